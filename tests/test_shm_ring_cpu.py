"""Native shared-memory broadcast ring: ordering, multi-slot messages, back-pressure,
several reader processes, close wake-up."""
import multiprocessing as mp
import os
import time

import pytest

from vllm_distributed_amd.runtime.shm_ring import MessageQueue, RingClosed, RingHandle, ShmRing


def _reader(handle, n_msgs, q):
    mq = MessageQueue.attach(RingHandle(*handle))
    got = []
    try:
        for _ in range(n_msgs):
            got.append(mq.dequeue(timeout=30))
        q.put(("ok", [(g["i"], len(g["blob"])) for g in got]))
        try:
            mq.dequeue(timeout=30)
            q.put(("bad", "expected close"))
        except RingClosed:
            q.put(("closed", None))
    except Exception as e:   # noqa: BLE001
        q.put(("err", repr(e)))


def test_ring_same_process_roundtrip():
    w = ShmRing.create(n_readers=1, n_slots=4, slot_bytes=256)
    r = ShmRing(w.handle, create=False)
    for i in range(10):                    # more messages than slots: slots are reused
        msg = bytes([i]) * (i * 100 + 1)   # some span several slots
        if len(msg) <= 4 * 256:
            w.write(msg)
            assert r.read(timeout=5) == msg
    with pytest.raises(TimeoutError):
        r.read(timeout=0.05)
    # back-pressure: the ring holds 4 unread slots, the 5th write must time out
    for i in range(4):
        w.write(b"x")
    with pytest.raises(TimeoutError):
        w.write(b"y", timeout=0.05)
    r.close()
    w.close()


def test_message_queue_three_reader_processes():
    ctx = mp.get_context("spawn")
    mq = MessageQueue.create(n_readers=3, max_chunk_bytes=4096, max_chunks=4)
    h = mq.export_handle()
    q = ctx.Queue()
    n = 40
    procs = [ctx.Process(target=_reader, args=((h.name, h.n_readers, h.n_slots, h.slot_bytes), n, q)) for _ in range(3)]
    for p in procs:
        p.start()
    mq.wait_until_ready(timeout=60)
    sizes = [(i * 977) % 20000 for i in range(n)]      # up to ~5 slots: longer than the ring is legal too
    for i, sz in enumerate(sizes):
        mq.enqueue({"i": i, "blob": os.urandom(sz)}, timeout=30)
    res = [q.get(timeout=60) for _ in range(3)]
    assert all(r[0] == "ok" for r in res), res
    assert all(r[1] == list(enumerate(sizes)) for r in res)
    time.sleep(0.1)
    mq.close()
    res = [q.get(timeout=60) for _ in range(3)]
    assert all(r[0] == "closed" for r in res), res
    for p in procs:
        p.join(10)
