"""KV-cache change events for external routers: block stored / removed / all
cleared, batched per engine step and published to subscribers with a replay
buffer (reference: vllm/distributed/kv_events.py).

The reference publishes over ZMQ PUB; this build has its own framed-msgpack
socket transport (engine/transport.py), so the publisher is a small TCP/UDS
fan-out server with the same sequence-number + replay contract.
"""
from __future__ import annotations

import queue
import threading
import time
from collections import deque
from dataclasses import asdict, dataclass, field
from typing import Any, Optional


@dataclass
class KVCacheEvent:
    pass


@dataclass
class BlockStored(KVCacheEvent):
    block_hashes: list[int]
    parent_block_hash: Optional[int]
    token_ids: list[int]
    block_size: int
    lora_id: Optional[int] = None


@dataclass
class BlockRemoved(KVCacheEvent):
    block_hashes: list[int]


@dataclass
class AllBlocksCleared(KVCacheEvent):
    pass


@dataclass
class KVEventBatch:
    ts: float
    events: list[KVCacheEvent]
    data_parallel_rank: Optional[int] = None

    def to_wire(self) -> dict:
        return {"ts": self.ts, "dp_rank": self.data_parallel_rank,
                "events": [{"type": type(e).__name__, **asdict(e)} for e in self.events]}


class EventPublisher:
    def publish(self, batch: KVEventBatch) -> None:
        raise NotImplementedError

    def shutdown(self) -> None:
        pass


class NullEventPublisher(EventPublisher):
    def publish(self, batch: KVEventBatch) -> None:
        pass


class SocketEventPublisher(EventPublisher):
    """Background thread that pushes ``(seq, batch)`` frames to every connected
    subscriber and answers replay requests ``{"replay_from": seq}`` from a ring
    of the last ``buffer_steps`` batches."""

    def __init__(self, endpoint: str, buffer_steps: int = 10000, max_queue_size: int = 100000,
                 topic: str = "", data_parallel_rank: int = 0):
        from ..engine.transport import Listener
        self._dp_rank = data_parallel_rank
        self._topic = topic
        self._q: "queue.Queue[Optional[KVEventBatch]]" = queue.Queue(maxsize=max_queue_size)
        self._ring: deque[tuple[int, dict]] = deque(maxlen=buffer_steps)
        self._seq = 0
        self._listener = Listener(endpoint)
        self.endpoint = self._listener.address
        self._subs: list = []
        self._lock = threading.Lock()
        self._running = True
        self._t = threading.Thread(target=self._run, name="kv-events", daemon=True)
        self._t.start()

    def publish(self, batch: KVEventBatch) -> None:
        if not self._running:
            raise RuntimeError("publisher is shut down")
        batch.data_parallel_rank = self._dp_rank
        self._q.put(batch)

    def _accept_new(self) -> None:
        while True:
            conn = self._listener.accept(timeout=0)
            if conn is None:
                return
            with self._lock:
                self._subs.append(conn)

    def _run(self) -> None:
        while self._running or not self._q.empty():
            self._accept_new()
            # serve replay requests
            with self._lock:
                subs = list(self._subs)
            for c in subs:
                try:
                    while c.poll(0):
                        req = c.recv()
                        start = int(req.get("replay_from", 0))
                        for seq, payload in list(self._ring):
                            if seq >= start:
                                c.send({"topic": self._topic, "seq": seq, "batch": payload, "replay": True})
                except (EOFError, OSError):
                    self._drop(c)
            try:
                batch = self._q.get(timeout=0.05)
            except queue.Empty:
                continue
            if batch is None:
                break
            payload = batch.to_wire()
            seq = self._seq
            self._seq += 1
            self._ring.append((seq, payload))
            for c in subs:
                try:
                    c.send({"topic": self._topic, "seq": seq, "batch": payload})
                except (EOFError, OSError):
                    self._drop(c)

    def _drop(self, c) -> None:
        with self._lock:
            if c in self._subs:
                self._subs.remove(c)
        try:
            c.close()
        except OSError:
            pass

    def shutdown(self) -> None:
        self._running = False
        self._q.put(None)
        self._t.join(timeout=2)
        self._listener.close()
        for c in list(self._subs):
            self._drop(c)


def make_event_publisher(cfg, data_parallel_rank: int = 0) -> EventPublisher:
    if cfg is None or not cfg.enable_kv_cache_events or cfg.publisher == "null":
        return NullEventPublisher()
    if cfg.publisher in ("socket", "zmq"):
        return SocketEventPublisher(cfg.endpoint, cfg.buffer_steps, cfg.max_queue_size, cfg.topic,
                                    data_parallel_rank)
    raise ValueError(f"unknown KV event publisher {cfg.publisher!r}")
