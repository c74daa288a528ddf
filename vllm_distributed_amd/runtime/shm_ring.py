"""ctypes binding of the native shared-memory broadcast ring + the object-level
``MessageQueue`` used by the multi-process executor (one writer, N local readers).

Objects are encoded with the framed-msgpack codec of the engine transport, so nothing is
unpickled from another process.
Reference: vllm/distributed/device_communicators/shm_broadcast.py (ShmRingBuffer :83,
MessageQueue :176, create_from_process_group :538).
"""
from __future__ import annotations

import ctypes
import os
import uuid
from dataclasses import dataclass
from typing import Any, Optional

from .. import envs
from ..logger import init_logger

logger = init_logger(__name__)
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[ctypes.CDLL] = None


def host_library() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libvda_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `python -m vllm_distributed_amd.ops.build`")
        lib = ctypes.CDLL(path)
        vp, u32, u64, dbl = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_double
        lib.vda_ring_create.restype, lib.vda_ring_create.argtypes = vp, [ctypes.c_char_p, u32, u64, u32]
        lib.vda_ring_attach.restype, lib.vda_ring_attach.argtypes = vp, [ctypes.c_char_p]
        lib.vda_ring_slot_bytes.restype, lib.vda_ring_slot_bytes.argtypes = u64, [vp]
        lib.vda_ring_readers_attached.restype, lib.vda_ring_readers_attached.argtypes = u32, [vp]
        lib.vda_ring_write.restype, lib.vda_ring_write.argtypes = ctypes.c_int, [vp, ctypes.c_char_p, u64, dbl]
        lib.vda_ring_next_size.restype, lib.vda_ring_next_size.argtypes = ctypes.c_int64, [vp, dbl]
        lib.vda_ring_read.restype, lib.vda_ring_read.argtypes = ctypes.c_int64, [vp, vp, u64, dbl]
        lib.vda_ring_close.restype, lib.vda_ring_close.argtypes = None, [vp, ctypes.c_int]
        _LIB = lib
    return _LIB


class RingClosed(RuntimeError):
    pass


@dataclass
class RingHandle:
    """What a reader needs to attach; travels to workers inside their start-up arguments."""
    name: str
    n_readers: int
    n_slots: int
    slot_bytes: int


class ShmRing:
    def __init__(self, handle: RingHandle, create: bool):
        self.handle = handle
        self.lib = host_library()
        self.is_writer = create
        nm = handle.name.encode()
        self.h = self.lib.vda_ring_create(nm, handle.n_slots, handle.slot_bytes, handle.n_readers) if create \
            else self.lib.vda_ring_attach(nm)
        if not self.h:
            raise OSError(f"cannot {'create' if create else 'attach'} shared-memory ring {handle.name}")
        self._buf = ctypes.create_string_buffer(64 * 1024)

    @classmethod
    def create(cls, n_readers: int, n_slots: int = 16, slot_bytes: int = 1 << 20) -> "ShmRing":
        return cls(RingHandle(f"/vda_ring_{os.getpid()}_{uuid.uuid4().hex[:12]}", n_readers, n_slots, slot_bytes), True)

    @staticmethod
    def _check(rc: int) -> None:
        if rc == -1:
            raise TimeoutError("shared-memory ring timed out")
        if rc == -2:
            raise RingClosed("shared-memory ring closed by the writer")
        if rc < 0:
            raise RuntimeError(f"shared-memory ring error {rc}")

    def write(self, data: bytes, timeout: Optional[float] = None) -> None:
        self._check(self.lib.vda_ring_write(self.h, data, len(data), -1.0 if timeout is None else timeout))

    def read(self, timeout: Optional[float] = None) -> bytes:
        t = -1.0 if timeout is None else timeout
        n = self.lib.vda_ring_next_size(self.h, t)
        self._check(n)
        if n > len(self._buf):
            self._buf = ctypes.create_string_buffer(max(n, 2 * len(self._buf)))
        got = self.lib.vda_ring_read(self.h, self._buf, len(self._buf), t)
        self._check(got)
        return self._buf.raw[:got] if got < 4096 else ctypes.string_at(self._buf, got)

    def readers_attached(self) -> int:
        return int(self.lib.vda_ring_readers_attached(self.h))

    def close(self) -> None:
        if self.h:
            self.lib.vda_ring_close(self.h, 1 if self.is_writer else 0)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MessageQueue:
    """Object broadcast from one writer to ``n_readers`` processes of the same node."""

    def __init__(self, ring: ShmRing):
        self.ring = ring
        from ..engine import serial
        self._enc, self._dec = serial.encode, serial.decode

    @classmethod
    def create(cls, n_readers: int, max_chunk_bytes: Optional[int] = None, max_chunks: int = 16) -> "MessageQueue":
        return cls(ShmRing.create(n_readers, max_chunks, max_chunk_bytes or envs.VLLM_MQ_MAX_CHUNK_BYTES_MB << 20))

    @classmethod
    def attach(cls, handle: RingHandle) -> "MessageQueue":
        return cls(ShmRing(handle, create=False))

    def export_handle(self) -> RingHandle:
        return self.ring.handle

    def wait_until_ready(self, timeout: float = 300.0) -> None:
        """Writer side: block until every reader has attached."""
        import time
        t0 = time.monotonic()
        while self.ring.readers_attached() < self.ring.handle.n_readers:
            if time.monotonic() - t0 > timeout:
                raise TimeoutError("workers did not attach to the RPC ring")
            time.sleep(0.002)

    def enqueue(self, obj: Any, timeout: Optional[float] = None) -> None:
        self.ring.write(self._enc(obj), timeout)

    def dequeue(self, timeout: Optional[float] = None) -> Any:
        return self._dec(self.ring.read(timeout))

    def broadcast_object(self, obj: Any = None) -> Any:
        if self.ring.is_writer:
            self.enqueue(obj)
            return obj
        return self.dequeue()

    @staticmethod
    def create_from_process_group(pg, max_chunk_bytes: int, max_chunks: int, writer_rank: int = 0) -> "MessageQueue":
        """Collective over ``pg`` (all ranks on one node): the writer creates the ring and the
        handle reaches the others through the group's object broadcast."""
        import torch.distributed as dist
        rank, world = dist.get_rank(pg), dist.get_world_size(pg)
        ranks = dist.get_process_group_ranks(pg)
        if rank == writer_rank:
            mq = MessageQueue.create(world - 1, max_chunk_bytes, max_chunks)
            h = mq.export_handle()
            box = [[h.name, h.n_readers, h.n_slots, h.slot_bytes]]
            dist.broadcast_object_list(box, src=ranks[writer_rank], group=pg)
        else:
            box = [None]
            dist.broadcast_object_list(box, src=ranks[writer_rank], group=pg)
            mq = MessageQueue.attach(RingHandle(*box[0]))
        dist.barrier(group=pg)
        return mq

    def close(self) -> None:
        self.ring.close()
