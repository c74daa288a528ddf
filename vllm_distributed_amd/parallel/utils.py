"""Small helpers shared by the parallel layers
(reference: vllm/distributed/utils.py:41-135)."""
from __future__ import annotations

import socket
from typing import Sequence

import torch

from .. import envs


def cdiv(a: int, b: int) -> int:
    return -(-a // b)


def ensure_divisible(a: int, b: int) -> int:
    if a % b:
        raise ValueError(f"{a} is not divisible by {b}")
    return a // b


def split_last_dim(t: torch.Tensor, parts: int, contiguous: bool = False) -> Sequence[torch.Tensor]:
    size = ensure_divisible(t.shape[-1], parts)
    out = torch.split(t, size, dim=-1)
    return tuple(x.contiguous() for x in out) if contiguous else out


def get_pp_indices(num_layers: int, pp_rank: int, pp_size: int) -> tuple[int, int]:
    """Half-open layer range ``[start, end)`` owned by a pipeline stage.

    ``VLLM_PP_LAYER_PARTITION="a,b,c"`` gives explicit per-stage counts.  The
    default spreads the remainder over the middle stages: the first stage also
    runs the embedding and the last the LM head, so they get the fewest layers.
    """
    spec = envs.VLLM_PP_LAYER_PARTITION
    if spec:
        try:
            counts = [int(x) for x in spec.split(",")]
        except ValueError as e:
            raise ValueError(f"bad VLLM_PP_LAYER_PARTITION {spec!r}") from e
        if len(counts) != pp_size or sum(counts) != num_layers:
            raise ValueError(f"VLLM_PP_LAYER_PARTITION {spec!r} must list {pp_size} counts "
                             f"summing to {num_layers}")
    else:
        base, rem = divmod(num_layers, pp_size)
        counts = [base] * pp_size
        # hand the remainder to stages 1.. (skipping stage 0), wrapping if needed
        order = list(range(1, pp_size - 1)) + ([pp_size - 1] if pp_size > 1 else []) + [0]
        for i in range(rem):
            counts[order[i % len(order)]] += 1
    start = sum(counts[:pp_rank])
    return start, start + counts[pp_rank]


def get_open_port() -> int:
    port = envs.VLLM_PORT
    if port is not None:
        while True:
            try:
                with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
                    s.bind(("", port))
                    return port
            except OSError:
                port += 1
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("", 0))
        return s.getsockname()[1]


def get_ip() -> str:
    return envs.VLLM_HOST_IP or "127.0.0.1"


def get_distributed_init_method(ip: str, port: int) -> str:
    return f"tcp://[{ip}]:{port}" if ":" in ip else f"tcp://{ip}:{port}"


class StatelessProcessGroup:
    """A process group that leaves ``torch.distributed``'s global state alone: ranks meet on a TCPStore
    and exchange msgpack-encoded objects through it.  Used for control-plane agreement between
    processes that are not part of the engine's own world (data-parallel replicas, prefill/decode
    pairs) and to hand a unique id to a dedicated RCCL communicator (``RcclCommunicator`` accepts it).
    Keys are consumed in order, so broadcasts and gathers from different call sites cannot mix.
    Reference: vllm/distributed/utils.py:137-340 (StatelessProcessGroup)."""

    def __init__(self, rank: int, world_size: int, store, data_expiration_seconds: int = 3600):
        import time as _time
        self.rank, self.world_size, self.store = rank, world_size, store
        self.expire = data_expiration_seconds
        self._time = _time
        self.send_dst_counter = {i: 0 for i in range(world_size)}
        self.recv_src_counter = {i: 0 for i in range(world_size)}
        self.broadcast_send_counter = 0
        self.broadcast_recv_src_counter = {i: 0 for i in range(world_size)}
        self._sent: list[tuple[str, float]] = []

    @classmethod
    def create(cls, host: str, port: int, rank: int, world_size: int, data_expiration_seconds: int = 3600,
               store_timeout: int = 300) -> "StatelessProcessGroup":
        import datetime

        from torch.distributed import TCPStore
        store = TCPStore(host_name=host, port=port, world_size=world_size, is_master=(rank == 0),
                         timeout=datetime.timedelta(seconds=store_timeout))
        return cls(rank, world_size, store, data_expiration_seconds)

    @staticmethod
    def _enc(obj) -> bytes:
        from ..engine.serial import decode as _d, encode as _e  # noqa: F401  (registered wire types)
        return _e(obj)

    @staticmethod
    def _dec(raw: bytes):
        from ..engine.serial import decode
        return decode(raw)

    def _expire(self) -> None:
        now = self._time.time()
        while self._sent and now - self._sent[0][1] > self.expire:
            key, _ = self._sent.pop(0)
            try:
                self.store.delete_key(key)
            except Exception:
                pass

    def send_obj(self, obj, dst: int) -> None:
        self._expire()
        key = f"send_to/{dst}/{self.send_dst_counter[dst]}"
        self.store.set(key, self._enc(obj))
        self.send_dst_counter[dst] += 1
        self._sent.append((key, self._time.time()))

    def recv_obj(self, src: int):
        obj = self._dec(self.store.get(f"send_to/{self.rank}/{self.recv_src_counter[src]}"))
        self.recv_src_counter[src] += 1
        return obj

    def broadcast_obj(self, obj, src: int):
        if self.rank == src:
            self._expire()
            key = f"broadcast_from/{src}/{self.broadcast_send_counter}"
            self.store.set(key, self._enc(obj))
            self.broadcast_send_counter += 1
            self._sent.append((key, self._time.time()))
            return obj
        key = f"broadcast_from/{src}/{self.broadcast_recv_src_counter[src]}"
        self.broadcast_recv_src_counter[src] += 1
        return self._dec(self.store.get(key))

    def all_gather_obj(self, obj) -> list:
        out = []
        for i in range(self.world_size):
            out.append(self.broadcast_obj(obj if i == self.rank else None, src=i))
        return out

    def barrier(self) -> None:
        """Every rank has arrived when this returns.  The rank that hosts the store (rank 0) additionally waits until
        every rank has READ what the barrier published: it may exit right after a final barrier, and the others'
        pending reads would find the store's socket closed."""
        self.all_gather_obj(self.rank)
        n = getattr(self, "_barriers", 0)
        self._barriers = n + 1
        key = f"barrier_read/{n}"
        arrived = self.store.add(key, 1)
        if self.rank == 0:
            deadline = self._time.time() + 60.0
            while arrived < self.world_size and self._time.time() < deadline:
                self._time.sleep(0.002)
                arrived = self.store.add(key, 0)
