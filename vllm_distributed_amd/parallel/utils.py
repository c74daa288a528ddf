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
