"""Token exchange around expert layers when attention runs data-parallel.

xGMI is point-to-point with 7 links per GPU, and RCCL's all-gather / reduce-scatter use all of
them at once, so "dispatch" is ONE padded all-gather of every replica's tokens and "combine" ONE
reduce-scatter of the partial expert outputs -- two large collectives per layer instead of
per-peer sends.  Replicas pad to the largest token count of the step (known from the forward
context; all replicas enter every step together in lock-step mode).

``AllToAllManager`` additionally offers a true variable-split all-to-all (``all_to_all_single``)
for callers that route tokens to expert owners explicitly.
Reference: vllm/distributed/device_communicators/all2all.py (NaiveAll2AllManager :14-79,
pplx / DeepEP managers :82-), vllm/model_executor/layers/fused_moe/layer.py:1470-1494.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from ..forward_context import get_forward_context
from .state import GroupCoordinator, get_dp_group


@dataclass
class DispatchMeta:
    sizes: list[int]
    max_tokens: int
    rank: int


def dp_dispatch(hidden: torch.Tensor, router_logits: torch.Tensor):
    """[t_r, H] per replica -> [sum_r t_r, H] on every replica (plus the matching logits)."""
    g = get_dp_group()
    dp = get_forward_context().dp
    if dp is None:
        raise RuntimeError("data-parallel expert dispatch needs DPMetadata in the forward context")
    sizes, mx = dp.num_tokens_across_dp, dp.max_tokens
    H, E = hidden.shape[1], router_logits.shape[1]
    # one buffer, one collective: hidden and logits travel together
    pack = torch.zeros((mx, H + E), dtype=hidden.dtype, device=hidden.device)
    t = hidden.shape[0]
    pack[:t, :H] = hidden
    pack[:t, H:] = router_logits.to(hidden.dtype)
    full = g.all_gather(pack, dim=0).view(g.world_size, mx, H + E)
    if all(s == mx for s in sizes):
        flat = full.view(-1, H + E)
    else:
        flat = torch.cat([full[r, :s] for r, s in enumerate(sizes)], dim=0)
    return flat[:, :H].contiguous(), flat[:, H:].contiguous(), DispatchMeta(list(sizes), mx, g.rank_in_group)


def dp_combine(out: torch.Tensor, meta: DispatchMeta) -> torch.Tensor:
    """Partial outputs for all replicas' tokens -> summed outputs for this replica's tokens."""
    g = get_dp_group()
    H = out.shape[1]
    if all(s == meta.max_tokens for s in meta.sizes):
        padded = out
    else:
        padded = torch.zeros((g.world_size, meta.max_tokens, H), dtype=out.dtype, device=out.device)
        off = 0
        for r, s in enumerate(meta.sizes):
            padded[r, :s] = out[off:off + s]
            off += s
        padded = padded.view(-1, H)
    mine = g.reduce_scatter(padded, dim=0)
    return mine[: meta.sizes[meta.rank]]


class AllToAllManager:
    """Variable-split all-to-all over a group (expert-owner routing)."""

    def __init__(self, group: GroupCoordinator):
        self.group = group

    def exchange_counts(self, send_counts: torch.Tensor) -> torch.Tensor:
        recv = torch.empty_like(send_counts)
        self.group.all_to_all_single(recv, send_counts)
        return recv

    def dispatch(self, rows: torch.Tensor, send_counts: list[int], recv_counts: Optional[list[int]] = None
                 ) -> tuple[torch.Tensor, list[int]]:
        """``rows`` sorted by destination rank; returns the rows received and their per-source counts."""
        if recv_counts is None:
            sc = torch.tensor(send_counts, dtype=torch.int64, device=rows.device)
            recv_counts = self.exchange_counts(sc).tolist()
        out = torch.empty((sum(recv_counts), *rows.shape[1:]), dtype=rows.dtype, device=rows.device)
        self.group.all_to_all_single(out, rows, recv_counts, send_counts)
        return out, recv_counts

    def combine(self, rows: torch.Tensor, send_counts: list[int], recv_counts: list[int]) -> torch.Tensor:
        """Inverse of dispatch: what was received goes back to where it came from."""
        out = torch.empty((sum(send_counts), *rows.shape[1:]), dtype=rows.dtype, device=rows.device)
        self.group.all_to_all_single(out, rows, send_counts, recv_counts)
        return out
