"""Token-parallel (TKNP) placement: which rank owns the attention + KV cache
of each request.

What the reference does (vllm/v1/core/sched/scheduler.py:55-252,387-411): every
rank runs the same scheduler; new requests are split into contiguous runs whose
lengths are proportional to each rank's free KV blocks, and those free-block
counts are all-gathered on the *device* process group at every ``schedule()``.

What is different here:
  * No collective on the scheduling path.  All ranks see the same requests in
    the same order, so each rank can keep an exact model of every rank's block
    usage ("virtual pools") from the per-rank capacities exchanged once at
    start-up.  Admission and preemption decisions are taken on the virtual
    pools and are therefore identical on every rank by construction.
  * Placement is token-aware: a request goes to the rank with the most free
    blocks *after* charging the blocks its prompt needs, which balances ragged
    prompt lengths.  The reference's block-wise policy (contiguous runs,
    largest-remainder rounding) is kept as ``policy="blockwise"``.
"""
from __future__ import annotations

import math
from typing import Iterable, Optional, Sequence

import numpy as np

from ..parallel.utils import cdiv
from .output import TokenParallelAllocation


class TokenParallelScheduler:
    def __init__(self, world_size: int, rank: int, block_size: int,
                 capacities: Sequence[int], policy: str = "balanced"):
        if len(capacities) != world_size:
            raise ValueError("need one KV capacity per token-parallel rank")
        if policy not in ("balanced", "blockwise"):
            raise ValueError("policy must be balanced or blockwise")
        self.world_size = world_size
        self.rank = rank
        self.block_size = block_size
        self.policy = policy
        # block 0 of every pool is the null block
        self.capacity = np.asarray([c - 1 for c in capacities], dtype=np.int64)
        self.used = np.zeros(world_size, dtype=np.int64)
        self.req_to_rank: dict[str, int] = {}
        self.req_blocks: dict[str, int] = {}
        self.reqs_per_rank = np.zeros(world_size, dtype=np.int64)

    # placement ------------------------------------------------------------
    def free_blocks(self) -> np.ndarray:
        return self.capacity - self.used

    def assign_ranks(self, waiting: Iterable, free_blocks_per_rank: Optional[Sequence[int]] = None) -> None:
        new = [r for r in waiting if r.request_id not in self.req_to_rank]
        if not new:
            return
        free = np.asarray(free_blocks_per_rank, dtype=np.float64) if free_blocks_per_rank is not None \
            else self.free_blocks().astype(np.float64)
        if self.policy == "blockwise":
            counts = self._proportional_counts(len(new), free)
            it = iter(new)
            for rank, c in enumerate(counts):
                for _ in range(int(c)):
                    self._place(next(it), rank)
            return
        # balanced: greedy by projected free blocks, charging each prompt as it is placed
        projected = free.copy()
        for req in new:
            need = cdiv(req.num_tokens, self.block_size)
            rank = int(np.argmax(projected))
            projected[rank] -= need
            self._place(req, rank)

    def _proportional_counts(self, n: int, free: np.ndarray) -> np.ndarray:
        total = float(free.sum())
        if total <= 0.0:
            per = max(1, math.ceil(n / self.world_size))
            counts = np.zeros(self.world_size, dtype=np.int64)
            left = n
            for r in range(self.world_size):
                counts[r] = min(per, left)
                left -= counts[r]
            return counts
        ideal = free * (n / total)
        counts = np.floor(ideal).astype(np.int64)
        rest = n - int(counts.sum())
        if rest > 0:
            order = np.argsort(-(ideal - counts), kind="stable")
            counts[order[:rest]] += 1
        return counts

    def _place(self, req, rank: int) -> None:
        self.req_to_rank[req.request_id] = rank
        self.reqs_per_rank[rank] += 1

    # virtual pools --------------------------------------------------------
    def rank_of(self, request_id: str) -> int:
        return self.req_to_rank[request_id]

    def is_mine(self, request) -> bool:
        return self.req_to_rank.get(request.request_id) == self.rank

    def blocks_needed(self, request_id: str, total_tokens: int) -> int:
        return max(cdiv(total_tokens, self.block_size) - self.req_blocks.get(request_id, 0), 0)

    def can_reserve(self, request_id: str, total_tokens: int) -> bool:
        r = self.req_to_rank[request_id]
        return self.used[r] + self.blocks_needed(request_id, total_tokens) <= self.capacity[r]

    def reserve(self, request_id: str, total_tokens: int) -> None:
        r = self.req_to_rank[request_id]
        n = self.blocks_needed(request_id, total_tokens)
        self.used[r] += n
        self.req_blocks[request_id] = self.req_blocks.get(request_id, 0) + n

    def release(self, request_id: str) -> None:
        """Give the blocks back (preemption or finish) but keep the placement."""
        n = self.req_blocks.pop(request_id, 0)
        if n:
            self.used[self.req_to_rank[request_id]] -= n

    def free_request(self, request) -> None:
        rid = request.request_id
        self.release(rid)
        r = self.req_to_rank.pop(rid, None)
        if r is not None:
            self.reqs_per_rank[r] -= 1

    # per-step summary -----------------------------------------------------
    def allocation(self, num_scheduled_tokens: dict[str, int]) -> TokenParallelAllocation:
        per_rank = [0] * self.world_size
        for rid, n in num_scheduled_tokens.items():
            per_rank[self.req_to_rank[rid]] += n
        live = {rid: self.req_to_rank[rid] for rid in num_scheduled_tokens}
        return TokenParallelAllocation(live, per_rank, [int(x) for x in self.reqs_per_rank])
