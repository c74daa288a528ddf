"""Token exchange around expert layers when attention runs data-parallel.

xGMI is point-to-point with 7 links per GPU, and RCCL's all-gather / reduce-scatter use all of
them at once, so "dispatch" is ONE padded all-gather of every replica's tokens and "combine" ONE
reduce-scatter of the partial expert outputs -- two large collectives per layer instead of
per-peer sends.  Replicas pad to the largest token count of the step (known from the forward
context; all replicas enter every step together in lock-step mode).

``RowExchange`` is the all-to-all-v of rows (peer memory over xGMI on one node, torch.distributed
otherwise) and ``ExpertDispatcher`` the expert-parallel dispatch / combine built on it: tokens go
to the ranks that own their experts instead of every rank seeing every token.
Reference: vllm/distributed/device_communicators/all2all.py (NaiveAll2AllManager :14-79,
pplx / DeepEP managers :82-), vllm/model_executor/layers/fused_moe/layer.py:1470-1494.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from .. import envs
from ..forward_context import get_forward_context
from ..logger import init_logger
from .state import GroupCoordinator, get_dp_group

logger = init_logger(__name__)


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


class RowExchange:
    """All-to-all-v of rows over a group with fixed result slots: ``recv[src * cap + j]`` is the j-th row
    rank ``src`` sent here.  On GPUs of one node the rows move through peer memory over xGMI
    (``PeerAllToAll``: no host sync, hipGraph-capturable); otherwise (CPU tensors, several nodes, the
    one-device rehearsal when the kernel is unavailable) the same contract is served by padded
    ``all_to_all_single`` calls of torch.distributed."""

    def __init__(self, group: GroupCoordinator, cap: int, row_bytes: int):
        self.group = group
        self.world, self.rank = group.world_size, group.rank_in_group
        self.cap, self.row_bytes = int(cap), int(row_bytes)
        self.peer = None
        if group.device.type == "cuda" and self.world > 1 and not envs.VDA_DISABLE_PEER_ALL2ALL:
            from .custom_allreduce import PeerAllToAll
            try:
                self.peer = PeerAllToAll.maybe_create(group.cpu_group, group.device, self.cap, self.row_bytes)
            except Exception as e:   # noqa: BLE001
                logger.warning("peer-memory all-to-all unavailable for group %s: %s", group.name, e)

    # ---- host / torch.distributed form of the same contract -------------------------------------------
    def _host_exchange(self, send: torch.Tensor, counts: torch.Tensor, meta: Optional[torch.Tensor]):
        """send [world, cap, W] padded regions by destination, counts [world]."""
        g = self.group
        recv = torch.empty_like(send)
        g.all_to_all_single(recv.view(-1, send.shape[-1]), send.view(-1, send.shape[-1]))
        rc = torch.empty_like(counts)
        g.all_to_all_single(rc, counts)
        rmeta = None
        if meta is not None:
            rmeta = torch.empty_like(meta)
            g.all_to_all_single(rmeta.view(-1), meta.view(-1))
        return recv, rmeta, rc

    def exchange(self, x: torch.Tensor, sorted_ids: torch.Tensor, offsets: torch.Tensor, k_div: int = 1,
                 meta: Optional[torch.Tensor] = None, cap: int = 0):
        """Row ``sorted_ids[i] // k_div`` of ``x`` goes to rank d for offsets[d] <= i < offsets[d + 1];
        ``meta[sorted_ids[i]]`` (int32) travels with it.  Returns (recv [world * cap, W], recv_meta
        [world * cap] or None, recv_counts int32 [world])."""
        cap = int(cap) or self.cap
        if self.peer is not None and x.is_cuda:
            return self.peer.exchange(x, sorted_ids, offsets, k_div, meta, cap)
        W = x.shape[1]
        send = torch.zeros((self.world, cap, W), dtype=x.dtype, device=x.device)
        smeta = torch.zeros((self.world, cap), dtype=torch.int32, device=x.device) if meta is not None else None
        off = offsets.tolist()
        for d in range(self.world):
            n = min(off[d + 1] - off[d], cap)
            if n > 0:
                idx = sorted_ids[off[d]:off[d] + n].long()
                send[d, :n] = x[idx // k_div]
                if smeta is not None:
                    smeta[d, :n] = meta.view(-1)[idx]
        counts = (offsets[1:] - offsets[:-1]).clamp(max=cap).to(torch.int32)
        recv, rmeta, rc = self._host_exchange(send, counts, smeta)
        return recv.view(-1, W), (rmeta.view(-1) if rmeta is not None else None), rc

    def exchange_regions(self, y: torch.Tensor, counts: torch.Tensor, cap: int = 0):
        """``y`` [world * cap, W] already grouped by destination with counts[d] valid rows each."""
        cap = int(cap) or self.cap
        if self.peer is not None and y.is_cuda:
            return self.peer.exchange_regions(y, counts, cap)
        W = y.shape[1]
        recv, _, rc = self._host_exchange(y[: self.world * cap].reshape(self.world, cap, W).contiguous(),
                                          counts.to(torch.int32), None)
        return recv.view(-1, W), rc

    def all_gather_rows(self, x: torch.Tensor) -> torch.Tensor:
        """[n, W] on every rank (same n) -> [world * n, W]: every rank pulls every peer's rows at once, the
        direct all-gather of a fully connected xGMI node, through the same kernel (and as capturable)."""
        n, W = x.shape
        cap = -(-max(n, 1) // 16) * 16
        if self.peer is None or not x.is_cuda or cap > self.cap:
            return self.group.all_gather(x, dim=0)
        ids = torch.arange(n, dtype=torch.int32, device=x.device).repeat(self.world)
        off = torch.arange(self.world + 1, dtype=torch.int32, device=x.device) * n
        recv, _, _ = self.peer.exchange(x, ids, off, 1, None, cap)
        return recv.view(self.world, cap, W)[:, :n].reshape(self.world * n, W)

    def check_error(self) -> int:
        return self.peer.check_error() if self.peer is not None else 0

    def close(self) -> None:
        if self.peer is not None:
            self.peer.close()
            self.peer = None


@dataclass
class EpState:
    dest: torch.Tensor          # [P] destination rank of every (token, slot) pair, -1 = padding
    slot: torch.Tensor          # [P] position of pair p inside its destination's region (-1 = padding)
    offsets: torch.Tensor       # [world + 1]
    recv_counts: torch.Tensor   # [world]
    cap: int


class ExpertDispatcher:
    """Expert-parallel token routing: every (token, slot) pair travels to the rank that owns its expert,
    the owner runs its experts on what it received, and the results travel back to be weighted and
    summed at the source.  Per link this moves T * k * H / world bytes each way instead of the
    all-gather + all-reduce of replicate-and-mask.
    Reference: vllm/model_executor/layers/fused_moe/layer.py:1439-1484 (dispatch / combine around the
    expert computation), vllm/distributed/device_communicators/all2all.py."""

    def __init__(self, group: GroupCoordinator, experts_per_rank: int, top_k: int, row_bytes: int, max_tokens: int):
        self.group = group
        self.world, self.rank = group.world_size, group.rank_in_group
        self.epr, self.top_k = experts_per_rank, top_k
        cap = -(-max(max_tokens, 1) * top_k // 16) * 16
        self.rx = RowExchange(group, cap, row_bytes)

    def dispatch(self, x: torch.Tensor, topk_ids: torch.Tensor, cap_tokens: int = 0):
        """x [T, H]; topk_ids int32 [T, k] global (physical) expert ids, negative = padding.  ``cap_tokens``: the
        largest T of any rank in this call when the ranks hold different numbers of tokens (data-parallel
        attention); every rank must pass the same value.  Returns (rows received [world * cap, H], their local
        expert ids int32 [world * cap, 1] with -1 on empty slots, state for ``combine``).
        On the GPU: one bookkeeping kernel (ops.ep_prep) and the exchange, which hands out local expert ids itself."""
        from .. import ops
        T, k = topk_ids.shape
        cap = min(self.rx.cap, -(-max(int(cap_tokens) or T, 1) * k // 16) * 16)
        ids, dest, sorted_ids, offsets, slot = ops.ep_prep(topk_ids.reshape(-1), self.world, self.epr)
        if self.rx.peer is not None and x.is_cuda:
            recv, local, rc = self.rx.peer.exchange(x, sorted_ids, offsets, k, ids, cap, meta_sub=self.rank * self.epr,
                                                    fill_meta=True)
            return recv, local.view(-1, 1), EpState(dest, slot, offsets, rc, cap)
        recv, rmeta, rc = self.rx.exchange(x, sorted_ids, offsets, k, ids.contiguous(), cap)
        pos = torch.arange(cap, device=x.device, dtype=torch.int32).unsqueeze(0)
        local = torch.where(pos < rc.unsqueeze(1), rmeta.view(self.world, cap) - self.rank * self.epr,
                            torch.full((), -1, dtype=torch.int32, device=x.device))
        return recv, local.reshape(-1, 1).contiguous(), EpState(dest, slot, offsets, rc, cap)

    def combine(self, y: torch.Tensor, topk_w: torch.Tensor, st: EpState) -> torch.Tensor:
        """y [world * cap, H]: outputs of the rows received by ``dispatch`` (same slots).  Returns
        out [T, H] = sum_j topk_w[t, j] * y(pair t, j), fp32 sums.  On the GPU one kernel: the rows return through
        peer memory and are weighted and summed where they land."""
        T, k = topk_w.shape
        if self.rx.peer is not None and y.is_cuda and y.dtype in (torch.bfloat16, torch.float16):
            return self.rx.peer.combine(y, st.recv_counts, st.dest, st.slot, topk_w, st.cap)
        ret, _ = self.rx.exchange_regions(y, st.recv_counts, st.cap)
        d = st.dest.long()
        valid = d >= 0
        pos = (d.clamp(min=0) * st.cap + st.slot.long().clamp(min=0)).clamp(max=ret.shape[0] - 1)
        w = torch.where(valid.view(T, k), topk_w.float(), torch.zeros((), device=y.device))
        rows = ret[pos].view(T, k, -1).float()
        rows = torch.where(valid.view(T, k, 1), rows, torch.zeros((), device=y.device))   # unwritten slots may hold NaN
        return (rows * w.unsqueeze(-1)).sum(dim=1).to(y.dtype)


_DISPATCHERS: dict = {}


def get_expert_dispatcher(group: GroupCoordinator, experts_per_rank: int, top_k: int, row_bytes: int,
                          max_tokens: int) -> ExpertDispatcher:
    """One dispatcher (one arena) per (group, shape): every MoE layer of a model shares it, calls are
    ordered by the stream."""
    key = (group.name, tuple(group.ranks), experts_per_rank, top_k, row_bytes, max_tokens)
    d = _DISPATCHERS.get(key)
    if d is None:
        d = _DISPATCHERS[key] = ExpertDispatcher(group, experts_per_rank, top_k, row_bytes, max_tokens)
    return d


def check_dispatcher_errors() -> None:
    for d in _DISPATCHERS.values():
        e = d.rx.check_error()
        if e:
            raise RuntimeError(f"peer-memory all-to-all of group {d.group.name} timed out waiting for rank {e - 1}; "
                               "the step's results are invalid")


def close_dispatchers() -> None:
    for d in _DISPATCHERS.values():
        d.rx.close()
    _DISPATCHERS.clear()
