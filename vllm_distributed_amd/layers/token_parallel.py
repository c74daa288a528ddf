"""Token parallelism (TKNP): attention and its KV cache are sharded *by request* over the
ranks of the TKNP group, while only the group's root holds the dense weights.

Per attention layer on the root:
    qkv = W_qkv h                                  (all T rows of the step)
    rows are permuted so that every rank's rows are contiguous (explicit permutation: under
    continuous batching one rank's requests are not contiguous in the batch), then ONE
    all-to-all-v sends rank r its rows -- on MI355X the root has a private xGMI link to each of up
    to 7 peers, and the peer-memory exchange (parallel/all2all.py::RowExchange) lets every peer pull
    its rows at link rate with device-side counts, so the step is hipGraph-capturable;
    every rank runs RoPE + KV write + paged attention for its own requests against its own pool;
    a second all-to-all returns the attention rows to the root, which undoes the permutation and
    applies W_o (+ the TP all-reduce inside the root's TP group).
Non-root ranks execute only the attention of each layer (``TokenParallelWorkerModel``): no
embedding, norms, MLP or weights, so their whole HBM is KV cache.

Reference: vllm/model_executor/layers/token_parallel_linear.py (TokenParallelQKVLinear :39-242,
TokenParallelRowLinear :244-381, init_tknp_layer :444-554); the reference scatters contiguous row
ranges and runs stand-in dense layers on non-root ranks, both replaced here.
"""
from __future__ import annotations

from typing import Optional

import torch

from ..forward_context import TokenParallelMetadata, get_forward_context
from ..parallel.state import get_tknp_group


def tknp_metadata() -> Optional[TokenParallelMetadata]:
    from ..forward_context import has_forward_context
    return get_forward_context().tknp if has_forward_context() else None


def scatter_rows(x: Optional[torch.Tensor], md: TokenParallelMetadata, width: int, dtype: torch.dtype,
                 device: torch.device) -> torch.Tensor:
    """Root passes its [T, width] rows in batch order; every rank gets its own [T_local, width] rows."""
    if md.exchange is not None:
        # root: rows grouped by owner through the index list; the others contribute nothing.  Region 0 of
        # what arrives (the root's rows for this rank, in batch order) is the local batch.
        src = x if md.is_root else md.dummy[:, :width]
        recv, _, _ = md.exchange.exchange(src, md.sorted_ids, md.offsets, 1, None, md.cap)
        return recv[: md.num_local_tokens]
    g = get_tknp_group()
    n = g.world_size
    out = torch.empty((md.num_local_tokens, width), dtype=dtype, device=device)
    out_splits = [md.num_local_tokens if r == 0 else 0 for r in range(n)]
    if md.is_root:
        send = x.index_select(0, md.perm) if md.perm is not None else x
        in_splits = list(md.tokens_per_rank)
    else:
        send = torch.empty((0, width), dtype=dtype, device=device)
        in_splits = [0] * n
    g.all_to_all_single(out, send.contiguous(), out_splits, in_splits)
    return out


def gather_rows(x_local: torch.Tensor, md: TokenParallelMetadata) -> Optional[torch.Tensor]:
    """Inverse of scatter_rows: the root gets all rows back in batch order, the others None."""
    if md.exchange is not None:
        src = x_local if x_local.shape[0] else md.dummy[:, : x_local.shape[1]]
        recv, _, _ = md.exchange.exchange(src, md.iota, md.gather_offsets, 1, None, md.cap)
        return recv.index_select(0, md.back) if md.is_root else None
    g = get_tknp_group()
    n = g.world_size
    width = x_local.shape[1]
    in_splits = [x_local.shape[0] if r == 0 else 0 for r in range(n)]
    if md.is_root:
        out = torch.empty((md.num_total_tokens, width), dtype=x_local.dtype, device=x_local.device)
        out_splits = list(md.tokens_per_rank)
    else:
        out = torch.empty((0, width), dtype=x_local.dtype, device=x_local.device)
        out_splits = [0] * n
    g.all_to_all_single(out, x_local.contiguous(), out_splits, in_splits)
    if not md.is_root:
        return None
    return out.index_select(0, md.inv_perm) if md.inv_perm is not None else out


def token_parallel_attention(attn, rotary, qkv: Optional[torch.Tensor], positions: torch.Tensor,
                             md: TokenParallelMetadata, qkv_width: int, out_width: int, dtype: torch.dtype,
                             device: torch.device) -> Optional[torch.Tensor]:
    """scatter -> local attention -> gather.  ``positions`` are the LOCAL tokens' positions."""
    local = scatter_rows(qkv, md, qkv_width, dtype, device)
    if md.num_local_tokens > 0:
        o = attn(local, positions[: md.num_local_tokens], rotary)
    else:
        o = torch.empty((0, out_width), dtype=dtype, device=device)
    return gather_rows(o.reshape(o.shape[0], out_width), md)
