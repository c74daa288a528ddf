"""Per-forward global context: attention metadata, token-parallel routing,
DP metadata, and the optional per-batch-size timing log.

Reference: vllm/forward_context.py:28-213 (TokenParallelMetadata :28-48,
DPMetadata :57-70, batch-size timing :177-211).
"""
from __future__ import annotations

import contextlib
import time
from collections import defaultdict
from dataclasses import dataclass, field
from typing import Any, Optional

import torch

from . import envs
from .logger import init_logger

logger = init_logger(__name__)


@dataclass
class AttentionMetadata:
    """Batch layout: decode sequences first (one or a few query tokens each, split-KV
    kernel), then prefill / chunked-prefill sequences (tiled kernel).  All tensors
    live on the device and have fixed addresses when the step is graph-captured."""
    num_actual_tokens: int
    num_decode_seqs: int
    num_decode_tokens: int
    num_prefill_seqs: int
    num_prefill_tokens: int
    slot_mapping: torch.Tensor          # int64 [T]
    block_table: torch.Tensor           # int32 [S, max_blocks]
    seq_lens: torch.Tensor              # int32 [S]
    query_start_loc: torch.Tensor       # int32 [S+1]
    # prefill part, rebased so its first token is 0 (None when no prefill)
    prefill_query_start_loc: Optional[torch.Tensor] = None
    max_query_len: int = 1
    max_seq_len: int = 0
    max_decode_query_len: int = 1
    decode_num_splits: int = 1
    decode_workspace: Optional[tuple] = None
    # hybrid attention + state-space models: int32 [S, 1], the state slot of every sequence (None: the
    # slot is column 0 of the block table, as in attention-free models)
    state_table: Optional[torch.Tensor] = None
    # cascade attention (eager steps whose sequences all start with the same cached blocks): the first
    # ``cascade_prefix_len`` keys are attended once by all query tokens, the rest per sequence
    cascade_prefix_len: int = 0
    cascade_seq_len: Optional[torch.Tensor] = None        # int32 [1] = cascade_prefix_len
    cascade_query_start_loc: Optional[torch.Tensor] = None   # int32 [2] = (0, T)
    cascade_suffix_seq_lens: Optional[torch.Tensor] = None   # int32 [S] = seq_lens - cascade_prefix_len
    # chunked local attention (layers whose tokens only see their own chunk of ``local_chunk`` positions):
    # every (sequence, chunk) pair touched by the step is a virtual sequence whose block table starts at the
    # chunk's first block.  Decode-form sequences map to one virtual sequence each, in the same order.
    local_block_table: Optional[torch.Tensor] = None        # int32 [S_virtual, chunk / block_size]
    local_seq_lens: Optional[torch.Tensor] = None           # int32 [S_virtual]: keys inside the chunk
    local_query_start_loc: Optional[torch.Tensor] = None    # int32 [S_virtual + 1]
    local_prefill_query_start_loc: Optional[torch.Tensor] = None
    local_num_prefill_seqs: int = 0
    # dual-chunk attention: per virtual sequence (sequence, chunk c) also the keys of chunk c - 1 (table that starts
    # at that chunk's first block) and of chunks 0 .. c - 2 (the owner's own table), zero keys where there is none
    dca_succ_block_table: Optional[torch.Tensor] = None     # int32 [S_virtual, chunk / block_size]
    dca_succ_lens: Optional[torch.Tensor] = None            # int32 [S_virtual]: chunk length or 0
    dca_inter_block_table: Optional[torch.Tensor] = None    # int32 [S_virtual, blocks of the longest sequence]
    dca_inter_lens: Optional[torch.Tensor] = None           # int32 [S_virtual]: (c - 1) * chunk length or 0
    dca_token_scale: Optional[torch.Tensor] = None          # model dtype [T, 1]: length-dependent q / k factor
    dca_passes: int = 3                                      # 1: no virtual sequence has older chunks (host knows)
    # host copies for connectors / debugging
    seq_lens_cpu: Any = None
    query_start_loc_cpu: Any = None


@dataclass
class TokenParallelMetadata:
    """Which rows of the root's token batch go to which token-parallel rank.
    ``perm`` sorts the batch rows by owning rank (explicit permutation: rows of
    one rank need not be contiguous under continuous batching)."""
    tokens_per_rank: list[int]
    perm: Optional[torch.Tensor] = None        # int64 [T_total] on root: rows grouped by rank
    inv_perm: Optional[torch.Tensor] = None    # inverse of perm
    num_local_tokens: int = 0
    num_total_tokens: int = 0
    is_root: bool = True
    # peer-memory form (parallel/all2all.py::RowExchange): every index lives in persistent device buffers
    # the runner refills each step, so the same launches serve eager steps and captured graphs
    exchange: object = None                    # RowExchange of the token-parallel group, None = torch.distributed
    sorted_ids: Optional[torch.Tensor] = None  # int32, root: batch rows grouped by owning rank
    offsets: Optional[torch.Tensor] = None     # int32 [world + 1], root: row range per rank (zeros elsewhere)
    gather_offsets: Optional[torch.Tensor] = None   # int32 [world + 1]: this rank's local rows all go to the root
    iota: Optional[torch.Tensor] = None        # int32 0, 1, 2, ...
    back: Optional[torch.Tensor] = None        # int64 [T], root: slot (owner * cap + local index) of batch row i
    cap: int = 0                               # rows per (source, destination) pair of this step's exchanges
    dummy: Optional[torch.Tensor] = None       # one row of zeros (what a rank with nothing to send passes)


@dataclass
class DPMetadata:
    num_tokens_across_dp: list[int]

    @property
    def max_tokens(self) -> int:
        return max(self.num_tokens_across_dp)


@dataclass
class MicroBatch:
    """A run of whole sequences of the step's batch: token rows [start, end) and their own metadata
    (query offsets rebased to the first row).  Two of them let TP collectives of one overlap with the
    GEMMs of the other."""
    start: int
    end: int
    attn_metadata: AttentionMetadata


@dataclass
class ForwardContext:
    attn_metadata: Optional[AttentionMetadata]
    tknp: Optional[TokenParallelMetadata] = None
    micro_batches: Optional[list] = None
    dp: Optional[DPMetadata] = None
    kv_connector: Any = None
    extra: dict = field(default_factory=dict)


_ctx: Optional[ForwardContext] = None
_timings: dict[int, list[float]] = defaultdict(list)
_last_log = 0.0


def get_forward_context() -> ForwardContext:
    if _ctx is None:
        raise RuntimeError("no forward context is set; run the model inside set_forward_context(...)")
    return _ctx


def has_forward_context() -> bool:
    return _ctx is not None


@contextlib.contextmanager
def set_forward_context(attn_metadata, tknp=None, dp=None, kv_connector=None, num_tokens: int = 0,
                        micro_batches=None):
    global _ctx, _last_log
    track = envs.VLLM_LOG_BATCHSIZE_INTERVAL >= 0 and num_tokens > 0
    t0 = time.perf_counter() if track else 0.0
    prev, _ctx = _ctx, ForwardContext(attn_metadata, tknp, micro_batches, dp, kv_connector)
    try:
        yield _ctx
    finally:
        _ctx = prev
        if track:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            now = time.perf_counter()
            _timings[num_tokens].append((now - t0) * 1000)
            if now - _last_log > envs.VLLM_LOG_BATCHSIZE_INTERVAL:
                _last_log = now
                rows = sorted(((bs, len(v), sorted(v)[len(v) // 2]) for bs, v in _timings.items() if len(v) > 1),
                              key=lambda r: -r[1])
                if rows:
                    logger.info("forward time by batch size (tokens, count, median ms): %s",
                                [(b, c, round(m, 2)) for b, c, m in rows])
