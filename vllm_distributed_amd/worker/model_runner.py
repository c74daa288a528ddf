"""Runs the model for one scheduler step on one device.

Per step: apply the scheduler's deltas to the persistent batch, order the
scheduled requests decode-first, build the flat inputs with vectorised numpy,
ship them to fixed device buffers, run the model (a replayed hipGraph when the
step is pure decode and fits a captured size, eager otherwise), compute logits
for the last token of every request and sample.  The only host sync is the
read-back of the sampled ids.

Reference: vllm/v1/worker/gpu_model_runner.py (_update_states :670-870,
_prepare_inputs :892-1109, execute_model :1613-1923, _dummy_run :2303-2471,
capture_model :2683-2723, KV allocation :2808-2987, get_kv_cache_spec :3010-3089)
and its token-parallel hooks (:331-510, :1813-1843).
"""
from __future__ import annotations

import gc
import time
from typing import Optional, Union

import numpy as np
import torch
import torch.nn as nn

from .. import envs, ops
from ..config import VllmConfig
from ..forward_context import AttentionMetadata, MicroBatch, TokenParallelMetadata, set_forward_context
from ..layers.attention import Attention
from ..layers.sampler import Sampler, SamplingMetadata
from ..logger import init_logger
from ..models.loader import get_model
from ..models.utils import IntermediateTensors
from ..parallel.state import get_pp_group, get_tknp_group, get_tp_group, graph_capture
from ..request import LogprobsLists
from ..sched.kv_cache_interface import KVCacheConfig, KVCacheSpec, MLASpec, MambaSpec
from ..sched.output import EMPTY_MODEL_RUNNER_OUTPUT, ModelRunnerOutput, SchedulerOutput
from .input_batch import InputBatch

logger = init_logger(__name__)
_CAPTURE_STATE_ANCHOR = None      # see ModelRunner.capture_model


class DeferredOutput:
    """Result of a step that the engine collects after it has launched the next one (run-ahead scheduling)."""

    def __init__(self, runner, gen, req_ids, done, buf, S, event, logprobs_t):
        self.runner, self.gen, self.req_ids, self.done = runner, gen, req_ids, done
        self.buf, self.S, self.event, self.logprobs_t = buf, S, event, logprobs_t

        self._result: Optional[ModelRunnerOutput] = None
        self.prev: Optional["DeferredOutput"] = None

    def get(self) -> ModelRunnerOutput:
        """Wait for the ids, apply them to the runner's host-side token arrays (once) and return the step's
        result.  Steps are collected in the order they ran."""
        if self._result is not None:
            return self._result
        if self.prev is not None:
            self.prev.get()
            self.prev = None
        r = self.runner
        if self.event is not None:
            self.event.synchronize()
        ids = self.buf[:self.S].tolist()
        r._check_comm_errors()
        lists = [[t] if self.done[i] else [] for i, t in enumerate(ids)]
        r._apply_sampled(self.req_ids, lists)
        if r._pend_gen == self.gen:             # no later step took over: the host arrays are complete again
            r._pend_of, r._pend_ids = {}, None
        if r._deferred_open is self:
            r._deferred_open = None
        self._result = ModelRunnerOutput(
            req_ids=self.req_ids, req_id_to_index={rid: i for i, rid in enumerate(self.req_ids)},
            sampled_token_ids=lists, spec_token_ids=None, logprobs=Sampler.to_lists(self.logprobs_t),
            prompt_logprobs_dict={}, finished_sending=None, finished_recving=None, failed_recving=None,
            num_nans_in_logits=None)
        return self._result


class ModelRunner:
    def __init__(self, vllm_config: VllmConfig, device: torch.device):
        self.vllm_config = vllm_config
        self.device = device
        self.is_gpu = device.type == "cuda"
        mc, cc, sc, pc = (vllm_config.model_config, vllm_config.cache_config, vllm_config.scheduler_config,
                          vllm_config.parallel_config)
        self.model_config, self.cache_config, self.scheduler_config, self.parallel_config = mc, cc, sc, pc
        self.dtype = mc.dtype
        self.block_size = cc.block_size
        self.max_model_len = mc.max_model_len
        self.max_num_tokens = sc.max_num_batched_tokens
        self.max_num_reqs = sc.max_num_seqs
        self.vocab_size = mc.get_vocab_size()
        self.hidden_size = mc.get_hidden_size()
        self.num_heads = mc.get_num_attention_heads(pc)
        self.num_kv_heads = mc.get_num_kv_heads(pc)
        self.head_size = mc.get_head_size()
        spec = vllm_config.speculative_config
        self.num_spec_tokens = spec.num_speculative_tokens if spec else 0
        group = max(self.num_heads // self.num_kv_heads, 1)
        # query tokens per sequence the split-KV kernel form can take (<= 32 MFMA columns per kv head)
        self.max_decode_q = max(1, min(1 + self.num_spec_tokens, 32 // group if self.head_size <= 128 else 16 // group))
        self.tknp_size = pc.token_parallel_size
        self.tknp_rank = get_tknp_group().rank_in_group if self.tknp_size > 1 else 0
        # token parallelism: the rows every rank attends for change from step to step, but the peer-memory
        # row exchange reads its counts and index lists from device buffers, so decode steps are captured
        # at the padded batch size like any other (without that exchange: eager only)
        self.use_graphs = self.is_gpu and vllm_config.compilation_config.level > 0 and not mc.enforce_eager
        self.capture_sizes = list(vllm_config.compilation_config.cudagraph_capture_sizes or []) if self.use_graphs else []
        self.model: Optional[nn.Module] = None
        self.sampler = Sampler()
        self.kv_caches: list[tuple[torch.Tensor, torch.Tensor]] = []
        self.kv_arena: Optional[torch.Tensor] = None
        self.attn_layers: dict[str, Attention] = {}
        self.state_table: Optional[torch.Tensor] = None      # hybrid models only, see _init_hybrid_state
        self.uses_mrope = False
        self._step_mrope: Optional[torch.Tensor] = None
        self.input_batch: Optional[InputBatch] = None
        self.graphs: dict[int, tuple] = {}
        self.sleep_allocator = None
        pcfg = vllm_config.parallel_config
        self.sp_multiple = pcfg.tensor_parallel_size if vllm_config.compilation_config.enable_sequence_parallelism \
            else 1
        # async tensor parallelism: large eager batches run as two micro-batches whose all-reduces
        # overlap with each other's GEMMs (models/llama.py::_forward_micro_batched)
        self.async_tp = bool(vllm_config.compilation_config.enable_async_tp
                             and pcfg.tensor_parallel_size > 1 and self.tknp_size == 1 and self.sp_multiple == 1)
        self.graph_pool = None
        self._step = 0
        self._rng = np.random.default_rng(mc.seed)
        self.drafter = None
        self.rejection_sampler = None
        self.kv_connector = None
        self.lora_manager = None
        self.eplb_state = None
        self.tk_rx = None              # peer-memory row exchange of the token-parallel group (set by load_model)
        self._tknp_step_bs = None
        self.encoder_cache: dict[str, dict[int, torch.Tensor]] = {}     # request -> item index -> embeddings
        self.local_chunk = 0           # chunk length of chunked-local attention layers (set by load_model)
        self._l_counts = (0, 0)        # (virtual sequences, virtual prefill-form sequences) of the step
        self.dual_chunk = False        # some layer runs dual-chunk attention (set by load_model)
        self._init_buffers()

    # ------------------------------------------------------------------
    def _init_buffers(self) -> None:
        dev, T, S = self.device, self.max_num_tokens, self.max_num_reqs
        Tpad = max(T, max(self.capture_sizes, default=0))
        self.input_ids = torch.zeros(Tpad, dtype=torch.int32, device=dev)
        self.positions = torch.zeros(Tpad, dtype=torch.int64, device=dev)
        self.slot_mapping = torch.full((Tpad,), -1, dtype=torch.int64, device=dev)
        Spad = max(S, max(self.capture_sizes, default=0))
        self.max_blocks = -(-self.max_model_len // self.block_size)
        self.block_table = torch.zeros((Spad, self.max_blocks), dtype=torch.int32, device=dev)
        self.seq_lens = torch.zeros(Spad, dtype=torch.int32, device=dev)
        self.query_start_loc = torch.zeros(Spad + 1, dtype=torch.int32, device=dev)
        self.prefill_qsl = torch.zeros(Spad + 1, dtype=torch.int32, device=dev)
        self.logits_indices = torch.zeros(Tpad, dtype=torch.int64, device=dev)
        self.mb_qsl = torch.zeros(Spad + 1, dtype=torch.int32, device=dev)          # second micro-batch, rebased
        self.mb_prefill_qsl = torch.zeros(Spad + 1, dtype=torch.int32, device=dev)
        self.cascade_seq_len = torch.zeros(1, dtype=torch.int32, device=dev)
        self.cascade_qsl = torch.zeros(2, dtype=torch.int32, device=dev)
        self.cascade_suffix_lens = torch.zeros(Spad, dtype=torch.int32, device=dev)
        self.pin = self.is_gpu
        self._stage_bufs, self._stage_idx, self._stage_off = None, 0, 0
        # one step uploads ~24 bytes per token plus one block-table row per sequence
        self._stage_bytes = int(Tpad * 40 + Spad * (self.max_blocks * 4 + 64) + (1 << 20))
        self.intermediate: Optional[IntermediateTensors] = None
        self._dp_moe = False
        # run-ahead scheduling: sampled ids of the step in flight, still on the device, and which request
        # takes which of them as its next input token
        self._pend_ids: Optional[torch.Tensor] = None
        self._pend_of: dict[str, int] = {}
        self._pend_gen = 0
        self._pend_dst = torch.zeros(Spad, dtype=torch.int64, device=dev)
        self._pend_src = torch.zeros(Spad, dtype=torch.int64, device=dev)
        self._deferred_bufs = None
        self._sample_pos: Optional[np.ndarray] = None
        self._deferred_open: Optional[DeferredOutput] = None     # handed out, not collected yet

    def _stage(self, src: np.ndarray) -> torch.Tensor:
        """Host array -> a slice of a persistent pinned arena (what an asynchronous upload reads from).
        Three arenas rotate by step: a slice is reused two steps later, after the step that read it has
        passed its end-of-step synchronisation.  Pinning a fresh tensor per upload, as before, cost a
        page-locking system call each time."""
        src = np.ascontiguousarray(src)
        if not self.pin:
            return torch.from_numpy(src)
        nb = src.nbytes
        if self._stage_bufs is None:
            self._stage_bufs = [torch.empty(self._stage_bytes, dtype=torch.uint8).pin_memory() for _ in range(3)]
        off = (self._stage_off + 15) // 16 * 16
        if off + nb > self._stage_bytes:
            return torch.from_numpy(src).pin_memory()          # oversize upload: rare, pay the pin
        self._stage_off = off + nb
        view = self._stage_bufs[self._stage_idx][off:off + nb].view(torch.from_numpy(src[:0]).dtype).view(src.shape)
        view.numpy()[...] = src
        return view

    def _next_stage(self) -> None:
        self._stage_idx = (self._stage_idx + 1) % 3
        self._stage_off = 0

    def _to_dev(self, dst: torch.Tensor, src: np.ndarray) -> None:
        t = self._stage(src)
        dst[: t.shape[0]].copy_(t, non_blocking=True)

    # ------------------------------------------------------------------
    def _sleep_pool(self, tag: str):
        """Allocation context of the address-stable sleep allocator, or a no-op when sleep mode is off."""
        import contextlib
        if not (self.is_gpu and self.vllm_config.model_config.enable_sleep_mode):
            return contextlib.nullcontext()
        from ..device_allocator import SleepAllocator
        self.sleep_allocator = SleepAllocator.get()
        return self.sleep_allocator.use_memory_pool(tag)

    def load_model(self) -> None:
        t0 = time.time()
        if self.is_gpu:
            torch.cuda.synchronize()
            before = torch.cuda.memory_allocated(self.device)
        with self._sleep_pool("weights"):
            self.model = get_model(self.vllm_config, self.device)
        if self.is_gpu:
            self.model_memory = torch.cuda.memory_allocated(self.device) - before
            logger.info("weights take %.2f GiB, loaded in %.1f s", self.model_memory / 2**30, time.time() - t0)
        # expert layers behind data-parallel attention trade tokens between replicas: every step then starts with the
        # replicas' row counts, and decode graphs are off (a captured exchange would freeze another replica's size)
        from ..layers.moe import FusedMoE
        from ..parallel.state import get_dp_group
        self._dp_moe = get_dp_group().world_size > 1 and any(isinstance(m, FusedMoE) for m in self.model.modules())
        if getattr(self.model, "requires_eager", False) and self.use_graphs:
            logger.info("%s runs without decode graphs", type(self.model).__name__)
            self.use_graphs, self.capture_sizes = False, []
        if self._dp_moe and self.use_graphs:
            logger.info("data-parallel attention with expert layers: decode graphs off")
            self.use_graphs, self.capture_sizes = False, []
        # every layer that owns per-request cache memory: attention layers and state-space mixers
        self.attn_layers = {n: m for n, m in self.model.named_modules()
                            if (isinstance(m, Attention) or getattr(m, "is_state_space_layer", False)
                                or getattr(m, "is_mla", False)) and getattr(m, "kv_source", None) is None}
        for n, m in self.attn_layers.items():
            m.layer_name = n
        # heads of a size the kernels do not have run zero-padded (layers/attention.py): size buffers for that
        padded = [m.head_size for m in self.attn_layers.values() if isinstance(m, Attention)]
        if padded:
            self.head_size = max(self.head_size, max(padded))
        self._init_hybrid_state()
        self._init_local_attention()
        self.uses_mrope = bool(getattr(self.model, "uses_mrope", False))
        spec = self.vllm_config.speculative_config
        if spec is not None and get_pp_group().is_last_rank:
            from ..spec_decode import make_proposer
            from ..layers.rejection_sampler import RejectionSampler
            self.drafter = make_proposer(self.vllm_config, self)
            self.rejection_sampler = RejectionSampler()
            if hasattr(self.drafter, "attention_layers"):
                # the draft layers cache K/V like any other layer (same blocks, own tensors)
                for n, m in self.drafter.attention_layers().items():
                    m.layer_name = n
                    self.attn_layers[n] = m
        pc = self.parallel_config
        if pc.enable_eplb and pc.enable_expert_parallel:
            self._init_eplb()
        self._init_tknp_exchange()
        if self.vllm_config.lora_config is not None:
            from ..lora.manager import LoRAManager
            self.lora_manager = LoRAManager(self.model, self.vllm_config, self.device)

    def _init_local_attention(self) -> None:
        """Chunked local attention (Llama-4's rotary layers): tokens attend inside their own chunk of
        ``local_chunk`` positions.  The step's sequences are re-cut into one virtual sequence per
        (sequence, chunk) with a block table that starts at the chunk's first block, so the ordinary
        paged kernels serve these layers unchanged.  Reference: vllm/v1/attention/backends/utils.py:242-420
        (make_local_attention_virtual_batches)."""
        chunks = {m.local_chunk for m in self.attn_layers.values() if getattr(m, "local_chunk", 0)}
        if not chunks:
            return
        if len(chunks) > 1:
            raise ValueError(f"layers disagree on the local attention chunk: {sorted(chunks)}")
        C = chunks.pop()
        if C % self.block_size:
            raise ValueError(f"attention chunk {C} must be a multiple of the KV block size {self.block_size}")
        self.local_chunk = C
        # a decode-form sequence must lie inside one chunk: one query token each
        self.max_decode_q = 1
        dev = self.device
        Spad = self.block_table.shape[0]
        # a prefill of n tokens touches at most n / C + 2 chunks
        self.l_max = Spad * 2 + self.max_num_tokens // C + 1
        self.l_width = C // self.block_size
        self.l_block_table = torch.zeros((self.l_max, self.l_width), dtype=torch.int32, device=dev)
        self.l_seq_lens = torch.zeros(self.l_max, dtype=torch.int32, device=dev)
        self.l_qsl = torch.zeros(self.l_max + 1, dtype=torch.int32, device=dev)
        self.l_prefill_qsl = torch.zeros(self.l_max + 1, dtype=torch.int32, device=dev)
        self._stage_bytes += self.l_max * (self.l_width * 4 + 64)
        # dual-chunk attention (layers/attention.py::DualChunkAttention): the same virtual sequences also read the
        # chunk before theirs and everything older
        dca = [m for m in self.attn_layers.values() if getattr(m, "dual_chunk", False)]
        self.dual_chunk = bool(dca)
        if dca:
            self.d_succ_bt = torch.zeros((self.l_max, self.l_width), dtype=torch.int32, device=dev)
            self.d_succ_lens = torch.zeros(self.l_max, dtype=torch.int32, device=dev)
            self.d_inter_lens = torch.zeros(self.l_max, dtype=torch.int32, device=dev)
            self.d_owner = torch.zeros(self.l_max, dtype=torch.int64, device=dev)
            self.d_orig_max = max(m.original_max for m in dca)
            self.d_scale = torch.ones((self.max_num_tokens, 1), dtype=self.dtype, device=dev) if self.d_orig_max else None
            self._d_inter_bt, self._d_passes = None, 3
            self._stage_bytes += self.l_max * (self.l_width * 4 + 64) + self.max_num_tokens * 4

    def _upload_local_attention(self, rows, n_sched, cu, seq_lens, nd: int) -> None:
        """Per-chunk view of the step's sequences (see _init_local_attention)."""
        C, bs, W = self.local_chunk, self.block_size, self.l_width
        S = len(rows)
        n = n_sched.astype(np.int64)
        end = seq_lens.astype(np.int64)                 # one past the last position of this step
        start = end - n
        c_lo, c_hi = start // C, (end - 1) // C
        per_seq = np.maximum(c_hi - c_lo + 1, 0)
        if nd and (per_seq[:nd] != 1).any():
            raise RuntimeError("a decode-form sequence spans two attention chunks")
        Sv = int(per_seq.sum())
        if Sv > self.l_max:
            raise RuntimeError(f"{Sv} (sequence, chunk) pairs exceed the local-attention tables ({self.l_max})")
        owner = np.repeat(np.arange(S, dtype=np.int64), per_seq)
        first = np.zeros(S + 1, dtype=np.int64)
        np.cumsum(per_seq, out=first[1:])
        chunk = c_lo[owner] + (np.arange(Sv, dtype=np.int64) - first[:-1][owner])
        q_lo = np.maximum(start[owner], chunk * C)
        q_hi = np.minimum(end[owner], (chunk + 1) * C)
        l_len = (q_hi - chunk * C).astype(np.int32)
        l_cu = np.zeros(Sv + 1, dtype=np.int32)
        np.cumsum(q_hi - q_lo, out=l_cu[1:])
        bt = self.input_batch.block_tables[0]
        cols = chunk[:, None] * (C // bs) + np.arange(W, dtype=np.int64)[None, :]
        valid = cols < bt.shape[1]
        l_bt = np.where(valid, bt[rows[owner][:, None], np.minimum(cols, bt.shape[1] - 1)], 0).astype(np.int32)
        self._to_dev(self.l_seq_lens, l_len)
        self._to_dev(self.l_qsl, l_cu)
        if Sv:
            self.l_block_table[:Sv].copy_(self._stage(l_bt), non_blocking=True)
        if nd < Sv:
            self._to_dev(self.l_prefill_qsl, (l_cu[nd:] - l_cu[nd]).astype(np.int32))
        self._l_counts = (Sv, Sv - nd)
        if not self.dual_chunk:
            return
        prev = chunk - 1
        self._d_passes = 3 if Sv and int(chunk.max()) >= 1 else 1
        self._to_dev(self.d_succ_lens, np.where(prev >= 0, C, 0).astype(np.int32))
        self._to_dev(self.d_inter_lens, (np.maximum(prev, 0) * C).astype(np.int32))
        if self._d_passes == 3:
            cols = prev[:, None] * (C // bs) + np.arange(W, dtype=np.int64)[None, :]
            valid = (prev >= 0)[:, None] & (cols < bt.shape[1])
            s_bt = np.where(valid, bt[rows[owner][:, None], np.clip(cols, 0, bt.shape[1] - 1)], 0).astype(np.int32)
            self.d_succ_bt[:Sv].copy_(self._stage(s_bt), non_blocking=True)
            if Sv == S:                    # every sequence inside one chunk: the step's own table serves
                self._d_inter_bt = None
            else:
                self._to_dev(self.d_owner, owner)
                self._d_inter_bt = self.block_table.index_select(0, self.d_owner[:Sv])
        if self.d_scale is not None:
            # tokens known of the sequence (the whole prompt while it is prefilled, the current length afterwards)
            known = np.maximum(self.input_batch.num_tokens[rows].astype(np.int64), end)
            f = np.maximum(0.1 * np.log(np.maximum(known, 1) / self.d_orig_max) + 1.0, 1.0)
            self._to_dev(self.d_scale.view(-1), np.repeat(f, n).astype(np.float32))

    def _init_tknp_exchange(self) -> None:
        """Token parallelism on GPUs of one node: rows travel through the peer-memory all-to-all-v
        (device-side counts and index lists in persistent buffers), which is what lets decode steps run
        as captured graphs; without it (CPU, several nodes, kernel unavailable) the layers fall back to
        torch.distributed all-to-all with host-side split lists, eager only."""
        self.tk_rx = None
        if self.tknp_size == 1:
            return
        if self.is_gpu and not envs.VDA_DISABLE_PEER_ALL2ALL:
            from ..parallel.all2all import RowExchange
            it = torch.empty((), dtype=self.dtype).element_size()
            width = (self.num_heads + 2 * self.num_kv_heads) * self.head_size
            cap = max(self.max_num_tokens, max(self.capture_sizes, default=0))
            # a row also carries the sampled token ids of a step (4 bytes per sequence) from the root
            row_bytes = -(-max(width * it, 4 * max(self.max_num_reqs, max(self.capture_sizes, default=0))) // 16) * 16
            rx = RowExchange(get_tknp_group(), cap, row_bytes)
            if rx.peer is not None:
                self.tk_rx = rx
                n, dev = self.tknp_size, self.device
                self.tk_sorted = torch.zeros(cap, dtype=torch.int32, device=dev)
                self.tk_offsets = torch.zeros(n + 1, dtype=torch.int32, device=dev)
                self.tk_goff = torch.zeros(n + 1, dtype=torch.int32, device=dev)
                self.tk_iota = torch.arange(cap, dtype=torch.int32, device=dev)
                self.tk_back = torch.zeros(cap, dtype=torch.int64, device=dev)
                self.tk_dummy = torch.zeros((1, row_bytes // it), dtype=self.dtype, device=dev)
                self.tk_ids_row = torch.zeros((1, row_bytes // 4), dtype=torch.int32, device=dev)
                self.tk_bcast_off = torch.arange(n + 1, dtype=torch.int32, device=dev)      # the root: one row to everyone
                self.tk_row0 = torch.zeros(n, dtype=torch.int32, device=dev)                # ... always its row 0
                self.tk_zero_off = torch.zeros(n + 1, dtype=torch.int32, device=dev)
                return
            rx.close()
        if self.use_graphs:
            logger.info("token parallelism without the peer-memory row exchange: decode graphs off")
        self.use_graphs = False
        self.capture_sizes = []

    def _tknp_graph_bs(self, S: int, n_sched: np.ndarray) -> Optional[int]:
        """Padded batch size of the captured graph this step replays, None for an eager step.  Decided
        from the GLOBAL batch, which every token-parallel rank sees alike."""
        if not self.use_graphs or not self.graphs or self.lora_manager is not None or not len(n_sched):
            return None
        if int(n_sched.max()) != 1 or S > self.capture_sizes[0]:
            return None
        return self.vllm_config.pad_for_cudagraph(S)

    def _init_hybrid_state(self) -> None:
        """Hybrid models (attention and state-space layers in one stack, e.g. Jamba): the paged KV cache
        serves the attention layers only.  Every state-space layer gets one state slot per batch row,
        allocated here next to the weights (so memory profiling counts it), and a sequence's slot is its
        row of the persistent batch + 1; slot 0 absorbs the padding rows of captured decode graphs."""
        self.state_table: Optional[torch.Tensor] = None
        ss = {n: m for n, m in self.attn_layers.items() if getattr(m, "is_state_space_layer", False)}
        if not ss or len(ss) == len(self.attn_layers):
            return
        nslots = max(self.max_num_reqs, max(self.capture_sizes, default=0)) + 1
        total = 0
        with self._sleep_pool("kv_cache"):
            for n, m in ss.items():
                conv_shape, ssm_shape = m.state_shapes()
                dt0, dt1 = getattr(m, "state_dtypes", (m.conv_dtype, torch.float32))
                conv = torch.zeros((nslots, *conv_shape), dtype=dt0, device=self.device)
                ssm = torch.zeros((nslots, *ssm_shape), dtype=dt1, device=self.device)
                m.bind_kv_cache(conv, ssm)
                total += conv.numel() * conv.element_size() + ssm.numel() * ssm.element_size()
                del self.attn_layers[n]
            self.state_table = torch.zeros((nslots - 1, 1), dtype=torch.int32, device=self.device)
        self.state_layers = ss
        logger.info("hybrid model: %d state-space layers, %d state slots, %.2f GiB of recurrent state", len(ss),
                    nslots, total / 2**30)

    def _init_eplb(self) -> None:
        from ..layers.moe import FusedMoE
        from ..parallel.eplb import EplbState
        from ..parallel.state import get_ep_group
        layers = [m for m in self.model.modules() if isinstance(m, FusedMoE)]
        if not layers or get_ep_group().world_size == 1:
            return
        pc = self.parallel_config
        hf = self.model_config.hf_config
        first = layers[0]
        self.eplb_state = EplbState(len(layers), first.global_num_experts, first.num_physical_experts,
                                    get_ep_group().world_size, window=pc.eplb_window_size,
                                    step_interval=pc.eplb_step_interval,
                                    num_groups=int(getattr(hf, "n_group", 1) or 1))
        for i, m in enumerate(layers):
            m.eplb_state, m.eplb_layer = self.eplb_state, i
        self._eplb_layers = layers
        # routing tables and load counters are persistent device tensors updated in place and expert weights
        # move inside their own storage, so captured decode graphs stay valid across rearrangements
        logger.info("EPLB on: %d MoE layers, %d logical -> %d physical experts", len(layers),
                    first.global_num_experts, first.num_physical_experts)

    def _eplb_step(self) -> None:
        from ..parallel.state import get_ep_group
        st = self.eplb_state
        weights = [m.expert_weight_tensors() for m in self._eplb_layers]
        g = get_ep_group()
        st.step_done(g.cpu_group, weights, p2p_group=g.device_group if (self.is_gpu and not g.staged) else g.cpu_group)

    def get_model(self) -> nn.Module:
        return self.model

    def get_kv_cache_spec(self) -> dict[str, KVCacheSpec]:
        dt = self.dtype if self.cache_config.cache_dtype == "auto" else \
            {"fp8": torch.float8_e4m3fn, "bfloat16": torch.bfloat16, "float16": torch.float16}[self.cache_config.cache_dtype]
        if dt == torch.float8_e4m3fn and any(isinstance(m, Attention) and m.head_size not in (64, 128)
                                             for m in self.attn_layers.values()):
            raise ValueError("the fp8 KV cache supports head sizes up to 128 (and latent caches)")
        return {n: m.get_kv_cache_spec(self.block_size, dt) for n, m in self.attn_layers.items()}

    def initialize_kv_cache(self, cfg: KVCacheConfig) -> None:
        """The cache is one zero-filled allocation carved into per-layer K and V views (a single
        allocation is a single IPC handle for the xGMI KV connector).  An instance that exports KV
        to peers allocates it IPC-shareable; ``VDA_KV_IPC_CHUNK_GB`` > 0 splits it into several
        regions at layer boundaries."""
        self.kv_cache_config = cfg
        kt = self.vllm_config.kv_transfer_config
        shareable = self.is_gpu and kt is not None and kt.is_kv_producer and kt.kv_connector != "SharedStorageConnector"
        tensors = list(cfg.kv_cache_tensors)
        # group layer tensors into chunks
        limit = int(envs.VDA_KV_IPC_CHUNK_GB * 2**30) if (shareable and envs.VDA_KV_IPC_CHUNK_GB > 0) else None
        groups: list[list] = [[]]
        used = 0
        for t in tensors:
            if limit is not None and t.size > limit:
                raise ValueError(f"one layer's KV cache ({t.size / 2**30:.1f} GiB) exceeds the IPC chunk limit "
                                 f"({envs.VDA_KV_IPC_CHUNK_GB} GiB); lower gpu_memory_utilization on this exporter")
            if limit is not None and used + t.size > limit and groups[-1]:
                groups.append([])
                used = 0
            groups[-1].append(t)
            used += t.size
        self.kv_regions = []
        self.kv_chunks: list[torch.Tensor] = []
        for g in groups:
            nbytes = sum(t.size for t in g)
            if shareable:
                from ..device_allocator import alloc_shareable
                r = alloc_shareable(nbytes, self.device)
                self.kv_regions.append(r)
                self.kv_chunks.append(r.tensor)
            else:
                with self._sleep_pool("kv_cache"):
                    self.kv_chunks.append(torch.zeros(nbytes, dtype=torch.uint8, device=self.device))
        self.kv_arena = self.kv_chunks[0] if len(self.kv_chunks) == 1 else None
        spec_of = {}
        for g in cfg.kv_cache_groups:
            for n in g.layer_names:
                spec_of[n] = (g.per_layer or {}).get(n, g.kv_cache_spec)
        self.kv_caches = []
        # layer name -> (chunk index, byte offset of K inside the chunk, bytes of K == bytes of V)
        self.kv_layer_loc: dict[str, tuple[int, int, int]] = {}
        # layer name -> [(chunk index, byte offset, bytes, KV heads)] for every per-block tensor of the layer
        # (K and V; latent + transposed latent; state tensors).  Heads = 0: not sharded by head (copied whole).
        self.kv_layer_parts: dict[str, list[tuple[int, int, int, int]]] = {}
        nb = cfg.num_blocks
        for ci, g in enumerate(groups):
            off = 0
            for t in g:
                raw = self.kv_chunks[ci][off: off + t.size]
                sp = spec_of[t.shared_by[0]]
                if isinstance(sp, MambaSpec):
                    # one region per state tensor, each [num_blocks, *shape]; a request's slot is its block id
                    parts, o2 = [], 0
                    for i, (shape, pb) in enumerate(zip(sp.shapes, sp.part_bytes())):
                        dt_i = sp.dtypes[i] if sp.dtypes else sp.dtype
                        numel = 1
                        for d in shape:
                            numel *= d
                        if pb != numel * dt_i.itemsize:
                            raise ValueError(f"state tensor of {numel * dt_i.itemsize} bytes per request is not "
                                             "a multiple of 16 bytes")
                        parts.append(raw[o2: o2 + nb * pb].view(dt_i).view(nb, *shape))
                        o2 += nb * pb
                    locs, o3 = [], 0
                    for pb in sp.part_bytes():
                        locs.append((ci, off + o3, nb * pb, 0))
                        o3 += nb * pb
                    for n in t.shared_by:
                        self.attn_layers[n].bind_kv_cache(*parts)
                        self.kv_layer_loc[n] = (ci, off, t.size)
                        self.kv_layer_parts[n] = locs
                    self.kv_caches.append(tuple(parts))
                    off += t.size
                    continue
                if isinstance(sp, MLASpec):
                    kb = nb * sp.block_size * sp.head_size * sp.dtype.itemsize
                    k = raw[:kb].view(sp.dtype).view(nb, sp.block_size, sp.head_size)
                    v = raw[kb:].view(sp.dtype).view(nb, sp.v_size, sp.block_size)
                    for n in t.shared_by:
                        self.attn_layers[n].bind_kv_cache(k, v)
                        self.kv_layer_loc[n] = (ci, off, t.size)
                        self.kv_layer_parts[n] = [(ci, off, kb, 0), (ci, off + kb, t.size - kb, 0)]
                    self.kv_caches.append((k, v))
                    off += t.size
                    continue
                half = t.size // 2
                k = raw[:half].view(sp.dtype).view(nb, sp.num_kv_heads, sp.block_size, sp.head_size)
                v = raw[half:].view(sp.dtype).view(nb, sp.num_kv_heads, sp.head_size, sp.block_size)
                for n in t.shared_by:
                    self.attn_layers[n].bind_kv_cache(k, v)
                    self.kv_layer_loc[n] = (ci, off, half)
                    self.kv_layer_parts[n] = [(ci, off, half, sp.num_kv_heads), (ci, off + half, half, sp.num_kv_heads)]
                self.kv_caches.append((k, v))
                off += t.size
        self.input_batch = InputBatch(self.max_num_reqs, self.max_model_len, self.block_size, self.vocab_size,
                                      max(len(cfg.kv_cache_groups), 1), self.pin)
        if self.kv_connector is not None:
            self.kv_connector.register_kv_caches(self)

    # ------------------------------------------------------------------
    # state updates
    # ------------------------------------------------------------------
    def _update_states(self, so: SchedulerOutput) -> None:
        ib = self.input_batch
        for rid in so.finished_req_ids:
            ib.remove_request(rid)
        for nr in so.scheduled_new_reqs:
            ib.add_request(nr.req_id, nr.prompt_token_ids, nr.sampling_params, nr.block_ids, nr.num_computed_tokens,
                           nr.pooling_params, nr.lora_request, nr.mm_inputs, nr.mm_positions)
            if self.uses_mrope and nr.mm_inputs:
                st = ib.reqs[nr.req_id]
                st.mrope_positions, st.mrope_delta = self.model.get_mrope_positions(
                    len(nr.prompt_token_ids), nr.mm_positions, nr.mm_inputs)
        cr = so.scheduled_cached_reqs
        for i, rid in enumerate(cr.req_ids):
            st = ib.reqs.get(rid)
            if st is None:
                raise RuntimeError(f"scheduler references unknown request {rid}")
            row = st.row
            ib.num_computed[row] = cr.num_computed_tokens[i]
            if cr.resumed_from_preemption[i]:
                ib.reset_blocks(row, cr.new_block_ids[i])
            else:
                ib.append_blocks(row, cr.new_block_ids[i])
            new_toks = cr.new_token_ids[i]
            if new_toks:
                # tokens sampled elsewhere (other PP stage / before preemption): place them by position
                s = int(cr.num_computed_tokens[i])
                e = s + len(new_toks)
                ib.token_ids[row, s:e] = new_toks
                if e > ib.num_tokens[row]:
                    ib.num_tokens[row] = e
            spec = so.scheduled_spec_decode_tokens.get(rid)
            if spec:
                s = int(ib.num_tokens[row])
                ib.token_ids[row, s:s + len(spec)] = spec

    # ------------------------------------------------------------------
    def _flat_arrays(self, rows: np.ndarray, n_sched: np.ndarray):
        """Per-token arrays of a list of requests (batch rows ``rows``, ``n_sched`` tokens each)."""
        ib = self.input_batch
        S = len(rows)
        T = int(n_sched.sum())
        cu = np.zeros(S + 1, dtype=np.int32)
        np.cumsum(n_sched, out=cu[1:])
        req_idx = np.repeat(np.arange(S, dtype=np.int64), n_sched)
        computed = ib.num_computed[rows].astype(np.int64)
        pos = computed[req_idx] + (np.arange(T, dtype=np.int64) - cu[:-1].astype(np.int64)[req_idx])
        tok_rows = rows[req_idx]
        bt = ib.block_tables[0]
        slot = bt[tok_rows, pos // self.block_size].astype(np.int64) * self.block_size + pos % self.block_size
        seq_lens = (computed + n_sched).astype(np.int32)
        return cu, req_idx, pos, tok_rows, slot, seq_lens

    def _upload_attention_inputs(self, rows, n_sched, cu, pos, slot, seq_lens) -> tuple[int, int]:
        """Ship the attention-side arrays of the requests this rank attends for; returns
        (decode-form sequence count, decode-form token count)."""
        ib = self.input_batch
        S = len(rows)
        nd = int((n_sched <= self.max_decode_q).sum())
        ndt = int(n_sched[:nd].sum())
        max_seq = int(seq_lens.max()) if S else 0
        W = max(-(-max_seq // self.block_size), 1)
        self._to_dev(self.positions, pos)
        self._to_dev(self.slot_mapping, slot)
        self._to_dev(self.seq_lens, seq_lens)
        self._to_dev(self.query_start_loc, cu)
        if S:
            self.block_table[:S, :W].copy_(self._stage(ib.block_tables[0][rows, :W]), non_blocking=True)
        if nd < S:
            self._to_dev(self.prefill_qsl, (cu[nd:] - cu[nd]).astype(np.int32))
        if self.state_table is not None and S:
            self._to_dev(self.state_table.view(-1), (rows + 1).astype(np.int32))
        if self.local_chunk:
            self._upload_local_attention(rows, n_sched, cu, seq_lens, nd)
        return nd, ndt

    def _prepare_inputs(self, so: SchedulerOutput):
        ib = self.input_batch
        req_ids = list(so.num_scheduled_tokens)
        n_sched = np.fromiter((so.num_scheduled_tokens[r] for r in req_ids), dtype=np.int32, count=len(req_ids))
        rows = np.fromiter((ib.reqs[r].row for r in req_ids), dtype=np.int64, count=len(req_ids))
        # decode-form requests first (stable), then prefill-form
        is_dec = n_sched <= self.max_decode_q
        order = np.argsort(~is_dec, kind="stable")
        req_ids = [req_ids[i] for i in order]
        n_sched, rows = n_sched[order], rows[order]
        S = len(req_ids)
        cu, req_idx, pos, tok_rows, slot, seq_lens = self._flat_arrays(rows, n_sched)
        T = int(cu[-1])
        tmd = None
        if self.tknp_size > 1:
            tmd, att = self._prepare_token_parallel(so, req_ids, rows, n_sched, req_idx, tok_rows, pos)
        else:
            self._to_dev(self.input_ids, ib.token_ids[tok_rows, pos])
            if self._pend_of:
                self._feed_pending_tokens(req_ids, rows, cu)
            if self.uses_mrope:
                pos = self._mrope_positions(req_ids, rows, n_sched, cu, pos)
            nd, ndt = self._upload_attention_inputs(rows, n_sched, cu, pos, slot, seq_lens)
            att = (S, T, nd, ndt, int(seq_lens.max()) if S else 0, int(n_sched.max()) if S else 0)
        logits_idx = (cu[1:] - 1).astype(np.int64)
        return req_ids, rows, n_sched, cu, S, T, logits_idx, att, tmd

    def _feed_pending_tokens(self, req_ids, rows, cu) -> None:
        """Run-ahead scheduling: requests whose newest token was sampled by the step still in flight read it
        from that step's device result (the host copy of the token arrives later)."""
        ib = self.input_batch
        dst, src = [], []
        for i, rid in enumerate(req_ids):
            j = self._pend_of.get(rid)
            if j is not None and ib.num_computed[rows[i]] == ib.num_tokens[rows[i]]:
                dst.append(int(cu[i]))
                src.append(j)
        if not dst:
            return
        n = len(dst)
        self._to_dev(self._pend_dst, np.asarray(dst, dtype=np.int64))
        self._to_dev(self._pend_src, np.asarray(src, dtype=np.int64))
        vals = self._pend_ids.index_select(0, self._pend_src[:n]).to(self.input_ids.dtype)
        self.input_ids.index_copy_(0, self._pend_dst[:n], vals)

    def _mrope_positions(self, req_ids, rows, n_sched, cu, pos: np.ndarray) -> np.ndarray:
        """Multimodal rotary positions.  Requests with images carry three position rows for their prompt
        and an offset for everything after it; text-only requests and generated tokens have one number
        on all three axes.  Returns the 1-D positions (row 0) and leaves the [3, T] tensor of a step that
        touches image prompts in ``self._step_mrope`` (None otherwise: the step runs the plain rotary path)."""
        ib = self.input_batch
        self._step_mrope = None
        sts = [ib.reqs[r] for r in req_ids]
        if not any(getattr(st, "mrope_positions", None) is not None for st in sts):
            return pos
        pos = pos.copy()
        prompt_parts = []
        for k, st in enumerate(sts):
            mp = getattr(st, "mrope_positions", None)
            if mp is None:
                continue
            a, b = int(cu[k]), int(cu[k + 1])
            idx = int(ib.num_computed[rows[k]]) + np.arange(b - a)
            inside = idx < mp.shape[1]
            pos[a:b][~inside] = idx[~inside] + st.mrope_delta
            if inside.any():
                prompt_parts.append((a, inside, mp[:, idx[inside]]))
        if prompt_parts:
            p3 = np.broadcast_to(pos, (3, len(pos))).copy()
            for a, inside, vals in prompt_parts:
                p3[:, a:a + len(inside)][:, inside] = vals
            self._step_mrope = torch.from_numpy(p3).to(self.device)
            pos = p3[0].copy()
        return pos

    def _prepare_token_parallel(self, so, req_ids, rows, n_sched, req_idx, tok_rows, pos):
        """Token parallelism: the root feeds all T rows through the dense layers, but every rank
        (root included) attends only for the requests it owns.  Rows are grouped by owner with an
        explicit permutation; the local request list keeps the batch order (decode-form first)."""
        ib = self.input_batch
        alloc = so.tknp
        if alloc is None:
            raise RuntimeError("token parallelism is enabled but the scheduler sent no rank assignment")
        is_root = self.tknp_rank == 0
        owner = np.fromiter((alloc.req_to_rank[r] for r in req_ids), dtype=np.int64, count=len(req_ids))
        tok_owner = owner[req_idx]
        per_rank = np.bincount(tok_owner, minlength=self.tknp_size).astype(np.int64)
        T = len(tok_owner)
        sel = owner == self.tknp_rank
        rows_l, n_l = rows[sel], n_sched[sel]
        cu_l, _, pos_l, _, slot_l, seq_l = self._flat_arrays(rows_l, n_l)
        nd, ndt = self._upload_attention_inputs(rows_l, n_l, cu_l, pos_l, slot_l, seq_l)
        S_l, T_l = len(rows_l), int(cu_l[-1])
        perm_t = inv_t = None
        if is_root:
            self._to_dev(self.input_ids, ib.token_ids[tok_rows, pos])
        att = (S_l, T_l, nd, ndt, int(seq_l.max()) if S_l else 0, int(n_l.max()) if S_l else 0)
        if self.tk_rx is not None:
            # peer-memory form: index lists and counts go into the persistent device buffers
            gbs = self._tknp_graph_bs(len(req_ids), n_sched)
            self._tknp_step_bs = gbs
            cap = gbs if gbs is not None else max(T, 1)
            goff = np.full(self.tknp_size + 1, T_l, dtype=np.int32)
            goff[0] = 0
            self._to_dev(self.tk_goff, goff)
            if is_root:
                perm = np.argsort(tok_owner, kind="stable")
                self._to_dev(self.tk_sorted, perm.astype(np.int32))
                off = np.zeros(self.tknp_size + 1, dtype=np.int32)
                off[1:] = np.cumsum(per_rank)
                self._to_dev(self.tk_offsets, off)
                local_idx = np.empty(T, dtype=np.int64)
                local_idx[perm] = np.arange(T) - off[:-1][tok_owner[perm]]
                back = tok_owner * cap + local_idx
                if gbs is not None and gbs > T:
                    back = np.concatenate([back, np.zeros(gbs - T, dtype=np.int64)])
                self._to_dev(self.tk_back, back)
            n_tot = gbs if gbs is not None else T
            tmd = TokenParallelMetadata(
                tokens_per_rank=per_rank.tolist(), num_local_tokens=gbs if gbs is not None else T_l,
                num_total_tokens=n_tot, is_root=is_root, exchange=self.tk_rx, sorted_ids=self.tk_sorted,
                offsets=self.tk_offsets, gather_offsets=self.tk_goff, iota=self.tk_iota, back=self.tk_back[:n_tot],
                cap=cap, dummy=self.tk_dummy)
            return tmd, att
        if is_root:
            perm = np.argsort(tok_owner, kind="stable")
            if not np.array_equal(perm, np.arange(T)):
                inv = np.empty_like(perm)
                inv[perm] = np.arange(T)
                perm_t = torch.from_numpy(perm).to(self.device, non_blocking=True)
                inv_t = torch.from_numpy(inv).to(self.device, non_blocking=True)
        tmd = TokenParallelMetadata(tokens_per_rank=per_rank.tolist(), perm=perm_t, inv_perm=inv_t,
                                    num_local_tokens=T_l, num_total_tokens=T, is_root=is_root)
        return tmd, att

    # ---- multimodal -----------------------------------------------------------------------------------
    def _run_mm_encoder(self, so: SchedulerOutput) -> None:
        """Run the vision encoder for the items scheduled this step; keep outputs until the scheduler frees them."""
        ib = self.input_batch
        for rid in so.finished_req_ids:
            self.encoder_cache.pop(rid, None)
        for rid, i in so.free_encoder_input_ids:
            self.encoder_cache.get(rid, {}).pop(i, None)
        if not so.scheduled_encoder_inputs:
            return
        todo = [(rid, i) for rid, idxs in so.scheduled_encoder_inputs.items() for i in idxs]
        if hasattr(self.model, "encode_mm_items"):
            # items of different shapes (patch matrices with their grids)
            emb = self.model.encode_mm_items([[torch.as_tensor(x).to(self.device) for x in ib.reqs[rid].mm_inputs[i]]
                                              for rid, i in todo])
        else:
            px = torch.stack([ib.reqs[rid].mm_inputs[i] for rid, i in todo]).to(self.device, self.dtype)
            emb = self.model.encode_images(px)
        for (rid, i), e in zip(todo, emb):
            self.encoder_cache.setdefault(rid, {})[i] = e

    def _gather_mm_embeddings(self, req_ids, rows, n_sched, cu):
        """Embeddings and the token mask of every placeholder position covered by this step."""
        ib = self.input_batch
        parts, idx = [], []
        for k, rid in enumerate(req_ids):
            st = ib.reqs[rid]
            if not st.mm_positions:
                continue
            a = int(ib.num_computed[rows[k]])
            b = a + int(n_sched[k])
            for i, pos in enumerate(st.mm_positions):
                lo, hi = max(pos.offset, a), min(pos.offset + pos.length, b)
                if lo >= hi:
                    continue
                e = self.encoder_cache.get(rid, {}).get(i)
                if e is None:
                    raise RuntimeError(f"encoder output {i} of request {rid} is missing")
                parts.append(e[lo - pos.offset: hi - pos.offset])
                base = int(cu[k]) - a
                idx.extend(range(base + lo, base + hi))
        if not parts:
            return None, None
        mask = torch.zeros(int(cu[-1]), dtype=torch.bool)
        mask[idx] = True
        return torch.cat(parts, dim=0), mask.to(self.device)

    def _decode_splits(self, num_seqs: int, max_seq_len: int, fixed: bool) -> int:
        if num_seqs == 0:
            return 1
        if fixed:   # captured graphs: a function of the batch size only
            max_seq_len = self.max_model_len
        return ops.decode_num_splits(num_seqs, self.num_kv_heads, max_seq_len)

    def _workspace(self, tokens: int, nsplit: int):
        if nsplit <= 1 or not self.is_gpu:
            return None
        need = tokens * self.num_heads * nsplit
        ws = getattr(self, "_ws", None)
        if ws is None or ws[2] < need:
            po = torch.empty(need * self.head_size, dtype=torch.float32, device=self.device)
            pl = torch.empty(need, dtype=torch.float32, device=self.device)
            self._ws = ws = (po, pl, need)
        return ws[0], ws[1]

    def _build_attn_metadata(self, S, T, nd, ndt, max_seq, max_q, nsplit, decode_only=False) -> AttentionMetadata:
        local = {}
        if self.local_chunk:
            # decode_only (captured graphs, warm-up): the virtual sequences are the S sequences themselves
            Sv, n_pre = (S, 0) if decode_only else self._l_counts
            local = dict(local_block_table=self.l_block_table[:Sv], local_seq_lens=self.l_seq_lens[:Sv],
                         local_query_start_loc=self.l_qsl[:Sv + 1], local_num_prefill_seqs=n_pre,
                         local_prefill_query_start_loc=self.l_prefill_qsl[:n_pre + 1] if n_pre else None)
            if self.dual_chunk:
                inter = None if decode_only else self._d_inter_bt
                local.update(dca_succ_block_table=self.d_succ_bt[:Sv], dca_succ_lens=self.d_succ_lens[:Sv],
                             dca_inter_block_table=inter if inter is not None else self.block_table[:Sv],
                             dca_inter_lens=self.d_inter_lens[:Sv], dca_token_scale=self.d_scale,
                             dca_passes=3 if decode_only else self._d_passes)
        return AttentionMetadata(**local,
            num_actual_tokens=T, num_decode_seqs=nd, num_decode_tokens=ndt, num_prefill_seqs=S - nd,
            num_prefill_tokens=T - ndt, slot_mapping=self.slot_mapping, block_table=self.block_table[:S],
            seq_lens=self.seq_lens[:S], query_start_loc=self.query_start_loc[:S + 1],
            prefill_query_start_loc=self.prefill_qsl[:S - nd + 1] if nd < S else None,
            max_query_len=max_q, max_seq_len=max_seq, max_decode_query_len=self.max_decode_q,
            decode_num_splits=nsplit, decode_workspace=self._workspace(max(ndt, 1), nsplit),
            state_table=self.state_table[:S] if self.state_table is not None else None)

    def _token_type_ids(self, req_ids, n_sched, T: int) -> Optional[torch.Tensor]:
        """Segment ids of the step's tokens (requests that carry none are all zero); None when no request has any."""
        ib = self.input_batch
        if not any(getattr(ib.reqs[r].pooling_params, "token_type_ids", None) for r in req_ids):
            return None
        tt = np.zeros(T, dtype=np.int64)
        o = 0
        for r, n in zip(req_ids, n_sched):
            ids = getattr(ib.reqs[r].pooling_params, "token_type_ids", None)
            if ids:
                a = int(ib.num_computed[ib.reqs[r].row])
                tt[o:o + int(n)] = np.asarray(ids[a:a + int(n)], dtype=np.int64)
            o += int(n)
        return torch.from_numpy(tt).to(self.device)

    def _cascade_prefix_len(self, so: SchedulerOutput, rows, n_sched, S: int) -> int:
        """Number of leading keys shared by every sequence of this step that is worth a shared attention
        pass (0: attend the ordinary way).  Reference: gpu_model_runner.py (_compute_cascade_attn_prefix_len)."""
        if self.vllm_config.model_config.disable_cascade_attn or S < envs.VDA_CASCADE_MIN_SEQS \
                or not so.num_common_prefix_blocks or self.tknp_size > 1:
            return 0
        blocks = int(so.num_common_prefix_blocks[0])
        if blocks * self.block_size < envs.VDA_CASCADE_MIN_PREFIX:
            return 0
        if getattr(self, "_cascade_ok", None) is None:
            from ..layers.attention import Attention as _A
            self._cascade_ok = all(isinstance(m, _A) and m.alibi_slopes is None for m in self.attn_layers.values()) \
                and len(so.num_common_prefix_blocks) == 1
        if not self._cascade_ok:
            return 0
        # every query token must lie behind the shared keys
        P = min(blocks * self.block_size, int(self.input_batch.num_computed[rows].min()))
        P = P // self.block_size * self.block_size
        return P if P >= envs.VDA_CASCADE_MIN_PREFIX else 0

    def _make_micro_batches(self, rows, n_sched, cu, S, T, nd) -> Optional[list]:
        """Split the step's sequences into two runs of about equal token count (None when one sequence
        dominates).  The first run reuses the step's own device arrays; the second gets its query
        offsets rebased to its first row."""
        if S < 2:
            return None
        k = int(np.argmin(np.abs(cu[1:S].astype(np.int64) - T // 2))) + 1
        if min(int(cu[k]), T - int(cu[k])) * 4 < T:
            return None
        ib = self.input_batch
        seq_lens = (ib.num_computed[rows] + n_sched).astype(np.int64)
        out = []
        for a, b in ((0, k), (k, S)):
            nd_u = min(max(nd - a, 0), b - a)
            t0, t1 = int(cu[a]), int(cu[b])
            ndt_u = int(cu[a + nd_u]) - t0
            if a == 0:
                qsl, pqsl = self.query_start_loc[: b + 1], self.prefill_qsl[: b - nd_u + 1]
            else:
                self._to_dev(self.mb_qsl, (cu[a: b + 1] - cu[a]).astype(np.int32))
                qsl = self.mb_qsl[: b - a + 1]
                p = a + nd_u
                self._to_dev(self.mb_prefill_qsl, (cu[p: b + 1] - cu[p]).astype(np.int32))
                pqsl = self.mb_prefill_qsl[: b - p + 1]
            max_seq = int(seq_lens[a:b].max())
            nsplit = self._decode_splits(nd_u, max_seq, False)
            md = AttentionMetadata(
                num_actual_tokens=t1 - t0, num_decode_seqs=nd_u, num_decode_tokens=ndt_u,
                num_prefill_seqs=b - a - nd_u, num_prefill_tokens=t1 - t0 - ndt_u,
                slot_mapping=self.slot_mapping[t0:], block_table=self.block_table[a:b], seq_lens=self.seq_lens[a:b],
                query_start_loc=qsl, prefill_query_start_loc=pqsl if nd_u < b - a else None,
                max_query_len=int(n_sched[a:b].max()), max_seq_len=max_seq,
                max_decode_query_len=self.max_decode_q, decode_num_splits=nsplit,
                decode_workspace=self._workspace(max(ndt_u, 1), nsplit))
            out.append(MicroBatch(t0, t1, md))
        return out

    # ------------------------------------------------------------------
    @torch.inference_mode()
    def execute_model(self, so: SchedulerOutput, intermediate_tensors: Optional[IntermediateTensors] = None
                      ) -> Union[ModelRunnerOutput, IntermediateTensors]:
        self._update_states(so)
        if not so.total_num_scheduled_tokens:
            return self._finish_kv_only(so)
        self._next_stage()
        req_ids, rows, n_sched, cu, S, T, logits_idx, att, tmd = self._prepare_inputs(so)
        inputs_embeds = None
        encoder_outputs, starting_seqs = (), ()
        if getattr(self.model, "is_multimodal", False):
            self._run_mm_encoder(so)
            if getattr(self.model, "is_encoder_decoder", False):
                # the encoder output feeds the decoder's cross-attention (written to its cache in the step
                # that holds the request's first token), not the token embeddings
                encoder_outputs = [(k, self.encoder_cache[rid][0]) for k, rid in enumerate(req_ids)
                                   if self.input_batch.num_computed[rows[k]] == 0
                                   and 0 in self.encoder_cache.get(rid, {})]
                # every request whose first token runs in this step: models with per-slot book-keeping reset it
                starting_seqs = [k for k in range(len(req_ids)) if self.input_batch.num_computed[rows[k]] == 0]
            else:
                mm_e, mm_mask = self._gather_mm_embeddings(req_ids, rows, n_sched, cu)
                if mm_e is not None:
                    inputs_embeds = self.model.get_input_embeddings(self.input_ids[:T], mm_e, mm_mask)
        S_a, T_a, nd, ndt, max_seq, max_q = att        # what THIS rank attends for (== the batch without TKNP)
        ib = self.input_batch
        pure_decode = nd == S and max_q == 1
        pp = get_pp_group()
        graph_bs = None
        if tmd is not None:
            graph_bs = self._tknp_step_bs if self.tk_rx is not None else None     # decided on the global batch
        elif self.use_graphs and pure_decode and S <= (self.capture_sizes[0] if self.capture_sizes else 0) \
                and self.graphs and self.lora_manager is None and inputs_embeds is None and not starting_seqs:
            # (a one-token decoder prompt looks like a decode step, but the step in which an encoder-decoder request
            # starts writes its cross-attention keys from Python: that is not in any captured graph)
            graph_bs = self.vllm_config.pad_for_cudagraph(S)
        if self.lora_manager is not None:
            self.lora_manager.set_active(req_ids, rows, n_sched, ib)
        conn = self.kv_connector
        if conn is not None:
            conn.bind_connector_metadata(so.kv_connector_metadata)

        if graph_bs is not None:
            Sg = S_a                 # sequences THIS rank attends for (== S without token parallelism)
            if graph_bs > Sg:
                # padded sequences: zero keys, slot -1 (no cache write), one query token each
                self.seq_lens[Sg:graph_bs].zero_()
                self.slot_mapping[Sg:graph_bs].fill_(-1)
                if self.state_table is not None:
                    self.state_table[Sg:graph_bs].zero_()
                self.query_start_loc[Sg + 1: graph_bs + 1].copy_(
                    torch.arange(Sg + 1, graph_bs + 1, dtype=torch.int32, device=self.device))
                if self.local_chunk:
                    self.l_seq_lens[Sg:graph_bs].zero_()
                    self.l_qsl[Sg + 1: graph_bs + 1].copy_(self.query_start_loc[Sg + 1: graph_bs + 1])
                    if self.dual_chunk:
                        self.d_succ_lens[Sg:graph_bs].zero_()
                        self.d_inter_lens[Sg:graph_bs].zero_()
            g, hidden_buf, md = self.graphs[graph_bs]
            if not pp.is_first_rank and intermediate_tensors is not None:
                # the graph reads the previous stage's activations from the buffers it was captured with
                for k, buf in self._capture_inter(graph_bs).tensors.items():
                    src = intermediate_tensors.tensors.get(k)
                    if src is not None and src.numel():
                        n_rows = min(src.shape[0], graph_bs)
                        buf[:n_rows].copy_(src[:n_rows])
            with set_forward_context(md, kv_connector=conn, num_tokens=T):
                if conn is not None:
                    conn.start_load_kv()
                g.replay()
            hidden = hidden_buf
        else:
            nsplit = self._decode_splits(nd, max_seq, False)
            md = self._build_attn_metadata(S_a, T_a, nd, ndt, max_seq, max_q, nsplit)
            P = self._cascade_prefix_len(so, rows, n_sched, S) if tmd is None else 0
            if P:
                self._to_dev(self.cascade_seq_len, np.array([P], dtype=np.int32))
                self._to_dev(self.cascade_qsl, np.array([0, T], dtype=np.int32))
                self._to_dev(self.cascade_suffix_lens, (ib.num_computed[rows] + n_sched - P).astype(np.int32))
                md.cascade_prefix_len = P
                md.cascade_seq_len, md.cascade_query_start_loc = self.cascade_seq_len, self.cascade_qsl
                md.cascade_suffix_seq_lens = self.cascade_suffix_lens[:S]
            inter = None
            if not pp.is_first_rank:
                inter = intermediate_tensors
            T_in, P_in = T, T_a
            tp = self.sp_multiple
            if tp > 1 and tmd is None and inputs_embeds is None and T % tp and T + tp <= self.input_ids.shape[0] \
                    and not self.uses_mrope:
                # sequence parallelism splits the rows over the TP ranks: pad to a multiple of tp with
                # rows that attention ignores (they lie beyond num_actual_tokens)
                T_in = P_in = (T + tp - 1) // tp * tp
                self.input_ids[T:T_in].zero_()
                self.positions[T:T_in].zero_()
            mbs = None
            if self.async_tp and T >= envs.VDA_ASYNC_TP_MIN_TOKENS and conn is None and self.lora_manager is None \
                    and inputs_embeds is None and T_in == T and not P and not self.local_chunk:
                mbs = self._make_micro_batches(rows, n_sched, cu, S, T, nd)
            with set_forward_context(md, tknp=tmd, kv_connector=conn, num_tokens=T, micro_batches=mbs,
                                     dp=self._dp_metadata(T_in)) as fctx:
                if encoder_outputs:
                    fctx.extra["encoder_outputs"] = encoder_outputs
                if starting_seqs:
                    fctx.extra["starting_seqs"] = starting_seqs
                if conn is not None:
                    conn.start_load_kv()
                extra = {}
                if getattr(self.model, "uses_token_type_ids", False):
                    extra["token_type_ids"] = self._token_type_ids(req_ids, n_sched, T_in)
                positions = self.positions[:P_in]
                if self.uses_mrope and getattr(self, "_step_mrope", None) is not None and P_in == T:
                    positions = self._step_mrope
                hidden = self.model(self.input_ids[:T_in], positions, inter, inputs_embeds, **extra)
            if T_in > T and torch.is_tensor(hidden):
                hidden = hidden[:T]
        if conn is not None:
            conn.wait_for_save()
            conn.clear_connector_metadata()
        if self.lora_manager is not None:
            self.lora_manager.clear_active()       # warm-up runs are not adapted; the head names its rows
        if self.eplb_state is not None:
            self._eplb_step()
        tk_plain = tmd is not None and self._tknp_plain_step(so, rows, conn)
        if tmd is not None and not tmd.is_root:
            # the root samples; everybody applies the same tokens to the persistent batch
            if not pp.is_last_rank:
                return EMPTY_MODEL_RUNNER_OUTPUT
            if tk_plain:
                ids = self._tknp_share_sampled(None, S)
                self._check_comm_errors()
                done = (ib.num_computed[rows] + n_sched) >= ib.num_tokens[rows]
                sampled_lists = [[t] if done[i] else [] for i, t in enumerate(ids)]
                self._apply_sampled(req_ids, sampled_lists)
                self._step += 1
                return ModelRunnerOutput(
                    req_ids=req_ids, req_id_to_index={r: i for i, r in enumerate(req_ids)},
                    sampled_token_ids=sampled_lists, spec_token_ids=None, logprobs=None, prompt_logprobs_dict={},
                    finished_sending=None, finished_recving=None, num_nans_in_logits=None)
            out = get_tknp_group().broadcast_object(None, src=0)
            self._apply_sampled(out.req_ids, out.sampled_token_ids)
            return out
        if not pp.is_last_rank:
            return hidden
        finished_sending, finished_recving = (conn.get_finished(so.finished_req_ids) if conn is not None
                                              else (None, None))
        failed_recving = (conn.failed_recving() or None) if conn is not None else None

        if getattr(self.model, "is_pooling_model", False):
            return self._pool(so, hidden, req_ids, rows, n_sched, cu, finished_sending, finished_recving)

        # ---- logits + sampling ------------------------------------------------
        spec_sched = so.scheduled_spec_decode_tokens
        if spec_sched and self.rejection_sampler is not None:
            out = self._sample_with_spec(so, hidden, req_ids, rows, n_sched, cu, S)
        else:
            self._to_dev(self.logits_indices, logits_idx)
            sample_hidden = hidden[self.logits_indices[:S]]
            logits = self._compute_logits(sample_hidden, logits_idx[:S])
            if envs.VLLM_COMPUTE_NANS_IN_LOGITS:
                self._last_nans = {r: int(x) for r, x in zip(req_ids, logits.isnan().sum(-1).tolist())}
            if so.grammar_bitmask is not None:
                self._apply_grammar_bitmask(so, logits, req_ids)
            # index of the token being sampled, from the scheduler's counters (the host-side token count lags
            # by one under run-ahead scheduling)
            self._sample_pos = (ib.num_computed[rows] + n_sched).astype(np.int64)
            smd = self._sampling_metadata(req_ids, rows)
            self._sample_pos = None
            sout = self.sampler(logits, smd)
            sampled = sout.sampled_token_ids[:, 0]
            out = (sampled, sout.logprobs_tensors, None)
        sampled_t, logprobs_t, multi = out
        prompt_lp = self._prompt_logprobs(so, hidden, req_ids, rows, n_sched, cu)

        if tk_plain:
            self._tknp_share_sampled(sampled_t, S)          # device to device, queued before the host sync below
        done = (ib.num_computed[rows] + n_sched) >= ib.num_tokens[rows]
        if so.defer_output and multi is None and tmd is None and self.drafter is None and not prompt_lp \
                and conn is None:
            return self._defer_output(req_ids, done, sampled_t, logprobs_t, S)
        if self._deferred_open is not None:
            self._deferred_open.get()        # the previous step's tokens enter the host arrays before this step's
        self._pend_of, self._pend_ids = {}, None
        # the one host sync of the step
        if multi is not None:
            sampled_lists = multi
        else:
            sampled_lists = [[t] for t in sampled_t.tolist()]
        self._check_comm_errors()            # the stream is drained: a timed-out collective has reported by now
        # requests still in the middle of their prompt discard the sample
        for i in range(len(req_ids)):
            if not done[i]:
                sampled_lists[i] = []
        self._apply_sampled(req_ids, sampled_lists)
        spec_tokens = None
        if self.drafter is not None:
            spec_tokens = self.drafter.propose(self, req_ids, rows, sampled_lists, hidden, cu, n_sched, so)
        lp_lists = Sampler.to_lists(logprobs_t)
        self._step += 1
        result = ModelRunnerOutput(
            req_ids=req_ids, req_id_to_index={r: i for i, r in enumerate(req_ids)},
            sampled_token_ids=sampled_lists, spec_token_ids=spec_tokens, logprobs=lp_lists,
            prompt_logprobs_dict=prompt_lp, finished_sending=finished_sending, finished_recving=finished_recving,
            failed_recving=failed_recving,
            num_nans_in_logits=getattr(self, "_last_nans", None))
        if tmd is not None and not tk_plain:
            get_tknp_group().broadcast_object(result, src=0)
        return result

    def _defer_output(self, req_ids, done, sampled_t, logprobs_t, S) -> "DeferredOutput":
        """End of a step whose results the engine collects later: the ids start their way to a pinned host buffer,
        the next step (scheduled before they arrive) takes them from ``sampled_t`` on the device."""
        if self._deferred_bufs is None:
            self._deferred_bufs = [torch.empty(self.max_num_reqs, dtype=torch.int64, pin_memory=self.pin)
                                   for _ in range(3)]
        buf = self._deferred_bufs[self._step % 3]
        buf[:S].copy_(sampled_t, non_blocking=True)
        ev = None
        if self.is_gpu:
            ev = torch.cuda.Event()
            ev.record()
        self._pend_gen += 1
        self._pend_ids = sampled_t
        self._pend_of = {rid: i for i, rid in enumerate(req_ids) if done[i]}
        self._step += 1
        if self._deferred_open is not None:
            prev = self._deferred_open
            out = DeferredOutput(self, self._pend_gen, req_ids, done, buf, S, ev, logprobs_t)
            out.prev = prev                  # collected first (order of the host-side appends)
        else:
            out = DeferredOutput(self, self._pend_gen, req_ids, done, buf, S, ev, logprobs_t)
        self._deferred_open = out
        return out

    def _check_comm_errors(self) -> None:
        if not self.is_gpu:
            return
        get_tp_group().check_comm_errors()
        if self.tk_rx is not None and self.tk_rx.check_error():
            raise RuntimeError("token-parallel row exchange timed out waiting for a peer; the step's results are invalid")
        from ..parallel.all2all import check_dispatcher_errors
        check_dispatcher_errors()

    def _tknp_plain_step(self, so, rows, conn) -> bool:
        """True when the step's result is just one sampled id per request (no log-probabilities, drafts,
        transfer bookkeeping): then the root's ids reach the other token-parallel ranks as one int32 row
        through the peer-memory exchange instead of a pickled object over gloo.  Every rank evaluates this
        on the same scheduler output and the same mirrored batch, so all take the same branch."""
        ib = self.input_batch
        return (self.tk_rx is not None and not so.scheduled_spec_decode_tokens and self.drafter is None
                and conn is None and not envs.VLLM_COMPUTE_NANS_IN_LOGITS
                and not getattr(self.model, "is_pooling_model", False)
                and not bool((ib.num_logprobs[rows] >= 0).any()) and not bool((ib.num_prompt_logprobs[rows] >= 0).any()))

    def _tknp_share_sampled(self, sampled: Optional[torch.Tensor], S: int) -> Optional[list[int]]:
        """Root: send the S sampled ids to every rank.  Others: receive them (one host sync)."""
        rx = self.tk_rx
        if sampled is not None:
            self.tk_ids_row[0, :S] = sampled.to(torch.int32)
            rx.exchange(self.tk_ids_row, self.tk_row0, self.tk_bcast_off, 1, None, 1)
            return None
        recv, _, _ = rx.exchange(self.tk_ids_row, self.tk_row0, self.tk_zero_off, 1, None, 1)
        return recv[0, :S].tolist()

    def _pool(self, so, hidden, req_ids, rows, n_sched, cu, fs, fr) -> ModelRunnerOutput:
        """Pooling models: one vector (or score) per request once its whole prompt has been run."""
        ib = self.input_batch
        done = (ib.num_computed[rows] + n_sched) >= ib.num_tokens[rows]
        whole = self.model.pooler.needs_whole_prompt
        outs = []
        for i, rid in enumerate(req_ids):
            if not done[i]:
                outs.append(None)
                continue
            if whole and ib.num_computed[rows[i]] > 0:
                raise RuntimeError("mean/cls pooling needs the whole prompt in one step: disable chunked prefill "
                                   "and prefix caching for this model")
            h = hidden[int(cu[i]): int(cu[i + 1])]
            outs.append(self.model.pool(h, ib.reqs[rid].pooling_params).cpu())
        return ModelRunnerOutput(req_ids=req_ids, req_id_to_index={r: i for i, r in enumerate(req_ids)},
                                 sampled_token_ids=[[] for _ in req_ids], pooler_output=outs,
                                 finished_sending=fs, finished_recving=fr)

    def _apply_sampled(self, req_ids: list[str], sampled_lists: list[list[int]]) -> None:
        ib = self.input_batch
        for rid, toks in zip(req_ids, sampled_lists):
            if not toks:
                continue
            st = ib.reqs.get(rid)
            if st is None:
                continue             # finished or aborted while its step was in flight
            for tok in toks:
                ib.append_token(st.row, tok)
                st.output_token_ids.append(tok)

    def _finish_kv_only(self, so: SchedulerOutput) -> ModelRunnerOutput:
        """No tokens scheduled: still drive the KV connector (async transfers)."""
        conn = self.kv_connector
        if conn is None:
            return EMPTY_MODEL_RUNNER_OUTPUT
        conn.bind_connector_metadata(so.kv_connector_metadata)
        conn.start_load_kv()
        fs, fr = conn.get_finished(so.finished_req_ids)
        failed = conn.failed_recving() or None
        conn.clear_connector_metadata()
        if not fs and not fr:
            return EMPTY_MODEL_RUNNER_OUTPUT
        return ModelRunnerOutput(finished_sending=fs, finished_recving=fr, failed_recving=failed)

    # ------------------------------------------------------------------
    def _sampling_metadata(self, req_ids: list[str], rows: np.ndarray) -> SamplingMetadata:
        ib = self.input_batch
        dev = self.device
        temp = ib.temperature[rows]
        all_greedy = bool((temp < 1e-5).all())
        all_random = bool((temp >= 1e-5).all())
        nlp = ib.num_logprobs[rows]
        max_lp = int(nlp.max()) if (nlp >= 0).any() else None
        md = SamplingMetadata(temperature=None, all_greedy=all_greedy, all_random=all_random, max_num_logprobs=max_lp)

        def dev_t(a):
            t = torch.from_numpy(np.ascontiguousarray(a))
            return (t.pin_memory() if self.pin else t).to(dev, non_blocking=True)

        if not all_greedy:
            md.temperature = dev_t(temp)
            tk, tp, mp = ib.top_k[rows], ib.top_p[rows], ib.min_p[rows]
            if (tk > 0).any():
                md.top_k = dev_t(tk)
            if (tp < 1.0).any():
                md.top_p = dev_t(tp)
            if (mp > 0).any():
                md.min_p = dev_t(mp)
            # per-step seeds: seeded requests are reproducible (seed, position); others draw fresh entropy
            pos = self._sample_pos if self._sample_pos is not None else ib.num_tokens[rows].astype(np.int64)
            fresh = self._rng.integers(0, 2**62, size=len(rows), dtype=np.int64)
            seeds = np.where(ib.has_seed[rows], ib.seed[rows] * np.int64(1000003) + pos, fresh)
            md.seeds = dev_t(seeds)
        sps = [ib.reqs[r].sampling_params for r in req_ids]
        pen = (ib.freq_pen[rows] != 0) | (ib.pres_pen[rows] != 0) | (ib.rep_pen[rows] != 1.0)
        need_out = bool(pen.any())
        for i, sp in enumerate(sps):
            if sp is None:
                continue
            if sp.min_tokens > 0 and sp.all_stop_token_ids:
                md.min_tokens[i] = (sp.min_tokens, set(sp.all_stop_token_ids))
                need_out = True
            if sp.bad_words_token_ids:
                md.bad_words_token_ids[i] = sp.bad_words_token_ids
                need_out = True
        if any(sp is not None and sp.logit_bias for sp in sps):
            md.logit_bias = [sp.logit_bias if sp is not None else None for sp in sps]
        if any(sp is not None and sp.allowed_token_ids for sp in sps):
            mask = torch.zeros((len(sps), self.vocab_size), dtype=torch.bool)
            for i, sp in enumerate(sps):
                if sp is not None and sp.allowed_token_ids:
                    mask[i] = True
                    mask[i, sp.allowed_token_ids] = False
            md.allowed_token_ids_mask = mask.to(dev)
        if need_out:
            md.output_token_ids = [ib.reqs[r].output_token_ids for r in req_ids]
        if pen.any():
            md.no_penalties = False
            md.frequency_penalties, md.presence_penalties, md.repetition_penalties = \
                dev_t(ib.freq_pen[rows]), dev_t(ib.pres_pen[rows]), dev_t(ib.rep_pen[rows])
            Lp = max(ib.reqs[r].num_prompt_tokens for r in req_ids)
            pt = np.full((len(rows), Lp), self.vocab_size, dtype=np.int64)
            for i, r in enumerate(req_ids):
                n = ib.reqs[r].num_prompt_tokens
                pt[i, :n] = ib.token_ids[rows[i], :n]
            md.prompt_token_ids = dev_t(pt)
        return md

    def _prompt_logprobs(self, so, hidden, req_ids, rows, n_sched, cu) -> dict[str, Optional[LogprobsLists]]:
        ib = self.input_batch
        out: dict[str, Optional[LogprobsLists]] = {}
        want = ib.num_prompt_logprobs[rows]
        if not (want >= 0).any():
            return out
        for i, rid in enumerate(req_ids):
            k = int(want[i])
            if k < 0:
                continue
            st = ib.reqs[rid]
            row = int(rows[i])
            start = int(ib.num_computed[row])
            n = int(n_sched[i])
            # positions [start, start+n) predict tokens [start+1, start+n]; only prompt targets count
            n_tgt = min(start + n, st.num_prompt_tokens - 1) - start
            if n_tgt <= 0:
                continue
            h = hidden[int(cu[i]): int(cu[i]) + n_tgt]
            lp = torch.log_softmax(self._compute_logits(h, np.arange(int(cu[i]), int(cu[i]) + n_tgt)).float(), dim=-1)
            tgt = torch.from_numpy(ib.token_ids[row, start + 1: start + 1 + n_tgt].astype(np.int64)).to(self.device)
            ids, lps, ranks = Sampler.gather_logprobs(lp, k, tgt)
            out[rid] = LogprobsLists(ids.tolist(), lps.tolist(), ranks.tolist())
        return out

    def _compute_logits(self, hidden_rows: torch.Tensor, index: np.ndarray) -> torch.Tensor:
        """Logits of rows ``index`` of this step's tokens (the index tells LoRA which adapter each row uses)."""
        if self.lora_manager is None:
            return self.model.compute_logits(hidden_rows)
        with self.lora_manager.head_rows(index):
            return self.model.compute_logits(hidden_rows)

    def _apply_grammar_bitmask(self, so, logits, req_ids) -> None:
        bm = so.grammar_bitmask
        idx = so.structured_output_request_ids
        if bm is None or not idx:
            return
        bm_t = torch.as_tensor(bm)
        for i, rid in enumerate(req_ids):
            j = idx.get(rid)
            if j is None:
                continue
            words = bm_t[j].to(self.device).to(torch.int32)
            bits = (words[:, None] >> torch.arange(32, device=self.device, dtype=torch.int32)) & 1
            allowed = bits.reshape(-1)[: logits.shape[-1]].bool()
            logits[i].masked_fill_(~allowed, float("-inf"))

    def _sample_with_spec(self, so, hidden, req_ids, rows, n_sched, cu, S):
        """Verify draft tokens: logits at every scheduled position of requests with drafts."""
        ib = self.input_batch
        drafts = [so.scheduled_spec_decode_tokens.get(r, []) for r in req_ids]
        idx = []
        for i in range(S):
            nd_i = len(drafts[i])
            last = int(cu[i + 1]) - 1
            idx.extend(range(last - nd_i, last + 1))
        sel = torch.tensor(idx, dtype=torch.int64, device=self.device)
        logits = self._compute_logits(hidden[sel], np.asarray(idx)).float()
        smd = self._sampling_metadata(req_ids, rows)
        lists = self.rejection_sampler(logits, drafts, smd, self.sampler)
        return None, None, lists

    # ------------------------------------------------------------------
    # profiling / warm-up / capture
    # ------------------------------------------------------------------
    @torch.inference_mode()
    def _dummy_run(self, num_tokens: int, with_metadata: bool = False, num_seqs: Optional[int] = None):
        """Forward pass on zero inputs.  Without metadata the attention kernels are
        skipped (memory profiling); with metadata a decode batch of ``num_seqs``
        empty sequences runs the full kernel path."""
        pp = get_pp_group()
        inter = None
        if not pp.is_first_rank:
            inter = IntermediateTensors({
                "hidden_states": torch.zeros((num_tokens, self.hidden_size), dtype=self.dtype, device=self.device),
                "residual": torch.zeros((num_tokens, self.hidden_size), dtype=self.dtype, device=self.device)})
        md = None
        if with_metadata:
            S = num_seqs or num_tokens
            self.seq_lens[:S].zero_()
            self.slot_mapping[:num_tokens].fill_(-1)
            self.query_start_loc[:S + 1].copy_(torch.arange(S + 1, dtype=torch.int32, device=self.device))
            nsplit = self._decode_splits(S, 0, True)
            if self.local_chunk:
                self.l_seq_lens[:S].zero_()
                self.l_qsl[:S + 1].copy_(self.query_start_loc[:S + 1])
                if self.dual_chunk:
                    self.d_succ_lens[:S].zero_()
                    self.d_inter_lens[:S].zero_()
            md = self._build_attn_metadata(S, num_tokens, S, num_tokens, 0, 1, nsplit, decode_only=True)
        with set_forward_context(md, num_tokens=0, dp=self._dp_metadata(num_tokens)):
            out = self.model(self.input_ids[:num_tokens], self.positions[:num_tokens], inter, None)
        return out

    def _dp_metadata(self, rows: int):
        """Row counts of every data-parallel replica for this step (one small host all-reduce), for expert layers
        that exchange tokens between replicas; None when nothing needs it.  Every replica makes the same sequence
        of calls: steps run in lock step and idle replicas run dummy batches.
        Reference: vllm/forward_context.py:57-70 (DPMetadata.num_tokens_across_dp)."""
        if not self._dp_moe:
            return None
        import torch.distributed as dist
        from ..forward_context import DPMetadata
        from ..parallel.state import get_dp_group
        g = get_dp_group()
        t = torch.zeros(g.world_size, dtype=torch.int32)
        t[g.rank_in_group] = int(rows)
        dist.all_reduce(t, group=g.cpu_group)
        return DPMetadata([int(v) for v in t.tolist()])

    @torch.inference_mode()
    def profile_run(self) -> None:
        hidden = self._dummy_run(self.max_num_tokens)
        if get_pp_group().is_last_rank and hidden is not None and hasattr(self.model, "compute_logits"):
            n = min(self.max_num_reqs, self.max_num_tokens)
            logits = self.model.compute_logits(hidden[:n])
            self.sampler(logits, SamplingMetadata(None, True, False))
            del logits
        if self.is_gpu:
            torch.cuda.synchronize()
        del hidden
        gc.collect()

    @torch.inference_mode()
    def capture_model(self) -> None:
        if not self.use_graphs or not self.capture_sizes:
            return
        t0 = time.time()
        free0 = torch.cuda.mem_get_info(self.device)[0]
        global _CAPTURE_STATE_ANCHOR
        if _CAPTURE_STATE_ANCHOR is None:
            with torch.inference_mode(False):
                # The default generator creates its capture-state tensors when the first graph registers
                # with it and drops them when the last one is destroyed; every capture updates them in
                # place.  Created inside inference mode they would make a capture outside it (user code
                # sharing the process) fail, so a tiny graph captured here, outside inference mode, is
                # kept for the life of the process: the tensors are made once and stay ordinary ones.
                probe = torch.zeros(1, device=self.device)
                anchor = torch.cuda.CUDAGraph()
                with torch.cuda.graph(anchor):
                    probe.add_(1)
                _CAPTURE_STATE_ANCHOR = (anchor, probe)
        self.graph_pool = torch.cuda.graph_pool_handle()
        if self.tk_rx is not None:
            # nothing moves during warm-up and capture: every rank sends zero rows
            for t in (self.tk_offsets, self.tk_goff, self.tk_back):
                t.zero_()
        with graph_capture(self.device) as cap:
            for bs in sorted(self.capture_sizes, reverse=True):
                nsplit = self._decode_splits(bs, 0, True)
                self.seq_lens[:bs].zero_()
                self.slot_mapping[:bs].fill_(-1)
                self.query_start_loc[:bs + 1].copy_(torch.arange(bs + 1, dtype=torch.int32, device=self.device))
                if self.local_chunk:
                    self.l_seq_lens[:bs].zero_()
                    self.l_qsl[:bs + 1].copy_(self.query_start_loc[:bs + 1])
                    if self.dual_chunk:
                        self.d_succ_lens[:bs].zero_()
                        self.d_inter_lens[:bs].zero_()
                md = self._build_attn_metadata(bs, bs, bs, bs, 0, 1, nsplit, decode_only=True)
                tmd = None
                if self.tk_rx is not None:
                    # token parallelism: bs rows on the root, up to bs local rows on every rank; counts and
                    # index lists are read from the persistent buffers at replay
                    tmd = TokenParallelMetadata(
                        tokens_per_rank=[], num_local_tokens=bs, num_total_tokens=bs, is_root=self.tknp_rank == 0,
                        exchange=self.tk_rx, sorted_ids=self.tk_sorted, offsets=self.tk_offsets,
                        gather_offsets=self.tk_goff, iota=self.tk_iota, back=self.tk_back[:bs], cap=bs,
                        dummy=self.tk_dummy)
                for _ in range(self.vllm_config.compilation_config.cudagraph_num_of_warmups):
                    with set_forward_context(md, tknp=tmd):
                        self.model(self.input_ids[:bs], self.positions[:bs], self._capture_inter(bs), None)
                g = torch.cuda.CUDAGraph()
                with set_forward_context(md, tknp=tmd):
                    with torch.cuda.graph(g, pool=self.graph_pool, stream=cap.stream):
                        out = self.model(self.input_ids[:bs], self.positions[:bs], self._capture_inter(bs), None)
                self.graphs[bs] = (g, out, md)
        torch.cuda.synchronize()
        used = free0 - torch.cuda.mem_get_info(self.device)[0]
        logger.info("captured %d decode graphs in %.1f s (%.2f GiB)", len(self.graphs), time.time() - t0,
                    used / 2**30)

    def _capture_inter(self, bs: int):
        if get_pp_group().is_first_rank:
            return None
        if self.intermediate is None:
            n = max(self.capture_sizes)
            self.intermediate = IntermediateTensors({
                "hidden_states": torch.zeros((n, self.hidden_size), dtype=self.dtype, device=self.device),
                "residual": torch.zeros((n, self.hidden_size), dtype=self.dtype, device=self.device)})
        return self.intermediate[:bs]
