"""Speculative decoding proposers: n-gram prompt lookup (native matcher), Medusa heads, EAGLE.

A proposer is asked once per step, after sampling, for up to k draft tokens per request; the
scheduler schedules them next step and the rejection sampler verifies them against the target.
Reference: vllm/v1/spec_decode/{ngram_proposer,medusa,eagle}.py, gpu_model_runner.py:1927-2190.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from ..config import VllmConfig
from ..forward_context import set_forward_context
from ..logger import init_logger

logger = init_logger(__name__)


def accepted_bounds(n_sched: np.ndarray, drafts: list[int], sampled_lists: list[list[int]]) -> np.ndarray:
    """Per request: number of scheduled positions whose *input* token turned out correct
    (the real tokens plus the accepted drafts)."""
    v = np.empty(len(n_sched), dtype=np.int64)
    for i, n in enumerate(n_sched):
        a = len(sampled_lists[i])
        v[i] = n - drafts[i] + (a - 1) if a > 0 else n
    return v


class NgramProposer:
    def __init__(self, vllm_config: VllmConfig):
        sc = vllm_config.speculative_config
        self.k = sc.num_speculative_tokens
        self.min_n, self.max_n = sc.prompt_lookup_min, sc.prompt_lookup_max
        self.max_model_len = vllm_config.model_config.max_model_len
        from ..runtime.shm_ring import host_library
        self.lib = host_library()
        self.lib.vda_ngram_propose_batch.restype = None
        self.lib.vda_ngram_propose_batch.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                     ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                     ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]

    def propose(self, runner, req_ids, rows, sampled_lists, hidden, cu, n_sched=None, so=None) -> list[list[int]]:
        ib = runner.input_batch
        S = len(req_ids)
        rows64 = np.ascontiguousarray(rows, dtype=np.int64)
        lens = ib.num_tokens[rows].astype(np.int64)
        do = np.fromiter((1 if (sampled_lists[i] and lens[i] + self.k < self.max_model_len) else 0 for i in range(S)),
                         dtype=np.uint8, count=S)
        out = np.zeros((S, self.k), dtype=np.int32)
        counts = np.zeros(S, dtype=np.int32)
        toks = ib.token_ids
        assert toks.flags.c_contiguous and toks.dtype == np.int32
        self.lib.vda_ngram_propose_batch(toks.ctypes.data, toks.shape[1], rows64.ctypes.data, lens.ctypes.data,
                                         do.ctypes.data, S, self.min_n, self.max_n, self.k, out.ctypes.data,
                                         counts.ctypes.data)
        return [out[i, :counts[i]].tolist() for i in range(S)]


def _load_draft(vllm_config: VllmConfig, runner, archs_default: str):
    """Build the draft network from ``speculative_config.model`` on the runner's device."""
    import copy
    from ..config import load_hf_config
    from ..models.loader import get_model
    sc = vllm_config.speculative_config
    cfg = copy.copy(vllm_config)
    cfg.model_config = copy.copy(vllm_config.model_config)
    cfg.model_config.model = sc.model
    hf = load_hf_config(sc.model)
    cfg.model_config.hf_config = hf
    archs = list(getattr(hf, "architectures", None) or [archs_default])
    # checkpoints that carry their own multi-token-prediction layers draft with those
    mtp = {"DeepseekV3ForCausalLM": "DeepSeekMTPModel", "MiMoForCausalLM": "MiMoMTPModel"}
    if archs and archs[0] in mtp and int(getattr(hf, "num_nextn_predict_layers", 0) or 0) > 0:
        archs = [mtp[archs[0]]]
    elif archs_default == "EagleLlamaForCausalLM" and not any("Eagle" in a or "MTP" in a for a in archs):
        archs = [archs_default]
    hf.architectures = archs
    cfg.model_config.quantization = None
    cfg.speculative_config = None
    cfg.lora_config = None
    return get_model(cfg, runner.device), hf


class MedusaProposer:
    """k independent heads on the target's last hidden state; head j guesses token t+1+j."""

    def __init__(self, vllm_config: VllmConfig, runner):
        self.k = vllm_config.speculative_config.num_speculative_tokens
        self.model, _ = _load_draft(vllm_config, runner, "MedusaModel")
        self.max_model_len = vllm_config.model_config.max_model_len

    @torch.inference_mode()
    def propose(self, runner, req_ids, rows, sampled_lists, hidden, cu, n_sched=None, so=None) -> list[list[int]]:
        S = len(req_ids)
        drafts = [len(so.scheduled_spec_decode_tokens.get(r, ())) for r in req_ids]
        v = accepted_bounds(n_sched, drafts, sampled_lists)
        idx = torch.from_numpy(cu[:-1].astype(np.int64) + v - 1).to(hidden.device)
        logits = self.model.compute_logits(self.model(hidden[idx]))      # list of [S, V]
        toks = torch.stack([lg.argmax(dim=-1) for lg in logits[: self.k]], dim=1).tolist()
        lens = runner.input_batch.num_tokens[rows]
        return [toks[i] if (sampled_lists[i] and lens[i] + self.k < self.max_model_len) else [] for i in range(S)]


class MLPSpeculatorProposer:
    """Stage-wise MLP speculator on the target's last hidden state and the token just sampled."""

    def __init__(self, vllm_config: VllmConfig, runner):
        self.k = vllm_config.speculative_config.num_speculative_tokens
        self.model, _ = _load_draft(vllm_config, runner, "MLPSpeculatorPreTrainedModel")
        self.max_model_len = vllm_config.model_config.max_model_len

    @torch.inference_mode()
    def propose(self, runner, req_ids, rows, sampled_lists, hidden, cu, n_sched=None, so=None) -> list[list[int]]:
        S = len(req_ids)
        drafts = [len(so.scheduled_spec_decode_tokens.get(r, ())) for r in req_ids]
        v = accepted_bounds(n_sched, drafts, sampled_lists)
        want = [i for i in range(S) if sampled_lists[i]
                and runner.input_batch.num_tokens[rows[i]] + self.k < self.max_model_len]
        out: list[list[int]] = [[] for _ in range(S)]
        if not want:
            return out
        idx = torch.from_numpy((cu[:-1].astype(np.int64) + v - 1)[want]).to(hidden.device)
        last = torch.tensor([sampled_lists[i][-1] for i in want], dtype=torch.int64, device=hidden.device)
        toks = self.model.propose(last, hidden.index_select(0, idx), self.k).tolist()
        for n, i in enumerate(want):
            out[i] = toks[n]
        return out


class EagleProposer:
    """One-layer draft transformer fed with the target's hidden states (EAGLE).  The draft layer
    keeps its own KV cache in the same blocks as the target (the scheduler reserves k look-ahead
    slots), so its attention runs through the regular paged kernels."""

    def __init__(self, vllm_config: VllmConfig, runner):
        self.k = vllm_config.speculative_config.num_speculative_tokens
        self.max_model_len = vllm_config.model_config.max_model_len
        self.block_size = vllm_config.cache_config.block_size
        self.model, _ = _load_draft(vllm_config, runner, "EagleLlamaForCausalLM")
        target = runner.model
        # EAGLE checkpoints reuse the target's embedding and output head
        if getattr(self.model, "share_embeddings", True) and hasattr(target, "model"):
            self.model.set_shared(target.model.embed_tokens, target.lm_head, target.logits_processor)

    def attention_layers(self) -> dict:
        from ..layers.attention import Attention, MLAAttention
        return {f"draft.{n}": m for n, m in self.model.named_modules() if isinstance(m, (Attention, MLAAttention))}

    def _run(self, runner, ids: np.ndarray, hidden: torch.Tensor, rows, n_q: np.ndarray, start: np.ndarray,
             decode: bool) -> torch.Tensor:
        """One draft forward over ``n_q[i]`` tokens of request rows[i] beginning at position start[i]."""
        ib = runner.input_batch
        S = len(rows)
        T = int(n_q.sum())
        cu = np.zeros(S + 1, dtype=np.int32)
        np.cumsum(n_q, out=cu[1:])
        req_idx = np.repeat(np.arange(S, dtype=np.int64), n_q)
        pos = start[req_idx] + (np.arange(T, dtype=np.int64) - cu[:-1].astype(np.int64)[req_idx])
        bt = ib.block_tables[0]
        tok_rows = rows[req_idx]
        slot = bt[tok_rows, pos // self.block_size].astype(np.int64) * self.block_size + pos % self.block_size
        seq_lens = (start + n_q).astype(np.int32)
        max_seq = int(seq_lens.max())
        W = max(-(-max_seq // self.block_size), 1)
        runner._to_dev(runner.input_ids, ids.astype(np.int32))
        runner._to_dev(runner.positions, pos)
        runner._to_dev(runner.slot_mapping, slot)
        runner._to_dev(runner.seq_lens, seq_lens)
        runner._to_dev(runner.query_start_loc, cu)
        runner.block_table[:S, :W].copy_(torch.from_numpy(np.ascontiguousarray(bt[rows, :W])), non_blocking=True)
        nd, ndt = (S, T) if decode else (0, 0)
        if not decode:
            runner._to_dev(runner.prefill_qsl, cu)
        nsplit = runner._decode_splits(nd, max_seq, False)
        md = runner._build_attn_metadata(S, T, nd, ndt, max_seq, int(n_q.max()), nsplit)
        with set_forward_context(md, num_tokens=0):
            return self.model(runner.input_ids[:T], runner.positions[:T], hidden)

    @torch.inference_mode()
    def propose(self, runner, req_ids, rows, sampled_lists, hidden, cu, n_sched=None, so=None) -> list[list[int]]:
        ib = runner.input_batch
        S = len(req_ids)
        drafts = [len(so.scheduled_spec_decode_tokens.get(r, ())) for r in req_ids]
        v = accepted_bounds(n_sched, drafts, sampled_lists)
        # the step's scheduled positions started here (num_computed is advanced by the scheduler later)
        start = ib.num_computed[rows].astype(np.int64)
        # pass 1: all verified positions; inputs are the *next* tokens and the target's hidden states
        req_idx = np.repeat(np.arange(S, dtype=np.int64), v)
        cuv = np.zeros(S + 1, dtype=np.int64)
        np.cumsum(v, out=cuv[1:])
        off = np.arange(int(cuv[-1]), dtype=np.int64) - cuv[:-1][req_idx]
        pos = start[req_idx] + off
        ids = ib.token_ids[rows[req_idx], pos + 1]
        src = torch.from_numpy(cu[:-1].astype(np.int64)[req_idx] + off).to(hidden.device)
        h = self._run(runner, ids, hidden.index_select(0, src), rows, v.astype(np.int32), start, decode=False)
        last = torch.from_numpy(cuv[1:] - 1).to(h.device)
        want = np.array([bool(sampled_lists[i]) and ib.num_tokens[rows[i]] + self.k < self.max_model_len
                         for i in range(S)])
        if not want.any():
            return [[] for _ in range(S)]
        sel = np.nonzero(want)[0]
        sel_t = torch.from_numpy(sel).to(h.device)
        h_last = h.index_select(0, last).index_select(0, sel_t)
        tok = self.model.compute_logits(h_last).argmax(dim=-1)
        out = [tok]
        rows_s = rows[sel]
        nxt = (start + v)[sel]                   # position of the first draft token's own input
        one = np.ones(len(sel), dtype=np.int32)
        for j in range(1, self.k):
            h_last = self._run(runner, tok.cpu().numpy(), h_last, rows_s, one, nxt + (j - 1), decode=True)
            tok = self.model.compute_logits(h_last).argmax(dim=-1)
            out.append(tok)
        toks = torch.stack(out, dim=1).tolist()
        res: list[list[int]] = [[] for _ in range(S)]
        for n, i in enumerate(sel):
            res[int(i)] = toks[n]
        return res


class Eagle3Proposer(EagleProposer):
    """EAGLE-3: the draft reads the target's residual stream at three depths instead of its final hidden state.
    Reference: vllm/v1/spec_decode/eagle.py (use_aux_hidden_state), vllm/v1/worker/gpu_model_runner.py:1340-1380."""

    def __init__(self, vllm_config: VllmConfig, runner):
        super().__init__(vllm_config, runner)
        if not getattr(self.model, "is_eagle3", False):
            raise ValueError("speculative method eagle3 needs an EAGLE-3 draft checkpoint")
        inner = getattr(runner.model, "model", None)
        if inner is None or not hasattr(inner, "enable_aux_hidden_states"):
            raise ValueError(f"{type(runner.model).__name__} cannot hand its intermediate states to an EAGLE-3 draft")
        n = len(inner.layers)
        layers = getattr(self.model.config, "eagle_aux_hidden_state_layer_ids", None) or (2, n // 2, n - 3)
        cap = max(runner.max_num_tokens, max(runner.capture_sizes, default=0))
        inner.enable_aux_hidden_states(tuple(max(min(int(x), n - 1), 0) for x in layers), cap)
        self.target_inner = inner

    def propose(self, runner, req_ids, rows, sampled_lists, hidden, cu, n_sched=None, so=None):
        return super().propose(runner, req_ids, rows, sampled_lists, self.target_inner.aux_hidden[: hidden.shape[0]],
                               cu, n_sched, so)


def make_proposer(vllm_config: VllmConfig, runner):
    m = vllm_config.speculative_config.method
    if vllm_config.parallel_config.token_parallel_size > 1:
        raise ValueError("speculative decoding is not supported together with token parallelism")
    if m == "ngram":
        return NgramProposer(vllm_config)
    if m == "medusa":
        return MedusaProposer(vllm_config, runner)
    if m == "mlp_speculator":
        return MLPSpeculatorProposer(vllm_config, runner)
    if m == "eagle3":
        return Eagle3Proposer(vllm_config, runner)
    if m in ("eagle", "draft_model", "mtp", "deepseek_mtp", "mimo_mtp"):
        return EagleProposer(vllm_config, runner)
    raise ValueError(f"unknown speculative method {m!r}")
