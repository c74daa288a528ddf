"""Llama-family decoder (Llama 2/3, Mistral, Qwen2, Qwen3 dense): pre-norm
blocks, GQA attention with RoPE, gated SiLU MLP.

Per layer the GPU executes: fused residual-add+RMSNorm -> QKV GEMM -> fused
RoPE + KV-cache write -> paged attention -> O GEMM (+TP all-reduce) -> fused
residual-add+RMSNorm -> gate/up GEMM -> SiLU*mul -> down GEMM (+TP all-reduce).

Reference: vllm/model_executor/models/llama.py (MLP :60, attention :100-271,
decoder layer :274, model :331-483, weight mapping :485-549, causal LM :552-693)
including the fork's token-parallel variants (:85-127,179-195,256-271).
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch
import torch.nn as nn

from ..config import VllmConfig
from ..layers.attention import Attention
from ..layers.basic import (GemmaRMSNorm, LogitsProcessor, ParallelLMHead, RMSNorm, SiluAndMul, VocabParallelEmbedding,
                            get_act_and_mul_fn)
from ..forward_context import get_forward_context, has_forward_context
from ..layers.linear import (MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear,
                             set_defer_reduce, set_sequence_parallel)
from ..layers.rotary import get_rope
from ..layers.token_parallel import tknp_metadata, token_parallel_attention
from ..parallel.state import (get_pp_group, get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                              get_tp_group, tensor_model_parallel_all_gather)
from .utils import (IntermediateTensors, PPMissingLayer, load_weights_with_mapping, make_layers, maybe_prefix)


class LlamaMLP(nn.Module):
    def __init__(self, hidden_size: int, intermediate_size: int, bias: bool = False, quant_config=None,
                 prefix: str = "", reduce_results: bool = True, act: str = "silu"):
        super().__init__()
        self.gate_up_proj = MergedColumnParallelLinear(hidden_size, [intermediate_size] * 2, bias=bias,
                                                       quant_config=quant_config, prefix=f"{prefix}.gate_up_proj")
        self.down_proj = RowParallelLinear(intermediate_size, hidden_size, bias=bias, quant_config=quant_config,
                                           reduce_results=reduce_results, prefix=f"{prefix}.down_proj")
        self.act_fn = SiluAndMul() if act == "silu" else get_act_and_mul_fn(act)
        self._silu = act == "silu"

    def forward(self, x):
        if self._silu:
            h = self.gate_up_proj.forward_silu_mul(x)      # decode batches: activation inside the GEMM epilogue
            if h is not None:
                return self.down_proj(h)
        return self.down_proj(self.act_fn(self.gate_up_proj(x)))


class PerHeadLayerNorm(nn.Module):
    """LayerNorm over the head dimension with a scale (and optional shift) of its own per head: weight
    ``[heads, head_dim]``, this rank's heads only (Chameleon, Cohere / StableLM q-k norms).  KV heads replicated
    over ranks take the shard of the head they hold."""

    def __init__(self, total_heads: int, head_dim: int, eps: float = 1e-5, bias: bool = True):
        super().__init__()
        tp, r = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        self.heads = max(1, total_heads // tp)
        self.first = (r * total_heads // tp) if total_heads >= tp else r // (tp // total_heads)
        self.head_dim, self.eps = head_dim, eps
        self.weight = nn.Parameter(torch.ones(self.heads, head_dim))
        self.weight.weight_loader = self._load
        if bias:
            self.bias = nn.Parameter(torch.zeros(self.heads, head_dim))
            self.bias.weight_loader = self._load
        else:
            self.bias = None

    def _load(self, param, tensor):
        param.data.copy_(tensor.reshape(-1, self.head_dim)[self.first:self.first + self.heads])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: [T, heads * head_dim] (a view is fine) -> same shape."""
        T = x.shape[0]
        y = nn.functional.layer_norm(x.reshape(T, self.heads, self.head_dim).float(), (self.head_dim,), None, None,
                                     self.eps)
        y = y * self.weight.float()
        if self.bias is not None:
            y = y + self.bias.float()
        return y.to(x.dtype).view(T, self.heads * self.head_dim)


class LlamaAttention(nn.Module):
    def __init__(self, config, hidden_size: int, num_heads: int, num_kv_heads: int, rope_theta: float,
                 rope_scaling: Optional[dict], max_position: int, bias: bool = False, o_bias: bool = False,
                 sliding_window: Optional[int] = None, quant_config=None, prefix: str = "", qk_norm: bool = False,
                 cache_dtype: str = "auto", attention_only: bool = False, scale: Optional[float] = None,
                 logits_soft_cap: Optional[float] = None, neox_style: bool = True,
                 qkv_clip: Optional[float] = None, alibi_slopes: Optional[list] = None, use_rope: bool = True,
                 blocksparse: Optional[dict] = None):
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        self.qkv_clip = qkv_clip
        self.total_num_heads, self.total_num_kv_heads = num_heads, num_kv_heads
        self.num_heads = num_heads // tp
        self.num_kv_heads = max(1, num_kv_heads // tp)
        self.head_dim = getattr(config, "head_dim", None) or hidden_size // num_heads
        self.q_size, self.kv_size = self.num_heads * self.head_dim, self.num_kv_heads * self.head_dim
        self.attention_only = attention_only
        if not attention_only:
            self._make_projections(hidden_size, num_heads, num_kv_heads, bias, o_bias, quant_config, prefix)
        partial = getattr(config, "partial_rotary_factor", None) or \
            (getattr(config, "rope_parameters", None) or {}).get("partial_rotary_factor") or 1.0
        self.rotary_emb = get_rope(self.head_dim, self.head_dim, max_position, rope_theta, neox_style, rope_scaling,
                                   partial) if use_rope else None
        if alibi_slopes is not None:       # this rank's heads
            r = get_tensor_model_parallel_rank()
            alibi_slopes = list(alibi_slopes)[r * self.num_heads:(r + 1) * self.num_heads]
        dca = getattr(config, "dual_chunk_attention_config", None)
        if dca and use_rope:
            # Qwen2.5-1M: keys rotated with their position inside a chunk, three query rotations (layers/attention.py
            # ::DualChunkAttention); the table only needs chunk_size + 1 rows and takes no length scaling
            # (reference: vllm/model_executor/models/qwen2.py:117-175 passes the same dict to rope and attention)
            from ..layers.attention import DualChunkAttention
            cs, ls = int(dca.get("chunk_size", 8192)), int(dca.get("local_size", 1024))
            # sparse prefill (optional): per layer {head: [kind, vertical lines, slash lines, score]}; its probe
            # rotations reach position chunk_size + chunk_len
            sparse, per_layer = None, dca.get("sparse_attention_config")
            if per_layer and dca.get("sparse_attention_enabled", True):
                import re
                layer = int(re.search(r"layers\.(\d+)\.", prefix).group(1))
                row = {int(h): v for h, v in per_layer[layer].items()}
                h0 = get_tensor_model_parallel_rank() * self.num_heads
                if any(row[h][0] != "vertical_and_slash" for h in range(h0, h0 + self.num_heads)):
                    raise ValueError("sparse attention config: only the vertical_and_slash pattern is served")
                sparse = dict(heads=[(row[h][1], row[h][2]) for h in range(h0, h0 + self.num_heads)],
                              threshold=dca.get("sparse_attention_threshold", 32768),
                              last_q=dca.get("sparse_attention_last_q", 64))
            self.rotary_emb = get_rope(self.head_dim, self.head_dim, 2 * cs - ls + 1, rope_theta, neox_style, None,
                                       partial)
            self.attn = DualChunkAttention(
                self.num_heads, self.head_dim, scale if scale is not None else self.head_dim ** -0.5, self.num_kv_heads,
                logits_soft_cap=logits_soft_cap, prefix=f"{prefix}.attn", cache_dtype=cache_dtype, chunk_size=cs,
                local_size=ls, original_max_position_embeddings=int(dca.get("original_max_position_embeddings", 0)),
                sparse=sparse)
        else:
            self.attn = Attention(self.num_heads, self.head_dim, scale if scale is not None else self.head_dim ** -0.5,
                                  self.num_kv_heads, alibi_slopes=alibi_slopes, sliding_window=sliding_window,
                                  logits_soft_cap=logits_soft_cap, prefix=f"{prefix}.attn", cache_dtype=cache_dtype,
                                  blocksparse=blocksparse)
        # True: RMSNorm per head (Qwen3); "gemma": the (1 + w) form per head; "full": one RMSNorm over the
        # whole q / k projection (OLMo-2); "layernorm" / "layernorm_nobias": LayerNorm over the head dimension
        # with per-head scale (and shift)
        self.qk_norm = qk_norm if not attention_only else False
        if self.qk_norm:
            eps = getattr(config, "rms_norm_eps", 1e-6)
            if self.qk_norm in ("layernorm", "layernorm_nobias"):
                ln_eps = float(getattr(config, "qk_layernorm_eps", None) or getattr(config, "layer_norm_eps", 1e-5))
                self.q_norm = PerHeadLayerNorm(num_heads, self.head_dim, ln_eps, self.qk_norm == "layernorm")
                self.k_norm = PerHeadLayerNorm(num_kv_heads, self.head_dim, ln_eps, self.qk_norm == "layernorm")
            elif self.qk_norm == "full":
                self.q_norm = RMSNorm(num_heads * self.head_dim, eps)
                self.k_norm = RMSNorm(num_kv_heads * self.head_dim, eps)
            else:
                cls = GemmaRMSNorm if self.qk_norm == "gemma" else RMSNorm
                self.q_norm = cls(self.head_dim, eps)
                self.k_norm = cls(self.head_dim, eps)

    def _make_projections(self, hidden_size, num_heads, num_kv_heads, bias, o_bias, quant_config, prefix):
        self.qkv_proj = QKVParallelLinear(hidden_size, self.head_dim, num_heads, num_kv_heads, bias=bias,
                                          quant_config=quant_config, prefix=f"{prefix}.qkv_proj")
        self.o_proj = RowParallelLinear(num_heads * self.head_dim, hidden_size, bias=o_bias,
                                        quant_config=quant_config, prefix=f"{prefix}.o_proj")

    def forward_attention_only(self, positions: torch.Tensor, dtype: torch.dtype) -> None:
        """Non-root token-parallel rank: this layer's share of the attention, nothing else."""
        md = tknp_metadata()
        token_parallel_attention(self.attn, self.rotary_emb, None, positions, md, self.q_size + 2 * self.kv_size,
                                 self.q_size, dtype, positions.device)

    def _full_width_qk_norm(self, qkv: torch.Tensor) -> None:
        """RMSNorm over all heads of q (and of k) at once; with TP the other ranks' heads are gathered first.
        KV heads replicated over ranks (fewer KV heads than ranks) are counted once."""
        tp = get_tensor_model_parallel_world_size()
        r = get_tensor_model_parallel_rank()
        rep = max(1, tp // self.total_num_kv_heads)          # ranks holding the same KV head
        for lo, size, norm, dup in ((0, self.q_size, self.q_norm, 1), (self.q_size, self.kv_size, self.k_norm, rep)):
            x = qkv[:, lo:lo + size].contiguous()
            if tp > 1:
                x = tensor_model_parallel_all_gather(x, -1)
                if dup > 1:
                    x = x.view(x.shape[0], tp // dup, dup, size)[:, :, 0].reshape(x.shape[0], -1)
            if x.shape[-1] != norm.weight.shape[0]:
                raise ValueError("full-width q/k norm: gathered width does not match the norm weight")
            y = norm(x)
            own = r // dup
            qkv[:, lo:lo + size] = y[:, own * size:(own + 1) * size] if tp > 1 else y

    def forward(self, positions: torch.Tensor, hidden_states: torch.Tensor) -> torch.Tensor:
        qkv = self.qkv_proj(hidden_states)
        if self.qkv_clip is not None:
            qkv.clamp_(-self.qkv_clip, self.qkv_clip)
        if self.qk_norm == "full":
            self._full_width_qk_norm(qkv)
        elif self.qk_norm in ("layernorm", "layernorm_nobias"):
            qkv[:, : self.q_size] = self.q_norm(qkv[:, : self.q_size])
            qkv[:, self.q_size: self.q_size + self.kv_size] = self.k_norm(qkv[:, self.q_size: self.q_size + self.kv_size])
        elif self.qk_norm:
            T = qkv.shape[0]
            q = qkv[:, : self.q_size].reshape(T * self.num_heads, self.head_dim)
            k = qkv[:, self.q_size: self.q_size + self.kv_size].reshape(T * self.num_kv_heads, self.head_dim)
            qkv[:, : self.q_size] = self.q_norm(q).view(T, self.q_size)
            qkv[:, self.q_size: self.q_size + self.kv_size] = self.k_norm(k).view(T, self.kv_size)
        md = tknp_metadata()
        if md is not None:
            o = token_parallel_attention(self.attn, self.rotary_emb, qkv, positions, md, qkv.shape[1], self.q_size,
                                         qkv.dtype, qkv.device)
            return self.o_proj(o)
        return self.o_proj(self.attn(qkv, positions, self.rotary_emb))


class LlamaDecoderLayer(nn.Module):
    norm_cls = RMSNorm
    supports_sequence_parallel = True      # subclasses with their own forward / MLP turn this off

    def _attention_extras(self, config, layer_idx: int) -> dict:
        """Per-family keyword arguments for LlamaAttention (softmax scale, soft cap, window...)."""
        return {}

    def _rope_of(self, config, layer_idx: int) -> tuple[float, Optional[dict]]:
        """(theta, scaling dict or None) of this layer's rotary embedding."""
        rope_scaling = getattr(config, "rope_scaling", None) or getattr(config, "rope_parameters", None)
        # newer HF configs keep theta inside rope_parameters
        rope_theta = getattr(config, "rope_theta", None) or (rope_scaling or {}).get("rope_theta") or 10000.0
        if rope_scaling and "rope_type" not in rope_scaling and "type" not in rope_scaling:
            rope_scaling = None
        return float(rope_theta), rope_scaling

    def __init__(self, config, vllm_config: VllmConfig, quant_config=None, prefix: str = "",
                 attention_only: bool = False):
        super().__init__()
        mc = vllm_config.model_config
        layer_idx = int(prefix.split(".")[-1])
        rope_theta, rope_scaling = self._rope_of(config, layer_idx)
        bias = bool(getattr(config, "attention_bias", False) or getattr(config, "bias", False))
        arch = (mc.architectures or [""])[0]
        if arch.startswith("Qwen2"):
            bias = True
        window = mc.get_sliding_window()
        if window is not None and getattr(config, "max_window_layers", None) is not None \
                and arch.startswith("Qwen") and layer_idx < config.max_window_layers:
            window = None
        kw = dict(bias=bias, o_bias=False if arch.startswith("Qwen") else bias, sliding_window=window,
                  quant_config=quant_config, prefix=f"{prefix}.self_attn", qk_norm=arch.startswith("Qwen3"),
                  cache_dtype=vllm_config.cache_config.cache_dtype, attention_only=attention_only)
        kw.update(self._attention_extras(config, layer_idx))
        self.self_attn = LlamaAttention(
            config, config.hidden_size, config.num_attention_heads,
            getattr(config, "num_key_value_heads", None) or config.num_attention_heads,
            rope_theta, rope_scaling, mc.max_model_len, **kw)
        if attention_only:
            return
        self.mlp = self._make_mlp(config, vllm_config, quant_config, f"{prefix}.mlp")
        self.input_layernorm = self.make_norm(config)
        self.post_attention_layernorm = self.make_norm(config)

    @classmethod
    def make_norm(cls, config) -> nn.Module:
        """The block's normalisation layer (families with LayerNorm or other epsilon names override)."""
        return cls.norm_cls(config.hidden_size, getattr(config, "rms_norm_eps", 1e-6))

    def _make_mlp(self, config, vllm_config, quant_config, prefix):
        return LlamaMLP(config.hidden_size, config.intermediate_size, bool(getattr(config, "mlp_bias", False)),
                        quant_config, prefix)

    def forward(self, positions, hidden_states, residual, sp: bool = False):
        """With ``sp`` the layer's input, output and residual are this rank's T/tp rows: the norms
        run on the shard, the GEMMs on all rows (all-gather in front, reduce-scatter behind)."""
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        if sp:
            hidden_states = tensor_model_parallel_all_gather(hidden_states, 0)
        hidden_states = self.self_attn(positions, hidden_states)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        if sp:
            hidden_states = tensor_model_parallel_all_gather(hidden_states, 0)
        hidden_states = self.mlp(hidden_states)
        return hidden_states, residual

    # The two halves of ``forward``, split where the TP reductions sit (used with deferred reductions).
    def attention_phase(self, positions, hidden_states, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        return self.self_attn(positions, hidden_states), residual

    def mlp_phase(self, hidden_states, residual):
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.mlp(hidden_states), residual


class LlamaModel(nn.Module):
    layer_cls = LlamaDecoderLayer
    embed_scale_of = staticmethod(lambda config: 1.0)      # Gemma: sqrt(hidden), Granite: embedding_multiplier

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        quant_config = getattr(vllm_config, "quant_config", None)
        self.config = config
        pp = get_pp_group()
        if pp.is_first_rank or (getattr(config, "tie_word_embeddings", False) and pp.is_last_rank):
            self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size,
                                                       prefix=f"{prefix}.embed_tokens")
        else:
            self.embed_tokens = PPMissingLayer()
        self.start_layer, self.end_layer, self.layers = make_layers(
            config.num_hidden_layers,
            lambda p: self.layer_cls(config, vllm_config, quant_config, p), f"{prefix}.layers")
        self.norm = self.layer_cls.make_norm(config) if pp.is_last_rank else PPMissingLayer()
        self.embed_scale = float(self.embed_scale_of(config))
        self.aux_layers: tuple = ()
        self.aux_hidden: Optional[torch.Tensor] = None
        self.tp = get_tensor_model_parallel_world_size()
        # all-reduce + add + norm in one kernel (decode-sized batches): the standard block only -- families that
        # override the block's forward, its MLP or its norm class keep the separate reductions
        self.fuse_reduce_norm_ok = bool(
            getattr(self.layer_cls, "supports_sequence_parallel", False)
            and self.layer_cls.forward is LlamaDecoderLayer.forward and isinstance(self.norm, RMSNorm)
            and not vllm_config.parallel_config.enable_token_parallel)
        # sequence parallelism: dense Llama-style layers only (the MoE blocks reduce on their own)
        self.sequence_parallel = bool(vllm_config.compilation_config.enable_sequence_parallelism and self.tp > 1
                                      and getattr(self.layer_cls, "supports_sequence_parallel", False)
                                      and not vllm_config.parallel_config.enable_token_parallel)

    def get_input_embeddings(self, input_ids):
        return self.embed_tokens(input_ids)

    def enable_aux_hidden_states(self, layers: tuple, max_tokens: int) -> None:
        """Keep ``hidden + residual`` as it enters each of ``layers`` in one persistent [max_tokens, n * H] buffer
        (written inside captured decode graphs too).  Reference: vllm/model_executor/models/llama.py:383-397."""
        if get_pp_group().world_size > 1 or self.sequence_parallel:
            raise ValueError("EAGLE-3 drafting needs the whole target stack on one pipeline stage, without "
                             "sequence parallelism")
        w = self.embed_tokens.weight
        self.aux_layers = tuple(int(x) for x in layers)
        self.aux_hidden = torch.zeros((max_tokens, len(self.aux_layers) * self.config.hidden_size), dtype=w.dtype,
                                      device=w.device)

    def forward(self, input_ids, positions, intermediate_tensors: Optional[IntermediateTensors] = None,
                inputs_embeds: Optional[torch.Tensor] = None):
        pp = get_pp_group()
        if pp.is_first_rank:
            hidden = inputs_embeds if inputs_embeds is not None else self.embed_tokens(input_ids)
            if self.embed_scale != 1.0:
                hidden = hidden * self.embed_scale
            residual = None
        else:
            hidden, residual = intermediate_tensors["hidden_states"], intermediate_tensors["residual"]
        sp = self.sequence_parallel and hidden.shape[0] % self.tp == 0 and hidden.shape[0] > 0
        mbs = get_forward_context().micro_batches if (self.tp > 1 and has_forward_context()) else None
        if mbs and not sp and getattr(self.layer_cls, "supports_sequence_parallel", False):
            hidden, residual = self._forward_micro_batched(positions, hidden, residual, mbs)
        elif not sp and self.aux_layers:
            # EAGLE-3 drafting: the residual stream in front of a few layers is kept for the draft network
            T, H = hidden.shape
            for i, layer in enumerate(self.layers[self.start_layer:self.end_layer], self.start_layer):
                for j, a in enumerate(self.aux_layers):
                    if a == i:
                        self.aux_hidden[:T, j * H:(j + 1) * H] = hidden if residual is None else hidden + residual
                hidden, residual = layer(positions, hidden, residual)
        elif not sp and self._fuse_reduce_norm(hidden):
            return self._forward_fused_reduce(positions, hidden, residual)
        elif not sp:
            for layer in self.layers[self.start_layer:self.end_layer]:
                hidden, residual = layer(positions, hidden, residual)
        else:
            # every stage starts and ends with full rows; in between each rank holds T/tp of them
            r = get_tensor_model_parallel_rank()
            hidden = hidden.chunk(self.tp)[r]
            if residual is not None:
                residual = residual.chunk(self.tp)[r]
            old = set_sequence_parallel(True)
            try:
                for layer in self.layers[self.start_layer:self.end_layer]:
                    hidden, residual = layer(positions, hidden, residual, sp=True)
            finally:
                set_sequence_parallel(old)
            if pp.is_last_rank:
                hidden, _ = self.norm(hidden, residual)
                return tensor_model_parallel_all_gather(hidden, 0)
            hidden = tensor_model_parallel_all_gather(hidden, 0)
            residual = tensor_model_parallel_all_gather(residual, 0)
        if not pp.is_last_rank:
            return IntermediateTensors({"hidden_states": hidden, "residual": residual})
        hidden, _ = self.norm(hidden, residual)
        return hidden


def _fuse_reduce_norm(self, hidden) -> bool:
    """Decode-sized batch of a dense Llama-style stack under tensor parallelism, peer-memory kernel up."""
    return (self.tp > 1 and self.fuse_reduce_norm_ok and get_tp_group().can_fuse_reduce_norm(hidden))


def _forward_fused_reduce(self, positions, hidden, residual):
    """Every tensor-parallel all-reduce of the stack is followed by a residual add and an RMSNorm: the
    row-parallel layers hand over their partial sums and the three run as one peer-memory kernel
    (``RMSNorm.forward_reduce``).  Role of the reference's all-reduce + norm fusion pass
    (vllm/compilation/collective_fusion.py), chosen at run time by batch size."""
    pp = get_pp_group()
    partial = False                 # is ``hidden`` an un-reduced partial sum?
    old = set_defer_reduce(True)
    try:
        for layer in self.layers[self.start_layer:self.end_layer]:
            if residual is None:
                residual, h = hidden, layer.input_layernorm(hidden)
            elif partial:
                h, residual = layer.input_layernorm.forward_reduce(hidden, residual)
            else:
                h, residual = layer.input_layernorm(hidden, residual)
            o = layer.self_attn(positions, h)
            h, residual = layer.post_attention_layernorm.forward_reduce(o, residual)
            hidden, partial = layer.mlp(h), True
    finally:
        set_defer_reduce(old)
    if not pp.is_last_rank:
        if partial:
            hidden = get_tp_group().all_reduce(hidden)
        return IntermediateTensors({"hidden_states": hidden, "residual": residual})
    if partial:
        hidden, _ = self.norm.forward_reduce(hidden, residual)
    else:
        hidden, _ = self.norm(hidden, residual)
    return hidden


LlamaModel._fuse_reduce_norm = _fuse_reduce_norm
LlamaModel._forward_fused_reduce = _forward_fused_reduce


def _forward_micro_batched(self, positions, hidden, residual, mbs):
    """Async tensor parallelism for large (prefill) batches: the batch runs as micro-batches of whole
    sequences, and every TP all-reduce is started on the communication stream while this stream goes
    on with the other micro-batch, so xGMI traffic hides behind GEMMs instead of adding to them:

        A.attn -> AR(A) | B.attn -> AR(B) | A.mlp -> AR(A') | B.mlp -> AR(B') | next layer A.attn ...

    Every rank issues the collectives in this same order.  Role of the reference's collective-fusion
    pass (vllm/compilation/collective_fusion.py), done at run time instead of by graph rewriting."""
    ctx = get_forward_context()
    tp = get_tp_group()
    full_md = ctx.attn_metadata
    n = len(mbs)
    pos = [positions[m.start:m.end] for m in mbs]
    h = [hidden[m.start:m.end] for m in mbs]
    res = [None if residual is None else residual[m.start:m.end] for m in mbs]
    pending = [None] * n
    old = set_defer_reduce(True)
    try:
        for layer in self.layers[self.start_layer:self.end_layer]:
            for i, m in enumerate(mbs):
                ctx.attn_metadata = m.attn_metadata
                x = pending[i].wait() if pending[i] is not None else h[i]
                o, res[i] = layer.attention_phase(pos[i], x, res[i])
                pending[i] = tp.all_reduce_async(o)
            for i in range(n):
                y, res[i] = layer.mlp_phase(pending[i].wait(), res[i])
                pending[i] = tp.all_reduce_async(y)
    finally:
        set_defer_reduce(old)
        ctx.attn_metadata = full_md
    h = [pending[i].wait() if pending[i] is not None else h[i] for i in range(n)]
    tail = hidden[mbs[-1].end:]          # padding rows beyond the scheduled tokens, if any
    if tail.shape[0]:
        h, res = h + [tail], res + [tail]
    return torch.cat(h, dim=0), torch.cat(res, dim=0)


LlamaModel._forward_micro_batched = _forward_micro_batched


class LlamaForCausalLM(nn.Module):
    model_cls = LlamaModel
    stacked_params_mapping = [
        (".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"), (".qkv_proj", ".v_proj", "v"),
        (".gate_up_proj", ".gate_proj", 0), (".gate_up_proj", ".up_proj", 1),
    ]
    packed_modules_mapping = {"qkv_proj": ["q_proj", "k_proj", "v_proj"], "gate_up_proj": ["gate_proj", "up_proj"]}
    supports_lora = True
    supports_pp = True

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        self.model = self.model_cls(vllm_config, prefix=maybe_prefix(prefix, "model"))
        pp = get_pp_group()
        self.tie = bool(getattr(config, "tie_word_embeddings", False))
        if pp.is_last_rank:
            self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size,
                                          prefix=maybe_prefix(prefix, "lm_head"))
            if self.tie:
                self.lm_head.tie_weights(self.model.embed_tokens)
            self.logits_processor = LogitsProcessor(config.vocab_size, config.vocab_size, self._logit_scale(config),
                                                    soft_cap=getattr(config, "final_logit_softcapping", None))
        else:
            self.lm_head = PPMissingLayer()

    @staticmethod
    def _logit_scale(config) -> float:
        return float(getattr(config, "logit_scale", 1.0) or 1.0)

    def get_input_embeddings(self, input_ids):
        return self.model.get_input_embeddings(input_ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        return self.model(input_ids, positions, intermediate_tensors, inputs_embeds)

    def compute_logits(self, hidden_states: torch.Tensor) -> torch.Tensor:
        return self.logits_processor(self.lm_head, hidden_states)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]) -> set[str]:
        return load_weights_with_mapping(self, weights, self.stacked_params_mapping, tie_lm_head=self.tie)


class MistralForCausalLM(LlamaForCausalLM):
    pass


class Qwen2ForCausalLM(LlamaForCausalLM):
    pass


class Qwen3ForCausalLM(LlamaForCausalLM):
    pass


class TokenParallelWorkerModel(nn.Module):
    """What a non-root token-parallel rank runs: the attention of every local layer (same module
    names as the full model, so KV-cache binding is identical) and nothing that needs weights."""
    is_tknp_worker = True

    def __init__(self, vllm_config: VllmConfig, prefix: str = "", layer_cls=LlamaDecoderLayer):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        self.dtype = vllm_config.model_config.dtype
        self.model = nn.Module()
        mp = maybe_prefix(prefix, "model")
        self.model.start_layer, self.model.end_layer, self.model.layers = make_layers(
            config.num_hidden_layers, lambda p: layer_cls(config, vllm_config, None, p, attention_only=True),
            f"{mp}.layers")

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        if tknp_metadata() is None:          # memory profiling: nothing to run without the root
            return None
        m = self.model
        for layer in m.layers[m.start_layer:m.end_layer]:
            layer.self_attn.forward_attention_only(positions, self.dtype)
        return None

    def load_weights(self, weights) -> None:
        for _ in ():                          # no parameters: the checkpoint is not even opened
            pass
        return None


LlamaForCausalLM.tknp_worker_cls = TokenParallelWorkerModel
