"""Llama-4 text decoder (Scout 16E, Maverick 128E), also the language half of the
``Llama4ForConditionalGeneration`` checkpoints.

What differs from the Llama block:
  * rotary layers use interleaved (complex-pair) RoPE, then a weightless RMS norm per head on q and k, and
    attend inside chunks of ``attention_chunk_size`` positions (chunked local attention);
  * every ``nope_layer_interval``-th layer has no positional encoding, attends globally and scales q by
    ``log1p(floor((pos + 1) / floor_scale)) * attn_scale + 1`` (temperature tuning for long contexts);
  * MoE layers route top-k on the raw router logits, weigh the experts' INPUT by sigmoid(logit), and add a
    shared dense expert; the other layers are a plain gated MLP of ``intermediate_size_mlp``;
  * checkpoints stack the experts: ``experts.gate_up_proj [E, H, 2I]`` and ``experts.down_proj [E, I, H]``.

Reference: vllm/model_executor/models/llama4.py (Llama4MoE :50-118, Llama4Attention :121-243, decoder layer
:246-329, expert weight loading :350-470), vllm/model_executor/models/mllama4.py for the checkpoint prefix.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch
import torch.nn as nn

from .. import ops
from ..config import VllmConfig
from ..layers.attention import Attention
from ..layers.basic import RMSNorm
from ..layers.linear import QKVParallelLinear, ReplicatedLinear, RowParallelLinear
from ..layers.moe import FusedMoE
from ..layers.rotary import get_rope
from ..parallel.state import get_tensor_model_parallel_world_size, tensor_model_parallel_all_reduce
from .llama import LlamaForCausalLM, LlamaMLP, LlamaModel
from .utils import load_weights_with_mapping


def _uses_rope(config, layer_idx: int) -> bool:
    flags = getattr(config, "no_rope_layers", None)
    if flags:                       # 1 = rotary layer, 0 = no positional encoding (the config's naming)
        return bool(flags[layer_idx])
    interval = int(getattr(config, "nope_layer_interval", 0) or 0)
    return not (interval and (layer_idx + 1) % interval == 0)


def _is_moe_layer(config, layer_idx: int) -> bool:
    layers = getattr(config, "moe_layers", None)
    if layers is not None:
        return layer_idx in layers
    step = int(getattr(config, "interleave_moe_layer_step", 1) or 1)
    return (layer_idx + 1) % step == 0


class Llama4MoE(nn.Module):
    """router -> top-k on logits -> experts run on sigmoid(logit) * x -> sum, plus the shared expert."""

    def __init__(self, config, quant_config=None, prefix: str = ""):
        super().__init__()
        H, I = config.hidden_size, config.intermediate_size
        self.top_k = int(config.num_experts_per_tok)
        self.num_experts = int(config.num_local_experts)
        self.tp = get_tensor_model_parallel_world_size()
        self.router = ReplicatedLinear(H, self.num_experts, bias=False, quant_config=None, prefix=f"{prefix}.router")
        # the layer sees one (token, expert) pair per row: the score already sits in the row
        self.experts = FusedMoE(self.num_experts, 1, H, I, reduce_results=False, renormalize=False,
                                quant_config=quant_config, prefix=f"{prefix}.experts")
        self.shared_expert = LlamaMLP(H, I, quant_config=quant_config, prefix=f"{prefix}.shared_expert",
                                      reduce_results=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        T, H = x.shape
        k = self.top_k
        logits = self.router(x)
        top, ids = torch.topk(logits, k, dim=-1)
        score = torch.sigmoid(top.float()).to(x.dtype)
        if k == 1:
            rows = x * score
        else:
            rows = (x.unsqueeze(1) * score.unsqueeze(-1)).reshape(T * k, H)
        ones = torch.ones((T * k, 1), dtype=torch.float32, device=x.device)
        out = self.experts(rows, None, routed=(ones, ids.reshape(T * k, 1).to(torch.int32)))
        if k > 1:
            out = out.view(T, k, H).sum(dim=1)
        out = out + self.shared_expert(x)
        if self.tp > 1 or self.experts.ep_size > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out


class Llama4Attention(nn.Module):
    def __init__(self, config, vllm_config: VllmConfig, layer_idx: int, quant_config=None, prefix: str = ""):
        super().__init__()
        mc = vllm_config.model_config
        tp = get_tensor_model_parallel_world_size()
        H, nh = config.hidden_size, config.num_attention_heads
        nkv = getattr(config, "num_key_value_heads", None) or nh
        self.num_heads, self.num_kv_heads = nh // tp, max(1, nkv // tp)
        self.head_dim = getattr(config, "head_dim", None) or H // nh
        self.q_size, self.kv_size = self.num_heads * self.head_dim, self.num_kv_heads * self.head_dim
        self.use_rope = _uses_rope(config, layer_idx)
        self.eps = float(getattr(config, "rms_norm_eps", 1e-5))
        self.qk_norm = bool(getattr(config, "use_qk_norm", False)) and self.use_rope
        self.temperature_tuning = bool(getattr(config, "attn_temperature_tuning", False)) and not self.use_rope
        self.floor_scale = float(getattr(config, "floor_scale", 8192))
        self.attn_scale = float(getattr(config, "attn_scale", 0.1))
        bias = bool(getattr(config, "attention_bias", False))
        self.qkv_proj = QKVParallelLinear(H, self.head_dim, nh, nkv, bias=bias, quant_config=quant_config,
                                          prefix=f"{prefix}.qkv_proj")
        self.o_proj = RowParallelLinear(nh * self.head_dim, H, bias=bias, quant_config=quant_config,
                                        prefix=f"{prefix}.o_proj")
        self.rotary_emb = None
        if self.use_rope:
            rs = getattr(config, "rope_scaling", None) or getattr(config, "rope_parameters", None)
            theta = getattr(config, "rope_theta", None) or (rs or {}).get("rope_theta") or 500000.0
            if rs and "rope_type" not in rs and "type" not in rs:
                rs = None
            if rs:
                rs = {k: v for k, v in dict(rs).items() if k != "rope_theta"}
            self.rotary_emb = get_rope(self.head_dim, self.head_dim, mc.max_model_len, float(theta), False, rs)
        chunk = int(getattr(config, "attention_chunk_size", 0) or 0) if self.use_rope else 0
        self.attn = Attention(self.num_heads, self.head_dim, self.head_dim ** -0.5, self.num_kv_heads,
                              prefix=f"{prefix}.attn", cache_dtype=vllm_config.cache_config.cache_dtype,
                              local_chunk=chunk)
        self.register_buffer("_unit", torch.ones(self.head_dim), persistent=False)

    def forward(self, positions: torch.Tensor, hidden_states: torch.Tensor) -> torch.Tensor:
        qkv = self.qkv_proj(hidden_states)
        T = qkv.shape[0]
        q, k = qkv[:, : self.q_size], qkv[:, self.q_size: self.q_size + self.kv_size]
        if self.rotary_emb is not None:
            self.rotary_emb(positions[:T], q, k)
        if self.qk_norm:
            unit = self._unit.to(qkv.dtype)
            qn = ops.rms_norm(q.reshape(T * self.num_heads, self.head_dim), unit, self.eps)
            kn = ops.rms_norm(k.reshape(T * self.num_kv_heads, self.head_dim), unit, self.eps)
            qkv[:, : self.q_size] = qn.view(T, self.q_size)
            qkv[:, self.q_size: self.q_size + self.kv_size] = kn.view(T, self.kv_size)
        if self.temperature_tuning:
            s = torch.log1p(torch.floor((positions[:T].float() + 1.0) / self.floor_scale)) * self.attn_scale + 1.0
            qkv[:, : self.q_size] = (q * s.unsqueeze(-1)).to(qkv.dtype)
        return self.o_proj(self.attn(qkv, None, None))


class Llama4DecoderLayer(nn.Module):
    supports_sequence_parallel = False

    @classmethod
    def make_norm(cls, config) -> nn.Module:
        return RMSNorm(config.hidden_size, getattr(config, "rms_norm_eps", 1e-5))

    def __init__(self, config, vllm_config: VllmConfig, quant_config=None, prefix: str = ""):
        super().__init__()
        layer_idx = int(prefix.split(".")[-1])
        self.self_attn = Llama4Attention(config, vllm_config, layer_idx, quant_config, f"{prefix}.self_attn")
        if _is_moe_layer(config, layer_idx):
            self.feed_forward = Llama4MoE(config, quant_config, f"{prefix}.feed_forward")
        else:
            self.feed_forward = LlamaMLP(config.hidden_size, int(getattr(config, "intermediate_size_mlp", None)
                                                                 or config.intermediate_size),
                                         False, quant_config, f"{prefix}.feed_forward")
        self.input_layernorm = self.make_norm(config)
        self.post_attention_layernorm = self.make_norm(config)

    def forward(self, positions, hidden_states, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.feed_forward(hidden_states), residual


class Llama4Model(LlamaModel):
    layer_cls = Llama4DecoderLayer


class Llama4ForCausalLM(LlamaForCausalLM):
    model_cls = Llama4Model
    supports_lora = False
    tknp_worker_cls = None

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]) -> set[str]:
        E = int(self.config.num_local_experts)

        def fix(items):
            for name, t in items:
                if name.startswith("language_model."):          # checkpoints of the image+text model
                    name = name[len("language_model."):]
                elif name.startswith(("vision_model.", "multi_modal_projector.")):
                    continue
                if name.endswith(".feed_forward.experts.gate_up_proj"):
                    base = name[: -len("gate_up_proj")]
                    half = t.shape[2] // 2
                    for e in range(t.shape[0]):
                        yield f"{base}{e}.w1.weight", t[e, :, :half].t()
                        yield f"{base}{e}.w3.weight", t[e, :, half:].t()
                elif name.endswith(".feed_forward.experts.down_proj"):
                    base = name[: -len("down_proj")]
                    for e in range(t.shape[0]):
                        yield f"{base}{e}.w2.weight", t[e].t()
                else:
                    yield name, t

        experts = FusedMoE.make_expert_params_mapping("w1", "w2", "w3", E)
        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping, tie_lm_head=self.tie,
                                         expert_mapping=experts)


class Llama4ForConditionalGeneration(Llama4ForCausalLM):
    """The image+text checkpoints served as a text model: the language weights load, the vision tower and
    projector are skipped (prompts carry text only)."""
