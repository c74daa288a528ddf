"""More decoder families built from the Llama block and the shared MoE layers:

  Ernie 4.5 (dense and MoE)   interleaved-pair rotary, optional biases; MoE with shared experts and a
                              softmax router whose selection (not weights) is shifted by a learned bias
  dots.llm1                   Qwen3-style attention (per-head q/k RMSNorm), DeepSeek-V3-style MoE
  HunYuan v1 (dense and MoE)  per-head q/k RMSNorm applied AFTER the rotary embedding; MoE = softmax top-k
                              experts + a shared dense MLP
  MiMo                        Qwen2 with multi-token-prediction layers that serving ignores
  Persimmon                   LayerNorm blocks, per-head LayerNorm on q/k, partial rotary, squared-ReLU MLP
  MiniCPM                     Llama with scaled embeddings, depth-scaled residual branches and scaled logits

Reference: vllm/model_executor/models/{ernie45.py, ernie45_moe.py, dots1.py, hunyuan_v1_moe.py, mimo.py,
persimmon.py, minicpm.py}.
"""
from __future__ import annotations

import math
from typing import Iterable, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..layers.basic import LayerNorm, RMSNorm
from ..layers.linear import ColumnParallelLinear, QKVParallelLinear, ReplicatedLinear, RowParallelLinear
from ..layers.moe import FusedMoE
from ..parallel.state import get_tensor_model_parallel_world_size, tensor_model_parallel_all_reduce
from .deepseek import DeepseekMoE
from .llama import LlamaAttention, LlamaDecoderLayer, LlamaForCausalLM, LlamaMLP, LlamaModel
from .utils import load_weights_with_mapping


# ---- Ernie 4.5 ----------------------------------------------------------------------------------------
class Ernie45DecoderLayer(LlamaDecoderLayer):
    def _attention_extras(self, config, layer_idx: int) -> dict:
        b = bool(getattr(config, "use_bias", False))
        return {"neox_style": False, "bias": b, "o_bias": b}

    def _make_mlp(self, config, vllm_config, quant_config, prefix):
        return LlamaMLP(config.hidden_size, config.intermediate_size, bool(getattr(config, "use_bias", False)),
                        quant_config, prefix)


class Ernie45Model(LlamaModel):
    layer_cls = Ernie45DecoderLayer


class Ernie4_5ForCausalLM(LlamaForCausalLM):
    model_cls = Ernie45Model


class Ernie45MoE(nn.Module):
    """softmax scores; the top-k choice is made on scores + e_score_correction_bias, the weights are the
    unshifted scores of the chosen experts, renormalised; shared experts run beside the routed ones."""

    def __init__(self, config, quant_config=None, prefix: str = ""):
        super().__init__()
        E = int(config.moe_num_experts)
        H, I = config.hidden_size, int(config.moe_intermediate_size)
        self.tp = get_tensor_model_parallel_world_size()
        self.top_k = int(config.moe_k)
        self.norm_min = float(getattr(config, "moe_norm_min", 1e-12) or 1e-12)
        self.gate = ReplicatedLinear(H, E, bias=False, quant_config=None, prefix=f"{prefix}.gate")
        self.e_score_correction_bias = nn.Parameter(torch.zeros(E, dtype=torch.float32), requires_grad=False)
        n_shared = int(getattr(config, "moe_num_shared_experts", 0) or 0)
        self.experts = FusedMoE(E, self.top_k, H, I, reduce_results=n_shared == 0, renormalize=False,
                                quant_config=quant_config, prefix=f"{prefix}.experts",
                                custom_routing_function=self._route)
        self.shared_experts = LlamaMLP(H, I * n_shared, bool(getattr(config, "use_bias", False)), quant_config,
                                       f"{prefix}.shared_experts", reduce_results=False) if n_shared else None

    def _route(self, logits: torch.Tensor, k: int, renormalize: bool):
        scores = torch.softmax(logits.float(), dim=-1)
        _, ids = torch.topk(scores + self.e_score_correction_bias.view(1, -1), k, dim=-1)
        w = scores.gather(-1, ids)
        w = w / w.sum(-1, keepdim=True).clamp(min=self.norm_min)
        return w, ids.to(torch.int32)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        logits = F.linear(x.float(), self.gate.weight.float())
        out = self.experts(x, logits)
        if self.shared_experts is not None:
            out = out + self.shared_experts(x)
            if self.tp > 1 or self.experts.ep_size > 1:
                out = tensor_model_parallel_all_reduce(out)
        return out


class Ernie45MoeDecoderLayer(Ernie45DecoderLayer):
    supports_sequence_parallel = False

    def _make_mlp(self, config, vllm_config, quant_config, prefix):
        idx = int(prefix.split(".")[-2])
        start = int(getattr(config, "moe_layer_start_index", 0) or 0)
        end = getattr(config, "moe_layer_end_index", None)
        end = config.num_hidden_layers - 1 if end is None or end < 0 else int(end)
        step = int(getattr(config, "moe_layer_interval", 1) or 1)
        if int(getattr(config, "moe_num_experts", 0) or 0) > 0 and start <= idx <= end and (idx + 1) % step == 0:
            return Ernie45MoE(config, quant_config, prefix)
        return super()._make_mlp(config, vllm_config, quant_config, prefix)

    def forward(self, positions, hidden_states, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.mlp(hidden_states), residual


class Ernie45MoeModel(LlamaModel):
    layer_cls = Ernie45MoeDecoderLayer


class Ernie4_5_MoeForCausalLM(LlamaForCausalLM):
    model_cls = Ernie45MoeModel
    supports_lora = False

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]) -> set[str]:
        experts = FusedMoE.make_expert_params_mapping("gate_proj", "down_proj", "up_proj",
                                                      int(getattr(self.config, "moe_num_experts", 0) or 0))

        def rename(name: str):
            if ".mtp_" in name or name.startswith("model.mtp"):           # multi-token-prediction extras
                return None
            if name.endswith(".mlp.moe_statics.e_score_correction_bias"):
                return name.replace(".moe_statics.", ".")
            return name

        def fix(items):
            for name, t in items:
                if name.endswith("e_score_correction_bias") and t.dim() > 1:
                    t = t.reshape(t.shape[-1])                 # stored [1, E]
                yield name, t

        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping, tie_lm_head=self.tie,
                                         expert_mapping=experts, rename=rename)


# ---- dots.llm1 ----------------------------------------------------------------------------------------
class Dots1DecoderLayer(LlamaDecoderLayer):
    supports_sequence_parallel = False

    def _attention_extras(self, config, layer_idx: int) -> dict:
        kinds = getattr(config, "layer_types", None)
        window = getattr(config, "sliding_window", None) if kinds and kinds[layer_idx] == "sliding_attention" else None
        b = bool(getattr(config, "attention_bias", False))
        return {"qk_norm": True, "sliding_window": window, "bias": b, "o_bias": b}

    def _make_mlp(self, config, vllm_config, quant_config, prefix):
        idx = int(prefix.split(".")[-2])
        if getattr(config, "n_routed_experts", None) and idx >= int(getattr(config, "first_k_dense_replace", 0) or 0):
            if getattr(config, "topk_method", None) is None:
                config.topk_method = "noaux_tc"      # sigmoid scores, grouped selection with the learned bias
            return DeepseekMoE(config, quant_config, prefix)
        return LlamaMLP(config.hidden_size, config.intermediate_size, False, quant_config, prefix)

    def forward(self, positions, hidden_states, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.mlp(hidden_states), residual


class Dots1Model(LlamaModel):
    layer_cls = Dots1DecoderLayer


class Dots1ForCausalLM(LlamaForCausalLM):
    model_cls = Dots1Model
    supports_lora = False

    def load_weights(self, weights):
        experts = FusedMoE.make_expert_params_mapping("gate_proj", "down_proj", "up_proj",
                                                      int(getattr(self.config, "n_routed_experts", 0) or 0))
        return load_weights_with_mapping(self, weights, self.stacked_params_mapping, tie_lm_head=self.tie,
                                         expert_mapping=experts)


# ---- HunYuan v1 ---------------------------------------------------------------------------------------
class PostRopeNormAttention(LlamaAttention):
    """q/k RMSNorm per head applied to the ROTATED vectors (HunYuan), so the rotary embedding cannot be
    fused with the cache write: rotate in place, normalise, then hand the finished q/k/v to the cache."""

    def __init__(self, *a, norm_eps: float = 1e-5, **kw):
        super().__init__(*a, **kw)
        self.query_layernorm = RMSNorm(self.head_dim, norm_eps)
        self.key_layernorm = RMSNorm(self.head_dim, norm_eps)

    def forward(self, positions: torch.Tensor, hidden_states: torch.Tensor) -> torch.Tensor:
        qkv = self.qkv_proj(hidden_states)
        T = qkv.shape[0]
        q, k = qkv[:, : self.q_size], qkv[:, self.q_size: self.q_size + self.kv_size]
        self.rotary_emb(positions[:T], q, k)
        qkv[:, : self.q_size] = self.query_layernorm(q.reshape(T * self.num_heads, self.head_dim)).view(T, self.q_size)
        qkv[:, self.q_size: self.q_size + self.kv_size] = \
            self.key_layernorm(k.reshape(T * self.num_kv_heads, self.head_dim)).view(T, self.kv_size)
        return self.o_proj(self.attn(qkv, None, None))


def _per_layer(v, idx: int):
    return v[idx] if isinstance(v, (list, tuple)) else v


class HunYuanMoE(nn.Module):
    def __init__(self, config, layer_idx: int, quant_config=None, prefix: str = ""):
        super().__init__()
        H = config.hidden_size
        E = int(_per_layer(config.num_experts, layer_idx))
        k = int(_per_layer(config.moe_topk, layer_idx))
        I = int(_per_layer(getattr(config, "moe_intermediate_size", None) or config.intermediate_size, layer_idx))
        n_shared = int(_per_layer(getattr(config, "num_shared_expert", 1) or 1, layer_idx))
        self.tp = get_tensor_model_parallel_world_size()
        self.gate = nn.Module()
        self.gate.wg = ReplicatedLinear(H, E, bias=False, quant_config=None, prefix=f"{prefix}.gate.wg")
        self.experts = FusedMoE(E, k, H, I, reduce_results=False, renormalize=True, quant_config=quant_config,
                                prefix=f"{prefix}.experts")
        self.shared_mlp = LlamaMLP(H, int(config.intermediate_size) * n_shared, False, quant_config,
                                   f"{prefix}.shared_mlp", reduce_results=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        logits = F.linear(x.float(), self.gate.wg.weight.float())
        out = self.experts(x, logits) + self.shared_mlp(x)
        if self.tp > 1 or self.experts.ep_size > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out


class HunYuanDecoderLayer(LlamaDecoderLayer):
    supports_sequence_parallel = False

    def __init__(self, config, vllm_config, quant_config=None, prefix: str = "", attention_only: bool = False):
        if getattr(config, "use_cla", False):
            raise NotImplementedError("HunYuan checkpoints with cross-layer attention (use_cla) share K/V between "
                                      "layers; that layout is not supported")
        super().__init__(config, vllm_config, quant_config, prefix, attention_only)
        mc = vllm_config.model_config
        layer_idx = int(prefix.split(".")[-1])
        rope_theta, rope_scaling = self._rope_of(config, layer_idx)
        if rope_scaling and rope_scaling.get("rope_type", rope_scaling.get("type")) == "dynamic" \
                and rope_scaling.get("alpha"):
            # NTK-alpha scaling: a fixed rescaling of the base
            d = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
            rope_theta = rope_theta * float(rope_scaling["alpha"]) ** (d / (d - 2))
            rope_scaling = None
        b = bool(getattr(config, "attention_bias", False))
        self.self_attn = PostRopeNormAttention(
            config, config.hidden_size, config.num_attention_heads,
            getattr(config, "num_key_value_heads", None) or config.num_attention_heads, rope_theta, rope_scaling,
            mc.max_model_len, bias=b, o_bias=b, quant_config=quant_config, prefix=f"{prefix}.self_attn",
            cache_dtype=vllm_config.cache_config.cache_dtype, norm_eps=float(getattr(config, "rms_norm_eps", 1e-5)))

    def _make_mlp(self, config, vllm_config, quant_config, prefix):
        idx = int(prefix.split(".")[-2])
        if int(_per_layer(getattr(config, "num_experts", 1) or 1, idx)) > 1:
            return HunYuanMoE(config, idx, quant_config, prefix)
        return LlamaMLP(config.hidden_size, config.intermediate_size, bool(getattr(config, "mlp_bias", False)),
                        quant_config, prefix)

    def forward(self, positions, hidden_states, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.mlp(hidden_states), residual


class HunYuanModel(LlamaModel):
    layer_cls = HunYuanDecoderLayer


class HunYuanDenseV1ForCausalLM(LlamaForCausalLM):
    model_cls = HunYuanModel
    supports_lora = False


class HunYuanMoEV1ForCausalLM(HunYuanDenseV1ForCausalLM):
    def load_weights(self, weights):
        n = getattr(self.config, "num_experts", 1)
        E = max(n) if isinstance(n, (list, tuple)) else int(n)
        experts = FusedMoE.make_expert_params_mapping("gate_proj", "down_proj", "up_proj", E)
        return load_weights_with_mapping(self, weights, self.stacked_params_mapping, tie_lm_head=self.tie,
                                         expert_mapping=experts)


# ---- MiMo ---------------------------------------------------------------------------------------------
class MiMoForCausalLM(LlamaForCausalLM):
    """Qwen2 weights plus ``model.mtp_layers`` (multi-token prediction heads used for speculative
    decoding only): the serving model skips them."""

    def load_weights(self, weights):
        return load_weights_with_mapping(self, weights, self.stacked_params_mapping, tie_lm_head=self.tie,
                                         rename=lambda n: None if ".mtp_layers." in n else n)


# ---- MiniCPM ------------------------------------------------------------------------------------------
class MiniCPMDecoderLayer(LlamaDecoderLayer):
    supports_sequence_parallel = False

    def __init__(self, config, vllm_config, quant_config=None, prefix: str = "", attention_only: bool = False):
        super().__init__(config, vllm_config, quant_config, prefix, attention_only)
        self.branch_scale = float(getattr(config, "scale_depth", 1.0)) / math.sqrt(config.num_hidden_layers)

    def forward(self, positions, hidden_states, residual):
        # x + s * f(norm(x)): the scaled branch is folded into the residual stream through the fused add + norm
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states) * self.branch_scale
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.mlp(hidden_states) * self.branch_scale, residual


class MiniCPMModel(LlamaModel):
    layer_cls = MiniCPMDecoderLayer
    embed_scale_of = staticmethod(lambda config: float(getattr(config, "scale_emb", 1.0) or 1.0))


class MiniCPMForCausalLM(LlamaForCausalLM):
    model_cls = MiniCPMModel

    @staticmethod
    def _logit_scale(config) -> float:
        base = float(getattr(config, "dim_model_base", 0) or 0)
        return base / config.hidden_size if base else 1.0


# ---- MiniCPM3 -----------------------------------------------------------------------------------------
class MiniCPM3DecoderLayer(nn.Module):
    """MiniCPM's scaled residual branches around multi-head latent attention (the DeepSeek-V2 attention with
    half-split rotary pairs) and a dense gated MLP.  Reference: vllm/model_executor/models/minicpm3.py."""
    make_norm = classmethod(lambda cls, config: RMSNorm(config.hidden_size, getattr(config, "rms_norm_eps", 1e-6)))

    def __init__(self, config, vllm_config, quant_config=None, prefix: str = ""):
        super().__init__()
        from .deepseek import DeepseekAttention
        from .llama import LlamaMLP
        self.self_attn = DeepseekAttention(config, vllm_config, quant_config, f"{prefix}.self_attn")
        self.mlp = LlamaMLP(config.hidden_size, config.intermediate_size, False, quant_config, f"{prefix}.mlp")
        self.input_layernorm = self.make_norm(config)
        self.post_attention_layernorm = self.make_norm(config)
        self.branch_scale = float(getattr(config, "scale_depth", 1.0)) / math.sqrt(config.num_hidden_layers)

    def forward(self, positions, hidden_states, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states) * self.branch_scale
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.mlp(hidden_states) * self.branch_scale, residual


class MiniCPM3Model(MiniCPMModel):
    layer_cls = MiniCPM3DecoderLayer


class MiniCPM3ForCausalLM(MiniCPMForCausalLM):
    model_cls = MiniCPM3Model
    stacked_params_mapping = [(".gate_up_proj", ".gate_proj", 0), (".gate_up_proj", ".up_proj", 1)]
    packed_modules_mapping = {"gate_up_proj": ["gate_proj", "up_proj"]}
    supports_lora = False


# ---- Persimmon ----------------------------------------------------------------------------------------
class PersimmonAttention(nn.Module):
    def __init__(self, config, vllm_config, quant_config=None, prefix: str = ""):
        super().__init__()
        from ..layers.attention import Attention
        from ..layers.rotary import get_rope
        tp = get_tensor_model_parallel_world_size()
        H, nh = config.hidden_size, config.num_attention_heads
        self.num_heads, self.head_dim = nh // tp, H // nh
        self.q_size = self.num_heads * self.head_dim
        self.query_key_value = QKVParallelLinear(H, self.head_dim, nh, nh, bias=True, quant_config=quant_config,
                                                 prefix=f"{prefix}.query_key_value")
        self.dense = RowParallelLinear(H, H, bias=True, quant_config=quant_config, prefix=f"{prefix}.dense")
        self.qk_layernorm = bool(getattr(config, "qk_layernorm", True))
        if self.qk_layernorm:
            eps = float(getattr(config, "layer_norm_eps", 1e-5))
            self.q_layernorm = LayerNorm(self.head_dim, eps)
            self.k_layernorm = LayerNorm(self.head_dim, eps)
        rs = getattr(config, "rope_scaling", None) or getattr(config, "rope_parameters", None) or {}
        theta = getattr(config, "rope_theta", None) or rs.get("rope_theta") or 25000.0
        partial = getattr(config, "partial_rotary_factor", None) or rs.get("partial_rotary_factor") or 0.5
        self.rotary_emb = get_rope(self.head_dim, self.head_dim, vllm_config.model_config.max_model_len, float(theta),
                                   True, None, float(partial))
        self.attn = Attention(self.num_heads, self.head_dim, self.head_dim ** -0.5, self.num_heads,
                              prefix=f"{prefix}.attn", cache_dtype=vllm_config.cache_config.cache_dtype)

    def forward(self, positions, hidden_states):
        qkv = self.query_key_value(hidden_states)
        if self.qk_layernorm:
            T = qkv.shape[0]
            q = qkv[:, : self.q_size].reshape(T * self.num_heads, self.head_dim)
            k = qkv[:, self.q_size: 2 * self.q_size].reshape(T * self.num_heads, self.head_dim)
            qkv[:, : self.q_size] = self.q_layernorm(q).view(T, self.q_size)
            qkv[:, self.q_size: 2 * self.q_size] = self.k_layernorm(k).view(T, self.q_size)
        return self.dense(self.attn(qkv, positions, self.rotary_emb))


class PersimmonMLP(nn.Module):
    def __init__(self, config, quant_config=None, prefix: str = ""):
        super().__init__()
        self.dense_h_to_4h = ColumnParallelLinear(config.hidden_size, config.intermediate_size, bias=True,
                                                  quant_config=quant_config, prefix=f"{prefix}.dense_h_to_4h")
        self.dense_4h_to_h = RowParallelLinear(config.intermediate_size, config.hidden_size, bias=True,
                                               quant_config=quant_config, prefix=f"{prefix}.dense_4h_to_h")

    def forward(self, x):
        return self.dense_4h_to_h(ops.activation(self.dense_h_to_4h(x), "relu2"))


class PersimmonDecoderLayer(nn.Module):
    def __init__(self, config, vllm_config, quant_config=None, prefix: str = ""):
        super().__init__()
        eps = float(getattr(config, "layer_norm_eps", 1e-5))
        self.self_attn = PersimmonAttention(config, vllm_config, quant_config, f"{prefix}.self_attn")
        self.mlp = PersimmonMLP(config, quant_config, f"{prefix}.mlp")
        self.input_layernorm = LayerNorm(config.hidden_size, eps)
        self.post_attention_layernorm = LayerNorm(config.hidden_size, eps)

    def forward(self, positions, h):
        h = h + self.self_attn(positions, self.input_layernorm(h))
        return h + self.mlp(self.post_attention_layernorm(h))


class PersimmonForCausalLM(nn.Module):
    supports_lora = False
    supports_pp = False

    def __init__(self, vllm_config, prefix: str = ""):
        super().__init__()
        from ..layers.basic import LogitsProcessor, ParallelLMHead, VocabParallelEmbedding
        config = vllm_config.model_config.hf_config
        quant_config = getattr(vllm_config, "quant_config", None)
        self.config = config
        self.model = nn.Module()
        self.model.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.model.layers = nn.ModuleList([PersimmonDecoderLayer(config, vllm_config, quant_config, f"model.layers.{i}")
                                           for i in range(config.num_hidden_layers)])
        self.model.final_layernorm = LayerNorm(config.hidden_size, float(getattr(config, "layer_norm_eps", 1e-5)))
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        self.tie = bool(getattr(config, "tie_word_embeddings", False))
        if self.tie:
            self.lm_head.tie_weights(self.model.embed_tokens)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    def get_input_embeddings(self, ids):
        return self.model.embed_tokens(ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        h = inputs_embeds if inputs_embeds is not None else self.model.embed_tokens(input_ids)
        for layer in self.model.layers:
            h = layer(positions, h)
        return self.model.final_layernorm(h)

    def compute_logits(self, hidden):
        return self.logits_processor(self.lm_head, hidden)

    def load_weights(self, weights):
        nh = self.config.num_attention_heads
        hd = self.config.hidden_size // nh

        def fix(items):
            for name, t in items:
                if ".self_attn.query_key_value." in name:
                    # stored head by head as (q, k, v) triples: regroup into all-q, all-k, all-v rows
                    parts = t.view(nh, 3, hd, *t.shape[1:]).transpose(0, 1).reshape(3, nh * hd, *t.shape[1:])
                    for i, part in enumerate(("q", "k", "v")):
                        yield name.replace("query_key_value", f"query_key_value__{part}"), parts[i].contiguous()
                    continue
                yield name, t

        stacked = [(".query_key_value", f".query_key_value__{p}", p) for p in ("q", "k", "v")]
        return load_weights_with_mapping(self, fix(weights), stacked, tie_lm_head=self.tie)
