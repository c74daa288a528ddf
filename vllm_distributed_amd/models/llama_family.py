"""Decoder families that are the Llama block with small changes: Gemma / Gemma-2 (offset RMSNorm,
GeGLU, scaled embeddings, soft-capped logits, sandwich norms, alternating sliding window), Phi-3
(fused projections in the checkpoint), Granite (embedding / residual / attention / logit multipliers).
Reference: vllm/model_executor/models/gemma.py, gemma2.py, phi3.py, granite.py.
"""
from __future__ import annotations

import torch

from ..layers.basic import GemmaRMSNorm
from .llama import LlamaDecoderLayer, LlamaForCausalLM, LlamaMLP, LlamaModel
from .utils import load_weights_with_mapping


# ---- Gemma -------------------------------------------------------------------------------------------
class GemmaDecoderLayer(LlamaDecoderLayer):
    norm_cls = GemmaRMSNorm
    supports_sequence_parallel = False

    def _make_mlp(self, config, vllm_config, quant_config, prefix):
        act = getattr(config, "hidden_activation", None) or getattr(config, "hidden_act", "gelu_pytorch_tanh")
        return LlamaMLP(config.hidden_size, config.intermediate_size, False, quant_config, prefix,
                        act="gelu_tanh" if "tanh" in act or act == "gelu" else act)


class GemmaModel(LlamaModel):
    layer_cls = GemmaDecoderLayer
    embed_scale_of = staticmethod(lambda config: config.hidden_size ** 0.5)


class GemmaForCausalLM(LlamaForCausalLM):
    model_cls = GemmaModel


class Gemma2DecoderLayer(GemmaDecoderLayer):
    def _attention_extras(self, config, layer_idx: int) -> dict:
        kinds = getattr(config, "layer_types", None)
        sliding = (kinds[layer_idx] == "sliding_attention") if kinds else (layer_idx % 2 == 0)
        return {"scale": float(config.query_pre_attn_scalar) ** -0.5,
                "logits_soft_cap": getattr(config, "attn_logit_softcapping", None),
                "sliding_window": getattr(config, "sliding_window", None) if sliding else None}

    def __init__(self, config, vllm_config, quant_config=None, prefix: str = "", attention_only: bool = False):
        super().__init__(config, vllm_config, quant_config, prefix, attention_only)
        eps = getattr(config, "rms_norm_eps", 1e-6)
        self.pre_feedforward_layernorm = GemmaRMSNorm(config.hidden_size, eps)
        self.post_feedforward_layernorm = GemmaRMSNorm(config.hidden_size, eps)

    def forward(self, positions, hidden_states, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.post_attention_layernorm(self.self_attn(positions, hidden_states))
        hidden_states, residual = self.pre_feedforward_layernorm(hidden_states, residual)
        hidden_states = self.post_feedforward_layernorm(self.mlp(hidden_states))
        return hidden_states, residual


class Gemma2Model(GemmaModel):
    layer_cls = Gemma2DecoderLayer


class Gemma2ForCausalLM(LlamaForCausalLM):
    model_cls = Gemma2Model


# ---- Phi-3 -------------------------------------------------------------------------------------------
class Phi3ForCausalLM(LlamaForCausalLM):
    """Llama block; the checkpoint already stores qkv_proj and gate_up_proj fused, which the fused
    layers' loaders split per TP rank."""
    stacked_params_mapping = []
    packed_modules_mapping = {"qkv_proj": ["qkv_proj"], "gate_up_proj": ["gate_up_proj"]}


# ---- Granite -----------------------------------------------------------------------------------------
class GraniteDecoderLayer(LlamaDecoderLayer):
    supports_sequence_parallel = False

    def _attention_extras(self, config, layer_idx: int) -> dict:
        return {"scale": float(getattr(config, "attention_multiplier", None) or
                               (config.hidden_size // config.num_attention_heads) ** -0.5)}

    def __init__(self, config, vllm_config, quant_config=None, prefix: str = "", attention_only: bool = False):
        super().__init__(config, vllm_config, quant_config, prefix, attention_only)
        self.residual_multiplier = float(getattr(config, "residual_multiplier", 1.0) or 1.0)

    def forward(self, positions, hidden_states, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states) * self.residual_multiplier
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.mlp(hidden_states) * self.residual_multiplier, residual


class GraniteModel(LlamaModel):
    layer_cls = GraniteDecoderLayer
    embed_scale_of = staticmethod(lambda config: float(getattr(config, "embedding_multiplier", 1.0) or 1.0))


class GraniteForCausalLM(LlamaForCausalLM):
    model_cls = GraniteModel

    @staticmethod
    def _logit_scale(config) -> float:
        return 1.0 / float(getattr(config, "logits_scaling", 1.0) or 1.0)
