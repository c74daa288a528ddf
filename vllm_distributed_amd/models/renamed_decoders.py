"""Decoder families that are the Llama block under other checkpoint names and config keys: EXAONE 3,
TeleChat2, TeleFLM, Orion, Qwen (v1), ChatGLM / GLM-2/3/4 in their original layout, GritLM.

Their config files use other key names (``num_layers``, ``layer_norm_epsilon``, ``ffn_hidden_size``, ...);
``normalise_config`` rewrites them to the names the Llama block reads, so the block, the fused projections and
the kernels are shared and only the weight routing differs.
Reference: vllm/model_executor/models/{exaone,telechat2,teleflm,orion,qwen,chatglm,gritlm}.py and
vllm/transformers_utils/configs/{exaone,telechat2,chatglm}.py.
"""
from __future__ import annotations

import torch

from .llama import LlamaDecoderLayer, LlamaForCausalLM, LlamaMLP, LlamaModel
from .llama_variants import ResidualLayerNorm
from .utils import load_weights_with_mapping


def normalise_config(d: dict) -> dict:
    """Config keys of these families -> the names used throughout the engine (returns a new dict)."""
    mt = d.get("model_type")
    out = dict(d)

    def alias(new: str, old: str, default=None):
        if new not in out or out[new] is None:
            if old in d and d[old] is not None:
                out[new] = d[old]
            elif default is not None:
                out[new] = default

    if mt == "exaone":
        alias("num_hidden_layers", "num_layers")
        alias("rms_norm_eps", "layer_norm_epsilon")
        alias("hidden_act", "activation_function", "silu")
    elif mt == "telechat":
        alias("num_hidden_layers", "n_layer")
        alias("num_attention_heads", "n_head")
        alias("intermediate_size", "ffn_hidden_size")
        alias("rms_norm_eps", "layer_norm_epsilon")
        alias("max_position_embeddings", "training_seqlen", 8192)
        alias("num_key_value_heads", "n_head")
    elif mt == "qwen":
        alias("rms_norm_eps", "layer_norm_epsilon")
        alias("rope_theta", "rotary_emb_base", 10000.0)
        alias("max_position_embeddings", "seq_length", 8192)
        alias("num_key_value_heads", "num_attention_heads")
        alias("head_dim", "kv_channels")
    elif mt == "chatglm":
        alias("num_hidden_layers", "num_layers")
        if d.get("multi_query_attention"):
            alias("num_key_value_heads", "multi_query_group_num")
        alias("num_key_value_heads", "num_attention_heads")
        alias("intermediate_size", "ffn_hidden_size")
        alias("rms_norm_eps", "layernorm_epsilon")
        out["org_vocab_size"] = d.get("vocab_size", d.get("padded_vocab_size"))
        if d.get("padded_vocab_size"):
            out["vocab_size"] = d["padded_vocab_size"]
        alias("head_dim", "kv_channels")
        alias("max_position_embeddings", "seq_length", 8192)
        out.setdefault("partial_rotary_factor", 0.5)
        out.setdefault("rope_theta", 10000.0 * float(d.get("rope_ratio", 1.0)))
        out.setdefault("attention_bias", bool(d.get("add_qkv_bias") or d.get("add_bias_linear")))
        out.setdefault("tie_word_embeddings", False)
    return out


def _renamed(pairs):
    def f(name: str) -> str:
        for a, b in pairs:
            name = name.replace(a, b)
        return name
    return f


# ---- EXAONE 3 ----------------------------------------------------------------------------------------------
class ExaoneForCausalLM(LlamaForCausalLM):
    """``transformer.h.N.attn.attention.{q,k,v,out}_proj``, ``mlp.c_fc_0`` (gate) / ``c_fc_1`` (up) / ``c_proj``."""
    _names = staticmethod(_renamed((("transformer.wte.", "model.embed_tokens."), ("transformer.ln_f.", "model.norm."),
                       ("transformer.h.", "model.layers."), (".attn.attention.out_proj.", ".self_attn.o_proj."),
                       (".attn.attention.", ".self_attn."), (".ln_1.", ".input_layernorm."),
                       (".ln_2.", ".post_attention_layernorm."), (".mlp.c_fc_0.", ".mlp.gate_proj."),
                       (".mlp.c_fc_1.", ".mlp.up_proj."), (".mlp.c_proj.", ".mlp.down_proj."))))

    def load_weights(self, weights):
        return load_weights_with_mapping(self, ((self._names(n), t) for n, t in weights),
                                         self.stacked_params_mapping, tie_lm_head=self.tie)


# ---- TeleChat2 ---------------------------------------------------------------------------------------------
class TeleChat2DecoderLayer(LlamaDecoderLayer):
    """Biases on the attention output and the MLP down projection only."""
    supports_sequence_parallel = False

    def _attention_extras(self, config, layer_idx: int) -> dict:
        return {"bias": False, "o_bias": True}

    def _make_mlp(self, config, vllm_config, quant_config, prefix):
        mlp = LlamaMLP(config.hidden_size, config.intermediate_size, True, quant_config, prefix)
        mlp.gate_up_proj.bias = None
        return mlp


class TeleChat2Model(LlamaModel):
    layer_cls = TeleChat2DecoderLayer


class TeleChat2ForCausalLM(LlamaForCausalLM):
    """``self_attention.query`` + ``self_attention.key_value`` (per KV head: its key rows, then its value rows)."""
    model_cls = TeleChat2Model
    _names = staticmethod(_renamed((("transformer.word_embeddings.", "model.embed_tokens."), ("transformer.ln_f.", "model.norm."),
                       ("transformer.h.", "model.layers."), (".self_attention.dense.", ".self_attn.o_proj."),
                       (".self_attention.", ".self_attn."))))

    def load_weights(self, weights):
        c = self.config
        KV = getattr(c, "num_key_value_heads", None) or c.num_attention_heads
        D = c.hidden_size // c.num_attention_heads

        def fix(items):
            for name, t in items:
                name = self._names(name)
                if ".self_attn.key_value." in name:
                    t = t.view(KV, 2, D, *t.shape[1:])
                    yield name.replace(".key_value.", ".k_proj."), t[:, 0].reshape(KV * D, *t.shape[3:])
                    yield name.replace(".key_value.", ".v_proj."), t[:, 1].reshape(KV * D, *t.shape[3:])
                elif ".self_attn.query." in name:
                    yield name.replace(".query.", ".q_proj."), t
                else:
                    yield name, t
        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping, tie_lm_head=self.tie)


# ---- TeleFLM -----------------------------------------------------------------------------------------------
class TeleFLMModel(LlamaModel):
    """Llama with maximal-update parametrisation: the embedding output is scaled by ``input_mult``."""
    embed_scale_of = staticmethod(lambda c: float(getattr(c, "input_mult", 1.0)) if getattr(c, "use_mup", False) else 1.0)


class TeleFLMForCausalLM(LlamaForCausalLM):
    model_cls = TeleFLMModel

    @staticmethod
    def _logit_scale(config) -> float:
        if getattr(config, "use_mup", False):
            return float(config.output_mult) / float(config.mup_scale_factor)
        return 1.0


# ---- Orion -------------------------------------------------------------------------------------------------
class OrionDecoderLayer(LlamaDecoderLayer):
    """Llama block with LayerNorm (scale and shift) in place of RMSNorm."""
    supports_sequence_parallel = False

    @classmethod
    def make_norm(cls, config):
        return ResidualLayerNorm(config.hidden_size, getattr(config, "rms_norm_eps", 1e-5))


class OrionModel(LlamaModel):
    layer_cls = OrionDecoderLayer


class OrionForCausalLM(LlamaForCausalLM):
    model_cls = OrionModel


# ---- Qwen (first generation) -------------------------------------------------------------------------------
class QWenDecoderLayer(LlamaDecoderLayer):
    supports_sequence_parallel = False

    def _attention_extras(self, config, layer_idx: int) -> dict:
        return {"bias": True, "o_bias": False}

    def _make_mlp(self, config, vllm_config, quant_config, prefix):
        # the config counts gate and up together
        return LlamaMLP(config.hidden_size, config.intermediate_size // 2, False, quant_config, prefix)


class QWenModel(LlamaModel):
    layer_cls = QWenDecoderLayer


class QWenLMHeadModel(LlamaForCausalLM):
    """``attn.c_attn`` is q, k, v stacked; ``mlp.w2`` gates ``mlp.w1``."""
    model_cls = QWenModel
    packed_modules_mapping = {"qkv_proj": ["c_attn"], "gate_up_proj": ["w2", "w1"]}
    _names = staticmethod(_renamed((("transformer.wte.", "model.embed_tokens."), ("transformer.ln_f.", "model.norm."),
                       ("transformer.h.", "model.layers."), (".attn.c_proj.", ".self_attn.o_proj."),
                       (".attn.c_attn.", ".self_attn.c_attn."), (".ln_1.", ".input_layernorm."),
                       (".ln_2.", ".post_attention_layernorm."), (".mlp.w2.", ".mlp.gate_proj."),
                       (".mlp.w1.", ".mlp.up_proj."), (".mlp.c_proj.", ".mlp.down_proj."))))

    def load_weights(self, weights):
        d = self.config.hidden_size

        def fix(items):
            for name, t in items:
                if ".visual." in name:
                    continue                     # Qwen-VL's vision tower: text decoding only
                name = self._names(name)
                if ".self_attn.c_attn." in name:
                    for i, proj in enumerate(("q_proj", "k_proj", "v_proj")):
                        yield name.replace(".c_attn.", f".{proj}."), t[i * d:(i + 1) * d]
                else:
                    yield name, t
        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping, tie_lm_head=self.tie)


# ---- ChatGLM (GLM-2 / 3 / 4 in the original layout) --------------------------------------------------------
class ChatGLMDecoderLayer(LlamaDecoderLayer):
    """Rotary embedding over half of each head, pairs interleaved unless ``original_rope`` is off."""
    supports_sequence_parallel = False

    def _attention_extras(self, config, layer_idx: int) -> dict:
        if getattr(config, "apply_residual_connection_post_layernorm", False) or not getattr(config, "rmsnorm", True):
            raise ValueError("ChatGLM checkpoints with post-LayerNorm residuals or LayerNorm blocks are not supported")
        return {"neox_style": not getattr(config, "original_rope", True),
                "bias": bool(getattr(config, "attention_bias", False)),
                "o_bias": bool(getattr(config, "add_bias_linear", False))}


class ChatGLMInnerModel(LlamaModel):
    layer_cls = ChatGLMDecoderLayer


class ChatGLMModel(LlamaForCausalLM):
    """``query_key_value`` is q, k, v stacked (grouped-query sizes), ``dense_h_to_4h`` gate and up stacked."""
    model_cls = ChatGLMInnerModel
    packed_modules_mapping = {"qkv_proj": ["query_key_value"], "gate_up_proj": ["dense_h_to_4h"]}
    _names = staticmethod(_renamed((("transformer.embedding.word_embeddings.", "model.embed_tokens."),
                       ("transformer.encoder.final_layernorm.", "model.norm."),
                       ("transformer.output_layer.", "lm_head."),
                       ("transformer.encoder.layers.", "model.layers."),
                       (".self_attention.dense.", ".self_attn.o_proj."), (".self_attention.", ".self_attn."),
                       (".mlp.dense_4h_to_h.", ".mlp.down_proj."))))

    def __init__(self, vllm_config, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        org = getattr(self.config, "org_vocab_size", None)
        if org and hasattr(self, "logits_processor"):
            self.logits_processor.org_vocab_size = int(org)

    def load_weights(self, weights):
        c = self.config
        H = c.num_attention_heads
        KV = getattr(c, "num_key_value_heads", None) or H
        D = getattr(c, "head_dim", None) or c.hidden_size // H
        F = c.intermediate_size

        def fix(items):
            for name, t in items:
                if "rotary_pos_emb" in name or ".vision." in name:
                    continue
                name = self._names(name)
                if ".self_attn.query_key_value." in name:
                    q, k, v = t.split([H * D, KV * D, KV * D], dim=0)
                    for proj, part in (("q_proj", q), ("k_proj", k), ("v_proj", v)):
                        yield name.replace(".query_key_value.", f".{proj}."), part
                elif ".mlp.dense_h_to_4h." in name:
                    yield name.replace(".dense_h_to_4h.", ".gate_proj."), t[:F]
                    yield name.replace(".dense_h_to_4h.", ".up_proj."), t[F:]
                else:
                    yield name, t
        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping, tie_lm_head=self.tie)


ChatGLMForConditionalGeneration = ChatGLMModel


# ---- GritLM ------------------------------------------------------------------------------------------------
class GritLM(LlamaForCausalLM):
    """Mistral weights trained for generation and for embeddings; generation is the plain causal model
    (the embedding mode's bidirectional instruction pooling is not offered)."""
