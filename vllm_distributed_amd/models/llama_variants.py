"""More decoder families built from the Llama block: StableLM, OLMo, OLMo-2, Cohere (Command-R),
Nemotron, GLM / GLM-4, Gemma-3 (text), InternLM2 and Baichuan.  Each differs from Llama in its
normalisation layer, the residual wiring, the rotary convention or the checkpoint layout only, so
they share the fused QKV / gate-up projections, the fused RoPE + cache-write kernel and the paged
attention kernels.
Reference: vllm/model_executor/models/{stablelm,olmo,olmo2,commandr,nemotron,glm,glm4,gemma3,
internlm2,baichuan}.py.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from ..layers.basic import GemmaRMSNorm, LayerNorm, RMSNorm, get_act_fn
from ..layers.linear import ColumnParallelLinear, RowParallelLinear
from .llama import LlamaDecoderLayer, LlamaForCausalLM, LlamaMLP, LlamaModel
from .llama_family import Gemma2DecoderLayer, GemmaModel
from .utils import load_weights_with_mapping


class ResidualLayerNorm(LayerNorm):
    """LayerNorm with the (x, residual) -> (norm(x + residual), x + residual) form the Llama block uses."""

    def forward(self, x, residual: Optional[torch.Tensor] = None):
        if residual is None:
            return super().forward(x)
        residual.add_(x)
        return super().forward(residual), residual


class PlainLayerNorm(nn.Module):
    """LayerNorm without learnable scale or shift (OLMo)."""

    def __init__(self, hidden_size: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(hidden_size), persistent=False)

    def forward(self, x, residual: Optional[torch.Tensor] = None):
        if residual is None:
            return ops.layer_norm(x, self.weight, None, self.eps)
        residual.add_(x)
        return ops.layer_norm(residual, self.weight, None, self.eps), residual


def alibi_slopes(num_heads: int, alibi_bias_max: float = 8.0) -> list[float]:
    """Geometric head slopes of ALiBi (Press et al.); heads beyond a power of two take every other slope
    of the next power."""
    p = 2 ** math.floor(math.log2(num_heads))
    base = 2.0 ** (-(2.0 ** -(math.log2(p) - math.log2(alibi_bias_max))))
    slopes = [base ** (i + 1) for i in range(p)]
    if p != num_heads:
        extra = 2.0 ** (-(2.0 ** -(math.log2(2 * p) - math.log2(alibi_bias_max))))
        slopes += [extra ** (i + 1) for i in range(0, 2 * (num_heads - p), 2)]
    return slopes


# ---- StableLM ----------------------------------------------------------------------------------------
class StableLmDecoderLayer(LlamaDecoderLayer):
    supports_sequence_parallel = False

    @classmethod
    def make_norm(cls, config):
        return ResidualLayerNorm(config.hidden_size, getattr(config, "layer_norm_eps", 1e-5))

    def _attention_extras(self, config, layer_idx: int) -> dict:
        if getattr(config, "qk_layernorm", False):
            raise ValueError("StableLM checkpoints with per-head q/k LayerNorm are not supported")
        if getattr(config, "use_parallel_residual", False):
            raise ValueError("StableLM checkpoints with parallel residual blocks are not supported")
        return {"bias": bool(getattr(config, "use_qkv_bias", False)), "o_bias": False}


class StableLmModel(LlamaModel):
    layer_cls = StableLmDecoderLayer


class StableLmForCausalLM(LlamaForCausalLM):
    model_cls = StableLmModel


StablelmForCausalLM = StableLmForCausalLM


# ---- OLMo --------------------------------------------------------------------------------------------
class OlmoDecoderLayer(LlamaDecoderLayer):
    supports_sequence_parallel = False

    @classmethod
    def make_norm(cls, config):
        return PlainLayerNorm(config.hidden_size, 1e-5)

    def _attention_extras(self, config, layer_idx: int) -> dict:
        return {"qkv_clip": getattr(config, "clip_qkv", None)}


class OlmoModel(LlamaModel):
    layer_cls = OlmoDecoderLayer


class OlmoForCausalLM(LlamaForCausalLM):
    model_cls = OlmoModel


# ---- OLMo-2 ------------------------------------------------------------------------------------------
class Olmo2DecoderLayer(LlamaDecoderLayer):
    """Norms sit behind the attention and the MLP (x + norm(f(x))); q and k are normalised over their
    whole width before the rotary embedding."""
    supports_sequence_parallel = False

    def _attention_extras(self, config, layer_idx: int) -> dict:
        return {"qk_norm": "full"}

    def __init__(self, config, vllm_config, quant_config=None, prefix: str = "", attention_only: bool = False):
        super().__init__(config, vllm_config, quant_config, prefix, attention_only)
        if attention_only:
            return
        del self.input_layernorm
        self.post_feedforward_layernorm = self.make_norm(config)

    def forward(self, positions, hidden_states, residual):
        # ``residual`` is unused: the stream itself is carried in hidden_states
        h = hidden_states + self.post_attention_layernorm(self.self_attn(positions, hidden_states))
        return h + self.post_feedforward_layernorm(self.mlp(h)), None


class _NoResidualFinalNorm(LlamaModel):
    """Blocks that keep the whole stream in hidden_states hand ``None`` as residual to the last norm."""

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        from ..parallel.state import get_pp_group
        from .utils import IntermediateTensors
        pp = get_pp_group()
        if pp.is_first_rank:
            hidden = inputs_embeds if inputs_embeds is not None else self.embed_tokens(input_ids)
            if self.embed_scale != 1.0:
                hidden = hidden * self.embed_scale
        else:
            hidden = intermediate_tensors["hidden_states"]
        for layer in self.layers[self.start_layer:self.end_layer]:
            hidden, _ = layer(positions, hidden, None)
        if not pp.is_last_rank:
            return IntermediateTensors({"hidden_states": hidden, "residual": torch.zeros_like(hidden[:0])})
        return self.norm(hidden)


class Olmo2Model(_NoResidualFinalNorm):
    layer_cls = Olmo2DecoderLayer


class Olmo2ForCausalLM(LlamaForCausalLM):
    model_cls = Olmo2Model
    supports_pp = False


# ---- Cohere (Command-R) ------------------------------------------------------------------------------
class CohereDecoderLayer(LlamaDecoderLayer):
    """One LayerNorm (scale only) feeds attention and MLP side by side: x + attn(n(x)) + mlp(n(x)).
    Rotary pairs are interleaved."""
    supports_sequence_parallel = False

    @classmethod
    def make_norm(cls, config):
        return LayerNorm(config.hidden_size, getattr(config, "layer_norm_eps", 1e-5), bias=False)

    def _attention_extras(self, config, layer_idx: int) -> dict:
        if getattr(config, "use_qk_norm", False):      # per-head LayerNorm (scale only) on q and k before the rotation
            return {"neox_style": False, "qk_norm": "layernorm_nobias"}
        return {"neox_style": False}

    def __init__(self, config, vllm_config, quant_config=None, prefix: str = "", attention_only: bool = False):
        super().__init__(config, vllm_config, quant_config, prefix, attention_only)
        if not attention_only:
            del self.post_attention_layernorm

    def forward(self, positions, hidden_states, residual):
        n = self.input_layernorm(hidden_states)
        return hidden_states + self.self_attn(positions, n) + self.mlp(n), None


class CohereModel(_NoResidualFinalNorm):
    layer_cls = CohereDecoderLayer


class CohereForCausalLM(LlamaForCausalLM):
    model_cls = CohereModel
    supports_pp = False


class Cohere2DecoderLayer(CohereDecoderLayer):
    """Command-R7B / Command-A: sliding-window layers carry the rotary embedding, the full-attention
    layers between them use no positional encoding at all."""

    def _attention_extras(self, config, layer_idx: int) -> dict:
        kinds = getattr(config, "layer_types", None)
        if kinds:
            sliding = kinds[layer_idx] == "sliding_attention"
        else:
            sliding = (layer_idx + 1) % int(getattr(config, "sliding_window_pattern", 4)) != 0
        return {"neox_style": False, "use_rope": sliding,
                "sliding_window": getattr(config, "sliding_window", None) if sliding else None}


class Cohere2Model(_NoResidualFinalNorm):
    layer_cls = Cohere2DecoderLayer


class Cohere2ForCausalLM(LlamaForCausalLM):
    model_cls = Cohere2Model
    supports_pp = False


# ---- Nemotron ----------------------------------------------------------------------------------------
def _plus_one_loader(param, tensor):
    param.data.copy_(tensor.to(param.dtype) + 1.0)


class NemotronLayerNorm(ResidualLayerNorm):
    """LayerNorm whose stored scale is (w - 1); the loader adds the one back."""

    def __init__(self, hidden_size, eps):
        super().__init__(hidden_size, eps)
        self.weight.weight_loader = _plus_one_loader


class NemotronMLP(nn.Module):
    def __init__(self, hidden: int, inter: int, bias: bool, act: str, quant_config, prefix: str):
        super().__init__()
        self.up_proj = ColumnParallelLinear(hidden, inter, bias=bias, quant_config=quant_config,
                                            prefix=f"{prefix}.up_proj")
        self.down_proj = RowParallelLinear(inter, hidden, bias=bias, quant_config=quant_config,
                                           prefix=f"{prefix}.down_proj")
        self.act = get_act_fn(act)

    def forward(self, x):
        return self.down_proj(self.act(self.up_proj(x)))


class NemotronDecoderLayer(LlamaDecoderLayer):
    supports_sequence_parallel = False

    @classmethod
    def make_norm(cls, config):
        return NemotronLayerNorm(config.hidden_size, getattr(config, "norm_eps", 1e-5))

    def _make_mlp(self, config, vllm_config, quant_config, prefix):
        return NemotronMLP(config.hidden_size, config.intermediate_size, bool(getattr(config, "mlp_bias", False)),
                           config.hidden_act, quant_config, prefix)


class NemotronModel(LlamaModel):
    layer_cls = NemotronDecoderLayer


class NemotronForCausalLM(LlamaForCausalLM):
    model_cls = NemotronModel
    stacked_params_mapping = LlamaForCausalLM.stacked_params_mapping[:3]
    packed_modules_mapping = {"qkv_proj": ["q_proj", "k_proj", "v_proj"]}


# ---- GLM / GLM-4 -------------------------------------------------------------------------------------
class GlmDecoderLayer(LlamaDecoderLayer):
    """Llama block with interleaved rotary pairs over half of each head; gate and up projections are
    stored fused in the checkpoint."""
    supports_sequence_parallel = False

    def _attention_extras(self, config, layer_idx: int) -> dict:
        return {"neox_style": False, "bias": bool(getattr(config, "attention_bias", False)), "o_bias": False}


class GlmModel(LlamaModel):
    layer_cls = GlmDecoderLayer


class GlmForCausalLM(LlamaForCausalLM):
    model_cls = GlmModel
    stacked_params_mapping = LlamaForCausalLM.stacked_params_mapping[:3]
    packed_modules_mapping = {"qkv_proj": ["q_proj", "k_proj", "v_proj"], "gate_up_proj": ["gate_up_proj"]}


class Glm4DecoderLayer(GlmDecoderLayer):
    """GLM block with a second norm behind the attention and behind the MLP (sandwich norms)."""

    def __init__(self, config, vllm_config, quant_config=None, prefix: str = "", attention_only: bool = False):
        super().__init__(config, vllm_config, quant_config, prefix, attention_only)
        if attention_only:
            return
        self.post_self_attn_layernorm = self.make_norm(config)
        self.post_mlp_layernorm = self.make_norm(config)

    def forward(self, positions, hidden_states, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.post_self_attn_layernorm(self.self_attn(positions, hidden_states))
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.post_mlp_layernorm(self.mlp(hidden_states)), residual


class Glm4Model(LlamaModel):
    layer_cls = Glm4DecoderLayer


class Glm4ForCausalLM(GlmForCausalLM):
    model_cls = Glm4Model


# ---- Gemma-3 (text) ----------------------------------------------------------------------------------
class Gemma3DecoderLayer(Gemma2DecoderLayer):
    """Gemma-2 block plus per-head (1 + w) RMSNorm on q and k, and a rotary base of its own for the
    sliding-window layers."""

    @staticmethod
    def _is_sliding(config, layer_idx: int) -> bool:
        kinds = getattr(config, "layer_types", None)
        if kinds:
            return kinds[layer_idx] == "sliding_attention"
        pattern = getattr(config, "sliding_window_pattern", 6)
        return (layer_idx + 1) % pattern != 0

    def _rope_of(self, config, layer_idx: int):
        rp = getattr(config, "rope_parameters", None) or {}
        sliding = self._is_sliding(config, layer_idx)
        if "full_attention" in rp or "sliding_attention" in rp:
            d = dict(rp.get("sliding_attention" if sliding else "full_attention") or {})
            theta = float(d.get("rope_theta") or 10000.0)
            return theta, (d if d.get("rope_type", "default") != "default" else None)
        if sliding:
            return float(getattr(config, "rope_local_base_freq", 10000.0)), None
        return super()._rope_of(config, layer_idx)

    def _attention_extras(self, config, layer_idx: int) -> dict:
        return {"scale": float(config.query_pre_attn_scalar) ** -0.5,
                "logits_soft_cap": getattr(config, "attn_logit_softcapping", None),
                "sliding_window": getattr(config, "sliding_window", None) if self._is_sliding(config, layer_idx)
                else None,
                "qk_norm": "gemma"}


class Gemma3Model(GemmaModel):
    layer_cls = Gemma3DecoderLayer


class Gemma3ForCausalLM(LlamaForCausalLM):
    model_cls = Gemma3Model


# ---- InternLM2 ---------------------------------------------------------------------------------------
class InternLM2ForCausalLM(LlamaForCausalLM):
    """Llama weights under other names; ``wqkv`` stores, per KV head, its query heads followed by the
    key and the value head."""
    packed_modules_mapping = {"qkv_proj": ["wqkv"], "gate_up_proj": ["w1", "w3"]}

    def load_weights(self, weights):
        c = self.config
        H, KV = c.num_attention_heads, getattr(c, "num_key_value_heads", None) or c.num_attention_heads
        D = getattr(c, "head_dim", None) or c.hidden_size // H
        g = H // KV
        table = (("tok_embeddings", "embed_tokens"), (".attention.wo.", ".self_attn.o_proj."),
                 (".attention_norm.", ".input_layernorm."), (".ffn_norm.", ".post_attention_layernorm."),
                 (".feed_forward.w1.", ".mlp.gate_proj."), (".feed_forward.w3.", ".mlp.up_proj."),
                 (".feed_forward.w2.", ".mlp.down_proj."), (".feed_forward_ve.w1.", ".mlp_ve.gate_proj."),
                 (".feed_forward_ve.w3.", ".mlp_ve.up_proj."), (".feed_forward_ve.w2.", ".mlp_ve.down_proj."))

        def fix(items):
            for name, t in items:
                if name.startswith("output."):
                    name = "lm_head." + name[len("output."):]
                for a, b in table:
                    name = name.replace(a, b)
                if ".attention.wqkv." in name:
                    base = name.replace(".attention.wqkv.", ".self_attn.{}.")
                    t = t.view(KV, g + 2, D, *t.shape[1:])
                    yield base.format("q_proj"), t[:, :g].reshape(H * D, *t.shape[3:])
                    yield base.format("k_proj"), t[:, g].reshape(KV * D, *t.shape[3:])
                    yield base.format("v_proj"), t[:, g + 1].reshape(KV * D, *t.shape[3:])
                    continue
                yield name, t

        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping, tie_lm_head=self.tie)


class _VisualMask:
    """Which rows of the step are image positions (set by the multimodal wrapper when it merges the image
    embeddings, consumed by the next forward of the visual-expert decoder)."""
    rows: Optional[torch.Tensor] = None


class InternLM2VEDecoderLayer(LlamaDecoderLayer):
    """InternLM2 block with a second MLP for image positions ("visual expert"): rows flagged as image tokens run
    ``feed_forward_ve``, all others ``feed_forward``.
    Reference: vllm/model_executor/models/internlm2_ve.py (InternLM2VEDecoderLayer :26-103)."""
    supports_sequence_parallel = False

    def __init__(self, config, vllm_config, quant_config=None, prefix: str = "", attention_only: bool = False):
        super().__init__(config, vllm_config, quant_config, prefix, attention_only)
        if not attention_only:
            self.mlp_ve = self._make_mlp(config, vllm_config, quant_config, f"{prefix}.mlp_ve")

    def forward(self, positions, hidden_states, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        rows = _VisualMask.rows
        if rows is None:
            return self.mlp(hidden_states), residual
        # both experts over the step's rows, picked per row: no host synchronisation, and steps with image
        # positions are prefill steps anyway
        return torch.where(rows.unsqueeze(1), self.mlp_ve(hidden_states), self.mlp(hidden_states)), residual


class InternLM2VEModel(LlamaModel):
    layer_cls = InternLM2VEDecoderLayer

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        rows = _VisualMask.rows
        T = positions.shape[0] if inputs_embeds is None else inputs_embeds.shape[0]
        if rows is not None and rows.shape[0] != T:       # the step was padded behind the last real row
            rows = torch.nn.functional.pad(rows[:T], (0, max(T - rows.shape[0], 0)))
            _VisualMask.rows = rows
        try:
            return super().forward(input_ids, positions, intermediate_tensors, inputs_embeds)
        finally:
            _VisualMask.rows = None


class InternLM2VEForCausalLM(InternLM2ForCausalLM):
    """The decoder of InternVL checkpoints with visual experts.  ``set_visual_rows`` tells the next forward which
    rows are image positions (the wrapper calls it while it merges the image embeddings)."""
    model_cls = InternLM2VEModel

    @staticmethod
    def set_visual_rows(rows: Optional[torch.Tensor]) -> None:
        _VisualMask.rows = rows if rows is not None and bool(rows.numel()) else None


# ---- Baichuan ----------------------------------------------------------------------------------------
class BaichuanDecoderLayer(LlamaDecoderLayer):
    """Llama block; every size but the 7B one (hidden 4096) uses ALiBi instead of rotary embeddings."""
    supports_sequence_parallel = False

    def _attention_extras(self, config, layer_idx: int) -> dict:
        if config.hidden_size == 4096:
            return {}
        return {"use_rope": False, "alibi_slopes": alibi_slopes(config.num_attention_heads)}


class BaichuanModel(LlamaModel):
    layer_cls = BaichuanDecoderLayer


class BaichuanForCausalLM(LlamaForCausalLM):
    """``W_pack`` is q, k and v stacked; Baichuan-2 normalises the rows of the LM head."""
    model_cls = BaichuanModel
    packed_modules_mapping = {"qkv_proj": ["W_pack"], "gate_up_proj": ["gate_proj", "up_proj"]}

    def load_weights(self, weights):
        c = self.config
        d = c.hidden_size
        norm_head = getattr(c, "vocab_size", 0) == 125696       # Baichuan-2 releases

        def fix(items):
            for name, t in items:
                if ".W_pack." in name:
                    base = name.replace(".W_pack.", ".{}.")
                    for i, proj in enumerate(("q_proj", "k_proj", "v_proj")):
                        yield base.format(proj), t[i * d:(i + 1) * d]
                    continue
                if name == "lm_head.weight" and norm_head:
                    t = torch.nn.functional.normalize(t.float()).to(t.dtype)
                yield name, t

        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping, tie_lm_head=self.tie)


BaiChuanForCausalLM = BaichuanForCausalLM


class Gemma3TextOnlyFromMultimodal(Gemma3ForCausalLM):
    """Gemma-3 multimodal checkpoints (architecture ``Gemma3ForConditionalGeneration``) served as text
    models: the decoder's weights are read from the checkpoint, the vision tower and projector are not
    loaded, and image inputs are not accepted."""

    def load_weights(self, weights):
        def fix(items):
            for name, t in items:
                if name.startswith("model.language_model."):
                    yield "model." + name[len("model.language_model."):], t
                elif name.startswith("language_model."):
                    yield name[len("language_model."):], t
                elif name.startswith("lm_head."):
                    yield name, t
                # vision_tower.*, multi_modal_projector.*: not part of the text path
        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping, tie_lm_head=self.tie)


# ---- Llama checkpoints written by fairseq2 --------------------------------------------------------------------
class Fairseq2LlamaForCausalLM(LlamaForCausalLM):
    """Llama weights in fairseq2's naming (``decoder_frontend.embed``, ``decoder.layers.N.ffn.inner_proj``,
    ``final_proj`` ...) inside its ``{"model_key": name, name: state_dict}`` wrapper (unwrapped by the .pt reader),
    with q / k projection rows in the interleaved rotary order: they are permuted to the half-split order the
    rotary kernel uses.  Reference: vllm/model_executor/models/fairseq2_llama.py:48-152."""
    _PREFIX = (("decoder_frontend.embed.", "model.embed_tokens."), ("decoder.", "model."), ("final_proj.", "lm_head."))
    _SUBSTR = ((".self_attn_layer_norm.", ".input_layernorm."), (".ffn_layer_norm.", ".post_attention_layernorm."),
               (".self_attn.output_proj.", ".self_attn.o_proj."), (".ffn.gate_proj.", ".mlp.gate_proj."),
               (".ffn.inner_proj.", ".mlp.up_proj."), (".ffn.output_proj.", ".mlp.down_proj."),
               (".layer_norm.", ".norm."))

    def load_weights(self, weights):
        c = self.config
        hd = getattr(c, "head_dim", None) or c.hidden_size // c.num_attention_heads

        def convert():
            for n, t in weights:
                for a, b in self._PREFIX:
                    if n.startswith(a):
                        n = b + n[len(a):]
                        break
                for a, b in self._SUBSTR:
                    n = n.replace(a, b)
                if ".q_proj." in n or ".k_proj." in n:
                    heads = t.shape[0] // hd
                    t = t.view(heads, hd // 2, 2, *t.shape[1:]).transpose(1, 2).reshape(t.shape)
                yield n, t
        return super().load_weights(convert())


# ---- DeciLM / Nemotron-NAS -------------------------------------------------------------------------
def _find_multiple(n: int, k: int) -> int:
    return n if n % k == 0 else n + k - n % k


def deci_intermediate_size(ffn_mult: float, hidden: int) -> int:
    """Width of a block's gated MLP from its ``ffn_mult``: 2/3 of mult x hidden, rounded up to 256."""
    return _find_multiple(int(2 * ffn_mult * hidden / 3), 256)


def _field(obj, name, default=None):
    return obj.get(name, default) if isinstance(obj, dict) else getattr(obj, name, default)


class DeciLMDecoderLayer(LlamaDecoderLayer):
    """Llama block whose shape comes from a per-layer entry of ``block_configs`` (a neural-architecture search
    picked them): the attention keeps ``num_attention_heads / n_heads_in_group`` KV heads or is skipped altogether
    (``no_op``), the MLP is ``ffn_mult`` wide or skipped.  Skipped halves have no weights and no norm.
    Reference: vllm/model_executor/models/nemotron_nas.py (DeciLMDecoderLayer :110-208)."""
    supports_sequence_parallel = False

    def __init__(self, config, vllm_config, quant_config=None, prefix: str = ""):
        import copy
        from ..config import HFConfig
        bc = config.block_configs[int(prefix.split(".")[-1])]
        attn, ffn = _field(bc, "attention", {}), _field(bc, "ffn", {})
        no_attn, no_ffn = bool(_field(attn, "no_op", False)), bool(_field(ffn, "no_op", False))
        if _field(attn, "replace_with_linear", False) or _field(ffn, "replace_with_linear", False):
            raise NotImplementedError("DeciLM blocks replaced by a linear layer")
        own = HFConfig(config.to_dict()) if isinstance(config, HFConfig) else copy.copy(config)
        group = _field(attn, "n_heads_in_group", None)
        if not no_attn and group:
            own.num_key_value_heads = config.num_attention_heads // int(group)
        mult = _field(ffn, "ffn_mult", None)
        if not no_ffn and mult:
            own.intermediate_size = deci_intermediate_size(float(mult), config.hidden_size)
        super().__init__(own, vllm_config, quant_config, prefix)
        self.no_attn, self.no_ffn = no_attn, no_ffn
        if no_attn:
            del self.self_attn, self.input_layernorm
        if no_ffn:
            del self.mlp, self.post_attention_layernorm

    def forward(self, positions, hidden_states, residual):
        if not self.no_attn:
            if residual is None:
                residual = hidden_states
                hidden_states = self.input_layernorm(hidden_states)
            else:
                hidden_states, residual = self.input_layernorm(hidden_states, residual)
            hidden_states = self.self_attn(positions, hidden_states)
        if not self.no_ffn:
            if residual is None:                        # a first block without attention
                residual = hidden_states
                hidden_states = self.post_attention_layernorm(hidden_states)
            else:
                hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
            hidden_states = self.mlp(hidden_states)
        return hidden_states, residual


class DeciLMModel(LlamaModel):
    layer_cls = DeciLMDecoderLayer


class DeciLMForCausalLM(LlamaForCausalLM):
    model_cls = DeciLMModel
