"""Hybrid stacks.  Jamba: a hybrid stack of Mamba-1 mixers and (rotary-free) attention layers, each followed by a dense
or a routed-expert MLP.  Attention layers use the paged KV cache; the Mamba layers keep one recurrent
state per request in slots the model runner allocates beside the weights (worker/model_runner.py,
``_init_hybrid_state``), so both kinds of per-request memory live under one scheduler.
Reference: vllm/model_executor/models/jamba.py.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Iterable

import torch
import torch.nn as nn

from ..config import VllmConfig
from ..layers.basic import LogitsProcessor, ParallelLMHead, RMSNorm, VocabParallelEmbedding
from .llama import LlamaAttention, LlamaMLP
from .mamba import MambaMixer
from .mixtral import SparseMoeBlock
from .utils import load_weights_with_mapping, maybe_prefix


def _layer_kinds(config) -> list[str]:
    kinds = getattr(config, "layers_block_type", None)
    if kinds:
        return list(kinds)
    p, o = config.attn_layer_period, config.attn_layer_offset
    return ["attention" if i % p == o else "mamba" for i in range(config.num_hidden_layers)]


def _layer_experts(config) -> list[int]:
    n = getattr(config, "layers_num_experts", None)
    if n:
        return list(n)
    p, o = config.expert_layer_period, config.expert_layer_offset
    return [config.num_experts if i % p == o else 1 for i in range(config.num_hidden_layers)]


class JambaLayer(nn.Module):
    def __init__(self, config, vllm_config: VllmConfig, kind: str, num_experts: int, prefix: str):
        super().__init__()
        quant_config = getattr(vllm_config, "quant_config", None)
        d = config.hidden_size
        eps = config.rms_norm_eps
        self.kind = kind
        if kind == "attention":
            self.self_attn = LlamaAttention(
                config, d, config.num_attention_heads, config.num_key_value_heads, 10000.0, None,
                vllm_config.model_config.max_model_len, quant_config=quant_config, prefix=f"{prefix}.self_attn",
                cache_dtype=vllm_config.cache_config.cache_dtype, use_rope=False)
        else:
            inner = int(config.mamba_expand * d)
            shim = SimpleNamespace(hidden_size=d, intermediate_size=inner, state_size=config.mamba_d_state,
                                   conv_kernel=config.mamba_d_conv, time_step_rank=config.mamba_dt_rank,
                                   use_bias=bool(config.mamba_proj_bias), use_conv_bias=bool(config.mamba_conv_bias))
            self.mamba = MambaMixer(shim, f"{prefix}.mamba", rms_eps=eps, learned_norms=True)
        if num_experts > 1:
            # softmax over all experts, top-k, no renormalisation
            self.feed_forward = SparseMoeBlock(d, config.intermediate_size, num_experts, config.num_experts_per_tok,
                                               False, quant_config, f"{prefix}.feed_forward")
        else:
            self.feed_forward = LlamaMLP(d, config.intermediate_size, False, quant_config, f"{prefix}.feed_forward")
        self.input_layernorm = RMSNorm(d, eps)
        self.pre_ff_layernorm = RMSNorm(d, eps)

    def forward(self, positions, hidden, residual):
        if residual is None:
            residual = hidden
            hidden = self.input_layernorm(hidden)
        else:
            hidden, residual = self.input_layernorm(hidden, residual)
        hidden = self.self_attn(positions, hidden) if self.kind == "attention" else self.mamba(hidden)
        hidden, residual = self.pre_ff_layernorm(hidden, residual)
        return self.feed_forward(hidden), residual


class JambaForCausalLM(nn.Module):
    supports_pp = False
    supports_lora = False
    packed_modules_mapping = {"qkv_proj": ["q_proj", "k_proj", "v_proj"], "gate_up_proj": ["gate_proj", "up_proj"]}
    stacked_params_mapping = [
        (".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"), (".qkv_proj", ".v_proj", "v"),
        (".gate_up_proj", ".gate_proj", 0), (".gate_up_proj", ".up_proj", 1),
    ]

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        kinds, experts = _layer_kinds(config), _layer_experts(config)
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([JambaLayer(config, vllm_config, kinds[i], experts[i], f"model.layers.{i}")
                                     for i in range(config.num_hidden_layers)])
        self.final_layernorm = RMSNorm(config.hidden_size, config.rms_norm_eps)
        self.tie = bool(getattr(config, "tie_word_embeddings", False))
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if self.tie:
            self.lm_head.tie_weights(self.embed_tokens)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    def get_input_embeddings(self, ids):
        return self.embed_tokens(ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        hidden = inputs_embeds if inputs_embeds is not None else self.embed_tokens(input_ids)
        residual = None
        for layer in self.layers:
            hidden, residual = layer(positions, hidden, residual)
        hidden, _ = self.final_layernorm(hidden, residual)
        return hidden

    def compute_logits(self, hidden):
        return self.logits_processor(self.lm_head, hidden)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        from ..layers.moe import FusedMoE
        n_exp = max(_layer_experts(self.config))
        experts = FusedMoE.make_expert_params_mapping("gate_proj", "down_proj", "up_proj", n_exp) if n_exp > 1 else []
        mamba_names = {".mamba.conv1d.weight": ".mamba.conv_weight", ".mamba.conv1d.bias": ".mamba.conv_bias",
                       ".mamba.A_log": ".mamba.A", ".mamba.dt_proj.bias": ".mamba.dt_bias"}

        def fix(items):
            for name, t in items:
                name = name.removeprefix("model.")
                for a, b in mamba_names.items():
                    if name.endswith(a):
                        name = name[: -len(a)] + b
                if name.endswith(".feed_forward.router.weight"):
                    name = name.replace(".router.", ".gate.")
                if name.endswith(".feed_forward.experts.gate_up_proj"):
                    # recent transformers store all experts stacked: [E, 2I, H] (gate rows, then up rows)
                    base, half = name[: -len("gate_up_proj")], t.shape[1] // 2
                    for e in range(t.shape[0]):
                        yield f"{base}{e}.gate_proj.weight", t[e, :half]
                        yield f"{base}{e}.up_proj.weight", t[e, half:]
                    continue
                if name.endswith(".feed_forward.experts.down_proj"):
                    base = name[: -len("down_proj")]
                    for e in range(t.shape[0]):
                        yield f"{base}{e}.down_proj.weight", t[e]
                    continue
                yield name, t

        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping, tie_lm_head=self.tie,
                                         expert_mapping=experts)


# ---- Bamba -------------------------------------------------------------------------------------------
class BambaLayer(nn.Module):
    """Mamba-2 mixer or rotary (half of each head) attention, then a gated MLP.
    Reference: vllm/model_executor/models/bamba.py."""

    def __init__(self, config, vllm_config: VllmConfig, kind: str, prefix: str):
        super().__init__()
        from .mamba import Mamba2Mixer
        quant_config = getattr(vllm_config, "quant_config", None)
        d, eps = config.hidden_size, config.rms_norm_eps
        self.kind = kind
        if kind == "attention":
            rp = getattr(config, "rope_parameters", None) or {}
            theta = float(getattr(config, "rope_theta", None) or rp.get("rope_theta") or 10000.0)
            self.self_attn = LlamaAttention(
                config, d, config.num_attention_heads, config.num_key_value_heads, theta, None,
                vllm_config.model_config.max_model_len, bias=bool(getattr(config, "attention_bias", False)),
                o_bias=bool(getattr(config, "attention_bias", False)), quant_config=quant_config,
                prefix=f"{prefix}.self_attn", cache_dtype=vllm_config.cache_config.cache_dtype)
        else:
            shim = SimpleNamespace(hidden_size=d, num_heads=config.mamba_n_heads, head_dim=config.mamba_d_head,
                                   intermediate_size=config.mamba_n_heads * config.mamba_d_head,
                                   state_size=config.mamba_d_state, conv_kernel=config.mamba_d_conv,
                                   n_groups=config.mamba_n_groups, use_bias=bool(config.mamba_proj_bias),
                                   use_conv_bias=bool(config.mamba_conv_bias), layer_norm_epsilon=eps,
                                   gated_norm_groups=getattr(config, "gated_norm_groups", None))
            self.mamba = Mamba2Mixer(shim, f"{prefix}.mamba")
        self.feed_forward = LlamaMLP(d, config.intermediate_size, bool(getattr(config, "mlp_bias", False)),
                                     quant_config, f"{prefix}.feed_forward")
        self.input_layernorm = RMSNorm(d, eps)
        self.pre_ff_layernorm = RMSNorm(d, eps)

    forward = JambaLayer.forward


class BambaForCausalLM(JambaForCausalLM):
    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        nn.Module.__init__(self)
        config = vllm_config.model_config.hf_config
        self.config = config
        attn = set(getattr(config, "attn_layer_indices", None) or [])
        kinds = getattr(config, "layers_block_type", None) or \
            ["attention" if i in attn else "mamba" for i in range(config.num_hidden_layers)]
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([BambaLayer(config, vllm_config, kinds[i], f"model.layers.{i}")
                                     for i in range(config.num_hidden_layers)])
        self.final_layernorm = RMSNorm(config.hidden_size, config.rms_norm_eps)
        self.tie = bool(getattr(config, "tie_word_embeddings", False))
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if self.tie:
            self.lm_head.tie_weights(self.embed_tokens)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        names = {".mamba.conv1d.weight": ".mamba.conv_weight", ".mamba.conv1d.bias": ".mamba.conv_bias",
                 ".mamba.A_log": ".mamba.A", ".mamba.norm.weight": ".mamba.norm_weight"}

        def fix(items):
            for name, t in items:
                name = name.removeprefix("model.")
                for a, b in names.items():
                    if name.endswith(a):
                        name = name[: -len(a)] + b
                yield name, t

        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping, tie_lm_head=self.tie)


# ---- Falcon-H1 ------------------------------------------------------------------------------------------------
class FalconH1Layer(nn.Module):
    """Attention and a Mamba-2 mixer side by side on the same normalised input, their outputs added to the residual
    together; then a gated MLP.  The family's fixed multipliers (on the mixer input segments, keys, attention input
    and output, MLP gate and output) are constants: they are folded into the projection weights while loading.
    Reference: vllm/model_executor/models/falcon_h1.py (FalconH1ParallelHybrid :360-440, the multipliers :60-110)."""

    def __init__(self, config, vllm_config: VllmConfig, prefix: str):
        super().__init__()
        from .mamba import Mamba2Mixer
        quant_config = getattr(vllm_config, "quant_config", None)
        d, eps = config.hidden_size, config.rms_norm_eps
        rp = getattr(config, "rope_parameters", None) or {}
        theta = float(getattr(config, "rope_theta", None) or rp.get("rope_theta") or 10000.0)
        ab = bool(getattr(config, "attention_bias", False))
        self.self_attn = LlamaAttention(config, d, config.num_attention_heads, config.num_key_value_heads, theta, None,
                                        vllm_config.model_config.max_model_len, bias=ab, o_bias=ab,
                                        quant_config=quant_config, prefix=f"{prefix}.self_attn",
                                        cache_dtype=vllm_config.cache_config.cache_dtype)
        if bool(getattr(config, "mamba_rms_norm", False)) and bool(getattr(config, "mamba_norm_before_gate", False)):
            raise ValueError("Falcon-H1 with a gated norm applied before the gate is not supported")
        d_ssm = int(getattr(config, "mamba_d_ssm", None) or config.mamba_expand * d)
        shim = SimpleNamespace(hidden_size=d, num_heads=config.mamba_n_heads, head_dim=d_ssm // config.mamba_n_heads,
                               intermediate_size=d_ssm, state_size=config.mamba_d_state,
                               conv_kernel=config.mamba_d_conv, n_groups=config.mamba_n_groups,
                               use_bias=bool(config.mamba_proj_bias), use_conv_bias=bool(config.mamba_conv_bias),
                               layer_norm_epsilon=eps, gated_norm=bool(getattr(config, "mamba_rms_norm", False)),
                               gated_norm_groups=getattr(config, "gated_norm_groups", None))
        self.mamba = Mamba2Mixer(shim, f"{prefix}.mamba")
        self.feed_forward = LlamaMLP(d, config.intermediate_size, bool(getattr(config, "mlp_bias", False)),
                                     quant_config, f"{prefix}.feed_forward")
        self.input_layernorm = RMSNorm(d, eps)
        self.pre_ff_layernorm = RMSNorm(d, eps)

    def forward(self, positions, hidden, residual):
        if residual is None:
            residual = hidden
            hidden = self.input_layernorm(hidden)
        else:
            hidden, residual = self.input_layernorm(hidden, residual)
        hidden = self.self_attn(positions, hidden) + self.mamba(hidden)
        hidden, residual = self.pre_ff_layernorm(hidden, residual)
        return self.feed_forward(hidden), residual


class FalconH1ForCausalLM(JambaForCausalLM):
    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        nn.Module.__init__(self)
        config = vllm_config.model_config.hf_config
        self.config = config
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.embed_scale = float(getattr(config, "embedding_multiplier", 1.0) or 1.0)
        self.layers = nn.ModuleList([FalconH1Layer(config, vllm_config, f"model.layers.{i}")
                                     for i in range(config.num_hidden_layers)])
        self.final_layernorm = RMSNorm(config.hidden_size, config.rms_norm_eps)
        self.tie = bool(getattr(config, "tie_word_embeddings", False))
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if self.tie:
            self.lm_head.tie_weights(self.embed_tokens)
        self.logits_processor = LogitsProcessor(config.vocab_size, config.vocab_size,
                                                float(getattr(config, "lm_head_multiplier", 1.0) or 1.0))

    forward = None          # set below: GraniteMoeHybridForCausalLM.forward applies the embedding scale

    def _mup_rows(self) -> torch.Tensor:
        """Per-row factor of the mixer's input projection: [gate | x | B | C | dt] segments."""
        c = self.config
        m = list(getattr(c, "ssm_multipliers", None) or [1.0] * 5)
        d_ssm = int(getattr(c, "mamba_d_ssm", None) or c.mamba_expand * c.hidden_size)
        gs = int(c.mamba_n_groups) * int(c.mamba_d_state)
        v = torch.ones(2 * d_ssm + 2 * gs + int(c.mamba_n_heads))
        v[:d_ssm] *= m[0]
        v[d_ssm:2 * d_ssm] *= m[1]
        v[2 * d_ssm:2 * d_ssm + gs] *= m[2]
        v[2 * d_ssm + gs:2 * d_ssm + 2 * gs] *= m[3]
        v[2 * d_ssm + 2 * gs:] *= m[4]
        return v

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        c = self.config
        f = lambda n, d=1.0: float(getattr(c, n, d) if getattr(c, n, None) is not None else d)
        gate_m, down_m = (list(getattr(c, "mlp_multipliers", None) or [1.0, 1.0]) + [1.0, 1.0])[:2]
        a_in, a_out, key_m = f("attention_in_multiplier"), f("attention_out_multiplier"), f("key_multiplier")
        s_in, s_out = f("ssm_in_multiplier"), f("ssm_out_multiplier")
        mup = self._mup_rows()
        names = {".mamba.conv1d.weight": ".mamba.conv_weight", ".mamba.conv1d.bias": ".mamba.conv_bias",
                 ".mamba.A_log": ".mamba.A", ".mamba.norm.weight": ".mamba.norm_weight"}

        def fix(items):
            for name, t in items:
                name = name.removeprefix("model.")
                w = name.endswith(".weight")
                if ".mamba.in_proj." in name:
                    r = mup.to(t.dtype)
                    t = t * (r[:, None] * s_in if w else r)
                elif ".mamba.out_proj." in name and w:
                    t = t * s_out
                elif ".self_attn.k_proj." in name:
                    t = t * (a_in * key_m if w else key_m)
                elif (".self_attn.q_proj." in name or ".self_attn.v_proj." in name) and w:
                    t = t * a_in
                elif ".self_attn.o_proj." in name and w:
                    t = t * a_out
                elif ".feed_forward.gate_proj." in name:
                    t = t * gate_m
                elif ".feed_forward.down_proj." in name and w:
                    t = t * down_m
                for a, b in names.items():
                    if name.endswith(a):
                        name = name[: -len(a)] + b
                yield name, t

        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping, tie_lm_head=self.tie)


# ---- Granite-4 hybrid (GraniteMoeHybrid) -----------------------------------------------------------------
class GraniteMoeHybridLayer(nn.Module):
    """Mamba-2 mixer or attention (softmax scale and residual scale from the Granite multipliers), then
    routed experts plus a shared gated MLP on the same normalised input.
    Reference: vllm/model_executor/models/granitemoehybrid.py."""

    def __init__(self, config, vllm_config: VllmConfig, kind: str, prefix: str):
        super().__init__()
        from .mamba import Mamba2Mixer
        quant_config = getattr(vllm_config, "quant_config", None)
        d, eps = config.hidden_size, config.rms_norm_eps
        self.kind = kind
        self.residual_multiplier = float(getattr(config, "residual_multiplier", 1.0) or 1.0)
        if kind == "attention":
            rp = getattr(config, "rope_parameters", None) or {}
            theta = float(getattr(config, "rope_theta", None) or rp.get("rope_theta") or 10000.0)
            bias = bool(getattr(config, "attention_bias", False))
            self.self_attn = LlamaAttention(
                config, d, config.num_attention_heads, config.num_key_value_heads, theta, None,
                vllm_config.model_config.max_model_len, bias=bias, o_bias=bias, quant_config=quant_config,
                prefix=f"{prefix}.self_attn", cache_dtype=vllm_config.cache_config.cache_dtype,
                scale=float(getattr(config, "attention_multiplier", None) or (d // config.num_attention_heads) ** -0.5),
                use_rope=getattr(config, "position_embedding_type", "rope") == "rope")
        else:
            shim = SimpleNamespace(hidden_size=d, num_heads=config.mamba_n_heads, head_dim=config.mamba_d_head,
                                   intermediate_size=config.mamba_n_heads * config.mamba_d_head,
                                   state_size=config.mamba_d_state, conv_kernel=config.mamba_d_conv,
                                   n_groups=config.mamba_n_groups, use_bias=bool(config.mamba_proj_bias),
                                   use_conv_bias=bool(config.mamba_conv_bias), layer_norm_epsilon=eps,
                                   gated_norm_groups=getattr(config, "gated_norm_groups", None))
            self.mamba = Mamba2Mixer(shim, f"{prefix}.mamba")
        n_exp = int(getattr(config, "num_local_experts", 0) or 0)
        self.block_sparse_moe = SparseMoeBlock(d, config.intermediate_size, n_exp, config.num_experts_per_tok, True,
                                               quant_config, f"{prefix}.block_sparse_moe") if n_exp > 0 else None
        shared = int(getattr(config, "shared_intermediate_size", 0) or 0)
        self.shared_mlp = LlamaMLP(d, shared, False, quant_config, f"{prefix}.shared_mlp") if shared > 0 else None
        self.input_layernorm = RMSNorm(d, eps)
        self.post_attention_layernorm = RMSNorm(d, eps)

    def forward(self, positions, hidden, residual):
        if residual is None:
            residual = hidden
            hidden = self.input_layernorm(hidden)
        else:
            hidden, residual = self.input_layernorm(hidden, residual)
        hidden = self.self_attn(positions, hidden) if self.kind == "attention" else self.mamba(hidden)
        hidden, residual = self.post_attention_layernorm(hidden * self.residual_multiplier, residual)
        out = self.block_sparse_moe(hidden) if self.block_sparse_moe is not None else 0
        if self.shared_mlp is not None:
            out = out + self.shared_mlp(hidden)
        return out * self.residual_multiplier, residual


class GraniteMoeHybridForCausalLM(JambaForCausalLM):
    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        nn.Module.__init__(self)
        config = vllm_config.model_config.hf_config
        self.config = config
        kinds = ["attention" if k in ("attention", "full_attention") else "mamba"
                 for k in (getattr(config, "layer_types", None) or getattr(config, "layers_block_type"))]
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.embed_scale = float(getattr(config, "embedding_multiplier", 1.0) or 1.0)
        self.layers = nn.ModuleList([GraniteMoeHybridLayer(config, vllm_config, kinds[i], f"model.layers.{i}")
                                     for i in range(config.num_hidden_layers)])
        self.final_layernorm = RMSNorm(config.hidden_size, config.rms_norm_eps)
        self.tie = bool(getattr(config, "tie_word_embeddings", False))
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if self.tie:
            self.lm_head.tie_weights(self.embed_tokens)
        self.logits_processor = LogitsProcessor(config.vocab_size, config.vocab_size,
                                                1.0 / float(getattr(config, "logits_scaling", 1.0) or 1.0))

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        hidden = inputs_embeds if inputs_embeds is not None else self.embed_tokens(input_ids)
        hidden = hidden * self.embed_scale
        residual = None
        for layer in self.layers:
            hidden, residual = layer(positions, hidden, residual)
        hidden, _ = self.final_layernorm(hidden, residual)
        return hidden

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        from ..layers.moe import FusedMoE
        n_exp = int(getattr(self.config, "num_local_experts", 0) or 0)
        experts = FusedMoE.make_expert_params_mapping("gate_proj", "down_proj", "up_proj", n_exp) if n_exp else []
        tails = {".mamba.conv1d.weight": ".mamba.conv_weight", ".mamba.conv1d.bias": ".mamba.conv_bias",
                 ".mamba.A_log": ".mamba.A", ".mamba.norm.weight": ".mamba.norm_weight",
                 ".shared_mlp.input_linear.weight": ".shared_mlp.gate_up_proj.weight",
                 ".shared_mlp.output_linear.weight": ".shared_mlp.down_proj.weight",
                 ".block_sparse_moe.router.weight": ".block_sparse_moe.gate.weight",
                 ".block_sparse_moe.router.layer.weight": ".block_sparse_moe.gate.weight"}

        def fix(items):
            for name, t in items:
                name = name.removeprefix("model.")
                if name == "norm.weight":
                    name = "final_layernorm.weight"
                for a, b in tails.items():
                    if name.endswith(a):
                        name = name[: -len(a)] + b
                # all experts stacked: [E, 2I, H] (gate rows, then up rows) and [E, H, I]
                if name.endswith((".block_sparse_moe.experts.gate_up_proj", ".block_sparse_moe.input_linear.weight")):
                    base = name[: name.index(".block_sparse_moe.")] + ".block_sparse_moe.experts."
                    half = t.shape[1] // 2
                    for e in range(t.shape[0]):
                        yield f"{base}{e}.gate_proj.weight", t[e, :half]
                        yield f"{base}{e}.up_proj.weight", t[e, half:]
                    continue
                if name.endswith((".block_sparse_moe.experts.down_proj", ".block_sparse_moe.output_linear.weight")):
                    base = name[: name.index(".block_sparse_moe.")] + ".block_sparse_moe.experts."
                    for e in range(t.shape[0]):
                        yield f"{base}{e}.down_proj.weight", t[e]
                    continue
                yield name, t

        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping, tie_lm_head=self.tie,
                                         expert_mapping=experts)


# ---- Nemotron-H ----------------------------------------------------------------------------------------
class NemotronHBlock(nn.Module):
    """One pre-norm residual block around a single mixer: a Mamba-2 layer, an attention layer without
    positional encoding, or a squared-ReLU MLP.  Reference: vllm/model_executor/models/nemotron_h.py."""

    def __init__(self, config, vllm_config: VllmConfig, kind: str, prefix: str):
        super().__init__()
        from .llama_variants import NemotronMLP
        from .mamba import Mamba2Mixer
        quant_config = getattr(vllm_config, "quant_config", None)
        d, eps = config.hidden_size, float(getattr(config, "layer_norm_epsilon", 1e-5))
        self.kind = kind
        self.norm = RMSNorm(d, eps)
        if kind == "attention":
            bias = bool(getattr(config, "attention_bias", False))
            self.mixer = LlamaAttention(
                config, d, config.num_attention_heads, config.num_key_value_heads, 10000.0, None,
                vllm_config.model_config.max_model_len, bias=bias, o_bias=bias, quant_config=quant_config,
                prefix=f"{prefix}.mixer", cache_dtype=vllm_config.cache_config.cache_dtype, use_rope=False)
        elif kind == "mamba":
            shim = SimpleNamespace(hidden_size=d, num_heads=config.mamba_num_heads, head_dim=config.mamba_head_dim,
                                   intermediate_size=config.mamba_num_heads * config.mamba_head_dim,
                                   state_size=config.ssm_state_size, conv_kernel=config.conv_kernel,
                                   n_groups=config.n_groups, use_bias=bool(getattr(config, "use_bias", False)),
                                   use_conv_bias=bool(getattr(config, "use_conv_bias", True)), layer_norm_epsilon=eps,
                                   gated_norm_groups=getattr(config, "gated_norm_groups", None),
                                   time_step_limit=getattr(config, "time_step_limit", None))
            self.mixer = Mamba2Mixer(shim, f"{prefix}.mixer")
        elif kind == "mlp":
            self.mixer = NemotronMLP(d, config.intermediate_size, bool(getattr(config, "mlp_bias", False)),
                                     getattr(config, "mlp_hidden_act", "relu2"), quant_config, f"{prefix}.mixer")
        else:
            raise ValueError(f"Nemotron-H block type {kind!r} is not supported (mamba, attention, mlp)")

    def forward(self, positions, hidden, residual):
        if residual is None:
            residual = hidden
            hidden = self.norm(hidden)
        else:
            hidden, residual = self.norm(hidden, residual)
        hidden = self.mixer(positions, hidden) if self.kind == "attention" else self.mixer(hidden)
        return hidden, residual


class NemotronHForCausalLM(JambaForCausalLM):
    stacked_params_mapping = [(".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"), (".qkv_proj", ".v_proj", "v")]
    packed_modules_mapping = {"qkv_proj": ["q_proj", "k_proj", "v_proj"]}

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        nn.Module.__init__(self)
        config = vllm_config.model_config.hf_config
        self.config = config
        names = {"linear_attention": "mamba", "full_attention": "attention", "M": "mamba", "*": "attention", "-": "mlp"}
        kinds = getattr(config, "layers_block_type", None) or list(getattr(config, "hybrid_override_pattern", ""))
        kinds = [names.get(k, k) for k in kinds]
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([NemotronHBlock(config, vllm_config, k, f"model.layers.{i}")
                                     for i, k in enumerate(kinds)])
        self.final_layernorm = RMSNorm(config.hidden_size, float(getattr(config, "layer_norm_epsilon", 1e-5)))
        self.tie = bool(getattr(config, "tie_word_embeddings", False))
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if self.tie:
            self.lm_head.tie_weights(self.embed_tokens)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        tails = {".mixer.conv1d.weight": ".mixer.conv_weight", ".mixer.conv1d.bias": ".mixer.conv_bias",
                 ".mixer.A_log": ".mixer.A", ".mixer.norm.weight": ".mixer.norm_weight"}

        def fix(items):
            for name, t in items:
                name = name.removeprefix("backbone.").removeprefix("model.")
                for old in ("embeddings.", "embedding."):
                    if name.startswith(old):
                        name = "embed_tokens." + name[len(old):]
                if name == "norm_f.weight":
                    name = "final_layernorm.weight"
                for a, b in tails.items():
                    if name.endswith(a):
                        name = name[: -len(a)] + b
                yield name, t

        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping, tie_lm_head=self.tie)


FalconH1ForCausalLM.forward = GraniteMoeHybridForCausalLM.forward
