"""Sparse mixture-of-experts decoders: Mixtral (8x7B / 8x22B) and Qwen2-MoE / Qwen3-MoE.

Attention is the Llama block; the MLP is a replicated router GEMM followed by the fused
routed-expert layer (top-k softmax -> counting sort -> grouped gate/up + down GEMMs -> sum),
optionally plus a shared dense expert (Qwen2-MoE).
Reference: vllm/model_executor/models/mixtral.py (MixtralMoE :60-122, load_weights :362-470),
qwen2_moe.py (Qwen2MoeSparseMoeBlock :102-177), qwen3_moe.py.
"""
from __future__ import annotations

from typing import Iterable

import torch
import torch.nn as nn

from ..layers.linear import ReplicatedLinear
from ..layers.moe import FusedMoE
from ..parallel.state import get_tensor_model_parallel_world_size, tensor_model_parallel_all_reduce
from .llama import LlamaDecoderLayer, LlamaForCausalLM, LlamaMLP, LlamaModel
from .utils import load_weights_with_mapping


def _num_experts(config) -> int:
    # "num_experts" in Qwen checkpoints, "num_local_experts" in Mixtral ones and in configs
    # re-saved by recent transformers
    for k in ("num_experts", "num_local_experts", "n_routed_experts"):
        v = getattr(config, k, None)
        if v:
            return int(v)
    return 0


class SparseMoeBlock(nn.Module):
    def __init__(self, hidden_size: int, intermediate_size: int, num_experts: int, top_k: int, renormalize: bool,
                 quant_config=None, prefix: str = "", shared_intermediate: int = 0, shared_gate: bool = False,
                 routing=None, activation: str = "silu"):
        super().__init__()
        self.gate = ReplicatedLinear(hidden_size, num_experts, bias=False, quant_config=None, prefix=f"{prefix}.gate")
        self.tp = get_tensor_model_parallel_world_size()
        self.has_shared = shared_intermediate > 0
        # with a shared expert both partial results are summed once, after adding them
        self.experts = FusedMoE(num_experts, top_k, hidden_size, intermediate_size, reduce_results=not self.has_shared,
                                renormalize=renormalize, quant_config=quant_config, prefix=f"{prefix}.experts",
                                custom_routing_function=routing, activation=activation)
        if self.has_shared:
            self.shared_expert = LlamaMLP(hidden_size, shared_intermediate, quant_config=quant_config,
                                          prefix=f"{prefix}.shared_expert", reduce_results=False)
            self.shared_expert_gate = ReplicatedLinear(hidden_size, 1, bias=False, prefix=f"{prefix}.shared_expert_gate") \
                if shared_gate else None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        out = self.experts(x, self.gate(x))
        if self.has_shared:
            s = self.shared_expert(x)
            if self.shared_expert_gate is not None:
                s = torch.sigmoid(self.shared_expert_gate(x)) * s
            out = out + s
            if self.tp > 1 or self.experts.ep_size > 1:
                out = tensor_model_parallel_all_reduce(out)
        return out


class MixtralDecoderLayer(LlamaDecoderLayer):
    supports_sequence_parallel = False
    def _make_mlp(self, config, vllm_config, quant_config, prefix):
        # checkpoint names: model.layers.N.block_sparse_moe.{gate, experts.E.{w1,w2,w3}}
        return SparseMoeBlock(config.hidden_size, config.intermediate_size, _num_experts(config),
                              config.num_experts_per_tok, True, quant_config,
                              prefix.replace(".mlp", ".block_sparse_moe"))

    def __init__(self, config, vllm_config, quant_config=None, prefix: str = ""):
        super().__init__(config, vllm_config, quant_config, prefix)
        self.block_sparse_moe = self.mlp           # parameter names follow the checkpoint
        del self.mlp

    def forward(self, positions, hidden_states, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.block_sparse_moe(hidden_states), residual


class MixtralModel(LlamaModel):
    layer_cls = MixtralDecoderLayer


class MixtralForCausalLM(LlamaForCausalLM):
    model_cls = MixtralModel
    stacked_params_mapping = [(".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"), (".qkv_proj", ".v_proj", "v")]
    packed_modules_mapping = {"qkv_proj": ["q_proj", "k_proj", "v_proj"]}
    supports_lora = False
    expert_names = ("w1", "w2", "w3")              # gate, down, up

    def num_experts(self) -> int:
        return _num_experts(self.config)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]) -> set[str]:
        g, d, u = self.expert_names
        experts = FusedMoE.make_expert_params_mapping(g, d, u, self.num_experts())
        return load_weights_with_mapping(self, weights, self.stacked_params_mapping, tie_lm_head=self.tie,
                                         expert_mapping=experts)


class QwenMoeDecoderLayer(LlamaDecoderLayer):
    supports_sequence_parallel = False
    def _make_mlp(self, config, vllm_config, quant_config, prefix):
        layer_idx = int(prefix.split(".")[-2])
        sparse_step = getattr(config, "decoder_sparse_step", 1) or 1
        only_dense = layer_idx in (getattr(config, "mlp_only_layers", None) or [])
        if only_dense or _num_experts(config) <= 0 or (layer_idx + 1) % sparse_step:
            return LlamaMLP(config.hidden_size, config.intermediate_size, False, quant_config, prefix)
        return SparseMoeBlock(config.hidden_size, config.moe_intermediate_size, _num_experts(config),
                              config.num_experts_per_tok, bool(getattr(config, "norm_topk_prob", False)), quant_config,
                              prefix, shared_intermediate=int(getattr(config, "shared_expert_intermediate_size", 0) or 0),
                              shared_gate=hasattr(config, "shared_expert_intermediate_size"))


class QwenMoeModel(LlamaModel):
    layer_cls = QwenMoeDecoderLayer


class Qwen2MoeForCausalLM(MixtralForCausalLM):
    model_cls = QwenMoeModel
    stacked_params_mapping = LlamaForCausalLM.stacked_params_mapping
    packed_modules_mapping = LlamaForCausalLM.packed_modules_mapping
    expert_names = ("gate_proj", "down_proj", "up_proj")

    def num_experts(self) -> int:
        return _num_experts(self.config)


class Qwen3MoeForCausalLM(Qwen2MoeForCausalLM):
    pass


# ---- OLMoE -------------------------------------------------------------------------------------------
class OlmoeDecoderLayer(LlamaDecoderLayer):
    """Llama attention with full-width RMSNorm on q and k; every MLP is a routed expert block.
    Reference: vllm/model_executor/models/olmoe.py."""
    supports_sequence_parallel = False

    def _attention_extras(self, config, layer_idx: int) -> dict:
        return {"qk_norm": "full", "qkv_clip": getattr(config, "clip_qkv", None)}

    def _make_mlp(self, config, vllm_config, quant_config, prefix):
        return SparseMoeBlock(config.hidden_size, config.intermediate_size, _num_experts(config),
                              config.num_experts_per_tok, bool(getattr(config, "norm_topk_prob", False)),
                              quant_config, prefix)


class OlmoeModel(LlamaModel):
    layer_cls = OlmoeDecoderLayer


class OlmoeForCausalLM(Qwen2MoeForCausalLM):
    model_cls = OlmoeModel


# ---- Granite-MoE -------------------------------------------------------------------------------------
class GraniteMoeDecoderLayer(LlamaDecoderLayer):
    """Granite block (attention / residual multipliers) with a routed expert MLP.
    Reference: vllm/model_executor/models/granitemoe.py."""
    supports_sequence_parallel = False

    def _attention_extras(self, config, layer_idx: int) -> dict:
        return {"scale": float(getattr(config, "attention_multiplier", None) or
                               (config.hidden_size // config.num_attention_heads) ** -0.5)}

    def _make_mlp(self, config, vllm_config, quant_config, prefix):
        return SparseMoeBlock(config.hidden_size, config.intermediate_size, _num_experts(config),
                              config.num_experts_per_tok, True, quant_config,
                              prefix.replace(".mlp", ".block_sparse_moe"))

    def __init__(self, config, vllm_config, quant_config=None, prefix: str = ""):
        super().__init__(config, vllm_config, quant_config, prefix)
        self.block_sparse_moe = self.mlp
        del self.mlp
        self.residual_multiplier = float(getattr(config, "residual_multiplier", 1.0) or 1.0)

    def forward(self, positions, hidden_states, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states) * self.residual_multiplier
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.block_sparse_moe(hidden_states) * self.residual_multiplier, residual


class GraniteMoeModel(LlamaModel):
    layer_cls = GraniteMoeDecoderLayer
    embed_scale_of = staticmethod(lambda config: float(getattr(config, "embedding_multiplier", 1.0) or 1.0))


class GraniteMoeForCausalLM(MixtralForCausalLM):
    """The checkpoint stacks all experts: input_linear [E, 2I, H] (gate rows, then up rows) and
    output_linear [E, H, I]; they are dealt out per expert while loading."""
    model_cls = GraniteMoeModel

    @staticmethod
    def _logit_scale(config) -> float:
        return 1.0 / float(getattr(config, "logits_scaling", 1.0) or 1.0)

    def load_weights(self, weights):
        def fix(items):
            for name, t in items:
                if name.endswith(".block_sparse_moe.input_linear.weight"):
                    base = name[: -len("input_linear.weight")]
                    half = t.shape[1] // 2
                    for e in range(t.shape[0]):
                        yield f"{base}experts.{e}.w1.weight", t[e, :half]
                        yield f"{base}experts.{e}.w3.weight", t[e, half:]
                elif name.endswith(".block_sparse_moe.output_linear.weight"):
                    base = name[: -len("output_linear.weight")]
                    for e in range(t.shape[0]):
                        yield f"{base}experts.{e}.w2.weight", t[e]
                elif name.endswith(".block_sparse_moe.router.layer.weight"):
                    yield name.replace(".router.layer.", ".gate."), t
                else:
                    yield name, t
        return super().load_weights(fix(weights))


class GraniteMoeSharedDecoderLayer(GraniteMoeDecoderLayer):
    """Granite-MoE with a dense gated MLP beside the routed experts (its checkpoint stores the gate and up
    halves stacked in ``shared_mlp.input_linear``)."""

    def __init__(self, config, vllm_config, quant_config=None, prefix: str = ""):
        super().__init__(config, vllm_config, quant_config, prefix)
        inter = int(getattr(config, "shared_intermediate_size", 0) or 0)
        self.shared_mlp = LlamaMLP(config.hidden_size, inter, False, quant_config, f"{prefix}.shared_mlp") \
            if inter > 0 else None

    def forward(self, positions, hidden_states, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states) * self.residual_multiplier
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        out = self.block_sparse_moe(hidden_states)
        if self.shared_mlp is not None:
            out = out + self.shared_mlp(hidden_states)
        return out * self.residual_multiplier, residual


class GraniteMoeSharedModel(GraniteMoeModel):
    layer_cls = GraniteMoeSharedDecoderLayer


class GraniteMoeSharedForCausalLM(GraniteMoeForCausalLM):
    model_cls = GraniteMoeSharedModel

    def load_weights(self, weights):
        def fix(items):
            for name, t in items:
                if name.endswith(".shared_mlp.input_linear.weight"):
                    yield name.replace(".input_linear.", ".gate_up_proj."), t      # already [gate rows | up rows]
                elif name.endswith(".shared_mlp.output_linear.weight"):
                    yield name.replace(".output_linear.", ".down_proj."), t
                else:
                    yield name, t
        return super().load_weights(fix(weights))


# ---- Phi-3.5-MoE -------------------------------------------------------------------------------------
def sparsemixer_routing(jitter_eps: float):
    """Phi-MoE's top-2 router at inference: each pick is the arg-max of the (remaining) logits, weighted
    by a softmax over the logits that lie within a relative band of 2 * jitter_eps below that maximum.
    Reference: vllm/model_executor/models/phimoe.py (phimoe_routing_function)."""
    def route(logits: torch.Tensor, top_k: int, renormalize: bool):
        if top_k != 2:
            raise ValueError("the sparsemixer router is defined for two experts per token")
        scores = logits.float()
        ws, ids = [], []
        work = scores
        for _ in range(2):
            mx, ind = work.max(dim=-1, keepdim=True)
            factor = scores.abs().clamp(min=mx)
            outside = ((mx - scores) / factor) > (2 * jitter_eps)
            gates = torch.softmax(work.masked_fill(outside, float("-inf")), dim=-1)
            ws.append(gates.gather(-1, ind))
            ids.append(ind)
            work = scores.scatter(-1, ind, float("-inf"))
        return torch.cat(ws, dim=-1), torch.cat(ids, dim=-1).to(torch.int32)
    return route


class PhiMoEDecoderLayer(MixtralDecoderLayer):
    @classmethod
    def make_norm(cls, config):
        from .llama_variants import ResidualLayerNorm
        return ResidualLayerNorm(config.hidden_size, getattr(config, "rms_norm_eps", 1e-5))

    def _attention_extras(self, config, layer_idx: int) -> dict:
        b = bool(getattr(config, "attention_bias", False))
        return {"bias": b, "o_bias": b}

    def _make_mlp(self, config, vllm_config, quant_config, prefix):
        return SparseMoeBlock(config.hidden_size, config.intermediate_size, _num_experts(config),
                              config.num_experts_per_tok, False, quant_config,
                              prefix.replace(".mlp", ".block_sparse_moe"),
                              routing=sparsemixer_routing(float(getattr(config, "router_jitter_noise", 0.01))))


class PhiMoEModel(LlamaModel):
    layer_cls = PhiMoEDecoderLayer


class PhiMoEForCausalLM(MixtralForCausalLM):
    model_cls = PhiMoEModel

    def __init__(self, vllm_config, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        if getattr(self.config, "lm_head_bias", False) and hasattr(self.lm_head, "weight"):
            from ..layers.basic import ParallelLMHead
            self.lm_head = ParallelLMHead(self.config.vocab_size, self.config.hidden_size, bias=True)

    def compute_logits(self, hidden_states):
        return self.logits_processor(self.lm_head, hidden_states, getattr(self.lm_head, "bias", None))


# ---- DBRX --------------------------------------------------------------------------------------------
class DbrxDecoderLayer(MixtralDecoderLayer):
    """Pre-norm block with scale-only LayerNorms, clipped fused QKV and a 16-expert top-4 router.
    Reference: vllm/model_executor/models/dbrx.py."""

    @classmethod
    def make_norm(cls, config):
        from .llama_variants import ResidualLayerNorm
        return ResidualLayerNorm(config.hidden_size, 1e-5, bias=False)

    def _attention_extras(self, config, layer_idx: int) -> dict:
        return {"qkv_clip": getattr(config, "clip_qkv", None), "bias": False, "o_bias": False}


class DbrxModel(LlamaModel):
    layer_cls = DbrxDecoderLayer


class DbrxForCausalLM(MixtralForCausalLM):
    model_cls = DbrxModel

    def __init__(self, vllm_config, prefix: str = ""):
        c = vllm_config.model_config.hf_config
        if not getattr(c, "hidden_size", None):
            # the checkpoint's config nests its sizes; give them the names the shared layers read
            ac, fc = c.attn_config, c.ffn_config
            g = (lambda o, k, d=None: o.get(k, d)) if isinstance(ac, dict) else (lambda o, k, d=None: getattr(o, k, d))
            c.hidden_size, c.num_attention_heads, c.num_hidden_layers = c.d_model, c.n_heads, c.n_layers
            c.num_key_value_heads = g(ac, "kv_n_heads", c.n_heads)
            c.clip_qkv, c.rope_theta = g(ac, "clip_qkv"), float(g(ac, "rope_theta", 10000.0))
            c.intermediate_size = g(fc, "ffn_hidden_size")
            c.num_local_experts, c.num_experts_per_tok = g(fc, "moe_num_experts"), g(fc, "moe_top_k")
            c.max_position_embeddings = c.max_seq_len
            c.tie_word_embeddings = bool(getattr(c, "tie_word_embeddings", False))
        super().__init__(vllm_config, prefix)

    def load_weights(self, weights):
        E, F_ = self.num_experts(), int(self.config.intermediate_size)
        top = {"transformer.wte.": "model.embed_tokens.", "transformer.norm_f.": "model.norm."}
        inner = ((".norm_attn_norm.norm_1.", ".input_layernorm."), (".norm_attn_norm.norm_2.", ".post_attention_layernorm."),
                 (".norm_attn_norm.attn.Wqkv.", ".self_attn.qkv_proj."), (".norm_attn_norm.attn.out_proj.", ".self_attn.o_proj."),
                 (".ffn.router.layer.", ".block_sparse_moe.gate."))

        def fix(items):
            for name, t in items:
                for a, b in top.items():
                    if name.startswith(a):
                        name = b + name[len(a):]
                name = name.replace("transformer.blocks.", "model.layers.")
                for a, b in inner:
                    name = name.replace(a, b)
                if ".ffn.experts.mlp." in name:
                    base, leaf = name.split(".ffn.experts.mlp.")
                    per = t.view(E, F_, -1)             # experts stacked along the rows
                    for e in range(E):
                        if leaf == "w2":                # stored [F, d]: the down projection's transpose
                            yield f"{base}.block_sparse_moe.experts.{e}.w2.weight", per[e].t().contiguous()
                        else:
                            yield f"{base}.block_sparse_moe.experts.{e}.{'w1' if leaf == 'w1' else 'w3'}.weight", per[e]
                    continue
                yield name, t

        g, d, u = self.expert_names
        experts = FusedMoE.make_expert_params_mapping(g, d, u, E)
        return load_weights_with_mapping(self, fix(weights), [], tie_lm_head=self.tie, expert_mapping=experts)


# ---- Snowflake Arctic --------------------------------------------------------------------------------
class ArcticDecoderLayer(LlamaDecoderLayer):
    """Every ``moe_layer_frequency``-th layer routes over experts, the others carry a dense gated MLP; with
    ``use_residual`` an expert layer also runs a small dense MLP on the post-attention stream while the experts read
    the normalised layer INPUT, and both are added to the post-attention stream.
    Reference: vllm/model_executor/models/arctic.py:296-360."""
    supports_sequence_parallel = False

    def _make_mlp(self, config, vllm_config, quant_config, prefix):
        idx = int(prefix.split(".")[-2])
        self.is_moe = (idx + 1) % int(getattr(config, "moe_layer_frequency", 2)) == 0
        self.use_residual = bool(getattr(config, "use_residual", True)) and self.is_moe
        p = prefix.replace(".mlp", ".block_sparse_moe")
        if self.use_residual:
            self.residual_layernorm = self.make_norm(config)
            self.residual_mlp = LlamaMLP(config.hidden_size, config.hidden_size, False, quant_config,
                                         prefix.replace(".mlp", ".residual_mlp"))
        if self.is_moe:
            return SparseMoeBlock(config.hidden_size, config.intermediate_size, _num_experts(config),
                                  config.num_experts_per_tok, int(config.num_experts_per_tok) > 1, quant_config, p)
        m = nn.Module()
        m.mlp = LlamaMLP(config.hidden_size, config.intermediate_size, False, quant_config, f"{p}.mlp")
        m.forward = lambda x: m.mlp(x)
        return m

    def __init__(self, config, vllm_config, quant_config=None, prefix: str = ""):
        super().__init__(config, vllm_config, quant_config, prefix)
        self.block_sparse_moe = self.mlp
        del self.mlp

    def forward(self, positions, hidden_states, residual):
        x = hidden_states if residual is None else hidden_states + residual          # the layer's input
        attn = x + self.self_attn(positions, self.input_layernorm(x))
        if self.use_residual:
            out = self.residual_mlp(self.residual_layernorm(attn)) + \
                self.block_sparse_moe(self.post_attention_layernorm(x))
        else:
            out = self.block_sparse_moe(self.post_attention_layernorm(attn))
        return out, attn


class ArcticModel(LlamaModel):
    layer_cls = ArcticDecoderLayer


class ArcticForCausalLM(MixtralForCausalLM):
    model_cls = ArcticModel
    stacked_params_mapping = MixtralForCausalLM.stacked_params_mapping + [
        (".gate_up_proj", ".w1", 0), (".gate_up_proj", ".w3", 1)]

    def num_experts(self) -> int:
        return _num_experts(self.config)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]) -> set[str]:
        experts = FusedMoE.make_expert_params_mapping("w1", "w2", "w3", self.num_experts())

        def fix(items):
            for n, t in items:
                if (".residual_mlp." in n or ".block_sparse_moe.mlp." in n) and ".w2." in n:
                    n = n.replace(".w2.", ".down_proj.")
                yield n, t
        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping, tie_lm_head=self.tie,
                                         expert_mapping=experts)


# ---- Grok-1 ------------------------------------------------------------------------------------------
class Grok1DecoderLayer(LlamaDecoderLayer):
    """Grok-1 block: soft-capped attention whose output is scaled by ``attn_output_multiplier``, a routed expert MLP
    with a GELU gate (fused into the first expert GEMM), and a norm on BOTH sides of each of the two sub-blocks.
    Reference: vllm/model_executor/models/grok1.py (Grok1MoE :62-111, Grok1Attention :114-200,
    Grok1DecoderLayer :203-284)."""
    supports_sequence_parallel = False

    def _attention_extras(self, config, layer_idx: int) -> dict:
        cap = getattr(config, "attn_logit_softcapping", 30.0)
        return {"logits_soft_cap": max(float(30.0 if cap is None else cap), 0.0) or None}

    def _make_mlp(self, config, vllm_config, quant_config, prefix):
        return SparseMoeBlock(config.hidden_size, config.intermediate_size, _num_experts(config) or 8,
                              config.num_experts_per_tok, True, quant_config, prefix.replace(".mlp", ".moe_block"),
                              activation="gelu")

    def __init__(self, config, vllm_config, quant_config=None, prefix: str = ""):
        super().__init__(config, vllm_config, quant_config, prefix)
        self.moe_block = self.mlp                  # parameter names follow the checkpoint
        del self.mlp
        self.pre_moe_norm = self.make_norm(config)
        self.post_moe_norm = self.make_norm(config)
        self.attn_multiplier = float(getattr(config, "attn_output_multiplier", 0.08838834764831845))

    def forward(self, positions, hidden_states, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.post_attention_layernorm(self.self_attn(positions, hidden_states) * self.attn_multiplier)
        hidden_states, residual = self.pre_moe_norm(hidden_states, residual)
        return self.post_moe_norm(self.moe_block(hidden_states)), residual


class Grok1Model(LlamaModel):
    layer_cls = Grok1DecoderLayer
    embed_scale_of = staticmethod(lambda config: float(getattr(config, "embedding_multiplier_scale", 78.38367176906169)))


class Grok1ModelForCausalLM(MixtralForCausalLM):
    """Checkpoint names: ``attn`` / ``pre_attn_norm`` / ``post_attn_norm`` / ``moe_block`` with experts
    ``linear`` (gate), ``linear_v`` (up), ``linear_1`` (down), and norm weights called ``scale``.  Logits are
    multiplied by ``output_multiplier_scale``."""
    model_cls = Grok1Model
    expert_names = ("linear", "linear_1", "linear_v")
    _RENAMES = ((".attn.", ".self_attn."), (".pre_attn_norm.", ".input_layernorm."),
                (".post_attn_norm.", ".post_attention_layernorm."))

    @staticmethod
    def _logit_scale(config) -> float:
        return float(getattr(config, "output_multiplier_scale", 0.5773502691896257))

    def num_experts(self) -> int:
        return _num_experts(self.config) or 8

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]) -> set[str]:
        def renamed():
            for name, t in weights:
                if name.endswith("norm.scale"):
                    name = name[: -len("scale")] + "weight"
                for a, b in self._RENAMES:
                    name = name.replace(a, b)
                yield name, t
        return super().load_weights(renamed())
