"""OPT and GPT-2: learned absolute positions, LayerNorm, plain MLP, MHA.
Reference: vllm/model_executor/models/opt.py, gpt2.py."""
from __future__ import annotations

from typing import Iterable, Optional

import torch
import torch.nn as nn

from ..config import VllmConfig
from ..layers.attention import Attention
from ..layers.basic import LayerNorm, LogitsProcessor, ParallelLMHead, VocabParallelEmbedding, get_act_fn
from ..layers.linear import ColumnParallelLinear, QKVParallelLinear, ReplicatedLinear, RowParallelLinear
from ..parallel.state import get_pp_group, get_tensor_model_parallel_world_size
from .utils import (IntermediateTensors, PPMissingLayer, load_weights_with_mapping, make_layers, maybe_prefix)


class OPTLearnedPositionalEmbedding(nn.Embedding):
    def __init__(self, num_embeddings: int, embedding_dim: int):
        self.offset = 2      # OPT reserves the first two rows
        super().__init__(num_embeddings + self.offset, embedding_dim)
        self.weight.requires_grad_(False)

    def forward(self, positions: torch.Tensor):
        return super().forward(positions + self.offset)


class OPTAttention(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, bias: bool, prefix: str):
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        self.num_heads = num_heads // tp
        self.head_dim = embed_dim // num_heads
        self.qkv_proj = QKVParallelLinear(embed_dim, self.head_dim, num_heads, bias=bias, prefix=f"{prefix}.qkv_proj")
        self.out_proj = RowParallelLinear(embed_dim, embed_dim, bias=bias, prefix=f"{prefix}.out_proj")
        self.attn = Attention(self.num_heads, self.head_dim, self.head_dim ** -0.5, prefix=f"{prefix}.attn")

    def forward(self, x):
        return self.out_proj(self.attn(self.qkv_proj(x)))


class OPTDecoderLayer(nn.Module):
    def __init__(self, config, prefix: str):
        super().__init__()
        d = config.hidden_size
        bias = getattr(config, "enable_bias", True)
        self.self_attn = OPTAttention(d, config.num_attention_heads, bias, f"{prefix}.self_attn")
        self.do_layer_norm_before = config.do_layer_norm_before
        affine = getattr(config, "layer_norm_elementwise_affine", True)
        self.self_attn_layer_norm = LayerNorm(d, bias=affine)
        self.fc1 = ColumnParallelLinear(d, config.ffn_dim, bias=bias, prefix=f"{prefix}.fc1")
        self.activation_fn = get_act_fn(config.activation_function)
        self.fc2 = RowParallelLinear(config.ffn_dim, d, bias=bias, prefix=f"{prefix}.fc2")
        self.final_layer_norm = LayerNorm(d, bias=affine)

    def forward(self, x):
        r = x
        if self.do_layer_norm_before:
            x = self.self_attn_layer_norm(x)
        x = r + self.self_attn(x)
        if not self.do_layer_norm_before:
            x = self.self_attn_layer_norm(x)
        r = x
        if self.do_layer_norm_before:
            x = self.final_layer_norm(x)
        x = r + self.fc2(self.activation_fn(self.fc1(x)))
        if not self.do_layer_norm_before:
            x = self.final_layer_norm(x)
        return x


class OPTDecoder(nn.Module):
    def __init__(self, vllm_config: VllmConfig, prefix: str):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        wd = config.word_embed_proj_dim
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, wd)
        self.embed_positions = OPTLearnedPositionalEmbedding(config.max_position_embeddings, config.hidden_size)
        self.project_out = ReplicatedLinear(config.hidden_size, wd, bias=False) if wd != config.hidden_size else None
        self.project_in = ReplicatedLinear(wd, config.hidden_size, bias=False) if wd != config.hidden_size else None
        if config.do_layer_norm_before and not getattr(config, "_remove_final_layer_norm", False):
            self.final_layer_norm = LayerNorm(config.hidden_size,
                                              bias=getattr(config, "layer_norm_elementwise_affine", True))
        else:
            self.final_layer_norm = None
        self.start_layer, self.end_layer, self.layers = make_layers(
            config.num_hidden_layers, lambda p: OPTDecoderLayer(config, p), f"{prefix}.layers")

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        pp = get_pp_group()
        if pp.is_first_rank:
            x = inputs_embeds if inputs_embeds is not None else self.embed_tokens(input_ids)
            if self.project_in is not None:
                x = self.project_in(x)
            x = x + self.embed_positions(positions)
        else:
            x = intermediate_tensors["hidden_states"]
        for layer in self.layers[self.start_layer:self.end_layer]:
            x = layer(x)
        if not pp.is_last_rank:
            return IntermediateTensors({"hidden_states": x})
        if self.final_layer_norm is not None:
            x = self.final_layer_norm(x)
        if self.project_out is not None:
            x = self.project_out(x)
        return x


class OPTModel(nn.Module):
    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        self.decoder = OPTDecoder(vllm_config, f"{prefix}.decoder")

    def forward(self, *a, **kw):
        return self.decoder(*a, **kw)


class OPTForCausalLM(nn.Module):
    stacked_params_mapping = [(".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"),
                              (".qkv_proj", ".v_proj", "v")]
    packed_modules_mapping = {"qkv_proj": ["q_proj", "k_proj", "v_proj"]}
    supports_pp = True

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        self.model = OPTModel(vllm_config, maybe_prefix(prefix, "model"))
        self.tie = bool(getattr(config, "tie_word_embeddings", True))
        self.lm_head = ParallelLMHead(config.vocab_size, config.word_embed_proj_dim)
        if self.tie:
            self.lm_head.tie_weights(self.model.decoder.embed_tokens)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    def get_input_embeddings(self, ids):
        return self.model.decoder.embed_tokens(ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        return self.model(input_ids, positions, intermediate_tensors, inputs_embeds)

    def compute_logits(self, hidden):
        return self.logits_processor(self.lm_head, hidden)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]) -> set[str]:
        def rename(n: str):
            if n.startswith("decoder."):
                n = "model." + n
            return n
        return load_weights_with_mapping(self, weights, self.stacked_params_mapping, rename=rename,
                                         tie_lm_head=self.tie)


# ---------------------------------------------------------------------------
class GPT2Block(nn.Module):
    def __init__(self, config, prefix: str):
        super().__init__()
        d = config.n_embd
        inner = getattr(config, "n_inner", None) or 4 * d
        tp = get_tensor_model_parallel_world_size()
        self.ln_1 = LayerNorm(d, config.layer_norm_epsilon)
        self.c_attn = QKVParallelLinear(d, d // config.n_head, config.n_head, bias=True, prefix=f"{prefix}.attn.c_attn")
        self.c_proj = RowParallelLinear(d, d, bias=True, prefix=f"{prefix}.attn.c_proj")
        self.attn = Attention(config.n_head // tp, d // config.n_head, (d // config.n_head) ** -0.5,
                              prefix=f"{prefix}.attn")
        self.ln_2 = LayerNorm(d, config.layer_norm_epsilon)
        self.c_fc = ColumnParallelLinear(d, inner, bias=True, prefix=f"{prefix}.mlp.c_fc")
        self.mlp_proj = RowParallelLinear(inner, d, bias=True, prefix=f"{prefix}.mlp.c_proj")
        self.act = get_act_fn(config.activation_function)

    def forward(self, x):
        x = x + self.c_proj(self.attn(self.c_attn(self.ln_1(x))))
        return x + self.mlp_proj(self.act(self.c_fc(self.ln_2(x))))


class GPT2LMHeadModel(nn.Module):
    supports_pp = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        self.wte = VocabParallelEmbedding(config.vocab_size, config.n_embd)
        self.wpe = nn.Embedding(config.n_positions, config.n_embd)
        self.wpe.weight.requires_grad_(False)
        self.h = nn.ModuleList([GPT2Block(config, f"transformer.h.{i}") for i in range(config.n_layer)])
        self.ln_f = LayerNorm(config.n_embd, config.layer_norm_epsilon)
        self.lm_head = ParallelLMHead(config.vocab_size, config.n_embd).tie_weights(self.wte)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    def get_input_embeddings(self, ids):
        return self.wte(ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        x = (inputs_embeds if inputs_embeds is not None else self.wte(input_ids)) + self.wpe(positions)
        for blk in self.h:
            x = blk(x)
        return self.ln_f(x)

    def compute_logits(self, hidden):
        return self.logits_processor(self.lm_head, hidden)

    def load_weights(self, weights):
        params = dict(self.named_parameters())
        loaded = set()
        for name, t in weights:
            if name.startswith("lm_head") or ".attn.bias" in name or ".attn.masked_bias" in name:
                continue
            name = name.removeprefix("transformer.")
            name = name.replace(".attn.c_attn", ".c_attn").replace(".attn.c_proj", ".c_proj") \
                       .replace(".mlp.c_fc", ".c_fc").replace(".mlp.c_proj", ".mlp_proj")
            if name not in params:
                continue
            # GPT-2 stores Conv1D weights as [in, out]
            if any(k in name for k in ("c_attn", "c_proj", "c_fc", "mlp_proj")) and name.endswith(".weight"):
                t = t.t()
            p = params[name]
            getattr(p, "weight_loader", lambda p_, t_: p_.data.copy_(t_))(p, t)
            loaded.add(name)
        return loaded


# ---------------------------------------------------------------------------
def _alibi_slopes(n: int) -> list[float]:
    import math

    def pow2(m):
        start = 2 ** (-(2 ** -(math.log2(m) - 3)))
        return [start * start ** i for i in range(m)]
    if math.log2(n).is_integer():
        return pow2(n)
    c = 2 ** math.floor(math.log2(n))
    return pow2(c) + _alibi_slopes(2 * c)[0::2][: n - c]


class JAISBlock(GPT2Block):
    """GPT-2 block with ALiBi attention (softmax scale 1/d under muP), and x1 * silu(x2) over two up projections."""

    def __init__(self, config, prefix: str):
        super().__init__(config, prefix)
        d, H = config.n_embd, config.n_head
        tp = get_tensor_model_parallel_world_size()
        from ..parallel.state import get_tensor_model_parallel_rank
        r = get_tensor_model_parallel_rank()
        by_d = bool(getattr(config, "scale_qk_dot_by_d", getattr(config, "mup_scale_qk_dot_by_d", False)))
        self.attn = Attention(H // tp, d // H, (d // H) ** (-1.0 if by_d else -0.5),
                              alibi_slopes=_alibi_slopes(H)[r * (H // tp):(r + 1) * (H // tp)], prefix=f"{prefix}.attn")
        inner = getattr(config, "n_inner", None) or 4 * d
        self.swiglu = bool(getattr(config, "jais_swiglu", False))
        if self.swiglu:
            self.c_fc2 = ColumnParallelLinear(d, inner, bias=True, prefix=f"{prefix}.mlp.c_fc2")

    def forward(self, x):
        x = x + self.c_proj(self.attn(self.c_attn(self.ln_1(x))))
        h = self.ln_2(x)
        h = self.c_fc(h) * torch.nn.functional.silu(self.c_fc2(h)) if self.swiglu else self.act(self.c_fc(h))
        return x + self.mlp_proj(h)


class JAISLMHeadModel(GPT2LMHeadModel):
    """JAIS: GPT-2 layout with ALiBi instead of learned positions, SwiGLU MLPs and muP scales on the embedding,
    the attention logits and the output logits.  Reference: vllm/model_executor/models/jais.py:60-373."""

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        nn.Module.__init__(self)
        config = vllm_config.model_config.hf_config
        self.config = config
        self.wte = VocabParallelEmbedding(config.vocab_size, config.n_embd)
        self.alibi = getattr(config, "position_embedding_type", "alibi") == "alibi"
        if not self.alibi:
            self.wpe = nn.Embedding(config.n_positions, config.n_embd)
            self.wpe.weight.requires_grad_(False)
        real_act = getattr(config, "activation_function", "swiglu")
        config.jais_swiglu = real_act == "swiglu"
        if config.jais_swiglu:
            config.activation_function = "silu"          # the base block builds an (unused) plain activation
        self.h = nn.ModuleList([JAISBlock(config, f"transformer.h.{i}") for i in range(config.n_layer)])
        config.activation_function = real_act
        self.ln_f = LayerNorm(config.n_embd, config.layer_norm_epsilon)
        self.lm_head = ParallelLMHead(config.vocab_size, config.n_embd).tie_weights(self.wte)
        self.embeddings_scale = float(getattr(config, "embeddings_scale", None)
                                      or getattr(config, "mup_embeddings_scale", 1.0))
        out_scale = getattr(config, "width_scale", None)
        if out_scale is None:
            out_scale = float(getattr(config, "mup_output_alpha", 1.0)) * float(getattr(config, "mup_width_scale", 1.0))
        self.logits_processor = LogitsProcessor(config.vocab_size, config.vocab_size, float(out_scale))

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        x = inputs_embeds if inputs_embeds is not None else self.wte(input_ids)
        if not self.alibi:
            x = x + self.wpe(positions)
        x = x * self.embeddings_scale
        for blk in self.h:
            x = blk(x)
        return self.ln_f(x)

    def load_weights(self, weights):
        def fix():
            for name, t in weights:
                if "relative_pe" in name:
                    continue
                if ".mlp.c_fc2." in name:
                    name = name.replace(".mlp.c_fc2.", ".c_fc2.")
                    t = t.t() if name.endswith(".weight") else t
                    p = dict(self.named_parameters())[name.removeprefix("transformer.")]
                    getattr(p, "weight_loader", lambda p_, t_: p_.data.copy_(t_))(p, t)
                    done.add(name.removeprefix("transformer."))
                    continue
                yield name, t
        done: set[str] = set()
        loaded = super().load_weights(fix())
        return loaded | done
