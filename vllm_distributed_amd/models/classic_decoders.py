"""LayerNorm decoders with ALiBi or rotary positions and a plain (non-gated) MLP: Falcon (all three
checkpoint generations), BLOOM, MPT and Phi-1.5 / Phi-2.  ALiBi is a per-head slope handed to the
paged-attention kernels; nothing else distinguishes these from the rotary models on the GPU.
Reference: vllm/model_executor/models/{falcon,bloom,mpt,phi}.py.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ..config import VllmConfig
from ..layers.attention import Attention
from ..layers.basic import LayerNorm, LogitsProcessor, ParallelLMHead, VocabParallelEmbedding, get_act_fn
from ..layers.linear import ColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from ..layers.rotary import get_rope
from ..parallel.state import get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size
from .gpt_family import _Base, _rope_theta
from .llama_variants import alibi_slopes
from .utils import load_weights_with_mapping


def mpt_alibi_slopes(num_heads: int, alibi_bias_max: float = 8.0) -> list[float]:
    """MPT's variant: slopes of the next power of two, odd positions first when the head count is not one."""
    p = 2 ** math.ceil(math.log2(num_heads))
    slopes = [1.0 / 2.0 ** ((i + 1) * alibi_bias_max / p) for i in range(p)]
    if p != num_heads:
        slopes = (slopes[1::2] + slopes[::2])[:num_heads]
    return slopes


class _FusedAttention(nn.Module):
    """Fused QKV GEMM -> rotary (optional) + cache write -> attention (optional ALiBi) -> output GEMM.
    The projection attributes carry the checkpoint's names."""

    def __init__(self, vllm_config: VllmConfig, hidden: int, heads: int, kv_heads: int, bias: bool, out_bias: bool,
                 prefix: str, qkv_name: str, out_name: str, rotary=None, slopes=None, scale=None, clip=None):
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        r = get_tensor_model_parallel_rank()
        D = hidden // heads
        self.num_heads, self.num_kv_heads = heads // tp, max(1, kv_heads // tp)
        setattr(self, qkv_name, QKVParallelLinear(hidden, D, heads, kv_heads, bias=bias, prefix=f"{prefix}.{qkv_name}"))
        setattr(self, out_name, RowParallelLinear(hidden, hidden, bias=out_bias, prefix=f"{prefix}.{out_name}"))
        self._qkv, self._out = qkv_name, out_name
        self.rotary_emb = rotary
        self.clip = clip
        if slopes is not None:
            slopes = list(slopes)[r * self.num_heads:(r + 1) * self.num_heads]
        self.attn = Attention(self.num_heads, D, scale if scale is not None else D ** -0.5, self.num_kv_heads,
                              alibi_slopes=slopes, prefix=f"{prefix}.attn",
                              cache_dtype=vllm_config.cache_config.cache_dtype)

    def forward(self, positions, x):
        qkv = getattr(self, self._qkv)(x)
        if self.clip:
            qkv.clamp_(-self.clip, self.clip)
        return getattr(self, self._out)(self.attn(qkv, positions, self.rotary_emb))


def _heads_first_to_qkv(t: torch.Tensor, heads: int, head_dim: int) -> torch.Tensor:
    """[heads, (q, k, v), head_dim, ...] rows -> [all q | all k | all v] rows."""
    return t.view(heads, 3, head_dim, *t.shape[1:]).transpose(0, 1).reshape(t.shape)


# ---- Falcon ------------------------------------------------------------------------------------------
class FalconLayer(nn.Module):
    def __init__(self, config, vllm_config, prefix: str):
        super().__init__()
        d, H = config.hidden_size, config.num_attention_heads
        self.new_arch = bool(getattr(config, "new_decoder_architecture", False))
        self.parallel = bool(getattr(config, "parallel_attn", True)) or self.new_arch
        if self.new_arch:
            kv = getattr(config, "num_kv_heads", None) or H
        else:
            kv = 1 if getattr(config, "multi_query", True) else H
        self.kv_heads = kv
        D = d // H
        bias = bool(getattr(config, "bias", False))
        eps = config.layer_norm_epsilon
        rotary = slopes = None
        if getattr(config, "alibi", False):
            # the checkpoint's attention scales the ALiBi term together with the scores
            slopes = [s * D ** -0.5 for s in alibi_slopes(H)]
        else:
            rotary = get_rope(D, D, vllm_config.model_config.max_model_len, _rope_theta(config), True)
        self.self_attention = _FusedAttention(vllm_config, d, H, kv, bias, bias, f"{prefix}.self_attention",
                                              "query_key_value", "dense", rotary, slopes)
        n_ln = getattr(config, "num_ln_in_parallel_attn", None)
        if n_ln is None and self.new_arch:
            n_ln = 2
        self.two_norms = self.new_arch and n_ln == 2
        if not self.parallel:
            self.input_layernorm = LayerNorm(d, eps)
            self.post_attention_layernorm = LayerNorm(d, eps)
        elif self.two_norms:
            self.ln_attn = LayerNorm(d, eps)
            self.ln_mlp = LayerNorm(d, eps)
        else:
            self.input_layernorm = LayerNorm(d, eps)
        inner = getattr(config, "ffn_hidden_size", None) or 4 * d
        self.dense_h_to_4h = ColumnParallelLinear(d, inner, bias=bias, prefix=f"{prefix}.mlp.dense_h_to_4h")
        self.dense_4h_to_h = RowParallelLinear(inner, d, bias=bias, prefix=f"{prefix}.mlp.dense_4h_to_h")
        self.act = get_act_fn(getattr(config, "activation", "gelu"))

    def _mlp(self, x):
        return self.dense_4h_to_h(self.act(self.dense_h_to_4h(x)))

    def forward(self, positions, x):
        if self.two_norms:
            a, m = self.ln_attn(x), self.ln_mlp(x)
        else:
            a = m = self.input_layernorm(x)
        attn = self.self_attention(positions, a)
        if not self.parallel:
            x = x + attn
            return x + self._mlp(self.post_attention_layernorm(x))
        return x + attn + self._mlp(m)


class FalconForCausalLM(_Base):
    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        self.word_embeddings = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.h = nn.ModuleList([FalconLayer(config, vllm_config, f"transformer.h.{i}")
                                for i in range(config.num_hidden_layers)])
        self.ln_f = LayerNorm(config.hidden_size, config.layer_norm_epsilon)
        self.tie = bool(getattr(config, "tie_word_embeddings", True))
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if self.tie:
            self.lm_head.tie_weights(self.word_embeddings)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    def get_input_embeddings(self, ids):
        return self.word_embeddings(ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        x = inputs_embeds if inputs_embeds is not None else self.word_embeddings(input_ids)
        for layer in self.h:
            x = layer(positions, x)
        return self.ln_f(x)

    def load_weights(self, weights):
        c = self.config
        H = c.num_attention_heads
        D = c.hidden_size // H
        layer0 = self.h[0]
        KV, new_arch = layer0.kv_heads, layer0.new_arch

        def fix(items):
            for name, t in items:
                name = name.removeprefix("transformer.").replace(".mlp.dense_", ".dense_")
                if "query_key_value" in name:
                    if new_arch:
                        # per KV head: its query heads, then the key head, then the value head
                        g = H // KV
                        v = t.view(KV, g + 2, D, *t.shape[1:])
                        t = torch.cat([v[:, :g].reshape(H * D, *t.shape[1:]), v[:, g].reshape(KV * D, *t.shape[1:]),
                                       v[:, g + 1].reshape(KV * D, *t.shape[1:])], dim=0)
                    elif KV == H:
                        t = _heads_first_to_qkv(t, H, D)
                    # multi-query checkpoints are already [q heads | k | v]
                yield name, t

        return load_weights_with_mapping(self, fix(weights), [], tie_lm_head=self.tie)


RWForCausalLM = FalconForCausalLM


# ---- BLOOM -------------------------------------------------------------------------------------------
class BloomBlock(nn.Module):
    def __init__(self, config, vllm_config, prefix: str):
        super().__init__()
        d, H = config.hidden_size, config.n_head
        eps = config.layer_norm_epsilon
        self.post_ln_residual = bool(getattr(config, "apply_residual_connection_post_layernorm", False))
        self.input_layernorm = LayerNorm(d, eps)
        self.self_attention = _FusedAttention(vllm_config, d, H, H, True, True, f"{prefix}.self_attention",
                                              "query_key_value", "dense", None, alibi_slopes(H))
        self.post_attention_layernorm = LayerNorm(d, eps)
        self.dense_h_to_4h = ColumnParallelLinear(d, 4 * d, bias=True, prefix=f"{prefix}.mlp.dense_h_to_4h")
        self.dense_4h_to_h = RowParallelLinear(4 * d, d, bias=True, prefix=f"{prefix}.mlp.dense_4h_to_h")
        self.act = get_act_fn("gelu_tanh")

    def forward(self, positions, x):
        n = self.input_layernorm(x)
        x = (n if self.post_ln_residual else x) + self.self_attention(positions, n)
        n = self.post_attention_layernorm(x)
        return (n if self.post_ln_residual else x) + self.dense_4h_to_h(self.act(self.dense_h_to_4h(n)))


class BloomForCausalLM(_Base):
    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        d = config.hidden_size
        self.word_embeddings = VocabParallelEmbedding(config.vocab_size, d)
        self.word_embeddings_layernorm = LayerNorm(d, config.layer_norm_epsilon)
        self.h = nn.ModuleList([BloomBlock(config, vllm_config, f"transformer.h.{i}") for i in range(config.n_layer)])
        self.ln_f = LayerNorm(d, config.layer_norm_epsilon)
        self.lm_head = ParallelLMHead(config.vocab_size, d).tie_weights(self.word_embeddings)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    def get_input_embeddings(self, ids):
        return self.word_embeddings_layernorm(self.word_embeddings(ids))

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        x = inputs_embeds if inputs_embeds is not None else self.get_input_embeddings(input_ids)
        for blk in self.h:
            x = blk(positions, x)
        return self.ln_f(x)

    def load_weights(self, weights):
        H = self.config.n_head
        D = self.config.hidden_size // H

        def fix(items):
            for name, t in items:
                if name.startswith("lm_head"):
                    continue
                name = name.removeprefix("transformer.").replace(".mlp.dense_", ".dense_")
                if "query_key_value" in name:
                    t = _heads_first_to_qkv(t, H, D)
                yield name, t

        return load_weights_with_mapping(self, fix(weights), [])


# ---- MPT ---------------------------------------------------------------------------------------------
class MptBlock(nn.Module):
    def __init__(self, config, vllm_config, prefix: str):
        super().__init__()
        d, H = config.d_model, config.n_heads
        ac = config.attn_config
        get = (lambda k, dflt=None: ac.get(k, dflt)) if isinstance(ac, dict) else (lambda k, dflt=None: getattr(ac, k, dflt))
        if get("qk_ln", False):
            raise ValueError("MPT checkpoints with q/k LayerNorm are not supported")
        if not get("alibi", True):
            raise ValueError("MPT checkpoints without ALiBi (learned positions) are not supported")
        bias = not bool(getattr(config, "no_bias", True))
        eps = getattr(config, "layer_norm_epsilon", 1e-5)
        self.norm_1 = LayerNorm(d, eps, bias=bias)
        self.attn = _FusedAttention(vllm_config, d, H, H, bias, bias, f"{prefix}.attn", "Wqkv", "out_proj", None,
                                    mpt_alibi_slopes(H, float(get("alibi_bias_max", 8))), get("softmax_scale", None),
                                    get("clip_qkv", None))
        self.norm_2 = LayerNorm(d, eps, bias=bias)
        inner = int(getattr(config, "expansion_ratio", 4) * d)
        self.up_proj = ColumnParallelLinear(d, inner, bias=bias, prefix=f"{prefix}.ffn.up_proj")
        self.down_proj = RowParallelLinear(inner, d, bias=bias, prefix=f"{prefix}.ffn.down_proj")
        self.act = get_act_fn("gelu")

    def forward(self, positions, x):
        x = x + self.attn(positions, self.norm_1(x))
        return x + self.down_proj(self.act(self.up_proj(self.norm_2(x))))


class MptForCausalLM(_Base):
    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        d = config.d_model
        bias = not bool(getattr(config, "no_bias", True))
        self.wte = VocabParallelEmbedding(config.vocab_size, d)
        self.blocks = nn.ModuleList([MptBlock(config, vllm_config, f"transformer.blocks.{i}")
                                     for i in range(config.n_layers)])
        self.norm_f = LayerNorm(d, getattr(config, "layer_norm_epsilon", 1e-5), bias=bias)
        self.lm_head = ParallelLMHead(config.vocab_size, d).tie_weights(self.wte)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    def get_input_embeddings(self, ids):
        return self.wte(ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        x = inputs_embeds if inputs_embeds is not None else self.wte(input_ids)
        for blk in self.blocks:
            x = blk(positions, x)
        return self.norm_f(x)

    def load_weights(self, weights):
        def fix(items):
            for name, t in items:
                if name.startswith("lm_head"):
                    continue
                yield name.removeprefix("transformer.").replace(".ffn.up_proj", ".up_proj") \
                          .replace(".ffn.down_proj", ".down_proj"), t
        return load_weights_with_mapping(self, fix(weights), [])


MPTForCausalLM = MptForCausalLM


# ---- Phi-1.5 / Phi-2 ---------------------------------------------------------------------------------
class PhiLayer(nn.Module):
    """One LayerNorm, attention and MLP side by side; rotary over a fraction of each head."""

    def __init__(self, config, vllm_config, prefix: str):
        super().__init__()
        d, H = config.hidden_size, config.num_attention_heads
        kv = getattr(config, "num_key_value_heads", None) or H
        D = d // H
        if getattr(config, "qk_layernorm", False):
            raise ValueError("Phi checkpoints with q/k LayerNorm are not supported")
        pct = getattr(config, "partial_rotary_factor", None) or \
            (getattr(config, "rope_parameters", None) or {}).get("partial_rotary_factor") or 0.5
        rotary = get_rope(D, int(D * pct), vllm_config.model_config.max_model_len, _rope_theta(config), True)
        self.input_layernorm = LayerNorm(d, config.layer_norm_eps)
        self.self_attn = _FusedAttention(vllm_config, d, H, kv, True, True, f"{prefix}.self_attn", "qkv_proj",
                                         "dense", rotary)
        self.fc1 = ColumnParallelLinear(d, config.intermediate_size, bias=True, prefix=f"{prefix}.mlp.fc1")
        self.fc2 = RowParallelLinear(config.intermediate_size, d, bias=True, prefix=f"{prefix}.mlp.fc2")
        self.act = get_act_fn(config.hidden_act)

    def forward(self, positions, x):
        n = self.input_layernorm(x)
        return x + self.self_attn(positions, n) + self.fc2(self.act(self.fc1(n)))


class PhiForCausalLM(_Base):
    stacked_params_mapping = [(".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"),
                              (".qkv_proj", ".v_proj", "v")]
    packed_modules_mapping = {"qkv_proj": ["q_proj", "k_proj", "v_proj"]}

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([PhiLayer(config, vllm_config, f"model.layers.{i}")
                                     for i in range(config.num_hidden_layers)])
        self.final_layernorm = LayerNorm(config.hidden_size, config.layer_norm_eps)
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size, bias=True)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    def get_input_embeddings(self, ids):
        return self.embed_tokens(ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        x = inputs_embeds if inputs_embeds is not None else self.embed_tokens(input_ids)
        for layer in self.layers:
            x = layer(positions, x)
        return self.final_layernorm(x)

    def compute_logits(self, hidden):
        return self.logits_processor(self.lm_head, hidden, getattr(self.lm_head, "bias", None))

    def load_weights(self, weights):
        def fix(items):
            for name, t in items:
                yield name.removeprefix("model.").replace(".mlp.fc", ".fc"), t
        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping)
