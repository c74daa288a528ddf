"""LayerNorm-based decoders with rotary or learned positions: GPT-NeoX (Pythia), GPT-J,
StarCoder2, GPT-BigCode (SantaCoder / StarCoder).
Reference: vllm/model_executor/models/gpt_neox.py, gpt_j.py, starcoder2.py, gpt_bigcode.py.
"""
from __future__ import annotations

from typing import Iterable

import torch
import torch.nn as nn

from ..config import VllmConfig
from ..layers.attention import Attention
from ..layers.basic import LayerNorm, LogitsProcessor, ParallelLMHead, VocabParallelEmbedding, get_act_fn
from ..layers.linear import ColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from ..layers.rotary import get_rope
from ..parallel.state import get_tensor_model_parallel_world_size
from .utils import load_weights_with_mapping, maybe_prefix


def _rope_theta(config, default=10000.0) -> float:
    rs = getattr(config, "rope_parameters", None) or {}
    return float(getattr(config, "rope_theta", None) or rs.get("rope_theta") or
                 getattr(config, "rotary_emb_base", None) or default)


class _RotaryLNAttention(nn.Module):
    """QKV GEMM -> (partial) rotary + cache write -> attention -> output GEMM."""

    def __init__(self, config, vllm_config: VllmConfig, hidden: int, heads: int, kv_heads: int, rotary_dim: int,
                 neox_style: bool, bias: bool, out_bias: bool, prefix: str, qkv_name: str, out_name: str,
                 sliding_window=None):
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        self.head_dim = hidden // heads
        self.num_heads, self.num_kv_heads = heads // tp, max(1, kv_heads // tp)
        qkv = QKVParallelLinear(hidden, self.head_dim, heads, kv_heads, bias=bias, prefix=f"{prefix}.{qkv_name}")
        out = RowParallelLinear(hidden, hidden, bias=out_bias, prefix=f"{prefix}.{out_name}")
        setattr(self, qkv_name, qkv)
        setattr(self, out_name, out)
        self._qkv, self._out = qkv_name, out_name
        self.rotary_emb = get_rope(self.head_dim, rotary_dim, vllm_config.model_config.max_model_len,
                                   _rope_theta(config), neox_style)
        self.attn = Attention(self.num_heads, self.head_dim, self.head_dim ** -0.5, self.num_kv_heads,
                              sliding_window=sliding_window, prefix=f"{prefix}.attn",
                              cache_dtype=vllm_config.cache_config.cache_dtype)

    def forward(self, positions, x):
        return getattr(self, self._out)(self.attn(getattr(self, self._qkv)(x), positions, self.rotary_emb))


class _Base(nn.Module):
    supports_pp = False

    def compute_logits(self, hidden):
        return self.logits_processor(self.lm_head, hidden)


# ---- GPT-NeoX ----------------------------------------------------------------------------------------
class GPTNeoXLayer(nn.Module):
    def __init__(self, config, vllm_config, prefix: str):
        super().__init__()
        d = config.hidden_size
        self.parallel = bool(config.use_parallel_residual)
        self.input_layernorm = LayerNorm(d, config.layer_norm_eps)
        self.post_attention_layernorm = LayerNorm(d, config.layer_norm_eps)
        head_dim = d // config.num_attention_heads
        pct = getattr(config, "rotary_pct", None)
        if pct is None:
            pct = (getattr(config, "rope_parameters", None) or {}).get("partial_rotary_factor", 1.0)
        bias = bool(getattr(config, "attention_bias", True))
        self.attention = _RotaryLNAttention(config, vllm_config, d, config.num_attention_heads,
                                            config.num_attention_heads, int(head_dim * pct), True, bias, bias,
                                            f"{prefix}.attention", "query_key_value", "dense")
        self.dense_h_to_4h = ColumnParallelLinear(d, config.intermediate_size, bias=True,
                                                  prefix=f"{prefix}.mlp.dense_h_to_4h")
        self.dense_4h_to_h = RowParallelLinear(config.intermediate_size, d, bias=True,
                                               prefix=f"{prefix}.mlp.dense_4h_to_h")
        self.act = get_act_fn(config.hidden_act)

    def _mlp(self, x):
        return self.dense_4h_to_h(self.act(self.dense_h_to_4h(x)))

    def forward(self, positions, x):
        a = self.attention(positions, self.input_layernorm(x))
        if self.parallel:
            return x + a + self._mlp(self.post_attention_layernorm(x))
        x = x + a
        return x + self._mlp(self.post_attention_layernorm(x))


class GPTNeoXForCausalLM(_Base):
    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        self.embed_in = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([GPTNeoXLayer(config, vllm_config, f"gpt_neox.layers.{i}")
                                     for i in range(config.num_hidden_layers)])
        self.final_layer_norm = LayerNorm(config.hidden_size, config.layer_norm_eps)
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        self.tie = bool(getattr(config, "tie_word_embeddings", False))
        if self.tie:
            self.lm_head.tie_weights(self.embed_in)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    def get_input_embeddings(self, ids):
        return self.embed_in(ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        x = inputs_embeds if inputs_embeds is not None else self.embed_in(input_ids)
        for layer in self.layers:
            x = layer(positions, x)
        return self.final_layer_norm(x)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]) -> set[str]:
        H = self.config.num_attention_heads
        D = self.config.hidden_size // H

        def fix(items):
            for name, t in items:
                if "attention.bias" in name or "masked_bias" in name or "rotary_emb" in name:
                    continue
                name = name.replace("gpt_neox.", "").replace("embed_out.", "lm_head.") \
                           .replace(".mlp.dense_", ".dense_")
                if "query_key_value" in name:
                    # checkpoint: per head [q | k | v]; fused layer: [all q | all k | all v]
                    t = t.view(H, 3, D, *t.shape[1:]).transpose(0, 1).reshape(t.shape)
                yield name, t

        return load_weights_with_mapping(self, fix(weights), [], tie_lm_head=self.tie)


# ---- GPT-J -------------------------------------------------------------------------------------------
class GPTJBlock(nn.Module):
    def __init__(self, config, vllm_config, prefix: str):
        super().__init__()
        d = config.n_embd
        inner = getattr(config, "n_inner", None) or 4 * d
        self.ln_1 = LayerNorm(d, config.layer_norm_epsilon)
        self.attn = _RotaryLNAttention(config, vllm_config, d, config.n_head, config.n_head,
                                       config.rotary_dim or d // config.n_head, False, False, False,
                                       f"{prefix}.attn", "qkv_proj", "out_proj")
        self.fc_in = ColumnParallelLinear(d, inner, bias=True, prefix=f"{prefix}.mlp.fc_in")
        self.fc_out = RowParallelLinear(inner, d, bias=True, prefix=f"{prefix}.mlp.fc_out")
        self.act = get_act_fn(config.activation_function)

    def forward(self, positions, x):
        h = self.ln_1(x)
        return x + self.attn(positions, h) + self.fc_out(self.act(self.fc_in(h)))


class GPTJForCausalLM(_Base):
    stacked_params_mapping = [(".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"),
                              (".qkv_proj", ".v_proj", "v")]

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        self.wte = VocabParallelEmbedding(config.vocab_size, config.n_embd)
        self.h = nn.ModuleList([GPTJBlock(config, vllm_config, f"transformer.h.{i}") for i in range(config.n_layer)])
        self.ln_f = LayerNorm(config.n_embd, config.layer_norm_epsilon)
        self.lm_head = ParallelLMHead(config.vocab_size, config.n_embd, bias=True)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    def get_input_embeddings(self, ids):
        return self.wte(ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        x = inputs_embeds if inputs_embeds is not None else self.wte(input_ids)
        for blk in self.h:
            x = blk(positions, x)
        return self.ln_f(x)

    def compute_logits(self, hidden):
        return self.logits_processor(self.lm_head, hidden, getattr(self.lm_head, "bias", None))

    def load_weights(self, weights):
        def fix(items):
            for name, t in items:
                if ".attn.bias" in name or "masked_bias" in name:
                    continue
                yield name.removeprefix("transformer.").replace(".mlp.fc_", ".fc_"), t
        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping)


# ---- StarCoder2 --------------------------------------------------------------------------------------
class Starcoder2Layer(nn.Module):
    def __init__(self, config, vllm_config, prefix: str):
        super().__init__()
        d = config.hidden_size
        self.input_layernorm = LayerNorm(d, config.norm_epsilon)
        self.post_attention_layernorm = LayerNorm(d, config.norm_epsilon)
        self.self_attn = _RotaryLNAttention(config, vllm_config, d, config.num_attention_heads,
                                            config.num_key_value_heads, d // config.num_attention_heads, True,
                                            config.use_bias, config.use_bias, f"{prefix}.self_attn", "qkv_proj",
                                            "o_proj", sliding_window=getattr(config, "sliding_window", None))
        self.c_fc = ColumnParallelLinear(d, config.intermediate_size, bias=config.use_bias, prefix=f"{prefix}.mlp.c_fc")
        self.c_proj = RowParallelLinear(config.intermediate_size, d, bias=config.use_bias,
                                        prefix=f"{prefix}.mlp.c_proj")
        self.act = get_act_fn(config.hidden_act)

    def forward(self, positions, x):
        x = x + self.self_attn(positions, self.input_layernorm(x))
        return x + self.c_proj(self.act(self.c_fc(self.post_attention_layernorm(x))))


class Starcoder2ForCausalLM(_Base):
    stacked_params_mapping = GPTJForCausalLM.stacked_params_mapping

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([Starcoder2Layer(config, vllm_config, f"model.layers.{i}")
                                     for i in range(config.num_hidden_layers)])
        self.norm = LayerNorm(config.hidden_size, config.norm_epsilon)
        self.tie = bool(getattr(config, "tie_word_embeddings", True))
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if self.tie:
            self.lm_head.tie_weights(self.embed_tokens)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    def get_input_embeddings(self, ids):
        return self.embed_tokens(ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        x = inputs_embeds if inputs_embeds is not None else self.embed_tokens(input_ids)
        for layer in self.layers:
            x = layer(positions, x)
        return self.norm(x)

    def load_weights(self, weights):
        def fix(items):
            for name, t in items:
                yield name.removeprefix("model.").replace(".mlp.c_", ".c_"), t
        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping, tie_lm_head=self.tie)


# ---- GPT-BigCode -------------------------------------------------------------------------------------
class GPTBigCodeBlock(nn.Module):
    def __init__(self, config, vllm_config, prefix: str):
        super().__init__()
        d = config.n_embd
        inner = getattr(config, "n_inner", None) or 4 * d
        tp = get_tensor_model_parallel_world_size()
        kv_heads = 1 if config.multi_query else config.n_head
        self.ln_1 = LayerNorm(d, config.layer_norm_epsilon)
        self.c_attn = QKVParallelLinear(d, d // config.n_head, config.n_head, kv_heads, bias=True,
                                        prefix=f"{prefix}.attn.c_attn")
        self.c_proj = RowParallelLinear(d, d, bias=True, prefix=f"{prefix}.attn.c_proj")
        self.attn = Attention(config.n_head // tp, d // config.n_head, (d // config.n_head) ** -0.5,
                              max(1, kv_heads // tp), prefix=f"{prefix}.attn",
                              cache_dtype=vllm_config.cache_config.cache_dtype)
        self.ln_2 = LayerNorm(d, config.layer_norm_epsilon)
        self.c_fc = ColumnParallelLinear(d, inner, bias=True, prefix=f"{prefix}.mlp.c_fc")
        self.mlp_proj = RowParallelLinear(inner, d, bias=True, prefix=f"{prefix}.mlp.c_proj")
        self.act = get_act_fn(config.activation_function)

    def forward(self, x):
        x = x + self.c_proj(self.attn(self.c_attn(self.ln_1(x))))
        return x + self.mlp_proj(self.act(self.c_fc(self.ln_2(x))))


class GPTBigCodeForCausalLM(_Base):
    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        self.wte = VocabParallelEmbedding(config.vocab_size, config.n_embd)
        self.wpe = nn.Embedding(config.n_positions, config.n_embd)
        self.wpe.weight.requires_grad_(False)
        self.h = nn.ModuleList([GPTBigCodeBlock(config, vllm_config, f"transformer.h.{i}")
                                for i in range(config.n_layer)])
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

    def load_weights(self, weights):
        def fix(items):
            for name, t in items:
                if name.startswith("lm_head") or ".attn.bias" in name or "masked_bias" in name:
                    continue
                name = name.removeprefix("transformer.")
                yield name.replace(".attn.c_attn", ".c_attn").replace(".attn.c_proj", ".c_proj") \
                          .replace(".mlp.c_fc", ".c_fc").replace(".mlp.c_proj", ".mlp_proj"), t
        return load_weights_with_mapping(self, fix(weights), [])
