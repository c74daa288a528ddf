"""Zamba2: a Mamba-2 backbone with a few SHARED transformer blocks.  Every ``hybrid`` layer first runs one of the
``num_mem_blocks`` shared blocks (taken in turn) on the concatenation [current stream | token embeddings] (so the
block works at twice the model width), projects the result with a linear layer of its own and adds it to the input
of that layer's Mamba-2 mixer.  A shared block's weights serve several depths; what differs per depth is a low-rank
adapter on q, k, v and on the MLP's gate/up projection, and the KV cache (one paged cache per hybrid layer).
Attention logits are scaled by 1 / sqrt(head_dim / 2).

The Mamba-2 states live in the per-request state slots, the attention layers in the paged cache.
Reference: vllm/model_executor/models/zamba2.py (Zamba2LoRA :49-95, Zamba2Attention :98-271, Zamba2MLP :274-363,
Zamba2AttentionDecoderLayer :366-466, Zamba2HybridLayer :563-646, Zamba2Model :649-827).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Iterable

import torch
import torch.nn as nn

from ..config import VllmConfig
from ..layers.attention import Attention
from ..layers.basic import LogitsProcessor, ParallelLMHead, RMSNorm, VocabParallelEmbedding, get_act_fn
from ..layers.linear import ColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from ..layers.rotary import get_rope
from .mamba import Mamba2Mixer
from .utils import load_weights_with_mapping


class _Adapter(nn.Sequential):
    """x -> B(A(x)), rank ``adapter_rank``; checkpoint names ``<list>.<depth>.0.weight`` / ``.1.weight``."""

    def __init__(self, d_in: int, rank: int, d_out: int):
        super().__init__(nn.Linear(d_in, rank, bias=False), nn.Linear(rank, d_out, bias=False))


def _adapter_list(depths: list[int], total: int, make):
    """ModuleList indexed by hybrid depth; depths this block does not serve hold an empty module."""
    return nn.ModuleList(make() if k in depths else nn.Identity() for k in range(total))


class Zamba2SharedAttention(nn.Module):
    def __init__(self, config, vllm_config: VllmConfig, depths: list[int], total: int, prefix: str):
        super().__init__()
        quant_config = getattr(vllm_config, "quant_config", None)
        d2 = int(config.attention_hidden_size)
        self.num_heads, self.num_kv_heads = int(config.num_attention_heads), int(config.num_key_value_heads)
        self.head_dim = int(config.attention_head_dim)
        self.q_size, self.kv_size = self.num_heads * self.head_dim, self.num_kv_heads * self.head_dim
        self.qkv_proj = QKVParallelLinear(d2, self.head_dim, self.num_heads, self.num_kv_heads, bias=False,
                                          quant_config=quant_config, prefix=f"{prefix}.qkv_proj")
        self.o_proj = RowParallelLinear(self.q_size, int(config.hidden_size), bias=False, quant_config=quant_config,
                                        prefix=f"{prefix}.o_proj")
        self.use_adapter = bool(getattr(config, "use_shared_attention_adapter", False))
        if self.use_adapter:
            r = int(config.adapter_rank)
            self.linear_q_adapter_list = _adapter_list(depths, total, lambda: _Adapter(d2, r, self.q_size))
            self.linear_k_adapter_list = _adapter_list(depths, total, lambda: _Adapter(d2, r, self.kv_size))
            self.linear_v_adapter_list = _adapter_list(depths, total, lambda: _Adapter(d2, r, self.kv_size))
        rp = getattr(config, "rope_parameters", None) or {}
        theta = float(getattr(config, "rope_theta", None) or rp.get("rope_theta") or 10000.0)
        self.rotary_emb = get_rope(self.head_dim, self.head_dim, vllm_config.model_config.max_model_len, theta, True,
                                   None, 1.0) if getattr(config, "use_mem_rope", False) else None
        # one paged cache per depth this block serves
        self.attn = nn.ModuleDict({
            str(k): Attention(self.num_heads, self.head_dim, (self.head_dim / 2) ** -0.5, self.num_kv_heads,
                              prefix=f"{prefix}.attn.{k}", cache_dtype=vllm_config.cache_config.cache_dtype)
            for k in depths})

    def forward(self, positions, x, depth: int):
        qkv = self.qkv_proj(x)
        if self.use_adapter:
            qkv = qkv + torch.cat([self.linear_q_adapter_list[depth](x), self.linear_k_adapter_list[depth](x),
                                   self.linear_v_adapter_list[depth](x)], -1)
        return self.o_proj(self.attn[str(depth)](qkv, positions, self.rotary_emb))


class Zamba2SharedMLP(nn.Module):
    def __init__(self, config, vllm_config: VllmConfig, depths: list[int], total: int, prefix: str):
        super().__init__()
        quant_config = getattr(vllm_config, "quant_config", None)
        d, inter = int(config.hidden_size), int(config.intermediate_size)
        bias = bool(getattr(config, "add_bias_linear", False))
        self.gate_up_proj = ColumnParallelLinear(d, 2 * inter, bias=bias, quant_config=quant_config,
                                                 prefix=f"{prefix}.gate_up_proj")
        self.down_proj = RowParallelLinear(inter, d, bias=bias, quant_config=quant_config, prefix=f"{prefix}.down_proj")
        self.gate_up_proj_adapter_list = _adapter_list(depths, total,
                                                       lambda: _Adapter(d, int(config.adapter_rank), 2 * inter))
        self.act = get_act_fn(getattr(config, "hidden_act", "gelu"))

    def forward(self, x, depth: int):
        gu = self.gate_up_proj(x) + self.gate_up_proj_adapter_list[depth](x)
        gate, up = gu.chunk(2, -1)
        return self.down_proj(self.act(gate) * up)


class Zamba2SharedBlock(nn.Module):
    def __init__(self, config, vllm_config: VllmConfig, depths: list[int], total: int, prefix: str):
        super().__init__()
        self.self_attn = Zamba2SharedAttention(config, vllm_config, depths, total, f"{prefix}.self_attn")
        self.feed_forward = Zamba2SharedMLP(config, vllm_config, depths, total, f"{prefix}.feed_forward")
        self.input_layernorm = RMSNorm(int(config.attention_hidden_size), config.rms_norm_eps)
        self.pre_ff_layernorm = RMSNorm(int(config.hidden_size), config.rms_norm_eps)

    def forward(self, positions, hidden, embeds, depth: int):
        x = self.input_layernorm(torch.cat([hidden, embeds], -1))
        x = self.self_attn(positions, x, depth)
        return self.feed_forward(self.pre_ff_layernorm(x), depth)


class Zamba2MambaLayer(nn.Module):
    def __init__(self, config, prefix: str):
        super().__init__()
        d = int(config.hidden_size)
        heads = int(config.n_mamba_heads)
        inner = int(config.mamba_expand * d)
        shim = SimpleNamespace(hidden_size=d, num_heads=heads, head_dim=inner // heads, intermediate_size=inner,
                               state_size=config.mamba_d_state, conv_kernel=config.mamba_d_conv,
                               n_groups=config.mamba_ngroups, use_bias=bool(getattr(config, "add_bias_linear", False)),
                               use_conv_bias=bool(getattr(config, "use_conv_bias", True)), layer_norm_epsilon=1e-5,
                               time_step_limit=(float(getattr(config, "time_step_min", 0.0)), float("inf")))
        self.mamba = Mamba2Mixer(shim, f"{prefix}.mamba")
        self.input_layernorm = RMSNorm(d, config.rms_norm_eps)

    def forward(self, hidden, from_block=None):
        x = hidden if from_block is None else hidden + from_block
        return hidden + self.mamba(self.input_layernorm(x))


class Zamba2HybridLayer(nn.Module):
    def __init__(self, config, block: Zamba2SharedBlock, depth: int, prefix: str):
        super().__init__()
        self.depth = depth
        self._block = [block]                      # not a submodule: the block is owned by the model
        self.linear = nn.Linear(int(config.hidden_size), int(config.hidden_size), bias=False)
        self.mamba_decoder = Zamba2MambaLayer(config, f"{prefix}.mamba_decoder")

    def forward(self, positions, hidden, embeds):
        t = self.linear(self._block[0](positions, hidden, embeds, self.depth))
        return self.mamba_decoder(hidden, t)


class Zamba2ForCausalLM(nn.Module):
    supports_pp = False
    supports_lora = False
    stacked_params_mapping = [(".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"), (".qkv_proj", ".v_proj", "v")]

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        kinds = list(config.layers_block_type)
        hybrid_ids = [i for i, k in enumerate(kinds) if k == "hybrid"]
        nb, total = int(config.num_mem_blocks), len(hybrid_ids)
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        # shared block b serves hybrid depths b, b + nb, b + 2 nb ...
        self.blocks = nn.ModuleList(
            Zamba2SharedBlock(config, vllm_config, list(range(b, total, nb)), total, f"model.blocks.{b}")
            for b in range(min(nb, total)))
        self.block_home = {b: hybrid_ids[b] for b in range(min(nb, total))}     # layer whose checkpoint entry holds it
        layers = []
        for i, kind in enumerate(kinds):
            if kind == "hybrid":
                k = hybrid_ids.index(i)
                layers.append(Zamba2HybridLayer(config, self.blocks[k % nb], k, f"model.layers.{i}"))
            else:
                layers.append(Zamba2MambaLayer(config, f"model.layers.{i}"))
        self.layers = nn.ModuleList(layers)
        self.final_layernorm = RMSNorm(config.hidden_size, config.rms_norm_eps)
        self.tie = bool(getattr(config, "tie_word_embeddings", True))
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if self.tie:
            self.lm_head.tie_weights(self.embed_tokens)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    def get_input_embeddings(self, ids):
        return self.embed_tokens(ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        embeds = inputs_embeds if inputs_embeds is not None else self.embed_tokens(input_ids)
        hidden = embeds
        for layer in self.layers:
            if isinstance(layer, Zamba2HybridLayer):
                hidden = layer(positions, hidden, embeds)
            else:
                hidden = layer(hidden)
        return self.final_layernorm(hidden)

    def compute_logits(self, hidden):
        return self.logits_processor(self.lm_head, hidden)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        names = {".mamba.conv1d.weight": ".mamba.conv_weight", ".mamba.conv1d.bias": ".mamba.conv_bias",
                 ".mamba.A_log": ".mamba.A", ".mamba.norm.weight": ".mamba.norm_weight"}
        home = {f"layers.{layer}.shared_transformer.": f"blocks.{b}." for b, layer in self.block_home.items()}

        def fix(items):
            for name, t in items:
                name = name.removeprefix("model.")
                for a, b in names.items():
                    if name.endswith(a):
                        name = name[: -len(a)] + b
                if ".shared_transformer." in name:
                    head = name[: name.index(".shared_transformer.") + len(".shared_transformer.")]
                    if head not in home:
                        continue                   # a tied copy of a block stored under its first layer
                    name = home[head] + name[len(head):]
                yield name, t

        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping, tie_lm_head=self.tie)
