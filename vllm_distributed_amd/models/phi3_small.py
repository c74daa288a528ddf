"""Phi-3-small: LayerNorm blocks, a gated MLP whose fused projection INTERLEAVES the GELU half and the linear half
(``x1 * sigmoid(1.702 x1) * (x2 + 1)`` with both halves clamped), a fused q / k / v projection laid out per KV
group, muP scales (embedding multiplier, attention scale ``mup_attn_multiplier / head_dim``, logits divided by the
width multiplier), and block-sparse causal attention on every layer that is not a multiple of
``dense_attention_every_n_layers``: a query block sees the last ``blocksparse_num_local_blocks`` key blocks plus, per
head, every ``blocksparse_vert_stride``-th block.

Sequences that fit the local span see every causal key, so they run the ordinary kernels; longer ones take the
masked PyTorch form over the paged cache, which is exact but slow and cannot be graph-captured (the model runs
eager).  No transformers class exists for this family: parity unpinned; tested against the network written out
densely in fp32.
Reference: vllm/model_executor/models/phi3_small.py (gegelu :67-83, Phi3SmallSelfAttention :115-270,
Phi3SmallForCausalLM :372-465), vllm/attention/ops/blocksparse_attention/utils.py (get_sparse_attn_mask :163-245).
"""
from __future__ import annotations

from typing import Iterable

import torch
import torch.nn as nn

from ..layers.linear import ColumnParallelLinear, RowParallelLinear
from ..parallel.state import get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size
from .llama import LlamaDecoderLayer, LlamaForCausalLM, LlamaModel
from .llama_variants import ResidualLayerNorm
from .utils import load_weights_with_mapping


def gegelu(x: torch.Tensor, limit: float | None) -> torch.Tensor:
    a, b = x[..., ::2], x[..., 1::2]
    if limit is not None:
        a = torch.where(torch.isinf(a), a, a.clamp(max=limit))
        b = torch.where(torch.isinf(b), b, b.clamp(min=-limit, max=limit))
    return a * torch.sigmoid(1.702 * a) * (b + 1.0)


class Phi3SmallMLP(nn.Module):
    def __init__(self, config, quant_config, prefix: str):
        super().__init__()
        if getattr(config, "hidden_act", "gegelu") != "gegelu":
            raise ValueError("Phi-3-small uses the gegelu activation")
        d, inter = int(config.hidden_size), int(config.intermediate_size)
        self.limit = getattr(config, "gegelu_limit", None)
        self.up_proj = ColumnParallelLinear(d, 2 * inter, bias=True, quant_config=quant_config, prefix=f"{prefix}.up_proj")
        self.down_proj = RowParallelLinear(inter, d, bias=True, quant_config=quant_config, prefix=f"{prefix}.down_proj")

    def forward(self, x):
        return self.down_proj(gegelu(self.up_proj(x), self.limit))


class Phi3SmallDecoderLayer(LlamaDecoderLayer):
    supports_sequence_parallel = False

    @classmethod
    def make_norm(cls, config):
        return ResidualLayerNorm(config.hidden_size, getattr(config, "layer_norm_epsilon", 1e-5))

    def _rope_of(self, config, layer_idx: int):
        scaling = getattr(config, "rope_scaling", None)
        if scaling:
            scaling = dict(scaling)
            scaling.setdefault("factor", float(getattr(config, "rope_position_scale", 1.0)))
        elif float(getattr(config, "rope_position_scale", 1.0)) != 1.0:
            scaling = {"rope_type": "linear", "factor": float(config.rope_position_scale)}
        return float(getattr(config, "rope_embedding_base", 10000.0)), scaling

    def _attention_extras(self, config, layer_idx: int) -> dict:
        hd = config.hidden_size // config.num_attention_heads
        scale = float(config.mup_attn_multiplier) / hd if getattr(config, "mup_use_scaling", True) else hd ** -0.5
        every = int(getattr(config, "dense_attention_every_n_layers", 0) or 0)
        sparse = None
        if not (every and (layer_idx + 1) % every == 0):
            tp, r = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
            sparse = dict(block_size=int(config.blocksparse_block_size),
                          local_blocks=int(config.blocksparse_num_local_blocks),
                          vert_stride=int(config.blocksparse_vert_stride),
                          homo_head=bool(getattr(config, "blocksparse_homo_head_pattern", False)),
                          total_heads=int(config.num_attention_heads),
                          first_head=r * (int(config.num_attention_heads) // tp))
        return {"bias": True, "o_bias": True, "scale": scale, "blocksparse": sparse}

    def _make_mlp(self, config, vllm_config, quant_config, prefix):
        return Phi3SmallMLP(config, quant_config, prefix)


class Phi3SmallModel(LlamaModel):
    layer_cls = Phi3SmallDecoderLayer
    embed_scale_of = staticmethod(lambda config: float(getattr(config, "mup_embedding_multiplier", 0.0) or 1.0))


class Phi3SmallForCausalLM(LlamaForCausalLM):
    model_cls = Phi3SmallModel
    requires_eager = True
    supports_lora = False

    @staticmethod
    def _logit_scale(config) -> float:
        return 1.0 / float(getattr(config, "mup_width_multiplier", 1.0) or 1.0)

    def __init__(self, vllm_config, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        dummy = getattr(self.config, "dummy_token_indices", None)
        self.register_buffer("dummy_token_indices", torch.tensor(list(dummy), dtype=torch.long) if dummy else None,
                             persistent=False)

    def compute_logits(self, hidden_states):
        logits = super().compute_logits(hidden_states)
        if logits is not None and self.dummy_token_indices is not None:
            logits.index_fill_(-1, self.dummy_token_indices.to(logits.device), float("-inf"))
        return logits

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        c = self.config
        heads, kv = int(c.num_attention_heads), int(c.num_key_value_heads)
        hd, per = c.hidden_size // heads, heads // kv

        def fix(items):
            for name, t in items:
                name = name.replace("model.final_layernorm.", "model.norm.").replace(".self_attn.dense.", ".self_attn.o_proj.")
                if ".self_attn.query_key_value." in name:
                    # rows are grouped per KV head: its ``per`` query heads, then its key, then its value
                    g = t.view(kv, per + 2, hd, *t.shape[1:])
                    for proj, part in (("q_proj", g[:, :per]), ("k_proj", g[:, per]), ("v_proj", g[:, per + 1])):
                        yield name.replace(".query_key_value.", f".{proj}."), part.reshape(-1, *t.shape[1:])
                else:
                    yield name, t

        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping[:3], tie_lm_head=self.tie)
