"""Norm, activation, embedding and LM-head modules.

Reference: vllm/model_executor/layers/layernorm.py (RMSNorm :88, GemmaRMSNorm :223),
activation.py, vocab_parallel_embedding.py (:162 embedding, :436 LM head; vocab
padded to a multiple of 64 per rank :22,53-57), logits_processor.py:88-119.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..parallel.state import (get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                              get_tp_group, tensor_model_parallel_all_reduce)
from .linear import UnquantizedLinearMethod, set_weight_attrs


class RMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6, dtype: Optional[torch.dtype] = None,
                 weight_bias: float = 0.0):
        super().__init__()
        self.eps = eps
        self.weight_bias = weight_bias   # 1.0 for Gemma-style (x * (1 + w))
        init = torch.zeros if weight_bias else torch.ones
        self.weight = nn.Parameter(init(hidden_size, dtype=dtype or torch.get_default_dtype()), requires_grad=False)

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None):
        """With ``residual``: residual += x (in place), returns (norm(residual), residual)."""
        if residual is None:
            return ops.rms_norm(x, self.weight, self.eps, None, self.weight_bias)
        return ops.rms_norm(x, self.weight, self.eps, residual, self.weight_bias), residual


    def forward_reduce(self, x_partial: torch.Tensor, residual: torch.Tensor):
        """``x_partial`` holds this tensor-parallel rank's partial sums of a row-parallel layer:
        residual += all_reduce(x_partial); returns (norm(residual), residual).  One peer-memory kernel
        for decode-sized batches (parallel/csrc/custom_allreduce.hip far_kernel)."""
        from ..parallel.state import get_tp_group
        return get_tp_group().all_reduce_rms_norm(x_partial, residual, self.weight, self.eps, self.weight_bias)


class GemmaRMSNorm(RMSNorm):
    def __init__(self, hidden_size: int, eps: float = 1e-6, dtype=None):
        super().__init__(hidden_size, eps, dtype, weight_bias=1.0)


class LayerNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-5, bias: bool = True, dtype=None):
        super().__init__()
        dtype = dtype or torch.get_default_dtype()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(hidden_size, dtype=dtype), requires_grad=False) if bias else None

    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias, self.eps)


class SiluAndMul(nn.Module):
    def forward(self, x):
        return ops.act_and_mul(x, "silu")


class GeluAndMul(nn.Module):
    def __init__(self, approximate: str = "none"):
        super().__init__()
        self.kind = "gelu_tanh" if approximate == "tanh" else "gelu"

    def forward(self, x):
        return ops.act_and_mul(x, self.kind)


class MulAndSilu(nn.Module):
    def forward(self, x):
        return ops.act_and_mul(x, "silu", act_first=False)


class FatreluAndMul(nn.Module):
    def __init__(self, threshold: float = 0.0):
        super().__init__()
        self.threshold = threshold

    def forward(self, x):
        return ops.act_and_mul(x, "fatrelu", threshold=self.threshold)


class Activation(nn.Module):
    def __init__(self, kind: str):
        super().__init__()
        self.kind = kind

    def forward(self, x):
        return ops.activation(x, self.kind)


_ACT_ALIASES = {"gelu_pytorch_tanh": "gelu_tanh", "quick_gelu": "gelu_quick"}


def get_act_fn(name: str) -> nn.Module:
    name = name.lower()
    name = _ACT_ALIASES.get(name, name)
    if name not in ("gelu", "gelu_tanh", "gelu_new", "gelu_fast", "gelu_quick", "relu", "relu2", "silu"):
        raise ValueError(f"unsupported activation {name!r}")
    return Activation(name)


def get_act_and_mul_fn(name: str) -> nn.Module:
    name = name.lower()
    if name == "silu":
        return SiluAndMul()
    if name == "gelu":
        return GeluAndMul()
    if name in ("gelu_tanh", "gelu_pytorch_tanh"):
        return GeluAndMul("tanh")
    raise ValueError(f"unsupported gated activation {name!r}")


# ---- vocabulary-parallel embedding / LM head ------------------------------------
VOCAB_PAD = 64


def pad_vocab_size(v: int, to: int = VOCAB_PAD) -> int:
    return (v + to - 1) // to * to


class VocabParallelEmbedding(nn.Module):
    """Rows [rank * per_rank, (rank+1) * per_rank) of the (padded) vocabulary."""

    def __init__(self, num_embeddings: int, embedding_dim: int, params_dtype=None,
                 org_num_embeddings: Optional[int] = None, padding_size: int = VOCAB_PAD,
                 quant_config=None, prefix: str = ""):
        super().__init__()
        self.tp_size = get_tensor_model_parallel_world_size()
        self.tp_rank = get_tensor_model_parallel_rank()
        self.num_embeddings = num_embeddings
        self.org_vocab_size = org_num_embeddings or num_embeddings
        self.embedding_dim = embedding_dim
        self.num_embeddings_padded = pad_vocab_size(num_embeddings, padding_size * self.tp_size)
        self.per_rank = self.num_embeddings_padded // self.tp_size
        self.vocab_start = self.tp_rank * self.per_rank
        self.vocab_end = self.vocab_start + self.per_rank
        dtype = params_dtype or torch.get_default_dtype()
        self.weight = nn.Parameter(torch.zeros(self.per_rank, embedding_dim, dtype=dtype), requires_grad=False)
        set_weight_attrs(self.weight, {"weight_loader": self.weight_loader, "output_dim": 0, "input_dim": 1})
        self.quant_method = UnquantizedLinearMethod()

    def weight_loader(self, param, loaded, *_):
        if loaded.shape[0] < self.num_embeddings and loaded.shape[0] != self.org_vocab_size:
            raise ValueError(f"embedding rows {loaded.shape[0]} do not match vocab {self.org_vocab_size}")
        lo, hi = self.vocab_start, min(self.vocab_end, loaded.shape[0])
        param.data.zero_()
        if hi > lo:
            param.data[: hi - lo].copy_(loaded[lo:hi])

    def forward(self, ids: torch.Tensor) -> torch.Tensor:
        if self.tp_size == 1:
            return F.embedding(ids, self.weight)
        mask = (ids < self.vocab_start) | (ids >= self.vocab_end)
        local = (ids - self.vocab_start).masked_fill(mask, 0)
        out = F.embedding(local, self.weight)
        out.masked_fill_(mask.unsqueeze(-1), 0)
        return tensor_model_parallel_all_reduce(out)


class ParallelLMHead(VocabParallelEmbedding):
    def __init__(self, num_embeddings, embedding_dim, bias=False, params_dtype=None, org_num_embeddings=None,
                 padding_size=VOCAB_PAD, quant_config=None, prefix=""):
        super().__init__(num_embeddings, embedding_dim, params_dtype, org_num_embeddings, padding_size,
                         quant_config, prefix)
        if bias:
            self.bias = nn.Parameter(torch.zeros(self.per_rank, dtype=self.weight.dtype), requires_grad=False)
            set_weight_attrs(self.bias, {"weight_loader": self.weight_loader, "output_dim": 0})
        else:
            self.register_parameter("bias", None)

    def tie_weights(self, embed: VocabParallelEmbedding) -> "ParallelLMHead":
        self.weight = embed.weight
        return self

    def forward(self, ids):
        raise RuntimeError("the LM head is applied through LogitsProcessor")


class LogitsProcessor(nn.Module):
    """hidden states -> full-vocabulary logits (TP: all-gather of the vocab shards)."""

    def __init__(self, vocab_size: int, org_vocab_size: Optional[int] = None, scale: float = 1.0,
                 logits_as_input: bool = False, soft_cap: Optional[float] = None):
        super().__init__()
        self.vocab_size = vocab_size
        self.org_vocab_size = org_vocab_size or vocab_size
        self.scale = scale
        self.soft_cap = soft_cap
        self.logits_as_input = logits_as_input

    def forward(self, lm_head: VocabParallelEmbedding, hidden: torch.Tensor,
                embedding_bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.logits_as_input:
            logits = hidden
        else:
            logits = lm_head.quant_method.apply(lm_head, hidden, embedding_bias)
            if get_tensor_model_parallel_world_size() > 1:
                logits = get_tp_group().all_gather(logits, dim=-1)
            logits = logits[..., : self.org_vocab_size]
        if self.soft_cap is not None:
            logits = torch.tanh(logits / self.soft_cap) * self.soft_cap
        if self.scale != 1.0:
            logits = logits * self.scale
        return logits
