"""Sequence pooling for embedding / classification / scoring models.
Reference: vllm/model_executor/layers/pooler.py (PoolingType, normalize / softmax heads)."""
from __future__ import annotations

import enum
from typing import Optional

import torch
import torch.nn as nn


class PoolingType(enum.Enum):
    LAST = "last"
    MEAN = "mean"
    CLS = "cls"
    ALL = "all"


class Pooler(nn.Module):
    def __init__(self, pooling_type: PoolingType = PoolingType.LAST, normalize: bool = True, softmax: bool = False):
        super().__init__()
        self.pooling_type, self.normalize, self.softmax = pooling_type, normalize, softmax

    @classmethod
    def from_config(cls, pooler_config, default_type: PoolingType, normalize: bool, softmax: bool) -> "Pooler":
        pt = default_type
        if pooler_config is not None:
            if getattr(pooler_config, "pooling_type", None):
                pt = PoolingType(str(pooler_config.pooling_type).lower())
            if getattr(pooler_config, "normalize", None) is not None:
                normalize = pooler_config.normalize
            if getattr(pooler_config, "softmax", None) is not None:
                softmax = pooler_config.softmax
        return cls(pt, normalize, softmax)

    @property
    def needs_whole_prompt(self) -> bool:
        return self.pooling_type in (PoolingType.MEAN, PoolingType.CLS, PoolingType.ALL)

    def extract(self, hidden: torch.Tensor) -> torch.Tensor:
        """hidden: [n_tokens, H] of one sequence."""
        if self.pooling_type == PoolingType.LAST:
            return hidden[-1]
        if self.pooling_type == PoolingType.CLS:
            return hidden[0]
        if self.pooling_type == PoolingType.MEAN:
            return hidden.float().mean(dim=0).to(hidden.dtype)
        return hidden

    def head(self, x: torch.Tensor, params=None) -> torch.Tensor:
        normalize, softmax = self.normalize, self.softmax
        if params is not None:
            normalize = params.normalize if params.normalize is not None else normalize
            softmax = params.softmax if params.softmax is not None else softmax
            if params.dimensions:
                x = x[..., : params.dimensions]
        x = x.float()
        if normalize:
            x = torch.nn.functional.normalize(x, p=2, dim=-1)
        if softmax:
            x = torch.softmax(x, dim=-1) if x.shape[-1] > 1 else torch.sigmoid(x)
        return x

    def forward(self, hidden: torch.Tensor, params=None) -> torch.Tensor:
        return self.head(self.extract(hidden), params)
