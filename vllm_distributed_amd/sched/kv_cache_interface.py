"""What a layer needs from the KV cache, and the resulting memory plan.

Reference: vllm/v1/kv_cache_interface.py:19-223.  The device layout differs
from the reference on purpose (see ops/csrc/attention.hip): per layer one K
tensor ``[num_blocks, kv_heads, block_size, head_size]`` and one V tensor
``[num_blocks, kv_heads, head_size, block_size]`` so that both feed gfx950
MFMA operands with 16-byte per-lane loads and no LDS transpose.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import torch

from ..parallel.utils import cdiv


def _dtype_size(dt: torch.dtype) -> int:
    return torch.empty((), dtype=dt).element_size()


@dataclass(frozen=True)
class KVCacheSpec:
    block_size: int

    @property
    def type_id(self) -> str:
        raise NotImplementedError

    @property
    def page_size_bytes(self) -> int:
        raise NotImplementedError

    def max_memory_usage_bytes(self, max_model_len: int, max_num_batched_tokens: int) -> int:
        raise NotImplementedError


@dataclass(frozen=True)
class AttentionSpec(KVCacheSpec):
    num_kv_heads: int
    head_size: int
    dtype: torch.dtype
    use_mla: bool = False

    @property
    def page_size_bytes(self) -> int:
        coef = 1 if self.use_mla else 2
        return coef * self.block_size * self.num_kv_heads * self.head_size * _dtype_size(self.dtype)


@dataclass(frozen=True)
class FullAttentionSpec(AttentionSpec):
    sliding_window: Optional[int] = None   # kept so hybrid models can be served with one group

    @property
    def type_id(self) -> str:
        return f"full_{self.block_size}_{self.page_size_bytes}"

    def max_memory_usage_bytes(self, max_model_len: int, max_num_batched_tokens: int) -> int:
        return cdiv(max_model_len, self.block_size) * self.page_size_bytes


@dataclass(frozen=True)
class MLASpec(FullAttentionSpec):
    """Latent cache of multi-head latent attention: per token one row of ``head_size`` = kv_lora_rank +
    rope_dim elements (what queries dot against) plus a transposed copy of its first ``v_size`` =
    kv_lora_rank elements (what the P.V MFMA reads); num_kv_heads is 1."""
    v_size: int = 0

    @property
    def type_id(self) -> str:
        return f"mla_{self.block_size}_{self.page_size_bytes}"

    @property
    def page_size_bytes(self) -> int:
        return self.block_size * (self.head_size + self.v_size) * _dtype_size(self.dtype)


@dataclass(frozen=True)
class SlidingWindowSpec(AttentionSpec):
    sliding_window: int = 0

    @property
    def type_id(self) -> str:
        return f"swa_{self.sliding_window}_{self.block_size}_{self.page_size_bytes}"

    def max_memory_usage_bytes(self, max_model_len: int, max_num_batched_tokens: int) -> int:
        # a chunk of new tokens plus the window behind it, plus one block of misalignment
        tokens = min(self.sliding_window - 1 + max_num_batched_tokens, max_model_len)
        return (cdiv(tokens, self.block_size) + 1) * self.page_size_bytes


@dataclass(frozen=True)
class MambaSpec(KVCacheSpec):
    """Recurrent state of one request: several tensors (conv window, SSM state) that may differ in dtype."""
    shapes: tuple = ()
    dtype: torch.dtype = torch.float32
    dtypes: tuple = ()          # per shape; empty = ``dtype`` for all

    @property
    def type_id(self) -> str:
        return f"mamba_{self.shapes}_{self.dtypes or self.dtype}"

    def part_bytes(self) -> list[int]:
        """Bytes of each state tensor of one request, rounded up to 16 so parts stay aligned."""
        out = []
        for i, s in enumerate(self.shapes):
            k = 1
            for d in s:
                k *= d
            k *= _dtype_size(self.dtypes[i] if self.dtypes else self.dtype)
            out.append((k + 15) // 16 * 16)
        return out

    @property
    def page_size_bytes(self) -> int:
        return sum(self.part_bytes())

    def max_memory_usage_bytes(self, max_model_len: int, max_num_batched_tokens: int) -> int:
        return self.page_size_bytes   # one state per request, independent of length


@dataclass
class KVCacheTensor:
    """One device allocation shared by the named layers."""
    size: int
    shared_by: list[str]


@dataclass
class KVCacheGroupSpec:
    layer_names: list[str]
    kv_cache_spec: KVCacheSpec
    # layers of one group that keep different numbers of KV heads: their own spec (the group's spec is the largest)
    per_layer: Optional[dict] = None


@dataclass
class KVCacheConfig:
    num_blocks: int
    kv_cache_tensors: list[KVCacheTensor] = field(default_factory=list)
    kv_cache_groups: list[KVCacheGroupSpec] = field(default_factory=list)
