"""The one attention backend: fused RoPE + KV-cache write, then the MFMA paged
attention kernels (decode form for the leading decode sequences of the batch,
tiled form for the prefill sequences).

Reference: vllm/attention/layer.py:27-481 (Attention module, KV-connector
hooks :380-395), vllm/v1/attention/backends/triton_attn.py:189-449 (the ROCm
default backend whose role this plays).  There is no backend selector here.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from ..forward_context import AttentionMetadata, get_forward_context
from ..sched.kv_cache_interface import FullAttentionSpec, KVCacheSpec, SlidingWindowSpec


class Attention(nn.Module):
    def __init__(self, num_heads: int, head_size: int, scale: float, num_kv_heads: Optional[int] = None,
                 alibi_slopes: Optional[list[float]] = None, sliding_window: Optional[int] = None,
                 logits_soft_cap: Optional[float] = None, prefix: str = "", cache_dtype: str = "auto"):
        super().__init__()
        self.num_heads, self.head_size, self.scale = num_heads, head_size, float(scale)
        self.num_kv_heads = num_kv_heads or num_heads
        self.sliding_window = sliding_window or 0
        self.softcap = float(logits_soft_cap or 0.0)
        self.layer_name = prefix
        self.cache_dtype = cache_dtype
        # fp8 KV cache scales: stored value = x / scale (1.0 unless the checkpoint carries calibrated scales)
        self._k_scale_float = self._v_scale_float = 1.0
        from ..config import get_current_vllm_config_or_none
        cfg = get_current_vllm_config_or_none()
        qc = getattr(cfg, "quant_config", None) if cfg is not None else None
        self.quant_method = qc.get_quant_method(self, prefix) if qc is not None else None
        if self.quant_method is not None:
            self.quant_method.create_weights(self)
        if alibi_slopes is not None:
            self.register_buffer("alibi_slopes", torch.tensor(alibi_slopes, dtype=torch.float32), persistent=False)
        else:
            self.alibi_slopes = None
        # bound by the model runner after KV memory planning
        self.k_cache: Optional[torch.Tensor] = None
        self.v_cache: Optional[torch.Tensor] = None
        if head_size not in (64, 128, 256):
            raise ValueError(f"head size {head_size} is not supported by the MFMA attention kernels (64/128/256)")

    def get_kv_cache_spec(self, block_size: int, dtype: torch.dtype) -> KVCacheSpec:
        if self.sliding_window:
            return SlidingWindowSpec(block_size=block_size, num_kv_heads=self.num_kv_heads,
                                     head_size=self.head_size, dtype=dtype, sliding_window=self.sliding_window)
        return FullAttentionSpec(block_size=block_size, num_kv_heads=self.num_kv_heads, head_size=self.head_size,
                                 dtype=dtype)

    def bind_kv_cache(self, k_cache: torch.Tensor, v_cache: torch.Tensor) -> None:
        self.k_cache, self.v_cache = k_cache, v_cache

    def forward(self, qkv: torch.Tensor, positions: Optional[torch.Tensor] = None, rotary=None) -> torch.Tensor:
        """qkv: [T, (Hq + 2 Hkv) * D] fused projection output.  Returns [T, Hq * D]."""
        ctx = get_forward_context()
        md: Optional[AttentionMetadata] = ctx.attn_metadata
        T = qkv.shape[0]
        Hq, Hkv, D = self.num_heads, self.num_kv_heads, self.head_size
        qs, ks = Hq * D, Hkv * D
        out = torch.empty((T, Hq, D), dtype=qkv.dtype, device=qkv.device)
        if md is None or self.k_cache is None:
            # memory-profiling run: no cache bound yet, the projections are what matters
            if rotary is not None and positions is not None:
                rotary(positions, qkv[:, :qs], qkv[:, qs:qs + ks])
            out.zero_()
            return out.view(T, qs)
        n = md.num_actual_tokens
        ksc, vsc = self._k_scale_float, self._v_scale_float
        conn = ctx.kv_connector
        if conn is not None:
            conn.wait_for_layer_load(self.layer_name)
        if rotary is not None:
            ops.rope_and_cache(positions[:n], qkv[:n], rotary.cos_sin_cache, self.k_cache, self.v_cache,
                               md.slot_mapping[:n], rotary.rotary_dim, D, Hq, Hkv, rotary.is_neox_style, ksc, vsc)
        else:
            ops.reshape_and_cache(qkv[:n, qs:qs + ks].view(n, Hkv, D), qkv[:n, qs + ks:].view(n, Hkv, D),
                                  self.k_cache, self.v_cache, md.slot_mapping[:n], ksc, vsc)
        q = qkv[:, :qs].view(T, Hq, D)
        nd, ndt = md.num_decode_seqs, md.num_decode_tokens
        if nd > 0:
            ops.paged_attention_decode(out[:ndt], q[:ndt], self.k_cache, self.v_cache, md.block_table[:nd],
                                       md.seq_lens[:nd], md.query_start_loc[:nd + 1], self.scale,
                                       num_splits=md.decode_num_splits, max_q_len=md.max_decode_query_len,
                                       window=self.sliding_window, softcap=self.softcap, alibi=self.alibi_slopes,
                                       workspace=md.decode_workspace, k_scale=ksc, v_scale=vsc)
        if md.num_prefill_seqs > 0:
            e = ndt + md.num_prefill_tokens
            ops.paged_attention_prefill(out[ndt:e], q[ndt:e], self.k_cache, self.v_cache, md.block_table[nd:],
                                        md.seq_lens[nd:], md.prefill_query_start_loc, self.scale,
                                        window=self.sliding_window, softcap=self.softcap, alibi=self.alibi_slopes,
                                        k_scale=ksc, v_scale=vsc)
        if conn is not None:
            conn.save_kv_layer(self.layer_name, (self.k_cache, self.v_cache), md)
        return out.view(T, qs)
