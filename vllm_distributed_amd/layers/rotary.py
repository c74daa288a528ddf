"""Rotary position embeddings.

Every variant only differs in how the fp32 ``cos_sin_cache [max_pos, rot_dim]``
(``[cos | sin]``) is built; applying it is one HIP kernel (ops.rotary_embedding,
or fused with the KV-cache write in ops.rope_and_cache).
Reference: vllm/model_executor/layers/rotary_embedding.py (default :93, linear
:376, dynamic NTK :467, YaRN :556, LongRoPE :653, Llama-3 :885, M-RoPE :968).
"""
from __future__ import annotations

import math
from typing import Any, Optional

import torch
import torch.nn as nn

from .. import ops


def _inv_freq(base: float, rot_dim: int) -> torch.Tensor:
    return 1.0 / (base ** (torch.arange(0, rot_dim, 2, dtype=torch.float32) / rot_dim))


def _yarn_find_dim(num_rot: float, dim: int, base: float, max_pos: int) -> float:
    return dim * math.log(max_pos / (num_rot * 2 * math.pi)) / (2 * math.log(base))


def _yarn_range(low_rot, high_rot, dim, base, max_pos) -> tuple[int, int]:
    lo = math.floor(_yarn_find_dim(low_rot, dim, base, max_pos))
    hi = math.ceil(_yarn_find_dim(high_rot, dim, base, max_pos))
    return max(lo, 0), min(hi, dim - 1)


def _ramp(lo: float, hi: float, n: int) -> torch.Tensor:
    if lo == hi:
        hi += 0.001
    return ((torch.arange(n, dtype=torch.float32) - lo) / (hi - lo)).clamp(0, 1)


def yarn_mscale(scale: float = 1.0, mscale: float = 1.0) -> float:
    return 1.0 if scale <= 1 else 0.1 * mscale * math.log(scale) + 1.0


class RotaryEmbedding(nn.Module):
    def __init__(self, head_size: int, rotary_dim: int, max_position: int, base: float, is_neox_style: bool):
        super().__init__()
        self.head_size, self.rotary_dim = head_size, rotary_dim
        self.max_position, self.base, self.is_neox_style = max_position, base, is_neox_style
        self.register_buffer("cos_sin_cache", self._build().float().contiguous(), persistent=False)

    def _freqs(self) -> torch.Tensor:
        return _inv_freq(self.base, self.rotary_dim)

    def _positions(self) -> torch.Tensor:
        return torch.arange(self.max_position, dtype=torch.float32)

    def _mscale(self) -> float:
        return 1.0

    def _build(self) -> torch.Tensor:
        f = torch.outer(self._positions(), self._freqs())
        m = self._mscale()
        return torch.cat((f.cos() * m, f.sin() * m), dim=-1)

    def forward(self, positions: torch.Tensor, q: torch.Tensor, k: Optional[torch.Tensor] = None):
        ops.rotary_embedding(positions, q, k, self.cos_sin_cache, self.rotary_dim, self.head_size,
                             self.is_neox_style)
        return q, k


class LinearScalingRotaryEmbedding(RotaryEmbedding):
    def __init__(self, head_size, rotary_dim, max_position, base, is_neox_style, factor: float):
        self.factor = factor
        super().__init__(head_size, rotary_dim, int(max_position * factor), base, is_neox_style)

    def _positions(self):
        return torch.arange(self.max_position, dtype=torch.float32) / self.factor


class NTKScalingRotaryEmbedding(RotaryEmbedding):
    """Fixed NTK scaling (base * factor, every frequency divided by factor^(2 / rotary_dim)) and the "mixed" form
    whose divisor grows with the frequency index.  Reference: vllm/model_executor/layers/rotary_embedding.py:462-493."""

    def __init__(self, head_size, rotary_dim, max_position, base, is_neox_style, factor: float,
                 mixed_b: Optional[float] = None):
        self.factor, self.mixed_b = float(factor), mixed_b
        super().__init__(head_size, rotary_dim, max_position, base, is_neox_style)

    def _freqs(self):
        if self.mixed_b is None:
            return _inv_freq(self.base * self.factor, self.rotary_dim) / self.factor ** (2 / self.rotary_dim)
        half = self.rotary_dim // 2
        a = math.log(self.factor) / half ** self.mixed_b
        lam = (a * torch.arange(1, half + 1).float() ** self.mixed_b).exp()
        return _inv_freq(self.base, self.rotary_dim) / lam


class DynamicNTKScalingRotaryEmbedding(RotaryEmbedding):
    def __init__(self, head_size, rotary_dim, max_position, base, is_neox_style, factor: float):
        self.factor = factor
        self.orig_max = max_position
        super().__init__(head_size, rotary_dim, int(max_position * factor), base, is_neox_style)

    def _freqs(self):
        b = self.base * ((self.factor * self.max_position / self.orig_max) - (self.factor - 1)) ** (
            self.rotary_dim / (self.rotary_dim - 2))
        return _inv_freq(b, self.rotary_dim)


class YaRNScalingRotaryEmbedding(RotaryEmbedding):
    def __init__(self, head_size, rotary_dim, max_position, base, is_neox_style, factor: float,
                 extrapolation_factor: float = 1, attn_factor: float = 1, beta_fast: int = 32, beta_slow: int = 1,
                 mscale: Optional[float] = None):
        self.factor, self.orig_max = factor, max_position
        self.extrapolation_factor, self.beta_fast, self.beta_slow = extrapolation_factor, beta_fast, beta_slow
        self._m = (mscale if mscale is not None else yarn_mscale(factor)) * attn_factor
        super().__init__(head_size, rotary_dim, int(max_position * factor), base, is_neox_style)

    def _freqs(self):
        extra = _inv_freq(self.base, self.rotary_dim)
        inter = extra / self.factor
        lo, hi = _yarn_range(self.beta_fast, self.beta_slow, self.rotary_dim, self.base, self.orig_max)
        mask = (1 - _ramp(lo, hi, self.rotary_dim // 2)) * self.extrapolation_factor
        return inter * (1 - mask) + extra * mask

    def _mscale(self):
        return self._m


class Llama3RotaryEmbedding(RotaryEmbedding):
    def __init__(self, head_size, rotary_dim, max_position, base, is_neox_style, factor: float,
                 low_freq_factor: float, high_freq_factor: float, orig_max_position: int):
        self.factor, self.low, self.high, self.orig_max = factor, low_freq_factor, high_freq_factor, orig_max_position
        super().__init__(head_size, rotary_dim, max_position, base, is_neox_style)

    def _freqs(self):
        inv = _inv_freq(self.base, self.rotary_dim)
        wave = 2 * math.pi / inv
        smooth = (self.orig_max / wave - self.low) / (self.high - self.low) if self.high != self.low else \
            torch.zeros_like(inv)
        low_w, high_w = self.orig_max / self.low, self.orig_max / self.high
        mid = (1 - smooth) * inv / self.factor + smooth * inv
        return torch.where(wave < high_w, inv, torch.where(wave > low_w, inv / self.factor, mid))


class LongRoPEScaledRotaryEmbedding(RotaryEmbedding):
    """Phi-3 style: per-dimension rescale factors, a short and a long table stacked
    (positions >= original_max index the long table by adding an offset)."""

    def __init__(self, head_size, rotary_dim, max_position, orig_max_position, base, is_neox_style,
                 short_factor: list[float], long_factor: list[float]):
        self.orig_max, self.short_factor, self.long_factor = orig_max_position, short_factor, long_factor
        scale = max_position / orig_max_position
        self._m = 1.0 if scale <= 1.0 else math.sqrt(1 + math.log(scale) / math.log(orig_max_position))
        self.full_max = max_position
        super().__init__(head_size, rotary_dim, max_position, base, is_neox_style)

    def _table(self, n: int, factors: list[float]) -> torch.Tensor:
        inv = 1.0 / (torch.tensor(factors, dtype=torch.float32) * self.base ** (
            torch.arange(0, self.rotary_dim, 2, dtype=torch.float32) / self.rotary_dim))
        f = torch.outer(torch.arange(n, dtype=torch.float32), inv)
        return torch.cat((f.cos() * self._m, f.sin() * self._m), dim=-1)

    def _build(self):
        return torch.cat((self._table(self.orig_max, self.short_factor),
                          self._table(self.full_max, self.long_factor)), dim=0)

    def forward(self, positions, q, k=None):
        # a sequence longer than the original window reads the long table for all of its tokens
        use_long = bool((positions >= self.orig_max).any()) if positions.numel() else False
        pos = positions + self.orig_max if use_long else positions
        ops.rotary_embedding(pos, q, k, self.cos_sin_cache, self.rotary_dim, self.head_size, self.is_neox_style)
        return q, k


_CACHE: dict[Any, RotaryEmbedding] = {}


class MRotaryEmbedding(RotaryEmbedding):
    """Multimodal rotary embedding (Qwen2-VL): the frequency pairs are split into three sections that
    take their angle from the temporal, the height and the width position of a token.  Text tokens
    carry the same number on all three axes, which is the ordinary embedding, so steps without image
    tokens (every decode step) run the plain path on 1-D positions.  For a step with 3-D positions the
    model calls ``prepare`` once: it gathers one cos/sin row per token from the position table, and the
    fused RoPE + cache-write kernel then indexes that per-step table by token number.
    Reference: vllm/model_executor/layers/rotary_embedding.py (MRotaryEmbedding)."""

    def __init__(self, head_size, rotary_dim, max_position, base, is_neox_style, mrope_section: list[int]):
        super().__init__(head_size, rotary_dim, max_position, base, is_neox_style)
        if sum(mrope_section) != rotary_dim // 2:
            raise ValueError(f"mrope_section {mrope_section} does not add up to {rotary_dim // 2} frequency pairs")
        axis = torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(mrope_section)])
        self.register_buffer("axis_of", axis.repeat(2), persistent=False)      # cos half, sin half
        self.step_table: Optional[torch.Tensor] = None
        self.step_index: Optional[torch.Tensor] = None

    def prepare(self, positions: Optional[torch.Tensor]) -> None:
        """positions [3, T] (or None / 1-D to return to the plain path)."""
        if positions is None or positions.dim() == 1:
            self.step_table = self.step_index = None
            return
        cs = self.cos_sin_cache[positions]                                # [3, T, rot]
        T = cs.shape[1]
        sel = self.axis_of.view(1, -1, 1).expand(T, -1, 1)
        self.step_table = torch.gather(cs.permute(1, 2, 0), 2, sel).squeeze(-1).contiguous()
        self.step_index = torch.arange(T, dtype=torch.long, device=cs.device)

    def forward(self, positions: torch.Tensor, q: torch.Tensor, k: Optional[torch.Tensor] = None):
        if positions.dim() == 2:
            self.prepare(positions)
            try:
                ops.rotary_embedding(self.step_index, q, k, self.step_table, self.rotary_dim, self.head_size,
                                     self.is_neox_style)
            finally:
                self.prepare(None)
            return q, k
        return super().forward(positions, q, k)


def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float, is_neox_style: bool = True,
             rope_scaling: Optional[dict] = None, partial_rotary_factor: float = 1.0) -> RotaryEmbedding:
    if partial_rotary_factor < 1.0:
        rotary_dim = int(rotary_dim * partial_rotary_factor)
    key = (head_size, rotary_dim, max_position, base, is_neox_style,
           None if rope_scaling is None else tuple(sorted((k, str(v)) for k, v in rope_scaling.items())))
    if key in _CACHE:
        return _CACHE[key]
    kind = None if rope_scaling is None else rope_scaling.get("rope_type", rope_scaling.get("type"))
    rs = rope_scaling or {}
    if rs.get("mrope_section"):
        rope = MRotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style,
                                [int(x) for x in rs["mrope_section"]])
    elif kind in (None, "default", "mrope"):
        rope = RotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style)
    elif kind == "llama3":
        rope = Llama3RotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style, rs["factor"],
                                     rs["low_freq_factor"], rs["high_freq_factor"],
                                     rs["original_max_position_embeddings"])
    elif kind == "linear":
        rope = LinearScalingRotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style, rs["factor"])
    elif kind == "ntk":
        rope = NTKScalingRotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style, rs["factor"],
                                         rs.get("mixed_b"))
    elif kind == "dynamic":
        rope = DynamicNTKScalingRotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style,
                                                rs["factor"])
    elif kind == "yarn":
        orig = rs.get("original_max_position_embeddings", max_position)
        extra = {k: rs[k] for k in ("extrapolation_factor", "attn_factor", "beta_fast", "beta_slow") if k in rs}
        if rs.get("mscale") and rs.get("mscale_all_dim"):
            # DeepSeek flavour: the cos/sin magnitude is the ratio of two mscale terms, the rest of
            # the correction goes into the softmax scale (applied by the attention layer)
            extra["mscale"] = yarn_mscale(rs["factor"], float(rs["mscale"])) / \
                yarn_mscale(rs["factor"], float(rs["mscale_all_dim"]))
        rope = YaRNScalingRotaryEmbedding(head_size, rotary_dim, orig, base, is_neox_style, rs["factor"], **extra)
    elif kind in ("longrope", "su"):
        rope = LongRoPEScaledRotaryEmbedding(head_size, rotary_dim, max_position,
                                             rs["original_max_position_embeddings"], base, is_neox_style,
                                             rs["short_factor"], rs["long_factor"])
    else:
        raise ValueError(f"unsupported rope scaling type {kind!r}")
    _CACHE[key] = rope
    return rope
