"""Qwen-VL (first generation): an OpenCLIP-style ViT over a 448 x 448 image, pooled to 256 vectors by one layer of
cross-attention from learned queries (2-D sine / cosine positions on queries and keys), in front of the first-generation
Qwen decoder.  In the prompt an image is ``<img>`` + 256 pad tokens + ``</img>``; the pad positions take the pooled
vectors.

No transformers class exists for this family (its modelling code ships with the checkpoint): parity unpinned; the
tower and the resampler are tested against the arithmetic written out with ``nn.MultiheadAttention`` pieces.
Reference: vllm/model_executor/models/qwen_vl.py (VisualAttention :75-170, VisionTransformer :279-366,
QwenVLForConditionalGeneration :659-790), vllm/model_executor/layers/resampler.py (get_abs_pos :45-62,
get_2d_sincos_pos_embed :115-150, Resampler2 :205-275).
"""
from __future__ import annotations

import math
from typing import Iterable, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from .renamed_decoders import QWenLMHeadModel
from .utils import load_weights_with_mapping


def sincos_2d(dim: int, h: int, w: int) -> torch.Tensor:
    """[h * w, dim]: first half of the channels encodes the column, second half the row (sine block, cosine block
    each), frequencies 1 / 10000^(2i / (dim / 2))."""
    def axis(pos: torch.Tensor, d: int) -> torch.Tensor:
        omega = 1.0 / 10000 ** (torch.arange(d // 2, dtype=torch.float32) / (d / 2.0))
        out = pos.reshape(-1, 1).float() * omega[None]
        return torch.cat([out.sin(), out.cos()], 1)
    rows, cols = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.cat([axis(cols, dim // 2), axis(rows, dim // 2)], 1)


def resize_positions(table: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """A square position table [s * s, C] resampled (bicubic) to h x w."""
    s = int(math.sqrt(table.shape[0]))
    if (s, s) == (h, w):
        return table
    t = table.float().view(1, s, s, -1).permute(0, 3, 1, 2)
    return F.interpolate(t, size=(h, w), mode="bicubic", align_corners=False).permute(0, 2, 3, 1).flatten(0, 2) \
        .to(table.dtype)


class QueryResampler(nn.Module):
    """grid^2 learned queries cross-attend once to the patch features; optional LayerNorm + projection behind."""

    def __init__(self, grid: int, dim: int, heads: int, kv_dim: int, post_projection: bool, adaptive: bool = False,
                 eps: float = 1e-6, num_queries: Optional[int] = None):
        """``num_queries`` given: that many queries WITHOUT a position table of their own (MiniCPM-V 2.5+), keys
        positioned by a table computed for the input's own grid (``adaptive``)."""
        super().__init__()
        self.dim, self.adaptive = dim, adaptive
        self.query = nn.Parameter(torch.zeros(num_queries or grid * grid, dim))
        self.kv_proj = nn.Linear(kv_dim, dim, bias=False) if kv_dim != dim else nn.Identity()
        self.attn = nn.MultiheadAttention(dim, heads)
        self.ln_q, self.ln_kv = nn.LayerNorm(dim, eps=eps), nn.LayerNorm(dim, eps=eps)
        self.pos_embed = nn.Parameter(sincos_2d(dim, grid, grid), requires_grad=False) if num_queries is None else None
        if post_projection:
            self.ln_post = nn.LayerNorm(dim, eps=eps)
            self.proj = nn.Parameter(torch.zeros(dim, dim))
        else:
            self.ln_post = self.proj = None

    def forward(self, x: torch.Tensor, size: Optional[tuple[int, int]] = None) -> torch.Tensor:
        """x [B, N, kv_dim] -> [B, grid^2, dim]"""
        h, w = size if size is not None else (int(math.sqrt(x.shape[1])),) * 2
        pos = sincos_2d(self.dim, h, w).to(x.device) if self.adaptive else resize_positions(self.pos_embed, h, w)
        kv = self.ln_kv(self.kv_proj(x)).transpose(0, 1)                       # [N, B, dim]
        q = self.ln_q(self.query)[:, None].expand(-1, x.shape[0], -1)
        if self.pos_embed is not None:
            q = q + self.pos_embed[:, None].to(kv.dtype)
        out = self.attn(q, kv + pos[:, None].to(kv.dtype), kv, need_weights=False)[0].transpose(0, 1)
        if self.proj is not None:
            out = self.ln_post(out) @ self.proj
        return out


class _Block(nn.Module):
    def __init__(self, width: int, heads: int, mlp_ratio: float):
        super().__init__()
        self.heads = heads
        self.ln_1, self.ln_2 = nn.LayerNorm(width, eps=1e-6), nn.LayerNorm(width, eps=1e-6)
        self.attn = nn.Module()
        self.attn.in_proj, self.attn.out_proj = nn.Linear(width, 3 * width), nn.Linear(width, width)
        self.mlp = nn.Module()
        self.mlp.c_fc, self.mlp.c_proj = nn.Linear(width, int(width * mlp_ratio)), nn.Linear(int(width * mlp_ratio), width)

    def forward(self, x):
        B, N, W = x.shape
        # the fused projection is laid out head-major: [heads, (q | k | v), head_dim]
        q, k, v = self.attn.in_proj(self.ln_1(x)).view(B, N, self.heads, 3, W // self.heads).permute(3, 0, 2, 1, 4)
        o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, W)
        x = x + self.attn.out_proj(o)
        return x + self.mlp.c_proj(F.gelu(self.mlp.c_fc(self.ln_2(x))))


class QwenVisionTransformer(nn.Module):
    def __init__(self, image_size: int, patch_size: int, width: int, layers: int, heads: int, mlp_ratio: float,
                 n_queries: int = 256, output_dim: int = 512, **_):
        super().__init__()
        self.conv1 = nn.Conv2d(3, width, patch_size, patch_size, bias=False)
        self.positional_embedding = nn.Parameter(torch.zeros(256, width))
        self.ln_pre = nn.LayerNorm(width, eps=1e-6)
        self.transformer = nn.Module()
        self.transformer.resblocks = nn.ModuleList(_Block(width, heads, mlp_ratio) for _ in range(layers))
        self.attn_pool = QueryResampler(int(math.sqrt(n_queries)), output_dim, output_dim // 128, width, False)
        self.ln_post = nn.LayerNorm(output_dim, eps=1e-6)
        self.proj = nn.Parameter(torch.zeros(output_dim, output_dim))

    def forward(self, pixels: torch.Tensor) -> torch.Tensor:
        """[B, 3, S, S] -> [B, n_queries, output_dim]"""
        x = self.conv1(pixels.to(self.conv1.weight.dtype)).flatten(2).transpose(1, 2)
        side = int(math.sqrt(x.shape[1]))
        x = self.ln_pre(x + resize_positions(self.positional_embedding, side, side))
        for blk in self.transformer.resblocks:
            x = blk(x)
        return self.ln_post(self.attn_pool(x)) @ self.proj


class QwenVLForConditionalGeneration(QWenLMHeadModel):
    is_multimodal = True
    supports_pp = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        vis = dict(self.config.visual)
        self.image_start_id = int(vis.get("image_start_id", 151857))
        self.image_pad_id = self.image_start_id + 2
        self.visual = QwenVisionTransformer(**vis)

    def encode_images(self, pixel_values: torch.Tensor) -> torch.Tensor:
        return self.visual(pixel_values)

    def get_input_embeddings(self, input_ids, mm_embeds=None, mm_mask=None):
        x = super().get_input_embeddings(input_ids)
        if mm_embeds is not None and mm_embeds.numel():
            x[mm_mask] = mm_embeds.to(x.dtype)
        return x

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        text, vis = [], []
        for n, t in weights:
            if n.startswith("transformer.visual."):
                vis.append((n[len("transformer."):], t))
            else:
                text.append((n, t))
        loaded = super().load_weights(text)
        loaded |= load_weights_with_mapping(self, vis, [])
        return loaded


def qwen_vl_processor(vllm_config):
    """A run of image pad tokens (or one) -> ``<img>`` + n_queries pads + ``</img>``; pixels: plain bicubic resize
    to the square input, CLIP mean / std."""
    from ..multimodal import CLIP_MEAN, CLIP_STD, PlaceholderRange, hash_item
    vis = dict(vllm_config.model_config.hf_config.visual)
    start = int(vis.get("image_start_id", 151857))
    end, pad = start + 1, start + 2
    size, nq = int(vis["image_size"]), int(vis.get("n_queries", 256))

    def pixels(img):
        if isinstance(img, torch.Tensor) and img.dtype.is_floating_point:
            return img
        import numpy as np
        a = np.asarray(img.convert("RGB")) if hasattr(img, "convert") else np.asarray(img)
        t = torch.as_tensor(a).permute(2, 0, 1).float()[None]
        t = F.interpolate(t, size=(size, size), mode="bicubic", align_corners=False, antialias=True)
        t = t[0].clamp(0, 255) / 255.0
        return (t - torch.tensor(CLIP_MEAN).view(3, 1, 1)) / torch.tensor(CLIP_STD).view(3, 1, 1)

    def process(ids, mm_data):
        images = mm_data.get("image")
        if images is None:
            raise ValueError("multi_modal_data needs an 'image' entry")
        images = images if isinstance(images, (list, tuple)) else [images]
        out, ranges, inputs, hashes = [], [], [], []
        i, k = 0, 0
        while i < len(ids):
            if ids[i] != pad:
                out.append(ids[i])
                i += 1
                continue
            j = i
            while j < len(ids) and ids[j] == pad:
                j += 1
            if k >= len(images):
                raise ValueError("more image placeholders than images")
            if not out or out[-1] != start:
                out.append(start)
            ranges.append(PlaceholderRange(offset=len(out), length=nq))
            out.extend([pad] * nq)
            if not (j < len(ids) and ids[j] == end):
                out.append(end)
            inputs.append(pixels(images[k]))
            hashes.append(hash_item(images[k]))
            k, i = k + 1, j
        if k != len(images):
            raise ValueError(f"prompt has {k} image placeholders but {len(images)} images were given")
        return out, inputs, hashes, ranges

    return process
