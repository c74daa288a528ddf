"""Kimi-VL: MoonViT (images at their own resolution: a learned position table resampled to every image's patch
grid, a 2-D rotary embedding whose frequency pairs alternate between the column and the row of a patch, attention
within one image), a 2 x 2 patch merge, a two-layer projector behind a LayerNorm, and a DeepSeek-V3 decoder.

No transformers class exists for this family: parity unpinned; rotary angles, the resampled table and a tower block
are tested against the arithmetic written out with complex numbers.
Reference: vllm/model_executor/models/moonvit.py (Learnable2DInterpPosEmb :188-218, Rope2DPosEmb :265-376,
MoonVitEncoderLayer :403-487, patch_merger :527-551), vllm/model_executor/models/kimi_vl.py
(KimiVLMultiModalProjector :82-108, KimiVLForConditionalGeneration :283-480).
"""
from __future__ import annotations

from typing import Iterable

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from ..layers.basic import get_act_fn
from .llava import _cfg
from .llava_next import build_language_model
from .utils import load_weights_with_mapping, maybe_prefix


def rope_2d_angles(h: int, w: int, head_dim: int, base: float = 10000.0) -> torch.Tensor:
    """[h * w, head_dim / 2] rotation angles, row-major over the grid: pair 2i turns with the column, pair 2i + 1
    with the row, both at frequency base^(-4i / head_dim)."""
    freqs = 1.0 / base ** (torch.arange(0, head_dim, 4)[: head_dim // 4].float() / head_dim)
    rows, cols = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    ang = torch.stack([cols.reshape(-1, 1) * freqs, rows.reshape(-1, 1) * freqs], -1)      # [N, C/4, 2]
    return ang.reshape(h * w, head_dim // 2)


def rotate_pairs(x: torch.Tensor, ang: torch.Tensor) -> torch.Tensor:
    """x [N, heads, D] with (even, odd) channel pairs; ang [N, D / 2]."""
    x2 = x.float().view(*x.shape[:-1], -1, 2)
    c, s = ang.cos()[:, None, :], ang.sin()[:, None, :]
    out = torch.stack([x2[..., 0] * c - x2[..., 1] * s, x2[..., 0] * s + x2[..., 1] * c], -1)
    return out.flatten(-2).to(x.dtype)


class _MoonBlock(nn.Module):
    def __init__(self, d: int, heads: int, mlp: int):
        super().__init__()
        self.heads = heads
        self.norm0, self.norm1 = nn.LayerNorm(d), nn.LayerNorm(d)
        self.wqkv, self.wo = nn.Linear(d, 3 * d), nn.Linear(d, d)
        self.mlp = nn.Module()
        self.mlp.fc0, self.mlp.fc1 = nn.Linear(d, mlp), nn.Linear(mlp, d)

    def forward(self, x, ang, spans: list[int]):
        N, d = x.shape
        q, k, v = self.wqkv(self.norm0(x)).view(N, 3, self.heads, d // self.heads).unbind(1)
        q, k = rotate_pairs(q, ang), rotate_pairs(k, ang)
        outs, o = [], 0
        for n in spans:                                         # one image attends within itself
            qs, ks, vs = (t[o:o + n].transpose(0, 1) for t in (q, k, v))
            outs.append(F.scaled_dot_product_attention(qs, ks, vs).transpose(0, 1))
            o += n
        x = x + self.wo(torch.cat(outs).reshape(N, d))
        return x + self.mlp.fc1(F.gelu(self.mlp.fc0(self.norm1(x)), approximate="tanh"))


class MoonViT(nn.Module):
    def __init__(self, vc):
        super().__init__()
        d, p = int(vc.hidden_size), int(vc.patch_size)
        self.head_dim = d // int(vc.num_attention_heads)
        self.merge = tuple(int(v) for v in getattr(vc, "merge_kernel_size", (2, 2)))
        self.patch_embed = nn.Module()
        self.patch_embed.proj = nn.Conv2d(3, d, p, p)
        self.patch_embed.pos_emb = nn.Module()
        self.patch_embed.pos_emb.weight = nn.Parameter(
            torch.zeros(int(vc.init_pos_emb_height), int(vc.init_pos_emb_width), d))
        self.encoder = nn.Module()
        self.encoder.blocks = nn.ModuleList(_MoonBlock(d, int(vc.num_attention_heads), int(vc.intermediate_size))
                                            for _ in range(int(vc.num_hidden_layers)))
        self.encoder.final_layernorm = nn.LayerNorm(d)

    def positions(self, h: int, w: int) -> torch.Tensor:
        t = self.patch_embed.pos_emb.weight
        if t.shape[:2] != (h, w):
            t = F.interpolate(t.permute(2, 0, 1)[None].float(), size=(h, w), mode="bicubic")[0].permute(1, 2, 0) \
                .to(t.dtype)
        return t.reshape(h * w, -1)

    def forward(self, patches: torch.Tensor, grids: list[tuple[int, int]]) -> list[torch.Tensor]:
        """patches [sum h * w, 3, p, p] row-major per image -> per image [h / 2 * w / 2, 4, d]"""
        pe = self.patch_embed
        x = pe.proj(patches.to(pe.proj.weight.dtype)).flatten(1)
        x = x + torch.cat([self.positions(h, w) for h, w in grids]).to(x.dtype)
        ang = torch.cat([rope_2d_angles(h, w, self.head_dim) for h, w in grids]).to(x.device)
        spans = [h * w for h, w in grids]
        for blk in self.encoder.blocks:
            x = blk(x, ang, spans)
        x = self.encoder.final_layernorm(x)
        outs, o = [], 0
        kh, kw = self.merge
        for h, w in grids:
            s = x[o:o + h * w].view(h // kh, kh, w // kw, kw, -1).permute(0, 2, 1, 3, 4)
            outs.append(s.reshape(h // kh * (w // kw), kh * kw, -1))
            o += h * w
        return outs


class KimiVLProjector(nn.Module):
    def __init__(self, hf, vc, out_dim: int):
        super().__init__()
        d = int(vc.hidden_size)
        kh, kw = (int(v) for v in getattr(vc, "merge_kernel_size", (2, 2)))
        self.wide = d * kh * kw
        self.pre_norm = nn.LayerNorm(d, eps=1e-5)
        self.linear_1, self.linear_2 = nn.Linear(self.wide, self.wide), nn.Linear(self.wide, out_dim)
        self.act = get_act_fn("gelu")

    def forward(self, x):
        return self.linear_2(self.act(self.linear_1(self.pre_norm(x).view(-1, self.wide))))


class KimiVLForConditionalGeneration(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vc, tc = _cfg(hf.vision_config), _cfg(hf.text_config)
        self.vision_tower = MoonViT(vc)
        self.multi_modal_projector = KimiVLProjector(hf, vc, int(tc.hidden_size))
        if not getattr(tc, "architectures", None):
            tc.architectures = ["DeepseekV3ForCausalLM"]
        self.language_model = build_language_model(vllm_config, tc, maybe_prefix(prefix, "language_model"))

    def encode_mm_items(self, items: list) -> list[torch.Tensor]:
        """items: (patches [h * w, 3, p, p], grid [2] = (h, w)) per image"""
        grids = [tuple(int(v) for v in g.tolist()) for _, g in items]
        merged = self.vision_tower(torch.cat([p for p, _ in items]), grids)
        return [self.multi_modal_projector(m) for m in merged]

    def get_input_embeddings(self, input_ids, mm_embeds=None, mm_mask=None):
        x = self.language_model.get_input_embeddings(input_ids)
        if mm_embeds is not None and mm_embeds.numel():
            x[mm_mask] = mm_embeds.to(x.dtype)
        return x

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        return self.language_model(input_ids, positions, intermediate_tensors, inputs_embeds)

    def compute_logits(self, hidden_states):
        return self.language_model.compute_logits(hidden_states)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        dec, own = [], []
        for n, t in weights:
            if n.startswith("language_model."):
                dec.append((n[len("language_model."):], t))
            else:
                own.append((n, t))
        loaded = {"language_model." + n for n in self.language_model.load_weights(dec)}
        loaded |= load_weights_with_mapping(self, own, [])
        return loaded


def kimi_vl_processor(vllm_config):
    """One placeholder per image -> (h / 2) * (w / 2) positions for an h x w patch grid.  Pixels: scaled down when
    the grid would exceed ``in_token_limit`` patches, padded bottom / right to a multiple of two patches, mean =
    std = 0.5.  Pre-processed inputs: a dict with ``pixel_values`` [h * w, 3, p, p] and ``image_grid_hws`` (h, w)."""
    from ..multimodal import PlaceholderRange, hash_item
    hf = vllm_config.model_config.hf_config
    vc = _cfg(hf.vision_config)
    p = int(vc.patch_size)
    kh, kw = (int(v) for v in getattr(vc, "merge_kernel_size", (2, 2)))
    tok = int(getattr(hf, "media_placeholder_token_id", None) or getattr(hf, "image_token_id", 163605))
    kwargs = getattr(vllm_config.model_config, "mm_processor_kwargs", None) or {}
    limit = int(kwargs.get("in_token_limit", 4096))

    def pixels(img):
        if isinstance(img, dict):
            return torch.as_tensor(img["pixel_values"]).float(), torch.as_tensor(img["image_grid_hws"]).long().view(2)
        import numpy as np
        a = np.asarray(img.convert("RGB")) if hasattr(img, "convert") else np.asarray(img)
        t = torch.as_tensor(a).permute(2, 0, 1).float()[None]
        H, W = t.shape[-2:]
        if (H // p) * (W // p) > limit:
            s = (limit / ((H // p) * (W // p))) ** 0.5
            H, W = max(p, int(H * s)), max(p, int(W * s))
            t = F.interpolate(t, size=(H, W), mode="bicubic", align_corners=False, antialias=True).clamp(0, 255)
        ph, pw = -(-H // (kh * p)) * kh * p, -(-W // (kw * p)) * kw * p
        full = torch.zeros(3, ph, pw)
        full[:, :H, :W] = t[0]
        full = (full / 255.0 - 0.5) / 0.5
        gh, gw = ph // p, pw // p
        return full.view(3, gh, p, gw, p).permute(1, 3, 0, 2, 4).reshape(gh * gw, 3, p, p), torch.tensor([gh, gw])

    def process(ids, mm_data):
        images = mm_data.get("image")
        if images is None:
            raise ValueError("multi_modal_data needs an 'image' entry")
        images = images if isinstance(images, (list, tuple)) else [images]
        out, ranges, inputs, hashes = [], [], [], []
        i, k = 0, 0
        while i < len(ids):
            if ids[i] != tok:
                out.append(ids[i])
                i += 1
                continue
            j = i
            while j < len(ids) and ids[j] == tok:
                j += 1
            if k >= len(images):
                raise ValueError("more image placeholders than images")
            px, g = pixels(images[k])
            n = (int(g[0]) // kh) * (int(g[1]) // kw)
            ranges.append(PlaceholderRange(offset=len(out), length=n))
            out.extend([tok] * n)
            inputs.append([px, g])
            hashes.append(hash_item(px if isinstance(images[k], dict) else images[k]))
            k, i = k + 1, j
        if k != len(images):
            raise ValueError(f"prompt has {k} image placeholders but {len(images)} images were given")
        return out, inputs, hashes, ranges

    return process
