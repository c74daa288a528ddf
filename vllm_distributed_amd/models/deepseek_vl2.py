"""DeepSeek-VL2: a SigLIP-SO400M tower (stored under timm's parameter names) over one padded 384-pixel overview
and the 384-pixel tiles of the image at its best candidate resolution; a projector that concatenates 2 x 2
neighbouring patches and runs an MLP; tiles stitched into one map, every map row closed by a learned "newline"
vector, the overview and the tile map separated by a learned "view separator"; then a DeepSeek-V2 decoder.

No transformers class exists for this family: parity unpinned; the tower block and the stitching are tested
against the arithmetic written out.
Reference: vllm/model_executor/models/deepseek_vl2.py (MlpProjector :100-160, DeepseekVLV2ForCausalLM :330-480,
_pixel_values_to_embedding :482-574).
"""
from __future__ import annotations

import math
from typing import Iterable

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from .llava import _cfg
from .llava_next import build_language_model, select_best_resolution
from .utils import load_weights_with_mapping, maybe_prefix


class _TimmBlock(nn.Module):
    def __init__(self, d: int, heads: int, hidden: int):
        super().__init__()
        self.heads = heads
        self.norm1, self.norm2 = nn.LayerNorm(d, eps=1e-6), nn.LayerNorm(d, eps=1e-6)
        self.attn = nn.Module()
        self.attn.qkv, self.attn.proj = nn.Linear(d, 3 * d), nn.Linear(d, d)
        self.mlp = nn.Module()
        self.mlp.fc1, self.mlp.fc2 = nn.Linear(d, hidden), nn.Linear(hidden, d)

    def forward(self, x):
        B, N, d = x.shape
        q, k, v = self.attn.qkv(self.norm1(x)).view(B, N, 3, self.heads, d // self.heads).permute(2, 0, 3, 1, 4)
        x = x + self.attn.proj(F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, d))
        return x + self.mlp.fc2(F.gelu(self.mlp.fc1(self.norm2(x)), approximate="tanh"))


class SiglipTimmTower(nn.Module):
    """SigLIP vision transformer under timm's names: no class token, learned positions, closing LayerNorm; the
    attention-pooling head of the checkpoint is not used."""

    def __init__(self, vc):
        super().__init__()
        d, p = int(vc.width), int(vc.patch_size)
        self.patch_embed = nn.Module()
        self.patch_embed.proj = nn.Conv2d(3, d, p, p)
        self.pos_embed = nn.Parameter(torch.zeros(1, (int(vc.image_size) // p) ** 2, d))
        hidden = int(d * float(getattr(vc, "mlp_ratio", 3.7362)))
        self.blocks = nn.ModuleList(_TimmBlock(d, int(vc.heads), hidden) for _ in range(int(vc.layers)))
        self.norm = nn.LayerNorm(d, eps=1e-6)

    def forward(self, pixels):
        x = self.patch_embed.proj(pixels.to(self.pos_embed.dtype)).flatten(2).transpose(1, 2) + self.pos_embed
        for blk in self.blocks:
            x = blk(x)
        return self.norm(x)


class MlpProjector(nn.Module):
    def __init__(self, pc):
        super().__init__()
        if getattr(pc, "projector_type", "downsample_mlp_gelu") != "downsample_mlp_gelu":
            raise NotImplementedError(f"projector type {pc.projector_type!r}")
        self.ratio = r = int(getattr(pc, "downsample_ratio", 2))
        wide = int(pc.n_embed) * int(getattr(pc, "mlp_ratio", 1))
        layers = [nn.Linear(int(pc.input_dim) * r * r, wide)]
        for _ in range(1, int(getattr(pc, "depth", 2)) - 1):
            layers += [nn.GELU(), nn.Linear(wide, wide)]
        layers += [nn.GELU(), nn.Linear(wide, int(pc.n_embed))]
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        """[B, g * g, C] -> [B, ceil(g / r)^2, n_embed]; an r x r neighbourhood is concatenated channel-major."""
        B, n, C = x.shape
        g, r = int(n ** 0.5), self.ratio
        x = x.view(B, g, g, C).permute(0, 3, 1, 2)
        pad = (-g) % r
        if pad:
            x = F.pad(x, (0, pad, 0, pad))
        return self.layers(F.unfold(x, kernel_size=r, stride=r).transpose(1, 2))


class DeepseekVLV2ForCausalLM(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vc, pc = _cfg(hf.vision_config), _cfg(hf.projector_config)
        tc = _cfg(getattr(hf, "language_config", None) or hf.text_config)
        if getattr(hf, "tile_tag", "2D") != "2D":
            raise NotImplementedError("only the 2-D tile layout is built")
        self.global_first = getattr(hf, "global_view_pos", "head") == "head"
        self.vision = SiglipTimmTower(vc)
        self.projector = MlpProjector(pc)
        n = int(pc.n_embed)
        self.image_newline = nn.Parameter(torch.zeros(n))
        self.view_seperator = nn.Parameter(torch.zeros(n))                  # the checkpoint's spelling
        if not getattr(tc, "architectures", None):
            tc.architectures = ["DeepseekV2ForCausalLM" if getattr(tc, "kv_lora_rank", None) or
                                getattr(tc, "n_routed_experts", None) else "DeepseekForCausalLM"]
        self.language = build_language_model(vllm_config, tc, maybe_prefix(prefix, "language"))

    def _rows(self, m: torch.Tensor) -> torch.Tensor:
        """[h, w, D] -> [h * (w + 1), D] with the newline vector behind every row."""
        return torch.cat([m, self.image_newline.to(m.dtype).expand(m.shape[0], 1, -1)], 1).reshape(-1, m.shape[-1])

    def encode_mm_items(self, items: list) -> list[torch.Tensor]:
        """items: (tiles [1 + tw * th, 3, S, S] with the overview first, crop [2] = (tiles across, tiles down))"""
        outs = []
        for px, crop in items:
            tw, th = (int(v) for v in crop.tolist())
            e = self.projector(self.vision(px[: 1 + tw * th]))                  # [1 + n, g * g, D]
            g, D = int(e.shape[1] ** 0.5), e.shape[-1]
            overview = self._rows(e[0].view(g, g, D))
            tiles = e[1:].view(th, tw, g, g, D).permute(0, 2, 1, 3, 4).reshape(th * g, tw * g, D)
            parts = [overview, self.view_seperator.to(e.dtype)[None], self._rows(tiles)]
            outs.append(torch.cat(parts if self.global_first else parts[::-1]))
        return outs

    def get_input_embeddings(self, input_ids, mm_embeds=None, mm_mask=None):
        x = self.language.get_input_embeddings(input_ids)
        if mm_embeds is not None and mm_embeds.numel():
            x[mm_mask] = mm_embeds.to(x.dtype)
        return x

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        return self.language(input_ids, positions, intermediate_tensors, inputs_embeds)

    def compute_logits(self, hidden_states):
        return self.language.compute_logits(hidden_states)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        dec, own = [], []
        for n, t in weights:
            if n.startswith("language."):
                dec.append((n[len("language."):], t))
            elif n.startswith("vision.attn_pool.") or n.startswith("vision.head."):
                continue
            else:
                own.append((n, t))
        loaded = {"language." + n for n in self.language.load_weights(dec)}
        loaded |= load_weights_with_mapping(self, own, [])
        return loaded


def deepseek_vl2_processor(vllm_config):
    """One placeholder per image -> g * (g + 1) overview positions + 1 + (th * g) * (tw * g + 1) tile positions with
    g = ceil(image_size / patch / downsample).  Pixels: the image padded (mean colour) to a square overview and to
    its best candidate resolution, cut into tiles; mean = std = 0.5.  Pre-processed inputs: a dict with
    ``pixel_values`` [1 + n, 3, S, S] and ``images_spatial_crop`` (tiles across, tiles down)."""
    from ..multimodal import PlaceholderRange, hash_item
    hf = vllm_config.model_config.hf_config
    vc, pc = _cfg(hf.vision_config), _cfg(hf.projector_config)
    S, patch = int(vc.image_size), int(vc.patch_size)
    g = math.ceil(S // patch / int(getattr(pc, "downsample_ratio", 2)))
    tok = int(getattr(hf, "image_token_id", None) or getattr(hf, "image_token_index", None) or 128815)
    candidates = [tuple(int(v) for v in c) for c in (getattr(hf, "candidate_resolutions", None) or [(S, S)])]

    def fit(t: torch.Tensor, h: int, w: int) -> torch.Tensor:
        """Resize keeping the aspect ratio, centre on an h x w canvas of the mean colour (0 after normalisation)."""
        H, W = t.shape[-2:]
        s = min(h / H, w / W)
        nh, nw = max(1, round(H * s)), max(1, round(W * s))
        t = F.interpolate(t, size=(nh, nw), mode="bicubic", align_corners=False, antialias=True).clamp(0, 255)
        out = torch.zeros(3, h, w)
        top, left = (h - nh) // 2, (w - nw) // 2
        out[:, top:top + nh, left:left + nw] = (t[0] / 255.0 - 0.5) / 0.5
        return out

    def pixels(img):
        if isinstance(img, dict):
            return torch.as_tensor(img["pixel_values"]).float(), torch.as_tensor(img["images_spatial_crop"]).long().view(2)
        import numpy as np
        a = np.asarray(img.convert("RGB")) if hasattr(img, "convert") else np.asarray(img)
        t = torch.as_tensor(a).permute(2, 0, 1).float()[None]
        bh, bw = select_best_resolution((t.shape[-2], t.shape[-1]), candidates)
        tiles = fit(t, bh, bw).view(3, bh // S, S, bw // S, S).permute(1, 3, 0, 2, 4).reshape(-1, 3, S, S)
        return torch.cat([fit(t, S, S)[None], tiles]), torch.tensor([bw // S, bh // S])

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
            px, crop = pixels(images[k])
            tw, th = int(crop[0]), int(crop[1])
            n = g * (g + 1) + 1 + (th * g) * (tw * g + 1)
            ranges.append(PlaceholderRange(offset=len(out), length=n))
            out.extend([tok] * n)
            inputs.append([px, crop])
            hashes.append(hash_item(px if isinstance(images[k], dict) else images[k]))
            k, i = k + 1, j
        if k != len(images):
            raise ValueError(f"prompt has {k} image placeholders but {len(images)} images were given")
        return out, inputs, hashes, ranges

    return process
