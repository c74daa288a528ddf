"""Phi-3-vision / Phi-3.5-vision: a CLIP ViT-L/14-336 tower applied to a 336-pixel overview of the image and to the
336-pixel crops of its padded high-resolution version; patch features (layer -2, class token dropped) are merged
2 x 2, crops are stitched back into one map, a learned separator vector closes every row of the map, another one
sits between the crop map and the overview map, and a two-layer MLP projects everything into the Phi-3 decoder.

No transformers class exists for this family: parity unpinned (the CLIP tower itself is the one checked against
transformers with LLaVA); the stitching is tested against a loop over crops and rows.
Reference: vllm/model_executor/models/phi3v.py (CLIP_VIT_LARGE_PATCH14_336_CONFIG :67-78, Phi3HDImageEmbedding
:117-283, Phi3VForCausalLM :513-720).
"""
from __future__ import annotations

import math
from typing import Iterable

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import HFConfig, VllmConfig
from .llama_family import Phi3ForCausalLM
from .llava import ClipVisionModel
from .utils import load_weights_with_mapping

_CLIP_L_336 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                   image_size=336, patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5, num_channels=3)


class Phi3HDImageEmbedding(nn.Module):
    def __init__(self, config):
        super().__init__()
        ip = dict(getattr(config, "img_processor", None) or {})
        clip = HFConfig({**_CLIP_L_336, **{k: v for k, v in ip.items() if k in _CLIP_L_336}})
        self.crop = int(clip.image_size)
        self.grid = self.crop // int(clip.patch_size)                    # patches per crop side (24)
        self.img_processor = nn.Module()
        self.img_processor.vision_model = ClipVisionModel(clip, int(ip.get("layer_idx", -2)))
        self.dim = C = int(ip.get("image_dim_out", clip.hidden_size))
        self.glb_GN = nn.Parameter(torch.zeros(1, 1, 4 * C))
        self.sub_GN = nn.Parameter(torch.zeros(1, 1, 1, 4 * C))
        H = int(config.hidden_size)
        self.img_projection = nn.Sequential(nn.Linear(4 * C, H), nn.GELU(), nn.Linear(H, H))

    def _merged_map(self, feats: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
        """[rows * cols, grid^2, C] crop features -> [rows * grid / 2, cols * grid / 2, 4C]: neighbouring 2 x 2
        patches concatenated, crops put back side by side."""
        g, C = self.grid, feats.shape[-1]
        x = feats.view(rows, cols, g // 2, 2, g // 2, 2, C).permute(0, 2, 1, 4, 3, 5, 6)
        return x.reshape(rows * g // 2, cols * g // 2, 4 * C)

    def _with_row_ends(self, m: torch.Tensor) -> torch.Tensor:
        return torch.cat([m, self.sub_GN[0].expand(m.shape[0], 1, -1).to(m.dtype)], 1).reshape(-1, m.shape[-1])

    def forward(self, crops: torch.Tensor, size: tuple[int, int]) -> torch.Tensor:
        """crops [1 + n, 3, crop, crop] (overview first), size (h, w) of the padded image -> [tokens, H]"""
        rows, cols = size[0] // self.crop, size[1] // self.crop
        feats = self.img_processor.vision_model(crops[: 1 + rows * cols])[:, 1:]
        overview = self._with_row_ends(self._merged_map(feats[:1], 1, 1))
        detail = self._with_row_ends(self._merged_map(feats[1:], rows, cols))
        x = torch.cat([detail, self.glb_GN[0].to(detail.dtype), overview])
        return self.img_projection(x)


class Phi3VForCausalLM(Phi3ForCausalLM):
    is_multimodal = True
    supports_pp = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        self.vision_embed_tokens = Phi3HDImageEmbedding(self.config)

    def encode_mm_items(self, items: list) -> list[torch.Tensor]:
        w = self.vision_embed_tokens.img_projection[0].weight
        return [self.vision_embed_tokens(px.to(w.dtype), tuple(int(v) for v in size.tolist())) for px, size in items]

    def get_input_embeddings(self, input_ids, mm_embeds=None, mm_mask=None):
        ids = input_ids.clamp(min=0, max=self.config.vocab_size - 1)
        x = super().get_input_embeddings(ids)
        if mm_embeds is not None and mm_embeds.numel():
            x[mm_mask] = mm_embeds.to(x.dtype)
        return x

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        text, vis = [], []
        for n, t in weights:
            if n.startswith("model.vision_embed_tokens."):
                n = n[len("model."):]
                if n.startswith("vision_embed_tokens.wte."):
                    continue                                    # a copy of the token embedding
                if ".post_layernorm." in n:
                    continue
                vis.append((n, t))
            else:
                text.append((n, t))
        loaded = super().load_weights(text)
        params = dict(self.named_parameters())
        # only the tower layers up to the feature layer exist here
        vis = [(n, t) for n, t in vis if n in params]
        loaded |= load_weights_with_mapping(self, vis, [])
        return loaded


def hd_size(width: int, height: int, crop: int = 336, max_crops: int = 16) -> tuple[int, int, int, int]:
    """(resized width, resized height, padded width, padded height) of the high-resolution version: the longer side
    becomes ``scale * crop`` for the largest scale whose crop count stays within ``max_crops``, the shorter side
    follows the aspect ratio and is padded to a multiple of ``crop``."""
    tall = width < height
    w, h = (height, width) if tall else (width, height)
    ratio = w / h
    scale = 1
    while scale * math.ceil(scale / ratio) <= max_crops:
        scale += 1
    scale -= 1
    nw = int(scale * crop)
    nh = int(nw / ratio)
    ph = math.ceil(nh / crop) * crop
    return (nh, nw, ph, nw) if tall else (nw, nh, nw, ph)


def phi3v_processor(vllm_config):
    """One placeholder per image -> (crops + 1) * 144 + 1 + (crop rows + 1) * 12 positions (for 336-pixel crops).
    Pre-processed inputs: a dict with ``pixel_values`` [1 + n, 3, crop, crop] and ``image_sizes`` (h, w)."""
    from ..multimodal import CLIP_MEAN, CLIP_STD, PlaceholderRange, hash_item
    hf = vllm_config.model_config.hf_config
    ip = dict(getattr(hf, "img_processor", None) or {})
    crop = int(ip.get("image_size", 336))
    g2 = (crop // int(ip.get("patch_size", 14)) // 2)
    tok = int(getattr(hf, "image_token_id", None) or 32044)
    kw = getattr(vllm_config.model_config, "mm_processor_kwargs", None) or {}
    max_crops = int(kw.get("num_crops", getattr(hf, "num_crops", 16) or 16))

    def pixels(img):
        if isinstance(img, dict):
            return torch.as_tensor(img["pixel_values"]).float(), torch.as_tensor(img["image_sizes"]).long().view(2)
        import numpy as np
        a = np.asarray(img.convert("RGB")) if hasattr(img, "convert") else np.asarray(img)
        t = torch.as_tensor(a).permute(2, 0, 1).float()[None]
        H, W = t.shape[-2:]
        nw, nh, pw, ph = hd_size(W, H, crop, max_crops)
        t = F.interpolate(t, size=(nh, nw), mode="bilinear", align_corners=False, antialias=True).clamp(0, 255)
        top, left = (ph - nh) // 2, (pw - nw) // 2
        full = torch.full((1, 3, ph, pw), 255.0)                       # white padding, image centred
        full[:, :, top:top + nh, left:left + nw] = t
        full = (full / 255.0 - torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)) / torch.tensor(CLIP_STD).view(1, 3, 1, 1)
        overview = F.interpolate(full, size=(crop, crop), mode="bicubic", align_corners=False)
        tiles = full[0].view(3, ph // crop, crop, pw // crop, crop).permute(1, 3, 0, 2, 4).reshape(-1, 3, crop, crop)
        return torch.cat([overview, tiles]), torch.tensor([ph, pw])

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
            px, size = pixels(images[k])
            rows, cols = int(size[0]) // crop, int(size[1]) // crop
            n = (rows * cols + 1) * g2 * g2 + 1 + (rows + 1) * g2
            ranges.append(PlaceholderRange(offset=len(out), length=n))
            out.extend([tok] * n)
            inputs.append([px, size])
            hashes.append(hash_item(px if isinstance(images[k], dict) else images[k]))
            k, i = k + 1, j
        if k != len(images):
            raise ValueError(f"prompt has {k} image placeholders but {len(images)} images were given")
        return out, inputs, hashes, ranges

    return process
