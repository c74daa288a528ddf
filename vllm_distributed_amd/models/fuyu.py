"""Fuyu: no vision tower.  The image is cut into 30 x 30 pixel patches, every patch is one linear projection away
from an input embedding, rows of patches are separated by the ordinary ``|NEWLINE|`` token, and the decoder is
Persimmon.  An item is (patches [rows * cols, 3 * p * p], grid [rows, cols]); its embedding run covers the patch
positions and the newline tokens between them (those get the decoder's own embedding of that token).
Reference: vllm/model_executor/models/fuyu.py."""
from __future__ import annotations

from typing import Any, Iterable

import numpy as np
import torch
import torch.nn as nn

from ..config import VllmConfig
from .llava import _cfg
from .llava_next import build_language_model
from .utils import load_weights_with_mapping, maybe_prefix

IMAGE_TOKEN_ID = 71011          # |SPEAKER|
NEWLINE_TOKEN_ID = 71019        # |NEWLINE|


class FuyuForCausalLM(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        tc = _cfg(hf.text_config) if getattr(hf, "text_config", None) else hf
        self.patch = int(getattr(hf, "patch_size", 30))
        self.vision_embed_tokens = nn.Linear(self.patch * self.patch * int(getattr(hf, "num_channels", 3)),
                                             tc.hidden_size)
        if not getattr(tc, "architectures", None) or tc.architectures == ["FuyuForCausalLM"]:
            tc.architectures = ["PersimmonForCausalLM"]
        self.language_model = build_language_model(vllm_config, tc, maybe_prefix(prefix, "language_model"))
        self.image_token_id = int(getattr(hf, "image_token_id", IMAGE_TOKEN_ID) or IMAGE_TOKEN_ID)
        self.newline_token_id = NEWLINE_TOKEN_ID

    def encode_mm_items(self, items: list[list[torch.Tensor]]) -> list[torch.Tensor]:
        w = self.vision_embed_tokens.weight
        out = []
        for patches, grid in items:
            rows, cols = int(grid[0]), int(grid[1])
            e = self.vision_embed_tokens(patches.to(w.dtype)).view(rows, cols, -1)
            nl = self.language_model.get_input_embeddings(
                torch.tensor([self.newline_token_id], device=w.device)).to(e.dtype)
            out.append(torch.cat((e, nl.view(1, 1, -1).expand(rows, 1, -1)), dim=1).reshape(rows * (cols + 1), -1))
        return out

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
            if n.startswith("model.language_model."):
                dec.append(("model." + n[len("model.language_model."):], t))
            elif n.startswith("language_model."):
                dec.append((n[len("language_model."):], t))
            elif n.startswith("lm_head."):
                dec.append((n, t))
            elif n.startswith("model.vision_embed_tokens."):
                own.append((n[len("model."):], t))
            else:
                own.append((n, t))
        loaded = {"language_model." + n for n in (self.language_model.load_weights(dec) or ())}
        loaded |= load_weights_with_mapping(self, own, [])
        return loaded


class FuyuImageProcessor:
    """Image -> rows of patches.  Input: an [H, W, 3] image (padded with white to whole patches, scaled to [-1, 1]),
    or pre-cut patches as {"patches": [rows * cols, 3 * p * p], "grid": (rows, cols)}."""

    def __init__(self, patch: int, image_token_id: int = IMAGE_TOKEN_ID, newline_token_id: int = NEWLINE_TOKEN_ID,
                 max_hw: tuple[int, int] = (1080, 1920)):
        self.p, self.image_token_id, self.newline_token_id, self.max_hw = patch, image_token_id, newline_token_id, max_hw

    def patches(self, img: Any) -> tuple[torch.Tensor, torch.Tensor]:
        if isinstance(img, dict):
            return torch.as_tensor(img["patches"]).float(), torch.as_tensor(img["grid"], dtype=torch.int64)
        if hasattr(img, "convert"):
            img = np.asarray(img.convert("RGB"))
        t = torch.as_tensor(np.asarray(img)).permute(2, 0, 1).float()
        H, W = t.shape[-2:]
        mh, mw = self.max_hw
        if H > mh or W > mw:
            s = min(mh / H, mw / W)
            t = torch.nn.functional.interpolate(t[None], size=(int(H * s), int(W * s)), mode="bilinear",
                                                align_corners=False, antialias=True)[0]
            H, W = t.shape[-2:]
        p = self.p
        ph, pw = -(-H // p) * p, -(-W // p) * p
        canvas = torch.full((3, ph, pw), 255.0)
        canvas[:, :H, :W] = t
        x = (canvas / 255.0 - 0.5) / 0.5
        rows, cols = ph // p, pw // p
        # a patch is stored row-major with the channel last: [p, p, 3]
        patches = x.unfold(1, p, p).unfold(2, p, p).permute(1, 2, 3, 4, 0).reshape(rows * cols, p * p * 3)
        return patches, torch.tensor([rows, cols], dtype=torch.int64)

    def __call__(self, prompt_token_ids: list[int], mm_data: dict):
        from ..multimodal import PlaceholderRange, hash_item
        images = mm_data.get("image")
        if images is None:
            raise ValueError("multi_modal_data needs an 'image' entry")
        if not isinstance(images, (list, tuple)):
            images = [images]
        spots = [i for i, t in enumerate(prompt_token_ids) if t == self.image_token_id]
        if len(spots) != len(images):
            raise ValueError(f"prompt has {len(spots)} image placeholders but {len(images)} images were given")
        out, positions, inputs, hashes = [], [], [], []
        cur = 0
        for start, img in zip(spots, images):
            patches, grid = self.patches(img)
            rows, cols = int(grid[0]), int(grid[1])
            out.extend(prompt_token_ids[cur:start])
            positions.append(PlaceholderRange(offset=len(out), length=rows * (cols + 1)))
            for _ in range(rows):
                out.extend([self.image_token_id] * cols + [self.newline_token_id])
            cur = start + 1
            inputs.append([patches, grid])
            hashes.append(hash_item(patches))
        out.extend(prompt_token_ids[cur:])
        return out, inputs, hashes, positions
