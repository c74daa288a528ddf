"""Idefics3 / SmolVLM: SigLIP-style tower (closing LayerNorm included) -> pixel shuffle (scale_factor x scale_factor
patches stacked on the channel axis) -> one bias-free projection -> a Llama decoder.  An image is resized so that its
longest edge is at most ``longest_edge`` and both edges are multiples of the tower's input size, cut into square
tiles (row-major) and, when there is more than one tile, followed by the whole image scaled to one tile.
Reference: vllm/model_executor/models/idefics3.py (Idefics3Connector :430-465, Idefics3Model :468-560,
Idefics3ForConditionalGeneration :563-786), vllm/model_executor/models/smolvlm.py.
"""
from __future__ import annotations

import math
from typing import Any, Iterable

import numpy as np
import torch
import torch.nn as nn

from ..config import VllmConfig
from .llava import _cfg
from .llava_next import SiglipVisionModel, build_language_model
from .utils import load_weights_with_mapping, maybe_prefix


class Idefics3ForConditionalGeneration(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vc, tc = _cfg(hf.vision_config), _cfg(hf.text_config)
        self.image_token_id = int(getattr(hf, "image_token_id", 128257))
        self.scale = int(getattr(hf, "scale_factor", 2))
        self.vision_model = SiglipVisionModel(vc, -1)
        self.vision_model.post_layernorm = nn.LayerNorm(vc.hidden_size, eps=float(getattr(vc, "layer_norm_eps", 1e-6)))
        self.connector = nn.Module()
        self.connector.modality_projection = nn.Module()
        self.connector.modality_projection.proj = nn.Linear(vc.hidden_size * self.scale ** 2, tc.hidden_size, bias=False)
        self.language_model = build_language_model(vllm_config, tc, maybe_prefix(prefix, "language_model"))

    def _shuffle(self, x: torch.Tensor) -> torch.Tensor:
        n, t, c = x.shape
        s, f = int(t ** 0.5), self.scale
        x = x.reshape(n, s, s // f, c * f).permute(0, 2, 1, 3)
        x = x.reshape(n, s // f, s // f, c * f * f).permute(0, 2, 1, 3)
        return x.reshape(n, -1, c * f * f)

    def encode_mm_items(self, items: list[list[torch.Tensor]]) -> list[torch.Tensor]:
        dtype = self.connector.modality_projection.proj.weight.dtype
        tiles = torch.cat([it[0].to(dtype) for it in items], dim=0)
        vm = self.vision_model
        f = self.connector.modality_projection.proj(self._shuffle(vm.post_layernorm(vm(tiles))))
        out, k = [], 0
        for it in items:
            n = it[0].shape[0]
            out.append(f[k:k + n].flatten(0, 1))
            k += n
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
        dec, vis = [], []
        for n, t in weights:
            if n.startswith("model.text_model."):
                dec.append(("language_model.model." + n[len("model.text_model."):], t))
            elif n.startswith("lm_head."):
                dec.append(("language_model." + n, t))
            elif n.startswith("model."):
                vis.append((n[len("model."):], t))
            else:
                vis.append((n, t))
        lm = self.language_model
        loaded = load_weights_with_mapping(self, dec, lm.stacked_params_mapping, tie_lm_head=lm.tie)
        loaded |= load_weights_with_mapping(self, vis, [])
        return loaded


class SmolVLMForConditionalGeneration(Idefics3ForConditionalGeneration):
    """The same network under SmolVLM's name (other sizes and prompt tags only)."""


class Idefics3ImageProcessor:
    """Image -> tiles [n, 3, S, S] (+ the whole image as the last tile when n > 1).  A prompt marks an image either
    with one run of image tokens (expanded here to n * tokens_per_tile) or, when a chat template already expanded it,
    with one run of tokens_per_tile per tile between its text tags (then every tile is an item of its own)."""

    def __init__(self, image_token_id: int, S: int, tokens_per_tile: int, longest_edge: int, split: bool = True):
        self.image_token_id, self.S, self.tpt, self.longest, self.split = image_token_id, S, tokens_per_tile, longest_edge, split

    def tiles(self, img: Any) -> torch.Tensor:
        if isinstance(img, dict):
            return torch.as_tensor(img["pixel_values"]).float()
        if hasattr(img, "convert"):
            img = np.asarray(img.convert("RGB"))
        t = torch.as_tensor(np.asarray(img)).permute(2, 0, 1).float()[None]
        H, W = t.shape[-2:]
        S = self.S

        def resize(h, w):
            return nn.functional.interpolate(t, size=(h, w), mode="bicubic", align_corners=False,
                                             antialias=True).clamp(0, 255)
        if self.split:
            r = min(self.longest / max(H, W), 1e9)
            h, w = max(int(round(H * r)), 1), max(int(round(W * r)), 1)
            rows, cols = max(math.ceil(h / S), 1), max(math.ceil(w / S), 1)
        else:
            rows = cols = 1
        big = resize(rows * S, cols * S)[0]
        parts = big.unfold(1, S, S).unfold(2, S, S).permute(1, 2, 0, 3, 4).reshape(-1, 3, S, S)
        if parts.shape[0] > 1:
            parts = torch.cat((parts, resize(S, S)), dim=0)
        return (parts / 255.0 - 0.5) / 0.5

    def __call__(self, prompt_token_ids: list[int], mm_data: dict):
        from ..multimodal import PlaceholderRange, hash_item
        images = mm_data.get("image")
        if images is None:
            raise ValueError("multi_modal_data needs an 'image' entry")
        if not isinstance(images, (list, tuple)):
            images = [images]
        tiles = [self.tiles(i) for i in images]
        runs, prev = [], -2
        for i, tok in enumerate(prompt_token_ids):
            if tok == self.image_token_id:
                if i != prev + 1:
                    runs.append([i, 1])
                else:
                    runs[-1][1] += 1
                prev = i
        if len(runs) == len(images):
            units = [[t] for t in tiles]                                 # one item per image
        elif len(runs) == sum(t.shape[0] for t in tiles) and all(n == self.tpt for _, n in runs):
            units = [[t[j:j + 1]] for t in tiles for j in range(t.shape[0])]     # one item per tile
        else:
            raise ValueError(f"prompt has {len(runs)} runs of image tokens for {len(images)} images "
                             f"({sum(t.shape[0] for t in tiles)} tiles)")
        out, positions, inputs, hashes, cur = [], [], [], [], 0
        for (start, n), unit in zip(runs, units):
            ntok = self.tpt * unit[0].shape[0]
            out.extend(prompt_token_ids[cur:start])
            positions.append(PlaceholderRange(offset=len(out), length=ntok))
            out.extend([self.image_token_id] * ntok)
            cur = start + n
            inputs.append(unit)
            hashes.append(hash_item(unit[0]))
        out.extend(prompt_token_ids[cur:])
        return out, inputs, hashes, positions


def idefics3_processor(vllm_config):
    hf = vllm_config.model_config.hf_config
    vc = _cfg(hf.vision_config)
    S, P, f = int(vc.image_size), int(vc.patch_size), int(getattr(hf, "scale_factor", 2))
    return Idefics3ImageProcessor(int(getattr(hf, "image_token_id", 128257)), S, (S // P // f) ** 2,
                                  int(getattr(hf, "longest_edge", 4 * S)))
