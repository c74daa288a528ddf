"""Multimodal inputs: registry of per-architecture processors, content hashing, placeholder expansion.

A processor turns ``{"prompt_token_ids": [...], "multi_modal_data": {"image": ...}}`` into
  * the prompt with every image token expanded to as many placeholder tokens as the encoder emits,
  * one encoder input per item (pre-processed pixel tensor),
  * one content hash per item (prefix caching keys and encoder-cache identity),
  * one ``PlaceholderRange`` per item (where its embeddings land in the prompt).
Images may be PIL images, ``[H, W, 3]`` uint8 arrays / tensors, or already pre-processed
``[3, S, S]`` float tensors.
Reference: vllm/multimodal/{registry.py, processing.py, hasher.py, inputs.py}, vllm/v1/engine/mm_input_cache.py.
"""
from __future__ import annotations

import hashlib
from typing import Any, Callable, Optional

import numpy as np
import torch

from ..request import PlaceholderRange

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def hash_item(x: Any) -> str:
    h = hashlib.sha256()
    if isinstance(x, torch.Tensor):
        h.update(str((tuple(x.shape), str(x.dtype))).encode())
        h.update(x.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())
    elif isinstance(x, np.ndarray):
        h.update(str((x.shape, x.dtype.str)).encode())
        h.update(np.ascontiguousarray(x).tobytes())
    elif hasattr(x, "tobytes") and hasattr(x, "size"):        # PIL image
        h.update(str((x.size, getattr(x, "mode", ""))).encode())
        h.update(x.tobytes())
    else:
        h.update(repr(x).encode())
    return h.hexdigest()


def to_pixel_values(img: Any, size: int, mean=CLIP_MEAN, std=CLIP_STD) -> torch.Tensor:
    """CLIP-style preprocessing: shortest side to ``size`` (bicubic), centre crop, scale to [0,1], normalise."""
    if isinstance(img, torch.Tensor) and img.dtype.is_floating_point and img.dim() == 3 and img.shape[0] == 3:
        return img.float()                                   # already pre-processed
    if hasattr(img, "convert"):                              # PIL
        img = np.asarray(img.convert("RGB"))
    t = torch.as_tensor(np.asarray(img) if not isinstance(img, torch.Tensor) else img)
    if t.dim() != 3 or t.shape[-1] != 3:
        raise ValueError(f"image must be [H, W, 3] or a pre-processed [3, S, S] float tensor, got {tuple(t.shape)}")
    t = t.permute(2, 0, 1).float().unsqueeze(0)
    H, W = t.shape[-2:]
    s = size / min(H, W)
    nh, nw = max(int(round(H * s)), size), max(int(round(W * s)), size)
    if (nh, nw) != (H, W):
        t = torch.nn.functional.interpolate(t, size=(nh, nw), mode="bicubic", align_corners=False, antialias=True)
    top, left = (nh - size) // 2, (nw - size) // 2
    t = t[0, :, top:top + size, left:left + size].clamp(0, 255) / 255.0
    m = torch.tensor(mean).view(3, 1, 1)
    sd = torch.tensor(std).view(3, 1, 1)
    return (t - m) / sd


class MultiModalRegistry:
    """architecture name -> processor factory; processors are built once per model config."""

    def __init__(self):
        self._factories: dict[str, Callable] = {}
        self._cache: dict[int, Any] = {}

    def register_processor(self, arch: str, factory: Callable) -> None:
        self._factories[arch] = factory

    def supports(self, arch: str) -> bool:
        return arch in self._factories

    def get_processor(self, vllm_config):
        key = id(vllm_config.model_config)
        if key not in self._cache:
            arch = (vllm_config.model_config.architectures or [""])[0]
            if arch not in self._factories:
                raise ValueError(f"{arch} takes no multimodal inputs")
            self._cache[key] = self._factories[arch](vllm_config)
        return self._cache[key]


MULTIMODAL_REGISTRY = MultiModalRegistry()


class ImageTokenProcessor:
    """Models whose prompt marks each image with ONE special token that the encoder output replaces
    (LLaVA family)."""

    def __init__(self, image_token_id: int, tokens_per_image: int, image_size: int):
        self.image_token_id, self.tokens_per_image, self.image_size = image_token_id, tokens_per_image, image_size

    def __call__(self, prompt_token_ids: list[int], mm_data: dict):
        images = mm_data.get("image")
        if images is None:
            raise ValueError("multi_modal_data needs an 'image' entry")
        if not isinstance(images, (list, tuple)):
            images = [images]
        spots = [i for i, t in enumerate(prompt_token_ids) if t == self.image_token_id]
        # prompts may already carry the expanded run of image tokens: collapse runs first
        runs, prev = [], -2
        for i in spots:
            if i != prev + 1:
                runs.append([i, 1])
            else:
                runs[-1][1] += 1
            prev = i
        if len(runs) != len(images):
            raise ValueError(f"prompt has {len(runs)} image placeholders but {len(images)} images were given")
        out, positions, inputs, hashes = [], [], [], []
        cur = 0
        for (start, n), img in zip(runs, images):
            out.extend(prompt_token_ids[cur:start])
            positions.append(PlaceholderRange(offset=len(out), length=self.tokens_per_image))
            out.extend([self.image_token_id] * self.tokens_per_image)
            cur = start + n
            inputs.append(to_pixel_values(img, self.image_size))
            hashes.append(hash_item(img))
        out.extend(prompt_token_ids[cur:])
        return out, inputs, hashes, positions


def process_multimodal(vllm_config, prompt_token_ids: list[int], mm_data: dict):
    """(expanded prompt ids, encoder inputs, content hashes, placeholder ranges)."""
    return MULTIMODAL_REGISTRY.get_processor(vllm_config)(list(prompt_token_ids), mm_data)


def _llava_processor(vllm_config):
    hf = vllm_config.model_config.hf_config
    vc = hf.vision_config if isinstance(hf.vision_config, dict) else hf.vision_config.to_dict()
    size, patch = int(vc.get("image_size", 336)), int(vc.get("patch_size", 14))
    n = (size // patch) ** 2
    if getattr(hf, "vision_feature_select_strategy", "default") == "full":
        n += 1
    return ImageTokenProcessor(int(getattr(hf, "image_token_index", None) or getattr(hf, "image_token_id", 32000)),
                               n, size)


MULTIMODAL_REGISTRY.register_processor("LlavaForConditionalGeneration", _llava_processor)


# ---- Qwen2-VL: variable-size patch grids ------------------------------------------------------------
def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56,
                 max_pixels: int = 14 * 14 * 4 * 1280) -> tuple[int, int]:
    """Closest size to (height, width) whose sides are multiples of ``factor`` and whose area lies in
    [min_pixels, max_pixels], keeping the aspect ratio as far as the rounding allows."""
    import math
    if max(height, width) / max(min(height, width), 1) > 200:
        raise ValueError("aspect ratio above 200 is not supported")
    h = max(factor, round(height / factor) * factor)
    w = max(factor, round(width / factor) * factor)
    if h * w > max_pixels:
        beta = math.sqrt(height * width / max_pixels)
        h = max(factor, math.floor(height / beta / factor) * factor)
        w = max(factor, math.floor(width / beta / factor) * factor)
    elif h * w < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h = math.ceil(height * beta / factor) * factor
        w = math.ceil(width * beta / factor) * factor
    return h, w


class PatchGridProcessor:
    """Qwen2-VL style inputs: an image becomes a [grid_t * grid_h * grid_w, C * tps * patch * patch] matrix of
    flattened patches (ordered so that each 2x2 merge block is contiguous) plus its grid; the prompt's image
    token is expanded to grid_t * grid_h * grid_w / merge^2 placeholders.  Already pre-processed inputs are
    accepted as ``{"pixel_values": [N, K], "image_grid_thw": [3]}``."""

    def __init__(self, image_token_id: int, patch: int, merge: int, tps: int, min_pixels: int, max_pixels: int):
        self.image_token_id, self.patch, self.merge, self.tps = image_token_id, patch, merge, tps
        self.min_pixels, self.max_pixels = min_pixels, max_pixels

    def patches(self, img: Any) -> tuple[torch.Tensor, torch.Tensor]:
        if isinstance(img, dict):
            return torch.as_tensor(img["pixel_values"]).float(), torch.as_tensor(img["image_grid_thw"]).long().view(3)
        if hasattr(img, "convert"):
            img = np.asarray(img.convert("RGB"))
        t = torch.as_tensor(np.asarray(img) if not isinstance(img, torch.Tensor) else img)
        if t.dim() != 3 or t.shape[-1] != 3:
            raise ValueError(f"image must be [H, W, 3], got {tuple(t.shape)}")
        t = t.permute(2, 0, 1).float().unsqueeze(0)
        H, W = t.shape[-2:]
        h, w = smart_resize(H, W, self.patch * self.merge, self.min_pixels, self.max_pixels)
        if (h, w) != (H, W):
            t = torch.nn.functional.interpolate(t, size=(h, w), mode="bicubic", align_corners=False, antialias=True)
        t = t.clamp(0, 255) / 255.0
        t = (t - torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)) / torch.tensor(CLIP_STD).view(1, 3, 1, 1)
        t = t.repeat(self.tps, 1, 1, 1)                       # a still image fills one temporal patch
        p, m = self.patch, self.merge
        gh, gw = h // p, w // p
        x = t.view(1, self.tps, 3, gh // m, m, p, gw // m, m, p).permute(0, 3, 6, 4, 7, 2, 1, 5, 8)
        return x.reshape(gh * gw, 3 * self.tps * p * p).contiguous(), torch.tensor([1, gh, gw])

    def __call__(self, prompt_token_ids: list[int], mm_data: dict):
        images = mm_data.get("image")
        if images is None:
            raise ValueError("multi_modal_data needs an 'image' entry")
        if not isinstance(images, (list, tuple)):
            images = [images]
        runs, prev = [], -2
        for i, tok in enumerate(prompt_token_ids):
            if tok != self.image_token_id:
                continue
            if i != prev + 1:
                runs.append([i, 1])
            else:
                runs[-1][1] += 1
            prev = i
        if len(runs) != len(images):
            raise ValueError(f"prompt has {len(runs)} image placeholders but {len(images)} images were given")
        out, positions, inputs, hashes, cur = [], [], [], [], 0
        for (start, n), img in zip(runs, images):
            px, grid = self.patches(img)
            length = int(grid.prod()) // self.merge ** 2
            out.extend(prompt_token_ids[cur:start])
            positions.append(PlaceholderRange(offset=len(out), length=length))
            out.extend([self.image_token_id] * length)
            cur = start + n
            inputs.append([px, grid])
            hashes.append(hash_item(px if isinstance(img, dict) else img))
        out.extend(prompt_token_ids[cur:])
        return out, inputs, hashes, positions


def _qwen2_vl_processor(vllm_config):
    hf = vllm_config.model_config.hf_config
    vc = hf.vision_config if isinstance(hf.vision_config, dict) else hf.vision_config.to_dict()
    return PatchGridProcessor(int(getattr(hf, "image_token_id", 151655)), int(vc.get("patch_size", 14)),
                              int(vc.get("spatial_merge_size", 2)), int(vc.get("temporal_patch_size", 2)),
                              56 * 56, 14 * 14 * 4 * 1280)


MULTIMODAL_REGISTRY.register_processor("Qwen2VLForConditionalGeneration", _qwen2_vl_processor)
