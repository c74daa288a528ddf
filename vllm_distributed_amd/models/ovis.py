"""Ovis (1.6 / 2): images are "tokenised" softly.  A ViT backbone (AIMv2: RMSNorm blocks with a gated MLP, or
SigLIP) turns a 2 x 2 (``hidden_stride``) neighbourhood of patches into one vector, a linear head + LayerNorm turns
that into logits over a visual vocabulary, and the softmax of the logits weights the rows of a visual embedding
table: every image position is a convex mixture of learned visual words.  The last five words of that table are
structure markers (image start, prefix, column separator, row separator, image end) placed around the overview and
between the crops of a partitioned image.

No transformers class exists for this family: parity unpinned; the backbone block, the neighbourhood merge, the soft
lookup and the marker layout are tested against the arithmetic written out.
Reference: vllm/model_executor/models/ovis.py (VisualTokenizer :60-180, VisualEmbedding :207-228,
_process_image_input :479-509), vllm/model_executor/models/aimv2.py (AIMv2Block :121-139),
vllm/transformers_utils/processors/ovis.py (construct_image_indicators :256-269, preprocess_image :287-366).
"""
from __future__ import annotations

import math
from typing import Iterable

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from .llava import _cfg
from .llava_next import SiglipVisionModel, build_language_model
from .qwen2_vl import _RMS
from .utils import load_weights_with_mapping, maybe_prefix

N_MARKERS = 5          # start, prefix, column separator, row separator, end: the last rows of the visual table


class _Aimv2Block(nn.Module):
    def __init__(self, c):
        super().__init__()
        d, self.heads = int(c.hidden_size), int(c.num_attention_heads)
        eps, bias = float(getattr(c, "rms_norm_eps", 1e-5)), bool(getattr(c, "use_bias", False))
        self.norm_1, self.norm_2 = _RMS(d, eps), _RMS(d, eps)
        self.attn = nn.Module()
        self.attn.qkv = nn.Linear(d, 3 * d, bias=bool(getattr(c, "qkv_bias", False)))
        self.attn.proj = nn.Linear(d, d, bias=bias)
        self.mlp = nn.Module()
        inter = int(c.intermediate_size)
        self.mlp.fc1, self.mlp.fc3 = nn.Linear(d, inter, bias=bias), nn.Linear(d, inter, bias=bias)
        self.mlp.fc2 = nn.Linear(inter, d, bias=bias)

    def forward(self, x):
        B, N, d = x.shape
        q, k, v = (t.view(B, N, self.heads, -1).transpose(1, 2) for t in self.attn.qkv(self.norm_1(x)).chunk(3, -1))
        x = x + self.attn.proj(F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, d))
        h = self.norm_2(x)
        return x + self.mlp.fc2(F.silu(self.mlp.fc1(h)) * self.mlp.fc3(h))


class Aimv2Model(nn.Module):
    def __init__(self, c):
        super().__init__()
        d, p = int(c.hidden_size), int(c.patch_size)
        pre = self.preprocessor = nn.Module()
        pre.patchifier = nn.Module()
        pre.patchifier.proj = nn.Conv2d(int(getattr(c, "num_channels", 3)), d, p, p)
        pre.patchifier.norm = _RMS(d, float(getattr(c, "rms_norm_eps", 1e-5)))
        pre.pos_embed = nn.Parameter(torch.zeros(1, (int(c.image_size) // p) ** 2, d))
        self.trunk = nn.Module()
        self.trunk.blocks = nn.ModuleList(_Aimv2Block(c) for _ in range(int(c.num_hidden_layers)))

    def forward(self, pixels):
        pre = self.preprocessor
        x = pre.patchifier.norm(pre.patchifier.proj(pixels.to(pre.pos_embed.dtype)).flatten(2).transpose(1, 2))
        x = x + pre.pos_embed[:, : x.shape[1]]
        for blk in self.trunk.blocks:
            x = blk(x)
        return x


class VisualTokenizer(nn.Module):
    def __init__(self, c):
        super().__init__()
        bc = _cfg(c.backbone_config)
        kind = getattr(bc, "model_type", "aimv2")
        if kind == "aimv2":
            self.backbone = Aimv2Model(bc)
        elif kind == "siglip_vision_model":
            self.backbone = nn.Module()
            vm = self.backbone.vision_model = SiglipVisionModel(bc, -1)
            vm.post_layernorm = nn.LayerNorm(int(bc.hidden_size), eps=float(getattr(bc, "layer_norm_eps", 1e-6)))
        else:
            raise ValueError(f"visual tokenizer backbone {kind!r} is not supported")
        self.kind, self.stride = kind, int(getattr(c, "hidden_stride", 1))
        self.drop_cls = bool(getattr(c, "drop_cls_token", False))
        self.fn = getattr(c, "tokenize_function", "softmax")
        words = int(c.vocab_size) - N_MARKERS
        self.head = nn.Sequential(nn.Linear(int(bc.hidden_size) * self.stride ** 2, words, bias=False),
                                  nn.LayerNorm(words))

    def features(self, pixels: torch.Tensor) -> torch.Tensor:
        """[n, 3, S, S] -> [n, ceil(g / stride)^2, stride^2 * d]: ``stride`` x ``stride`` patches concatenated."""
        if self.kind == "aimv2":
            x = self.backbone(pixels)
        else:
            vm = self.backbone.vision_model
            x = vm.post_layernorm(vm(pixels))
        if self.drop_cls:
            x = x[:, 1:]
        s = self.stride
        if s > 1:
            n, L, d = x.shape
            g = int(L ** 0.5)
            x = x.view(n, g, g, d)
            pad = (-g) % s
            if pad:
                x = F.pad(x, (0, 0, 0, pad, 0, pad))
                g += pad
            x = x.view(n, g // s, s, g // s, s, d).permute(0, 1, 3, 2, 4, 5).reshape(n, (g // s) ** 2, s * s * d)
        return x

    def forward(self, pixels: torch.Tensor) -> torch.Tensor:
        """-> [n, tokens, visual vocabulary]: a distribution over the visual words, zero on the markers."""
        logits = self.head(self.features(pixels))
        if self.fn == "softmax":
            probs = torch.softmax(logits, -1)
        elif self.fn in ("st_argmax", "gumbel_argmax"):      # inference: the hard choice
            probs = torch.zeros_like(logits).scatter_(-1, logits.argmax(-1, keepdim=True), 1.0)
        else:
            raise ValueError(f"unknown tokenize_function {self.fn!r}")
        return F.pad(probs, (0, N_MARKERS))


def marker_rows(n_crops: int, grid: tuple[int, int], vocab: int) -> list[int]:
    """Visual-table rows of the markers of one image, in prompt order: start, prefix, then column / row separators
    between the crops of a ``rows x cols`` partition, end."""
    start, prefix, col, row, end = (vocab - 1 - i for i in range(5))
    out = [start, prefix]
    if grid[0] * grid[1] > 1:
        for r in range(grid[0]):
            out.extend([col] * (grid[1] - 1))
            if r < grid[0] - 1:
                out.append(row)
    return out + [end]


class Ovis(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vt = _cfg(hf.visual_tokenizer_config)
        tc = _cfg(getattr(hf, "llm_config", None) or hf.text_config)
        self.visual_tokenizer = VisualTokenizer(vt)
        self.visual_vocab = int(vt.vocab_size)
        self.vte = nn.Embedding(self.visual_vocab, int(tc.hidden_size))
        self.llm = build_language_model(vllm_config, tc, maybe_prefix(prefix, "llm"))

    def encode_mm_items(self, items: list) -> list[torch.Tensor]:
        """items: (crops [n, 3, S, S] with the overview first when n > 1, grid [2] = (rows, cols))"""
        outs = []
        w = self.vte.weight
        for px, grid in items:
            g = tuple(int(v) for v in grid.tolist())
            words = self.visual_tokenizer(px.to(w.dtype)) @ w                      # [n, tokens, H]
            marks = w[torch.tensor(marker_rows(px.shape[0], g, self.visual_vocab), device=w.device)]
            parts = []
            for i in range(px.shape[0]):
                parts += [marks[i:i + 1], words[i]]
            parts.append(marks[px.shape[0]:])
            outs.append(torch.cat(parts))
        return outs

    def get_input_embeddings(self, input_ids, mm_embeds=None, mm_mask=None):
        x = self.llm.get_input_embeddings(input_ids)
        if mm_embeds is not None and mm_embeds.numel():
            x[mm_mask] = mm_embeds.to(x.dtype)
        return x

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        return self.llm(input_ids, positions, intermediate_tensors, inputs_embeds)

    def compute_logits(self, hidden_states):
        return self.llm.compute_logits(hidden_states)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        dec, own = [], []
        for n, t in weights:
            if n.startswith("llm."):
                dec.append((n[len("llm."):], t))
            elif "post_trunk_norm" in n or ".vision_model.head." in n:
                continue
            else:
                own.append((n, t))
        loaded = {"llm." + n for n in self.llm.load_weights(dec)}
        loaded |= load_weights_with_mapping(self, own, [])
        return loaded


def best_partition(w: int, h: int, side: int, max_partition: int = 9, threshold: float = 0.9) -> tuple[int, int]:
    """(rows, cols): the smallest partition whose crops, each shown at up to ``side`` pixels, cover more than
    ``threshold`` of the image's pixels; the best covering one when none does."""
    def cover(pw, ph):
        a, b = max(pw, ph), min(pw, ph)
        if a > side:
            b, a = b / a * side, side
        return a * b
    good, every = [], []
    for r in range(1, max_partition + 1):
        for c in range(1, max_partition + 1):
            if r * c > max_partition:
                continue
            rh, cw = h // r, w // c
            total = 0.0
            for i in range(r):
                for j in range(c):
                    total += cover((w - j * cw) if j == c - 1 else cw, (h - i * rh) if i == r - 1 else rh)
            ratio = total / (w * h)
            every.append(((r, c), ratio))
            if ratio > threshold:
                good.append(((r, c), ratio))
    if good:
        return sorted(good, key=lambda x: (x[0][0] * x[0][1], -x[1]))[0][0]
    return sorted(every, key=lambda x: (-x[1], x[0][0] * x[0][1]))[0][0]


def ovis_processor(vllm_config):
    """One placeholder per image -> markers and visual positions: start, overview, prefix, then the crops with column /
    row separators, end.  Pixels: every crop resized (aspect kept) into a zero-padded square, CLIP mean / std.
    Pre-processed inputs: a dict with ``pixel_values`` [n, 3, S, S] and ``grid`` (rows, cols)."""
    from ..multimodal import CLIP_MEAN, CLIP_STD, PlaceholderRange, hash_item
    hf = vllm_config.model_config.hf_config
    vt = _cfg(hf.visual_tokenizer_config)
    bc = _cfg(vt.backbone_config)
    side, patch, stride = int(bc.image_size), int(bc.patch_size), int(getattr(vt, "hidden_stride", 1))
    per_crop = math.ceil(math.ceil(side / patch) / stride) ** 2
    tc = _cfg(getattr(hf, "llm_config", None) or hf.text_config)
    pads = {"gemma2": 7, "llama": 128002, "qwen2": 151655}
    kw = dict(getattr(vllm_config.model_config, "mm_processor_kwargs", None) or {})
    tok = int(kw.get("image_pad_token_id", getattr(hf, "image_pad_token_id", None)
                     or pads.get(getattr(tc, "model_type", ""), 151655)))
    max_partition = int(kw.get("max_partition", 9))
    mean, std = torch.tensor(CLIP_MEAN).view(3, 1, 1), torch.tensor(CLIP_STD).view(3, 1, 1)

    def square(t: torch.Tensor) -> torch.Tensor:
        h, w = t.shape[-2:]
        nw, nh = (side, side) if w == h else ((side, max(1, int(h / w * side))) if w > h else
                                              (max(1, int(w / h * side)), side))
        t = F.interpolate(t[None], size=(nh, nw), mode="bicubic", align_corners=False, antialias=True)[0]
        t = (t.clamp(0, 255) / 255.0 - mean) / std
        out = torch.zeros(3, side, side)
        top, left = (side - nh) // 2, (side - nw) // 2
        out[:, top:top + nh, left:left + nw] = t
        return out

    def pixels(img):
        if isinstance(img, dict):
            return torch.as_tensor(img["pixel_values"]).float(), torch.as_tensor(img["grid"]).long().view(2)
        import numpy as np
        a = np.asarray(img.convert("RGB")) if hasattr(img, "convert") else np.asarray(img)
        t = torch.as_tensor(a).permute(2, 0, 1).float()
        H, W = t.shape[-2:]
        rows, cols = best_partition(W, H, side, max_partition)
        crops = []
        rh, cw = H // rows, W // cols
        for i in range(rows):
            for j in range(cols):
                crops.append(t[:, i * rh: H if i == rows - 1 else (i + 1) * rh, j * cw: W if j == cols - 1 else (j + 1) * cw])
        if len(crops) > 1:
            crops.insert(0, t)
        return torch.stack([square(c) for c in crops]), torch.tensor([rows, cols])

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
            px, grid = pixels(images[k])
            n = px.shape[0]
            total = n * per_crop + (n + 1 if n > 1 else 3)
            ranges.append(PlaceholderRange(offset=len(out), length=total))
            out.extend([tok] * total)
            inputs.append([px, grid])
            hashes.append(hash_item(px if isinstance(images[k], dict) else images[k]))
            k, i = k + 1, j
        if k != len(images):
            raise ValueError(f"prompt has {k} image placeholders but {len(images)} images were given")
        return out, inputs, hashes, ranges

    return process
