"""Pixtral vision tower and the decoders built on it: Mistral-Small-3.x (``Mistral3ForConditionalGeneration``:
RMSNorm + learned 2x2 patch merger + two-layer projector) and Pixtral in the transformers layout (a LLaVA
checkpoint whose ``vision_config.model_type`` is ``pixtral``: two-layer projector only).

The tower takes images of any size (a multiple of the patch size): every image is cut into patches by one
strided convolution, the patch sequences of all images of a call are concatenated, attention is block-diagonal
(a patch sees its own image only) and positions are two-dimensional rotary embeddings indexed by (row, column).
In the prompt an image is ``rows`` runs of ``cols`` image tokens, each run closed by a break token and the last
one by an end token; the features replace the image tokens, the break / end tokens keep their text embeddings.

Reference: vllm/model_executor/models/pixtral.py (PixtralHFVisionModel :1006-1267), mistral3.py
(Mistral3PatchMerger :81-125, Mistral3MultiModalProjector :127-164, Mistral3ForConditionalGeneration :399-590).
Checkpoints in Mistral's own ``consolidated.safetensors`` / ``params.json`` layout are not read; the
architecture name ``PixtralForConditionalGeneration`` resolves to the transformers layout.
"""
from __future__ import annotations

import math
from typing import Any, Iterable

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from ..layers.basic import get_act_fn
from .llava import _cfg
from .llava_next import build_language_model
from .utils import load_weights_with_mapping, maybe_prefix


class _RMSNorm(nn.Module):
    def __init__(self, n: int, eps: float):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n))
        self.eps = eps

    def forward(self, x):
        xf = x.float()
        return self.weight * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps)).to(x.dtype)


def _rotate_half(x):
    a, b = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-b, a), dim=-1)


class PixtralBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        H, self.heads = c.hidden_size, c.num_attention_heads
        self.hd = getattr(c, "head_dim", None) or H // self.heads
        self.attention_norm, self.ffn_norm = _RMSNorm(H, 1e-5), _RMSNorm(H, 1e-5)
        self.attention = nn.Module()
        for n in ("q_proj", "k_proj", "v_proj"):
            setattr(self.attention, n, nn.Linear(H, self.heads * self.hd, bias=False))
        self.attention.o_proj = nn.Linear(self.heads * self.hd, H, bias=False)
        self.feed_forward = nn.Module()
        self.feed_forward.gate_proj = nn.Linear(H, c.intermediate_size, bias=False)
        self.feed_forward.up_proj = nn.Linear(H, c.intermediate_size, bias=False)
        self.feed_forward.down_proj = nn.Linear(c.intermediate_size, H, bias=False)
        self.act = get_act_fn(getattr(c, "hidden_act", "silu"))

    def forward(self, x, cos, sin, mask):
        N = x.shape[0]
        a = self.attention
        h = self.attention_norm(x)
        q, k, v = (p(h).view(N, self.heads, self.hd).transpose(0, 1) for p in (a.q_proj, a.k_proj, a.v_proj))
        q = q * cos + _rotate_half(q) * sin
        k = k * cos + _rotate_half(k) * sin
        o = F.scaled_dot_product_attention(q[None], k[None], v[None], attn_mask=mask)[0]
        x = x + a.o_proj(o.transpose(0, 1).reshape(N, -1))
        f = self.feed_forward
        return x + f.down_proj(self.act(f.gate_proj(h2 := self.ffn_norm(x))) * f.up_proj(h2))


class PixtralVisionTower(nn.Module):
    """Layers up to the one whose output is used (``feature_layer``: an index into the list of hidden states,
    embeddings first, negative from the end)."""

    def __init__(self, c, feature_layer: int = -1):
        super().__init__()
        self.c = c
        H, P = c.hidden_size, c.patch_size
        self.patch_size = P
        self.patch_conv = nn.Conv2d(getattr(c, "num_channels", 3), H, P, P, bias=False)
        self.ln_pre = _RMSNorm(H, 1e-5)
        n = c.num_hidden_layers
        keep = feature_layer if feature_layer >= 0 else n + feature_layer + 1
        self.transformer = nn.Module()
        self.transformer.layers = nn.ModuleList([PixtralBlock(c) for _ in range(keep)])
        hd = getattr(c, "head_dim", None) or H // c.num_attention_heads
        rp = getattr(c, "rope_parameters", None) or {}
        theta = float(getattr(c, "rope_theta", None) or rp.get("rope_theta", 10000.0))
        side = c.image_size // P
        freqs = 1.0 / (theta ** (torch.arange(0, hd, 2).float() / hd))
        fh = torch.outer(torch.arange(side).float(), freqs[::2])
        fw = torch.outer(torch.arange(side).float(), freqs[1::2])
        table = torch.cat((fh[:, None, :].repeat(1, side, 1), fw[None, :, :].repeat(side, 1, 1)), dim=-1)
        self.register_buffer("rope_table", torch.cat((table, table), dim=-1).reshape(side * side, hd), persistent=False)
        self.side = side

    def forward(self, images: list[torch.Tensor]) -> torch.Tensor:
        """images: [3, H_i, W_i] each -> features [sum of patches, hidden] (images concatenated, row-major)."""
        dt = self.patch_conv.weight.dtype
        seqs, ids, lens = [], [], []
        for im in images:
            e = self.patch_conv(im.to(dt)[None])[0]                  # [H, h, w]
            h, w = e.shape[-2:]
            seqs.append(e.flatten(1).t())
            rows, cols = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
            ids.append((rows * self.side + cols).reshape(-1))
            lens.append(h * w)
        x = self.ln_pre(torch.cat(seqs, dim=0))
        emb = self.rope_table[torch.cat(ids).to(x.device)]
        cos, sin = emb.cos().to(x.dtype), emb.sin().to(x.dtype)
        mask = None
        if len(lens) > 1:                                            # a patch attends within its own image
            owner = torch.repeat_interleave(torch.arange(len(lens)), torch.tensor(lens)).to(x.device)
            mask = (owner[:, None] == owner[None, :])[None, None]
        for layer in self.transformer.layers:
            x = layer(x, cos, sin, mask)
        return x


class Mistral3ForConditionalGeneration(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vc, tc = _cfg(hf.vision_config), _cfg(hf.text_config)
        self.vc = vc
        if getattr(hf, "tie_word_embeddings", False):          # the wrapper's setting is the checkpoint's truth
            tc.tie_word_embeddings = True
        self.image_token_id = int(getattr(hf, "image_token_index", None) or getattr(hf, "image_token_id", 10))
        self.break_id = int(getattr(hf, "image_break_token_id", 12))
        self.end_id = int(getattr(hf, "image_end_token_id", 13))
        fl = getattr(hf, "vision_feature_layer", -1)
        self.vision_tower = PixtralVisionTower(vc, int(fl))
        self.merge = int(getattr(hf, "spatial_merge_size", 1) or 1)
        self.has_merger = "Mistral3" in (vllm_config.model_config.architectures or [""])[0] or self.merge > 1
        bias = bool(getattr(hf, "multimodal_projector_bias", not self.has_merger))
        p = self.multi_modal_projector = nn.Module()
        if self.has_merger:
            p.norm = _RMSNorm(vc.hidden_size, float(getattr(tc, "rms_norm_eps", 1e-5)))
            p.patch_merger = nn.Module()
            p.patch_merger.merging_layer = nn.Linear(vc.hidden_size * self.merge ** 2, vc.hidden_size, bias=False)
        p.linear_1 = nn.Linear(vc.hidden_size, tc.hidden_size, bias=bias)
        p.linear_2 = nn.Linear(tc.hidden_size, tc.hidden_size, bias=bias)
        self.proj_act = get_act_fn(getattr(hf, "projector_hidden_act", "gelu"))
        self.language_model = build_language_model(vllm_config, tc, maybe_prefix(prefix, "language_model"))

    def _project(self, feats: torch.Tensor, grids: list[tuple[int, int]]) -> torch.Tensor:
        p = self.multi_modal_projector
        if self.has_merger:
            feats = p.norm(feats)
            m, d, parts = self.merge, feats.shape[-1], []
            for f, (h, w) in zip(feats.split([h * w for h, w in grids]), grids):
                g = F.unfold(f.view(h, w, d).permute(2, 0, 1)[None], kernel_size=m, stride=m)
                parts.append(g.view(d * m * m, -1).t())
            feats = p.patch_merger.merging_layer(torch.cat(parts, dim=0))
        return p.linear_2(self.proj_act(p.linear_1(feats)))

    def encode_mm_items(self, items: list[list[torch.Tensor]]) -> list[torch.Tensor]:
        """Per image: features in row-major order with the break / end tokens' text embeddings at the row ends."""
        P = self.vc.patch_size
        images = [it[0] for it in items]
        grids = [(im.shape[-2] // P, im.shape[-1] // P) for im in images]
        feats = self._project(self.vision_tower(images), grids)
        seps = self.language_model.get_input_embeddings(
            torch.tensor([self.break_id, self.end_id], device=feats.device)).to(feats.dtype)
        out, k = [], 0
        for h, w in grids:
            h, w = h // self.merge, w // self.merge
            f = feats[k:k + h * w].view(h, w, -1)
            k += h * w
            tail = seps[0].expand(h, 1, -1).clone()
            tail[-1, 0] = seps[1]
            out.append(torch.cat((f, tail), dim=1).flatten(0, 1))
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
        n_layers = len(self.vision_tower.transformer.layers)

        def rename(n: str):
            if n.startswith("model.language_model."):
                return "language_model.model." + n[len("model.language_model."):]
            for p in ("vision_tower.", "multi_modal_projector."):
                if n.startswith("model." + p):
                    n = n[len("model."):]
            if n.startswith("lm_head."):
                return "language_model." + n
            if n.startswith("vision_tower.transformer.layers.") and int(n.split(".")[3]) >= n_layers:
                return None
            return n
        dec, vis = [], []
        for n, t in weights:
            r = rename(n)
            if r is not None:
                (dec if r.startswith("language_model.") else vis).append((r, t))
        lm = self.language_model
        sub = set(lm.load_weights((n[len("language_model."):], t) for n, t in dec) or ())
        if "lm_head.weight" not in sub and not getattr(lm, "tie", False) and hasattr(lm, "lm_head"):
            # transformers leaves the head out of a checkpoint whose wrapper config ties it (the default of the
            # Mistral-3 wrapper, which it does not write into config.json either)
            lm.lm_head.tie_weights(lm.model.embed_tokens)
            lm.tie = True
            sub.add("lm_head.weight")
        loaded = {"language_model." + n for n in sub}
        loaded |= load_weights_with_mapping(self, vis, [])
        return loaded


class PixtralForConditionalGeneration(Mistral3ForConditionalGeneration):
    """Pixtral in the transformers layout: the tower above and LLaVA's two-layer projector."""


# ---- processor ------------------------------------------------------------------------------------------------
class PixtralImageProcessor:
    """Image -> pixels [3, H, W] with H, W multiples of patch x merge (longest side at most ``longest_edge``), and
    the prompt's image token -> rows of image tokens closed by break tokens, the last by the end token.
    Pre-processed input is accepted as {"pixel_values": [3, H, W]}."""

    def __init__(self, image_token_id: int, break_id: int, end_id: int, patch: int, merge: int, longest_edge: int,
                 mean, std):
        self.image_token_id, self.break_id, self.end_id = image_token_id, break_id, end_id
        self.P, self.M, self.longest, self.mean, self.std = patch, merge, longest_edge, mean, std

    def pixels(self, img: Any) -> torch.Tensor:
        if isinstance(img, dict):
            return torch.as_tensor(img["pixel_values"]).float()
        if hasattr(img, "convert"):
            img = np.asarray(img.convert("RGB"))
        t = torch.as_tensor(np.asarray(img)).permute(2, 0, 1).float()[None]
        H, W = t.shape[-2:]
        ratio = max(H / self.longest, W / self.longest)
        if ratio > 1:
            H, W = int(math.floor(H / ratio)), int(math.floor(W / ratio))
        step = self.P * self.M
        nh, nw = ((H - 1) // step + 1) * step, ((W - 1) // step + 1) * step
        r = F.interpolate(t, size=(nh, nw), mode="bicubic", align_corners=False, antialias=True).clamp(0, 255)
        m, sd = torch.tensor(self.mean).view(1, 3, 1, 1), torch.tensor(self.std).view(1, 3, 1, 1)
        return ((r / 255.0 - m) / sd)[0]

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
        step = self.P * self.M
        for start, img in zip(spots, images):
            px = self.pixels(img)
            rows, cols = px.shape[-2] // step, px.shape[-1] // step
            out.extend(prompt_token_ids[cur:start])
            positions.append(PlaceholderRange(offset=len(out), length=rows * (cols + 1)))
            for r in range(rows):
                out.extend([self.image_token_id] * cols + [self.end_id if r == rows - 1 else self.break_id])
            cur = start + 1
            inputs.append([px])
            hashes.append(hash_item(px))
        out.extend(prompt_token_ids[cur:])
        return out, inputs, hashes, positions


def pixtral_processor(vllm_config):
    from ..multimodal import CLIP_MEAN, CLIP_STD
    hf = vllm_config.model_config.hf_config
    vc = hf.vision_config if isinstance(hf.vision_config, dict) else hf.vision_config.to_dict()
    return PixtralImageProcessor(int(getattr(hf, "image_token_index", None) or getattr(hf, "image_token_id", 10)),
                                 int(getattr(hf, "image_break_token_id", 12)), int(getattr(hf, "image_end_token_id", 13)),
                                 int(vc.get("patch_size", 16)), int(getattr(hf, "spatial_merge_size", 1) or 1),
                                 int(vc.get("image_size", 1024)), CLIP_MEAN, CLIP_STD)
