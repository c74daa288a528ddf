"""Keye-VL: a SigLIP tower run NaViT-style (patches of any grid, row-major; the learned square position table resampled
bilinearly to each image's grid; 2-D rotary angles -- row frequencies on the first quarter of the head pairs, column
frequencies on the second -- and attention across all patches of an item, frames of a video included), a projector
that normalises, concatenates every 2 x 2 block of neighbours and applies a two-layer MLP, and a Qwen3 decoder with
the three-axis rotary positions of Qwen2-VL.

No transformers class exists for this family (its modelling code ships with the checkpoint): parity unpinned; the
tower and projector are tested against the arithmetic written out with complex rotations and per-head loops.
Reference: vllm/model_executor/models/keye.py (KeyeVisionEmbeddings :181-321, KeyeSiglipAttention :340-474,
KeyeSiglipEncoder :556-647, Projector :859-934, KeyeForConditionalGeneration :1327-1739).
"""
from __future__ import annotations

import math
from typing import Iterable

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from .llama import LlamaDecoderLayer, LlamaForCausalLM, LlamaModel
from .qwen2_vl import Qwen2VLForConditionalGeneration, VisionBlock
from .utils import load_weights_with_mapping, maybe_prefix


class KeyeVisionTower(nn.Module):
    def __init__(self, vc: dict):
        super().__init__()
        d, self.heads = int(vc["hidden_size"]), int(vc["num_attention_heads"])
        self.dim, self.patch = d, int(vc.get("patch_size", 14))
        side = int(vc.get("image_size", 384)) // self.patch
        self.patch_embedding = nn.Conv2d(int(vc.get("num_channels", 3)), d, self.patch, self.patch)
        self.position_embedding = nn.Embedding(side * side, d)
        self.blocks = nn.ModuleList(VisionBlock(d, self.heads, int(vc["intermediate_size"]),
                                                str(vc.get("hidden_act", "gelu_pytorch_tanh")))
                                    for _ in range(int(vc["num_hidden_layers"])))
        eps = float(vc.get("layer_norm_eps", 1e-6))
        for b in self.blocks:
            b.norm1.eps = b.norm2.eps = eps
        self.post_layernorm = nn.LayerNorm(d, eps=eps)

    def _positions(self, h: int, w: int) -> torch.Tensor:
        table = self.position_embedding.weight
        s = int(math.sqrt(table.shape[0]))
        t = table.view(1, s, s, -1).permute(0, 3, 1, 2).float()
        return F.interpolate(t, size=(h, w), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).reshape(h * w, -1)

    def forward(self, patches: torch.Tensor, grid: tuple[int, int, int]) -> torch.Tensor:
        """patches [t * h * w, C * p * p] row-major -> [t * h * w, d]"""
        t, h, w = grid
        wt = self.patch_embedding.weight
        x = F.linear(patches.to(wt.dtype), wt.view(self.dim, -1), self.patch_embedding.bias)
        x = x + self._positions(h, w).to(x.dtype).repeat(t, 1)
        hd = self.dim // self.heads
        inv = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float32, device=x.device) / (hd // 2)))
        rows = torch.arange(h, device=x.device).repeat_interleave(w).repeat(t).float()
        cols = torch.arange(w, device=x.device).repeat(h * t).float()
        ang = torch.cat([rows[:, None] * inv[None], cols[:, None] * inv[None]], -1)          # [N, hd / 2]
        ang = torch.cat([ang, ang], -1)
        cos, sin = ang.cos()[:, None], ang.sin()[:, None]
        for b in self.blocks:
            x = b(x, cos, sin, [x.shape[0]])
        return self.post_layernorm(x)


class KeyeProjector(nn.Module):
    def __init__(self, vision_dim: int, text_dim: int):
        super().__init__()
        self.pre_norm = nn.LayerNorm(vision_dim, eps=1e-5)
        self.linear_1 = nn.Linear(4 * vision_dim, 4 * vision_dim)
        self.linear_2 = nn.Linear(4 * vision_dim, text_dim)

    def forward(self, x: torch.Tensor, grid: tuple[int, int, int]) -> torch.Tensor:
        t, h, w = grid
        x = self.pre_norm(x).view(t, h // 2, 2, w // 2, 2, -1).permute(0, 1, 3, 2, 4, 5).reshape(t * h * w // 4, -1)
        return self.linear_2(F.gelu(self.linear_1(x)))


class _Qwen3Layer(LlamaDecoderLayer):
    def _attention_extras(self, config, layer_idx: int) -> dict:
        return {"qk_norm": True, "bias": bool(getattr(config, "attention_bias", False)), "o_bias": False}


class _Qwen3Model(LlamaModel):
    layer_cls = _Qwen3Layer


class _Qwen3Decoder(LlamaForCausalLM):
    model_cls = _Qwen3Model


class KeyeForConditionalGeneration(Qwen2VLForConditionalGeneration):
    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        nn.Module.__init__(self)
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vc = hf.vision_config if isinstance(hf.vision_config, dict) else hf.vision_config.to_dict()
        self.merge = 2
        self.image_token_id = int(getattr(hf, "image_token_id", 151655))
        self.visual = nn.Module()
        self.visual.vision_model = KeyeVisionTower(vc)
        self.mlp_AR = KeyeProjector(int(vc["hidden_size"]), int(hf.hidden_size))
        self.language_model = _Qwen3Decoder(vllm_config, prefix=maybe_prefix(prefix, "language_model"))
        self.rotary = self.language_model.model.layers[self.language_model.model.start_layer].self_attn.rotary_emb

    def encode_mm_items(self, items: list) -> list[torch.Tensor]:
        dev = self.mlp_AR.linear_1.weight.device
        out = []
        for px, g in items:
            grid = tuple(int(v) for v in g.tolist())
            out.append(self.mlp_AR(self.visual.vision_model(px.to(dev), grid), grid))
        return out

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        dec, vis, qkv = [], [], {}
        for n, t in weights:
            if n.startswith("visual.vision_model.head.") or "packing_position_embedding" in n or "inv_freq" in n:
                continue                          # pooling head and packed-position table: not on this path
            if n.startswith("visual."):
                n = n.replace(".embeddings.", ".").replace(".encoder.layers.", ".blocks.")
                n = n.replace(".layer_norm1.", ".norm1.").replace(".layer_norm2.", ".norm2.")
                n = n.replace(".self_attn.out_proj.", ".attn.proj.")
                m = [p for p in ("q_proj", "k_proj", "v_proj") if f".self_attn.{p}." in n]
                if m:
                    qkv.setdefault(n.replace(f".self_attn.{m[0]}.", ".attn.qkv."), {})[m[0]] = t
                else:
                    vis.append((n, t))
            elif n.startswith("mlp_AR."):
                vis.append((n, t))
            else:
                dec.append(("language_model." + n, t))
        for n, parts in qkv.items():
            vis.append((n, torch.cat([parts["q_proj"], parts["k_proj"], parts["v_proj"]], 0)))
        loaded = load_weights_with_mapping(self, dec, LlamaForCausalLM.stacked_params_mapping,
                                           tie_lm_head=self.language_model.tie)
        loaded |= load_weights_with_mapping(self, vis, [])
        return loaded


def keye_processor(vllm_config):
    """Images and video frames to row-major patch matrices [t * h * w, C * p * p] (one frame per temporal step, sides
    rounded to multiples of 2 patches, mean / std 0.5); the image / video token becomes t * h * w / 4 placeholders."""
    import numpy as np
    from ..multimodal import PatchGridProcessor, smart_resize
    hf = vllm_config.model_config.hf_config
    vc = hf.vision_config if isinstance(hf.vision_config, dict) else hf.vision_config.to_dict()
    kw = vllm_config.model_config.mm_processor_kwargs or {}

    class _Keye(PatchGridProcessor):
        def _frames(self, t: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:      # [F, 3, H, W] in 0..255
            H, W = t.shape[-2:]
            h, w = smart_resize(H, W, self.patch * self.merge, self.min_pixels, self.max_pixels)
            if (h, w) != (H, W):
                t = F.interpolate(t, size=(h, w), mode="bicubic", align_corners=False, antialias=True)
            t = (t.clamp(0, 255) / 255.0 - 0.5) / 0.5
            p, gh, gw = self.patch, h // self.patch, w // self.patch
            x = t.view(-1, 3, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5)
            return x.reshape(-1, 3 * p * p).contiguous(), torch.tensor([t.shape[0], gh, gw])

        def patches(self, img):
            if isinstance(img, dict):
                return torch.as_tensor(img["pixel_values"]).float().flatten(1), \
                    torch.as_tensor(img["image_grid_thw"]).long().view(3)
            a = np.asarray(img.convert("RGB")) if hasattr(img, "convert") else np.asarray(img)
            return self._frames(torch.as_tensor(a).permute(2, 0, 1).float()[None])

        def video_patches(self, video):
            if isinstance(video, dict):
                key = "pixel_values_videos" if "pixel_values_videos" in video else "pixel_values"
                return torch.as_tensor(video[key]).float().flatten(1), \
                    torch.as_tensor(video["video_grid_thw"]).long().view(3)
            return self._frames(torch.as_tensor(np.asarray(video)).permute(0, 3, 1, 2).float())

    return _Keye(int(getattr(hf, "image_token_id", 151655)), int(vc.get("patch_size", 14)), 2, 1,
                 int(kw.get("min_pixels", 28 * 28 * 4)), int(kw.get("max_pixels", 28 * 28 * 1280)),
                 video_token_id=int(getattr(hf, "video_token_id", 151656)))
