"""GLM-4V (``GLM4VForCausalLM``, the ChatGLM-layout checkpoints): an EVA2-CLIP tower whose blocks normalise the
OUTPUT of attention and MLP (x + LN(f(x))), a 2 x 2 strided convolution, a projector (linear -> LayerNorm -> GELU ->
gated MLP), learned begin / end-of-image vectors around the patch vectors, everything divided by
``scaling_factor``; then the ChatGLM decoder.  An image takes (image_size / patch / 2)^2 + 2 prompt positions: the
begin and end markers of the prompt receive the learned vectors.

No transformers class exists for this family: parity unpinned; the tower is tested against the same arithmetic
written out with plain tensor ops.
Reference: vllm/model_executor/models/glm4v.py (EVA2CLIPPatchEmbedding :52-88, EVA2CLIPTransformerLayer :172-203,
EVA2CLIPGLU :229-300, EVA2CLIPModel :303-354, GLM4VForCausalLM :530-650).
"""
from __future__ import annotations

from typing import Iterable

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from ..layers.basic import get_act_fn
from .renamed_decoders import ChatGLMModel
from .utils import load_weights_with_mapping


class _EvaLayer(nn.Module):
    def __init__(self, vc: dict):
        super().__init__()
        d, self.heads = int(vc["hidden_size"]), int(vc["num_heads"])
        eps = float(vc.get("layer_norm_eps", 1e-6))
        self.input_layernorm, self.post_attention_layernorm = nn.LayerNorm(d, eps=eps), nn.LayerNorm(d, eps=eps)
        self.attention = nn.Module()
        self.attention.query_key_value, self.attention.dense = nn.Linear(d, 3 * d), nn.Linear(d, d)
        self.mlp = nn.Module()
        self.mlp.fc1, self.mlp.fc2 = nn.Linear(d, int(vc["intermediate_size"])), nn.Linear(int(vc["intermediate_size"]), d)
        self.act = get_act_fn(vc.get("hidden_act", "gelu"))

    def forward(self, x):
        B, N, d = x.shape
        q, k, v = (t.view(B, N, self.heads, -1).transpose(1, 2) for t in self.attention.query_key_value(x).chunk(3, -1))
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, d)
        x = x + self.input_layernorm(self.attention.dense(a))
        return x + self.post_attention_layernorm(self.mlp.fc2(self.act(self.mlp.fc1(x))))


class EVA2CLIPModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        vc = dict(config.vision_config)
        d, H = int(vc["hidden_size"]), int(config.hidden_size)
        n = (int(vc["image_size"]) // int(vc["patch_size"])) ** 2
        pe = self.patch_embedding = nn.Module()
        pe.proj = nn.Conv2d(int(vc.get("in_channels", 3)), d, int(vc["patch_size"]), int(vc["patch_size"]))
        pe.cls_embedding = nn.Parameter(torch.zeros(1, d))
        pe.position_embedding = nn.Embedding(n + 1, d)
        self.transformer = nn.Module()
        self.transformer.layers = nn.ModuleList(_EvaLayer(vc) for _ in range(int(vc["num_hidden_layers"])))
        self.conv = nn.Conv2d(d, H, 2, 2)
        lp = self.linear_proj = nn.Module()
        ffn = int(config.ffn_hidden_size if hasattr(config, "ffn_hidden_size") else config.intermediate_size)
        lp.linear_proj = nn.Linear(H, H, bias=False)
        lp.norm1 = nn.LayerNorm(H)
        lp.gate_proj, lp.dense_h_to_4h = nn.Linear(H, ffn, bias=False), nn.Linear(H, ffn, bias=False)
        lp.dense_4h_to_h = nn.Linear(ffn, H, bias=False)
        self.boi, self.eoi = nn.Parameter(torch.zeros(1, 1, H)), nn.Parameter(torch.zeros(1, 1, H))
        self.scaling_factor = float(vc.get("scaling_factor", 1.0))

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        """[B, 3, S, S] -> [B, (S / patch / 2)^2 + 2, H]"""
        pe = self.patch_embedding
        x = pe.proj(images.to(pe.proj.weight.dtype)).flatten(2).transpose(1, 2)
        x = torch.cat([pe.cls_embedding.expand(x.shape[0], 1, -1), x], 1) + pe.position_embedding.weight[None]
        for layer in self.transformer.layers:
            x = layer(x)
        x = x[:, 1:]
        B, s, d = x.shape
        g = int(s ** 0.5)
        x = self.conv(x.view(B, g, g, d).permute(0, 3, 1, 2)).flatten(2).transpose(1, 2)
        lp = self.linear_proj
        x = F.gelu(lp.norm1(lp.linear_proj(x)))
        x = lp.dense_4h_to_h(F.silu(lp.gate_proj(x)) * lp.dense_h_to_4h(x))
        x = torch.cat([self.boi.expand(B, -1, -1), x, self.eoi.expand(B, -1, -1)], 1)
        return x / self.scaling_factor


class GLM4VForCausalLM(ChatGLMModel):
    is_multimodal = True
    supports_pp = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        self.vision = EVA2CLIPModel(self.config)

    def encode_images(self, pixel_values: torch.Tensor) -> torch.Tensor:
        return self.vision(pixel_values)

    def get_input_embeddings(self, input_ids, mm_embeds=None, mm_mask=None):
        x = super().get_input_embeddings(input_ids)
        if mm_embeds is not None and mm_embeds.numel():
            x[mm_mask] = mm_embeds.to(x.dtype)
        return x

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        text, vis = [], []
        for n, t in weights:
            if n.startswith("transformer.vision."):
                vis.append((n[len("transformer."):], t))
            else:
                text.append((n, t))
        loaded = super().load_weights(text)
        loaded |= load_weights_with_mapping(self, vis, [])
        return loaded


def glm4v_chat_processor(vllm_config):
    """begin marker + placeholder (one, or a run) + end marker -> begin + n placeholders + end, all n + 2 positions
    filled by the tower; a bare placeholder gets the markers added.  Pixels: bicubic resize to the square, CLIP mean /
    std."""
    from ..multimodal import CLIP_MEAN, CLIP_STD, PlaceholderRange, hash_item
    hf = vllm_config.model_config.hf_config
    vc = dict(hf.vision_config)
    size = int(vc["image_size"])
    n = (size // int(vc["patch_size"]) // 2) ** 2
    boi, eoi, pad = int(hf.boi_token_id), int(hf.eoi_token_id), int(hf.pad_token_id)

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
            framed = ids[i] == boi and i + 1 < len(ids) and ids[i + 1] == pad
            if ids[i] != pad and not framed:
                out.append(ids[i])
                i += 1
                continue
            j = i + 1 if framed else i
            while j < len(ids) and ids[j] == pad:
                j += 1
            if j < len(ids) and ids[j] == eoi:
                j += 1
            if k >= len(images):
                raise ValueError("more image placeholders than images")
            ranges.append(PlaceholderRange(offset=len(out), length=n + 2))
            out.extend([boi] + [pad] * n + [eoi])
            inputs.append(pixels(images[k]))
            hashes.append(hash_item(images[k]))
            k, i = k + 1, j
        if k != len(images):
            raise ValueError(f"prompt has {k} image placeholders but {len(images)} images were given")
        return out, inputs, hashes, ranges

    return process
