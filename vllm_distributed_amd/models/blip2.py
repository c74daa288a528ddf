"""BLIP-2: a frozen ViT (class token, fused qkv), a Q-Former (a small BERT whose learned query vectors attend to
themselves and, every ``cross_attention_frequency`` layers, to the image tokens), one linear projection, and a
decoder-only language model (OPT in the released checkpoints).  Every image costs ``num_query_tokens`` prompt
positions whatever its size.

The towers run once per image outside the paged-attention path, in plain PyTorch ops.
Reference: vllm/model_executor/models/blip2.py (Blip2QFormerMultiHeadAttention :80-196, Blip2QFormerLayer
:293-370, Blip2QFormerModel :419-458, Blip2ForConditionalGeneration :492-720), vllm/model_executor/models/blip.py
(BlipVisionModel :250-320).
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


class _VisionLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        d, eps = int(c.hidden_size), float(getattr(c, "layer_norm_eps", 1e-6))
        self.heads = int(c.num_attention_heads)
        self.self_attn = nn.Module()
        self.self_attn.qkv = nn.Linear(d, 3 * d, bias=bool(getattr(c, "qkv_bias", True)))
        self.self_attn.projection = nn.Linear(d, d)
        self.layer_norm1, self.layer_norm2 = nn.LayerNorm(d, eps=eps), nn.LayerNorm(d, eps=eps)
        self.mlp = nn.Module()
        self.mlp.fc1, self.mlp.fc2 = nn.Linear(d, int(c.intermediate_size)), nn.Linear(int(c.intermediate_size), d)
        self.act = get_act_fn(getattr(c, "hidden_act", "gelu"))

    def forward(self, x):
        b, n, d = x.shape
        q, k, v = self.self_attn.qkv(self.layer_norm1(x)).view(b, n, 3, self.heads, d // self.heads) \
            .permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b, n, d)
        x = x + self.self_attn.projection(o)
        return x + self.mlp.fc2(self.act(self.mlp.fc1(self.layer_norm2(x))))


class Blip2VisionModel(nn.Module):
    def __init__(self, c):
        super().__init__()
        d, p, s = int(c.hidden_size), int(c.patch_size), int(c.image_size)
        self.embeddings = nn.Module()
        self.embeddings.class_embedding = nn.Parameter(torch.zeros(1, 1, d))
        self.embeddings.patch_embedding = nn.Conv2d(3, d, p, p)
        self.embeddings.position_embedding = nn.Parameter(torch.zeros(1, (s // p) ** 2 + 1, d))
        self.encoder = nn.Module()
        self.encoder.layers = nn.ModuleList(_VisionLayer(c) for _ in range(int(c.num_hidden_layers)))
        self.post_layernorm = nn.LayerNorm(d, eps=float(getattr(c, "layer_norm_eps", 1e-6)))

    def forward(self, pixel_values):
        e = self.embeddings
        x = e.patch_embedding(pixel_values.to(e.patch_embedding.weight.dtype)).flatten(2).transpose(1, 2)
        x = torch.cat([e.class_embedding.expand(x.shape[0], -1, -1), x], 1) + e.position_embedding[:, : x.shape[1] + 1]
        for layer in self.encoder.layers:
            x = layer(x)
        return self.post_layernorm(x)


class _QAttention(nn.Module):
    """BERT attention block: projections, softmax attention, output dense, LayerNorm(out + input)."""

    def __init__(self, d: int, heads: int, kv_dim: int, eps: float):
        super().__init__()
        self.heads = heads
        self.attention = nn.Module()
        self.attention.query = nn.Linear(d, d)
        self.attention.key, self.attention.value = nn.Linear(kv_dim, d), nn.Linear(kv_dim, d)
        self.output = nn.Module()
        self.output.dense, self.output.LayerNorm = nn.Linear(d, d), nn.LayerNorm(d, eps=eps)

    def forward(self, x, kv=None):
        kv = x if kv is None else kv
        b, n, d = x.shape
        a = self.attention
        q = a.query(x).view(b, n, self.heads, -1).transpose(1, 2)
        k = a.key(kv).view(b, kv.shape[1], self.heads, -1).transpose(1, 2)
        v = a.value(kv).view(b, kv.shape[1], self.heads, -1).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b, n, d)
        return self.output.LayerNorm(self.output.dense(o) + x)


class _QLayer(nn.Module):
    def __init__(self, c, idx: int):
        super().__init__()
        d, eps = int(c.hidden_size), float(getattr(c, "layer_norm_eps", 1e-12))
        heads = int(c.num_attention_heads)
        self.attention = _QAttention(d, heads, d, eps)
        if idx % int(getattr(c, "cross_attention_frequency", 2)) == 0:
            self.crossattention = _QAttention(d, heads, int(c.encoder_hidden_size), eps)
        self.intermediate_query = nn.Module()
        self.intermediate_query.dense = nn.Linear(d, int(c.intermediate_size))
        self.output_query = nn.Module()
        self.output_query.dense = nn.Linear(int(c.intermediate_size), d)
        self.output_query.LayerNorm = nn.LayerNorm(d, eps=eps)
        self.act = get_act_fn(getattr(c, "hidden_act", "gelu"))

    def forward(self, x, image):
        x = self.attention(x)
        if hasattr(self, "crossattention"):
            x = self.crossattention(x, image)
        return self.output_query.LayerNorm(self.output_query.dense(self.act(self.intermediate_query.dense(x))) + x)


class Blip2QFormer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layernorm = nn.LayerNorm(int(c.hidden_size), eps=float(getattr(c, "layer_norm_eps", 1e-12)))
        self.encoder = nn.Module()
        self.encoder.layer = nn.ModuleList(_QLayer(c, i) for i in range(int(c.num_hidden_layers)))

    def forward(self, queries, image):
        x = self.layernorm(queries)
        for layer in self.encoder.layer:
            x = layer(x, image)
        return x


class Blip2ForConditionalGeneration(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vc, qc, tc = _cfg(hf.vision_config), _cfg(hf.qformer_config), _cfg(hf.text_config)
        self.image_token_id = int(getattr(hf, "image_token_index", None) or getattr(hf, "image_token_id", 50265))
        self.vision_model = Blip2VisionModel(vc)
        self.query_tokens = nn.Parameter(torch.zeros(1, int(hf.num_query_tokens), int(qc.hidden_size)))
        self.qformer = Blip2QFormer(qc)
        self.language_projection = nn.Linear(int(qc.hidden_size), int(tc.hidden_size))
        if not getattr(tc, "architectures", None) and getattr(tc, "model_type", "") == "opt":
            tc.architectures = ["OPTForCausalLM"]
        self.language_model = build_language_model(vllm_config, tc, maybe_prefix(prefix, "language_model"))

    def encode_images(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """[n, 3, S, S] -> [n, num_query_tokens, H_text]"""
        image = self.vision_model(pixel_values)
        q = self.qformer(self.query_tokens.expand(image.shape[0], -1, -1).to(image.dtype), image)
        return self.language_projection(q)

    def get_input_embeddings(self, input_ids, mm_embeds=None, mm_mask=None):
        ids = input_ids
        if self.image_token_id >= self.language_model.config.vocab_size:
            ids = torch.where(input_ids == self.image_token_id, 0, input_ids)
        x = self.language_model.get_input_embeddings(ids)
        if mm_embeds is not None and mm_embeds.numel():
            x[mm_mask] = mm_embeds.to(x.dtype)
        return x

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        return self.language_model(input_ids, positions, intermediate_tensors, inputs_embeds)

    def compute_logits(self, hidden_states):
        return self.language_model.compute_logits(hidden_states)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        dec, own, split_bias = [], [], {}
        for n, t in weights:
            if n.startswith("model."):                       # transformers >= 5 nests everything but the head
                n = n[len("model."):]
            if n.startswith("language_model."):
                dec.append((n[len("language_model."):], t))
            elif n.startswith("lm_head."):
                dec.append((n, t))
            elif n.endswith((".self_attn.q_bias", ".self_attn.v_bias")):
                # the original checkpoints store q and v biases only (k has none)
                split_bias.setdefault(n.rsplit(".", 1)[0], {})[n[-6]] = t
            elif "embeddings.position_ids" not in n:
                own.append((n, t))
        for base, parts in split_bias.items():
            q, v = parts["q"], parts["v"]
            own.append((base + ".qkv.bias", torch.cat([q, torch.zeros_like(q), v])))
        loaded = {"language_model." + n for n in self.language_model.load_weights(dec)}
        loaded |= load_weights_with_mapping(self, own, [])
        return loaded


def blip2_processor(vllm_config):
    """One placeholder (or an expanded run) per image -> ``num_query_tokens`` positions; pixels: plain bicubic
    resize to the square input, CLIP mean / std."""
    from ..multimodal import ImageTokenProcessor
    hf = vllm_config.model_config.hf_config
    vc = _cfg(hf.vision_config)
    tok = int(getattr(hf, "image_token_index", None) or getattr(hf, "image_token_id", 50265))
    size = int(vc.image_size)

    class _P(ImageTokenProcessor):
        def __call__(self, ids, mm_data):
            imgs = mm_data.get("image")
            if imgs is not None:
                imgs = imgs if isinstance(imgs, (list, tuple)) else [imgs]
                mm_data = dict(mm_data, image=[self._pre(i) for i in imgs])
            return super().__call__(ids, mm_data)

        def _pre(self, img):
            if isinstance(img, torch.Tensor) and img.dtype.is_floating_point:
                return img
            import numpy as np
            from ..multimodal import CLIP_MEAN, CLIP_STD
            a = np.asarray(img.convert("RGB")) if hasattr(img, "convert") else np.asarray(img)
            t = torch.as_tensor(a).permute(2, 0, 1).float()[None]
            t = F.interpolate(t, size=(size, size), mode="bicubic", align_corners=False, antialias=True)
            t = t[0].clamp(0, 255) / 255.0
            return (t - torch.tensor(CLIP_MEAN).view(3, 1, 1)) / torch.tensor(CLIP_STD).view(3, 1, 1)

    return _P(tok, int(hf.num_query_tokens), size)
