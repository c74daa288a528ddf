"""Florence-2: a DaViT image tower (stages of a strided convolution followed by blocks that alternate attention inside
spatial windows with attention ACROSS channel groups, each preceded by a depth-wise 3 x 3 convolution) in front of a
BART encoder-decoder.  The last stage's feature map gets learned row / column position embeddings and a sinusoidal
frame embedding, is reduced to the configured sources (mean over positions: one vector; mean over frames: one vector
per position), projected and normalised; those vectors followed by the embedded task prompt are the ENCODER input.
The decoder side is models/bart.py unchanged (paged self-attention, cross-attention from the per-request slot cache).

Both checkpoint layouts load: transformers' (``model.vision_tower`` / ``model.multi_modal_projector`` /
``model.language_model``) and the original one the reference reads (``vision_tower.convs.N.proj``, ``...fn.dw``,
``image_projection`` as a bare matrix, ``language_model.model``).  Checked token for token against transformers.
Reference: vllm/model_executor/models/florence2.py (DaViT :450-583, Florence2ForConditionalGeneration :877-1113,
_encode_image :996-1040).
"""
from __future__ import annotations

import math
from typing import Iterable

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from .bart import POS_OFFSET, BartForConditionalGeneration
from .utils import load_weights_with_mapping

TASK_PROMPTS = {          # the task tokens of the model card and the sentences they stand for
    "<OCR>": "What is the text in the image?", "<OCR_WITH_REGION>": "What is the text in the image, with regions?",
    "<CAPTION>": "What does the image describe?", "<DETAILED_CAPTION>": "Describe in detail what is shown in the image.",
    "<MORE_DETAILED_CAPTION>": "Describe with a paragraph what is shown in the image.",
    "<OD>": "Locate the objects with category name in the image.",
    "<DENSE_REGION_CAPTION>": "Locate the objects in the image, with their descriptions.",
    "<REGION_PROPOSAL>": "Locate the region proposals in the image.",
    "<CAPTION_TO_PHRASE_GROUNDING>": "Locate the phrases in the caption: {input}",
    "<REFERRING_EXPRESSION_SEGMENTATION>": "Locate {input} in the image with mask",
    "<REGION_TO_SEGMENTATION>": "What is the polygon mask of region {input}",
    "<OPEN_VOCABULARY_DETECTION>": "Locate {input} in the image.",
    "<REGION_TO_CATEGORY>": "What is the region {input}?", "<REGION_TO_DESCRIPTION>": "What does the region {input} describe?",
    "<REGION_TO_OCR>": "What text is in the region {input}?",
}


def expand_task_prompt(text: str) -> str:
    """``<OD>`` -> its sentence; ``<CAPTION_TO_PHRASE_GROUNDING> a cat`` -> the sentence with the rest filled in."""
    for token, sentence in TASK_PROMPTS.items():
        if token in text:
            return sentence.format(input=text.replace(token, "").strip()) if "{input}" in sentence else sentence
    return text


class _Mlp(nn.Module):
    def __init__(self, dim: int, ratio: float):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(dim, int(dim * ratio)), nn.Linear(int(dim * ratio), dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class _WindowAttention(nn.Module):
    def __init__(self, dim: int, heads: int, window: int, qkv_bias: bool):
        super().__init__()
        self.heads, self.window = heads, window
        self.qkv, self.proj = nn.Linear(dim, 3 * dim, bias=qkv_bias), nn.Linear(dim, dim)

    def forward(self, x):                                   # [B, H, W, C] -> [B, H * W, C]
        B, H, W, C = x.shape
        w = self.window
        x = F.pad(x, (0, 0, 0, (w - W % w) % w, 0, (w - H % w) % w))
        Hp, Wp = x.shape[1:3]
        x = x.view(B, Hp // w, w, Wp // w, w, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, w * w, C)
        q, k, v = self.qkv(x).view(-1, w * w, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        x = self.proj(F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(-1, w * w, C))
        x = x.view(B, Hp // w, Wp // w, w, w, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
        return x[:, :H, :W].reshape(B, H * W, C)


class _ChannelAttention(nn.Module):
    """Attention whose "tokens" are the channels of a group and whose feature axis is the positions."""

    def __init__(self, dim: int, groups: int, qkv_bias: bool):
        super().__init__()
        self.groups = groups
        self.qkv, self.proj = nn.Linear(dim, 3 * dim, bias=qkv_bias), nn.Linear(dim, dim)

    def forward(self, x):                                   # [B, N, C]
        B, N, C = x.shape
        q, k, v = self.qkv(x).view(B, N, 3, self.groups, C // self.groups).permute(2, 0, 3, 4, 1)   # [B, g, C/g, N]
        att = torch.softmax((q @ k.transpose(-1, -2)) * N ** -0.5, -1)
        return self.proj((att @ v).permute(0, 3, 1, 2).reshape(B, N, C))


class _MixBlock(nn.Module):
    """depth-wise conv (residual) -> norm -> mixer (residual) -> depth-wise conv (residual) -> norm -> MLP (residual)"""

    def __init__(self, dim: int, mixer: nn.Module, name: str, ratio: float):
        super().__init__()
        self.conv1 = nn.Conv2d(dim, dim, 3, padding=1, groups=dim)
        self.norm1 = nn.LayerNorm(dim)
        setattr(self, name, mixer)
        self.mixer_name, self.spatial = name, name == "window_attn"
        self.conv2 = nn.Conv2d(dim, dim, 3, padding=1, groups=dim)
        self.norm2 = nn.LayerNorm(dim)
        self.ffn = _Mlp(dim, ratio)

    def forward(self, x):                                   # [B, C, H, W]
        B, C, H, W = x.shape
        t = (self.conv1(x) + x).flatten(2).transpose(1, 2)
        h = self.norm1(t)
        t = t + getattr(self, self.mixer_name)(h.view(B, H, W, C) if self.spatial else h)
        x = t.transpose(1, 2).reshape(B, C, H, W)
        t = (self.conv2(x) + x).flatten(2).transpose(1, 2)
        t = t + self.ffn(self.norm2(t))
        return t.transpose(1, 2).reshape(B, C, H, W)


class _Stage(nn.Module):
    def __init__(self, dim, heads, groups, window, qkv_bias, ratio):
        super().__init__()
        self.spatial_block = _MixBlock(dim, _WindowAttention(dim, heads, window, qkv_bias), "window_attn", ratio)
        self.channel_block = _MixBlock(dim, _ChannelAttention(dim, groups, qkv_bias), "channel_attn", ratio)

    def forward(self, x):
        return self.channel_block(self.spatial_block(x))


class _ConvEmbed(nn.Module):
    def __init__(self, cin, cout, patch, stride, padding, pre_norm):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, patch, stride, padding)
        self.norm = nn.LayerNorm(cin if pre_norm else cout)
        self.pre_norm = pre_norm

    def forward(self, x):
        if self.pre_norm:
            x = self.norm(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        x = self.conv(x)
        return x if self.pre_norm else self.norm(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)


class DaViT(nn.Module):
    def __init__(self, vc: dict):
        super().__init__()
        dims = list(vc.get("embed_dim") or vc["dim_embed"])
        ratio, bias = float(vc.get("mlp_ratio", 4.0)), bool(vc.get("qkv_bias", True))
        self.convs = nn.ModuleList(
            _ConvEmbed(int(vc.get("in_channels", vc.get("in_chans", 3))) if i == 0 else dims[i - 1], dims[i],
                       vc["patch_size"][i], vc["patch_stride"][i], vc["patch_padding"][i], bool(vc["patch_prenorm"][i]))
            for i in range(len(dims)))
        self.blocks = nn.ModuleList(
            nn.ModuleList(_Stage(dims[i], vc["num_heads"][i], vc["num_groups"][i], int(vc["window_size"]), bias, ratio)
                          for _ in range(vc["depths"][i])) for i in range(len(dims)))

    def forward(self, x):                                   # [B, 3, H, W] -> [B, C, h, w]
        x = x.to(self.convs[0].conv.weight.dtype)
        for conv, blocks in zip(self.convs, self.blocks):
            x = conv(x)
            for b in blocks:
                x = b(x)
        return x


class Florence2Projector(nn.Module):
    def __init__(self, vc: dict, projection_dim: int):
        super().__init__()
        dim = list(vc.get("embed_dim") or vc["dim_embed"])[-1]
        npos = int(vc.get("max_position_embeddings") or (vc.get("image_pos_embed") or {}).get("max_pos_embeddings", 50))
        nt = int(vc.get("max_temporal_embeddings")
                 or (vc.get("visual_temporal_embedding") or {}).get("max_temporal_embeddings", 100))
        self.image_projection = nn.Linear(dim, projection_dim, bias=False)
        self.image_proj_norm = nn.LayerNorm(projection_dim)
        self.image_position_embed = nn.Module()
        self.image_position_embed.row_embeddings = nn.Embedding(npos, dim // 2)
        self.image_position_embed.column_embeddings = nn.Embedding(npos, dim - dim // 2)
        freq = torch.exp(torch.arange(dim // 2).float() * -(math.log(10000) / (dim // 2)))
        ang = torch.arange(nt).float()[:, None] * freq[None]
        table = torch.zeros(nt, dim)
        table[:, 0::2], table[:, 1::2] = ang.sin(), ang.cos()
        self.register_buffer("temporal", table, persistent=False)
        self.sources = list(vc.get("image_feature_source") or ["spatial_avg_pool", "temporal_avg_pool"])

    def forward(self, feat):                                # [B, C, h, w] (B frames of one image) -> [tokens, proj]
        B, C, h, w = feat.shape
        pe = self.image_position_embed
        pos = torch.cat([pe.column_embeddings.weight[:w][None].expand(h, -1, -1),
                         pe.row_embeddings.weight[:h][:, None].expand(-1, w, -1)], -1)          # [h, w, C]
        x = (feat.permute(0, 2, 3, 1) + pos.to(feat.dtype)).reshape(B, h * w, C)
        x = x + self.temporal[:B, None].to(x.dtype)
        parts = {"spatial_avg_pool": x.mean(1), "temporal_avg_pool": x.mean(0), "last_frame": x[-1]}
        x = torch.cat([parts[s] for s in self.sources], 0)
        return self.image_proj_norm(self.image_projection(x))


class Florence2ForConditionalGeneration(BartForConditionalGeneration):
    is_text_encoder_decoder = False         # a request carries an image; its prompt is the encoder's text

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        c = self.config
        vc = dict(c.vision_config)
        self.model.vision_tower = DaViT(vc)
        self.model.multi_modal_projector = Florence2Projector(vc, int(vc.get("projection_dim") or c.projection_dim))

    def encode_mm_items(self, items: list[list[torch.Tensor]]) -> list[torch.Tensor]:
        """items: [prompt token ids [L], pixels [3, H, W] (absent: text only)] -> encoder states [tokens + L, d]."""
        enc, out = self.model.encoder, []
        for it in items:
            ids = it[0].long().view(-1)
            x = self.model.shared(ids) * self.embed_scale
            if len(it) > 1:
                px = it[1] if it[1].dim() == 4 else it[1][None]
                img = self.model.multi_modal_projector(self.model.vision_tower(px.to(x.device)))
                x = torch.cat([img.to(x.dtype), x], 0)
            pos = torch.arange(x.shape[0], device=x.device) + POS_OFFSET
            x = enc.layernorm_embedding(x + enc.embed_positions(pos))
            for layer in enc.layers:
                x = layer(x)
            out.append(x)
        return out

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        swaps = ((".conv1.fn.dw.", ".conv1."), (".conv2.fn.dw.", ".conv2."), (".window_attn.norm.", ".norm1."),
                 (".window_attn.fn.", ".window_attn."), (".channel_attn.norm.", ".norm1."),
                 (".channel_attn.fn.", ".channel_attn."), (".ffn.norm.", ".norm2."), (".ffn.fn.net.", ".ffn."))
        text, vis = [], []
        for n, t in weights:
            if n.startswith("model.language_model."):                        # transformers' layout
                text.append(("model." + n[len("model.language_model."):], t))
            elif n.startswith("language_model."):                            # the original layout
                text.append((n[len("language_model."):], t))
            elif n == "lm_head.weight" or n.endswith("pos_idx_to_embed"):
                continue
            else:
                n = n[len("model."):] if n.startswith("model.") else n
                if n.startswith("vision_tower."):
                    for a, b in swaps:
                        n = n.replace(a, b)
                    if ".convs." in n:
                        n = n.replace(".proj.", ".conv.")
                elif n == "image_projection":
                    n, t = "multi_modal_projector.image_projection.weight", t.t()
                elif n.startswith("image_proj_norm."):
                    n = "multi_modal_projector." + n
                elif n.startswith("image_pos_embed."):
                    n = "multi_modal_projector.image_position_embed." + n[len("image_pos_embed."):]
                vis.append(("model." + n, t))
        loaded = super().load_weights(text)
        loaded |= load_weights_with_mapping(self, vis, [])
        return loaded


def florence2_processor(vllm_config):
    """The request's prompt tokens are the encoder's text; the decoder starts from its start token.  Pixels: bicubic
    resize to the square input, ImageNet mean / std."""
    from ..multimodal import PlaceholderRange, hash_item
    hf = vllm_config.model_config.hf_config
    size = int((dict(hf.vision_config)).get("image_size", 768))
    start = int(hf.get("decoder_start_token_id", 2) or 0)
    mean, std = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)

    def pixels(img):
        if isinstance(img, torch.Tensor) and img.dtype.is_floating_point:
            return img
        import numpy as np
        a = np.asarray(img.convert("RGB")) if hasattr(img, "convert") else np.asarray(img)
        t = torch.as_tensor(a).permute(2, 0, 1).float()[None]
        t = F.interpolate(t, size=(size, size), mode="bicubic", align_corners=False, antialias=True)[0]
        return (t.clamp(0, 255) / 255.0 - mean) / std

    def process(ids, mm_data):
        img = mm_data.get("image")
        if isinstance(img, (list, tuple)):
            if len(img) != 1:
                raise ValueError("Florence-2 takes one image per request")
            img = img[0]
        src = mm_data.get("encoder_token_ids")
        dec = list(ids) if src is not None else [start]
        text = torch.as_tensor(list(src if src is not None else ids), dtype=torch.int64)
        item = [text] + ([pixels(img)] if img is not None else [])
        return dec, [item], [hash_item("".join(hash_item(t) for t in item))], [PlaceholderRange(0, 1)]

    return process
