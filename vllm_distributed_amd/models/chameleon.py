"""Chameleon: an early-fusion model.  Images are not embedded by a vision tower: a VQ-VAE encoder turns a 512 x 512
image into a 32 x 32 grid of codebook indices, a fixed table maps every index to an ordinary vocabulary id, and those
ids go through the decoder's own embedding table.  The decoder is a Llama block with a per-head LayerNorm on q and k
(scale and shift per head) in front of the rotary embedding; the 30B checkpoints put the block norms BEHIND attention
and MLP ("swin" placement).  Vocabulary ids of image codes are masked out of the logits: the engine generates text.

Reference: vllm/model_executor/models/chameleon.py (ChameleonLayerNorm :219-232, ChameleonAttention :265-374,
ChameleonSwinDecoderLayer :441-504, ChameleonVQVAE :542-792, ChameleonImageVocabularyMapping :796-841,
logits mask :1033-1044).
"""
from __future__ import annotations

from typing import Iterable

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from .llama import LlamaDecoderLayer, LlamaForCausalLM, LlamaModel
from .llama_variants import _NoResidualFinalNorm
from .llava import _cfg
from .utils import load_weights_with_mapping, maybe_prefix


# ---- decoder -----------------------------------------------------------------------------------------------
class ChameleonDecoderLayer(LlamaDecoderLayer):
    def _attention_extras(self, config, layer_idx: int) -> dict:
        return {"qk_norm": "layernorm"}


class ChameleonSwinDecoderLayer(ChameleonDecoderLayer):
    """x + norm(attn(x)), then x + norm(mlp(x)): ``input_layernorm`` is the norm behind the attention."""
    supports_sequence_parallel = False

    def forward(self, positions, hidden_states, residual):
        h = hidden_states + self.input_layernorm(self.self_attn(positions, hidden_states))
        return h + self.post_attention_layernorm(self.mlp(h)), None


class ChameleonModel(LlamaModel):
    layer_cls = ChameleonDecoderLayer


class ChameleonSwinModel(_NoResidualFinalNorm):
    layer_cls = ChameleonSwinDecoderLayer


class _ChameleonText(LlamaForCausalLM):
    model_cls = ChameleonModel


class _ChameleonSwinText(LlamaForCausalLM):
    model_cls = ChameleonSwinModel
    supports_pp = False


# ---- VQ-VAE encoder -----------------------------------------------------------------------------------------
def _swish_(x):
    return x * torch.sigmoid(x)


class _ResBlock(nn.Module):
    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1)
        self.norm2 = nn.GroupNorm(32, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x):
        h = self.conv1(_swish_(self.norm1(x)))
        h = self.conv2(_swish_(self.norm2(h)))
        return (self.nin_shortcut(x) if hasattr(self, "nin_shortcut") else x) + h


class _AttnBlock(nn.Module):
    """Single-head attention over the pixels of a feature map (1 x 1 convolutions as projections)."""

    def __init__(self, c: int):
        super().__init__()
        self.norm = nn.GroupNorm(32, c, eps=1e-6)
        self.q, self.k, self.v, self.proj_out = (nn.Conv2d(c, c, 1) for _ in range(4))

    def forward(self, x):
        b, c, hh, ww = x.shape
        n = self.norm(x)
        q, k, v = (f(n).reshape(b, c, hh * ww).transpose(1, 2)[:, None] for f in (self.q, self.k, self.v))
        o = F.scaled_dot_product_attention(q, k, v)[:, 0].transpose(1, 2).reshape(b, c, hh, ww)
        return x + self.proj_out(o)


class _Down(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 2, 0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))       # the padding is one-sided


class VQVAEEncoder(nn.Module):
    def __init__(self, vq):
        super().__init__()
        base, mult, nres = int(vq.base_channels), tuple(vq.channel_multiplier), int(vq.num_res_blocks)
        attn_res = vq.attn_resolutions if getattr(vq, "attn_type", "vanilla") == "vanilla" else None
        self.conv_in = nn.Conv2d(int(vq.in_channels), base, 3, 1, 1)
        res, cin = int(vq.resolution), base
        self.down = nn.ModuleList()
        for lvl, m in enumerate(mult):
            d = nn.Module()
            d.block, d.attn = nn.ModuleList(), nn.ModuleList()
            for _ in range(nres):
                d.block.append(_ResBlock(cin, base * m))
                cin = base * m
                if attn_res and res in attn_res:
                    d.attn.append(_AttnBlock(cin))
            if lvl != len(mult) - 1:
                d.downsample = _Down(cin)
                res //= 2
            self.down.append(d)
        self.mid = nn.Module()
        self.mid.block_1 = _ResBlock(cin, cin)
        self.mid.attn_1 = _AttnBlock(cin) if getattr(vq, "attn_type", "vanilla") == "vanilla" else nn.Identity()
        self.mid.block_2 = _ResBlock(cin, cin)
        self.norm_out = nn.GroupNorm(32, cin, eps=1e-6)
        lat = int(vq.latent_channels)
        self.conv_out = nn.Conv2d(cin, 2 * lat if getattr(vq, "double_latent", False) else lat, 3, 1, 1)

    def forward(self, x):
        h = self.conv_in(x)
        for d in self.down:
            for i, blk in enumerate(d.block):
                h = blk(h)
                if len(d.attn):
                    h = d.attn[i](h)
            if hasattr(d, "downsample"):
                h = d.downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        return self.conv_out(_swish_(self.norm_out(h)))


class VQVAE(nn.Module):
    """Encoder half only: pixels -> codebook indices (nearest code in squared distance)."""

    def __init__(self, vq):
        super().__init__()
        self.encoder = VQVAEEncoder(vq)
        self.quant_conv = nn.Conv2d(int(vq.latent_channels), int(vq.embed_dim), 1)
        self.quantize = nn.Module()
        self.quantize.embedding = nn.Embedding(int(vq.num_embeddings), int(vq.embed_dim))

    def forward(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """[n, 3, S, S] -> [n, (S / 2^(levels-1))^2] code indices, row-major."""
        z = self.quant_conv(self.encoder(pixel_values)).permute(0, 2, 3, 1)
        n = z.shape[0]
        z = z.reshape(-1, z.shape[-1]).float()
        e = self.quantize.embedding.weight.float()
        d = (z * z).sum(1, keepdim=True) + (e * e).sum(1) - 2.0 * z @ e.t()
        return d.argmin(1).view(n, -1)


def image_vocabulary(vocab_map: dict) -> tuple[list[int], dict[int, int]]:
    """(sorted vocabulary ids of the image codes, code index -> vocabulary id).  A code's vocabulary entry is
    named ``IMGIMG`` + its decimal index spelled with A..J for 0..9 + ``Z``."""
    digit = {chr(ord("A") + i): str(i) for i in range(10)}
    ids = sorted(v for k, v in vocab_map.items() if k.startswith("IMGIMG"))
    name = {v: k for k, v in vocab_map.items()}
    img2bpe = {int("".join(digit.get(c, c) for c in name[t][len("IMGIMG"):-1])): t for t in ids}
    return ids, img2bpe


# ---- the model ----------------------------------------------------------------------------------------------
class ChameleonForConditionalGeneration(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        swin = bool(getattr(hf, "swin_norm", False))
        self.language_model = (_ChameleonSwinText if swin else _ChameleonText)(
            vllm_config, prefix=maybe_prefix(prefix, "language_model"))
        vq = _cfg(hf.vq_config)
        self.vqmodel = VQVAE(vq)
        ids, img2bpe = image_vocabulary(dict(hf.vocabulary_map))
        table = torch.zeros(max(img2bpe) + 1 if img2bpe else 1, dtype=torch.long)
        for k, v in img2bpe.items():
            table[k] = v
        self.register_buffer("img2bpe", table, persistent=False)
        self.register_buffer("image_vocab_ids", torch.tensor(ids, dtype=torch.long), persistent=False)

    def image_tokens(self, pixel_values: torch.Tensor) -> torch.Tensor:
        return self.img2bpe[self.vqmodel(pixel_values)]

    def encode_images(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """[n, 3, S, S] -> [n, codes per image, H]: the text embeddings of the image's vocabulary ids."""
        return self.language_model.get_input_embeddings(self.image_tokens(pixel_values))

    def get_input_embeddings(self, input_ids, mm_embeds=None, mm_mask=None):
        x = self.language_model.get_input_embeddings(input_ids)
        if mm_embeds is not None and mm_embeds.numel():
            x[mm_mask] = mm_embeds.to(x.dtype)
        return x

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        return self.language_model(input_ids, positions, intermediate_tensors, inputs_embeds)

    def compute_logits(self, hidden_states):
        logits = self.language_model.compute_logits(hidden_states)
        if logits is not None and self.image_vocab_ids.numel():
            logits[:, self.image_vocab_ids] = torch.finfo(logits.dtype).min
        return logits

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        dec, vis = [], []
        for n, t in weights:
            if n.startswith("model.vqmodel."):
                n = n[len("model."):]
                if ".post_quant_conv." in n or n.startswith("vqmodel.decoder."):
                    continue
                vis.append((n, t))
            elif n.startswith("model."):
                dec.append(("language_model." + n, t))
            elif n.startswith("lm_head."):
                dec.append(("language_model." + n, t))
        lm = self.language_model
        loaded = load_weights_with_mapping(self, dec, lm.stacked_params_mapping, tie_lm_head=lm.tie)
        loaded |= load_weights_with_mapping(self, vis, [])
        return loaded


def chameleon_processor(vllm_config):
    """One ``<image>`` placeholder (or an already expanded run) -> begin-of-image + one placeholder per code +
    end-of-image; pixels: shortest edge to the VQ resolution, centre crop, values in [-1, 1]."""
    from ..multimodal import PlaceholderRange, hash_item
    hf = vllm_config.model_config.hf_config
    vq = _cfg(hf.vq_config)
    size = int(vq.resolution)
    per = (size // 2 ** (len(vq.channel_multiplier) - 1)) ** 2
    vocab = dict(hf.vocabulary_map)
    img = int(getattr(hf, "image_token_id", None) or vocab.get("<image>"))
    boi, eoi = vocab.get("<racm3:break>"), vocab.get("<eoss>")

    def pixels(im):
        if isinstance(im, torch.Tensor) and im.dtype.is_floating_point:
            return im
        import numpy as np
        a = np.asarray(im.convert("RGB")) if hasattr(im, "convert") else np.asarray(im)
        t = torch.as_tensor(a).permute(2, 0, 1).float()[None]
        h, w = t.shape[-2:]
        s = size / min(h, w)
        nh, nw = max(size, round(h * s)), max(size, round(w * s))
        t = F.interpolate(t, size=(nh, nw), mode="bicubic", align_corners=False, antialias=True).clamp(0, 255)
        top, left = (nh - size) // 2, (nw - size) // 2
        return t[0, :, top:top + size, left:left + size] / 127.5 - 1.0

    def process(ids, mm_data):
        images = mm_data.get("image")
        if images is None:
            raise ValueError("multi_modal_data needs an 'image' entry")
        images = images if isinstance(images, (list, tuple)) else [images]
        out, ranges, inputs, hashes = [], [], [], []
        i, k = 0, 0
        while i < len(ids):
            if ids[i] != img:
                out.append(ids[i])
                i += 1
                continue
            j = i
            while j < len(ids) and ids[j] == img:
                j += 1
            if k >= len(images):
                raise ValueError("more image placeholders than images")
            framed = i > 0 and ids[i - 1] == boi          # a chat template may have written the frame already
            if boi is not None and not framed:
                out.append(boi)
            ranges.append(PlaceholderRange(offset=len(out), length=per))
            out.extend([img] * per)
            if eoi is not None and not (j < len(ids) and ids[j] == eoi):
                out.append(eoi)
            inputs.append(pixels(images[k]))
            hashes.append(hash_item(images[k]))
            k += 1
            i = j
        if k != len(images):
            raise ValueError(f"prompt has {k} image placeholders but {len(images)} images were given")
        return out, inputs, hashes, ranges

    return process
