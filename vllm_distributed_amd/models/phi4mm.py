"""Phi-4-multimodal: a Phi-4-mini decoder whose image and audio placeholder tokens take vectors from two towers.

Image: a SigLIP tower run NaViT-style over 448 x 448 crops (position ids bucketised over each crop's valid patch
rectangle, padded patches masked as keys), the state entering the last layer, 2 x 2 average pooling, then the "HD"
arrangement: the crops' pooled grids stitched into one image and cut to its valid part, a learned separator vector
closing every row, a learned vector between the stitched crops and the low-resolution global view, and a two-layer
GELU MLP.  Audio: 80-bin log-mel frames, mean / variance normalised, reduced 8x in time by strided (depth-wise
separable) convolutions, a conformer stack (half-step MLP, attention with a learned relative-position bias per head,
GLU point-wise + causal depth-wise convolution module, half-step MLP, norm) over windows of 500 frames, and a two-layer
GELU MLP -- one for speech-only prompts, another when the prompt also has an image.

The vision and speech LoRA adapters that ship beside the checkpoint are ordinary adapters for the engine
(``lora_request``); ``.base_layer.`` in the original checkpoint's names is dropped on load.  Both layouts load:
transformers' (``img_projection_up`` ...) and the original (``img_projection.0`` ..., ``glb_GN`` / ``sub_GN``,
``audio_projection.speech.0`` ...).  Checked token for token against transformers' Phi4MultimodalForCausalLM.
Reference: vllm/model_executor/models/phi4mm.py (Phi4MMImageEncoder :145-420, Phi4MMForCausalLM :878-1258),
phi4mm_audio.py (ConformerEncoder :920-1130, AudioEmbedding :1135-1233), vision_siglip_navit.py.
"""
from __future__ import annotations

import math
from typing import Iterable

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from .llama import LlamaForCausalLM
from .llama_family import Phi3ForCausalLM
from .utils import load_weights_with_mapping, maybe_prefix


def _cfg(d, **defaults):
    d = dict(d) if isinstance(d, dict) else d.to_dict()
    return {**defaults, **{k: v for k, v in d.items() if v is not None}}


# ---- vision --------------------------------------------------------------------------------------------------------
class _SiglipLayer(nn.Module):
    def __init__(self, d: int, heads: int, inter: int, eps: float):
        super().__init__()
        self.heads = heads
        a = self.self_attn = nn.Module()
        a.q_proj, a.k_proj, a.v_proj, a.out_proj = (nn.Linear(d, d) for _ in range(4))
        self.layer_norm1, self.layer_norm2 = nn.LayerNorm(d, eps=eps), nn.LayerNorm(d, eps=eps)
        self.mlp = nn.Module()
        self.mlp.fc1, self.mlp.fc2 = nn.Linear(d, inter), nn.Linear(inter, d)

    def forward(self, x, key_mask):                          # [B, N, d], [B, N] True = a real patch
        B, N, d = x.shape
        a, h = self.self_attn, self.layer_norm1(x)
        q, k, v = (p(h).view(B, N, self.heads, -1).transpose(1, 2) for p in (a.q_proj, a.k_proj, a.v_proj))
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=key_mask[:, None, None, :])
        x = x + a.out_proj(o.transpose(1, 2).reshape(B, N, d))
        return x + self.mlp.fc2(F.gelu(self.mlp.fc1(self.layer_norm2(x)), approximate="tanh"))


class Phi4MMVisionTower(nn.Module):
    def __init__(self, vc: dict):
        super().__init__()
        d, self.patch = int(vc["hidden_size"]), int(vc["patch_size"])
        self.side = int(vc["image_size"]) // self.patch
        self.embeddings = nn.Module()
        self.embeddings.patch_embedding = nn.Conv2d(int(vc.get("num_channels", 3)), d, self.patch, self.patch)
        self.embeddings.position_embedding = nn.Embedding(self.side ** 2, d)
        n, fl = int(vc["num_hidden_layers"]), int(vc.get("feature_layer", -2))
        keep = fl if fl >= 0 else n + fl + 1
        self.encoder = nn.Module()
        self.encoder.layers = nn.ModuleList(_SiglipLayer(d, int(vc["num_attention_heads"]), int(vc["intermediate_size"]),
                                                         float(vc.get("layer_norm_eps", 1e-6))) for _ in range(keep))

    def forward(self, pixels: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """[crops, 3, S, S], [crops, side, side] bool -> [crops, side^2, d]"""
        e = self.embeddings
        x = e.patch_embedding(pixels.to(e.patch_embedding.weight.dtype)).flatten(2).transpose(1, 2)
        B, s = x.shape[0], self.side
        bounds = torch.arange(1 / s, 1.0, 1 / s, device=x.device)
        nh, nw = mask[:, :, 0].sum(1), mask[:, 0, :].sum(1)
        idx = torch.arange(s, device=x.device, dtype=torch.float32)
        fh = (idx[None] / nh[:, None]).clamp(max=1.0 - 1e-6).to(pixels.dtype)
        fw = (idx[None] / nw[:, None]).clamp(max=1.0 - 1e-6).to(pixels.dtype)
        ids = torch.bucketize(fh, bounds, right=True)[:, :, None] * s + torch.bucketize(fw, bounds, right=True)[:, None, :]
        flat = mask.reshape(B, -1)
        ids = torch.where(flat, ids.reshape(B, -1), torch.zeros_like(ids.reshape(B, -1)))
        x = x + e.position_embedding(ids)
        for layer in self.encoder.layers:
            x = layer(x, flat)
        return x


class Phi4MMImageEmbedding(nn.Module):
    def __init__(self, vc: dict, hidden: int):
        super().__init__()
        d = int(vc["hidden_size"])
        self.crop, self.dim = int(vc.get("crop_size", 448)), d
        self.img_processor = Phi4MMVisionTower(vc)
        self.odd = self.img_processor.side % 2 == 1
        self.img_projection_up, self.img_projection_down = nn.Linear(d, hidden), nn.Linear(hidden, hidden)
        self.global_img_feature_extensor = nn.Parameter(torch.zeros(1, 1, d))
        self.sub_img_feature_extensor = nn.Parameter(torch.zeros(1, 1, 1, d))

    def forward(self, pixels: torch.Tensor, size: tuple[int, int], mask: torch.Tensor) -> torch.Tensor:
        """One image: pixels [1 + crops, 3, S, S] (global view first), its padded (height, width), the patch masks
        [1 + crops, side, side] -> [tokens, hidden]"""
        mask = mask.to(pixels.device).bool()
        f = self.img_processor(pixels, mask)
        s = self.img_processor.side
        f = f.view(-1, s, s, self.dim).permute(0, 3, 1, 2)
        if self.odd:
            f = F.pad(f, (0, 1, 0, 1), mode="reflect")
        f = F.avg_pool2d(f, 2, 2).permute(0, 2, 3, 1)                           # [1 + crops, b, b, d]
        b = f.shape[1]
        hr, wr = int(size[0]) // self.crop, int(size[1]) // self.crop
        sep = self.sub_img_feature_extensor[0]                                  # [1, 1, d]
        glob = torch.cat([f[0], sep.expand(b, 1, -1)], 1).reshape(-1, self.dim)
        sub = f[1:1 + hr * wr].reshape(hr, wr, b, b, -1).transpose(1, 2).reshape(hr * b, wr * b, -1)
        m = mask[1:1 + hr * wr, 0::2, 0::2].reshape(hr, wr, b, b).transpose(1, 2).reshape(hr * b, wr * b)
        uh, uw = int(m[:, 0].sum()), int(m[0, :].sum())
        sub = torch.cat([sub[:uh, :uw], sep.expand(uh, 1, -1)], 1).reshape(-1, self.dim)
        x = torch.cat([sub, self.global_img_feature_extensor[0], glob], 0)
        return self.img_projection_down(F.gelu(self.img_projection_up(x)))


# ---- audio ---------------------------------------------------------------------------------------------------------
def _swish(x):
    return x * torch.sigmoid(x)


class _AudioMLP(nn.Module):
    def __init__(self, d: int, inter: int):
        super().__init__()
        self.layer_norm = nn.LayerNorm(d)
        self.gate_up_proj, self.down_proj = nn.Linear(d, 2 * inter), nn.Linear(inter, d)

    def forward(self, x):
        up, gate = self.gate_up_proj(self.layer_norm(x)).chunk(2, -1)
        return self.down_proj(up * _swish(gate))


class _ConvModule(nn.Module):
    def __init__(self, ac: dict):
        super().__init__()
        d, self.k, ext = int(ac["hidden_size"]), int(ac["kernel_size"]), int(ac["ext_pw_out_channel"])
        self.ext = ext
        self.layer_norm = nn.LayerNorm(d)
        self.glu = nn.Module()
        self.glu.ext_pw_conv_1d = nn.Conv1d(d, 2 * ext, 1)
        self.glu.b1, self.glu.b2 = nn.Parameter(torch.zeros(1, ext, 1)), nn.Parameter(torch.zeros(1, ext, 1))
        mult = int(ac.get("depthwise_multiplier", 1))
        self.dw_sep_conv_1d = nn.Module()
        self.dw_sep_conv_1d.dw_conv = nn.Conv1d(d, d * mult, self.k, 1, padding=self.k - 1, groups=d)
        self.dw_sep_conv_1d.pw_conv = nn.Conv1d(d * mult, int(ac["depthwise_separable_out_channel"]), 1)
        self.ext_pw_conv_1d = nn.Conv1d(d, ext, 1)

    def forward(self, x):                                    # [B, T, d]
        h = self.glu.ext_pw_conv_1d(self.layer_norm(x).transpose(1, 2))
        h = (h[:, : self.ext] + self.glu.b1) * _swish(h[:, self.ext:] + self.glu.b2)
        h = self.dw_sep_conv_1d.pw_conv(self.dw_sep_conv_1d.dw_conv(h))
        if self.k > 1:
            h = h[:, :, : -(self.k - 1)]                     # the left-padded (causal) part
        return self.ext_pw_conv_1d(_swish(h)).transpose(1, 2)


class _ConformerLayer(nn.Module):
    def __init__(self, ac: dict):
        super().__init__()
        d, inter, self.heads = int(ac["hidden_size"]), int(ac["intermediate_size"]), int(ac["num_attention_heads"])
        self.feed_forward_in, self.feed_forward_out = _AudioMLP(d, inter), _AudioMLP(d, inter)
        a = self.self_attn = nn.Module()
        a.q_proj, a.k_proj, a.v_proj, a.o_proj = (nn.Linear(d, d) for _ in range(4))
        self.conv = _ConvModule(ac)
        self.layer_norm_att, self.layer_norm = nn.LayerNorm(d), nn.LayerNorm(d)

    def forward(self, x, bias):                              # [B, T, d], [1, heads, T, T]
        B, T, d = x.shape
        r = x + 0.5 * self.feed_forward_in(x)
        a, h = self.self_attn, self.layer_norm_att(r)
        q, k, v = (p(h).view(B, T, self.heads, -1).transpose(1, 2) for p in (a.q_proj, a.k_proj, a.v_proj))
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=bias.to(q.dtype))
        x = r + a.o_proj(o.transpose(1, 2).reshape(B, T, d))
        x = x + self.conv(x)
        return self.layer_norm(x + 0.5 * self.feed_forward_out(x))


class Phi4MMAudioEncoder(nn.Module):
    WINDOW = 500                                             # frames the stack sees at once

    def __init__(self, ac: dict):
        super().__init__()
        d, ch = int(ac["hidden_size"]), int(ac["nemo_conv_channels"])
        self.reduction = int(ac["time_reduction"])
        self.chunk, self.left = int(ac.get("chunk_size", -1)), int(ac.get("left_chunk", 18))
        self.encoder_embedding = nn.Module()
        self.encoder_embedding.global_mean = nn.Parameter(torch.zeros(int(ac["input_size"])), requires_grad=False)
        self.encoder_embedding.global_invstd = nn.Parameter(torch.ones(int(ac["input_size"])), requires_grad=False)
        layers: list[nn.Module] = [nn.Conv2d(1, ch, 3, 2, 1), nn.ReLU()]
        for _ in range(int(math.log2(self.reduction)) - 1):
            layers += [nn.Conv2d(ch, ch, 3, 2, 1, groups=ch), nn.Conv2d(ch, ch, 1), nn.ReLU()]
        self.embed = nn.Module()
        self.embed.conv = nn.Sequential(*layers)
        self.embed.out = nn.Linear(ch * int(ac["nemo_final_size"]), d)
        self.max_distance, self.symmetric = int(ac["bias_max_distance"]), bool(ac.get("bias_symmetric", False))
        self.relative_attention_bias_layer = nn.Module()
        self.relative_attention_bias_layer.bias_values = nn.Embedding(
            self.max_distance * (1 if self.symmetric else 2), int(ac["num_attention_heads"]))
        self.encoders = nn.ModuleList(_ConformerLayer(ac) for _ in range(int(ac["num_blocks"])))

    def _bias(self, T: int, device) -> torch.Tensor:
        pos = torch.arange(T, device=device)
        rel = (pos[None] - pos[:, None]).clamp(-self.max_distance, self.max_distance - 1)
        idx = rel.abs() if self.symmetric else rel + self.max_distance
        bias = self.relative_attention_bias_layer.bias_values(idx).permute(2, 0, 1)[None]
        # the chunk mask joins the bias as 0 / 1 ADDED to it (not as minus infinity): what the released weights were
        # trained with; with one window of un-chunked audio it is the constant 1
        if self.chunk > 0:
            cq, ck = pos[:, None] // self.chunk, pos[None] // self.chunk
            seen = (ck <= cq) & (ck >= cq - self.left)
        else:
            seen = torch.ones(T, T, dtype=torch.bool, device=device)
        return bias + seen.to(bias.dtype)[None, None]

    def forward(self, feats: torch.Tensor) -> torch.Tensor:
        """log-mel frames [T, bins] of one clip -> [ceil(T / reduction), d]"""
        e = self.encoder_embedding
        x = ((feats.to(self.embed.out.weight.dtype) - e.global_mean) * e.global_invstd)[None, None]
        x = self.embed.conv(x)                                                   # [1, ch, T', bins']
        x = self.embed.out(x.transpose(1, 2).reshape(1, x.shape[2], -1))
        T = x.shape[1]
        pad = (-T) % self.WINDOW if T > self.WINDOW else 0
        if T > self.WINDOW:                      # long clips: independent windows, zero frames filling the last one
            x = F.pad(x, (0, 0, 0, pad)).view(-1, self.WINDOW, x.shape[-1])
        bias = self._bias(x.shape[1], x.device)
        for layer in self.encoders:
            x = layer(x, bias)
        x = x.reshape(-1, x.shape[-1])
        return x[: x.shape[0] - pad] if pad else x


class Phi4MMAudioEmbedding(nn.Module):
    def __init__(self, ac: dict, hidden: int):
        super().__init__()
        d = int(ac["hidden_size"]) * int(ac.get("downsample_rate", 1))
        self.encoder = Phi4MMAudioEncoder(ac)
        self.up_proj_for_speech, self.down_proj_for_speech = nn.Linear(d, hidden), nn.Linear(hidden, hidden)
        self.up_proj_for_vision_speech, self.down_proj_for_vision_speech = nn.Linear(d, hidden), nn.Linear(hidden, hidden)

    def forward(self, feats: torch.Tensor, with_vision: bool) -> torch.Tensor:
        x = self.encoder(feats)
        up, down = (self.up_proj_for_vision_speech, self.down_proj_for_vision_speech) if with_vision else \
            (self.up_proj_for_speech, self.down_proj_for_speech)
        return down(F.gelu(up(x)))


# ---- model ---------------------------------------------------------------------------------------------------------
class Phi4MMForCausalLM(nn.Module):
    is_multimodal = True
    supports_pp = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vc, ac = self.tower_configs(hf)
        self.language_model = Phi3ForCausalLM(vllm_config, prefix=maybe_prefix(prefix, "language_model"))
        ext = self.embed_tokens_extend = nn.Module()
        ext.image_embed = Phi4MMImageEmbedding(vc, int(hf.hidden_size))
        ext.audio_embed = Phi4MMAudioEmbedding(ac, int(hf.hidden_size))

    @staticmethod
    def tower_configs(hf) -> tuple[dict, dict]:
        """transformers' layout carries ``vision_config`` / ``audio_config``; the original one describes the audio
        encoder under ``audio_processor.config`` and fixes the SigLIP-so400m geometry in code."""
        vision = dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16,
                      image_size=448, patch_size=14, crop_size=448, feature_layer=-2, layer_norm_eps=1e-6)
        audio = dict(hidden_size=1024, intermediate_size=1536, num_blocks=24, num_attention_heads=16, input_size=80,
                     nemo_conv_channels=1024, nemo_final_size=10, time_reduction=8, ext_pw_out_channel=1024,
                     depthwise_separable_out_channel=1024, depthwise_multiplier=1, kernel_size=3, bias_max_distance=500,
                     bias_symmetric=False, chunk_size=-1, left_chunk=18, downsample_rate=1)
        if hf.get("vision_config"):
            vision = _cfg(hf.vision_config, **vision)
        if hf.get("audio_config"):
            audio = _cfg(hf.audio_config, **audio)
        elif hf.get("audio_processor"):
            o = dict(hf.audio_processor.get("config", {}))
            rel = o.get("relative_attention_bias_args") or {}
            audio.update(hidden_size=o.get("attention_dim", 1024), intermediate_size=o.get("linear_units", 1536),
                         num_blocks=o.get("num_blocks", 24), num_attention_heads=o.get("attention_heads", 16),
                         input_size=o.get("input_size", 80), time_reduction=o.get("time_reduction", 8),
                         kernel_size=o.get("kernel_size", 3), chunk_size=o.get("chunk_size", -1),
                         left_chunk=o.get("left_chunk", 18), bias_max_distance=rel.get("t5_bias_max_distance", 500),
                         bias_symmetric=rel.get("t5_bias_symmetric", False),
                         nemo_conv_channels=(o.get("nemo_conv_settings") or {}).get("conv_channels", 1024))
            audio["ext_pw_out_channel"] = audio["depthwise_separable_out_channel"] = audio["hidden_size"]
        return vision, audio

    def encode_mm_items(self, items: list[list[torch.Tensor]]) -> list[torch.Tensor]:
        """items: images [pixels [1 + crops, 3, S, S], (height, width), masks [1 + crops, side, side]];
        audio [frames [T, bins], flag (1: the prompt also has an image)]"""
        ext, out = self.embed_tokens_extend, []
        dev = ext.image_embed.img_projection_up.weight.device
        for it in items:
            if len(it) == 3:
                out.append(ext.image_embed(it[0].to(dev), tuple(int(v) for v in it[1].tolist()), it[2]))
            else:
                out.append(ext.audio_embed(it[0].to(dev), bool(int(it[1]))))
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
        swaps = ((".img_projection.0.", ".img_projection_up."), (".img_projection.2.", ".img_projection_down."),
                 (".audio_projection.speech.0.", ".up_proj_for_speech."), (".audio_projection.speech.2.", ".down_proj_for_speech."),
                 (".audio_projection.vision.0.", ".up_proj_for_vision_speech."),
                 (".audio_projection.vision.2.", ".down_proj_for_vision_speech."),
                 (".image_embed.glb_GN", ".image_embed.global_img_feature_extensor"),
                 (".image_embed.sub_GN", ".image_embed.sub_img_feature_extensor"))
        keep = len(self.embed_tokens_extend.image_embed.img_processor.encoder.layers)
        dec, ext = [], []
        for n, t in weights:
            if ".lora_A." in n or ".lora_B." in n:
                continue                                    # adapters travel as LoRA requests
            n = n.replace(".base_layer.", ".")
            if n.startswith("model.embed_tokens_extend."):
                n = n[len("model."):]
                for a, b in swaps:
                    n = n.replace(a, b)
                n = n.replace(".img_processor.vision_model.", ".img_processor.")
                if ".img_processor.head." in n or ".img_processor.post_layernorm." in n:
                    continue                                # pooling head: not on this path
                if ".img_processor.encoder.layers." in n and int(n.split(".encoder.layers.")[1].split(".")[0]) >= keep:
                    continue                                # layers behind the tapped one
                ext.append((n, t))
            else:
                dec.append(("language_model." + n, t))
        loaded = load_weights_with_mapping(self, dec, [], tie_lm_head=self.language_model.tie)
        loaded |= load_weights_with_mapping(self, ext, [])
        return loaded


class Phi4MultimodalForCausalLM(Phi4MMForCausalLM):
    """transformers' name for the same model."""


# ---- preprocessing -------------------------------------------------------------------------------------------------
def hd_preprocess(img: torch.Tensor, size: int = 448, patch: int = 14, max_crops: int = 36):
    """[3, H, W] in 0..255 -> (pixels [1 + crops, 3, size, size], (height, width) of the padded canvas, patch masks
    [1 + crops, size / patch, size / patch]).  The canvas is the smallest grid of crops that holds the image (or, above
    ``max_crops``, the grid whose aspect ratio is closest); the image is scaled to touch two of its edges and padded
    with white at the right / bottom; the global view is the canvas scaled down to one crop.
    The resampling filters follow transformers' image processor (bilinear with antialiasing for the canvas, bicubic
    for the global view); torchvision, which it calls, is not available here: parity unpinned."""
    H, W = img.shape[-2:]
    wn, hn = math.ceil(W / size), math.ceil(H / size)
    if wn * hn > max_crops:
        ratios = sorted({(i, j) for i in range(1, max_crops + 1) for j in range(1, max_crops + 1) if i * j <= max_crops},
                        key=lambda r: r[0] * r[1])
        best, diff = (1, 1), float("inf")
        for r in ratios:
            d = abs(W / H - r[0] / r[1])
            if d < diff or (d == diff and W * H > 0.5 * size * size * r[0] * r[1]):
                best, diff = r, min(d, diff)
        wn, hn = best
    tw, th = size * wn, size * hn
    if tw / W < th / H:
        nw, nh = tw, int(H * (tw / W))
    else:
        nw, nh = int(W * (th / H)), th
    pw, ph = tw - nw, th - nh
    side = size // patch
    mask = torch.ones(side * hn, side * wn)
    if pw >= patch:
        mask[:, -(pw // patch):] = 0
    if ph >= patch:
        mask[-(ph // patch):, :] = 0
    if min(nw, nh) < 10:
        raise ValueError(f"the aspect ratio is very extreme {(nw, nh)}")
    r = F.interpolate(img[None].float(), size=(nh, nw), mode="bilinear", align_corners=False, antialias=True)[0]
    canvas = torch.full((3, th, tw), 255.0)
    canvas[:, :nh, :nw] = r.clamp(0, 255)
    canvas = (canvas / 255.0 - 0.5) / 0.5
    glob = F.interpolate(canvas[None], size=(size, size), mode="bicubic", align_corners=False)[0]
    crops = canvas.view(3, hn, size, wn, size).permute(1, 3, 0, 2, 4).reshape(hn * wn, 3, size, size)
    masks = mask.view(hn, side, wn, side).permute(0, 2, 1, 3).reshape(hn * wn, side, side)
    return (torch.cat([glob[None], crops], 0), (th, tw), torch.cat([torch.ones(1, side, side), masks], 0).bool())


def image_token_count(size: tuple[int, int], masks: torch.Tensor, crop: int = 448) -> int:
    """Rows of the stitched crops' valid part (each closed by a separator) + the vector between + the global view's
    rows with their separators."""
    hr, wr = size[0] // crop, size[1] // crop
    b = (masks.shape[-1] + 1) // 2
    m = masks[1:1 + hr * wr, 0::2, 0::2].reshape(hr, wr, b, b).transpose(1, 2).reshape(hr * b, wr * b)
    uh, uw = int(m[:, 0].sum()), int(m[0, :].sum())
    return uh * (uw + 1) + 1 + b * (b + 1)


def log_mel_frames(wave: torch.Tensor, sampling_rate: int = 16000) -> torch.Tensor:
    """Mono waveform in -1..1 at 16 kHz -> [frames, 80]: 25 ms Hamming windows every 10 ms, pre-emphasis 0.97 inside
    each window, 512-point power spectrum of the 16-bit-scaled signal, 80 Kaldi-style mel bins up to 7690 Hz, natural
    log floored at 0."""
    from transformers.audio_utils import mel_filter_bank
    if sampling_rate != 16000:
        raise ValueError("Phi-4-multimodal audio must be sampled at 16 kHz")
    win, hop, n_fft = 400, 160, 512
    frames = wave.double().unfold(-1, win, hop)
    prev = torch.roll(frames, 1, dims=-1)
    prev[:, 0] = prev[:, 1]
    frames = (frames - 0.97 * prev) * 32768
    spec = torch.fft.rfft(torch.hamming_window(win, periodic=False, dtype=torch.float64) * frames, n=n_fft, dim=1)
    power = spec.to(torch.complex64).abs() ** 2
    mel = torch.from_numpy(mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=80, min_frequency=0,
                                           max_frequency=7690, sampling_rate=16000, triangularize_in_mel_space=True,
                                           mel_scale="kaldi")).float()
    return torch.log(torch.clamp(power @ mel, min=1.0))


def phi4mm_processor(vllm_config):
    """Every image token (or run) stands for one image, every audio token for one clip; each becomes as many
    placeholders as its tower returns vectors.  Audio in a prompt that also has an image takes the second projector."""
    import numpy as np
    from ..multimodal import PlaceholderRange, hash_item
    hf = vllm_config.model_config.hf_config
    vc, ac = Phi4MMForCausalLM.tower_configs(hf)
    kw = vllm_config.model_config.mm_processor_kwargs or {}
    img_id = int(kw.get("image_token_id", vc.get("image_token_id", 200010)))
    aud_id = int(kw.get("audio_token_id", ac.get("audio_token_id", 200011)))
    size, patch, crop = int(vc["image_size"]), int(vc["patch_size"]), int(vc.get("crop_size", 448))
    max_crops = int(kw.get("dynamic_hd", 36))
    rate = int(ac["time_reduction"]) * int(ac.get("downsample_rate", 1))

    def image_item(img):
        if isinstance(img, dict):
            px, sz, mk = (torch.as_tensor(img[k]) for k in ("image_pixel_values", "image_sizes", "image_attention_mask"))
            return [px.float(), sz.long().view(2), mk.bool()]
        if not (isinstance(img, torch.Tensor) and img.dim() == 3 and img.shape[0] == 3):
            a = np.asarray(img.convert("RGB")) if hasattr(img, "convert") else np.asarray(img)
            img = torch.as_tensor(a).permute(2, 0, 1)
        px, sz, mk = hd_preprocess(img.float(), size, patch, max_crops)
        return [px, torch.tensor(sz), mk]

    def audio_item(clip, with_vision: bool):
        if isinstance(clip, dict):
            feats = torch.as_tensor(clip["audio_input_features"]).float()
        else:
            wave, sr = clip if isinstance(clip, (tuple, list)) else (clip, 16000)
            wave = torch.as_tensor(np.asarray(wave)).float()
            feats = log_mel_frames(wave.mean(-1) if wave.dim() == 2 else wave, int(sr))
        return [feats.reshape(-1, feats.shape[-1]), torch.tensor(int(with_vision))]

    def process(prompt: list[int], mm_data: dict):
        def as_list(x):
            return [] if x is None else (list(x) if isinstance(x, list) else [x])
        images, clips = as_list(mm_data.get("image")), as_list(mm_data.get("audio"))
        if not images and not clips:
            raise ValueError("multi_modal_data needs an 'image' or an 'audio' entry")
        out, items, hashes, ranges, i, ki, ka = [], [], [], [], 0, 0, 0
        while i < len(prompt):
            t = prompt[i]
            if t not in (img_id, aud_id):
                out.append(t)
                i += 1
                continue
            while i < len(prompt) and prompt[i] == t:
                i += 1
            if t == img_id:
                if ki >= len(images):
                    raise ValueError("more image placeholders than images")
                item = image_item(images[ki])
                n = image_token_count(tuple(int(v) for v in item[1].tolist()), item[2], crop)
                ki += 1
            else:
                if ka >= len(clips):
                    raise ValueError("more audio placeholders than clips")
                item = audio_item(clips[ka], bool(images))
                n = -(-item[0].shape[0] // rate)
                ka += 1
            ranges.append(PlaceholderRange(offset=len(out), length=n))
            out.extend([t] * n)
            items.append(item)
            hashes.append(hash_item(item[0]))
        if ki != len(images) or ka != len(clips):
            raise ValueError(f"prompt has {ki} image / {ka} audio placeholders but {len(images)} images / "
                             f"{len(clips)} clips were given")
        return out, items, hashes, ranges

    return process
