"""InternVL family: InternViT tower (class token, learned positions, layer-scaled residual branches, optional
query / key RMSNorm), pixel-shuffle down-sampling (four patches -> one token of four times the width), a
LayerNorm + two-layer projector, and an InternLM2 / Qwen2 / Llama decoder.  An image is cut into up to
``max_dynamic_patch`` tiles of the encoder's size on the grid whose aspect ratio is closest to its own, plus a
thumbnail; every tile yields (size / patch)^2 / 4 tokens, which replace a run of context tokens in the prompt.

Two checkpoint layouts are read:
  * the original one (``InternVLChatModel``, also SkyworkR1V / H2OVL which share it): ``vision_model.*`` with a
    fused ``attn.qkv``, ``mlp1.{0,1,3}``, ``language_model.*``, configuration keys ``llm_config`` /
    ``select_layer`` / ``downsample_ratio``;
  * the transformers one (``InternVLForConditionalGeneration``): ``model.vision_tower.*`` with separate q / k / v,
    ``model.multi_modal_projector.*``, ``model.language_model.*``.
Both load into the module tree below (named after the original layout).
Reference: vllm/model_executor/models/internvl.py (InternVLChatModel :1021-1420, dynamic tiling :60-180),
intern_vit.py, skyworkr1v.py, h2ovl.py (H2OVL's multi-scale tiling policy is not reproduced: its checkpoints
are served with InternVL's single-scale tiling).
"""
from __future__ import annotations

from typing import Any, Iterable

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import HFConfig, VllmConfig
from ..layers.basic import get_act_fn
from .llava import _cfg
from .llava_next import build_language_model
from .utils import load_weights_with_mapping, maybe_prefix

IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


class _RMS(nn.Module):
    def __init__(self, n, eps):
        super().__init__()
        self.weight, self.eps = nn.Parameter(torch.ones(n)), eps

    def forward(self, x):
        xf = x.float()
        return (self.weight * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps))).to(x.dtype)


def _norm(kind: str, n: int, eps: float) -> nn.Module:
    return _RMS(n, eps) if kind == "rms_norm" else nn.LayerNorm(n, eps=eps)


class InternVisionLayer(nn.Module):
    def __init__(self, H, heads, inter, qkv_bias, qk_norm, norm_type, eps, act, dummy_heads: int = 0):
        super().__init__()
        # dummy heads: extra all-zero heads some checkpoints (NVLM-D) carry so that the count divides 8
        self.hd = H // heads
        self.heads = heads + dummy_heads
        A = self.heads * self.hd
        self.attn = nn.Module()
        self.attn.qkv = nn.Linear(H, 3 * A, bias=qkv_bias)
        self.attn.proj = nn.Linear(A, H)
        if qk_norm:
            self.attn.q_norm, self.attn.k_norm = _RMS(A, eps), _RMS(A, eps)
        self.qk_norm = qk_norm
        self.norm1, self.norm2 = _norm(norm_type, H, eps), _norm(norm_type, H, eps)
        self.mlp = nn.Module()
        self.mlp.fc1, self.mlp.fc2 = nn.Linear(H, inter), nn.Linear(inter, H)
        self.act = get_act_fn(act)
        self.ls1, self.ls2 = nn.Parameter(torch.ones(H)), nn.Parameter(torch.ones(H))

    def forward(self, x):
        B, N, H = x.shape
        q, k, v = self.attn.qkv(self.norm1(x)).chunk(3, dim=-1)
        if self.qk_norm:
            q, k = self.attn.q_norm(q), self.attn.k_norm(k)
        q, k, v = (t.reshape(B, N, self.heads, self.hd).transpose(1, 2) for t in (q, k, v))
        o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, self.heads * self.hd)
        x = x + self.ls1 * self.attn.proj(o)
        return x + self.ls2 * self.mlp.fc2(self.act(self.mlp.fc1(self.norm2(x))))


class InternVisionModel(nn.Module):
    def __init__(self, vc: dict, select_layer: int, dummy_heads: int = 0):
        super().__init__()
        g = vc.get
        H = int(g("hidden_size"))
        size, patch = g("image_size", 448), g("patch_size", 14)
        self.S = int(size[0] if isinstance(size, (list, tuple)) else size)
        self.P = int(patch[0] if isinstance(patch, (list, tuple)) else patch)
        n = int(g("num_hidden_layers"))
        keep = n + select_layer + 1 if select_layer < 0 else select_layer
        self.embeddings = nn.Module()
        self.embeddings.class_embedding = nn.Parameter(torch.zeros(1, 1, H))
        self.embeddings.patch_embedding = nn.Conv2d(3, H, self.P, self.P)
        self.embeddings.position_embedding = nn.Parameter(torch.zeros(1, (self.S // self.P) ** 2 + 1, H))
        qkv_bias = bool(g("qkv_bias", g("attention_bias", False)))
        qk_norm = bool(g("qk_normalization", g("use_qk_norm", False)))
        self.encoder = nn.Module()
        self.encoder.layers = nn.ModuleList([
            InternVisionLayer(H, int(g("num_attention_heads")), int(g("intermediate_size")), qkv_bias, qk_norm,
                              str(g("norm_type", "layer_norm")), float(g("layer_norm_eps", 1e-6)),
                              str(g("hidden_act", "gelu")), dummy_heads) for _ in range(keep)])

    def forward(self, px):
        e = self.embeddings
        x = e.patch_embedding(px.to(e.patch_embedding.weight.dtype)).flatten(2).transpose(1, 2)
        x = torch.cat((e.class_embedding.expand(x.shape[0], -1, -1), x), dim=1) + e.position_embedding
        for layer in self.encoder.layers:
            x = layer(x)
        return x


def pixel_shuffle(x: torch.Tensor, scale: float, v2: bool = True) -> torch.Tensor:
    """[n, w, h, c] -> [n, w * scale, h * scale, c / scale^2]"""
    n, w, h, c = x.shape
    x = x.view(n, w, int(h * scale), int(c / scale)).permute(0, 2, 1, 3).contiguous()
    x = x.view(n, int(h * scale), int(w * scale), int(c / (scale * scale)))
    return x.permute(0, 2, 1, 3).contiguous() if v2 else x


class InternVLChatModel(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vc = _cfg(hf.vision_config).to_dict()
        tc = _cfg(getattr(hf, "llm_config", None) or hf.text_config)
        self.ratio = float(getattr(hf, "downsample_ratio", 0.5))
        self.ps_v2 = str(getattr(hf, "ps_version", "v2")) != "v1"
        select = int(getattr(hf, "select_layer", getattr(hf, "vision_feature_layer", -1)))
        self.vision_model = InternVisionModel(vc, select, self.vision_dummy_heads)
        Hv = int(vc["hidden_size"]) * int(1 / self.ratio) ** 2
        self.mlp1 = self._make_projector(Hv, tc)
        self.language_model = build_language_model(vllm_config, tc, maybe_prefix(prefix, "language_model"))

    vision_dummy_heads = 0

    def _make_projector(self, Hv: int, tc) -> nn.Sequential:
        return nn.Sequential(nn.LayerNorm(Hv), nn.Linear(Hv, tc.hidden_size), nn.GELU(),
                             nn.Linear(tc.hidden_size, tc.hidden_size))

    def encode_mm_items(self, items: list[list[torch.Tensor]]) -> list[torch.Tensor]:
        tiles = torch.cat([it[0] for it in items], dim=0)
        f = self.vision_model(tiles)[:, 1:]                        # class token dropped
        side = int(f.shape[1] ** 0.5)
        f = pixel_shuffle(f.reshape(f.shape[0], side, side, -1), self.ratio, self.ps_v2)
        f = self.mlp1(f.reshape(f.shape[0], -1, f.shape[-1]))      # [tiles, tokens per tile, H_text]
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
            if hasattr(self.language_model, "set_visual_rows"):      # decoders with visual experts
                self.language_model.set_visual_rows(mm_mask)
        return x

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        return self.language_model(input_ids, positions, intermediate_tensors, inputs_embeds)

    def compute_logits(self, hidden_states):
        return self.language_model.compute_logits(hidden_states)

    # ---- weights ------------------------------------------------------------------------------------------
    _HF_VISION = (("embeddings.cls_token", "embeddings.class_embedding"),
                  ("embeddings.patch_embeddings.projection.", "embeddings.patch_embedding."),
                  ("embeddings.position_embeddings", "embeddings.position_embedding"),
                  ("encoder.layer.", "encoder.layers."), (".attention.projection_layer.", ".attn.proj."),
                  (".attention.q_norm.", ".attn.q_norm."), (".attention.k_norm.", ".attn.k_norm."),
                  (".lambda_1", ".ls1"), (".lambda_2", ".ls2"), (".layernorm_before.", ".norm1."),
                  (".layernorm_after.", ".norm2."))
    _HF_PROJ = (("layer_norm.", "0."), ("linear_1.", "1."), ("linear_2.", "3."))

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        n_layers = len(self.vision_model.encoder.layers)
        params = dict(self.named_parameters())
        dec, vis, loaded = [], [], set()
        for n, t in weights:
            if n.startswith("model.language_model."):
                dec.append(("model." + n[len("model.language_model."):], t))
                continue
            if n.startswith("model.") and n[6:].startswith(("vision_tower.", "multi_modal_projector.")):
                n = n[6:]
            if n.startswith("lm_head."):
                dec.append((n, t))
            elif n.startswith("language_model."):
                dec.append((n[len("language_model."):], t))
            elif n.startswith("vision_tower."):                    # transformers layout
                n = n[len("vision_tower."):]
                for a, b in self._HF_VISION:
                    n = n.replace(a, b)
                if n.startswith("layernorm."):
                    continue                                       # final norm: not on the feature path
                vis.append(("vision_model." + n, t))
            elif n.startswith("multi_modal_projector."):
                n = n[len("multi_modal_projector."):]
                for a, b in self._HF_PROJ:
                    if n.startswith(a):
                        n = b + n[len(a):]
                        break
                vis.append(("mlp1." + n, t))
            else:
                vis.append((n, t))
        for n, t in vis:
            if n.startswith("vision_model.encoder.layers.") and int(n.split(".")[3]) >= n_layers:
                continue
            hit = next((s for s in (".attention.q_proj.", ".attention.k_proj.", ".attention.v_proj.") if s in n), None)
            if hit is not None:                                    # separate projections -> rows of the fused one
                target = n.replace(hit, ".attn.qkv.")
                p = params[target]
                H = p.shape[0] // 3
                i = "qkv".index(hit[len(".attention.")])
                p.data[i * H:(i + 1) * H].copy_(t)
                loaded.add(target)
                continue
            if n not in params:
                continue
            params[n].data.copy_(t.reshape(params[n].shape))
            loaded.add(n)
        lm = self.language_model
        sub = set(lm.load_weights(iter(dec)) or ())
        if "lm_head.weight" not in sub and not getattr(lm, "tie", False) and hasattr(lm, "lm_head") \
                and hasattr(getattr(lm, "model", None), "embed_tokens"):
            lm.lm_head.tie_weights(lm.model.embed_tokens)
            lm.tie = True
            sub.add("lm_head.weight")
        return loaded | {"language_model." + n for n in sub}


class NVLM_D_Model(InternVLChatModel):
    """NVLM-D: InternVL whose InternViT-6B carries 7 all-zero heads beside its 25 (32 divides over 8 ranks) and
    whose projector is bias-free and as wide as the decoder's MLP.  The tile tags (``<tile_1>`` ... in front of
    each tile's positions) are text: a prompt that carries them keeps them, this package's processor only expands
    the image positions.  Reference: vllm/model_executor/models/nvlm_d.py:171-216."""
    vision_dummy_heads = 7

    def _make_projector(self, Hv: int, tc) -> nn.Sequential:
        return nn.Sequential(nn.LayerNorm(Hv), nn.Linear(Hv, tc.intermediate_size, bias=False), nn.GELU(),
                             nn.Linear(tc.intermediate_size, tc.hidden_size, bias=False))


class InternVLForConditionalGeneration(InternVLChatModel):
    """The same model under transformers' architecture name and configuration keys."""


# ---- processor ------------------------------------------------------------------------------------------------
def closest_grid(w: int, h: int, min_num: int, max_num: int, S: int) -> tuple[int, int]:
    """(columns, rows) of tiles whose aspect ratio is closest to the image's; on ties the larger grid wins when the
    image has more than half its pixels."""
    ratios = sorted({(i, j) for n in range(min_num, max_num + 1) for i in range(1, n + 1) for j in range(1, n + 1)
                     if min_num <= i * j <= max_num}, key=lambda r: r[0] * r[1])
    best, best_diff, aspect, area = (1, 1), float("inf"), w / h, w * h
    for r in ratios:
        diff = abs(aspect - r[0] / r[1])
        if diff < best_diff:
            best, best_diff = r, diff
        elif diff == best_diff and area > 0.5 * S * S * r[0] * r[1]:
            best = r
    return best


class InternVLImageProcessor:
    """Image -> tiles [n, 3, S, S] (grid tiles row-major, then the thumbnail when there is more than one), and the
    prompt's context token -> tokens_per_tile * n copies.  Pre-processed input: {"pixel_values": tiles}."""

    mean, std = IMAGENET_MEAN, IMAGENET_STD

    def __init__(self, context_token_id: int, S: int, tokens_per_tile: int, min_num: int, max_num: int,
                 thumbnail: bool, dynamic: bool):
        self.context_token_id, self.S, self.tpt = context_token_id, S, tokens_per_tile
        self.min_num, self.max_num, self.thumbnail, self.dynamic = min_num, max_num, thumbnail, dynamic

    def tiles(self, img: Any) -> torch.Tensor:
        if isinstance(img, dict):
            return torch.as_tensor(img["pixel_values"]).float()
        if hasattr(img, "convert"):
            img = np.asarray(img.convert("RGB"))
        t = torch.as_tensor(np.asarray(img)).permute(2, 0, 1).float()[None]
        H, W = t.shape[-2:]
        S = self.S

        def resize(h, w):
            return F.interpolate(t, size=(h, w), mode="bicubic", align_corners=False, antialias=True).clamp(0, 255)
        cols, rows = closest_grid(W, H, self.min_num, self.max_num, S) if self.dynamic else (1, 1)
        big = resize(rows * S, cols * S)[0]
        parts = big.unfold(1, S, S).unfold(2, S, S).permute(1, 2, 0, 3, 4).reshape(-1, 3, S, S)
        if self.thumbnail and parts.shape[0] > 1:
            parts = torch.cat((parts, resize(S, S)), dim=0)
        m, sd = torch.tensor(self.mean).view(1, 3, 1, 1), torch.tensor(self.std).view(1, 3, 1, 1)
        return (parts / 255.0 - m) / sd

    def __call__(self, prompt_token_ids: list[int], mm_data: dict):
        from ..multimodal import PlaceholderRange, hash_item
        images = mm_data.get("image")
        if images is None:
            raise ValueError("multi_modal_data needs an 'image' entry")
        if not isinstance(images, (list, tuple)):
            images = [images]
        runs, prev = [], -2
        for i, tok in enumerate(prompt_token_ids):
            if tok == self.context_token_id:
                if i != prev + 1:
                    runs.append([i, 1])
                else:
                    runs[-1][1] += 1
                prev = i
        if len(runs) != len(images):
            raise ValueError(f"prompt has {len(runs)} image placeholders but {len(images)} images were given")
        out, positions, inputs, hashes, cur = [], [], [], [], 0
        for (start, n), img in zip(runs, images):
            tiles = self.tiles(img)
            ntok = self.tpt * tiles.shape[0]
            out.extend(prompt_token_ids[cur:start])
            positions.append(PlaceholderRange(offset=len(out), length=ntok))
            out.extend([self.context_token_id] * ntok)
            cur = start + n
            inputs.append([tiles])
            hashes.append(hash_item(tiles))
        out.extend(prompt_token_ids[cur:])
        return out, inputs, hashes, positions


def internvl_processor(vllm_config):
    hf = vllm_config.model_config.hf_config
    vc = _cfg(hf.vision_config).to_dict()
    size, patch = vc.get("image_size", 448), vc.get("patch_size", 14)
    S = int(size[0] if isinstance(size, (list, tuple)) else size)
    P = int(patch[0] if isinstance(patch, (list, tuple)) else patch)
    ratio = float(getattr(hf, "downsample_ratio", 0.5))
    tok = getattr(hf, "image_token_id", None) or getattr(hf, "img_context_token_id", None)
    if tok is None:
        tc = _cfg(getattr(hf, "llm_config", None) or hf.text_config)
        tok = 92546 if "internlm" in str(getattr(tc, "model_type", "")).lower() else 151667
    return InternVLImageProcessor(int(tok), S, int((S // P) ** 2 * ratio * ratio),
                                  int(getattr(hf, "min_dynamic_patch", 1)), int(getattr(hf, "max_dynamic_patch", 12)),
                                  bool(getattr(hf, "use_thumbnail", True)), bool(getattr(hf, "dynamic_image_size", True)))
