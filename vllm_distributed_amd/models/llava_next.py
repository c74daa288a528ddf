"""LLaVA-NeXT and LLaVA-OneVision (images): an image of any resolution is resized and padded to the best-fitting
grid of encoder-sized tiles; the tiles and a down-scaled copy of the whole image go through the vision tower (CLIP
with its class token dropped, or SigLIP) and the two-layer projector; the tile features are laid back out as one
feature map, the padding is cut off again, a learned ``image_newline`` vector closes every row, and the result
follows the base image's features into the prompt.

An item reaches the model as (tiles [1 + n, 3, S, S], original size [2]); its token count is a function of the
original size alone, which is what lets the processor expand the prompt before anything runs.
Reference: vllm/model_executor/models/{llava_next,llava_onevision,siglip}.py.
"""
from __future__ import annotations

import math
from typing import Any, Iterable

import numpy as np
import torch
import torch.nn as nn

from ..config import VllmConfig
from ..layers.basic import get_act_fn
from .llava import ClipLayer, ClipVisionModel, _cfg
from .utils import load_weights_with_mapping, maybe_prefix


# ---- geometry (shared by the processor and the model) ---------------------------------------------------------
def select_best_resolution(size: tuple[int, int], candidates: list) -> tuple[int, int]:
    """(height, width) of the candidate that keeps most of the image's pixels and wastes least."""
    h, w = size
    best, best_eff, best_waste = None, -1, float("inf")
    for ch, cw in candidates:
        s = min(cw / w, ch / h)
        eff = min(int(w * s) * int(h * s), w * h)
        waste = ch * cw - eff
        if eff > best_eff or (eff == best_eff and waste < best_waste):
            best, best_eff, best_waste = (int(ch), int(cw)), eff, waste
    return best


def _unpadded(cur_h: int, cur_w: int, orig_h: int, orig_w: int) -> tuple[int, int, int, int]:
    """(top, bottom, left, right) bounds of the part of a [cur_h, cur_w] map that shows image, not padding."""
    if orig_w / orig_h > cur_w / cur_h:
        new_h = int(round(orig_h * (cur_w / orig_w), 7))
        pad = (cur_h - new_h) // 2
        return pad, cur_h - pad, 0, cur_w
    new_w = int(round(orig_w * (cur_h / orig_h), 7))
    pad = (cur_w - new_w) // 2
    return 0, cur_h, pad, cur_w - pad


def anyres_feature_shape(orig: tuple[int, int], pinpoints: list, image_size: int, patch: int,
                         max_tiles: int = 0) -> tuple[int, int, int, int]:
    """(tile rows, tile columns, feature rows, feature columns after unpadding [and the OneVision size cap])."""
    bh, bw = select_best_resolution(orig, pinpoints)
    nth, ntw = bh // image_size, bw // image_size
    side = image_size // patch
    t, b, l, r = _unpadded(nth * side, ntw * side, orig[0], orig[1])
    fh, fw = b - t, r - l
    if max_tiles:
        ratio = math.sqrt(fh * fw / (max_tiles * side * side))
        if ratio > 1.1:
            fh, fw = int(fh // ratio), int(fw // ratio)
    return nth, ntw, fh, fw


def anyres_num_tokens(orig, pinpoints, image_size, patch, max_tiles=0) -> int:
    _, _, fh, fw = anyres_feature_shape(orig, pinpoints, image_size, patch, max_tiles)
    return (image_size // patch) ** 2 + fh * (fw + 1)


class SiglipVisionModel(nn.Module):
    """SigLIP's vision transformer up to the layer whose output is used (no class token, biased patch projection)."""

    def __init__(self, c, feature_layer: int):
        super().__init__()
        H, P, S = c.hidden_size, c.patch_size, c.image_size
        self.embeddings = nn.Module()
        self.embeddings.patch_embedding = nn.Conv2d(getattr(c, "num_channels", 3), H, P, P, bias=True)
        self.embeddings.position_embedding = nn.Embedding((S // P) ** 2, H)
        n = c.num_hidden_layers
        keep = feature_layer + 1 if feature_layer >= 0 else n + feature_layer + 1
        self.encoder = nn.Module()
        self.encoder.layers = nn.ModuleList([ClipLayer(c) for _ in range(keep)])

    def forward(self, pixel_values):
        e = self.embeddings
        x = e.patch_embedding(pixel_values.to(e.patch_embedding.weight.dtype)).flatten(2).transpose(1, 2)
        x = x + e.position_embedding.weight.unsqueeze(0)
        for layer in self.encoder.layers:
            x = layer(x)
        return x


def build_language_model(vllm_config: VllmConfig, text_config, prefix: str):
    """The decoder a multimodal checkpoint wraps, from its ``text_config``."""
    import copy
    from .registry import ModelRegistry
    by_type = {"llama": "LlamaForCausalLM", "mistral": "MistralForCausalLM", "qwen2": "Qwen2ForCausalLM",
               "qwen3": "Qwen3ForCausalLM", "gemma": "GemmaForCausalLM", "gemma2": "Gemma2ForCausalLM",
               "cohere2": "Cohere2ForCausalLM", "phi3": "Phi3ForCausalLM", "persimmon": "PersimmonForCausalLM",
               "minimax": "MiniMaxForCausalLM", "minimax_text_01": "MiniMaxText01ForCausalLM"}
    archs = list(getattr(text_config, "architectures", None) or [])
    if not archs:
        mt = getattr(text_config, "model_type", "llama")
        if mt not in by_type:
            raise ValueError(f"no decoder registered for text model type {mt!r}")
        archs = [by_type[mt]]
        text_config.architectures = archs
    sub = copy.copy(vllm_config)
    sub.model_config = copy.copy(vllm_config.model_config)
    sub.model_config.hf_config = text_config
    return ModelRegistry.resolve_model_cls(archs)[0](sub, prefix=prefix)


class LlavaNextForConditionalGeneration(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False
    max_tiles = 0                       # OneVision: cap of the tile feature map, in tiles

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vc, tc = _cfg(hf.vision_config), _cfg(hf.text_config)
        self.vc = vc
        self.image_token_id = int(getattr(hf, "image_token_index", None) or getattr(hf, "image_token_id", 32000))
        self.select = getattr(hf, "vision_feature_select_strategy", "default")
        self.pinpoints = [list(p) for p in hf.image_grid_pinpoints]
        fl = int(getattr(hf, "vision_feature_layer", -2))
        self.vision_tower = nn.Module()
        siglip = "siglip" in str(getattr(vc, "model_type", "clip"))
        self.vision_tower.vision_model = SiglipVisionModel(vc, fl) if siglip else ClipVisionModel(vc, fl)
        self.has_cls = not siglip
        bias = bool(getattr(hf, "multimodal_projector_bias", True))
        self.multi_modal_projector = nn.Module()
        self.multi_modal_projector.linear_1 = nn.Linear(vc.hidden_size, tc.hidden_size, bias=bias)
        self.multi_modal_projector.linear_2 = nn.Linear(tc.hidden_size, tc.hidden_size, bias=bias)
        self.proj_act = get_act_fn(getattr(hf, "projector_hidden_act", "gelu"))
        self.image_newline = nn.Parameter(torch.zeros(tc.hidden_size), requires_grad=False)
        self.language_model = build_language_model(vllm_config, tc, maybe_prefix(prefix, "language_model"))

    # ---- encoder ------------------------------------------------------------------------------------------
    def _tile_features(self, tiles: torch.Tensor) -> torch.Tensor:
        feats = self.vision_tower.vision_model(tiles)
        if self.select == "default" and self.has_cls:
            feats = feats[:, 1:]
        p = self.multi_modal_projector
        return p.linear_2(self.proj_act(p.linear_1(feats)))

    def encode_mm_items(self, items: list[list[torch.Tensor]]) -> list[torch.Tensor]:
        dtype = self.image_newline.dtype
        feats = self._tile_features(torch.cat([it[0].to(dtype) for it in items], dim=0))
        out, k = [], 0
        side = self.vc.image_size // self.vc.patch_size
        for tiles, size in ((it[0], it[1]) for it in items):
            n = tiles.shape[0]
            f = feats[k:k + n]
            k += n
            if n == 1:
                out.append(torch.cat((f[0], self.image_newline[None].to(f.dtype)), dim=0))
                continue
            oh, ow = int(size[0]), int(size[1])
            nth, ntw, fh, fw = anyres_feature_shape((oh, ow), self.pinpoints, self.vc.image_size, self.vc.patch_size,
                                                    self.max_tiles)
            grid = f[1:].view(nth, ntw, side, side, -1).permute(4, 0, 2, 1, 3).reshape(-1, nth * side, ntw * side)
            t, b, l, r = _unpadded(nth * side, ntw * side, oh, ow)
            grid = grid[:, t:b, l:r]
            if (grid.shape[1], grid.shape[2]) != (fh, fw):          # OneVision: the map is scaled down to its cap
                grid = nn.functional.interpolate(grid[None], [fh, fw], mode="bilinear")[0]
            nl = self.image_newline.to(grid.dtype)[:, None, None].expand(-1, fh, 1)
            grid = torch.cat((grid, nl), dim=-1).flatten(1, 2).transpose(0, 1)
            out.append(torch.cat((f[0], grid), dim=0))
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
        n_layers = len(self.vision_tower.vision_model.encoder.layers)

        def rename(n: str):
            if n.startswith("model.language_model."):
                return "language_model.model." + n[len("model.language_model."):]
            for p in ("vision_tower.", "multi_modal_projector.", "image_newline"):
                if n.startswith("model." + p):
                    n = n[len("model."):]
            if n.startswith("lm_head."):
                return "language_model." + n
            if n.startswith("vision_tower.") and not n.startswith("vision_tower.vision_model."):
                n = "vision_tower.vision_model." + n[len("vision_tower."):]
            if n.startswith(("vision_tower.vision_model.post_layernorm", "vision_tower.vision_model.head.")):
                return None
            if n.startswith("vision_tower.vision_model.encoder.layers.") and int(n.split(".")[4]) >= n_layers:
                return None
            return n
        dec, vis = [], []
        for n, t in weights:
            r = rename(n)
            if r is not None:
                (dec if r.startswith("language_model.") else vis).append((r, t))
        from .llama import LlamaForCausalLM
        lm = self.language_model
        if type(lm).load_weights is LlamaForCausalLM.load_weights:
            loaded = load_weights_with_mapping(self, dec, lm.stacked_params_mapping, tie_lm_head=lm.tie)
        else:                                # decoders with a routing of their own see their own names
            sub = lm.load_weights((n[len("language_model."):], t) for n, t in dec)
            loaded = {"language_model." + n for n in (sub or ())}
        loaded |= load_weights_with_mapping(self, vis, [])
        return loaded


class LlavaOnevisionForConditionalGeneration(LlavaNextForConditionalGeneration):
    """SigLIP tower, every token of it used, and the tile feature map capped at ``anyres_max_N`` tiles' worth of
    tokens (images; the video path of the checkpoint family is not served)."""

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        ar = str(getattr(self.config, "vision_aspect_ratio", "anyres_max_9"))
        self.max_tiles = int(ar.rsplit("_", 1)[-1]) if ar.startswith("anyres_max_") else 0


# ---- processor ------------------------------------------------------------------------------------------------
class AnyResImageProcessor:
    """Image -> (tiles [1 + n, 3, S, S], original size), and the prompt's image token -> as many copies as the
    item will produce features.  Pre-processed input is accepted as {"pixel_values": tiles, "image_size": (h, w)}."""

    def __init__(self, image_token_id: int, image_size: int, patch: int, pinpoints: list, mean, std, max_tiles: int,
                 drop_cls_only: bool = True):
        self.image_token_id, self.S, self.P = image_token_id, image_size, patch
        self.pinpoints, self.mean, self.std, self.max_tiles = pinpoints, mean, std, max_tiles

    def tiles(self, img: Any) -> tuple[torch.Tensor, torch.Tensor]:
        if isinstance(img, dict):
            return torch.as_tensor(img["pixel_values"]).float(), torch.as_tensor(img["image_size"], dtype=torch.int64)
        if hasattr(img, "convert"):
            img = np.asarray(img.convert("RGB"))
        t = torch.as_tensor(np.asarray(img)).permute(2, 0, 1).float()[None]
        H, W = t.shape[-2:]
        bh, bw = select_best_resolution((H, W), self.pinpoints)
        s = min(bw / W, bh / H)
        nh, nw = min(math.ceil(H * s), bh), min(math.ceil(W * s), bw)
        F = torch.nn.functional
        r = F.interpolate(t, size=(nh, nw), mode="bicubic", align_corners=False, antialias=True).clamp(0, 255)
        top, left = (bh - nh) // 2, (bw - nw) // 2
        canvas = torch.zeros(1, 3, bh, bw)
        canvas[:, :, top:top + nh, left:left + nw] = r
        S = self.S
        parts = canvas[0].unfold(1, S, S).unfold(2, S, S).permute(1, 2, 0, 3, 4).reshape(-1, 3, S, S)
        base = F.interpolate(t, size=(S, S), mode="bicubic", align_corners=False, antialias=True).clamp(0, 255)
        px = torch.cat((base, parts), dim=0) / 255.0
        m, sd = torch.tensor(self.mean).view(1, 3, 1, 1), torch.tensor(self.std).view(1, 3, 1, 1)
        return (px - m) / sd, torch.tensor([H, W], dtype=torch.int64)

    def __call__(self, prompt_token_ids: list[int], mm_data: dict):
        from ..multimodal import PlaceholderRange, hash_item
        images = mm_data.get("image")
        if images is None:
            raise ValueError("multi_modal_data needs an 'image' entry")
        if not isinstance(images, (list, tuple)):
            images = [images]
        runs, prev = [], -2
        for i, tok in enumerate(prompt_token_ids):
            if tok == self.image_token_id:
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
            tiles, size = self.tiles(img)
            if tiles.shape[0] == 1:
                ntok = (self.S // self.P) ** 2 + 1
            else:
                ntok = anyres_num_tokens((int(size[0]), int(size[1])), self.pinpoints, self.S, self.P, self.max_tiles)
            out.extend(prompt_token_ids[cur:start])
            positions.append(PlaceholderRange(offset=len(out), length=ntok))
            out.extend([self.image_token_id] * ntok)
            cur = start + n
            inputs.append([tiles, size])
            hashes.append(hash_item(tiles))
        out.extend(prompt_token_ids[cur:])
        return out, inputs, hashes, positions


def _anyres_processor(vllm_config):
    from ..multimodal import CLIP_MEAN, CLIP_STD
    hf = vllm_config.model_config.hf_config
    vc = hf.vision_config if isinstance(hf.vision_config, dict) else hf.vision_config.to_dict()
    arch = (vllm_config.model_config.architectures or [""])[0]
    one = "Onevision" in arch
    siglip = "siglip" in str(vc.get("model_type", "clip"))
    mean, std = ((0.5, 0.5, 0.5), (0.5, 0.5, 0.5)) if siglip else (CLIP_MEAN, CLIP_STD)
    ar = str(getattr(hf, "vision_aspect_ratio", "anyres_max_9"))
    max_tiles = int(ar.rsplit("_", 1)[-1]) if (one and ar.startswith("anyres_max_")) else 0
    return AnyResImageProcessor(int(getattr(hf, "image_token_index", None) or getattr(hf, "image_token_id", 32000)),
                                int(vc.get("image_size", 336)), int(vc.get("patch_size", 14)),
                                [list(p) for p in hf.image_grid_pinpoints], mean, std, max_tiles)


# ---- LLaVA-NeXT-Video -----------------------------------------------------------------------------------------
class LlavaNextVideoForConditionalGeneration(LlavaNextForConditionalGeneration):
    """LLaVA-NeXT plus videos: every frame goes through the tower at the encoder's own size, its feature map is
    pooled with stride ``spatial_pool_stride`` (average, max or a learned convolution) and projected; the frames'
    tokens follow each other in the prompt in place of a run of video tokens (no newline vectors).  A video item
    reaches the model as [frames [F, 3, S, S]], an image item as LLaVA-NeXT's [tiles, size].
    Reference: vllm/model_executor/models/llava_next_video.py (LlavaNextVideoPooler :155-193, video path :360-420)."""

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        hf = self.config
        mode, stride = str(getattr(hf, "spatial_pool_mode", "average")), int(getattr(hf, "spatial_pool_stride", 2))
        self.video_token_id = int(getattr(hf, "video_token_index", None) or getattr(hf, "video_token_id", 32000))
        self.vision_resampler = nn.Module()
        if mode == "average":
            self.vision_resampler.pool = nn.AvgPool2d(stride, stride)
        elif mode == "max":
            self.vision_resampler.pool = nn.MaxPool2d(stride, stride)
        elif mode == "conv":
            H = self.vc.hidden_size
            self.vision_resampler.pool = nn.Conv2d(H, int(getattr(hf, "spatial_pool_out_channels", H)), stride, stride)
        else:
            raise ValueError(f"unknown spatial_pool_mode {mode!r}")

    def _video_features(self, frames: torch.Tensor) -> torch.Tensor:
        feats = self.vision_tower.vision_model(frames.to(self.image_newline.dtype))
        if self.select == "default" and self.has_cls:
            feats = feats[:, 1:]
        n, t, d = feats.shape
        side = int(math.sqrt(t))
        pooled = self.vision_resampler.pool(feats.view(n, side, side, d).permute(0, 3, 1, 2))
        p = self.multi_modal_projector
        return p.linear_2(self.proj_act(p.linear_1(pooled.flatten(2).transpose(1, 2)))).flatten(0, 1)

    def encode_mm_items(self, items: list[list[torch.Tensor]]) -> list[torch.Tensor]:
        images = [(i, it) for i, it in enumerate(items) if len(it) == 2]
        out: list = [None] * len(items)
        if images:
            for (i, _), f in zip(images, super().encode_mm_items([it for _, it in images])):
                out[i] = f
        for i, it in enumerate(items):
            if len(it) == 1:
                out[i] = self._video_features(it[0])
        return out


class FrameVideoProcessor(AnyResImageProcessor):
    """Adds videos to the any-resolution image processor: frames [F, H, W, 3] are resized (shortest side to the
    encoder size, centre crop) and normalised; a run of video tokens becomes F * (side / stride)^2 of them.
    Pre-processed input: {"pixel_values_videos": [F, 3, S, S]}."""

    def __init__(self, *a, video_token_id: int, pool_stride: int, **kw):
        super().__init__(*a, **kw)
        self.video_token_id, self.stride = video_token_id, pool_stride

    def frames(self, video: Any) -> torch.Tensor:
        if isinstance(video, dict):
            return torch.as_tensor(video["pixel_values_videos"]).float()
        t = torch.as_tensor(np.asarray(video)).permute(0, 3, 1, 2).float()
        H, W = t.shape[-2:]
        S = self.S
        s = S / min(H, W)
        nh, nw = max(S, round(H * s)), max(S, round(W * s))
        t = torch.nn.functional.interpolate(t, size=(nh, nw), mode="bicubic", align_corners=False,
                                            antialias=True).clamp(0, 255)
        top, left = (nh - S) // 2, (nw - S) // 2
        t = t[:, :, top:top + S, left:left + S] / 255.0
        m, sd = torch.tensor(self.mean).view(1, 3, 1, 1), torch.tensor(self.std).view(1, 3, 1, 1)
        return (t - m) / sd

    def __call__(self, prompt_token_ids: list[int], mm_data: dict):
        from ..multimodal import PlaceholderRange, hash_item
        videos = mm_data.get("video")
        if videos is None:
            return super().__call__(prompt_token_ids, mm_data)
        if not isinstance(videos, (list, tuple)):
            videos = [videos]
        # videos first pass: expand their runs; images (if any) are expanded afterwards on the result
        runs, prev = [], -2
        for i, tok in enumerate(prompt_token_ids):
            if tok == self.video_token_id:
                if i != prev + 1:
                    runs.append([i, 1])
                else:
                    runs[-1][1] += 1
                prev = i
        if len(runs) != len(videos):
            raise ValueError(f"prompt has {len(runs)} video placeholders but {len(videos)} videos were given")
        per_frame = (self.S // self.P // self.stride) ** 2
        out, vid, cur = [], [], 0
        for (start, n), v in zip(runs, videos):
            fr = self.frames(v)
            out.extend(prompt_token_ids[cur:start])
            vid.append((len(out), fr.shape[0] * per_frame, fr))
            out.extend([self.video_token_id] * (fr.shape[0] * per_frame))
            cur = start + n
        out.extend(prompt_token_ids[cur:])
        entries = [(off, PlaceholderRange(offset=off, length=ln), [fr], hash_item(fr)) for off, ln, fr in vid]
        if mm_data.get("image") is not None:
            out2, inputs, hashes, positions = super().__call__(out, {"image": mm_data["image"]})
            # image expansion shifted everything behind each image run: recompute the video offsets by scanning
            entries, k = [], 0
            i = 0
            while i < len(out2) and k < len(vid):
                if out2[i] == self.video_token_id:
                    ln, fr = vid[k][1], vid[k][2]
                    entries.append((i, PlaceholderRange(offset=i, length=ln), [fr], hash_item(fr)))
                    i += ln
                    k += 1
                else:
                    i += 1
            entries += [(p.offset, p, inp, h) for p, inp, h in zip(positions, inputs, hashes)]
            out = out2
        entries.sort(key=lambda e: e[0])
        return out, [e[2] for e in entries], [e[3] for e in entries], [e[1] for e in entries]


def _video_processor(vllm_config):
    base = _anyres_processor(vllm_config)
    hf = vllm_config.model_config.hf_config
    return FrameVideoProcessor(base.image_token_id, base.S, base.P, base.pinpoints, base.mean, base.std, base.max_tiles,
                               video_token_id=int(getattr(hf, "video_token_index", None)
                                                  or getattr(hf, "video_token_id", 32000)),
                               pool_stride=int(getattr(hf, "spatial_pool_stride", 2)))


class MiniMaxVL01ForConditionalGeneration(LlavaNextForConditionalGeneration):
    """MiniMax-VL-01: the LLaVA-NeXT image path (tower, two-layer projector, any-resolution tiles with a newline
    embedding per row) in front of the MiniMax-Text-01 decoder (lightning attention state slots + paged KV).
    Reference: vllm/model_executor/models/minimax_vl_01.py:170-360."""
