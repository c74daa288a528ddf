"""Molmo: an OpenAI-CLIP style ViT run over a low-resolution view of the image plus overlapping high-resolution crops;
the states of two late layers are concatenated, every 2 x 2 block of patches is pooled by one attention step whose
query is the block's mean, a gated MLP projects the result, and the vectors replace the image-patch tokens of the
prompt (row by row over the stitched crops, each row closed by a column token).  The decoder is an OLMo / Qwen2 style
stack with fused ``att_proj`` / ``ff_proj`` (``ff_proj`` holds [up | gate]), optional full-width q / k norms and the
norm-after variant.  The embedding table has 128 extra rows (``wte.new_embedding``) that the head does not.

No transformers class exists for this family (modelling and preprocessing ship with the checkpoint): parity unpinned.
The tower, pooling and projector are tested against the arithmetic written out; the crop / token layout is written
from the published preprocessing algorithm (overlap-and-resize, margins of 4 patches, crop windows of 16 patches).
Already preprocessed inputs -- ``{"images": [crops, 576, 588], "image_masks": [crops, 576],
"image_input_idx": [crops, 144]}`` with the expanded prompt -- are taken as they are.
Reference: vllm/model_executor/models/molmo.py (VisionTransformer :283-357, MolmoVisionBackbone :619-742,
MolmoAttention :360-463, decoder layers :543-616, select_tiling :944-964, MolmoForCausalLM :1333-1576).
"""
from __future__ import annotations

import copy
import math
from typing import Iterable, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import HFConfig, VllmConfig
from .llama import LlamaDecoderLayer, LlamaForCausalLM, LlamaModel
from .llama_variants import Olmo2DecoderLayer, _NoResidualFinalNorm
from .utils import load_weights_with_mapping

VIT_LAYERS, POOL, EXTRA_VOCAB = (-2, -9), 2, 128
VIT = dict(size=336, patch=14, dim=1024, heads=16, layers=23, mlp=4096, num_pos=577, eps=1e-5)


class _MHA(nn.Module):
    def __init__(self, dim: int, heads: int, in_dim: int):
        super().__init__()
        self.heads = heads
        self.wq, self.wk, self.wv = nn.Linear(in_dim, dim), nn.Linear(in_dim, dim), nn.Linear(in_dim, dim)
        self.wo = nn.Linear(dim, dim)

    def forward(self, q_in, kv_in=None):                   # [B, Nq, in], [B, Nk, in]
        kv_in = q_in if kv_in is None else kv_in
        B, Nq, _ = q_in.shape
        q, k, v = (p(t).view(B, t.shape[1], self.heads, -1).transpose(1, 2)
                   for p, t in ((self.wq, q_in), (self.wk, kv_in), (self.wv, kv_in)))
        return self.wo(F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, Nq, -1))


class _VitBlock(nn.Module):
    def __init__(self, v: dict):
        super().__init__()
        d = v["dim"]
        self.attention = _MHA(d, v["heads"], d)
        self.feed_forward = nn.Module()
        self.feed_forward.w1, self.feed_forward.w2 = nn.Linear(d, v["mlp"]), nn.Linear(v["mlp"], d)
        self.attention_norm, self.ffn_norm = nn.LayerNorm(d, eps=v["eps"]), nn.LayerNorm(d, eps=v["eps"])

    def forward(self, x):
        x = x + self.attention(self.attention_norm(x))
        h = self.feed_forward.w1(self.ffn_norm(x))
        return x + self.feed_forward.w2(h * torch.sigmoid(1.702 * h))


class MolmoVit(nn.Module):
    def __init__(self, v: dict):
        super().__init__()
        d = v["dim"]
        self.side = v["size"] // v["patch"]
        self.class_embedding = nn.Parameter(torch.zeros(d))
        self.positional_embedding = nn.Parameter(torch.zeros(v["num_pos"], d))
        self.patch_embedding = nn.Linear(v["patch"] ** 2 * 3, d, bias=False)
        self.pre_ln = nn.LayerNorm(d, eps=v["eps"])
        self.transformer = nn.Module()
        self.transformer.resblocks = nn.ModuleList(_VitBlock(v) for _ in range(v["layers"]))

    def forward(self, x) -> list[torch.Tensor]:            # [B, N, 3 * p * p] -> the state after every block
        x = self.patch_embedding(x)
        x = torch.cat([self.class_embedding.to(x.dtype).expand(x.shape[0], 1, -1), x], 1)
        pos = self.positional_embedding
        s = int(math.sqrt(pos.shape[0] - 1))
        if s != self.side:
            grid = F.interpolate(pos[1:].view(1, s, s, -1).permute(0, 3, 1, 2).float(), size=(self.side, self.side),
                                 mode="bicubic", align_corners=False, antialias=True)
            pos = torch.cat([pos[:1], grid.permute(0, 2, 3, 1).reshape(self.side ** 2, -1).to(pos.dtype)], 0)
        x = self.pre_ln(x + pos.to(x.dtype))
        states = []
        for blk in self.transformer.resblocks:
            x = blk(x)
            states.append(x)
        return states


class MolmoVisionBackbone(nn.Module):
    def __init__(self, hidden: int, inter: int, v: dict):
        super().__init__()
        d = v["dim"]
        self.side = v["size"] // v["patch"]
        self.image_vit = MolmoVit(v)
        self.image_pooling_2d = _MHA(d, v["heads"], d * len(VIT_LAYERS))
        self.image_projector = nn.Module()
        p = self.image_projector                                # w1: gate, w3: up, w2: down
        p.w1, p.w3 = nn.Linear(d, inter, bias=False), nn.Linear(d, inter, bias=False)
        p.w2 = nn.Linear(inter, hidden, bias=False)
        self.pad_embed = nn.Parameter(torch.zeros(2, d * len(VIT_LAYERS)))

    def forward(self, images: torch.Tensor, masks: torch.Tensor) -> torch.Tensor:
        """images [crops, N, 3 * p * p], masks [crops, N] (share of real pixels per patch; -1 = unknown) ->
        [crops, ((side + 1) // 2)^2, hidden]"""
        w = self.image_vit.patch_embedding.weight
        images = images.to(w.device, w.dtype)
        T, N, _ = images.shape
        live = ~(images == -1).all(-1).all(-1)                 # crops that are nothing but filler
        states = self.image_vit(images)
        x = torch.cat([states[i] for i in VIT_LAYERS], -1)[:, 1:] * live[:, None, None].to(w.dtype)
        masks = masks.to(w.device)
        all_pad = (masks == 0).to(torch.float32)[..., None]
        part_pad = ((masks < 1) & (masks != 0)).to(torch.float32)[..., None]
        x = (x + self.pad_embed[0] * all_pad + self.pad_embed[1] * part_pad).to(w.dtype)
        s = self.side
        x = x.view(T, s, s, -1)
        if s % POOL:
            x = F.pad(x, (0, 0, 0, s % POOL, 0, s % POOL))
        g = x.shape[1] // POOL
        x = x.view(T, g, POOL, g, POOL, -1).permute(0, 1, 3, 2, 4, 5).reshape(T * g * g, POOL * POOL, -1)
        x = self.image_pooling_2d(x.mean(-2, keepdim=True), x).view(T, g * g, -1)
        p = self.image_projector
        return p.w2(F.silu(p.w1(x)) * p.w3(x))


class _MolmoLayer(LlamaDecoderLayer):
    def _attention_extras(self, config, layer_idx: int) -> dict:
        return {"bias": bool(getattr(config, "qkv_bias", False)), "o_bias": False,
                "qk_norm": "full" if getattr(config, "attention_layer_norm", False) else False}


class _MolmoNormAfterLayer(Olmo2DecoderLayer):
    def _attention_extras(self, config, layer_idx: int) -> dict:
        return _MolmoLayer._attention_extras(self, config, layer_idx)


class _MolmoModel(LlamaModel):
    layer_cls = _MolmoLayer


class _MolmoNormAfterModel(_NoResidualFinalNorm):
    layer_cls = _MolmoNormAfterLayer


class _Decoder(LlamaForCausalLM):
    model_cls = _MolmoModel


class _NormAfterDecoder(LlamaForCausalLM):
    model_cls = _MolmoNormAfterModel
    supports_pp = False


class MolmoForCausalLM(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        self.vocab = int(hf.get("embedding_size") or hf.vocab_size)
        self.norm_after = bool(hf.get("norm_after", False))
        d = hf.to_dict()
        d.update(intermediate_size=int(hf.intermediate_size) // 2, rms_norm_eps=float(hf.get("layer_norm_eps", 1e-5)),
                 vocab_size=self.vocab + EXTRA_VOCAB, tie_word_embeddings=bool(hf.get("weight_tying", False)),
                 hidden_act="silu")
        mc = copy.copy(vllm_config.model_config)
        mc.hf_config = HFConfig(d)
        sub = copy.copy(vllm_config)
        sub.model_config = mc
        self.language_model = (_NormAfterDecoder if self.norm_after else _Decoder)(sub, prefix)
        v = dict(VIT, **(hf.get("vision_backbone") or {}))
        self.vision_backbone = MolmoVisionBackbone(int(hf.hidden_size), int(hf.intermediate_size) // 2, v)

    def encode_mm_items(self, items: list[list[torch.Tensor]]) -> list[torch.Tensor]:
        """items: [images [crops, N, K], masks [crops, N], order (which pooled vector each image-patch token takes),
        the token ids of the placeholder range, the image-patch token id] -> one row per token of the range: the
        pooled vector for a patch token, the token's own embedding for the brackets and column marks."""
        out = []
        for images, masks, order, toks, patch_id in items:
            feats = self.vision_backbone(images, masks).flatten(0, 1)            # [crops * pooled, hidden]
            toks = toks.to(feats.device).long()
            x = self.language_model.get_input_embeddings(toks).to(feats.dtype)
            x[toks == int(patch_id)] = feats[order.to(feats.device).long()]
            out.append(x)
        return out

    def get_input_embeddings(self, input_ids, mm_embeds=None, mm_mask=None):
        x = self.language_model.get_input_embeddings(input_ids)
        if mm_embeds is not None and mm_embeds.numel():
            x[mm_mask] = mm_embeds.to(x.dtype)
        return x

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        return self.language_model(input_ids, positions, intermediate_tensors, inputs_embeds)

    def compute_logits(self, hidden_states):
        logits = self.language_model.compute_logits(hidden_states)
        return logits[..., : self.vocab] if logits is not None else logits

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        c = self.config
        hq, hkv = int(c.num_attention_heads), int(c.get("num_key_value_heads") or c.num_attention_heads)
        hd = int(c.hidden_size) // hq
        norms = {"attn_norm": "post_attention_layernorm", "ff_norm": "post_feedforward_layernorm"} if self.norm_after \
            else {"attn_norm": "input_layernorm", "ff_norm": "post_attention_layernorm"}
        wte: dict[str, torch.Tensor] = {}
        dec, vis = [], []
        for n, t in weights:
            if n.startswith("model.vision_backbone."):
                vis.append((n[len("model."):], t))
                continue
            if "wte.embedding" in n or "wte.new_embedding" in n:
                wte["new" if "new_embedding" in n else "old"] = t
                continue
            n = n.replace("model.transformer.blocks.", "model.layers.").replace("model.transformer.ln_f.", "model.norm.")
            if n.startswith("model.transformer.ff_out."):
                t = torch.cat([t, t.new_zeros(EXTRA_VOCAB, t.shape[1])], 0)      # rows the head never had
                dec.append(("language_model.lm_head." + n.rsplit(".", 1)[1], t))
                continue
            part = n.rsplit(".", 1)[1]
            if ".att_proj." in n:
                q, k, v = t.split([hq * hd, hkv * hd, hkv * hd], 0)
                for name, piece in (("q_proj", q), ("k_proj", k), ("v_proj", v)):
                    dec.append(("language_model." + n.replace(".att_proj.", f".self_attn.{name}."), piece))
                continue
            if ".ff_proj." in n:
                up, gate = t.chunk(2, 0)                                          # x[:d] * silu(x[d:])
                dec.append(("language_model." + n.replace(".ff_proj.", ".mlp.gate_proj."), gate))
                dec.append(("language_model." + n.replace(".ff_proj.", ".mlp.up_proj."), up))
                continue
            n = n.replace(".attn_out.", ".self_attn.o_proj.").replace(".ff_out.", ".mlp.down_proj.")
            n = n.replace(".q_norm.", ".self_attn.q_norm.").replace(".k_norm.", ".self_attn.k_norm.")
            for a, b in norms.items():
                n = n.replace(f".{a}.", f".{b}.")
            dec.append(("language_model." + n, t))
            del part
        if wte:
            dec.append(("language_model.model.embed_tokens.weight", torch.cat([wte["old"], wte["new"]], 0)))
        loaded = load_weights_with_mapping(self, dec, LlamaForCausalLM.stacked_params_mapping,
                                           tie_lm_head=self.language_model.tie)
        loaded |= load_weights_with_mapping(self, vis, [])
        return loaded


# ---- preprocessing ---------------------------------------------------------------------------------------------
def select_tiling(h: int, w: int, window: int, max_crops: int) -> tuple[int, int]:
    """(rows, columns) of crop windows: the smallest grid that needs no down-scaling, else the one that needs the least."""
    grids = sorted(((i, j) for i in range(1, max_crops + 1) for j in range(1, max_crops + 1) if i * j <= max_crops),
                   key=lambda g: g[0] * g[1])
    scale = np.array([min(i * window / h, j * window / w) for i, j in grids], dtype=np.float32)
    ix = int(scale.argmax()) if (scale < 1).all() else int(np.where(scale < 1.0, 10e9, scale).argmin())
    return grids[ix]


def _resize_and_pad(img: torch.Tensor, size: tuple[int, int]) -> tuple[torch.Tensor, torch.Tensor]:
    """[3, H, W] in 0..1 -> ([3, h, w] scaled to fit and centred on zeros, [h, w] True where the image is)"""
    H, W = img.shape[-2:]
    s = min(size[0] / H, size[1] / W)
    nh, nw = int(H * s), int(W * s)
    r = F.interpolate(img[None], size=(nh, nw), mode="bilinear", align_corners=False, antialias=True)[0].clamp(0, 1)
    top, left = (size[0] - nh) // 2, (size[1] - nw) // 2
    out = torch.zeros(3, *size)
    out[:, top:top + nh, left:left + nw] = r
    mask = torch.zeros(size, dtype=torch.bool)
    mask[top:top + nh, left:left + nw] = True
    return out, mask


def _to_patches(x: torch.Tensor, p: int) -> torch.Tensor:
    """[..., C, H, W] -> [..., (H / p) * (W / p), p * p * C] (pixels of a patch row-major, channels last)"""
    *lead, C, H, W = x.shape
    x = x.reshape(*lead, C, H // p, p, W // p, p)
    n = len(lead)
    return x.permute(*range(n), n + 1, n + 3, n + 2, n + 4, n).reshape(*lead, (H // p) * (W // p), p * p * C)


def crops_and_layout(img: torch.Tensor, max_crops: int = 12, size: int = 336, patch: int = 14,
                     margins: tuple[int, int] = (4, 4)):
    """An image [3, H, W] in 0..1 -> (patches [1 + crops, N, K], masks [1 + crops, N], order, rows, columns):
    crop 0 is the whole image at the base size; the others are overlapping windows of the image resized to the
    chosen tiling, of which only the inner part (plus the outer margin at the image border) is used.  ``order``
    lists, row by row over the stitched crops and after the 144 vectors of crop 0, the index of the pooled vector
    (crop * pooled_per_crop + position) that each image token takes; (rows, columns) is the token grid of the crops."""
    from ..multimodal import CLIP_MEAN, CLIP_STD
    mean, std = torch.tensor(CLIP_MEAN).view(3, 1, 1), torch.tensor(CLIP_STD).view(3, 1, 1)
    left, right = margins
    if left % POOL or right % POOL:
        raise ValueError("overlap margins must be multiples of the pooling size")
    cp = size // patch                                          # patches per crop side
    win = cp - left - right                                     # patches a crop contributes in the middle
    H, W = img.shape[-2:]
    th, tw = select_tiling(H - patch * (left + right), W - patch * (left + right), win * patch, max_crops)
    src, real = _resize_and_pad(img, (th * win * patch + patch * (left + right), tw * win * patch + patch * (left + right)))
    src = (src - mean) / std
    tok = (cp + POOL - 1) // POOL                                # pooled vectors per crop side
    crops, masks = [], []
    place = torch.full((th, tw, tok, tok), -1, dtype=torch.long)  # pooled position -> running number, crop by crop
    on = 0
    for i in range(th):
        y0 = 0 if i == 0 else left // POOL
        ch = win + (left if i == 0 else 0) + (right if i == th - 1 else 0)
        for j in range(tw):
            x0 = 0 if j == 0 else left // POOL
            cw = win + (left if j == 0 else 0) + (right if j == tw - 1 else 0)
            ph, pw = (ch + POOL - 1) // POOL, (cw + POOL - 1) // POOL
            place[i, j, y0:y0 + ph, x0:x0 + pw] = torch.arange(on, on + ph * pw).view(ph, pw)
            on += ph * pw
            ys, xs = i * win * patch, j * win * patch
            crops.append(src[:, ys:ys + size, xs:xs + size])
            masks.append(real[ys:ys + size, xs:xs + size])
    patches = _to_patches(torch.stack(crops), patch)
    mask = _to_patches(torch.stack(masks)[:, None].float(), patch).mean(-1)
    # the running numbers in reading order over the stitched crops (row of crops, row inside, crop, column)
    rowwise = place.permute(0, 2, 1, 3).reshape(-1)
    used = rowwise[rowwise >= 0]                                 # token k of the crops takes running number used[k]
    flat = place.reshape(th * tw, tok * tok)
    where = torch.full((on,), -1, dtype=torch.long)              # running number -> crop * tok^2 + position
    idx = (flat >= 0).nonzero()
    where[flat[idx[:, 0], idx[:, 1]]] = idx[:, 0] * tok * tok + idx[:, 1]
    glob, _ = _resize_and_pad(img, (size, size))
    patches = torch.cat([_to_patches(((glob - mean) / std)[None], patch), patches], 0)
    # the published preprocessing appends the low-resolution view's mask row (unknown: -1) BEHIND the crops' rows
    mask = torch.cat([mask, torch.full((1, mask.shape[1]), -1.0)], 0)
    order = torch.cat([torch.arange(tok * tok), where[used] + tok * tok])
    rows = (th * win + left + right + POOL - 1) // POOL
    cols = (tw * win + left + right + POOL - 1) // POOL
    assert used.numel() == rows * cols
    return patches, mask, order, rows, cols


def molmo_processor(vllm_config):
    """Every image-patch token (or run of them) in the prompt stands for one image and becomes
    ``<im_start>`` 12 rows of 12 patch tokens + ``<im_col>`` ``<im_end>`` (the whole image), then the same bracket
    around the stitched crops' rows."""
    from ..multimodal import PlaceholderRange, hash_item
    hf = vllm_config.model_config.hf_config
    kw = vllm_config.model_config.mm_processor_kwargs or {}
    E = int(hf.get("embedding_size") or hf.vocab_size)
    ids = {k: int(kw.get(k, hf.get(k, E + i))) for i, k in
           enumerate(("image_start_token_id", "image_end_token_id", "image_patch_token_id", "image_col_token_id"))}
    start, end, patch_id, col = (ids[k] for k in ("image_start_token_id", "image_end_token_id", "image_patch_token_id",
                                                 "image_col_token_id"))
    max_crops = int(kw.get("max_crops", 12))
    margins = tuple(int(m) for m in kw.get("overlap_margins", (4, 4)))
    v = dict(VIT, **(hf.get("vision_backbone") or {}))

    def pixels(img) -> torch.Tensor:
        if isinstance(img, torch.Tensor) and img.dtype.is_floating_point:
            return img
        a = np.asarray(img.convert("RGB")) if hasattr(img, "convert") else np.asarray(img)
        return torch.as_tensor(a).permute(2, 0, 1).float() / 255.0

    def block(rows: int, cols: int) -> list[int]:
        return [start] + ([patch_id] * cols + [col]) * rows + [end]

    def process(prompt: list[int], mm_data: dict):
        images = mm_data.get("image")
        if images is None:
            raise ValueError("multi_modal_data needs an 'image' entry")
        images = images if isinstance(images, (list, tuple)) else [images]
        if isinstance(images[0], dict):                          # the checkpoint's own processor already ran
            out, items, hashes, ranges = list(prompt), [], [], []
            pos = [i for i, t in enumerate(out) if t == patch_id]
            o = 0
            for d in images:
                idx = torch.as_tensor(d["image_input_idx"]).long()
                used = (idx >= 0).flatten()
                # features are handed over in token order: sort the used ones by the token they land on
                order = used.nonzero().flatten()[idx.flatten()[used].argsort()]
                n = int(used.sum())
                a, b = pos[o], pos[o + n - 1] + 1
                items.append([torch.as_tensor(d["images"]).float(), torch.as_tensor(d["image_masks"]).float(), order,
                              torch.tensor(out[a:b]), torch.tensor(patch_id)])
                hashes.append(hash_item(items[-1][0]))
                ranges.append(PlaceholderRange(offset=a, length=b - a))
                o += n
            return out, items, hashes, ranges
        out, items, hashes, ranges, k, i = [], [], [], [], 0, 0
        while i < len(prompt):
            if prompt[i] != patch_id:
                out.append(prompt[i])
                i += 1
                continue
            while i < len(prompt) and prompt[i] == patch_id:
                i += 1
            if k >= len(images):
                raise ValueError("more image placeholders than images")
            px = pixels(images[k])
            patches, mask, order, rows, cols = crops_and_layout(px, max_crops, v["size"], v["patch"], margins)
            side = (v["size"] // v["patch"] + POOL - 1) // POOL
            toks = block(side, side) + block(rows, cols)
            ranges.append(PlaceholderRange(offset=len(out), length=len(toks)))
            out.extend(toks)
            items.append([patches, mask, order, torch.tensor(toks), torch.tensor(patch_id)])
            hashes.append(hash_item(images[k] if not isinstance(images[k], torch.Tensor) else px))
            k += 1
        if k != len(images):
            raise ValueError(f"prompt has {k} image placeholders but {len(images)} images were given")
        return out, items, hashes, ranges

    return process
