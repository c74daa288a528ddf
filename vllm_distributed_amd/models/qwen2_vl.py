"""Qwen2-VL: a vision transformer over variable-size patch grids (2-D rotary positions, one attention
span per frame, 2x2 patch merger) in front of a Qwen2 decoder with multimodal rotary positions.
The decoder is the shared Llama-family stack; image positions are three numbers per token
(layers/rotary.py, MRotaryEmbedding) computed here from the prompt and the image grids.
The vision tower runs once per image (replicated over TP ranks) in plain PyTorch.
Reference: vllm/model_executor/models/qwen2_vl.py.
"""
from __future__ import annotations

import copy
from typing import Iterable, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import HFConfig, VllmConfig
from .llama import LlamaForCausalLM
from .utils import load_weights_with_mapping, maybe_prefix


def _rotate_half(x):
    a, b = x.chunk(2, dim=-1)
    return torch.cat((-b, a), dim=-1)


class VisionBlock(nn.Module):
    def __init__(self, dim: int, heads: int, mlp_dim: int, act: str):
        super().__init__()
        self.heads = heads
        self.norm1, self.norm2 = nn.LayerNorm(dim, eps=1e-6), nn.LayerNorm(dim, eps=1e-6)
        self.attn = nn.Module()
        self.attn.qkv, self.attn.proj = nn.Linear(dim, 3 * dim), nn.Linear(dim, dim)
        self.mlp = nn.Module()
        self.mlp.fc1, self.mlp.fc2 = nn.Linear(dim, mlp_dim), nn.Linear(mlp_dim, dim)
        self.act = act

    def _act(self, x):
        if self.act == "quick_gelu":
            return x * torch.sigmoid(1.702 * x)
        return F.gelu(x, approximate="tanh" if "tanh" in self.act else "none")

    def forward(self, x, cos, sin, spans: list[int]):
        N, dim = x.shape
        q, k, v = self.attn.qkv(self.norm1(x)).view(N, 3, self.heads, dim // self.heads).unbind(1)
        q = (q.float() * cos + _rotate_half(q.float()) * sin).to(x.dtype)
        k = (k.float() * cos + _rotate_half(k.float()) * sin).to(x.dtype)
        outs, o = [], 0
        for n in spans:                 # one frame of one image attends within itself
            qs, ks, vs = (t[o:o + n].transpose(0, 1) for t in (q, k, v))
            outs.append(F.scaled_dot_product_attention(qs, ks, vs).transpose(0, 1))
            o += n
        x = x + self.attn.proj(torch.cat(outs, dim=0).reshape(N, dim))
        return x + self.mlp.fc2(self._act(self.mlp.fc1(self.norm2(x))))


class Qwen2VisionTransformer(nn.Module):
    def __init__(self, vc: dict, out_dim: int):
        super().__init__()
        g = vc.get
        self.embed_dim = int(g("embed_dim", 1280))
        self.heads = int(g("num_heads", 16))
        self.patch, self.tps = int(g("patch_size", 14)), int(g("temporal_patch_size", 2))
        self.merge = int(g("spatial_merge_size", 2))
        cin = int(g("in_channels", None) or g("in_chans", 3))
        self.patch_embed = nn.Module()
        self.patch_embed.proj = nn.Conv3d(cin, self.embed_dim, kernel_size=(self.tps, self.patch, self.patch),
                                          stride=(self.tps, self.patch, self.patch), bias=False)
        mlp_dim = int(self.embed_dim * float(g("mlp_ratio", 4)))
        self.blocks = nn.ModuleList([VisionBlock(self.embed_dim, self.heads, mlp_dim, g("hidden_act", "quick_gelu"))
                                     for _ in range(int(g("depth", 32)))])
        m = self.merge ** 2
        self.merger = nn.Module()
        self.merger.ln_q = nn.LayerNorm(self.embed_dim, eps=1e-6)
        self.merger.mlp = nn.Sequential(nn.Linear(self.embed_dim * m, self.embed_dim * m), nn.GELU(),
                                        nn.Linear(self.embed_dim * m, out_dim))
        for p in self.parameters():
            p.requires_grad_(False)

    def _rotary(self, grids: list[tuple[int, int, int]], device) -> tuple[torch.Tensor, torch.Tensor]:
        rot = self.embed_dim // self.heads // 2           # pairs rotated by h, pairs rotated by w: rot / 2 each
        inv = 1.0 / (10000.0 ** (torch.arange(0, rot, 2, dtype=torch.float32, device=device) / rot))
        ids, m = [], self.merge
        for t, h, w in grids:
            hp, wp = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
            shape = (h // m, m, w // m, m)
            hp = hp.reshape(shape).transpose(1, 2).flatten()
            wp = wp.reshape(shape).transpose(1, 2).flatten()
            ids.append(torch.stack([hp, wp], dim=-1).repeat(t, 1))
        ids = torch.cat(ids, dim=0).float()                                    # [N, 2]
        f = (ids[:, :, None] * inv[None, None, :]).flatten(1)                  # [N, rot]
        emb = torch.cat((f, f), dim=-1)
        return emb.cos()[:, None, :], emb.sin()[:, None, :]

    def forward(self, patches: torch.Tensor, grids: list[tuple[int, int, int]]) -> torch.Tensor:
        """patches [N, C * tps * patch * patch] of all images, grids (t, h, w) in patches -> [N / merge^2, out]"""
        w = self.patch_embed.proj.weight
        x = F.linear(patches.to(w.dtype), w.view(self.embed_dim, -1))
        cos, sin = self._rotary(grids, x.device)
        spans = [h * w_ for t, h, w_ in grids for _ in range(t)]
        for blk in self.blocks:
            x = blk(x, cos, sin, spans)
        m = self.merge ** 2
        return self.merger.mlp(self.merger.ln_q(x).view(-1, self.embed_dim * m))


def mrope_positions(prompt_len: int, image_spans: list[tuple[int, int]], grids: list[tuple[int, int, int]],
                    merge: int, t_scales: Optional[list[float]] = None) -> tuple[np.ndarray, int]:
    """Three position rows for a prompt whose images occupy ``image_spans`` (offset, length) with patch
    grids ``grids``.  Text continues from the largest position so far on all axes; an image's tokens take
    (frame, row, column) offsets from where it starts.  ``t_scales``: positions per temporal step of every item
    (Qwen2.5-VL spaces video frames by their distance in time), truncated to integers.  Returns
    ([3, prompt_len], delta) where positions of later tokens are their index + delta."""
    pos = np.zeros((3, prompt_len), dtype=np.int64)
    cur, nxt = 0, 0            # index into the prompt, next free position
    for k, ((off, length), (t, h, w)) in enumerate(zip(image_spans, grids)):
        n = off - cur
        pos[:, cur:off] = nxt + np.arange(n)
        nxt += n
        gh, gw = h // merge, w // merge
        if t * gh * gw != length:
            raise ValueError(f"image placeholder of {length} tokens does not match its grid {(t, h, w)}")
        steps = np.arange(t) if t_scales is None else (np.arange(t) * float(t_scales[k])).astype(np.int64)
        ti = np.repeat(steps, gh * gw)
        hi = np.tile(np.repeat(np.arange(gh), gw), t)
        wi = np.tile(np.arange(gw), t * gh)
        pos[:, off:off + length] = np.stack([ti, hi, wi]) + nxt
        nxt += max(int(steps[-1]) + 1, gh, gw)
        cur = off + length
    pos[:, cur:] = nxt + np.arange(prompt_len - cur)
    nxt += prompt_len - cur
    return pos, int(nxt - prompt_len)


class Qwen2VLForConditionalGeneration(nn.Module):
    is_multimodal = True
    uses_mrope = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vc = hf.vision_config if isinstance(hf.vision_config, dict) else hf.vision_config.to_dict()
        self.merge = int(vc.get("spatial_merge_size", 2))
        self.image_token_id = int(getattr(hf, "image_token_id", 151655))
        self.visual = Qwen2VisionTransformer(vc, int(hf.hidden_size))
        self.language_model = LlamaForCausalLM(vllm_config, prefix=maybe_prefix(prefix, "language_model"))
        self.rotary = self.language_model.model.layers[self.language_model.model.start_layer].self_attn.rotary_emb

    # ---- encoder ----------------------------------------------------------------------------------
    def encode_mm_items(self, items: list) -> list[torch.Tensor]:
        """items: (patches [n_i, C*tps*p*p], grid [3]) per image -> embeddings [n_i / merge^2, H] per image."""
        grids = [tuple(int(x) for x in it[1].tolist()) for it in items]
        dev = self.visual.patch_embed.proj.weight.device
        out = self.visual(torch.cat([it[0] for it in items], dim=0).to(dev), grids)
        sizes = [t * h * w // self.merge ** 2 for t, h, w in grids]
        return list(out.split(sizes, dim=0))

    def get_mrope_positions(self, prompt_len: int, mm_positions: list, mm_inputs: list) -> tuple[np.ndarray, int]:
        spans = [(p.offset, p.length) for p in mm_positions]
        grids = [tuple(int(x) for x in it[1].tolist()) for it in mm_inputs]
        # a third entry: seconds per temporal step of a video (Qwen2.5-VL); images and Qwen2-VL count steps
        tps = float(getattr(self, "tokens_per_second", 0.0) or 0.0)
        scales = [float(it[2]) * tps if len(it) > 2 and tps else 1.0 for it in mm_inputs]
        return mrope_positions(prompt_len, spans, grids, self.merge, scales if any(x != 1.0 for x in scales) else None)

    def get_input_embeddings(self, input_ids, mm_embeds: Optional[torch.Tensor] = None,
                             mm_mask: Optional[torch.Tensor] = None):
        x = self.language_model.get_input_embeddings(input_ids)
        if mm_embeds is not None and mm_embeds.numel():
            x[mm_mask] = mm_embeds.to(x.dtype)
        return x

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        if positions.dim() == 2 and hasattr(self.rotary, "prepare"):
            self.rotary.prepare(positions)
            try:
                return self.language_model(input_ids, positions[0], intermediate_tensors, inputs_embeds)
            finally:
                self.rotary.prepare(None)
        return self.language_model(input_ids, positions, intermediate_tensors, inputs_embeds)

    def compute_logits(self, hidden_states):
        return self.language_model.compute_logits(hidden_states)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        def rename(n: str) -> str:
            # current HF layout: model.{language_model,visual}.* + lm_head.*; original: model.* , visual.*, lm_head.*
            if n.startswith("model.language_model."):
                return "language_model.model." + n[len("model.language_model."):]
            if n.startswith("model.visual."):
                return n[len("model."):]
            if n.startswith("visual."):
                return n
            return "language_model." + n
        dec, vis = [], []
        for n, t in weights:
            r = rename(n)
            (vis if r.startswith("visual.") else dec).append((r, t))
        loaded = load_weights_with_mapping(self, dec, LlamaForCausalLM.stacked_params_mapping,
                                           tie_lm_head=self.language_model.tie)
        loaded |= load_weights_with_mapping(self, vis, [])
        return loaded


class JinaVLForRanking(Qwen2VLForConditionalGeneration):
    """jina-reranker-m0: Qwen2-VL whose last hidden state goes through a two-layer scorer; the score is
    sigmoid(logit - 2.65).  Reference: vllm/model_executor/models/jina_vl.py:30-150."""
    is_pooling_model = True
    LOGIT_BIAS = 2.65

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        from ..layers.pooler import Pooler, PoolingType
        H = int(self.config.hidden_size)
        n = int(getattr(self.config, "num_labels", 1) or 1)
        self.score = nn.Sequential(nn.Linear(H, H), nn.ReLU(), nn.Linear(H, n))
        self.pooler = Pooler.from_config(getattr(vllm_config.model_config, "pooler_config", None), PoolingType.LAST,
                                         normalize=False, softmax=True)

    def pool(self, hidden: torch.Tensor, params=None) -> torch.Tensor:
        x = self.pooler.extract(hidden)
        x = self.score(x.unsqueeze(0) if x.dim() == 1 else x).squeeze(0) - self.LOGIT_BIAS
        return self.pooler.head(x, params)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        own, rest = {}, []
        for n, t in weights:
            (own.__setitem__(n, t) if n.startswith("score.") else rest.append((n, t)))
        loaded = super().load_weights(rest)
        params = dict(self.named_parameters())
        for n, t in own.items():
            params[n].data.copy_(t)
            loaded.add(n)
        loaded |= {n for n in params if n.startswith("language_model.lm_head.")}       # unused head
        return loaded


# ---- Qwen2.5-VL --------------------------------------------------------------------------------------
class _RMS(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim), requires_grad=False)
        self.eps = eps

    def forward(self, x):
        f = x.float()
        return (f * torch.rsqrt(f.pow(2).mean(-1, keepdim=True) + self.eps)).to(x.dtype) * self.weight


class VisionBlock25(nn.Module):
    """RMSNorm, gated SiLU MLP with biases; attention spans are handed in per call (windows or frames)."""

    def __init__(self, dim: int, heads: int, mlp_dim: int):
        super().__init__()
        self.heads = heads
        self.norm1, self.norm2 = _RMS(dim), _RMS(dim)
        self.attn = nn.Module()
        self.attn.qkv, self.attn.proj = nn.Linear(dim, 3 * dim), nn.Linear(dim, dim)
        self.mlp = nn.Module()
        self.mlp.gate_proj, self.mlp.up_proj = nn.Linear(dim, mlp_dim), nn.Linear(dim, mlp_dim)
        self.mlp.down_proj = nn.Linear(mlp_dim, dim)

    def forward(self, x, cos, sin, spans: list[int]):
        N, dim = x.shape
        q, k, v = self.attn.qkv(self.norm1(x)).view(N, 3, self.heads, dim // self.heads).unbind(1)
        q = (q.float() * cos + _rotate_half(q.float()) * sin).to(x.dtype)
        k = (k.float() * cos + _rotate_half(k.float()) * sin).to(x.dtype)
        outs, o = [], 0
        for n in spans:
            qs, ks, vs = (t[o:o + n].transpose(0, 1) for t in (q, k, v))
            outs.append(F.scaled_dot_product_attention(qs, ks, vs).transpose(0, 1))
            o += n
        x = x + self.attn.proj(torch.cat(outs, dim=0).reshape(N, dim))
        h = self.norm2(x)
        return x + self.mlp.down_proj(F.silu(self.mlp.gate_proj(h)) * self.mlp.up_proj(h))


class Qwen25VisionTransformer(Qwen2VisionTransformer):
    """Most blocks attend inside windows of ``window_size`` pixels; the blocks listed in
    ``fullatt_block_indexes`` attend over the whole frame.  Tokens are reordered window by window (in units
    of one merge group) for the whole tower and put back after the merger."""

    def __init__(self, vc: dict, out_dim: int):
        nn.Module.__init__(self)
        g = vc.get
        self.embed_dim = int(g("hidden_size", 1280))
        self.heads = int(g("num_heads", 16))
        self.patch, self.tps = int(g("patch_size", 14)), int(g("temporal_patch_size", 2))
        self.merge = int(g("spatial_merge_size", 2))
        self.window = int(g("window_size", 112)) // self.merge // self.patch      # window edge in merge groups
        self.full_blocks = set(g("fullatt_block_indexes", None) or [])
        cin = int(g("in_channels", None) or g("in_chans", 3))
        self.patch_embed = nn.Module()
        self.patch_embed.proj = nn.Conv3d(cin, self.embed_dim, kernel_size=(self.tps, self.patch, self.patch),
                                          stride=(self.tps, self.patch, self.patch), bias=False)
        self.blocks = nn.ModuleList([VisionBlock25(self.embed_dim, self.heads, int(g("intermediate_size", 3420)))
                                     for _ in range(int(g("depth", 32)))])
        m = self.merge ** 2
        self.merger = nn.Module()
        self.merger.ln_q = _RMS(self.embed_dim)
        self.merger.mlp = nn.Sequential(nn.Linear(self.embed_dim * m, self.embed_dim * m), nn.GELU(),
                                        nn.Linear(self.embed_dim * m, out_dim))
        for p in self.parameters():
            p.requires_grad_(False)

    def _windows(self, grids) -> tuple[torch.Tensor, list[int]]:
        """Order of the merge groups window by window, and the length (in patches) of every window."""
        order, spans, base, ws, unit = [], [], 0, self.window, self.merge ** 2
        for t, h, w in grids:
            gh, gw = h // self.merge, w // self.merge
            idx = torch.arange(t * gh * gw).view(t, gh, gw)
            ph, pw = ws - gh % ws, ws - gw % ws
            idx = F.pad(idx, (0, pw, 0, ph), value=-1)
            nh, nw = (gh + ph) // ws, (gw + pw) // ws
            idx = idx.view(t, nh, ws, nw, ws).permute(0, 1, 3, 2, 4).reshape(t, nh * nw, ws * ws)
            for row in idx.reshape(-1, ws * ws):
                keep = row[row >= 0]
                if keep.numel():
                    order.append(keep + base)
                    spans.append(int(keep.numel()) * unit)
            base += t * gh * gw
        return torch.cat(order), spans

    def forward(self, patches: torch.Tensor, grids: list[tuple[int, int, int]]) -> torch.Tensor:
        w = self.patch_embed.proj.weight
        x = F.linear(patches.to(w.dtype), w.view(self.embed_dim, -1))
        cos, sin = self._rotary(grids, x.device)
        order, win_spans = self._windows(grids)
        order = order.to(x.device)
        unit = self.merge ** 2
        regroup = lambda t: t.reshape(t.shape[0] // unit, unit, *t.shape[1:])[order].reshape(t.shape)   # noqa: E731
        x, cos, sin = regroup(x), regroup(cos), regroup(sin)
        frame_spans = [h * w_ for t, h, w_ in grids for _ in range(t)]
        for i, blk in enumerate(self.blocks):
            x = blk(x, cos, sin, frame_spans if i in self.full_blocks else win_spans)
        out = self.merger.mlp(self.merger.ln_q(x).view(-1, self.embed_dim * unit))
        return out[torch.argsort(order)]


class Qwen2_5_VLForConditionalGeneration(Qwen2VLForConditionalGeneration):
    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        nn.Module.__init__(self)
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vc = hf.vision_config if isinstance(hf.vision_config, dict) else hf.vision_config.to_dict()
        self.merge = int(vc.get("spatial_merge_size", 2))
        self.image_token_id = int(getattr(hf, "image_token_id", 151655))
        self.tokens_per_second = float(vc.get("tokens_per_second", 2) or 2)     # video frames are spaced by their time
        self.visual = Qwen25VisionTransformer(vc, int(vc.get("out_hidden_size", None) or hf.hidden_size))
        self.language_model = LlamaForCausalLM(vllm_config, prefix=maybe_prefix(prefix, "language_model"))
        self.rotary = self.language_model.model.layers[self.language_model.model.start_layer].self_attn.rotary_emb
