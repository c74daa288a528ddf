"""Llama-3.2-Vision (Mllama): a Llama decoder in which some layers do not attend to the text but to the image:
tanh-gated cross-attention over the projected output of a two-stage vision transformer.  The image is cut into up
to ``max_num_tiles`` tiles of 448 pixels; the tower adds tile-position embeddings chosen by the aspect ratio, runs
32 layers per tile sequence and 8 gated "global" layers, and hands out the last state concatenated with five
intermediate ones.

The cross-attention keys and values of a request are computed once, in the step that holds its first token, and
kept in a per-request slot cache in paged layout (the machinery of the encoder-decoder models), so the paged MFMA
kernels serve these layers too.  A row sees the image from its ``<|image|>`` token on; earlier rows and text-only
requests pass the cross-attention layers unchanged.  One image per request.
Reference: vllm/model_executor/models/mllama.py (MllamaVisionModel :626-800, MllamaTextCrossAttention :805-1020,
MllamaCrossAttentionDecoderLayer :1023-1077, MllamaForConditionalGeneration :1259-1600).
"""
from __future__ import annotations

import math
from typing import Iterable

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..config import VllmConfig
from ..forward_context import get_forward_context
from ..layers.basic import LogitsProcessor, ParallelLMHead, RMSNorm, VocabParallelEmbedding, get_act_fn
from ..layers.linear import ColumnParallelLinear, RowParallelLinear
from ..parallel.state import get_tensor_model_parallel_world_size
from .llama import LlamaDecoderLayer, LlamaMLP
from .llava import _cfg
from .utils import load_weights_with_mapping
from .whisper import CrossAttentionCache


# ---- vision tower --------------------------------------------------------------------------------------------
class _TileEmbedding(nn.Module):
    """One learned vector per (aspect ratio, tile), gated."""

    def __init__(self, c):
        super().__init__()
        self.tiles, self.d = int(c.max_num_tiles), int(c.hidden_size)
        self.embedding = nn.Embedding(int(c.max_aspect_ratio_id) + 1, self.tiles * self.d)
        self.gate = nn.Parameter(torch.zeros(1))

    def forward(self, x, ratio_id):                                   # x [tiles, n, d]
        return x + self.embedding(ratio_id).view(self.tiles, 1, self.d) * self.gate.tanh()


class _PositionEmbedding(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.tiles, self.d = int(c.max_num_tiles), int(c.hidden_size)
        self.n = (int(c.image_size) // int(c.patch_size)) ** 2 + 1
        self.gate = nn.Parameter(torch.zeros(1))
        self.embedding = nn.Parameter(torch.zeros(self.n, self.d))
        self.tile_embedding = nn.Embedding(int(c.max_aspect_ratio_id) + 1, self.tiles * self.n * self.d)

    def forward(self, x, ratio_id):                                   # x [tiles, n, d]
        x = x + (1 - self.gate.tanh()) * self.embedding[None]
        return x + self.gate.tanh() * self.tile_embedding(ratio_id).view(self.tiles, self.n, self.d)


class _VisionLayer(nn.Module):
    def __init__(self, c, gated: bool):
        super().__init__()
        d, self.heads = int(c.hidden_size), int(c.attention_heads)
        self.self_attn = nn.Module()
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            setattr(self.self_attn, n, nn.Linear(d, d, bias=False))
        self.mlp = nn.Module()
        self.mlp.fc1, self.mlp.fc2 = nn.Linear(d, int(c.intermediate_size)), nn.Linear(int(c.intermediate_size), d)
        eps = float(getattr(c, "norm_eps", 1e-5))
        self.input_layernorm, self.post_attention_layernorm = nn.LayerNorm(d, eps=eps), nn.LayerNorm(d, eps=eps)
        self.act = get_act_fn(getattr(c, "hidden_act", "gelu"))
        if gated:
            self.gate_attn, self.gate_ffn = nn.Parameter(torch.zeros(1)), nn.Parameter(torch.zeros(1))
        self.gated = gated

    def forward(self, x, mask):                                       # x [1, N, d]
        a = self.self_attn
        h = self.input_layernorm(x)
        q, k, v = (p(h).view(1, -1, self.heads, h.shape[-1] // self.heads).transpose(1, 2)
                   for p in (a.q_proj, a.k_proj, a.v_proj))
        o = a.o_proj(F.scaled_dot_product_attention(q, k, v, attn_mask=mask).transpose(1, 2).reshape(x.shape))
        x = x + (self.gate_attn.tanh() * o if self.gated else o)
        o = self.mlp.fc2(self.act(self.mlp.fc1(self.post_attention_layernorm(x))))
        return x + (self.gate_ffn.tanh() * o if self.gated else o)


class MllamaVisionModel(nn.Module):
    def __init__(self, c):
        super().__init__()
        d, p = int(c.hidden_size), int(c.patch_size)
        self.tiles = int(c.max_num_tiles)
        self.n = (int(c.image_size) // p) ** 2 + 1
        # hidden state ENTERING layer i, as in the original model (and the reference); see the module docstring of
        # the test for how recent transformers versions count
        self.taps = [int(i) for i in c.intermediate_layers_indices]
        self.patch_embedding = nn.Conv2d(int(getattr(c, "num_channels", 3)), d, p, p, bias=False)
        self.class_embedding = nn.Parameter(torch.zeros(d))
        self.gated_positional_embedding = _PositionEmbedding(c)
        self.pre_tile_positional_embedding, self.post_tile_positional_embedding = _TileEmbedding(c), _TileEmbedding(c)
        self.layernorm_pre, self.layernorm_post = nn.LayerNorm(d), nn.LayerNorm(d)
        self.transformer = nn.Module()
        self.transformer.layers = nn.ModuleList(_VisionLayer(c, False) for _ in range(int(c.num_hidden_layers)))
        self.global_transformer = nn.Module()
        self.global_transformer.layers = nn.ModuleList(_VisionLayer(c, True) for _ in range(int(c.num_global_layers)))

    def forward(self, pixels: torch.Tensor, ratio_id: torch.Tensor, valid_tiles: int) -> torch.Tensor:
        """pixels [max_tiles, 3, S, S] (unused tiles zero), -> [max_tiles, n, d * (1 + taps)]"""
        T, n, w = self.tiles, self.n, self.patch_embedding.weight
        x = self.patch_embedding(pixels.to(w.dtype)).flatten(2).transpose(1, 2)              # [T, n - 1, d]
        x = self.pre_tile_positional_embedding(x, ratio_id)
        x = torch.cat([self.class_embedding.expand(T, 1, -1), x], 1)
        x = self.layernorm_pre(self.gated_positional_embedding(x, ratio_id))
        pad = (-n) % 8
        x = F.pad(x, (0, 0, 0, pad))
        L = n + pad
        # a pair of positions is masked only when BOTH are padding (unused tile, or the rows that pad a tile to a
        # multiple of eight): real patches do see the padding, as in the original model
        is_pad = torch.ones(T, L, dtype=w.dtype, device=w.device)
        is_pad[:valid_tiles, :n] = 0
        is_pad = is_pad.view(1, T * L, 1)
        mask = ((is_pad @ is_pad.transpose(1, 2)) * torch.finfo(w.dtype).min)[:, None]
        x = x.reshape(1, T * L, -1)
        taps = []
        for i, layer in enumerate(self.transformer.layers):
            if i in self.taps:
                taps.append(x)
            x = layer(x, mask)
        if len(self.transformer.layers) in self.taps:
            taps.append(x)
        x = self.layernorm_post(x).view(T, L, -1)
        x = self.post_tile_positional_embedding(x, ratio_id).reshape(1, T * L, -1)
        for layer in self.global_transformer.layers:
            x = layer(x, mask)
        x = x.view(T, L, -1)[:, :n]
        mid = torch.stack(taps, -1).view(T, L, -1)[:, :n]
        return torch.cat([x, mid], -1)


# ---- decoder -------------------------------------------------------------------------------------------------
class MllamaCrossAttention(nn.Module):
    def __init__(self, config, vllm_config: VllmConfig, frames: int, prefix: str):
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        d, hq, hkv = int(config.hidden_size), int(config.num_attention_heads), int(config.num_key_value_heads)
        self.head_dim = d // hq
        self.num_heads, self.num_kv_heads = hq // tp, max(1, hkv // tp)
        quant_config = getattr(vllm_config, "quant_config", None)
        self.q_proj = ColumnParallelLinear(d, hq * self.head_dim, bias=False, quant_config=quant_config,
                                           prefix=f"{prefix}.q_proj")
        self.k_proj = ColumnParallelLinear(d, hkv * self.head_dim, bias=False, quant_config=quant_config,
                                           prefix=f"{prefix}.k_proj")
        self.v_proj = ColumnParallelLinear(d, hkv * self.head_dim, bias=False, quant_config=quant_config,
                                           prefix=f"{prefix}.v_proj")
        self.o_proj = RowParallelLinear(hq * self.head_dim, d, bias=False, quant_config=quant_config,
                                        prefix=f"{prefix}.o_proj")
        self.q_norm, self.k_norm = RMSNorm(self.head_dim, config.rms_norm_eps), RMSNorm(self.head_dim, config.rms_norm_eps)
        self.scale = self.head_dim ** -0.5
        self.cache = CrossAttentionCache(self.num_kv_heads, self.head_dim, frames, vllm_config.cache_config.block_size,
                                         vllm_config.model_config.dtype, vllm_config.model_config.max_model_len)

    def forward(self, x: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
        """``lens`` [S]: image keys of every sequence of the step (at least 1: rows without an image are masked by
        the caller)."""
        ctx = get_forward_context()
        md, c = ctx.attn_metadata, self.cache
        T = x.shape[0]
        q = self.q_proj(x)
        if md is None or c.k_cache is None or md.state_table is None:
            return self.o_proj(torch.zeros_like(q))
        for seq, enc in ctx.extra.get("encoder_outputs", ()):
            k = self.k_norm(self.k_proj(enc).view(-1, self.head_dim)).view(enc.shape[0], -1)
            c.write(md.state_table[seq, 0], k, self.v_proj(enc))
        q3 = self.q_norm(q.view(T * self.num_heads, self.head_dim)).view(T, self.num_heads, self.head_dim)
        nd, ndt = md.num_decode_seqs, md.num_decode_tokens
        S = nd + md.num_prefill_seqs
        bt = c.block_table(md.state_table[:S, 0])
        out = torch.empty_like(q3)
        if T > md.num_actual_tokens:
            out[md.num_actual_tokens:].zero_()
        if nd > 0:
            ops.paged_attention_decode(out[:ndt], q3[:ndt], c.k_cache, c.v_cache, bt[:nd], lens[:nd],
                                       md.query_start_loc[:nd + 1], self.scale, num_splits=md.decode_num_splits,
                                       max_q_len=md.max_decode_query_len, workspace=md.decode_workspace)
        if md.num_prefill_seqs > 0:
            e = ndt + md.num_prefill_tokens
            ops.paged_attention_prefill(out[ndt:e], q3[ndt:e], c.k_cache, c.v_cache, bt[nd:], lens[nd:S],
                                        md.prefill_query_start_loc, self.scale, causal=False)
        return self.o_proj(out.view(T, -1))


class MllamaCrossAttentionDecoderLayer(nn.Module):
    def __init__(self, config, vllm_config: VllmConfig, frames: int, prefix: str):
        super().__init__()
        quant_config = getattr(vllm_config, "quant_config", None)
        self.cross_attn = MllamaCrossAttention(config, vllm_config, frames, f"{prefix}.cross_attn")
        self.input_layernorm = RMSNorm(config.hidden_size, config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, config.rms_norm_eps)
        self.cross_attn_attn_gate, self.cross_attn_mlp_gate = nn.Parameter(torch.zeros(1)), nn.Parameter(torch.zeros(1))
        self.mlp = LlamaMLP(config.hidden_size, config.intermediate_size, False, quant_config, f"{prefix}.mlp")

    def forward(self, x, lens, visible):
        a = self.cross_attn(self.input_layernorm(x), lens)
        x = x + self.cross_attn_attn_gate.tanh().to(x.dtype) * (a * visible)
        m = self.mlp(self.post_attention_layernorm(x))
        return x + self.cross_attn_mlp_gate.tanh().to(x.dtype) * (m * visible)


class MllamaForConditionalGeneration(nn.Module):
    is_multimodal = True
    is_encoder_decoder = True
    supports_pp = False
    supports_lora = False
    stacked_params_mapping = [(".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"), (".qkv_proj", ".v_proj", "v"),
                              (".gate_up_proj", ".gate_proj", 0), (".gate_up_proj", ".up_proj", 1)]

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        import copy
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vc, tc = _cfg(hf.vision_config), _cfg(hf.text_config)
        if getattr(vc, "max_aspect_ratio_id", None) is None:
            vc.max_aspect_ratio_id = len(vc.supported_aspect_ratios)
        self.image_token_id = int(getattr(hf, "image_token_index", None) or getattr(hf, "image_token_id", 128256))
        self.vision_model = MllamaVisionModel(vc)
        self.multi_modal_projector = nn.Linear(int(vc.vision_output_dim), int(tc.hidden_size))
        frames = int(vc.max_num_tiles) * self.vision_model.n
        sub = copy.copy(vllm_config)
        sub.model_config = copy.copy(vllm_config.model_config)
        sub.model_config.hf_config = tc
        lm = self.language_model = nn.Module()
        lm.model = nn.Module()
        lm.model.embed_tokens = VocabParallelEmbedding(int(tc.vocab_size) + 8, int(tc.hidden_size))
        cross = set(int(i) for i in tc.cross_attention_layers)
        lm.model.layers = nn.ModuleList(
            MllamaCrossAttentionDecoderLayer(tc, vllm_config, frames, f"language_model.model.layers.{i}") if i in cross
            else LlamaDecoderLayer(tc, sub, getattr(vllm_config, "quant_config", None), f"language_model.model.layers.{i}")
            for i in range(int(tc.num_hidden_layers)))
        lm.model.norm = RMSNorm(tc.hidden_size, tc.rms_norm_eps)
        lm.lm_head = ParallelLMHead(int(tc.vocab_size), int(tc.hidden_size))
        self.logits_processor = LogitsProcessor(int(tc.vocab_size))
        n = max(vllm_config.scheduler_config.max_num_seqs, 1) + 512
        self.register_buffer("enc_len", torch.zeros(n, dtype=torch.int32), persistent=False)
        self.register_buffer("img_pos", torch.full((n,), 2 ** 30, dtype=torch.int64), persistent=False)

    # ---- encoder ---------------------------------------------------------------------------------------
    def encode_mm_items(self, items: list) -> list[torch.Tensor]:
        """items: (tiles [max_tiles, 3, S, S], meta [2] = (aspect-ratio id, valid tiles)) -> the projected states of
        the valid tiles, [valid * patches, H]: the keys of the cross-attention layers."""
        outs = []
        for px, meta in items:
            rid, valid = int(meta[0]), int(meta[1])
            v = self.vision_model(px, torch.tensor(rid, device=px.device), valid)
            outs.append(self.multi_modal_projector(v[:valid]).reshape(-1, self.multi_modal_projector.out_features))
        return outs

    def get_input_embeddings(self, input_ids, mm_embeds=None, mm_mask=None):
        return self.language_model.model.embed_tokens(input_ids)

    # ---- decoder ---------------------------------------------------------------------------------------
    def _visibility(self, input_ids, positions, T: int):
        """(keys per sequence [S], row mask [T, 1]) and the slot book-keeping of requests that start in this step."""
        ctx = get_forward_context()
        md = ctx.attn_metadata
        if md is None or md.state_table is None:
            return None, None
        S = md.num_decode_seqs + md.num_prefill_seqs
        slots = md.state_table[:S, 0].long()
        for seq in ctx.extra.get("starting_seqs", ()):
            self.enc_len[slots[seq]] = 0
            self.img_pos[slots[seq]] = 2 ** 30
        for seq, enc in ctx.extra.get("encoder_outputs", ()):
            self.enc_len[slots[seq]] = enc.shape[0]
        n = md.num_actual_tokens
        seq_of = torch.bucketize(torch.arange(n, device=slots.device), md.query_start_loc[1:S + 1].long(), right=True)
        seq_of = seq_of.clamp(max=S - 1)
        slot_of = slots[seq_of]
        if input_ids is not None:
            is_img = input_ids[:n] == self.image_token_id
            cand = torch.where(is_img, positions[:n].long(), torch.full_like(slot_of, 2 ** 30))
            self.img_pos.scatter_reduce_(0, slot_of, cand, reduce="amin")
        visible = torch.zeros(T, 1, dtype=torch.bool, device=slots.device)
        visible[:n, 0] = (self.enc_len[slot_of] > 0) & (positions[:n].long() >= self.img_pos[slot_of])
        return self.enc_len[slots].clamp(min=1), visible

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        m = self.language_model.model
        x = inputs_embeds if inputs_embeds is not None else m.embed_tokens(input_ids)
        lens, visible = self._visibility(input_ids, positions, x.shape[0])
        residual = None
        for layer in m.layers:
            if isinstance(layer, MllamaCrossAttentionDecoderLayer):
                if lens is None:
                    continue                                     # memory-profiling run without state slots
                if residual is not None:
                    x, residual = x + residual, None
                x = layer(x, lens, visible.to(x.dtype))
            else:
                x, residual = layer(positions, x, residual)
        if residual is None:
            return m.norm(x)
        return m.norm(x, residual)[0]

    def compute_logits(self, hidden_states):
        return self.logits_processor(self.language_model.lm_head, hidden_states)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        def fix(items):
            for n, t in items:
                if n.startswith("model.language_model."):
                    n = "language_model.model." + n[len("model.language_model."):]
                elif n.startswith("model.vision_model.") or n.startswith("model.multi_modal_projector."):
                    n = n[len("model."):]
                elif n.startswith("lm_head."):
                    n = "language_model." + n
                yield n, t
        own, dec = [], []
        for n, t in fix(weights):
            # the fused projections exist in the self-attention layers only
            (dec if n.startswith("language_model.") and ".cross_attn." not in n else own).append((n, t))
        loaded = load_weights_with_mapping(self, dec, self.stacked_params_mapping)
        loaded |= load_weights_with_mapping(self, own, [])
        return loaded


# ---- processor -----------------------------------------------------------------------------------------------
def supported_ratios(max_tiles: int) -> list[tuple[int, int]]:
    return [(a, b) for a in range(1, max_tiles + 1) for b in range(1, max_tiles + 1) if a * b <= max_tiles]


def best_canvas(h: int, w: int, max_tiles: int, tile: int) -> tuple[int, int]:
    """(canvas height, canvas width): the smallest up-scaling canvas, or the largest down-scaling one; ties go to the
    smaller area."""
    cands = [(a * tile, b * tile) for a, b in supported_ratios(max_tiles)]
    scales = [min(cw / w, ch / h) for ch, cw in cands]
    ups = [s for s in scales if s >= 1]
    pick = min(ups) if ups else max(scales)
    best = [c for c, s in zip(cands, scales) if s == pick]
    return min(best, key=lambda c: c[0] * c[1])


def mllama_processor(vllm_config):
    """The ``<|image|>`` token stays ONE prompt token (the image enters through cross-attention).  Pixels: resized into
    the best canvas (aspect kept, never below one tile), zero-padded, CLIP mean / std, cut into tiles.  Pre-processed
    inputs: a dict with ``pixel_values`` [max_tiles, 3, S, S], ``aspect_ratio_id`` and ``num_tiles``."""
    from ..multimodal import CLIP_MEAN, CLIP_STD, PlaceholderRange, hash_item
    hf = vllm_config.model_config.hf_config
    vc = _cfg(hf.vision_config)
    tile, max_tiles = int(vc.image_size), int(vc.max_num_tiles)
    tok = int(getattr(hf, "image_token_index", None) or getattr(hf, "image_token_id", 128256))
    ratios = supported_ratios(max_tiles)
    mean, std = torch.tensor(CLIP_MEAN).view(3, 1, 1), torch.tensor(CLIP_STD).view(3, 1, 1)

    def pixels(img):
        if isinstance(img, dict):
            return torch.as_tensor(img["pixel_values"]).float(), \
                torch.tensor([int(img["aspect_ratio_id"]), int(img["num_tiles"])])
        import numpy as np
        a = np.asarray(img.convert("RGB")) if hasattr(img, "convert") else np.asarray(img)
        t = torch.as_tensor(a).permute(2, 0, 1).float()[None]
        H, W = t.shape[-2:]
        ch, cw = best_canvas(H, W, max_tiles, tile)
        s = min(cw / W, ch / H)
        nh, nw = min(max(int(H * s), tile), ch), min(max(int(W * s), tile), cw)
        t = F.interpolate(t, size=(nh, nw), mode="bilinear", align_corners=False, antialias=True).clamp(0, 255)
        full = torch.zeros(3, ch, cw)
        full[:, :nh, :nw] = t[0]
        full = (full / 255.0 - mean) / std
        rows, cols = ch // tile, cw // tile
        tiles = full.view(3, rows, tile, cols, tile).permute(1, 3, 0, 2, 4).reshape(-1, 3, tile, tile)
        out = torch.zeros(max_tiles, 3, tile, tile)
        out[: rows * cols] = tiles
        return out, torch.tensor([ratios.index((rows, cols)) + 1, rows * cols])

    def process(ids, mm_data):
        images = mm_data.get("image")
        if images is None:
            raise ValueError("multi_modal_data needs an 'image' entry")
        images = images if isinstance(images, (list, tuple)) else [images]
        if len(images) != 1 or list(ids).count(tok) != 1:
            raise ValueError("one image (and one image token) per request")
        px, meta = pixels(images[0])
        key = hash_item(px if isinstance(images[0], dict) else images[0])
        # tied to the first decoder token, like the audio of an encoder-decoder speech model: the cross-attention keys
        # must exist before any row of the request runs
        return list(ids), [[px, meta]], [key], [PlaceholderRange(0, 1)]

    return process
