"""BART: a text encoder-decoder.  The encoder reads the request's source tokens once (bidirectional attention,
learned positions); the decoder is served like any decoder here (paged self-attention cache, chunked prefill,
hipGraph decode) and reads the encoder's states through cross-attention whose keys / values sit in a per-request
slot cache with paged layout, written in the step that holds the request's first decoder token and read by the
paged attention kernels with a per-slot length -- the mechanism of models/whisper.py with variable source length.

A request's plain prompt is the ENCODER input; the decoder starts from ``decoder_start_token_id`` (an explicit
decoder prompt can be given as ``{"prompt_token_ids": decoder ids, "multi_modal_data": {"encoder_token_ids": ...}}``).
Reference: vllm/model_executor/models/bart.py (BartEncoderLayer :371-440, BartDecoderLayer :443-560,
BartForConditionalGeneration :870-1010), vllm/inputs/preprocess.py (encoder / decoder prompt split).
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
from ..layers.attention import Attention
from ..layers.basic import LayerNorm, LogitsProcessor, ParallelLMHead, VocabParallelEmbedding
from ..layers.linear import ColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from ..parallel.state import get_tensor_model_parallel_world_size
from .utils import load_weights_with_mapping
from .whisper import WhisperCrossAttention

POS_OFFSET = 2          # BART's learned position tables start two rows in


class BartEncoderLayer(nn.Module):
    """Post-norm block, plain PyTorch (the encoder runs once per request, outside the paged cache)."""

    def __init__(self, c):
        super().__init__()
        d, self.heads = int(c.d_model), int(c.encoder_attention_heads)
        self.self_attn = nn.Module()
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            setattr(self.self_attn, n, nn.Linear(d, d))
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.fc1, self.fc2 = nn.Linear(d, int(c.encoder_ffn_dim)), nn.Linear(int(c.encoder_ffn_dim), d)
        self.final_layer_norm = nn.LayerNorm(d)
        self.act = str(getattr(c, "activation_function", "gelu"))

    def forward(self, x):
        L, d = x.shape
        a = self.self_attn
        q, k, v = (p(x).view(L, self.heads, d // self.heads).transpose(0, 1) for p in (a.q_proj, a.k_proj, a.v_proj))
        o = F.scaled_dot_product_attention(q[None], k[None], v[None])[0].transpose(0, 1).reshape(L, d)
        x = self.self_attn_layer_norm(x + a.out_proj(o))
        return self.final_layer_norm(x + self.fc2(ops.activation(self.fc1(x), self.act)))


class BartCrossAttention(WhisperCrossAttention):
    """Cross-attention over a source of any length up to ``max_position_embeddings``: the number of keys of a
    slot is recorded when its keys are written."""

    def __init__(self, config, vllm_config: VllmConfig, prefix: str):
        super().__init__(config, vllm_config, prefix)
        d = int(config.d_model)
        self.k_proj = ColumnParallelLinear(d, d, bias=True, prefix=f"{prefix}.k_proj")
        self.register_buffer("slot_lens", torch.zeros(self.lens.shape[0], dtype=torch.int32), persistent=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ctx = get_forward_context()
        md = ctx.attn_metadata
        T = x.shape[0]
        q = self.q_proj(x)
        c = self.cache
        if md is None or c.k_cache is None or md.state_table is None:
            return self.out_proj(torch.zeros_like(q))                 # memory-profiling run
        for seq, enc in ctx.extra.get("encoder_outputs", ()):      # requests that start in this step
            slot = md.state_table[seq, 0]
            c.write(slot, self.k_proj(enc), self.v_proj(enc))
            self.slot_lens[slot.long()] = enc.shape[0]
        nd, ndt = md.num_decode_seqs, md.num_decode_tokens
        S = nd + md.num_prefill_seqs
        slots = md.state_table[:S, 0]
        bt = c.block_table(slots)
        lens = self.slot_lens[slots.long()]
        q3 = q.view(T, self.num_heads, self.head_dim)
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
        return self.out_proj(out.view(T, -1))


class _CrossCfg:
    """The attribute names WhisperCrossAttention reads, filled from a BART config."""

    def __init__(self, c):
        self.d_model, self.decoder_attention_heads = c.d_model, c.decoder_attention_heads
        self.max_source_positions = int(c.max_position_embeddings)


class BartDecoderLayer(nn.Module):
    def __init__(self, c, vllm_config: VllmConfig, prefix: str):
        super().__init__()
        d, nh = int(c.d_model), int(c.decoder_attention_heads)
        tp = get_tensor_model_parallel_world_size()
        self.self_attn = nn.Module()
        self.self_attn.qkv_proj = QKVParallelLinear(d, d // nh, nh, nh, bias=True, prefix=f"{prefix}.self_attn.qkv_proj")
        self.self_attn.out_proj = RowParallelLinear(d, d, bias=True, prefix=f"{prefix}.self_attn.out_proj")
        self.self_attn.attn = Attention(nh // tp, d // nh, (d // nh) ** -0.5, nh // tp, prefix=f"{prefix}.self_attn.attn",
                                        cache_dtype=vllm_config.cache_config.cache_dtype)
        self.self_attn_layer_norm = LayerNorm(d, 1e-5)
        self.encoder_attn = BartCrossAttention(_CrossCfg(c), vllm_config, f"{prefix}.encoder_attn")
        self.encoder_attn_layer_norm = LayerNorm(d, 1e-5)
        self.fc1 = ColumnParallelLinear(d, int(c.decoder_ffn_dim), bias=True, prefix=f"{prefix}.fc1")
        self.fc2 = RowParallelLinear(int(c.decoder_ffn_dim), d, bias=True, prefix=f"{prefix}.fc2")
        self.final_layer_norm = LayerNorm(d, 1e-5)
        self.act = str(getattr(c, "activation_function", "gelu"))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        a = self.self_attn
        x = self.self_attn_layer_norm(x + a.out_proj(a.attn(a.qkv_proj(x), None, None)))
        x = self.encoder_attn_layer_norm(x + self.encoder_attn(x))
        return self.final_layer_norm(x + self.fc2(ops.activation(self.fc1(x), self.act)))


class BartForConditionalGeneration(nn.Module):
    is_multimodal = True            # the source text travels as the request's encoder item
    is_encoder_decoder = True
    is_text_encoder_decoder = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        c = vllm_config.model_config.hf_config
        self.config = c
        d = int(c.d_model)
        m = self.model = nn.Module()
        m.shared = VocabParallelEmbedding(c.vocab_size, d)
        enc = m.encoder = nn.Module()
        enc.embed_positions = nn.Embedding(int(c.max_position_embeddings) + POS_OFFSET, d)
        enc.layernorm_embedding = nn.LayerNorm(d)
        enc.layers = nn.ModuleList([BartEncoderLayer(c) for _ in range(int(c.encoder_layers))])
        dec = m.decoder = nn.Module()
        dec.embed_positions = nn.Embedding(int(c.max_position_embeddings) + POS_OFFSET, d)
        dec.layernorm_embedding = LayerNorm(d, 1e-5)
        dec.layers = nn.ModuleList([BartDecoderLayer(c, vllm_config, f"model.decoder.layers.{i}")
                                    for i in range(int(c.decoder_layers))])
        self.embed_scale = math.sqrt(d) if getattr(c, "scale_embedding", False) else 1.0
        self.lm_head = ParallelLMHead(c.vocab_size, d)
        self.lm_head.tie_weights(m.shared)
        self.register_buffer("final_logits_bias", torch.zeros(1, c.vocab_size), persistent=False)
        self.logits_processor = LogitsProcessor(c.vocab_size)

    # ---- encoder ----------------------------------------------------------------------------------
    def encode_mm_items(self, items: list[list[torch.Tensor]]) -> list[torch.Tensor]:
        """items: [source token ids [L]] per request -> encoder states [L, d] per request."""
        enc, out = self.model.encoder, []
        for it in items:
            ids = it[0].long().view(-1)
            pos = torch.arange(ids.shape[0], device=ids.device) + POS_OFFSET
            x = enc.layernorm_embedding(self.model.shared(ids) * self.embed_scale + enc.embed_positions(pos))
            for layer in enc.layers:
                x = layer(x)
            out.append(x)
        return out

    def get_input_embeddings(self, input_ids, mm_embeds=None, mm_mask=None):
        return self.model.shared(input_ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        dec = self.model.decoder
        x = inputs_embeds if inputs_embeds is not None else self.model.shared(input_ids)
        if self.embed_scale != 1.0:
            x = x * self.embed_scale
        x = dec.layernorm_embedding(x + dec.embed_positions(positions[: x.shape[0]] + POS_OFFSET))
        for layer in dec.layers:
            x = layer(x)
        return x

    def compute_logits(self, hidden):
        logits = self.logits_processor(self.lm_head, hidden)
        return logits + self.final_logits_bias.to(logits.dtype) if logits is not None else logits

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        stacked = [(".self_attn.qkv_proj", ".self_attn.q_proj", "q"), (".self_attn.qkv_proj", ".self_attn.k_proj", "k"),
                   (".self_attn.qkv_proj", ".self_attn.v_proj", "v")]
        dec, rest = [], []
        for n, t in weights:
            if n == "final_logits_bias":
                self.final_logits_bias.copy_(t.reshape(self.final_logits_bias.shape))
                continue
            if n.startswith("lm_head.") or n.endswith(".embed_tokens.weight"):
                continue                                              # all tied to model.shared
            if not n.startswith("model."):
                n = "model." + n                                      # BartModel checkpoints carry no wrapper level
            (dec if ".decoder." in n and ".self_attn." in n else rest).append((n, t))
        loaded = load_weights_with_mapping(self, dec, stacked, tie_lm_head=True)
        loaded |= load_weights_with_mapping(self, rest, [], tie_lm_head=True)
        return loaded


class BartModel(BartForConditionalGeneration):
    """The bare encoder-decoder checkpoint (no ``model.`` level, head tied to the embedding)."""
