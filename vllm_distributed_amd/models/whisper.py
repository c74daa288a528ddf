"""Whisper: audio encoder + text decoder with cross-attention (the encoder-decoder family).

How the decoder's cross-attention maps onto the paged engine: the encoder runs once per request (with the
first decoder step), and every decoder layer projects its output to K and V right then and stores them in
a cross-attention cache that has the LAYOUT OF THE PAGED KV CACHE: one run of ceil(frames / block_size)
blocks per batch slot.  A sequence's block table for the cross-attention is computed from its slot number,
its length is the (fixed) number of encoder frames, and the ordinary paged MFMA attention kernels serve
the cross-attention of decode steps (captured in the decode graphs like everything else) and of the prompt
tokens (non-causal prefill form).  The slots are the per-request state slots the runner already keeps for
hybrid models, so their memory is part of what memory profiling sees.

Reference: vllm/model_executor/models/whisper.py (attention :80-330, encoder / decoder :330-560, processor
:560-700), vllm/attention/layer.py (cross-attention path of the V0 engine).
"""
from __future__ import annotations

import math
from typing import Iterable, Optional

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
from ..sched.kv_cache_interface import KVCacheSpec, MambaSpec
from .qwen2_audio import WhisperEncoder
from .utils import load_weights_with_mapping


class CrossAttentionCache(nn.Module):
    """K/V of the encoder output, per batch slot, in paged-cache layout (see the module docstring).
    The runner allocates it through the state-slot protocol of hybrid models."""
    is_state_space_layer = True

    def __init__(self, num_kv_heads: int, head_size: int, frames: int, block_size: int, dtype: torch.dtype,
                 max_model_len: int):
        super().__init__()
        self.h, self.d, self.frames, self.bs = num_kv_heads, head_size, frames, block_size
        self.nb = -(-frames // block_size)
        self.conv_dtype = dtype
        self.state_dtypes = (dtype, dtype)
        self.max_model_len = max_model_len
        self.k_cache: Optional[torch.Tensor] = None
        self.v_cache: Optional[torch.Tensor] = None
        self.layer_name = ""
        self.register_buffer("block_iota", torch.arange(self.nb, dtype=torch.int32), persistent=False)
        self.register_buffer("frame_iota", torch.arange(frames, dtype=torch.int64), persistent=False)

    def state_shapes(self) -> tuple:
        return (self.nb, self.h, self.bs, self.d), (self.nb, self.h, self.d, self.bs)

    def get_kv_cache_spec(self, block_size: int, dtype: torch.dtype) -> KVCacheSpec:
        return MambaSpec(block_size=self.max_model_len, shapes=self.state_shapes(), dtypes=self.state_dtypes)

    def bind_kv_cache(self, k: torch.Tensor, v: torch.Tensor) -> None:
        self.k_cache = k.view(-1, self.h, self.bs, self.d)
        self.v_cache = v.view(-1, self.h, self.d, self.bs)

    def write(self, slot: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> None:
        """k, v [frames, h * d] of one request -> the blocks of batch slot ``slot`` (0-d device tensor)."""
        slots = slot.to(torch.int64) * (self.nb * self.bs) + self.frame_iota[: k.shape[0]]
        ops.reshape_and_cache(k.view(-1, self.h, self.d), v.view(-1, self.h, self.d), self.k_cache, self.v_cache,
                              slots)

    def block_table(self, slots: torch.Tensor) -> torch.Tensor:
        return slots.view(-1, 1).to(torch.int32) * self.nb + self.block_iota.view(1, -1)


class WhisperCrossAttention(nn.Module):
    def __init__(self, config, vllm_config: VllmConfig, prefix: str):
        super().__init__()
        d, nh = int(config.d_model), int(config.decoder_attention_heads)
        tp = get_tensor_model_parallel_world_size()
        self.num_heads, self.head_dim = nh // tp, d // nh
        self.scale = self.head_dim ** -0.5
        self.frames = int(config.max_source_positions)
        self.q_proj = ColumnParallelLinear(d, d, bias=True, prefix=f"{prefix}.q_proj")
        self.k_proj = ColumnParallelLinear(d, d, bias=False, prefix=f"{prefix}.k_proj")
        self.v_proj = ColumnParallelLinear(d, d, bias=True, prefix=f"{prefix}.v_proj")
        self.out_proj = RowParallelLinear(d, d, bias=True, prefix=f"{prefix}.out_proj")
        self.cache = CrossAttentionCache(self.num_heads, self.head_dim, self.frames, vllm_config.cache_config.block_size,
                                         vllm_config.model_config.dtype, vllm_config.model_config.max_model_len)
        if self.head_dim not in (64, 128, 256):
            raise ValueError(f"cross-attention heads of {self.head_dim} values are not supported (64 / 128 / 256)")
        n = max(vllm_config.scheduler_config.max_num_seqs, 1) + 512
        self.register_buffer("lens", torch.full((n,), self.frames, dtype=torch.int32), persistent=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ctx = get_forward_context()
        md = ctx.attn_metadata
        T = x.shape[0]
        q = self.q_proj(x)
        c = self.cache
        if md is None or c.k_cache is None or md.state_table is None:
            return self.out_proj(torch.zeros_like(q))                 # memory-profiling run
        for seq, enc in ctx.extra.get("encoder_outputs", ()):      # requests that start in this step
            c.write(md.state_table[seq, 0], self.k_proj(enc), self.v_proj(enc))
        nd, ndt = md.num_decode_seqs, md.num_decode_tokens
        S = nd + md.num_prefill_seqs
        bt = c.block_table(md.state_table[:S, 0])
        lens = self.lens[:S]
        q3 = q.view(T, self.num_heads, self.head_dim)
        out = torch.empty_like(q3)
        if T > md.num_actual_tokens:
            out[md.num_actual_tokens:].zero_()
        if nd > 0:
            # one query row per sequence placed "at the end" of its frames: it sees all of them
            ops.paged_attention_decode(out[:ndt], q3[:ndt], c.k_cache, c.v_cache, bt[:nd], lens[:nd],
                                       md.query_start_loc[:nd + 1], self.scale, num_splits=md.decode_num_splits,
                                       max_q_len=md.max_decode_query_len, workspace=md.decode_workspace)
        if md.num_prefill_seqs > 0:
            e = ndt + md.num_prefill_tokens
            ops.paged_attention_prefill(out[ndt:e], q3[ndt:e], c.k_cache, c.v_cache, bt[nd:], lens[nd:S],
                                        md.prefill_query_start_loc, self.scale, causal=False)
        return self.out_proj(out.view(T, -1))


class WhisperDecoderLayer(nn.Module):
    def __init__(self, config, vllm_config: VllmConfig, prefix: str):
        super().__init__()
        d, nh = int(config.d_model), int(config.decoder_attention_heads)
        tp = get_tensor_model_parallel_world_size()
        self.self_attn = nn.Module()
        self.self_attn.qkv_proj = QKVParallelLinear(d, d // nh, nh, nh, bias=True, prefix=f"{prefix}.self_attn.qkv_proj")
        self.self_attn.qkv_proj.bias.data.zero_()                     # the key projection has no bias
        self.self_attn.out_proj = RowParallelLinear(d, d, bias=True, prefix=f"{prefix}.self_attn.out_proj")
        self.self_attn.attn = Attention(nh // tp, d // nh, (d // nh) ** -0.5, nh // tp, prefix=f"{prefix}.self_attn.attn",
                                        cache_dtype=vllm_config.cache_config.cache_dtype)
        self.self_attn_layer_norm = LayerNorm(d, 1e-5)
        self.encoder_attn = WhisperCrossAttention(config, vllm_config, f"{prefix}.encoder_attn")
        self.encoder_attn_layer_norm = LayerNorm(d, 1e-5)
        self.fc1 = ColumnParallelLinear(d, int(config.decoder_ffn_dim), bias=True, prefix=f"{prefix}.fc1")
        self.fc2 = RowParallelLinear(int(config.decoder_ffn_dim), d, bias=True, prefix=f"{prefix}.fc2")
        self.final_layer_norm = LayerNorm(d, 1e-5)
        self.act = str(getattr(config, "activation_function", "gelu"))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        a = self.self_attn
        x = x + a.out_proj(a.attn(a.qkv_proj(self.self_attn_layer_norm(x)), None, None))
        x = x + self.encoder_attn(self.encoder_attn_layer_norm(x))
        return x + self.fc2(ops.activation(self.fc1(self.final_layer_norm(x)), self.act))


class WhisperForConditionalGeneration(nn.Module):
    is_multimodal = True
    is_encoder_decoder = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        c = vllm_config.model_config.hf_config
        self.config = c
        d = int(c.d_model)
        self.model = nn.Module()
        self.model.encoder = WhisperEncoder(c, avg_pool=False)
        dec = self.model.decoder = nn.Module()
        dec.embed_tokens = VocabParallelEmbedding(c.vocab_size, d)
        dec.embed_positions = nn.Embedding(int(c.max_target_positions), d)
        dec.layers = nn.ModuleList([WhisperDecoderLayer(c, vllm_config, f"model.decoder.layers.{i}")
                                    for i in range(int(c.decoder_layers))])
        dec.layer_norm = LayerNorm(d, 1e-5)
        self.embed_scale = math.sqrt(d) if getattr(c, "scale_embedding", False) else 1.0
        self.proj_out = ParallelLMHead(c.vocab_size, d)
        self.proj_out.tie_weights(dec.embed_tokens)
        self.logits_processor = LogitsProcessor(c.vocab_size)

    # ---- encoder ----------------------------------------------------------------------------------
    def encode_mm_items(self, items: list[list[torch.Tensor]]) -> list[torch.Tensor]:
        """items: [log-mel features [n_mel, 2 * frames]] per clip -> encoder states [frames, d] per clip."""
        feats = torch.stack([it[0] for it in items])
        out = self.model.encoder(feats)
        return [out[i] for i in range(out.shape[0])]

    def get_input_embeddings(self, input_ids, mm_embeds=None, mm_mask=None):
        return self.model.decoder.embed_tokens(input_ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        dec = self.model.decoder
        x = inputs_embeds if inputs_embeds is not None else dec.embed_tokens(input_ids)
        if self.embed_scale != 1.0:
            x = x * self.embed_scale
        x = x + dec.embed_positions(positions[: x.shape[0]])
        for layer in dec.layers:
            x = layer(x)
        return dec.layer_norm(x)

    def compute_logits(self, hidden):
        return self.logits_processor(self.proj_out, hidden)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        stacked = [(".self_attn.qkv_proj", ".self_attn.q_proj", "q"), (".self_attn.qkv_proj", ".self_attn.k_proj", "k"),
                   (".self_attn.qkv_proj", ".self_attn.v_proj", "v")]
        dec, rest = [], []
        for n, t in weights:
            if n.startswith("proj_out."):
                continue                                              # tied to the token embedding
            # only the decoder's self-attention fuses q/k/v; the encoder and the cross-attention keep them apart
            (dec if ".decoder." in n and ".self_attn." in n else rest).append((n, t))
        loaded = load_weights_with_mapping(self, dec, stacked, tie_lm_head=True)
        loaded |= load_weights_with_mapping(self, rest, [], tie_lm_head=True)
        return loaded
