"""Audio-language models: the Whisper audio encoder (two strided convolutions over the log-mel spectrogram,
learned positions, pre-norm transformer layers) and Qwen2-Audio, which average-pools the encoder output by
two, projects it to the decoder width and feeds it to a Qwen2 decoder in place of the prompt's audio
placeholder tokens.

The encoder runs once per audio clip, outside the paged-attention path (bidirectional attention over at
most 1500 frames: plain scaled-dot-product attention with the padded frames masked out).
Reference: vllm/model_executor/models/qwen2_audio.py (:60-420), whisper.py (WhisperEncoder :330-410).
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from .llama import LlamaForCausalLM
from .utils import load_weights_with_mapping, maybe_prefix


def _cfg(c):
    from ..config import HFConfig
    return c if not isinstance(c, dict) else HFConfig(c)


class WhisperEncoderLayer(nn.Module):
    def __init__(self, d: int, heads: int, ffn: int, act: str = "gelu"):
        super().__init__()
        self.heads = heads
        self.self_attn = nn.Module()
        self.self_attn.q_proj = nn.Linear(d, d, bias=True)
        self.self_attn.k_proj = nn.Linear(d, d, bias=False)
        self.self_attn.v_proj = nn.Linear(d, d, bias=True)
        self.self_attn.out_proj = nn.Linear(d, d, bias=True)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, ffn)
        self.fc2 = nn.Linear(ffn, d)
        self.final_layer_norm = nn.LayerNorm(d)
        self.act = act

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor]) -> torch.Tensor:
        B, S, d = x.shape
        a = self.self_attn
        h = self.self_attn_layer_norm(x)
        q = a.q_proj(h).view(B, S, self.heads, -1).transpose(1, 2)
        k = a.k_proj(h).view(B, S, self.heads, -1).transpose(1, 2)
        v = a.v_proj(h).view(B, S, self.heads, -1).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
        x = x + a.out_proj(o.transpose(1, 2).reshape(B, S, d))
        h = self.final_layer_norm(x)
        return x + self.fc2(F.gelu(self.fc1(h)))


class WhisperEncoder(nn.Module):
    """log-mel [B, n_mel, 2 * max_source_positions] -> [B, max_source_positions (/ 2 when pooled), d_model]"""

    def __init__(self, c, avg_pool: bool):
        super().__init__()
        d = int(c.d_model)
        self.conv1 = nn.Conv1d(int(c.num_mel_bins), d, kernel_size=3, padding=1)
        self.conv2 = nn.Conv1d(d, d, kernel_size=3, stride=2, padding=1)
        self.embed_positions = nn.Embedding(int(c.max_source_positions), d)
        self.layers = nn.ModuleList([WhisperEncoderLayer(d, int(c.encoder_attention_heads), int(c.encoder_ffn_dim))
                                     for _ in range(int(c.encoder_layers))])
        self.layer_norm = nn.LayerNorm(d)
        self.avg_pool = avg_pool

    def forward(self, features: torch.Tensor, frames: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``frames`` [B]: valid frames after the strided convolution; the others are padding no frame attends."""
        x = F.gelu(self.conv1(features.to(self.conv1.weight.dtype)))
        x = F.gelu(self.conv2(x)).permute(0, 2, 1)
        x = x + self.embed_positions.weight[: x.shape[1]]
        mask = None
        if frames is not None:
            S = x.shape[1]
            keep = torch.arange(S, device=x.device)[None, :] < frames.to(x.device)[:, None]      # [B, S]
            mask = torch.zeros((x.shape[0], 1, 1, S), dtype=x.dtype, device=x.device)
            mask.masked_fill_(~keep[:, None, None, :], torch.finfo(x.dtype).min)
        for layer in self.layers:
            x = layer(x, mask)
        if self.avg_pool:
            x = F.avg_pool1d(x.permute(0, 2, 1), 2, stride=2).permute(0, 2, 1)
        return self.layer_norm(x)


class Qwen2AudioForConditionalGeneration(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        ac, tc = _cfg(hf.audio_config), _cfg(hf.text_config)
        self.audio_token_id = int(getattr(hf, "audio_token_index", None) or getattr(hf, "audio_token_id", 151646))
        self.audio_tower = WhisperEncoder(ac, avg_pool=True)
        self.multi_modal_projector = nn.Module()
        self.multi_modal_projector.linear = nn.Linear(int(ac.d_model), int(tc.hidden_size), bias=True)
        import copy
        sub = copy.copy(vllm_config)
        sub.model_config = copy.copy(vllm_config.model_config)
        sub.model_config.hf_config = tc
        self.language_model = LlamaForCausalLM(sub, prefix=maybe_prefix(prefix, "language_model"))

    # ---- encoder ----------------------------------------------------------------------------------
    def encode_mm_items(self, items: list[list[torch.Tensor]]) -> list[torch.Tensor]:
        """items: [log-mel features [n_mel, T_mel], valid mel frames (scalar)] per clip ->
        [tokens_i, H_text] per clip."""
        feats = torch.stack([it[0] for it in items])
        mel_len = torch.stack([it[1].reshape(()) for it in items]).long()
        frames = (mel_len - 1) // 2 + 1
        tokens = (frames - 2) // 2 + 1
        out = self.multi_modal_projector.linear(self.audio_tower(feats, frames))
        return [out[i, : int(tokens[i])] for i in range(len(items))]

    def get_input_embeddings(self, input_ids, mm_embeds: Optional[torch.Tensor] = None,
                             mm_mask: Optional[torch.Tensor] = None):
        x = self.language_model.get_input_embeddings(input_ids)
        if mm_embeds is not None and mm_embeds.numel():
            x[mm_mask] = mm_embeds.to(x.dtype)
        return x

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        return self.language_model(input_ids, positions, intermediate_tensors, inputs_embeds)

    def compute_logits(self, hidden_states):
        return self.language_model.compute_logits(hidden_states)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        def rename(n: str):
            if n.startswith("model."):                                  # layout with one more level
                n = n[len("model."):]
                if n.startswith("language_model."):
                    n = "language_model.model." + n[len("language_model."):]
            n = n.replace("language_model.model.model.", "language_model.model.")
            if n.startswith("lm_head."):
                n = "language_model." + n
            return n
        stacked = [(".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"), (".qkv_proj", ".v_proj", "v"),
                   (".gate_up_proj", ".gate_proj", 0), (".gate_up_proj", ".up_proj", 1)]
        dec, enc = [], []
        for n, t in weights:
            r = rename(n)
            (dec if r.startswith("language_model.") else enc).append((r, t))
        loaded = load_weights_with_mapping(self, dec, stacked, tie_lm_head=self.language_model.tie)
        loaded |= load_weights_with_mapping(self, enc, [])
        return loaded
