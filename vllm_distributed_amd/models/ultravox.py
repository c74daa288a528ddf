"""Ultravox: Whisper encoder over the clip's log-mel frames -> ``stack_factor`` neighbouring encoder frames stacked
on the channel axis -> RMSNorm, linear, SwiGLU (or another activation), RMSNorm (in the middle from v0.5 on, at the
end before), linear -> the text decoder's embedding space.
Reference: vllm/model_executor/models/ultravox.py (StackAudioFrames :240-257, UltravoxProjector :260-297,
ModifiedWhisperEncoder :300-400, UltravoxModel :403-670).
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from .llava_next import build_language_model
from .qwen2_audio import WhisperEncoder, _cfg
from .utils import load_weights_with_mapping, maybe_prefix


class _RMS(nn.Module):
    def __init__(self, n: int, eps: float = 1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n), requires_grad=False)
        self.eps = eps

    def forward(self, x):
        f = x.float()
        return (f * torch.rsqrt(f.pow(2).mean(-1, keepdim=True) + self.eps)).to(x.dtype) * self.weight


class UltravoxProjector(nn.Module):
    def __init__(self, audio_hidden: int, hidden: int, text_hidden: int, stack: int, act: str, ln_mid: bool):
        super().__init__()
        from ..layers.basic import get_act_fn
        self.stack = stack
        self.ln_pre = _RMS(audio_hidden * stack)
        self.linear_1 = nn.Linear(audio_hidden * stack, hidden, bias=False)
        self.swiglu = act == "swiglu"
        mid = hidden // 2 if self.swiglu else hidden
        self.act = None if self.swiglu else get_act_fn(act)
        self.linear_2 = nn.Linear(mid, text_hidden, bias=False)
        self.ln_mid = _RMS(mid) if ln_mid else nn.Identity()
        self.ln_post = nn.Identity() if ln_mid else _RMS(text_hidden)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, T, C = x.shape
        pad = -T % self.stack
        x = F.pad(x, (0, 0, 0, pad)).reshape(B, (T + pad) // self.stack, C * self.stack)
        h = self.linear_1(self.ln_pre(x))
        if self.swiglu:
            a, b = h.chunk(2, dim=-1)
            h = a * F.silu(b)
        else:
            h = self.act(h)
        return self.ln_post(self.linear_2(self.ln_mid(h)))


class UltravoxModel(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        ac, tc = _cfg(hf.audio_config), _cfg(hf.text_config)
        self.audio_token_id = int(getattr(hf, "audio_token_index", None) or getattr(hf, "audio_token_id", 32000))
        self.stack = int(getattr(hf, "stack_factor", 8))
        self.audio_tower = WhisperEncoder(ac, avg_pool=False)
        self.multi_modal_projector = UltravoxProjector(
            int(ac.d_model), int(getattr(hf, "top_hidden_size", None) or getattr(hf, "hidden_size", 4096)),
            int(tc.hidden_size), self.stack,
            str(getattr(hf, "projector_act", "swiglu")), bool(getattr(hf, "projector_ln_mid", False)))
        self.language_model = build_language_model(vllm_config, tc, maybe_prefix(prefix, "language_model"))

    def encode_mm_items(self, items: list[list[torch.Tensor]]) -> list[torch.Tensor]:
        """items: [log-mel features [n_mel, T_mel], valid mel frames (scalar)] per clip -> [tokens_i, H_text]."""
        feats = torch.stack([it[0] for it in items])
        mel_len = torch.stack([it[1].reshape(()) for it in items]).long()
        frames = (mel_len - 1) // 2 + 1
        tokens = torch.clamp((frames + self.stack - 1) // self.stack, min=1)
        out = self.multi_modal_projector(self.audio_tower(feats, frames))
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
        dec, enc = [], []
        for n, t in weights:
            if n.startswith("audio_tower.model.encoder."):
                n = "audio_tower." + n[len("audio_tower.model.encoder."):]
            (dec if n.startswith("language_model.") else enc).append((n, t))
        lm = self.language_model
        loaded = load_weights_with_mapping(self, dec, lm.stacked_params_mapping, tie_lm_head=lm.tie)
        loaded |= load_weights_with_mapping(self, enc, [])
        return loaded


def ultravox_processor(vllm_config):
    from ..multimodal import AudioTokenProcessor
    hf = vllm_config.model_config.hf_config
    ac = _cfg(hf.audio_config)
    stack = int(getattr(hf, "stack_factor", 8))

    class _P(AudioTokenProcessor):
        def num_tokens(self, mel_frames: int) -> int:
            frames = (mel_frames - 1) // 2 + 1
            return max((frames + stack - 1) // stack, 1)

    return _P(int(getattr(hf, "audio_token_index", None) or getattr(hf, "audio_token_id", 32000)),
              int(ac.num_mel_bins))
