"""Granite Speech: a conformer encoder (block-local attention with Shaw relative positions, GLU / depth-wise
convolution module, a CTC head fed back into the stream half way up), a windowed Q-Former that turns every
``window_size`` encoder frames into ``window_size / downsample_rate`` query outputs, and a Granite decoder.

Inputs are 80-bin log-mel frames stacked in pairs (160 features every 20 ms).  The checkpoint's LoRA adapter for
audio requests is an ordinary adapter to this engine: load it through the LoRA manager.
Reference: vllm/model_executor/models/granite_speech.py (GraniteSpeechEncoderProjector :223-275,
GraniteSpeechConformer* :280-470, GraniteSpeechCTCEncoder :473-548, GraniteSpeechForConditionalGeneration
:551-790).
"""
from __future__ import annotations

import math
from typing import Iterable

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from .blip2 import Blip2QFormer
from .llava import _cfg
from .llava_next import build_language_model
from .utils import load_weights_with_mapping, maybe_prefix


class _FeedForward(nn.Module):
    def __init__(self, d: int, mult: int):
        super().__init__()
        self.pre_norm = nn.LayerNorm(d)
        self.up_proj, self.down_proj = nn.Linear(d, d * mult), nn.Linear(d * mult, d)

    def forward(self, x):
        return self.down_proj(F.silu(self.up_proj(self.pre_norm(x))))


class _BlockAttention(nn.Module):
    """Frames attend inside consecutive blocks of ``context_size`` frames; the logits get q . r(i - j) for a learned
    vector r per relative distance."""

    def __init__(self, c):
        super().__init__()
        d, self.heads, self.dh = int(c.hidden_dim), int(c.num_heads), int(c.dim_head)
        self.ctx = int(c.context_size)
        inner = self.heads * self.dh
        self.pre_norm = nn.LayerNorm(d)
        self.to_q, self.to_kv = nn.Linear(d, inner, bias=False), nn.Linear(d, 2 * inner, bias=False)
        self.to_out = nn.Linear(inner, d)
        self.rel_pos_emb = nn.Embedding(2 * int(c.max_pos_emb) + 1, self.dh)

    def forward(self, x, dists):
        """x [T, d] (one clip)"""
        T = x.shape[0]
        nb = -(-T // self.ctx)
        x = F.pad(self.pre_norm(x), (0, 0, 0, nb * self.ctx - T))
        q = self.to_q(x).view(nb, self.ctx, self.heads, self.dh).transpose(1, 2)
        k, v = (t.reshape(nb, self.ctx, self.heads, self.dh).transpose(1, 2) for t in self.to_kv(x).chunk(2, -1))
        scale = self.dh ** -0.5
        bias = torch.einsum("bhcd,crd->bhcr", q, self.rel_pos_emb(dists)) * scale
        rem = T % self.ctx
        if rem:                                       # the padded tail of the last block is no key for real frames
            bias[-1, :, :rem, rem:] = float("-inf")
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=bias, scale=scale)
        return self.to_out(o.transpose(1, 2).reshape(nb * self.ctx, -1)[:T])


class _ConvModule(nn.Module):
    def __init__(self, c):
        super().__init__()
        d, inner, ks = int(c.hidden_dim), int(c.hidden_dim) * int(c.conv_expansion_factor), int(c.conv_kernel_size)
        self.norm = nn.LayerNorm(d)
        self.up_conv = nn.Conv1d(d, 2 * inner, 1)
        self.depth_conv = nn.Module()
        self.depth_conv.conv = nn.Conv1d(inner, inner, ks, groups=inner, bias=False)
        self.pad = (ks // 2, ks // 2 - (ks + 1) % 2)
        self.batch_norm = nn.BatchNorm1d(inner)
        self.down_conv = nn.Conv1d(inner, d, 1)

    def forward(self, x):
        h = F.glu(self.up_conv(self.norm(x).t()[None]), dim=1)
        h = F.silu(self.batch_norm(self.depth_conv.conv(F.pad(h, self.pad))))
        return self.down_conv(h)[0].t()


class _ConformerBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        d, mult = int(c.hidden_dim), int(c.feedforward_mult)
        self.ff1, self.attn, self.conv, self.ff2 = _FeedForward(d, mult), _BlockAttention(c), _ConvModule(c), \
            _FeedForward(d, mult)
        self.post_norm = nn.LayerNorm(d)

    def forward(self, x, dists):
        x = x + 0.5 * self.ff1(x)
        x = x + self.attn(x, dists)
        x = x + self.conv(x)
        x = x + 0.5 * self.ff2(x)
        return self.post_norm(x)


class GraniteSpeechCTCEncoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        d = int(c.hidden_dim)
        seq = torch.arange(int(c.context_size))
        dists = (seq[:, None] - seq[None, :]).clamp(-int(c.context_size), int(c.context_size)) + int(c.max_pos_emb)
        self.register_buffer("attention_dists", dists, persistent=False)
        self.input_linear = nn.Linear(int(c.input_dim), d)
        self.layers = nn.ModuleList(_ConformerBlock(c) for _ in range(int(c.num_layers)))
        self.out, self.out_mid = nn.Linear(d, int(c.output_dim)), nn.Linear(int(c.output_dim), d)

    def forward(self, feats: torch.Tensor) -> torch.Tensor:
        """[T, input_dim] -> [T, hidden_dim]"""
        x = self.input_linear(feats)
        for i, layer in enumerate(self.layers, 1):
            x = layer(x, self.attention_dists)
            if i == len(self.layers) // 2:            # the CTC posteriors of the middle layer re-enter the stream
                x = x + self.out_mid(torch.softmax(self.out(x), -1))
        return x


class GraniteSpeechProjector(nn.Module):
    def __init__(self, hf, pc, text_hidden: int):
        super().__init__()
        self.window, self.rate = int(hf.window_size), int(hf.downsample_rate)
        self.query = nn.Parameter(torch.zeros(1, self.window // self.rate, int(pc.hidden_size)))
        self.qformer = Blip2QFormer(pc)
        self.linear = nn.Linear(int(pc.hidden_size), text_hidden)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """[T, d] -> [ceil(T / window) * window / rate, H_text]"""
        T, d = x.shape
        nb = -(-T // self.window)
        x = F.pad(x, (0, 0, 0, nb * self.window - T)).view(nb, self.window, d)
        q = self.qformer(self.query.expand(nb, -1, -1).to(x.dtype), x)
        return self.linear(q.reshape(nb * (self.window // self.rate), -1))


class GraniteSpeechForConditionalGeneration(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        ec, pc, tc = _cfg(hf.encoder_config), _cfg(hf.projector_config), _cfg(hf.text_config)
        self.audio_token_id = int(getattr(hf, "audio_token_index", None) or getattr(hf, "audio_token_id", 49155))
        self.encoder = GraniteSpeechCTCEncoder(ec)
        self.projector = GraniteSpeechProjector(hf, pc, int(tc.hidden_size))
        if not getattr(tc, "architectures", None):
            tc.architectures = ["GraniteForCausalLM"]
        if getattr(tc, "tie_word_embeddings", None) is None:
            tc.tie_word_embeddings = bool(getattr(hf, "tie_word_embeddings", True))
        self.language_model = build_language_model(vllm_config, tc, maybe_prefix(prefix, "language_model"))

    def encode_mm_items(self, items: list) -> list[torch.Tensor]:
        """items: [features [T, 160], number of valid frames] per clip."""
        outs = []
        w = self.encoder.input_linear.weight
        for feats, n in items:
            outs.append(self.projector(self.encoder(feats[: int(n)].to(w.dtype))))
        return outs

    def get_input_embeddings(self, input_ids, mm_embeds=None, mm_mask=None):
        ids = torch.where(input_ids == self.audio_token_id, 0, input_ids)
        x = self.language_model.get_input_embeddings(ids)
        if mm_embeds is not None and mm_embeds.numel():
            x[mm_mask] = mm_embeds.to(x.dtype)
        return x

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        return self.language_model(input_ids, positions, intermediate_tensors, inputs_embeds)

    def compute_logits(self, hidden_states):
        return self.language_model.compute_logits(hidden_states)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        dec, own = [], []
        buffers = dict(self.named_buffers())
        loaded = set()
        for n, t in weights:
            if n.startswith("model.language_model."):
                dec.append(("model." + n[len("model.language_model."):], t))
            elif n.startswith("language_model."):
                n = n[len("language_model."):]
                while n.startswith("model.model."):          # some exporters repeat the wrapper level
                    n = n[len("model."):]
                dec.append((n, t))
            elif n.startswith("lm_head."):
                dec.append((n, t))
            else:
                n = n[len("model."):] if n.startswith("model.") else n
                if n in buffers:                            # running statistics of the batch norms
                    buffers[n].copy_(t)
                    continue
                own.append((n, t))
        loaded |= {"language_model." + n for n in self.language_model.load_weights(dec)}
        loaded |= load_weights_with_mapping(self, own, [])
        return loaded


def mel_filterbank(n_freqs: int, n_mels: int, rate: int) -> torch.Tensor:
    """[n_freqs, n_mels] triangular filters on the HTK mel scale from 0 to rate / 2, unnormalised."""
    def to_mel(f):
        return 2595.0 * math.log10(1.0 + f / 700.0)
    freqs = torch.linspace(0, rate // 2, n_freqs)
    pts = torch.linspace(to_mel(0.0), to_mel(rate / 2.0), n_mels + 2)
    hz = 700.0 * (10.0 ** (pts / 2595.0) - 1.0)
    diff = hz[1:] - hz[:-1]
    slopes = hz[None, :] - freqs[:, None]
    down = -slopes[:, :-2] / diff[:-1]
    up = slopes[:, 2:] / diff[1:]
    return torch.clamp(torch.minimum(down, up), min=0.0)


def granite_speech_processor(vllm_config):
    """Audio placeholder runs sized to the projector's output.  ``mm_data["audio"]``: a waveform (16 kHz), a
    (waveform, rate) pair, or a dict with ready ``input_features`` [T, 160]."""
    from ..multimodal import AudioTokenProcessor
    hf = vllm_config.model_config.hf_config
    window, rate = int(hf.window_size), int(hf.downsample_rate)

    class _P(AudioTokenProcessor):
        def __init__(self, tok):
            self.audio_token_id, self.rate = tok, 16000
            self.n_fft, self.win, self.hop, self.n_mels = 512, 400, 160, 80
            self.fb = mel_filterbank(self.n_fft // 2 + 1, self.n_mels, self.rate)

        def _waveform(self, a):
            if isinstance(a, dict):
                return torch.as_tensor(a["input_features"]).float().numpy()
            return super()._waveform(a)

        def features(self, a):
            if isinstance(a, dict):
                f = torch.as_tensor(a["input_features"]).float()
                return f, int(f.shape[0])
            w = torch.as_tensor(self._waveform(a))
            spec = torch.stft(w, self.n_fft, self.hop, self.win, torch.hann_window(self.win), center=True,
                              pad_mode="reflect", return_complex=True).abs().pow(2)
            logmel = (spec.t() @ self.fb).clamp_min(1e-10).log10()
            logmel = torch.maximum(logmel, logmel.max() - 8.0) / 4.0 + 1.0
            if logmel.shape[0] % 2:
                logmel = logmel[:-1]
            f = logmel.reshape(-1, 2 * self.n_mels)
            return f, int(f.shape[0])

        def num_tokens(self, frames: int) -> int:
            return -(-frames // window) * (window // rate)

    return _P(int(getattr(hf, "audio_token_index", None) or getattr(hf, "audio_token_id", 49155)))
