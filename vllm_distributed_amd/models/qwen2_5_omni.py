"""Qwen2.5-Omni "thinker": the text-generating half of the checkpoint.  A Whisper-style audio encoder that works on
windows of ``2 * n_window`` mel frames (convolutions, positions and attention all restart at every window), the
Qwen2.5-VL vision tower (stored with separate q / k / v projections) and a Qwen2 decoder with multimodal rotary
positions.  Audio positions advance like text positions on all three axes, so only images and videos need grids.

The talker and the codec of the full checkpoint are not loaded; audio that rides inside a video
(``use_audio_in_video``) is not interleaved: pass the clip as an audio item of its own.
Reference: vllm/model_executor/models/qwen2_5_omni_thinker.py (Qwen2_5OmniThinkerForConditionalGeneration :710-930).
"""
from __future__ import annotations

from typing import Iterable

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from .llama import LlamaForCausalLM
from .llava import ClipAttention, _cfg
from .qwen2_vl import Qwen25VisionTransformer, Qwen2VLForConditionalGeneration, mrope_positions
from .utils import load_weights_with_mapping, maybe_prefix


class _AudioLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        d = int(c.d_model)
        self.self_attn = ClipAttention(d, int(c.encoder_attention_heads))
        self.self_attn.k_proj = nn.Linear(d, d, bias=False)
        self.self_attn_layer_norm, self.final_layer_norm = nn.LayerNorm(d), nn.LayerNorm(d)
        self.fc1, self.fc2 = nn.Linear(d, int(c.encoder_ffn_dim)), nn.Linear(int(c.encoder_ffn_dim), d)

    def forward(self, x):
        x = x + self.self_attn(self.self_attn_layer_norm(x))
        return x + self.fc2(F.gelu(self.fc1(self.final_layer_norm(x))))


class OmniAudioEncoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        d = int(c.d_model)
        self.n_window = int(c.n_window)
        self.conv1 = nn.Conv1d(int(c.num_mel_bins), d, 3, padding=1)
        self.conv2 = nn.Conv1d(d, d, 3, stride=2, padding=1)
        n = int(c.max_source_positions)
        inc = np.log(10000.0) / (d // 2 - 1)
        t = torch.arange(n)[:, None] * torch.exp(-inc * torch.arange(d // 2))[None, :]
        self.register_buffer("positions", torch.cat([t.sin(), t.cos()], 1), persistent=False)
        self.audio_bos_eos_token = nn.Embedding(2, int(c.output_dim))       # in the checkpoint, unused by the thinker
        self.layers = nn.ModuleList(_AudioLayer(c) for _ in range(int(c.encoder_layers)))
        self.ln_post = nn.LayerNorm(d)
        self.proj = nn.Linear(d, int(c.output_dim))

    def forward(self, mel: torch.Tensor) -> torch.Tensor:
        """[n_mels, T] of one clip -> [((T - 1) // 2 + 1 - 2) // 2 + 1, output_dim]"""
        outs = []
        for chunk in mel.to(self.conv1.weight.dtype).split(2 * self.n_window, dim=1):
            x = F.gelu(self.conv2(F.gelu(self.conv1(chunk[None]))))[0].t()
            x = (x + self.positions[: x.shape[0]].to(x.dtype))[None]
            for layer in self.layers:
                x = layer(x)
            outs.append(x[0])
        x = torch.cat(outs)
        n = (x.shape[0] - 2) // 2 + 1
        x = (x[0:2 * n:2] + x[1:2 * n:2]) / 2
        return self.proj(self.ln_post(x))


class Qwen2_5OmniThinkerForConditionalGeneration(Qwen2VLForConditionalGeneration):
    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        nn.Module.__init__(self)
        import copy
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vc, ac, tc = _cfg(hf.vision_config), _cfg(hf.audio_config), _cfg(hf.text_config)
        vcd = vc.to_dict() if hasattr(vc, "to_dict") else dict(vc.__dict__)
        self.merge = int(vcd.get("spatial_merge_size", 2))
        self.image_token_id = int(getattr(hf, "image_token_index", None) or getattr(hf, "image_token_id", 151655))
        self.visual = Qwen25VisionTransformer(vcd, int(vcd.get("out_hidden_size", None) or tc.hidden_size))
        self.audio_tower = OmniAudioEncoder(ac)
        if not getattr(tc, "architectures", None):
            tc.architectures = ["Qwen2ForCausalLM"]
        top_tie = getattr(hf, "top_tie_word_embeddings", None)
        if top_tie is not None:                       # the thinker has an output head of its own
            tc.tie_word_embeddings = bool(top_tie)
        sub = copy.copy(vllm_config)
        sub.model_config = copy.copy(vllm_config.model_config)
        sub.model_config.hf_config = tc
        self.language_model = LlamaForCausalLM(sub, prefix=maybe_prefix(prefix, "language_model"))
        lm = self.language_model.model
        self.rotary = lm.layers[lm.start_layer].self_attn.rotary_emb

    @staticmethod
    def _is_grid(item) -> bool:
        return item[1].numel() == 3

    def encode_mm_items(self, items: list) -> list[torch.Tensor]:
        vis = [i for i, it in enumerate(items) if self._is_grid(it)]
        outs: list = [None] * len(items)
        if vis:
            for i, e in zip(vis, super().encode_mm_items([items[i] for i in vis])):
                outs[i] = e
        for i, (feats, n) in enumerate(items):
            if outs[i] is None:
                outs[i] = self.audio_tower(feats[:, : int(n)])
        return outs

    def get_mrope_positions(self, prompt_len: int, mm_positions: list, mm_inputs: list):
        keep = [k for k, it in enumerate(mm_inputs) if torch.as_tensor(it[1]).numel() == 3]
        spans = [(mm_positions[k].offset, mm_positions[k].length) for k in keep]
        grids = [tuple(int(x) for x in torch.as_tensor(mm_inputs[k][1]).tolist()) for k in keep]
        return mrope_positions(prompt_len, spans, grids, self.merge)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        dec, own, qkv = [], [], {}
        for n, t in weights:
            if n.startswith(("talker.", "token2wav.")):
                continue
            n = n[len("thinker."):] if n.startswith("thinker.") else n
            if n.startswith(("model.", "lm_head.")):
                dec.append(("language_model." + n, t))
            elif n.startswith("visual.") and ".attn." in n and n.split(".attn.")[1][0] in "qkv" \
                    and n.split(".attn.")[1][1] == ".":
                base, rest = n.split(".attn.")
                qkv.setdefault((base, rest[2:]), {})[rest[0]] = t            # (block, "weight" | "bias") -> q, k, v
            else:
                own.append((n, t))
        for (base, kind), parts in qkv.items():
            own.append((f"{base}.attn.qkv.{kind}", torch.cat([parts["q"], parts["k"], parts["v"]])))
        loaded = load_weights_with_mapping(self, dec, LlamaForCausalLM.stacked_params_mapping,
                                           tie_lm_head=self.language_model.tie)
        loaded |= load_weights_with_mapping(self, own, [])
        return loaded


Qwen2_5OmniForConditionalGeneration = Qwen2_5OmniThinkerForConditionalGeneration
Qwen2_5OmniModel = Qwen2_5OmniThinkerForConditionalGeneration


def qwen2_5_omni_processor(vllm_config):
    """Image / video runs through the patch-grid processor, audio runs sized to the audio encoder's output."""
    from ..multimodal import AudioTokenProcessor, PatchGridProcessor, PlaceholderRange, hash_item
    hf = vllm_config.model_config.hf_config
    vc, ac = _cfg(hf.vision_config), _cfg(hf.audio_config)
    img = int(getattr(hf, "image_token_index", None) or getattr(hf, "image_token_id", 151655))
    vid = int(getattr(hf, "video_token_index", None) or getattr(hf, "video_token_id", 151656))
    aud = int(getattr(hf, "audio_token_index", None) or getattr(hf, "audio_token_id", 151646))
    grid = PatchGridProcessor(img, int(getattr(vc, "patch_size", 14)), int(getattr(vc, "spatial_merge_size", 2)),
                              int(getattr(vc, "temporal_patch_size", 2)), 56 * 56, 14 * 14 * 4 * 1280,
                              video_token_id=vid)
    audio = AudioTokenProcessor(aud, int(ac.num_mel_bins))

    def as_list(x):
        return [] if x is None else (list(x) if isinstance(x, (list, tuple)) and not
                                     (len(x) == 2 and not hasattr(x[1], "__len__")) else [x])

    def process(ids, mm_data):
        queues = {img: iter(as_list(mm_data.get("image"))), vid: iter(as_list(mm_data.get("video"))),
                  aud: iter(as_list(mm_data.get("audio")))}
        out, ranges, inputs, hashes = [], [], [], []
        i = 0
        while i < len(ids):
            tok = ids[i]
            if tok not in queues:
                out.append(tok)
                i += 1
                continue
            j = i
            while j < len(ids) and ids[j] == tok:
                j += 1
            try:
                item = next(queues[tok])
            except StopIteration:
                raise ValueError("more placeholders than multimodal items") from None
            if tok == aud:
                if isinstance(item, dict):
                    feats = torch.as_tensor(item["input_features"]).float()
                    n = int(item.get("feature_length", feats.shape[1]))
                    key = feats
                else:
                    feats, n = audio.features(item)
                    key = audio._waveform(item)
                length = audio.num_tokens(n)
                inputs.append([feats, torch.tensor(n, dtype=torch.int64)])
                hashes.append(hash_item(key))
            else:
                px, g = grid.patches(item) if tok == img else grid.video_patches(item)
                length = int(g.prod()) // grid.merge ** 2
                inputs.append([px, g])
                hashes.append(hash_item(px if isinstance(item, dict) else item))
            ranges.append(PlaceholderRange(offset=len(out), length=length))
            out.extend([tok] * length)
            i = j
        for q in queues.values():
            if next(q, None) is not None:
                raise ValueError("more multimodal items than placeholders")
        return out, inputs, hashes, ranges

    return process
