"""MiniCPM-V 2.5 / 2.6 / 4.0 (and the audio extension MiniCPM-o 2.6): a SigLIP tower in the NaViT manner (every
image slice keeps its own patch grid; position ids spread the slice over the tower's square table), a resampler whose
``query_num`` learned queries cross-attend to the slice's patches with 2-D sine / cosine positions on the keys, and a
Llama-3 (2.5, 4.0) or Qwen2 (2.6) decoder.  A large image is an overview plus a grid of slices; each of them takes
``query_num`` prompt positions between its own pair of marker tokens.

MiniCPM-o adds a Whisper encoder whose output is projected by a two-layer MLP and average-pooled; the speech
generation half of that checkpoint is not loaded.  Version 2.0 (timm tower) is not built.
No transformers class exists for these families: parity unpinned; the resampler, the slicing rule and the tower's
position ids are tested against the arithmetic written out.
Reference: vllm/model_executor/models/minicpmv.py (Resampler2_5 :113-203, MiniCPMV2_5 :1087-1174, MiniCPMV2_6
:1177-1258), vllm/model_executor/models/minicpmo.py (MiniCPMO :466-700).
"""
from __future__ import annotations

import math
from typing import Iterable

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from .aria import AriaVisionTower
from .llava import _cfg
from .llava_next import build_language_model
from .qwen_vl import QueryResampler
from .utils import load_weights_with_mapping, maybe_prefix


def config_version(config) -> tuple[int, ...]:
    v = getattr(config, "version", None)
    if v is None:
        return (2, 0) if (config.hidden_size == 2304 and config.query_num == 64) else (2, 5)
    return tuple(int(x) for x in str(v).split("."))


class MiniCPMVTower(AriaVisionTower):
    """SigLIP layers over one slice at a time, closing LayerNorm included."""

    def __init__(self, c, drop_last: bool):
        super().__init__(c, -2 if drop_last else -1)
        self.post_layernorm = nn.LayerNorm(int(c.hidden_size), eps=float(getattr(c, "layer_norm_eps", 1e-6)))

    def forward(self, pixels: torch.Tensor) -> torch.Tensor:
        return self.post_layernorm(super().forward(pixels))


class MiniCPMV(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        self.version = config_version(hf)
        if self.version < (2, 5):
            raise NotImplementedError("MiniCPM-V 2.0 (timm vision tower) is not built; 2.5, 2.6 and 4.0 are")
        vc = _cfg(hf.vision_config)
        self.patch = int(vc.patch_size)
        self.vpm = MiniCPMVTower(vc, bool(getattr(hf, "drop_vision_last_layer", False)))
        d = int(hf.hidden_size)
        self.resampler = QueryResampler(0, d, d // 128, int(vc.hidden_size), post_projection=True, adaptive=True,
                                        num_queries=int(hf.query_num))
        arch = "Qwen2ForCausalLM" if self.version == (2, 6) else "LlamaForCausalLM"
        tc = _cfg({**hf.to_dict(), "architectures": [arch]})
        self.llm = build_language_model(vllm_config, tc, maybe_prefix(prefix, "llm"))

    def encode_mm_items(self, items: list) -> list[torch.Tensor]:
        """items: [pixels [3, h * p, w * p]] per slice (the overview and every slice of an image are items of
        their own) -> [query_num, H] each."""
        outs = []
        w = self.vpm.embeddings.patch_embedding.weight
        for (px,) in items:
            px = px.to(w.dtype)
            outs.append(self.resampler(self.vpm(px)[None], (px.shape[-2] // self.patch, px.shape[-1] // self.patch))[0])
        return outs

    def get_input_embeddings(self, input_ids, mm_embeds=None, mm_mask=None):
        x = self.llm.get_input_embeddings(input_ids)
        if mm_embeds is not None and mm_embeds.numel():
            x[mm_mask] = mm_embeds.to(x.dtype)
        return x

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        return self.llm(input_ids, positions, intermediate_tensors, inputs_embeds)

    def compute_logits(self, hidden_states):
        return self.llm.compute_logits(hidden_states)

    def _skip(self, name: str) -> bool:
        return name.startswith(("tts.", "audio_avg_pooler.")) or ".head." in name

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        dec, own = [], []
        for n, t in weights:
            if n.startswith("llm."):
                dec.append((n[len("llm."):], t))
            elif not self._skip(n):
                own.append((n, t))
        loaded = {"llm." + n for n in self.llm.load_weights(dec)}
        loaded |= load_weights_with_mapping(self, own, [])
        return loaded


# ---- image geometry ---------------------------------------------------------------------------------------
def _divisible(x: float, d: int) -> int:
    return max(round(x / d) * d, d)


def best_resize(size: tuple[int, int], scale: int, patch: int, upscale: bool = False) -> tuple[int, int]:
    """(width, height) with about ``scale``^2 pixels at the image's aspect ratio, both multiples of ``patch``."""
    w, h = size
    if w * h > scale * scale or upscale:
        r = w / h
        h = int(scale / math.sqrt(r))
        w = int(h * r)
    return _divisible(w, patch), _divisible(h, patch)


def slice_grid(size: tuple[int, int], max_slices: int, scale: int) -> tuple[int, int] | None:
    """(columns, rows) of the slice grid, or None when the image is small enough for the overview alone."""
    w, h = size
    multiple = min(math.ceil(w * h / (scale * scale)), max_slices)
    if multiple <= 1:
        return None
    best, err = None, float("inf")
    for n in (multiple - 1, multiple, multiple + 1):
        if n == 1 or n > max_slices:
            continue
        for cols in range(1, n + 1):
            if n % cols == 0:
                e = abs(math.log(w / h) - math.log(cols / (n // cols)))
                if e < err:
                    best, err = (cols, n // cols), e
    return best


def refine_size(size: tuple[int, int], grid: tuple[int, int], scale: int, patch: int) -> tuple[int, int]:
    w, h = size
    gx, gy = grid
    cell = best_resize((_divisible(w, gx) / gx, _divisible(h, gy) / gy), scale, patch, upscale=True)
    return cell[0] * gx, cell[1] * gy


def minicpmv_processor(vllm_config):
    """``<image>`` + placeholder(s) + ``</image>`` in the prompt -> the overview's ``query_num`` positions between the
    image markers, then (for large images) every slice's positions between slice markers, rows separated by a
    newline token.  The marker ids are tokenizer properties of these checkpoints: they come from the tokenizer when
    one is loaded, else from ``mm_processor_kwargs`` (``im_start_id``, ``im_end_id``, ``slice_start_id``,
    ``slice_end_id``, ``unk_id``, ``newline_id``).  Image-id tags of 2.6 are not written."""
    from ..multimodal import PlaceholderRange, hash_item
    hf = vllm_config.model_config.hf_config
    vc = _cfg(hf.vision_config)
    patch, q = int(vc.patch_size), int(hf.query_num)
    sc = getattr(hf, "slice_config", None)
    sc = dict(sc) if isinstance(sc, dict) else (sc.to_dict() if sc is not None else {})
    max_slices = int(sc.get("max_slice_nums", 9))
    scale = int(sc.get("scale_resolution", getattr(hf, "image_size", 448)))
    kw = dict(getattr(vllm_config.model_config, "mm_processor_kwargs", None) or {})
    max_slices = int(kw.get("max_slice_nums", max_slices))
    ids_ = {}

    def marker(name: str, token: str) -> int:
        if name not in ids_:
            if name in kw:
                ids_[name] = int(kw[name])
            elif getattr(hf, name, None) is not None:
                ids_[name] = int(getattr(hf, name))
            elif not vllm_config.model_config.skip_tokenizer_init:
                from ..engine.processor import get_tokenizer
                mc = vllm_config.model_config
                tok = get_tokenizer(mc.tokenizer or mc.model, mc.tokenizer_mode)
                for n_, t_ in (("im_start_id", "<image>"), ("im_end_id", "</image>"), ("slice_start_id", "<slice>"),
                               ("slice_end_id", "</slice>"), ("unk_id", "<unk>")):
                    ids_.setdefault(n_, int(tok.convert_tokens_to_ids(t_)))
                ids_.setdefault("newline_id", int(tok.encode("\n", add_special_tokens=False)[-1]))
            if name not in ids_:
                raise ValueError(f"{name} ({token!r}) is a tokenizer property: load a tokenizer or pass it in "
                                 f"mm_processor_kwargs")
        return ids_[name]

    def norm(t: torch.Tensor) -> torch.Tensor:
        return (t.clamp(0, 255) / 255.0 - 0.5) / 0.5

    def slices_of(img) -> tuple[list[torch.Tensor], tuple[int, int] | None]:
        if isinstance(img, dict):                        # {"slices": [[3, H, W], ...], "grid": (cols, rows) | None}
            return [torch.as_tensor(s).float() for s in img["slices"]], img.get("grid")
        import numpy as np
        a = np.asarray(img.convert("RGB")) if hasattr(img, "convert") else np.asarray(img)
        t = torch.as_tensor(a).permute(2, 0, 1).float()[None]
        H, W = t.shape[-2:]

        def resized(w, h):
            return norm(F.interpolate(t, size=(h, w), mode="bicubic", align_corners=False, antialias=True))[0]
        grid = slice_grid((W, H), max_slices, scale)
        if grid is None:
            return [resized(*best_resize((W, H), scale, patch, upscale=True))], None
        out = [resized(*best_resize((W, H), scale, patch))]
        rw, rh = refine_size((W, H), grid, scale, patch)
        full = resized(rw, rh)
        cw, ch = rw // grid[0], rh // grid[1]
        for r in range(grid[1]):
            for c in range(grid[0]):
                out.append(full[:, r * ch:(r + 1) * ch, c * cw:(c + 1) * cw])
        return out, grid

    def process(ids, mm_data):
        images = mm_data.get("image")
        if images is None:
            raise ValueError("multi_modal_data needs an 'image' entry")
        images = images if isinstance(images, (list, tuple)) else [images]
        im_s, im_e, unk = marker("im_start_id", "<image>"), marker("im_end_id", "</image>"), marker("unk_id", "<unk>")
        out, ranges, inputs, hashes = [], [], [], []
        i, k = 0, 0
        while i < len(ids):
            if ids[i] != im_s:
                out.append(ids[i])
                i += 1
                continue
            j = i + 1
            while j < len(ids) and ids[j] == unk:
                j += 1
            if j >= len(ids) or ids[j] != im_e:
                out.append(ids[i])
                i += 1
                continue
            if k >= len(images):
                raise ValueError("more image placeholders than images")
            sl, grid = slices_of(images[k])
            key = hash_item(torch.cat([x.flatten() for x in sl]) if isinstance(images[k], dict) else images[k])

            def put(px, n_, start_tok, end_tok):
                out.append(start_tok)
                ranges.append(PlaceholderRange(offset=len(out), length=q))
                out.extend([unk] * q)
                out.append(end_tok)
                inputs.append([px])                       # every slice is an encoder item of its own
                hashes.append(f"{key}:{n_}")
            put(sl[0], 0, im_s, im_e)
            if grid is not None:
                s_s, s_e = marker("slice_start_id", "<slice>"), marker("slice_end_id", "</slice>")
                for r in range(grid[1]):
                    if r:
                        out.append(marker("newline_id", "\n"))
                    for c in range(grid[0]):
                        put(sl[1 + r * grid[0] + c], 1 + r * grid[0] + c, s_s, s_e)
            k, i = k + 1, j + 1
        if k != len(images):
            raise ValueError(f"prompt has {k} image placeholders but {len(images)} images were given")
        return out, inputs, hashes, ranges

    return process


# ---- MiniCPM-o: the same model plus an audio path ---------------------------------------------------------
class MiniCPMO(MiniCPMV):
    """MiniCPM-o 2.6.  Audio: Whisper encoder whose frames attend to their own and all earlier chunks of
    ``audio_chunk_length`` seconds (50 frames per second), a two-layer ReLU projector, average pooling over
    ``audio_pool_step`` frames."""

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        from .qwen2_audio import WhisperEncoder
        hf = self.config
        ac = _cfg(hf.audio_config)
        self.apm = WhisperEncoder(ac, avg_pool=False)
        d_audio, d = int(ac.encoder_ffn_dim) // 4, int(hf.hidden_size)
        self.audio_projection_layer = nn.Module()
        self.audio_projection_layer.linear1 = nn.Linear(d_audio, d)
        self.audio_projection_layer.linear2 = nn.Linear(d, d)
        self.pool = int(getattr(hf, "audio_pool_step", 2))
        self.chunk_frames = int(float(getattr(hf, "audio_chunk_length", 1.0)) * 50)

    def encode_audio(self, mel: torch.Tensor) -> torch.Tensor:
        """[n_mels, T] of one clip (at most 30 s) -> [((T - 1) // 2 + 1 - pool) // pool + 1, H]"""
        a = self.apm
        x = F.gelu(a.conv1(mel[None].to(a.conv1.weight.dtype)))
        x = F.gelu(a.conv2(x)).permute(0, 2, 1)
        S = x.shape[1]
        x = x + a.embed_positions.weight[:S]
        mask = None
        if self.chunk_frames > 0:
            idx = torch.arange(S, device=x.device)
            visible = idx[None, :] < ((idx // self.chunk_frames + 1) * self.chunk_frames)[:, None]
            mask = torch.zeros(1, 1, S, S, dtype=x.dtype, device=x.device).masked_fill_(~visible, float("-inf"))
        for layer in a.layers:
            x = layer(x, mask)
        x = a.layer_norm(x)
        p = self.audio_projection_layer
        x = p.linear2(F.relu(p.linear1(x)))
        return F.avg_pool1d(x.transpose(1, 2), self.pool, stride=self.pool).transpose(1, 2)[0]

    def encode_mm_items(self, items: list) -> list[torch.Tensor]:
        outs: list = [None] * len(items)
        vis = [i for i, it in enumerate(items) if len(it) == 1]
        for i, e in zip(vis, super().encode_mm_items([items[i] for i in vis])):
            outs[i] = e
        for i, it in enumerate(items):
            if outs[i] is None:
                mel, n = it
                outs[i] = self.encode_audio(mel[:, : int(n)])
        return outs

    def _skip(self, name: str) -> bool:
        return super()._skip(name) or name.startswith(("apm.proj_out.", "apm.decoder."))



def minicpmo_processor(vllm_config):
    """Images as for MiniCPM-V; audio: ``<|audio_start|>`` + placeholder(s) + ``<|audio_end|>`` -> one run per clip
    sized to the pooled encoder output (clips longer than 30 s must be split by the caller).  Marker ids
    ``audio_start_id`` / ``audio_end_id`` come from the config, ``mm_processor_kwargs`` or the tokenizer."""
    from ..multimodal import AudioTokenProcessor, PlaceholderRange, hash_item
    hf = vllm_config.model_config.hf_config
    image = minicpmv_processor(vllm_config)
    ac = _cfg(hf.audio_config)
    pool = int(getattr(hf, "audio_pool_step", 2))
    kw = dict(getattr(vllm_config.model_config, "mm_processor_kwargs", None) or {})
    audio = AudioTokenProcessor(-1, int(ac.num_mel_bins))

    def marker(name, token):
        if name in kw:
            return int(kw[name])
        if getattr(hf, name, None) is not None:
            return int(getattr(hf, name))
        from ..engine.processor import get_tokenizer
        mc = vllm_config.model_config
        return int(get_tokenizer(mc.tokenizer or mc.model, mc.tokenizer_mode).convert_tokens_to_ids(token))

    def process(ids, mm_data):
        out_ranges, out_inputs, out_hashes = [], [], []
        if mm_data.get("image") is not None:
            ids, inputs, hashes, ranges = image(ids, {"image": mm_data["image"]})
            out_ranges, out_inputs, out_hashes = list(ranges), list(inputs), list(hashes)
        clips = mm_data.get("audio")
        if clips is None:
            if not out_ranges:
                raise ValueError("multi_modal_data needs an 'image' or an 'audio' entry")
            return ids, out_inputs, out_hashes, out_ranges
        clips = clips if isinstance(clips, list) else [clips]
        a_s, a_e, unk = marker("audio_start_id", "<|audio_start|>"), marker("audio_end_id", "<|audio_end|>"), \
            marker("unk_id", "<unk>")
        new, shift_at, i, k = [], [], 0, 0                      # shift_at: (position in ids, tokens added)
        a_ranges, a_inputs, a_hashes = [], [], []
        while i < len(ids):
            j = i + 1
            while ids[i] == a_s and j < len(ids) and ids[j] == unk:
                j += 1
            if ids[i] != a_s or j >= len(ids) or ids[j] != a_e:
                new.append(ids[i])
                i += 1
                continue
            if k >= len(clips):
                raise ValueError("more audio placeholders than clips")
            clip = clips[k]
            if isinstance(clip, dict):
                mel = torch.as_tensor(clip["audio_features"]).float()
                n = int(clip.get("audio_feature_lens", mel.shape[-1]))
                key = mel
            else:
                mel, n = audio.features(clip)
                key = audio._waveform(clip)
            ntok = (((n - 1) // 2 + 1) - pool) // pool + 1
            new.append(a_s)
            a_ranges.append(PlaceholderRange(offset=len(new), length=ntok))
            new.extend([unk] * ntok)
            new.append(a_e)
            shift_at.append((i, ntok - (j - i - 1)))
            a_inputs.append([mel, torch.tensor(n, dtype=torch.int64)])
            a_hashes.append(hash_item(key))
            k, i = k + 1, j + 1
        if k != len(clips):
            raise ValueError(f"prompt has {k} audio placeholders but {len(clips)} clips were given")
        # image ranges were computed before the audio runs grew: move those that lie behind one
        moved = []
        for r in out_ranges:
            off = r.offset + sum(d for pos, d in shift_at if pos < r.offset)
            moved.append(PlaceholderRange(offset=off, length=r.length))
        merged = sorted(zip(moved + a_ranges, out_inputs + a_inputs, out_hashes + a_hashes), key=lambda t: t[0].offset)
        return new, [m[1] for m in merged], [m[2] for m in merged], [m[0] for m in merged]

    return process
