"""Multimodal inputs: registry of per-architecture processors, content hashing, placeholder expansion.

A processor turns ``{"prompt_token_ids": [...], "multi_modal_data": {"image": ...}}`` into
  * the prompt with every image token expanded to as many placeholder tokens as the encoder emits,
  * one encoder input per item (pre-processed pixel tensor),
  * one content hash per item (prefix caching keys and encoder-cache identity),
  * one ``PlaceholderRange`` per item (where its embeddings land in the prompt).
Images may be PIL images, ``[H, W, 3]`` uint8 arrays / tensors, or already pre-processed
``[3, S, S]`` float tensors.
Reference: vllm/multimodal/{registry.py, processing.py, hasher.py, inputs.py}, vllm/v1/engine/mm_input_cache.py.
"""
from __future__ import annotations

import hashlib
from typing import Any, Callable, Optional

import numpy as np
import torch

from ..request import PlaceholderRange

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def hash_item(x: Any) -> str:
    h = hashlib.sha256()
    if isinstance(x, torch.Tensor):
        h.update(str((tuple(x.shape), str(x.dtype))).encode())
        h.update(x.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())
    elif isinstance(x, np.ndarray):
        h.update(str((x.shape, x.dtype.str)).encode())
        h.update(np.ascontiguousarray(x).tobytes())
    elif hasattr(x, "tobytes") and hasattr(x, "size"):        # PIL image
        h.update(str((x.size, getattr(x, "mode", ""))).encode())
        h.update(x.tobytes())
    else:
        h.update(repr(x).encode())
    return h.hexdigest()


def to_pixel_values(img: Any, size: int, mean=CLIP_MEAN, std=CLIP_STD) -> torch.Tensor:
    """CLIP-style preprocessing: shortest side to ``size`` (bicubic), centre crop, scale to [0,1], normalise."""
    if isinstance(img, torch.Tensor) and img.dtype.is_floating_point and img.dim() == 3 and img.shape[0] == 3:
        return img.float()                                   # already pre-processed
    if hasattr(img, "convert"):                              # PIL
        img = np.asarray(img.convert("RGB"))
    t = torch.as_tensor(np.asarray(img) if not isinstance(img, torch.Tensor) else img)
    if t.dim() != 3 or t.shape[-1] != 3:
        raise ValueError(f"image must be [H, W, 3] or a pre-processed [3, S, S] float tensor, got {tuple(t.shape)}")
    t = t.permute(2, 0, 1).float().unsqueeze(0)
    H, W = t.shape[-2:]
    s = size / min(H, W)
    nh, nw = max(int(round(H * s)), size), max(int(round(W * s)), size)
    if (nh, nw) != (H, W):
        t = torch.nn.functional.interpolate(t, size=(nh, nw), mode="bicubic", align_corners=False, antialias=True)
    top, left = (nh - size) // 2, (nw - size) // 2
    t = t[0, :, top:top + size, left:left + size].clamp(0, 255) / 255.0
    m = torch.tensor(mean).view(3, 1, 1)
    sd = torch.tensor(std).view(3, 1, 1)
    return (t - m) / sd


class MultiModalRegistry:
    """architecture name -> processor factory; processors are built once per model config."""

    def __init__(self):
        self._factories: dict[str, Callable] = {}
        self._cache: dict[int, Any] = {}

    def register_processor(self, arch: str, factory: Callable) -> None:
        self._factories[arch] = factory

    def supports(self, arch: str) -> bool:
        return arch in self._factories

    def get_processor(self, vllm_config):
        # keyed by id() but checked against a weak reference: a bare id() table would hand a dead config's
        # processor to a new config that happens to be allocated at the same address
        import weakref
        mc = vllm_config.model_config
        hit = self._cache.get(id(mc))
        if hit is not None and hit[0]() is mc:
            return hit[1]
        arch = (mc.architectures or [""])[0]
        if arch not in self._factories:
            raise ValueError(f"{arch} takes no multimodal inputs")
        proc = self._factories[arch](vllm_config)
        self._cache = {k: v for k, v in self._cache.items() if v[0]() is not None}
        self._cache[id(mc)] = (weakref.ref(mc), proc)
        return proc


MULTIMODAL_REGISTRY = MultiModalRegistry()


class ImageTokenProcessor:
    """Models whose prompt marks each image with ONE special token that the encoder output replaces
    (LLaVA family)."""

    def __init__(self, image_token_id: int, tokens_per_image: int, image_size: int):
        self.image_token_id, self.tokens_per_image, self.image_size = image_token_id, tokens_per_image, image_size

    def __call__(self, prompt_token_ids: list[int], mm_data: dict):
        images = mm_data.get("image")
        if images is None:
            raise ValueError("multi_modal_data needs an 'image' entry")
        if not isinstance(images, (list, tuple)):
            images = [images]
        spots = [i for i, t in enumerate(prompt_token_ids) if t == self.image_token_id]
        # prompts may already carry the expanded run of image tokens: collapse runs first
        runs, prev = [], -2
        for i in spots:
            if i != prev + 1:
                runs.append([i, 1])
            else:
                runs[-1][1] += 1
            prev = i
        if len(runs) != len(images):
            raise ValueError(f"prompt has {len(runs)} image placeholders but {len(images)} images were given")
        out, positions, inputs, hashes = [], [], [], []
        cur = 0
        for (start, n), img in zip(runs, images):
            out.extend(prompt_token_ids[cur:start])
            positions.append(PlaceholderRange(offset=len(out), length=self.tokens_per_image))
            out.extend([self.image_token_id] * self.tokens_per_image)
            cur = start + n
            inputs.append(to_pixel_values(img, self.image_size))
            hashes.append(hash_item(img))
        out.extend(prompt_token_ids[cur:])
        return out, inputs, hashes, positions


def process_multimodal(vllm_config, prompt_token_ids: list[int], mm_data: dict):
    """(expanded prompt ids, encoder inputs, content hashes, placeholder ranges)."""
    return MULTIMODAL_REGISTRY.get_processor(vllm_config)(list(prompt_token_ids), mm_data)


def _llava_processor(vllm_config):
    hf = vllm_config.model_config.hf_config
    vc = hf.vision_config if isinstance(hf.vision_config, dict) else hf.vision_config.to_dict()
    if vc.get("model_type") == "pixtral":
        from ..models.pixtral import pixtral_processor
        return pixtral_processor(vllm_config)
    size, patch = int(vc.get("image_size", 336)), int(vc.get("patch_size", 14))
    n = (size // patch) ** 2
    if getattr(hf, "vision_feature_select_strategy", "default") == "full":
        n += 1
    return ImageTokenProcessor(int(getattr(hf, "image_token_index", None) or getattr(hf, "image_token_id", 32000)),
                               n, size)


MULTIMODAL_REGISTRY.register_processor("LlavaForConditionalGeneration", _llava_processor)


def _anyres(vllm_config):
    from ..models.llava_next import _anyres_processor
    return _anyres_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("LlavaNextForConditionalGeneration", _anyres)
MULTIMODAL_REGISTRY.register_processor("LlavaOnevisionForConditionalGeneration", _anyres)
MULTIMODAL_REGISTRY.register_processor("MiniMaxVL01ForConditionalGeneration", _anyres)


# ---- Qwen2-VL: variable-size patch grids ------------------------------------------------------------
def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56,
                 max_pixels: int = 14 * 14 * 4 * 1280) -> tuple[int, int]:
    """Closest size to (height, width) whose sides are multiples of ``factor`` and whose area lies in
    [min_pixels, max_pixels], keeping the aspect ratio as far as the rounding allows."""
    import math
    if max(height, width) / max(min(height, width), 1) > 200:
        raise ValueError("aspect ratio above 200 is not supported")
    h = max(factor, round(height / factor) * factor)
    w = max(factor, round(width / factor) * factor)
    if h * w > max_pixels:
        beta = math.sqrt(height * width / max_pixels)
        h = max(factor, math.floor(height / beta / factor) * factor)
        w = max(factor, math.floor(width / beta / factor) * factor)
    elif h * w < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h = math.ceil(height * beta / factor) * factor
        w = math.ceil(width * beta / factor) * factor
    return h, w


class PatchGridProcessor:
    """Qwen2-VL style inputs: an image becomes a [grid_t * grid_h * grid_w, C * tps * patch * patch] matrix of
    flattened patches (ordered so that each 2x2 merge block is contiguous) plus its grid; the prompt's image
    token is expanded to grid_t * grid_h * grid_w / merge^2 placeholders.  Videos ([F, H, W, 3] frames under
    ``multi_modal_data["video"]``, marked by the video token) take ``temporal_patch_size`` frames per temporal
    step.  Already pre-processed inputs are accepted as ``{"pixel_values": [N, K], "image_grid_thw": [3]}`` /
    ``{"pixel_values_videos": [N, K], "video_grid_thw": [3]}``."""

    def __init__(self, image_token_id: int, patch: int, merge: int, tps: int, min_pixels: int, max_pixels: int,
                 video_token_id: int = -1):
        self.image_token_id, self.patch, self.merge, self.tps = image_token_id, patch, merge, tps
        self.video_token_id = video_token_id
        self.min_pixels, self.max_pixels = min_pixels, max_pixels
        self.video_fps: Optional[float] = None     # set: videos carry seconds per temporal step (tps / fps) as a third entry

    def patches(self, img: Any) -> tuple[torch.Tensor, torch.Tensor]:
        if isinstance(img, dict):
            return torch.as_tensor(img["pixel_values"]).float(), torch.as_tensor(img["image_grid_thw"]).long().view(3)
        if hasattr(img, "convert"):
            img = np.asarray(img.convert("RGB"))
        t = torch.as_tensor(np.asarray(img) if not isinstance(img, torch.Tensor) else img)
        if t.dim() != 3 or t.shape[-1] != 3:
            raise ValueError(f"image must be [H, W, 3], got {tuple(t.shape)}")
        t = t.permute(2, 0, 1).float().unsqueeze(0)
        H, W = t.shape[-2:]
        h, w = smart_resize(H, W, self.patch * self.merge, self.min_pixels, self.max_pixels)
        if (h, w) != (H, W):
            t = torch.nn.functional.interpolate(t, size=(h, w), mode="bicubic", align_corners=False, antialias=True)
        t = t.clamp(0, 255) / 255.0
        t = (t - torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)) / torch.tensor(CLIP_STD).view(1, 3, 1, 1)
        t = t.repeat(self.tps, 1, 1, 1)                       # a still image fills one temporal patch
        p, m = self.patch, self.merge
        gh, gw = h // p, w // p
        x = t.view(1, self.tps, 3, gh // m, m, p, gw // m, m, p).permute(0, 3, 6, 4, 7, 2, 1, 5, 8)
        return x.reshape(gh * gw, 3 * self.tps * p * p).contiguous(), torch.tensor([1, gh, gw])

    def video_patches(self, video: Any) -> tuple[torch.Tensor, torch.Tensor]:
        """Frames [F, H, W, 3] (uint8) -> patches of F / tps temporal steps; a trailing odd frame is repeated."""
        if isinstance(video, dict):
            key = "pixel_values_videos" if "pixel_values_videos" in video else "pixel_values"
            return torch.as_tensor(video[key]).float(), torch.as_tensor(video["video_grid_thw"]).long().view(3)
        t = torch.as_tensor(np.asarray(video) if not isinstance(video, torch.Tensor) else video)
        if t.dim() != 4 or t.shape[-1] != 3:
            raise ValueError(f"video must be [F, H, W, 3], got {tuple(t.shape)}")
        t = t.permute(0, 3, 1, 2).float()
        F_, _, H, W = t.shape
        h, w = smart_resize(H, W, self.patch * self.merge, self.min_pixels, self.max_pixels)
        if (h, w) != (H, W):
            t = torch.nn.functional.interpolate(t, size=(h, w), mode="bicubic", align_corners=False, antialias=True)
        t = t.clamp(0, 255) / 255.0
        t = (t - torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)) / torch.tensor(CLIP_STD).view(1, 3, 1, 1)
        if F_ % self.tps:
            t = torch.cat([t, t[-1:].expand(self.tps - F_ % self.tps, -1, -1, -1)], dim=0)
        gt, p, m = t.shape[0] // self.tps, self.patch, self.merge
        gh, gw = h // p, w // p
        x = t.view(gt, self.tps, 3, gh // m, m, p, gw // m, m, p).permute(0, 3, 6, 4, 7, 2, 1, 5, 8)
        return x.reshape(gt * gh * gw, 3 * self.tps * p * p).contiguous(), torch.tensor([gt, gh, gw])

    def __call__(self, prompt_token_ids: list[int], mm_data: dict):
        def as_list(x):
            return [] if x is None else (list(x) if isinstance(x, (list, tuple)) else [x])
        images, videos = as_list(mm_data.get("image")), as_list(mm_data.get("video"))
        if not images and not videos:
            raise ValueError("multi_modal_data needs an 'image' or a 'video' entry")
        marks = (self.image_token_id, self.video_token_id)
        runs, prev = [], -2
        for i, tok in enumerate(prompt_token_ids):
            if tok not in marks:
                continue
            if i != prev + 1 or prompt_token_ids[i - 1] != tok:
                runs.append([i, 1, tok])
            else:
                runs[-1][1] += 1
            prev = i
        n_img = sum(1 for r in runs if r[2] == self.image_token_id)
        if n_img != len(images) or len(runs) - n_img != len(videos):
            raise ValueError(f"prompt has {n_img} image and {len(runs) - n_img} video placeholders but "
                             f"{len(images)} images and {len(videos)} videos were given")
        out, positions, inputs, hashes, cur = [], [], [], [], 0
        images, videos = iter(images), iter(videos)
        for start, n, tok in runs:
            if tok == self.image_token_id:
                item = next(images)
                px, grid = self.patches(item)
            else:
                item = next(videos)
                px, grid = self.video_patches(item)
            length = int(grid.prod()) // self.merge ** 2
            out.extend(prompt_token_ids[cur:start])
            positions.append(PlaceholderRange(offset=len(out), length=length))
            out.extend([tok] * length)
            cur = start + n
            extra = []
            if tok == self.video_token_id and self.video_fps:
                sec = item.get("second_per_grid_ts") if isinstance(item, dict) else None
                extra = [torch.tensor(float(sec if sec is not None else self.tps / self.video_fps))]
            inputs.append([px, grid] + extra)
            hashes.append(hash_item(px if isinstance(item, dict) else item))
        out.extend(prompt_token_ids[cur:])
        return out, inputs, hashes, positions


def _qwen2_vl_processor(vllm_config):
    hf = vllm_config.model_config.hf_config
    vc = hf.vision_config if isinstance(hf.vision_config, dict) else hf.vision_config.to_dict()
    proc = PatchGridProcessor(int(getattr(hf, "image_token_id", 151655)), int(vc.get("patch_size", 14)),
                              int(vc.get("spatial_merge_size", 2)), int(vc.get("temporal_patch_size", 2)),
                              56 * 56, 14 * 14 * 4 * 1280, video_token_id=int(getattr(hf, "video_token_id", 151656)))
    if "Qwen2_5_VL" in (vllm_config.model_config.architectures or [""])[0]:
        # frames sampled at ``fps`` (mm_processor_kwargs, default 2): a temporal step spans temporal_patch_size / fps s
        proc.video_fps = float((vllm_config.model_config.mm_processor_kwargs or {}).get("fps", 2.0))
    return proc


MULTIMODAL_REGISTRY.register_processor("Qwen2VLForConditionalGeneration", _qwen2_vl_processor)
MULTIMODAL_REGISTRY.register_processor("Qwen2_5_VLForConditionalGeneration", _qwen2_vl_processor)


# ---- audio: log-mel features for Whisper-style encoders ------------------------------------------------
class AudioTokenProcessor:
    """Prompts mark each clip with an audio placeholder token (one, or an already expanded run); the run is
    resized to the number of embeddings the encoder will produce for the clip:
    mel frames -> (frames - 1) // 2 + 1 after the strided convolution -> (that - 2) // 2 + 1 after pooling.
    ``mm_data["audio"]``: a waveform (1-D float array at the model's sampling rate), a (waveform, rate) pair,
    or a list of either.  Reference: vllm/model_executor/models/qwen2_audio.py (:130-250)."""

    def __init__(self, audio_token_id: int, n_mels: int, sampling_rate: int = 16000, chunk_seconds: int = 30):
        from transformers import WhisperFeatureExtractor
        self.audio_token_id = audio_token_id
        self.rate = sampling_rate
        self.extractor = WhisperFeatureExtractor(feature_size=n_mels, sampling_rate=sampling_rate,
                                                 chunk_length=chunk_seconds)

    def _waveform(self, a):
        import numpy as np
        rate = self.rate
        if isinstance(a, (tuple, list)) and len(a) == 2 and not hasattr(a[1], "__len__"):
            a, rate = a
        w = np.asarray(a.cpu() if torch.is_tensor(a) else a, dtype=np.float32).reshape(-1)
        if int(rate) != self.rate:          # linear resampling: enough for a front end without extra packages
            n = max(int(round(len(w) * self.rate / float(rate))), 1)
            w = np.interp(np.linspace(0, len(w) - 1, n), np.arange(len(w)), w).astype(np.float32)
        return w

    def features(self, a) -> tuple[torch.Tensor, int]:
        w = self._waveform(a)
        out = self.extractor(w, sampling_rate=self.rate, return_attention_mask=True, padding="max_length",
                             return_tensors="pt")
        return out["input_features"][0].float(), int(out["attention_mask"][0].sum())

    @staticmethod
    def num_tokens(mel_frames: int) -> int:
        return ((mel_frames - 1) // 2 + 1 - 2) // 2 + 1

    def __call__(self, prompt_token_ids: list[int], mm_data: dict):
        clips = mm_data.get("audio")
        if clips is None:
            raise ValueError("multi_modal_data needs an 'audio' entry")
        if not isinstance(clips, list):
            clips = [clips]
        runs, prev = [], -2
        for i, t in enumerate(prompt_token_ids):
            if t != self.audio_token_id:
                continue
            if i != prev + 1:
                runs.append([i, 1])
            else:
                runs[-1][1] += 1
            prev = i
        if len(runs) != len(clips):
            raise ValueError(f"prompt has {len(runs)} audio placeholders but {len(clips)} clips were given")
        out, positions, inputs, hashes = [], [], [], []
        cur = 0
        for (start, n), clip in zip(runs, clips):
            feats, mel = self.features(clip)
            ntok = self.num_tokens(mel)
            out.extend(prompt_token_ids[cur:start])
            positions.append(PlaceholderRange(offset=len(out), length=ntok))
            out.extend([self.audio_token_id] * ntok)
            cur = start + n
            inputs.append([feats, torch.tensor(mel, dtype=torch.int64)])
            hashes.append(hash_item(self._waveform(clip)))
        out.extend(prompt_token_ids[cur:])
        return out, inputs, hashes, positions


def _qwen2_audio_processor(vllm_config):
    hf = vllm_config.model_config.hf_config
    ac = hf.audio_config if isinstance(hf.audio_config, dict) else hf.audio_config.to_dict()
    return AudioTokenProcessor(int(getattr(hf, "audio_token_index", None) or getattr(hf, "audio_token_id", 151646)),
                               int(ac.get("num_mel_bins", 128)))


MULTIMODAL_REGISTRY.register_processor("Qwen2AudioForConditionalGeneration", _qwen2_audio_processor)


class WhisperAudioProcessor(AudioTokenProcessor):
    """Encoder-decoder: the clip does not replace prompt tokens.  It is tied to the first decoder token (a
    placeholder range of one token at offset 0), so the encoder is scheduled with the step that holds it."""

    def __call__(self, prompt_token_ids: list[int], mm_data: dict):
        clips = mm_data.get("audio")
        if clips is None:
            raise ValueError("an encoder-decoder audio model needs multi_modal_data with an 'audio' entry")
        if isinstance(clips, list):
            if len(clips) != 1:
                raise ValueError("one audio clip per request")
            clips = clips[0]
        if not prompt_token_ids:
            raise ValueError("the decoder prompt is empty (start-of-transcript and task tokens expected)")
        feats, _ = self.features(clips)
        return list(prompt_token_ids), [[feats]], [hash_item(self._waveform(clips))], [PlaceholderRange(0, 1)]


def _whisper_processor(vllm_config):
    hf = vllm_config.model_config.hf_config
    return WhisperAudioProcessor(-1, int(getattr(hf, "num_mel_bins", 80)))


MULTIMODAL_REGISTRY.register_processor("WhisperForConditionalGeneration", _whisper_processor)


def _fuyu(vllm_config):
    from ..models.fuyu import FuyuImageProcessor
    return FuyuImageProcessor(int(getattr(vllm_config.model_config.hf_config, "patch_size", 30)))


MULTIMODAL_REGISTRY.register_processor("FuyuForCausalLM", _fuyu)


def _pixtral(vllm_config):
    from ..models.pixtral import pixtral_processor
    return pixtral_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("Mistral3ForConditionalGeneration", _pixtral)
MULTIMODAL_REGISTRY.register_processor("PixtralForConditionalGeneration", _pixtral)


def _internvl(vllm_config):
    from ..models.internvl import internvl_processor
    return internvl_processor(vllm_config)


for _a in ("InternVLChatModel", "InternVLForConditionalGeneration", "SkyworkR1VChatModel", "H2OVLChatModel", "NVLM_D"):
    MULTIMODAL_REGISTRY.register_processor(_a, _internvl)


def _llava_next_video(vllm_config):
    from ..models.llava_next import _video_processor
    return _video_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("LlavaNextVideoForConditionalGeneration", _llava_next_video)


class EncoderTextProcessor:
    """Text encoder-decoder: the source tokens are the request's one encoder item, tied to the first decoder token
    (they replace no prompt tokens)."""

    def __call__(self, prompt_token_ids: list[int], mm_data: dict):
        src = mm_data.get("encoder_token_ids")
        if src is None or len(src) == 0:
            raise ValueError("a text encoder-decoder needs multi_modal_data with 'encoder_token_ids'")
        if not prompt_token_ids:
            raise ValueError("the decoder prompt is empty (the decoder start token is expected)")
        t = torch.as_tensor(list(src), dtype=torch.int64)
        return list(prompt_token_ids), [[t]], [hash_item(t)], [PlaceholderRange(0, 1)]


for _a in ("BartForConditionalGeneration", "BartModel"):
    MULTIMODAL_REGISTRY.register_processor(_a, lambda vllm_config: EncoderTextProcessor())


def _tarsier_processor(vllm_config):
    base = _llava_processor(vllm_config)
    side = int(round((base.tokens_per_image) ** 0.5))
    base.tokens_per_image = side * (side + 1) + 1          # a newline position per row and one image-end position
    return base


MULTIMODAL_REGISTRY.register_processor("TarsierForConditionalGeneration", _tarsier_processor)
MULTIMODAL_REGISTRY.register_processor("Tarsier2ForConditionalGeneration", _qwen2_vl_processor)
MULTIMODAL_REGISTRY.register_processor("JinaVLForRanking", _qwen2_vl_processor)


def _paligemma(vllm_config):
    from ..models.paligemma import paligemma_processor
    return paligemma_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("PaliGemmaForConditionalGeneration", _paligemma)


def _aya_vision(vllm_config):
    from ..models.aya_vision import aya_vision_processor
    return aya_vision_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("AyaVisionForConditionalGeneration", _aya_vision)


def _idefics3(vllm_config):
    from ..models.idefics3 import idefics3_processor
    return idefics3_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("Idefics3ForConditionalGeneration", _idefics3)
MULTIMODAL_REGISTRY.register_processor("SmolVLMForConditionalGeneration", _idefics3)


def _ultravox(vllm_config):
    from ..models.ultravox import ultravox_processor
    return ultravox_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("UltravoxModel", _ultravox)


def _chameleon(vllm_config):
    from ..models.chameleon import chameleon_processor
    return chameleon_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("ChameleonForConditionalGeneration", _chameleon)


def _blip2(vllm_config):
    from ..models.blip2 import blip2_processor
    return blip2_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("Blip2ForConditionalGeneration", _blip2)


def _glm4v(vllm_config):
    from ..models.glm4v import glm4v_processor
    return glm4v_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("Glm4vForConditionalGeneration", _glm4v)


def _aria(vllm_config):
    from ..models.aria import aria_processor
    return aria_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("AriaForConditionalGeneration", _aria)


def _granite_speech(vllm_config):
    from ..models.granite_speech import granite_speech_processor
    return granite_speech_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("GraniteSpeechForConditionalGeneration", _granite_speech)


def _qwen2_5_omni(vllm_config):
    from ..models.qwen2_5_omni import qwen2_5_omni_processor
    return qwen2_5_omni_processor(vllm_config)


for _name in ("Qwen2_5OmniForConditionalGeneration", "Qwen2_5OmniModel", "Qwen2_5OmniThinkerForConditionalGeneration"):
    MULTIMODAL_REGISTRY.register_processor(_name, _qwen2_5_omni)


def _qwen_vl(vllm_config):
    from ..models.qwen_vl import qwen_vl_processor
    return qwen_vl_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("QwenVLForConditionalGeneration", _qwen_vl)


def _glm4v_chat(vllm_config):
    from ..models.glm4v_chat import glm4v_chat_processor
    return glm4v_chat_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("GLM4VForCausalLM", _glm4v_chat)


def _phi3v(vllm_config):
    from ..models.phi3v import phi3v_processor
    return phi3v_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("Phi3VForCausalLM", _phi3v)


def _deepseek_vl2(vllm_config):
    from ..models.deepseek_vl2 import deepseek_vl2_processor
    return deepseek_vl2_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("DeepseekVLV2ForCausalLM", _deepseek_vl2)


def _kimi_vl(vllm_config):
    from ..models.kimi_vl import kimi_vl_processor
    return kimi_vl_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("KimiVLForConditionalGeneration", _kimi_vl)


def _minicpmv(vllm_config):
    from ..models.minicpmv import minicpmv_processor
    return minicpmv_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("MiniCPMV", _minicpmv)


def _minicpmo(vllm_config):
    from ..models.minicpmv import minicpmo_processor
    return minicpmo_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("MiniCPMO", _minicpmo)


def _ovis(vllm_config):
    from ..models.ovis import ovis_processor
    return ovis_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("Ovis", _ovis)


def _mllama(vllm_config):
    from ..models.mllama import mllama_processor
    return mllama_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("MllamaForConditionalGeneration", _mllama)


def _florence2(vllm_config):
    from ..models.florence2 import florence2_processor
    return florence2_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("Florence2ForConditionalGeneration", _florence2)


def _keye(vllm_config):
    from ..models.keye import keye_processor
    return keye_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("KeyeForConditionalGeneration", _keye)


def _molmo(vllm_config):
    from ..models.molmo import molmo_processor
    return molmo_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("MolmoForCausalLM", _molmo)


def _phi4mm(vllm_config):
    from ..models.phi4mm import phi4mm_processor
    return phi4mm_processor(vllm_config)


MULTIMODAL_REGISTRY.register_processor("Phi4MMForCausalLM", _phi4mm)
MULTIMODAL_REGISTRY.register_processor("Phi4MultimodalForCausalLM", _phi4mm)
