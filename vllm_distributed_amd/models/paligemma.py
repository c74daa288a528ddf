"""PaliGemma / PaliGemma 2: SigLIP tower (all layers and the closing LayerNorm) -> one linear projection -> a
Gemma (or Gemma-2) decoder.  The prompt carries ``(image_size / patch)^2`` image positions in front of the text.

Attention over the prompt is causal, as in the reference's engine (the original model lets the prefix attend in
both directions; generation from a causal prefix is what both the reference and transformers' ``generate`` without
``token_type_ids`` compute).
Reference: vllm/model_executor/models/paligemma.py (PaliGemmaMultiModalProjector :224-236,
PaliGemmaForConditionalGeneration :239-400, the 1/sqrt(hidden) scaling of image features :349-351).
"""
from __future__ import annotations

from typing import Iterable

import torch
import torch.nn as nn

from ..config import VllmConfig
from .llava import _cfg
from .llava_next import SiglipVisionModel, build_language_model
from .utils import load_weights_with_mapping, maybe_prefix


class PaliGemmaForConditionalGeneration(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vc, tc = _cfg(hf.vision_config), _cfg(hf.text_config)
        self.image_token_id = int(getattr(hf, "image_token_id", None) or getattr(hf, "image_token_index", 257152))
        self.vision_tower = nn.Module()
        vm = self.vision_tower.vision_model = SiglipVisionModel(vc, -1)
        vm.post_layernorm = nn.LayerNorm(vc.hidden_size, eps=float(getattr(vc, "layer_norm_eps", 1e-6)))
        self.multi_modal_projector = nn.Module()
        out = int(getattr(vc, "projection_dim", None) or tc.hidden_size)
        self.multi_modal_projector.linear = nn.Linear(vc.hidden_size, out, bias=True)
        self.language_model = build_language_model(vllm_config, tc, maybe_prefix(prefix, "language_model"))
        # the decoder multiplies every input embedding by sqrt(hidden); image features enter unscaled
        self.feature_scale = float(tc.hidden_size) ** -0.5

    def encode_images(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """[n, 3, S, S] -> [n, (S / patch)^2, H_text]"""
        vm = self.vision_tower.vision_model
        x = vm.post_layernorm(vm(pixel_values))
        return self.multi_modal_projector.linear(x) * self.feature_scale

    def get_input_embeddings(self, input_ids, mm_embeds=None, mm_mask=None):
        ids = input_ids
        if self.image_token_id >= self.language_model.config.vocab_size:      # placeholder outside the vocabulary
            ids = torch.where(input_ids == self.image_token_id, 0, input_ids)
        x = self.language_model.get_input_embeddings(ids)
        if mm_embeds is not None and mm_embeds.numel():
            x[mm_mask] = mm_embeds.to(x.dtype)
        return x

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        return self.language_model(input_ids, positions, intermediate_tensors, inputs_embeds)

    def compute_logits(self, hidden_states):
        return self.language_model.compute_logits(hidden_states)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        def rename(n: str):
            if n.startswith("model.language_model."):
                return "language_model.model." + n[len("model.language_model."):]
            if n.startswith(("model.vision_tower.", "model.multi_modal_projector.")):
                n = n[len("model."):]
            if n.startswith("lm_head."):
                return "language_model." + n
            if n.startswith("vision_tower.") and not n.startswith("vision_tower.vision_model."):
                n = "vision_tower.vision_model." + n[len("vision_tower."):]     # layout without the wrapper level
            if n.startswith("vision_tower.vision_model.head."):
                return None
            return n
        dec, vis = [], []
        for n, t in weights:
            r = rename(n)
            if r is not None:
                (dec if r.startswith("language_model.") else vis).append((r, t))
        lm = self.language_model
        loaded = load_weights_with_mapping(self, dec, lm.stacked_params_mapping, tie_lm_head=lm.tie)
        loaded |= load_weights_with_mapping(self, vis, [])
        return loaded


def paligemma_processor(vllm_config):
    from ..multimodal import ImageTokenProcessor, to_pixel_values
    hf = vllm_config.model_config.hf_config
    vc = _cfg(hf.vision_config)
    size, patch = int(vc.image_size), int(vc.patch_size)
    tok = int(getattr(hf, "image_token_id", None) or getattr(hf, "image_token_index", 257152))

    class _P(ImageTokenProcessor):
        """SigLIP preprocessing: plain resize to the square input (no crop), mean = std = 0.5."""

        def __call__(self, ids, mm_data):
            imgs = mm_data.get("image")
            if imgs is not None:
                imgs = imgs if isinstance(imgs, (list, tuple)) else [imgs]
                mm_data = dict(mm_data, image=[self._pre(i) for i in imgs])
            return super().__call__(ids, mm_data)

        def _pre(self, img):
            if isinstance(img, torch.Tensor) and img.dtype.is_floating_point:
                return img
            import numpy as np
            a = np.asarray(img.convert("RGB")) if hasattr(img, "convert") else np.asarray(img)
            t = torch.as_tensor(a).permute(2, 0, 1).float()[None]
            t = nn.functional.interpolate(t, size=(size, size), mode="bicubic", align_corners=False, antialias=True)
            return (t[0].clamp(0, 255) / 255.0 - 0.5) / 0.5

    return _P(tok, (size // patch) ** 2, size)
