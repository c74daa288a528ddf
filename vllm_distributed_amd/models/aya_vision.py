"""Aya Vision: SigLIP tower (last layer, no closing norm) -> 2x2 pixel shuffle -> LayerNorm + SwiGLU projector ->
a Cohere-2 decoder.  Images are cut into up to 12 square tiles plus a thumbnail, every tile contributes
(image_size / patch / 2)^2 positions.  The tile tags of the chat template are text and stay in the prompt as given.
Reference: vllm/model_executor/models/aya_vision.py (AyaVisionMultiModalProjector :59-112, AyaVisionForConditionalGeneration
:300-486).
"""
from __future__ import annotations

from typing import Iterable

import torch
import torch.nn as nn

from ..config import VllmConfig
from .llava import _cfg
from .llava_next import SiglipVisionModel, build_language_model
from .utils import load_weights_with_mapping, maybe_prefix


class AyaVisionProjector(nn.Module):
    def __init__(self, vision_hidden: int, text_hidden: int, factor: int, inter: int, eps: float):
        super().__init__()
        self.factor = factor
        self.layernorm = nn.LayerNorm(vision_hidden * factor * factor, eps=eps)
        self.linear_1 = nn.Linear(vision_hidden * factor * factor, inter, bias=True)
        self.linear_2 = nn.Linear(inter // 2, text_hidden, bias=True)

    def shuffle(self, x: torch.Tensor) -> torch.Tensor:
        """[n, s*s, c] -> [n, (s/f)^2, c*f*f]: f x f neighbouring patches stacked on the channel axis."""
        n, t, c = x.shape
        s, f = int(t ** 0.5), self.factor
        x = x.reshape(n, s, s // f, c * f).permute(0, 2, 1, 3)
        x = x.reshape(n, s // f, s // f, c * f * f).permute(0, 2, 1, 3)
        return x.reshape(n, -1, c * f * f)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        a, gate = self.linear_1(self.layernorm(self.shuffle(x))).chunk(2, dim=-1)
        return self.linear_2(nn.functional.silu(gate) * a)


class AyaVisionForConditionalGeneration(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vc, tc = _cfg(hf.vision_config), _cfg(hf.text_config)
        self.image_token_id = int(getattr(hf, "image_token_index", None) or getattr(hf, "image_token_id", 255036))
        fl = getattr(hf, "vision_feature_layer", -1)
        self.vision_tower = nn.Module()
        self.vision_tower.vision_model = SiglipVisionModel(vc, int(fl))
        self.multi_modal_projector = AyaVisionProjector(
            vc.hidden_size, tc.hidden_size, int(getattr(hf, "downsample_factor", 2)),
            int(getattr(hf, "alignment_intermediate_size", None) or tc.hidden_size),
            float(getattr(hf, "adapter_layer_norm_eps", 1e-6)))
        self.language_model = build_language_model(vllm_config, tc, maybe_prefix(prefix, "language_model"))

    def encode_mm_items(self, items: list[list[torch.Tensor]]) -> list[torch.Tensor]:
        dtype = self.multi_modal_projector.linear_2.weight.dtype
        tiles = torch.cat([it[0].to(dtype) for it in items], dim=0)
        f = self.multi_modal_projector(self.vision_tower.vision_model(tiles))          # [tiles, positions, H]
        out, k = [], 0
        for it in items:
            n = it[0].shape[0]
            out.append(f[k:k + n].flatten(0, 1))
            k += n
        return out

    def get_input_embeddings(self, input_ids, mm_embeds=None, mm_mask=None):
        x = self.language_model.get_input_embeddings(input_ids)
        if mm_embeds is not None and mm_embeds.numel():
            x[mm_mask] = mm_embeds.to(x.dtype)
        return x

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        return self.language_model(input_ids, positions, intermediate_tensors, inputs_embeds)

    def compute_logits(self, hidden_states):
        return self.language_model.compute_logits(hidden_states)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        n_layers = len(self.vision_tower.vision_model.encoder.layers)

        def rename(n: str):
            if n.startswith("model.language_model."):
                return "language_model.model." + n[len("model.language_model."):]
            if n.startswith(("model.vision_tower.", "model.multi_modal_projector.")):
                n = n[len("model."):]
            if n.startswith("lm_head."):
                return "language_model." + n
            if n.startswith("vision_tower.") and not n.startswith("vision_tower.vision_model."):
                n = "vision_tower.vision_model." + n[len("vision_tower."):]
            if n.startswith(("vision_tower.vision_model.post_layernorm", "vision_tower.vision_model.head.")):
                return None
            if n.startswith("vision_tower.vision_model.encoder.layers.") and int(n.split(".")[4]) >= n_layers:
                return None
            return n
        dec, vis = [], []
        for n, t in weights:
            r = rename(n)
            if r is not None:
                (dec if r.startswith("language_model.") else vis).append((r, t))
        lm = self.language_model
        from .llama import LlamaForCausalLM
        if type(lm).load_weights is LlamaForCausalLM.load_weights:
            loaded = load_weights_with_mapping(self, dec, lm.stacked_params_mapping, tie_lm_head=lm.tie)
        else:
            sub = lm.load_weights((n[len("language_model."):], t) for n, t in dec)
            loaded = {"language_model." + n for n in (sub or ())}
        loaded |= load_weights_with_mapping(self, vis, [])
        return loaded


def aya_vision_processor(vllm_config):
    from .internvl import InternVLImageProcessor
    hf = vllm_config.model_config.hf_config
    vc = _cfg(hf.vision_config)
    S, P, f = int(vc.image_size), int(vc.patch_size), int(getattr(hf, "downsample_factor", 2))

    class _P(InternVLImageProcessor):
        mean = std = (0.5, 0.5, 0.5)

    tok = int(getattr(hf, "image_token_index", None) or getattr(hf, "image_token_id", 255036))
    return _P(tok, S, (S // P // f) ** 2, 1, int(getattr(hf, "max_dynamic_patch", 12)), True, True)
