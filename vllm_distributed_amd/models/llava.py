"""LLaVA-1.5 style vision-language model: CLIP ViT encoder -> 2-layer MLP projector -> Llama decoder.
The image embeddings replace the image placeholder tokens of the prompt; the decoder is the regular
paged-attention Llama.  The encoder runs once per image (its output is cached by the runner).
Reference: vllm/model_executor/models/{llava.py, clip.py}."""
from __future__ import annotations

from typing import Iterable, Optional

import torch
import torch.nn as nn

from ..config import HFConfig, VllmConfig
from ..layers.basic import get_act_fn
from .llama import LlamaForCausalLM
from .utils import load_weights_with_mapping, maybe_prefix


def _cfg(x) -> HFConfig:
    return x if isinstance(x, HFConfig) else HFConfig(dict(x) if isinstance(x, dict) else x.to_dict())


class ClipAttention(nn.Module):
    def __init__(self, H: int, heads: int):
        super().__init__()
        self.heads, self.hd = heads, H // heads
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = (nn.Linear(H, H) for _ in range(4))

    def forward(self, x):
        B, N, H = x.shape
        q, k, v = (p(x).view(B, N, self.heads, self.hd).transpose(1, 2) for p in (self.q_proj, self.k_proj, self.v_proj))
        o = torch.nn.functional.scaled_dot_product_attention(q, k, v)
        return self.out_proj(o.transpose(1, 2).reshape(B, N, H))


class ClipLayer(nn.Module):
    def __init__(self, c):
        super().__init__()
        H = c.hidden_size
        eps = getattr(c, "layer_norm_eps", 1e-5)
        self.self_attn = ClipAttention(H, c.num_attention_heads)
        self.layer_norm1, self.layer_norm2 = nn.LayerNorm(H, eps=eps), nn.LayerNorm(H, eps=eps)
        self.mlp = nn.Module()
        self.mlp.fc1, self.mlp.fc2 = nn.Linear(H, c.intermediate_size), nn.Linear(c.intermediate_size, H)
        self.act = get_act_fn(getattr(c, "hidden_act", "quick_gelu"))

    def forward(self, x):
        x = x + self.self_attn(self.layer_norm1(x))
        return x + self.mlp.fc2(self.act(self.mlp.fc1(self.layer_norm2(x))))


class ClipVisionModel(nn.Module):
    """Only the layers needed up to ``feature_layer`` are built."""

    def __init__(self, c, feature_layer: int):
        super().__init__()
        H, P, S = c.hidden_size, c.patch_size, c.image_size
        self.embeddings = nn.Module()
        self.embeddings.class_embedding = nn.Parameter(torch.zeros(H))
        self.embeddings.patch_embedding = nn.Conv2d(getattr(c, "num_channels", 3), H, P, P, bias=False)
        self.embeddings.position_embedding = nn.Embedding((S // P) ** 2 + 1, H)
        eps = getattr(c, "layer_norm_eps", 1e-5)
        self.pre_layrnorm = nn.LayerNorm(H, eps=eps)
        n = c.num_hidden_layers
        keep = feature_layer + 1 if feature_layer >= 0 else n + feature_layer + 1
        self.encoder = nn.Module()
        self.encoder.layers = nn.ModuleList([ClipLayer(c) for _ in range(keep)])

    def forward(self, pixel_values):
        e = self.embeddings
        x = e.patch_embedding(pixel_values.to(e.patch_embedding.weight.dtype)).flatten(2).transpose(1, 2)
        cls = e.class_embedding.expand(x.shape[0], 1, -1)
        x = torch.cat((cls, x), dim=1) + e.position_embedding.weight.unsqueeze(0)
        x = self.pre_layrnorm(x)
        for layer in self.encoder.layers:
            x = layer(x)
        return x


class LlavaForConditionalGeneration(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False

    def __new__(cls, vllm_config: Optional[VllmConfig] = None, prefix: str = ""):
        # a LLaVA checkpoint over the Pixtral tower (images of any size, break / end tokens) is another model
        if cls is LlavaForConditionalGeneration and vllm_config is not None \
                and _cfg(vllm_config.model_config.hf_config.vision_config).get("model_type") == "pixtral":
            from .pixtral import PixtralForConditionalGeneration
            return PixtralForConditionalGeneration(vllm_config, prefix)
        return super().__new__(cls)

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vc, tc = _cfg(hf.vision_config), _cfg(hf.text_config)
        self.image_token_id = int(getattr(hf, "image_token_index", None) or getattr(hf, "image_token_id", 32000))
        self.select = getattr(hf, "vision_feature_select_strategy", "default")
        fl = getattr(hf, "vision_feature_layer", -2)
        self.vision_tower = nn.Module()
        self.vision_tower.vision_model = ClipVisionModel(vc, int(fl))
        self.multi_modal_projector = nn.Module()
        self.multi_modal_projector.linear_1 = nn.Linear(vc.hidden_size, tc.hidden_size,
                                                        bias=getattr(hf, "multimodal_projector_bias", True))
        self.multi_modal_projector.linear_2 = nn.Linear(tc.hidden_size, tc.hidden_size,
                                                        bias=getattr(hf, "multimodal_projector_bias", True))
        self.proj_act = get_act_fn(getattr(hf, "projector_hidden_act", "gelu"))
        # the decoder sees the text config
        import copy
        sub = copy.copy(vllm_config)
        sub.model_config = copy.copy(vllm_config.model_config)
        sub.model_config.hf_config = tc
        self.language_model = LlamaForCausalLM(sub, prefix=maybe_prefix(prefix, "language_model"))

    # ---- encoder ----------------------------------------------------------------------------------
    def encode_images(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """[n, 3, S, S] -> [n, tokens_per_image, H_text]"""
        feats = self.vision_tower.vision_model(pixel_values)
        if self.select == "default":
            feats = feats[:, 1:]
        p = self.multi_modal_projector
        return p.linear_2(self.proj_act(p.linear_1(feats)))

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
            # current HF layout: model.{language_model,vision_tower,multi_modal_projector}.* + lm_head.*
            # original layout:   language_model.{model,lm_head}.*, vision_tower.*, multi_modal_projector.*
            if n.startswith("model.language_model."):
                return "language_model.model." + n[len("model.language_model."):]
            if n.startswith("model.vision_tower.") or n.startswith("model.multi_modal_projector."):
                n = n[len("model."):]
            if n.startswith("lm_head."):
                return "language_model." + n
            if n.startswith("vision_tower.") and not n.startswith("vision_tower.vision_model."):
                n = "vision_tower.vision_model." + n[len("vision_tower."):]     # layout without the wrapper level
            if n.startswith("vision_tower.vision_model.post_layernorm"):
                return None
            if n.startswith("vision_tower.vision_model.encoder.layers."):
                idx = int(n.split(".")[4])
                if idx >= len(self.vision_tower.vision_model.encoder.layers):
                    return None                      # layers after the feature layer are never run
            return n
        stacked = [(".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"), (".qkv_proj", ".v_proj", "v"),
                   (".gate_up_proj", ".gate_proj", 0), (".gate_up_proj", ".up_proj", 1)]
        # the stacked mapping is for the decoder only: CLIP keeps separate q/k/v
        dec, vis = [], []
        for n, t in weights:
            r = rename(n)
            if r is None:
                continue
            (dec if r.startswith("language_model.") else vis).append((r, t))
        loaded = load_weights_with_mapping(self, dec, stacked, tie_lm_head=self.language_model.tie)
        loaded |= load_weights_with_mapping(self, vis, [])
        return loaded


class TarsierForConditionalGeneration(LlavaForConditionalGeneration):
    """Tarsier: LLaVA whose projected patch grid gets the text embedding of a newline token behind every row and
    of an image-end token behind the last one (h * (w + 1) + 1 positions per image).
    Reference: vllm/model_executor/models/tarsier.py (_add_tarsier_split_tokens :515-559)."""

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__(vllm_config, prefix)
        self.newline_id = int(self.config.image_newline_idx)
        self.new_id = int(self.config.image_new_idx)

    def encode_images(self, pixel_values: torch.Tensor) -> torch.Tensor:
        f = super().encode_images(pixel_values)                     # [n, h * w, H]
        n, t, d = f.shape
        h = int(t ** 0.5)
        w = t // h
        seps = self.language_model.get_input_embeddings(
            torch.tensor([self.newline_id, self.new_id], device=f.device)).to(f.dtype)
        grid = torch.cat((f.view(n, h, w, d), seps[0].expand(n, h, 1, d)), dim=2).reshape(n, h * (w + 1), d)
        return torch.cat((grid, seps[1].expand(n, 1, d)), dim=1)
