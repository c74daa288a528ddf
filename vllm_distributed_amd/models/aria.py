"""Aria: a SigLIP-style tower over images padded to a 490 or 980 pixel square, a projector whose learned queries
(128 or 256 of them, by the padded patch count) cross-attend to the image patches, and a Llama decoder whose every
MLP is a mixture of experts (top-k on the router logits, softmax over the chosen ones) beside always-on shared experts.

The padded part of an image never influences the valid part (padded patches are masked as keys everywhere), so the
tower runs on the valid top-left rectangle only; its position ids are the checkpoint's square table bucketed to the
rectangle, as for a masked run.
Reference: vllm/model_executor/models/aria.py (AriaProjector :127-193, AriaFusedMoE / AriaTextMoELayer :196-293,
AriaTextDecoderLayer :296-316, AriaForConditionalGeneration :493-660).
"""
from __future__ import annotations

from typing import Iterable

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from ..layers.linear import ReplicatedLinear
from ..layers.moe import FusedMoE
from ..parallel.state import get_tensor_model_parallel_world_size, tensor_model_parallel_all_reduce
from .llama import LlamaDecoderLayer, LlamaForCausalLM, LlamaMLP, LlamaModel
from .llava import ClipLayer, _cfg
from .utils import load_weights_with_mapping, maybe_prefix


# ---- decoder ---------------------------------------------------------------------------------------------
class AriaMoE(nn.Module):
    def __init__(self, config, quant_config=None, prefix: str = ""):
        super().__init__()
        E, H, I = int(config.moe_num_experts), int(config.hidden_size), int(config.intermediate_size)
        self.tp = get_tensor_model_parallel_world_size()
        self.router = ReplicatedLinear(H, E, bias=False, quant_config=None, prefix=f"{prefix}.router")
        # softmax over the top-k logits = softmax over all of them, renormalised over the chosen ones
        self.experts = FusedMoE(E, int(config.moe_topk), H, I, reduce_results=False, renormalize=True,
                                quant_config=quant_config, prefix=f"{prefix}.experts")
        self.shared_experts = LlamaMLP(H, I * int(config.moe_num_shared_experts),
                                       bool(getattr(config, "mlp_bias", False)), quant_config,
                                       f"{prefix}.shared_experts", reduce_results=False)

    def forward(self, x):
        out = self.experts(x, self.router(x)) + self.shared_experts(x)
        if self.tp > 1 or self.experts.ep_size > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out


class AriaTextDecoderLayer(LlamaDecoderLayer):
    supports_sequence_parallel = False

    def _make_mlp(self, config, vllm_config, quant_config, prefix):
        return AriaMoE(config, quant_config, prefix)


class AriaTextModel(LlamaModel):
    layer_cls = AriaTextDecoderLayer


class AriaTextForCausalLM(LlamaForCausalLM):
    model_cls = AriaTextModel
    supports_lora = False

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]) -> set[str]:
        experts = FusedMoE.make_expert_params_mapping("gate_proj", "down_proj", "up_proj",
                                                      int(self.config.moe_num_experts))

        def fix(items):
            for name, t in items:
                # all experts of a layer are stored stacked and input-major: fc1 [E, H, 2I] (gate | up), fc2 [E, I, H]
                if name.endswith(".mlp.experts.fc1.weight"):
                    base, half = name[: -len("fc1.weight")], t.shape[2] // 2
                    for e in range(t.shape[0]):
                        yield f"{base}{e}.gate_proj.weight", t[e, :, :half].t()
                        yield f"{base}{e}.up_proj.weight", t[e, :, half:].t()
                elif name.endswith(".mlp.experts.fc2.weight"):
                    base = name[: -len("fc2.weight")]
                    for e in range(t.shape[0]):
                        yield f"{base}{e}.down_proj.weight", t[e].t()
                else:
                    yield name, t

        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping, tie_lm_head=self.tie,
                                         expert_mapping=experts)


# ---- vision ----------------------------------------------------------------------------------------------
class AriaVisionTower(nn.Module):
    def __init__(self, c, feature_layer: int):
        super().__init__()
        H, self.patch = int(c.hidden_size), int(c.patch_size)
        self.side = int(c.image_size) // self.patch
        self.embeddings = nn.Module()
        self.embeddings.patch_embedding = nn.Conv2d(int(getattr(c, "num_channels", 3)), H, self.patch, self.patch)
        self.embeddings.position_embedding = nn.Embedding(self.side ** 2, H)
        n = int(c.num_hidden_layers)
        keep = feature_layer + 1 if feature_layer >= 0 else n + feature_layer + 1
        if not getattr(c, "hidden_act", None):
            c.hidden_act = "gelu_pytorch_tanh"
        self.encoder = nn.Module()
        self.encoder.layers = nn.ModuleList(ClipLayer(c) for _ in range(keep))

    def _position_ids(self, h: int, w: int, device) -> torch.Tensor:
        """Rows / columns of an h x w patch rectangle spread over the square position table."""
        bounds = torch.arange(1 / self.side, 1.0, 1 / self.side, device=device)

        def axis(n):
            frac = (torch.arange(n, device=device, dtype=torch.float32) * (1.0 / n)).clamp(max=1.0 - 1e-6)
            return torch.bucketize(frac, bounds, right=True)
        return (axis(h)[:, None] * self.side + axis(w)[None, :]).flatten()

    def forward(self, pixels: torch.Tensor) -> torch.Tensor:
        """[3, h * patch, w * patch] (the valid rectangle) -> [h * w, H]"""
        e = self.embeddings
        x = e.patch_embedding(pixels[None].to(e.patch_embedding.weight.dtype))
        h, w = x.shape[-2:]
        x = x.flatten(2).transpose(1, 2) + e.position_embedding(self._position_ids(h, w, x.device))[None]
        for layer in self.encoder.layers:
            x = layer(x)
        return x[0]


class AriaProjector(nn.Module):
    def __init__(self, hf, vc, tc):
        super().__init__()
        H, heads, T = int(vc.hidden_size), int(vc.num_attention_heads), int(tc.hidden_size)
        self.queries_for = {int(k): int(v) for k, v in dict(hf.projector_patch_to_query_dict).items()}
        self.query = nn.Parameter(torch.zeros(max(self.queries_for.values()), H))
        ca = self.cross_attn = nn.Module()
        ca.q_proj, ca.k_proj, ca.v_proj = (nn.Linear(H, H, bias=False) for _ in range(3))
        ca.multihead_attn = nn.MultiheadAttention(H, heads, batch_first=True)
        ca.linear = nn.Linear(H, H)
        ca.layer_norm, ca.layer_norm_kv = nn.LayerNorm(H), nn.LayerNorm(H)
        self.layer_norm = nn.LayerNorm(H)
        self.feed_forward = nn.Module()
        self.feed_forward.linear_in = nn.Linear(H, T, bias=False)
        self.feed_forward.linear_out = nn.Linear(T, T, bias=False)

    def forward(self, patches: torch.Tensor, padded_patches: int) -> torch.Tensor:
        """patches [n_valid, H]; the number of queries follows the padded image's patch count."""
        if padded_patches not in self.queries_for:
            raise KeyError(f"no query count for images of {padded_patches} patches (known: {sorted(self.queries_for)})")
        ca = self.cross_attn
        q = ca.q_proj(ca.layer_norm(self.query[: self.queries_for[padded_patches]]))[None]
        kv = ca.layer_norm_kv(patches)[None]
        a, _ = ca.multihead_attn(q, ca.k_proj(kv), ca.v_proj(kv), need_weights=False)
        x = self.layer_norm(ca.linear(a))
        ff = self.feed_forward
        return ff.linear_out(F.gelu(ff.linear_in(x), approximate="tanh"))[0]


class AriaForConditionalGeneration(nn.Module):
    is_multimodal = True
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        import copy
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vc, tc = _cfg(hf.vision_config), _cfg(hf.text_config)
        self.image_token_id = int(getattr(hf, "image_token_index", None) or getattr(hf, "image_token_id", 9))
        self.patch = int(vc.patch_size)
        self.vision_tower = AriaVisionTower(vc, int(getattr(hf, "vision_feature_layer", -1)))
        self.multi_modal_projector = AriaProjector(hf, vc, tc)
        sub = copy.copy(vllm_config)
        sub.model_config = copy.copy(vllm_config.model_config)
        sub.model_config.hf_config = tc
        self.language_model = AriaTextForCausalLM(sub, prefix=maybe_prefix(prefix, "language_model"))

    def encode_mm_items(self, items: list) -> list[torch.Tensor]:
        """items: (pixels [3, S, S], valid size [2] in pixels) per image."""
        outs = []
        for px, valid in items:
            h, w = (int(v) for v in valid.tolist())
            hp, wp = -(-h // self.patch), -(-w // self.patch)        # a patch with any valid pixel counts
            feats = self.vision_tower(px[:, : hp * self.patch, : wp * self.patch])
            outs.append(self.multi_modal_projector(feats, (px.shape[-1] // self.patch) ** 2))
        return outs

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
        dec, own = [], []
        for n, t in weights:
            if n.startswith("model.language_model."):
                dec.append(("model." + n[len("model.language_model."):], t))
            elif n.startswith("language_model."):
                dec.append((n[len("language_model."):], t))
            elif n.startswith("lm_head."):
                dec.append((n, t))
            else:
                n = n[len("model."):] if n.startswith("model.") else n
                n = n.replace("vision_tower.vision_model.", "vision_tower.")
                if ".post_layernorm." in n or ".head." in n:
                    continue
                own.append((n, t))
        loaded = {"language_model." + n for n in self.language_model.load_weights(dec)}
        loaded |= load_weights_with_mapping(self, own, [])
        return loaded


def aria_processor(vllm_config):
    """One placeholder per image -> 128 / 256 positions.  Pixels: longest edge to ``max_image_size`` (490, or 980
    with ``mm_processor_kwargs={"max_image_size": 980}``), aspect ratio kept, padded bottom / right to the square,
    mean = std = 0.5.  Pre-processed inputs: a dict with ``pixel_values`` [3, S, S] and ``valid_size`` (h, w)."""
    from ..multimodal import PlaceholderRange, hash_item
    hf = vllm_config.model_config.hf_config
    vc = _cfg(hf.vision_config)
    patch = int(vc.patch_size)
    tok = int(getattr(hf, "image_token_index", None) or getattr(hf, "image_token_id", 9))
    queries = {int(k): int(v) for k, v in dict(hf.projector_patch_to_query_dict).items()}
    sizes = sorted({int(round(k ** 0.5)) * patch for k in queries})
    kw = getattr(vllm_config.model_config, "mm_processor_kwargs", None) or {}
    target = int(kw.get("max_image_size", sizes[0]))

    def pixels(img):
        if isinstance(img, dict):
            px = torch.as_tensor(img["pixel_values"]).float()
            return px, torch.tensor([int(v) for v in img["valid_size"]])
        import numpy as np
        a = np.asarray(img.convert("RGB")) if hasattr(img, "convert") else np.asarray(img)
        t = torch.as_tensor(a).permute(2, 0, 1).float()[None]
        h, w = t.shape[-2:]
        s = target / max(h, w)
        nh, nw = max(int(h * s), 1), max(int(w * s), 1)
        t = F.interpolate(t, size=(nh, nw), mode="bicubic", align_corners=False, antialias=True).clamp(0, 255)
        px = torch.zeros(3, target, target)
        px[:, :nh, :nw] = (t[0] / 255.0 - 0.5) / 0.5
        return px, torch.tensor([nh, nw])

    def process(ids, mm_data):
        images = mm_data.get("image")
        if images is None:
            raise ValueError("multi_modal_data needs an 'image' entry")
        images = images if isinstance(images, (list, tuple)) else [images]
        out, ranges, inputs, hashes = [], [], [], []
        i, k = 0, 0
        while i < len(ids):
            if ids[i] != tok:
                out.append(ids[i])
                i += 1
                continue
            j = i
            while j < len(ids) and ids[j] == tok:
                j += 1
            if k >= len(images):
                raise ValueError("more image placeholders than images")
            px, valid = pixels(images[k])
            n = queries[(px.shape[-1] // patch) ** 2]
            ranges.append(PlaceholderRange(offset=len(out), length=n))
            out.extend([tok] * n)
            inputs.append((px, valid))
            hashes.append(hash_item(images[k]["pixel_values"] if isinstance(images[k], dict) else images[k]))
            k, i = k + 1, j
        if k != len(images):
            raise ValueError(f"prompt has {k} image placeholders but {len(images)} images were given")
        return out, inputs, hashes, ranges

    return process
