"""GLM-4.1V (``Glm4vForConditionalGeneration``): a patch-grid vision tower in front of a GLM-4 decoder with
multimodal rotary positions.

The tower differs from Qwen2.5-VL's in four places: a learned square position table is resampled (bicubic, at the
patch centres) to every image's own grid and added after an RMSNorm on the patch embedding; all blocks attend over
the whole frame (no windows); the 2 x 2 merge is a strided convolution; and the merger is projection -> LayerNorm
-> GELU -> gated MLP.  The decoder rotates interleaved pairs over half of each head, with the frequency pairs split
into temporal / height / width sections.

Images only: the family's video prompts (one framed run of placeholders per sampled frame) are not expanded here.
Reference: vllm/model_executor/models/glm4_1v.py (Glm4vVisionEmbeddings :492-560, Glm4vPatchMerger :430-488,
Glm4vVisionTransformer :620-815, Glm4vForConditionalGeneration :1240-1580).
"""
from __future__ import annotations

import copy

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from ..layers.basic import get_act_fn
from .llama_variants import Glm4ForCausalLM
from .llava import _cfg
from .qwen2_vl import Qwen2VisionTransformer, Qwen2VLForConditionalGeneration, VisionBlock25, _RMS
from .utils import maybe_prefix


class _GlmVisionBlock(VisionBlock25):
    def __init__(self, dim: int, heads: int, mlp_dim: int, qkv_bias: bool, eps: float):
        nn.Module.__init__(self)
        self.heads = heads
        self.norm1, self.norm2 = _RMS(dim, eps), _RMS(dim, eps)
        self.attn = nn.Module()
        self.attn.qkv, self.attn.proj = nn.Linear(dim, 3 * dim, bias=qkv_bias), nn.Linear(dim, dim, bias=False)
        self.mlp = nn.Module()
        self.mlp.gate_proj, self.mlp.up_proj = nn.Linear(dim, mlp_dim, bias=False), nn.Linear(dim, mlp_dim, bias=False)
        self.mlp.down_proj = nn.Linear(mlp_dim, dim, bias=False)


class Glm4vVisionTransformer(Qwen2VisionTransformer):
    def __init__(self, vc):
        nn.Module.__init__(self)
        self.embed_dim, self.heads = int(vc.hidden_size), int(vc.num_heads)
        self.patch, self.tps, self.merge = int(vc.patch_size), int(vc.temporal_patch_size), int(vc.spatial_merge_size)
        out, eps = int(vc.out_hidden_size), float(getattr(vc, "rms_norm_eps", 1e-5))
        self.out_dim = out
        self.patch_embed = nn.Module()
        self.patch_embed.proj = nn.Conv3d(int(vc.in_channels), self.embed_dim, (self.tps, self.patch, self.patch),
                                          (self.tps, self.patch, self.patch))
        self.post_conv_layernorm = _RMS(self.embed_dim, eps)
        self.embeddings = nn.Module()
        self.embeddings.position_embedding = nn.Embedding((int(vc.image_size) // self.patch) ** 2, self.embed_dim)
        self.blocks = nn.ModuleList(_GlmVisionBlock(self.embed_dim, self.heads, out,
                                                    bool(getattr(vc, "attention_bias", False)), eps)
                                    for _ in range(int(vc.depth)))
        self.post_layernorm = _RMS(self.embed_dim, eps)
        self.downsample = nn.Conv2d(self.embed_dim, out, self.merge, self.merge)
        mg = self.merger = nn.Module()
        mg.proj = nn.Linear(out, out, bias=False)
        mg.post_projection_norm = nn.LayerNorm(out)
        inter = int(vc.intermediate_size)
        mg.gate_proj, mg.up_proj = nn.Linear(out, inter, bias=False), nn.Linear(out, inter, bias=False)
        mg.down_proj = nn.Linear(inter, out, bias=False)
        self.act = get_act_fn(getattr(vc, "hidden_act", "silu"))
        for p in self.parameters():
            p.requires_grad_(False)

    def _coords(self, grids, device):
        """(row, column) of every patch, in the processor's order (2 x 2 merge groups kept together), and the
        (height, width) of the grid the patch belongs to."""
        ids, size, m = [], [], self.merge
        for t, h, w in grids:
            hp, wp = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
            shape = (h // m, m, w // m, m)
            hp = hp.reshape(shape).transpose(1, 2).flatten()
            wp = wp.reshape(shape).transpose(1, 2).flatten()
            ids.append(torch.stack([hp, wp], -1).repeat(t, 1))
            size.append(torch.tensor([[h, w]], device=device).expand(t * h * w, 2))
        return torch.cat(ids).float(), torch.cat(size).float()

    def _position_embedding(self, coords, size):
        table = self.embeddings.position_embedding.weight
        side = int(table.shape[0] ** 0.5)
        grid2d = table.view(side, side, -1).permute(2, 0, 1)[None].float()
        xy = torch.stack([(coords[:, 1] + 0.5) / size[:, 1], (coords[:, 0] + 0.5) / size[:, 0]], -1) * 2 - 1
        out = F.grid_sample(grid2d, xy[None, :, None, :], mode="bicubic", align_corners=False, padding_mode="border")
        return out[0, :, :, 0].t().to(table.dtype)

    def forward(self, patches: torch.Tensor, grids: list[tuple[int, int, int]]) -> torch.Tensor:
        w = self.patch_embed.proj.weight
        x = F.linear(patches.to(w.dtype), w.view(self.embed_dim, -1), self.patch_embed.proj.bias)
        x = self.post_conv_layernorm(x)
        cos, sin = self._rotary(grids, x.device)
        coords, size = self._coords(grids, x.device)
        x = x + self._position_embedding(coords, size)
        spans = [h * w_ for t, h, w_ in grids for _ in range(t)]
        for blk in self.blocks:
            x = blk(x, cos, sin, spans)
        x = self.post_layernorm(x)
        m = self.merge
        x = self.downsample(x.view(-1, m, m, self.embed_dim).permute(0, 3, 1, 2)).view(-1, self.out_dim)
        mg = self.merger
        x = F.gelu(mg.post_projection_norm(mg.proj(x)))
        return mg.down_proj(self.act(mg.gate_proj(x)) * mg.up_proj(x))


class Glm4vForConditionalGeneration(Qwen2VLForConditionalGeneration):
    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        nn.Module.__init__(self)
        hf = vllm_config.model_config.hf_config
        self.config = hf
        vc, tc = _cfg(hf.vision_config), _cfg(hf.text_config)
        self.merge = int(vc.spatial_merge_size)
        self.image_token_id = int(getattr(hf, "image_token_id", 151343))
        self.visual = Glm4vVisionTransformer(vc)
        tc.attention_bias = True                      # q / k / v carry a bias in this family, o does not
        rp = dict(getattr(tc, "rope_parameters", None) or getattr(tc, "rope_scaling", None) or {})
        rp.setdefault("rope_type", "default")
        rp.setdefault("mrope_section", [8, 12, 12])
        tc.rope_parameters = rp
        if getattr(tc, "tie_word_embeddings", None) is None:
            tc.tie_word_embeddings = bool(getattr(hf, "tie_word_embeddings", False))
        sub = copy.copy(vllm_config)
        sub.model_config = copy.copy(vllm_config.model_config)
        sub.model_config.hf_config = tc
        self.language_model = Glm4ForCausalLM(sub, prefix=maybe_prefix(prefix, "language_model"))
        lm = self.language_model.model
        self.rotary = lm.layers[lm.start_layer].self_attn.rotary_emb

    def load_weights(self, weights):
        from .utils import load_weights_with_mapping

        def rename(n: str) -> str:
            if n.startswith("model.language_model."):
                return "language_model.model." + n[len("model.language_model."):]
            if n.startswith("model.visual."):
                return n[len("model."):]
            if n.startswith("visual."):
                return n
            return "language_model." + n
        dec, vis = [], []
        for n, t in weights:
            r = rename(n)
            (vis if r.startswith("visual.") else dec).append((r, t))
        loaded = load_weights_with_mapping(self, dec, Glm4ForCausalLM.stacked_params_mapping,
                                           tie_lm_head=self.language_model.tie)
        loaded |= load_weights_with_mapping(self, vis, [])
        return loaded


def glm4v_processor(vllm_config):
    from ..multimodal import PatchGridProcessor
    hf = vllm_config.model_config.hf_config
    vc = _cfg(hf.vision_config)
    return PatchGridProcessor(int(getattr(hf, "image_token_id", 151343)), int(vc.patch_size),
                              int(vc.spatial_merge_size), int(vc.temporal_patch_size), 112 * 112, 28 * 28 * 2 * 30000,
                              video_token_id=int(getattr(hf, "video_token_id", 151344)))
