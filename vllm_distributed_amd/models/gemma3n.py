"""Gemma-3n (text): a Gemma decoder whose residual stream is ``altup_num_inputs`` parallel copies ("alternating
updates").  Per layer a small router predicts all copies from each other, only the active copy goes through
attention + MLP, and the result corrects every copy.  Around that core:

* LAuReL: a low-rank residual branch added to the attention output, the sum divided by sqrt(2);
* per-layer inputs: a second embedding table holds a narrow vector per (token, layer), mixed with a projection of the
  token embedding and gated into every copy but the active one;
* activation sparsity: the first MLPs keep only gate values above mean + z * std of their row;
* attention with softmax scale 1, RMS-normalised q, k (learned scale) and v (no scale), sliding / global layers
  with rotary bases of their own, and the last ``num_kv_shared_layers`` layers reuse the keys and values of the
  last earlier layer of their kind: they have no k / v projections and no cache of their own.

The vision and audio towers of the checkpoints are not loaded (text generation only), as in the reference.
Reference: vllm/model_executor/models/gemma3n.py (Gemma3nAltUp :55-164, Gemma3nLaurelBlock :167-195,
Gemma3nMLP :198-239, Gemma3nAttention :242-365, Gemma3nDecoderLayer :368-508, Gemma3nTextModel :511-730).
"""
from __future__ import annotations

import math
from typing import Iterable

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from ..layers.attention import Attention
from ..layers.basic import LogitsProcessor, ParallelLMHead, RMSNorm, VocabParallelEmbedding
from ..layers.linear import ColumnParallelLinear, MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from ..layers.rotary import get_rope
from ..parallel.state import (get_tensor_model_parallel_world_size, tensor_model_parallel_all_reduce)
from .llava import _cfg
from .utils import load_weights_with_mapping


def _layer_types(config) -> list[str]:
    kinds = getattr(config, "layer_types", None)
    if kinds:
        return list(kinds)
    return ["full_attention" if (i + 1) % 5 == 0 else "sliding_attention" for i in range(config.num_hidden_layers)]


def _rope_theta(config, kind: str) -> float:
    rp = getattr(config, "rope_parameters", None) or {}
    if kind in rp and isinstance(rp[kind], dict):
        return float(rp[kind].get("rope_theta", 10000.0))
    if kind == "sliding_attention":
        return float(getattr(config, "rope_local_base_freq", 10000.0))
    return float(getattr(config, "rope_theta", None) or rp.get("rope_theta") or 1000000.0)


class Gemma3nAltUp(nn.Module):
    def __init__(self, config):
        super().__init__()
        n, d = int(config.altup_num_inputs), int(config.hidden_size)
        self.n, self.active = n, int(config.altup_active_idx)
        self.correct_output_scale = nn.Parameter(torch.zeros(d))
        self.correction_coefs = nn.Linear(n, n, bias=False)
        self.prediction_coefs = nn.Linear(n, n * n, bias=False)
        self.modality_router = nn.Linear(d, n, bias=False)
        self.router_norm = RMSNorm(d, config.rms_norm_eps)
        self.router_scale = 1.0 / d

    def _modalities(self, x):
        return torch.tanh(self.modality_router(self.router_norm(x) * self.router_scale).float()).to(x.dtype)

    def predict(self, h: torch.Tensor) -> torch.Tensor:
        """h [n, T, d] -> predictions [n, T, d]: every copy plus a token-dependent mix of all copies."""
        coefs = self.prediction_coefs(self._modalities(h[self.active])).view(-1, self.n, self.n)     # [T, out, in]
        return (torch.einsum("itd,toi->otd", h.float(), coefs.float()) + h.float()).to(h.dtype)

    def correct(self, predictions: torch.Tensor, activated: torch.Tensor) -> torch.Tensor:
        coefs = self.correction_coefs(self._modalities(activated)) + 1.0                               # [T, n]
        innovation = activated - predictions[self.active]
        return predictions + innovation[None] * coefs.t()[:, :, None]


class Gemma3nLaurel(nn.Module):
    def __init__(self, config):
        super().__init__()
        d, r = int(config.hidden_size), int(config.laurel_rank)
        self.linear_left, self.linear_right = nn.Linear(d, r, bias=False), nn.Linear(r, d, bias=False)
        self.post_laurel_norm = RMSNorm(d, config.rms_norm_eps)

    def forward(self, x):
        return x + self.post_laurel_norm(self.linear_right(self.linear_left(x)))


class Gemma3nMLP(nn.Module):
    def __init__(self, d: int, inter: int, sparsity: float, quant_config, prefix: str):
        super().__init__()
        self.gate_up_proj = MergedColumnParallelLinear(d, [inter, inter], bias=False, quant_config=quant_config,
                                                       prefix=f"{prefix}.gate_up_proj")
        self.down_proj = RowParallelLinear(inter, d, bias=False, quant_config=quant_config, prefix=f"{prefix}.down_proj")
        self.inter = inter
        # cut-off in standard deviations above the row mean: the ``sparsity`` quantile of a normal distribution
        self.z = float(torch.distributions.Normal(0.0, 1.0).icdf(torch.tensor(float(sparsity)))) if sparsity > 0 else None

    def forward(self, x):
        gate, up = self.gate_up_proj(x).chunk(2, -1)
        if self.z is not None:
            g = gate.float()
            s1, s2 = g.sum(-1, keepdim=True), (g * g).sum(-1, keepdim=True)
            if get_tensor_model_parallel_world_size() > 1:           # the statistics run over the whole row
                both = tensor_model_parallel_all_reduce(torch.cat([s1, s2], -1))
                s1, s2 = both[:, :1], both[:, 1:]
            mean = s1 / self.inter
            std = (s2 / self.inter - mean * mean).clamp_min(0).sqrt()
            gate = F.relu(g - (mean + std * self.z)).to(x.dtype)
        return self.down_proj(F.gelu(gate, approximate="tanh") * up)


class Gemma3nAttention(nn.Module):
    def __init__(self, config, vllm_config: VllmConfig, layer_idx: int, kind: str, source, prefix: str):
        super().__init__()
        quant_config = getattr(vllm_config, "quant_config", None)
        tp = get_tensor_model_parallel_world_size()
        d, self.head_dim = int(config.hidden_size), int(config.head_dim)
        hq, hkv = int(config.num_attention_heads), int(config.num_key_value_heads)
        self.num_heads, self.num_kv_heads = hq // tp, max(1, hkv // tp)
        self.q_size, self.kv_size = self.num_heads * self.head_dim, self.num_kv_heads * self.head_dim
        bias = bool(getattr(config, "attention_bias", False))
        self.shared = source is not None
        if self.shared:
            self.q_proj = ColumnParallelLinear(d, hq * self.head_dim, bias=bias, quant_config=quant_config,
                                               prefix=f"{prefix}.q_proj")
        else:
            self.qkv_proj = QKVParallelLinear(d, self.head_dim, hq, hkv, bias=bias, quant_config=quant_config,
                                              prefix=f"{prefix}.qkv_proj")
            self.k_norm = RMSNorm(self.head_dim, config.rms_norm_eps)
        self.q_norm = RMSNorm(self.head_dim, config.rms_norm_eps)
        self.o_proj = RowParallelLinear(hq * self.head_dim, d, bias=bias, quant_config=quant_config,
                                        prefix=f"{prefix}.o_proj")
        self.eps = float(config.rms_norm_eps)
        self.rotary_emb = get_rope(self.head_dim, self.head_dim, vllm_config.model_config.max_model_len,
                                   _rope_theta(config, kind), True, None, 1.0)
        window = int(config.sliding_window) if kind == "sliding_attention" else None
        self.attn = Attention(self.num_heads, self.head_dim, 1.0, self.num_kv_heads, sliding_window=window,
                              prefix=f"{prefix}.attn", cache_dtype=vllm_config.cache_config.cache_dtype,
                              kv_source=source.attn if source is not None else None)

    def forward(self, positions, x):
        T = x.shape[0]
        if self.shared:
            q = self.q_proj(x)
            q = self.q_norm(q.view(T * self.num_heads, self.head_dim)).view(T, self.q_size)
            return self.o_proj(self.attn(q, positions, self.rotary_emb))
        qkv = self.qkv_proj(x)
        qs, ks = self.q_size, self.kv_size
        qkv[:, :qs] = self.q_norm(qkv[:, :qs].reshape(T * self.num_heads, self.head_dim)).view(T, qs)
        qkv[:, qs:qs + ks] = self.k_norm(qkv[:, qs:qs + ks].reshape(T * self.num_kv_heads, self.head_dim)).view(T, ks)
        v = qkv[:, qs + ks:].reshape(T, self.num_kv_heads, self.head_dim).float()
        v = v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + self.eps)
        qkv[:, qs + ks:] = v.to(qkv.dtype).view(T, ks)
        return self.o_proj(self.attn(qkv, positions, self.rotary_emb))


class Gemma3nDecoderLayer(nn.Module):
    def __init__(self, config, vllm_config: VllmConfig, layer_idx: int, kind: str, source, prefix: str):
        super().__init__()
        quant_config = getattr(vllm_config, "quant_config", None)
        d, eps = int(config.hidden_size), config.rms_norm_eps
        inter = config.intermediate_size
        inter = int(inter[layer_idx] if isinstance(inter, (list, tuple)) else inter)
        pattern = getattr(config, "activation_sparsity_pattern", None)
        sparsity = float(pattern[layer_idx]) if pattern else 0.0
        self.self_attn = Gemma3nAttention(config, vllm_config, layer_idx, kind, source, f"{prefix}.self_attn")
        self.mlp = Gemma3nMLP(d, inter, sparsity, quant_config, f"{prefix}.mlp")
        self.input_layernorm, self.post_attention_layernorm = RMSNorm(d, eps), RMSNorm(d, eps)
        self.pre_feedforward_layernorm, self.post_feedforward_layernorm = RMSNorm(d, eps), RMSNorm(d, eps)
        self.altup, self.laurel = Gemma3nAltUp(config), Gemma3nLaurel(config)
        p = int(config.hidden_size_per_layer_input)
        self.per_layer_input_gate = nn.Linear(d, p, bias=False)
        self.per_layer_projection = nn.Linear(p, d, bias=False)
        self.post_per_layer_input_norm = RMSNorm(d, eps)
        self.correct_scale = bool(getattr(config, "altup_correct_scale", True))

    def forward(self, positions, h: torch.Tensor, per_layer_input: torch.Tensor) -> torch.Tensor:
        """h [n, T, d] -> [n, T, d]"""
        a = self.altup
        pred = a.predict(h)
        x = pred[a.active]
        xn = self.input_layernorm(x)
        laurel = self.laurel(xn)
        attn = self.post_attention_layernorm(self.self_attn(positions, xn))
        y = (x + attn + laurel) * (1.0 / math.sqrt(2.0))
        y = y + self.post_feedforward_layernorm(self.mlp(self.pre_feedforward_layernorm(y)))
        out = a.correct(pred, y)
        first = out[a.active]
        if self.correct_scale:
            first = first * a.correct_output_scale
        g = F.gelu(self.per_layer_input_gate(first), approximate="tanh") * per_layer_input
        g = self.post_per_layer_input_norm(self.per_layer_projection(g))
        out[1:] += g
        return out


class Gemma3nForConditionalGeneration(nn.Module):
    supports_pp = False
    supports_lora = False
    stacked_params_mapping = [(".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"), (".qkv_proj", ".v_proj", "v"),
                              (".gate_up_proj", ".gate_proj", 0), (".gate_up_proj", ".up_proj", 1)]

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        config = _cfg(hf.text_config) if getattr(hf, "text_config", None) is not None else hf
        self.config = config
        d, L, p = int(config.hidden_size), int(config.num_hidden_layers), int(config.hidden_size_per_layer_input)
        self.n, self.L, self.p = int(config.altup_num_inputs), L, p
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, d)
        self.embed_tokens_per_layer = VocabParallelEmbedding(int(config.vocab_size_per_layer_input), L * p)
        self.vocab_per_layer = int(config.vocab_size_per_layer_input)
        self.per_layer_model_projection = nn.Linear(d, L * p, bias=False)
        self.per_layer_projection_norm = RMSNorm(p, config.rms_norm_eps)
        self.altup_projections = nn.ModuleList(nn.Linear(d, d, bias=False) for _ in range(1, self.n))
        self.altup_unembed_projections = nn.ModuleList(nn.Linear(d, d, bias=False) for _ in range(1, self.n))
        kinds = _layer_types(config)
        first_shared = L - int(getattr(config, "num_kv_shared_layers", 0) or 0)
        layers = []
        for i in range(L):
            source = None
            if 0 < first_shared <= i:        # the last earlier layer of the same kind that owns its keys
                j = max(k for k in range(first_shared) if kinds[k] == kinds[i])
                source = layers[j].self_attn
            layers.append(Gemma3nDecoderLayer(config, vllm_config, i, kinds[i], source, f"model.layers.{i}"))
        self.layers = nn.ModuleList(layers)
        self.norm = RMSNorm(d, config.rms_norm_eps)
        self.lm_head = ParallelLMHead(config.vocab_size, d)
        self.tie = bool(getattr(config, "tie_word_embeddings", True))
        if self.tie:
            self.lm_head.tie_weights(self.embed_tokens)
        self.logits_processor = LogitsProcessor(config.vocab_size,
                                                soft_cap=getattr(config, "final_logit_softcapping", None))

    def get_input_embeddings(self, ids):
        return self.embed_tokens(ids) * (self.config.hidden_size ** 0.5)

    @staticmethod
    def _match_magnitude(x, target):
        new = x.float().pow(2).mean(-1, keepdim=True).clamp_min(1e-5).sqrt()
        return (x.float() * target / new).to(x.dtype)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        h0 = inputs_embeds if inputs_embeds is not None else self.get_input_embeddings(input_ids)
        T = h0.shape[0]
        per_layer = self.per_layer_projection_norm(
            (self.per_layer_model_projection(h0) * (self.config.hidden_size ** -0.5)).view(T, self.L, self.p))
        if input_ids is not None:
            ids = torch.where(input_ids < self.vocab_per_layer, input_ids, 0)
            table = self.embed_tokens_per_layer(ids).view(T, self.L, self.p) * (self.p ** 0.5)
            per_layer = (per_layer + table) * (1.0 / math.sqrt(2.0))
        target = h0.float().pow(2).mean(-1, keepdim=True).sqrt()
        h = torch.stack([h0] + [self._match_magnitude(proj(h0), target) for proj in self.altup_projections])
        for i, layer in enumerate(self.layers):
            h = layer(positions, h, per_layer[:, i])
        target = h[0].float().pow(2).mean(-1, keepdim=True).sqrt()
        outs = [h[0]] + [self._match_magnitude(proj(h[i + 1]), target)
                         for i, proj in enumerate(self.altup_unembed_projections)]
        return self.norm(torch.stack(outs).mean(0))

    def compute_logits(self, hidden):
        return self.logits_processor(self.lm_head, hidden)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        shared = {i for i, layer in enumerate(self.layers) if layer.self_attn.shared}

        def fix(items):
            for name, t in items:
                for head in ("model.language_model.", "language_model.model.", "model."):
                    if name.startswith(head):
                        name = name[len(head):]
                        break
                else:
                    if name.startswith("language_model."):
                        name = name[len("language_model."):]
                if name.startswith(("vision_tower.", "audio_tower.", "embed_vision.", "embed_audio.")):
                    continue                                 # towers of the multimodal checkpoints
                if name.startswith("layers."):
                    i = int(name.split(".")[1])
                    if i in shared and any(k in name for k in (".k_proj.", ".v_proj.", ".k_norm.", ".v_norm.")):
                        continue                             # layers that read another layer's keys keep none
                yield name, t

        def stack(items):
            # shared layers have a plain q_proj, the others a fused qkv_proj
            for name, t in items:
                if name.startswith("layers.") and int(name.split(".")[1]) in shared and ".q_proj." in name:
                    yield name.replace(".q_proj.", ".q_proj_only."), t
                else:
                    yield name, t

        params = dict(self.named_parameters())
        direct, rest = [], []
        for name, t in stack(fix(weights)):
            if ".q_proj_only." in name:
                direct.append((name.replace(".q_proj_only.", ".q_proj."), t))
            else:
                rest.append((name, t))
        loaded = load_weights_with_mapping(self, rest, self.stacked_params_mapping, tie_lm_head=self.tie)
        for name, t in direct:
            p = params[name]
            getattr(p, "weight_loader", lambda p_, t_: p_.data.copy_(t_))(p, t)
            loaded.add(name)
        return loaded
