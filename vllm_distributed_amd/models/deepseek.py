"""DeepSeek-V2 / V3 decoders: multi-head latent attention (low-rank Q and joint KV projections,
decoupled rotary key) and fine-grained MoE with shared experts and group-limited routing.

Attention here takes the expanded-head route: per token the latent is up-projected to per-head
keys and values (k = [k_nope | k_rope], v), zero-padded to a head size the MFMA paged-attention
kernels support, and cached like any other model's K/V.  This is the behaviour the reference has
with ``VLLM_MLA_DISABLE=1``; a latent-cache decode kernel would shrink the cache by
``2*H*Dp / (kv_lora_rank + rope_dim)`` and is the natural next step.
Reference: vllm/model_executor/models/deepseek_v2.py (DeepseekV2MoE :101-186, attention :199-325,
decoder layer :515-604, load_weights :741-857), deepseek_v3 shares the file; rope:
vllm/model_executor/layers/rotary_embedding.py (DeepseekScalingRotaryEmbedding).
"""
from __future__ import annotations

from typing import Iterable

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..config import VllmConfig
from .. import envs, ops
from ..layers.attention import Attention, MLAAttention
from ..layers.basic import RMSNorm
from ..layers.linear import ColumnParallelLinear, ReplicatedLinear, RowParallelLinear
from ..layers.moe import FusedMoE
from ..layers.rotary import get_rope, yarn_mscale
from ..parallel.state import get_tensor_model_parallel_world_size, tensor_model_parallel_all_reduce
from .llama import LlamaForCausalLM, LlamaMLP, LlamaModel
from .utils import load_weights_with_mapping


def _padded_head_size(n: int) -> int:
    for d in (64, 128, 256):
        if n <= d:
            return d
    raise ValueError(f"head size {n} exceeds what the attention kernels support (256)")


class DeepseekAttention(nn.Module):
    def __init__(self, config, vllm_config: VllmConfig, quant_config=None, prefix: str = ""):
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        H = config.num_attention_heads
        self.num_heads = H // tp
        self.nope, self.rope_dim, self.v_dim = config.qk_nope_head_dim, config.qk_rope_head_dim, config.v_head_dim
        self.qk_dim = self.nope + self.rope_dim
        self.kv_lora_rank = config.kv_lora_rank
        self.q_lora_rank = getattr(config, "q_lora_rank", None)
        hid = config.hidden_size
        eps = getattr(config, "rms_norm_eps", 1e-6)
        bias = bool(getattr(config, "attention_bias", False))
        if self.q_lora_rank:
            self.q_a_proj = ReplicatedLinear(hid, self.q_lora_rank, bias=bias, quant_config=quant_config,
                                             prefix=f"{prefix}.q_a_proj")
            self.q_a_layernorm = RMSNorm(self.q_lora_rank, eps)
            self.q_b_proj = ColumnParallelLinear(self.q_lora_rank, H * self.qk_dim, bias=False,
                                                 quant_config=quant_config, prefix=f"{prefix}.q_b_proj")
        else:
            self.q_proj = ColumnParallelLinear(hid, H * self.qk_dim, bias=False, quant_config=quant_config,
                                               prefix=f"{prefix}.q_proj")
        self.kv_a_proj_with_mqa = ReplicatedLinear(hid, self.kv_lora_rank + self.rope_dim, bias=bias,
                                                   quant_config=quant_config, prefix=f"{prefix}.kv_a_proj_with_mqa")
        self.kv_a_layernorm = RMSNorm(self.kv_lora_rank, eps)
        self.kv_b_proj = ColumnParallelLinear(self.kv_lora_rank, H * (self.nope + self.v_dim), bias=False,
                                              quant_config=quant_config, prefix=f"{prefix}.kv_b_proj")
        self.o_proj = RowParallelLinear(H * self.v_dim, hid, bias=bias, quant_config=quant_config,
                                        prefix=f"{prefix}.o_proj")
        rs = getattr(config, "rope_scaling", None) or getattr(config, "rope_parameters", None) or {}
        theta = getattr(config, "rope_theta", None) or rs.get("rope_theta") or 10000.0
        kind = rs.get("rope_type", rs.get("type"))
        scale = self.qk_dim ** -0.5
        rope_cfg = None
        if kind not in (None, "default"):
            rope_cfg = dict(rs)
            rope_cfg["rope_type"] = "yarn" if kind == "deepseek_yarn" else kind
            factor = float(rs.get("factor", 1.0))
            if rs.get("mscale_all_dim"):
                m = yarn_mscale(factor, float(rs["mscale_all_dim"]))
                scale = scale * m * m
        # the rotary pairs are interleaved (x0,x1),(x2,x3),... in DeepSeek checkpoints, split in halves in MiniCPM3
        neox = getattr(config, "model_type", "") == "minicpm3"
        self.rotary_emb = get_rope(self.rope_dim, self.rope_dim, vllm_config.model_config.max_model_len, theta, neox,
                                   rope_cfg)
        # Latent cache + absorbed projections when the kernel is built for these sizes and kv_b_proj keeps
        # 16-bit weights (its two halves become W_UK and W_UV); expanded, padded heads otherwise.
        on_dev = vllm_config.device_config.device != "cpu"
        self.use_mla = bool(envs.VDA_MLA and (quant_config is None or getattr(quant_config, "mla_compatible", False))
                            and vllm_config.cache_config.cache_dtype in ("auto", "fp8")
                            and vllm_config.kv_transfer_config is None
                            and ops.mla_supported(self.kv_lora_rank + self.rope_dim, self.kv_lora_rank,
                                                  vllm_config.model_config.dtype, on_dev))
        if self.use_mla:
            self.attn = MLAAttention(self.num_heads, self.kv_lora_rank, self.rope_dim, scale, prefix=f"{prefix}.attn")
            self._w_uk = self._w_uv = None
            return
        self.pad = _padded_head_size(max(self.qk_dim, self.v_dim))
        self.attn = Attention(self.num_heads, self.pad, scale, self.num_heads, prefix=f"{prefix}.attn",
                              cache_dtype=vllm_config.cache_config.cache_dtype)

    def _absorbed_weights(self):
        """kv_b_proj's weight [H * (nope + v), R] seen as W_UK [H, nope, R] and W_UV^T [H, R, v]."""
        w = self.kv_b_proj.weight
        if self._w_uk is None or self._w_uk.data_ptr() != w.data_ptr():
            w3 = w.view(self.num_heads, self.nope + self.v_dim, self.kv_lora_rank)
            self._w_uk = w3[:, : self.nope, :]                                  # view: shares kv_b_proj's storage
            self._w_uv_t = w3[:, self.nope:, :].transpose(1, 2).contiguous()    # [H, R, v]
        return self._w_uk, self._w_uv_t

    def _forward_mla(self, positions: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        T, H = x.shape[0], self.num_heads
        if self.q_lora_rank:
            q = self.q_b_proj(self.q_a_layernorm(self.q_a_proj(x)))
        else:
            q = self.q_proj(x)
        q = q.view(T, H, self.qk_dim)
        latent = self.kv_a_proj_with_mqa(x)
        kv_c = self.kv_a_layernorm(latent[:, : self.kv_lora_rank].contiguous())
        k_pe = latent[:, self.kv_lora_rank:].contiguous()
        q_pe = q[:, :, self.nope:].contiguous().view(T, H * self.rope_dim)
        n = min(positions.shape[0], T)
        self.rotary_emb(positions[:n], q_pe[:n], k_pe[:n])
        w_uk, w_uv_t = self._absorbed_weights()
        qf = torch.empty((T, H, self.kv_lora_rank + self.rope_dim), dtype=q.dtype, device=q.device)
        # q_abs[t, h] = q_nope[t, h] . W_UK[h]: one batched GEMM over the heads
        torch.bmm(q[:, :, : self.nope].transpose(0, 1), w_uk, out=qf[:, :, : self.kv_lora_rank].transpose(0, 1))
        qf[:, :, self.kv_lora_rank:] = q_pe.view(T, H, self.rope_dim)
        o_lat = self.attn(qf, kv_c, k_pe)                                        # [T, H, R]
        o = torch.bmm(o_lat.transpose(0, 1), w_uv_t).transpose(0, 1)             # [T, H, v]
        return self.o_proj(o.reshape(T, H * self.v_dim))

    def forward(self, positions: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        if self.use_mla:
            return self._forward_mla(positions, x)
        T, H, P = x.shape[0], self.num_heads, self.pad
        if self.q_lora_rank:
            q = self.q_b_proj(self.q_a_layernorm(self.q_a_proj(x)))
        else:
            q = self.q_proj(x)
        q = q.view(T, H, self.qk_dim)
        latent = self.kv_a_proj_with_mqa(x)
        kv_c, k_pe = latent[:, : self.kv_lora_rank], latent[:, self.kv_lora_rank:].contiguous()
        kv = self.kv_b_proj(self.kv_a_layernorm(kv_c.contiguous())).view(T, H, self.nope + self.v_dim)
        q_pe = q[:, :, self.nope:].contiguous().view(T, H * self.rope_dim)
        n = min(positions.shape[0], T)
        self.rotary_emb(positions[:n], q_pe[:n], k_pe[:n])
        # [q | k | v], every head zero-padded to the kernel's head size
        qkv = x.new_zeros((T, 3, H, P))
        qkv[:, 0, :, : self.nope] = q[:, :, : self.nope]
        qkv[:, 0, :, self.nope: self.qk_dim] = q_pe.view(T, H, self.rope_dim)
        qkv[:, 1, :, : self.nope] = kv[:, :, : self.nope]
        qkv[:, 1, :, self.nope: self.qk_dim] = k_pe.unsqueeze(1)
        qkv[:, 2, :, : self.v_dim] = kv[:, :, self.nope:]
        o = self.attn(qkv.view(T, 3 * H * P)).view(T, H, P)[:, :, : self.v_dim]
        return self.o_proj(o.reshape(T, H * self.v_dim))


class DeepseekMoE(nn.Module):
    def __init__(self, config, quant_config=None, prefix: str = ""):
        super().__init__()
        E = int(config.n_routed_experts)
        self.tp = get_tensor_model_parallel_world_size()
        self.gate = ReplicatedLinear(config.hidden_size, E, bias=False, quant_config=None, prefix=f"{prefix}.gate")
        method = getattr(config, "topk_method", None) or \
            ("noaux_tc" if getattr(config, "model_type", "") == "deepseek_v3" else "greedy")
        scoring = getattr(config, "scoring_func", "softmax")
        bias = None
        if method == "noaux_tc":
            self.gate.e_score_correction_bias = nn.Parameter(torch.zeros(E, dtype=torch.float32),
                                                             requires_grad=False)
            bias = self.gate.e_score_correction_bias
            scoring = "sigmoid"
        grouped = method in ("group_limited_greedy", "noaux_tc")
        n_shared = int(getattr(config, "n_shared_experts", 0) or 0)
        self.experts = FusedMoE(E, config.num_experts_per_tok, config.hidden_size, config.moe_intermediate_size,
                                reduce_results=n_shared == 0, renormalize=bool(getattr(config, "norm_topk_prob", False)),
                                use_grouped_topk=grouped, num_expert_group=getattr(config, "n_group", None),
                                topk_group=getattr(config, "topk_group", None), quant_config=quant_config,
                                prefix=f"{prefix}.experts", scoring_func=scoring, e_score_correction_bias=bias,
                                routed_scaling_factor=float(getattr(config, "routed_scaling_factor", 1.0) or 1.0))
        self.shared_experts = LlamaMLP(config.hidden_size, config.moe_intermediate_size * n_shared,
                                       quant_config=quant_config, prefix=f"{prefix}.shared_experts",
                                       reduce_results=False) if n_shared else None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # the router runs in fp32: with hundreds of experts bf16 logits reorder the top-k
        logits = F.linear(x.float(), self.gate.weight.float())
        out = self.experts(x, logits)
        if self.shared_experts is not None:
            out = out + self.shared_experts(x)
            if self.tp > 1 or self.experts.ep_size > 1:
                out = tensor_model_parallel_all_reduce(out)
        return out


class DeepseekDecoderLayer(nn.Module):
    norm_cls = RMSNorm
    make_norm = classmethod(lambda cls, config: cls.norm_cls(config.hidden_size, getattr(config, "rms_norm_eps", 1e-6)))
    def __init__(self, config, vllm_config: VllmConfig, quant_config=None, prefix: str = ""):
        super().__init__()
        idx = int(prefix.split(".")[-1])
        self.self_attn = DeepseekAttention(config, vllm_config, quant_config, f"{prefix}.self_attn")
        moe = (getattr(config, "n_routed_experts", None) and idx >= getattr(config, "first_k_dense_replace", 0)
               and idx % (getattr(config, "moe_layer_freq", 1) or 1) == 0)
        self.mlp = DeepseekMoE(config, quant_config, f"{prefix}.mlp") if moe else \
            LlamaMLP(config.hidden_size, config.intermediate_size, False, quant_config, f"{prefix}.mlp")
        eps = getattr(config, "rms_norm_eps", 1e-6)
        self.input_layernorm = RMSNorm(config.hidden_size, eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps)

    def forward(self, positions, hidden_states, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.mlp(hidden_states), residual


class DeepseekModel(LlamaModel):
    layer_cls = DeepseekDecoderLayer


class DeepseekV2ForCausalLM(LlamaForCausalLM):
    model_cls = DeepseekModel
    stacked_params_mapping = [(".gate_up_proj", ".gate_proj", 0), (".gate_up_proj", ".up_proj", 1)]
    packed_modules_mapping = {"gate_up_proj": ["gate_proj", "up_proj"]}
    supports_lora = False

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]) -> set[str]:
        n_layers = self.config.num_hidden_layers
        experts = FusedMoE.make_expert_params_mapping("gate_proj", "down_proj", "up_proj",
                                                      int(getattr(self.config, "n_routed_experts", 0) or 0))

        def rename(name: str):
            # multi-token-prediction heads sit after the last decoder layer: not used for serving
            if name.startswith("model.layers."):
                if int(name.split(".")[2]) >= n_layers:
                    return None
            return name

        return load_weights_with_mapping(self, weights, self.stacked_params_mapping, tie_lm_head=self.tie,
                                         expert_mapping=experts, rename=rename)


class DeepseekV3ForCausalLM(DeepseekV2ForCausalLM):
    pass
