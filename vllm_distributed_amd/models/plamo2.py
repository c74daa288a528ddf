"""PLaMo 2: a hybrid stack in which most layers mix tokens with a state-space layer and every ``mamba_step``-th one
with attention.  The state-space mixer has one decay and one skip weight per head (like Mamba-2) but computes its
B, C and time-step inputs from the CONVOLVED signal through a projection of its own, each of them RMS-normalised;
gate and signal come interleaved per head out of ``in_proj``; the output is the scan times silu(gate), no norm.
Attention layers RMS-normalise q and k per head with per-head weights.  All four block norms store their weight as an
offset from a constant (1, 1/5, 1, 1/5^1.5), and so does the final norm.

No transformers class exists for this family: parity unpinned; tested against the recurrence written out in fp32.
Reference: vllm/model_executor/models/plamo2.py (is_mamba :91-98, Plamo2MambaMixer :120-320, Plamo2AttentionMixer
:352-436, Plamo2DecoderLayer :439-503, load_weights :687-745).
"""
from __future__ import annotations

from typing import Iterable

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..config import VllmConfig
from ..forward_context import get_forward_context
from ..layers.attention import Attention
from ..layers.basic import LogitsProcessor, ParallelLMHead, RMSNorm, VocabParallelEmbedding
from ..layers.linear import ColumnParallelLinear, MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from ..layers.rotary import get_rope
from ..parallel.state import get_tensor_model_parallel_world_size
from .mamba import StateSpaceLayer
from .utils import load_weights_with_mapping


def is_mamba_layer(config, i: int) -> bool:
    if not getattr(config, "mamba_enabled", True):
        return False
    step = int(config.mamba_step)
    if config.num_hidden_layers <= step // 2:
        return i != config.num_hidden_layers - 1            # short stacks: attention in the last layer only
    return (i % step) != (step // 2)


class Plamo2MambaMixer(StateSpaceLayer):
    def __init__(self, config, prefix: str):
        super().__init__()
        if get_tensor_model_parallel_world_size() > 1:
            raise NotImplementedError("PLaMo 2 state-space layers run with tensor_parallel_size=1")
        d = int(config.hidden_size)
        self.heads, self.hd = int(config.mamba_num_heads), int(config.hidden_size_per_head)
        self.d_inner, self.d_state, self.d_conv = self.heads * self.hd, int(config.mamba_d_state), int(config.mamba_d_conv)
        self.rank = max(64, d // 16)
        dtype = torch.get_default_dtype()
        self.conv_dtype = dtype
        self.in_proj = ColumnParallelLinear(d, 2 * self.d_inner, bias=False, prefix=f"{prefix}.in_proj")
        self.conv_weight = nn.Parameter(torch.zeros(self.d_inner, self.d_conv, dtype=dtype), requires_grad=False)
        self.bcdt_proj = nn.Linear(self.d_inner, 2 * self.d_state + self.rank, bias=False)
        self.dt_proj = nn.Linear(self.rank, self.heads, bias=False)
        self.dt_bias = nn.Parameter(torch.zeros(self.heads, dtype=torch.float32), requires_grad=False)
        self.A = nn.Parameter(torch.zeros(self.heads, dtype=torch.float32), requires_grad=False)
        self.D = nn.Parameter(torch.ones(self.heads, dtype=torch.float32), requires_grad=False)
        eps = config.rms_norm_eps
        self.dt_norm, self.B_norm, self.C_norm = RMSNorm(self.rank, eps), RMSNorm(self.d_state, eps), RMSNorm(self.d_state, eps)
        self.out_proj = RowParallelLinear(self.d_inner, d, bias=False, prefix=f"{prefix}.out_proj")

    def state_shapes(self) -> tuple:
        return ((self.d_inner, self.d_conv - 1), (self.d_inner, self.d_state))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        md = get_forward_context().attn_metadata
        T = x.shape[0]
        p = self.in_proj(x).view(T, self.heads, 2 * self.hd)
        gate, u = p[..., : self.hd].reshape(T, self.d_inner), p[..., self.hd:].reshape(T, self.d_inner)
        if md is None or self.conv_state is None:
            return self.out_proj((u * gate).contiguous())                   # memory-profiling run
        tbl = self.slot_table(md)
        u = ops.causal_conv1d(u.contiguous(), self.conv_weight, None, self.conv_state, md.query_start_loc, md.seq_lens,
                              tbl, md.max_query_len)
        bcdt = self.bcdt_proj(u)
        n = self.d_state
        B, C = self.B_norm(bcdt[:, :n].contiguous()), self.C_norm(bcdt[:, n:2 * n].contiguous())
        dt = self.dt_proj(self.dt_norm(bcdt[:, 2 * n:].contiguous()))
        y = ops.selective_scan(u, dt.contiguous(), self.A, B, C, self.D, gate.contiguous(), self.dt_bias, self.ssm_state,
                               md.query_start_loc, md.seq_lens, tbl, softplus=True, num_groups=1,
                               max_query_len=md.max_query_len)
        return self.out_proj(y)


class Plamo2AttentionMixer(nn.Module):
    def __init__(self, config, vllm_config: VllmConfig, prefix: str):
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        d, hq, hkv = int(config.hidden_size), int(config.num_attention_heads), int(config.num_key_value_heads)
        self.hd = int(config.hidden_size_per_head)
        self.num_heads, self.num_kv_heads = hq // tp, max(1, hkv // tp)
        self.q_size, self.kv_size = self.num_heads * self.hd, self.num_kv_heads * self.hd
        quant_config = getattr(vllm_config, "quant_config", None)
        self.qkv_proj = QKVParallelLinear(d, self.hd, hq, hkv, bias=False, quant_config=quant_config,
                                          prefix=f"{prefix}.qkv_proj")
        self.o_proj = RowParallelLinear(hq * self.hd, d, bias=False, quant_config=quant_config, prefix=f"{prefix}.o_proj")
        self.q_weight = nn.Parameter(torch.ones(self.num_heads, self.hd))
        self.k_weight = nn.Parameter(torch.ones(self.num_kv_heads, self.hd))
        self.rotary_emb = get_rope(self.hd, self.hd, vllm_config.model_config.max_model_len,
                                   float(getattr(config, "rope_theta", 10000.0)), True,
                                   getattr(config, "rope_scaling", None), 1.0)
        self.attn = Attention(self.num_heads, self.hd, self.hd ** -0.5, self.num_kv_heads, prefix=f"{prefix}.attn",
                              cache_dtype=vllm_config.cache_config.cache_dtype)

    @staticmethod
    def _norm(x, w):                                      # x [T, heads * hd], w [heads, hd]
        f = x.float().view(x.shape[0], *w.shape)
        f = f * torch.rsqrt(f.pow(2).mean(-1, keepdim=True) + 1e-6)
        return (f.to(x.dtype) * w).view(x.shape)

    def forward(self, positions, x):
        qkv = self.qkv_proj(x)
        qs, ks = self.q_size, self.kv_size
        qkv[:, :qs] = self._norm(qkv[:, :qs], self.q_weight)
        qkv[:, qs:qs + ks] = self._norm(qkv[:, qs:qs + ks], self.k_weight)
        return self.o_proj(self.attn(qkv, positions, self.rotary_emb))


class Plamo2DecoderLayer(nn.Module):
    def __init__(self, config, vllm_config: VllmConfig, idx: int, prefix: str):
        super().__init__()
        quant_config = getattr(vllm_config, "quant_config", None)
        d, eps = int(config.hidden_size), config.rms_norm_eps
        self.is_mamba = is_mamba_layer(config, idx)
        self.mixer = Plamo2MambaMixer(config, f"{prefix}.mixer") if self.is_mamba else \
            Plamo2AttentionMixer(config, vllm_config, f"{prefix}.mixer")
        self.mlp = nn.Module()
        inter = int(config.intermediate_size)
        self.mlp.gate_up_proj = MergedColumnParallelLinear(d, [inter, inter], bias=False, quant_config=quant_config,
                                                           prefix=f"{prefix}.mlp.gate_up_proj")
        self.mlp.down_proj = RowParallelLinear(inter, d, bias=False, quant_config=quant_config,
                                               prefix=f"{prefix}.mlp.down_proj")
        self.pre_mixer_norm, self.post_mixer_norm = RMSNorm(d, eps), RMSNorm(d, eps)
        self.pre_mlp_norm, self.post_mlp_norm = RMSNorm(d, eps), RMSNorm(d, eps)

    def forward(self, positions, x, residual):
        if residual is None:
            residual = x
            x = self.pre_mixer_norm(x)
        else:
            x, residual = self.pre_mixer_norm(x, residual)
        x = self.mixer(x) if self.is_mamba else self.mixer(positions, x)
        x, residual = self.pre_mlp_norm(self.post_mixer_norm(x), residual)
        gate, up = self.mlp.gate_up_proj(x).chunk(2, -1)
        return self.post_mlp_norm(self.mlp.down_proj(F.silu(gate) * up)), residual


# constants the checkpoint's norm weights are offsets from
_NORM_OFFSETS = {".pre_mixer_norm.": 1.0, ".post_mixer_norm.": 1.0 / 5, ".pre_mlp_norm.": 1.0,
                 ".post_mlp_norm.": 1.0 / 5 ** 1.5}


class Plamo2ForCausalLM(nn.Module):
    supports_pp = False
    supports_lora = False

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        self.model = nn.Module()
        self.model.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.model.layers = nn.Module()
        self.model.layers.layers = nn.ModuleList(
            Plamo2DecoderLayer(config, vllm_config, i, f"model.layers.layers.{i}") for i in range(config.num_hidden_layers))
        self.model.norm = RMSNorm(config.hidden_size, config.rms_norm_eps)
        self.tie = bool(getattr(config, "tie_word_embeddings", True))
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if self.tie:
            self.lm_head.tie_weights(self.model.embed_tokens)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    def get_input_embeddings(self, ids):
        return self.model.embed_tokens(ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        x = inputs_embeds if inputs_embeds is not None else self.model.embed_tokens(input_ids)
        residual = None
        for layer in self.model.layers.layers:
            x, residual = layer(positions, x, residual)
        return self.model.norm(x, residual)[0]

    def compute_logits(self, hidden):
        return self.logits_processor(self.lm_head, hidden)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        names = {".mixer.A_log": ".mixer.A", ".mixer.B_norm_weight": ".mixer.B_norm.weight",
                 ".mixer.C_norm_weight": ".mixer.C_norm.weight", ".mixer.dt_norm_weight": ".mixer.dt_norm.weight",
                 ".mixer.conv1d.weight": ".mixer.conv_weight"}

        def fix(items):
            for n, t in items:
                for a, b in names.items():
                    if n.endswith(a):
                        n = n[: -len(a)] + b
                if n.endswith(".mixer.A"):
                    t = -torch.exp(t.float())
                elif n.endswith(".mixer.conv_weight"):
                    t = t.reshape(t.shape[0], t.shape[-1])
                elif n.endswith((".mixer.D", ".mixer.dt_bias")):
                    t = t.float()
                elif n == "model.norm.weight":
                    t = t + 1.0
                else:
                    for frag, off in _NORM_OFFSETS.items():
                        if frag in n:
                            t = t + off
                yield n, t

        return load_weights_with_mapping(self, fix(weights), [], tie_lm_head=self.tie)
