"""MiniMax-Text-01 / MiniMax-M1: a stack of "lightning" (linear) attention layers with a softmax-attention layer
every few layers, every MLP a routed expert block, DeepNorm-style residual factors.

Lightning attention keeps ONE matrix per head and request, S = sum_j decay^(t-j) k_j^T v_j, instead of keys and
values: a decode step is S <- decay * S + k^T v, out = q S; a prompt runs block-wise (inside a block the decayed
(q k^T) v product, across blocks q S).  The state lives in the runner's per-request state slots (the mechanism of
the state-space layers), the softmax layers in the paged KV cache.
Reference: vllm/model_executor/models/minimax_text_01.py (MiniMaxText01LinearAttention :330-510,
MiniMaxText01DecoderLayer :586-775, weight names :1100-1200), vllm/model_executor/layers/lightning_attn.py.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch
import torch.nn as nn

from .. import ops
from ..config import VllmConfig
from ..forward_context import get_forward_context
from ..layers.basic import LogitsProcessor, ParallelLMHead, RMSNorm, VocabParallelEmbedding
from ..layers.linear import ColumnParallelLinear, RowParallelLinear
from ..layers.moe import FusedMoE
from ..parallel.state import (get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                              tensor_model_parallel_all_reduce)
from .llama import LlamaAttention, LlamaMLP
from .mamba import StateSpaceLayer
from .mixtral import SparseMoeBlock
from .utils import load_weights_with_mapping

BLOCK = 256


class LightningAttention(StateSpaceLayer):
    """qkv = act(W x) split per head into (q, k, v); out = W_o (sigmoid(W_g x) * RMSNorm(lightning(q, k, v)))."""
    state_dtypes = (torch.float32, torch.float32)

    def __init__(self, config, layer_idx: int, prefix: str):
        super().__init__()
        from ..layers.basic import get_act_fn
        tp, rank = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        d, H = config.hidden_size, config.num_attention_heads
        self.D = int(getattr(config, "head_dim", None) or d // H)
        self.H = H // tp
        self.tp = tp
        self.layer_name = prefix
        self.conv_dtype = torch.float32
        # rows of the fused projection are head-major ([q | k | v] of head 0, then head 1 ...): a column split
        # over ranks is a split over heads
        self.qkv_proj = ColumnParallelLinear(d, 3 * H * self.D, bias=False, prefix=f"{prefix}.qkv_proj")
        self.output_gate = ColumnParallelLinear(d, H * self.D, bias=False, prefix=f"{prefix}.output_gate")
        self.out_proj = RowParallelLinear(H * self.D, d, bias=False, prefix=f"{prefix}.out_proj")
        self.norm = RMSNorm(H * self.D, float(getattr(config, "rms_norm_eps", 1e-5)))
        if tp > 1:                                        # this rank's slice of the full-width norm weight
            w = self.norm.weight
            w.data = w.data[: self.H * self.D].clone()
            full, r = H * self.D, rank

            def load(param, t, *_):
                param.data.copy_(t[r * (full // tp):(r + 1) * (full // tp)])
            w.weight_loader = load
        self.full_width = H * self.D
        self.act = get_act_fn(getattr(config, "hidden_act", "silu"))
        n = int(config.num_hidden_layers)
        base = 1.0 / (2.0 ** (8.0 / H))
        slope = base ** (torch.arange(H, dtype=torch.float32) + 1) * (1 - layer_idx / (n - 1 + 1e-5) + 1e-5)
        self.register_buffer("slope", slope[rank * self.H:(rank + 1) * self.H].clone(), persistent=False)

    def state_shapes(self) -> tuple:
        return ((1, 1), (self.H, self.D, self.D))

    def _normed(self, o: torch.Tensor) -> torch.Tensor:
        if self.tp == 1:
            return self.norm(o)
        f = o.float()
        ss = tensor_model_parallel_all_reduce(f.pow(2).sum(-1, keepdim=True))
        return (f * torch.rsqrt(ss / self.full_width + self.norm.eps)).to(o.dtype) * self.norm.weight

    def _prompt(self, q, k, v, state: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """One sequence's new tokens, [H, n, D] each in fp32, against its state [H, D, D] -> (out [H, n, D], state)."""
        n = q.shape[1]
        s = self.slope.view(-1, 1, 1)
        outs = []
        for a in range(0, n, BLOCK):
            b = min(a + BLOCK, n)
            m = b - a
            idx = torch.arange(m, device=q.device, dtype=torch.float32)
            qb, kb, vb = q[:, a:b], k[:, a:b], v[:, a:b]
            diff = idx[:, None] - idx[None, :]
            w = torch.where(diff >= 0, torch.exp(-s * diff), torch.zeros((), device=q.device))
            intra = torch.matmul(torch.matmul(qb, kb.transpose(1, 2)) * w, vb)
            inter = torch.matmul(qb * torch.exp(-s[:, 0] * (idx + 1))[..., None], state)
            outs.append(intra + inter)
            kd = kb * torch.exp(-s[:, 0] * (m - 1 - idx))[..., None]
            state = state * torch.exp(-s * m) + torch.matmul(kd.transpose(1, 2), vb)
        return torch.cat(outs, dim=1), state

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        md = get_forward_context().attn_metadata
        T = x.shape[0]
        H, D = self.H, self.D
        qkv = self.act(self.qkv_proj(x))
        gate = self.output_gate(x)
        if md is None or self.ssm_state is None:          # memory-profiling run
            return self.out_proj(torch.sigmoid(gate) * qkv[:, : H * D])
        q, k, v = qkv.view(T, H, 3 * D).float().split(D, dim=-1)
        out = torch.zeros((T, H, D), dtype=torch.float32, device=x.device)
        tbl = self.slot_table(md)
        nd, ndt, S = md.num_decode_seqs, md.num_decode_tokens, md.num_decode_seqs + md.num_prefill_seqs
        state = self.ssm_state
        if nd and ndt == nd:
            # one token per sequence: S <- decay * S + k^T v (a sequence at its first token starts from zero)
            out[:nd] = ops.lightning_attention_decode(qkv[:nd], state, self.slope, md.seq_lens[:nd], tbl[:nd], H, D)
            first = nd
        else:
            first = 0
        if first < S:
            cu = md.query_start_loc[:S + 1].tolist()
            lens = md.seq_lens[:S].tolist()
            slots = tbl[:S, 0].tolist()
            for i in range(first, S):
                a, b = cu[i], cu[i + 1]
                if b <= a:
                    continue
                s0 = state[slots[i]] if lens[i] - (b - a) > 0 else torch.zeros_like(state[slots[i]])
                o, s1 = self._prompt(q[a:b].transpose(0, 1), k[a:b].transpose(0, 1), v[a:b].transpose(0, 1), s0)
                out[a:b] = o.transpose(0, 1)
                state[slots[i]] = s1
        o = self._normed(out.view(T, H * D).to(x.dtype))
        return self.out_proj(torch.sigmoid(gate) * o)


class MiniMaxDecoderLayer(nn.Module):
    """h = norm(x); x = alpha * (h or x) + beta * mixer(h); the same around the expert block.  ``postnorm``
    (the released checkpoints, and transformers' form) keeps the NORMED input as the residual."""

    def __init__(self, config, vllm_config: VllmConfig, idx: int, linear: bool, prefix: str):
        super().__init__()
        mc = vllm_config.model_config
        quant_config = getattr(vllm_config, "quant_config", None)
        g = lambda *names, default=1.0: next((float(getattr(config, n)) for n in names if getattr(config, n, None)
                                              is not None), default)
        eps = float(getattr(config, "rms_norm_eps", 1e-5))
        self.linear = linear
        if linear:
            self.self_attn = LightningAttention(config, idx, f"{prefix}.self_attn")
            self.a_alpha = g("layernorm_linear_attention_alpha", "linear_attn_alpha_factor")
            self.a_beta = g("layernorm_linear_attention_beta", "linear_attn_beta_factor")
        else:
            rp = getattr(config, "rope_parameters", None) or {}
            theta = float(getattr(config, "rope_theta", None) or rp.get("rope_theta") or 10000.0)
            self.self_attn = LlamaAttention(config, config.hidden_size, config.num_attention_heads,
                                            config.num_key_value_heads, theta, None, mc.max_model_len,
                                            quant_config=quant_config, prefix=f"{prefix}.self_attn",
                                            cache_dtype=vllm_config.cache_config.cache_dtype)
            rd = getattr(config, "rotary_dim", None)
            if rd and int(rd) != self.self_attn.head_dim:
                from ..layers.rotary import get_rope
                self.self_attn.rotary_emb = get_rope(self.self_attn.head_dim, int(rd), mc.max_model_len, theta, True)
            self.a_alpha = g("layernorm_full_attention_alpha", "full_attn_alpha_factor")
            self.a_beta = g("layernorm_full_attention_beta", "full_attn_beta_factor")
        self.m_alpha = g("layernorm_mlp_alpha", "mlp_alpha_factor")
        self.m_beta = g("layernorm_mlp_beta", "mlp_beta_factor")
        self.postnorm = bool(getattr(config, "postnorm", True))
        self.input_layernorm = RMSNorm(config.hidden_size, eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps)
        self.block_sparse_moe = SparseMoeBlock(config.hidden_size, config.intermediate_size,
                                               int(config.num_local_experts), int(config.num_experts_per_tok), True,
                                               quant_config, f"{prefix}.block_sparse_moe")
        shared = getattr(config, "shared_intermediate_size", 0) or 0
        if isinstance(shared, (list, tuple)):
            shared = shared[idx] if idx < len(shared) else 0
        if shared:
            raise ValueError("MiniMax checkpoints with a shared expert beside the routed ones are not supported")

    def forward(self, positions, x):
        h = self.input_layernorm(x)
        a = self.self_attn(h) if self.linear else self.self_attn(positions, h)
        x = (h if self.postnorm else x) * self.a_alpha + a * self.a_beta
        h = self.post_attention_layernorm(x)
        return (h if self.postnorm else x) * self.m_alpha + self.block_sparse_moe(h) * self.m_beta


class MiniMaxForCausalLM(nn.Module):
    supports_pp = False
    supports_lora = False
    stacked_params_mapping = [(".qkv_proj", ".q_proj", "q"), (".qkv_proj", ".k_proj", "k"), (".qkv_proj", ".v_proj", "v")]

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        kinds = getattr(config, "layer_types", None)
        if kinds:
            linear = [k == "linear_attention" for k in kinds]
        else:                                             # the original configuration: 0 = lightning, 1 = softmax
            linear = [int(t) == 0 for t in config.attn_type_list]
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([MiniMaxDecoderLayer(config, vllm_config, i, linear[i], f"model.layers.{i}")
                                     for i in range(config.num_hidden_layers)])
        self.norm = RMSNorm(config.hidden_size, float(getattr(config, "rms_norm_eps", 1e-5)))
        self.tie = bool(getattr(config, "tie_word_embeddings", False))
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if self.tie:
            self.lm_head.tie_weights(self.embed_tokens)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    def get_input_embeddings(self, input_ids):
        return self.embed_tokens(input_ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        x = inputs_embeds if inputs_embeds is not None else self.embed_tokens(input_ids)
        for layer in self.layers:
            x = layer(positions, x)
        return self.norm(x)

    def compute_logits(self, hidden_states):
        return self.logits_processor(self.lm_head, hidden_states)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]):
        E = int(self.config.num_local_experts)
        experts = FusedMoE.make_expert_params_mapping("w1", "w2", "w3", E)

        def fix(items):
            for name, t in items:
                name = name.removeprefix("model.")
                if name.endswith(("_decay", ".slope_rate")):
                    continue                                  # decay tables transformers saves as buffers
                name = name.replace(".mlp.gate.", ".block_sparse_moe.gate.").replace(".mlp.experts.", ".block_sparse_moe.experts.")
                # transformers' stacked experts: [E, 2I, H] (gate rows, then up rows) and [E, H, I]
                if name.endswith(".block_sparse_moe.experts.gate_up_proj"):
                    base, half = name[: -len("gate_up_proj")], t.shape[1] // 2
                    for e in range(t.shape[0]):
                        yield f"{base}{e}.w1.weight", t[e, :half]
                        yield f"{base}{e}.w3.weight", t[e, half:]
                    continue
                if name.endswith(".block_sparse_moe.experts.down_proj"):
                    base = name[: -len("down_proj")]
                    for e in range(t.shape[0]):
                        yield f"{base}{e}.w2.weight", t[e]
                    continue
                yield name, t
        return load_weights_with_mapping(self, fix(weights), self.stacked_params_mapping, tie_lm_head=self.tie,
                                         expert_mapping=experts)


class MiniMaxText01ForCausalLM(MiniMaxForCausalLM):
    """The original checkpoint's architecture name (``attn_type_list``, ``block_sparse_moe.experts.N.w1/w2/w3``)."""


class MiniMaxM1ForCausalLM(MiniMaxForCausalLM):
    """MiniMax-M1: the same network, other sizes."""
