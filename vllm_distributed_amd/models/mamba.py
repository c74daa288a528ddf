"""Attention-free state-space models: Mamba, Falcon-Mamba and Mamba-2.

Every request owns one state slot per layer (the last conv inputs and the SSM state); the
slot index arrives in column 0 of the request's block table, so the ordinary KV-cache
manager, scheduler (chunked prefill, preemption by recompute) and decode graphs serve
these models unchanged.  The two recurrent ops are HIP kernels over the ragged batch
(ops/csrc/mamba.hip); everything else is the shared TP linear layers.

Reference: vllm/model_executor/models/mamba.py, mamba2.py, falcon_mamba (through mamba.py),
vllm/model_executor/layers/mamba/mamba_mixer.py:27-250, mamba_mixer2.py:218-600,
csrc/mamba/mamba_ssm/selective_scan_fwd.cu, csrc/mamba/causal_conv1d/causal_conv1d.cu.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch
import torch.nn as nn

from .. import ops
from ..config import VllmConfig, get_current_vllm_config
from ..forward_context import get_forward_context
from ..layers.basic import LogitsProcessor, ParallelLMHead, RMSNorm, VocabParallelEmbedding
from ..layers.linear import (ColumnParallelLinear, MergedColumnParallelLinear, RowParallelLinear,
                             default_weight_loader, set_weight_attrs)
from ..parallel.state import (get_pp_group, get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size)
from ..sched.kv_cache_interface import KVCacheSpec, MambaSpec
from .utils import IntermediateTensors, is_pp_missing_parameter, make_layers, maybe_prefix


def _sharded_param(shape, dtype, tp_rank: int, transform=None) -> nn.Parameter:
    """Parameter sharded along dim 0 over the TP ranks; ``transform`` maps the checkpoint tensor first."""
    p = nn.Parameter(torch.zeros(shape, dtype=dtype), requires_grad=False)

    def loader(param, loaded, *_):
        if transform is not None:
            loaded = transform(loaded)
        n = param.shape[0]
        default_weight_loader(param, loaded.narrow(0, tp_rank * n, n).reshape(param.shape))
    set_weight_attrs(p, {"weight_loader": loader})
    return p


class StateSpaceLayer(nn.Module):
    """Common part of the mixers: state slots bound by the model runner."""
    is_state_space_layer = True

    def __init__(self):
        super().__init__()
        self.conv_state: Optional[torch.Tensor] = None
        self.ssm_state: Optional[torch.Tensor] = None
        self.layer_name = ""
        self.max_model_len = get_current_vllm_config().model_config.max_model_len

    def state_shapes(self) -> tuple:
        raise NotImplementedError

    def get_kv_cache_spec(self, block_size: int, dtype: torch.dtype) -> KVCacheSpec:
        # one "block" holds the whole request, whatever its length
        return MambaSpec(block_size=self.max_model_len, shapes=self.state_shapes(),
                         dtypes=(self.conv_dtype, torch.float32))

    def bind_kv_cache(self, conv_state: torch.Tensor, ssm_state: torch.Tensor) -> None:
        self.conv_state, self.ssm_state = conv_state, ssm_state

    @staticmethod
    def slot_table(md) -> torch.Tensor:
        """[S, >=1] int32 whose column 0 is each sequence's state slot."""
        return md.state_table if md.state_table is not None else md.block_table


class MambaMixer(StateSpaceLayer):
    """Mamba-1 block: in_proj -> causal conv -> (dt, B, C) projections -> selective scan -> out_proj.
    TP shards the inner channels; x_proj is row-parallel (one small all-reduce per layer)."""

    def __init__(self, config, prefix: str, rms_eps: Optional[float] = None, learned_norms: bool = False):
        super().__init__()
        tp, rank = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        self.learned_norms = learned_norms      # Jamba: RMSNorm with learned scales on dt, B and C
        if learned_norms:
            self.dt_layernorm = RMSNorm(int(config.time_step_rank), rms_eps or 1e-6)
            self.b_layernorm = RMSNorm(int(config.state_size), rms_eps or 1e-6)
            self.c_layernorm = RMSNorm(int(config.state_size), rms_eps or 1e-6)
        self.hidden = config.hidden_size
        self.d_inner = int(config.intermediate_size)
        self.d_state = int(config.state_size)
        self.d_conv = int(config.conv_kernel)
        self.dt_rank = int(config.time_step_rank)
        self.di = self.d_inner // tp
        dtype = torch.get_default_dtype()
        self.conv_dtype = dtype
        self.rms_eps = rms_eps          # Falcon-Mamba normalises dt, B and C (no learned scale)
        use_bias = bool(getattr(config, "use_bias", False))
        self.in_proj = MergedColumnParallelLinear(self.hidden, [self.d_inner, self.d_inner], bias=use_bias,
                                                  prefix=f"{prefix}.in_proj")
        self.conv_weight = _sharded_param((self.di, self.d_conv), dtype, rank,
                                          lambda t: t.reshape(t.shape[0], t.shape[-1]))
        self.conv_bias = _sharded_param((self.di,), dtype, rank) if getattr(config, "use_conv_bias", True) else None
        self.x_proj = RowParallelLinear(self.d_inner, self.dt_rank + 2 * self.d_state, bias=False,
                                        prefix=f"{prefix}.x_proj")
        self.dt_proj = ColumnParallelLinear(self.dt_rank, self.d_inner, bias=False, prefix=f"{prefix}.dt_proj")
        self.dt_bias = _sharded_param((self.di,), torch.float32, rank, lambda t: t.float())
        self.A = _sharded_param((self.di, self.d_state), torch.float32, rank, lambda t: -torch.exp(t.float()))
        self.D = _sharded_param((self.di,), torch.float32, rank, lambda t: t.float())
        self.out_proj = RowParallelLinear(self.d_inner, self.hidden, bias=use_bias, prefix=f"{prefix}.out_proj")

    def state_shapes(self) -> tuple:
        return ((self.di, self.d_conv - 1), (self.di, self.d_state))

    def _norm(self, t: torch.Tensor) -> torch.Tensor:
        f = t.float()
        return (f * torch.rsqrt(f.pow(2).mean(-1, keepdim=True) + self.rms_eps)).to(t.dtype)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        md = get_forward_context().attn_metadata
        xz = self.in_proj(x)
        xs, z = xz[:, :self.di], xz[:, self.di:]
        if md is None or self.conv_state is None:
            # memory-profiling run: the projections carry the activation footprint
            p = self.x_proj(xs.contiguous())
            dt = self.dt_proj(p[:, :self.dt_rank].contiguous())
            return self.out_proj(dt * z)
        tbl = self.slot_table(md)
        xc = ops.causal_conv1d(xs, self.conv_weight, self.conv_bias, self.conv_state, md.query_start_loc,
                               md.seq_lens, tbl, md.max_query_len)
        p = self.x_proj(xc)
        R, N = self.dt_rank, self.d_state
        dt_low, B, C = p[:, :R], p[:, R:R + N], p[:, R + N:]
        if self.learned_norms:
            dt_low, B, C = self.dt_layernorm(dt_low.contiguous()), self.b_layernorm(B.contiguous()), \
                self.c_layernorm(C.contiguous())
        elif self.rms_eps is not None:
            dt_low, B, C = self._norm(dt_low), self._norm(B), self._norm(C)
        dt = self.dt_proj(dt_low.contiguous())
        y = ops.selective_scan(xc, dt, self.A, B, C, self.D, z, self.dt_bias, self.ssm_state, md.query_start_loc,
                               md.seq_lens, tbl, softplus=True, max_query_len=md.max_query_len)
        return self.out_proj(y)


class Mamba2Mixer(StateSpaceLayer):
    """Mamba-2 block: one fused in_proj (gate, conv input x|B|C, per-head dt), causal conv, per-head
    scalar-decay scan with grouped B/C, gated group RMSNorm, out_proj."""

    def __init__(self, config, prefix: str):
        super().__init__()
        if get_tensor_model_parallel_world_size() > 1:
            raise NotImplementedError("Mamba-2 layers run with tensor_parallel_size=1 (use data or pipeline "
                                      "parallelism to scale out)")
        self.hidden = config.hidden_size
        self.num_heads = int(config.num_heads)
        self.head_dim = int(config.head_dim)
        self.d_inner = int(getattr(config, "expand", 2) * self.hidden) if not hasattr(config, "intermediate_size") \
            else int(config.intermediate_size)
        if self.d_inner != self.num_heads * self.head_dim:
            self.d_inner = self.num_heads * self.head_dim
        self.d_state = int(config.state_size)
        self.d_conv = int(config.conv_kernel)
        self.n_groups = int(config.n_groups)
        self.conv_dim = self.d_inner + 2 * self.n_groups * self.d_state
        dtype = torch.get_default_dtype()
        self.conv_dtype = dtype
        use_bias = bool(getattr(config, "use_bias", False))
        self.in_proj = ColumnParallelLinear(self.hidden, self.d_inner + self.conv_dim + self.num_heads, bias=use_bias,
                                            prefix=f"{prefix}.in_proj")
        self.conv_weight = _sharded_param((self.conv_dim, self.d_conv), dtype, 0,
                                          lambda t: t.reshape(t.shape[0], t.shape[-1]))
        self.conv_bias = _sharded_param((self.conv_dim,), dtype, 0) if getattr(config, "use_conv_bias", True) else None
        self.dt_bias = _sharded_param((self.num_heads,), torch.float32, 0, lambda t: t.float())
        self.A = _sharded_param((self.num_heads,), torch.float32, 0, lambda t: -torch.exp(t.float()))
        self.D = _sharded_param((self.num_heads,), torch.float32, 0, lambda t: t.float())
        if bool(getattr(config, "gated_norm", True)):
            self.norm_weight = _sharded_param((self.d_inner,), dtype, 0)
        self.norm_eps = float(getattr(config, "layer_norm_epsilon", 1e-5))
        # The gated norm is per B/C group, as in the kernels these checkpoints were trained with (and in the
        # reference, mamba_mixer2.py Mixer2RMSNormGated); transformers' eager path normalises the whole
        # width instead, which ``gated_norm_groups: 1`` in the config reproduces.
        self.norm_groups = int(getattr(config, "gated_norm_groups", None) or self.n_groups)
        self.use_norm = bool(getattr(config, "gated_norm", True))      # Falcon-H1: y * silu(gate) without a norm
        lim = getattr(config, "time_step_limit", None)

        def num(v, default):       # JSON has no infinity: configs carry it as a string or a tagged object
            try:
                return float(v)
            except (TypeError, ValueError):
                return default
        lim = (num(lim[0], 0.0), num(lim[1], float("inf"))) if lim else (0.0, float("inf"))
        self.dt_limit = lim if (lim[0] > 0.0 or lim[1] != float("inf")) else None
        self.out_proj = RowParallelLinear(self.d_inner, self.hidden, bias=use_bias, prefix=f"{prefix}.out_proj")

    def state_shapes(self) -> tuple:
        return ((self.conv_dim, self.d_conv - 1), (self.d_inner, self.d_state))

    def _gated_norm(self, y: torch.Tensor) -> torch.Tensor:
        # y already carries the silu(gate) factor (applied inside the scan kernel)
        if not self.use_norm:
            return y
        T = y.shape[0]
        f = y.float().view(T, self.norm_groups, -1)
        f = f * torch.rsqrt(f.pow(2).mean(-1, keepdim=True) + self.norm_eps)
        return (f.view(T, -1) * self.norm_weight.float()).to(y.dtype)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        md = get_forward_context().attn_metadata
        p = self.in_proj(x)
        di, cd = self.d_inner, self.conv_dim
        gate, xbc, dt = p[:, :di], p[:, di:di + cd], p[:, di + cd:]
        if md is None or self.conv_state is None:
            return self.out_proj(self._gated_norm((xbc[:, :di] * gate).contiguous()))
        tbl = self.slot_table(md)
        xbc = ops.causal_conv1d(xbc, self.conv_weight, self.conv_bias, self.conv_state, md.query_start_loc,
                                md.seq_lens, tbl, md.max_query_len)
        gn = self.n_groups * self.d_state
        xs, B, C = xbc[:, :di], xbc[:, di:di + gn], xbc[:, di + gn:]
        bias, softplus = self.dt_bias, True
        if self.dt_limit is not None:
            dt = torch.nn.functional.softplus(dt.float() + self.dt_bias).clamp(*self.dt_limit).to(x.dtype)
            bias, softplus = None, False
        y = ops.selective_scan(xs, dt, self.A, B, C, self.D, gate, bias, self.ssm_state, md.query_start_loc,
                               md.seq_lens, tbl, softplus=softplus, num_groups=self.n_groups,
                               max_query_len=md.max_query_len)
        return self.out_proj(self._gated_norm(y))


class MambaBlock(nn.Module):
    def __init__(self, config, prefix: str, mixer_cls, **kw):
        super().__init__()
        eps = getattr(config, "layer_norm_epsilon", None) or getattr(config, "rms_norm_eps", 1e-5)
        self.norm = RMSNorm(config.hidden_size, eps)
        self.mixer = mixer_cls(config, f"{prefix}.mixer", **kw)

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor]):
        if residual is None:
            residual = x
            x = self.norm(x)
        else:
            x, residual = self.norm(x, residual)
        return self.mixer(x), residual


class MambaModel(nn.Module):
    def __init__(self, vllm_config: VllmConfig, prefix: str, mixer_cls, **kw):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        self.embeddings = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.start_layer, self.end_layer, self.layers = make_layers(
            config.num_hidden_layers, lambda p: MambaBlock(config, p, mixer_cls, **kw), f"{prefix}.layers")
        eps = getattr(config, "layer_norm_epsilon", None) or getattr(config, "rms_norm_eps", 1e-5)
        self.norm_f = RMSNorm(config.hidden_size, eps)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        pp = get_pp_group()
        if pp.is_first_rank:
            x = inputs_embeds if inputs_embeds is not None else self.embeddings(input_ids)
            residual = None
        else:
            x, residual = intermediate_tensors["hidden_states"], intermediate_tensors["residual"]
        for layer in self.layers[self.start_layer:self.end_layer]:
            x, residual = layer(x, residual)
        if not pp.is_last_rank:
            return IntermediateTensors({"hidden_states": x, "residual": residual})
        x, _ = self.norm_f(x, residual)
        return x


_RENAME = ((".mixer.conv1d.weight", ".mixer.conv_weight"), (".mixer.conv1d.bias", ".mixer.conv_bias"),
           (".mixer.dt_proj.bias", ".mixer.dt_bias"), (".mixer.A_log", ".mixer.A"), (".mixer.norm.weight",
                                                                                      ".mixer.norm_weight"))


class MambaForCausalLM(nn.Module):
    supports_pp = True
    is_attention_free = True
    mixer_cls = MambaMixer

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        config = vllm_config.model_config.hf_config
        self.config = config
        self.backbone = MambaModel(vllm_config, maybe_prefix(prefix, "backbone"), self.mixer_cls, **self._mixer_kw())
        self.tie = bool(getattr(config, "tie_word_embeddings", True))
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if self.tie:
            self.lm_head.tie_weights(self.backbone.embeddings)
        self.logits_processor = LogitsProcessor(config.vocab_size)

    def _mixer_kw(self) -> dict:
        return {}

    def get_input_embeddings(self, ids):
        return self.backbone.embeddings(ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        return self.backbone(input_ids, positions, intermediate_tensors, inputs_embeds)

    def compute_logits(self, hidden):
        return self.logits_processor(self.lm_head, hidden)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]) -> set[str]:
        params = dict(self.named_parameters())
        loaded: set[str] = set()
        for name, t in weights:
            if self.tie and name.startswith("lm_head."):
                continue
            for a, b in _RENAME:
                if name.endswith(a):
                    name = name[: -len(a)] + b
            if is_pp_missing_parameter(name, self) or name not in params:
                continue
            p = params[name]
            getattr(p, "weight_loader", default_weight_loader)(p, t)
            loaded.add(name)
        return loaded


class FalconMambaForCausalLM(MambaForCausalLM):
    def _mixer_kw(self) -> dict:
        return {"rms_eps": float(getattr(self.config, "mixer_rms_eps", 1e-6))}


class Mamba2ForCausalLM(MambaForCausalLM):
    mixer_cls = Mamba2Mixer
