"""Fallback model implementation: any decoder-only language model that the installed ``transformers``
library can build runs through ITS module code, with this package's paged attention, KV cache,
tensor-parallel linear layers, sampler and scheduler around it.

Reference: vllm/model_executor/models/transformers.py (``TransformersForCausalLM``: attention function
registered with transformers :54-75, linear replacement by the config's tensor-parallel plan :82-110 and
:264-291, one ``Attention`` instance per layer :293-324, forward :377-405).

How it fits this package:
  * ``AutoModel.from_config(..., attn_implementation="vda")``: transformers calls ``_vda_attention`` in
    place of its own attention for every layer.  That function folds the rotated q / k / v it is handed
    into the fused [T, (Hq + 2 Hkv) * D] row layout of ``layers.attention.Attention`` and calls it: the
    cache write and the paged MFMA kernels are the ones every native model here uses (decode graphs
    included: the call is an ordinary kernel launch on the current stream).
  * The batch is ragged and flat ([T] token ids): transformers sees it as one sequence of batch size 1
    with explicit ``position_ids``; masks are never built (the mask hook returns None), causality and
    sequence boundaries come from this package's attention metadata.
  * With tensor parallelism the ``nn.Linear`` modules named by the config's ``base_model_tp_plan``
    ("colwise" / "rowwise") are replaced by Column/RowParallelLinear; transformers derives head counts
    from the projection width at run time, so sharded projections need no other change.
The registry resolves to this class for architectures it has no native implementation of
(``ModelConfig.model_impl`` "auto") and for every architecture with ``model_impl="transformers"``.
"""
from __future__ import annotations

import re
from typing import Iterable, Optional

import torch
import torch.nn as nn

from ..config import VllmConfig
from ..layers.attention import Attention
from ..layers.basic import LogitsProcessor, ParallelLMHead
from ..layers.linear import ColumnParallelLinear, RowParallelLinear
from ..logger import init_logger
from ..parallel.state import get_pp_group, get_tensor_model_parallel_world_size
from .utils import IntermediateTensors, load_weights_with_mapping

logger = init_logger(__name__)
_NAME = "vda"


def _vda_attention(module: nn.Module, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                   attention_mask=None, scaling: Optional[float] = None, **kwargs):
    """transformers' attention-function contract: query [1, Hq, T, D], key / value [1, Hkv, T, D] (rotary
    already applied) -> ([1, T, Hq, D], None)."""
    attn: Attention = module.__dict__["_vda_attn"]
    if scaling is not None:
        attn.scale = float(scaling)
    T = query.shape[-2]
    q, k, v = (x.transpose(1, 2).reshape(T, -1) for x in (query, key, value))
    out = attn(torch.cat((q, k, v), dim=-1))
    return out.view(1, T, query.shape[1], -1), None


def _register_with_transformers() -> None:
    from transformers import AttentionInterface
    AttentionInterface.register(_NAME, _vda_attention)
    try:        # no mask tensors: causality and sequence boundaries live in the attention metadata
        from transformers import AttentionMaskInterface
        AttentionMaskInterface.register(_NAME, lambda *a, **k: None)
    except ImportError:      # older transformers build the mask per implementation name and ignore unknown ones
        pass


def _replace_linears(model: nn.Module, plan: dict[str, str], dtype: torch.dtype) -> int:
    """Swap nn.Linear modules whose qualified name matches a plan pattern ("layers.*.mlp.up_proj")."""
    rules = [(re.compile("^(?:.*\\.)?" + pat.replace(".", "\\.").replace("*", "\\d+") + "$"), style)
             for pat, style in plan.items()]
    n = 0
    for parent_name, parent in list(model.named_modules()):
        for child_name, child in list(parent.named_children()):
            if not isinstance(child, nn.Linear):
                continue
            full = f"{parent_name}.{child_name}" if parent_name else child_name
            style = next((s for rx, s in rules if rx.match(full)), None)
            if style == "colwise":
                new = ColumnParallelLinear(child.in_features, child.out_features, bias=child.bias is not None,
                                           params_dtype=dtype, prefix=full)
            elif style == "rowwise":
                new = RowParallelLinear(child.in_features, child.out_features, bias=child.bias is not None,
                                        params_dtype=dtype, prefix=full)
            else:
                continue          # replicated (also the styles this backend does not shard)
            setattr(parent, child_name, new)
            n += 1
    return n


class TransformersForCausalLM(nn.Module):
    supports_pp = False
    is_transformers_backend = True

    def __init__(self, vllm_config: VllmConfig, prefix: str = ""):
        super().__init__()
        from transformers import AutoModel
        _register_with_transformers()
        mc = vllm_config.model_config
        from transformers import AutoConfig
        d = mc.hf_config.to_dict()
        d.pop("architectures", None)
        config = AutoConfig.for_model(d.pop("model_type"), **d)      # transformers' own config class
        text = config.get_text_config()
        self.config, self.text_config = config, text
        if get_pp_group().world_size > 1:
            raise ValueError("the transformers fallback does not split layers over pipeline stages")
        tp = get_tensor_model_parallel_world_size()
        dtype = mc.dtype
        config._attn_implementation = _NAME
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            self.model = AutoModel.from_config(config, attn_implementation=_NAME, dtype=dtype,
                                               trust_remote_code=bool(getattr(mc, "trust_remote_code", False)))
        finally:
            torch.set_default_dtype(prev)
        self.model.eval().requires_grad_(False)

        heads = text.num_attention_heads
        kv_heads = getattr(text, "num_key_value_heads", None) or heads
        head_dim = getattr(text, "head_dim", None) or text.hidden_size // heads
        if heads % tp or kv_heads % tp:
            raise ValueError(f"{heads} / {kv_heads} attention heads do not split over {tp} ranks")
        plan = getattr(self.model, "_tp_plan", None) or getattr(text, "base_model_tp_plan", None) or {}
        if tp > 1:
            if not plan:
                raise ValueError(f"{type(self.model).__name__} carries no tensor-parallel plan")
            n = _replace_linears(self.model, dict(plan), dtype)
            logger.info("transformers fallback: %d linear layers sharded over %d ranks", n, tp)

        # one Attention per decoder layer, found by the runner through named_modules()
        layer_types = getattr(text, "layer_types", None)
        window = getattr(text, "sliding_window", None)
        softcap = getattr(text, "attn_logit_softcapping", None)
        cache_dtype = vllm_config.cache_config.cache_dtype
        self.attention_instances = nn.ModuleDict()
        n_bound = 0
        for name, mod in self.model.named_modules():
            idx = getattr(mod, "layer_idx", None)
            if idx is None or not any(hasattr(mod, a) for a in ("q_proj", "qkv_proj", "query_key_value", "c_attn",
                                                                 "Wqkv")):
                continue
            sliding = window if (layer_types is not None and layer_types[idx] == "sliding_attention") or \
                (layer_types is None and getattr(mod, "sliding_window", None)) else None
            inst = Attention(heads // tp, head_dim, float(getattr(mod, "scaling", head_dim ** -0.5)), kv_heads // tp,
                             sliding_window=sliding, logits_soft_cap=softcap, cache_dtype=cache_dtype,
                             prefix=f"model.{name}.attn")
            self.attention_instances[str(idx)] = inst
            mod.__dict__["_vda_attn"] = inst          # not a second registration in the module tree
            n_bound += 1
        if n_bound != text.num_hidden_layers:
            raise ValueError(f"{type(self.model).__name__}: found {n_bound} attention modules for "
                             f"{text.num_hidden_layers} layers; this architecture needs a native implementation")

        self.tie = bool(getattr(text, "tie_word_embeddings", getattr(config, "tie_word_embeddings", False)))
        self.lm_head = ParallelLMHead(text.vocab_size, text.hidden_size, params_dtype=dtype)
        if self.tie:
            self.lm_head.weight = self.model.get_input_embeddings().weight
        scale = getattr(text, "logit_scale", None)
        self.logits_processor = LogitsProcessor(text.vocab_size, scale=float(scale) if scale else 1.0,
                                                soft_cap=getattr(text, "final_logit_softcapping", None))

    def get_input_embeddings(self, ids):
        return self.model.get_input_embeddings()(ids)

    def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
        kw = {"inputs_embeds": inputs_embeds[None]} if inputs_embeds is not None else {"input_ids": input_ids[None]}
        out = self.model(position_ids=positions[None], use_cache=False, return_dict=False, **kw)
        return out[0][0]

    def compute_logits(self, hidden):
        return self.logits_processor(self.lm_head, hidden)

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]) -> set[str]:
        base = getattr(self.model, "base_model_prefix", "model")

        def rename(n: str) -> Optional[str]:
            if n.startswith("lm_head."):
                return n
            if n.startswith(base + "."):
                n = n[len(base) + 1:]
            return "model." + n
        return load_weights_with_mapping(self, weights, (), rename=rename, tie_lm_head=self.tie)
