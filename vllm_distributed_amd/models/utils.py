"""Model-building helpers: pipeline-stage layer ranges, placeholder layers for
stages that do not own a layer, and the generic name-mapping weight loader.

Reference: vllm/model_executor/models/utils.py (AutoWeightsLoader :81,
PPMissingLayer :524, make_layers :625), vllm/sequence.py (IntermediateTensors).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Iterable, Optional

import torch
import torch.nn as nn

from ..layers.linear import default_weight_loader
from ..logger import init_logger
from ..parallel.state import get_pp_group
from ..parallel.utils import get_pp_indices

logger = init_logger(__name__)


class IntermediateTensors:
    """What one pipeline stage hands to the next (hidden_states, residual)."""

    def __init__(self, tensors: dict[str, torch.Tensor]):
        self.tensors = tensors

    def __getitem__(self, k):
        if isinstance(k, str):
            return self.tensors[k]
        return IntermediateTensors({n: t[k] for n, t in self.tensors.items()})

    def __setitem__(self, k, v):
        self.tensors[k] = v

    def items(self):
        return self.tensors.items()

    def __len__(self):
        return len(self.tensors)


class PPMissingLayer(nn.Identity):
    """Stands in for a layer owned by another pipeline stage."""

    def __init__(self, *a, **kw):
        super().__init__()

    def forward(self, *args, **kwargs):
        return args[0] if args else next(iter(kwargs.values()))


def make_layers(num_layers: int, layer_fn: Callable[[str], nn.Module], prefix: str) -> tuple[int, int, nn.ModuleList]:
    pp = get_pp_group()
    start, end = get_pp_indices(num_layers, pp.rank_in_group, pp.world_size)
    mods = [PPMissingLayer() for _ in range(start)] + \
           [layer_fn(f"{prefix}.{i}") for i in range(start, end)] + \
           [PPMissingLayer() for _ in range(end, num_layers)]
    return start, end, nn.ModuleList(mods)


def is_pp_missing_parameter(name: str, model: nn.Module) -> bool:
    missing = getattr(model, "_pp_missing_prefixes", None)
    if missing is None:
        missing = tuple(n + "." for n, m in model.named_modules() if isinstance(m, PPMissingLayer))
        model._pp_missing_prefixes = missing
    return any(name.startswith(p) for p in missing)


def maybe_prefix(prefix: str, name: str) -> str:
    return f"{prefix}.{name}" if prefix else name


def load_weights_with_mapping(model: nn.Module, weights: Iterable[tuple[str, torch.Tensor]],
                              stacked: list[tuple[str, str, object]] = (),
                              skip_substrings: tuple[str, ...] = ("rotary_emb.inv_freq",),
                              rename: Optional[Callable[[str], Optional[str]]] = None,
                              tie_lm_head: bool = False,
                              expert_mapping: list[tuple[str, str, int, str]] = ()) -> set[str]:
    """Route checkpoint tensors to parameters.

    ``stacked`` maps (fused_param_fragment, checkpoint_fragment, shard_id): a
    checkpoint name containing ``checkpoint_fragment`` is loaded into the
    parameter obtained by replacing it with ``fused_param_fragment``, with the
    parameter's ``weight_loader(param, tensor, shard_id)``.
    ``expert_mapping`` rows (param_fragment, checkpoint_fragment, expert_id, shard_id) route
    per-expert checkpoint tensors into the stacked expert parameters.
    Returns the set of parameter names that received data.
    """
    params = dict(model.named_parameters())
    loaded: set[str] = set()
    by_frag = {frag: (pfrag, eid, sid) for pfrag, frag, eid, sid in expert_mapping}
    for name, tensor in weights:
        if any(s in name for s in skip_substrings):
            continue
        if rename is not None:
            name = rename(name)
            if name is None:
                continue
        if tie_lm_head and name.startswith("lm_head."):
            continue
        if name.endswith(("k_scale", "v_scale", ".output_scale")):
            # calibrated KV-cache scales live on the Attention module
            from ..layers.quant import QuantizationConfig
            target = QuantizationConfig.get_cache_scale_name(None, name)
            if target is None or target not in params:
                continue
            params[target].data.fill_(float(tensor.float().max()))
            loaded.add(target)
            continue
        if by_frag and ".experts." in name:
            i = name.index("experts.")
            parts = name[i:].split(".")            # experts, <id>, <proj>, <suffix...>
            key = ".".join(parts[:3]) + "."
            if key in by_frag:
                pfrag, eid, sid = by_frag[key]
                target = name[:i] + pfrag + "_".join(parts[3:])
                if is_pp_missing_parameter(target, model):
                    continue
                if target not in params:
                    logger.warning_once("checkpoint tensor %s has no matching expert parameter %s", name, target)
                    continue
                p = params[target]
                if p.weight_loader(p, tensor, sid, eid):
                    loaded.add(target)
                continue
        for fused, frag, shard_id in stacked:
            if frag not in name:
                continue
            target = name.replace(frag, fused)
            if target.endswith(".bias") and target not in params:
                break
            if is_pp_missing_parameter(target, model):
                break
            if target not in params:
                continue
            p = params[target]
            p.weight_loader(p, tensor, shard_id)
            loaded.add(target)
            break
        else:
            if name.endswith(".bias") and name not in params:
                continue
            if is_pp_missing_parameter(name, model):
                continue
            if name not in params:
                logger.warning_once("checkpoint tensor %s has no matching parameter", name)
                continue
            p = params[name]
            getattr(p, "weight_loader", default_weight_loader)(p, tensor)
            loaded.add(name)
    return loaded
