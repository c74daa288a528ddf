"""Routed mixture-of-experts layer with tensor- and expert-parallel sharding.

Weights of all local experts live in two stacked tensors, K-contiguous per row:
    w13_weight [E_local, 2 * I_shard, H]   gate rows then up rows
    w2_weight  [E_local, H, I_shard]
Tensor parallel shards I; expert parallel (``enable_expert_parallel``) instead gives each rank
of the EP group (TP x DP ranks) a contiguous range of whole experts and leaves I unsharded.
Either way every rank computes a partial sum over its weights for the tokens it sees and the
partial sums are all-reduced over the TP group; with data-parallel attention in front, tokens
are first all-gathered over the DP group and the result is reduce-scattered back (dispatch /
combine of parallel/all2all.py).

Reference: vllm/model_executor/layers/fused_moe/layer.py (FusedMoE :829-1568: expert map :780,
weight_loader :1045-1259, select_experts :1261-1364, forward_impl :1496), fused_moe.py.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn as nn

from .. import envs, ops
from ..config import get_current_vllm_config
from ..logger import init_logger
from ..parallel.state import (get_dp_group, get_ep_group, get_tensor_model_parallel_rank,
                              get_tensor_model_parallel_world_size, tensor_model_parallel_all_reduce)
from .linear import set_weight_attrs

logger = init_logger(__name__)


def determine_expert_map(ep_size: int, ep_rank: int, global_num_experts: int) -> tuple[int, Optional[torch.Tensor]]:
    """Contiguous ranges, the first ``E % ep`` ranks take one extra expert.  Returns the local count
    and a [E_global] int32 map global id -> local id (-1 = elsewhere); None without EP."""
    if ep_size == 1:
        return global_num_experts, None
    base, rem = divmod(global_num_experts, ep_size)
    local = base + (1 if ep_rank < rem else 0)
    start = ep_rank * base + min(ep_rank, rem)
    m = torch.full((global_num_experts,), -1, dtype=torch.int32)
    m[start:start + local] = torch.arange(local, dtype=torch.int32)
    return local, m


def grouped_topk(scores: torch.Tensor, k: int, renormalize: bool, num_expert_group: int, topk_group: int,
                 bias: Optional[torch.Tensor] = None):
    """DeepSeek-style routing: keep the best ``topk_group`` groups of experts, then top-k inside."""
    T, E = scores.shape
    sel = scores + bias.unsqueeze(0) if bias is not None else scores
    g = sel.view(T, num_expert_group, -1)
    gscore = g.topk(2, dim=-1)[0].sum(-1) if bias is not None else g.max(dim=-1).values
    gidx = gscore.topk(topk_group, dim=-1, sorted=False)[1]
    gmask = torch.zeros_like(gscore).scatter_(1, gidx, 1.0)
    mask = gmask.unsqueeze(-1).expand(T, num_expert_group, E // num_expert_group).reshape(T, E)
    masked = sel.masked_fill(mask == 0, float("-inf"))
    ids = masked.topk(k, dim=-1, sorted=False)[1]
    w = scores.gather(1, ids)
    if renormalize:
        w = w / w.sum(dim=-1, keepdim=True)
    return w.float(), ids.to(torch.int32)


class UnquantizedMoEMethod:
    def create_weights(self, layer: "FusedMoE", E: int, H: int, I: int, dtype: torch.dtype) -> None:
        w13 = nn.Parameter(torch.empty(E, 2 * I, H, dtype=dtype), requires_grad=False)
        w2 = nn.Parameter(torch.empty(E, H, I, dtype=dtype), requires_grad=False)
        layer.register_parameter("w13_weight", w13)
        layer.register_parameter("w2_weight", w2)
        for p in (w13, w2):
            set_weight_attrs(p, {"weight_loader": layer.weight_loader})

    def process_weights_after_loading(self, layer: "FusedMoE") -> None:
        pass

    def apply(self, layer: "FusedMoE", x: torch.Tensor, topk_w: torch.Tensor, topk_ids: torch.Tensor) -> torch.Tensor:
        return ops.fused_moe(x, layer.w13_weight, layer.w2_weight, topk_w, topk_ids,
                             activation=getattr(layer, "activation", "silu"))


class FusedMoE(nn.Module):
    _fp8_wire_logged = False

    def __init__(self, num_experts: int, top_k: int, hidden_size: int, intermediate_size: int,
                 params_dtype: Optional[torch.dtype] = None, reduce_results: bool = True, renormalize: bool = True,
                 use_grouped_topk: bool = False, num_expert_group: Optional[int] = None,
                 topk_group: Optional[int] = None, quant_config=None, prefix: str = "",
                 custom_routing_function: Optional[Callable] = None, scoring_func: str = "softmax",
                 e_score_correction_bias: Optional[torch.Tensor] = None, routed_scaling_factor: float = 1.0,
                 activation: str = "silu"):
        super().__init__()
        if activation not in ops.MOE_ACTIVATIONS:
            raise ValueError(f"{prefix}: expert activation {activation!r} is not supported "
                             f"(supported: {', '.join(ops.MOE_ACTIVATIONS)})")
        self.activation = activation            # of the gate half, fused into the first expert GEMM
        self.params_dtype = params_dtype or torch.get_default_dtype()
        self.global_num_experts, self.top_k = num_experts, top_k
        self.hidden_size = hidden_size
        self.prefix = prefix
        self.reduce_results, self.renormalize = reduce_results, renormalize
        self.use_grouped_topk, self.num_expert_group, self.topk_group = use_grouped_topk, num_expert_group, topk_group
        self.custom_routing_function, self.scoring_func = custom_routing_function, scoring_func
        self.e_score_correction_bias = e_score_correction_bias
        self.routed_scaling_factor = routed_scaling_factor
        pc = get_current_vllm_config().parallel_config
        self.tp_size, self.tp_rank = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        self.dp_size = get_dp_group().world_size
        self.use_ep = bool(pc.enable_expert_parallel) and get_ep_group().world_size > 1
        # EPLB: physical slots (logical experts + replicas of hot ones), re-placed at run time
        self.eplb_state = None
        self.eplb_layer = -1
        self.num_physical_experts = num_experts
        if self.use_ep and pc.enable_eplb:
            self.num_physical_experts = num_experts + pc.num_redundant_experts
            if self.num_physical_experts % get_ep_group().world_size:
                raise ValueError("num_experts + num_redundant_experts must be divisible by the EP size for EPLB")
        if self.use_ep:
            ep = get_ep_group()
            self.ep_size, self.ep_rank = ep.world_size, ep.rank_in_group
            self.local_num_experts, emap = determine_expert_map(self.ep_size, self.ep_rank, self.num_physical_experts)
            self.register_buffer("expert_map", emap, persistent=False)
            self.shard_size, self.shard_rank = 1, 0           # whole experts: I is not split
        else:
            self.ep_size, self.ep_rank = 1, 0
            self.local_num_experts, self.expert_map = num_experts, None
            self.shard_size, self.shard_rank = self.tp_size, self.tp_rank
        if intermediate_size % self.shard_size:
            raise ValueError(f"{prefix}: intermediate size {intermediate_size} not divisible by TP {self.shard_size}")
        self.intermediate_size_per_partition = intermediate_size // self.shard_size
        from .. import envs
        self._ep_route = bool(envs.VDA_EP_DISPATCH)
        self._ep_disp = None
        self._max_tokens = int(get_current_vllm_config().scheduler_config.max_num_batched_tokens)
        qm = quant_config.get_quant_method(self, prefix) if quant_config is not None else None
        self.quant_method = qm or UnquantizedMoEMethod()
        self.quant_method.create_weights(self, self.local_num_experts, hidden_size,
                                         self.intermediate_size_per_partition, self.params_dtype)

    # ---- weight loading -------------------------------------------------------------------
    def weight_loader(self, param: nn.Parameter, loaded: torch.Tensor, shard_id: str, expert_id: int) -> bool:
        """``shard_id`` in {"w1" (gate), "w3" (up), "w2" (down)}; ``expert_id`` is global.
        Returns False when the expert is not local (nothing loaded)."""
        if self.num_physical_experts != self.global_num_experts:
            # initial EPLB placement: physical slot s hosts logical expert s % num_logical
            ok = False
            for slot in range(expert_id, self.num_physical_experts, self.global_num_experts):
                ok |= self._load_slot(param, loaded, shard_id, slot)
            return ok
        return self._load_slot(param, loaded, shard_id, expert_id)

    def _load_slot(self, param: nn.Parameter, loaded: torch.Tensor, shard_id: str, expert_id: int) -> bool:
        local = expert_id
        if self.expert_map is not None:
            local = int(self.expert_map[expert_id])
            if local < 0:
                return False
        I = self.intermediate_size_per_partition
        kind = getattr(param, "moe_kind", "weight")
        if kind == "scale_per_expert":
            # one scalar per (expert, projection); w1/w3 kept apart until post-processing
            if param.dim() == 2:
                param.data[local, {"w1": 0, "w3": 1}.get(shard_id, 0)] = loaded.reshape(()).to(param.dtype)
            else:
                param.data[local] = loaded.reshape(()).to(param.dtype)
            return True
        if kind == "block_scale":
            # one scale per 128 x 128 block: same sharding as the weights, in units of blocks
            Ib = I // 128
            if shard_id == "w2":
                src = loaded.narrow(1, self.shard_rank * Ib, Ib) if self.shard_size > 1 else loaded
                param.data[local].copy_(src)
            else:
                src = loaded.narrow(0, self.shard_rank * Ib, Ib) if self.shard_size > 1 else loaded
                off = 0 if shard_id == "w1" else Ib
                param.data[local, off:off + Ib].copy_(src)
            return True
        if shard_id == "w2":
            src = loaded.narrow(1, self.shard_rank * I, I) if self.shard_size > 1 else loaded
            param.data[local].copy_(src)
        else:
            src = loaded.narrow(0, self.shard_rank * I, I) if self.shard_size > 1 else loaded
            off = 0 if shard_id == "w1" else I
            param.data[local, off:off + I].copy_(src)
        return True

    @staticmethod
    def make_expert_params_mapping(gate_name: str, down_name: str, up_name: str, num_experts: int
                                   ) -> list[tuple[str, str, int, str]]:
        """(param fragment, checkpoint fragment, expert id, shard id) for every expert/projection."""
        out = []
        for e in range(num_experts):
            for sid, ck in (("w1", gate_name), ("w2", down_name), ("w3", up_name)):
                out.append(("experts.w13_" if sid != "w2" else "experts.w2_", f"experts.{e}.{ck}.", e, sid))
        return out

    def expert_weight_tensors(self) -> list[torch.Tensor]:
        """Tensors whose first dimension is the local physical expert slot (moved by EPLB)."""
        return [p.data for n, p in self.named_parameters(recurse=False)
                if p.dim() >= 1 and p.shape[0] == self.local_num_experts]

    # ---- routing + compute ----------------------------------------------------------------------
    def select_experts(self, router_logits: torch.Tensor):
        k = self.top_k
        if self.custom_routing_function is not None:
            w, ids = self.custom_routing_function(router_logits, k, self.renormalize)
            return w.float(), ids.to(torch.int32)
        if self.use_grouped_topk:
            scores = torch.softmax(router_logits.float(), -1) if self.scoring_func == "softmax" \
                else torch.sigmoid(router_logits.float())
            return grouped_topk(scores, k, self.renormalize, self.num_expert_group, self.topk_group,
                                self.e_score_correction_bias)
        if self.scoring_func == "sigmoid":
            scores = torch.sigmoid(router_logits.float())
            w, ids = scores.topk(k, dim=-1)
            if self.renormalize:
                w = w / w.sum(-1, keepdim=True)
            return w, ids.to(torch.int32)
        return ops.moe_topk_softmax(router_logits, k, self.renormalize)

    def _forward_ep_dispatch(self, x: torch.Tensor, lg: torch.Tensor, routed=None) -> torch.Tensor:
        """Expert parallel with token routing.  The ranks that hold the same rows (all of them after tensor-parallel
        attention; the tensor-parallel ranks of one replica after data-parallel attention) each take a slice of
        those rows, route it to the expert owners through the all-to-all-v, the owners compute, the results return
        and are weighted at the source, and an all-gather among the same ranks replicates the rows again.
        Per xGMI link: T k H / ep^2 bytes each way plus the all-gather, against the T H of an all-reduce -- or, with
        data-parallel attention, against an all-gather of EVERY replica's tokens to every rank.
        Reference: vllm/model_executor/layers/fused_moe/layer.py:1439-1484,
        vllm/distributed/device_communicators/all2all.py:77-264."""
        from ..parallel.all2all import get_expert_dispatcher
        ep = get_ep_group()
        T, H = x.shape
        dp_meta = None
        if self.dp_size > 1:
            from ..forward_context import get_forward_context
            dp_meta = get_forward_context().dp
            if dp_meta is None:
                raise RuntimeError("data-parallel expert dispatch needs DPMetadata in the forward context")
        parts, part_rank = (self.tp_size, self.tp_rank) if self.dp_size > 1 else (self.ep_size, self.ep_rank)
        Tp = -(-T // parts)
        # rows per (source, destination) pair of this call: the same on every rank of the expert-parallel group
        cap_tokens = -(-int(dp_meta.max_tokens) // parts) if dp_meta is not None else Tp
        if self._ep_disp is None:
            self._ep_disp = get_expert_dispatcher(ep, self.local_num_experts, self.top_k, H * x.element_size(),
                                                  -(-self._max_tokens // parts))
        lo = part_rank * Tp
        pad = parts * Tp - T
        if pad:
            x = torch.nn.functional.pad(x, (0, 0, 0, pad))
            if routed is None:
                lg = torch.nn.functional.pad(lg, (0, 0, 0, pad))
            else:
                routed = tuple(torch.nn.functional.pad(t, (0, 0, 0, pad)) for t in routed)
        xs = x[lo:lo + Tp]
        if routed is None:
            topk_w, topk_ids = self.select_experts(lg[lo:lo + Tp])
        else:
            topk_w, topk_ids = routed[0][lo:lo + Tp], routed[1][lo:lo + Tp]
        if self.eplb_state is not None:
            topk_ids = self.eplb_state.map_topk(self.eplb_layer, topk_ids)
        if pad:     # rows past T are padding: they go nowhere
            live = (torch.arange(lo, lo + Tp, device=x.device) < T).unsqueeze(1)
            topk_ids = torch.where(live, topk_ids, torch.full((), -1, dtype=topk_ids.dtype, device=x.device))
        if self.eplb_state is not None:
            self.eplb_state.record(self.eplb_layer, topk_ids)
        if self.routed_scaling_factor != 1.0:
            topk_w = topk_w * self.routed_scaling_factor
        # rows the owner would quantise per token anyway travel as e4m3 + their scale (H + 16 bytes instead of 2 H)
        cap = min(self._ep_disp.rx.cap, -(-max(cap_tokens, 1) * self.top_k // 16) * 16)
        fp8_rows = getattr(self.quant_method, "fp8_rows_ok", None)
        if envs.VDA_EP_FP8_WIRE and fp8_rows is not None and fp8_rows(self, x.dtype, ep.world_size * cap):
            if not FusedMoE._fp8_wire_logged:
                FusedMoE._fp8_wire_logged = True
                logger.info("expert-parallel dispatch: e4m3 rows on the wire (%d instead of %d bytes per row)",
                            H + 16, H * x.element_size())
            xq, xsc = ops.per_token_fp8_quant(xs)
            pack = torch.empty((Tp, H + 16), dtype=torch.uint8, device=x.device)
            pack[:, :H] = xq.view(torch.uint8)
            pack[:, H:H + 4] = xsc.reshape(Tp, 1).contiguous().view(torch.uint8)
            recv, local_ids, st = self._ep_disp.dispatch(pack, topk_ids.to(torch.int32), cap_tokens)
            ones = torch.ones((recv.shape[0], 1), dtype=torch.float32, device=x.device)
            y = self.quant_method.apply(self, recv[:, :H].view(torch.float8_e4m3fn), ones, local_ids,
                                        x_scale=recv[:, H:H + 4].contiguous().view(torch.float32), out_dtype=x.dtype)
        else:
            recv, local_ids, st = self._ep_disp.dispatch(xs.contiguous(), topk_ids.to(torch.int32), cap_tokens)
            ones = torch.ones((recv.shape[0], 1), dtype=torch.float32, device=x.device)
            y = self.quant_method.apply(self, recv, ones, local_ids)
        part = self._ep_disp.combine(y, topk_w, st)
        if self.dp_size == 1:
            out = self._ep_disp.rx.all_gather_rows(part)[:T]
        elif self.tp_size > 1:
            from ..parallel.state import tensor_model_parallel_all_gather
            out = tensor_model_parallel_all_gather(part, 0)[:T]
        else:
            out = part[:T]
        if not self.reduce_results:
            # the caller adds a tensor-parallel partial sum (shared experts) and all-reduces the total:
            # hand it the complete routed output in 1 / tp parts
            out = out / self.tp_size
        return out

    def forward(self, hidden_states: torch.Tensor, router_logits: Optional[torch.Tensor], routed=None) -> torch.Tensor:
        """``routed`` = (weights f32 [T, k], logical expert ids i32 [T, k]) when the caller did the routing
        itself (models that scale the experts' INPUT by the router score pass weights of one)."""
        x, lg = hidden_states, router_logits
        dispatch = self.dp_size > 1
        if self.use_ep and self._ep_route and (not dispatch or routed is None):
            # expert parallel: tokens travel to the expert owners through the all-to-all-v, also when attention
            # ran data-parallel (each replica routes its own tokens; nothing is all-gathered)
            return self._forward_ep_dispatch(x, lg, routed)
        if dispatch:
            if routed is not None:
                raise NotImplementedError("caller-side routing together with data-parallel dispatch")
            from ..parallel.all2all import dp_combine, dp_dispatch
            x, lg, meta = dp_dispatch(x, lg)
        topk_w, topk_ids = self.select_experts(lg) if routed is None else routed
        if self.eplb_state is not None:
            topk_ids = self.eplb_state.map_topk(self.eplb_layer, topk_ids)      # logical -> physical slot
            self.eplb_state.record(self.eplb_layer, topk_ids)
        if self.expert_map is not None:
            topk_ids = self.expert_map[topk_ids.long()]
        if self.routed_scaling_factor != 1.0:
            topk_w = topk_w * self.routed_scaling_factor
        out = self.quant_method.apply(self, x, topk_w, topk_ids)
        if dispatch:
            out = dp_combine(out, meta)
        if self.reduce_results and (self.tp_size > 1 or self.ep_size > 1):
            out = tensor_model_parallel_all_reduce(out)
        return out
