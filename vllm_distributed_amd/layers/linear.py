"""Tensor-parallel linear layers with shard-aware weight loading.

Megatron-style: column-parallel layers shard the output dimension,
row-parallel layers shard the input dimension and all-reduce the result.
Weights keep the checkpoint layout ``[out, in]`` (K-contiguous rows), which is
exactly what the skinny MFMA GEMM streams and what hipBLASLt takes as B^T.

Reference: vllm/model_executor/layers/linear.py (ColumnParallel :370-527,
MergedColumn :530, QKV :808 with GQA head replication :855-864, RowParallel
:1158-1333), vllm/model_executor/layers/utils.py:66-116 (GEMM dispatch).
"""
from __future__ import annotations

from typing import Optional, Sequence, Union

import torch
import torch.nn as nn

from .. import ops
from ..parallel.state import (get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                              tensor_model_parallel_all_gather, tensor_model_parallel_all_reduce,
                              tensor_model_parallel_reduce_scatter)
from ..parallel.utils import ensure_divisible


def set_weight_attrs(p: torch.Tensor, attrs: dict) -> None:
    for k, v in attrs.items():
        setattr(p, k, v)


def default_weight_loader(param: torch.Tensor, loaded: torch.Tensor) -> None:
    if param.numel() == 1 and loaded.numel() == 1:
        param.data.fill_(loaded.item())
        return
    if param.shape != loaded.shape:
        raise ValueError(f"shape mismatch loading weight: param {tuple(param.shape)} vs checkpoint {tuple(loaded.shape)}")
    param.data.copy_(loaded)


class UnquantizedLinearMethod:
    def create_weights(self, layer: nn.Module, in_per_rank: int, out_per_rank: int, dtype: torch.dtype) -> None:
        w = nn.Parameter(torch.empty(out_per_rank, in_per_rank, dtype=dtype), requires_grad=False)
        set_weight_attrs(w, {"input_dim": 1, "output_dim": 0})
        layer.register_parameter("weight", w)

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        pass

    def apply(self, layer: nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        if x.dim() == 2:
            return ops.linear(x, layer.weight, bias)
        shape = x.shape
        return ops.linear(x.reshape(-1, shape[-1]), layer.weight, bias).view(*shape[:-1], -1)


class LinearBase(nn.Module):
    def __init__(self, input_size: int, output_size: int, params_dtype: Optional[torch.dtype] = None,
                 quant_config=None, prefix: str = ""):
        super().__init__()
        self.input_size, self.output_size = input_size, output_size
        self.params_dtype = params_dtype or torch.get_default_dtype()
        self.prefix = prefix
        self.quant_method = quant_config.get_quant_method(self, prefix) if quant_config is not None \
            else UnquantizedLinearMethod()
        if self.quant_method is None:
            self.quant_method = UnquantizedLinearMethod()

    def _make_bias(self, n: int, loader) -> None:
        b = nn.Parameter(torch.zeros(n, dtype=self.params_dtype), requires_grad=False)
        set_weight_attrs(b, {"output_dim": 0, "weight_loader": loader})
        self.register_parameter("bias", b)


class ReplicatedLinear(LinearBase):
    def __init__(self, input_size, output_size, bias=True, params_dtype=None, quant_config=None, prefix=""):
        super().__init__(input_size, output_size, params_dtype, quant_config, prefix)
        self.quant_method.create_weights(self, input_size, output_size, self.params_dtype)
        set_weight_attrs(self.weight, {"weight_loader": self.weight_loader})
        if bias:
            self._make_bias(output_size, self.weight_loader)
        else:
            self.register_parameter("bias", None)

    def weight_loader(self, param, loaded, *_):
        default_weight_loader(param, loaded.reshape(param.shape) if loaded.numel() == param.numel() else loaded)

    def forward(self, x):
        return self.quant_method.apply(self, x, self.bias)


class ColumnParallelLinear(LinearBase):
    def __init__(self, input_size, output_size, bias=True, gather_output=False, params_dtype=None,
                 quant_config=None, output_sizes: Optional[Sequence[int]] = None, prefix=""):
        super().__init__(input_size, output_size, params_dtype, quant_config, prefix)
        self.tp_size = get_tensor_model_parallel_world_size()
        self.tp_rank = get_tensor_model_parallel_rank()
        self.gather_output = gather_output
        self.output_sizes = list(output_sizes) if output_sizes is not None else [output_size]
        self.output_size_per_partition = self._out_per_rank()
        self.quant_method.create_weights(self, input_size, self.output_size_per_partition, self.params_dtype)
        for name, p in self.named_parameters(recurse=False):
            set_weight_attrs(p, {"weight_loader": self.weight_loader})
        if bias:
            self._make_bias(self.output_size_per_partition, self.weight_loader)
        else:
            self.register_parameter("bias", None)

    def _out_per_rank(self) -> int:
        return ensure_divisible(self.output_size, self.tp_size)

    def weight_loader(self, param, loaded, shard_id=None):
        out_dim = getattr(param, "output_dim", None)
        if out_dim is not None and loaded.dim() > 0 and param.numel() > 1:
            size = param.shape[out_dim]
            loaded = loaded.narrow(out_dim, self.tp_rank * size, size)
        default_weight_loader(param, loaded)

    def forward(self, x):
        y = self.quant_method.apply(self, x, self.bias)
        if self.gather_output and self.tp_size > 1:
            y = tensor_model_parallel_all_gather(y)
        return y


def _in_blocks(prefix: str, param, off: int, size: int, tp: int) -> tuple[int, int]:
    """Offsets of a logical shard along the output dimension, in units of the parameter's scale blocks
    (``block_n`` attribute: one scale row per that many weight rows; 1 for ordinary parameters)."""
    blk = int(getattr(param, "block_n", 1) or 1)
    if blk == 1:
        return off, size
    if off % blk or (size % blk and tp > 1):
        raise ValueError(f"{prefix}: a shard of {size} rows at row {off} does not line up with {blk}-row scale blocks")
    return off // blk, -(-size // blk)


class MergedColumnParallelLinear(ColumnParallelLinear):
    """Several column-parallel projections of the same input fused into one GEMM
    (gate_proj + up_proj).  ``shard_id`` selects the logical projection."""

    def forward_silu_mul(self, x: torch.Tensor) -> Optional[torch.Tensor]:
        """SiLU(gate) * up of a gated MLP with the activation fused into the GEMM epilogue, or None when this
        layer / batch does not qualify (quantised or LoRA-wrapped layer, bias, gathered output, large batch): the
        caller then runs the projection and the activation kernel separately."""
        if (type(self) is not MergedColumnParallelLinear or type(self.quant_method) is not UnquantizedLinearMethod
                or self.bias is not None or self.gather_output or len(self.output_sizes) != 2
                or self.output_sizes[0] != self.output_sizes[1] or not x.is_cuda):
            return None
        shape = x.shape
        y = ops.gate_up_silu_mul(x.reshape(-1, shape[-1]), self.weight)
        return None if y is None else y.view(*shape[:-1], -1)

    def __init__(self, input_size, output_sizes: Sequence[int], bias=True, gather_output=False,
                 params_dtype=None, quant_config=None, prefix=""):
        tp = get_tensor_model_parallel_world_size()
        for s in output_sizes:
            ensure_divisible(s, tp)
        super().__init__(input_size, sum(output_sizes), bias, gather_output, params_dtype, quant_config,
                         output_sizes, prefix)

    def weight_loader(self, param, loaded, shard_id: Optional[int] = None):
        out_dim = getattr(param, "output_dim", None)
        if shard_id is None:
            # checkpoint already holds the fused tensor: load piece by piece
            if out_dim is None or loaded.dim() == 0:
                return default_weight_loader(param, loaded)
            off = 0
            for i, s in enumerate(self.output_sizes):
                self.weight_loader(param, loaded.narrow(out_dim, off, s), i)
                off += s
            return
        per_tensor_scale = getattr(param, "per_shard_scalar", False)
        if per_tensor_scale:
            # one scalar per logical projection (fp8 per-tensor weight scales)
            param.data[shard_id].copy_(loaded.reshape(()))
            return
        if out_dim is None:
            return default_weight_loader(param, loaded)
        off = sum(self.output_sizes[:shard_id]) // self.tp_size
        size = self.output_sizes[shard_id] // self.tp_size
        off, size = _in_blocks(self.prefix, param, off, size, self.tp_size)
        dst = param.data.narrow(out_dim, off, size)
        src = loaded.narrow(out_dim, self.tp_rank * size, size)
        if dst.shape != src.shape:
            raise ValueError(f"{self.prefix}: shard {shard_id} shape {tuple(src.shape)} != {tuple(dst.shape)}")
        dst.copy_(src)


class QKVParallelLinear(ColumnParallelLinear):
    """Fused q/k/v projection.  Query heads are split over TP ranks; KV heads are
    split when there are at least TP of them and replicated otherwise."""

    def __init__(self, hidden_size, head_size, total_num_heads, total_num_kv_heads=None, bias=True,
                 params_dtype=None, quant_config=None, prefix=""):
        tp = get_tensor_model_parallel_world_size()
        self.head_size = head_size
        self.total_num_heads = total_num_heads
        self.total_num_kv_heads = total_num_kv_heads or total_num_heads
        self.num_heads = ensure_divisible(total_num_heads, tp)
        if tp >= self.total_num_kv_heads:
            self.num_kv_heads, self.num_kv_head_replicas = 1, ensure_divisible(tp, self.total_num_kv_heads)
        else:
            self.num_kv_heads, self.num_kv_head_replicas = ensure_divisible(self.total_num_kv_heads, tp), 1
        out = (self.num_heads + 2 * self.num_kv_heads) * tp * head_size
        super().__init__(hidden_size, out, bias, False, params_dtype, quant_config,
                         [self.num_heads * head_size * tp, self.num_kv_heads * head_size * tp,
                          self.num_kv_heads * head_size * tp], prefix)

    def _offsets(self) -> dict[str, tuple[int, int]]:
        q, kv = self.num_heads * self.head_size, self.num_kv_heads * self.head_size
        return {"q": (0, q), "k": (q, kv), "v": (q + kv, kv)}

    def weight_loader(self, param, loaded, shard_id: Optional[str] = None):
        out_dim = getattr(param, "output_dim", None)
        if shard_id is None:
            if out_dim is None or loaded.dim() == 0:
                return default_weight_loader(param, loaded)
            # fused checkpoint tensor laid out [q_total | k_total | v_total]
            qt = self.total_num_heads * self.head_size
            kt = self.total_num_kv_heads * self.head_size
            for sid, off, size in (("q", 0, qt), ("k", qt, kt), ("v", qt + kt, kt)):
                self.weight_loader(param, loaded.narrow(out_dim, off, size), sid)
            return
        if getattr(param, "per_shard_scalar", False):
            param.data[{"q": 0, "k": 1, "v": 2}[shard_id]].copy_(loaded.reshape(()))
            return
        if out_dim is None:
            return default_weight_loader(param, loaded)
        off, size = self._offsets()[shard_id]
        off, size = _in_blocks(self.prefix, param, off, size, self.tp_size)
        src_rank = self.tp_rank if shard_id == "q" else self.tp_rank // self.num_kv_head_replicas
        dst = param.data.narrow(out_dim, off, size)
        src = loaded.narrow(out_dim, src_rank * size, size)
        if dst.shape != src.shape:
            raise ValueError(f"{self.prefix}: {shard_id} shard shape {tuple(src.shape)} != {tuple(dst.shape)}")
        dst.copy_(src)


_SEQUENCE_PARALLEL = False


def set_sequence_parallel(active: bool) -> bool:
    """While active, row-parallel layers end in a reduce-scatter over the token dimension instead of
    an all-reduce: each rank keeps T/tp rows, the norm that follows runs on those rows only, and the
    caller all-gathers before the next column-parallel GEMM.  Same bytes on the wire as an all-reduce,
    1/tp of the norm / residual work.  Returns the previous state.
    Reference: vllm/compilation/sequence_parallelism.py:87-482 (there an FX graph rewrite)."""
    global _SEQUENCE_PARALLEL
    old, _SEQUENCE_PARALLEL = _SEQUENCE_PARALLEL, bool(active)
    return old


_DEFER_REDUCE = False


def set_defer_reduce(active: bool) -> bool:
    """While active, row-parallel layers return their partial sums and the caller owns the reduction
    (the micro-batched forward starts it on the communication stream and overlaps it with the other
    micro-batch's GEMMs).  Returns the previous state."""
    global _DEFER_REDUCE
    old, _DEFER_REDUCE = _DEFER_REDUCE, bool(active)
    return old


class RowParallelLinear(LinearBase):
    def __init__(self, input_size, output_size, bias=True, input_is_parallel=True, params_dtype=None,
                 reduce_results=True, quant_config=None, prefix=""):
        super().__init__(input_size, output_size, params_dtype, quant_config, prefix)
        self.tp_size = get_tensor_model_parallel_world_size()
        self.tp_rank = get_tensor_model_parallel_rank()
        self.input_is_parallel = input_is_parallel
        self.reduce_results = reduce_results
        self.input_size_per_partition = ensure_divisible(input_size, self.tp_size)
        self.quant_method.create_weights(self, self.input_size_per_partition, output_size, self.params_dtype)
        for name, p in self.named_parameters(recurse=False):
            set_weight_attrs(p, {"weight_loader": self.weight_loader})
        if bias:
            self._make_bias(output_size, default_weight_loader)
        else:
            self.register_parameter("bias", None)
        if bias and not reduce_results and self.tp_size > 1:
            raise ValueError("bias would be added tp_size times when results are not reduced")

    def weight_loader(self, param, loaded, *_):
        in_dim = getattr(param, "input_dim", None)
        if in_dim is not None and loaded.dim() > 0 and param.numel() > 1:
            size = param.shape[in_dim]
            loaded = loaded.narrow(in_dim, self.tp_rank * size, size)
        default_weight_loader(param, loaded)

    def forward(self, x):
        if not self.input_is_parallel and self.tp_size > 1:
            x = x.chunk(self.tp_size, dim=-1)[self.tp_rank].contiguous()
        # bias only once: on rank 0 before the sum
        b = self.bias if (self.tp_rank == 0 or self.tp_size == 1) else None
        y = self.quant_method.apply(self, x, b)
        if self.reduce_results and self.tp_size > 1 and not _DEFER_REDUCE:
            if _SEQUENCE_PARALLEL:
                return tensor_model_parallel_reduce_scatter(y, 0)
            y = tensor_model_parallel_all_reduce(y)
        return y
