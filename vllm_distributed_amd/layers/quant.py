"""fp8 (OCP e4m3fn) quantisation: linear layers, routed experts, KV-cache scales.

Two regimes on MI355X:
  * decode-sized M: weights are the traffic, so the skinny MFMA kernel streams e4m3 weights and
    widens them in registers (W8A16) -- half the HBM bytes, activations untouched;
  * prefill-sized M: activations are quantised (per-tensor static scale from the checkpoint or
    per-token dynamic) and the GEMM runs on the fp8 matrix cores through hipBLASLt
    (``torch._scaled_mm``), ~2x the bf16 MFMA rate.
Checkpoints already in fp8 ("quant_method": "fp8") load directly; bf16 checkpoints can be
quantised while loading (``quantization="fp8"``).
Reference: vllm/model_executor/layers/quantization/fp8.py (Fp8Config :58, Fp8LinearMethod :169,
Fp8MoEMethod :439, Fp8KVCacheMethod), utils/w8a8_utils.py (requantize_with_max_scale, Fp8LinearOp),
kv_cache.py (BaseKVCacheMethod).
"""
from __future__ import annotations

import os

from typing import Any, Optional

import torch
import torch.nn as nn

from .. import envs, ops
from ..logger import init_logger
from .linear import LinearBase, UnquantizedLinearMethod, set_weight_attrs

logger = init_logger(__name__)
FP8 = torch.float8_e4m3fn
FP8_MAX = 448.0


class QuantizationConfig:
    """Maps layers to quantisation methods."""

    name = "none"

    @classmethod
    def from_config(cls, cfg: dict) -> "QuantizationConfig":
        raise NotImplementedError

    def get_quant_method(self, layer: nn.Module, prefix: str):
        raise NotImplementedError

    def get_cache_scale_name(self, name: str) -> Optional[str]:
        """Checkpoint name of a KV scale -> parameter name on the Attention module."""
        for a, b in ((".k_proj.output_scale", ".attn.k_scale"), (".v_proj.output_scale", ".attn.v_scale"),
                     (".self_attn.k_scale", ".self_attn.attn.k_scale"), (".self_attn.v_scale", ".self_attn.attn.v_scale")):
            if name.endswith(a):
                return name[: -len(a)] + b
        return None


def _is_ignored(prefix: str, ignored: list[str], fused: dict[str, list[str]]) -> bool:
    if not ignored:
        return False
    leaf = prefix.rsplit(".", 1)[-1]
    names = [prefix]
    if leaf in fused:           # a fused module is skipped when its parts are
        names = [prefix[: -len(leaf)] + part for part in fused[leaf]]
    hit = [any(n == ig or n.endswith("." + ig) or ig in n for ig in ignored) for n in names]
    if any(hit) and not all(hit):
        raise ValueError(f"{prefix}: the parts of a fused layer must be quantised alike")
    return all(hit)


class Fp8Config(QuantizationConfig):
    name = "fp8"
    FUSED = {"qkv_proj": ["q_proj", "k_proj", "v_proj"], "gate_up_proj": ["gate_proj", "up_proj"]}

    def __init__(self, is_checkpoint_fp8_serialized: bool = False, activation_scheme: str = "dynamic",
                 ignored_layers: Optional[list[str]] = None, weight_block_size: Optional[list[int]] = None):
        if activation_scheme not in ("static", "dynamic"):
            raise ValueError(f"unsupported activation scheme {activation_scheme!r}")
        self.is_checkpoint_fp8_serialized = is_checkpoint_fp8_serialized
        self.activation_scheme = activation_scheme
        self.ignored_layers = ignored_layers or []
        # [rows, columns] of one weight-scale block (DeepSeek-V3 style ``weight_scale_inv``); None: per tensor
        self.weight_block_size = [int(b) for b in weight_block_size] if weight_block_size else None
        if self.weight_block_size is not None:
            if not is_checkpoint_fp8_serialized:
                raise ValueError("block-wise fp8 needs an fp8 checkpoint")
            if self.weight_block_size != [128, 128]:
                raise ValueError(f"fp8 weight blocks of {self.weight_block_size} are not supported (128 x 128)")
            if activation_scheme != "dynamic":
                raise ValueError("block-wise fp8 weights go with dynamic (per 128-group) activation scales")

    @property
    def mla_compatible(self) -> bool:
        """Block-scaled checkpoints: kv_b_proj is expanded to 16 bit while loading (its halves are the
        absorbed W_UK / W_UV of the latent-attention path), so that path stays available."""
        return self.weight_block_size is not None

    @classmethod
    def from_config(cls, cfg: dict) -> "Fp8Config":
        return cls("fp8" in str(cfg.get("quant_method", "")), cfg.get("activation_scheme", "dynamic"),
                   cfg.get("ignored_layers") or cfg.get("modules_to_not_convert"), cfg.get("weight_block_size"))

    def get_quant_method(self, layer: nn.Module, prefix: str):
        from .attention import Attention
        from .moe import FusedMoE
        if isinstance(layer, LinearBase):
            if _is_ignored(prefix, self.ignored_layers, self.FUSED) or prefix.endswith("lm_head"):
                return UnquantizedLinearMethod()
            return Fp8LinearMethod(self)
        if isinstance(layer, FusedMoE):
            return Fp8MoEMethod(self)
        if isinstance(layer, Attention):
            return Fp8KVCacheMethod(self)
        return None


class Int8Config(QuantizationConfig):
    """W8A8 int8 (compressed-tensors): int8 weights with per-channel or per-tensor scales,
    activations quantised per token on the fly (or with a calibrated static scale)."""
    name = "int8"
    FUSED = Fp8Config.FUSED

    def __init__(self, weight_strategy: str = "channel", dynamic_activations: bool = True,
                 ignored_layers: Optional[list[str]] = None, symmetric_activations: bool = True):
        if weight_strategy not in ("channel", "tensor"):
            raise ValueError(f"int8: unsupported weight strategy {weight_strategy!r}")
        self.weight_strategy, self.dynamic = weight_strategy, dynamic_activations
        # False: activations carry a zero point (q = round(x / s) + azp); the GEMM result is corrected by
        # azp times the column sums of the weight (reference: cutlass_scaled_mm_azp, scaled_mm_entry.cu)
        self.symmetric = symmetric_activations
        self.ignored_layers = ignored_layers or []

    def get_quant_method(self, layer: nn.Module, prefix: str):
        if isinstance(layer, LinearBase):
            if _is_ignored(prefix, self.ignored_layers, self.FUSED) or prefix.endswith("lm_head"):
                return UnquantizedLinearMethod()
            return Int8LinearMethod(self)
        return None


_INT_MM_OK: Optional[bool] = None


def _int8_matmul(xq: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """int8 [M, K] x int8 [N, K]^T -> fp32 [M, N] of exact integer sums (int32 accumulation through
    the library's int8 GEMM on the GPU; rows are padded to its minimum of 32)."""
    global _INT_MM_OK
    M = xq.shape[0]
    if xq.is_cuda and _INT_MM_OK is not False and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0:
        try:
            a = xq
            if M < 32:
                a = torch.zeros((32, xq.shape[1]), dtype=torch.int8, device=xq.device)
                a[:M] = xq
            y = torch._int_mm(a, w.t())[:M].float()
            _INT_MM_OK = True
            return y
        except RuntimeError:
            if _INT_MM_OK:
                raise
            _INT_MM_OK = False
    return xq.float() @ w.float().t()


class Int8LinearMethod:
    def __init__(self, cfg: Int8Config):
        self.cfg = cfg

    def create_weights(self, layer: nn.Module, in_per_rank: int, out_per_rank: int, dtype: torch.dtype) -> None:
        w = nn.Parameter(torch.empty(out_per_rank, in_per_rank, dtype=torch.int8), requires_grad=False)
        set_weight_attrs(w, {"input_dim": 1, "output_dim": 0})
        layer.register_parameter("weight", w)
        layer.orig_dtype = dtype
        n_shards = len(getattr(layer, "output_sizes", [0]))
        layer.logical_widths = [s // getattr(layer, "tp_size", 1) for s in layer.output_sizes] \
            if hasattr(layer, "output_sizes") else [out_per_rank]
        if self.cfg.weight_strategy == "channel":
            ws = nn.Parameter(torch.ones(out_per_rank, 1, dtype=torch.float32), requires_grad=False)
            set_weight_attrs(ws, {"output_dim": 0})
        else:
            ws = nn.Parameter(torch.ones(n_shards, dtype=torch.float32), requires_grad=False)
            set_weight_attrs(ws, {"per_shard_scalar": n_shards > 1})
        layer.register_parameter("weight_scale", ws)
        if not self.cfg.dynamic:
            ins = nn.Parameter(torch.ones(n_shards, dtype=torch.float32), requires_grad=False)
            set_weight_attrs(ins, {"per_shard_scalar": n_shards > 1})
            layer.register_parameter("input_scale", ins)
            if not self.cfg.symmetric:
                zp = nn.Parameter(torch.zeros(n_shards, dtype=torch.int8), requires_grad=False)
                set_weight_attrs(zp, {"per_shard_scalar": n_shards > 1})
                layer.register_parameter("input_zero_point", zp)

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        ws = layer.weight_scale.data.float()
        if self.cfg.weight_strategy == "tensor":
            ws = torch.cat([ws.reshape(-1)[i].expand(w) for i, w in enumerate(layer.logical_widths)]) \
                if ws.numel() > 1 else ws.reshape(1).expand(layer.weight.shape[0])
        layer.weight_scale = nn.Parameter(ws.reshape(1, -1).contiguous(), requires_grad=False)     # [1, N]
        ins = getattr(layer, "input_scale", None)
        if ins is not None and not self.cfg.symmetric:
            # fused shards (q/k/v) were calibrated separately: one range covering all of them
            zp = layer.input_zero_point.data.float()
            lo, hi = (ins.data.float() * (-128 - zp)).min(), (ins.data.float() * (127 - zp)).max()
            sc = ((hi - lo) / 255.0).clamp(min=1e-10)
            layer.input_scale = nn.Parameter(sc.reshape(1), requires_grad=False)
            layer.input_zero_point = nn.Parameter(torch.round(-128 - lo / sc).to(torch.int32).reshape(1),
                                                  requires_grad=False)
        else:
            layer.input_scale = nn.Parameter(ins.data.float().max().reshape(1), requires_grad=False) \
                if ins is not None else None
        if not self.cfg.symmetric:
            layer.azp_adj = nn.Parameter(layer.weight.data.to(torch.int32).sum(dim=1).float().reshape(1, -1),
                                         requires_grad=False)                   # [1, N] column sums

    def apply(self, layer: nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        if self.cfg.symmetric:
            xq, xs = ops.int8_quant(x2, getattr(layer, "input_scale", None))
            acc = _int8_matmul(xq, layer.weight)
        else:
            xq, xs, azp = ops.int8_quant_azp(x2, getattr(layer, "input_scale", None),
                                             getattr(layer, "input_zero_point", None))
            acc = _int8_matmul(xq, layer.weight) - azp.float().reshape(-1, 1) * layer.azp_adj
        y = acc * (xs.float().reshape(-1, 1) if xs.numel() > 1 else xs.float().reshape(1, 1)) * layer.weight_scale
        if bias is not None:
            y = y + bias.float()
        return y.to(x.dtype).view(*shape[:-1], -1)


# ---- bitsandbytes-style NF4, quantised in flight ------------------------------------------------------------
NF4_CODE = (-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635,
            -0.18477343022823334, -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725,
            0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0)


def nf4_quantize(w: torch.Tensor, block: int = 64) -> tuple[torch.Tensor, torch.Tensor]:
    """[N, K] -> (codes packed two per byte [N, K / 2] (first element in the high nibble), absmax [N, K / block])."""
    N, K = w.shape
    g = w.float().reshape(N, K // block, block)
    absmax = g.abs().amax(-1).clamp(min=1e-12)
    code = torch.tensor(NF4_CODE, device=w.device)
    # nearest code: the codebook is sorted, so a search over the midpoints finds it
    mids = (code[1:] + code[:-1]) / 2
    idx = torch.bucketize((g / absmax[..., None]).reshape(N, K), mids).to(torch.uint8)
    return (idx[:, 0::2] << 4) | idx[:, 1::2], absmax


def nf4_dequantize(packed: torch.Tensor, absmax: torch.Tensor, dtype: torch.dtype, block: int = 64) -> torch.Tensor:
    N = packed.shape[0]
    code = torch.tensor(NF4_CODE, device=packed.device, dtype=torch.float32)
    idx = torch.stack((packed >> 4, packed & 15), dim=-1).reshape(N, -1).long()
    return (code[idx].reshape(N, -1, block) * absmax[..., None]).reshape(N, -1).to(dtype)


class CodeBook4LinearMethod:
    """Linear layer over 4-bit code-book weights: ``qweight`` uint8 [N, K/2] (two codes per byte), ``scales`` fp32
    [N, K/group], ``code_table`` fp32 [16]; weight = code_table[code] * scale.  The codes stay packed in device memory
    (4 bits per weight + 32 / group bits of scale): up to 64 rows stream them through the MFMA skinny kernel of
    ops/csrc/cb4.hip, larger batches expand a transient 16-bit copy in front of the library GEMM.  Serves
    bitsandbytes NF4 / FP4 (groups of 64, first element in the high nibble; double-quantised scales are decoded to
    fp32 while loading) and NVFP4 (E2M1 codes, groups of 16, first element in the low nibble; the per-tensor scale
    is folded into the group scales).  Reference: vllm/model_executor/layers/quantization/bitsandbytes.py
    (BitsAndBytesLinearMethod._apply_4bit_weight), modelopt.py (ModelOptNvFp4LinearMethod)."""

    def __init__(self, group: int, hi_first: bool, table: Optional[tuple] = None):
        self.group, self.hi_first, self.table = int(group), bool(hi_first), table

    def _create_packed(self, layer: nn.Module, in_per_rank: int, out_per_rank: int, device=None) -> None:
        if in_per_rank % self.group:
            raise ValueError(f"{getattr(layer, 'prefix', '')}: {in_per_rank} input features per rank do not divide "
                             f"into groups of {self.group}")
        qw = nn.Parameter(torch.zeros(out_per_rank, in_per_rank // 2, dtype=torch.uint8, device=device),
                          requires_grad=False)
        sc = nn.Parameter(torch.zeros(out_per_rank, in_per_rank // self.group, dtype=torch.float32, device=device),
                          requires_grad=False)
        for name, p in (("qweight", qw), ("scales", sc)):
            set_weight_attrs(p, {"input_dim": 1, "output_dim": 0})
            layer.register_parameter(name, p)
        tab = torch.tensor(self.table or [0.0] * 16, dtype=torch.float32, device=device)
        layer.register_parameter("code_table", nn.Parameter(tab, requires_grad=False))     # loaded whole, never cut

    def create_weights(self, layer: nn.Module, in_per_rank: int, out_per_rank: int, dtype: torch.dtype) -> None:
        self._create_packed(layer, in_per_rank, out_per_rank)

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        """On the GPU, groups of 32 and more take the chunk layout once (codes: word transpose inside every 64-byte
        chunk; scales: [K/group, N]), which lets the decode kernel scale accumulators instead of weights."""
        from .. import ops
        qw = layer.qweight
        if getattr(layer, "cb4_chunked", False) or not qw.is_cuda or qw.shape[0] % 16 \
                or not ops.cb4_chunk_layout_ok(qw.shape[1] * 2, self.group):
            return
        codes, scales_t = ops.cb4_to_chunk_layout(qw.data, layer.scales.data)
        layer.qweight = nn.Parameter(codes, requires_grad=False)
        layer.scales = nn.Parameter(scales_t, requires_grad=False)
        layer.cb4_chunked = True

    def apply(self, layer: nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        from .. import ops
        shape = x.shape
        y = ops.cb4_linear(x.reshape(-1, shape[-1]), layer.qweight, layer.scales, layer.code_table, self.group,
                           self.hi_first, bias, getattr(layer, "cb4_chunked", False))
        return y.view(*shape[:-1], -1)


class Bnb4InflightConfig(QuantizationConfig):
    """``quantization="bitsandbytes"`` on an unquantised checkpoint: every linear layer's weight is quantised to
    NF4 (blocks of 64 values, one fp32 absmax each: 4.5 bits per weight in HBM) right after it is loaded and runs
    as a code-book layer from then on (``CodeBook4LinearMethod``).  Reference:
    vllm/model_executor/model_loader/bitsandbytes_loader.py (in-flight path :418-520) and
    layers/quantization/bitsandbytes.py; the bitsandbytes package itself is not part of this build, the NF4 code
    book and blocking are its published ones (pre-quantised checkpoints: models/dequant.py)."""
    name = "bitsandbytes"
    FUSED = Fp8Config.FUSED

    def get_quant_method(self, layer: nn.Module, prefix: str):
        if isinstance(layer, LinearBase):
            return UnquantizedLinearMethod() if prefix.endswith("lm_head") else Bnb4LinearMethod()
        return None


class Bnb4LinearMethod(CodeBook4LinearMethod):
    BLOCK = 64

    def __init__(self):
        super().__init__(self.BLOCK, True, NF4_CODE)

    def create_weights(self, layer: nn.Module, in_per_rank: int, out_per_rank: int, dtype: torch.dtype) -> None:
        w = nn.Parameter(torch.empty(out_per_rank, in_per_rank, dtype=dtype), requires_grad=False)
        set_weight_attrs(w, {"input_dim": 1, "output_dim": 0})
        layer.register_parameter("weight", w)

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        w = getattr(layer, "weight", None)
        if w is None or w.shape[1] % self.BLOCK:
            return                                   # already processed, or a width the blocking does not divide
        packed, absmax = nf4_quantize(w.data, self.BLOCK)
        self._create_packed(layer, w.shape[1], w.shape[0], device=w.device)
        layer.qweight.data.copy_(packed)
        layer.scales.data.copy_(absmax)
        del layer.weight
        super().process_weights_after_loading(layer)

    def apply(self, layer: nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        if getattr(layer, "weight", None) is not None:
            return torch.nn.functional.linear(x, layer.weight, bias)
        return super().apply(layer, x, bias)


def get_quant_config(vllm_config) -> Optional[QuantizationConfig]:
    mc = vllm_config.model_config
    q = mc.quantization
    if q in (None, "gguf") and str(mc.model).endswith(".gguf") and os.path.isfile(mc.model) and envs.VDA_GGUF_PACKED:
        from ..models.gguf import GGUFPackedConfig
        return GGUFPackedConfig(mc.model)
    if q in (None, "gguf"):
        return None
    hf_q = mc.hf_config.get("quantization_config") if hasattr(mc.hf_config, "get") else \
        getattr(mc.hf_config, "quantization_config", None)
    if q == "bitsandbytes" and not hf_q:
        return Bnb4InflightConfig()          # 16-bit checkpoint, quantised to NF4 while loading
    if q == "modelopt_fp4":
        if int((hf_q or {}).get("group_size", 16) or 16) != 16:
            raise ValueError("modelopt fp4: only groups of 16 (NVFP4) are supported")
        from ..models.dequant import LoadTimeDequant
        return LoadTimeDequant("nvfp4", dict(hf_q or {}))
    if q == "fp8" and hf_q and hf_q.get("weight_block_size") and envs.VDA_FP8_BLOCK_NATIVE \
            and list(hf_q.get("weight_block_size")) == [128, 128]:
        # 128 x 128 block scales run natively: weights stay e4m3, activations are quantised per 128-group
        return Fp8Config.from_config(hf_q)
    if q in ("awq", "gptq", "bitsandbytes") or (q == "fp8" and hf_q and hf_q.get("weight_block_size")):
        # weight-only formats without a kernel of their own: dequantised while the checkpoint loads
        from ..models.dequant import LoadTimeDequant
        return LoadTimeDequant(q, dict(hf_q or {}), tp_size=vllm_config.parallel_config.tensor_parallel_size)
    if q == "fp8" and hf_q and str(hf_q.get("quant_method", "")).lower() == "modelopt":
        # ModelOpt FP8 export: fp8 weights with per-tensor weight_scale and a calibrated input_scale
        return Fp8Config(True, "static", list(hf_q.get("exclude_modules") or []))
    if q == "fp8":
        return Fp8Config.from_config(hf_q) if hf_q else Fp8Config(False, "dynamic")
    if q == "quark":
        wq = ((hf_q or {}).get("global_quant_config") or {}).get("weight") or {}
        if isinstance(wq, dict) and wq.get("dtype") == "fp4":
            if int(wq.get("group_size", 32)) != 32 or wq.get("scale_format", "e8m0") != "e8m0":
                raise ValueError("quark fp4: only the microscaling layout (groups of 32, e8m0 scales) is supported")
            from ..models.dequant import LoadTimeDequant
            return LoadTimeDequant("mxfp4", dict(hf_q or {}))
    if q in ("compressed-tensors", "quark"):
        # accept only the plain per-tensor fp8 flavour of these container formats
        groups = (hf_q or {}).get("config_groups") or {}
        kinds = {(str(g.get("weights", {}).get("type", "")), int(g.get("weights", {}).get("num_bits", 8)))
                 for g in groups.values()}
        if groups and kinds == {("int", 8)}:
            g0 = next(iter(groups.values()))
            act = g0.get("input_activations") or {}
            if int(act.get("num_bits", 8)) != 8 or not g0["weights"].get("symmetric", True):
                raise ValueError(f"{q}: W8A8 int8 needs 8-bit activations and symmetric weights")
            return Int8Config(str(g0["weights"].get("strategy", "channel")), bool(act.get("dynamic", True)),
                              (hf_q or {}).get("ignore"), symmetric_activations=bool(act.get("symmetric", True)))
        ok = all(str(g.get("weights", {}).get("type", "")).startswith("float") and
                 int(g.get("weights", {}).get("num_bits", 8)) == 8 for g in groups.values())
        if not groups or not ok:
            raise ValueError(f"{q}: only per-tensor fp8 and symmetric int8 W8A8 weight schemes are supported")
        dyn = any(g.get("input_activations", {}).get("dynamic", True) for g in groups.values())
        return Fp8Config(True, "dynamic" if dyn else "static", (hf_q or {}).get("ignore"))
    raise ValueError(f"unsupported quantization {q!r}")


def per_tensor_quantize(w: torch.Tensor, scale: Optional[torch.Tensor] = None) -> tuple[torch.Tensor, torch.Tensor]:
    if scale is None:
        scale = (w.abs().amax().float() / FP8_MAX).clamp(min=1e-12)
    q = (w.float() / scale).clamp(-FP8_MAX, FP8_MAX).to(FP8)
    return q, scale.reshape(()).float()


def requantize_with_max_scale(weight: torch.Tensor, scales: torch.Tensor, logical_widths: list[int]
                              ) -> tuple[torch.Tensor, torch.Tensor]:
    """Fused projections arrive with one scale per logical shard; the GEMM wants one scale."""
    mx = scales.max()
    if scales.numel() == 1 or bool((scales == mx).all()):
        return weight, mx.reshape(())
    off = 0
    for i, wdt in enumerate(logical_widths):
        shard = weight[off:off + wdt]
        shard.copy_((shard.float() * scales[i] / mx).clamp(-FP8_MAX, FP8_MAX).to(FP8))
        off += wdt
    return weight, mx.reshape(())


def _scaled_mm_ok(x: torch.Tensor, w: torch.Tensor) -> bool:
    return x.is_cuda and x.shape[0] >= 1 and w.shape[0] % 16 == 0 and w.shape[1] % 16 == 0


def apply_fp8_linear(x: torch.Tensor, weight: torch.Tensor, weight_scale: torch.Tensor,
                     input_scale: Optional[torch.Tensor], bias: Optional[torch.Tensor]) -> torch.Tensor:
    """x [M, K] 16-bit, weight [N, K] e4m3, weight_scale scalar."""
    M = x.shape[0]
    if x.is_cuda and x.stride(-1) == 1 and \
            ops.skinny_gemm_ok(M, weight.shape[0], weight.shape[1], x.dtype, envs.VDA_FP8_W8A16_MAX_M):
        return ops.skinny_gemm(x, weight, bias, weight_scale.reshape(1))
    if _scaled_mm_ok(x, weight) and not envs.VDA_FORCE_TORCH_OPS:
        if input_scale is not None:
            xq, xs = ops.scaled_fp8_quant(x, input_scale.reshape(1))
            sa, sb = xs.reshape(()), weight_scale.reshape(())
        else:
            xq, xs = ops.per_token_fp8_quant(x)          # [M,1] row scales
            sa, sb = xs.view(M, 1), weight_scale.reshape(1, 1).expand(1, weight.shape[0]).contiguous()
        return torch._scaled_mm(xq, weight.t(), scale_a=sa, scale_b=sb, out_dtype=x.dtype, bias=bias)
    # CPU / odd shapes: dequantise
    w = (weight.float() * weight_scale.float()).to(x.dtype)
    return torch.nn.functional.linear(x, w, bias)


class Fp8LinearMethod:
    def __init__(self, cfg: Fp8Config):
        self.cfg = cfg

    def create_weights(self, layer: nn.Module, in_per_rank: int, out_per_rank: int, dtype: torch.dtype) -> None:
        serialized = self.cfg.is_checkpoint_fp8_serialized
        w = nn.Parameter(torch.empty(out_per_rank, in_per_rank, dtype=FP8 if serialized else dtype), requires_grad=False)
        set_weight_attrs(w, {"input_dim": 1, "output_dim": 0})
        layer.register_parameter("weight", w)
        layer.orig_dtype = dtype
        if self.cfg.weight_block_size is not None:
            bn, bk = self.cfg.weight_block_size
            tp = getattr(layer, "tp_size", 1)
            if tp > 1 and ((hasattr(layer, "input_size_per_partition") and in_per_rank % bk)
                           or (hasattr(layer, "output_size_per_partition") and out_per_rank % bn)):
                raise ValueError(f"{layer.prefix}: a tensor-parallel shard of {out_per_rank} x {in_per_rank} does not "
                                 f"line up with {bn} x {bk} scale blocks")
            s = nn.Parameter(torch.ones(-(-out_per_rank // bn), -(-in_per_rank // bk), dtype=torch.float32),
                             requires_grad=False)
            set_weight_attrs(s, {"input_dim": 1, "output_dim": 0, "block_n": bn, "block_k": bk})
            layer.register_parameter("weight_scale_inv", s)
            return
        n_shards = len(getattr(layer, "output_sizes", [0]))
        layer.logical_widths = [s // getattr(layer, "tp_size", 1) for s in getattr(layer, "output_sizes", [out_per_rank])] \
            if hasattr(layer, "output_sizes") else [out_per_rank]
        if serialized:
            ws = nn.Parameter(torch.full((n_shards,), torch.finfo(torch.float32).min), requires_grad=False)
            set_weight_attrs(ws, {"per_shard_scalar": n_shards > 1})
            layer.register_parameter("weight_scale", ws)
            if self.cfg.activation_scheme == "static":
                ins = nn.Parameter(torch.full((n_shards,), torch.finfo(torch.float32).min), requires_grad=False)
                set_weight_attrs(ins, {"per_shard_scalar": n_shards > 1})
                layer.register_parameter("input_scale", ins)

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        if self.cfg.weight_block_size is not None:
            if str(getattr(layer, "prefix", "")).endswith("kv_b_proj") or layer.weight.shape[1] % 128:
                # the latent-attention path slices this weight into its absorbed halves (and a K that is no
                # multiple of 128 has no tiled kernel): expand to 16 bit once, here
                from ..ops import reference as R
                w = layer.weight.data.float() * R.expand_block_scales(layer.weight_scale_inv.data.float(),
                                                                      *layer.weight.shape)
                layer.weight = nn.Parameter(w.to(layer.orig_dtype), requires_grad=False)
                layer.weight_scale_inv = None
                layer.fp8_expanded = True
            return
        if not self.cfg.is_checkpoint_fp8_serialized:
            q, s = per_tensor_quantize(layer.weight.data)
            layer.weight = nn.Parameter(q, requires_grad=False)
            layer.weight_scale = nn.Parameter(s, requires_grad=False)
            layer.input_scale = None
            return
        w, s = requantize_with_max_scale(layer.weight.data, layer.weight_scale.data, layer.logical_widths)
        layer.weight = nn.Parameter(w, requires_grad=False)
        layer.weight_scale = nn.Parameter(s.float(), requires_grad=False)
        ins = getattr(layer, "input_scale", None)
        layer.input_scale = nn.Parameter(ins.data.max().reshape(()), requires_grad=False) if ins is not None else None

    def apply(self, layer: nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]) if x.dim() != 2 else x
        if getattr(layer, "fp8_expanded", False):
            y = torch.nn.functional.linear(x2, layer.weight, bias)
        elif self.cfg.weight_block_size is not None:
            xq, xs = ops.per_group_fp8_quant(x2.contiguous(), 128)
            y = ops.fp8_gemm(xq, layer.weight, xs, layer.weight_scale_inv, x.dtype, bias, block=True)
        else:
            y = apply_fp8_linear(x2, layer.weight, layer.weight_scale, getattr(layer, "input_scale", None), bias)
        return y if x.dim() == 2 else y.view(*shape[:-1], -1)


class Fp8MoEMethod:
    """Per-expert per-tensor weight scales; activations stay 16-bit inside the grouped kernels."""

    def __init__(self, cfg: Fp8Config):
        self.cfg = cfg

    def create_weights(self, layer, E: int, H: int, I: int, dtype: torch.dtype) -> None:
        serialized = self.cfg.is_checkpoint_fp8_serialized
        wdt = FP8 if serialized else dtype
        w13 = nn.Parameter(torch.empty(E, 2 * I, H, dtype=wdt), requires_grad=False)
        w2 = nn.Parameter(torch.empty(E, H, I, dtype=wdt), requires_grad=False)
        layer.register_parameter("w13_weight", w13)
        layer.register_parameter("w2_weight", w2)
        for p in (w13, w2):
            set_weight_attrs(p, {"weight_loader": layer.weight_loader})
        if self.cfg.weight_block_size is not None:
            if I % 128 or H % 128:
                raise ValueError(f"block-scaled experts need hidden ({H}) and expert width per rank ({I}) to be "
                                 "multiples of 128")
            s13 = nn.Parameter(torch.ones(E, 2 * I // 128, H // 128, dtype=torch.float32), requires_grad=False)
            s2 = nn.Parameter(torch.ones(E, H // 128, I // 128, dtype=torch.float32), requires_grad=False)
            layer.register_parameter("w13_weight_scale_inv", s13)
            layer.register_parameter("w2_weight_scale_inv", s2)
            for p in (s13, s2):
                set_weight_attrs(p, {"weight_loader": layer.weight_loader, "moe_kind": "block_scale"})
            return
        if serialized:
            s13 = nn.Parameter(torch.ones(E, 2, dtype=torch.float32), requires_grad=False)
            s2 = nn.Parameter(torch.ones(E, dtype=torch.float32), requires_grad=False)
            layer.register_parameter("w13_weight_scale", s13)
            layer.register_parameter("w2_weight_scale", s2)
            for p in (s13, s2):
                set_weight_attrs(p, {"weight_loader": layer.weight_loader, "moe_kind": "scale_per_expert"})
            if self.cfg.activation_scheme == "static":
                # loaded for completeness; the W8A16 kernels do not quantise activations
                a13 = nn.Parameter(torch.ones(E, 2, dtype=torch.float32), requires_grad=False)
                a2 = nn.Parameter(torch.ones(E, dtype=torch.float32), requires_grad=False)
                layer.register_parameter("w13_input_scale", a13)
                layer.register_parameter("w2_input_scale", a2)
                for p in (a13, a2):
                    set_weight_attrs(p, {"weight_loader": layer.weight_loader, "moe_kind": "scale_per_expert"})

    def process_weights_after_loading(self, layer) -> None:
        E = layer.w13_weight.shape[0]
        I = layer.w13_weight.shape[1] // 2
        if self.cfg.weight_block_size is not None:
            return
        if not self.cfg.is_checkpoint_fp8_serialized:
            q13 = torch.empty_like(layer.w13_weight.data, dtype=FP8)
            q2 = torch.empty_like(layer.w2_weight.data, dtype=FP8)
            s13 = torch.empty(E, dtype=torch.float32, device=q13.device)
            s2 = torch.empty(E, dtype=torch.float32, device=q13.device)
            for e in range(E):
                q13[e], s13[e] = per_tensor_quantize(layer.w13_weight.data[e])
                q2[e], s2[e] = per_tensor_quantize(layer.w2_weight.data[e])
            layer.w13_weight = nn.Parameter(q13, requires_grad=False)
            layer.w2_weight = nn.Parameter(q2, requires_grad=False)
            layer.w13_weight_scale = nn.Parameter(s13, requires_grad=False)
            layer.w2_weight_scale = nn.Parameter(s2, requires_grad=False)
            return
        s = layer.w13_weight_scale.data
        mx = s.max(dim=1).values
        w = layer.w13_weight.data
        for e in range(E):
            for half in range(2):
                if s[e, half] != mx[e]:
                    blk = w[e, half * I:(half + 1) * I]
                    blk.copy_((blk.float() * s[e, half] / mx[e]).clamp(-FP8_MAX, FP8_MAX).to(FP8))
        layer.w13_weight_scale = nn.Parameter(mx.contiguous(), requires_grad=False)

    def fp8_rows_ok(self, layer, dtype: torch.dtype, pairs: int) -> bool:
        """May rows for ``pairs`` (row, expert) pairs arrive as e4m3 + one scale per row?"""
        return self.cfg.weight_block_size is None and ops.moe_fp8_rows_ok(layer.w13_weight, layer.w2_weight, dtype, pairs)

    def apply(self, layer, x: torch.Tensor, topk_w: torch.Tensor, topk_ids: torch.Tensor, x_scale=None,
              out_dtype=None) -> torch.Tensor:
        if self.cfg.weight_block_size is not None:
            return ops.fused_moe(x, layer.w13_weight, layer.w2_weight, topk_w, topk_ids, layer.w13_weight_scale_inv,
                                 layer.w2_weight_scale_inv, block_scales=True,
                                 activation=getattr(layer, "activation", "silu"))
        return ops.fused_moe(x, layer.w13_weight, layer.w2_weight, topk_w, topk_ids, layer.w13_weight_scale,
                             layer.w2_weight_scale, activation=getattr(layer, "activation", "silu"), x_scale=x_scale,
                             out_dtype=out_dtype)


class Fp8KVCacheMethod:
    """Adds k_scale / v_scale to an Attention layer (fp8 KV cache); 1.0 when the checkpoint has none."""

    def __init__(self, cfg: QuantizationConfig):
        self.cfg = cfg

    def create_weights(self, layer: nn.Module) -> None:
        layer.k_scale = nn.Parameter(torch.tensor(-1.0), requires_grad=False)
        layer.v_scale = nn.Parameter(torch.tensor(-1.0), requires_grad=False)

    def process_weights_after_loading(self, layer: nn.Module) -> None:
        k, v = float(layer.k_scale), float(layer.v_scale)
        if k <= 0 and v <= 0:
            k = v = 1.0
        elif k <= 0 or v <= 0:
            k = v = max(k, v)
        if getattr(layer, "cache_dtype", "auto") not in ("fp8", "fp8_e4m3") and (k != 1.0 or v != 1.0):
            logger.warning_once("checkpoint has KV-cache scales but kv_cache_dtype is not fp8; they are unused")
        layer._k_scale_float, layer._v_scale_float = k, v
        del layer.k_scale, layer.v_scale
