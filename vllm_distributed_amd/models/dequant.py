"""Load-time dequantisation of weight-only checkpoint formats.

GPTQ and AWQ (4/8-bit integers with group scales and zero points), bitsandbytes (NF4 / FP4 with
optional double quantisation, LLM.int8()) and block-wise fp8 (``weight_scale_inv`` per 128x128
block, DeepSeek-V3 style) checkpoints are turned into 16-bit
weights while the checkpoint streams in: the iterator below pairs every packed tensor with its
scales / zero points and yields the ordinary ``<module>.weight``.  The model then runs with the
exact values of the quantised model on the bf16 GEMM path; the memory saving of the format is
not kept.
Reference: vllm/model_executor/layers/quantization/gptq.py, awq.py (tensor layouts),
utils/fp8_utils.py (block scales).
"""
from __future__ import annotations

from typing import Iterable, Iterator, Optional

import torch

_AWQ_ORDER = [0, 4, 1, 5, 2, 6, 3, 7]        # nibble j of a packed word holds column _AWQ_ORDER^-1


def _unpack_rows(packed: torch.Tensor, bits: int) -> torch.Tensor:
    """int32 [R, C] with 32/bits values per word along dim 0 (low bits first) -> int32 [R*32/bits, C]."""
    per = 32 // bits
    shifts = torch.arange(0, 32, bits, dtype=torch.int32).view(1, per, 1)
    v = (packed.to(torch.int32).unsqueeze(1) >> shifts) & ((1 << bits) - 1)
    return v.reshape(packed.shape[0] * per, packed.shape[1])


def _unpack_cols(packed: torch.Tensor, bits: int) -> torch.Tensor:
    """int32 [R, C] with 32/bits values per word along dim 1 -> int32 [R, C*32/bits]."""
    per = 32 // bits
    shifts = torch.arange(0, 32, bits, dtype=torch.int32).view(1, 1, per)
    v = (packed.to(torch.int32).unsqueeze(2) >> shifts) & ((1 << bits) - 1)
    return v.reshape(packed.shape[0], packed.shape[1] * per)


def dequantize_gptq(qweight, qzeros, scales, g_idx: Optional[torch.Tensor], bits: int, zero_offset: int,
                    dtype: torch.dtype) -> torch.Tensor:
    """-> [N, K].  qweight [K*bits/32, N], qzeros [G, N*bits/32], scales [G, N], g_idx [K]."""
    q = _unpack_rows(qweight, bits)                       # [K, N]
    z = _unpack_cols(qzeros, bits) + zero_offset          # [G, N]
    K = q.shape[0]
    if g_idx is None:
        g_idx = torch.arange(K, dtype=torch.int64) // (K // scales.shape[0])
    g = g_idx.to(torch.int64)
    w = (q - z[g]).to(torch.float32) * scales.to(torch.float32)[g]
    return w.t().contiguous().to(dtype)


def dequantize_awq(qweight, qzeros, scales, bits: int, dtype: torch.dtype) -> torch.Tensor:
    """-> [N, K].  qweight [K, N*bits/32], qzeros [G, N*bits/32], scales [G, N]; 4-bit words are
    interleaved (columns 0,2,4,6,1,3,5,7)."""
    def cols(p):
        v = _unpack_cols(p, bits)
        if bits == 4:
            v = v.view(v.shape[0], -1, 8)[:, :, _AWQ_ORDER].reshape(v.shape[0], -1)
        return v
    q, z = cols(qweight), cols(qzeros)
    K, G = q.shape[0], scales.shape[0]
    g = torch.arange(K, dtype=torch.int64) // (K // G)
    w = (q - z[g]).to(torch.float32) * scales.to(torch.float32)[g]
    return w.t().contiguous().to(dtype)


def gptq_to_packed4(qweight, qzeros, scales, zero_offset: int, dtype: torch.dtype):
    """GPTQ 4-bit tensors -> (packed int32 [N, K/8], scales [N, G], zero points [N, G]) for ops.w4_linear."""
    from ..ops import w4_pack
    q = _unpack_rows(qweight, 4).t().contiguous()                     # [N, K]
    z = (_unpack_cols(qzeros, 4) + zero_offset).t().contiguous()      # [N, G]
    return w4_pack(q), scales.t().contiguous().to(dtype), z.to(dtype)


def gptq_to_packed(qweight, qzeros, scales, g_idx, bits: int, zero_offset: int, dtype: torch.dtype) -> dict:
    """GPTQ 4- or 8-bit tensors -> the tensors of ``PackedIntLinearMethod``: qweight (4 bit: int32 [N, K/8] in the
    nibble order of ops.w4_pack; 8 bit: uint8 [N, K]), scales [N, G], zero points [N, G] and, for activation
    reordering (``g_idx`` not in natural order), ``perm`` int32 [K]: the weights' K columns are stored in the order
    that makes every group contiguous, and the activations' columns are gathered in that order before the GEMM."""
    from ..ops import w4_pack
    q = _unpack_rows(qweight, bits).t().contiguous()                   # [N, K]
    z = (_unpack_cols(qzeros, bits) + zero_offset).t().contiguous()    # [N, G]
    out = {}
    K, G = q.shape[1], scales.shape[0]
    if g_idx is not None:
        g = g_idx.to(torch.int64).reshape(-1)
        natural = torch.arange(K, dtype=torch.int64) // (K // G)
        if not torch.equal(g, natural):
            perm = torch.argsort(g, stable=True)
            if not torch.equal(g[perm], natural):
                raise ValueError("gptq act-order: groups of unequal size are not supported")
            q = q[:, perm].contiguous()
            out["perm"] = perm.to(torch.int32)
    out["qweight"] = w4_pack(q) if bits == 4 else q.to(torch.uint8)
    out["scales"] = scales.t().contiguous().to(dtype)
    out["qzeros"] = z.to(dtype)
    return out


def awq_to_packed4(qweight, qzeros, scales, dtype: torch.dtype):
    from ..ops import w4_pack

    def cols(p):
        v = _unpack_cols(p, 4)
        return v.view(v.shape[0], -1, 8)[:, :, _AWQ_ORDER].reshape(v.shape[0], -1)
    q = cols(qweight).t().contiguous()                                # [N, K]
    z = cols(qzeros).t().contiguous()                                 # [N, G]
    return w4_pack(q), scales.t().contiguous().to(dtype), z.to(dtype)


def dequantize_block_fp8(weight, scale_inv, block: tuple[int, int], dtype: torch.dtype) -> torch.Tensor:
    N, K = weight.shape
    s = scale_inv.to(torch.float32).repeat_interleave(block[0], 0)[:N].repeat_interleave(block[1], 1)[:, :K]
    return (weight.to(torch.float32) * s).to(dtype)


_E2M1 = (0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, -0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0)
MX_BLOCK = 32


def dequantize_mxfp4(packed: torch.Tensor, scale: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """OCP microscaling fp4: ``packed`` uint8 [N, K/2] holds two E2M1 codes per byte (element 2i in the low
    nibble), ``scale`` uint8 [N, K/32] one E8M0 exponent (2^(s-127)) per 32 elements.
    Reference: vllm/model_executor/layers/quantization/utils/mxfp4_utils.py (dequant_mxfp4)."""
    if packed.dtype != torch.uint8 or scale.dtype != torch.uint8:
        raise ValueError("mxfp4: packed weights and scales are uint8")
    N, K2 = packed.shape
    if scale.shape != (N, 2 * K2 // MX_BLOCK):
        raise ValueError(f"mxfp4: scale shape {tuple(scale.shape)} does not match weights {tuple(packed.shape)}")
    lut = torch.tensor(_E2M1, dtype=torch.float32, device=packed.device)
    codes = torch.stack((packed & 15, packed >> 4), dim=-1).reshape(N, 2 * K2).long()
    s = torch.exp2(scale.to(torch.float32) - 127.0)
    return (lut[codes].view(N, -1, MX_BLOCK) * s.unsqueeze(-1)).view(N, 2 * K2).to(dtype)


def mxfp4_to_packed8(packed: torch.Tensor, scale: torch.Tensor, dtype: torch.dtype) -> dict:
    """Microscaling fp4 as 8-bit integers, EXACTLY: every E2M1 value is a multiple of one half, so
    value = (q - 128) * 2^(s - 128) with q - 128 in {0, +-1, +-2, +-3, +-4, +-6, +-8, +-12} -- the W8A16 layout of
    ``PackedIntLinearMethod`` with groups of 32, zero point 128 and a power-of-two scale (exact in bf16; fp16 holds
    scales down to 2^-24).  Nine bits per weight in memory instead of sixteen, and no table look-up in the GEMM."""
    if packed.dtype != torch.uint8 or scale.dtype != torch.uint8:
        raise ValueError("mxfp4: packed weights and scales are uint8")
    N, K2 = packed.shape
    if scale.shape != (N, 2 * K2 // MX_BLOCK):
        raise ValueError(f"mxfp4: scale shape {tuple(scale.shape)} does not match weights {tuple(packed.shape)}")
    twice = torch.tensor([int(2 * v) for v in _E2M1], dtype=torch.int16, device=packed.device)
    codes = torch.stack((packed & 15, packed >> 4), dim=-1).reshape(N, 2 * K2).long()
    q = (twice[codes] + 128).to(torch.uint8)
    sc = torch.exp2(scale.to(torch.float32) - 128.0).to(dtype)
    return {"qweight": q.contiguous(), "scales": sc.contiguous(), "qzeros": torch.full_like(sc, 128.0)}


def dequantize_nvfp4(packed: torch.Tensor, scale: torch.Tensor, global_scale: torch.Tensor,
                     dtype: torch.dtype) -> torch.Tensor:
    """NVFP4: E2M1 codes two per byte (element 2i in the low nibble) [N, K/2], one fp8-e4m3 scale per 16
    elements [N, K/16], and a per-tensor fp32 scale multiplying them.
    Reference: vllm/model_executor/layers/quantization/modelopt.py (ModelOptNvFp4LinearMethod)."""
    N, K2 = packed.shape
    if packed.dtype != torch.uint8 or scale.shape != (N, 2 * K2 // 16):
        raise ValueError(f"nvfp4: weights {tuple(packed.shape)} {packed.dtype} / scales {tuple(scale.shape)} mismatch")
    lut = torch.tensor(_E2M1, dtype=torch.float32, device=packed.device)
    codes = torch.stack((packed & 15, packed >> 4), dim=-1).reshape(N, 2 * K2).long()
    s = scale.to(torch.float32) * global_scale.to(torch.float32).max()
    return (lut[codes].view(N, -1, 16) * s.unsqueeze(-1)).view(N, 2 * K2).to(dtype)


def quant_dequant_mxfp4(x: torch.Tensor) -> torch.Tensor:
    """Round ``x`` to what microscaling fp4 can hold (shared power-of-two scale per 32 elements of the last
    dimension, E2M1 elements, round to nearest even) and return it in x's dtype: what a W4A4 checkpoint's
    activations see.  Reference: mxfp4_utils.py (quant_dequant_mxfp4)."""
    shape = x.shape
    b = x.float().reshape(-1, MX_BLOCK)
    amax = b.abs().amax(dim=-1, keepdim=True)
    e = torch.floor(torch.log2(amax.clamp(min=2.0 ** -126))) - 2.0        # 6 = 1.5 * 2^2 is the largest element
    s = torch.exp2(e.clamp(-127.0, 127.0))
    v = (b / s).clamp(-6.0, 6.0)
    a = v.abs()
    # E2M1 grid: steps of 0.5 below 2, of 1 below 4, of 2 up to 6
    step = torch.where(a < 2.0, 0.5, torch.where(a < 4.0, 1.0, 2.0))
    q = torch.round(a / step) * step
    return (torch.sign(v) * q * s).reshape(shape).to(x.dtype)


def dequantize_bnb4(packed: torch.Tensor, absmax: torch.Tensor, quant_map: torch.Tensor, state: dict,
                    nested_absmax: Optional[torch.Tensor], nested_quant_map: Optional[torch.Tensor],
                    dtype: torch.dtype) -> torch.Tensor:
    """bitsandbytes 4-bit (NF4 / FP4) -> [N, K].  ``packed`` holds two codes per byte (first element in the
    high nibble), ``quant_map`` the 16 code values, ``absmax`` one scale per ``blocksize`` elements; with
    double quantisation the scales themselves are 8-bit codes with their own map, block scales and offset."""
    bs = int(state["blocksize"])
    shape = tuple(int(x) for x in state["shape"])
    n = 1
    for d in shape:
        n *= d
    if nested_absmax is not None:
        nbs = int(state.get("nested_blocksize", 256))
        a = nested_quant_map.float()[absmax.long().view(-1)]
        a = a * nested_absmax.float().repeat_interleave(nbs)[: a.numel()]
        absmax = a + float(state.get("nested_offset", 0.0))
    codes = packed.view(-1).to(torch.uint8)
    vals = torch.stack([codes >> 4, codes & 15], dim=-1).view(-1)[:n].long()
    w = quant_map.float()[vals]
    pad = (-n) % bs
    if pad:
        w = torch.cat([w, w.new_zeros(pad)])
    w = (w.view(-1, bs) * absmax.float().view(-1, 1)).view(-1)[:n]
    return w.view(shape).to(dtype)


BNB_BLOCK = 64


def bnb4_to_codebook(packed: torch.Tensor, absmax: torch.Tensor, quant_map: torch.Tensor, state: dict,
                     nested_absmax: Optional[torch.Tensor], nested_quant_map: Optional[torch.Tensor],
                     name: str = "") -> dict:
    """bitsandbytes 4-bit tensors as they are kept in memory by ``CodeBook4LinearMethod``: the byte stream re-shaped
    to [N, K/2], one fp32 scale per block (double-quantised scales decoded here, once), the 16 code values."""
    bs = int(state["blocksize"])
    shape = tuple(int(x) for x in state["shape"])
    if len(shape) != 2 or bs != BNB_BLOCK or shape[1] % bs:
        raise ValueError(f"{name}: bitsandbytes blocks of {bs} over a weight of shape {shape} do not line up with "
                         f"rows in groups of {BNB_BLOCK}; set VDA_INT4_WEIGHTS=0 to expand this checkpoint while loading")
    if nested_absmax is not None:
        nbs = int(state.get("nested_blocksize", 256))
        a = nested_quant_map.float()[absmax.long().view(-1)]
        a = a * nested_absmax.float().repeat_interleave(nbs)[: a.numel()]
        absmax = a + float(state.get("nested_offset", 0.0))
    N, K = shape
    return {"qweight": packed.reshape(-1).to(torch.uint8)[: N * K // 2].view(N, K // 2).contiguous(),
            "scales": absmax.float().view(N, K // bs).contiguous(),
            "code_table": quant_map.float().reshape(16).contiguous()}


def nvfp4_to_codebook(packed: torch.Tensor, scale: torch.Tensor, global_scale: torch.Tensor) -> dict:
    """NVFP4 tensors as ``CodeBook4LinearMethod`` keeps them: the codes untouched, the e4m3 group scales times the
    per-tensor scale as fp32 (so fused projections with different per-tensor scales share one layer)."""
    N, K2 = packed.shape
    if packed.dtype != torch.uint8 or scale.shape != (N, 2 * K2 // 16):
        raise ValueError(f"nvfp4: weights {tuple(packed.shape)} {packed.dtype} / scales {tuple(scale.shape)} mismatch")
    s = scale.to(torch.float32) * global_scale.to(torch.float32).max()
    return {"qweight": packed.contiguous(), "scales": s.contiguous(),
            "code_table": torch.tensor(_E2M1, dtype=torch.float32)}


def dequantize_bnb8(weight: torch.Tensor, scb: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """LLM.int8() weights: int8 rows with one fp scale per output row (value = q * SCB / 127)."""
    return (weight.float() * (scb.float().view(-1, 1) / 127.0)).to(dtype)


class PackedIntLinearMethod:
    """Linear layer over packed 4- or 8-bit weights with group scales and zero points (W4A16 / W8A16).  The weights
    stay packed in device memory (model memory = checkpoint size): decode-sized batches of 4-bit layers stream them
    through the MFMA skinny kernel (ops/csrc/w4.hip), everything else through the ping-pong GEMM that widens the
    weights on their way into the MFMA operands (ops/csrc/gemm_wq.hip), for any number of rows.  Activation
    reordering (GPTQ ``desc_act``) is a column gather of the activations in front of the GEMM.
    Reference: vllm/model_executor/layers/quantization/gptq.py:GPTQLinearMethod, awq.py:AWQLinearMethod,
    gptq_marlin.py."""

    def __init__(self, group_size: int, bits: int = 4, act_order: bool = False):
        self.group_size, self.bits, self.act_order = group_size, bits, act_order

    def create_weights(self, layer, in_per_rank: int, out_per_rank: int, dtype: torch.dtype) -> None:
        import torch.nn as nn
        from ..layers.linear import set_weight_attrs
        total_in = getattr(layer, "input_size", in_per_rank)
        G = self.group_size if self.group_size > 0 else total_in
        whole = G >= total_in                    # one group spans all of K: scales are not split by row-parallel TP
        g_per_rank = 1 if whole else in_per_rank // G
        if in_per_rank % 8 or (not whole and in_per_rank % G):
            raise ValueError(f"{getattr(layer, 'prefix', '')}: {in_per_rank} input features per rank do not "
                             f"divide into groups of {G}")
        if self.act_order and in_per_rank != total_in:
            raise ValueError(f"{getattr(layer, 'prefix', '')}: activation reordering with a row-parallel shard")
        layer.w4_group = in_per_rank if whole else G
        layer.wq_bits = self.bits
        if self.bits == 4:
            qw = nn.Parameter(torch.zeros(out_per_rank, in_per_rank // 8, dtype=torch.int32), requires_grad=False)
        else:
            qw = nn.Parameter(torch.zeros(out_per_rank, in_per_rank, dtype=torch.uint8), requires_grad=False)
        set_weight_attrs(qw, {"input_dim": 1, "output_dim": 0})
        layer.register_parameter("qweight", qw)
        for name in ("scales", "qzeros"):
            p = nn.Parameter(torch.zeros(out_per_rank, g_per_rank, dtype=dtype), requires_grad=False)
            set_weight_attrs(p, {"output_dim": 0} if whole else {"input_dim": 1, "output_dim": 0})
            layer.register_parameter(name, p)
        if self.act_order:
            # identity until a checkpoint tensor arrives (a layer of the file may be in natural order)
            p = nn.Parameter(torch.arange(in_per_rank, dtype=torch.int32), requires_grad=False)
            set_weight_attrs(p, {})
            layer.register_parameter("perm", p)

    def process_weights_after_loading(self, layer) -> None:
        # (scale, zero) pairs per group and row, the layout the tiled kernel copies into LDS
        layer.wq_sz = torch.stack([layer.scales.data.t(), layer.qzeros.data.t()], dim=-1).contiguous()
        perm = getattr(layer, "perm", None)
        if perm is not None and bool((perm.data.cpu() == torch.arange(perm.numel(), dtype=torch.int32)).all()):
            layer.perm = None                    # natural order after all: no gather

    def apply(self, layer, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        from .. import ops
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        perm = getattr(layer, "perm", None)
        if perm is not None:
            x2 = ops.permute_cols(x2, perm)
        sz = getattr(layer, "wq_sz", None)
        if sz is None:
            self.process_weights_after_loading(layer)
            sz = layer.wq_sz
        y = ops.wq_linear(x2, layer.qweight, layer.scales, layer.qzeros, sz, layer.w4_group, layer.wq_bits, bias)
        return y.view(*shape[:-1], -1)


Int4LinearMethod = PackedIntLinearMethod


class LoadTimeDequant:
    """What the loader needs to know about a weight-only format.  4-bit GPTQ / AWQ linear layers without
    activation reordering keep their 4-bit weights (repacked for the W4A16 kernel, ``Int4LinearMethod``);
    everything else (2/8-bit, act-order, expert weights, block fp8, microscaling fp4) is expanded to 16 bit
    while loading, so those layers are ordinary ones.  Microscaling fp4 checkpoints calibrated for 4-bit
    activations run with 16-bit activations here (``quant_dequant_mxfp4`` is the rounding they assume)."""

    def __init__(self, method: str, cfg: dict, tp_size: int = 1):
        self.method = method
        self.tp_size = tp_size
        self.bits = int(cfg.get("bits", cfg.get("w_bit", 4)) or 4)
        self.block = tuple(cfg.get("weight_block_size") or (128, 128))
        # AutoGPTQ v1 files store zero - 1
        self.zero_offset = 0 if str(cfg.get("checkpoint_format", "gptq")) == "gptq_v2" else 1
        if method in ("gptq", "awq") and self.bits not in (2, 4, 8):
            raise ValueError(f"{method}: {self.bits}-bit weights are not supported (2, 4, 8)")
        from .. import envs
        self.group_size = int(cfg.get("group_size", cfg.get("q_group_size", 128)) or 128)
        self.skip = [str(x).strip("*") for x in (cfg.get("modules_to_not_convert") or cfg.get("exclude")
                                                 or cfg.get("exclude_modules") or cfg.get("llm_int8_skip_modules") or [])]
        # activation reordering gathers whole-K activations: kept packed only without tensor parallelism (a
        # row-parallel shard would need every group's scales); 8-bit GPTQ stays 8 bit
        self.act_order = bool(cfg.get("desc_act", False)) and method == "gptq"
        self.pack4 = (method in ("gptq", "awq") and (self.bits == 4 or (self.bits == 8 and method == "gptq"))
                      and (not self.act_order or tp_size == 1)
                      and envs.VDA_INT4_WEIGHTS and (self.group_size == -1 or self.group_size % 32 == 0))
        if method == "mxfp4" and envs.VDA_INT4_WEIGHTS:
            # linear layers keep their codes as exact 8-bit integers with power-of-two group scales (mxfp4_to_packed8)
            self.pack4, self.bits, self.group_size = True, 8, MX_BLOCK
        # 4-bit code-book formats: linear layers keep codes + fp32 group scales (layers/quant.py::CodeBook4LinearMethod)
        self.codebook = None                     # (group, first element in the high nibble)
        if envs.VDA_INT4_WEIGHTS and method == "nvfp4":
            self.pack4, self.codebook = True, (16, False)
        if envs.VDA_INT4_WEIGHTS and method == "bitsandbytes" and not cfg.get("load_in_8bit"):
            self.pack4, self.codebook = True, (BNB_BLOCK, True)

    def _packed(self, prefix: str) -> bool:
        return self.pack4 and ".experts." not in prefix and not prefix.endswith("lm_head") \
            and not any(s and s in prefix for s in self.skip)

    def get_quant_method(self, layer, prefix):
        from ..layers.linear import LinearBase
        if isinstance(layer, LinearBase) and self._packed(prefix):
            if self.codebook is not None:
                from ..layers.quant import CodeBook4LinearMethod
                return CodeBook4LinearMethod(*self.codebook)
            return PackedIntLinearMethod(self.group_size, self.bits, self.act_order)
        return None

    def _wrap_bnb(self, weights, dtype: torch.dtype) -> Iterator[tuple[str, torch.Tensor]]:
        """bitsandbytes checkpoints: ``X.weight`` (packed uint8 or int8) travels with ``X.weight.absmax``,
        ``.quant_map``, ``.nested_absmax``, ``.nested_quant_map``, ``.quant_state.bitsandbytes__{nf4,fp4}``
        (a JSON blob) or, for 8-bit, ``X.SCB``.  Tensors of one weight may arrive in any order."""
        import json
        pending: dict[str, dict] = {}

        def try_finish(base: str):
            d = pending[base]
            if "weight" not in d:
                return None
            if d["weight"].dtype == torch.int8:
                return dequantize_bnb8(d["weight"], d["SCB"], dtype) if "SCB" in d else None
            st = d.get("state")
            if st is None or "absmax" not in d or "quant_map" not in d:
                return None
            nested = "nested_blocksize" in st or "nested_offset" in st
            if nested and not ("nested_absmax" in d and "nested_quant_map" in d):
                return None
            if self.codebook is not None and self._packed(base):
                return bnb4_to_codebook(d["weight"], d["absmax"], d["quant_map"], st, d.get("nested_absmax"),
                                        d.get("nested_quant_map"), base)
            return dequantize_bnb4(d["weight"], d["absmax"], d["quant_map"], st, d.get("nested_absmax"),
                                   d.get("nested_quant_map"), dtype)

        for name, t in weights:
            base = leaf = None
            if ".weight." in name:
                base, _, leaf = name.partition(".weight.")
            elif name.endswith(".weight_format"):
                continue                              # layout tag of LLM.int8() weights: row-major is the only one read
            elif name.endswith(".SCB"):
                base, _, leaf = name.rpartition(".")
            elif name.endswith(".weight") and t.dtype in (torch.uint8, torch.int8):
                base, leaf = name[: -len(".weight")], "weight"
            if base is None:
                yield name, t
                continue
            d = pending.setdefault(base, {})
            if leaf.startswith("quant_state."):
                d["state"] = json.loads(bytes(t.view(-1).to(torch.uint8).tolist()).decode("utf-8"))
            else:
                d[leaf] = t
            w = try_finish(base)
            if w is not None:
                del pending[base]
                if isinstance(w, dict):
                    for k, v in w.items():
                        yield f"{base}.{k}", v
                else:
                    yield f"{base}.weight", w
        if pending:
            raise ValueError(f"incomplete bitsandbytes tensors for {sorted(pending)[:4]}")

    def wrap(self, weights: Iterable[tuple[str, torch.Tensor]], dtype: torch.dtype) -> Iterator[tuple[str, torch.Tensor]]:
        if self.method == "bitsandbytes":
            yield from self._wrap_bnb(weights, dtype)
            return
        pending: dict[str, dict[str, torch.Tensor]] = {}
        parts = ("qweight", "qzeros", "scales", "g_idx", "weight_scale_inv")
        mx = self.method == "mxfp4"
        nv = self.method == "nvfp4"

        def finish(prefix: str, d: dict):
            if nv:
                if all(k in d for k in ("weight", "weight_scale", "weight_scale_2")):
                    if self._packed(prefix):
                        return nvfp4_to_codebook(d["weight"], d["weight_scale"], d["weight_scale_2"])
                    return dequantize_nvfp4(d["weight"], d["weight_scale"], d["weight_scale_2"], dtype)
                return None
            if mx:
                if "weight" in d and "weight_scale" in d:
                    if self._packed(prefix):
                        return mxfp4_to_packed8(d["weight"], d["weight_scale"], dtype)
                    return dequantize_mxfp4(d["weight"], d["weight_scale"], dtype)
                return None
            if self.method == "fp8":
                if "weight" in d and "weight_scale_inv" in d:
                    return dequantize_block_fp8(d["weight"], d["weight_scale_inv"], self.block, dtype)
                return None
            if not all(k in d for k in ("qweight", "qzeros", "scales")):
                return None
            if self._packed(prefix):
                if self.method == "awq":
                    qw, sc, z = awq_to_packed4(d["qweight"], d["qzeros"], d["scales"], dtype)
                    return {"qweight": qw, "scales": sc, "qzeros": z}
                return gptq_to_packed(d["qweight"], d["qzeros"], d["scales"], d.get("g_idx") if self.act_order else None,
                                      self.bits, self.zero_offset, dtype)
            if self.method == "awq":
                return dequantize_awq(d["qweight"], d["qzeros"], d["scales"], self.bits, dtype)
            return dequantize_gptq(d["qweight"], d["qzeros"], d["scales"], d.get("g_idx"), self.bits,
                                   self.zero_offset, dtype)

        def emit(prefix: str, w):
            if isinstance(w, dict):
                for leaf, t in w.items():
                    yield f"{prefix}.{leaf}", t
            else:
                yield f"{prefix}.weight", w

        for name, t in weights:
            prefix, _, leaf = name.rpartition(".")
            held = leaf in parts or (self.method == "fp8" and leaf == "weight" and t.dtype == torch.float8_e4m3fn)
            if mx:      # excluded layers keep 16-bit weights and pass through
                held = leaf in ("weight", "weight_scale") and t.dtype == torch.uint8
            if nv:
                if leaf == "input_scale":       # calibrated for 4-bit activations; activations stay 16-bit here
                    continue
                held = (leaf == "weight" and t.dtype == torch.uint8) or leaf == "weight_scale_2" \
                    or (leaf == "weight_scale" and t.dtype == torch.float8_e4m3fn)
            if not held:
                yield name, t
                continue
            d = pending.setdefault(prefix, {})
            d[leaf] = t
            # GPTQ: g_idx may come last, so integer formats are completed at the end unless it is here
            ready = self.method in ("fp8", "awq", "mxfp4", "nvfp4") or "g_idx" in d
            if ready:
                w = finish(prefix, d)
                if w is not None:
                    del pending[prefix]
                    yield from emit(prefix, w)
        for prefix, d in pending.items():
            w = finish(prefix, d)
            if w is None:
                raise ValueError(f"{prefix}: incomplete {self.method} tensors in the checkpoint ({sorted(d)})")
            yield from emit(prefix, w)
