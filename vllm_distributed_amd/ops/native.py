"""ctypes binding of libvda_ops.so (HIP kernels with a C ABI).

Kernels are launched on PyTorch's current HIP stream, so they order with
PyTorch work and are captured by ``torch.cuda.graph`` like any other launch.
On a machine with a GPU a missing library is a hard error: there is no silent
fallback to PyTorch ops on the device path.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_long, c_void_p
from typing import Optional

import torch

from .. import envs

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[ctypes.CDLL] = None

BF16, F16, F32, FP8 = 0, 1, 2, 3
_DT = {torch.bfloat16: BF16, torch.float16: F16, torch.float32: F32, torch.float8_e4m3fn: FP8}

ACT = {"silu": 0, "gelu": 1, "gelu_tanh": 2, "gelu_pytorch_tanh": 2, "relu": 3, "gelu_new": 4,
       "gelu_fast": 5, "quick_gelu": 6, "gelu_quick": 6, "fatrelu": 7, "relu2": 8}

P, I, L, F = c_void_p, c_int, c_long, c_float
_SIGS = {
    "vda_rms_norm": [P, P, P, P, F, I, I, L, L, L, F, I, P, P, P, I, P],
    "vda_layer_norm": [P, P, P, P, F, I, I, L, L, I, P],
    "vda_act_and_mul": [P, P, L, I, L, L, I, I, F, P, I, P],
    "vda_activation": [P, P, L, I, F, I, P],
    "vda_rotary_embedding": [P, P, P, P, I, I, I, I, I, L, L, I, I, P],
    "vda_reshape_and_cache": [P, P, P, P, P, I, I, I, I, L, L, I, I, F, F, P],
    "vda_rope_and_cache": [P, P, P, P, P, P, I, I, I, I, I, I, L, I, I, I, F, F, P],
    "vda_copy_blocks": [P, P, I, L, P],
    "vda_transfer_blocks": [P, P, P, P, I, L, P],
    "vda_transfer_blocks_ex": [P, P, P, P, I, L, L, L, L, L, P],
    "vda_gather_cache": [P, P, P, P, P, I, I, I, I, I, P],
    "vda_paged_attention_decode": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, L, L, F, F, I, I, I, F, F, P],
    "vda_paged_attention_prefill": [P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, L, L, F, F, I, I, I, F, F, I, P,
                                    I, P],
    "vda_paged_attention_prefill_sparse": [P] * 13 + [I] * 9 + [L, L, F, F, I, I, F, F, I, P, L, P],
    "vda_set_prefill_variant": [I],
    "vda_int8_quant_azp": [P, P, P, P, P, P, I, I, L, L, I, P],
    "vda_permute_cols": [P, P, P, I, I, L, L, P],
    "vda_set_prefill_wide_min_wgs": [L],
    "vda_merge_attn_states": [P, P, P, P, P, P, I, I, I, I, P],
    "vda_argmax": [P, P, I, I, L, I, P],
    "vda_sample": [P, P, P, P, P, P, P, I, I, L, P],
    "vda_topk_topp_mask": [P, P, P, P, I, I, L, P],
    "vda_apply_penalties": [P, P, P, P, P, P, I, I, L, P],
    "vda_scaled_fp8_quant": [P, P, P, L, I, I, P],
    "vda_per_token_fp8_quant": [P, P, P, P, I, I, L, L, I, P],
    "vda_int8_quant": [P, P, P, P, I, I, L, L, I, P],
    "vda_fp8_dequant": [P, P, P, L, I, P],
    "vda_skinny_gemm": [P, P, P, P, P, I, I, I, L, L, L, I, I, P],
    "vda_moe_topk_softmax": [P, P, P, I, I, I, I, L, I, P],
    "vda_moe_align": [P, P, P, I, I, P],
    "vda_moe_gate_up": [P, P, P, P, P, P, I, I, I, I, I, L, I, I, I, I, P],
    "vda_moe_down": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "vda_moe_sum": [P, P, P, I, I, I, L, I, P],
    "vda_ep_prep": [P, I, I, I, P, P, P, P, P, P],
    "vda_moe_tile_map": [P, P, P, P, P, I, I, I, P],
    "vda_moe_align_large": [P, P, P, P, P, I, I, P],
    "vda_tile_gemm": [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, L, L, L, I, I, I, I, I, I, P],
    "vda_decode_gemm": [P, P, P, P, P, L, I, I, I, L, L, I, P],
    "vda_wq_gemm": [P, P, P, P, P, P, L, I, I, I, I, I, L, L, I, I, I, P],
    "vda_dense_gemm_stamp": [P, P, P, I, I, I, I, P],
    "vda_dense_gemm": [P, P, P, P, P, P, L, I, I, I, L, L, L, I, I, I, I, I, P],
    "vda_per_group_fp8_quant": [P, P, P, I, I, I, L, L, I, P],
    "vda_w4_skinny_gemm": [P, P, P, P, P, P, I, I, I, I, L, L, I, P],
    "vda_w4_dequant": [P, P, P, P, I, I, I, I, P],
    "vda_cb4_skinny_gemm": [P, P, P, P, P, P, I, I, I, I, I, L, L, I, P],
    "vda_cb4_dequant": [P, P, P, P, I, I, I, I, I, P],
    "vda_cb4_skinny_gemm_post": [P, P, P, P, P, P, I, I, I, I, I, L, L, I, P, L, I, P],
    "vda_cb4_dequant_post": [P, P, P, P, I, I, I, I, I, P],
    "vda_w8_dequant": [P, P, P, P, I, I, I, I, P],
    "vda_mla_cache_write": [P, P, P, P, P, I, I, I, I, L, L, I, I, F, P],
    "vda_mla_attention": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, L, L, F, I, I, F, P],
    "vda_mla_supported": [I, I],
    "vda_lora_shrink": [P, P, P, P, I, I, I, L, I, P],
    "vda_lora_expand": [P, P, P, P, I, I, I, L, I, P],
    "vda_causal_conv1d": [P, P, P, P, P, P, P, P, L, I, I, I, I, L, L, I, I, P],
    "vda_lightning_decode": [P, P, P, P, P, P, L, I, I, I, L, I, P],
    "vda_selective_scan": [P, P, P, P, P, P, P, P, P, P, P, P, P, L, I, I, I, L, L, L, L, L, L, I, I, I, I, I, I, P, P,
                           I, P],
}


def library_path() -> str:
    return envs.VDA_OPS_LIBRARY or os.path.join(_HERE, "libvda_ops.so")


def is_available() -> bool:
    return os.path.exists(library_path())


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -m vllm_distributed_amd.ops.build` "
                "(the HIP kernels are mandatory on the GPU path)")
        _LIB = ctypes.CDLL(path)
        for name, sig in _SIGS.items():
            fn = getattr(_LIB, name, None)
            if fn is None:
                continue       # optional symbol (e.g. MoE kernels are registered by their module)
            fn.argtypes = sig
            fn.restype = c_int
    return _LIB


def register(name: str, sig: list) -> None:
    _SIGS[name] = sig
    if _LIB is not None:
        fn = getattr(_LIB, name)
        fn.argtypes = sig
        fn.restype = c_int


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def dt(t: torch.dtype) -> int:
    try:
        return _DT[t]
    except KeyError:
        raise TypeError(f"dtype {t} is not supported by the HIP kernels") from None


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def call(name: str, *args) -> None:
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed with code {rc}")
