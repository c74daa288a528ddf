"""Tensor-level op API.  Every function takes torch tensors; CUDA(HIP) tensors
run the hand-written gfx950 kernels of ops/csrc, CPU tensors run the PyTorch
reference (ops/reference.py).  Role of the reference's vllm/_custom_ops.py.

KV cache layout used throughout:
    k_cache [num_blocks, kv_heads, block_size, head_size]
    v_cache [num_blocks, kv_heads, head_size, block_size]
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import envs
from ..logger import init_logger
from . import native as N
from . import reference as _reference

logger = init_logger(__name__)


class _ReferenceOps:
    """``ops/reference.py`` behind a tripwire: the PyTorch forms are the CPU backend and the oracle of the tests, but
    a GPU tensor landing in one of them means a shape or dtype the HIP kernel does not take, silently orders of
    magnitude slower.  The first such call of every op is logged (once per op and process)."""

    def __init__(self, mod):
        self._mod, self._seen = mod, set()

    def __getattr__(self, name):
        fn = getattr(self._mod, name)
        if not callable(fn):
            return fn

        def call(*a, **k):
            if name not in self._seen and not envs.VDA_FORCE_TORCH_OPS:
                t = next((v for v in list(a) + list(k.values()) if isinstance(v, torch.Tensor)), None)
                if t is not None and t.is_cuda:
                    self._seen.add(name)
                    shapes = [tuple(v.shape) for v in a if isinstance(v, torch.Tensor)][:4]
                    logger.warning("ops.%s: no HIP kernel takes this call (tensor shapes %s, %s): it runs as plain "
                                   "PyTorch ops on the GPU, far below the kernel's speed", name, shapes, t.dtype)
            return fn(*a, **k)
        return call


R = _ReferenceOps(_reference)


def _dev(x: torch.Tensor) -> bool:
    return x.is_cuda and not envs.VDA_FORCE_TORCH_OPS


def _contig_last(x: torch.Tensor) -> torch.Tensor:
    return x if x.stride(-1) == 1 else x.contiguous()


# ---- normalisation -----------------------------------------------------------
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float, residual: Optional[torch.Tensor] = None,
             weight_bias: float = 0.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = rmsnorm(x [+ residual]) * (weight + weight_bias); residual is updated in place."""
    if not _dev(x):
        y = R.rms_norm(x, weight, eps, residual, weight_bias)
        if out is not None:
            out.copy_(y)
            return out
        return y
    H = x.shape[-1]
    x2 = _contig_last(x).view(-1, H) if x.dim() != 2 else _contig_last(x)
    if out is None:
        out = torch.empty(x2.shape, dtype=x.dtype, device=x.device)
    o2 = out.view(-1, H)
    r2 = residual.view(-1, H) if residual is not None else None
    N.call("vda_rms_norm", o2.data_ptr(), x2.data_ptr(), N.ptr(r2), weight.data_ptr(), eps, x2.shape[0], H,
           x2.stride(0), r2.stride(0) if r2 is not None else 0, o2.stride(0), weight_bias, 0, None, None, None,
           N.dt(x.dtype), N.stream())
    return out.view(x.shape)


def rms_norm_fp8(x, weight, eps, residual=None, static_scale: Optional[torch.Tensor] = None,
                 scale_ub: Optional[torch.Tensor] = None, weight_bias: float = 0.0):
    """RMSNorm with fp8 output.  static_scale given -> per-tensor; else dynamic per-token.
    Returns (fp8 tensor, scales)."""
    H = x.shape[-1]
    if not _dev(x):
        y = R.rms_norm(x, weight, eps, residual, weight_bias)
        return R.scaled_fp8_quant(y, static_scale) if static_scale is not None else R.per_token_fp8_quant(y, scale_ub)
    x2 = _contig_last(x).view(-1, H)
    out = torch.empty(x2.shape, dtype=torch.float8_e4m3fn, device=x.device)
    r2 = residual.view(-1, H) if residual is not None else None
    if static_scale is not None:
        scales, quant = static_scale, 1
    else:
        scales, quant = torch.empty((x2.shape[0], 1), dtype=torch.float32, device=x.device), 2
    N.call("vda_rms_norm", out.data_ptr(), x2.data_ptr(), N.ptr(r2), weight.data_ptr(), eps, x2.shape[0], H,
           x2.stride(0), r2.stride(0) if r2 is not None else 0, out.stride(0), weight_bias, quant,
           scales.data_ptr() if quant == 1 else None, scales.data_ptr() if quant == 2 else None,
           N.ptr(scale_ub), N.dt(x.dtype), N.stream())
    return out.view(x.shape), scales


def layer_norm(x, weight, bias, eps):
    if not _dev(x):
        return R.layer_norm(x, weight, bias, eps)
    H = x.shape[-1]
    x2 = _contig_last(x).view(-1, H)
    out = torch.empty_like(x2)
    N.call("vda_layer_norm", out.data_ptr(), x2.data_ptr(), weight.data_ptr(), N.ptr(bias), eps, x2.shape[0], H,
           x2.stride(0), out.stride(0), N.dt(x.dtype), N.stream())
    return out.view(x.shape)


# ---- activations ----------------------------------------------------------------
def act_and_mul(x, kind: str = "silu", act_first: bool = True, threshold: float = 0.0,
                out: Optional[torch.Tensor] = None, fp8_scale: Optional[torch.Tensor] = None):
    d = x.shape[-1] // 2
    if not _dev(x):
        y = R.act_and_mul(x, kind, act_first, threshold)
        if fp8_scale is not None:
            y = R.scaled_fp8_quant(y, fp8_scale)[0]
        return y
    x2 = _contig_last(x).view(-1, 2 * d)
    if out is None:
        out = torch.empty((x2.shape[0], d), device=x.device,
                          dtype=torch.float8_e4m3fn if fp8_scale is not None else x.dtype)
    N.call("vda_act_and_mul", out.data_ptr(), x2.data_ptr(), x2.shape[0], d, x2.stride(0), out.stride(0),
           N.ACT[kind], int(act_first), threshold, N.ptr(fp8_scale), N.dt(x.dtype), N.stream())
    return out.view(*x.shape[:-1], d)


def silu_and_mul(x, out=None):
    return act_and_mul(x, "silu", out=out)


def activation(x, kind: str, threshold: float = 0.0):
    if not _dev(x) or x.numel() % 8:
        return R.activation(x, kind, threshold)
    xc = x.contiguous()
    out = torch.empty_like(xc)
    N.call("vda_activation", out.data_ptr(), xc.data_ptr(), xc.numel(), N.ACT[kind], threshold, N.dt(x.dtype),
           N.stream())
    return out


# ---- rotary + cache -----------------------------------------------------------------
def rotary_embedding(positions, q, k, cos_sin_cache, rot_dim: int, head_size: int, is_neox: bool = True):
    """In place.  q: [T, Hq*D] (row stride arbitrary), k: [T, Hkv*D] or None."""
    if not _dev(q):
        return R.rotary_embedding(positions, q, k, cos_sin_cache, rot_dim, head_size, is_neox)
    T = q.shape[0]
    N.call("vda_rotary_embedding", positions.data_ptr(), q.data_ptr(), N.ptr(k), cos_sin_cache.data_ptr(), T,
           rot_dim, head_size, q.shape[-1] // head_size, (k.shape[-1] // head_size) if k is not None else 0,
           q.stride(0), k.stride(0) if k is not None else 0, int(is_neox), N.dt(q.dtype), N.stream())


def _kv_fp8(cache: torch.Tensor) -> int:
    return int(cache.dtype == torch.float8_e4m3fn)


def reshape_and_cache(k, v, k_cache, v_cache, slot_mapping, k_scale: float = 1.0, v_scale: float = 1.0):
    """k, v: [T, Hkv, D] (token stride arbitrary, heads contiguous).  fp8 caches store x / scale."""
    if not _dev(k):
        return R.reshape_and_cache(k, v, k_cache, v_cache, slot_mapping, k_scale, v_scale)
    T, Hkv, D = k.shape
    N.call("vda_reshape_and_cache", k.data_ptr(), v.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
           slot_mapping.data_ptr(), T, Hkv, D, k_cache.shape[2], k.stride(0), v.stride(0), N.dt(k.dtype),
           _kv_fp8(k_cache), k_scale, v_scale, N.stream())


def rope_and_cache(positions, qkv, cos_sin_cache, k_cache, v_cache, slot_mapping, rot_dim: int, head_size: int,
                   num_heads: int, num_kv_heads: int, is_neox: bool = True, k_scale: float = 1.0,
                   v_scale: float = 1.0):
    """Fused RoPE(q, k) in place + write of k, v into the paged cache.
    qkv: [T, (Hq + 2*Hkv) * D]."""
    T = qkv.shape[0]
    if not _dev(qkv):
        qs, ks = num_heads * head_size, num_kv_heads * head_size
        q, k, v = qkv[:, :qs], qkv[:, qs:qs + ks], qkv[:, qs + ks:]
        R.rotary_embedding(positions, q, k, cos_sin_cache, rot_dim, head_size, is_neox)
        R.reshape_and_cache(k.reshape(T, num_kv_heads, head_size), v.reshape(T, num_kv_heads, head_size),
                            k_cache, v_cache, slot_mapping, k_scale, v_scale)
        return
    N.call("vda_rope_and_cache", positions.data_ptr(), qkv.data_ptr(), cos_sin_cache.data_ptr(),
           k_cache.data_ptr(), v_cache.data_ptr(), slot_mapping.data_ptr(), T, rot_dim, head_size, num_heads,
           num_kv_heads, k_cache.shape[2], qkv.stride(0), int(is_neox), N.dt(qkv.dtype), _kv_fp8(k_cache), k_scale,
           v_scale, N.stream())


def copy_blocks(cache: torch.Tensor, pairs: torch.Tensor):
    """cache: [num_blocks, ...]; pairs: int64 [n, 2] (src, dst) on the same device."""
    if not _dev(cache):
        return R.copy_blocks(cache, pairs)
    N.call("vda_copy_blocks", cache.data_ptr(), pairs.data_ptr(), pairs.shape[0],
           cache.stride(0) * cache.element_size(), N.stream())


def transfer_blocks(dst: torch.Tensor, src: torch.Tensor, dst_ids: torch.Tensor, src_ids: torch.Tensor):
    """dst[dst_ids[i]] = src[src_ids[i]] for whole blocks; the id tensors live on the launching device.
    src may be peer-mapped memory of another GPU (xGMI) or pinned host memory."""
    if not (dst.is_cuda or src.is_cuda) or envs.VDA_FORCE_TORCH_OPS:
        return R.transfer_blocks(dst, src, dst_ids, src_ids)
    N.call("vda_transfer_blocks", dst.data_ptr(), src.data_ptr(), dst_ids.data_ptr(), src_ids.data_ptr(),
           dst_ids.numel(), dst.stride(0) * dst.element_size(), N.stream())


def transfer_block_slices(dst: torch.Tensor, src: torch.Tensor, dst_ids: torch.Tensor, src_ids: torch.Tensor,
                          copy_bytes: int, dst_off: int = 0, src_off: int = 0):
    """Bytes [dst_off, dst_off + copy_bytes) of block dst_ids[i] of ``dst`` [blocks, bytes] (uint8 rows) come
    from bytes [src_off, ...) of block src_ids[i] of ``src``; the two sides may have different row lengths."""
    if not (dst.is_cuda or src.is_cuda) or envs.VDA_FORCE_TORCH_OPS:
        d, s_ = dst_ids.long().to(dst.device), src_ids.long().to(src.device)
        dst[d, dst_off:dst_off + copy_bytes] = src[s_, src_off:src_off + copy_bytes].to(dst.device)
        return
    assert dst.dtype == torch.uint8 and src.dtype == torch.uint8 and dst.stride(1) == 1 and src.stride(1) == 1
    N.call("vda_transfer_blocks_ex", dst.data_ptr(), src.data_ptr(), dst_ids.data_ptr(), src_ids.data_ptr(),
           dst_ids.numel(), copy_bytes, dst.stride(0), src.stride(0), dst_off, src_off, N.stream())


def gather_cache(k_cache, v_cache, slots):
    """Token-major copies [T, Hkv, D] of the K and V rows at the given slots."""
    if not _dev(k_cache):
        return R.gather_cache(k_cache, v_cache, slots)
    T = slots.numel()
    Hkv, D = k_cache.shape[1], k_cache.shape[3]
    ko = torch.empty((T, Hkv, D), dtype=k_cache.dtype, device=k_cache.device)
    vo = torch.empty_like(ko)
    N.call("vda_gather_cache", ko.data_ptr(), vo.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
           slots.data_ptr(), T, Hkv, D, k_cache.shape[2], N.dt(k_cache.dtype), N.stream())
    return ko, vo


# ---- attention --------------------------------------------------------------------
def decode_num_splits(num_seqs: int, num_kv_heads: int, max_seq_len: int) -> int:
    """Split-KV factor so that a decode launch has enough workgroups for 256 CUs."""
    wgs = max(num_seqs * num_kv_heads, 1)
    target = envs.VDA_DECODE_SPLIT_TARGET_WGS
    if wgs >= target:
        return 1
    by_len = max(max_seq_len // 256, 1)       # at least 256 keys per partition
    return max(1, min(target // wgs, by_len, 64))


def paged_attention_decode(out, q, k_cache, v_cache, block_table, seq_lens, q_start_loc, scale: float,
                           num_splits: int = 1, max_q_len: int = 1, window: int = 0, softcap: float = 0.0,
                           alibi: Optional[torch.Tensor] = None, workspace=None, k_scale: float = 1.0,
                           v_scale: float = 1.0):
    """q, out: [T, Hq, D].  Each sequence has <= max_q_len query tokens (1 for plain decode)."""
    T, Hq, D = q.shape
    if not _dev(q):
        out.copy_(R.paged_attention(q, k_cache, v_cache, block_table, seq_lens, q_start_loc, scale, window,
                                    softcap, alibi, k_scale=k_scale, v_scale=v_scale))
        return out
    S = seq_lens.shape[0]
    part_out = part_lse = None
    if num_splits > 1:
        if workspace is not None:
            part_out, part_lse = workspace
        else:
            part_out = torch.empty((T, Hq, num_splits, D), dtype=torch.float32, device=q.device)
            part_lse = torch.empty((T, Hq, num_splits), dtype=torch.float32, device=q.device)
    N.call("vda_paged_attention_decode", out.data_ptr(), q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
           block_table.data_ptr(), seq_lens.data_ptr(), q_start_loc.data_ptr(), N.ptr(alibi), N.ptr(part_out),
           N.ptr(part_lse), S, T, Hq, k_cache.shape[1], D, k_cache.shape[2], block_table.stride(0), num_splits,
           max_q_len, q.stride(0), out.stride(0), scale, softcap, window, N.dt(q.dtype), _kv_fp8(k_cache), k_scale,
           v_scale, N.stream())
    return out


def paged_attention_prefill(out, q, k_cache, v_cache, block_table, seq_lens, q_start_loc, scale: float,
                            window: int = 0, softcap: float = 0.0, alibi: Optional[torch.Tensor] = None,
                            k_scale: float = 1.0, v_scale: float = 1.0, causal: bool = True,
                            return_lse: bool = False, max_q_len: int = 0):
    """Varlen attention of new tokens against the paged cache.  ``causal=False``: every query token sees
    all ``seq_lens[s]`` keys of its sequence (encoder models, the shared-prefix pass of cascade
    attention).  ``return_lse``: also return the log-sum-exp [Hq, T] (natural log) of every row.
    ``max_q_len``: the longest query run of the batch when the caller knows it on the host (0 = unknown);
    the wide kernel then starts the longest causal tiles of all sequences first."""
    T, Hq, D = q.shape
    if not _dev(q):
        o, lse = R.paged_attention(q, k_cache, v_cache, block_table, seq_lens, q_start_loc, scale, window,
                                   softcap, alibi, return_lse=True, k_scale=k_scale, v_scale=v_scale,
                                   causal=causal)
        out.copy_(o)
        return (out, lse) if return_lse else out
    lse = torch.full((Hq, T), float("-inf"), dtype=torch.float32, device=q.device) if return_lse else None
    N.call("vda_paged_attention_prefill", out.data_ptr(), q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
           block_table.data_ptr(), seq_lens.data_ptr(), q_start_loc.data_ptr(), N.ptr(alibi), seq_lens.shape[0], T,
           Hq, k_cache.shape[1], D, k_cache.shape[2], block_table.stride(0), q.stride(0), out.stride(0), scale,
           softcap, window, N.dt(q.dtype), _kv_fp8(k_cache), k_scale, v_scale, int(bool(causal)), N.ptr(lse),
           int(max_q_len), N.stream())
    return (out, lse) if return_lse else out


def paged_attention_prefill_sparse(out, q, k_cache, v_cache, block_table, seq_lens, q_start_loc, scale: float, vert,
                                   vcount, slash, scount, key_base, q_base, causal: bool = True, softcap: float = 0.0,
                                   k_scale: float = 1.0, v_scale: float = 1.0, max_keys: int = 0):
    """Vertical-slash sparse prefill over the paged cache: per (sequence, head) a list of key positions ``vert``
    [S, Hq, NV] (counts ``vcount`` [S, Hq]) and a list of query-minus-key distances ``slash`` [S, Hq, NS]
    (``scount``), int32, in absolute positions of the request; ``key_base`` / ``q_base`` [S] give the absolute position
    of key 0 of the sequence's table and of its first query row.  A 128-row query tile visits the 32-key steps that
    hold a listed position or that a listed diagonal crosses inside the tile.  Returns (out, lse [Hq, T])."""
    T, Hq, D = q.shape
    if not _dev(q):
        o, lse = R.paged_attention_sparse(q, k_cache, v_cache, block_table, seq_lens, q_start_loc, scale, vert, vcount,
                                          slash, scount, key_base, q_base, causal, softcap, k_scale, v_scale)
        out.copy_(o)
        return out, lse
    S = seq_lens.shape[0]
    for t in (vert, vcount, slash, scount, key_base, q_base):
        assert t.dtype == torch.int32 and t.is_contiguous() and t.device == q.device
    assert vert.shape[:2] == (S, Hq) and slash.shape[:2] == (S, Hq) and vcount.shape == (S, Hq) == scount.shape
    assert key_base.shape == (S,) == q_base.shape and q.stride(1) == D and out.stride(1) == D
    lse = torch.full((Hq, T), float("-inf"), dtype=torch.float32, device=q.device)
    N.call("vda_paged_attention_prefill_sparse", out.data_ptr(), q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
           block_table.data_ptr(), seq_lens.data_ptr(), q_start_loc.data_ptr(), vert.data_ptr(), vcount.data_ptr(),
           slash.data_ptr(), scount.data_ptr(), key_base.data_ptr(), q_base.data_ptr(), vert.shape[2], slash.shape[2],
           S, T, Hq, k_cache.shape[1], D, k_cache.shape[2], block_table.stride(0), q.stride(0), out.stride(0), scale,
           softcap, N.dt(q.dtype), _kv_fp8(k_cache), k_scale, v_scale, int(bool(causal)), lse.data_ptr(),
           int(max_keys or block_table.shape[1] * k_cache.shape[2]), N.stream())
    return out, lse


def merge_attn_states(prefix, prefix_lse, suffix, suffix_lse, return_lse: bool = False):
    if not _dev(prefix):
        o, l = R.merge_attn_states(prefix, prefix_lse, suffix, suffix_lse)
        return (o, l) if return_lse else o
    T, H, D = prefix.shape
    out = torch.empty_like(prefix)
    lse = torch.empty_like(prefix_lse) if return_lse else None
    N.call("vda_merge_attn_states", out.data_ptr(), N.ptr(lse), prefix.data_ptr(), prefix_lse.data_ptr(),
           suffix.data_ptr(), suffix_lse.data_ptr(), T, H, D, N.dt(prefix.dtype), N.stream())
    return (out, lse) if return_lse else out


# ---- multi-head latent attention ------------------------------------------------------------
def mla_supported(dk: int, dv: int, dtype: torch.dtype, on_device: bool) -> bool:
    """Latent sizes the MLA kernel is built for (DeepSeek-V2/V3: 576 / 512, plus two small ones)."""
    if not on_device or envs.VDA_FORCE_TORCH_OPS:
        return True
    return dtype in (torch.bfloat16, torch.float16) and (dk, dv) in ((576, 512), (96, 64), (160, 128))


def mla_cache_write(kv_c: torch.Tensor, k_pe: torch.Tensor, kc: torch.Tensor, vt: torch.Tensor,
                    slot_mapping: torch.Tensor, kv_scale: float = 1.0) -> None:
    """Write the normalised latent kv_c [T, R] and the rotated shared key k_pe [T, rope] of every token
    into the latent cache: kc [blocks, BS, R + rope] and the transposed copy vt [blocks, R, BS].  An e4m3 cache
    stores value / kv_scale."""
    if not _dev(kv_c):
        return R.mla_cache_write(kv_c, k_pe, kc, vt, slot_mapping, kv_scale)
    T, Rk = kv_c.shape
    assert kv_c.stride(1) == 1 and k_pe.stride(1) == 1 and kc.is_contiguous() and vt.is_contiguous()
    N.call("vda_mla_cache_write", kv_c.data_ptr(), k_pe.data_ptr(), kc.data_ptr(), vt.data_ptr(),
           slot_mapping.data_ptr(), T, Rk, k_pe.shape[1], kc.shape[1], kv_c.stride(0), k_pe.stride(0),
           N.dt(kv_c.dtype), int(kc.dtype == torch.float8_e4m3fn), float(kv_scale), N.stream())


def mla_attention(out: torch.Tensor, q: torch.Tensor, kc: torch.Tensor, vt: torch.Tensor, block_table: torch.Tensor,
                  seq_lens: torch.Tensor, q_start_loc: torch.Tensor, scale: float, max_query_len: int,
                  num_splits: int = 1, kv_scale: float = 1.0) -> torch.Tensor:
    """Causal multi-query attention of q [T, H, DK] over the latent cache; out [T, H, DV] stays in latent
    space (the caller applies the value up-projection).  Decode and (chunked) prefill are one path."""
    T, H, DK = q.shape
    DV = vt.shape[1]
    if not _dev(q):
        out.copy_(R.mla_attention(q, kc, vt, block_table, seq_lens, q_start_loc, scale, kv_scale))
        return out
    part_out = part_lse = None
    if num_splits > 1:
        part_out = torch.empty((T, H, num_splits, DV), dtype=torch.float32, device=q.device)
        part_lse = torch.empty((T, H, num_splits), dtype=torch.float32, device=q.device)
    assert q.stride(2) == 1 and q.stride(1) == DK and out.stride(2) == 1 and out.stride(1) == DV
    N.call("vda_mla_attention", out.data_ptr(), q.data_ptr(), kc.data_ptr(), vt.data_ptr(), block_table.data_ptr(),
           seq_lens.data_ptr(), q_start_loc.data_ptr(), N.ptr(part_out), N.ptr(part_lse), seq_lens.shape[0], T, H, DK,
           DV, kc.shape[1], block_table.stride(0), num_splits, max(int(max_query_len), 1), q.stride(0),
           out.stride(0), scale, N.dt(q.dtype), int(kc.dtype == torch.float8_e4m3fn), float(kv_scale), N.stream())
    return out


# ---- sampling --------------------------------------------------------------------------
def argmax(logits: torch.Tensor) -> torch.Tensor:
    if not _dev(logits):
        return logits.argmax(dim=-1)
    S, V = logits.shape
    out = torch.empty(S, dtype=torch.long, device=logits.device)
    N.call("vda_argmax", logits.data_ptr(), out.data_ptr(), S, V, logits.stride(0), N.dt(logits.dtype), N.stream())
    return out


def sample(logits: torch.Tensor, temperature=None, top_k=None, top_p=None, min_p=None, seeds=None,
           generator=None) -> torch.Tensor:
    """Draw one token per row of fp32 logits.  temperature < 1e-5 means greedy for that row."""
    if not _dev(logits):
        return R.sample(logits, temperature, top_k, top_p, min_p, seeds, generator)
    assert logits.dtype == torch.float32
    S, V = logits.shape
    out = torch.empty(S, dtype=torch.long, device=logits.device)
    N.call("vda_sample", logits.data_ptr(), out.data_ptr(), N.ptr(temperature), N.ptr(top_k), N.ptr(top_p),
           N.ptr(min_p), N.ptr(seeds), S, V, logits.stride(0), N.stream())
    return out


def topk_topp_mask(logits, top_k=None, top_p=None, min_p=None):
    if not _dev(logits):
        return R.topk_topp_mask(logits, top_k, top_p, min_p)
    S, V = logits.shape
    N.call("vda_topk_topp_mask", logits.data_ptr(), N.ptr(top_k), N.ptr(top_p), N.ptr(min_p), S, V,
           logits.stride(0), N.stream())
    return logits


def apply_penalties(logits, prompt_mask, out_counts, rep, freq, pres):
    """In place on fp32 logits [S, V]; prompt_mask bool/uint8 [S, V]; out_counts int32 [S, V]."""
    if not _dev(logits):
        return R.apply_penalties(logits, prompt_mask, out_counts, rep, freq, pres)
    S, V = logits.shape
    pm = prompt_mask.view(torch.uint8) if prompt_mask.dtype == torch.bool else prompt_mask
    N.call("vda_apply_penalties", logits.data_ptr(), pm.data_ptr(), out_counts.data_ptr(), rep.data_ptr(),
           freq.data_ptr(), pres.data_ptr(), S, V, logits.stride(0), N.stream())
    return logits


# ---- fp8 ---------------------------------------------------------------------------------
def scaled_fp8_quant(x: torch.Tensor, scale: Optional[torch.Tensor] = None):
    """Per-tensor fp8 (e4m3fn).  scale=None computes it (dynamic)."""
    if not _dev(x) or x.numel() % 8:
        return R.scaled_fp8_quant(x, scale)
    xc = x.contiguous()
    out = torch.empty(xc.shape, dtype=torch.float8_e4m3fn, device=x.device)
    mode = 0
    if scale is None:
        scale = torch.zeros(1, dtype=torch.float32, device=x.device)
        mode = 1
    N.call("vda_scaled_fp8_quant", out.data_ptr(), xc.data_ptr(), scale.data_ptr(), xc.numel(), mode,
           N.dt(x.dtype), N.stream())
    return out, scale


def per_token_fp8_quant(x: torch.Tensor, scale_ub: Optional[torch.Tensor] = None):
    if not _dev(x):
        return R.per_token_fp8_quant(x, scale_ub)
    H = x.shape[-1]
    x2 = _contig_last(x).view(-1, H)
    out = torch.empty(x2.shape, dtype=torch.float8_e4m3fn, device=x.device)
    scales = torch.empty((x2.shape[0], 1), dtype=torch.float32, device=x.device)
    N.call("vda_per_token_fp8_quant", out.data_ptr(), scales.data_ptr(), x2.data_ptr(), N.ptr(scale_ub),
           x2.shape[0], H, x2.stride(0), out.stride(0), N.dt(x.dtype), N.stream())
    return out.view(x.shape), scales.view(*x.shape[:-1], 1)


def int8_quant(x: torch.Tensor, scale: Optional[torch.Tensor] = None):
    """Symmetric int8 quantisation of the last dimension's rows: per-token dynamic scales
    ([..., 1] fp32, absmax / 127) when ``scale`` is None, the given scalar otherwise."""
    if not _dev(x) or x.shape[-1] % 8 or x.shape[-1] > 65536:
        return R.int8_quant(x, scale)
    H = x.shape[-1]
    x2 = _contig_last(x).view(-1, H)
    out = torch.empty(x2.shape, dtype=torch.int8, device=x.device)
    scales = torch.empty((x2.shape[0], 1), dtype=torch.float32, device=x.device) if scale is None else scale
    N.call("vda_int8_quant", out.data_ptr(), scales.data_ptr(), x2.data_ptr(), N.ptr(scale), x2.shape[0], H,
           x2.stride(0), out.stride(0), N.dt(x.dtype), N.stream())
    return out.view(x.shape), (scales.view(*x.shape[:-1], 1) if scale is None else scale)


def int8_quant_azp(x: torch.Tensor, scale: Optional[torch.Tensor] = None, azp: Optional[torch.Tensor] = None):
    """Asymmetric int8 quantisation of the last dimension's rows: q = clamp(round(x / scale) + azp).
    Dynamic per token when ``scale`` is None (scale = (max - min) / 255, azp = round(-128 - min / scale),
    both returned with a trailing dimension of 1), the given scalar scale / int32 zero point otherwise.
    Reference: vllm/_custom_ops.py scaled_int8_quant(symmetric=False)."""
    if not _dev(x) or x.shape[-1] % 8 or x.shape[-1] > 65536:
        return R.int8_quant_azp(x, scale, azp)
    H = x.shape[-1]
    x2 = _contig_last(x).view(-1, H)
    out = torch.empty(x2.shape, dtype=torch.int8, device=x.device)
    dyn = scale is None
    scales = torch.empty((x2.shape[0], 1), dtype=torch.float32, device=x.device) if dyn else scale
    azps = torch.empty((x2.shape[0], 1), dtype=torch.int32, device=x.device) if dyn else azp
    N.call("vda_int8_quant_azp", out.data_ptr(), scales.data_ptr(), N.ptr(azps), x2.data_ptr(), N.ptr(scale),
           N.ptr(azp), x2.shape[0], H, x2.stride(0), out.stride(0), N.dt(x.dtype), N.stream())
    if dyn:
        return out.view(x.shape), scales.view(*x.shape[:-1], 1), azps.view(*x.shape[:-1], 1)
    return out.view(x.shape), scale, azp


def permute_cols(x: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
    """``x[:, perm]`` for a 16-bit matrix [rows, cols] (perm int32 [cols]).  Reference: csrc/permute_cols.cu."""
    if not _dev(x) or x.element_size() != 2 or x.dim() != 2 or x.stride(1) != 1:
        return x[:, perm.long()]
    out = torch.empty((x.shape[0], perm.shape[0]), dtype=x.dtype, device=x.device)
    if out.stride(0) % 8:
        return x[:, perm.long()]
    N.call("vda_permute_cols", out.data_ptr(), x.data_ptr(), perm.to(torch.int32).contiguous().data_ptr(), x.shape[0],
           perm.shape[0], x.stride(0), out.stride(0), N.stream())
    return out


def convert_fp8(dst: torch.Tensor, src: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """Cache-dtype conversion in either direction (reference: csrc/cache_kernels.cu convert_fp8):
    16-bit / fp32 -> e4m3 stores ``src / scale``, e4m3 -> 16-bit / fp32 stores ``src * scale``."""
    f8 = torch.float8_e4m3fn
    s = torch.tensor([scale], dtype=torch.float32, device=src.device)
    if dst.dtype == f8 and src.dtype != f8:
        q, _ = scaled_fp8_quant(src.reshape(-1, src.shape[-1]), s)
        dst.copy_(q.view(dst.shape))
    elif src.dtype == f8 and dst.dtype != f8:
        dst.copy_(fp8_dequant(src.contiguous(), s, dst.dtype).view(dst.shape))
    else:
        raise ValueError(f"convert_fp8: exactly one side must be float8_e4m3fn, got {src.dtype} -> {dst.dtype}")
    return dst


def fp8_dequant(q: torch.Tensor, scale: torch.Tensor, dtype: torch.dtype):
    if not _dev(q) or q.numel() % 8:
        return R.fp8_dequant(q, scale, dtype)
    out = torch.empty(q.shape, dtype=dtype, device=q.device)
    N.call("vda_fp8_dequant", out.data_ptr(), q.contiguous().data_ptr(), scale.data_ptr(), q.numel(), N.dt(dtype),
           N.stream())
    return out


# ---- GEMM -----------------------------------------------------------------------------------
def skinny_gemm_ok(M: int, N_: int, K: int, dtype: torch.dtype, max_m: Optional[int] = None) -> bool:
    return (1 <= M <= min(max_m or envs.VDA_SKINNY_GEMM_MAX_M, 64) and N_ % 16 == 0 and K % 128 == 0
            and dtype in (torch.bfloat16, torch.float16) and not envs.VDA_DISABLE_SKINNY_GEMM)


def skinny_gemm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
                w_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x @ w.T (+ bias) for x [M<=64, K], w [N, K]; w may be e4m3 with a per-tensor ``w_scale``
    (weights widened in registers: half the HBM traffic of 16-bit weights)."""
    if not _dev(x):
        return R.skinny_gemm(x, w, bias, w_scale)
    M, K = x.shape
    N_ = w.shape[0]
    y = torch.empty((M, N_), dtype=x.dtype, device=x.device)
    N.call("vda_skinny_gemm", y.data_ptr(), x.data_ptr(), w.data_ptr(), N.ptr(bias), N.ptr(w_scale), M, N_, K,
           x.stride(0), w.stride(0), y.stride(0), N.dt(x.dtype), N.dt(w.dtype), N.stream())
    return y


_DECODE_WS: dict = {}


def decode_gemm_ok(M: int, N_: int, K: int, dtype: torch.dtype) -> bool:
    """Shapes the weight-streaming decode GEMM takes (and is meant for: the rows of a decode batch)."""
    return (envs.VDA_DECODE_GEMM_MIN_M <= M <= min(envs.VDA_DECODE_GEMM_MAX_M, 256) and K % 256 == 0 and N_ % 4 == 0
            and N_ >= 512 and dtype in (torch.bfloat16, torch.float16))


def decode_gemm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x @ w.T (+ bias) for x [M <= 256, K], w [N, K] contiguous: the weights stream once from HBM
    through registers into the MFMAs, K split over workgroups (ops/csrc/gemm_decode.hip)."""
    if not _dev(x):
        y = torch.nn.functional.linear(x, w, bias)
        if out is not None:
            out.copy_(y)
            return out
        return y
    M, K = x.shape
    N_ = w.shape[0]
    if w.shape[1] != K or not w.is_contiguous() or x.stride(1) != 1:
        raise ValueError("decode_gemm: x must be [M, K] with unit column stride and w [N, K] contiguous")
    if out is None:
        out = torch.empty((M, N_), dtype=x.dtype, device=x.device)
    ws = _DECODE_WS.get(x.device)
    if ws is None:          # split-K partial sums; one arena per device, reused by every call (stream order)
        ws = _DECODE_WS[x.device] = torch.empty(envs.VDA_DECODE_GEMM_WS_MB << 18, dtype=torch.float32, device=x.device)
    N.call("vda_decode_gemm", out.data_ptr(), x.data_ptr(), w.data_ptr(), N.ptr(bias), ws.data_ptr(),
           ws.numel() * 4, M, N_, K, x.stride(0), out.stride(0), N.dt(x.dtype), N.stream())
    return out


_DENSE_EPI = {"plain": 0, "gate_up": 1}


def dense_gemm_ok(x: torch.Tensor, w: torch.Tensor) -> bool:
    """Operands the ping-pong MFMA GEMM takes: 16-bit, K a multiple of 32, row strides multiples of 8 elements,
    both operands below 4 GiB (the kernel addresses them with 32-bit byte offsets)."""
    return (x.dim() == 2 and x.dtype in (torch.bfloat16, torch.float16) and w.dtype == x.dtype and w.is_contiguous()
            and x.stride(1) == 1 and x.shape[1] % 32 == 0 and x.stride(0) % 8 == 0 and x.shape[0] >= 1
            and w.numel() * 2 < (1 << 32) and x.shape[0] * x.stride(0) * 2 < (1 << 32))


def dense_gemm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
               residual: Optional[torch.Tensor] = None, epilogue: str = "plain",
               out: Optional[torch.Tensor] = None, wt: int = 0, splits: int = 0, variant: int = 0) -> torch.Tensor:
    """y = epilogue(x @ w.T) with the in-tree ping-pong MFMA kernel (ops/csrc/gemm_dense.hip), any M.
    ``plain``: y = x @ w.T (+ bias) (+ residual); ``gate_up``: w holds the gate rows followed by the up rows of a
    gated MLP and y = silu(x @ gate.T) * (x @ up.T) (no separate activation pass).  ``wt`` / ``splits`` override the
    tile height (2 or 4 = 128 or 256 weight rows), the K split and the kernel form (1 = ring of 32-element K units,
    2 = 64-element units on 256 x 256 tiles; 0 = chosen by the launcher)."""
    epi = _DENSE_EPI[epilogue]
    M, K = x.shape
    n_out = w.shape[0] // 2 if epi else w.shape[0]
    if not _dev(x):
        y = torch.nn.functional.linear(x, w, None if epi else bias)
        if epi:
            y = torch.nn.functional.silu(y[:, :n_out]) * y[:, n_out:]
        elif residual is not None:
            y = y + residual
        if out is not None:
            out.copy_(y)
            return out
        return y
    if not dense_gemm_ok(x, w) or w.shape[1] != K:
        raise ValueError("dense_gemm: x must be 16-bit [M, K] (K % 32 == 0), w [N, K] contiguous of the same dtype")
    if out is None:
        out = torch.empty((M, n_out), dtype=x.dtype, device=x.device)
    ws = _DECODE_WS.get(x.device)
    if ws is None:          # split-K partial sums; one arena per device, reused by every call (stream order)
        ws = _DECODE_WS[x.device] = torch.empty(envs.VDA_DECODE_GEMM_WS_MB << 18, dtype=torch.float32, device=x.device)
    N.call("vda_dense_gemm", out.data_ptr(), x.data_ptr(), w.data_ptr(), N.ptr(bias), N.ptr(residual), ws.data_ptr(),
           ws.numel() * 4, M, n_out, K, x.stride(0), out.stride(0), residual.stride(0) if residual is not None else 0,
           epi, N.dt(x.dtype), wt, splits, variant, N.stream())
    return out


def dense_gemm_wanted(x: torch.Tensor, w: torch.Tensor, max_n: Optional[int] = None) -> bool:
    """True when a 16-bit linear of this size takes the in-tree GEMM by default: decode batches of
    ``VDA_DENSE_GEMM_MIN_M`` .. ``VDA_DENSE_GEMM_MAX_M`` rows against weight matrices of up to ``VDA_DENSE_GEMM_MAX_N``
    rows (qkv / o / down projections, which need a K split to fill the chip).  There the in-tree kernel is 5 - 13 %
    faster per Llama-3-70B layer than the library with its offline selection table and up to 25 % at batch sizes the
    table does not list (profiles/dense_gemm_chain_r3.log); smaller batches, the wide gate/up matrix and prefill
    batches stay on the library GEMM."""
    return (x.is_cuda and envs.VDA_DENSE_GEMM_MIN_M <= x.shape[0] <= envs.VDA_DENSE_GEMM_MAX_M
            and 64 <= w.shape[0] <= (envs.VDA_DENSE_GEMM_MAX_N if max_n is None else max_n)
            and dense_gemm_ok(x, w) and w.shape[1] == x.shape[1])


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Unquantised linear: the in-tree ping-pong MFMA GEMM for decode-sized batches (skinny / split-K kernels when
    their switches are set), hipBLASLt otherwise."""
    if x.is_cuda and x.dim() == 2 and x.stride(-1) == 1 and w.stride(-1) == 1:
        M, K = x.shape
        if skinny_gemm_ok(M, w.shape[0], K, x.dtype):
            return skinny_gemm(x, w, bias)
        if decode_gemm_ok(M, w.shape[0], K, x.dtype) and w.dtype == x.dtype and w.is_contiguous() \
                and x.stride(0) % 8 == 0 and (bias is None or bias.dtype == x.dtype):
            return decode_gemm(x, w, bias)
        if dense_gemm_wanted(x, w) and (bias is None or (bias.dtype == x.dtype and w.shape[0] % 4 == 0)):
            return dense_gemm(x, w, bias)
    return torch.nn.functional.linear(x, w, bias)


def gate_up_silu_mul(x: torch.Tensor, w: torch.Tensor) -> Optional[torch.Tensor]:
    """silu(x @ gate.T) * (x @ up.T) for w = [gate rows; up rows] in ONE kernel (the activation runs in the GEMM
    epilogue, the [M, 2 I] intermediate never exists), or None when this batch does not take the in-tree GEMM."""
    if x.dim() != 2 or x.shape[0] > envs.VDA_DENSE_GATE_UP_MAX_M or w.shape[0] % 8 \
            or not dense_gemm_wanted(x, w, max_n=1 << 30):
        return None
    return dense_gemm(x, w, epilogue="gate_up")


# ---- 4-bit weight-only linear (GPTQ / AWQ) ------------------------------------------------------
def w4_pack(q: torch.Tensor) -> torch.Tensor:
    """Integer weights [N, K] with values 0..15 -> the kernel's packed form, int32 [N, K/8]: element e
    of every 8 sits in nibble e//2 + 4*(e%2), so ``(w >> 4 i) & 0x000F000F`` is the pair (2 i, 2 i + 1) as two
    16-bit fields, the layout of an MFMA operand register (see ops/csrc/w4.hip)."""
    N_, K = q.shape
    v = q.to(torch.int64).view(N_, K // 8, 8)
    word = torch.zeros((N_, K // 8), dtype=torch.int64, device=q.device)
    for e in range(8):
        word |= (v[:, :, e] & 0xF) << (4 * (e // 2 + 4 * (e % 2)))
    return word.to(torch.uint32).view(torch.int32) if hasattr(torch, "uint32") else word.to(torch.int32)


def w4_dequant(qw: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, group: int,
               dtype: torch.dtype) -> torch.Tensor:
    """Packed 4-bit weights -> [N, K] in ``dtype``: (q - zero) * scale per group of ``group`` K elements."""
    N_, K = qw.shape[0], qw.shape[1] * 8
    if not _dev(qw) or dtype not in (torch.bfloat16, torch.float16):
        return R.w4_dequant(qw, scales, zeros, group, dtype)
    out = torch.empty((N_, K), dtype=dtype, device=qw.device)
    N.call("vda_w4_dequant", out.data_ptr(), qw.data_ptr(), scales.data_ptr(), zeros.data_ptr(), N_, K, group,
           N.dt(dtype), N.stream())
    return out


def w4_linear(x: torch.Tensor, qw: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, group: int,
              bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = x @ dequant(qw).T (+ bias).  Decode-sized batches (``VDA_W4_SKINNY_MAX_M`` rows, 64 at most)
    stream the packed weights once through the MFMA skinny kernel (a quarter of the bytes of 16-bit
    weights); more rows expand them and use the library GEMM."""
    M, K = x.shape
    N_ = qw.shape[0]
    if _dev(x) and M <= envs.VDA_W4_SKINNY_MAX_M and N_ % 16 == 0 and K % 128 == 0 and group % 32 == 0 \
            and x.stride(1) == 1 \
            and x.dtype in (torch.bfloat16, torch.float16) and scales.dtype == x.dtype:
        y = torch.empty((M, N_), dtype=x.dtype, device=x.device)
        N.call("vda_w4_skinny_gemm", y.data_ptr(), x.data_ptr(), qw.data_ptr(), scales.data_ptr(), zeros.data_ptr(),
               N.ptr(bias), M, N_, K, group, x.stride(0), y.stride(0), N.dt(x.dtype), N.stream())
        return y
    return torch.nn.functional.linear(x, w4_dequant(qw, scales, zeros, group, x.dtype), bias)


# ---- 4-bit code-book weights (bitsandbytes NF4 / FP4, NVFP4) -------------------------------------
_CB4_WS: dict = {}


def cb4_chunk_layout_ok(K: int, group: int) -> bool:
    """Whether the kernels that scale accumulators instead of weights apply: whole 128-element chunks, and every run of
    32 consecutive elements inside one scale group."""
    return K % 128 == 0 and (group in (32, 64) or (group >= 128 and group % 128 == 0)) and K % group == 0


def cb4_to_chunk_layout(codes: torch.Tensor, scales: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """codes uint8 [N, K/2] in K order, scales [N, K/group] -> the layout of ``cb4_skinny_post_kernel``: the sixteen
    4-byte words of every 64-byte chunk re-ordered as a transposed 4 x 4 matrix (stored word 4 g + q = the codes of
    elements 32 q + 8 g .. + 7), scales transposed to [K/group, N]."""
    N_, K2 = codes.shape
    c = codes.view(N_, K2 // 64, 4, 4, 4).transpose(2, 3).contiguous().view(N_, K2)
    return c, scales.t().contiguous()


def cb4_from_chunk_layout(codes: torch.Tensor, scales_t: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    return cb4_to_chunk_layout(codes, scales_t)          # the word transpose is its own inverse, so is ``.t()``


def cb4_dequant(codes: torch.Tensor, scales: torch.Tensor, table: torch.Tensor, group: int, hi_first: bool,
                dtype: torch.dtype, chunked: bool = False) -> torch.Tensor:
    """Code-book weights -> [N, K] in ``dtype``: ``codes`` uint8 [N, K/2] (two per byte, the first element in the high
    nibble when ``hi_first``), ``scales`` fp32 [N, K/group], ``table`` fp32 [16]; value = table[code] * scale.
    ``chunked``: codes and scales are in the layout of ``cb4_to_chunk_layout``."""
    N_, K = codes.shape[0], codes.shape[1] * 2
    ok = _dev(codes) and dtype in (torch.bfloat16, torch.float16) and group % 8 == 0 \
        and codes.is_contiguous() and scales.is_contiguous() and scales.dtype == torch.float32
    if ok and chunked:
        out = torch.empty((N_, K), dtype=dtype, device=codes.device)
        N.call("vda_cb4_dequant_post", out.data_ptr(), codes.data_ptr(), scales.data_ptr(), table.data_ptr(), N_, K,
               group, int(hi_first), N.dt(dtype), N.stream())
        return out
    if chunked:
        codes, scales = cb4_from_chunk_layout(codes, scales)
    if ok and K % 8 == 0:
        out = torch.empty((N_, K), dtype=dtype, device=codes.device)
        N.call("vda_cb4_dequant", out.data_ptr(), codes.data_ptr(), scales.data_ptr(), table.data_ptr(), N_, K, group,
               int(hi_first), N.dt(dtype), N.stream())
        return out
    return R.cb4_dequant(codes, scales, table, group, hi_first, dtype)


def cb4_linear(x: torch.Tensor, codes: torch.Tensor, scales: torch.Tensor, table: torch.Tensor, group: int,
               hi_first: bool, bias: Optional[torch.Tensor] = None, chunked: bool = False,
               splits: int = 0) -> torch.Tensor:
    """y = x @ (table[codes] * scales).T (+ bias) over weights that stay 4-bit codes in device memory
    (ops/csrc/cb4.hip).  Decode-sized batches (``VDA_CB4_SKINNY_MAX_M`` rows, 64 at most) stream the codes once
    through an MFMA skinny kernel -- in the chunk layout (``chunked``) with operands straight out of a byte-indexed
    table and the scales applied to the accumulators, otherwise scaling every weight; more rows expand the codes into
    a transient 16-bit copy for the library GEMM."""
    M, K = x.shape
    N_ = codes.shape[0]
    if _dev(x) and M <= min(envs.VDA_CB4_SKINNY_MAX_M, 64) and N_ % 16 == 0 and K % 128 == 0 and K % group == 0 \
            and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0 \
            and x.dtype in (torch.bfloat16, torch.float16) and scales.dtype == torch.float32 \
            and codes.is_contiguous() and scales.is_contiguous() and (bias is None or bias.dtype == x.dtype) \
            and (cb4_chunk_layout_ok(K, group) if chunked else (group == 16 or group % 32 == 0)):
        y = torch.empty((M, N_), dtype=x.dtype, device=x.device)
        if chunked:
            # K splits write slabs of fp32 partial sums that a second small kernel adds (fewer splits if they do not fit)
            ws = _CB4_WS.get(x.device)
            if ws is None:
                ws = _CB4_WS[x.device] = torch.empty(1 << 23, dtype=torch.float32, device=x.device)
            N.call("vda_cb4_skinny_gemm_post", y.data_ptr(), x.data_ptr(), codes.data_ptr(), scales.data_ptr(),
                   table.data_ptr(), N.ptr(bias), M, N_, K, group, int(hi_first), x.stride(0), y.stride(0),
                   N.dt(x.dtype), ws.data_ptr(), ws.numel(), splits, N.stream())
        else:
            N.call("vda_cb4_skinny_gemm", y.data_ptr(), x.data_ptr(), codes.data_ptr(), scales.data_ptr(),
                   table.data_ptr(), N.ptr(bias), M, N_, K, group, int(hi_first), x.stride(0), y.stride(0),
                   N.dt(x.dtype), N.stream())
        return y
    return torch.nn.functional.linear(x, cb4_dequant(codes, scales, table, group, hi_first, x.dtype, chunked), bias)


def wq_dequant(qw: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, group: int, bits: int,
               dtype: torch.dtype) -> torch.Tensor:
    """Packed 4- / 8-bit weights -> [N, K] in ``dtype``: (q - zero) * scale per group of ``group`` K elements."""
    if bits == 4:
        return w4_dequant(qw, scales, zeros, group, dtype)
    N_, K = qw.shape
    if _dev(qw) and dtype in (torch.bfloat16, torch.float16) and scales.dtype == dtype and K % 8 == 0 and group % 8 == 0:
        out = torch.empty((N_, K), dtype=dtype, device=qw.device)
        N.call("vda_w8_dequant", out.data_ptr(), qw.data_ptr(), scales.data_ptr(), zeros.data_ptr(), N_, K, group,
               N.dt(dtype), N.stream())
        return out
    g = torch.arange(K, device=qw.device) // group
    return ((qw.float() - zeros.float()[:, g]) * scales.float()[:, g]).to(dtype)


def wq_gemm_ok(x: torch.Tensor, qw: torch.Tensor, group: int, bits: int) -> bool:
    K = x.shape[1]
    return (_dev(x) and x.dtype in (torch.bfloat16, torch.float16) and x.stride(1) == 1 and x.stride(0) % 8 == 0
            and K % 32 == 0 and group % 32 == 0 and K % group == 0 and qw.is_contiguous()
            and qw.numel() * qw.element_size() < (1 << 32) and x.shape[0] * x.stride(0) * 2 < (1 << 32))


def wq_gemm(x: torch.Tensor, qw: torch.Tensor, sz: torch.Tensor, group: int, bits: int,
            bias: Optional[torch.Tensor] = None, splits: int = 0, variant: int = 0) -> torch.Tensor:
    """y = x @ dequant(qw).T (+ bias) for any number of rows with the weights kept packed: they are widened inside
    the kernel (ops/csrc/gemm_wq.hip): up to 128 rows a weight-streaming kernel (packed words global -> registers),
    more rows a tiled ping-pong kernel.  ``qw``: int32 [N, K/8] (4 bit, nibble order of ``w4_pack``) or uint8 [N, K];
    ``sz``: [K/group, N, 2] (scale, zero) pairs in x's dtype.  ``variant`` 1 / 2 force the tiled / streaming kernel."""
    M, K = x.shape
    N_ = qw.shape[0]
    if sz.shape != (K // group, N_, 2) or sz.dtype != x.dtype or not sz.is_contiguous():
        raise ValueError(f"wq_gemm: sz must be contiguous [K/group, N, 2] of {x.dtype}, got {tuple(sz.shape)} {sz.dtype}")
    y = torch.empty((M, N_), dtype=x.dtype, device=x.device)
    ws = _DECODE_WS.get(x.device)
    if ws is None:
        ws = _DECODE_WS[x.device] = torch.empty(envs.VDA_DECODE_GEMM_WS_MB << 18, dtype=torch.float32, device=x.device)
    N.call("vda_wq_gemm", y.data_ptr(), x.data_ptr(), qw.data_ptr(), sz.data_ptr(), N.ptr(bias), ws.data_ptr(),
           ws.numel() * 4, M, N_, K, group, bits, x.stride(0), y.stride(0), N.dt(x.dtype), splits, variant, N.stream())
    return y


def wq_linear(x: torch.Tensor, qw: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, sz: torch.Tensor,
              group: int, bits: int, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Weight-only quantised linear over weights that stay packed in device memory.  Up to ``VDA_WQ_GEMM_MAX_M``
    rows the packed-weight kernels run (weight-streaming up to 128 rows, tiled ping-pong above: both read a quarter /
    half of the bytes of 16-bit weights).  Larger prefill batches are compute bound: a transient 16-bit copy made by
    the dequantisation kernel plus the library GEMM is faster there and is freed right after.  CPU: PyTorch."""
    M = x.shape[0]
    if M <= envs.VDA_WQ_GEMM_MAX_M and wq_gemm_ok(x, qw, group, bits) \
            and (bias is None or (bias.dtype == x.dtype and qw.shape[0] % 4 == 0)):
        return wq_gemm(x, qw, sz, group, bits, bias)
    if bits == 4 and M <= envs.VDA_W4_SKINNY_MAX_M:
        return w4_linear(x, qw, scales, zeros, group, bias)
    return torch.nn.functional.linear(x, wq_dequant(qw, scales, zeros, group, bits, x.dtype), bias)


# ---- multi-LoRA ----------------------------------------------------------------------------------
def lora_update_ok(x: torch.Tensor, A: torch.Tensor, B: torch.Tensor) -> bool:
    return (_dev(x) and x.dtype in (torch.bfloat16, torch.float16) and A.dtype == x.dtype and B.dtype == x.dtype
            and x.shape[-1] % 8 == 0 and B.shape[1] % 8 == 0 and A.shape[1] % 8 == 0 and A.shape[1] <= 256)


def lora_update(y: torch.Tensor, x: torch.Tensor, A: torch.Tensor, B: torch.Tensor, tok_slot: torch.Tensor) -> None:
    """y[t] += B[s] @ (A[s] @ x[t]) with s = tok_slot[t] (rows with s < 0 untouched), in place.
    x [T, K], y [T, N], A [slots, R, K], B [slots, N, R], tok_slot int32 [T]."""
    T, K = x.shape
    R_, N_ = A.shape[1], B.shape[1]
    if not lora_update_ok(x, A, B):
        return R.lora_update(y, x, A, B, tok_slot)
    assert x.stride(1) == 1 and y.stride(1) == 1 and A.is_contiguous() and B.is_contiguous()
    h = torch.empty((T, R_), dtype=torch.float32, device=x.device)
    st = N.stream()
    N.call("vda_lora_shrink", h.data_ptr(), x.data_ptr(), A.data_ptr(), tok_slot.data_ptr(), T, R_, K, x.stride(0),
           N.dt(x.dtype), st)
    N.call("vda_lora_expand", y.data_ptr(), h.data_ptr(), B.data_ptr(), tok_slot.data_ptr(), T, R_, N_, y.stride(0),
           N.dt(x.dtype), st)


# ---- mixture of experts ------------------------------------------------------------------------
def moe_topk_softmax(logits: torch.Tensor, k: int, renormalize: bool = True):
    """Router: softmax over experts then top-k.  Returns (weights f32 [T,k], ids i32 [T,k])."""
    if not _dev(logits) or logits.shape[-1] > 512:
        return R.moe_topk_softmax(logits, k, renormalize)
    lg = _contig_last(logits)
    T, E = lg.shape
    w = torch.empty((T, k), dtype=torch.float32, device=lg.device)
    ids = torch.empty((T, k), dtype=torch.int32, device=lg.device)
    N.call("vda_moe_topk_softmax", w.data_ptr(), ids.data_ptr(), lg.data_ptr(), T, E, k, int(renormalize),
           lg.stride(0), N.dt(lg.dtype), N.stream())
    return w, ids


def moe_grouped_ok(x: torch.Tensor, w13: torch.Tensor, w2: torch.Tensor) -> bool:
    I = w13.shape[1] // 2
    return (x.dtype in (torch.bfloat16, torch.float16) and w13.shape[2] % 128 == 0 and I % 128 == 0
            and w2.shape[1] % 16 == 0 and w13.dtype in (x.dtype, torch.float8_e4m3fn) and w13.dtype == w2.dtype
            and w13.shape[0] <= 1024)


EPI_PLAIN, EPI_GATE_UP, EPI_DOWN = 0, 1, 2
SC_NONE, SC_TOKEN, SC_BLOCK = 0, 1, 2


def per_group_fp8_quant(x: torch.Tensor, group: int = 128):
    """e4m3 quantisation with one dynamic scale per ``group`` consecutive elements of a row
    (absmax / 448).  Returns (fp8 [.., K], fp32 scales [.., K / group])."""
    K = x.shape[-1]
    if not _dev(x) or K % group or group % 8:
        return R.per_group_fp8_quant(x, group)
    x2 = _contig_last(x).view(-1, K)
    out = torch.empty(x2.shape, dtype=torch.float8_e4m3fn, device=x.device)
    scales = torch.empty((x2.shape[0], K // group), dtype=torch.float32, device=x.device)
    N.call("vda_per_group_fp8_quant", out.data_ptr(), scales.data_ptr(), x2.data_ptr(), x2.shape[0], K, group,
           x2.stride(0), out.stride(0), N.dt(x.dtype), N.stream())
    return out.view(x.shape), scales.view(*x.shape[:-1], K // group)


def _tile_rows(avg_rows: float) -> int:
    """Activation rows per tile of the tiled GEMM: 128, or 64 when the matrices see few rows each."""
    return 64 if avg_rows <= 64 else 128


def tile_gemm_ok(K: int, in_dtype: torch.dtype) -> bool:
    eb = 1 if in_dtype == torch.float8_e4m3fn else 2
    return (K * eb) % 128 == 0 and in_dtype in (torch.float8_e4m3fn, torch.bfloat16, torch.float16)


def fp8_gemm(xq: torch.Tensor, w: torch.Tensor, x_scale: torch.Tensor, w_scale: torch.Tensor,
             out_dtype: torch.dtype, bias: Optional[torch.Tensor] = None, block: bool = False) -> torch.Tensor:
    """y = (xq * x_scale) @ (w * w_scale).T on the fp8 matrix cores.  xq [M, K] e4m3, w [N, K] e4m3.
    ``block=False``: x_scale per row [M] (or one element, broadcast), w_scale one element.
    ``block=True``: x_scale [M, K/128] (groups of 128 along K), w_scale [ceil(N/128), K/128]."""
    M, K = xq.shape
    N_ = w.shape[0]
    if not _dev(xq) or not tile_gemm_ok(K, xq.dtype) or xq.stride(1) != 1 or (xq.stride(0) % 16) or N_ % 4:
        return R.fp8_gemm(xq, w, x_scale, w_scale, out_dtype, bias, block)
    xs = x_scale.reshape(M, -1) if block else (x_scale.reshape(-1) if x_scale.numel() == M
                                               else x_scale.reshape(1).expand(M))
    xs = xs.contiguous().float()
    ws = w_scale.contiguous().float() if block else w_scale.reshape(1).float()
    out = torch.empty((M, N_), dtype=out_dtype, device=xq.device)
    N.call("vda_tile_gemm", out.data_ptr(), xq.data_ptr(), w.data_ptr(), xs.data_ptr(), ws.data_ptr(), None,
           N.ptr(bias), None, None, None, None, None, M, N_, K, 1, 0, xq.stride(0), out.stride(0), 0, EPI_PLAIN,
           SC_BLOCK if block else SC_TOKEN, _tile_rows(M), N.dt(xq.dtype), N.dt(out_dtype), 0, N.stream())
    return out


def sort_by_key(keys: torch.Tensor, num_keys: int):
    """Counting sort of positions by key: returns (sorted positions int32 [n], offsets int32
    [num_keys + 1]); positions with a negative key are left out (they sit past offsets[num_keys])."""
    keys = keys.contiguous().view(-1)
    n = keys.numel()
    if not _dev(keys):
        k64 = keys.long()
        order = torch.argsort(torch.where(k64 < 0, num_keys, k64), stable=True).to(torch.int32)
        cnt = torch.bincount(k64[k64 >= 0], minlength=num_keys)
        off = torch.zeros(num_keys + 1, dtype=torch.int32, device=keys.device)
        off[1:] = cnt.cumsum(0).to(torch.int32)
        return order, off
    keys = keys.to(torch.int32)
    sorted_ids = torch.zeros(n, dtype=torch.int32, device=keys.device)
    meta = torch.empty(3 * num_keys + 2, dtype=torch.int32, device=keys.device)
    offsets = meta[:num_keys + 1]
    if n > 4096:
        N.call("vda_moe_align_large", sorted_ids.data_ptr(), offsets.data_ptr(), meta[num_keys + 1:].data_ptr(),
               meta[2 * num_keys + 1:].data_ptr(), keys.data_ptr(), n, num_keys, N.stream())
    else:
        N.call("vda_moe_align", sorted_ids.data_ptr(), offsets.data_ptr(), keys.data_ptr(), n, num_keys, N.stream())
    return sorted_ids, offsets


def ep_prep(topk_ids: torch.Tensor, world: int, experts_per_rank: int):
    """Expert-parallel dispatch bookkeeping for the flattened (token, slot) pairs with global expert ids
    ``topk_ids`` int32 [P]: returns (ids with invalid entries set to -1, destination rank per pair or -1, pair
    positions sorted by destination (stable), offsets int32 [world + 1], position of every pair inside its
    destination's region or -1).  One kernel on the GPU (ops/csrc/moe.hip ep_prep_kernel)."""
    ids = topk_ids.contiguous().view(-1).to(torch.int32)
    P_ = ids.numel()
    if not _dev(ids) or world > 8:
        valid = (ids >= 0) & (ids < world * experts_per_rank)
        clean = torch.where(valid, ids, torch.full_like(ids, -1))
        dest = torch.where(valid, ids // experts_per_rank, torch.full_like(ids, -1))
        sorted_ids, offsets = sort_by_key(dest, world)
        inv = torch.zeros(P_ + 1, dtype=torch.int64, device=ids.device)
        seq = torch.arange(P_, device=ids.device, dtype=torch.int64)
        inv[torch.where(seq < offsets[world].long(), sorted_ids.long(), torch.full_like(seq, P_))] = seq
        slot = torch.where(valid, (inv[:P_] - offsets.long()[dest.clamp(min=0).long()]).to(torch.int32),
                           torch.full_like(ids, -1))
        return clean, dest, sorted_ids, offsets, slot
    out = torch.empty(4 * P_ + world + 1, dtype=torch.int32, device=ids.device)
    clean, dest, sorted_ids, slot = (out[i * P_:(i + 1) * P_] for i in range(4))
    offsets = out[4 * P_:]
    N.call("vda_ep_prep", ids.data_ptr(), P_, world, experts_per_rank, clean.data_ptr(), dest.data_ptr(),
           sorted_ids.data_ptr(), offsets.data_ptr(), slot.data_ptr(), N.stream())
    return clean, dest, sorted_ids, offsets, slot


def tile_linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, rows_per_tile: int = 0
                ) -> torch.Tensor:
    """y = x @ w.T (+ bias) through the LDS-tiled 16-bit MFMA GEMM (x [M, K], w [N, K], same 16-bit dtype)."""
    M, K = x.shape
    N_ = w.shape[0]
    out = torch.empty((M, N_), dtype=x.dtype, device=x.device)
    N.call("vda_tile_gemm", out.data_ptr(), x.data_ptr(), w.data_ptr(), None, None, None, N.ptr(bias), None, None,
           None, None, None, M, N_, K, 1, 0, x.stride(0), out.stride(0), 0, EPI_PLAIN, SC_NONE,
           rows_per_tile or _tile_rows(M), N.dt(x.dtype), N.dt(x.dtype), 0, N.stream())
    return out


def moe_align_tiles(topk_ids: torch.Tensor, E: int, rows_per_tile: int):
    """Sort the (token, slot) pairs by expert and cut every expert's range into m-tiles of
    ``rows_per_tile`` pairs.  Returns (sorted ids, offsets [E + 1], tile table, bound on tiles);
    everything stays on the device, nothing synchronises."""
    ids = topk_ids.contiguous().view(-1)
    n = ids.numel()
    dev = ids.device
    sorted_ids = torch.empty(n, dtype=torch.int32, device=dev)
    meta = torch.empty(3 * E + 2, dtype=torch.int32, device=dev)
    offsets, counts, cursor = meta[:E + 1], meta[E + 1:2 * E + 1], meta[2 * E + 1:3 * E + 1]
    st = N.stream()
    if n > 4096:
        N.call("vda_moe_align_large", sorted_ids.data_ptr(), offsets.data_ptr(), counts.data_ptr(), cursor.data_ptr(),
               ids.data_ptr(), n, E, st)
    else:
        N.call("vda_moe_align", sorted_ids.data_ptr(), offsets.data_ptr(), ids.data_ptr(), n, E, st)
    max_tiles = min(n, n // rows_per_tile + E)
    tiles = torch.empty(3 * max_tiles + 1, dtype=torch.int32, device=dev)
    N.call("vda_moe_tile_map", tiles[:max_tiles].data_ptr(), tiles[max_tiles:2 * max_tiles].data_ptr(),
           tiles[2 * max_tiles:3 * max_tiles].data_ptr(), tiles[3 * max_tiles:].data_ptr(), offsets.data_ptr(), E,
           rows_per_tile, max_tiles, st)
    return sorted_ids, offsets, tiles, max_tiles


def _moe_tiled(x, w13, w2, topk_w, topk_ids, w13_scale, w2_scale, block_scales: bool = False,
               activation: str = "silu", x_scale=None, out_dtype=None):
    """Compute-bound batches: both expert GEMMs run as LDS-tiled MFMA GEMMs over the expert-sorted pairs
    (ops/csrc/gemm_tile.hip).  e4m3 weights: activations are quantised per token (or per group of 128
    with block-scaled weights) and the fp8 matrix cores do the products; 16-bit weights: 16-bit MFMA.
    ``x_scale`` [T, 1] fp32: x already is e4m3 with one scale per row (rows that arrived quantised from
    another rank); ``out_dtype`` then names the 16-bit type of the result."""
    odt = out_dtype or x.dtype
    T, H = x.shape
    k = topk_ids.shape[1]
    E, I2, _ = w13.shape
    I = I2 // 2
    P_ = T * k
    dev = x.device
    rows = _tile_rows(P_ / E)
    sorted_ids, _, tiles, max_tiles = moe_align_tiles(topk_ids, E, rows)
    te, tb, tend, ntl = (tiles[:max_tiles], tiles[max_tiles:2 * max_tiles], tiles[2 * max_tiles:3 * max_tiles],
                         tiles[3 * max_tiles:])
    fp8 = w13.dtype == torch.float8_e4m3fn
    st = N.stream()
    inter = torch.empty((P_, I), dtype=odt, device=dev)
    out2 = torch.empty((P_, H), dtype=odt, device=dev)
    out = torch.empty((T, H), dtype=odt, device=dev)
    tw = topk_w.contiguous()
    if fp8:
        mode = SC_BLOCK if block_scales else SC_TOKEN
        if x_scale is not None:
            xq, xs = x, x_scale.float().reshape(T, 1)
        else:
            xq, xs = per_group_fp8_quant(x) if block_scales else per_token_fp8_quant(x)
        xs = xs.contiguous()
    else:
        mode, xq, xs = SC_NONE, _contig_last(x), None
    N.call("vda_tile_gemm", inter.data_ptr(), xq.data_ptr(), w13.data_ptr(), N.ptr(xs), N.ptr(w13_scale), None, None,
           sorted_ids.data_ptr(), te.data_ptr(), tb.data_ptr(), tend.data_ptr(), ntl.data_ptr(), 0, I, H, k,
           max_tiles, xq.stride(0), I, I2 * H, EPI_GATE_UP, mode, rows, N.dt(xq.dtype), N.dt(odt), N.ACT[activation], st)
    if fp8:
        hq, hs = per_group_fp8_quant(inter) if block_scales else per_token_fp8_quant(inter)
        hs = hs.contiguous()
    else:
        hq, hs = inter, None
    N.call("vda_tile_gemm", out2.data_ptr(), hq.data_ptr(), w2.data_ptr(), N.ptr(hs), N.ptr(w2_scale),
           tw.data_ptr(), None, sorted_ids.data_ptr(), te.data_ptr(), tb.data_ptr(), tend.data_ptr(),
           ntl.data_ptr(), 0, H, I, 1, max_tiles, hq.stride(0), H, H * I, EPI_DOWN, mode, rows, N.dt(hq.dtype),
           N.dt(odt), 0, st)
    N.call("vda_moe_sum", out.data_ptr(), out2.data_ptr(), topk_ids.contiguous().data_ptr(), T, H, k,
           out.stride(0), N.dt(odt), st)
    return out


def moe_fp8_rows_ok(w13: torch.Tensor, w2: torch.Tensor, dtype: torch.dtype, pairs: int) -> bool:
    """Will ``fused_moe`` quantise these rows per token and run the fp8 tiles?  Then they may as well arrive
    quantised (expert-parallel dispatch sends e4m3 rows + their scale: half the bytes per xGMI link, and the
    receiver's quantisation pass disappears)."""
    probe = torch.empty(0, dtype=dtype)
    return (w13.dtype == torch.float8_e4m3fn and dtype in (torch.bfloat16, torch.float16) and w13.is_cuda
            and not envs.VDA_FORCE_TORCH_OPS and moe_tiled_ok(probe, w13, w2) and w13.shape[2] % 16 == 0
            and (pairs > min(envs.VDA_MOE_GROUPED_MAX_PAIRS, envs.VDA_MOE_GROUPED_MAX_PAIRS_FP8)
                 or not moe_grouped_ok(probe, w13, w2)))


def moe_tiled_ok(x: torch.Tensor, w13: torch.Tensor, w2: torch.Tensor) -> bool:
    I = w13.shape[1] // 2
    return (x.dtype in (torch.bfloat16, torch.float16) and w13.dtype in (x.dtype, torch.float8_e4m3fn)
            and w13.dtype == w2.dtype and tile_gemm_ok(w13.shape[2], w13.dtype) and tile_gemm_ok(I, w13.dtype)
            and I % 64 == 0 and w2.shape[1] % 4 == 0 and w13.shape[0] <= 1024)


def moe_rows_form(num_pairs: int, num_experts: int, rows: int, weight_bytes: int = 2) -> int:
    """Grouped expert GEMM, which form (0 K-split, 1 row-split with 64 tokens per pass, 2 with 128): the
    row-split kernel (each wave 16 output rows over the whole K, activations shared through LDS) pays once
    an expert sees enough tokens for the shared tile to matter and the launch still fills the chip; below
    that the K-split kernel (four waves per 16 rows) has more workgroups in flight.  An expert with more
    tokens than one pass holds streams its weights again, hence the wide pass once the average nears
    half of the narrow one.  Thresholds from benchmarks_dev/moe_probe.py on MI355X."""
    forced = envs.VDA_MOE_ROWS_FORM
    if forced >= 0:
        return int(forced)
    if rows % 64:
        return 0
    avg = num_pairs / max(num_experts, 1)
    fills = (rows // 64) * num_experts >= 512
    # measured: with 8-bit weights the 128-token pass is ahead at every size once the launch fills the chip;
    # with 16-bit weights from ~40 tokens per expert; a launch that does not fill the chip (one or two
    # experts on the rank) wants it only when experts would otherwise need several passes
    wide = 2 if ((fills and (weight_bytes == 1 or avg > 40)) or avg >= 192) else 1
    if weight_bytes == 1 and fills:
        return wide
    return wide if avg >= 8 and (fills or avg >= 64) else 0


MOE_ACTIVATIONS = ("silu", "gelu", "gelu_tanh", "gelu_pytorch_tanh", "relu", "relu2")


def fused_moe(x: torch.Tensor, w13: torch.Tensor, w2: torch.Tensor, topk_w: torch.Tensor, topk_ids: torch.Tensor,
              w13_scale: Optional[torch.Tensor] = None, w2_scale: Optional[torch.Tensor] = None,
              block_scales: bool = False, activation: str = "silu", x_scale: Optional[torch.Tensor] = None,
              out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """Routed gated expert MLPs, out[t] = sum_j w[t, j] * W2_e(act(Wg_e x_t) * (Wu_e x_t)) with ``activation`` one of
    MOE_ACTIVATIONS (reference: vllm/model_executor/layers/fused_moe/fused_moe.py:1003,1034 ``activation``), fused
    into the first GEMM's epilogue in every kernel form.  x [T,H]; w13 [E,2I,H] (gate rows then up rows); w2 [E,H,I];
    topk_w f32 [T,k]; topk_ids i32 [T,k] with local expert ids, negative = not on this rank.
    e4m3 weights carry one scale per expert, or with ``block_scales`` one per 128 x 128 block
    (w13_scale [E, 2I/128, H/128], w2_scale [E, H/128, I/128])."""
    if activation not in MOE_ACTIVATIONS:
        raise ValueError(f"expert activation {activation!r} is not supported (supported: {', '.join(MOE_ACTIVATIONS)})")
    if x_scale is not None:
        # x is e4m3 with one scale per row (see moe_fp8_rows_ok): straight into the fp8 tiles, or widened again
        odt = out_dtype or torch.bfloat16
        if _dev(x) and not block_scales and moe_tiled_ok(torch.empty(0, dtype=odt), w13, w2):
            return _moe_tiled(x, w13, w2, topk_w, topk_ids, w13_scale, w2_scale, False, activation, x_scale, odt)
        x = (x.float() * x_scale.float().reshape(-1, 1)).to(odt)
    if _dev(x) and block_scales and not moe_tiled_ok(x, w13, w2):
        raise RuntimeError(f"block-scaled expert shapes {tuple(w13.shape)} / {tuple(w2.shape)} are outside the tiled "
                           "fp8 GEMM (hidden and expert width must be multiples of 128)")
    if not _dev(x) or (not moe_grouped_ok(x, w13, w2) and not moe_tiled_ok(x, w13, w2)):
        if _dev(x) and torch.cuda.is_current_stream_capturing():
            raise RuntimeError(
                f"expert shapes {tuple(w13.shape)} / {tuple(w2.shape)} are outside the grouped MoE kernel (hidden and "
                "expert width must be multiples of 128); the PyTorch fallback synchronises and cannot be captured "
                "into a hipGraph: run with enforce_eager=True")
        return R.fused_moe(x, w13, w2, topk_w, topk_ids, w13_scale, w2_scale, block_scales, activation)
    if not moe_grouped_ok(x, w13, w2):
        return _moe_tiled(_contig_last(x), w13, w2, topk_w, topk_ids, w13_scale, w2_scale, block_scales, activation)
    T, H = x.shape
    k = topk_ids.shape[1]
    E, I2, _ = w13.shape
    I = I2 // 2
    # e4m3 experts: from a few hundred pairs on, quantised activations through the fp8 MFMA tiles beat
    # widening the weights for 16-bit MFMAs (Mixtral-8x7B layer, 256 pairs: 396 us against 420; 2048: 684 / 1367)
    limit = envs.VDA_MOE_GROUPED_MAX_PAIRS
    if w13.dtype == torch.float8_e4m3fn:
        limit = min(limit, envs.VDA_MOE_GROUPED_MAX_PAIRS_FP8)
    if T * k > limit or block_scales:
        return _moe_tiled(_contig_last(x), w13, w2, topk_w, topk_ids, w13_scale, w2_scale, block_scales, activation)
    x = _contig_last(x)
    dev = x.device
    ids = topk_ids.contiguous()
    tw = topk_w.contiguous()
    sorted_ids = torch.empty(T * k, dtype=torch.int32, device=dev)
    offsets = torch.empty(E + 1, dtype=torch.int32, device=dev)
    st = N.stream()
    N.call("vda_moe_align", sorted_ids.data_ptr(), offsets.data_ptr(), ids.data_ptr(), T * k, E, st)
    inter = torch.empty((T * k, I), dtype=x.dtype, device=dev)
    out2 = torch.empty((T * k, H), dtype=x.dtype, device=dev)
    out = torch.empty((T, H), dtype=x.dtype, device=dev)
    wd = N.dt(w13.dtype)
    # one expert can receive at most one slot of every token
    max_pairs = T
    N.call("vda_moe_gate_up", inter.data_ptr(), x.data_ptr(), w13.data_ptr(), N.ptr(w13_scale), sorted_ids.data_ptr(),
           offsets.data_ptr(), E, I, H, k, max_pairs, x.stride(0), N.dt(x.dtype), wd,
           int(moe_rows_form(T * k, E, I, w13.element_size())), N.ACT[activation], st)
    N.call("vda_moe_down", out2.data_ptr(), inter.data_ptr(), w2.data_ptr(), N.ptr(w2_scale), tw.data_ptr(),
           sorted_ids.data_ptr(), offsets.data_ptr(), E, H, I, max_pairs, N.dt(x.dtype), wd,
           int(moe_rows_form(T * k, E, H, w2.element_size())), st)
    N.call("vda_moe_sum", out.data_ptr(), out2.data_ptr(), ids.data_ptr(), T, H, k, out.stride(0), N.dt(x.dtype), st)
    return out


# ---- state-space layers (Mamba / Mamba-2) ------------------------------------------------------------
def causal_conv1d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], conv_state: torch.Tensor,
                  query_start_loc: torch.Tensor, seq_lens: torch.Tensor, block_table: torch.Tensor,
                  max_query_len: int, activation: bool = True) -> torch.Tensor:
    """Depthwise causal convolution over a ragged batch.  x [T, C] (row stride arbitrary), weight [C, W],
    conv_state [slots, C, W-1] holds the last W-1 inputs of every request and is rolled forward in place;
    sequence s uses slot ``block_table[s, 0]``, and rows with ``seq_lens[s] == 0`` are skipped."""
    S = seq_lens.shape[0]
    if not _dev(x):
        return R.causal_conv1d(x, weight, bias, conv_state, query_start_loc, seq_lens, block_table, activation)
    T, C = x.shape
    out = torch.empty((T, C), dtype=x.dtype, device=x.device)
    assert x.stride(1) == 1 and weight.is_contiguous() and conv_state.is_contiguous()
    assert conv_state.shape[1] == C and conv_state.shape[2] == weight.shape[1] - 1 and conv_state.dtype == x.dtype
    N.call("vda_causal_conv1d", out.data_ptr(), x.data_ptr(), weight.data_ptr(), N.ptr(bias), conv_state.data_ptr(),
           query_start_loc.data_ptr(), seq_lens.data_ptr(), block_table.data_ptr(), block_table.stride(0), S, C,
           weight.shape[1], max(int(max_query_len), 1), x.stride(0), out.stride(0), int(activation), N.dt(x.dtype),
           N.stream())
    return out


def selective_scan(u: torch.Tensor, delta: torch.Tensor, A: torch.Tensor, B: torch.Tensor, C: torch.Tensor,
                   D: Optional[torch.Tensor], z: Optional[torch.Tensor], delta_bias: Optional[torch.Tensor],
                   ssm_state: torch.Tensor, query_start_loc: torch.Tensor, seq_lens: torch.Tensor,
                   block_table: torch.Tensor, softplus: bool = True, num_groups: int = 1,
                   max_query_len: int = 1) -> torch.Tensor:
    """Selective state-space scan over a ragged batch with carried state.
    u, z [T, Ch]; delta [T, Ch] (Mamba) or [T, heads] (Mamba-2: one step size per head);
    A fp32 [Ch, N] or [heads]; B, C [T, groups * N]; D, delta_bias fp32 with delta's width;
    ssm_state fp32 [slots, Ch, N], updated in place.  Returns y * silu(z) (y when z is None)."""
    if not _dev(u):
        return R.selective_scan(u, delta, A, B, C, D, z, delta_bias, ssm_state, query_start_loc, seq_lens,
                                block_table, softplus, num_groups)
    T, Ch = u.shape
    Nst = ssm_state.shape[2]
    E = delta.shape[1]
    assert Ch % E == 0 and Ch % num_groups == 0 and ssm_state.shape[1] == Ch and ssm_state.dtype == torch.float32
    assert A.dtype == torch.float32 and A.is_contiguous() and ssm_state.is_contiguous()
    assert B.shape[1] == num_groups * Nst and C.shape[1] == num_groups * Nst
    for t in (u, delta, B, C) + ((z,) if z is not None else ()):
        assert t.stride(1) == 1 and t.dtype == u.dtype
    out = torch.empty((T, Ch), dtype=u.dtype, device=u.device)
    # Long prefills of few sequences leave the chip idle when every (sequence, channel) walks its tokens
    # alone, so the tokens are cut into chunks that run in parallel (summary, combine, output passes).
    # The passes cost workspace traffic of 6 x Ch x N floats per chunk, so only as many chunks are made
    # as it takes to reach ~32 waves per CU; wide states (Mamba-2) fill the chip almost without chunking.
    S = seq_lens.shape[0]
    chunk, nchunks, ws_p, ws_h = 0, 1, None, None
    if max_query_len >= envs.VDA_SCAN_CHUNK_MIN_TOKENS:
        want = -(-524288 // (S * Ch * (Nst // 16)))
        if want > 1:
            chunk = max(64, -(-(-(-max_query_len // want)) // 64) * 64)
            nchunks = -(-max_query_len // chunk)
        if nchunks > 1:
            ws_p = torch.empty((S, nchunks, Ch, Nst), dtype=torch.float32, device=u.device)
            ws_h = torch.empty_like(ws_p)
    N.call("vda_selective_scan", out.data_ptr(), u.data_ptr(), delta.data_ptr(), A.data_ptr(), B.data_ptr(),
           C.data_ptr(), N.ptr(D), N.ptr(z), N.ptr(delta_bias), ssm_state.data_ptr(), query_start_loc.data_ptr(),
           seq_lens.data_ptr(), block_table.data_ptr(), block_table.stride(0), seq_lens.shape[0], Ch, Nst,
           u.stride(0), delta.stride(0), B.stride(0), C.stride(0), z.stride(0) if z is not None else 0, out.stride(0),
           Ch // E, int(A.dim() == 1), Ch // num_groups, int(softplus), chunk, nchunks, N.ptr(ws_p), N.ptr(ws_h),
           N.dt(u.dtype), N.stream())
    return out


# ---- lightning (linear) attention -----------------------------------------------------------------------
def lightning_attention_decode(qkv: torch.Tensor, state: torch.Tensor, slope: torch.Tensor, seq_lens: torch.Tensor,
                               slot_table: torch.Tensor, num_heads: int, head_dim: int) -> torch.Tensor:
    """One token per sequence.  qkv [S, H * 3 D] (per head [q | k | v], already activated); state fp32
    [slots, H, D, D], updated in place: S <- exp(-slope) S + k^T v (from zero when ``seq_lens`` says this is the
    sequence's first token); slot_table [S, >= 1] int32 with the state slot in column 0.  Returns q S, fp32 [S, H, D]."""
    S = qkv.shape[0]
    H, D = num_heads, head_dim
    out = torch.empty((S, H, D), dtype=torch.float32, device=qkv.device)
    if _dev(qkv) and qkv.stride(1) == 1 and state.is_contiguous() and D in (16, 32, 64, 128, 256) \
            and state.dtype == torch.float32 and slope.dtype == torch.float32:
        N.call("vda_lightning_decode", out.data_ptr(), qkv.data_ptr(), state.data_ptr(), slope.data_ptr(),
               seq_lens.data_ptr(), slot_table.data_ptr(), slot_table.stride(0), S, H, D, qkv.stride(0),
               N.dt(qkv.dtype), N.stream())
        return out
    q, k, v = qkv.view(S, H, 3 * D).float().split(D, dim=-1)
    slots = slot_table[:S, 0].long()
    fresh = (seq_lens[:S] <= 1).view(S, 1, 1, 1)
    s = state[slots] * torch.where(fresh, 0.0, 1.0) * torch.exp(-slope.float()).view(1, H, 1, 1)
    s = s + k.unsqueeze(-1) * v.unsqueeze(-2)
    live = (seq_lens[:S] > 0).view(S, 1, 1, 1)
    state[slots] = torch.where(live, s, state[slots])
    return torch.matmul(q.unsqueeze(-2), s).squeeze(-2)
