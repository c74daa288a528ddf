"""GGUF checkpoints: a reader for the container (v2 / v3), block dequantisers for the common ggml
types, and the mapping from llama.cpp tensor / metadata names to the Hugging Face ones the model
classes load.  Linear layers whose tensors are all of a 4-, 5- or 8-bit block type with 32-element sub-blocks
(Q4_0, Q4_1, Q4_K / Q5_0, Q5_1, Q5_K, Q8_0) keep their codes packed in device memory (4 or 8 bits per weight, one
(scale, zero) pair per 32 weights) and run the packed-weight GEMM kernels (ops/csrc/gemm_wq.hip); every other tensor
(Q6_K with its 16-element sub-blocks, embeddings, norms) is expanded to 16 bit while loading.  The file is
memory-mapped and decoded tensor by tensor.

Only what is read from the file's own tables is trusted: sizes are checked against the file length
before any slice is taken.
Reference: vllm/model_executor/model_loader/gguf_loader.py, vllm/model_executor/layers/quantization/gguf.py
(which keep the blocks packed and dequantise in CUDA kernels, csrc/quantization/gguf/).
"""
from __future__ import annotations

import mmap
import struct
from dataclasses import dataclass
from typing import Any, Iterator, Optional

import numpy as np
import torch

GGUF_MAGIC = b"GGUF"
# metadata value types
_U8, _I8, _U16, _I16, _U32, _I32, _F32, _BOOL, _STR, _ARR, _U64, _I64, _F64 = range(13)
_SCALAR = {_U8: "<B", _I8: "<b", _U16: "<H", _I16: "<h", _U32: "<I", _I32: "<i", _F32: "<f", _BOOL: "<?",
           _U64: "<Q", _I64: "<q", _F64: "<d"}

# ggml tensor types: id -> (name, elements per block, bytes per block)
GGML_TYPES = {
    0: ("F32", 1, 4), 1: ("F16", 1, 2), 2: ("Q4_0", 32, 18), 3: ("Q4_1", 32, 20), 6: ("Q5_0", 32, 22),
    7: ("Q5_1", 32, 24), 8: ("Q8_0", 32, 34), 12: ("Q4_K", 256, 144), 13: ("Q5_K", 256, 176),
    14: ("Q6_K", 256, 210), 30: ("BF16", 1, 2),
}


@dataclass
class GGUFTensorInfo:
    name: str
    shape: tuple[int, ...]       # row-major (outermost first), i.e. the reverse of the file's order
    ggml_type: int
    offset: int                  # from the start of the data section


class GGUFReader:
    def __init__(self, path: str):
        self.path = path
        self._f = open(path, "rb")
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        self._pos = 0
        if self._read(4) != GGUF_MAGIC:
            raise ValueError(f"{path} is not a GGUF file")
        self.version = self._scalar(_U32)
        if self.version not in (2, 3):
            raise ValueError(f"GGUF version {self.version} is not supported (2 and 3 are)")
        n_tensors, n_kv = self._scalar(_U64), self._scalar(_U64)
        if n_tensors > 1 << 20 or n_kv > 1 << 20:
            raise ValueError("implausible GGUF table sizes")
        self.metadata: dict[str, Any] = {}
        for _ in range(n_kv):
            key = self._string()
            self.metadata[key] = self._value(self._scalar(_U32))
        self.tensors: dict[str, GGUFTensorInfo] = {}
        for _ in range(n_tensors):
            name = self._string()
            nd = self._scalar(_U32)
            if nd > 8:
                raise ValueError("implausible tensor rank")
            dims = [self._scalar(_U64) for _ in range(nd)]
            t, off = self._scalar(_U32), self._scalar(_U64)
            self.tensors[name] = GGUFTensorInfo(name, tuple(reversed(dims)), t, off)
        align = int(self.metadata.get("general.alignment", 32))
        self.data_start = (self._pos + align - 1) // align * align

    # ---- primitives ----------------------------------------------------------------------------------
    def _read(self, n: int) -> bytes:
        if n < 0 or self._pos + n > len(self._mm):
            raise ValueError("GGUF file is truncated")
        b = self._mm[self._pos:self._pos + n]
        self._pos += n
        return b

    def _scalar(self, t: int):
        fmt = _SCALAR[t]
        return struct.unpack(fmt, self._read(struct.calcsize(fmt)))[0]

    def _string(self) -> str:
        n = self._scalar(_U64)
        return self._read(n).decode("utf-8", errors="replace")

    def _value(self, t: int):
        if t in _SCALAR:
            return self._scalar(t)
        if t == _STR:
            return self._string()
        if t == _ARR:
            et, n = self._scalar(_U32), self._scalar(_U64)
            if et in _SCALAR and et != _BOOL:
                fmt = _SCALAR[et]
                size = struct.calcsize(fmt)
                return np.frombuffer(self._read(size * n), dtype=np.dtype(fmt)).tolist()
            return [self._value(et) for _ in range(n)]
        raise ValueError(f"unknown GGUF metadata type {t}")

    # ---- tensors -------------------------------------------------------------------------------------
    def raw(self, info: GGUFTensorInfo) -> np.ndarray:
        if info.ggml_type not in GGML_TYPES:
            raise ValueError(f"tensor {info.name}: ggml type {info.ggml_type} is not supported "
                             f"(supported: {', '.join(v[0] for v in GGML_TYPES.values())})")
        _, per, nbytes = GGML_TYPES[info.ggml_type]
        n = int(np.prod(info.shape)) if info.shape else 1
        if n % per:
            raise ValueError(f"tensor {info.name}: {n} elements is not a multiple of the block size {per}")
        size = n // per * nbytes
        a = self.data_start + info.offset
        if a + size > len(self._mm):
            raise ValueError(f"tensor {info.name} extends past the end of the file")
        return np.frombuffer(self._mm, dtype=np.uint8, count=size, offset=a)

    def load(self, name: str, dtype: torch.dtype = torch.float16) -> torch.Tensor:
        info = self.tensors[name]
        return dequantize(self.raw(info), info.ggml_type, info.shape).to(dtype)

    def close(self) -> None:
        self._mm.close()
        self._f.close()


# ---- block dequantisers (numpy, vectorised over blocks) ----------------------------------------------------
def _f16(b: np.ndarray) -> np.ndarray:
    return b.copy().view(np.float16).astype(np.float32)


def _q4_0(b):
    d = _f16(b[:, :2])
    q = b[:, 2:]
    lo, hi = (q & 15).astype(np.int8) - 8, (q >> 4).astype(np.int8) - 8
    return d * np.concatenate([lo, hi], axis=1)


def _q4_1(b):
    d, m = _f16(b[:, :2]), _f16(b[:, 2:4])
    q = b[:, 4:]
    return d * np.concatenate([q & 15, q >> 4], axis=1) + m


def _high_bits(qh: np.ndarray) -> np.ndarray:
    """[n, 4] bytes -> [n, 32] bit j of the little-endian 32-bit word."""
    w = qh.copy().view(np.uint32)                      # [n, 1]
    return ((w >> np.arange(32, dtype=np.uint32)) & 1).astype(np.uint8)


def _q5_0(b):
    d = _f16(b[:, :2])
    h = _high_bits(b[:, 2:6])
    q = b[:, 6:]
    v = np.concatenate([(q & 15) | (h[:, :16] << 4), (q >> 4) | (h[:, 16:] << 4)], axis=1).astype(np.int8) - 16
    return d * v


def _q5_1(b):
    d, m = _f16(b[:, :2]), _f16(b[:, 2:4])
    h = _high_bits(b[:, 4:8])
    q = b[:, 8:]
    v = np.concatenate([(q & 15) | (h[:, :16] << 4), (q >> 4) | (h[:, 16:] << 4)], axis=1)
    return d * v + m


def _q8_0(b):
    return _f16(b[:, :2]) * b[:, 2:].copy().view(np.int8)


def _k_scales(s: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """The 12-byte packed 6-bit (scale, min) pairs of the K-quants -> two [n, 8] arrays."""
    sc = np.empty((s.shape[0], 8), dtype=np.uint8)
    mn = np.empty_like(sc)
    sc[:, :4] = s[:, 0:4] & 63
    mn[:, :4] = s[:, 4:8] & 63
    sc[:, 4:] = (s[:, 8:12] & 15) | ((s[:, 0:4] >> 6) << 4)
    mn[:, 4:] = (s[:, 8:12] >> 4) | ((s[:, 4:8] >> 6) << 4)
    return sc.astype(np.float32), mn.astype(np.float32)


def _q4_k(b):
    d, dmin = _f16(b[:, 0:2]), _f16(b[:, 2:4])
    sc, mn = _k_scales(b[:, 4:16])
    q = b[:, 16:].reshape(-1, 4, 32)                       # 4 chunks of 64 values: low nibbles, then high
    v = np.stack([q & 15, q >> 4], axis=2).reshape(-1, 8, 32).astype(np.float32)
    return ((d * sc)[:, :, None] * v - (dmin * mn)[:, :, None]).reshape(-1, 256)


def _q5_k(b):
    d, dmin = _f16(b[:, 0:2]), _f16(b[:, 2:4])
    sc, mn = _k_scales(b[:, 4:16])
    qh = b[:, 16:48]                                       # bit k of byte l: high bit of value l in sub-block k
    q = b[:, 48:].reshape(-1, 4, 32)
    lo = np.stack([q & 15, q >> 4], axis=2).reshape(-1, 8, 32)
    hb = ((qh[:, None, :] >> np.arange(8, dtype=np.uint8)[None, :, None]) & 1).astype(np.uint8)
    v = (lo | (hb << 4)).astype(np.float32)
    return ((d * sc)[:, :, None] * v - (dmin * mn)[:, :, None]).reshape(-1, 256)


def _q6_k(b):
    ql, qh = b[:, :128].reshape(-1, 2, 64), b[:, 128:192].reshape(-1, 2, 32)
    sc = b[:, 192:208].copy().view(np.int8).astype(np.float32)          # 16 scales of 16 values
    d = _f16(b[:, 208:210])
    a, c = ql[:, :, :32], ql[:, :, 32:]
    v = np.stack([(a & 15) | ((qh & 3) << 4), (c & 15) | (((qh >> 2) & 3) << 4),
                  (a >> 4) | (((qh >> 4) & 3) << 4), (c >> 4) | (((qh >> 6) & 3) << 4)], axis=2)   # [n, 2, 4, 32]
    v = v.astype(np.int8).astype(np.float32) - 32.0
    return (d[:, :, None] * sc.reshape(-1, 16, 1) * v.reshape(-1, 16, 16)).reshape(-1, 256)


_DEQUANT = {2: _q4_0, 3: _q4_1, 6: _q5_0, 7: _q5_1, 8: _q8_0, 12: _q4_k, 13: _q5_k, 14: _q6_k}


def dequantize(raw: np.ndarray, ggml_type: int, shape: tuple[int, ...]) -> torch.Tensor:
    """Bytes of one tensor -> float32 / float16 / bfloat16 torch tensor of ``shape``."""
    name, per, nbytes = GGML_TYPES[ggml_type]
    if name == "F32":
        return torch.from_numpy(raw.copy().view(np.float32)).reshape(shape)
    if name == "F16":
        return torch.from_numpy(raw.copy().view(np.float16)).reshape(shape)
    if name == "BF16":
        return torch.from_numpy(raw.copy().view(np.int16)).view(torch.bfloat16).reshape(shape)
    out = _DEQUANT[ggml_type](raw.reshape(-1, nbytes))
    return torch.from_numpy(np.ascontiguousarray(out, dtype=np.float32)).reshape(shape)


# ---- codes for the packed-weight kernels: value = code * scale + offset per 32 elements -------------------------
def _blocks_q4_0(b):
    q = b[:, 2:]
    return np.concatenate([q & 15, q >> 4], axis=1), _f16(b[:, :2]), -8.0 * _f16(b[:, :2])


def _blocks_q4_1(b):
    q = b[:, 4:]
    return np.concatenate([q & 15, q >> 4], axis=1), _f16(b[:, :2]), _f16(b[:, 2:4])


def _blocks_q5_0(b):
    h, q = _high_bits(b[:, 2:6]), b[:, 6:]
    return (np.concatenate([(q & 15) | (h[:, :16] << 4), (q >> 4) | (h[:, 16:] << 4)], axis=1), _f16(b[:, :2]),
            -16.0 * _f16(b[:, :2]))


def _blocks_q5_1(b):
    h, q = _high_bits(b[:, 4:8]), b[:, 8:]
    return (np.concatenate([(q & 15) | (h[:, :16] << 4), (q >> 4) | (h[:, 16:] << 4)], axis=1), _f16(b[:, :2]),
            _f16(b[:, 2:4]))


def _blocks_q8_0(b):
    d = _f16(b[:, :2])
    return (b[:, 2:].copy().view(np.int8).astype(np.int16) + 128).astype(np.uint8), d, -128.0 * d


def _blocks_q4_k(b):
    d, dmin = _f16(b[:, 0:2]), _f16(b[:, 2:4])
    sc, mn = _k_scales(b[:, 4:16])
    q = b[:, 16:].reshape(-1, 4, 32)
    v = np.stack([q & 15, q >> 4], axis=2).reshape(-1, 256)
    return v, d * sc, -(dmin * mn)


def _blocks_q5_k(b):
    d, dmin = _f16(b[:, 0:2]), _f16(b[:, 2:4])
    sc, mn = _k_scales(b[:, 4:16])
    qh = b[:, 16:48]
    q = b[:, 48:].reshape(-1, 4, 32)
    lo = np.stack([q & 15, q >> 4], axis=2).reshape(-1, 8, 32)
    hb = ((qh[:, None, :] >> np.arange(8, dtype=np.uint8)[None, :, None]) & 1).astype(np.uint8)
    return (lo | (hb << 4)).reshape(-1, 256), d * sc, -(dmin * mn)


# ggml type id -> (bits of a code, codes / scales / offsets of the blocks)
_BLOCKS = {2: (4, _blocks_q4_0), 3: (4, _blocks_q4_1), 12: (4, _blocks_q4_k), 6: (5, _blocks_q5_0),
           7: (5, _blocks_q5_1), 13: (5, _blocks_q5_k), 8: (8, _blocks_q8_0)}


def packed_tensors(raw: np.ndarray, ggml_type: int, shape: tuple[int, ...], bits: int,
                   dtype: torch.dtype) -> dict[str, torch.Tensor]:
    """Bytes of one [N, K] tensor -> the tensors of a packed linear layer (models/dequant.py
    PackedIntLinearMethod, groups of 32): qweight (4 bit: int32 [N, K/8]; 8 bit: uint8 [N, K]), scales and zero
    points [N, K/32] with value = (code - zero) * scale.  A sub-block whose scale is 0 (every element equals the
    offset) gets a scale small enough for the codes not to matter."""
    from ..ops import w4_pack
    _, per, nbytes = GGML_TYPES[ggml_type]
    code_bits, fn = _BLOCKS[ggml_type]
    if code_bits > bits or len(shape) != 2:
        raise ValueError(f"ggml type {GGML_TYPES[ggml_type][0]} does not fit {bits}-bit codes")
    q, s, off = fn(raw.reshape(-1, nbytes))
    N, K = shape
    q = torch.from_numpy(np.ascontiguousarray(q)).reshape(N, K)
    s = np.ascontiguousarray(s, dtype=np.float32).reshape(N, K // 32)
    off = np.ascontiguousarray(off, dtype=np.float32).reshape(N, K // 32)
    flat = s == 0
    s = np.where(flat, np.abs(off) / (32768.0 if dtype == torch.float16 else 1048576.0), s)     # zero stays finite
    z = np.where(s != 0, -off / np.where(s != 0, s, 1.0), 0.0)
    return {"qweight": w4_pack(q) if bits == 4 else q.to(torch.uint8),
            "scales": torch.from_numpy(s).to(dtype), "qzeros": torch.from_numpy(z.astype(np.float32)).to(dtype)}


# ---- llama.cpp names -> Hugging Face names ---------------------------------------------------------------
_LAYER_NAMES = {
    "attn_norm": "input_layernorm", "attn_q": "self_attn.q_proj", "attn_k": "self_attn.k_proj",
    "attn_v": "self_attn.v_proj", "attn_output": "self_attn.o_proj", "ffn_norm": "post_attention_layernorm",
    "ffn_gate": "mlp.gate_proj", "ffn_up": "mlp.up_proj", "ffn_down": "mlp.down_proj",
    "attn_q_norm": "self_attn.q_norm", "attn_k_norm": "self_attn.k_norm",
}
_TOP_NAMES = {"token_embd": "model.embed_tokens", "output_norm": "model.norm", "output": "lm_head"}
# architectures whose q / k rows llama.cpp's converter reorders for its interleaved rotary kernel
_PERMUTED_QK = ("llama", "mistral")
_ARCH = {"llama": "LlamaForCausalLM", "mistral": "MistralForCausalLM", "qwen2": "Qwen2ForCausalLM",
         "qwen3": "Qwen3ForCausalLM"}


def hf_name(name: str) -> Optional[str]:
    base, _, suffix = name.rpartition(".")
    if base in _TOP_NAMES:
        return f"{_TOP_NAMES[base]}.{suffix}"
    parts = base.split(".")
    if len(parts) == 3 and parts[0] == "blk" and parts[2] in _LAYER_NAMES:
        return f"model.layers.{parts[1]}.{_LAYER_NAMES[parts[2]]}.{suffix}"
    return None


def _unpermute(w: torch.Tensor, n_head: int) -> torch.Tensor:
    """Inverse of the converter's row reorder (pairs (i, i + d/2) were made adjacent)."""
    d = w.shape[0] // n_head
    return w.reshape(n_head, d // 2, 2, *w.shape[1:]).transpose(1, 2).reshape(w.shape)


def config_from_gguf(path: str) -> dict:
    """A Hugging Face style config dictionary from the file's metadata."""
    r = GGUFReader(path)
    try:
        md = r.metadata
        arch = md.get("general.architecture", "llama")
        if arch not in _ARCH:
            raise ValueError(f"GGUF architecture {arch!r} is not supported (supported: {', '.join(_ARCH)})")
        g = lambda k, d=None: md.get(f"{arch}.{k}", d)        # noqa: E731
        heads = int(g("attention.head_count"))
        hidden = int(g("embedding_length"))
        vocab = g("vocab_size") or len(md.get("tokenizer.ggml.tokens", [])) or r.tensors["token_embd.weight"].shape[0]
        cfg = {
            "architectures": [_ARCH[arch]], "model_type": arch, "hidden_size": hidden,
            "intermediate_size": int(g("feed_forward_length")), "num_hidden_layers": int(g("block_count")),
            "num_attention_heads": heads, "num_key_value_heads": int(g("attention.head_count_kv", heads)),
            "rms_norm_eps": float(g("attention.layer_norm_rms_epsilon", 1e-5)),
            "rope_theta": float(g("rope.freq_base", 10000.0)),
            "max_position_embeddings": int(g("context_length", 2048)), "vocab_size": int(vocab),
            "tie_word_embeddings": "output.weight" not in r.tensors, "torch_dtype": "float16",
            "bos_token_id": md.get("tokenizer.ggml.bos_token_id"), "eos_token_id": md.get("tokenizer.ggml.eos_token_id"),
        }
        hd = g("attention.key_length")
        if hd and int(hd) != hidden // heads:
            cfg["head_dim"] = int(hd)
        if g("rope.scaling.type") == "linear" and g("rope.scaling.factor"):
            cfg["rope_scaling"] = {"rope_type": "linear", "factor": float(g("rope.scaling.factor"))}
        return cfg
    finally:
        r.close()


# projections the model classes fuse into one layer: all of them must be packable for the layer to be packed
_FUSED = (("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"), ("mlp.gate_proj", "mlp.up_proj"))
_FUSED_NAME = {"self_attn.qkv_proj": _FUSED[0], "mlp.gate_up_proj": _FUSED[1]}


class GGUFPackedConfig:
    """Quantisation config of a GGUF file: which linear layers keep their block codes packed, and at how many
    bits.  Decided from the file's tensor table alone, so the layer constructors and the weight iterator agree.
    Reference: vllm/model_executor/layers/quantization/gguf.py:GGUFConfig / GGUFLinearMethod."""

    def __init__(self, path: str):
        r = GGUFReader(path)
        try:
            self.types: dict[str, int] = {}            # HF module path -> ggml type of its weight
            for name, info in r.tensors.items():
                target = hf_name(name)
                if target is not None and target.endswith(".weight") and len(info.shape) == 2:
                    self.types[target[: -len(".weight")]] = info.ggml_type
        finally:
            r.close()

    def _members(self, prefix: str) -> list[str]:
        head, _, leaf2 = prefix.rpartition(".")
        head, _, leaf1 = head.rpartition(".")
        leaf = f"{leaf1}.{leaf2}"
        if leaf in _FUSED_NAME:
            return [f"{head}.{m}" for m in _FUSED_NAME[leaf]]
        for group in _FUSED:
            if leaf in group:
                return [f"{head}.{m}" for m in group]
        return [prefix]

    def bits_for(self, prefix: str) -> Optional[int]:
        """4 / 8 when the layer at ``prefix`` (a fused layer or one of its projections) stays packed, else None."""
        if ".layers." not in prefix:
            return None                                  # embeddings and the output head stay 16 bit
        kinds = [self.types.get(m) for m in self._members(prefix)]
        if any(k not in _BLOCKS for k in kinds):
            return None
        return 4 if all(_BLOCKS[k][0] == 4 for k in kinds) else 8

    def get_quant_method(self, layer, prefix: str):
        from ..layers.linear import LinearBase
        from .dequant import PackedIntLinearMethod
        if not isinstance(layer, LinearBase):
            return None
        bits = self.bits_for(prefix)
        return PackedIntLinearMethod(32, bits) if bits else None


def gguf_weights_iterator(path: str, dtype: torch.dtype,
                          packed: Optional[GGUFPackedConfig] = None) -> Iterator[tuple[str, torch.Tensor]]:
    """(Hugging Face name, tensor) for every tensor of the file the model classes know: 16-bit weights, or the
    packed codes, scales and zero points of the layers ``packed`` keeps quantised."""
    r = GGUFReader(path)
    try:
        arch = r.metadata.get("general.architecture", "llama")
        heads = int(r.metadata.get(f"{arch}.attention.head_count", 0) or 0)
        kv = int(r.metadata.get(f"{arch}.attention.head_count_kv", heads) or heads)
        for name, info in r.tensors.items():
            target = hf_name(name)
            if target is None:
                continue                    # rope_freqs and other derived tensors
            permuted = arch in _PERMUTED_QK and name.endswith((".attn_q.weight", ".attn_k.weight"))
            bits = packed.bits_for(target[: -len(".weight")]) if packed is not None and target.endswith(".weight") \
                and len(info.shape) == 2 else None
            if bits:
                for suffix, t in packed_tensors(r.raw(info), info.ggml_type, info.shape, bits, dtype).items():
                    if permuted:
                        t = _unpermute(t, heads if ".attn_q." in name else kv)
                    yield f"{target[: -len('.weight')]}.{suffix}", t
                continue
            w = dequantize(r.raw(info), info.ggml_type, info.shape).to(dtype)
            if permuted:
                w = _unpermute(w, heads if ".attn_q." in name else kv)
            yield target, w
    finally:
        r.close()
