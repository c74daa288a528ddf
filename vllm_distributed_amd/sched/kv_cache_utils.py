"""Prefix-cache hashing and the memory -> KVCacheConfig planner.

Reference: vllm/v1/core/kv_cache_utils.py (hash chain with extra keys for
multimodal items, LoRA id and cache salt :414-480; planner :644-722 and
:927-999).  Hashes are chained: the hash of block *i* covers the hash of block
*i-1*, so equal hashes mean equal whole prefixes.
"""
from __future__ import annotations

import hashlib
import os
import pickle
from typing import Any, Callable, NamedTuple, Optional

from ..logger import init_logger
from ..parallel.utils import cdiv
from .kv_cache_interface import (FullAttentionSpec, KVCacheConfig, KVCacheGroupSpec, KVCacheSpec,
                                 KVCacheTensor, SlidingWindowSpec)

logger = init_logger(__name__)


class BlockHash(NamedTuple):
    hash_value: int
    token_ids: tuple
    extra_keys: Optional[Any] = None


# Python's hash() of tuples is salted per process for str members only; ints are
# stable.  The first block's parent is a fixed seed unless PYTHONHASHSEED is set,
# matching the "builtin is fast, sha256 is collision-safe" trade-off.
NONE_HASH: int = int.from_bytes(os.urandom(8), "little") if os.getenv("PYTHONHASHSEED") is None \
    else hash(("vda-none-hash", os.getenv("PYTHONHASHSEED")))


def _hash_builtin(x) -> int:
    return hash(x)


def _hash_sha256(x) -> int:
    return int.from_bytes(hashlib.sha256(pickle.dumps(x, protocol=4)).digest()[:16], "little")


def get_hash_fn(name: str) -> Callable:
    return _hash_sha256 if name == "sha256" else _hash_builtin


def hash_block_tokens(hash_fn, parent_hash: Optional[int], token_ids, extra_keys=None) -> BlockHash:
    if parent_hash is None:
        parent_hash = NONE_HASH
    t = tuple(token_ids)
    return BlockHash(hash_fn((parent_hash, t, extra_keys)), t, extra_keys)


def _mm_extra_keys(request, start: int, end: int, next_idx: int) -> tuple[list, int]:
    """Hashes of multimodal items overlapping token range [start, end)."""
    keys: list = []
    pos = request.mm_positions
    if not pos:
        return keys, next_idx
    if len(request.mm_hashes) != len(pos):
        raise ValueError("every multimodal item needs a content hash for prefix caching")
    last = pos[-1]
    if last.offset + last.length <= start:
        return keys, next_idx
    i = next_idx
    while i < len(pos):
        off, ln = pos[i].offset, pos[i].length
        if end <= off:
            break
        if off + ln > start:
            keys.append(request.mm_hashes[i])
            if end < off + ln:
                break      # item continues into the next block: revisit it there
        i += 1
    return keys, i


def generate_extra_keys(request, start: int, end: int, next_mm_idx: int) -> tuple[Optional[tuple], int]:
    mm, next_mm_idx = _mm_extra_keys(request, start, end, next_mm_idx)
    lora = [request.lora_request.lora_int_id] if request.lora_request else []
    salt = [request.cache_salt] if (start == 0 and request.cache_salt) else []
    keys = lora + mm + salt
    return (tuple(keys) if keys else None), next_mm_idx


def extend_block_hashes(request, block_hashes: list, upto_blocks: int, block_size: int,
                        hash_fn: Optional[Callable]) -> None:
    """Append hashes so that ``block_hashes`` covers the request's first
    ``upto_blocks`` full blocks."""
    hash_fn = hash_fn or _hash_builtin
    toks = request.all_token_ids
    n = len(block_hashes)
    upto_blocks = min(upto_blocks, len(toks) // block_size)
    parent = block_hashes[-1].hash_value if n else None
    mm_idx = 0
    while n < upto_blocks:
        s, e = n * block_size, (n + 1) * block_size
        extra, mm_idx = generate_extra_keys(request, s, e, mm_idx)
        bh = hash_block_tokens(hash_fn, parent, toks[s:e], extra)
        block_hashes.append(bh)
        parent = bh.hash_value
        n += 1


def hash_request_tokens(hash_fn, block_size: int, request) -> list[BlockHash]:
    out: list[BlockHash] = []
    extend_block_hashes(request, out, request.num_tokens // block_size, block_size, hash_fn)
    return out


# --------------------------------------------------------------------------
# planner
# --------------------------------------------------------------------------
def _check_enough_memory(vllm_config, spec: dict[str, KVCacheSpec], available: int) -> None:
    if available <= 0:
        raise ValueError("no memory left for the KV cache; raise gpu_memory_utilization "
                         "or lower max_model_len / max_num_seqs")
    need = sum(s.max_memory_usage_bytes(vllm_config.model_config.max_model_len,
                                        vllm_config.scheduler_config.max_num_batched_tokens)
               for s in spec.values())
    if need > available:
        per_tok = need / vllm_config.model_config.max_model_len
        fit = int(available / per_tok) if per_tok else 0
        raise ValueError(
            f"one request of max_model_len={vllm_config.model_config.max_model_len} needs "
            f"{need / 2**30:.2f} GiB of KV cache but only {available / 2**30:.2f} GiB is available "
            f"(about {fit} tokens would fit); lower max_model_len or raise gpu_memory_utilization")


def _uniform_type(spec: dict[str, KVCacheSpec]) -> bool:
    return len({s.type_id for s in spec.values()}) == 1


def _uniform_page(spec: dict[str, KVCacheSpec]) -> bool:
    return len({s.page_size_bytes for s in spec.values()}) == 1


def unify_hybrid_kv_cache_specs(spec: dict[str, KVCacheSpec]) -> None:
    """If sliding-window and full-attention layers are mixed but hybrid
    allocation is not wanted, treat the windowed layers as full attention."""
    has_full = any(isinstance(s, FullAttentionSpec) for s in spec.values())
    has_swa = any(isinstance(s, SlidingWindowSpec) for s in spec.values())
    if has_full and has_swa:
        for name, s in list(spec.items()):
            if isinstance(s, SlidingWindowSpec):
                spec[name] = FullAttentionSpec(block_size=s.block_size, num_kv_heads=s.num_kv_heads,
                                               head_size=s.head_size, dtype=s.dtype, use_mla=s.use_mla,
                                               sliding_window=s.sliding_window)


def _num_blocks(vllm_config, available: int, page_size: int, num_layers: int) -> int:
    n = int(available // page_size // max(num_layers, 1))
    n = max(n, 0)
    override = vllm_config.cache_config.num_gpu_blocks_override
    if override is not None:
        logger.info("overriding num KV blocks %d -> %d", n, override)
        n = override
    return n


def get_kv_cache_config(vllm_config, spec: dict[str, KVCacheSpec], available: int) -> KVCacheConfig:
    """Turn a per-layer spec + a byte budget into tensors, groups and a block count."""
    if not spec:
        return KVCacheConfig(num_blocks=1, kv_cache_tensors=[], kv_cache_groups=[])
    if vllm_config.cache_config.num_gpu_blocks_override is None:
        _check_enough_memory(vllm_config, spec, available)
    if not _uniform_type(spec):
        unify_hybrid_kv_cache_specs(spec)
    names = list(spec)
    if _uniform_type(spec):
        page = spec[names[0]].page_size_bytes
        nb = _num_blocks(vllm_config, available, page, len(names))
        tensors = [KVCacheTensor(size=page * nb, shared_by=[n]) for n in names]
        groups = [KVCacheGroupSpec(names, spec[names[0]])]
    elif all(type(s_) is FullAttentionSpec and not s_.use_mla for s_ in spec.values()) \
            and len({s_.block_size for s_ in spec.values()}) == 1:
        # full attention everywhere, but layers keep different numbers of KV heads (DeciLM / Nemotron-NAS): one
        # group and one block table, every layer's tensor sized by its own page
        total = sum(spec[n].page_size_bytes for n in names)
        nb = _num_blocks(vllm_config, available, total, 1)
        tensors = [KVCacheTensor(size=spec[n].page_size_bytes * nb, shared_by=[n]) for n in names]
        groups = [KVCacheGroupSpec(names, max(spec.values(), key=lambda s_: s_.page_size_bytes),
                                   per_layer={n: spec[n] for n in names})]
    elif _uniform_page(spec):
        # hybrid model: group layers by type, pad groups to equal length, and let
        # the i-th layer of every group share one allocation
        by_type: dict[str, list[str]] = {}
        for n in names:
            by_type.setdefault(spec[n].type_id, []).append(n)
        group_size = min(len(v) for v in by_type.values())
        grouped: list[list[str]] = []
        for layers in by_type.values():
            for i in range(0, len(layers), group_size):
                grouped.append(layers[i:i + group_size])
        page = spec[names[0]].page_size_bytes
        nb = _num_blocks(vllm_config, available, page, group_size)
        tensors = []
        for i in range(group_size):
            shared = [g[i] for g in grouped if i < len(g)]
            tensors.append(KVCacheTensor(size=page * nb, shared_by=shared))
        groups = [KVCacheGroupSpec(g, spec[g[0]]) for g in grouped]
    else:
        raise NotImplementedError("layers with different KV page sizes are not supported")
    tokens = nb * groups[0].kv_cache_spec.block_size
    max_len = vllm_config.model_config.max_model_len
    logger.info("KV cache: %d blocks = %s tokens, %.2fx concurrency at %d tokens per request",
                nb, f"{tokens:,}", tokens / max_len, max_len)
    return KVCacheConfig(num_blocks=nb, kv_cache_tensors=tensors, kv_cache_groups=groups)


def unify_kv_cache_configs(configs: list[KVCacheConfig]) -> None:
    """Workers may have different free memory: shrink all to the smallest block count."""
    ref = sorted(g.kv_cache_spec.type_id for g in configs[0].kv_cache_groups)
    for c in configs[1:]:
        if sorted(g.kv_cache_spec.type_id for g in c.kv_cache_groups) != ref:
            raise ValueError("workers disagree on the KV cache group layout")
    nb = min(c.num_blocks for c in configs)
    for c in configs:
        if c.num_blocks != nb:
            for t in c.kv_cache_tensors:
                t.size = t.size // c.num_blocks * nb
            c.num_blocks = nb


def get_max_concurrency(vllm_config, cfg: KVCacheConfig) -> float:
    spec = cfg.kv_cache_groups[0].kv_cache_spec
    per_req = cdiv(spec.max_memory_usage_bytes(vllm_config.model_config.max_model_len,
                                               vllm_config.scheduler_config.max_num_batched_tokens),
                   spec.page_size_bytes)
    return cfg.num_blocks / max(per_req, 1)
