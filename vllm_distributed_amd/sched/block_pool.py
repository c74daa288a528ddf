"""Reference-counted KV blocks with an eviction-ordered free list and a
hash -> block index for prefix reuse.

Behaviour follows the reference's pool (vllm/v1/core/block_pool.py:19-349,
vllm/v1/core/kv_cache_utils.py:126-280): freed blocks keep their hash and stay
findable until they are handed out again; eviction order is "least recently
freed first, and within one request tail blocks first"; block 0 is a
never-freed null block used as a placeholder for skipped (out-of-window)
positions.  The free list is an index-linked list over two int arrays rather
than linked Python objects, which keeps the 100k+ block pools of a 288 GB
device cheap to build and walk.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Callable, Iterable, Optional

import numpy as np

from .kv_events import AllBlocksCleared, BlockRemoved, BlockStored, KVCacheEvent


class KVCacheBlock:
    __slots__ = ("block_id", "ref_cnt", "block_hash", "is_null")

    def __init__(self, block_id: int):
        self.block_id = block_id
        self.ref_cnt = 0
        self.block_hash = None     # BlockHashWithGroupId once the block is full and cached
        self.is_null = False

    def __repr__(self):
        return f"KVCacheBlock(id={self.block_id}, ref={self.ref_cnt}, hashed={self.block_hash is not None})"


class FreeBlockQueue:
    """Doubly linked list of free block ids, O(1) pop-front / append / remove."""

    _NIL = -1

    def __init__(self, num_blocks: int, skip: Iterable[int] = ()):
        self._prev = np.full(num_blocks, self._NIL, dtype=np.int64)
        self._next = np.full(num_blocks, self._NIL, dtype=np.int64)
        self._in = np.zeros(num_blocks, dtype=bool)
        self._head = self._tail = self._NIL
        self.num_free_blocks = 0
        skip = set(skip)
        ids = [i for i in range(num_blocks) if i not in skip]
        if ids:
            a = np.asarray(ids, dtype=np.int64)
            self._prev[a[1:]] = a[:-1]
            self._next[a[:-1]] = a[1:]
            self._in[a] = True
            self._head, self._tail = int(a[0]), int(a[-1])
            self.num_free_blocks = len(ids)

    def popleft(self) -> int:
        if self._head == self._NIL:
            raise ValueError("no free KV blocks")
        b = self._head
        self.remove(b)
        return b

    def popleft_n(self, n: int) -> list[int]:
        if n > self.num_free_blocks:
            raise ValueError(f"asked for {n} blocks, {self.num_free_blocks} free")
        out = []
        b = self._head
        for _ in range(n):
            out.append(b)
            b = int(self._next[b])
        if out:
            last = out[-1]
            self._head = b
            if b == self._NIL:
                self._tail = self._NIL
            else:
                self._prev[b] = self._NIL
            arr = np.asarray(out, dtype=np.int64)
            self._prev[arr] = self._NIL
            self._next[arr] = self._NIL
            self._in[arr] = False
            self.num_free_blocks -= len(out)
        return out

    def remove(self, b: int) -> None:
        if not self._in[b]:
            raise RuntimeError(f"block {b} is not in the free list")
        p, n = int(self._prev[b]), int(self._next[b])
        if p != self._NIL:
            self._next[p] = n
        else:
            self._head = n
        if n != self._NIL:
            self._prev[n] = p
        else:
            self._tail = p
        self._prev[b] = self._next[b] = self._NIL
        self._in[b] = False
        self.num_free_blocks -= 1

    def append(self, b: int) -> None:
        if self._in[b]:
            raise RuntimeError(f"block {b} is already free")
        self._prev[b] = self._tail
        self._next[b] = self._NIL
        if self._tail != self._NIL:
            self._next[self._tail] = b
        else:
            self._head = b
        self._tail = b
        self._in[b] = True
        self.num_free_blocks += 1

    def append_n(self, ids: list[int]) -> None:
        for b in ids:
            self.append(b)

    def __contains__(self, b: int) -> bool:
        return bool(self._in[b])

    def get_all_free_blocks(self) -> list[int]:
        out, b = [], self._head
        while b != self._NIL:
            out.append(b)
            b = int(self._next[b])
        return out


class NativeFreeBlockQueue:
    """The same queue kept by the host runtime library (runtime/csrc/block_pool.cpp): one call pops or appends a
    request's whole list of blocks.  Used when the library is built and ``VDA_NATIVE_BLOCK_POOL`` is on."""

    def __init__(self, num_blocks: int, skip: Iterable[int] = ()):
        import ctypes
        from ..runtime.shm_ring import host_library
        lib = host_library()
        c64, vp = ctypes.c_int64, ctypes.c_void_p
        lib.vda_fbq_create.restype = vp
        lib.vda_fbq_create.argtypes = [c64, vp, c64]
        lib.vda_fbq_destroy.argtypes = [vp]
        lib.vda_fbq_count.restype = c64
        lib.vda_fbq_count.argtypes = [vp]
        lib.vda_fbq_contains.argtypes = [vp, c64]
        lib.vda_fbq_popleft_n.restype = c64
        lib.vda_fbq_popleft_n.argtypes = [vp, c64, vp]
        lib.vda_fbq_remove.argtypes = [vp, c64]
        lib.vda_fbq_append_n.restype = c64
        lib.vda_fbq_append_n.argtypes = [vp, vp, c64]
        lib.vda_fbq_list.restype = c64
        lib.vda_fbq_list.argtypes = [vp, vp]
        self._lib = lib
        sk = np.asarray(sorted(set(skip)), dtype=np.int64)
        self._h = lib.vda_fbq_create(num_blocks, sk.ctypes.data, len(sk))
        if not self._h:
            raise RuntimeError("could not create the native free list")
        self._one = np.zeros(1, dtype=np.int64)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.vda_fbq_destroy(h)

    @property
    def num_free_blocks(self) -> int:
        return int(self._lib.vda_fbq_count(self._h))

    def popleft(self) -> int:
        if self._lib.vda_fbq_popleft_n(self._h, 1, self._one.ctypes.data) != 1:
            raise ValueError("no free KV blocks")
        return int(self._one[0])

    def popleft_n(self, n: int) -> list[int]:
        out = np.empty(max(n, 1), dtype=np.int64)
        if self._lib.vda_fbq_popleft_n(self._h, n, out.ctypes.data) != n:
            raise ValueError(f"asked for {n} blocks, {self.num_free_blocks} free")
        return out[:n].tolist()

    def remove(self, b: int) -> None:
        if self._lib.vda_fbq_remove(self._h, b):
            raise RuntimeError(f"block {b} is not in the free list")

    def append(self, b: int) -> None:
        self._one[0] = b
        if self._lib.vda_fbq_append_n(self._h, self._one.ctypes.data, 1) != 1:
            raise RuntimeError(f"block {b} is already free")

    def append_n(self, ids: list[int]) -> None:
        if not ids:
            return
        a = np.asarray(ids, dtype=np.int64)
        rc = self._lib.vda_fbq_append_n(self._h, a.ctypes.data, len(a))
        if rc != len(a):
            raise RuntimeError(f"block {ids[-1 - rc]} is already free")

    def __contains__(self, b: int) -> bool:
        return bool(self._lib.vda_fbq_contains(self._h, b))

    def get_all_free_blocks(self) -> list[int]:
        out = np.empty(max(self.num_free_blocks, 1), dtype=np.int64)
        n = self._lib.vda_fbq_list(self._h, out.ctypes.data)
        return out[:n].tolist()


def make_free_queue(num_blocks: int, skip: Iterable[int] = ()):
    from .. import envs
    if envs.VDA_NATIVE_BLOCK_POOL:
        try:
            return NativeFreeBlockQueue(num_blocks, skip)
        except Exception:   # noqa: BLE001 - library not built: the numpy form serves the same contract
            pass
    return FreeBlockQueue(num_blocks, skip)


class BlockPool:
    def __init__(self, num_gpu_blocks: int, enable_caching: bool, enable_kv_cache_events: bool = False):
        if num_gpu_blocks < 2:
            raise ValueError("need at least 2 KV blocks (one is the null block)")
        self.num_gpu_blocks = num_gpu_blocks
        self.enable_caching = enable_caching
        self.blocks = [KVCacheBlock(i) for i in range(num_gpu_blocks)]
        self.null_block = self.blocks[0]
        self.null_block.is_null = True
        self.free_queue = make_free_queue(num_gpu_blocks, skip=(0,))
        # hash -> {block_id: block}; several blocks may hold identical content
        self.cached: dict = defaultdict(dict)
        self.enable_kv_cache_events = enable_kv_cache_events
        self._events: list[KVCacheEvent] = []

    # lookup ---------------------------------------------------------------
    def get_cached_block(self, block_hash, kv_cache_group_ids: list[int]) -> Optional[list[KVCacheBlock]]:
        """One cached block per group for this hash, or None if any group misses."""
        out = []
        for gid in kv_cache_group_ids:
            hit = self.cached.get((block_hash, gid))
            if not hit:
                return None
            out.append(next(iter(hit.values())))
        return out

    # caching --------------------------------------------------------------
    def cache_full_blocks(self, request, blocks: list[KVCacheBlock], block_hashes: list,
                          num_cached_blocks: int, num_full_blocks: int, block_size: int,
                          kv_cache_group_id: int, hash_fn: Optional[Callable] = None) -> None:
        """Give blocks ``[num_cached_blocks, num_full_blocks)`` of the request their
        content hash and publish them in the index.  ``block_hashes`` is the
        request's own hash chain and is extended lazily here."""
        if num_cached_blocks >= num_full_blocks:
            return
        from .kv_cache_utils import extend_block_hashes
        extend_block_hashes(request, block_hashes, num_full_blocks, block_size, hash_fn)
        new_hashes = []
        parent = block_hashes[num_cached_blocks - 1] if num_cached_blocks > 0 else None
        for i in range(num_cached_blocks, num_full_blocks):
            blk = blocks[i]
            if blk.is_null:
                continue
            assert blk.block_hash is None, f"{blk} already cached"
            key = (block_hashes[i], kv_cache_group_id)
            blk.block_hash = key
            self.cached[key][blk.block_id] = blk
            new_hashes.append(block_hashes[i].hash_value)
        if self.enable_kv_cache_events and new_hashes:
            s, e = num_cached_blocks * block_size, num_full_blocks * block_size
            self._events.append(BlockStored(
                block_hashes=new_hashes,
                parent_block_hash=None if parent is None else parent.hash_value,
                token_ids=list(request.all_token_ids[s:e]), block_size=block_size,
                lora_id=request.lora_request.lora_int_id if request.lora_request else None))

    # allocation -----------------------------------------------------------
    def get_new_blocks(self, n: int) -> list[KVCacheBlock]:
        if n > self.get_num_free_blocks():
            raise ValueError(f"cannot allocate {n} blocks, {self.get_num_free_blocks()} free")
        out = []
        for bid in self.free_queue.popleft_n(n):
            blk = self.blocks[bid]
            assert blk.ref_cnt == 0
            if self.enable_caching:
                self._maybe_evict(blk)
            blk.ref_cnt = 1
            out.append(blk)
        return out

    def _maybe_evict(self, blk: KVCacheBlock) -> bool:
        key = blk.block_hash
        if key is None:
            return False
        bucket = self.cached.get(key)
        if bucket is not None and blk.block_id in bucket:
            del bucket[blk.block_id]
            if not bucket:
                del self.cached[key]
            if self.enable_kv_cache_events:
                self._events.append(BlockRemoved(block_hashes=[key[0].hash_value]))
        blk.block_hash = None
        return True

    def touch(self, blocks: Iterable[Iterable[KVCacheBlock]]) -> None:
        """A request reuses these cached blocks: take them off the free list."""
        for group in blocks:
            for blk in group:
                if blk.ref_cnt == 0 and not blk.is_null:
                    self.free_queue.remove(blk.block_id)
                blk.ref_cnt += 1

    def free_blocks(self, ordered: Iterable[KVCacheBlock]) -> None:
        """Drop one reference from each block; blocks that reach zero are appended
        to the free list in the given order (callers pass tail blocks first)."""
        freed = []
        for blk in ordered:
            if blk.is_null:
                continue
            blk.ref_cnt -= 1
            assert blk.ref_cnt >= 0, f"double free of {blk}"
            if blk.ref_cnt == 0:
                freed.append(blk.block_id)
        if freed:
            self.free_queue.append_n(freed)

    def reset_prefix_cache(self) -> bool:
        used = self.num_gpu_blocks - self.get_num_free_blocks()
        if used != 1:   # the null block is always "used"
            return False
        self.cached = defaultdict(dict)
        for b in self.blocks:
            b.block_hash = None
        if self.enable_kv_cache_events:
            self._events.append(AllBlocksCleared())
        return True

    # stats ----------------------------------------------------------------
    def get_num_free_blocks(self) -> int:
        return self.free_queue.num_free_blocks

    def get_usage(self) -> float:
        """Fraction of the allocatable blocks in use (the null block is never handed out and not counted)."""
        usable = max(self.num_gpu_blocks - 1, 1)
        return max(0.0, 1.0 - self.get_num_free_blocks() / usable)

    def take_events(self) -> list[KVCacheEvent]:
        ev, self._events = self._events, []
        return ev
