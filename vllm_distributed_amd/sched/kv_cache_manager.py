"""Per-request block bookkeeping on top of :class:`BlockPool`.

Layers (reference files in parentheses):
  * ``SingleTypeManager`` subclasses own the block lists of one KV group and
    know the reuse/skip rules of their attention type
    (vllm/v1/core/single_type_kv_cache_manager.py);
  * ``KVCacheCoordinator`` fans operations out over the groups and finds the
    longest prefix hit that every group agrees on (kv_cache_coordinator.py);
  * ``KVCacheManager`` is what the scheduler calls (kv_cache_manager.py:68-419).

Token parallelism: a request whose KV lives on another token-parallel rank is
tracked with zero blocks here (``owned=False``), so every rank runs the same
scheduler while only the owner pays for the blocks (the reference threads
``tknp_skip_allocation`` flags through the same calls: kv_cache_manager.py:139-155,203,243-245).
"""
from __future__ import annotations

from collections import defaultdict
from dataclasses import dataclass
from typing import Optional

from ..logger import init_logger
from ..metrics.stats import PrefixCacheStats
from ..parallel.utils import cdiv
from .block_pool import BlockPool, KVCacheBlock
from .kv_cache_interface import (FullAttentionSpec, KVCacheConfig, KVCacheSpec, MLASpec, MambaSpec,
                                 SlidingWindowSpec)
from .kv_cache_utils import BlockHash, extend_block_hashes, get_hash_fn

logger = init_logger(__name__)


@dataclass
class KVCacheBlocks:
    """blocks[g][i] = i-th block of KV group g."""
    blocks: tuple[list[KVCacheBlock], ...]

    def __add__(self, other: "KVCacheBlocks") -> "KVCacheBlocks":
        return KVCacheBlocks(tuple(a + b for a, b in zip(self.blocks, other.blocks)))

    def get_block_ids(self) -> tuple[list[int], ...]:
        return tuple([b.block_id for b in g] for g in self.blocks)

    def get_unhashed_block_ids(self) -> list[int]:
        assert len(self.blocks) == 1
        return [b.block_id for b in self.blocks[0] if b.block_hash is None]

    @classmethod
    def empty(cls, num_groups: int) -> "KVCacheBlocks":
        return cls(tuple([] for _ in range(num_groups)))


# --------------------------------------------------------------------------
class SingleTypeManager:
    def __init__(self, spec: KVCacheSpec, pool: BlockPool, group_id: int, caching_hash_fn):
        self.spec = spec
        self.block_size = spec.block_size
        self.pool = pool
        self.group_id = group_id
        self.hash_fn = caching_hash_fn
        self.req_to_blocks: dict[str, list[KVCacheBlock]] = defaultdict(list)
        self.num_cached_block: dict[str, int] = {}
        self._null = pool.null_block

    def get_num_blocks_to_allocate(self, request_id: str, num_tokens: int,
                                   new_computed: list[KVCacheBlock]) -> int:
        need = cdiv(num_tokens, self.block_size)
        new = need - len(new_computed) - len(self.req_to_blocks[request_id])
        # reused blocks currently sitting in the free list stop being evictable
        evictable = sum(1 for b in new_computed if b.ref_cnt == 0 and not b.is_null)
        return max(new, 0) + evictable

    def save_new_computed_blocks(self, request_id: str, new_computed: list[KVCacheBlock]) -> None:
        if request_id not in self.num_cached_block:
            blocks = self.req_to_blocks[request_id]
            assert not blocks
            blocks.extend(new_computed)
            self.num_cached_block[request_id] = len(new_computed)
        else:
            assert not new_computed, "a running request cannot gain prefix-cache blocks"

    def allocate_new_blocks(self, request_id: str, num_tokens: int) -> list[KVCacheBlock]:
        blocks = self.req_to_blocks[request_id]
        n = cdiv(num_tokens, self.block_size) - len(blocks)
        if n <= 0:
            return []
        new = self.pool.get_new_blocks(n)
        blocks.extend(new)
        return new

    def cache_blocks(self, request, block_hashes: list[BlockHash], num_tokens: int) -> None:
        rid = request.request_id
        done = self.num_cached_block.get(rid, 0)
        full = num_tokens // self.block_size
        self.pool.cache_full_blocks(request, self.req_to_blocks[rid], block_hashes, done, full,
                                    self.block_size, self.group_id, self.hash_fn)
        self.num_cached_block[rid] = max(done, full)

    def free(self, request_id: str) -> None:
        blocks = self.req_to_blocks.pop(request_id, [])
        # tail first: the tail of a sequence is the least likely to be reused
        self.pool.free_blocks(reversed(blocks))
        self.num_cached_block.pop(request_id, None)

    def get_num_common_prefix_blocks(self, request_id: str, num_running: int) -> int:
        raise NotImplementedError

    @classmethod
    def find_longest_cache_hit(cls, block_hashes, max_length, group_ids, pool, spec) -> tuple:
        raise NotImplementedError

    def remove_skipped_blocks(self, request_id: str, num_computed_tokens: int) -> None:
        raise NotImplementedError


class FullAttentionManager(SingleTypeManager):
    @classmethod
    def find_longest_cache_hit(cls, block_hashes, max_length, group_ids, pool, spec):
        out = tuple([] for _ in group_ids)
        for bh in block_hashes[: max_length // spec.block_size]:
            hit = pool.get_cached_block(bh, group_ids)
            if hit is None:
                break
            for lst, b in zip(out, hit):
                lst.append(b)
        return out

    def remove_skipped_blocks(self, request_id, num_computed_tokens):
        pass

    def get_num_common_prefix_blocks(self, request_id, num_running):
        n = 0
        for b in self.req_to_blocks[request_id]:
            if b.ref_cnt == num_running:
                n += 1
            else:
                break
        return n


class SlidingWindowManager(SingleTypeManager):
    def __init__(self, spec: SlidingWindowSpec, pool, group_id, caching_hash_fn):
        super().__init__(spec, pool, group_id, caching_hash_fn)
        self.sliding_window = spec.sliding_window

    @classmethod
    def find_longest_cache_hit(cls, block_hashes, max_length, group_ids, pool, spec):
        # need `contig` consecutive cached blocks ending at the hit point; search
        # from the right so the longest usable prefix wins
        contig = cdiv(spec.sliding_window - 1, spec.block_size)
        max_blocks = max_length // spec.block_size
        null = pool.null_block
        out = tuple([null] * max_blocks for _ in group_ids)
        run = 0
        found = False
        for i in range(max_blocks - 1, -1, -1):
            hit = pool.get_cached_block(block_hashes[i], group_ids)
            if hit is not None:
                for lst, b in zip(out, hit):
                    lst[i] = b
                run += 1
                if run >= contig:
                    for lst in out:
                        del lst[i + run:]
                    found = True
                    break
            else:
                run = 0
        if not found:
            # only a run anchored at block 0 is usable without its predecessors
            for lst in out:
                del lst[run:]
        return out

    def remove_skipped_blocks(self, request_id, num_computed_tokens):
        last_useful_token = num_computed_tokens - self.sliding_window + 1
        last_useful_block = last_useful_token // self.block_size
        blocks = self.req_to_blocks[request_id]
        removed = []
        for i in range(min(last_useful_block, len(blocks)) - 1, -1, -1):
            if blocks[i] is self._null:
                break
            removed.append(blocks[i])
            blocks[i] = self._null
        self.pool.free_blocks(removed)

    def get_num_common_prefix_blocks(self, request_id, num_running):
        return 0   # cascade attention is not used with windows


class MambaManager(SingleTypeManager):
    @classmethod
    def find_longest_cache_hit(cls, block_hashes, max_length, group_ids, pool, spec):
        return tuple([] for _ in group_ids)     # recurrent state is never shared

    def remove_skipped_blocks(self, request_id, num_computed_tokens):
        pass

    def get_num_common_prefix_blocks(self, request_id, num_running):
        return 0

    def get_num_blocks_to_allocate(self, request_id, num_tokens, new_computed):
        return 0 if self.req_to_blocks[request_id] else 1

    def allocate_new_blocks(self, request_id, num_tokens):
        blocks = self.req_to_blocks[request_id]
        if blocks:
            return []
        new = self.pool.get_new_blocks(1)
        blocks.extend(new)
        return new


_MANAGERS = {FullAttentionSpec: FullAttentionManager, MLASpec: FullAttentionManager,
             SlidingWindowSpec: SlidingWindowManager,
             MambaSpec: MambaManager}


def get_manager_for_spec(spec: KVCacheSpec, **kw) -> SingleTypeManager:
    return _MANAGERS[type(spec)](spec, **kw)


# --------------------------------------------------------------------------
class KVCacheCoordinator:
    def __init__(self, cfg: KVCacheConfig, max_model_len: int, enable_caching: bool, hash_fn,
                 enable_kv_cache_events: bool):
        self.cfg = cfg
        self.max_model_len = max_model_len
        self.pool = BlockPool(cfg.num_blocks, enable_caching, enable_kv_cache_events)
        self.managers = tuple(
            get_manager_for_spec(g.kv_cache_spec, pool=self.pool, group_id=i, caching_hash_fn=hash_fn)
            for i, g in enumerate(cfg.kv_cache_groups))
        # groups of the same type share one lookup
        self._by_type: dict[str, list[int]] = {}
        for i, g in enumerate(cfg.kv_cache_groups):
            self._by_type.setdefault(g.kv_cache_spec.type_id, []).append(i)
        types = list(self._by_type)
        if len(types) > 2:
            raise NotImplementedError("at most two attention types per model are supported")
        if len(types) == 2:
            full = [t for t in types if isinstance(
                cfg.kv_cache_groups[self._by_type[t][0]].kv_cache_spec, FullAttentionSpec)]
            if len(full) != 1:
                raise NotImplementedError("hybrid models need exactly one full-attention type")
            self._full_t = full[0]
            self._other_t = [t for t in types if t != self._full_t][0]
            bs = {cfg.kv_cache_groups[ids[0]].kv_cache_spec.block_size for ids in self._by_type.values()}
            if len(bs) != 1:
                raise NotImplementedError("hybrid groups must share a block size")

    def get_num_blocks_to_allocate(self, rid, num_tokens, new_computed) -> int:
        return sum(m.get_num_blocks_to_allocate(rid, num_tokens, new_computed[i])
                   for i, m in enumerate(self.managers))

    def save_new_computed_blocks(self, rid, new_computed) -> None:
        for i, m in enumerate(self.managers):
            m.save_new_computed_blocks(rid, new_computed[i])

    def allocate_new_blocks(self, rid, num_tokens) -> tuple:
        return tuple(m.allocate_new_blocks(rid, num_tokens) for m in self.managers)

    def cache_blocks(self, request, block_hashes, num_tokens) -> None:
        for m in self.managers:
            m.cache_blocks(request, block_hashes, num_tokens)

    def free(self, rid) -> None:
        for m in self.managers:
            m.free(rid)

    def remove_skipped_blocks(self, rid, num_computed) -> None:
        for m in self.managers:
            m.remove_skipped_blocks(rid, num_computed)

    def get_blocks(self, rid) -> tuple:
        return tuple(m.req_to_blocks.get(rid) or [] for m in self.managers)

    def get_num_common_prefix_blocks(self, rid, num_running) -> list[int]:
        return [m.get_num_common_prefix_blocks(rid, num_running) for m in self.managers]

    def find_longest_cache_hit(self, block_hashes, max_length) -> tuple[tuple, int]:
        n_groups = len(self.managers)
        if not self.managers:
            return (), 0
        if len(self._by_type) == 1:
            ids = list(range(n_groups))
            m = self.managers[0]
            hit = m.find_longest_cache_hit(block_hashes, max_length, ids, self.pool, m.spec)
            return hit, len(hit[0]) * m.block_size
        # hybrid: full attention is prefix-closed, so find its hit first, then the
        # longest window hit inside it, then trim the full hit to that length
        f_ids, o_ids = self._by_type[self._full_t], self._by_type[self._other_t]
        fm, om = self.managers[f_ids[0]], self.managers[o_ids[0]]
        f_hit = fm.find_longest_cache_hit(block_hashes, max_length, f_ids, self.pool, fm.spec)
        f_len = len(f_hit[0]) * fm.block_size
        o_hit = om.find_longest_cache_hit(block_hashes, f_len, o_ids, self.pool, om.spec)
        o_len = len(o_hit[0]) * om.block_size
        for lst in f_hit:
            del lst[o_len // fm.block_size:]
        out: list = [None] * n_groups
        for k, gid in enumerate(f_ids):
            out[gid] = f_hit[k]
        for k, gid in enumerate(o_ids):
            out[gid] = o_hit[k]
        return tuple(out), o_len


# --------------------------------------------------------------------------
class KVCacheManager:
    def __init__(self, kv_cache_config: KVCacheConfig, max_model_len: int, enable_caching: bool = True,
                 caching_hash_algo: str = "builtin", use_eagle: bool = False, log_stats: bool = False,
                 enable_kv_cache_events: bool = False):
        self.max_model_len = max_model_len
        self.enable_caching = enable_caching
        self.hash_fn = get_hash_fn(caching_hash_algo)
        self.use_eagle = use_eagle
        self.log_stats = log_stats
        self.prefix_cache_stats = PrefixCacheStats() if log_stats else None
        self.num_groups = len(kv_cache_config.kv_cache_groups)
        self.block_size = (kv_cache_config.kv_cache_groups[0].kv_cache_spec.block_size
                           if self.num_groups else None)
        self.coordinator = KVCacheCoordinator(kv_cache_config, max_model_len, enable_caching,
                                              self.hash_fn, enable_kv_cache_events)
        self.block_pool = self.coordinator.pool
        self.kv_cache_config = kv_cache_config
        self.req_to_block_hashes: dict[str, list[BlockHash]] = defaultdict(list)
        # requests whose blocks live on another token-parallel rank
        self._foreign: set[str] = set()
        # Decode steps mostly need nothing from the pool: a request's blocks already cover the next token and no
        # block fills up.  Per request: (token slots its blocks cover, first token count at which another block
        # becomes full and must be hashed).  While a step stays below both marks allocate_slots returns at once.
        # Only for layouts whose every group is plain full attention (windows release blocks as they go).
        self._covered: dict[str, tuple[int, int]] = {}
        self._fast_ok = all(type(m) is FullAttentionManager for m in self.coordinator.managers)
        self._no_blocks = self._empty()

    @property
    def usage(self) -> float:
        return self.block_pool.get_usage()

    def make_prefix_cache_stats(self) -> Optional[PrefixCacheStats]:
        if not self.log_stats:
            return None
        s, self.prefix_cache_stats = self.prefix_cache_stats, PrefixCacheStats()
        return s

    def _empty(self) -> KVCacheBlocks:
        return KVCacheBlocks.empty(self.num_groups)

    # ------------------------------------------------------------------
    def get_computed_blocks(self, request, owned: bool = True) -> tuple[KVCacheBlocks, int]:
        """Longest cached prefix of ``request`` (never the whole prompt: the last
        token must be recomputed to produce logits)."""
        sp = request.sampling_params
        if (not self.enable_caching or not owned
                or (sp is not None and sp.prompt_logprobs is not None)):
            return self._empty(), 0
        hashes = self.req_to_block_hashes[request.request_id]
        extend_block_hashes(request, hashes, request.num_tokens // self.block_size, self.block_size,
                            self.hash_fn)
        max_hit = request.num_tokens - 1
        blocks, n_tok = self.coordinator.find_longest_cache_hit(hashes, max_hit)
        if self.use_eagle and n_tok > 0:
            # the draft head needs the hidden state of the last cached token: drop one block
            blocks = tuple(b[:-1] for b in blocks)
            n_tok -= self.block_size
        if self.log_stats:
            st = self.prefix_cache_stats
            if request.num_preemptions == 0:
                st.requests += 1
                st.queries += request.num_tokens
                st.hits += n_tok
            else:
                st.preempted_requests += 1
                st.preempted_queries += request.num_tokens
                st.preempted_hits += n_tok
        return KVCacheBlocks(blocks), n_tok

    def allocate_slots(self, request, num_new_tokens: int, num_new_computed_tokens: int = 0,
                       new_computed_blocks: Optional[KVCacheBlocks] = None,
                       num_lookahead_tokens: int = 0, delay_cache_blocks: bool = False,
                       owned: bool = True) -> Optional[KVCacheBlocks]:
        """Make room for ``num_new_tokens`` more tokens of ``request``.
        Returns the newly allocated blocks, or None when the pool is exhausted."""
        if num_new_tokens <= 0:
            raise ValueError("num_new_tokens must be positive")
        rid = request.request_id
        if not owned:
            self._foreign.add(rid)
            return self._empty()
        if new_computed_blocks is None and not num_new_computed_tokens:
            mark = self._covered.get(rid)
            if mark is not None:
                total = request.num_computed_tokens + num_new_tokens
                if min(total + num_lookahead_tokens, self.max_model_len) <= mark[0] and total < mark[1]:
                    return self._no_blocks
        new_computed = new_computed_blocks.blocks if new_computed_blocks is not None \
            else tuple([] for _ in range(self.num_groups))
        # blocks that fell out of every window can be released before counting
        self.coordinator.remove_skipped_blocks(rid, request.num_computed_tokens)
        computed = request.num_computed_tokens + num_new_computed_tokens
        need_slots = min(computed + num_new_tokens + num_lookahead_tokens, self.max_model_len)
        n_alloc = self.coordinator.get_num_blocks_to_allocate(rid, need_slots, new_computed)
        if n_alloc > self.block_pool.get_num_free_blocks():
            return None
        if self.enable_caching:
            self.block_pool.touch(new_computed)
        else:
            assert not any(new_computed), "prefix hits with caching disabled"
        self.coordinator.save_new_computed_blocks(rid, new_computed)
        new_blocks = self.coordinator.allocate_new_blocks(rid, need_slots)
        if self.enable_caching and not delay_cache_blocks:
            # speculative tokens may be rejected: only hash what is certain
            certain = min(computed + num_new_tokens - len(request.spec_token_ids)
                          if request.spec_token_ids else computed + num_new_tokens,
                          request.num_tokens)
            self.coordinator.cache_blocks(request, self.req_to_block_hashes[rid], certain)
        if self._fast_ok:
            bs = self.block_size
            covered = min(len(m.req_to_blocks[rid]) for m in self.coordinator.managers) * bs
            # with caching off nothing is ever hashed; with delayed caching the caller hashes on its own
            hashed = min(m.num_cached_block.get(rid, 0) for m in self.coordinator.managers) \
                if self.enable_caching and not delay_cache_blocks else None
            self._covered[rid] = (covered, (hashed + 1) * bs if hashed is not None else (1 << 62)
                                  if not self.enable_caching else 0)
        return KVCacheBlocks(new_blocks)

    def cache_blocks(self, request, num_computed_tokens: int) -> None:
        """Hash blocks whose content arrived asynchronously (KV transfer)."""
        if self.enable_caching and request.request_id not in self._foreign:
            self.coordinator.cache_blocks(request, self.req_to_block_hashes[request.request_id],
                                          num_computed_tokens)

    def free(self, request) -> None:
        self._foreign.discard(request.request_id)
        self._covered.pop(request.request_id, None)
        self.coordinator.free(request.request_id)

    def free_block_hashes(self, request) -> None:
        self.req_to_block_hashes.pop(request.request_id, None)

    def reset_prefix_cache(self) -> bool:
        ok = self.block_pool.reset_prefix_cache()
        if ok and self.log_stats:
            self.prefix_cache_stats.reset = True
        return ok

    def get_num_common_prefix_blocks(self, request, num_running_requests: int) -> list[int]:
        assert request.status.name == "RUNNING"
        return self.coordinator.get_num_common_prefix_blocks(request.request_id, num_running_requests)

    def get_blocks(self, request_id: str) -> KVCacheBlocks:
        return KVCacheBlocks(self.coordinator.get_blocks(request_id))

    def get_block_ids(self, request_id: str) -> tuple[list[int], ...]:
        return self.get_blocks(request_id).get_block_ids()

    def get_num_free_blocks(self) -> int:
        return self.block_pool.get_num_free_blocks()

    def take_events(self):
        return self.block_pool.take_events()
