"""KV-cache offloading to host memory: a second, larger prefix cache behind the one in HBM.

Every full block of a finished prompt is copied into a pool of pinned host pages, keyed by the chained hash of
the tokens it covers (so any prompt that shares a block-aligned prefix finds it, not only an identical prompt);
a later request whose prefix is no longer in HBM gets the blocks back with one host -> device copy per layer
instead of recomputing them.  512 GB and more of host memory beside each MI355X hold an order of magnitude more
context than the 288 GB of HBM that also has to hold the weights.

This is the role the reference gives to the external LMCache package
(vllm/distributed/kv_transfer/kv_connector/v1/lmcache_connector.py, a thin adapter around
``lmcache.integration.vllm``) and, in its older engine, to the Mooncake store and the lookup-buffer connectors
(vllm/distributed/kv_transfer/kv_lookup_buffer/*): a store of KV chunks by token-prefix hash outside the GPU.
Neither package is part of this build; the store here is in-tree and registered under those connector names too.

Division of labour (the scheduler and the workers may be different processes):
  * the SCHEDULER-side object owns the index: hash -> host slot, least-recently-used order, slot allocation and
    eviction.  Per step it tells the workers which (host slot, device block) pairs to load before the forward
    pass and which to store after it.
  * every WORKER-side object owns a pool with the same slot numbering: per attention layer one pinned K and one
    pinned V tensor of ``slots`` cache blocks (its tensor-parallel shard), and only executes the copies.
Copies are issued on the compute stream (pinned memory, non-blocking), so they are ordered with the kernels that
produce and consume the blocks; the only host synchronisation is one per step that stores blocks.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Optional

import torch

from ...logger import init_logger
from ...sched.kv_cache_utils import get_hash_fn, hash_block_tokens
from .base import KVConnectorBase, KVConnectorMetadata, KVConnectorRole

logger = init_logger(__name__)


@dataclass
class HostOffloadMetadata(KVConnectorMetadata):
    loads: list[tuple[int, int]] = field(default_factory=list)     # (host slot, device block)
    stores: list[tuple[int, int]] = field(default_factory=list)    # (host slot, device block)


class HostOffloadConnector(KVConnectorBase):
    def __init__(self, vllm_config, role: KVConnectorRole):
        super().__init__(vllm_config, role)
        kt = vllm_config.kv_transfer_config
        extra = kt.kv_connector_extra_config or {}
        self.block_size = vllm_config.cache_config.block_size
        self.num_slots = int(extra.get("host_blocks", 0))
        self.host_gib = float(extra.get("host_gib", 4.0))           # used when host_blocks is not given
        self.hash_fn = get_hash_fn(getattr(vllm_config.cache_config, "prefix_caching_hash_algo", "builtin"))
        # scheduler side
        self.index: "OrderedDict[int, int]" = OrderedDict()         # block hash -> slot, oldest first
        self.free: list[int] = []
        self._slots_known = False
        self._loads: list[tuple[int, int]] = []
        self._pinned: set[int] = set()                              # slots read by a load not yet issued
        self._storing: dict[str, list] = {}                         # req id -> [prompt tokens, block ids]
        self._matched: dict[str, tuple[int, list[int]]] = {}        # req id -> (first block index, hashes found)
        self._identities: dict[str, tuple] = {}                     # req id -> (lora id, cache salt) when either is set
        self.stats = {"hit_blocks": 0, "stored_blocks": 0, "evicted_blocks": 0}
        # worker side
        self.runner = None
        self.pool: dict[str, tuple[torch.Tensor, torch.Tensor]] = {}
        self._staged: list = []

    # ---- sizing -----------------------------------------------------------------------------------------------
    def _ensure_slots(self) -> None:
        if self._slots_known:
            return
        if self.num_slots <= 0:
            # scheduler side: size from the model's KV bytes per block (all layers, this rank's shard unknown here:
            # use the whole model's, the workers allocate the same number of slots of their own shard size)
            mc = self.vllm_config.model_config
            per_tok = 2 * mc.get_num_layers(self.vllm_config.parallel_config) * mc.get_total_num_kv_heads() \
                * mc.get_head_size() * 2
            self.num_slots = max(int(self.host_gib * (1 << 30) // (per_tok * self.block_size)), 1)
        self.free = list(range(self.num_slots - 1, -1, -1))
        self._slots_known = True

    # ---- worker -----------------------------------------------------------------------------------------------
    def register_kv_caches(self, runner) -> None:
        self.runner = runner
        self._ensure_slots()
        pin = torch.cuda.is_available()
        total = 0
        for name, layer in runner.attn_layers.items():
            kc, vc = getattr(layer, "k_cache", None), getattr(layer, "v_cache", None)
            if kc is None or vc is None:
                continue                                    # state-space layers and latent caches are not offloaded
            k = torch.empty((self.num_slots,) + tuple(kc.shape[1:]), dtype=kc.dtype, pin_memory=pin and kc.is_cuda)
            v = torch.empty((self.num_slots,) + tuple(vc.shape[1:]), dtype=vc.dtype, pin_memory=pin and vc.is_cuda)
            self.pool[name] = (k, v)
            total += k.numel() * k.element_size() + v.numel() * v.element_size()
        logger.info("host KV pool: %d blocks of %d tokens, %.2f GiB%s", self.num_slots, self.block_size, total / 2**30,
                    " (pinned)" if pin else "")

    def start_load_kv(self) -> None:
        meta = self._meta
        if not isinstance(meta, HostOffloadMetadata) or not meta.loads or self.runner is None:
            return
        slots = torch.tensor([s for s, _ in meta.loads], dtype=torch.int64)
        for name, (hk, hv) in self.pool.items():
            layer = self.runner.attn_layers[name]
            dev = layer.k_cache.device
            blocks = torch.tensor([b for _, b in meta.loads], dtype=torch.int64, device=dev)
            layer.k_cache.index_copy_(0, blocks, hk[slots].to(dev, non_blocking=True))
            layer.v_cache.index_copy_(0, blocks, hv[slots].to(dev, non_blocking=True))

    def save_kv_layer(self, layer_name: str, kv_layer: tuple, attn_metadata) -> None:
        meta = self._meta
        if not isinstance(meta, HostOffloadMetadata) or not meta.stores or layer_name not in self.pool:
            return
        k_cache, v_cache = kv_layer
        hk, hv = self.pool[layer_name]
        slots = torch.tensor([s for s, _ in meta.stores], dtype=torch.int64)
        blocks = torch.tensor([b for _, b in meta.stores], dtype=torch.int64, device=k_cache.device)
        # device gather -> pinned staging (asynchronous, ordered behind the kernels that wrote the blocks); the
        # staging rows move into their pool slots in wait_for_save, after ONE stream synchronisation for the step
        n = len(meta.stores)
        pin = k_cache.is_cuda
        sk = torch.empty((n,) + tuple(k_cache.shape[1:]), dtype=k_cache.dtype, pin_memory=pin)
        sv = torch.empty((n,) + tuple(v_cache.shape[1:]), dtype=v_cache.dtype, pin_memory=pin)
        sk.copy_(k_cache.index_select(0, blocks), non_blocking=True)
        sv.copy_(v_cache.index_select(0, blocks), non_blocking=True)
        self._staged.append((hk, hv, slots, sk, sv))

    def wait_for_save(self) -> None:
        if not self._staged:
            return
        if self._staged[0][3].is_pinned():
            torch.cuda.current_stream().synchronize()
        for hk, hv, slots, sk, sv in self._staged:
            hk[slots] = sk
            hv[slots] = sv
        self._staged = []

    # ---- scheduler --------------------------------------------------------------------------------------------
    @staticmethod
    def _identity(request) -> Optional[tuple]:
        """What besides the tokens decides a block's contents: the LoRA adapter that computed it and the request's
        cache salt (the same extra keys as the device prefix cache, sched/kv_cache_utils.generate_extra_keys)."""
        lora = getattr(request, "lora_request", None)
        salt = getattr(request, "cache_salt", None)
        if lora is None and not salt:
            return None
        return (lora.lora_int_id if lora is not None else 0, salt or "")

    def _prompt_hashes(self, token_ids, identity: Optional[tuple] = None) -> list[int]:
        """Chained hashes of the full blocks of the prompt that may be reused (the last token is always computed).
        ``identity`` seeds the chain, so equal prompts under different adapters or salts never share blocks."""
        n = (len(token_ids) - 1) // self.block_size
        out, parent = [], (self.hash_fn(("host_offload", identity)) if identity is not None else None)
        for i in range(n):
            h = hash_block_tokens(self.hash_fn, parent, token_ids[i * self.block_size:(i + 1) * self.block_size])
            out.append(h.hash_value)
            parent = h.hash_value
        return out

    def get_num_new_matched_tokens(self, request, num_computed_tokens: int) -> tuple[int, bool]:
        self._ensure_slots()
        if getattr(request, "mm_inputs", None):
            return 0, False                             # placeholder tokens do not identify multimodal content
        ident = self._identity(request)
        if ident is not None:
            self._identities[request.request_id] = ident     # the store side sees only NewRequestData
        hashes = self._prompt_hashes(request.prompt_token_ids, ident)
        i0 = num_computed_tokens // self.block_size
        n = 0
        while i0 + n < len(hashes) and hashes[i0 + n] in self.index:
            n += 1
        self._matched[request.request_id] = (i0, [hashes[i0 + j] for j in range(n)])
        return n * self.block_size, False

    def update_state_after_alloc(self, request, blocks, num_external_tokens: int) -> None:
        i0, hashes = self._matched.pop(request.request_id, (0, []))
        if num_external_tokens <= 0 or not hashes:
            return
        ids = list(blocks.get_block_ids()[0]) if hasattr(blocks, "get_block_ids") else list(blocks)
        for j, h in enumerate(hashes[: num_external_tokens // self.block_size]):
            slot = self.index.get(h)
            if slot is None or i0 + j >= len(ids):
                raise RuntimeError("host KV index changed between the lookup and the allocation of a request")
            self.index.move_to_end(h)
            self._loads.append((slot, ids[i0 + j]))
            self._pinned.add(slot)
            self.stats["hit_blocks"] += 1

    def _take_slot(self) -> int:
        if self.free:
            return self.free.pop()
        for h, s in self.index.items():                 # oldest first
            if s not in self._pinned:
                del self.index[h]
                self.stats["evicted_blocks"] += 1
                return s
        return -1

    def build_connector_meta(self, so) -> HostOffloadMetadata:
        self._ensure_slots()
        meta = HostOffloadMetadata(loads=self._loads)
        self._loads = []

        def flat(ids):
            return list(ids[0]) if ids and isinstance(ids[0], (list, tuple)) else list(ids or [])

        computed = {}
        for nr in so.scheduled_new_reqs:
            if not getattr(nr, "mm_inputs", None):
                self._storing[nr.req_id] = [list(nr.prompt_token_ids), flat(nr.block_ids)]
            computed[nr.req_id] = nr.num_computed_tokens
        cr = so.scheduled_cached_reqs
        for i, rid in enumerate(cr.req_ids):
            st = self._storing.get(rid)
            if st is None:
                continue
            if cr.resumed_from_preemption[i]:
                st[1] = flat(cr.new_block_ids[i])
            else:
                st[1].extend(flat(cr.new_block_ids[i]))
            computed[rid] = cr.num_computed_tokens[i]
        for rid in list(self._storing):
            if rid in so.finished_req_ids:
                del self._storing[rid]
                self._identities.pop(rid, None)
                continue
            n = so.num_scheduled_tokens.get(rid)
            if n is None or computed.get(rid, 0) + n < len(self._storing[rid][0]):
                continue
            toks, ids = self._storing.pop(rid)          # the step that completes the prompt: store what is new
            for j, h in enumerate(self._prompt_hashes(toks, self._identities.pop(rid, None))):
                if h in self.index:
                    self.index.move_to_end(h)
                    continue
                slot = self._take_slot()
                if slot < 0 or j >= len(ids):
                    break
                self.index[h] = slot
                meta.stores.append((slot, ids[j]))
                self.stats["stored_blocks"] += 1
        self._pinned.clear()
        return meta
