"""Prefix KV on a shared file system: one safetensors file per (prompt prefix hash, layer).
A producer saves the KV of every fully prefilled prompt; any instance that later sees a prompt
with the same block-aligned prefix loads it instead of computing it.  Synchronous and simple:
the debugging / functional-test connector.
Reference: vllm/distributed/kv_transfer/kv_connector/v1/shared_storage_connector.py."""
from __future__ import annotations

import hashlib
import os
from dataclasses import dataclass, field

import numpy as np
import torch
from safetensors.torch import load_file, save_file

from ...logger import init_logger
from .base import KVConnectorBase, KVConnectorMetadata, KVConnectorRole

logger = init_logger(__name__)


@dataclass
class _Req:
    token_ids: list[int]
    block_ids: list[int]
    is_store: bool


@dataclass
class SharedStorageMetadata(KVConnectorMetadata):
    requests: list[_Req] = field(default_factory=list)


class SharedStorageConnector(KVConnectorBase):
    def __init__(self, vllm_config, role: KVConnectorRole):
        super().__init__(vllm_config, role)
        kt = vllm_config.kv_transfer_config
        self.block_size = vllm_config.cache_config.block_size
        self.path = kt.kv_connector_extra_config.get("shared_storage_path", "/tmp/vda_kv")
        os.makedirs(self.path, exist_ok=True)
        self.is_producer, self.is_consumer = kt.is_kv_producer, kt.kv_role in ("kv_consumer", "kv_both")
        self._need_load: dict[str, object] = {}
        self._storing: dict[str, list] = {}
        self.runner = None

    # ---- helpers -----------------------------------------------------------------------------
    def _aligned(self, n: int) -> int:
        return n // self.block_size * self.block_size

    def _folder(self, token_ids) -> str:
        h = hashlib.sha256(np.asarray(token_ids, dtype=np.int64).tobytes()).hexdigest()
        return os.path.join(self.path, h)

    def _found(self, token_ids) -> bool:
        n = self._aligned(len(token_ids) - 1)
        return n > 0 and os.path.isdir(self._folder(token_ids[:n]))

    # ---- worker ---------------------------------------------------------------------------------
    def register_kv_caches(self, runner) -> None:
        self.runner = runner

    def _slots(self, block_ids: list[int], n_tokens: int) -> torch.Tensor:
        b = torch.tensor(block_ids, dtype=torch.int64)
        s = (b[:, None] * self.block_size + torch.arange(self.block_size)[None, :]).reshape(-1)
        return s[:n_tokens]

    def start_load_kv(self) -> None:
        meta = self._meta
        if not isinstance(meta, SharedStorageMetadata) or self.runner is None:
            return
        from ... import ops
        for r in meta.requests:
            if r.is_store:
                continue
            n = self._aligned(len(r.token_ids) - 1)
            folder = self._folder(r.token_ids[:n])
            slots = self._slots(r.block_ids, n)
            for name, layer in self.runner.attn_layers.items():
                f = os.path.join(folder, f"{name}.safetensors")
                d = load_file(f)
                k, v = d["k"].to(layer.k_cache.device), d["v"].to(layer.k_cache.device)
                ops.reshape_and_cache(k, v, layer.k_cache, layer.v_cache, slots.to(k.device))
            logger.debug("loaded %d tokens of KV from %s", n, folder)

    def save_kv_layer(self, layer_name: str, kv_layer: tuple, attn_metadata) -> None:
        meta = self._meta
        if not isinstance(meta, SharedStorageMetadata):
            return
        from ... import ops
        k_cache, v_cache = kv_layer
        for r in meta.requests:
            if not r.is_store:
                continue
            n = self._aligned(len(r.token_ids) - 1)
            if n <= 0:
                continue
            folder = self._folder(r.token_ids[:n])
            os.makedirs(folder, exist_ok=True)
            k, v = ops.gather_cache(k_cache, v_cache, self._slots(r.block_ids, n).to(k_cache.device))
            tmp = os.path.join(folder, f".{layer_name}.{os.getpid()}.tmp")
            save_file({"k": k.cpu().contiguous(), "v": v.cpu().contiguous()}, tmp)
            os.replace(tmp, os.path.join(folder, f"{layer_name}.safetensors"))

    # ---- scheduler ------------------------------------------------------------------------------
    def get_num_new_matched_tokens(self, request, num_computed_tokens: int) -> tuple[int, bool]:
        if not self.is_consumer or not self._found(request.prompt_token_ids):
            return 0, False
        n = self._aligned(len(request.prompt_token_ids) - 1)
        return max(n - num_computed_tokens, 0), False

    def update_state_after_alloc(self, request, blocks, num_external_tokens: int) -> None:
        if num_external_tokens > 0:
            self._need_load[request.request_id] = request

    def build_connector_meta(self, so) -> SharedStorageMetadata:
        meta = SharedStorageMetadata()

        def flat(ids):
            return list(ids[0]) if ids and isinstance(ids[0], (list, tuple)) else list(ids or [])

        for nr in so.scheduled_new_reqs:
            if nr.req_id in self._need_load:
                meta.requests.append(_Req(list(nr.prompt_token_ids), flat(nr.block_ids), False))
            elif self.is_producer and not self._found(nr.prompt_token_ids):
                # [prompt, block ids so far]; stored in the step that finishes the prompt
                self._storing[nr.req_id] = [list(nr.prompt_token_ids), flat(nr.block_ids)]
        cr = so.scheduled_cached_reqs
        computed = {nr.req_id: nr.num_computed_tokens for nr in so.scheduled_new_reqs}
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
                continue
            n = so.num_scheduled_tokens.get(rid)
            if n is not None and computed.get(rid, 0) + n >= len(self._storing[rid][0]):
                toks, ids = self._storing.pop(rid)
                meta.requests.append(_Req(toks, ids, True))
        self._need_load.clear()
        return meta
