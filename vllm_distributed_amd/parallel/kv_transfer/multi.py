"""Fan-out over several connectors: loads come from the first connector that has the tokens,
saves go to all.  Reference: vllm/distributed/kv_transfer/kv_connector/v1/multi_connector.py."""
from __future__ import annotations

import copy
from dataclasses import dataclass

from .base import KVConnectorBase, KVConnectorMetadata, KVConnectorRole


@dataclass
class MultiMetadata(KVConnectorMetadata):
    metas: tuple


class MultiConnector(KVConnectorBase):
    def __init__(self, vllm_config, role: KVConnectorRole):
        super().__init__(vllm_config, role)
        from .factory import KVConnectorFactory
        self.children: list[KVConnectorBase] = []
        for spec in vllm_config.kv_transfer_config.kv_connector_extra_config.get("connectors", []):
            cfg = copy.copy(vllm_config)
            kt = copy.copy(vllm_config.kv_transfer_config)
            for k, v in spec.items():
                setattr(kt, k, v)
            kt.kv_connector_extra_config = spec.get("kv_connector_extra_config", {})
            cfg.kv_transfer_config = kt
            self.children.append(KVConnectorFactory.create(cfg, role))
        self._load_owner: dict[str, int] = {}
        self._extra_async_saves: dict[str, int] = {}

    def register_kv_caches(self, runner) -> None:
        for c in self.children:
            c.register_kv_caches(runner)

    def bind_connector_metadata(self, meta) -> None:
        assert isinstance(meta, MultiMetadata)
        for c, m in zip(self.children, meta.metas):
            c.bind_connector_metadata(m)

    def clear_connector_metadata(self) -> None:
        for c in self.children:
            c.clear_connector_metadata()

    def start_load_kv(self) -> None:
        for c in self.children:
            c.start_load_kv()

    def wait_for_layer_load(self, layer_name: str) -> None:
        for c in self.children:
            c.wait_for_layer_load(layer_name)

    def save_kv_layer(self, layer_name, kv_layer, attn_metadata) -> None:
        for c in self.children:
            c.save_kv_layer(layer_name, kv_layer, attn_metadata)

    def wait_for_save(self) -> None:
        for c in self.children:
            c.wait_for_save()

    def failed_recving(self):
        out = set()
        for c in self.children:
            out |= c.failed_recving()
        return out

    def get_finished(self, finished_req_ids):
        sending, recving = set(), set()
        for c in self.children:
            s, r = c.get_finished(finished_req_ids)
            recving |= r or set()
            for rid in s or ():
                # a request is "sent" once every connector that kept its blocks is done
                left = self._extra_async_saves.get(rid, 0)
                if left > 0:
                    self._extra_async_saves[rid] = left - 1
                else:
                    self._extra_async_saves.pop(rid, None)
                    sending.add(rid)
        return sending or None, recving or None

    def get_num_new_matched_tokens(self, request, num_computed_tokens):
        for i, c in enumerate(self.children):
            n, is_async = c.get_num_new_matched_tokens(request, num_computed_tokens)
            if n > 0:
                self._load_owner[request.request_id] = i
                return n, is_async
        return 0, False

    def update_state_after_alloc(self, request, blocks, num_external_tokens) -> None:
        owner = self._load_owner.pop(request.request_id, -1)
        for i, c in enumerate(self.children):
            c.update_state_after_alloc(request, blocks, num_external_tokens if i == owner else 0)

    def build_connector_meta(self, so) -> MultiMetadata:
        return MultiMetadata(tuple(c.build_connector_meta(so) for c in self.children))

    def request_finished(self, request, block_ids):
        n_async, params = 0, None
        for c in self.children:
            delay, p = c.request_finished(request, block_ids)
            n_async += bool(delay)
            if p is not None:
                if params is not None:
                    raise RuntimeError("only one connector may produce kv_transfer_params")
                params = p
        if n_async > 1:
            self._extra_async_saves[request.request_id] = n_async - 1
        return n_async > 0, params

    def shutdown(self) -> None:
        for c in self.children:
            c.shutdown()


class UnavailableConnector(KVConnectorBase):
    def __init__(self, vllm_config, role):
        raise RuntimeError("this connector needs the external LMCache package, which is not part of this build; "
                           "use XgmiConnector (GPU to GPU) or SharedStorageConnector")
