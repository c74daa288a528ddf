"""Connector registry (reference: vllm/distributed/kv_transfer/kv_connector/factory.py)."""
from __future__ import annotations

import importlib
from typing import Callable

from .base import KVConnectorBase, KVConnectorRole  # noqa: F401


class KVConnectorFactory:
    _registry: dict[str, Callable[[], type]] = {}

    @classmethod
    def register_connector(cls, name: str, module_path: str, class_name: str) -> None:
        if name in cls._registry:
            raise ValueError(f"connector {name!r} is already registered")
        cls._registry[name] = lambda: getattr(importlib.import_module(module_path), class_name)

    @classmethod
    def create(cls, vllm_config, role: KVConnectorRole) -> KVConnectorBase:
        kt = vllm_config.kv_transfer_config
        name = kt.kv_connector
        if name in cls._registry:
            klass = cls._registry[name]()
        elif kt.kv_connector_module_path:
            klass = getattr(importlib.import_module(kt.kv_connector_module_path), name)
        else:
            raise ValueError(f"unknown KV connector {name!r}; known: {sorted(cls._registry)}")
        return klass(vllm_config, role)


_PKG = __package__
KVConnectorFactory.register_connector("SharedStorageConnector", f"{_PKG}.shared_storage", "SharedStorageConnector")
KVConnectorFactory.register_connector("XgmiConnector", f"{_PKG}.xgmi", "XgmiConnector")
# names used by the reference's GPU-to-GPU connectors map onto the xGMI peer-memory connector
KVConnectorFactory.register_connector("NixlConnector", f"{_PKG}.xgmi", "XgmiConnector")
KVConnectorFactory.register_connector("XgmiPushConnector", f"{_PKG}.push", "XgmiPushConnector")
KVConnectorFactory.register_connector("P2pNcclConnector", f"{_PKG}.push", "p2p_connector")
KVConnectorFactory.register_connector("MultiConnector", f"{_PKG}.multi", "MultiConnector")
# KV offloading to host memory, in-tree; the names of the reference's adapters around external stores (LMCache,
# Mooncake) resolve to it as well: those packages are not part of this build
KVConnectorFactory.register_connector("HostOffloadConnector", f"{_PKG}.host_offload", "HostOffloadConnector")
KVConnectorFactory.register_connector("LMCacheConnectorV1", f"{_PKG}.host_offload", "HostOffloadConnector")
KVConnectorFactory.register_connector("MooncakeStoreConnector", f"{_PKG}.host_offload", "HostOffloadConnector")
