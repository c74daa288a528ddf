"""KV-cache transfer between engine instances (disaggregated prefill / decode, external KV stores)."""
from .base import KVConnectorBase, KVConnectorMetadata, KVConnectorRole  # noqa: F401
from .factory import KVConnectorFactory  # noqa: F401
