"""Connector interface.  One connector object lives in the scheduler (decides what to load /
what to keep for export) and one in every worker (moves the bytes); the scheduler-side object
sends the worker-side one a metadata record with each step.
Reference: vllm/distributed/kv_transfer/kv_connector/v1/base.py:1-288."""
from __future__ import annotations

import enum
from typing import Any, Optional


class KVConnectorRole(enum.Enum):
    SCHEDULER = 0
    WORKER = 1


class KVConnectorMetadata:
    """Marker base class of the per-step scheduler -> worker record."""


class KVConnectorBase:
    def __init__(self, vllm_config, role: KVConnectorRole):
        self.vllm_config = vllm_config
        self.role = role
        self._meta: Optional[KVConnectorMetadata] = None

    # ---- worker side ------------------------------------------------------------------------
    def register_kv_caches(self, runner) -> None:
        """Called once the KV arena exists (``runner.kv_chunks``, ``runner.kv_layer_loc``)."""

    def bind_connector_metadata(self, meta: Optional[KVConnectorMetadata]) -> None:
        self._meta = meta

    def clear_connector_metadata(self) -> None:
        self._meta = None

    def start_load_kv(self) -> None:
        """Begin loading external KV into the paged cache (may be asynchronous)."""

    def wait_for_layer_load(self, layer_name: str) -> None:
        """Block until this layer's external KV is in place (layer-pipelined loads)."""

    def save_kv_layer(self, layer_name: str, kv_layer: tuple, attn_metadata) -> None:
        """Export this layer's freshly written KV (may be asynchronous)."""

    def wait_for_save(self) -> None:
        """Block until all exports of the step are done."""

    def get_finished(self, finished_req_ids: set[str]) -> tuple[Optional[set[str]], Optional[set[str]]]:
        """(request ids whose export finished, request ids whose import finished) since last call."""
        return None, None

    def failed_recving(self) -> set[str]:
        """Of the imports reported by the last ``get_finished``: those that failed (nothing usable arrived;
        the scheduler has the prompt computed locally).  Cleared by the call."""
        return set()

    def shutdown(self) -> None:
        pass

    # ---- scheduler side --------------------------------------------------------------------------
    def get_num_new_matched_tokens(self, request, num_computed_tokens: int) -> tuple[int, bool]:
        """(tokens available externally beyond the locally cached ones, load asynchronously?)."""
        return 0, False

    def update_state_after_alloc(self, request, blocks, num_external_tokens: int) -> None:
        pass

    def build_connector_meta(self, scheduler_output) -> Optional[KVConnectorMetadata]:
        return None

    def request_finished(self, request, block_ids: list[int]) -> tuple[bool, Optional[dict[str, Any]]]:
        """(keep the blocks until the export completes?, kv_transfer_params for the response)."""
        return False, None
