"""Budgeted cache of multimodal encoder outputs (reference:
vllm/v1/core/encoder_cache_manager.py).  An item is identified by
(request id, index of the item inside that request) and costs as many slots as
the placeholder tokens it fills."""
from __future__ import annotations


class EncoderCacheManager:
    def __init__(self, cache_size: int):
        self.cache_size = cache_size
        self.num_free_slots = cache_size
        self.cached: dict[str, set[int]] = {}
        self.freed: list[tuple[str, int]] = []

    def has_cache(self, request, input_id: int) -> bool:
        return input_id in self.cached.get(request.request_id, ())

    def can_allocate(self, request, input_id: int) -> bool:
        return request.get_num_encoder_tokens(input_id) <= self.num_free_slots

    def allocate(self, request, input_id: int) -> None:
        self.cached.setdefault(request.request_id, set()).add(input_id)
        self.num_free_slots -= request.get_num_encoder_tokens(input_id)

    def get_cached_input_ids(self, request) -> set[int]:
        return self.cached.get(request.request_id, set())

    def free_encoder_input(self, request, input_id: int) -> None:
        rid = request.request_id
        ids = self.cached.get(rid)
        if not ids or input_id not in ids:
            return
        ids.discard(input_id)
        if not ids:
            del self.cached[rid]
        self.num_free_slots += request.get_num_encoder_tokens(input_id)
        self.freed.append((rid, input_id))

    def free(self, request) -> None:
        for i in list(self.get_cached_input_ids(request)):
            self.free_encoder_input(request, i)

    def get_freed_ids(self) -> list[tuple[str, int]]:
        out, self.freed = self.freed, []
        return out
