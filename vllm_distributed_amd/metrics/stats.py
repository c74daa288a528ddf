"""Statistics records that flow from the engine core to the stat loggers
(reference: vllm/v1/metrics/stats.py, vllm/v1/spec_decode/metrics.py)."""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Optional

from ..request import EngineCoreEvent, EngineCoreEventType, FinishReason


@dataclass
class PrefixCacheStats:
    reset: bool = False
    requests: int = 0
    queries: int = 0        # tokens looked up
    hits: int = 0           # tokens found
    preempted_requests: int = 0
    preempted_queries: int = 0
    preempted_hits: int = 0


@dataclass
class SpecDecodingStats:
    num_spec_tokens: int = 0
    num_drafts: int = 0
    num_draft_tokens: int = 0
    num_accepted_tokens: int = 0
    num_accepted_tokens_per_pos: list[int] = field(default_factory=list)

    @classmethod
    def new(cls, k: int) -> "SpecDecodingStats":
        return cls(num_spec_tokens=k, num_accepted_tokens_per_pos=[0] * k)

    def observe_draft(self, num_draft: int, num_accepted: int) -> None:
        self.num_drafts += 1
        self.num_draft_tokens += num_draft
        self.num_accepted_tokens += num_accepted
        for i in range(min(num_accepted, len(self.num_accepted_tokens_per_pos))):
            self.num_accepted_tokens_per_pos[i] += 1


@dataclass
class SchedulerStats:
    num_running_reqs: int = 0
    num_waiting_reqs: int = 0
    kv_cache_usage: float = 0.0
    prefix_cache_stats: PrefixCacheStats = field(default_factory=PrefixCacheStats)
    spec_decoding_stats: Optional[SpecDecodingStats] = None
    num_corrupted_reqs: int = 0
    step_counter: int = 0
    current_wave: int = 0


@dataclass
class RequestStateStats:
    """Per-request timing kept by the front end."""
    num_generation_tokens: int = 0
    arrival_time: float = 0.0          # wall clock, front end
    queued_ts: float = 0.0             # monotonic, engine core
    scheduled_ts: float = 0.0
    first_token_ts: float = 0.0
    last_token_ts: float = 0.0


@dataclass
class FinishedRequestStats:
    finish_reason: FinishReason
    e2e_latency: float = 0.0
    num_prompt_tokens: int = 0
    num_generation_tokens: int = 0
    max_tokens_param: Optional[int] = None
    queued_time: float = 0.0
    prefill_time: float = 0.0
    inference_time: float = 0.0
    decode_time: float = 0.0


class IterationStats:
    """Everything the front end learned while processing one batch of engine outputs."""

    def __init__(self):
        self.iteration_timestamp = time.time()
        self.num_generation_tokens = 0
        self.num_prompt_tokens = 0
        self.num_preempted_reqs = 0
        self.finished_requests: list[FinishedRequestStats] = []
        self.max_num_generation_tokens_iter: list[int] = []
        self.n_params_iter: list[int] = []
        self.time_to_first_tokens_iter: list[float] = []
        self.time_per_output_tokens_iter: list[float] = []
        self.waiting_lora_adapters: dict[str, int] = {}
        self.running_lora_adapters: dict[str, int] = {}

    def _since_arrival(self, start: float) -> float:
        return self.iteration_timestamp - start

    def update_from_output(self, output, engine_core_timestamp: float, is_prefilling: bool,
                           prompt_len: int, req_stats: RequestStateStats, lora_stats=None) -> None:
        n_new = len(output.new_token_ids)
        self.num_generation_tokens += n_new
        if is_prefilling:
            # generation requests emit their first token here; pooling requests finish without tokens
            self.num_prompt_tokens += prompt_len
            self.time_to_first_tokens_iter.append(self._since_arrival(req_stats.arrival_time))
        req_stats.num_generation_tokens += n_new
        if output.events:
            self.update_from_events(output.request_id, output.events, is_prefilling, req_stats, lora_stats)
        if is_prefilling:
            req_stats.first_token_ts = engine_core_timestamp
        else:
            self.time_per_output_tokens_iter.append(engine_core_timestamp - req_stats.last_token_ts)
        req_stats.last_token_ts = engine_core_timestamp

    def update_from_events(self, req_id: str, events: list[EngineCoreEvent], is_prefilling: bool,
                           req_stats: RequestStateStats, lora_stats=None) -> None:
        for ev in events:
            if ev.type == EngineCoreEventType.QUEUED:
                req_stats.queued_ts = ev.timestamp
                if lora_stats is not None:
                    lora_stats.waiting_requests.add(req_id)
            elif ev.type == EngineCoreEventType.SCHEDULED:
                if req_stats.scheduled_ts == 0.0:
                    req_stats.scheduled_ts = ev.timestamp
                if lora_stats is not None:
                    lora_stats.waiting_requests.discard(req_id)
                    lora_stats.running_requests.add(req_id)
            elif ev.type == EngineCoreEventType.PREEMPTED:
                self.num_preempted_reqs += 1
                if lora_stats is not None:
                    lora_stats.running_requests.discard(req_id)
                    lora_stats.waiting_requests.add(req_id)

    def update_from_finished_request(self, finish_reason: FinishReason, num_prompt_tokens: int,
                                     max_tokens_param: Optional[int], req_stats: RequestStateStats) -> None:
        e2e = self._since_arrival(req_stats.arrival_time)
        # intervals on the engine core's monotonic clock
        queued = req_stats.scheduled_ts - req_stats.queued_ts
        prefill = req_stats.first_token_ts - req_stats.scheduled_ts
        decode = req_stats.last_token_ts - req_stats.first_token_ts
        inference = req_stats.last_token_ts - req_stats.scheduled_ts
        self.finished_requests.append(FinishedRequestStats(
            finish_reason, e2e, num_prompt_tokens, req_stats.num_generation_tokens, max_tokens_param,
            queued, prefill, inference, decode))


@dataclass
class LoRARequestStates:
    waiting_requests: set = field(default_factory=set)
    running_requests: set = field(default_factory=set)
