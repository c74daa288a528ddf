"""Request-level types shared by the front end, the engine core and the scheduler.

``EngineCoreRequest`` / ``EngineCoreOutput(s)`` are the wire structs between
the API process and the engine-core process (reference: vllm/v1/engine/__init__.py:44-179);
``Request`` is the scheduler's mutable view of one request
(reference: vllm/v1/request.py).
"""
from __future__ import annotations

import enum
import time
from dataclasses import dataclass, field
from typing import Any, Optional, Union

from .sampling_params import PoolingParams, SamplingParams


class FinishReason(enum.IntEnum):
    STOP = 0
    LENGTH = 1
    ABORT = 2

    def __str__(self) -> str:
        return ("stop", "length", "abort")[self.value]


class RequestStatus(enum.IntEnum):
    WAITING = 0
    WAITING_FOR_FSM = 1
    WAITING_FOR_REMOTE_KVS = 2
    RUNNING = 3
    PREEMPTED = 4
    # anything above PREEMPTED is terminal
    FINISHED_STOPPED = 5
    FINISHED_LENGTH_CAPPED = 6
    FINISHED_ABORTED = 7
    FINISHED_IGNORED = 8

    @staticmethod
    def is_finished(s: "RequestStatus") -> bool:
        return s > RequestStatus.PREEMPTED

    @staticmethod
    def finish_reason(s: "RequestStatus") -> Optional[FinishReason]:
        return _FINISH_MAP.get(s)


_FINISH_MAP = {
    RequestStatus.FINISHED_STOPPED: FinishReason.STOP,
    RequestStatus.FINISHED_LENGTH_CAPPED: FinishReason.LENGTH,
    RequestStatus.FINISHED_ABORTED: FinishReason.ABORT,
    RequestStatus.FINISHED_IGNORED: FinishReason.LENGTH,
}


class EngineCoreEventType(enum.IntEnum):
    QUEUED = 1
    SCHEDULED = 2
    PREEMPTED = 3


@dataclass
class EngineCoreEvent:
    type: EngineCoreEventType
    timestamp: float

    @classmethod
    def now(cls, t: EngineCoreEventType) -> "EngineCoreEvent":
        return cls(t, time.monotonic())


@dataclass
class LoRARequest:
    lora_name: str
    lora_int_id: int
    lora_path: str = ""
    base_model_name: Optional[str] = None

    def __post_init__(self):
        if self.lora_int_id < 1:
            raise ValueError("lora_int_id must be >= 1")

    def __hash__(self):
        return hash(self.lora_int_id)

    def __eq__(self, other):
        return isinstance(other, LoRARequest) and other.lora_int_id == self.lora_int_id


@dataclass
class PlaceholderRange:
    offset: int
    length: int


@dataclass
class EngineCoreRequest:
    request_id: str
    prompt_token_ids: list[int]
    sampling_params: Optional[SamplingParams]
    eos_token_id: Optional[int] = None
    arrival_time: float = 0.0
    lora_request: Optional[LoRARequest] = None
    cache_salt: Optional[str] = None
    priority: int = 0
    pooling_params: Optional[PoolingParams] = None
    # multimodal: opaque per-item tensors + hashes + where they land in the prompt
    mm_inputs: Optional[list[Any]] = None
    mm_hashes: Optional[list[str]] = None
    mm_placeholders: Optional[list[PlaceholderRange]] = None
    data_parallel_rank: Optional[int] = None
    kv_transfer_params: Optional[dict[str, Any]] = None
    trace_headers: Optional[dict[str, str]] = None
    # routing info so a DP front end can send the abort to the right engine
    current_wave: int = 0
    client_index: int = 0


@dataclass
class LogprobsLists:
    """top-k+1 ids, their logprobs and the rank of the sampled token, per position."""
    logprob_token_ids: list[list[int]]
    logprobs: list[list[float]]
    sampled_token_ranks: list[int]

    def slice(self, a: int, b: int) -> "LogprobsLists":
        return LogprobsLists(self.logprob_token_ids[a:b], self.logprobs[a:b], self.sampled_token_ranks[a:b])


@dataclass
class EngineCoreOutput:
    request_id: str
    new_token_ids: list[int]
    new_logprobs: Optional[LogprobsLists] = None
    new_prompt_logprobs: Optional[LogprobsLists] = None
    pooling_output: Any = None
    finish_reason: Optional[FinishReason] = None
    stop_reason: Union[int, str, None] = None
    events: Optional[list[EngineCoreEvent]] = None
    kv_transfer_params: Optional[dict[str, Any]] = None
    num_cached_tokens: int = 0

    @property
    def finished(self) -> bool:
        return self.finish_reason is not None


@dataclass
class UtilityOutput:
    call_id: int
    failure_message: Optional[str] = None
    result: Any = None


@dataclass
class EngineCoreOutputs:
    engine_index: int = 0
    outputs: list[EngineCoreOutput] = field(default_factory=list)
    scheduler_stats: Any = None
    timestamp: float = 0.0
    utility_output: Optional[UtilityOutput] = None
    finished_requests: Optional[set[str]] = None
    # DP wave protocol
    wave_complete: Optional[int] = None
    start_wave: Optional[int] = None

    def __post_init__(self):
        if self.timestamp == 0.0:
            self.timestamp = time.monotonic()


class EngineCoreRequestType(enum.Enum):
    ADD = b"\x00"
    ABORT = b"\x01"
    START_DP_WAVE = b"\x02"
    UTILITY = b"\x03"
    EXECUTOR_FAILED = b"\x04"
    CLIENT_GONE = b"\x05"     # internal: a front-end connection closed


class Request:
    """Scheduler-side state of one request."""

    __slots__ = (
        "request_id", "client_index", "priority", "sampling_params", "pooling_params",
        "eos_token_id", "lora_request", "structured_output_request", "arrival_time",
        "status", "events", "stop_reason", "kv_transfer_params", "max_tokens",
        "prompt_token_ids", "num_prompt_tokens", "_output_token_ids", "_all_token_ids",
        "spec_token_ids", "num_computed_tokens", "cache_salt", "mm_inputs", "mm_hashes",
        "mm_positions", "num_cached_tokens", "num_preemptions", "block_hashes",
        "trace_headers", "num_output_placeholders", "use_structured_output", "plain_sampling",
    )

    def __init__(self, request_id: str, prompt_token_ids: list[int],
                 sampling_params: Optional[SamplingParams], eos_token_id: Optional[int] = None,
                 client_index: int = 0, arrival_time: Optional[float] = None,
                 lora_request: Optional[LoRARequest] = None, cache_salt: Optional[str] = None,
                 priority: int = 0, pooling_params: Optional[PoolingParams] = None,
                 mm_inputs=None, mm_hashes=None, mm_positions=None,
                 structured_output_request=None, kv_transfer_params=None, trace_headers=None):
        self.request_id = request_id
        self.client_index = client_index
        self.priority = priority
        self.sampling_params = sampling_params
        self.pooling_params = pooling_params
        self.eos_token_id = eos_token_id
        self.lora_request = lora_request
        self.structured_output_request = structured_output_request
        self.use_structured_output = structured_output_request is not None
        self.arrival_time = arrival_time if arrival_time is not None else time.time()
        self.status = (RequestStatus.WAITING_FOR_FSM if self.use_structured_output
                       and not getattr(structured_output_request, "ready", True)
                       else RequestStatus.WAITING)
        self.events: list[EngineCoreEvent] = []
        self.stop_reason: Union[int, str, None] = None
        self.kv_transfer_params = kv_transfer_params
        if pooling_params is not None:
            self.max_tokens = 1
        else:
            assert sampling_params is not None and sampling_params.max_tokens is not None
            self.max_tokens = sampling_params.max_tokens
        self.prompt_token_ids = prompt_token_ids
        self.num_prompt_tokens = len(prompt_token_ids)
        self._output_token_ids: list[int] = []
        self._all_token_ids: list[int] = list(prompt_token_ids)
        self.spec_token_ids: list[int] = []
        self.num_computed_tokens = 0
        self.cache_salt = cache_salt
        self.mm_inputs = mm_inputs or []
        self.mm_hashes = mm_hashes or []
        self.mm_positions = mm_positions or []
        self.num_cached_tokens = -1   # set on first schedule (prefix-cache hit length)
        self.num_preemptions = 0
        self.block_hashes: list = []  # filled lazily by the KV manager
        self.trace_headers = trace_headers
        self.num_output_placeholders = 0
        # nothing but the sampled token itself carries over from one step of this request to the next
        # (run-ahead scheduling may then start step N + 1 before the results of step N are on the host)
        sp = sampling_params
        self.plain_sampling = bool(
            sp is not None and not self.use_structured_output and not sp.presence_penalty
            and not sp.frequency_penalty and sp.repetition_penalty == 1.0 and not sp.bad_words
            and not sp.min_tokens and sp.prompt_logprobs is None and not getattr(sp, "logits_processors", None)
            and not kv_transfer_params and not self.mm_inputs)

    @classmethod
    def from_engine_core_request(cls, r: EngineCoreRequest, structured_output_request=None) -> "Request":
        return cls(r.request_id, r.prompt_token_ids, r.sampling_params, r.eos_token_id,
                   client_index=r.client_index, arrival_time=r.arrival_time, lora_request=r.lora_request,
                   cache_salt=r.cache_salt, priority=r.priority, pooling_params=r.pooling_params,
                   mm_inputs=r.mm_inputs, mm_hashes=r.mm_hashes, mm_positions=r.mm_placeholders,
                   structured_output_request=structured_output_request,
                   kv_transfer_params=r.kv_transfer_params, trace_headers=r.trace_headers)

    # token bookkeeping --------------------------------------------------
    def append_output_token_ids(self, ids: Union[int, list[int]]) -> None:
        if isinstance(ids, int):
            self._output_token_ids.append(ids)
            self._all_token_ids.append(ids)
        else:
            self._output_token_ids.extend(ids)
            self._all_token_ids.extend(ids)

    @property
    def output_token_ids(self) -> list[int]:
        return self._output_token_ids

    @property
    def all_token_ids(self) -> list[int]:
        return self._all_token_ids

    @property
    def num_tokens(self) -> int:
        return len(self._all_token_ids)

    @property
    def num_tokens_with_spec(self) -> int:
        return len(self._all_token_ids) + len(self.spec_token_ids)

    @property
    def num_output_tokens(self) -> int:
        return len(self._output_token_ids)

    @property
    def has_encoder_inputs(self) -> bool:
        return bool(self.mm_inputs)

    def get_num_encoder_tokens(self, i: int) -> int:
        return self.mm_positions[i].length

    def is_finished(self) -> bool:
        return RequestStatus.is_finished(self.status)

    def get_finished_reason(self) -> Optional[FinishReason]:
        return RequestStatus.finish_reason(self.status)

    def record_event(self, t: EngineCoreEventType, ts: Optional[float] = None) -> None:
        self.events.append(EngineCoreEvent(t, time.monotonic() if ts is None else ts))

    def take_events(self) -> Optional[list[EngineCoreEvent]]:
        if not self.events:
            return None
        ev, self.events = self.events, []
        return ev

    def __lt__(self, other: "Request") -> bool:
        # priority queue order: smaller priority value first, then earlier arrival
        return (self.priority, self.arrival_time, self.request_id) < \
               (other.priority, other.arrival_time, other.request_id)

    def __repr__(self) -> str:
        return (f"Request({self.request_id}, status={self.status.name}, prompt={self.num_prompt_tokens}, "
                f"out={self.num_output_tokens}, computed={self.num_computed_tokens})")
