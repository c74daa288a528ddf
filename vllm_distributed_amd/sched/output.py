"""The scheduler -> worker contract for one engine step, and the worker's reply
(reference: vllm/v1/core/sched/output.py, vllm/v1/outputs.py).

Token-parallel extension: ``tknp`` carries the request -> rank assignment so
every rank can slice the batch to its own requests
(reference: sched/output.py:83-89,167-169).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Optional

from ..request import LogprobsLists, LoRARequest, PlaceholderRange
from ..sampling_params import PoolingParams, SamplingParams


@dataclass
class NewRequestData:
    req_id: str
    prompt_token_ids: list[int]
    sampling_params: Optional[SamplingParams]
    block_ids: tuple[list[int], ...]
    num_computed_tokens: int
    lora_request: Optional[LoRARequest] = None
    pooling_params: Optional[PoolingParams] = None
    mm_inputs: list = field(default_factory=list)
    mm_hashes: list[str] = field(default_factory=list)
    mm_positions: list[PlaceholderRange] = field(default_factory=list)

    @classmethod
    def from_request(cls, request, block_ids) -> "NewRequestData":
        return cls(request.request_id, request.prompt_token_ids, request.sampling_params, block_ids,
                   request.num_computed_tokens, request.lora_request, request.pooling_params,
                   request.mm_inputs, request.mm_hashes, request.mm_positions)


@dataclass
class CachedRequestData:
    """Requests the worker already knows, in parallel lists."""
    req_ids: list[str] = field(default_factory=list)
    # True: the request was preempted and is resumed -> replace the block table
    resumed_from_preemption: list[bool] = field(default_factory=list)
    # tokens the worker has not seen yet (pipeline parallel: sampled on the last stage)
    new_token_ids: list[list[int]] = field(default_factory=list)
    new_block_ids: list[tuple[list[int], ...]] = field(default_factory=list)
    num_computed_tokens: list[int] = field(default_factory=list)

    @property
    def num_reqs(self) -> int:
        return len(self.req_ids)


@dataclass
class TokenParallelAllocation:
    req_to_rank: dict[str, int]
    tokens_per_rank: list[int]
    reqs_per_rank: list[int]


@dataclass
class SchedulerOutput:
    scheduled_new_reqs: list[NewRequestData] = field(default_factory=list)
    scheduled_cached_reqs: CachedRequestData = field(default_factory=CachedRequestData)
    num_scheduled_tokens: dict[str, int] = field(default_factory=dict)
    total_num_scheduled_tokens: int = 0
    scheduled_spec_decode_tokens: dict[str, list[int]] = field(default_factory=dict)
    scheduled_encoder_inputs: dict[str, list[int]] = field(default_factory=dict)
    num_common_prefix_blocks: list[int] = field(default_factory=list)
    finished_req_ids: set[str] = field(default_factory=set)
    free_encoder_input_ids: list[tuple[str, int]] = field(default_factory=list)
    structured_output_request_ids: dict[str, int] = field(default_factory=dict)
    grammar_bitmask: Any = None
    kv_connector_metadata: Any = None
    tknp: Optional[TokenParallelAllocation] = None
    # run-ahead scheduling: the runner may hand back a handle and leave the sampled ids on the device until asked
    defer_output: bool = False


@dataclass
class ModelRunnerOutput:
    req_ids: list[str] = field(default_factory=list)
    req_id_to_index: dict[str, int] = field(default_factory=dict)
    # generated token ids per request (several with speculative decoding, none mid-prefill)
    sampled_token_ids: list[list[int]] = field(default_factory=list)
    spec_token_ids: Optional[list[list[int]]] = None
    logprobs: Optional[LogprobsLists] = None
    prompt_logprobs_dict: dict[str, Optional[LogprobsLists]] = field(default_factory=dict)
    pooler_output: list = field(default_factory=list)
    finished_sending: Optional[set[str]] = None
    finished_recving: Optional[set[str]] = None
    # subset of finished_recving whose transfer failed: nothing usable arrived, the prompt is computed locally
    failed_recving: Optional[set[str]] = None
    num_nans_in_logits: Optional[dict[str, int]] = None


EMPTY_MODEL_RUNNER_OUTPUT = ModelRunnerOutput()
