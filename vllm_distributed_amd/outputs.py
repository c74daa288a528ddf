"""Result objects handed back to users (reference: vllm/outputs.py)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Optional, Sequence, Union

from .sampling_params import RequestOutputKind


@dataclass
class Logprob:
    logprob: float
    rank: Optional[int] = None
    decoded_token: Optional[str] = None


# one dict (token id -> Logprob) per position
SampleLogprobs = list[dict[int, Logprob]]
PromptLogprobs = list[Optional[dict[int, Logprob]]]


@dataclass
class CompletionOutput:
    index: int
    text: str
    token_ids: Sequence[int]
    cumulative_logprob: Optional[float]
    logprobs: Optional[SampleLogprobs]
    finish_reason: Optional[str] = None
    stop_reason: Union[int, str, None] = None
    lora_request: Any = None

    def finished(self) -> bool:
        return self.finish_reason is not None


@dataclass
class RequestMetrics:
    arrival_time: float = 0.0
    first_scheduled_time: Optional[float] = None
    first_token_time: Optional[float] = None
    last_token_time: Optional[float] = None
    finished_time: Optional[float] = None
    time_in_queue: Optional[float] = None
    num_generation_tokens: int = 0


@dataclass
class RequestOutput:
    request_id: str
    prompt: Optional[str]
    prompt_token_ids: Optional[list[int]]
    prompt_logprobs: Optional[PromptLogprobs]
    outputs: list[CompletionOutput]
    finished: bool
    metrics: Optional[RequestMetrics] = None
    lora_request: Any = None
    encoder_prompt: Optional[str] = None
    encoder_prompt_token_ids: Optional[list[int]] = None
    num_cached_tokens: Optional[int] = None
    kv_transfer_params: Optional[dict[str, Any]] = None

    def add(self, nxt: "RequestOutput", aggregate: bool) -> None:
        """Merge a later output of the same request into this one (used when a
        consumer is slower than the producer of streaming outputs)."""
        self.finished |= nxt.finished
        self.kv_transfer_params = nxt.kv_transfer_params
        for nc in nxt.outputs:
            for c in self.outputs:
                if c.index == nc.index:
                    if aggregate:
                        c.text += nc.text
                        if not isinstance(c.token_ids, list):
                            c.token_ids = list(c.token_ids)
                        c.token_ids.extend(nc.token_ids)
                        if nc.logprobs:
                            c.logprobs = (c.logprobs or []) + nc.logprobs
                        c.cumulative_logprob = nc.cumulative_logprob
                    else:
                        c.text, c.token_ids = nc.text, nc.token_ids
                        c.logprobs, c.cumulative_logprob = nc.logprobs, nc.cumulative_logprob
                    c.finish_reason, c.stop_reason = nc.finish_reason, nc.stop_reason
                    break
            else:
                self.outputs.append(nc)


@dataclass
class PoolingOutput:
    data: Any  # torch.Tensor

    def __eq__(self, other):
        return isinstance(other, PoolingOutput) and bool((self.data == other.data).all())


@dataclass
class PoolingRequestOutput:
    request_id: str
    outputs: PoolingOutput
    prompt_token_ids: list[int]
    finished: bool


@dataclass
class EmbeddingOutput:
    embedding: list[float]

    @property
    def hidden_size(self) -> int:
        return len(self.embedding)


@dataclass
class EmbeddingRequestOutput:
    request_id: str
    outputs: EmbeddingOutput
    prompt_token_ids: list[int]
    finished: bool

    @staticmethod
    def from_base(o: PoolingRequestOutput) -> "EmbeddingRequestOutput":
        return EmbeddingRequestOutput(o.request_id, EmbeddingOutput(o.outputs.data.flatten().tolist()),
                                      o.prompt_token_ids, o.finished)


@dataclass
class ClassificationOutput:
    probs: list[float]

    @property
    def num_classes(self) -> int:
        return len(self.probs)


@dataclass
class ClassificationRequestOutput:
    request_id: str
    outputs: ClassificationOutput
    prompt_token_ids: list[int]
    finished: bool

    @staticmethod
    def from_base(o: PoolingRequestOutput) -> "ClassificationRequestOutput":
        return ClassificationRequestOutput(o.request_id, ClassificationOutput(o.outputs.data.flatten().tolist()),
                                           o.prompt_token_ids, o.finished)


@dataclass
class ScoringOutput:
    score: float


@dataclass
class ScoringRequestOutput:
    request_id: str
    outputs: ScoringOutput
    prompt_token_ids: list[int]
    finished: bool


@dataclass
class BeamSearchSequence:
    tokens: list[int]
    logprobs: list[dict[int, Logprob]]
    cum_logprob: float = 0.0
    text: Optional[str] = None
    finish_reason: Optional[str] = None
    stop_reason: Union[int, str, None] = None


@dataclass
class BeamSearchOutput:
    sequences: list[BeamSearchSequence] = field(default_factory=list)


__all__ = [n for n in dir() if n[0].isupper()] + ["RequestOutputKind"]
