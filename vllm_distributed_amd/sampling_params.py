"""User-facing generation parameters.

Same fields and validation rules a user of the reference expects
(reference: vllm/sampling_params.py; pooling: vllm/pooling_params.py), stored
as a plain dataclass so they travel through the msgpack wire codec unchanged.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field, fields
from enum import IntEnum
from typing import Any, Optional, Union

_EPS = 1e-5
_MAX_TEMP_FLOOR = 1e-2


class SamplingType(IntEnum):
    GREEDY = 0
    RANDOM = 1
    RANDOM_SEED = 2


class RequestOutputKind(IntEnum):
    CUMULATIVE = 0   # every output holds everything generated so far
    DELTA = 1        # every output holds only what is new
    FINAL_ONLY = 2   # a single output when the request finishes


@dataclass
class GuidedDecodingParams:
    json: Optional[Union[str, dict]] = None
    regex: Optional[str] = None
    choice: Optional[list[str]] = None
    grammar: Optional[str] = None
    json_object: Optional[bool] = None
    backend: Optional[str] = None
    whitespace_pattern: Optional[str] = None
    structural_tag: Optional[Union[str, dict]] = None

    def __post_init__(self):
        n = sum(x is not None and x is not False for x in
                (self.json, self.regex, self.choice, self.grammar, self.json_object, self.structural_tag))
        if n > 1:
            raise ValueError("only one of json/regex/choice/grammar/json_object/structural_tag may be set")

    @property
    def kind(self) -> Optional[str]:
        for k in ("json", "regex", "choice", "grammar", "structural_tag"):
            if getattr(self, k) is not None:
                return k
        return "json_object" if self.json_object else None


@dataclass
class SamplingParams:
    n: int = 1
    best_of: Optional[int] = None
    presence_penalty: float = 0.0
    frequency_penalty: float = 0.0
    repetition_penalty: float = 1.0
    temperature: float = 1.0
    top_p: float = 1.0
    top_k: int = 0
    min_p: float = 0.0
    seed: Optional[int] = None
    stop: Optional[Union[str, list[str]]] = None
    stop_token_ids: Optional[list[int]] = None
    bad_words: Optional[list[str]] = None
    ignore_eos: bool = False
    max_tokens: Optional[int] = 16
    min_tokens: int = 0
    logprobs: Optional[int] = None
    prompt_logprobs: Optional[int] = None
    detokenize: bool = True
    skip_special_tokens: bool = True
    spaces_between_special_tokens: bool = True
    include_stop_str_in_output: bool = False
    output_kind: RequestOutputKind = RequestOutputKind.CUMULATIVE
    truncate_prompt_tokens: Optional[int] = None
    logit_bias: Optional[dict[int, float]] = None
    allowed_token_ids: Optional[list[int]] = None
    guided_decoding: Optional[GuidedDecodingParams] = None
    extra_args: Optional[dict[str, Any]] = None
    # filled in by the input processor (tokenised bad words, eos handling)
    _bad_words_token_ids: Optional[list[list[int]]] = field(default=None, repr=False)
    _all_stop_token_ids: set = field(default_factory=set, repr=False)
    output_text_buffer_length: int = field(default=0, repr=False)

    def __post_init__(self):
        if self.top_k is None or self.top_k == -1:
            self.top_k = 0
        if isinstance(self.stop, str):
            self.stop = [self.stop]
        elif self.stop is None:
            self.stop = []
        else:
            self.stop = list(self.stop)
        self.stop_token_ids = list(self.stop_token_ids or [])
        self.bad_words = list(self.bad_words or [])
        if self.logprobs is True:
            self.logprobs = 1
        if self.prompt_logprobs is True:
            self.prompt_logprobs = 1
        if isinstance(self.output_kind, int) and not isinstance(self.output_kind, RequestOutputKind):
            self.output_kind = RequestOutputKind(self.output_kind)
        if isinstance(self.guided_decoding, dict):
            self.guided_decoding = GuidedDecodingParams(**self.guided_decoding)
        if self.logit_bias:
            self.logit_bias = {int(k): float(v) for k, v in self.logit_bias.items()}
        if 0 < self.temperature < _MAX_TEMP_FLOOR:
            self.temperature = _MAX_TEMP_FLOOR
        if self.stop and not self.include_stop_str_in_output:
            # hold back enough characters that a stop string is never streamed
            self.output_text_buffer_length = max(len(s) for s in self.stop) - 1
        self._verify()
        if self.temperature < _EPS:
            # greedy: the filters are irrelevant, neutralise them
            self.top_p, self.top_k, self.min_p = 1.0, 0, 0.0
        self._all_stop_token_ids = set(self.stop_token_ids)

    def _verify(self) -> None:
        if not isinstance(self.n, int) or self.n < 1:
            raise ValueError(f"n must be a positive int, got {self.n}")
        if self.best_of is not None and self.best_of < self.n:
            raise ValueError("best_of must be >= n")
        for name in ("presence_penalty", "frequency_penalty"):
            v = getattr(self, name)
            if not -2.0 <= v <= 2.0:
                raise ValueError(f"{name} must be in [-2, 2], got {v}")
        if not 0.0 < self.repetition_penalty <= 2.0:
            raise ValueError("repetition_penalty must be in (0, 2]")
        if self.temperature < 0.0:
            raise ValueError("temperature must be non-negative")
        if not 0.0 < self.top_p <= 1.0:
            raise ValueError("top_p must be in (0, 1]")
        if self.top_k < 0:
            raise ValueError("top_k must be 0/-1 (disabled) or >= 1")
        if not 0.0 <= self.min_p <= 1.0:
            raise ValueError("min_p must be in [0, 1]")
        if self.max_tokens is not None and self.max_tokens < 1:
            raise ValueError("max_tokens must be at least 1")
        if self.min_tokens < 0:
            raise ValueError("min_tokens must be >= 0")
        if self.max_tokens is not None and self.min_tokens > self.max_tokens:
            raise ValueError("min_tokens must be <= max_tokens")
        if self.logprobs is not None and self.logprobs < 0:
            raise ValueError("logprobs must be non-negative")
        if self.prompt_logprobs is not None and self.prompt_logprobs < 0:
            raise ValueError("prompt_logprobs must be non-negative")
        if self.truncate_prompt_tokens is not None and self.truncate_prompt_tokens < 1:
            raise ValueError("truncate_prompt_tokens must be >= 1")
        if any(not s for s in self.stop):
            raise ValueError("stop cannot contain an empty string")
        if self.stop and not self.detokenize:
            raise ValueError("stop strings need detokenize=True")
        if self.best_of is not None and self.best_of != self.n:
            raise ValueError("best_of != n is not supported by this engine")

    # ------------------------------------------------------------------
    @property
    def sampling_type(self) -> SamplingType:
        if self.temperature < _EPS:
            return SamplingType.GREEDY
        return SamplingType.RANDOM_SEED if self.seed is not None else SamplingType.RANDOM

    @property
    def all_stop_token_ids(self) -> set:
        return self._all_stop_token_ids

    @property
    def bad_words_token_ids(self):
        return self._bad_words_token_ids

    def update_from_generation_config(self, gen_cfg: dict, eos_token_id: Optional[int] = None) -> None:
        """Fold model-level eos ids into the stop set (reference: sampling_params.py
        ``update_from_generation_config``)."""
        if eos_token_id is not None and not self.ignore_eos:
            self._all_stop_token_ids.add(eos_token_id)
        extra = gen_cfg.get("eos_token_id") if gen_cfg else None
        if extra is not None:
            ids = {extra} if isinstance(extra, int) else set(extra)
            ids.discard(eos_token_id)
            if ids and not self.ignore_eos:
                self._all_stop_token_ids.update(ids)
                self.stop_token_ids = sorted(set(self.stop_token_ids) | ids)

    def update_from_tokenizer(self, tokenizer) -> None:
        if not self.bad_words:
            return
        out: list[list[int]] = []
        for word in self.bad_words:
            # a word can be tokenised differently at the start of the text and
            # after a space: ban both spellings
            plain = list(tokenizer.encode(word.lstrip(), add_special_tokens=False))
            spaced = list(tokenizer.encode(" " + word.lstrip(), add_special_tokens=False))
            if plain:
                out.append(plain)
            if spaced and spaced != plain and len(spaced) == len(plain) and spaced[0] != plain[0]:
                out.append(spaced)
        vocab = len(tokenizer) if hasattr(tokenizer, "__len__") else None
        if vocab:
            for ids in out:
                if any(t < 0 or t >= vocab for t in ids):
                    raise ValueError("bad_words produced a token outside the vocabulary")
        self._bad_words_token_ids = out

    def clone(self) -> "SamplingParams":
        return copy.deepcopy(self)

    # wire helpers -------------------------------------------------------
    def to_wire(self) -> dict:
        d = {}
        for f in fields(self):
            v = getattr(self, f.name)
            if f.name == "_all_stop_token_ids":
                v = sorted(v)
            elif f.name == "output_kind":
                v = int(v)
            elif f.name == "guided_decoding" and v is not None:
                v = {k: getattr(v, k) for k in v.__dataclass_fields__}
            d[f.name] = v
        return d

    @classmethod
    def from_wire(cls, d: dict) -> "SamplingParams":
        d = dict(d)
        stops = set(d.pop("_all_stop_token_ids", ()))
        bw = d.pop("_bad_words_token_ids", None)
        d.pop("output_text_buffer_length", None)
        if d.get("logit_bias"):
            d["logit_bias"] = {int(k): v for k, v in d["logit_bias"].items()}
        sp = cls(**d)
        sp._all_stop_token_ids = stops | set(sp.stop_token_ids)
        sp._bad_words_token_ids = bw
        return sp


@dataclass
class BeamSearchParams:
    beam_width: int
    max_tokens: int
    ignore_eos: bool = False
    temperature: float = 0.0
    length_penalty: float = 1.0
    include_stop_str_in_output: bool = False


@dataclass
class PoolingParams:
    """Parameters for embedding / classification / scoring requests."""
    dimensions: Optional[int] = None
    normalize: Optional[bool] = None
    softmax: Optional[bool] = None
    task: Optional[str] = None
    additional_data: Optional[Any] = None
    output_kind: RequestOutputKind = RequestOutputKind.FINAL_ONLY
    #: Segment id of every prompt token (BERT cross-encoders: 0 for the first text, 1 for the second).
    token_type_ids: Optional[list[int]] = None

    def clone(self) -> "PoolingParams":
        return copy.deepcopy(self)

    def to_wire(self) -> dict:
        return {"dimensions": self.dimensions, "normalize": self.normalize,
                "softmax": self.softmax, "task": self.task, "token_type_ids": self.token_type_ids}

    @classmethod
    def from_wire(cls, d: dict) -> "PoolingParams":
        return cls(**d)
