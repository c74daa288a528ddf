"""Output side of the front end: incremental detokenisation with stop-string
handling, logprob assembly, n>1 fan-in, and RequestOutput construction in
cumulative / delta / final-only mode.

Reference: vllm/v1/engine/output_processor.py, detokenizer.py, logprobs.py,
parallel_sampling.py.
"""
from __future__ import annotations

import asyncio
from dataclasses import dataclass, field
from typing import Any, Optional, Union

from ..logger import init_logger
from ..metrics.stats import IterationStats, LoRARequestStates, RequestStateStats
from ..outputs import (CompletionOutput, Logprob, PoolingOutput, PoolingRequestOutput, RequestOutput)
from ..request import EngineCoreOutput, EngineCoreRequest, FinishReason
from ..sampling_params import RequestOutputKind, SamplingParams

logger = init_logger(__name__)


# ---------------------------------------------------------------------------
class IncrementalDetokenizer:
    """Token ids -> text, one step at a time, holding back text that could still
    turn out to be (the start of) a stop string."""

    def __init__(self, tokenizer, params: Optional[SamplingParams], prompt_ids: list[int]):
        self.tokenizer = tokenizer
        self.token_ids: list[int] = []
        self.output_text = ""
        self.stop = list(params.stop) if params else []
        self.include_stop = bool(params and params.include_stop_str_in_output)
        self.hold_back = params.output_text_buffer_length if params else 0
        self.skip_special = params.skip_special_tokens if params else True
        self.spaces_between = params.spaces_between_special_tokens if params else True
        self.min_tokens = params.min_tokens if params else 0
        self._sent = 0
        self.active = tokenizer is not None and (params is None or params.detokenize)
        if self.active:
            # decode with a little prompt context so that the first generated token gets its leading space
            self._ctx = list(prompt_ids[-6:])
            self._prefix_offset = 0
            self._read_offset = len(self._ctx)
            self._all = list(self._ctx)

    def _decode(self, ids: list[int]) -> str:
        return self.tokenizer.decode(ids, skip_special_tokens=self.skip_special,
                                     spaces_between_special_tokens=self.spaces_between) \
            if _accepts_sbst(self.tokenizer) else self.tokenizer.decode(ids, skip_special_tokens=self.skip_special)

    def update(self, new_ids: list[int], stop_terminated: bool) -> Optional[str]:
        """Append tokens; returns the stop string that matched, if any."""
        if not new_ids:
            return None
        if stop_terminated and not self.include_stop:
            # the final token is a stop/eos token: keep it out of the text
            skipped = new_ids[-1]
            new_ids = new_ids[:-1]
        else:
            skipped = None
        self.token_ids.extend(new_ids)
        if skipped is not None:
            self.token_ids.append(skipped)
        if not self.active:
            return None
        before = len(self.output_text)
        for t in new_ids:
            self._all.append(t)
            prefix = self._decode(self._all[self._prefix_offset:self._read_offset])
            full = self._decode(self._all[self._prefix_offset:])
            if len(full) > len(prefix) and not full.endswith("�"):
                self.output_text += full[len(prefix):]
                self._prefix_offset = self._read_offset
                self._read_offset = len(self._all)
        if self.stop and len(self.token_ids) > self.min_tokens:
            hit = self._check_stop(len(self.output_text) - before)
            if hit is not None:
                s, cut = hit
                if cut >= 0:
                    self.output_text = self.output_text[:cut]
                return s
        return None

    def _check_stop(self, new_chars: int) -> Optional[tuple[str, int]]:
        if not new_chars:
            return None
        text = self.output_text
        for s in self.stop:
            start = max(0, len(text) - new_chars - len(s) + 1)
            i = text.find(s, start)
            if i < 0:
                continue
            if self.include_stop:
                end = i + len(s)
                return s, (end if end < len(text) else -1)
            return s, i
        return None

    def get_next_output_text(self, finished: bool, delta: bool) -> str:
        hold = 0 if finished else self.hold_back
        length = len(self.output_text) - hold
        if not delta:
            return self.output_text[:length] if hold else self.output_text
        if length > self._sent:
            out = self.output_text[self._sent:length]
            self._sent = length
            return out
        return ""


_SBST: dict[int, bool] = {}


def _accepts_sbst(tok) -> bool:
    k = id(type(tok))
    if k not in _SBST:
        import inspect
        try:
            _SBST[k] = "spaces_between_special_tokens" in inspect.signature(tok.decode).parameters or \
                any(p.kind == p.VAR_KEYWORD for p in inspect.signature(tok.decode).parameters.values())
        except (TypeError, ValueError):
            _SBST[k] = False
    return _SBST[k]


# ---------------------------------------------------------------------------
class LogprobsProcessor:
    def __init__(self, tokenizer, params: Optional[SamplingParams]):
        self.tokenizer = tokenizer
        self.num_logprobs = params.logprobs if params else None
        self.num_prompt_logprobs = params.prompt_logprobs if params else None
        self.logprobs: Optional[list] = [] if self.num_logprobs is not None else None
        self.prompt_logprobs: Optional[list] = [None] if self.num_prompt_logprobs is not None else None
        self.cumulative_logprob: Optional[float] = 0.0 if self.num_logprobs is not None else None
        self.detok = bool(tokenizer is not None and (params is None or params.detokenize))

    def _make(self, ids: list[int], lps: list[float], rank: int, k: int) -> dict[int, Logprob]:
        out: dict[int, Logprob] = {}
        for j, (t, lp) in enumerate(zip(ids[: k + 1], lps[: k + 1])):
            if t in out:
                continue
            r = rank if j == 0 else j
            out[t] = Logprob(lp, r, self.tokenizer.decode([t]) if self.detok else None)
        return out

    def update(self, out: EngineCoreOutput) -> None:
        if out.new_logprobs is not None and self.logprobs is not None:
            ll = out.new_logprobs
            for ids, lps, rank in zip(ll.logprob_token_ids, ll.logprobs, ll.sampled_token_ranks):
                self.logprobs.append(self._make(ids, lps, rank, self.num_logprobs))
                self.cumulative_logprob += lps[0]
        if out.new_prompt_logprobs is not None and self.prompt_logprobs is not None:
            ll = out.new_prompt_logprobs
            for ids, lps, rank in zip(ll.logprob_token_ids, ll.logprobs, ll.sampled_token_ranks):
                self.prompt_logprobs.append(self._make(ids, lps, rank, self.num_prompt_logprobs))

    def pop_prompt_logprobs(self):
        p = self.prompt_logprobs
        if p:
            self.prompt_logprobs = []
        return p


# ---------------------------------------------------------------------------
class ParentRequest:
    """Fan-out of one user request with n > 1 into n engine requests and fan-in of their outputs."""

    def __init__(self, request_id: str, params: SamplingParams):
        self.request_id = request_id
        self.params = params
        self.child_ids: set[str] = set()
        self.outputs: list[Optional[CompletionOutput]] = [None] * params.n
        self.max_num_generation_tokens = 0

    def child(self, index: int) -> tuple[str, SamplingParams]:
        p = self.params.clone()
        p.n = 1
        if p.seed is not None:
            p.seed += index
        cid = f"{index}_{self.request_id}"
        self.child_ids.add(cid)
        return cid, p

    def add_output(self, child_id: str, co: CompletionOutput) -> tuple[list[CompletionOutput], bool]:
        if co.finished():
            self.child_ids.discard(child_id)
        if self.params.output_kind != RequestOutputKind.FINAL_ONLY:
            return [co], not self.child_ids
        self.outputs[co.index] = co
        done = not self.child_ids
        return ([o for o in self.outputs if o is not None] if done else []), done


class RequestOutputCollector:
    """Single-producer single-consumer hand-off between the output handler task and one
    ``generate()`` coroutine; merges outputs when the consumer is slower."""

    def __init__(self, kind: RequestOutputKind):
        self.aggregate = kind == RequestOutputKind.DELTA
        self.output: Union[RequestOutput, PoolingRequestOutput, Exception, None] = None
        self.ready = asyncio.Event()

    def put(self, out) -> None:
        if self.output is None or isinstance(out, Exception):
            self.output = out
            self.ready.set()
        elif isinstance(self.output, RequestOutput):
            self.output.add(out, self.aggregate)

    async def get(self):
        while self.output is None:
            await self.ready.wait()
        out, self.output = self.output, None
        self.ready.clear()
        if isinstance(out, Exception):
            raise out
        return out

    def get_nowait(self):
        out, self.output = self.output, None
        if out is not None:
            self.ready.clear()
        if isinstance(out, Exception):
            raise out
        return out


@dataclass
class OutputProcessorOutput:
    request_outputs: list = field(default_factory=list)
    reqs_to_abort: list[str] = field(default_factory=list)


class RequestState:
    def __init__(self, request_id, parent: Optional[ParentRequest], index: int, lora_name, kind, prompt,
                 prompt_ids, logprobs: Optional[LogprobsProcessor], detok: Optional[IncrementalDetokenizer],
                 max_tokens_param, arrival_time, queue: Optional[RequestOutputCollector], log_stats: bool,
                 is_pooling: bool):
        self.request_id, self.parent, self.index = request_id, parent, index
        self.lora_name, self.kind, self.prompt, self.prompt_ids = lora_name, kind, prompt, prompt_ids
        self.logprobs, self.detok = logprobs, detok
        self.max_tokens_param = max_tokens_param
        self.queue = queue
        self.is_prefilling = True
        self.is_pooling = is_pooling
        self.stats = RequestStateStats(arrival_time=arrival_time) if log_stats else None
        self._sent_tokens = 0
        self.num_cached_tokens = 0


class OutputProcessor:
    def __init__(self, tokenizer_group, log_stats: bool):
        self.tokenizer = tokenizer_group
        self.log_stats = log_stats
        self.request_states: dict[str, RequestState] = {}
        self.parent_requests: dict[str, ParentRequest] = {}
        self.lora_states = LoRARequestStates()

    def get_num_unfinished_requests(self) -> int:
        return len(self.request_states)

    def has_unfinished_requests(self) -> bool:
        return bool(self.request_states)

    def propagate_error(self, e: Exception) -> None:
        for st in self.request_states.values():
            if st.queue is not None:
                st.queue.put(e)

    def abort_requests(self, request_ids) -> list[str]:
        aborted = []
        for rid in request_ids:
            st = self.request_states.pop(rid, None)
            if st is not None:
                aborted.append(rid)
                if st.parent is not None:
                    st.parent.child_ids.discard(rid)
                    if not st.parent.child_ids:
                        self.parent_requests.pop(st.parent.request_id, None)
            elif rid in self.parent_requests:
                parent = self.parent_requests.pop(rid)
                aborted.extend(self.abort_requests(list(parent.child_ids)))
        return aborted

    def add_request(self, request: EngineCoreRequest, prompt: Optional[str], parent: Optional[ParentRequest] = None,
                    index: int = 0, queue: Optional[RequestOutputCollector] = None) -> None:
        rid = request.request_id
        if rid in self.request_states:
            raise ValueError(f"request id {rid} is already running")
        sp = request.sampling_params
        tok = self.tokenizer.get_lora_tokenizer(request.lora_request) if self.tokenizer is not None else None
        kind = sp.output_kind if sp else request.pooling_params.output_kind
        self.request_states[rid] = RequestState(
            rid, parent, index, request.lora_request.lora_name if request.lora_request else None, kind, prompt,
            request.prompt_token_ids, LogprobsProcessor(tok, sp) if sp else None,
            IncrementalDetokenizer(tok, sp, request.prompt_token_ids) if sp else None,
            sp.max_tokens if sp else None, request.arrival_time, queue, self.log_stats, sp is None)
        if parent is not None:
            self.parent_requests[parent.request_id] = parent

    # ------------------------------------------------------------------
    def process_outputs(self, outputs: list[EngineCoreOutput], engine_core_timestamp: Optional[float] = None,
                        iteration_stats: Optional[IterationStats] = None) -> OutputProcessorOutput:
        result = OutputProcessorOutput()
        for eco in outputs:
            st = self.request_states.get(eco.request_id)
            if st is None:
                continue   # aborted by the client in the meantime
            if iteration_stats is not None and st.stats is not None:
                iteration_stats.update_from_output(eco, engine_core_timestamp, st.is_prefilling, len(st.prompt_ids),
                                                   st.stats, self.lora_states)
            new_ids = eco.new_token_ids
            finish, stop_reason = eco.finish_reason, eco.stop_reason
            st.num_cached_tokens = eco.num_cached_tokens
            st.is_prefilling = False
            if eco.pooling_output is not None or st.is_pooling:
                ro = self._pooling_output(st, eco)
            else:
                stop_str = st.detok.update(new_ids, finish == FinishReason.STOP)
                if stop_str is not None:
                    finish, stop_reason = FinishReason.STOP, stop_str
                st.logprobs.update(eco)
                ro = self._request_output(st, new_ids, finish, stop_reason, eco.kv_transfer_params)
            if ro is not None:
                if st.queue is not None:
                    st.queue.put(ro)
                else:
                    result.request_outputs.append(ro)
            if finish is not None:
                self.request_states.pop(eco.request_id, None)
                if st.parent is not None and not st.parent.child_ids:
                    self.parent_requests.pop(st.parent.request_id, None)
                if not eco.finished:
                    # stopped by a stop string the engine core cannot see: tell it
                    result.reqs_to_abort.append(eco.request_id)
                if iteration_stats is not None and st.stats is not None:
                    iteration_stats.update_from_finished_request(finish, len(st.prompt_ids), st.max_tokens_param,
                                                                 st.stats)
                    self.lora_states.running_requests.discard(eco.request_id)
                    # per USER request: the longest of its n samples, and n (reference: ParentRequest.
                    # observe_finished_request, vllm/v1/engine/parallel_sampling.py:110-133)
                    gen = st.stats.num_generation_tokens
                    if st.parent is None:
                        iteration_stats.max_num_generation_tokens_iter.append(gen)
                        iteration_stats.n_params_iter.append(1)
                    else:
                        st.parent.max_num_generation_tokens = max(st.parent.max_num_generation_tokens, gen)
                        if not st.parent.child_ids:
                            iteration_stats.max_num_generation_tokens_iter.append(st.parent.max_num_generation_tokens)
                            iteration_stats.n_params_iter.append(st.parent.params.n)
        return result

    def _pooling_output(self, st: RequestState, eco: EngineCoreOutput):
        if eco.pooling_output is None:
            return None
        return PoolingRequestOutput(st.request_id, PoolingOutput(eco.pooling_output), st.prompt_ids, True)

    def _request_output(self, st: RequestState, new_ids, finish, stop_reason, kv_params) -> Optional[RequestOutput]:
        finished = finish is not None
        final_only = st.kind == RequestOutputKind.FINAL_ONLY
        if not finished and final_only:
            return None
        delta = st.kind == RequestOutputKind.DELTA
        text = st.detok.get_next_output_text(finished, delta)
        if delta:
            ids = st.detok.token_ids[st._sent_tokens:]
            st._sent_tokens = len(st.detok.token_ids)
            lps = st.logprobs.logprobs[-len(ids):] if (st.logprobs.logprobs is not None and ids) else \
                ([] if st.logprobs.logprobs is not None else None)
            if not ids and not text and not finished:
                return None
        else:
            ids = list(st.detok.token_ids)
            lps = st.logprobs.logprobs
        co = CompletionOutput(index=st.index, text=text, token_ids=ids,
                              cumulative_logprob=st.logprobs.cumulative_logprob, logprobs=lps,
                              finish_reason=str(finish) if finished else None,
                              stop_reason=stop_reason if finished else None)
        rid, outputs, all_done = st.request_id, [co], finished
        if st.parent is not None:
            rid = st.parent.request_id
            outputs, all_done = st.parent.add_output(st.request_id, co)
            if not outputs:
                return None
        prompt_lp = st.logprobs.pop_prompt_logprobs() if delta else st.logprobs.prompt_logprobs
        return RequestOutput(request_id=rid, prompt=st.prompt, prompt_token_ids=st.prompt_ids,
                             prompt_logprobs=prompt_lp, outputs=outputs, finished=all_done,
                             num_cached_tokens=st.num_cached_tokens, kv_transfer_params=kv_params)
