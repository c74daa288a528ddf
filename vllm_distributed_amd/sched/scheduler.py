"""Continuous-batching scheduler with a unified token budget.

There is no prefill phase and no decode phase: every request has
``num_computed_tokens`` and a target ``num_tokens_with_spec``; each step hands
out tokens from one budget, running requests first, then waiting ones.  That
one rule gives chunked prefill, prefix-cache hits, speculative decoding and
preemption-by-recompute for free.

Reference behaviour: vllm/v1/core/sched/scheduler.py (schedule :413-856,
update_from_output :1012-1178, finish/free :1203-1290, KV-connector hooks
:1348-1398), vllm/v1/core/sched/utils.py (check_stop).
"""
from __future__ import annotations

import itertools
import time
from collections import defaultdict, deque
from typing import Iterable, Optional, Union

from ..config import VllmConfig
from ..logger import init_logger
from ..metrics.stats import SchedulerStats, SpecDecodingStats
from ..request import (EngineCoreEventType, EngineCoreOutput, EngineCoreOutputs, Request, RequestStatus)
from .encoder_cache import EncoderCacheManager
from .kv_cache_interface import KVCacheConfig
from .kv_cache_manager import KVCacheBlocks, KVCacheManager
from .kv_events import KVEventBatch, make_event_publisher
from .output import (CachedRequestData, ModelRunnerOutput, NewRequestData, SchedulerOutput)
from .request_queue import create_request_queue
from .token_parallel import TokenParallelScheduler

logger = init_logger(__name__)


def check_stop(request: Request, max_model_len: int) -> bool:
    """Decide whether the last appended token ends the request; sets status."""
    if request.num_tokens >= max_model_len or request.num_output_tokens >= request.max_tokens:
        request.status = RequestStatus.FINISHED_LENGTH_CAPPED
        return True
    if request.pooling_params is not None:
        return False
    sp = request.sampling_params
    last = request.output_token_ids[-1]
    if request.num_output_tokens < sp.min_tokens:
        return False
    if not sp.ignore_eos and last == request.eos_token_id:
        request.status = RequestStatus.FINISHED_STOPPED
        return True
    if last in (sp.stop_token_ids or ()):
        request.status = RequestStatus.FINISHED_STOPPED
        request.stop_reason = last
        return True
    return False


class SchedulerInterface:
    def schedule(self) -> SchedulerOutput: raise NotImplementedError
    def update_from_output(self, so, mro) -> dict[int, EngineCoreOutputs]: raise NotImplementedError
    def add_request(self, request: Request) -> None: raise NotImplementedError
    def finish_requests(self, request_ids, finished_status) -> None: raise NotImplementedError
    def get_num_unfinished_requests(self) -> int: raise NotImplementedError
    def has_finished_requests(self) -> bool: raise NotImplementedError
    def reset_prefix_cache(self) -> bool: raise NotImplementedError
    def get_request_counts(self) -> tuple[int, int]: raise NotImplementedError
    def make_stats(self): raise NotImplementedError
    def shutdown(self) -> None: raise NotImplementedError

    def has_unfinished_requests(self) -> bool:
        return self.get_num_unfinished_requests() > 0

    def has_requests(self) -> bool:
        return self.has_unfinished_requests() or self.has_finished_requests()


class Scheduler(SchedulerInterface):
    def __init__(self, vllm_config: VllmConfig, kv_cache_config: KVCacheConfig,
                 structured_output_manager=None, include_finished_set: bool = False,
                 log_stats: bool = False, tknp: Optional[TokenParallelScheduler] = None,
                 kv_connector=None):
        self.vllm_config = vllm_config
        sc, cc = vllm_config.scheduler_config, vllm_config.cache_config
        self.scheduler_config = sc
        self.cache_config = cc
        self.kv_cache_config = kv_cache_config
        self.structured_output_manager = structured_output_manager
        self.log_stats = log_stats
        self.max_num_running_reqs = sc.max_num_seqs
        self.max_num_scheduled_tokens = sc.max_num_batched_tokens
        self.max_model_len = sc.max_model_len
        self.max_loras = sc.max_loras
        self.finished_req_ids_dict: Optional[dict[int, set[str]]] = defaultdict(set) \
            if include_finished_set else None

        # KV connector (scheduler role) -------------------------------------
        self.connector = kv_connector
        if self.connector is None and vllm_config.kv_transfer_config is not None:
            from ..parallel.kv_transfer.factory import KVConnectorFactory, KVConnectorRole
            self.connector = KVConnectorFactory.create(vllm_config, KVConnectorRole.SCHEDULER)
        ev = vllm_config.kv_events_config
        self.kv_event_publisher = make_event_publisher(ev, vllm_config.parallel_config.data_parallel_rank)
        enable_events = bool(ev and ev.enable_kv_cache_events)

        # token parallelism -------------------------------------------------
        self.tknp = tknp
        enable_caching = cc.enable_prefix_caching
        if self.tknp is not None and enable_caching:
            logger.warning_once("prefix caching is disabled under token parallelism: placement "
                                "decisions must be identical on every rank")
            enable_caching = False

        spec = vllm_config.speculative_config
        self.num_spec_tokens = spec.num_speculative_tokens if spec else 0
        self.use_eagle = bool(spec and spec.use_eagle())
        self.num_lookahead_tokens = self.num_spec_tokens if self.use_eagle else 0

        self.kv_cache_manager = KVCacheManager(
            kv_cache_config, self.max_model_len, enable_caching, cc.prefix_caching_hash_algo,
            use_eagle=self.use_eagle, log_stats=log_stats, enable_kv_cache_events=enable_events)
        self.block_size = cc.block_size

        self.requests: dict[str, Request] = {}
        self.policy = sc.policy
        self.waiting = create_request_queue(sc.policy)
        self.running: list[Request] = []
        # finished between two steps; sent to workers so they drop cached state
        self.finished_req_ids: set[str] = set()
        # P/D: requests whose remote KV arrived
        self.finished_recving_kv_req_ids: set[str] = set()
        self.failed_recving_kv_req_ids: set[str] = set()
        # requests finished (aborted) while their remote KV was still arriving: the blocks stay allocated
        # until the worker reports the transfer over, or the copy would land in somebody else's blocks
        self.abandoned_recving: dict[str, Request] = {}

        self.encoder_cache_manager = EncoderCacheManager(sc.encoder_cache_size)
        self.max_num_encoder_input_tokens = sc.max_num_encoder_input_tokens
        self._step = 0
        self.async_mode = bool(self.scheduler_config.async_scheduling)
        self.last_schedule_preempted = False

    # ------------------------------------------------------------------
    # schedule
    # ------------------------------------------------------------------
    def schedule(self) -> SchedulerOutput:
        new_reqs: list[Request] = []
        resumed_reqs: list[Request] = []
        running_reqs: list[Request] = []
        preempted: list[Request] = []
        req_to_new_blocks: dict[str, KVCacheBlocks] = {}
        num_scheduled: dict[str, int] = {}
        spec_tokens: dict[str, list[int]] = {}
        encoder_inputs: dict[str, list[int]] = {}
        budget = self.max_num_scheduled_tokens
        encoder_budget = self.max_num_encoder_input_tokens
        structured_ids: dict[str, int] = {}
        now = time.monotonic()
        tknp = self.tknp

        # 1) running requests ------------------------------------------------
        i = 0
        while i < len(self.running) and budget > 0:
            req = self.running[i]
            if req.num_output_placeholders and self._complete_with_placeholders(req):
                i += 1               # its last token is in flight: nothing more to compute for it
                continue
            n_new = req.num_tokens_with_spec + req.num_output_placeholders - req.num_computed_tokens
            thr = self.scheduler_config.long_prefill_token_threshold
            if 0 < thr < n_new:
                n_new = thr
            n_new = min(n_new, budget, self.max_model_len - 1 - req.num_computed_tokens)
            enc_ids, n_new, new_enc_budget = self._schedule_encoder(req, req.num_computed_tokens, n_new,
                                                                   encoder_budget)
            if n_new <= 0:
                # nothing to do for it this step (e.g. PP: its token is still in flight)
                i += 1
                continue
            owned = True
            while True:
                if tknp is not None:
                    owned = tknp.is_mine(req)
                    ok = tknp.can_reserve(req.request_id, req.num_computed_tokens + n_new
                                          + self.num_lookahead_tokens)
                    blocks = self.kv_cache_manager.allocate_slots(
                        req, n_new, num_lookahead_tokens=self.num_lookahead_tokens, owned=owned) if ok else None
                    if ok and blocks is None:
                        raise RuntimeError("token-parallel virtual pool and real pool disagree")
                else:
                    blocks = self.kv_cache_manager.allocate_slots(
                        req, n_new, num_lookahead_tokens=self.num_lookahead_tokens)
                if blocks is not None:
                    break
                victim = self._pick_victim(req)
                vpos = self.running.index(victim)
                self._preempt(victim, now)
                preempted.append(victim)
                if victim is req:
                    blocks = None
                    break
                if vpos < i:
                    # the victim was already given tokens in this step: take them back
                    i -= 1
                    vid = victim.request_id
                    if vid in num_scheduled:
                        running_reqs.remove(victim)
                        budget += num_scheduled.pop(vid)
                        req_to_new_blocks.pop(vid, None)
                        spec_tokens.pop(vid, None)
                        encoder_inputs.pop(vid, None)
                        structured_ids.pop(vid, None)
            if blocks is None:
                # `req` itself was preempted; whatever sits at index i now is unvisited
                continue
            if tknp is not None:
                tknp.reserve(req.request_id, req.num_computed_tokens + n_new + self.num_lookahead_tokens)
            running_reqs.append(req)
            req_to_new_blocks[req.request_id] = blocks
            num_scheduled[req.request_id] = n_new
            budget -= n_new
            i += 1
            if req.spec_token_ids:
                n_spec = n_new + req.num_computed_tokens - req.num_tokens
                if n_spec > 0:
                    del req.spec_token_ids[n_spec:]
                    spec_tokens[req.request_id] = list(req.spec_token_ids)
            if enc_ids:
                encoder_inputs[req.request_id] = enc_ids
                for k in enc_ids:
                    self.encoder_cache_manager.allocate(req, k)
                encoder_budget = new_enc_budget
            if req.use_structured_output:
                structured_ids[req.request_id] = len(running_reqs) - 1

        # 2) waiting requests -----------------------------------------------
        scheduled_loras: set[int] = set()
        if self.max_loras:
            scheduled_loras = {r.lora_request.lora_int_id for r in running_reqs
                               if r.lora_request and r.lora_request.lora_int_id > 0}
        skipped = create_request_queue(self.policy)
        if not preempted:
            if tknp is not None and self.waiting:
                tknp.assign_ranks(self.waiting)
            while self.waiting and budget > 0:
                if len(self.running) >= self.max_num_running_reqs:
                    break
                req = self.waiting.peek_request()

                if req.status == RequestStatus.WAITING_FOR_REMOTE_KVS:
                    if not self._update_waiting_for_remote_kv(req):
                        self.waiting.pop_request()
                        skipped.prepend_request(req)
                        continue
                    req.status = RequestStatus.WAITING
                if req.status == RequestStatus.WAITING_FOR_FSM:
                    so = req.structured_output_request
                    if so is not None and so.ready:
                        req.status = RequestStatus.WAITING
                    else:
                        self.waiting.pop_request()
                        skipped.prepend_request(req)
                        continue
                if (self.max_loras and req.lora_request and req.lora_request.lora_int_id > 0
                        and len(scheduled_loras) >= self.max_loras
                        and req.lora_request.lora_int_id not in scheduled_loras):
                    self.waiting.pop_request()
                    skipped.prepend_request(req)
                    continue

                owned = tknp.is_mine(req) if tknp is not None else True
                n_ext = 0
                load_async = False
                if req.num_computed_tokens == 0:
                    hit_blocks, n_local = self.kv_cache_manager.get_computed_blocks(req, owned=owned)
                    if self.connector is not None and owned:
                        n_ext, load_async = self.connector.get_num_new_matched_tokens(req, n_local)
                    n_computed = n_local + n_ext
                else:
                    # resumed after async KV load: blocks already hold the tokens
                    hit_blocks = self.kv_cache_manager._empty()
                    n_local = 0
                    n_computed = req.num_computed_tokens

                enc_ids, new_enc_budget = None, encoder_budget
                if load_async:
                    n_new = 0   # KV arrives on its own; compute nothing now
                else:
                    n_new = req.num_tokens - n_computed
                    thr = self.scheduler_config.long_prefill_token_threshold
                    if 0 < thr < n_new:
                        n_new = thr
                    if not self.scheduler_config.chunked_prefill_enabled and n_new > budget:
                        # whole prompts only
                        self.waiting.pop_request()
                        skipped.prepend_request(req)
                        continue
                    n_new = min(n_new, budget)
                    assert n_new > 0
                    enc_ids, n_new, new_enc_budget = self._schedule_encoder(req, n_computed, n_new,
                                                                           encoder_budget)
                    if n_new == 0:
                        break   # encoder budget exhausted: nothing behind it may overtake

                if tknp is not None and not tknp.can_reserve(req.request_id,
                                                             n_computed + max(n_new, 1) + self.num_lookahead_tokens):
                    break
                blocks = self.kv_cache_manager.allocate_slots(
                    req, n_new + n_ext, n_local, hit_blocks,
                    num_lookahead_tokens=self.num_lookahead_tokens, delay_cache_blocks=load_async,
                    owned=owned)
                if blocks is None:
                    break   # out of KV space: stop admitting
                if tknp is not None:
                    tknp.reserve(req.request_id, n_computed + max(n_new, 1) + self.num_lookahead_tokens)

                if self.connector is not None and owned:
                    self.connector.update_state_after_alloc(req, hit_blocks + blocks, n_ext)
                self.waiting.pop_request()
                if load_async:
                    skipped.add_request(req)
                    req.status = RequestStatus.WAITING_FOR_REMOTE_KVS
                    continue

                if req.use_structured_output:
                    structured_ids[req.request_id] = len(running_reqs) + len(new_reqs) + len(resumed_reqs)
                self.running.append(req)
                if self.log_stats:
                    req.record_event(EngineCoreEventType.SCHEDULED, now)
                if req.status == RequestStatus.WAITING:
                    new_reqs.append(req)
                elif req.status == RequestStatus.PREEMPTED:
                    resumed_reqs.append(req)
                else:
                    raise RuntimeError(f"unexpected status {req.status.name} in the waiting queue")
                if self.max_loras and req.lora_request:
                    scheduled_loras.add(req.lora_request.lora_int_id)
                req_to_new_blocks[req.request_id] = self.kv_cache_manager.get_blocks(req.request_id)
                num_scheduled[req.request_id] = n_new
                budget -= n_new
                req.status = RequestStatus.RUNNING
                req.num_computed_tokens = n_computed
                if req.num_cached_tokens < 0:
                    req.num_cached_tokens = n_computed
                if enc_ids:
                    encoder_inputs[req.request_id] = enc_ids
                    for k in enc_ids:
                        self.encoder_cache_manager.allocate(req, k)
                    encoder_budget = new_enc_budget
        if skipped:
            self.waiting.prepend_requests(skipped)

        total = sum(num_scheduled.values())
        assert total <= self.max_num_scheduled_tokens
        assert budget >= 0
        assert len(self.running) <= self.max_num_running_reqs
        assert len(new_reqs) + len(resumed_reqs) + len(running_reqs) <= len(self.running)

        # cascade attention hint: blocks shared by every running request
        common = [0] * len(self.kv_cache_config.kv_cache_groups)
        if self.running and tknp is None:
            common = self.kv_cache_manager.get_num_common_prefix_blocks(self.running[0], len(self.running))

        bitmask = None
        if structured_ids and self.structured_output_manager is not None:
            bitmask = self.structured_output_manager.grammar_bitmask(self.requests, structured_ids, spec_tokens)

        new_data = [NewRequestData.from_request(r, req_to_new_blocks[r.request_id].get_block_ids())
                    for r in new_reqs]
        cached = self._make_cached_request_data(running_reqs, resumed_reqs, num_scheduled, spec_tokens,
                                                req_to_new_blocks)
        out = SchedulerOutput(
            scheduled_new_reqs=new_data, scheduled_cached_reqs=cached, num_scheduled_tokens=num_scheduled,
            total_num_scheduled_tokens=total, scheduled_spec_decode_tokens=spec_tokens,
            scheduled_encoder_inputs=encoder_inputs, num_common_prefix_blocks=common,
            finished_req_ids=self.finished_req_ids,
            free_encoder_input_ids=self.encoder_cache_manager.get_freed_ids(),
            structured_output_request_ids=structured_ids, grammar_bitmask=bitmask)
        if tknp is not None:
            out.tknp = tknp.allocation(num_scheduled)
        if self.connector is not None:
            out.kv_connector_metadata = self.connector.build_connector_meta(out)
        events = self.kv_cache_manager.take_events()
        if events:
            self.kv_event_publisher.publish(KVEventBatch(ts=time.time(), events=events))

        # the step is now committed: advance the computed-token counters so the
        # next schedule() (pipeline parallel / async scheduling) starts after it
        for rid, n in num_scheduled.items():
            req = self.requests[rid]
            req.num_computed_tokens += n
            if self.async_mode and req.num_computed_tokens == req.num_tokens + req.num_output_placeholders:
                # this step samples a token the scheduler will only see one step later: stand-in until then
                req.num_output_placeholders += 1
        self.last_schedule_preempted = bool(preempted)
        self.finished_req_ids = set()
        self._step += 1
        return out

    def _complete_with_placeholders(self, req: Request) -> bool:
        """Run-ahead scheduling: would the tokens already in flight end the request by length?"""
        n_out = req.num_output_tokens + req.num_output_placeholders
        return (req.max_tokens is not None and n_out >= req.max_tokens) or \
            req.num_tokens + req.num_output_placeholders >= self.max_model_len

    def can_run_ahead(self, so: SchedulerOutput) -> bool:
        """May the next step be scheduled before the results of ``so`` are known?  Only when nothing about the
        next step's inputs other than the sampled token itself depends on those results: no grammar state, no
        penalties or token-history processors, no prompt log-probabilities (their extraction synchronises)."""
        if not self.async_mode or so.total_num_scheduled_tokens == 0:
            return False
        for rid in so.num_scheduled_tokens:
            req = self.requests.get(rid)
            if req is not None and not req.plain_sampling:
                return False
        return True

    def _pick_victim(self, current: Request) -> Request:
        if self.tknp is not None:
            # only freeing blocks on the same rank helps
            rank = self.tknp.rank_of(current.request_id)
            cands = [r for r in self.running if self.tknp.rank_of(r.request_id) == rank]
        else:
            cands = self.running
        if self.policy == "priority":
            return max(cands, key=lambda r: (r.priority, r.arrival_time))
        return cands[-1]

    def _preempt(self, req: Request, now: float) -> None:
        self.running.remove(req)
        self.kv_cache_manager.free(req)
        if self.tknp is not None:
            self.tknp.release(req.request_id)
        req.status = RequestStatus.PREEMPTED
        req.num_computed_tokens = 0
        req.num_output_placeholders = 0
        req.spec_token_ids = []
        req.num_preemptions += 1
        if self.log_stats:
            req.record_event(EngineCoreEventType.PREEMPTED, now)
        self.waiting.prepend_request(req)

    def _make_cached_request_data(self, running_reqs, resumed_reqs, num_scheduled, spec_tokens,
                                  req_to_new_blocks) -> CachedRequestData:
        d = CachedRequestData()
        use_pp = self.vllm_config.parallel_config.pipeline_parallel_size > 1
        # decode steps mostly allocate nothing: those requests share one (never mutated) empty id tuple
        nothing = self.kv_cache_manager._no_blocks
        no_ids = nothing.get_block_ids()
        for req, resumed in itertools.chain(zip(running_reqs, itertools.repeat(False)),
                                            zip(resumed_reqs, itertools.repeat(True))):
            rid = req.request_id
            d.req_ids.append(rid)
            d.resumed_from_preemption.append(resumed)
            if use_pp or resumed:
                # first-stage workers never see sampled tokens: ship the ones this step consumes
                n_tok = num_scheduled[rid] - len(spec_tokens.get(rid, ()))
                s = req.num_computed_tokens
                d.new_token_ids.append(list(req.all_token_ids[s:s + n_tok]))
            else:
                d.new_token_ids.append([])
            blocks = req_to_new_blocks[rid]
            d.new_block_ids.append(no_ids if blocks is nothing else blocks.get_block_ids())
            d.num_computed_tokens.append(req.num_computed_tokens)
        return d

    def _schedule_encoder(self, req: Request, n_computed: int, n_new: int, enc_budget: int):
        """Multimodal: encoder outputs needed by tokens [n_computed, n_computed+n_new)."""
        if n_new <= 0 or not req.has_encoder_inputs:
            return None, n_new, enc_budget
        chosen: list[int] = []
        for i, pos in enumerate(req.mm_positions):
            start, length = pos.offset, pos.length
            if start >= n_computed + n_new:
                break
            if start + length <= n_computed:
                continue
            if self.encoder_cache_manager.has_cache(req, i):
                continue
            if (self.scheduler_config.disable_chunked_mm_input and n_computed < start
                    and n_computed + n_new < start + length):
                n_new = start - n_computed
                break
            if not self.encoder_cache_manager.can_allocate(req, i) or length > enc_budget:
                # stop right before this item
                n_new = max(start - n_computed, 0) if n_computed < start else 0
                break
            enc_budget -= length
            chosen.append(i)
        return chosen, n_new, enc_budget

    # ------------------------------------------------------------------
    # results
    # ------------------------------------------------------------------
    def update_from_output(self, scheduler_output: SchedulerOutput,
                           mro: ModelRunnerOutput) -> dict[int, EngineCoreOutputs]:
        sampled = mro.sampled_token_ids
        spec = mro.spec_token_ids
        num_scheduled = scheduler_output.num_scheduled_tokens
        outputs: dict[int, list[EngineCoreOutput]] = defaultdict(list)
        spec_stats: Optional[SpecDecodingStats] = None
        stopped: list[Request] = []
        stopped_preempted: list[Request] = []   # run-ahead: finished by a token that was in flight at preemption

        for rid, n_sched in num_scheduled.items():
            if n_sched == 0:
                continue
            req = self.requests.get(rid)
            if req is None:
                continue   # aborted while the step was in flight
            idx = mro.req_id_to_index.get(rid)
            new_ids = sampled[idx] if (sampled and idx is not None) else []
            drafted = scheduler_output.scheduled_spec_decode_tokens.get(rid)
            if drafted:
                # rejected drafts were counted as computed in schedule(): roll them back
                n_rejected = len(drafted) + 1 - len(new_ids)
                req.num_computed_tokens -= n_rejected
                if self.log_stats:
                    spec_stats = spec_stats or SpecDecodingStats.new(self.num_spec_tokens)
                    spec_stats.observe_draft(len(drafted), len(new_ids) - 1)

            # encoder outputs whose tokens are now all in the KV cache can go
            if req.has_encoder_inputs:
                for k in list(self.encoder_cache_manager.get_cached_input_ids(req)):
                    p = req.mm_positions[k]
                    if p.offset + p.length <= req.num_computed_tokens:
                        self.encoder_cache_manager.free_encoder_input(req, k)

            finished = False
            kv_params = None
            was_preempted = req.status == RequestStatus.PREEMPTED
            if req.num_output_placeholders:
                req.num_output_placeholders = max(0, req.num_output_placeholders - len(new_ids))
            for n, tok in enumerate(new_ids, 1):
                req.append_output_token_ids(tok)
                if check_stop(req, self.max_model_len):
                    finished = True
                    del new_ids[n:]
                    break
            pooled = mro.pooler_output[idx] if (mro.pooler_output and idx is not None) else None
            if pooled is not None and not finished:
                req.status = RequestStatus.FINISHED_STOPPED
                finished = True
            if finished:
                kv_params = self._free_request(req)
                (stopped_preempted if was_preempted else stopped).append(req)

            new_lp = mro.logprobs.slice(idx, idx + 1) if (mro.logprobs and new_ids and idx is not None
                                                           and sp_logprobs(req)) else None
            if new_ids and self.structured_output_manager is not None and req.use_structured_output:
                req.structured_output_request.accept_tokens(rid, new_ids)
            if spec is not None and idx is not None and not finished:
                drafts = spec[idx]
                if req.use_structured_output and self.structured_output_manager is not None:
                    drafts = req.structured_output_request.validate_tokens(drafts)
                req.spec_token_ids = list(drafts)

            prompt_lp = mro.prompt_logprobs_dict.get(rid)
            if new_ids or pooled is not None or kv_params or prompt_lp is not None:
                outputs[req.client_index].append(EngineCoreOutput(
                    request_id=rid, new_token_ids=list(new_ids), new_logprobs=new_lp,
                    new_prompt_logprobs=prompt_lp, pooling_output=pooled,
                    finish_reason=req.get_finished_reason(), stop_reason=req.stop_reason,
                    events=req.take_events(), kv_transfer_params=kv_params,
                    num_cached_tokens=max(req.num_cached_tokens, 0)))

        if stopped:
            gone = set(id(r) for r in stopped)
            self.running = [r for r in self.running if id(r) not in gone]
        if stopped_preempted:
            # the victim went back to the waiting queue while its last token was still in flight: take it out
            # again, or the next schedule() would admit a finished request
            self.waiting.remove_requests(stopped_preempted)

        self._update_from_kv_xfer_finished(mro)

        result = {ci: EngineCoreOutputs(outputs=outs) for ci, outs in outputs.items()}
        if self.finished_req_ids_dict:
            for ci, ids in self.finished_req_ids_dict.items():
                if ci in result:
                    result[ci].finished_requests = ids
                else:
                    result[ci] = EngineCoreOutputs(finished_requests=ids)
            self.finished_req_ids_dict.clear()
        if self.log_stats and result:
            # stats ride on exactly one client's message
            next(iter(result.values())).scheduler_stats = self.make_stats(spec_stats)
        return result

    # ------------------------------------------------------------------
    # request lifecycle
    # ------------------------------------------------------------------
    def add_request(self, request: Request) -> None:
        self.waiting.add_request(request)
        self.requests[request.request_id] = request
        if self.log_stats:
            request.record_event(EngineCoreEventType.QUEUED)

    def finish_requests(self, request_ids: Union[str, Iterable[str]], finished_status: RequestStatus) -> None:
        assert RequestStatus.is_finished(finished_status)
        ids = [request_ids] if isinstance(request_ids, str) else list(request_ids)
        run_drop, wait_drop = [], []
        for rid in ids:
            req = self.requests.get(rid)
            if req is None:
                continue
            (run_drop if req.status == RequestStatus.RUNNING else wait_drop).append(req)
        if run_drop:
            gone = set(id(r) for r in run_drop)
            self.running = [r for r in self.running if id(r) not in gone]
        if wait_drop:
            self.waiting.remove_requests(wait_drop)
        for req in run_drop + wait_drop:
            recving = req.status == RequestStatus.WAITING_FOR_REMOTE_KVS \
                and req.request_id not in self.finished_recving_kv_req_ids
            req.status = finished_status
            self._free_request(req, keep_blocks=recving)
            if recving:
                self.abandoned_recving[req.request_id] = req

    def _free_request(self, req: Request, keep_blocks: bool = False) -> Optional[dict]:
        assert req.is_finished()
        delay, kv_params = self._connector_finished(req)
        delay = delay or keep_blocks
        self.encoder_cache_manager.free(req)
        rid = req.request_id
        self.finished_req_ids.add(rid)
        if self.finished_req_ids_dict is not None:
            self.finished_req_ids_dict[req.client_index].add(rid)
        if self.tknp is not None:
            self.tknp.free_request(req)
        if not delay:
            self._free_blocks(req)
        return kv_params

    def _free_blocks(self, req: Request) -> None:
        self.kv_cache_manager.free(req)
        self.kv_cache_manager.free_block_hashes(req)
        self.requests.pop(req.request_id, None)

    def get_num_unfinished_requests(self) -> int:
        return len(self.waiting) + len(self.running)

    def has_finished_requests(self) -> bool:
        return bool(self.finished_req_ids)

    def get_request_counts(self) -> tuple[int, int]:
        return len(self.running), len(self.waiting)

    def reset_prefix_cache(self) -> bool:
        return self.kv_cache_manager.reset_prefix_cache()

    def make_stats(self, spec_stats: Optional[SpecDecodingStats] = None) -> Optional[SchedulerStats]:
        if not self.log_stats:
            return None
        return SchedulerStats(num_running_reqs=len(self.running), num_waiting_reqs=len(self.waiting),
                              kv_cache_usage=self.kv_cache_manager.usage,
                              prefix_cache_stats=self.kv_cache_manager.make_prefix_cache_stats(),
                              spec_decoding_stats=spec_stats, step_counter=self._step)

    def shutdown(self) -> None:
        self.kv_event_publisher.shutdown()

    # ------------------------------------------------------------------
    # KV connector (disaggregated prefill / decode)
    # ------------------------------------------------------------------
    def get_kv_connector(self):
        return self.connector

    def _connector_finished(self, req: Request) -> tuple[bool, Optional[dict]]:
        if self.connector is None or (self.tknp is not None and not self.tknp.is_mine(req)):
            return False, None
        ids = self.kv_cache_manager.get_block_ids(req.request_id)
        return self.connector.request_finished(req, ids[0] if ids else [])

    def _update_waiting_for_remote_kv(self, req: Request) -> bool:
        """True once the remote KV of ``req`` has landed in its blocks."""
        rid = req.request_id
        if rid not in self.finished_recving_kv_req_ids:
            return False
        if rid in self.failed_recving_kv_req_ids:
            # the transfer failed: the blocks are there but hold nothing; compute the prompt here
            self.failed_recving_kv_req_ids.discard(rid)
            self.finished_recving_kv_req_ids.discard(rid)
            req.num_computed_tokens = 0
            return True
        block_ids = self.kv_cache_manager.get_block_ids(rid)[0]
        n_computed = len(block_ids) * self.block_size
        # always leave the last token to be computed locally
        n_computed = min(n_computed, req.num_tokens - 1) if n_computed >= req.num_tokens else n_computed
        self.kv_cache_manager.cache_blocks(req, n_computed)
        req.num_computed_tokens = n_computed
        if req.num_cached_tokens < 0:
            req.num_cached_tokens = n_computed
        self.finished_recving_kv_req_ids.discard(rid)
        return True

    def _update_from_kv_xfer_finished(self, mro: ModelRunnerOutput) -> None:
        for rid in mro.failed_recving or ():
            self.failed_recving_kv_req_ids.add(rid)
        for rid in mro.finished_recving or ():
            gone = self.abandoned_recving.pop(rid, None)
            if gone is not None:
                self.failed_recving_kv_req_ids.discard(rid)
                self._free_blocks(gone)              # aborted while receiving: now nobody writes there any more
                continue
            self.finished_recving_kv_req_ids.add(rid)
        for rid in mro.finished_sending or ():
            req = self.requests.get(rid)
            if req is not None:
                self._free_blocks(req)


def sp_logprobs(req: Request) -> bool:
    return req.sampling_params is not None and req.sampling_params.logprobs is not None
