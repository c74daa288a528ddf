"""EngineCore: scheduler + executor + the step loop.

``EngineCore`` is the in-process object; ``EngineCoreProc`` (engine/core_proc.py)
wraps it in a busy loop behind the socket transport for the multi-process
deployment.  Reference: vllm/v1/engine/core.py (EngineCore :58-359: KV sizing
handshake :133-174, step :223-240, batch-queue step for pipeline parallelism
:242-288, utility methods :290-359).
"""
from __future__ import annotations

import queue
import time
from collections import deque
from concurrent.futures import Future
from typing import Any, Callable, Optional, Union

from ..config import VllmConfig
from ..logger import init_logger
from ..request import EngineCoreOutputs, EngineCoreRequest, Request, RequestStatus
from ..sched.kv_cache_utils import get_kv_cache_config, unify_kv_cache_configs
from ..sched.output import ModelRunnerOutput, SchedulerOutput
from ..sched.scheduler import Scheduler, SchedulerInterface
from ..sched.token_parallel import TokenParallelScheduler
from ..utils import resolve_obj_by_qualname

logger = init_logger(__name__)


class EngineCore:
    def __init__(self, vllm_config: VllmConfig, executor_class: type, log_stats: bool,
                 executor_fail_callback: Optional[Callable] = None):
        from ..plugins import load_general_plugins
        load_general_plugins()
        self.vllm_config = vllm_config
        self.log_stats = log_stats
        logger.info("initialising engine core: %s", vllm_config)
        self.model_executor = executor_class(vllm_config)
        if executor_fail_callback is not None:
            self.model_executor.register_failure_callback(executor_fail_callback)
        num_blocks, kv_cache_config = self._initialize_kv_caches(vllm_config)
        vllm_config.cache_config.num_gpu_blocks = num_blocks
        vllm_config.cache_config.num_cpu_blocks = 0
        from ..structured_output import StructuredOutputManager
        self.structured_output_manager = StructuredOutputManager(vllm_config)
        tknp = self._make_tknp(vllm_config, num_blocks)
        sched_cls = resolve_obj_by_qualname(vllm_config.scheduler_config.scheduler_cls) \
            if isinstance(vllm_config.scheduler_config.scheduler_cls, str) else vllm_config.scheduler_config.scheduler_cls
        self.scheduler: SchedulerInterface = sched_cls(
            vllm_config=vllm_config, kv_cache_config=kv_cache_config,
            structured_output_manager=self.structured_output_manager,
            include_finished_set=vllm_config.parallel_config.data_parallel_size > 1,
            log_stats=log_stats, tknp=tknp)
        # pipeline parallelism keeps pp_size batches in flight
        self.batch_queue_size = self.model_executor.max_concurrent_batches
        self.batch_queue: Optional[deque] = None
        if self.batch_queue_size > 1:
            logger.info("batch queue enabled with %d slots", self.batch_queue_size)
            self.batch_queue = deque()
        self.step_fn = self.step if self.batch_queue is None else self.step_with_batch_queue
        self._inflight = None            # run-ahead scheduling: (scheduler output, result handle) of the step in flight
        self._ready: deque = deque()     # outputs of steps collected ahead of their turn
        if vllm_config.scheduler_config.async_scheduling and self.batch_queue is None:
            logger.info("run-ahead scheduling: step N + 1 is scheduled and launched while step N runs")
            self.step_fn = self.step_run_ahead

    def _make_tknp(self, cfg: VllmConfig, num_blocks: int) -> Optional[TokenParallelScheduler]:
        n = cfg.parallel_config.token_parallel_size
        if n <= 1:
            return None
        from ..parallel.state import get_tknp_group
        g = get_tknp_group()
        # every rank's pool size, exchanged ONCE; afterwards placement needs no collective
        caps = g.all_gather_object(num_blocks)
        policy = cfg.additional_config.get("tknp_policy", "balanced")
        return TokenParallelScheduler(n, g.rank_in_group, cfg.cache_config.block_size, caps, policy)

    def _initialize_kv_caches(self, vllm_config: VllmConfig):
        t0 = time.time()
        specs = self.model_executor.get_kv_cache_specs()
        avail = self.model_executor.determine_available_memory()
        assert len(specs) == len(avail)
        configs = [get_kv_cache_config(vllm_config, dict(s), a) for s, a in zip(specs, avail)]
        unify_kv_cache_configs(configs)
        assert all(c.num_blocks == configs[0].num_blocks for c in configs)
        self.model_executor.initialize_from_config(configs)
        logger.info("KV cache + warm-up took %.1f s", time.time() - t0)
        return configs[0].num_blocks, configs[0]

    # ---- requests -------------------------------------------------------------
    def add_request(self, request: EngineCoreRequest) -> None:
        so_req = None
        sp = request.sampling_params
        if sp is not None and sp.guided_decoding is not None:
            so_req = self.structured_output_manager.make_request(request)
        req = Request.from_engine_core_request(request, so_req)
        if req.kv_transfer_params is not None and self.scheduler.get_kv_connector() is None:
            logger.warning_once("request carries kv_transfer_params but no KV connector is configured")
        self.scheduler.add_request(req)

    def abort_requests(self, request_ids: list[str]) -> None:
        self.scheduler.finish_requests(request_ids, RequestStatus.FINISHED_ABORTED)

    # ---- stepping ---------------------------------------------------------------
    def execute_model(self, so: SchedulerOutput) -> ModelRunnerOutput:
        try:
            return self.model_executor.execute_model(so)
        except Exception:
            from ..utils.dump_input import dump_engine_exception
            dump_engine_exception(self.vllm_config, so, self.scheduler.make_stats())
            raise

    def step(self) -> tuple[dict[int, EngineCoreOutputs], bool]:
        if not self.scheduler.has_requests():
            return {}, False
        so = self.scheduler.schedule()
        mro = self.execute_model(so)
        if isinstance(mro, Future):
            mro = mro.result()
        outs = self.scheduler.update_from_output(so, mro)
        return outs, so.total_num_scheduled_tokens > 0

    def _execute_maybe_deferred(self, so: SchedulerOutput):
        """Execute ``so``; the result is a ModelRunnerOutput, or a handle whose ``get()`` yields one later."""
        so.defer_output = self.scheduler.can_run_ahead(so)
        h = self.execute_model(so)
        return h.result() if isinstance(h, Future) else h

    def _settle(self, so: SchedulerOutput, h) -> Optional[dict[int, EngineCoreOutputs]]:
        """Keep a deferred result in flight (returns None) or fold a finished one into the scheduler."""
        if isinstance(h, ModelRunnerOutput):
            return self.scheduler.update_from_output(so, h)
        self._inflight = (so, h)
        return None

    def step_run_ahead(self) -> tuple[dict[int, EngineCoreOutputs], bool]:
        """One engine step with the host ahead of the GPU by one step: while step N runs, step N + 1 is scheduled
        (every request of N that will sample a token holds a stand-in for it), its inputs are prepared, the one
        unknown input token per request is read from N's result on the device, and N + 1 is launched; only then
        are N's ids collected and its requests updated.  A request that N finished (stop token, stop string)
        has then already been given one more token in N + 1, which is dropped.  Steps whose next inputs depend
        on more than the sampled id (grammars, penalties, ...) are collected before anything else is scheduled.
        Reference: the async scheduler of vllm/v1/core/sched/async_scheduler.py and EngineCore.step_with_batch_queue
        (vllm/v1/engine/core.py:245-290); here the ids never leave the device between two steps."""
        sch = self.scheduler
        if self._ready:
            return self._ready.popleft(), True
        if self._inflight is None:
            if not sch.has_requests():
                return {}, False
            so = sch.schedule()
            outs = self._settle(so, self._execute_maybe_deferred(so))
            return (outs if outs is not None else {}), so.total_num_scheduled_tokens > 0
        so, h = self._inflight
        self._inflight = None
        so2 = sch.schedule()
        if sch.last_schedule_preempted:
            # a preempted request's blocks and batch row go away: finish N on the host before N + 1 is prepared
            outs = sch.update_from_output(so, h.get())
            h2 = self._execute_maybe_deferred(so2)
        else:
            h2 = self._execute_maybe_deferred(so2)
            outs = sch.update_from_output(so, h.get())       # N before N + 1, whichever way N + 1 ran
        outs2 = self._settle(so2, h2)
        if outs2:
            self._ready.append(outs2)
        return outs, True

    def step_with_batch_queue(self) -> tuple[Optional[dict[int, EngineCoreOutputs]], bool]:
        """Keep the pipeline full: schedule a new batch whenever a slot is free, and only
        block on the oldest batch when nothing new can be scheduled."""
        bq = self.batch_queue
        scheduled = False
        if len(bq) < self.batch_queue_size and self.scheduler.has_requests():
            so = self.scheduler.schedule()
            if so.total_num_scheduled_tokens > 0:
                fut = self.model_executor.execute_model(so)
                bq.appendleft((fut, so))
                scheduled = True
            elif so.finished_req_ids or so.kv_connector_metadata is not None:
                # nothing to run, but workers must still learn about finished requests
                fut = self.model_executor.execute_model(so)
                bq.appendleft((fut, so))
        if scheduled and len(bq) < self.batch_queue_size:
            # a slot is still free: come back and try to fill it before blocking
            return None, True
        if not bq:
            return None, False
        fut, so = bq.pop()
        mro = fut.result() if isinstance(fut, Future) else fut
        outs = self.scheduler.update_from_output(so, mro)
        return outs, scheduled or so.total_num_scheduled_tokens > 0

    # ---- utilities -----------------------------------------------------------------
    def shutdown(self) -> None:
        self.structured_output_manager.shutdown()
        if self.model_executor:
            self.model_executor.shutdown()
        if self.scheduler:
            self.scheduler.shutdown()

    def profile(self, is_start: bool = True) -> None:
        self.model_executor.profile(is_start)

    def reset_prefix_cache(self) -> bool:
        return self.scheduler.reset_prefix_cache()

    def reset_mm_cache(self) -> None:
        pass

    def sleep(self, level: int = 1) -> None:
        """Release device memory.  The KV cache is dropped, not saved, so nothing may be running (its
        blocks would be gone at wake-up) and cached prefixes are forgotten first."""
        if self.scheduler.has_unfinished_requests():
            raise RuntimeError("cannot sleep while requests are running: their KV cache would be dropped")
        if not self.scheduler.reset_prefix_cache():
            raise RuntimeError("cannot sleep: the prefix cache still holds referenced blocks")
        self.model_executor.sleep(level)

    def wake_up(self, tags: Optional[list[str]] = None) -> None:
        self.model_executor.wake_up(tags)

    def is_sleeping(self) -> bool:
        return self.model_executor.is_sleeping()

    def add_lora(self, r) -> bool: return self.model_executor.add_lora(r)
    def remove_lora(self, i: int) -> bool: return self.model_executor.remove_lora(i)
    def list_loras(self) -> set[int]: return self.model_executor.list_loras()
    def pin_lora(self, i: int) -> bool: return self.model_executor.pin_lora(i)

    def save_sharded_state(self, path: str, pattern: Optional[str] = None, max_size: Optional[int] = None) -> None:
        self.model_executor.save_sharded_state(path, pattern, max_size)

    def collective_rpc(self, method: Union[str, Callable], timeout: Optional[float] = None, args: tuple = (),
                       kwargs: Optional[dict] = None) -> list:
        return self.model_executor.collective_rpc(method, timeout, args, kwargs)

    def execute_dummy_batch(self) -> None:
        self.model_executor.collective_rpc("execute_dummy_batch")

    def get_request_counts(self) -> tuple[int, int]:
        return self.scheduler.get_request_counts()

    def is_healthy(self) -> bool:
        self.model_executor.check_health()
        return True
