"""asyncio front end used by the API server.

One background task pulls EngineCoreOutputs, runs the output processor in chunks
(so a huge batch does not starve the event loop) and feeds per-request
collectors that ``generate()`` coroutines await.  Cancelling a ``generate()``
aborts the request in the engine.
Reference: vllm/v1/engine/async_llm.py:46-625.
"""
from __future__ import annotations

import asyncio
import copy
import time
from typing import Any, AsyncGenerator, Optional, Union

from .. import envs
from ..config import VllmConfig
from ..logger import init_logger
from ..metrics.loggers import StatLoggerManager
from ..metrics.stats import IterationStats
from ..outputs import PoolingRequestOutput, RequestOutput
from ..request import LoRARequest
from ..sampling_params import PoolingParams, RequestOutputKind, SamplingParams
from .core_client import EngineCoreClient, EngineDeadError
from .output_processor import OutputProcessor, ParentRequest, RequestOutputCollector
from .processor import Processor, TokenizerGroup, get_tokenizer

logger = init_logger(__name__)


class AsyncLLM:
    def __init__(self, vllm_config: VllmConfig, executor_class: Optional[type] = None, log_stats: bool = True,
                 log_requests: bool = False, start_engine_loop: bool = True, stat_loggers: Optional[list] = None,
                 client_index: int = 0, attach: Optional[dict] = None):
        from ..executor.base import Executor
        self.vllm_config = vllm_config
        self.model_config = vllm_config.model_config
        self.log_stats = log_stats
        self.log_requests = log_requests
        executor_class = executor_class or Executor.get_class(vllm_config)
        self.tokenizer: Optional[TokenizerGroup] = None
        if not self.model_config.skip_tokenizer_init:
            self.tokenizer = TokenizerGroup(get_tokenizer(self.model_config.tokenizer, self.model_config.tokenizer_mode),
                                            lora_enabled=vllm_config.lora_config is not None)
        self.processor = Processor(vllm_config, self.tokenizer)
        self.output_processor = OutputProcessor(self.tokenizer, log_stats)
        self.client_index = client_index
        self.engine_core = EngineCoreClient.make_client(True, True, vllm_config, executor_class, log_stats,
                                                        attach=attach)
        self.stat_logger = StatLoggerManager(vllm_config, custom=stat_loggers) if log_stats else None
        self.output_handler: Optional[asyncio.Task] = None
        self._errored_with: Optional[BaseException] = None
        from ..tracing import init_tracer
        ep = vllm_config.observability_config.otlp_traces_endpoint
        self.tracer = init_tracer("vda.engine", ep) if ep else None

    @classmethod
    def from_vllm_config(cls, vllm_config: VllmConfig, **kw) -> "AsyncLLM":
        return cls(vllm_config, **kw)

    @classmethod
    def from_engine_args(cls, engine_args, start_engine_loop: bool = True, usage_context=None, stat_loggers=None,
                         client_index: int = 0, attach: Optional[dict] = None):
        cfg = engine_args.create_engine_config(usage_context)
        return cls(cfg, log_stats=not engine_args.disable_log_stats,
                   log_requests=not getattr(engine_args, "disable_log_requests", True), stat_loggers=stat_loggers,
                   client_index=client_index, attach=attach)

    def shutdown(self) -> None:
        if self.output_handler is not None:
            self.output_handler.cancel()
            self.output_handler = None
        ec = getattr(self, "engine_core", None)
        if ec is not None:
            ec.shutdown()
            self.engine_core = None

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass

    # ------------------------------------------------------------------
    async def add_request(self, request_id: str, prompt, params: Union[SamplingParams, PoolingParams],
                          arrival_time: Optional[float] = None, lora_request: Optional[LoRARequest] = None,
                          tokenization_kwargs: Optional[dict] = None, trace_headers: Optional[dict] = None,
                          priority: int = 0, data_parallel_rank: Optional[int] = None) -> RequestOutputCollector:
        if self.errored:
            raise EngineDeadError()
        kind = params.output_kind
        queue = RequestOutputCollector(kind)
        text, req = self.processor.process_inputs(request_id, prompt, params, arrival_time, lora_request,
                                                  tokenization_kwargs, trace_headers, priority, data_parallel_rank)
        req.client_index = self.client_index
        n = params.n if isinstance(params, SamplingParams) else 1
        if n == 1:
            self.output_processor.add_request(req, text, None, 0, queue)
            await self.engine_core.add_request_async(req)
        else:
            parent = ParentRequest(request_id, params)
            for i in range(n):
                cid, cparams = parent.child(i)
                child = req if i == n - 1 else copy.copy(req)
                child.request_id = cid
                cparams._all_stop_token_ids = set(req.sampling_params._all_stop_token_ids)
                cparams._bad_words_token_ids = req.sampling_params._bad_words_token_ids
                child.sampling_params = cparams
                self.output_processor.add_request(child, text, parent, i, queue)
                await self.engine_core.add_request_async(child)
        if self.log_requests:
            logger.info("added request %s (%d prompt tokens)", request_id, len(req.prompt_token_ids))
        return queue

    async def generate(self, prompt, sampling_params: SamplingParams, request_id: str,
                       lora_request: Optional[LoRARequest] = None, trace_headers: Optional[dict] = None,
                       priority: int = 0, data_parallel_rank: Optional[int] = None
                       ) -> AsyncGenerator[RequestOutput, None]:
        try:
            self._run_output_handler()
            q = await self.add_request(request_id, prompt, sampling_params, lora_request=lora_request,
                                       trace_headers=trace_headers, priority=priority,
                                       data_parallel_rank=data_parallel_rank)
            finished = False
            t0 = time.time()
            while not finished:
                out = q.get_nowait() or await q.get()
                finished = out.finished
                yield out
            if self.tracer is not None:
                from ..tracing import record_request_span
                record_request_span(self.tracer, request_id, t0, out, trace_headers)
        except (asyncio.CancelledError, GeneratorExit):
            # task cancelled (client disconnect) or generator closed early by the caller
            await self.abort(request_id)
            if self.log_requests:
                logger.info("request %s aborted (client cancelled)", request_id)
            raise
        except EngineDeadError:
            raise
        except ValueError:
            raise
        except Exception:
            await self.abort(request_id)
            raise

    async def encode(self, prompt, pooling_params: PoolingParams, request_id: str, lora_request=None,
                     trace_headers=None, priority: int = 0) -> AsyncGenerator[PoolingRequestOutput, None]:
        try:
            self._run_output_handler()
            q = await self.add_request(request_id, prompt, pooling_params, lora_request=lora_request,
                                       trace_headers=trace_headers, priority=priority)
            finished = False
            while not finished:
                out = q.get_nowait() or await q.get()
                finished = out.finished
                yield out
        except (asyncio.CancelledError, GeneratorExit):
            await self.abort(request_id)
            raise

    def _run_output_handler(self) -> None:
        if self.output_handler is not None:
            return
        engine_core, output_processor, stat_logger = self.engine_core, self.output_processor, self.stat_logger
        log_stats = self.log_stats
        chunk = envs.VLLM_V1_OUTPUT_PROC_CHUNK_SIZE

        async def handler():
            try:
                while True:
                    outputs = await engine_core.get_output_async()
                    n = len(outputs.outputs)
                    it_stats = IterationStats() if (log_stats and n) else None
                    slices = [outputs.outputs] if n <= chunk else \
                        [outputs.outputs[i:i + chunk] for i in range(0, n, chunk)]
                    for i, sl in enumerate(slices):
                        processed = output_processor.process_outputs(sl, outputs.timestamp, it_stats)
                        assert not processed.request_outputs      # everything goes through the collectors
                        if i + 1 < len(slices):
                            await asyncio.sleep(0)
                        if processed.reqs_to_abort:
                            await engine_core.abort_requests_async(processed.reqs_to_abort)
                    if stat_logger is not None and (it_stats is not None or outputs.scheduler_stats is not None):
                        stat_logger.record(outputs.scheduler_stats, it_stats)
            except asyncio.CancelledError:
                raise
            except Exception as e:   # noqa: BLE001 - surfaces in every waiting generate()
                logger.exception("output handler failed")
                self._errored_with = e
                output_processor.propagate_error(e if isinstance(e, EngineDeadError) else EngineDeadError(str(e)))

        self.output_handler = asyncio.create_task(handler())

    async def abort(self, request_id: str) -> None:
        ids = self.output_processor.abort_requests([request_id])
        if self.engine_core is not None:
            await self.engine_core.abort_requests_async(ids)

    # ---- EngineClient protocol ---------------------------------------------------
    @property
    def is_running(self) -> bool:
        return self.output_handler is None or not self.output_handler.done()

    @property
    def is_stopped(self) -> bool:
        return self.errored

    @property
    def errored(self) -> bool:
        return self._errored_with is not None or self.engine_core is None or self.engine_core.engine_dead

    @property
    def dead_error(self) -> BaseException:
        return EngineDeadError()

    async def get_vllm_config(self): return self.vllm_config
    async def get_model_config(self): return self.model_config
    async def get_decoding_config(self): return self.vllm_config.decoding_config
    async def get_input_preprocessor(self): return self.processor

    async def get_tokenizer(self, lora_request=None):
        if self.tokenizer is None:
            raise ValueError("the engine was started with skip_tokenizer_init=True")
        return self.tokenizer.get_lora_tokenizer(lora_request)

    async def is_tracing_enabled(self) -> bool: return self.tracer is not None

    async def do_log_stats(self) -> None:
        if self.stat_logger is not None:
            self.stat_logger.log()

    async def check_health(self) -> None:
        if self.errored:
            raise self.dead_error

    async def start_profile(self): await self.engine_core.profile_async(True)
    async def stop_profile(self): await self.engine_core.profile_async(False)
    async def reset_mm_cache(self): await self.engine_core.reset_mm_cache_async()
    async def reset_prefix_cache(self, device=None): return await self.engine_core.reset_prefix_cache_async()
    async def sleep(self, level: int = 1): await self.engine_core.sleep_async(level)
    async def wake_up(self, tags=None): await self.engine_core.wake_up_async(tags)
    async def is_sleeping(self) -> bool: return await self.engine_core.is_sleeping_async()
    async def add_lora(self, r) -> bool: return await self.engine_core.add_lora_async(r)
    async def remove_lora(self, i: int) -> bool: return await self.engine_core.remove_lora_async(i)
    async def list_loras(self) -> set[int]: return await self.engine_core.list_loras_async()
    async def pin_lora(self, i: int) -> bool: return await self.engine_core.pin_lora_async(i)

    async def collective_rpc(self, method: str, timeout: Optional[float] = None, args: tuple = (),
                             kwargs: Optional[dict] = None):
        return await self.engine_core.collective_rpc_async(method, timeout, args, kwargs)


AsyncLLMEngine = AsyncLLM
