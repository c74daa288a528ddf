"""Synchronous engine: Processor -> EngineCoreClient -> OutputProcessor.
Reference: vllm/v1/engine/llm_engine.py:41-322."""
from __future__ import annotations

import time
from typing import Any, Callable, Optional, Union

from .. import envs
from ..config import VllmConfig
from ..logger import init_logger
from ..metrics.loggers import StatLoggerManager
from ..metrics.stats import IterationStats
from ..outputs import PoolingRequestOutput, RequestOutput
from ..request import LoRARequest
from ..sampling_params import PoolingParams, SamplingParams
from .core_client import EngineCoreClient
from .output_processor import OutputProcessor, ParentRequest
from .processor import Processor, TokenizerGroup, get_tokenizer

logger = init_logger(__name__)


class LLMEngine:
    def __init__(self, vllm_config: VllmConfig, executor_class: Optional[type] = None, log_stats: bool = False,
                 multiprocess_mode: bool = False, stat_loggers: Optional[list] = None):
        from ..executor.base import Executor
        self.vllm_config = vllm_config
        self.model_config = vllm_config.model_config
        self.log_stats = log_stats
        executor_class = executor_class or Executor.get_class(vllm_config)
        from ..usage import UsageContext, usage_message
        usage_message.report_usage((self.model_config.architectures or ["unknown"])[0], UsageContext.ENGINE_CONTEXT,
                                   {"dtype": str(self.model_config.dtype),
                                    "tensor_parallel_size": vllm_config.parallel_config.tensor_parallel_size,
                                    "quantization": self.model_config.quantization})
        self.tokenizer: Optional[TokenizerGroup] = None
        if not self.model_config.skip_tokenizer_init:
            self.tokenizer = TokenizerGroup(
                get_tokenizer(self.model_config.tokenizer, self.model_config.tokenizer_mode),
                lora_enabled=vllm_config.lora_config is not None)
        self.processor = Processor(vllm_config, self.tokenizer)
        self.output_processor = OutputProcessor(self.tokenizer, log_stats)
        self.engine_core = EngineCoreClient.make_client(multiprocess_mode, False, vllm_config, executor_class,
                                                        log_stats)
        self.stat_logger = StatLoggerManager(vllm_config, custom=stat_loggers) if log_stats else None
        self.should_execute_dummy_batch = False

    @classmethod
    def from_vllm_config(cls, vllm_config: VllmConfig, **kw) -> "LLMEngine":
        return cls(vllm_config, **kw)

    @classmethod
    def from_engine_args(cls, engine_args, usage_context=None, stat_loggers=None,
                         enable_multiprocessing: Optional[bool] = None) -> "LLMEngine":
        cfg = engine_args.create_engine_config(usage_context)
        mp = envs.VLLM_ENABLE_V1_MULTIPROCESSING if enable_multiprocessing is None else enable_multiprocessing
        if cfg.parallel_config.distributed_executor_backend == "external_launcher":
            mp = False
        return cls(cfg, log_stats=not engine_args.disable_log_stats, multiprocess_mode=mp, stat_loggers=stat_loggers)

    # ------------------------------------------------------------------
    def get_num_unfinished_requests(self) -> int:
        return self.output_processor.get_num_unfinished_requests()

    def has_unfinished_requests(self) -> bool:
        has = self.output_processor.has_unfinished_requests()
        dp = self.vllm_config.parallel_config.data_parallel_size
        if dp > 1 and hasattr(self.engine_core, "dp_has_unfinished"):
            return self.engine_core.dp_has_unfinished(has)
        return has

    def abort_request(self, request_ids: list[str]) -> None:
        ids = self.output_processor.abort_requests(request_ids)
        self.engine_core.abort_requests(ids)

    def add_request(self, request_id: str, prompt, params: Union[SamplingParams, PoolingParams],
                    arrival_time: Optional[float] = None, lora_request: Optional[LoRARequest] = None,
                    tokenization_kwargs: Optional[dict] = None, trace_headers: Optional[dict] = None,
                    priority: int = 0) -> None:
        text, req = self.processor.process_inputs(request_id, prompt, params, arrival_time, lora_request,
                                                  tokenization_kwargs, trace_headers, priority)
        n = params.n if isinstance(params, SamplingParams) else 1
        if n == 1:
            self.output_processor.add_request(req, text)
            self.engine_core.add_request(req)
            return
        parent = ParentRequest(request_id, params)
        import copy
        for i in range(n):
            cid, cparams = parent.child(i)
            child = req if i == n - 1 else copy.copy(req)
            child.request_id = cid
            cparams._all_stop_token_ids = set(req.sampling_params._all_stop_token_ids)
            cparams._bad_words_token_ids = req.sampling_params._bad_words_token_ids
            child.sampling_params = cparams
            self.output_processor.add_request(child, text, parent, i)
            self.engine_core.add_request(child)

    def step(self) -> list[Union[RequestOutput, PoolingRequestOutput]]:
        if self.should_execute_dummy_batch:
            self.should_execute_dummy_batch = False
            self.engine_core.execute_dummy_batch()
            return []
        outputs = self.engine_core.get_output()
        it_stats = IterationStats() if self.log_stats else None
        processed = self.output_processor.process_outputs(outputs.outputs, outputs.timestamp, it_stats)
        if processed.reqs_to_abort:
            self.engine_core.abort_requests(processed.reqs_to_abort)
        if self.stat_logger is not None:
            self.stat_logger.record(outputs.scheduler_stats, it_stats)
        return processed.request_outputs

    # ---- pass-throughs -------------------------------------------------------------
    def get_vllm_config(self): return self.vllm_config
    def get_model_config(self): return self.model_config
    def get_tokenizer_group(self) -> TokenizerGroup:
        if self.tokenizer is None:
            raise ValueError("the engine was started with skip_tokenizer_init=True")
        return self.tokenizer
    def get_tokenizer(self, lora_request=None): return self.get_tokenizer_group().get_lora_tokenizer(lora_request)
    def start_profile(self): self.engine_core.profile(True)
    def stop_profile(self): self.engine_core.profile(False)
    def reset_prefix_cache(self, device=None): return self.engine_core.reset_prefix_cache()
    def reset_mm_cache(self): self.engine_core.reset_mm_cache()
    def sleep(self, level: int = 1): self.engine_core.sleep(level)
    def wake_up(self, tags: Optional[list[str]] = None): self.engine_core.wake_up(tags)
    def is_sleeping(self) -> bool: return self.engine_core.is_sleeping()
    def add_lora(self, r) -> bool: return self.engine_core.add_lora(r)
    def remove_lora(self, i: int) -> bool: return self.engine_core.remove_lora(i)
    def list_loras(self) -> set[int]: return self.engine_core.list_loras()
    def pin_lora(self, i: int) -> bool: return self.engine_core.pin_lora(i)

    def get_metrics(self) -> list:
        from ..metrics.reader import get_metrics_snapshot
        return get_metrics_snapshot()

    def collective_rpc(self, method: Union[str, Callable], timeout: Optional[float] = None, args: tuple = (),
                       kwargs: Optional[dict] = None) -> list[Any]:
        return self.engine_core.collective_rpc(method, timeout, args, kwargs)

    def do_log_stats(self) -> None:
        if self.stat_logger is not None:
            self.stat_logger.log()

    def shutdown(self) -> None:
        ec = getattr(self, "engine_core", None)
        if ec is not None:
            ec.shutdown()
            self.engine_core = None

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass
