"""How the front end reaches the engine core.

  InprocClient   engine core lives in the caller's process (offline LLM, torchrun mode)
  SyncMPClient   engine core in a child process, blocking calls (LLM with multiprocessing)
  AsyncMPClient  same, asyncio (AsyncLLM / API server)
  DPAsyncMPClient / DPLBAsyncMPClient   several engine cores (data parallel), the
                 latter picks the least loaded one from coordinator statistics

The process boundary is the framed-msgpack socket transport of engine/transport.py
(the reference uses ZMQ ROUTER/DEALER + PUSH/PULL with msgspec structs:
vllm/v1/engine/core_client.py:44-1046).
"""
from __future__ import annotations

import asyncio
import queue
import threading
import time
import weakref
from typing import Any, Callable, Optional, Union

from ..config import VllmConfig
from ..logger import init_logger
from ..request import EngineCoreOutputs, EngineCoreRequest, EngineCoreRequestType, UtilityOutput

logger = init_logger(__name__)


class EngineDeadError(RuntimeError):
    def __init__(self, msg: str = "the engine core died; see the log of the engine process for the root cause"):
        super().__init__(msg)


class EngineCoreClient:
    @staticmethod
    def make_client(multiprocess_mode: bool, asyncio_mode: bool, vllm_config: VllmConfig, executor_class: type,
                    log_stats: bool, attach: Optional[dict] = None) -> "EngineCoreClient":
        if asyncio_mode and not multiprocess_mode:
            raise NotImplementedError("asyncio front ends need the multiprocess engine core")
        if not multiprocess_mode:
            return InprocClient(vllm_config, executor_class, log_stats)
        from .mp_client import AsyncMPClient, DPAsyncMPClient, DPLBAsyncMPClient, SyncMPClient
        if not asyncio_mode:
            return SyncMPClient(vllm_config, executor_class, log_stats)
        pc = vllm_config.parallel_config
        if pc.data_parallel_size > 1:
            if pc.data_parallel_external_lb:
                return DPAsyncMPClient(vllm_config, executor_class, log_stats, attach=attach)
            return DPLBAsyncMPClient(vllm_config, executor_class, log_stats, attach=attach)
        return AsyncMPClient(vllm_config, executor_class, log_stats, attach=attach)

    # sync API
    def get_output(self) -> EngineCoreOutputs: raise NotImplementedError
    def add_request(self, request: EngineCoreRequest) -> None: raise NotImplementedError
    def abort_requests(self, request_ids: list[str]) -> None: raise NotImplementedError
    def profile(self, is_start: bool = True) -> None: raise NotImplementedError
    def reset_prefix_cache(self) -> bool: raise NotImplementedError
    def reset_mm_cache(self) -> None: raise NotImplementedError
    def sleep(self, level: int = 1) -> None: raise NotImplementedError
    def wake_up(self, tags=None) -> None: raise NotImplementedError
    def is_sleeping(self) -> bool: raise NotImplementedError
    def execute_dummy_batch(self) -> None: raise NotImplementedError
    def add_lora(self, r) -> bool: raise NotImplementedError
    def remove_lora(self, i: int) -> bool: raise NotImplementedError
    def list_loras(self) -> set[int]: raise NotImplementedError
    def pin_lora(self, i: int) -> bool: raise NotImplementedError
    def save_sharded_state(self, path, pattern=None, max_size=None) -> None: raise NotImplementedError
    def collective_rpc(self, method, timeout=None, args=(), kwargs=None) -> list: raise NotImplementedError
    def shutdown(self) -> None: raise NotImplementedError
    # async API
    async def get_output_async(self) -> EngineCoreOutputs: raise NotImplementedError
    async def add_request_async(self, request: EngineCoreRequest) -> None: raise NotImplementedError
    async def abort_requests_async(self, request_ids: list[str]) -> None: raise NotImplementedError
    async def call_utility_async(self, method: str, *args) -> Any: raise NotImplementedError


class InprocClient(EngineCoreClient):
    def __init__(self, vllm_config: VllmConfig, executor_class: type, log_stats: bool):
        from .core import EngineCore
        self.engine_core = EngineCore(vllm_config, executor_class, log_stats)

    def get_output(self) -> EngineCoreOutputs:
        outs, _ = self.engine_core.step_fn()
        while outs is None:          # pipeline parallel: keep stepping until a batch completes
            outs, _ = self.engine_core.step_fn()
        return outs.get(0) or EngineCoreOutputs()

    def add_request(self, request): self.engine_core.add_request(request)

    def abort_requests(self, request_ids):
        if request_ids:
            self.engine_core.abort_requests(request_ids)

    def shutdown(self): self.engine_core.shutdown()
    def profile(self, is_start=True): self.engine_core.profile(is_start)
    def reset_prefix_cache(self): return self.engine_core.reset_prefix_cache()
    def reset_mm_cache(self): self.engine_core.reset_mm_cache()
    def sleep(self, level=1): self.engine_core.sleep(level)
    def wake_up(self, tags=None): self.engine_core.wake_up(tags)
    def is_sleeping(self): return self.engine_core.is_sleeping()
    def execute_dummy_batch(self): self.engine_core.collective_rpc("execute_dummy_batch")
    def add_lora(self, r): return self.engine_core.add_lora(r)
    def remove_lora(self, i): return self.engine_core.remove_lora(i)
    def list_loras(self): return self.engine_core.list_loras()
    def pin_lora(self, i): return self.engine_core.pin_lora(i)
    def save_sharded_state(self, path, pattern=None, max_size=None):
        self.engine_core.save_sharded_state(path, pattern, max_size)
    def collective_rpc(self, method, timeout=None, args=(), kwargs=None):
        return self.engine_core.collective_rpc(method, timeout, args, kwargs)
