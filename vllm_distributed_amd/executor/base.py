"""Executor interface: the engine core talks to "the workers" only through this
(reference: vllm/v1/executor/abstract.py:24-116, vllm/executor/executor_base.py)."""
from __future__ import annotations

from concurrent.futures import Future
from typing import Any, Callable, Optional, Union

import torch.distributed as dist

from ..config import VllmConfig
from ..logger import init_logger
from ..sched.kv_cache_interface import KVCacheConfig, KVCacheSpec
from ..sched.output import ModelRunnerOutput, SchedulerOutput

logger = init_logger(__name__)
FailureCallback = Callable[[], None]


class Executor:
    uses_ray = False

    @staticmethod
    def get_class(vllm_config: VllmConfig) -> type["Executor"]:
        backend = vllm_config.parallel_config.distributed_executor_backend
        if isinstance(backend, type):
            return backend
        if backend == "mp":
            from .multiproc import MultiprocExecutor
            return MultiprocExecutor
        if backend == "uni":
            from .uniproc import UniProcExecutor
            return UniProcExecutor
        if backend == "external_launcher":
            from .uniproc import ExternalLauncherExecutor
            return ExternalLauncherExecutor
        if backend == "ray":
            from .ray_executor import RayDistributedExecutor
            return RayDistributedExecutor
        raise ValueError(f"unknown executor backend {backend!r}")

    def __init__(self, vllm_config: VllmConfig):
        self.vllm_config = vllm_config
        self.parallel_config = vllm_config.parallel_config
        self.is_failed = False
        self._failure_callback: Optional[FailureCallback] = None
        self._init_executor()

    def _init_executor(self) -> None:
        raise NotImplementedError

    def collective_rpc(self, method: Union[str, Callable], timeout: Optional[float] = None, args: tuple = (),
                       kwargs: Optional[dict] = None) -> list[Any]:
        raise NotImplementedError

    def register_failure_callback(self, cb: FailureCallback) -> None:
        if self.is_failed:
            cb()
        else:
            self._failure_callback = cb

    # ---- KV cache handshake ---------------------------------------------------
    def get_kv_cache_specs(self) -> list[dict[str, KVCacheSpec]]:
        return self.collective_rpc("get_kv_cache_spec")

    def determine_available_memory(self) -> list[int]:
        return self.collective_rpc("determine_available_memory")

    def initialize_from_config(self, kv_cache_configs: list[KVCacheConfig]) -> None:
        self.collective_rpc("initialize_from_config", args=(kv_cache_configs,))
        self.collective_rpc("compile_or_warm_up_model")

    def execute_model(self, scheduler_output: SchedulerOutput) -> Union[ModelRunnerOutput, Future]:
        return self.collective_rpc("execute_model", args=(scheduler_output,))[0]

    @property
    def max_concurrent_batches(self) -> int:
        return 1

    def profile(self, is_start: bool = True) -> None:
        self.collective_rpc("profile", args=(is_start,))

    def sleep(self, level: int = 1) -> None:
        self.collective_rpc("sleep", args=(level,))

    def wake_up(self, tags: Optional[list[str]] = None) -> None:
        self.collective_rpc("wake_up", args=(tags,))

    def is_sleeping(self) -> bool:
        return all(self.collective_rpc("is_sleeping"))

    def save_sharded_state(self, path: str, pattern: Optional[str] = None, max_size: Optional[int] = None) -> None:
        self.collective_rpc("save_sharded_state", kwargs=dict(path=path, pattern=pattern, max_size=max_size))

    def add_lora(self, r) -> bool: return all(self.collective_rpc("add_lora", args=(r,)))
    def remove_lora(self, i: int) -> bool: return all(self.collective_rpc("remove_lora", args=(i,)))
    def pin_lora(self, i: int) -> bool: return all(self.collective_rpc("pin_lora", args=(i,)))
    def list_loras(self) -> set[int]: return self.collective_rpc("list_loras")[0]

    def check_health(self) -> None:
        if self.is_failed:
            raise RuntimeError("executor failed")
        self.collective_rpc("check_health", timeout=10)

    def shutdown(self) -> None:
        pass
