"""In-process executors: one worker inside the engine-core process.

``UniProcExecutor`` serves a single device.  ``ExternalLauncherExecutor`` is the
torchrun mode: every rank is a full engine (own scheduler, own worker) started
by an external launcher with ``env://`` rendezvous; all ranks see the same
requests and must schedule identically (the mode token parallelism runs in).
Reference: vllm/executor/uniproc_executor.py:20-139, vllm/v1/executor/abstract.py:107-116.
"""
from __future__ import annotations

import os
from typing import Any, Callable, Optional, Union

import torch
import torch.distributed as dist

from ..logger import init_logger
from ..parallel.utils import get_distributed_init_method, get_ip, get_open_port
from ..worker.worker import Worker
from .base import Executor

logger = init_logger(__name__)


def run_method(obj: Any, method: Union[str, Callable], args: tuple, kwargs: dict) -> Any:
    if isinstance(method, str):
        return getattr(obj, method)(*args, **kwargs)
    return method(obj, *args, **kwargs)


class UniProcExecutor(Executor):
    def _init_executor(self) -> None:
        init = get_distributed_init_method(get_ip(), get_open_port())
        local_rank = 0
        if self.vllm_config.device_config.device == "cuda" and \
                ":" in str(os.environ.get("VDA_DEVICE", "")):
            local_rank = int(os.environ["VDA_DEVICE"].split(":")[1])
        self.worker = Worker(self.vllm_config, local_rank=local_rank, rank=0, distributed_init_method=init)
        self.worker.init_device()
        self.worker.load_model()

    def collective_rpc(self, method, timeout=None, args=(), kwargs=None) -> list[Any]:
        return [run_method(self.worker, method, args, kwargs or {})]

    def initialize_from_config(self, kv_cache_configs) -> None:
        self.worker.initialize_from_config(kv_cache_configs[0])
        self.worker.compile_or_warm_up_model()

    def check_health(self) -> None:
        return

    def shutdown(self) -> None:
        self.worker.shutdown()


class ExternalLauncherExecutor(UniProcExecutor):
    def _init_executor(self) -> None:
        pc = self.parallel_config
        rank = int(os.environ["RANK"])
        local_rank = int(os.environ.get("LOCAL_RANK", rank))
        world = int(os.environ.get("WORLD_SIZE", 1))
        if world != pc.world_size:
            raise ValueError(f"launcher started {world} ranks but the parallel config needs {pc.world_size} "
                             f"(PP x TP x TKNP)")
        self.worker = Worker(self.vllm_config, local_rank=local_rank, rank=rank, distributed_init_method="env://",
                             is_driver_worker=True)
        self.worker.init_device()
        self.worker.load_model()

    def determine_available_memory(self) -> list[int]:
        """All ranks must agree on the number of KV blocks (identical schedulers), so take
        the minimum -- except under token parallelism, where every rank has its own pool
        and weight-less ranks keep their larger one."""
        mem = super().determine_available_memory()
        if self.parallel_config.token_parallel_size > 1:
            return mem
        from ..parallel.state import get_world_group
        t = torch.tensor([mem[0]], dtype=torch.int64)
        g = get_world_group()
        if g.world_size > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=g.cpu_group)
        return [int(t.item())]
