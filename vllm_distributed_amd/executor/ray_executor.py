"""Ray executor: one Ray actor per GPU (possibly on several nodes), driven by RPC fan-out.

Each actor owns a ``Worker``; the driver calls ``collective_rpc`` on all of them and, for
``execute_model``, reads the answer of the last pipeline stage's first tensor-parallel rank.
Actors are placed with a STRICT_PACK-per-node placement group (``PACK`` across nodes), and rank 0's
node provides the rendezvous address for ``torch.distributed``.  Ray is optional: selecting this
executor without it installed is a configuration error with a clear message.
Reference: vllm/executor/ray_distributed_executor.py:60-678, vllm/executor/ray_utils.py:35-399.
"""
from __future__ import annotations

import os
from typing import Any, Optional

from ..logger import init_logger
from ..parallel.utils import get_distributed_init_method, get_open_port
from .base import Executor
from .uniproc import run_method

logger = init_logger(__name__)


def ray_is_available() -> bool:
    try:
        import ray  # noqa: F401
        return True
    except ImportError:
        return False


class RayWorkerWrapper:
    """Body of the actor: builds the ``Worker`` lazily so that device visibility is fixed first."""

    def __init__(self):
        self.worker = None

    def node_ip(self) -> str:
        import ray
        return ray.util.get_node_ip_address()

    def gpu_ids(self) -> list:
        import ray
        return ray.get_gpu_ids()

    def init_worker(self, vllm_config, rank: int, local_rank: int, init_method: str, env: dict) -> None:
        os.environ.update(env)
        from ..worker.worker import Worker
        self.worker = Worker(vllm_config, local_rank=local_rank, rank=rank, distributed_init_method=init_method,
                             is_driver_worker=(rank % vllm_config.parallel_config.tensor_parallel_size == 0))

    def execute_method(self, method, args: tuple, kwargs: dict) -> Any:
        return run_method(self.worker, method, args, kwargs)


class RayDistributedExecutor(Executor):
    uses_ray = True

    def _init_executor(self) -> None:
        if not ray_is_available():
            raise RuntimeError("distributed_executor_backend='ray' needs the `ray` package, which is not installed; "
                               "use 'mp' on one node or 'external_launcher' (torchrun) across nodes")
        import ray
        from ray.util.placement_group import placement_group
        from ray.util.scheduling_strategies import PlacementGroupSchedulingStrategy
        if not ray.is_initialized():
            ray.init(address=os.environ.get("RAY_ADDRESS"), ignore_reinit_error=True)
        pc = self.parallel_config
        world = pc.world_size
        pg = placement_group([{"GPU": 1, "CPU": 1}] * world, strategy="PACK")
        ray.get(pg.ready(), timeout=float(os.environ.get("VLLM_RAY_PG_TIMEOUT", "600")))
        actor_cls = ray.remote(num_cpus=1, num_gpus=1)(RayWorkerWrapper)
        self.workers = [actor_cls.options(scheduling_strategy=PlacementGroupSchedulingStrategy(
            placement_group=pg, placement_group_bundle_index=i)).remote() for i in range(world)]
        self.placement_group = pg
        ips = ray.get([w.node_ip.remote() for w in self.workers])
        # ranks follow the bundle order, which PACK keeps node-contiguous: TP groups stay inside a node
        init = get_distributed_init_method(ips[0], get_open_port())
        local_of: dict[str, int] = {}
        calls = []
        for rank, (w, ip) in enumerate(zip(self.workers, ips)):
            local_rank = local_of.get(ip, 0)
            local_of[ip] = local_rank + 1
            # Ray pins the visible device per actor, so inside the actor the GPU is always device 0
            env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "VLLM_HOST_IP": ip}
            calls.append(w.init_worker.remote(self.vllm_config, rank, 0, init, env))
        ray.get(calls)
        self._run_workers("init_device")
        self._run_workers("load_model")
        self.output_rank = world - pc.tensor_parallel_size      # first TP rank of the last pipeline stage

    def _run_workers(self, method, args: tuple = (), kwargs: Optional[dict] = None, timeout=None) -> list[Any]:
        import ray
        refs = [w.execute_method.remote(method, args, kwargs or {}) for w in self.workers]
        try:
            return ray.get(refs, timeout=timeout)
        except Exception:
            self.is_failed = True
            if self._failure_callback is not None:
                cb, self._failure_callback = self._failure_callback, None
                cb()
            raise

    def collective_rpc(self, method, timeout=None, args=(), kwargs=None) -> list[Any]:
        return self._run_workers(method, args, kwargs, timeout)

    def execute_model(self, scheduler_output):
        return self._run_workers("execute_model", (scheduler_output,))[self.output_rank]

    def check_health(self) -> None:
        self._run_workers("check_health", timeout=10)

    def shutdown(self) -> None:
        if not getattr(self, "workers", None):
            return
        import ray
        for w in self.workers:
            try:
                ray.kill(w)
            except Exception:
                pass
        self.workers = []
        try:
            ray.util.remove_placement_group(self.placement_group)
        except Exception:
            pass
