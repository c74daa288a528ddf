"""One worker = one device = one process (or the engine's own process with the
uniproc executor).  Owns device set-up, the distributed environment, memory
profiling for KV sizing, warm-up / hipGraph capture, and the per-step call.

Reference: vllm/v1/worker/gpu_worker.py (init_device :129-180,
determine_available_memory :199-248, compile_or_warm_up_model :267-307,
execute_model :312-352, profile :354-362, sleep/wake :86-122).
"""
from __future__ import annotations

import gc
import os
from typing import Any, Optional

import torch

from .. import envs
from ..config import VllmConfig
from ..logger import init_logger
from ..models.utils import IntermediateTensors
from ..parallel import state as pstate
from ..platforms import current_platform
from ..sched.kv_cache_interface import KVCacheConfig, KVCacheSpec
from ..sched.output import ModelRunnerOutput, SchedulerOutput
from .model_runner import ModelRunner

logger = init_logger(__name__)


class Worker:
    def __init__(self, vllm_config: VllmConfig, local_rank: int = 0, rank: int = 0,
                 distributed_init_method: Optional[str] = None, is_driver_worker: bool = True):
        self.vllm_config = vllm_config
        self.local_rank, self.rank = local_rank, rank
        self.distributed_init_method = distributed_init_method or "env://"
        self.is_driver_worker = is_driver_worker
        self.parallel_config = vllm_config.parallel_config
        self.parallel_config.rank = rank
        self.device: Optional[torch.device] = None
        self.model_runner: Optional[ModelRunner] = None
        self.profiler = None
        self._sleep_saved: dict[str, torch.Tensor] = {}
        self._sleeping = 0
        pdir = envs.VLLM_TORCH_PROFILER_DIR
        if pdir:
            acts = [torch.profiler.ProfilerActivity.CPU]
            if torch.cuda.is_available():
                acts.append(torch.profiler.ProfilerActivity.CUDA)
            self.profiler = torch.profiler.profile(
                activities=acts, with_stack=True,
                on_trace_ready=torch.profiler.tensorboard_trace_handler(pdir, use_gzip=True))

    # ------------------------------------------------------------------
    def init_device(self) -> None:
        from ..plugins import load_general_plugins
        load_general_plugins()               # worker processes need the same registrations as the front end
        dc = self.vllm_config.device_config
        pc = self.parallel_config
        if dc.device == "cuda":
            rehearse = envs.VDA_REHEARSE_ON_ONE_GPU
            self.device = torch.device("cuda:0" if rehearse else f"cuda:{self.local_rank}")
            torch.cuda.set_device(self.device)
            gc.collect()
            torch.cuda.empty_cache()
            free, total = torch.cuda.mem_get_info(self.device)
            self.init_free, self.total_memory = free, total
            want = total * self.vllm_config.cache_config.gpu_memory_utilization
            if free < want:
                raise ValueError(f"device has {free / 2**30:.1f} GiB free of {total / 2**30:.1f} GiB but "
                                 f"gpu_memory_utilization asks for {want / 2**30:.1f} GiB")
            backend = "gloo+hip" if rehearse else "nccl"
            self._load_gemm_selections()
        else:
            self.device = torch.device("cpu")
            self.init_free = self.total_memory = current_platform.total_memory()
            backend = "gloo"
        world, rank, init, dp = pc.world_size, self.rank, self.distributed_init_method, 1
        from ..engine.dp import needs_lockstep
        if needs_lockstep(self.vllm_config):
            # expert parallelism spans the data-parallel replicas: one world over all of them
            dp = pc.data_parallel_size
            world, rank = dp * pc.world_size, pc.data_parallel_rank * pc.world_size + self.rank
            init = f"tcp://{pc.data_parallel_master_ip}:{pc.data_parallel_master_port}"
        pstate.init_distributed_environment(world, rank, init, self.local_rank, backend)
        pstate.ensure_model_parallel_initialized(pc.tensor_parallel_size, pc.pipeline_parallel_size, dp,
                                                 pc.token_parallel_size)
        current_platform.seed_everything(self.vllm_config.model_config.seed)
        self.model_runner = ModelRunner(self.vllm_config, self.device)
        if self.vllm_config.kv_transfer_config is not None:
            from ..parallel.kv_transfer.factory import KVConnectorFactory, KVConnectorRole
            self.model_runner.kv_connector = KVConnectorFactory.create(self.vllm_config, KVConnectorRole.WORKER)

    @staticmethod
    def _load_gemm_selections() -> None:
        """Library-GEMM algorithm table for decode-sized batches, produced offline on this GPU generation
        (benchmarks_dev/tune_gemms.py).  Lookup only: nothing is tuned at run time, unknown shapes use
        the library heuristic."""
        path = envs.VDA_GEMM_SELECTIONS
        if path is None:
            path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ops", "tuned",
                                "gfx950_gemm.csv")
        if not path or path == "off" or not os.path.isfile(path):
            return
        try:
            import torch.cuda.tunable as tn
            tn.enable(True)
            tn.tuning_enable(False)
            tn.record_untuned_enable(False)
            ok = tn.read_file(path)
            logger.info("GEMM selection table %s %s", path, "loaded" if ok else "rejected (library versions differ)")
            if not ok:
                tn.enable(False)
        except Exception as e:   # noqa: BLE001 - an optimisation only
            logger.warning("GEMM selection table not used: %s", e)

    def load_model(self) -> None:
        self.model_runner.load_model()
        pp = pstate.get_pp_group()
        if pp.world_size > 1 and self.device.type == "cuda":
            # activations of the largest step, one tensor-parallel slice of them per pipe
            mr = self.model_runner
            tp = pstate.get_tp_group().world_size
            per_row = mr.hidden_size * torch.empty((), dtype=mr.dtype).element_size()
            pp.enable_pipe(-(-mr.max_num_tokens * per_row // tp))

    def get_model(self):
        return self.model_runner.get_model()

    def get_kv_cache_spec(self) -> dict[str, KVCacheSpec]:
        return self.model_runner.get_kv_cache_spec()

    @torch.inference_mode()
    def determine_available_memory(self) -> int:
        """Bytes left for the KV cache after weights, peak activations, comm arenas
        and everything else already allocated on the device."""
        cc = self.vllm_config.cache_config
        if self.device.type != "cuda":
            # CPU plumbing path: a fixed modest budget
            return int(min(cc.swap_space, 4.0) * 2**30) if cc.num_gpu_blocks_override is None else 0
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(self.device)
        self.model_runner.profile_run()
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info(self.device)
        peak_torch = torch.cuda.max_memory_allocated(self.device)
        now_torch = torch.cuda.memory_allocated(self.device)
        torch.cuda.empty_cache()
        free_after, _ = torch.cuda.mem_get_info(self.device)
        # memory this process holds outside the torch allocator (RCCL buffers, IPC arenas, HIP context):
        # measured against the free memory seen at start-up, so that other processes on the device
        # (a co-located instance, a trainer) are not billed to this one
        non_torch = (self.init_free - free_after) - torch.cuda.memory_reserved(self.device)
        activation_peak = peak_torch - now_torch
        budget = total * cc.gpu_memory_utilization
        # hipGraph pools are captured after the KV cache exists: reserve for them up front
        graph_reserve = 0
        if self.model_runner.use_graphs:
            graph_reserve = int(1.5 * 2**30) + 64 * 2**20 * len(self.model_runner.capture_sizes) // 8
        avail = budget - now_torch - activation_peak - max(non_torch, 0) - graph_reserve
        logger.info("memory: total %.1f GiB, budget %.1f, weights+static %.1f, activation peak %.1f, "
                    "non-torch %.1f, graph reserve %.1f -> KV cache %.1f GiB",
                    total / 2**30, budget / 2**30, now_torch / 2**30, activation_peak / 2**30,
                    max(non_torch, 0) / 2**30, graph_reserve / 2**30, avail / 2**30)
        if envs.VDA_REHEARSE_ON_ONE_GPU and torch.distributed.is_initialized() \
                and torch.distributed.get_world_size() > 1:
            # ranks sharing one device: nobody allocates its KV cache while another rank still measures
            torch.distributed.barrier()
        return int(max(avail, 0))

    def initialize_from_config(self, kv_cache_config) -> None:
        if isinstance(kv_cache_config, (list, tuple)):      # one entry per rank, or one for all
            kv_cache_config = kv_cache_config[self.rank if len(kv_cache_config) > 1 else 0]
        self.model_runner.initialize_kv_cache(kv_cache_config)

    def compile_or_warm_up_model(self) -> None:
        mr = self.model_runner
        if self.device.type == "cuda":
            # touch the eager kernels at a few sizes so first requests do not pay lazy initialisation
            for n in (1, 16, min(512, mr.max_num_tokens)):
                mr._dummy_run(n, with_metadata=True, num_seqs=min(n, mr.max_num_reqs))
            mr.capture_model()
            torch.cuda.synchronize()
        current_platform.seed_everything(self.vllm_config.model_config.seed)

    # ------------------------------------------------------------------
    @torch.inference_mode()
    def execute_model(self, scheduler_output: SchedulerOutput) -> Optional[ModelRunnerOutput]:
        pp = pstate.get_pp_group()
        tp = pstate.get_tp_group()
        inter = None
        # non-root token-parallel ranks run attention only: no activations travel between their stages
        attn_only = getattr(self.model_runner.model, "is_tknp_worker", False)
        # rows of the activations that matter: known to both stages from the scheduler output they share, which is
        # what lets the peer-memory pipe do without per-hop metadata
        rows = scheduler_output.total_num_scheduled_tokens
        if not pp.is_first_rank and rows and not attn_only:
            d = pp.recv_tensor_dict(all_gather_group=tp, rows=rows)
            inter = IntermediateTensors(d)
        out = self.model_runner.execute_model(scheduler_output, inter)
        # torchrun mode: every rank runs its own scheduler and needs the sampled tokens, so the
        # last stage shares its result with the earlier stages
        share = pp.world_size > 1 and scheduler_output.total_num_scheduled_tokens > 0 and \
            self.parallel_config.distributed_executor_backend == "external_launcher"
        if isinstance(out, IntermediateTensors):
            pp.send_tensor_dict(out.tensors, all_gather_group=tp, rows=rows)
            return pp.broadcast_object(None, src=pp.world_size - 1) if share else None
        if share:
            out = pp.broadcast_object(out if pp.is_last_rank else None, src=pp.world_size - 1)
        return out

    def profile(self, is_start: bool = True) -> None:
        if self.profiler is None:
            raise RuntimeError("set VLLM_TORCH_PROFILER_DIR to enable profiling")
        if is_start:
            self.profiler.start()
        else:
            self.profiler.stop()

    def comm_info(self) -> dict:
        """Per parallel group of this rank: the ranks its RCCL communicator and its peer-memory arena span and the
        all-reduce calls per path (GroupCoordinator.comm_info); read by bench.py for its result line."""
        from ..parallel import state as ps
        out = {}
        for key, getter in (("tp", ps.get_tp_group), ("pp", ps.get_pp_group)):
            try:
                out[key] = getter().comm_info()
            except Exception:   # noqa: BLE001 - group not initialised (single process)
                out[key] = None
        return out

    def check_health(self) -> bool:
        return True

    @torch.inference_mode()
    def execute_dummy_batch(self) -> None:
        """One-token forward pass that produces nothing: keeps cross-replica collectives matched
        while this data-parallel replica has no requests."""
        self.model_runner._dummy_run(1)

    # ---- sleep / wake (RLHF co-location) -------------------------------------
    def sleep(self, level: int = 1) -> None:
        """Level 1: weights to pinned host memory, KV cache dropped.  Level 2: both dropped
        (weights must be reloaded or pushed again before waking)."""
        if self.device.type != "cuda" or self._sleeping:
            return
        mr = self.model_runner
        free0 = torch.cuda.mem_get_info(self.device)[0]
        if mr.sleep_allocator is not None and not getattr(mr, "kv_regions", None):
            # address-stable path: pages go away, tensors / views / captured graphs stay valid
            n = mr.sleep_allocator.sleep(offload_tags=("weights",) if level == 1 else ())
            self._sleeping = level
            logger.info("sleep level %d released %.2f GiB (addresses kept)", level, n / 2**30)
            return
        from ..device_allocator import offload_module
        self._sleep_saved = offload_module(mr.model, keep_copy=(level == 1))
        # KV cache: drop every view and chunk (exported IPC regions are closed too)
        for m in mr.attn_layers.values():
            m.k_cache = m.v_cache = None
        mr.kv_caches = []
        for r in getattr(mr, "kv_regions", []):
            r.close()
        mr.kv_regions, mr.kv_chunks, mr.kv_arena = [], [], None
        mr.graphs.clear()
        gc.collect()
        torch.cuda.empty_cache()
        self._sleeping = level
        logger.info("sleep level %d freed %.2f GiB", level,
                    (torch.cuda.mem_get_info(self.device)[0] - free0) / 2**30)

    def wake_up(self, tags: Optional[list[str]] = None) -> None:
        if not self._sleeping:
            return
        mr = self.model_runner
        if mr.sleep_allocator is not None and mr.sleep_allocator.asleep:
            mr.sleep_allocator.wake_up(tags)
            if not mr.sleep_allocator.asleep:
                self._sleeping = 0
            return
        from ..device_allocator import restore_module
        if tags is None or "weights" in tags:
            restore_module(mr.model, self._sleep_saved, self.device)
            self._sleep_saved = {}
        if (tags is None or "kv_cache" in tags) and not mr.kv_caches:
            mr.initialize_kv_cache(mr.kv_cache_config)
            mr.capture_model()
        if tags is None or set(tags) >= {"weights", "kv_cache"}:
            self._sleeping = 0

    def is_sleeping(self) -> bool:
        return bool(self._sleeping)

    # ---- misc RPC targets --------------------------------------------------------
    def save_sharded_state(self, path: str, pattern: Optional[str] = None, max_size: Optional[int] = None) -> None:
        from ..models.loader import ShardedStateLoader
        ShardedStateLoader.save_model(self.model_runner.model, path, pattern, max_size)

    def add_lora(self, lora_request) -> bool:
        return self.model_runner.lora_manager.add_adapter(lora_request)

    def remove_lora(self, lora_id: int) -> bool:
        return self.model_runner.lora_manager.remove_adapter(lora_id)

    def list_loras(self) -> set[int]:
        return self.model_runner.lora_manager.list_adapters()

    def pin_lora(self, lora_id: int) -> bool:
        return self.model_runner.lora_manager.pin_adapter(lora_id)

    def shutdown(self) -> None:
        """Orderly end of a multi-rank run: the ranks meet (bounded wait), then every communicator, IPC
        mapping and process group is closed here, in a known order, rather than by interpreter teardown,
        where a peer that has already gone makes a communication helper thread abort the process."""
        if getattr(self, "_shut", False):
            return
        self._shut = True
        mr = getattr(self, "model_runner", None)
        if mr is not None:
            conn = getattr(mr, "kv_connector", None)
            if conn is not None:
                try:
                    conn.shutdown()
                except Exception as e:   # noqa: BLE001
                    logger.warning("KV connector shutdown: %s", e)
            if getattr(mr, "tk_rx", None) is not None:
                try:
                    mr.tk_rx.close()
                except Exception:        # noqa: BLE001
                    pass
                mr.tk_rx = None
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        from ..parallel.all2all import close_dispatchers
        try:
            if torch.cuda.is_available() and self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            world = pstate.get_world_group()
            if world.cpu_group is not None:
                import datetime
                dist.monitored_barrier(world.cpu_group, timeout=datetime.timedelta(seconds=10),
                                       wait_all_ranks=False)
        except Exception as e:           # noqa: BLE001 - a peer is gone: close what is ours anyway
            logger.warning("peers did not meet at shutdown: %s", e)
        try:
            close_dispatchers()
            pstate.cleanup_dist_env_and_memory()
        except Exception as e:           # noqa: BLE001
            logger.warning("distributed teardown: %s", e)
