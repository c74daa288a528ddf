"""Clients of engine cores that run in child processes.

Reference: vllm/v1/engine/core_client.py (MPClient :354, SyncMPClient :498,
AsyncMPClient :657, DPAsyncMPClient :853, DPLBAsyncMPClient :966),
vllm/v1/engine/utils.py (process launch + handshake).
"""
from __future__ import annotations

import asyncio
import contextlib
import multiprocessing
import os
import queue
import threading
import time
import weakref
from concurrent.futures import Future
from typing import Any, Optional

from .. import envs
from ..config import VllmConfig
from ..logger import init_logger
from ..request import EngineCoreOutputs, EngineCoreRequest, EngineCoreRequestType
from ..utils import get_mp_context, kill_process_tree
from .core_client import EngineCoreClient, EngineDeadError
from .core_proc import ENGINE_CORE_DEAD, HANDSHAKE_TIMEOUT_S, EngineCoreProc
from .transport import AsyncConnection, Connection, Listener, make_ipc_address

logger = init_logger(__name__)


class CoreEngineProcManager:
    """Spawns engine-core processes and performs the HELLO/READY handshake.

    Three shapes:
    * default: listen on a fresh Unix socket, spawn ``num_engines`` cores, wait for all of them;
    * ``num_remote > 0``: listen on TCP (``listen_address``) instead, so that cores started on other
      nodes by ``vda serve --headless`` can join; wait for local + remote cores;
    * ``spawn=False`` (attach): only listen on ``listen_address`` and wait; somebody else (the
      multi-API-server launcher) starts the cores and tells them about this listener.
    """

    def __init__(self, vllm_config: VllmConfig, executor_class: type, log_stats: bool, num_engines: int = 1,
                 start_index: int = 0, local_start_index: int = 0, coordinator_address: Optional[str] = None,
                 listen_address: Optional[str] = None, num_remote: int = 0, spawn: bool = True):
        self.listener = Listener(listen_address or make_ipc_address("engine"))
        self.procs: list[multiprocessing.Process] = []
        self.conns: dict[int, Connection] = {}
        if spawn:
            self.procs = launch_engine_cores(vllm_config, executor_class, log_stats, self.listener.address,
                                             num_engines, start_index, local_start_index, coordinator_address)
        self._wait_ready(num_engines + num_remote)

    def _wait_ready(self, n: int) -> None:
        deadline = time.monotonic() + HANDSHAKE_TIMEOUT_S
        pending: list[Connection] = []
        ready = 0
        while ready < n:
            if time.monotonic() > deadline:
                self.close()
                raise TimeoutError("engine core did not become ready in time")
            for p in self.procs:
                if not p.is_alive() and p.exitcode not in (0, None):
                    self.close()
                    raise RuntimeError(f"engine core process {p.name} failed to start (exit code {p.exitcode}); "
                                       "see its log above for the root cause")
            c = self.listener.accept(timeout=0.2)
            if c is not None:
                pending.append(c)
            for c in list(pending):
                try:
                    while c.poll(0):
                        msg = c.recv()
                        st = msg.get("status")
                        if st == "READY":
                            self.conns[msg["engine_index"]] = c
                            pending.remove(c)
                            ready += 1
                            break
                        if st == "FAILED":
                            self.close()
                            raise RuntimeError(f"engine core failed during start-up:\n{msg.get('error')}")
                except EOFError:
                    pending.remove(c)

    def sentinels(self) -> list:
        return [p.sentinel for p in self.procs]

    def finished_procs(self) -> dict[str, int]:
        return {p.name: p.exitcode for p in self.procs if p.exitcode is not None}

    def close(self) -> None:
        for c in self.conns.values():
            c.close()
        for p in self.procs:
            if p.is_alive():
                p.terminate()
        deadline = time.monotonic() + 5
        for p in self.procs:
            p.join(max(0.0, deadline - time.monotonic()))
            if p.is_alive() and p.pid is not None:
                kill_process_tree(p.pid)
        self.listener.close()


def launch_engine_cores(vllm_config: VllmConfig, executor_class: type, log_stats: bool, address,
                        num_engines: int, start_index: int = 0, local_start_index: int = 0,
                        coordinator_address: Optional[str] = None) -> list:
    """Start ``num_engines`` engine-core processes that connect to ``address`` (one front-end address
    or a list of them).  Engine ``i`` gets global DP rank ``start_index + i``."""
    ctx = get_mp_context()
    procs = []
    for i in range(num_engines):
        idx = start_index + i
        p = ctx.Process(target=EngineCoreProc.run_engine_core, name=f"EngineCore_{idx}",
                        kwargs=dict(vllm_config=vllm_config, executor_class=executor_class, log_stats=log_stats,
                                    address=address, engine_index=idx, dp_rank=idx,
                                    local_dp_rank=local_start_index + i, coordinator_address=coordinator_address))
        env_patch = {}
        if vllm_config.parallel_config.data_parallel_size > 1 and vllm_config.device_config.device == "cuda":
            # each DP replica owns a contiguous slice of the node's GPUs
            ws = vllm_config.parallel_config.world_size
            lo = (local_start_index + i) * ws
            env_patch["HIP_VISIBLE_DEVICES"] = ",".join(str(d) for d in range(lo, lo + ws))
        with _patched_env(env_patch):
            p.start()
        procs.append(p)
    return procs


@contextlib.contextmanager
def _patched_env(patch: dict):
    old = {k: os.environ.get(k) for k in patch}
    os.environ.update(patch)
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


class MPClient(EngineCoreClient):
    def __init__(self, vllm_config: VllmConfig, executor_class: type, log_stats: bool, num_engines: int = 1,
                 coordinator_address: Optional[str] = None, attach: Optional[dict] = None, num_remote: int = 0,
                 listen_address: Optional[str] = None):
        """``attach = {"address": ..., "num_engines": n, "coordinator_address": ...}``: join engine cores
        that another process launches (one API server of several) instead of spawning them."""
        self.vllm_config = vllm_config
        self.attach = attach
        if attach is not None:
            self.manager = CoreEngineProcManager(vllm_config, executor_class, log_stats, attach["num_engines"],
                                                 listen_address=attach["address"], spawn=False)
        else:
            self.manager = CoreEngineProcManager(vllm_config, executor_class, log_stats, num_engines,
                                                 coordinator_address=coordinator_address,
                                                 listen_address=listen_address, num_remote=num_remote)
        self.conns = [self.manager.conns[i] for i in sorted(self.manager.conns)]
        self.engine_dead = False
        self._call_id = 0
        self._finalizer = weakref.finalize(self, self.manager.close)
        self._monitor = threading.Thread(target=self._monitor_procs, daemon=True, name="engine-monitor")
        self._monitor.start()

    def _monitor_procs(self) -> None:
        from multiprocessing.connection import wait
        sent = self.manager.sentinels()
        if not sent:
            return
        wait(sent)
        if self._finalizer.alive and not getattr(self, "_shutting_down", False):
            self.engine_dead = True
            logger.error("engine core process died unexpectedly: %s", self.manager.finished_procs())
            self._on_engine_dead()

    def _on_engine_dead(self) -> None:
        pass

    def _ensure_alive(self) -> None:
        if self.engine_dead:
            raise EngineDeadError()

    def _next_call_id(self) -> int:
        self._call_id += 1
        return self._call_id

    def shutdown(self) -> None:
        self._shutting_down = True
        self._finalizer()


class SyncMPClient(MPClient):
    def __init__(self, vllm_config, executor_class, log_stats):
        super().__init__(vllm_config, executor_class, log_stats)
        self.conn = self.conns[0]
        self.outputs: "queue.Queue[Any]" = queue.Queue()
        self._utility: dict[int, Future] = {}
        self._reader = threading.Thread(target=self._read_loop, daemon=True, name="engine-output-reader")
        self._reader.start()

    def _read_loop(self) -> None:
        try:
            while True:
                msg = self.conn.recv()
                if isinstance(msg, str) and msg == ENGINE_CORE_DEAD:
                    self.engine_dead = True
                    self._on_engine_dead()
                    return
                if msg.utility_output is not None:
                    uo = msg.utility_output
                    fut = self._utility.pop(uo.call_id, None)
                    if fut is not None:
                        if uo.failure_message is not None:
                            fut.set_exception(RuntimeError(uo.failure_message))
                        else:
                            fut.set_result(uo.result)
                    continue
                self.outputs.put(msg)
        except (EOFError, OSError):
            if not getattr(self, "_shutting_down", False):
                self.engine_dead = True
                self._on_engine_dead()

    def _on_engine_dead(self) -> None:
        self.outputs.put(EngineDeadError())
        for f in list(self._utility.values()):
            if not f.done():
                f.set_exception(EngineDeadError())

    def _send(self, kind: EngineCoreRequestType, payload: Any) -> None:
        self._ensure_alive()
        try:
            self.conn.send((kind, payload))
        except (OSError, EOFError):
            raise EngineDeadError() from None

    def get_output(self) -> EngineCoreOutputs:
        out = self.outputs.get()
        if isinstance(out, Exception):
            raise out
        return out

    def add_request(self, request: EngineCoreRequest) -> None:
        self._send(EngineCoreRequestType.ADD, request)

    def abort_requests(self, request_ids: list[str]) -> None:
        if request_ids and not self.engine_dead:
            self._send(EngineCoreRequestType.ABORT, request_ids)

    def call_utility(self, method: str, *args) -> Any:
        cid = self._next_call_id()
        fut: Future = Future()
        self._utility[cid] = fut
        self._send(EngineCoreRequestType.UTILITY, (cid, method, list(args)))
        return fut.result()

    def profile(self, is_start=True): self.call_utility("profile", is_start)
    def reset_prefix_cache(self): return self.call_utility("reset_prefix_cache")
    def reset_mm_cache(self): self.call_utility("reset_mm_cache")
    def sleep(self, level=1): self.call_utility("sleep", level)
    def wake_up(self, tags=None): self.call_utility("wake_up", tags)
    def is_sleeping(self): return self.call_utility("is_sleeping")
    def execute_dummy_batch(self): self.call_utility("execute_dummy_batch")
    def add_lora(self, r): return self.call_utility("add_lora", r)
    def remove_lora(self, i): return self.call_utility("remove_lora", i)
    def list_loras(self): return self.call_utility("list_loras")
    def pin_lora(self, i): return self.call_utility("pin_lora", i)
    def save_sharded_state(self, path, pattern=None, max_size=None):
        self.call_utility("save_sharded_state", path, pattern, max_size)

    def collective_rpc(self, method, timeout=None, args=(), kwargs=None):
        if not isinstance(method, str):
            raise TypeError("collective_rpc across processes needs a method name, not a callable")
        return self.call_utility("collective_rpc", method, timeout, tuple(args), kwargs)


class AsyncMPClient(MPClient):
    def __init__(self, vllm_config, executor_class, log_stats, num_engines: int = 1,
                 coordinator_address: Optional[str] = None, attach: Optional[dict] = None, num_remote: int = 0,
                 listen_address: Optional[str] = None):
        super().__init__(vllm_config, executor_class, log_stats, num_engines, coordinator_address, attach,
                         num_remote, listen_address)
        self.aconns: list[Optional[AsyncConnection]] = [None] * len(self.conns)
        self.outputs: Optional[asyncio.Queue] = None
        self._utility: dict[int, asyncio.Future] = {}
        self._reader_tasks: list[asyncio.Task] = []
        self._loop: Optional[asyncio.AbstractEventLoop] = None
        self._started: Optional[asyncio.Future] = None

    async def _ensure_started(self) -> None:
        if self._started is not None:
            if not self._started.done():      # another coroutine is half-way through the start
                await asyncio.shield(self._started)
            return
        self._loop = asyncio.get_running_loop()
        self._started = self._loop.create_future()
        self.outputs = asyncio.Queue()
        for i, c in enumerate(self.conns):
            self.aconns[i] = await AsyncConnection.from_connection(c)
            self._reader_tasks.append(asyncio.create_task(self._read_loop(i)))
        self._started.set_result(None)

    async def _read_loop(self, i: int) -> None:
        ac = self.aconns[i]
        try:
            while True:
                msg = await ac.recv()
                if isinstance(msg, str) and msg == ENGINE_CORE_DEAD:
                    break
                if msg.utility_output is not None:
                    uo = msg.utility_output
                    fut = self._utility.pop(uo.call_id, None)
                    if fut is not None and not fut.done():
                        if uo.failure_message is not None:
                            fut.set_exception(RuntimeError(uo.failure_message))
                        else:
                            fut.set_result(uo.result)
                    continue
                self._on_outputs(i, msg)
                await self.outputs.put(msg)
        except (asyncio.IncompleteReadError, ConnectionError, OSError):
            pass
        except asyncio.CancelledError:
            return
        if not getattr(self, "_shutting_down", False):
            self.engine_dead = True
            self._fail_all()

    def _on_outputs(self, engine: int, msg: EngineCoreOutputs) -> None:
        pass

    def _on_engine_dead(self) -> None:
        if self._loop is not None:
            self._loop.call_soon_threadsafe(self._fail_all)

    def _fail_all(self) -> None:
        if self.outputs is not None:
            self.outputs.put_nowait(EngineDeadError())
        for f in list(self._utility.values()):
            if not f.done():
                f.set_exception(EngineDeadError())

    async def _send(self, engine: int, kind: EngineCoreRequestType, payload: Any) -> None:
        self._ensure_alive()
        await self._ensure_started()
        try:
            await self.aconns[engine].send((kind, payload))
        except (ConnectionError, OSError):
            raise EngineDeadError() from None

    def _engine_for(self, request: EngineCoreRequest) -> int:
        return 0

    async def get_output_async(self) -> EngineCoreOutputs:
        await self._ensure_started()
        out = await self.outputs.get()
        if isinstance(out, Exception):
            raise out
        return out

    async def add_request_async(self, request: EngineCoreRequest) -> None:
        await self._send(self._engine_for(request), EngineCoreRequestType.ADD, request)

    async def abort_requests_async(self, request_ids: list[str]) -> None:
        if request_ids and not self.engine_dead:
            await self._send(0, EngineCoreRequestType.ABORT, request_ids)

    async def call_utility_async(self, method: str, *args, engine: int = 0) -> Any:
        await self._ensure_started()
        cid = self._next_call_id()
        fut = asyncio.get_running_loop().create_future()
        self._utility[cid] = fut
        await self._send(engine, EngineCoreRequestType.UTILITY, (cid, method, list(args)))
        return await fut

    async def call_all_async(self, method: str, *args) -> list:
        return await asyncio.gather(*(self.call_utility_async(method, *args, engine=i)
                                      for i in range(len(self.conns))))

    async def profile_async(self, is_start=True): await self.call_all_async("profile", is_start)
    async def reset_prefix_cache_async(self): return all(await self.call_all_async("reset_prefix_cache"))
    async def reset_mm_cache_async(self): await self.call_all_async("reset_mm_cache")
    async def sleep_async(self, level=1): await self.call_all_async("sleep", level)
    async def wake_up_async(self, tags=None): await self.call_all_async("wake_up", tags)
    async def is_sleeping_async(self): return all(await self.call_all_async("is_sleeping"))
    async def add_lora_async(self, r): return all(await self.call_all_async("add_lora", r))
    async def remove_lora_async(self, i): return all(await self.call_all_async("remove_lora", i))
    async def list_loras_async(self): return (await self.call_all_async("list_loras"))[0]
    async def pin_lora_async(self, i): return all(await self.call_all_async("pin_lora", i))
    async def is_healthy_async(self): return all(await self.call_all_async("is_healthy"))

    async def collective_rpc_async(self, method, timeout=None, args=(), kwargs=None):
        res = await self.call_all_async("collective_rpc", method, timeout, tuple(args), kwargs)
        return [x for r in res for x in r]

    def shutdown(self) -> None:
        self._shutting_down = True
        for t in self._reader_tasks:
            t.cancel()
        super().shutdown()


class DPAsyncMPClient(AsyncMPClient):
    """One engine core per data-parallel rank behind one front end; requests carry the
    engine they were sent to so aborts go to the right place.  Placement: round robin."""

    def __init__(self, vllm_config, executor_class, log_stats, attach: Optional[dict] = None):
        pc = vllm_config.parallel_config
        from .dp import CoordinatorSubscription, DPCoordinator
        self.coordinator = None
        self.subscription = None
        if attach is not None:
            # one API server of several: the launcher owns the coordinator, we read the load it publishes
            super().__init__(vllm_config, executor_class, log_stats, attach=attach)
            if attach.get("coordinator_address"):
                self.subscription = CoordinatorSubscription(attach["coordinator_address"])
        else:
            local = pc.data_parallel_size_local or pc.data_parallel_size
            remote = pc.data_parallel_size - local
            host, listen = "127.0.0.1", None
            coord_port = 0
            if remote > 0:
                # engines on other nodes find us at the agreed address; the coordinator sits one port above
                host = pc.data_parallel_master_ip
                listen = f"tcp://{host}:{pc.data_parallel_rpc_port}"
                coord_port = pc.data_parallel_rpc_port + 1
            self.coordinator = DPCoordinator(pc.data_parallel_size, host=host, port=coord_port)
            super().__init__(vllm_config, executor_class, log_stats, local,
                             coordinator_address=self.coordinator.engine_address, num_remote=remote,
                             listen_address=listen)
        self.req_to_engine: dict[str, int] = {}
        self._rr = 0
        self.current_wave = 0
        self.engines_running = False
        self.counts = [[0, 0] for _ in self.conns]   # [waiting, running] per engine

    def _engine_for(self, request: EngineCoreRequest) -> int:
        if request.data_parallel_rank is not None:
            e = request.data_parallel_rank
        else:
            e = self._pick()
        self.req_to_engine[request.request_id] = e
        request.current_wave = self.current_wave
        return e

    def _pick(self) -> int:
        self._rr = (self._rr + 1) % len(self.conns)
        return self._rr

    def _on_outputs(self, engine: int, msg: EngineCoreOutputs) -> None:
        for rid in msg.finished_requests or ():
            self.req_to_engine.pop(rid, None)
        for o in msg.outputs:
            if o.finished:
                self.req_to_engine.pop(o.request_id, None)
        if msg.wave_complete is not None:
            self.current_wave = max(self.current_wave, msg.wave_complete + 1)
            self.engines_running = False
        if msg.start_wave is not None and (not self.engines_running or msg.start_wave >= self.current_wave):
            # an engine got a request stamped with a wave that is already over: wake the others
            self.current_wave = msg.start_wave
            self.engines_running = True
            for i in range(len(self.conns)):
                if i != engine:
                    asyncio.ensure_future(self._send(i, EngineCoreRequestType.START_DP_WAVE,
                                                     (self.current_wave, engine)))
        if msg.scheduler_stats is not None:
            self.counts[engine] = [msg.scheduler_stats.num_waiting_reqs, msg.scheduler_stats.num_running_reqs]

    async def add_request_async(self, request: EngineCoreRequest) -> None:
        e = self._engine_for(request)
        await self._send(e, EngineCoreRequestType.ADD, request)
        self.counts[e][0] += 1
        if not self.engines_running:
            # wake the other replicas so their collectives stay in lock-step
            self.engines_running = True
            await asyncio.gather(*(self._send(i, EngineCoreRequestType.START_DP_WAVE, (self.current_wave, e))
                                   for i in range(len(self.conns)) if i != e))

    async def abort_requests_async(self, request_ids: list[str]) -> None:
        by_engine: dict[int, list[str]] = {}
        for rid in request_ids:
            e = self.req_to_engine.pop(rid, None)
            if e is not None:
                by_engine.setdefault(e, []).append(rid)
        for e, ids in by_engine.items():
            if not self.engine_dead:
                await self._send(e, EngineCoreRequestType.ABORT, ids)

    def shutdown(self) -> None:
        super().shutdown()
        if self.coordinator is not None:
            self.coordinator.close()
        if self.subscription is not None:
            self.subscription.close()


class DPLBAsyncMPClient(DPAsyncMPClient):
    """Least-loaded placement from the queue lengths the engines report
    (waiting requests weigh 4x running ones, as in the reference :985-1013)."""

    def _pick(self) -> int:
        best, best_score = 0, None
        n = len(self.conns)
        sub = self.subscription
        if sub is not None and len(sub.counts) == n:
            # several front ends: the published load covers everybody's requests
            self.counts = [list(c) for c in sub.counts]
            sub.counts = []
        for k in range(n):
            i = (self._rr + 1 + k) % n
            w, r = self.counts[i]
            score = w * 4 + r
            if best_score is None or score < best_score:
                best, best_score = i, score
        self._rr = best
        return best
