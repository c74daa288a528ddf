"""Multi-process executor: one worker process per GPU of this node.

RPCs go out once through a native shared-memory ring that every worker reads (no per-worker
copy on the scheduler's critical path); replies come back over one pipe per worker, and for
``execute_model`` only the output rank (first TP rank of the last PP stage) replies.  A
monitor thread watches the worker processes: if one dies the executor marks itself failed,
fires the failure callback (the engine core then reports ENGINE_CORE_DEAD) and stops the rest.
With pipeline parallelism ``execute_model`` returns a Future and up to ``pp_size`` batches are
in flight; replies of one worker are matched to calls in FIFO order.

Nothing is pickled between the processes after start-up: messages use the msgpack wire codec.
Reference: vllm/v1/executor/multiproc_executor.py:41-549.
"""
from __future__ import annotations

import multiprocessing as mp
import os
import queue
import signal
import threading
import time
import traceback
import weakref
from concurrent.futures import Future
from multiprocessing.connection import Connection
from typing import Any, Optional

from .. import envs
from ..engine import serial
from ..logger import init_logger
from ..parallel.utils import get_distributed_init_method, get_open_port
from ..runtime.shm_ring import MessageQueue, RingClosed, RingHandle
from .base import Executor
from .uniproc import run_method

logger = init_logger(__name__)

_OK, _ERR = 0, 1


class _WorkerHandle:
    def __init__(self, proc: mp.Process, rank: int, pipe: Connection):
        self.proc, self.rank, self.pipe = proc, rank, pipe
        self.pending: "queue.Queue[Future]" = queue.Queue()
        self.reader: Optional[threading.Thread] = None


def _worker_main(vllm_config, local_rank: int, rank: int, init_method: str, ring: tuple, pipe: Connection,
                 ready_pipe: Connection) -> None:
    """Entry of a worker process: build the worker, attach to the RPC ring, serve until closed."""
    from ..logger import decorate_process_output
    from ..worker.worker import Worker
    signal.signal(signal.SIGINT, signal.SIG_IGN)            # the parent decides when to stop
    stop = {"flag": False}

    def on_term(*_):
        stop["flag"] = True
        raise SystemExit(0)
    signal.signal(signal.SIGTERM, on_term)
    decorate_process_output(f"Worker_r{rank}")
    worker = None
    try:
        worker = Worker(vllm_config, local_rank=local_rank, rank=rank, distributed_init_method=init_method,
                        is_driver_worker=(rank % vllm_config.parallel_config.tensor_parallel_size == 0))
        worker.init_device()
        mqueue = MessageQueue.attach(RingHandle(*ring))
        worker.load_model()
        ready_pipe.send_bytes(serial.encode({"status": "READY", "rank": rank}))
        ready_pipe.close()
    except BaseException:   # noqa: BLE001 - reported to the parent, which raises
        try:
            ready_pipe.send_bytes(serial.encode({"status": "FAILED", "rank": rank, "error": traceback.format_exc()}))
        except Exception:
            pass
        raise
    try:
        while True:
            try:
                method, args, kwargs, reply_rank = mqueue.dequeue()
            except RingClosed:
                break
            try:
                out = run_method(worker, method, tuple(args), kwargs or {})
                if reply_rank is None or reply_rank == rank:
                    pipe.send_bytes(serial.encode((_OK, out)))
            except SystemExit:
                raise
            except BaseException as e:   # noqa: BLE001 - every failure is forwarded to the caller
                logger.exception("worker %d: %s failed", rank, method)
                if reply_rank is None or reply_rank == rank:
                    pipe.send_bytes(serial.encode((_ERR, f"{type(e).__name__}: {e}\n{traceback.format_exc()}")))
    except (SystemExit, KeyboardInterrupt):
        pass
    finally:
        try:
            if worker is not None:
                worker.shutdown()
            mqueue.close()
        except Exception:
            pass


class MultiprocExecutor(Executor):
    def _init_executor(self) -> None:
        self._finalizer = weakref.finalize(self, lambda: None)
        pc = self.parallel_config
        self.world_size = pc.world_size
        tp, pp = pc.tensor_parallel_size, pc.pipeline_parallel_size
        # first TP rank of the last PP stage (of token-parallel group 0: ranks are laid out
        # [token-parallel][pipeline][tensor], fastest last)
        self.output_rank = pp * tp - tp
        self.shutting_down = False
        self.rpc_lock = threading.Lock()
        os.environ.setdefault("OMP_NUM_THREADS", "1")       # one busy thread per worker, not n_cpu each
        init = get_distributed_init_method("127.0.0.1", get_open_port())
        self.mq = MessageQueue.create(self.world_size)
        h = self.mq.export_handle()
        ring = (h.name, h.n_readers, h.n_slots, h.slot_bytes)
        ctx = mp.get_context(envs.VLLM_WORKER_MULTIPROC_METHOD)
        self.workers: list[_WorkerHandle] = []
        ready = []
        # a data-parallel engine sees only its own window of the node's devices (the engine
        # manager masks them), so local ranks always start at 0
        base = 0
        try:
            for rank in range(self.world_size):
                parent, child = ctx.Pipe(duplex=False)
                r_parent, r_child = ctx.Pipe(duplex=False)
                p = ctx.Process(target=_worker_main, name=f"VdaWorker-{rank}", daemon=True,
                                args=(self.vllm_config, base + rank, rank, init, ring, child, r_child))
                p.start()
                child.close()
                r_child.close()
                self.workers.append(_WorkerHandle(p, rank, parent))
                ready.append(r_parent)
            for w, rp in zip(self.workers, ready):
                self._wait_ready(w, rp)
            self.mq.wait_until_ready()
        except BaseException:
            self.shutdown()
            raise
        for w in self.workers:
            w.reader = threading.Thread(target=self._reader_loop, args=(w,), daemon=True,
                                        name=f"vda-worker-reply-{w.rank}")
            w.reader.start()
        self._monitor = threading.Thread(target=self._monitor_loop, daemon=True, name="vda-worker-monitor")
        self._monitor.start()

    def _wait_ready(self, w: _WorkerHandle, rp: Connection) -> None:
        while not rp.poll(0.2):
            if not w.proc.is_alive():
                raise RuntimeError(f"worker {w.rank} exited with code {w.proc.exitcode} during start-up")
        try:
            msg = serial.decode(rp.recv_bytes())
        except EOFError:
            raise RuntimeError(f"worker {w.rank} died during start-up") from None
        finally:
            rp.close()
        if msg.get("status") != "READY":
            raise RuntimeError(f"worker {w.rank} failed to start:\n{msg.get('error')}")

    # ---- reply plumbing -------------------------------------------------------------
    def _reader_loop(self, w: _WorkerHandle) -> None:
        try:
            while True:
                data = w.pipe.recv_bytes()
                status, payload = serial.decode(data)
                fut = w.pending.get_nowait()
                if status == _OK:
                    fut.set_result(payload)
                else:
                    fut.set_exception(RuntimeError(f"worker {w.rank} raised:\n{payload}"))
        except (EOFError, OSError, queue.Empty):
            pass
        self._fail_pending(w, f"worker {w.rank} closed its reply pipe")

    def _fail_pending(self, w: _WorkerHandle, why: str) -> None:
        while True:
            try:
                f = w.pending.get_nowait()
            except queue.Empty:
                return
            if not f.done():
                f.set_exception(RuntimeError(why))

    def _monitor_loop(self) -> None:
        sentinels = {w.proc.sentinel: w for w in self.workers}
        died = mp.connection.wait(list(sentinels))
        if self.shutting_down or not died:
            return
        w = sentinels[died[0]]
        logger.error("worker %d (pid %s) died unexpectedly (exit code %s); shutting the executor down",
                     w.rank, w.proc.pid, w.proc.exitcode)
        self.is_failed = True
        for x in self.workers:
            self._fail_pending(x, f"worker {w.rank} died")
        cb, self._failure_callback = self._failure_callback, None
        self.shutdown()
        if cb is not None:
            cb()

    # ---- RPC ----------------------------------------------------------------------------
    def _rpc(self, method, args: tuple, kwargs: Optional[dict], reply_rank: Optional[int]) -> list[Future]:
        if self.is_failed:
            raise RuntimeError("executor failed: a worker process died")
        if not isinstance(method, str):
            raise TypeError("the multi-process executor takes method names (callables cannot cross processes "
                            "without pickle)")
        targets = self.workers if reply_rank is None else [self.workers[reply_rank]]
        futs = []
        with self.rpc_lock:            # ring order == pending-queue order on every worker
            for w in targets:
                f: Future = Future()
                w.pending.put(f)
                futs.append(f)
            self.mq.enqueue((method, list(args), kwargs or {}, reply_rank))
        return futs

    def collective_rpc(self, method, timeout: Optional[float] = None, args: tuple = (), kwargs: Optional[dict] = None,
                       unique_reply_rank: Optional[int] = None, non_block: bool = False) -> list[Any]:
        futs = self._rpc(method, args, kwargs, unique_reply_rank)
        if non_block:
            return futs
        deadline = None if timeout is None else time.monotonic() + timeout
        out = []
        for f in futs:
            left = None if deadline is None else max(deadline - time.monotonic(), 0.0)
            try:
                out.append(f.result(timeout=left))
            except TimeoutError:
                raise TimeoutError(f"RPC {method} timed out after {timeout} s") from None
        return out

    def execute_model(self, scheduler_output):
        non_block = self.max_concurrent_batches > 1
        res = self.collective_rpc("execute_model", args=(scheduler_output,), unique_reply_rank=self.output_rank,
                                  non_block=non_block, timeout=envs.VLLM_EXECUTE_MODEL_TIMEOUT_SECONDS)
        return res[0]

    @property
    def max_concurrent_batches(self) -> int:
        return self.parallel_config.pipeline_parallel_size

    def check_health(self) -> None:
        if self.is_failed:
            raise RuntimeError("executor failed: a worker process died")
        self.collective_rpc("check_health", timeout=10)

    def shutdown(self) -> None:
        if getattr(self, "shutting_down", True):
            return
        self.shutting_down = True
        try:
            self.mq.close()               # wakes every reader with "closed"
        except Exception:
            pass
        deadline = time.monotonic() + 4.0
        for w in self.workers:
            w.proc.join(max(deadline - time.monotonic(), 0.0))
        for w in self.workers:
            if w.proc.is_alive():
                w.proc.terminate()
        for w in self.workers:
            w.proc.join(2.0)
            if w.proc.is_alive():
                w.proc.kill()
            try:
                w.pipe.close()
            except Exception:
                pass

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass
