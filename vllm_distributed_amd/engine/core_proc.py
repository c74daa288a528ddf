"""Engine core as its own process: socket IO threads around a busy loop.

    client --(ADD / ABORT / UTILITY / START_DP_WAVE)--> input thread -> input_queue
    busy loop: drain input_queue; step; outputs -> output_queue -> output thread --> client

Start-up handshake: the engine connects to the client's listener, sends HELLO,
builds the EngineCore (loads the model, sizes the KV cache, captures graphs),
then sends READY.  Any fatal error is reported as an ENGINE_CORE_DEAD frame so
that clients fail fast with EngineDeadError.
Reference: vllm/v1/engine/core.py (EngineCoreProc :362-809, handshake :426-540,
busy loop :598-606, DPEngineCoreProc :812-978).
"""
from __future__ import annotations

import os
import queue
import signal
import threading
import time
import traceback
from typing import Any, Optional

from .. import envs
from ..config import VllmConfig
from ..logger import decorate_process_output, init_logger
from ..request import (EngineCoreOutputs, EngineCoreRequest, EngineCoreRequestType, UtilityOutput)
from ..utils import set_process_title
from .core import EngineCore
from .transport import Connection, connect

logger = init_logger(__name__)
ENGINE_CORE_DEAD = "ENGINE_CORE_DEAD"
HANDSHAKE_TIMEOUT_S = 600


class EngineCoreProc(EngineCore):
    def __init__(self, vllm_config: VllmConfig, executor_class: type, log_stats: bool, address,
                 engine_index: int = 0):
        """``address``: one front-end address, or a list of them (one per API-server process; the
        position in the list is the ``client_index`` that requests carry and outputs are routed by)."""
        self.engine_index = engine_index
        self.input_queue: "queue.Queue[tuple]" = queue.Queue()
        self.output_queue: "queue.Queue[Any]" = queue.Queue()
        addresses = [address] if isinstance(address, str) else list(address)
        self.conns: list[Connection] = [connect(a, timeout=HANDSHAKE_TIMEOUT_S) for a in addresses]
        self.conn = self.conns[0]
        for c in self.conns:
            c.send({"status": "HELLO", "engine_index": engine_index, "pid": os.getpid()})
        super().__init__(vllm_config, executor_class, log_stats, executor_fail_callback=self._executor_failed)
        for c in self.conns:
            c.send({"status": "READY", "engine_index": engine_index,
                    "num_gpu_blocks": vllm_config.cache_config.num_gpu_blocks})
        self.engines_running = False
        self._live_clients = len(self.conns)
        self._in_ts = [threading.Thread(target=self._input_loop, args=(i,), daemon=True, name=f"core-input-{i}")
                       for i in range(len(self.conns))]
        self._out_t = threading.Thread(target=self._output_loop, daemon=True, name="core-output")
        for t in self._in_ts:
            t.start()
        self._out_t.start()

    def _executor_failed(self) -> None:
        self.input_queue.put((EngineCoreRequestType.EXECUTOR_FAILED, None))

    # ---- IO threads ---------------------------------------------------------
    def _input_loop(self, client: int) -> None:
        try:
            while True:
                kind, payload = self.conns[client].recv()
                if kind == EngineCoreRequestType.UTILITY:
                    payload = (*payload, client)
                self.input_queue.put((kind, payload))
        except (EOFError, OSError):
            # the engine stops when its last client has gone away
            self.input_queue.put((EngineCoreRequestType.CLIENT_GONE, client))

    def _output_loop(self) -> None:
        while True:
            item = self.output_queue.get()
            if item is None:
                return
            client = getattr(item, "_client", 0)
            targets = self.conns if client < 0 else [self.conns[client]]
            for c in targets:
                try:
                    c.send(item)
                except (EOFError, OSError):
                    if len(self.conns) == 1:
                        return

    # ---- busy loop -------------------------------------------------------------
    def run_busy_loop(self) -> None:
        while True:
            self._process_input_queue()
            self._process_engine_step()

    def _has_work(self) -> bool:
        return self.scheduler.has_requests() or bool(self.batch_queue) or self._inflight is not None or bool(self._ready)

    def _process_input_queue(self) -> None:
        waited = False
        while not self._has_work() and not self.engines_running:
            if not waited and self.log_stats:
                logger.debug("engine core idle")
                waited = True
            self._handle(*self.input_queue.get())
        while True:
            try:
                item = self.input_queue.get_nowait()
            except queue.Empty:
                return
            self._handle(*item)

    def _process_engine_step(self) -> bool:
        outs, executed = self.step_fn()
        for client_idx, o in (outs.items() if outs else ()):
            o.engine_index = self.engine_index
            o._client = client_idx if client_idx < len(self.conns) else 0
            self.output_queue.put(o)
        return executed

    def _handle(self, kind, payload) -> None:
        if kind is None:
            raise SystemExit(0)
        if kind == EngineCoreRequestType.ADD:
            self.add_request(payload)
        elif kind == EngineCoreRequestType.ABORT:
            self.abort_requests(payload)
        elif kind == EngineCoreRequestType.UTILITY:
            call_id, method, args, *who = payload
            out = UtilityOutput(call_id)
            try:
                out.result = getattr(self, method)(*args)
            except BaseException as e:   # noqa: BLE001 - reported to the caller
                logger.exception("utility %s failed", method)
                out.failure_message = f"{method} failed: {e}\n{traceback.format_exc()}"
            reply = EngineCoreOutputs(engine_index=self.engine_index, utility_output=out)
            reply._client = who[0] if who else 0
            self.output_queue.put(reply)
        elif kind == EngineCoreRequestType.CLIENT_GONE:
            self._live_clients -= 1
            if self._live_clients <= 0:
                raise SystemExit(0)
            self.abort_requests([rid for rid, r in self.scheduler.requests.items() if r.client_index == payload])
        elif kind == EngineCoreRequestType.EXECUTOR_FAILED:
            raise RuntimeError("a worker process died")
        elif kind == EngineCoreRequestType.START_DP_WAVE:
            self._start_wave(payload)
        else:
            logger.error("unknown request type %r", kind)

    def _start_wave(self, payload) -> None:
        pass

    # ---- entry point ---------------------------------------------------------------
    @staticmethod
    def run_engine_core(vllm_config: VllmConfig, executor_class: type, log_stats: bool, address: str,
                        engine_index: int = 0, dp_rank: int = 0, local_dp_rank: int = 0,
                        coordinator_address: Optional[str] = None) -> None:
        set_process_title(f"vda-engine-core-{engine_index}")
        decorate_process_output(f"EngineCore_{engine_index}")
        stop = {"flag": False}

        def on_signal(signum, frame):
            if not stop["flag"]:
                stop["flag"] = True
                raise SystemExit(0)

        signal.signal(signal.SIGTERM, on_signal)
        signal.signal(signal.SIGINT, signal.SIG_IGN)    # the parent decides
        core: Optional[EngineCoreProc] = None
        try:
            pc = vllm_config.parallel_config
            if pc.data_parallel_size > 1:
                from .dp import DPEngineCoreProc
                pc.data_parallel_rank, pc.data_parallel_rank_local = dp_rank, local_dp_rank
                core = DPEngineCoreProc(vllm_config, executor_class, log_stats, address, engine_index,
                                        coordinator_address)
            else:
                core = EngineCoreProc(vllm_config, executor_class, log_stats, address, engine_index)
            core.run_busy_loop()
        except SystemExit:
            logger.debug("engine core exiting")
        except BaseException:
            logger.exception("engine core crashed")
            if core is not None:
                for c in core.conns:
                    try:
                        c.send(ENGINE_CORE_DEAD)
                    except Exception:
                        pass
            else:
                for a in ([address] if isinstance(address, str) else address):
                    try:
                        c = connect(a, timeout=2)
                        c.send({"status": "FAILED", "engine_index": engine_index, "error": traceback.format_exc()})
                    except Exception:
                        pass
            raise
        finally:
            if core is not None:
                core.output_queue.put(None)
                try:
                    core.shutdown()
                except Exception:
                    pass
