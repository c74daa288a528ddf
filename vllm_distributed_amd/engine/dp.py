"""Data-parallel engine cores and their coordinator.

Replicas of a dense model share nothing at run time, so they run *independently*: no
dummy batches, no per-step agreement, and an idle replica sleeps.  Only a model whose
layers communicate across replicas (expert parallelism spanning the DP ranks) needs every
replica to enter each forward pass together; those run in *lock-step*: a replica without
work executes a dummy batch while any replica is busy, and every ``SYNC_EVERY`` steps the
replicas agree through the coordinator on whether the current wave of requests is over.

The coordinator is a small socket service in the front-end process: engines report queue
lengths and take part in the lock-step agreement; additional API-server processes subscribe
to the published load so that each can balance requests on its own.
Reference: vllm/v1/engine/core.py:812-978 (DPEngineCoreProc), vllm/v1/engine/coordinator.py.
"""
from __future__ import annotations

import threading
import time
from typing import Optional

from ..config import VllmConfig
from ..logger import init_logger
from ..request import EngineCoreOutputs, EngineCoreRequest
from .core_proc import EngineCoreProc
from .transport import Connection, Listener, connect

logger = init_logger(__name__)
SYNC_EVERY = 32
PUBLISH_INTERVAL_S = 0.1


def needs_lockstep(vllm_config: VllmConfig) -> bool:
    pc = vllm_config.parallel_config
    hf = vllm_config.model_config.hf_config
    n_experts = max(int(getattr(hf, k, 0) or 0) for k in ("num_local_experts", "num_experts", "n_routed_experts"))
    return pc.data_parallel_size > 1 and pc.enable_expert_parallel and n_experts > 0


class DPCoordinator:
    def __init__(self, n_engines: int, host: str = "127.0.0.1", port: int = 0):
        self.n = n_engines
        self.listener = Listener(f"tcp://{host}:{port}")
        self.engine_address = self.listener.address
        self.lock = threading.Lock()
        self.engines: dict[int, Connection] = {}
        self.subscribers: list[Connection] = []
        self.counts = [[0, 0] for _ in range(n_engines)]
        self.current_wave = 0
        self.engines_running = False
        self._sync: dict[int, dict[int, bool]] = {}      # round -> engine -> has_unfinished
        self._closed = False
        threading.Thread(target=self._accept_loop, daemon=True, name="dp-coord-accept").start()
        threading.Thread(target=self._publish_loop, daemon=True, name="dp-coord-publish").start()

    # ---- threads ---------------------------------------------------------------------
    def _accept_loop(self) -> None:
        while not self._closed:
            try:
                c = self.listener.accept(timeout=0.5)
            except OSError:
                return
            if c is not None:
                threading.Thread(target=self._serve, args=(c,), daemon=True, name="dp-coord-conn").start()

    def _serve(self, c: Connection) -> None:
        engine = None
        try:
            while True:
                msg = c.recv()
                kind = msg[0]
                if kind == "HELLO":
                    engine = msg[1]
                    with self.lock:
                        self.engines[engine] = c
                elif kind == "SUBSCRIBE":
                    with self.lock:
                        self.subscribers.append(c)
                elif kind == "STATS":
                    _, e, waiting, running = msg
                    self.counts[e] = [waiting, running]
                elif kind == "SYNC":
                    self._on_sync(msg[1], msg[2], bool(msg[3]))
                elif kind == "WAVE":                     # a front end saw a request for a finished wave
                    with self.lock:
                        self.current_wave = max(self.current_wave, msg[1])
                        self.engines_running = True
        except (EOFError, OSError):
            pass
        finally:
            with self.lock:
                if engine is not None and self.engines.get(engine) is c:
                    del self.engines[engine]
                if c in self.subscribers:
                    self.subscribers.remove(c)

    def _on_sync(self, engine: int, rnd: int, unfinished: bool) -> None:
        with self.lock:
            votes = self._sync.setdefault(rnd, {})
            votes[engine] = unfinished
            if len(votes) < self.n:
                return
            del self._sync[rnd]
            result = any(votes.values())
            if not result:
                self.current_wave += 1
            self.engines_running = result
            conns = list(self.engines.values())
            wave = self.current_wave
        for c in conns:
            try:
                c.send(("SYNC_RESULT", rnd, result, wave))
            except (EOFError, OSError):
                pass

    def _publish_loop(self) -> None:
        while not self._closed:
            time.sleep(PUBLISH_INTERVAL_S)
            with self.lock:
                subs = list(self.subscribers)
                state = ("STATE", [list(c) for c in self.counts], self.current_wave, self.engines_running)
            for c in subs:
                try:
                    c.send(state)
                except (EOFError, OSError):
                    with self.lock:
                        if c in self.subscribers:
                            self.subscribers.remove(c)

    def close(self) -> None:
        self._closed = True
        self.listener.close()
        with self.lock:
            for c in list(self.engines.values()) + self.subscribers:
                c.close()


class CoordinatorSubscription:
    """Front-end side: latest published load, refreshed by a reader thread."""

    def __init__(self, address: str):
        self.conn = connect(address)
        self.conn.send(("SUBSCRIBE",))
        self.counts: list[list[int]] = []
        self.current_wave = 0
        self.engines_running = False
        threading.Thread(target=self._loop, daemon=True, name="dp-coord-sub").start()

    def _loop(self) -> None:
        try:
            while True:
                _, self.counts, self.current_wave, self.engines_running = self.conn.recv()
        except (EOFError, OSError):
            pass

    def notify_wave(self, wave: int) -> None:
        self.conn.send(("WAVE", wave))

    def close(self) -> None:
        self.conn.close()


class DPEngineCoreProc(EngineCoreProc):
    def __init__(self, vllm_config: VllmConfig, executor_class: type, log_stats: bool, address: str,
                 engine_index: int = 0, coordinator_address: Optional[str] = None):
        self.lockstep = needs_lockstep(vllm_config)
        self.dp_size = vllm_config.parallel_config.data_parallel_size
        self.current_wave = 0
        self.step_counter = 0
        self.sync_round = 0
        self._last_stats = (-1, -1)
        self.coord: Optional[Connection] = None
        if coordinator_address is not None:
            self.coord = connect(coordinator_address)
            self.coord.send(("HELLO", engine_index))
        if self.lockstep and self.coord is None:
            raise ValueError("lock-step data parallelism needs a coordinator address")
        super().__init__(vllm_config, executor_class, log_stats, address, engine_index)
        logger.info("DP engine %d/%d running %s", engine_index, self.dp_size,
                    "in lock-step (cross-replica expert parallelism)" if self.lockstep else "independently")

    # ---- requests ----------------------------------------------------------------------
    def add_request(self, request: EngineCoreRequest) -> None:
        if self.lockstep and request.current_wave != self.current_wave:
            if request.current_wave > self.current_wave:
                self.current_wave = request.current_wave
            elif not self.engines_running:
                # the front end still thinks an old wave is running: tell it to wake the others
                self.engines_running = True
                self._announce(EngineCoreOutputs(engine_index=self.engine_index, start_wave=self.current_wave))
        super().add_request(request)

    def _announce(self, msg: EngineCoreOutputs) -> None:
        msg._client = -1            # wave state concerns every front end
        self.output_queue.put(msg)

    def _start_wave(self, payload) -> None:
        wave, _origin = payload
        if self.lockstep and wave >= self.current_wave and not self.engines_running:
            self.current_wave = wave
            self.engines_running = True

    # ---- loop ----------------------------------------------------------------------------
    def _report_stats(self) -> None:
        if self.coord is None:
            return
        counts = self.scheduler.get_request_counts()
        if counts != self._last_stats:
            self._last_stats = counts
            try:
                self.coord.send(("STATS", self.engine_index, counts[0], counts[1]))
            except (EOFError, OSError):
                self.coord = None

    def run_busy_loop(self) -> None:
        while True:
            self._process_input_queue()
            executed = self._process_engine_step()
            self._report_stats()
            if not self.lockstep:
                continue
            local_unfinished = self._has_work()
            if not executed:
                if not local_unfinished and not self.engines_running:
                    continue
                self.execute_dummy_batch()          # keep the cross-replica collectives matched
            self.engines_running = self._agree_unfinished(local_unfinished)
            if not self.engines_running:
                if self.engine_index == 0:
                    self._announce(EngineCoreOutputs(engine_index=0, wave_complete=self.current_wave))
                self.current_wave += 1
                self.step_counter = 0

    def _agree_unfinished(self, local_unfinished: bool) -> bool:
        """Every SYNC_EVERY steps all replicas vote; in between, assume the wave goes on."""
        self.step_counter += 1
        if self.step_counter % SYNC_EVERY:
            return True
        self.sync_round += 1
        self.coord.send(("SYNC", self.engine_index, self.sync_round, local_unfinished))
        while True:
            msg = self.coord.recv()
            if msg[0] == "SYNC_RESULT" and msg[1] == self.sync_round:
                return bool(msg[2])

    def execute_dummy_batch(self) -> None:
        self.model_executor.collective_rpc("execute_dummy_batch")
