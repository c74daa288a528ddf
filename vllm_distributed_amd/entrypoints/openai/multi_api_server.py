"""``vda serve --api-server-count N``: N HTTP front-end processes in front of one set of engine cores.

Tokenisation, detokenisation and JSON handling are Python work that saturates one process long
before a GPU does at high request rates.  The launcher (this process) starts the engine cores and
the DP coordinator; every API server listens on its own engine socket, and each core connects to all
of them.  Requests carry the index of the server they came from (``client_index``) and the core
routes outputs back by it.  The HTTP port is shared with ``SO_REUSEPORT``: the kernel spreads
incoming connections over the servers.
Reference: vllm/entrypoints/cli/serve.py:175-249 (run_multi_api_server), vllm/v1/utils.py
(APIServerProcessManager).
"""
from __future__ import annotations

import asyncio
import signal
import socket
import threading
from multiprocessing.connection import wait
from typing import Optional

from ...logger import init_logger

logger = init_logger(__name__)


def _reuseport_socket(host: str, port: int, listen: bool) -> socket.socket:
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEPORT, 1)
    s.bind((host, port))
    if listen:
        s.listen(2048)
        s.set_inheritable(True)
    return s


def _api_server_main(args, index: int, attach: dict, ready=None) -> None:
    from ...logger import decorate_process_output
    from ...utils import set_process_title
    from .api_server import run_server
    set_process_title(f"vda-api-server-{index}")
    decorate_process_output(f"APIServer_{index}")
    sock = _reuseport_socket(args.host or "0.0.0.0", args.port, listen=True)
    asyncio.run(run_server(args, sock=sock, client_index=index, attach=attach, on_ready=ready.set if ready else None))


class MultiAPIServer:
    """Launcher state; ``start()`` returns once every API server accepts HTTP connections."""

    def __init__(self, args):
        from ...engine.arg_utils import AsyncEngineArgs
        from ...engine.transport import make_ipc_address
        from ...executor.base import Executor
        if getattr(args, "model", None) is None and getattr(args, "model_tag", None):
            args.model = args.model_tag
        self.args = args
        self.n = int(args.api_server_count)
        engine_args = AsyncEngineArgs.from_cli_args(args)
        self.cfg = engine_args.create_engine_config()
        self.log_stats = not engine_args.disable_log_stats
        self.executor_class = Executor.get_class(self.cfg)
        self.addresses = [make_ipc_address(f"api{i}") for i in range(self.n)]
        self.coordinator = None
        self.engine_procs: list = []
        self.api_procs: list = []
        self._hold: Optional[socket.socket] = None

    def start(self, timeout: float = 600.0) -> int:
        from ...engine.mp_client import launch_engine_cores
        from ...utils import get_mp_context
        pc = self.cfg.parallel_config
        dp = pc.data_parallel_size
        coord_addr = None
        if dp > 1:
            from ...engine.dp import DPCoordinator
            self.coordinator = DPCoordinator(dp)
            coord_addr = self.coordinator.engine_address
        # reserve the HTTP port (bound, never listening: the kernel delivers nothing to this socket)
        self._hold = _reuseport_socket(self.args.host or "0.0.0.0", self.args.port, listen=False)
        self.args.port = self._hold.getsockname()[1]
        ctx = get_mp_context()
        events = []
        for i, addr in enumerate(self.addresses):
            ev = ctx.Event()
            attach = {"address": addr, "num_engines": dp, "coordinator_address": coord_addr}
            p = ctx.Process(target=_api_server_main, name=f"APIServer_{i}", args=(self.args, i, attach, ev))
            p.start()
            self.api_procs.append(p)
            events.append(ev)
        self.engine_procs = launch_engine_cores(self.cfg, self.executor_class, self.log_stats, self.addresses, dp,
                                                coordinator_address=coord_addr)
        import time
        deadline = time.monotonic() + timeout
        for ev in events:
            while not ev.wait(0.5):
                dead = [p.name for p in self.api_procs + self.engine_procs if p.exitcode not in (None, 0)]
                if dead or time.monotonic() > deadline:
                    self.close()
                    raise RuntimeError(f"start-up failed: {dead or 'timed out'}")
        logger.info("%d API servers on port %d in front of %d engine core(s)", self.n, self.args.port, dp)
        return self.args.port

    def sentinels(self) -> list:
        return [p.sentinel for p in self.api_procs + self.engine_procs]

    def close(self) -> None:
        from ...engine.headless import shutdown_procs
        shutdown_procs(self.api_procs)
        shutdown_procs(self.engine_procs)
        if self.coordinator is not None:
            self.coordinator.close()
        if self._hold is not None:
            self._hold.close()
            self._hold = None


def run_multi_api_server(args) -> None:
    m = MultiAPIServer(args)
    stop = threading.Event()
    signal.signal(signal.SIGTERM, lambda *_: stop.set())
    signal.signal(signal.SIGINT, lambda *_: stop.set())
    try:
        m.start()
        while not stop.is_set():
            if wait(m.sentinels(), timeout=0.5):
                logger.error("a server or engine process exited; shutting down")
                break
    finally:
        m.close()
