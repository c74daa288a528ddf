"""``vda serve --headless``: run engine cores on this node for a front end that lives on another.

The head node runs ``vda serve MODEL --data-parallel-size N --data-parallel-size-local L
--data-parallel-address HEAD --data-parallel-rpc-port P``; it listens for engine cores on
``tcp://HEAD:P`` and runs the DP coordinator on port ``P + 1``.  Every other node runs the same
command with ``--headless --data-parallel-start-rank R`` and its own ``--data-parallel-size-local``;
the cores started here connect to the head, take requests from it and send their outputs back.
Reference: vllm/entrypoints/cli/serve.py:112-172 (run_headless).
"""
from __future__ import annotations

import signal
import threading
from multiprocessing.connection import wait

from ..logger import init_logger

logger = init_logger(__name__)


def run_headless(args, block: bool = True):
    from ..executor.base import Executor
    from .arg_utils import AsyncEngineArgs
    from .mp_client import launch_engine_cores
    if getattr(args, "model", None) is None and getattr(args, "model_tag", None):
        args.model = args.model_tag
    if getattr(args, "api_server_count", 1) > 1:
        raise ValueError("--api-server-count cannot be combined with --headless")
    engine_args = AsyncEngineArgs.from_cli_args(args)
    cfg = engine_args.create_engine_config()
    pc = cfg.parallel_config
    local = pc.data_parallel_size_local
    start = int(getattr(args, "data_parallel_start_rank", 0) or 0)
    if local <= 0:
        raise ValueError("--data-parallel-size-local must be at least 1 in headless mode")
    if start + local > pc.data_parallel_size:
        raise ValueError(f"ranks {start}..{start + local - 1} exceed --data-parallel-size {pc.data_parallel_size}")
    head = f"tcp://{pc.data_parallel_master_ip}:{pc.data_parallel_rpc_port}"
    coord = f"tcp://{pc.data_parallel_master_ip}:{pc.data_parallel_rpc_port + 1}"
    logger.info("headless: starting DP ranks %d..%d for the front end at %s", start, start + local - 1, head)
    procs = launch_engine_cores(cfg, Executor.get_class(cfg), not engine_args.disable_log_stats, head, local,
                                start_index=start, local_start_index=0, coordinator_address=coord)
    if not block:
        return procs
    stop = threading.Event()

    def on_signal(*_):
        stop.set()

    signal.signal(signal.SIGTERM, on_signal)
    signal.signal(signal.SIGINT, on_signal)
    try:
        while not stop.is_set():
            done = wait([p.sentinel for p in procs], timeout=0.5)
            if done:
                bad = {p.name: p.exitcode for p in procs if p.exitcode not in (None, 0)}
                if bad:
                    logger.error("engine core(s) exited: %s", bad)
                break
    finally:
        shutdown_procs(procs)
    return None


def shutdown_procs(procs, grace: float = 5.0) -> None:
    import time

    from ..utils import kill_process_tree
    for p in procs:
        if p.is_alive():
            p.terminate()
    deadline = time.monotonic() + grace
    for p in procs:
        p.join(max(0.0, deadline - time.monotonic()))
        if p.is_alive() and p.pid is not None:
            kill_process_tree(p.pid)
