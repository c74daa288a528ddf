"""Small shared helpers."""
from __future__ import annotations

import importlib
import itertools
import os
import threading
import uuid
from typing import Any, Callable, Iterable, TypeVar

T = TypeVar("T")


def cdiv(a: int, b: int) -> int:
    return -(-a // b)


def random_uuid() -> str:
    return uuid.uuid4().hex


def resolve_obj_by_qualname(qualname: str) -> Any:
    mod, name = qualname.rsplit(".", 1)
    return getattr(importlib.import_module(mod), name)


class Counter:
    def __init__(self, start: int = 0):
        self._it = itertools.count(start)
        self._lock = threading.Lock()

    def __next__(self) -> int:
        with self._lock:
            return next(self._it)

    def reset(self) -> None:
        self._it = itertools.count(0)


def is_list_of(x: Any, typ: type) -> bool:
    return isinstance(x, list) and all(isinstance(v, typ) for v in x)


def flatten(xs: Iterable[Iterable[T]]) -> list[T]:
    return [y for x in xs for y in x]


def get_mp_context():
    import multiprocessing
    from .. import envs
    return multiprocessing.get_context(envs.VLLM_WORKER_MULTIPROC_METHOD)


def kill_process_tree(pid: int) -> None:
    """Terminate exactly this process and its descendants (never by name pattern)."""
    import psutil
    try:
        parent = psutil.Process(pid)
    except psutil.NoSuchProcess:
        return
    procs = parent.children(recursive=True) + [parent]
    for p in procs:
        try:
            p.terminate()
        except psutil.NoSuchProcess:
            pass
    _, alive = psutil.wait_procs(procs, timeout=4)
    for p in alive:
        try:
            p.kill()
        except psutil.NoSuchProcess:
            pass


def set_process_title(name: str) -> None:
    try:
        import setproctitle  # type: ignore
        setproctitle.setproctitle(name)
    except Exception:
        pass
