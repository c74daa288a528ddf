"""Logging set-up: one ``vda`` root logger, optional JSON dictConfig override,
``info_once``/``warning_once`` helpers, a per-process stdout prefix used by
worker processes, and the opt-in call tracer.

Reference behaviour: vllm/logger.py (dictConfig + *_once helpers),
vllm/v1/executor/multiproc_executor.py:441-443 (process prefix),
vllm/utils/__init__.py:1388-1395 (function tracing).
"""
from __future__ import annotations

import json
import logging
import logging.config
import os
import sys
import threading
from functools import lru_cache

from . import envs

_FORMAT = "%(levelname)s %(asctime)s [%(name)s:%(lineno)d] %(message)s"
_DATEFMT = "%m-%d %H:%M:%S"
_ROOT = "vda"
_configured = False
_lock = threading.Lock()


def _configure() -> None:
    global _configured
    with _lock:
        if _configured:
            return
        _configured = True
        if not envs.VLLM_CONFIGURE_LOGGING:
            return
        path = envs.VLLM_LOGGING_CONFIG_PATH
        if path:
            with open(path) as f:
                cfg = json.load(f)
            if not isinstance(cfg, dict):
                raise ValueError(f"{path} must hold a JSON object for logging.config.dictConfig")
            logging.config.dictConfig(cfg)
            return
        root = logging.getLogger(_ROOT)
        root.setLevel(envs.VLLM_LOGGING_LEVEL)
        handler = logging.StreamHandler(sys.stderr)
        prefix = envs.VLLM_LOGGING_PREFIX
        handler.setFormatter(logging.Formatter(prefix + _FORMAT, _DATEFMT))
        root.addHandler(handler)
        root.propagate = False


class _Logger(logging.LoggerAdapter):
    def process(self, msg, kwargs):
        return msg, kwargs

    @lru_cache(maxsize=4096)
    def _once(self, level: int, msg: str) -> None:
        self.logger.log(level, msg, stacklevel=4)

    def info_once(self, msg: str, *args) -> None:
        self._once(logging.INFO, msg % args if args else msg)

    def warning_once(self, msg: str, *args) -> None:
        self._once(logging.WARNING, msg % args if args else msg)


def init_logger(name: str) -> _Logger:
    _configure()
    if not name.startswith(_ROOT):
        name = f"{_ROOT}.{name.split('.', 1)[-1]}" if "." in name else f"{_ROOT}.{name}"
    return _Logger(logging.getLogger(name), {})


def decorate_process_output(tag: str) -> None:
    """Prefix every line this process prints with ``(tag pid=N)``."""
    pid = os.getpid()
    label = f"({tag} pid={pid}) "

    class _Prefixed:
        def __init__(self, stream):
            self._s = stream
            self._bol = True

        def write(self, data):
            if not data:
                return 0
            out = []
            for chunk in data.splitlines(keepends=True):
                if self._bol:
                    out.append(label)
                out.append(chunk)
                self._bol = chunk.endswith("\n")
            return self._s.write("".join(out))

        def __getattr__(self, k):
            return getattr(self._s, k)

    sys.stdout = _Prefixed(sys.stdout)
    sys.stderr = _Prefixed(sys.stderr)


def enable_function_trace(log_path: str, root_dir: str | None = None) -> None:
    """Log every call/return inside this package to ``log_path`` (very slow)."""
    root_dir = root_dir or os.path.dirname(os.path.abspath(__file__))
    fh = open(log_path, "a", buffering=1)

    def tracer(frame, event, arg):
        if event not in ("call", "return"):
            return tracer
        fn = frame.f_code.co_filename
        if not fn.startswith(root_dir):
            return None
        fh.write(f"{event} {frame.f_code.co_name} {fn}:{frame.f_lineno}\n")
        return tracer

    sys.settrace(tracer)
    threading.settrace(tracer)
