"""Usage record of a process: what was run, on what hardware, with which features.

The record is written to a local JSON-lines file only (``$VLLM_CONFIG_ROOT/usage_stats.json`` or
``~/.config/vda/usage_stats.json``); nothing is ever sent over the network.  It is off unless
``VLLM_NO_USAGE_STATS=0`` (and ``VLLM_DO_NOT_TRACK`` / ``DO_NOT_TRACK`` unset), mirroring the opt-out
switches of the reference.  Reference: vllm/usage/usage_lib.py:30-260.
"""
from __future__ import annotations

import datetime
import json
import os
import platform
import time
import uuid
from enum import Enum
from typing import Any, Optional

from .. import envs

_GLOBAL: dict[str, Any] = {}


class UsageContext(str, Enum):
    UNKNOWN_CONTEXT = "UNKNOWN_CONTEXT"
    LLM_CLASS = "LLM_CLASS"
    API_SERVER = "API_SERVER"
    OPENAI_API_SERVER = "OPENAI_API_SERVER"
    OPENAI_BATCH_RUNNER = "OPENAI_BATCH_RUNNER"
    ENGINE_CONTEXT = "ENGINE_CONTEXT"


def set_runtime_usage_data(key: str, value) -> None:
    _GLOBAL[key] = value


def is_usage_stats_enabled() -> bool:
    if envs.VLLM_NO_USAGE_STATS or envs.VLLM_DO_NOT_TRACK or os.environ.get("DO_NOT_TRACK") == "1":
        return False
    return True


def _path() -> str:
    root = os.environ.get("VLLM_CONFIG_ROOT") or os.path.join(os.path.expanduser("~"), ".config", "vda")
    return os.path.join(root, "usage_stats.json")


class UsageMessage:
    def __init__(self):
        self.uuid = str(uuid.uuid4())

    def report_usage(self, model_architecture: str, usage_context: UsageContext,
                     extra_kvs: Optional[dict[str, Any]] = None) -> Optional[dict]:
        if not is_usage_stats_enabled():
            return None
        from ..platforms import current_platform
        from ..version import __version__
        rec = {
            "uuid": self.uuid, "log_time": int(time.time() * 1e9),
            "date": datetime.datetime.now(datetime.timezone.utc).isoformat(), "source": envs.VLLM_USAGE_SOURCE,
            "context": usage_context.value, "version": __version__, "model_architecture": model_architecture,
            "platform": platform.platform(), "architecture": platform.machine(), "num_cpu": os.cpu_count(),
            "device": getattr(current_platform, "device_name", "unknown"),
        }
        try:
            import torch
            if torch.cuda.is_available():
                p = torch.cuda.get_device_properties(0)
                rec.update(gpu_count=torch.cuda.device_count(), gpu_type=p.name, gpu_memory_per_device=p.total_memory)
        except Exception:
            pass
        rec.update(_GLOBAL)
        rec.update(extra_kvs or {})
        try:
            os.makedirs(os.path.dirname(_path()), exist_ok=True)
            with open(_path(), "a") as f:
                f.write(json.dumps(rec, default=str) + "\n")
        except OSError:
            pass
        return rec


usage_message = UsageMessage()
