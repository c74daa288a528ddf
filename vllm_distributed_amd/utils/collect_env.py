"""Environment report for bug reports (reference: vllm/collect_env.py)."""
from __future__ import annotations

import os
import platform
import shutil
import subprocess
import sys


def _run(cmd: list[str]) -> str:
    try:
        return subprocess.run(cmd, capture_output=True, text=True, timeout=20).stdout.strip()
    except Exception as e:   # noqa: BLE001
        return f"<{type(e).__name__}>"


def collect_env() -> str:
    import torch
    from .. import envs
    from ..ops import native
    from ..version import __version__
    lines = [f"vllm_distributed_amd: {__version__}", f"python: {sys.version.split()[0]}",
             f"platform: {platform.platform()}", f"torch: {torch.__version__} (hip {torch.version.hip})",
             f"gpu available: {torch.cuda.is_available()}"]
    if torch.cuda.is_available():
        for i in range(torch.cuda.device_count()):
            p = torch.cuda.get_device_properties(i)
            lines.append(f"  cuda:{i}: {p.name} {p.gcnArchName} {p.total_memory / 2**30:.0f} GiB, {p.multi_processor_count} CUs")
    lines.append(f"hipcc: {_run([shutil.which('hipcc') or 'hipcc', '--version']).splitlines()[0:1]}")
    lines.append(f"ops library: {native.library_path()} (present: {native.is_available()})")
    comm = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "parallel", "libvda_comm.so")
    lines.append(f"comm library: {comm} (present: {os.path.exists(comm)})")
    lines.append("environment knobs set:")
    for k in dir(envs):
        if k.isupper() and k in os.environ:
            lines.append(f"  {k}={os.environ[k]}")
    if shutil.which("rocm-smi"):
        lines.append(_run(["rocm-smi", "--showtopo"])[:4000])
    return "\n".join(lines)
