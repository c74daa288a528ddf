"""Layer-wise profile of a model: how much device time every module of the tree takes, and which module
classes the time goes to.

Built on forward hooks and HIP events rather than on a trace post-processor: a pre-hook and a post-hook
record an event pair on the current stream around every module call, the pairs are resolved after one
synchronisation, and a module's own ("self") time is its elapsed time minus its children's.  Works on the
CPU too (wall-clock time).  Steps replayed from captured graphs run no Python modules, so profile with
``enforce_eager=True`` (the runner's eager path launches the very same kernels).

    with layerwise_profile(model, num_running_seqs=32) as prof:
        llm.generate(...)
    prof.results.print_model_table()
    prof.results.print_summary_table()
    prof.results.convert_stats_to_dict()

Counterpart of the reference's ``vllm/profiler/layerwise_profile.py`` (model table :86, summary table :104,
CSV export :120-133, dict export :134) and of ``examples/offline_inference/profiling.py``.
"""
from __future__ import annotations

import csv
import time
from dataclasses import dataclass, field
from typing import Any, Optional

import torch
import torch.nn as nn


@dataclass
class _Call:
    name: str                      # qualified module name ("model.layers.0.self_attn")
    cls: str                       # class name
    depth: int
    start: Any
    end: Any = None
    children: list = field(default_factory=list)
    total_us: float = 0.0
    self_us: float = 0.0
    cpu_us: float = 0.0


@dataclass
class ModelStatsEntry:
    name: str
    cls: str
    depth: int
    calls: int
    cpu_time_us: float
    device_time_us: float
    self_device_time_us: float
    pct_device_time: float


@dataclass
class SummaryStatsEntry:
    name: str                      # module class
    device_time_us: float          # self time summed over all instances and calls
    pct_device_time: float
    invocations: int


class LayerwiseProfileResults:
    def __init__(self, roots: list[_Call], num_running_seqs: Optional[int], on_gpu: bool):
        self.num_running_seqs = num_running_seqs
        self.on_gpu = on_gpu
        self._roots = roots
        self.total_us = sum(r.total_us for r in roots) or 1e-9
        self.model_stats = self._model_table()
        self.summary_stats = self._summary_table()

    # ---- aggregation ---------------------------------------------------------------------------------
    def _model_table(self) -> list[ModelStatsEntry]:
        """One row per module of the tree (calls of the same module merged), in tree order."""
        order: list[tuple[str, str, int]] = []
        acc: dict[str, list[float]] = {}

        def walk(c: _Call):
            if c.name not in acc:
                acc[c.name] = [0, 0.0, 0.0, 0.0]
                order.append((c.name, c.cls, c.depth))
            a = acc[c.name]
            a[0] += 1
            a[1] += c.cpu_us
            a[2] += c.total_us
            a[3] += c.self_us
            for ch in c.children:
                walk(ch)
        for r in self._roots:
            walk(r)
        return [ModelStatsEntry(n, cls, d, int(acc[n][0]), acc[n][1], acc[n][2], acc[n][3],
                                100.0 * acc[n][2] / self.total_us) for n, cls, d in order]

    def _summary_table(self) -> list[SummaryStatsEntry]:
        acc: dict[str, list[float]] = {}

        def walk(c: _Call):
            a = acc.setdefault(c.cls, [0.0, 0])
            a[0] += c.self_us
            a[1] += 1
            for ch in c.children:
                walk(ch)
        for r in self._roots:
            walk(r)
        rows = [SummaryStatsEntry(k, v[0], 100.0 * v[0] / self.total_us, int(v[1])) for k, v in acc.items()]
        return sorted(rows, key=lambda e: -e.device_time_us)

    # ---- output --------------------------------------------------------------------------------------
    def print_model_table(self, max_depth: Optional[int] = None, file=None) -> None:
        unit = "device" if self.on_gpu else "wall"
        print(f"{'module':<64} {'calls':>6} {'cpu us':>10} {unit + ' us':>12} {'self us':>10} {'%':>6}", file=file)
        for e in self.model_stats:
            if max_depth is not None and e.depth > max_depth:
                continue
            label = ("  " * e.depth + (e.name.rsplit(".", 1)[-1] or "<root>") + f" ({e.cls})")[:64]
            print(f"{label:<64} {e.calls:>6} {e.cpu_time_us:>10.1f} {e.device_time_us:>12.1f} "
                  f"{e.self_device_time_us:>10.1f} {e.pct_device_time:>6.2f}", file=file)

    def print_summary_table(self, file=None) -> None:
        print(f"{'module class':<40} {'self us':>12} {'%':>7} {'invocations':>12}", file=file)
        for e in self.summary_stats:
            print(f"{e.name[:40]:<40} {e.device_time_us:>12.1f} {e.pct_device_time:>7.2f} {e.invocations:>12}", file=file)

    def export_model_stats_table_csv(self, filename: str) -> None:
        with open(filename, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["name", "class", "depth", "calls", "cpu_time_us", "device_time_us", "self_device_time_us",
                        "pct_device_time"])
            for e in self.model_stats:
                w.writerow([e.name, e.cls, e.depth, e.calls, f"{e.cpu_time_us:.3f}", f"{e.device_time_us:.3f}",
                            f"{e.self_device_time_us:.3f}", f"{e.pct_device_time:.4f}"])

    def export_summary_stats_table_csv(self, filename: str) -> None:
        with open(filename, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["class", "self_device_time_us", "pct_device_time", "invocations"])
            for e in self.summary_stats:
                w.writerow([e.name, f"{e.device_time_us:.3f}", f"{e.pct_device_time:.4f}", e.invocations])

    def convert_stats_to_dict(self) -> dict[str, Any]:
        return {"metadata": {"num_running_seqs": self.num_running_seqs, "device": "gpu" if self.on_gpu else "cpu",
                             "total_time_us": self.total_us},
                "model_stats": [e.__dict__ for e in self.model_stats],
                "summary_stats": [e.__dict__ for e in self.summary_stats]}


class LayerwiseProfiler:
    def __init__(self, model: nn.Module, num_running_seqs: Optional[int] = None, max_depth: Optional[int] = None):
        self.model = model
        self.num_running_seqs = num_running_seqs
        self.max_depth = max_depth
        self.results: Optional[LayerwiseProfileResults] = None
        self._handles: list = []
        self._stack: list[_Call] = []
        self._roots: list[_Call] = []
        p = next(model.parameters(), None)
        self._gpu = bool(p is not None and p.is_cuda)

    def _mark(self):
        if self._gpu:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        return time.perf_counter()

    def __enter__(self) -> "LayerwiseProfiler":
        names = {m: n for n, m in self.model.named_modules()}

        def pre(mod, args):
            depth = len(self._stack)
            if self.max_depth is not None and depth > self.max_depth:
                self._stack.append(None)            # keep the stack balanced, record nothing
                return
            c = _Call(names.get(mod, ""), type(mod).__name__, depth, None)
            c.cpu_us = time.perf_counter()
            (self._stack[-1].children if self._stack and self._stack[-1] is not None else self._roots).append(c) \
                if not (self._stack and self._stack[-1] is None) else None
            c.start = self._mark()
            self._stack.append(c)

        def post(mod, args, out):
            c = self._stack.pop()
            if c is None:
                return
            c.end = self._mark()
            c.cpu_us = (time.perf_counter() - c.cpu_us) * 1e6

        for m in self.model.modules():
            self._handles.append(m.register_forward_pre_hook(pre))
            self._handles.append(m.register_forward_hook(post, always_call=True))
        return self

    def __exit__(self, *exc) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []
        if self._gpu:
            torch.cuda.synchronize()

        def resolve(c: _Call):
            if c.end is None:                       # forward raised: no time for this call
                c.total_us = 0.0
            elif self._gpu:
                c.total_us = c.start.elapsed_time(c.end) * 1e3
            else:
                c.total_us = (c.end - c.start) * 1e6
            kids = 0.0
            for ch in c.children:
                resolve(ch)
                kids += ch.total_us
            c.self_us = max(c.total_us - kids, 0.0)
        for r in self._roots:
            resolve(r)
        self.results = LayerwiseProfileResults(self._roots, self.num_running_seqs, self._gpu)


def layerwise_profile(model: nn.Module, num_running_seqs: Optional[int] = None,
                      max_depth: Optional[int] = None) -> LayerwiseProfiler:
    return LayerwiseProfiler(model, num_running_seqs, max_depth)
