"""Snapshot of the in-process Prometheus registry as plain objects
(reference: vllm/v1/metrics/reader.py; behind ``LLM.get_metrics()``)."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass
class Metric:
    name: str
    labels: dict


@dataclass
class Counter(Metric):
    value: int


@dataclass
class Gauge(Metric):
    value: float


@dataclass
class Histogram(Metric):
    count: int
    sum: float
    buckets: dict


def get_metrics_snapshot() -> list[Metric]:
    from prometheus_client import REGISTRY
    out: list[Metric] = []
    for fam in REGISTRY.collect():
        if not fam.name.startswith("vllm:"):
            continue
        if fam.type == "gauge":
            out += [Gauge(s.name, dict(s.labels), s.value) for s in fam.samples]
        elif fam.type == "counter":
            out += [Counter(s.name, dict(s.labels), int(s.value)) for s in fam.samples if s.name.endswith("_total")]
        elif fam.type == "histogram":
            groups: dict = {}
            for s in fam.samples:
                lab = {k: v for k, v in s.labels.items() if k != "le"}
                g = groups.setdefault(tuple(sorted(lab.items())), {"labels": lab, "buckets": {}, "count": 0, "sum": 0.0})
                if s.name.endswith("_bucket"):
                    g["buckets"][s.labels["le"]] = int(s.value)
                elif s.name.endswith("_count"):
                    g["count"] = int(s.value)
                elif s.name.endswith("_sum"):
                    g["sum"] = s.value
            out += [Histogram(fam.name, g["labels"], g["count"], g["sum"], g["buckets"]) for g in groups.values()]
    return out
