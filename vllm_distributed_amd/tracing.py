"""Request tracing.  With the OpenTelemetry SDK installed spans go to the OTLP
endpoint; without it (this image) spans are written as JSON lines to the
endpoint when it is a ``file://`` URL, so the span contents stay testable.
Attribute names follow the gen-ai semantic conventions used by the reference
(vllm/tracing.py:52-131)."""
from __future__ import annotations

import json
import os
import time
import uuid
from typing import Any, Mapping, Optional

from . import envs
from .logger import init_logger

logger = init_logger(__name__)
TRACE_HEADERS = ("traceparent", "tracestate")


class SpanAttributes:
    GEN_AI_USAGE_COMPLETION_TOKENS = "gen_ai.usage.completion_tokens"
    GEN_AI_USAGE_PROMPT_TOKENS = "gen_ai.usage.prompt_tokens"
    GEN_AI_REQUEST_MAX_TOKENS = "gen_ai.request.max_tokens"
    GEN_AI_REQUEST_TOP_P = "gen_ai.request.top_p"
    GEN_AI_REQUEST_TEMPERATURE = "gen_ai.request.temperature"
    GEN_AI_RESPONSE_MODEL = "gen_ai.response.model"
    GEN_AI_REQUEST_ID = "gen_ai.request.id"
    GEN_AI_REQUEST_N = "gen_ai.request.n"
    GEN_AI_USAGE_NUM_SEQUENCES = "gen_ai.usage.num_sequences"
    GEN_AI_LATENCY_TIME_IN_QUEUE = "gen_ai.latency.time_in_queue"
    GEN_AI_LATENCY_TIME_TO_FIRST_TOKEN = "gen_ai.latency.time_to_first_token"
    GEN_AI_LATENCY_E2E = "gen_ai.latency.e2e"
    GEN_AI_LATENCY_TIME_IN_SCHEDULER = "gen_ai.latency.time_in_scheduler"
    GEN_AI_LATENCY_TIME_IN_MODEL_FORWARD = "gen_ai.latency.time_in_model_forward"
    GEN_AI_LATENCY_TIME_IN_MODEL_EXECUTE = "gen_ai.latency.time_in_model_execute"


def is_otel_available() -> bool:
    try:
        import opentelemetry  # noqa: F401
        return True
    except ImportError:
        return False


def extract_trace_headers(headers: Mapping[str, str]) -> Optional[dict[str, str]]:
    out = {k: headers[k] for k in TRACE_HEADERS if k in headers}
    return out or None


def contains_trace_headers(headers: Mapping[str, str]) -> bool:
    return any(h in headers for h in TRACE_HEADERS)


class FileTracer:
    def __init__(self, name: str, path: str):
        self.name, self.path = name, path
        os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)

    def emit(self, span_name: str, start_ns: int, end_ns: int, attributes: dict, parent: Optional[dict]) -> None:
        rec = {"name": span_name, "scope": self.name, "trace_id": (parent or {}).get("traceparent", "").split("-")[1]
               if parent and parent.get("traceparent", "").count("-") >= 3 else uuid.uuid4().hex,
               "span_id": uuid.uuid4().hex[:16], "start_time_unix_nano": start_ns, "end_time_unix_nano": end_ns,
               "attributes": attributes}
        with open(self.path, "a") as f:
            f.write(json.dumps(rec) + "\n")


class OtelTracer:
    def __init__(self, name: str, endpoint: str):
        from opentelemetry import trace
        from opentelemetry.sdk.trace import TracerProvider
        from opentelemetry.sdk.trace.export import BatchSpanProcessor
        if envs.OTEL_EXPORTER_OTLP_TRACES_PROTOCOL == "http/protobuf":
            from opentelemetry.exporter.otlp.proto.http.trace_exporter import OTLPSpanExporter
        else:
            from opentelemetry.exporter.otlp.proto.grpc.trace_exporter import OTLPSpanExporter
        provider = TracerProvider()
        provider.add_span_processor(BatchSpanProcessor(OTLPSpanExporter(endpoint=endpoint)))
        trace.set_tracer_provider(provider)
        self.tracer = provider.get_tracer(name)

    def emit(self, span_name, start_ns, end_ns, attributes, parent):
        from opentelemetry.trace import SpanKind
        from opentelemetry.trace.propagation.tracecontext import TraceContextTextMapPropagator
        ctx = TraceContextTextMapPropagator().extract(parent) if parent else None
        span = self.tracer.start_span(span_name, context=ctx, kind=SpanKind.SERVER, start_time=start_ns)
        for k, v in attributes.items():
            span.set_attribute(k, v)
        span.end(end_time=end_ns)


def init_tracer(name: str, endpoint: str):
    if endpoint.startswith("file://"):
        return FileTracer(name, endpoint[len("file://"):])
    if not is_otel_available():
        raise ValueError("OpenTelemetry is not installed; use a file:// endpoint or install the OTLP exporter")
    return OtelTracer(name, endpoint)


def record_request_span(tracer, request_id: str, arrival: float, out, trace_headers: Optional[dict]) -> None:
    try:
        A = SpanAttributes
        n_out = sum(len(c.token_ids) for c in out.outputs)
        attrs = {A.GEN_AI_REQUEST_ID: request_id, A.GEN_AI_USAGE_PROMPT_TOKENS: len(out.prompt_token_ids or ()),
                 A.GEN_AI_USAGE_COMPLETION_TOKENS: n_out, A.GEN_AI_USAGE_NUM_SEQUENCES: len(out.outputs),
                 A.GEN_AI_LATENCY_E2E: time.time() - arrival}
        tracer.emit("llm_request", int(arrival * 1e9), int(time.time() * 1e9), attrs, trace_headers)
    except Exception:   # tracing must never break serving
        logger.exception("failed to record a request span")
