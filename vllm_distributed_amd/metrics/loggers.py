"""Stat loggers: periodic log line and the Prometheus ``vllm:*`` metric family.
Reference: vllm/v1/metrics/loggers.py:50-557 (metric names :174-391)."""
from __future__ import annotations

import time
from typing import Optional

from ..logger import init_logger
from .stats import IterationStats, SchedulerStats

logger = init_logger(__name__)


class StatLoggerBase:
    def record(self, scheduler_stats: Optional[SchedulerStats], iteration_stats: Optional[IterationStats]) -> None:
        raise NotImplementedError

    def log(self) -> None:
        pass

    def log_engine_initialized(self) -> None:
        pass


class LoggingStatLogger(StatLoggerBase):
    def __init__(self, vllm_config, engine_index: int = 0, interval: float = 10.0):
        self.engine_index = engine_index
        self.interval = interval
        self.vllm_config = vllm_config
        self._reset(time.monotonic())
        self.last: SchedulerStats = SchedulerStats()
        self.hits = self.queries = 0
        self.spec_drafts = self.spec_draft_tokens = self.spec_accepted = 0

    def _reset(self, now: float) -> None:
        self.t0 = now
        self.prompt_tokens = self.gen_tokens = 0

    def record(self, scheduler_stats, iteration_stats) -> None:
        if iteration_stats is not None:
            self.prompt_tokens += iteration_stats.num_prompt_tokens
            self.gen_tokens += iteration_stats.num_generation_tokens
        if scheduler_stats is not None:
            self.last = scheduler_stats
            pc = scheduler_stats.prefix_cache_stats
            if pc is not None:
                if pc.reset:
                    self.hits = self.queries = 0
                self.hits += pc.hits
                self.queries += pc.queries
            sd = scheduler_stats.spec_decoding_stats
            if sd is not None:
                self.spec_drafts += sd.num_drafts
                self.spec_draft_tokens += sd.num_draft_tokens
                self.spec_accepted += sd.num_accepted_tokens
        if time.monotonic() - self.t0 >= self.interval:
            self.log()

    def log(self) -> None:
        now = time.monotonic()
        dt = max(now - self.t0, 1e-9)
        s = self.last
        msg = ("engine %03d: prompt %.1f tok/s, generation %.1f tok/s, running %d, waiting %d, KV cache %.1f%%, "
               "prefix-cache hit rate %.1f%%")
        args = [self.engine_index, self.prompt_tokens / dt, self.gen_tokens / dt, s.num_running_reqs,
                s.num_waiting_reqs, s.kv_cache_usage * 100, 100.0 * self.hits / self.queries if self.queries else 0.0]
        if self.spec_draft_tokens:
            msg += ", draft acceptance %.1f%% (mean accepted length %.2f)"
            args += [100.0 * self.spec_accepted / self.spec_draft_tokens,
                     1 + self.spec_accepted / max(self.spec_drafts, 1)]
        logger.info(msg, *args)
        self._reset(now)


class PrometheusStatLogger(StatLoggerBase):
    _BUCKETS_TIME = [0.001, 0.005, 0.01, 0.02, 0.04, 0.06, 0.08, 0.1, 0.25, 0.5, 0.75, 1.0, 2.5, 5.0, 7.5, 10.0,
                     20.0, 40.0, 80.0, 160.0, 640.0, 2560.0]
    _BUCKETS_TPOT = [0.01, 0.025, 0.05, 0.075, 0.1, 0.15, 0.2, 0.3, 0.4, 0.5, 0.75, 1.0, 2.5, 5.0, 7.5, 10.0, 20.0,
                     40.0, 80.0]

    def __init__(self, vllm_config, engine_index: int = 0):
        import prometheus_client as pc
        self.pc = pc
        labels = {"model_name": str(vllm_config.model_config.served_model_name if not isinstance(
            vllm_config.model_config.served_model_name, list) else vllm_config.model_config.served_model_name[0]),
            "engine": str(engine_index)}
        self.labels = labels
        names = list(labels)
        max_len = vllm_config.model_config.max_model_len
        tok_buckets = [b for b in (1, 2, 5, 10, 20, 50, 100, 200, 500, 1000, 2000, 5000, 10000, 20000, 50000,
                                   100000, 200000) if b <= max_len] + [max_len]

        def _mk(kind, name, doc, **kw):
            try:
                m = kind(name, doc, labelnames=names, **kw)
            except ValueError:      # already registered in this process (several engines / tests)
                m = pc.REGISTRY._names_to_collectors[name.removesuffix("_total")]
            return m.labels(**labels)

        G, C, H = pc.Gauge, pc.Counter, pc.Histogram
        self.g_running = _mk(G, "vllm:num_requests_running", "Requests in model execution batches.")
        self.g_waiting = _mk(G, "vllm:num_requests_waiting", "Requests waiting to be processed.")
        self.g_kv = _mk(G, "vllm:kv_cache_usage_perc", "KV-cache usage. 1 means 100 percent usage.")
        self.g_gpu_cache = _mk(G, "vllm:gpu_cache_usage_perc", "Deprecated alias of kv_cache_usage_perc.")
        self.c_pc_q = _mk(C, "vllm:prefix_cache_queries", "Prefix cache queries, in tokens.")
        self.c_pc_h = _mk(C, "vllm:prefix_cache_hits", "Prefix cache hits, in tokens.")
        self.c_preempt = _mk(C, "vllm:num_preemptions", "Cumulative number of preemptions.")
        self.c_prompt = _mk(C, "vllm:prompt_tokens", "Prefill tokens processed.")
        self.c_gen = _mk(C, "vllm:generation_tokens", "Generation tokens processed.")
        self.c_success = {}
        try:
            self._success = C("vllm:request_success", "Finished requests by reason.", labelnames=names + ["finished_reason"])
        except ValueError:
            self._success = pc.REGISTRY._names_to_collectors["vllm:request_success"]
        self.h_ttft = _mk(H, "vllm:time_to_first_token_seconds", "Time to first token.", buckets=self._BUCKETS_TIME)
        self.h_tpot = _mk(H, "vllm:time_per_output_token_seconds", "Time per output token.", buckets=self._BUCKETS_TPOT)
        self.h_e2e = _mk(H, "vllm:e2e_request_latency_seconds", "End-to-end request latency.", buckets=self._BUCKETS_TIME)
        self.h_queue = _mk(H, "vllm:request_queue_time_seconds", "Time in WAITING.", buckets=self._BUCKETS_TIME)
        self.h_infer = _mk(H, "vllm:request_inference_time_seconds", "Time in RUNNING.", buckets=self._BUCKETS_TIME)
        self.h_prefill = _mk(H, "vllm:request_prefill_time_seconds", "Time in PREFILL.", buckets=self._BUCKETS_TIME)
        self.h_decode = _mk(H, "vllm:request_decode_time_seconds", "Time in DECODE.", buckets=self._BUCKETS_TIME)
        self.h_prompt_len = _mk(H, "vllm:request_prompt_tokens", "Prompt tokens per request.", buckets=tok_buckets)
        self.h_gen_len = _mk(H, "vllm:request_generation_tokens", "Generated tokens per request.", buckets=tok_buckets)
        self.h_iter = _mk(H, "vllm:iteration_tokens_total", "Tokens per engine step.",
                          buckets=[1, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
        self.h_max_tokens = _mk(H, "vllm:request_params_max_tokens", "max_tokens request parameter.", buckets=tok_buckets)
        self.h_n = _mk(H, "vllm:request_params_n", "n request parameter.", buckets=[1, 2, 5, 10, 20])
        self.c_spec_drafts = _mk(C, "vllm:spec_decode_num_drafts", "Speculative drafts.")
        self.c_spec_draft_tokens = _mk(C, "vllm:spec_decode_num_draft_tokens", "Draft tokens.")
        self.c_spec_accepted = _mk(C, "vllm:spec_decode_num_accepted_tokens", "Accepted draft tokens.")
        self.h_max_gen = _mk(H, "vllm:request_max_num_generation_tokens",
                             "Histogram of maximum number of requested generation tokens.", buckets=tok_buckets)
        # info metrics: a gauge permanently at 1 whose labels carry the values (multiprocess collectors have no Info)
        cc = vllm_config.cache_config
        info = {k: str(v) for k, v in sorted(vars(cc).items()) if not k.startswith("_")
                and isinstance(v, (int, float, str, bool, type(None)))}
        info["engine"] = str(engine_index)
        try:
            g = G("vllm:cache_config_info", "Information of the LLMEngine CacheConfig", labelnames=list(info))
        except ValueError:
            g = pc.REGISTRY._names_to_collectors["vllm:cache_config_info"]
        try:
            g.labels(**info).set(1)
        except ValueError:          # registered earlier with another label set (a second engine in one process)
            pass
        self.g_lora = None
        if getattr(vllm_config, "lora_config", None) is not None:
            self.max_lora = str(vllm_config.lora_config.max_loras)
            try:
                self.g_lora = G("vllm:lora_requests_info", "Running stats on lora requests.",
                                labelnames=["max_lora", "waiting_lora_adapters", "running_lora_adapters"])
            except ValueError:
                self.g_lora = pc.REGISTRY._names_to_collectors["vllm:lora_requests_info"]

    def record(self, scheduler_stats, iteration_stats) -> None:
        s = scheduler_stats
        if s is not None:
            self.g_running.set(s.num_running_reqs)
            self.g_waiting.set(s.num_waiting_reqs)
            self.g_kv.set(s.kv_cache_usage)
            self.g_gpu_cache.set(s.kv_cache_usage)
            if s.prefix_cache_stats is not None:
                self.c_pc_q.inc(s.prefix_cache_stats.queries)
                self.c_pc_h.inc(s.prefix_cache_stats.hits)
            if s.spec_decoding_stats is not None:
                self.c_spec_drafts.inc(s.spec_decoding_stats.num_drafts)
                self.c_spec_draft_tokens.inc(s.spec_decoding_stats.num_draft_tokens)
                self.c_spec_accepted.inc(s.spec_decoding_stats.num_accepted_tokens)
        it = iteration_stats
        if it is None:
            return
        if self.g_lora is not None:
            # the value is the time of the sample; the adapter names ride in the labels
            self.g_lora.labels(max_lora=self.max_lora,
                               waiting_lora_adapters=",".join(sorted(getattr(it, "waiting_lora_adapters", {}) or {})),
                               running_lora_adapters=",".join(sorted(getattr(it, "running_lora_adapters", {}) or {}))
                               ).set_to_current_time()
        for v in getattr(it, "max_num_generation_tokens_iter", ()) or ():
            self.h_max_gen.observe(v)
        for v in getattr(it, "n_params_iter", ()) or ():
            self.h_n.observe(v)
        self.c_preempt.inc(it.num_preempted_reqs)
        self.c_prompt.inc(it.num_prompt_tokens)
        self.c_gen.inc(it.num_generation_tokens)
        self.h_iter.observe(it.num_prompt_tokens + it.num_generation_tokens)
        for v in it.time_to_first_tokens_iter:
            self.h_ttft.observe(v)
        for v in it.time_per_output_tokens_iter:
            self.h_tpot.observe(v)
        for f in it.finished_requests:
            self._success.labels(**self.labels, finished_reason=str(f.finish_reason)).inc()
            self.h_e2e.observe(f.e2e_latency)
            self.h_queue.observe(max(f.queued_time, 0))
            self.h_infer.observe(max(f.inference_time, 0))
            self.h_prefill.observe(max(f.prefill_time, 0))
            self.h_decode.observe(max(f.decode_time, 0))
            self.h_prompt_len.observe(f.num_prompt_tokens)
            self.h_gen_len.observe(f.num_generation_tokens)
            if f.max_tokens_param:
                self.h_max_tokens.observe(f.max_tokens_param)


class StatLoggerManager:
    def __init__(self, vllm_config, engine_index: int = 0, custom: Optional[list] = None, prometheus: bool = True,
                 log_interval: float = 10.0):
        self.loggers: list[StatLoggerBase] = []
        if custom:
            self.loggers = [c(vllm_config, engine_index) if isinstance(c, type) else c for c in custom]
            return
        self.loggers.append(LoggingStatLogger(vllm_config, engine_index, log_interval))
        if prometheus:
            try:
                self.loggers.append(PrometheusStatLogger(vllm_config, engine_index))
            except Exception as e:
                logger.warning("prometheus metrics disabled: %s", e)

    def record(self, scheduler_stats, iteration_stats) -> None:
        for lg in self.loggers:
            lg.record(scheduler_stats, iteration_stats)

    def log(self) -> None:
        for lg in self.loggers:
            lg.log()
