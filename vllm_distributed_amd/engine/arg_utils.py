"""Flat engine arguments -> VllmConfig, and the auto-generated CLI.

Every field of :class:`EngineArgs` becomes a ``--kebab-case`` flag whose type,
choices and help text are derived from the type hint and from the ``#:`` doc
comment of the config field it maps to.  JSON-valued flags accept inline JSON
(``--kv-transfer-config '{...}'``) and dotted overrides
(``--speculative-config.method ngram``).
Reference: vllm/engine/arg_utils.py (EngineArgs :301, add_cli_args :509, TKNP
flags :732-737, create_engine_config :1093, default batch sizes :1632).
"""
from __future__ import annotations

import argparse
import dataclasses
import json
import typing
from dataclasses import dataclass, field, fields
from typing import Any, Optional, Union, get_args, get_origin

from .. import config as C
from .. import envs
from ..logger import init_logger

logger = init_logger(__name__)


def _json(v: str) -> Any:
    try:
        return json.loads(v)
    except json.JSONDecodeError as e:
        raise argparse.ArgumentTypeError(f"not valid JSON: {e}") from None


def _optional_type(tp):
    origin = get_origin(tp)
    if origin is Union:
        args = [a for a in get_args(tp) if a is not type(None)]
        if len(args) == 1:
            return args[0], True
        return args[0], True
    return tp, False


@dataclass
class EngineArgs:
    model: str = "facebook/opt-125m"
    served_model_name: Optional[Union[str, list[str]]] = None
    tokenizer: Optional[str] = None
    skip_tokenizer_init: bool = False
    enable_sleep_mode: bool = False
    tokenizer_mode: str = "auto"
    trust_remote_code: bool = False
    model_impl: str = "auto"
    dtype: str = "auto"
    seed: int = 0
    max_model_len: Optional[int] = None
    quantization: Optional[str] = None
    enforce_eager: bool = False
    max_seq_len_to_capture: int = 8192
    max_logprobs: int = 20
    revision: Optional[str] = None
    hf_overrides: Optional[dict] = None
    mm_processor_kwargs: Optional[dict] = None
    task: str = "auto"
    generation_config: str = "auto"
    disable_sliding_window: bool = False
    disable_cascade_attn: bool = False
    # load
    load_format: str = "auto"
    download_dir: Optional[str] = None
    model_loader_extra_config: Optional[dict] = None
    ignore_patterns: Optional[list[str]] = None
    # cache
    block_size: int = 16
    gpu_memory_utilization: float = 0.9
    swap_space: float = 4.0
    kv_cache_dtype: str = "auto"
    num_gpu_blocks_override: Optional[int] = None
    enable_prefix_caching: Optional[bool] = None
    prefix_caching_hash_algo: str = "builtin"
    cpu_offload_gb: float = 0.0
    calculate_kv_scales: bool = False
    # parallel
    pipeline_parallel_size: int = 1
    tensor_parallel_size: int = 1
    data_parallel_size: int = 1
    data_parallel_size_local: Optional[int] = None
    data_parallel_rank: Optional[int] = None
    data_parallel_address: Optional[str] = None
    data_parallel_rpc_port: Optional[int] = None
    data_parallel_backend: str = "mp"
    enable_expert_parallel: bool = False
    enable_eplb: bool = False
    num_redundant_experts: int = 0
    eplb_window_size: int = 1000
    eplb_step_interval: int = 3000
    eplb_log_balancedness: bool = False
    enable_token_parallel: bool = False
    token_parallel_size: int = 1
    distributed_executor_backend: Optional[str] = None
    disable_custom_all_reduce: bool = False
    worker_cls: str = "auto"
    max_parallel_loading_workers: Optional[int] = None
    # scheduler
    max_num_batched_tokens: Optional[int] = None
    max_num_seqs: Optional[int] = None
    enable_chunked_prefill: Optional[bool] = None
    max_num_partial_prefills: int = 1
    max_long_partial_prefills: int = 1
    long_prefill_token_threshold: int = 0
    scheduling_policy: str = "fcfs"
    scheduler_cls: str = "vllm_distributed_amd.sched.scheduler.Scheduler"
    async_scheduling: Optional[bool] = None      # None: VDA_ASYNC_SCHEDULING
    disable_chunked_mm_input: bool = False
    # lora
    enable_lora: bool = False
    max_loras: int = 1
    max_lora_rank: int = 16
    max_cpu_loras: Optional[int] = None
    lora_dtype: str = "auto"
    lora_extra_vocab_size: int = 256
    fully_sharded_loras: bool = False
    # speculative / structured / observability / transfer (JSON valued)
    speculative_config: Optional[dict] = None
    guided_decoding_backend: str = "auto"
    reasoning_parser: str = ""
    otlp_traces_endpoint: Optional[str] = None
    collect_detailed_traces: Optional[str] = None
    kv_transfer_config: Optional[dict] = None
    kv_events_config: Optional[dict] = None
    compilation_config: Optional[dict] = None
    additional_config: Optional[dict] = None
    limit_mm_per_prompt: Optional[dict] = None
    disable_log_stats: bool = False
    device: str = "auto"

    def __post_init__(self):
        if self.tokenizer is None:
            self.tokenizer = self.model
        for name in ("speculative_config", "kv_transfer_config", "kv_events_config", "compilation_config",
                     "hf_overrides", "additional_config", "model_loader_extra_config", "limit_mm_per_prompt",
                     "mm_processor_kwargs"):
            v = getattr(self, name)
            if isinstance(v, str):
                setattr(self, name, json.loads(v))
        if isinstance(self.compilation_config, int):
            self.compilation_config = {"level": self.compilation_config}

    # ---- CLI -------------------------------------------------------------------
    _DOC_SOURCES = (C.ModelConfig, C.CacheConfig, C.LoadConfig, C.ParallelConfig, C.SchedulerConfig,
                    C.LoRAConfig, C.ObservabilityConfig, C.DeviceConfig)
    _ALIASES = {"tensor_parallel_size": ["-tp"], "pipeline_parallel_size": ["-pp"], "data_parallel_size": ["-dp"],
                "compilation_config": ["-O"], "token_parallel_size": ["-tknp"]}
    _RENAMED_DOC = {"kv_cache_dtype": "cache_dtype", "scheduling_policy": "policy"}

    @classmethod
    def _docs(cls) -> dict[str, str]:
        docs: dict[str, str] = {}
        for src in cls._DOC_SOURCES:
            for k, v in C.get_attr_docs(src).items():
                docs.setdefault(k, v)
        return docs

    @classmethod
    def add_cli_args(cls, parser: argparse.ArgumentParser) -> argparse.ArgumentParser:
        docs = cls._docs()
        hints = typing.get_type_hints(cls)
        for f in fields(cls):
            tp, optional = _optional_type(hints[f.name])
            flag = "--" + f.name.replace("_", "-")
            names = [flag] + cls._ALIASES.get(f.name, [])
            help_ = docs.get(cls._RENAMED_DOC.get(f.name, f.name), "").replace("%", "%%")   # argparse expands %
            default = f.default if f.default is not dataclasses.MISSING else None
            if f.name == "model":
                continue    # positional / --model handled by the caller
            if tp is bool:
                parser.add_argument(*names, action=argparse.BooleanOptionalAction, default=default, help=help_)
            elif tp is dict or get_origin(tp) is dict:
                parser.add_argument(*names, type=_json, default=default, help=help_ + " (JSON)")
            elif tp is list or get_origin(tp) is list:
                parser.add_argument(*names, nargs="+", default=default, help=help_)
            elif tp in (int, float, str):
                parser.add_argument(*names, type=tp, default=default, help=help_)
            else:
                parser.add_argument(*names, type=str, default=default, help=help_)
        return parser

    @classmethod
    def from_cli_args(cls, args: argparse.Namespace) -> "EngineArgs":
        names = {f.name for f in fields(cls)}
        return cls(**{k: v for k, v in vars(args).items() if k in names})

    # ---- config --------------------------------------------------------------------
    def create_model_config(self) -> C.ModelConfig:
        return C.ModelConfig(
            model=self.model, tokenizer=self.tokenizer, skip_tokenizer_init=self.skip_tokenizer_init, enable_sleep_mode=self.enable_sleep_mode,
            tokenizer_mode=self.tokenizer_mode, trust_remote_code=self.trust_remote_code, model_impl=self.model_impl, dtype=self.dtype,
            seed=self.seed, max_model_len=self.max_model_len, quantization=self.quantization,
            enforce_eager=self.enforce_eager, max_seq_len_to_capture=self.max_seq_len_to_capture,
            max_logprobs=self.max_logprobs, served_model_name=self.served_model_name, revision=self.revision,
            hf_overrides=self.hf_overrides, task=self.task, generation_config=self.generation_config,
            disable_sliding_window=self.disable_sliding_window, disable_cascade_attn=self.disable_cascade_attn,
            limit_mm_per_prompt=self.limit_mm_per_prompt, mm_processor_kwargs=self.mm_processor_kwargs)

    def create_engine_config(self, usage_context=None) -> C.VllmConfig:
        from ..plugins import load_general_plugins
        load_general_plugins()               # plugins may register model architectures the config names
        mc = self.create_model_config()
        device = C.DeviceConfig(self.device)
        on_gpu = device.device == "cuda"
        prefix = self.enable_prefix_caching
        if prefix is None:
            prefix = mc.runner_type == "generate"
        if mc.is_encoder_only:
            if prefix or self.enable_chunked_prefill:
                raise ValueError("encoder-only models attend in both directions: prefix caching and chunked "
                                 "prefill cannot be used with them")
        cache = C.CacheConfig(
            block_size=self.block_size, gpu_memory_utilization=self.gpu_memory_utilization,
            swap_space=self.swap_space, cache_dtype=self.kv_cache_dtype,
            num_gpu_blocks_override=self.num_gpu_blocks_override, enable_prefix_caching=prefix,
            prefix_caching_hash_algo=self.prefix_caching_hash_algo, cpu_offload_gb=self.cpu_offload_gb,
            calculate_kv_scales=self.calculate_kv_scales)
        pkw = {}
        if self.data_parallel_rank is not None:
            pkw["data_parallel_rank"] = self.data_parallel_rank
        if self.data_parallel_address:
            pkw["data_parallel_master_ip"] = self.data_parallel_address
        if self.data_parallel_rpc_port:
            pkw["data_parallel_rpc_port"] = self.data_parallel_rpc_port
        parallel = C.ParallelConfig(
            pipeline_parallel_size=self.pipeline_parallel_size, tensor_parallel_size=self.tensor_parallel_size,
            data_parallel_size=self.data_parallel_size, data_parallel_size_local=self.data_parallel_size_local,
            data_parallel_backend=self.data_parallel_backend,
            data_parallel_external_lb=self.data_parallel_rank is not None,
            enable_expert_parallel=self.enable_expert_parallel, enable_eplb=self.enable_eplb,
            num_redundant_experts=self.num_redundant_experts, eplb_window_size=self.eplb_window_size,
            eplb_step_interval=self.eplb_step_interval, eplb_log_balancedness=self.eplb_log_balancedness,
            enable_token_parallel=self.enable_token_parallel, token_parallel_size=self.token_parallel_size,
            distributed_executor_backend=self.distributed_executor_backend,
            disable_custom_all_reduce=self.disable_custom_all_reduce, worker_cls=self.worker_cls,
            max_parallel_loading_workers=self.max_parallel_loading_workers, **pkw)
        chunked = self.enable_chunked_prefill
        if chunked is None:
            chunked = mc.runner_type == "generate"
        max_seqs = self.max_num_seqs
        max_tokens = self.max_num_batched_tokens
        if max_seqs is None:
            # a 288 GB device keeps a thousand Llama-class sequences resident; the CPU path stays small
            max_seqs = 1024 if on_gpu else 128
        if max_tokens is None:
            max_tokens = (16384 if on_gpu else 4096)
            if not chunked:
                max_tokens = max(max_tokens, mc.max_model_len)
            max_tokens = max(max_tokens, max_seqs)
        sched = C.SchedulerConfig(
            max_num_batched_tokens=max_tokens, max_num_seqs=max_seqs, max_model_len=mc.max_model_len,
            enable_chunked_prefill=chunked, max_num_partial_prefills=self.max_num_partial_prefills,
            max_long_partial_prefills=self.max_long_partial_prefills,
            long_prefill_token_threshold=self.long_prefill_token_threshold, policy=self.scheduling_policy,
            scheduler_cls=self.scheduler_cls, async_scheduling=(bool(envs.VDA_ASYNC_SCHEDULING) if self.async_scheduling is None
                                                                    else self.async_scheduling),
            disable_chunked_mm_input=self.disable_chunked_mm_input)
        load = C.LoadConfig(load_format=self.load_format, download_dir=self.download_dir,
                            model_loader_extra_config=self.model_loader_extra_config,
                            ignore_patterns=self.ignore_patterns)
        lora = C.LoRAConfig(max_lora_rank=self.max_lora_rank, max_loras=self.max_loras,
                            max_cpu_loras=self.max_cpu_loras, lora_dtype=self.lora_dtype,
                            lora_extra_vocab_size=self.lora_extra_vocab_size,
                            fully_sharded_loras=self.fully_sharded_loras) if self.enable_lora else None
        spec = C.SpeculativeConfig(**self.speculative_config) if self.speculative_config else None
        kvt = C.KVTransferConfig(**self.kv_transfer_config) if self.kv_transfer_config else None
        kve = C.KVEventsConfig(**self.kv_events_config) if self.kv_events_config else None
        comp = C.CompilationConfig(**{k: v for k, v in (self.compilation_config or {}).items()
                                      if k in {f.name for f in fields(C.CompilationConfig) if f.init}})
        obs = C.ObservabilityConfig(otlp_traces_endpoint=self.otlp_traces_endpoint,
                                    collect_detailed_traces=self.collect_detailed_traces)
        dec = C.DecodingConfig(backend=self.guided_decoding_backend, reasoning_backend=self.reasoning_parser)
        return C.VllmConfig(model_config=mc, cache_config=cache, parallel_config=parallel, scheduler_config=sched,
                            device_config=device, load_config=load, lora_config=lora, speculative_config=spec,
                            decoding_config=dec, observability_config=obs, kv_transfer_config=kvt,
                            kv_events_config=kve, compilation_config=comp,
                            additional_config=self.additional_config or {})


@dataclass
class AsyncEngineArgs(EngineArgs):
    disable_log_requests: bool = True

    @classmethod
    def add_cli_args(cls, parser: argparse.ArgumentParser, async_args_only: bool = False):
        return super().add_cli_args(parser)
