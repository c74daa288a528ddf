"""Configuration objects.

One dataclass per concern, aggregated by :class:`VllmConfig`.  Field comments
written as ``#: text`` directly above a field are harvested by
:func:`get_attr_docs` and become the ``--help`` text of the auto-generated CLI
flag (engine/arg_utils.py).

Reference for fields and cross-validation rules: vllm/config.py
(ModelConfig:230, CacheConfig:1511, LoadConfig:1711, ParallelConfig:1798 with the
token-parallel fields :1898-1902 and the world-size rule :1990-2010,
SchedulerConfig:2139, SpeculativeConfig:2502, LoRAConfig:2999,
ObservabilityConfig:3735, KVTransferConfig:3826, KVEventsConfig:3922,
CompilationConfig:4014, VllmConfig:4364).  Defaults that depend on device
memory are sized for a 288 GB MI355X rather than an 80 GB part.
"""
from __future__ import annotations

import contextlib
import copy
import hashlib
import inspect
import json
import os
import re
import uuid
from dataclasses import asdict, dataclass, field, fields, is_dataclass
from typing import Any, Literal, Optional, Union

import torch

from . import envs
from .logger import init_logger

logger = init_logger(__name__)

_STR_DTYPE = {
    "half": torch.float16, "float16": torch.float16, "fp16": torch.float16,
    "bfloat16": torch.bfloat16, "bf16": torch.bfloat16,
    "float": torch.float32, "float32": torch.float32, "fp32": torch.float32,
}
# OCP e4m3fn is the only fp8 flavour gfx950 MFMA understands.
FP8_DTYPE = torch.float8_e4m3fn


def get_attr_docs(cls) -> dict[str, str]:
    """Map field name -> the ``#:`` comment block directly above it."""
    try:
        src = inspect.getsource(cls)
    except (OSError, TypeError):
        return {}
    docs: dict[str, str] = {}
    pending: list[str] = []
    for line in src.splitlines():
        s = line.strip()
        if s.startswith("#:"):
            pending.append(s[2:].strip())
            continue
        m = re.match(r"^(\w+)\s*:", s)
        if m and pending:
            docs[m.group(1)] = " ".join(pending)
        pending = []
    return docs


def _hash_of(obj: Any, skip: tuple[str, ...] = ()) -> str:
    def norm(v):
        if is_dataclass(v) and not isinstance(v, type):
            return {f.name: norm(getattr(v, f.name)) for f in fields(v) if f.name not in skip}
        if isinstance(v, dict):
            return {str(k): norm(x) for k, x in sorted(v.items(), key=lambda kv: str(kv[0]))}
        if isinstance(v, (list, tuple, set)):
            return [norm(x) for x in v]
        if isinstance(v, (str, int, float, bool)) or v is None:
            return v
        return str(v)
    return hashlib.sha256(json.dumps(norm(obj), sort_keys=True).encode()).hexdigest()[:16]


class HFConfig:
    """Attribute view over a Hugging Face ``config.json`` dictionary."""

    def __init__(self, d: dict):
        self.__dict__["_d"] = dict(d)

    def __getattr__(self, k):
        try:
            return self.__dict__["_d"][k]
        except KeyError:
            raise AttributeError(k) from None

    def __getstate__(self):
        return self.__dict__["_d"]

    def __setstate__(self, d):
        self.__dict__["_d"] = d

    def __setattr__(self, k, v):
        self._d[k] = v

    def get(self, k, default=None):
        return self._d.get(k, default)

    def to_dict(self) -> dict:
        return dict(self._d)

    def __contains__(self, k):
        return k in self._d

    def __repr__(self):
        return f"HFConfig({self._d.get('architectures')}, hidden={self._d.get('hidden_size')})"


def load_hf_config(model: str, overrides: Optional[dict] = None) -> HFConfig:
    """``config.json`` of a local checkpoint directory, or a built-in preset when
    ``model`` names a well-known architecture and no such directory exists
    (there is no hub access on the target machines)."""
    from .models.presets import PRESETS
    path = os.path.join(model, "config.json")
    if os.path.isfile(path):
        with open(path) as f:
            d = json.load(f)
    elif model.endswith(".gguf") and os.path.isfile(model):
        from .models.gguf import config_from_gguf
        sidecar = os.path.join(os.path.dirname(model), "config.json")
        if os.path.isfile(sidecar):
            with open(sidecar) as f:
                d = json.load(f)
        else:
            d = config_from_gguf(model)
    else:
        key = model.rstrip("/").split("/")[-1].lower()
        if key not in PRESETS:
            raise FileNotFoundError(
                f"{model!r} is neither a local checkpoint directory nor a known preset "
                f"({', '.join(sorted(PRESETS))})")
        d = copy.deepcopy(PRESETS[key])
    dca = d.get("dual_chunk_attention_config")
    side = os.path.join(model, "sparse_attention_config.json")
    if isinstance(dca, dict) and "sparse_attention_config" not in dca and os.path.isfile(side):
        # per layer, per head: ["vertical_and_slash", vertical lines, slash lines, score] (reference:
        # vllm/config.py:1036-1045, model_loader/weight_utils.py:226-262)
        with open(side) as f:
            dca["sparse_attention_config"] = json.load(f)
    if "thinker_config" in d:                  # Qwen2.5-Omni: only the thinker generates text
        d = {**{k: v for k, v in d.items() if k not in ("thinker_config", "talker_config", "token2wav_config")},
             **{k: v for k, v in d["thinker_config"].items() if k not in ("architectures", "model_type")
                and v is not None}}
    if "text_config" in d and ("hidden_size" not in d or "num_attention_heads" not in d):
        if d.get("tie_word_embeddings") is not None:      # the wrapper's own flag (its head may differ from the decoder's)
            d["top_tie_word_embeddings"] = d["tie_word_embeddings"]
        if "hidden_size" in d:                 # a wrapper-level width (Ultravox: its projector's) stays reachable
            d["top_hidden_size"] = d["hidden_size"]
        d = {**d, **{k: v for k, v in d["text_config"].items()
                     if v is not None and not (k == "architectures" and d.get("architectures"))}}
    elif "language_config" in d and "hidden_size" not in d:     # DeepSeek-VL2
        d = {**d, **{k: v for k, v in d["language_config"].items() if k not in ("architectures", "model_type")}}
    elif "llm_config" in d and "hidden_size" not in d:          # InternVL's original layout
        d = {**d, **{k: v for k, v in d["llm_config"].items() if k not in ("architectures", "model_type")}}
    if d.get("model_type") == "whisper" and "hidden_size" not in d:
        # encoder-decoder configs name the decoder's sizes differently: give them the names the engine reads
        d = {**d, "hidden_size": d["d_model"], "num_attention_heads": d["decoder_attention_heads"],
             "num_key_value_heads": d["decoder_attention_heads"], "num_hidden_layers": d["decoder_layers"],
             "max_position_embeddings": d.get("max_target_positions", 448)}
    if d.get("model_type") in ("bart", "mbart") and "hidden_size" not in d:
        d = {**d, "hidden_size": d["d_model"], "num_attention_heads": d["decoder_attention_heads"],
             "num_key_value_heads": d["decoder_attention_heads"], "num_hidden_layers": d["decoder_layers"]}
    if d.get("model_type") in ("exaone", "telechat", "qwen", "chatglm"):
        from .models.renamed_decoders import normalise_config
        d = normalise_config(d)
    if overrides:
        d.update(overrides)
    return HFConfig(d)


# --------------------------------------------------------------------------
@dataclass
class ModelConfig:
    #: Local checkpoint directory (HF layout) or the name of a built-in preset.
    model: str = "facebook/opt-125m"
    #: Tokenizer directory; defaults to the model directory.
    tokenizer: Optional[str] = None
    #: Skip tokenizer construction; prompts must then be given as token ids.
    skip_tokenizer_init: bool = False
    #: Allocate weights and KV cache from the address-stable sleep allocator, so that sleep / wake-up
    #: releases and restores device memory without invalidating captured graphs.
    enable_sleep_mode: bool = False
    #: "auto" | "slow": prefer the Rust tokenizer or force the Python one.
    tokenizer_mode: str = "auto"
    #: Allow custom code from the checkpoint directory (never executed here; kept for CLI parity).
    trust_remote_code: bool = False
    #: "auto": this package's own model code, the transformers fallback for architectures it lacks;
    #: "vllm": own model code only; "transformers": always the fallback (transformers' module code around
    #: this package's attention, KV cache and parallel layers).
    model_impl: str = "auto"
    #: Weight/activation dtype: auto, bfloat16, float16 or float32.
    dtype: Union[str, torch.dtype] = "auto"
    #: Seed for sampling and for dummy-weight initialisation.
    seed: int = 0
    #: Context length; derived from the model config when unset.
    max_model_len: Optional[int] = None
    #: Weight quantisation method (fp8 or None).
    quantization: Optional[str] = None
    #: Never capture hipGraphs; run every step eagerly.
    enforce_eager: bool = False
    #: Largest decode batch captured as a hipGraph.
    max_seq_len_to_capture: int = 8192
    #: Maximum number of logprobs a request may ask for.
    max_logprobs: int = 20
    #: Name(s) under which the model is served.
    served_model_name: Optional[Union[str, list[str]]] = None
    #: Revision string (kept for CLI parity; local checkpoints have none).
    revision: Optional[str] = None
    #: JSON overrides applied on top of the checkpoint's config.json.
    hf_overrides: Optional[dict] = None
    #: Keyword overrides for the model's multimodal input processor (e.g. ``{"num_crops": 4}``).
    mm_processor_kwargs: Optional[dict] = None
    #: "generate", "embed", "classify", "score" or "auto".
    task: str = "auto"
    #: Pooling behaviour overrides for embedding models.
    override_pooler_config: Optional[dict] = None
    #: Skip sliding-window layers' window (treat as full attention).
    disable_sliding_window: bool = False
    #: Never use cascade attention (shared-prefix pass + per-request suffix pass) for batches with a long common prefix.
    disable_cascade_attn: bool = False
    #: Per-modality item limits for multimodal prompts.
    limit_mm_per_prompt: Optional[dict] = None
    #: Default sampling parameters source: "auto" reads generation_config.json.
    generation_config: str = "auto"
    #: Which logits processors pattern is allowed (kept for parity).
    logits_processor_pattern: Optional[str] = None
    #: Enable prompt embeddings as input.
    enable_prompt_embeds: bool = False

    hf_config: HFConfig = field(default=None, repr=False)  # type: ignore[assignment]

    def __post_init__(self):
        if self.tokenizer is None:
            self.tokenizer = self.model
        if self.hf_config is None:
            self.hf_config = load_hf_config(self.model, self.hf_overrides)
        hf = self.hf_config
        self.dtype = self._resolve_dtype(self.dtype, hf)
        derived = self._derive_max_len(hf)
        if self.max_model_len is None:
            self.max_model_len = derived
        elif self.max_model_len > derived and not envs.VLLM_ALLOW_LONG_MAX_MODEL_LEN:
            raise ValueError(
                f"max_model_len={self.max_model_len} exceeds the model's {derived}; "
                "set VLLM_ALLOW_LONG_MAX_MODEL_LEN=1 to override")
        if self.quantization is None:
            qc = hf.get("quantization_config")
            if qc:
                self.quantization = str(qc.get("quant_method", "")).lower() or None
        if self.quantization in ("awq_marlin", "gptq_marlin"):
            self.quantization = self.quantization.split("_")[0]
        if self.quantization == "modelopt":
            # ModelOpt exports name the scheme in quant_algo: FP8 is the plain per-tensor fp8 layout
            algo = str((hf.get("quantization_config") or {}).get("quant_algo", "FP8")).upper()
            self.quantization = "modelopt_fp4" if "FP4" in algo else "fp8"
        if self.quantization not in (None, "fp8", "compressed-tensors", "quark", "awq", "gptq", "bitsandbytes",
                                     "modelopt_fp4"):
            raise ValueError(f"unsupported quantization {self.quantization!r} (supported: fp8, compressed-tensors, "
                             "quark, awq, gptq, bitsandbytes, modelopt, modelopt_fp4)")
        if self.served_model_name is None:
            self.served_model_name = self.model
        if self.task == "auto":
            arch = (self.architectures or [""])[0]
            generative = "CausalLM" in arch or "LMHead" in arch or "ConditionalGeneration" in arch \
                or arch in ("ChatGLMModel", "GritLM", "MiniCPMV", "MiniCPMO", "Ovis", "NVLM_D", "InternVLChatModel",
                            "H2OVLChatModel", "SkyworkR1VChatModel", "UltravoxModel", "Qwen2_5OmniModel")
            self.task = ("generate" if generative else "classify" if "SequenceClassification" in arch
                         or arch.endswith("ForRanking")
                         else "embed" if arch.endswith(("Model", "ForMaskedLM")) else "generate")

    @staticmethod
    def _resolve_dtype(dtype, hf: HFConfig) -> torch.dtype:
        if isinstance(dtype, torch.dtype):
            return dtype
        if dtype == "auto":
            cfg = hf.get("torch_dtype") or hf.get("dtype") or "bfloat16"
            cfg = str(cfg).replace("torch.", "")
            # fp32 checkpoints are served in bf16: MFMA has no fast fp32 path
            return torch.bfloat16 if cfg in ("float32", "float") else _STR_DTYPE.get(cfg, torch.bfloat16)
        if dtype not in _STR_DTYPE:
            raise ValueError(f"unknown dtype {dtype!r}")
        return _STR_DTYPE[dtype]

    @staticmethod
    def _derive_max_len(hf: HFConfig) -> int:
        best = None
        for k in ("max_position_embeddings", "n_positions", "max_seq_len", "seq_length",
                  "max_sequence_length", "model_max_length"):
            v = hf.get(k)
            if v is not None:
                best = v if best is None else min(best, v)
        if best is None:
            best = 2048
        rs = hf.get("rope_scaling") or hf.get("rope_parameters")
        if rs and rs.get("rope_type", rs.get("type")) in ("yarn", "longrope", "linear", "dynamic"):
            factor = rs.get("factor", 1.0)
            base = rs.get("original_max_position_embeddings")
            if rs.get("rope_type", rs.get("type")) == "yarn" and base:
                best = int(base * factor)
            elif rs.get("rope_type", rs.get("type")) in ("linear", "dynamic"):
                best = int(best * factor)
        return int(best)

    # derived geometry ----------------------------------------------------
    @property
    def architectures(self) -> list[str]:
        return list(self.hf_config.get("architectures") or [])

    def get_hidden_size(self) -> int:
        return self.hf_config.get("hidden_size") or self.hf_config.get("n_embd") or self.hf_config.get("d_model")

    def get_vocab_size(self) -> int:
        return self.hf_config.vocab_size

    @property
    def is_attention_free(self) -> bool:
        """State-space models: no attention layers, one fixed-size state per request instead of a KV cache."""
        return any(a in ("MambaForCausalLM", "FalconMambaForCausalLM", "Mamba2ForCausalLM")
                   for a in self.architectures)

    @property
    def is_hybrid(self) -> bool:
        """Attention and state-space layers in one stack: paged KV for the former, one state slot per
        request for the latter; recurrent state rules out prefix reuse."""
        return any(a in ("JambaForCausalLM", "BambaForCausalLM", "GraniteMoeHybridForCausalLM", "NemotronHForCausalLM",
                         "MiniMaxForCausalLM", "MiniMaxText01ForCausalLM", "MiniMaxM1ForCausalLM", "FalconH1ForCausalLM", "MiniMaxVL01ForConditionalGeneration",
                         # encoder-decoder: the cross-attention cache is per-request state of the same kind
                         "WhisperForConditionalGeneration", "BartForConditionalGeneration", "BartModel",
                         "MllamaForConditionalGeneration", "Zamba2ForCausalLM", "Plamo2ForCausalLM",
                         "Florence2ForConditionalGeneration")
                   for a in self.architectures)

    @property
    def is_text_encoder_decoder(self) -> bool:
        """Encoder-decoder over text: a request's plain prompt is the encoder's input."""
        return any(a in ("BartForConditionalGeneration", "BartModel", "Florence2ForConditionalGeneration")
                   for a in self.architectures)

    def get_total_num_attention_heads(self) -> int:
        if self.is_attention_free:
            return 1
        return self.hf_config.get("num_attention_heads") or self.hf_config.get("n_head") or \
            self.hf_config.get("n_heads")

    def get_total_num_kv_heads(self) -> int:
        hf = self.hf_config
        if hf.get("model_type") in ("falcon", "RefinedWeb", "RefinedWebModel") \
                and not hf.get("new_decoder_architecture"):
            # the first two Falcon generations: one shared KV head with multi_query, else one per head
            return 1 if hf.get("multi_query", True) else self.get_total_num_attention_heads()
        for k in ("num_key_value_heads", "n_head_kv", "num_kv_heads", "multi_query_group_num"):
            if hf.get(k) is not None:
                return hf.get(k)
        return self.get_total_num_attention_heads()

    def get_head_size(self) -> int:
        hf = self.hf_config
        if hf.get("head_dim"):
            return hf.head_dim
        return self.get_hidden_size() // self.get_total_num_attention_heads()

    def get_num_attention_heads(self, pc: "ParallelConfig") -> int:
        return self.get_total_num_attention_heads() // pc.tensor_parallel_size

    def get_num_kv_heads(self, pc: "ParallelConfig") -> int:
        # KV heads are replicated once TP exceeds their count
        return max(1, self.get_total_num_kv_heads() // pc.tensor_parallel_size)

    def get_total_num_hidden_layers(self) -> int:
        n = self.hf_config.get("num_hidden_layers") or self.hf_config.get("n_layer") or self.hf_config.get("n_layers")
        if not n:      # configs that list their blocks instead (Nemotron-H)
            n = len(self.hf_config.get("layers_block_type") or self.hf_config.get("hybrid_override_pattern") or [])
        return n

    def get_num_layers(self, pc: "ParallelConfig", pp_rank: Optional[int] = None) -> int:
        from .parallel.utils import get_pp_indices
        n = self.get_total_num_hidden_layers()
        if pp_rank is None:
            pp_rank = pc.rank // pc.tensor_parallel_size % pc.pipeline_parallel_size
        s, e = get_pp_indices(n, pp_rank, pc.pipeline_parallel_size)
        return e - s

    def get_sliding_window(self) -> Optional[int]:
        if self.disable_sliding_window:
            return None
        hf = self.hf_config
        if hf.get("use_sliding_window") is False:
            return None
        return hf.get("sliding_window")

    @property
    def is_moe(self) -> bool:
        hf = self.hf_config
        return bool(hf.get("num_local_experts") or hf.get("num_experts") or hf.get("n_routed_experts"))

    @property
    def runner_type(self) -> str:
        return "generate" if self.task == "generate" else "pooling"

    @property
    def is_encoder_only(self) -> bool:
        """Bidirectional encoders: a prompt runs in one step and its KV blocks are never reused."""
        return any(a.startswith(("Bert", "Roberta", "XLMRoberta", "NomicBert", "GteNew", "GteModel", "ModernBert"))
                   for a in self.architectures)

    def verify_with_parallel_config(self, pc: "ParallelConfig") -> None:
        heads = self.get_total_num_attention_heads()
        if heads % pc.tensor_parallel_size:
            raise ValueError(f"attention heads ({heads}) not divisible by tensor_parallel_size "
                             f"({pc.tensor_parallel_size})")
        kv = self.get_total_num_kv_heads()
        tp = pc.tensor_parallel_size
        if kv % tp and tp % kv:
            raise ValueError(f"kv heads ({kv}) and tensor_parallel_size ({tp}) must divide one another")
        if pc.pipeline_parallel_size > self.get_total_num_hidden_layers():
            raise ValueError("pipeline_parallel_size exceeds the number of layers")
        if pc.enable_expert_parallel and not self.is_moe:
            logger.warning_once("enable_expert_parallel has no effect on a dense model")

    def compute_hash(self) -> str:
        return _hash_of({"model": self.model, "dtype": str(self.dtype), "q": self.quantization,
                         "len": self.max_model_len, "hf": self.hf_config.to_dict()})


# --------------------------------------------------------------------------
@dataclass
class CacheConfig:
    #: Tokens per KV block. 16 matches one MFMA 16x16 token tile; 32 and 64 are also accepted.
    block_size: int = 16
    #: Fraction of device memory this engine may use (weights + KV + graphs + comm arenas).
    gpu_memory_utilization: float = 0.9
    #: Host swap space in GiB (only used for accounting of the pinned KV spill pool).
    swap_space: float = 4.0
    #: KV dtype: "auto" (model dtype) or "fp8" (OCP e4m3).
    cache_dtype: str = "auto"
    #: Override the number of device KV blocks instead of profiling.
    num_gpu_blocks_override: Optional[int] = None
    #: Reuse KV blocks across requests that share a prefix.
    enable_prefix_caching: bool = True
    #: Hash for prefix blocks: "builtin" or "sha256".
    prefix_caching_hash_algo: str = "builtin"
    #: Fraction of the CPU memory to use for the host-side KV spill pool.
    cpu_offload_gb: float = 0.0
    #: Sliding window (filled from the model).
    sliding_window: Optional[int] = None
    #: Compute k/v scales for fp8 caches on the fly.
    calculate_kv_scales: bool = False
    num_gpu_blocks: Optional[int] = None
    num_cpu_blocks: Optional[int] = None

    def __post_init__(self):
        if self.block_size not in (16, 32, 64):
            raise ValueError("block_size must be 16, 32 or 64")
        if not 0.0 < self.gpu_memory_utilization <= 1.0:
            raise ValueError("gpu_memory_utilization must be in (0, 1]")
        if self.cache_dtype not in ("auto", "fp8", "fp8_e4m3", "bfloat16", "float16"):
            raise ValueError(f"unknown kv cache dtype {self.cache_dtype!r}")
        if self.cache_dtype == "fp8_e4m3":
            self.cache_dtype = "fp8"
        if self.prefix_caching_hash_algo not in ("builtin", "sha256"):
            raise ValueError("prefix_caching_hash_algo must be builtin or sha256")

    def compute_hash(self) -> str:
        return _hash_of({"bs": self.block_size, "dt": self.cache_dtype})


@dataclass
class LoadConfig:
    #: auto | safetensors | pt | dummy | sharded_state | npcache | fastsafetensors | runai_streamer (the last two:
    #: the native multi-threaded range reader; extra config "concurrency", "memory_limit") | tensorizer (one
    #: pre-sharded, post-processed file per rank written by ``TensorizerLoader.tensorize_model``; extra config
    #: "tensorizer_uri", "verify_hash", "num_readers")
    load_format: str = "auto"
    #: Directory for downloaded weights (unused: no hub access).
    download_dir: Optional[str] = None
    #: Extra options for the loader (e.g. sharded_state pattern).
    model_loader_extra_config: Optional[dict] = None
    #: Glob patterns of checkpoint files to ignore.
    ignore_patterns: Optional[list[str]] = None
    #: Show a progress bar while loading.
    use_tqdm_on_load: bool = True
    #: Read safetensors shards with this many parallel reader threads.
    num_loader_threads: int = 8

    def __post_init__(self):
        self.load_format = self.load_format.lower()
        ok = ("auto", "safetensors", "pt", "bin", "dummy", "sharded_state", "npcache", "fastsafetensors",
              "runai_streamer", "tensorizer")
        if self.load_format not in ok:
            raise ValueError(f"load_format must be one of {ok}")
        if self.ignore_patterns is None:
            self.ignore_patterns = ["original/**/*"]


@dataclass
class ParallelConfig:
    #: Number of pipeline stages.
    pipeline_parallel_size: int = 1
    #: Number of tensor-parallel ranks per stage.
    tensor_parallel_size: int = 1
    #: Number of engine-level data-parallel replicas.
    data_parallel_size: int = 1
    #: Local data-parallel replicas on this node.
    data_parallel_size_local: Optional[int] = None
    #: Rank of this replica.
    data_parallel_rank: int = 0
    #: Local rank of this replica on its node.
    data_parallel_rank_local: Optional[int] = None
    #: Rendezvous address of the DP group.
    data_parallel_master_ip: str = "127.0.0.1"
    #: RPC port of the DP head node.
    data_parallel_rpc_port: int = 29550
    #: Rendezvous port of the DP group.
    data_parallel_master_port: int = 29500
    #: "mp" (hybrid LB inside one front end) or "external".
    data_parallel_backend: str = "mp"
    #: An external load balancer picks the replica; every replica runs its own API server.
    data_parallel_external_lb: bool = False
    #: Shard MoE experts across TP x DP ranks instead of slicing every expert.
    enable_expert_parallel: bool = False
    #: Rebalance expert placement from observed load.
    enable_eplb: bool = False
    #: Extra physical expert slots for EPLB.
    num_redundant_experts: int = 0
    #: Length of the load window in steps.
    eplb_window_size: int = 1000
    #: Rebalance every this many steps.
    eplb_step_interval: int = 3000
    #: Log per-step balancedness.
    eplb_log_balancedness: bool = False
    #: Shard attention + KV cache by request across extra GPUs (root rank holds the weights).
    enable_token_parallel: bool = False
    #: Number of token-parallel ranks.
    token_parallel_size: int = 1
    #: mp | uni | external_launcher | ray (ray only when the package is installed).
    distributed_executor_backend: Optional[str] = None
    #: Use only RCCL for all-reduce (disable the peer-memory kernels).
    disable_custom_all_reduce: bool = False
    #: Worker class path override.
    worker_cls: str = "auto"
    #: Cap on parallel loader workers to bound host RAM during load.
    max_parallel_loading_workers: Optional[int] = None
    #: Steps between DP "still running" syncs.
    dp_sync_interval: int = 32

    world_size: int = field(init=False, default=1)
    rank: int = 0

    def __post_init__(self):
        if self.enable_token_parallel and self.token_parallel_size < 2:
            self.token_parallel_size = max(self.token_parallel_size, 1)
        if not self.enable_token_parallel:
            self.token_parallel_size = 1
        if self.token_parallel_size > 1 and self.data_parallel_size > 1:
            raise ValueError("token parallelism and data parallelism are mutually exclusive")
        for n in ("pipeline_parallel_size", "tensor_parallel_size", "data_parallel_size",
                  "token_parallel_size"):
            if getattr(self, n) < 1:
                raise ValueError(f"{n} must be >= 1")
        # ranks inside ONE engine: PP x TP x TKNP (DP replicas are separate engines)
        self.world_size = (self.pipeline_parallel_size * self.tensor_parallel_size
                           * self.token_parallel_size)
        if self.data_parallel_size_local is None:
            self.data_parallel_size_local = self.data_parallel_size
        if self.data_parallel_size > 1 and envs.is_set("VLLM_DP_SIZE"):
            self.data_parallel_rank = envs.VLLM_DP_RANK
            self.data_parallel_rank_local = envs.VLLM_DP_RANK_LOCAL
            self.data_parallel_master_ip = envs.VLLM_DP_MASTER_IP
            self.data_parallel_master_port = envs.VLLM_DP_MASTER_PORT
        if self.distributed_executor_backend is None:
            self.distributed_executor_backend = "mp" if self.world_size > 1 else "uni"
        if self.distributed_executor_backend == "ray":
            from .executor.ray_executor import ray_is_available
            if not ray_is_available():
                raise ValueError("distributed_executor_backend='ray' needs the `ray` package, which is not "
                                 "installed; use 'mp' or 'external_launcher'")
        if self.distributed_executor_backend not in ("mp", "uni", "external_launcher", "ray"):
            raise ValueError(f"unknown executor backend {self.distributed_executor_backend!r}")
        if self.distributed_executor_backend == "uni" and self.world_size > 1:
            raise ValueError("the uni executor runs exactly one rank")
        if envs.VLLM_DISABLE_CUSTOM_ALL_REDUCE:
            self.disable_custom_all_reduce = True

    @property
    def world_size_across_dp(self) -> int:
        return self.world_size * self.data_parallel_size

    @property
    def use_ep(self) -> bool:
        return self.enable_expert_parallel and \
            self.tensor_parallel_size * self.data_parallel_size * self.token_parallel_size > 1

    def get_next_dp_init_port(self) -> int:
        p = self.data_parallel_master_port
        self.data_parallel_master_port += 1
        return p

    def compute_hash(self) -> str:
        return _hash_of({"pp": self.pipeline_parallel_size, "tp": self.tensor_parallel_size,
                         "dp": self.data_parallel_size, "tknp": self.token_parallel_size,
                         "ep": self.enable_expert_parallel})


@dataclass
class SchedulerConfig:
    #: Token budget of one engine step (prefill chunks + decode tokens).
    max_num_batched_tokens: Optional[int] = None
    #: Maximum number of requests running concurrently.
    max_num_seqs: Optional[int] = None
    #: Filled from the model config.
    max_model_len: Optional[int] = None
    #: Split long prompts into budget-sized chunks mixed with decodes.
    enable_chunked_prefill: bool = True
    #: Upper bound on prompts being prefilled concurrently.
    max_num_partial_prefills: int = 1
    #: How many of those may be "long" prompts.
    max_long_partial_prefills: int = 1
    #: Prompts longer than this are "long" (0 = 4 percent of max_model_len when partial prefills > 1).
    long_prefill_token_threshold: int = 0
    #: fcfs or priority.
    policy: str = "fcfs"
    #: Extra decode slots reserved per request for speculative tokens.
    num_lookahead_slots: int = 0
    #: Keep the prefix-cache hit of preempted requests disabled.
    disable_chunked_mm_input: bool = False
    #: Scheduler class path override.
    scheduler_cls: str = "vllm_distributed_amd.sched.scheduler.Scheduler"
    #: Delay (fraction of last prompt latency) before scheduling the next prompt.
    delay_factor: float = 0.0
    #: Schedule step N+1 while the GPU still runs step N.
    async_scheduling: bool = False
    #: Encoder (multimodal) compute budget per step, in tokens.
    max_num_encoder_input_tokens: int = 0
    #: Encoder cache capacity in tokens.
    encoder_cache_size: int = 0
    #: LoRA cap on distinct adapters per batch (filled from LoRAConfig).
    max_loras: int = 0
    chunked_prefill_enabled: bool = field(init=False, default=True)

    def __post_init__(self):
        if self.max_model_len is None:
            self.max_model_len = 8192
        if self.max_num_seqs is None:
            # 288 GB of HBM keeps thousands of Llama-class sequences resident
            self.max_num_seqs = 1024
        if self.max_num_batched_tokens is None:
            self.max_num_batched_tokens = 16384 if self.enable_chunked_prefill else \
                max(self.max_model_len, 16384)
        if not self.enable_chunked_prefill and self.max_num_batched_tokens < self.max_model_len:
            raise ValueError("without chunked prefill max_num_batched_tokens must cover max_model_len")
        if self.max_num_batched_tokens < self.max_num_seqs:
            raise ValueError("max_num_batched_tokens must be >= max_num_seqs")
        if self.policy not in ("fcfs", "priority"):
            raise ValueError("policy must be fcfs or priority")
        self.chunked_prefill_enabled = self.enable_chunked_prefill
        if self.max_num_partial_prefills > 1 and self.long_prefill_token_threshold == 0:
            self.long_prefill_token_threshold = int(self.max_model_len * 0.04)
        if self.max_num_encoder_input_tokens == 0:
            self.max_num_encoder_input_tokens = self.max_num_batched_tokens
        if self.encoder_cache_size == 0:
            self.encoder_cache_size = self.max_num_batched_tokens


@dataclass
class DeviceConfig:
    #: "auto", "cuda" (= HIP device under PyTorch-ROCm) or "cpu".
    device: str = "auto"

    def __post_init__(self):
        if self.device == "auto":
            self.device = "cuda" if torch.cuda.is_available() else "cpu"
        if self.device in ("rocm", "hip", "gpu"):
            self.device = "cuda"
        self.device_type = self.device


@dataclass
class SpeculativeConfig:
    #: ngram | eagle | eagle3 | medusa | draft_model
    method: Optional[str] = None
    #: Draft checkpoint (eagle / medusa heads).
    model: Optional[str] = None
    #: Draft tokens proposed per step.
    num_speculative_tokens: int = 0
    #: Longest n-gram to match in the prompt.
    prompt_lookup_max: Optional[int] = None
    #: Shortest n-gram to match.
    prompt_lookup_min: Optional[int] = None
    #: Draft TP size (1 or the target's).
    draft_tensor_parallel_size: Optional[int] = None
    #: "rejection_sampler" is the only acceptance method.
    acceptance_method: str = "rejection_sampler"
    #: Disable speculation when the running batch is larger than this.
    disable_by_batch_size: Optional[int] = None

    def __post_init__(self):
        if self.method is None and self.model is None and self.num_speculative_tokens == 0:
            return
        if self.method is None:
            self.method = "ngram" if self.model in (None, "ngram", "[ngram]") else "eagle"
        if self.method not in ("ngram", "eagle", "eagle3", "medusa", "draft_model", "mtp", "deepseek_mtp", "mimo_mtp",
                               "mlp_speculator"):
            raise ValueError(f"unknown speculative method {self.method!r}")
        if self.num_speculative_tokens < 1:
            raise ValueError("num_speculative_tokens must be >= 1")
        if self.method == "ngram":
            if self.prompt_lookup_max is None:
                self.prompt_lookup_max = 5
            if self.prompt_lookup_min is None:
                self.prompt_lookup_min = min(2, self.prompt_lookup_max)
            if self.prompt_lookup_min > self.prompt_lookup_max or self.prompt_lookup_min < 1:
                raise ValueError("need 1 <= prompt_lookup_min <= prompt_lookup_max")
        elif self.model is None and self.method not in ("mtp", "deepseek_mtp", "mimo_mtp"):
            raise ValueError(f"method {self.method} needs a draft model path")

    @property
    def enabled(self) -> bool:
        return self.method is not None

    def use_eagle(self) -> bool:
        return self.method in ("eagle", "eagle3", "mtp", "deepseek_mtp", "mimo_mtp")


@dataclass
class LoRAConfig:
    #: Largest adapter rank supported.
    max_lora_rank: int = 16
    #: Adapters active in one batch.
    max_loras: int = 1
    #: Adapters kept in host memory.
    max_cpu_loras: Optional[int] = None
    #: dtype of adapter weights.
    lora_dtype: Union[str, torch.dtype] = "auto"
    #: Extra vocabulary rows adapters may add.
    lora_extra_vocab_size: int = 256
    #: Shard the LoRA A matrices too (fully sharded).
    fully_sharded_loras: bool = False
    #: Bias terms in adapters.
    bias_enabled: bool = False

    def __post_init__(self):
        if self.max_lora_rank not in (8, 16, 32, 64, 128, 256, 320, 512):
            raise ValueError("max_lora_rank must be one of 8,16,32,64,128,256,320,512")
        if self.max_loras < 1:
            raise ValueError("max_loras must be >= 1")
        if self.max_cpu_loras is None:
            self.max_cpu_loras = self.max_loras
        if self.max_cpu_loras < self.max_loras:
            raise ValueError("max_cpu_loras must be >= max_loras")


@dataclass
class ObservabilityConfig:
    #: Collector endpoint for request spans.
    otlp_traces_endpoint: Optional[str] = None
    #: Comma list of model|worker|all: which detailed timings to attach to spans.
    collect_detailed_traces: Optional[list[str]] = None
    #: Also export the deprecated metric names.
    show_hidden_metrics_for_version: Optional[str] = None

    def __post_init__(self):
        if isinstance(self.collect_detailed_traces, str):
            self.collect_detailed_traces = [x for x in self.collect_detailed_traces.split(",") if x]
        if self.collect_detailed_traces:
            bad = set(self.collect_detailed_traces) - {"model", "worker", "all"}
            if bad:
                raise ValueError(f"unknown detailed trace modules {sorted(bad)}")
            if not self.otlp_traces_endpoint:
                raise ValueError("collect_detailed_traces needs otlp_traces_endpoint")

    @property
    def collect_model_forward_time(self) -> bool:
        return bool(self.collect_detailed_traces) and bool({"model", "all"} & set(self.collect_detailed_traces))

    @property
    def collect_model_execute_time(self) -> bool:
        return bool(self.collect_detailed_traces) and bool({"worker", "all"} & set(self.collect_detailed_traces))


@dataclass
class KVTransferConfig:
    #: Connector class name registered in the connector factory.
    kv_connector: Optional[str] = None
    #: Engine id; random when unset.
    engine_id: Optional[str] = None
    #: Receive buffer budget in bytes.
    kv_buffer_size: float = 1e9
    #: Device of the transfer buffer.
    kv_buffer_device: str = "cuda"
    #: kv_producer | kv_consumer | kv_both
    kv_role: Optional[str] = None
    #: Rank of this instance among the transfer peers (producer 0, consumer 1).
    kv_rank: Optional[int] = None
    #: Number of instances taking part.
    kv_parallel_size: int = 1
    #: Side-channel address.
    kv_ip: str = "127.0.0.1"
    #: Side-channel port.
    kv_port: int = 14579
    #: Free-form connector options.
    kv_connector_extra_config: dict = field(default_factory=dict)
    #: Module path to import a connector class from.
    kv_connector_module_path: Optional[str] = None

    def __post_init__(self):
        if self.engine_id is None:
            self.engine_id = str(uuid.uuid4())
        # secret the exporting scheduler signs block lists with and its workers check them against (it
        # travels to the workers inside this config); ranks started separately (external launcher) must
        # be given the same "auth_key"
        self.kv_connector_extra_config.setdefault("auth_key", uuid.uuid4().hex)
        if self.kv_role not in (None, "kv_producer", "kv_consumer", "kv_both"):
            raise ValueError("kv_role must be kv_producer, kv_consumer or kv_both")
        if self.kv_connector is not None and self.kv_role is None:
            raise ValueError("kv_connector needs kv_role")

    @property
    def is_kv_transfer_instance(self) -> bool:
        return self.kv_connector is not None and self.kv_role is not None

    @property
    def is_kv_producer(self) -> bool:
        return self.kv_connector is not None and self.kv_role in ("kv_producer", "kv_both")

    @property
    def is_kv_consumer(self) -> bool:
        return self.kv_connector is not None and self.kv_role in ("kv_consumer", "kv_both")

    def get_from_extra_config(self, key, default):
        return self.kv_connector_extra_config.get(key, default)


@dataclass
class KVEventsConfig:
    #: Publish block stored/removed events.
    enable_kv_cache_events: bool = False
    #: null | socket
    publisher: str = "null"
    #: Address subscribers connect to.
    endpoint: str = "tcp://127.0.0.1:5557"
    #: Address of the replay service.
    replay_endpoint: Optional[str] = None
    #: Batches kept for replay.
    buffer_steps: int = 10_000
    #: High-water mark of the publish queue.
    hwm: int = 100_000
    #: Queue capacity between scheduler and publisher thread.
    max_queue_size: int = 100_000
    #: Topic prefix.
    topic: str = ""


@dataclass
class DecodingConfig:
    #: auto | regex_fsm | json_fsm (built-in finite-state backends)
    backend: str = "auto"
    #: Do not fall back to another backend on error.
    disable_fallback: bool = False
    #: Forbid whitespace in guided JSON.
    disable_any_whitespace: bool = False
    #: Reasoning parser name.
    reasoning_backend: str = ""


@dataclass
class CompilationConfig:
    """Graph-capture plan.  There is no tracing compiler here: fusion is done by
    hand in the HIP kernels, so this only carries the hipGraph settings."""
    #: 0 = eager, >=1 = capture decode steps as hipGraphs.
    level: int = 1
    #: Decode batch sizes to capture; derived from max_num_seqs when empty.
    cudagraph_capture_sizes: Optional[list[int]] = None
    #: Capture the full decode step (attention included) in one graph.
    full_cuda_graph: bool = True
    #: Number of warm-up runs before capture.
    cudagraph_num_of_warmups: int = 1
    #: Enable the sequence-parallel (reduce-scatter + all-gather) layer variant.
    enable_sequence_parallelism: bool = False
    #: Overlap TP collectives with the next GEMM on a side stream.
    enable_async_tp: bool = False
    max_capture_size: int = field(init=False, default=0)
    bs_to_padded_graph_size: list = field(init=False, default_factory=list, repr=False)

    def init_with_cudagraph_sizes(self, max_num_seqs: int) -> None:
        if self.cudagraph_capture_sizes is None:
            sizes = [1, 2, 4] + list(range(8, 257, 8)) + list(range(288, 1025, 32))
            self.cudagraph_capture_sizes = [s for s in sizes if s <= max_num_seqs] or [max_num_seqs]
            if max_num_seqs not in self.cudagraph_capture_sizes and max_num_seqs <= 2048:
                self.cudagraph_capture_sizes.append(max_num_seqs)
        self.cudagraph_capture_sizes = sorted(set(self.cudagraph_capture_sizes), reverse=True)
        self.max_capture_size = self.cudagraph_capture_sizes[0] if self.cudagraph_capture_sizes else 0
        table = [0] * (self.max_capture_size + 1)
        asc = sorted(self.cudagraph_capture_sizes)
        j = 0
        for bs in range(self.max_capture_size + 1):
            while asc[j] < bs:
                j += 1
            table[bs] = asc[j]
        self.bs_to_padded_graph_size = table


@dataclass
class VllmConfig:
    model_config: ModelConfig = field(default_factory=ModelConfig)
    cache_config: CacheConfig = field(default_factory=CacheConfig)
    parallel_config: ParallelConfig = field(default_factory=ParallelConfig)
    scheduler_config: SchedulerConfig = field(default_factory=SchedulerConfig)
    device_config: DeviceConfig = field(default_factory=DeviceConfig)
    load_config: LoadConfig = field(default_factory=LoadConfig)
    lora_config: Optional[LoRAConfig] = None
    speculative_config: Optional[SpeculativeConfig] = None
    decoding_config: DecodingConfig = field(default_factory=DecodingConfig)
    observability_config: ObservabilityConfig = field(default_factory=ObservabilityConfig)
    kv_transfer_config: Optional[KVTransferConfig] = None
    kv_events_config: Optional[KVEventsConfig] = None
    compilation_config: CompilationConfig = field(default_factory=CompilationConfig)
    additional_config: dict = field(default_factory=dict)
    instance_id: str = ""

    def __post_init__(self):
        mc, pc, sc, cc = self.model_config, self.parallel_config, self.scheduler_config, self.cache_config
        mc.verify_with_parallel_config(pc)
        sc.max_model_len = mc.max_model_len
        if sc.enable_chunked_prefill is False and sc.max_num_batched_tokens < mc.max_model_len:
            sc.max_num_batched_tokens = mc.max_model_len
        cc.sliding_window = mc.get_sliding_window()
        if mc.is_attention_free or mc.is_hybrid:
            cc.enable_prefix_caching = False      # recurrent state cannot be shared by prefix
        if self.lora_config is not None:
            sc.max_loras = self.lora_config.max_loras
        if self.speculative_config is not None and not self.speculative_config.enabled:
            self.speculative_config = None
        if self.speculative_config is not None:
            sc.num_lookahead_slots = self.speculative_config.num_speculative_tokens
            if self.speculative_config.model is None and self.speculative_config.method in ("mtp", "deepseek_mtp",
                                                                                            "mimo_mtp"):
                self.speculative_config.model = mc.model       # the prediction layers ship inside the target
        if self.device_config.device == "cpu" or mc.enforce_eager:
            self.compilation_config.level = 0
        if pc.token_parallel_size > 1:
            # per-rank row counts vary from step to step; decode graphs are captured all the same when the
            # ranks exchange rows through peer memory (the runner turns graphs off when they cannot)
            if pc.distributed_executor_backend == "mp":
                pc.distributed_executor_backend = "external_launcher"
        if sc.async_scheduling:
            # run-ahead scheduling keeps a step's sampled ids on the device of the process that schedules: the
            # worker must live in the engine-core process (one GPU, or one scheduler per rank under torchrun),
            # and nothing else may feed on a step's results before the next one starts
            why = None
            if pc.world_size > 1 and pc.distributed_executor_backend != "external_launcher":
                why = "workers run in other processes"
            elif pc.pipeline_parallel_size > 1:
                why = "pipeline stages already overlap through the batch queue"
            elif self.speculative_config is not None:
                why = "draft tokens of step N + 1 depend on the result of step N"
            elif self.kv_transfer_config is not None and self.kv_transfer_config.is_kv_transfer_instance:
                why = "KV transfer bookkeeping rides on every step's result"
            elif pc.token_parallel_size > 1:
                why = "token-parallel ranks share sampled ids at the end of every step"
            elif mc.runner_type == "pooling":
                why = "pooling models produce no tokens"
            if why is not None:
                logger.info("async_scheduling is off: %s", why)
                sc.async_scheduling = False
        if self.compilation_config.level > 0:
            self.compilation_config.init_with_cudagraph_sizes(sc.max_num_seqs)
        if self.kv_transfer_config is not None and not self.kv_transfer_config.is_kv_transfer_instance:
            self.kv_transfer_config = None
        if not self.instance_id:
            self.instance_id = uuid.uuid4().hex[:5]
        from .platforms import current_platform
        current_platform.check_and_update_config(self)

    def compute_hash(self) -> str:
        parts = [self.model_config.compute_hash(), self.cache_config.compute_hash(),
                 self.parallel_config.compute_hash(), envs.compute_hash()]
        return hashlib.sha256("|".join(parts).encode()).hexdigest()[:10]

    def pad_for_cudagraph(self, batch_size: int) -> int:
        return self.compilation_config.bs_to_padded_graph_size[batch_size]

    def __str__(self):
        mc, pc = self.model_config, self.parallel_config
        return (f"model={mc.model!r} dtype={mc.dtype} max_len={mc.max_model_len} quant={mc.quantization} "
                f"tp={pc.tensor_parallel_size} pp={pc.pipeline_parallel_size} dp={pc.data_parallel_size} "
                f"tknp={pc.token_parallel_size} ep={pc.enable_expert_parallel} "
                f"executor={pc.distributed_executor_backend} block={self.cache_config.block_size} "
                f"kv_dtype={self.cache_config.cache_dtype} prefix_cache={self.cache_config.enable_prefix_caching} "
                f"chunked_prefill={self.scheduler_config.enable_chunked_prefill} "
                f"graphs={self.compilation_config.level > 0}")


# ---- "current config" context: layers read it while being constructed ------
_current: Optional[VllmConfig] = None


@contextlib.contextmanager
def set_current_vllm_config(cfg: VllmConfig):
    global _current
    old, _current = _current, cfg
    try:
        yield
    finally:
        _current = old


def get_current_vllm_config() -> VllmConfig:
    if _current is None:
        raise RuntimeError("no VllmConfig is active; wrap model construction in set_current_vllm_config")
    return _current


def get_current_vllm_config_or_none() -> Optional[VllmConfig]:
    return _current
