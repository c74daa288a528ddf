"""Process-level switches read lazily from the environment.

Every knob is declared once in ``_REGISTRY`` as ``name -> (parser, default)``.
Attribute access (``envs.VLLM_LOGGING_LEVEL``) parses ``os.environ`` at the
moment of the read, so tests can monkeypatch the environment freely.

The ``VLLM_*`` spelling is kept for the knobs that have the same meaning as in
the reference (reference: vllm/envs.py:198-970) so that deployment scripts
keep working; knobs that only exist here use the ``VDA_*`` prefix.
"""
from __future__ import annotations

import hashlib
import os
import tempfile
from typing import Any, Callable


def _bool(v: str) -> bool:
    return v.strip().lower() in ("1", "true", "yes", "on")


def _opt_int(v: str):
    return int(v) if v not in ("", "None", "none") else None


def _path(v: str) -> str:
    return os.path.abspath(os.path.expanduser(v))


def _csv(v: str) -> list[str]:
    return [x.strip() for x in v.split(",") if x.strip()]


# name -> (parser, default-as-python-value)
_REGISTRY: dict[str, tuple[Callable[[str], Any], Any]] = {
    # ---- logging / debugging
    "VLLM_CONFIGURE_LOGGING": (_bool, True),
    "VLLM_LOGGING_LEVEL": (str.upper, "INFO"),
    "VLLM_LOGGING_CONFIG_PATH": (str, None),
    "VLLM_LOGGING_PREFIX": (str, ""),
    "VLLM_TRACE_FUNCTION": (_bool, False),
    "VLLM_LOG_BATCHSIZE_INTERVAL": (float, -1.0),
    "VLLM_COMPUTE_NANS_IN_LOGITS": (_bool, False),
    "VLLM_TORCH_PROFILER_DIR": (_path, None),
    "VLLM_DEBUG_DUMP_PATH": (_path, None),
    # ---- engine / process topology
    "VLLM_ENABLE_V1_MULTIPROCESSING": (_bool, True),
    "VLLM_WORKER_MULTIPROC_METHOD": (str, "spawn"),
    "VLLM_EXECUTE_MODEL_TIMEOUT_SECONDS": (int, 300),
    "VLLM_ENGINE_ITERATION_TIMEOUT_S": (int, 60),
    "VLLM_V1_OUTPUT_PROC_CHUNK_SIZE": (int, 128),
    "VLLM_RPC_BASE_PATH": (_path, tempfile.gettempdir()),
    "VLLM_HOST_IP": (str, ""),
    "VLLM_PORT": (_opt_int, None),
    "VLLM_RINGBUFFER_WARNING_INTERVAL": (int, 60),
    "VLLM_SLEEP_WHEN_IDLE": (_bool, False),
    "VLLM_MQ_MAX_CHUNK_BYTES_MB": (int, 16),
    "VLLM_KEEP_ALIVE_ON_ENGINE_DEATH": (_bool, False),
    # ---- data parallel
    "VLLM_DP_RANK": (int, 0),
    "VLLM_DP_RANK_LOCAL": (int, -1),
    "VLLM_DP_SIZE": (int, 1),
    "VLLM_DP_MASTER_IP": (str, "127.0.0.1"),
    "VLLM_DP_MASTER_PORT": (int, 0),
    # ---- pipeline parallel
    "VLLM_PP_LAYER_PARTITION": (str, None),
    # ---- kernels / device
    "VDA_OPS_LIBRARY": (_path, None),            # override path of libvda_ops.so
    "VDA_COMM_LIBRARY": (_path, None),           # override path of libvda_comm.so
    "VDA_FORCE_TORCH_OPS": (_bool, False),       # CPU/reference ops even on GPU (debug)
    "VDA_DECODE_SPLIT_TARGET_WGS": (int, 1024),  # split-KV until grid has this many WGs
    # 16-bit dense layers: rows served by the skinny MFMA kernel instead of the library GEMM.  0 = never:
    # hipBLASLt streams 16-bit weights at 85-95 % of the HBM floor from M = 1 on gfx950 (Llama-3-8B, 32 in /
    # 128 out: 0.699 s against 0.745 s at batch 8 and 0.698 s against 0.706 s at batch 1 with the skinny kernel).
    # The kernel keeps its role for 8-bit weights (VDA_FP8_W8A16_MAX_M) and packed 4-bit weights.
    "VDA_SKINNY_GEMM_MAX_M": (int, 0),
    # 16-bit dense layers with this many rows run the weight-streaming split-K MFMA kernel (ops/csrc/gemm_decode.hip)
    "VDA_DECODE_GEMM_MIN_M": (int, 33),
    "VDA_DECODE_GEMM_MAX_M": (int, 0),           # 0 = never (default: the tuned library GEMMs are faster, see docs/kernels.md)
    "VDA_DECODE_GEMM_WS_MB": (int, 128),         # fp32 split-K workspace per device
    "VDA_DENSE_GEMM_MAX_M": (int, 192),          # rows up to which 16-bit linears run the in-tree ping-pong MFMA GEMM (0 = never;
                                                 # at 256 rows the library's selected algorithms win inside the engine)
    "VDA_DENSE_GEMM_MIN_M": (int, 33),           # below: the library GEMM is as fast or faster (measured, graph-captured layer chains)
    "VDA_DENSE_GEMM_MAX_N": (int, 16384),        # wider weight matrices (gate/up) fill the chip without a K split: library
    "VDA_DENSE_GATE_UP_MAX_M": (int, 0),         # rows up to which a gated MLP runs SiLU * up inside the GEMM epilogue (0 = never:
                                                 # library GEMM + activation kernel measured faster on the wide gate/up matrix)
    "VDA_ASYNC_SCHEDULING": (int, 1),            # default of --async-scheduling when the flag is not given (the
                                                 # config turns it off again where it cannot apply)
    "VDA_REHEARSE_ON_ONE_GPU": (_bool, False),   # all ranks share cuda:0: gloo groups + peer-memory all-reduce (tests)
    "VDA_GEMM_SELECTIONS": (str, None),          # CSV of library-GEMM algorithm choices; "off" disables
    "VDA_DISABLE_SKINNY_GEMM": (_bool, False),
    "VDA_LORA_KERNEL_MAX_TOKENS": (int, 1024),   # multi-LoRA: batches up to this many rows use the fused shrink/expand kernels
    "VDA_SCAN_CHUNK_MIN_TOKENS": (int, 256),     # state-space scan: sequences of at least this many new tokens run chunk-parallel
    "VDA_W4_SKINNY_MAX_M": (int, 32),            # W4A16: rows served by the packed-weight skinny kernel
    "VDA_CB4_SKINNY_MAX_M": (int, 64),           # 4-bit code-book weights (NF4 / FP4 / NVFP4): rows served by the skinny kernel
    "VDA_GGUF_PACKED": (_bool, True),            # GGUF: keep 4- / 5- / 8-bit block codes packed (False: expand to 16 bit at load)
    "VDA_WQ_GEMM_MAX_M": (int, 512),             # W4A16 / W8A16: rows served by the packed-weight GEMM kernels
    "VDA_MLA": (_bool, True),                    # DeepSeek: latent KV cache + absorbed projections; 0 = expanded heads
    "VDA_INT4_WEIGHTS": (_bool, True),           # keep GPTQ / AWQ / fp4 / NF4 linear weights packed in memory; 0 = expand at load
    "VDA_FP8_BLOCK_NATIVE": (_bool, True),       # 128x128 block-scaled fp8 checkpoints: keep weights in fp8 (0: expand to 16 bit at load)
    "VDA_FP8_W8A16_MAX_M": (int, 64),            # fp8 linear: up to this M stream weights, beyond use fp8 MFMA GEMM
    "VDA_DISABLE_PEER_ALL2ALL": (_bool, False),  # expert / token parallel row exchange through torch.distributed instead of peer memory
    "VDA_EP_DISPATCH": (_bool, True),            # expert parallel without DP attention: route tokens to expert owners (False: replicate + mask + all-reduce)
    "VDA_MOE_GROUPED_MAX_PAIRS": (int, 2048),    # (token, slot) pairs served by the grouped skinny MoE kernel
    "VDA_MOE_GROUPED_MAX_PAIRS_FP8": (int, 192), # the same for e4m3 expert weights (beyond: fp8 MFMA tiles, W8A8)
    "VDA_EP_FP8_WIRE": (_bool, True),            # expert-parallel dispatch sends e4m3 rows + scale when the owner would quantise them anyway
    "VLLM_FUSED_MOE_CHUNK_SIZE": (int, 32768),
    "VLLM_MOE_DP_CHUNK_SIZE": (int, 256),
    "VLLM_ROCM_FP8_PADDING": (_bool, False),
    # ---- collectives
    "VLLM_DISABLE_CUSTOM_ALL_REDUCE": (_bool, False),
    "VDA_CUSTOM_AR_MAX_BYTES": (int, 16 * 1024 * 1024),
    "VDA_CUSTOM_AR_ONESHOT_BYTES": (int, 512 * 1024),
    # all-reduce + residual add + RMSNorm in one peer-memory kernel for batches of up to this many tokens (0: off)
    "VDA_FUSED_AR_NORM_MAX_TOKENS": (int, 512),
    "VDA_CUSTOM_AR_CHUNKED": (_bool, False),     # messages beyond one arena: chunk through it instead of RCCL
    "VDA_MOE_ROWS_FORM": (int, -1),              # grouped expert GEMM: 0 K-split kernel, 1 / 2 row-split kernel (64 / 128 tokens per pass), -1 by shape
    "VDA_CASCADE_MIN_PREFIX": (int, 256),        # cascade attention: shortest common prefix (tokens) worth a shared pass
    "VDA_CASCADE_MIN_SEQS": (int, 8),            # ... and fewest sequences in the step
    "VDA_ASYNC_TP_MIN_TOKENS": (int, 2048),      # async TP: batches of at least this many tokens run as 2 micro-batches
    "VDA_NATIVE_BLOCK_POOL": (_bool, True),      # scheduler free list in the host runtime library (0: numpy arrays)
    "VDA_PP_PEER_PIPE": (_bool, True),           # pipeline hops through the peer-memory pipe on one node (0: RCCL + metadata)
    "VDA_COMM_SPIN_TIMEOUT_MS": (int, 20000),
    "VLLM_ROCM_QUICK_REDUCE_QUANTIZATION": (str.upper, "NONE"),
    "VLLM_ROCM_QUICK_REDUCE_CAST_BF16_TO_FP16": (_bool, False),
    "VLLM_ROCM_QUICK_REDUCE_MAX_SIZE_BYTES_MB": (_opt_int, None),
    "VLLM_ALL2ALL_BACKEND": (str, "xgmi"),
    "VDA_RCCL_SO_PATH": (str, None),
    # ---- KV transfer
    "VLLM_NIXL_SIDE_CHANNEL_HOST": (str, "127.0.0.1"),
    "VLLM_NIXL_SIDE_CHANNEL_PORT": (int, 5557),
    "VLLM_NIXL_ABORT_REQUEST_TIMEOUT": (int, 120),
    "VDA_KV_IPC_CHUNK_GB": (float, 0.0),         # >0: split an exported KV cache into IPC regions of at most this size
    "VDA_KV_SIDE_CHANNEL_HOST": (str, "127.0.0.1"),
    "VDA_KV_SIDE_CHANNEL_PORT": (int, 5600),
    # ---- serving
    "VLLM_API_KEY": (str, None),
    "VLLM_HTTP_TIMEOUT_KEEP_ALIVE": (int, 5),
    "VLLM_SERVER_DEV_MODE": (_bool, False),
    "VLLM_ALLOW_LONG_MAX_MODEL_LEN": (_bool, False),
    "VLLM_ALLOW_RUNTIME_LORA_UPDATING": (_bool, False),
    "VLLM_USE_MODELSCOPE": (_bool, False),
    "VLLM_NO_USAGE_STATS": (_bool, True),
    "VLLM_DO_NOT_TRACK": (_bool, True),
    "VLLM_USAGE_SOURCE": (str, "production"),
    "VLLM_CACHE_ROOT": (_path, os.path.join(os.path.expanduser("~"), ".cache", "vda")),
    "VLLM_PLUGINS": (_csv, None),
    "VLLM_USE_V1": (_bool, True),
    "OTEL_EXPORTER_OTLP_TRACES_PROTOCOL": (str, "grpc"),
    # ---- torch.distributed launch variables (read-only convenience)
    "RANK": (int, 0),
    "LOCAL_RANK": (int, 0),
    "WORLD_SIZE": (int, 1),
    "MASTER_ADDR": (str, "127.0.0.1"),
    "MASTER_PORT": (int, 29500),
}

# Knobs that change numerics or graph shape: part of the config fingerprint.
_HASHED = (
    "VDA_FORCE_TORCH_OPS", "VDA_SKINNY_GEMM_MAX_M", "VDA_DISABLE_SKINNY_GEMM", "VDA_INT4_WEIGHTS", "VDA_MLA",
    "VDA_DECODE_GEMM_MIN_M", "VDA_DECODE_GEMM_MAX_M", "VDA_DENSE_GEMM_MAX_M", "VDA_DENSE_GEMM_MIN_M", "VDA_DENSE_GEMM_MAX_N", "VDA_DENSE_GATE_UP_MAX_M", "VDA_WQ_GEMM_MAX_M",
    "VLLM_ROCM_QUICK_REDUCE_QUANTIZATION", "VLLM_DISABLE_CUSTOM_ALL_REDUCE",
    "VLLM_FUSED_MOE_CHUNK_SIZE",
)


def __getattr__(name: str) -> Any:
    try:
        parser, default = _REGISTRY[name]
    except KeyError:
        raise AttributeError(f"unknown environment knob {name!r}") from None
    raw = os.environ.get(name)
    if raw is None:
        return default
    return parser(raw)


def __dir__() -> list[str]:
    return sorted(_REGISTRY)


def is_set(name: str) -> bool:
    if name not in _REGISTRY:
        raise AttributeError(f"unknown environment knob {name!r}")
    return name in os.environ


def snapshot() -> dict[str, Any]:
    """All knobs with their current values (for logs and engine dumps)."""
    return {k: __getattr__(k) for k in _REGISTRY}


def compute_hash() -> str:
    h = hashlib.sha256()
    for k in _HASHED:
        h.update(f"{k}={__getattr__(k)!r};".encode())
    return h.hexdigest()[:16]
