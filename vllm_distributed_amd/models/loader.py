"""Checkpoint loading: HF safetensors / .bin / .pt directories, dummy (random)
weights, and per-rank sharded-state save/load.

Reference: vllm/model_executor/model_loader/default_loader.py:30-283 (iterator
over checkpoint files, index de-duplication, strict all-weights-loaded check),
dummy_loader.py, sharded_state_loader.py, weight_utils.py.
"""
from __future__ import annotations

import contextlib
import fnmatch
import glob
import hashlib
import json
import os
import time
from typing import Iterable, Iterator, Optional

import torch
import torch.nn as nn

from ..config import VllmConfig, set_current_vllm_config
from ..logger import init_logger
from .registry import ModelRegistry

logger = init_logger(__name__)


@contextlib.contextmanager
def set_default_torch_dtype(dtype: torch.dtype):
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        yield
    finally:
        torch.set_default_dtype(old)


def _quant_config(vllm_config: VllmConfig):
    q = vllm_config.model_config.quantization
    if q is None and not str(vllm_config.model_config.model).endswith(".gguf"):
        return None
    from ..layers.quant import get_quant_config
    return get_quant_config(vllm_config)


def initialize_model(vllm_config: VllmConfig, prefix: str = "") -> nn.Module:
    mc = vllm_config.model_config
    impl = getattr(mc, "model_impl", "auto")
    if impl not in ("auto", "vllm", "transformers"):
        raise ValueError(f"model_impl must be auto, vllm or transformers, not {impl!r}")
    archs = mc.architectures
    if impl == "transformers" or (impl == "auto" and not any(ModelRegistry.is_supported(a) for a in archs)
                                  and ModelRegistry.is_text_generation_model(archs)):
        if impl == "auto":
            logger.warning("no native implementation of %s: running it through the transformers fallback", archs)
        archs = ["TransformersForCausalLM"]
    cls, arch = ModelRegistry.resolve_model_cls(archs)
    vllm_config.quant_config = _quant_config(vllm_config)
    if vllm_config.parallel_config.token_parallel_size > 1:
        from ..parallel.state import is_tknp_root
        worker_cls = getattr(cls, "tknp_worker_cls", None)
        if worker_cls is None:
            raise ValueError(f"{arch} does not support token parallelism")
        if not is_tknp_root():
            # non-root token-parallel ranks hold no weights: attention + KV cache only
            with set_current_vllm_config(vllm_config):
                return worker_cls(vllm_config=vllm_config, prefix=prefix)
    with set_current_vllm_config(vllm_config):
        return cls(vllm_config=vllm_config, prefix=prefix)


# ---- weight iterators ---------------------------------------------------------
def _filter_by_index(files: list[str], folder: str, index_name: str) -> list[str]:
    idx = os.path.join(folder, index_name)
    if not os.path.isfile(idx):
        return files
    with open(idx) as f:
        wm = json.load(f).get("weight_map", {})
    keep = {os.path.join(folder, v) for v in wm.values()}
    return [f for f in files if f in keep] or files


def find_weight_files(folder: str, load_format: str, ignore_patterns: Optional[list[str]] = None) -> tuple[list[str], bool]:
    if not os.path.isdir(folder):
        raise FileNotFoundError(f"{folder} is not a checkpoint directory (no hub access on this machine); "
                                "use load_format='dummy' for random weights")
    patterns = {"auto": ["*.safetensors", "*.bin", "*.pt"], "safetensors": ["*.safetensors"],
                "pt": ["*.pt", "*.bin"], "bin": ["*.bin"], "npcache": ["*.bin"]}[load_format]
    for pat in patterns:
        files = sorted(glob.glob(os.path.join(folder, pat)))
        files = [f for f in files if not any(fnmatch.fnmatch(os.path.relpath(f, folder), ip)
                                             for ip in (ignore_patterns or []))]
        files = [f for f in files if os.path.basename(f) not in ("training_args.bin", "optimizer.bin",
                                                                 "optimizer.pt", "scheduler.pt", "scaler.pt")]
        if files:
            st = pat == "*.safetensors"
            if st:
                files = _filter_by_index(files, folder, "model.safetensors.index.json")
            else:
                files = _filter_by_index(files, folder, "pytorch_model.bin.index.json")
            return files, st
    raise FileNotFoundError(f"no weight files matching {patterns} in {folder}")


def safetensors_iterator(files: list[str]) -> Iterator[tuple[str, torch.Tensor]]:
    from safetensors import safe_open
    for path in files:
        with safe_open(path, framework="pt", device="cpu") as f:
            for name in f.keys():
                yield name, f.get_tensor(name)


_ST_DTYPES = {"F64": torch.float64, "F32": torch.float32, "F16": torch.float16, "BF16": torch.bfloat16,
              "I64": torch.int64, "I32": torch.int32, "I16": torch.int16, "I8": torch.int8, "U8": torch.uint8,
              "BOOL": torch.bool, "F8_E4M3": torch.float8_e4m3fn, "F8_E5M2": torch.float8_e5m2}


def read_safetensors_header(path: str) -> tuple[dict, int]:
    """(tensor table {name: {dtype, shape, data_offsets}}, byte offset of the data section)."""
    with open(path, "rb") as f:
        n = int.from_bytes(f.read(8), "little")
        if n <= 0 or n > 1 << 30:
            raise ValueError(f"{path}: not a safetensors file")
        table = json.loads(f.read(n))
    table.pop("__metadata__", None)
    return table, 8 + n


def streamed_safetensors_iterator(files: list[str], concurrency: int = 16, window_bytes: int = 1 << 30,
                                  pin: bool = False, zero_copy: bool = False) -> Iterator[tuple[str, torch.Tensor]]:
    """Tensors of safetensors files read by the native range reader (runtime/csrc/weight_reader.cpp): a window of
    tensors at a time, by ``concurrency`` threads, into one of two (pinned) host buffers, the next window being read
    while the caller consumes this one.

    Lifetime: the read of window k + 1 is started when window k is handed out and lands in the buffer that held
    window k - 1, so a VIEW into a buffer is only good until the first tensor of the next window is requested.  By
    default every tensor is therefore yielded as its own copy, safe to keep for as long as the caller likes (weight loaders park tensors: packed-weight triples, per-module lists).  A caller
    that consumes each tensor before asking for the next may pass ``zero_copy=True`` and receive views.
    Role of the reference's fastsafetensors / Run:ai streamer loaders
    (vllm/model_executor/model_loader/weight_utils.py:480-560, runai_streamer_loader.py)."""
    import ctypes
    from concurrent.futures import ThreadPoolExecutor
    from ..runtime.shm_ring import host_library
    lib = host_library()
    lib.vda_read_ranges.restype = ctypes.c_int
    lib.vda_read_ranges.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_int]
    windows: list[tuple[str, list]] = []             # (file, [(name, dtype, shape, file offset, nbytes, window offset)])
    for path in files:
        table, base = read_safetensors_header(path)
        items = sorted(table.items(), key=lambda kv: kv[1]["data_offsets"][0])
        cur, used = [], 0
        for name, meta in items:
            lo, hi = meta["data_offsets"]
            if meta["dtype"] not in _ST_DTYPES:
                raise ValueError(f"{path}: tensor {name} has unsupported dtype {meta['dtype']}")
            nb = hi - lo
            off = (used + 63) // 64 * 64
            if cur and off + nb > window_bytes:
                windows.append((path, cur))
                cur, off = [], 0
            cur.append((name, _ST_DTYPES[meta["dtype"]], meta["shape"], base + lo, nb, off))
            used = off + nb
        if cur:
            windows.append((path, cur))
    if not windows:
        return
    cap = max(max(it[5] + it[4] for it in w) for _, w in windows)
    bufs = [torch.empty(max(cap, 64), dtype=torch.uint8, pin_memory=pin) for _ in range(min(2, len(windows)))]

    def fill(k: int):
        path, items = windows[k]
        buf = bufs[k % len(bufs)]
        n = len(items)
        offs = (ctypes.c_uint64 * n)(*[it[3] for it in items])
        sizes = (ctypes.c_uint64 * n)(*[it[4] for it in items])
        dsts = (ctypes.c_void_p * n)(*[buf.data_ptr() + it[5] for it in items])
        rc = lib.vda_read_ranges(path.encode(), n, offs, sizes, dsts, int(concurrency))
        if rc != 0:
            raise OSError(-rc, f"reading {path}: {os.strerror(-rc)}")
        return buf

    with ThreadPoolExecutor(max_workers=1, thread_name_prefix="weights") as pool:
        fut = pool.submit(fill, 0)
        for k, (_, items) in enumerate(windows):
            buf = fut.result()
            if k + 1 < len(windows):
                fut = pool.submit(fill, k + 1)
            for name, dtype, shape, _, nb, off in items:
                t = buf[off:off + nb].view(dtype).view(shape)
                yield name, (t if zero_copy else t.clone())


def pt_iterator(files: list[str]) -> Iterator[tuple[str, torch.Tensor]]:
    for path in files:
        state = torch.load(path, map_location="cpu", weights_only=True)
        if "state_dict" in state and isinstance(state["state_dict"], dict):
            state = state["state_dict"]
        elif isinstance(state.get("model_key"), str) and isinstance(state.get(state["model_key"]), dict):
            state = state[state["model_key"]]          # fairseq2: {"model_key": name, name: state_dict}
        yield from state.items()
        del state


def initialize_dummy_weights(model: nn.Module, low: float = -1e-3, high: float = 1e-3, seed: int = 1234) -> None:
    """Small uniform random weights, generated on the parameter's device so that
    a 70B model is filled in seconds; norm weights become ~1 so activations stay finite."""
    for i, (name, p) in enumerate(model.state_dict().items()):
        if not torch.is_floating_point(p):
            continue
        g = torch.Generator(device=p.device)
        g.manual_seed(seed + i)
        if p.dtype.itemsize < 2:       # fp8: fill through a wider dtype
            tmp = torch.empty(p.shape, dtype=torch.float16, device=p.device).uniform_(low, high, generator=g)
            p.copy_(tmp.to(p.dtype))
        elif name.endswith("norm.weight") or "layernorm.weight" in name or name.endswith("ln_f.weight"):
            p.uniform_(1 - 1e-3, 1 + 1e-3, generator=g)
        elif "scale" in name.rsplit(".", 1)[-1]:
            p.fill_(1.0)
        else:
            p.uniform_(low, high, generator=g)


# ---- loaders ----------------------------------------------------------------------
class BaseModelLoader:
    def __init__(self, load_config):
        self.load_config = load_config

    def load_weights(self, model: nn.Module, vllm_config: VllmConfig) -> None:
        raise NotImplementedError

    def load_model(self, vllm_config: VllmConfig, device: Optional[torch.device] = None) -> nn.Module:
        mc = vllm_config.model_config
        device = device or torch.device(vllm_config.device_config.device)
        t0 = time.time()
        with set_default_torch_dtype(mc.dtype), device:
            model = initialize_model(vllm_config)
        if not getattr(model, "is_tknp_worker", False):
            self.load_weights(model, vllm_config)
        for m in model.modules():
            qm = getattr(m, "quant_method", None)
            if qm is not None and hasattr(qm, "process_weights_after_loading"):
                qm.process_weights_after_loading(m)
        logger.info("model %s ready in %.1f s (%s)", mc.model, time.time() - t0, type(self).__name__)
        return model.eval()


class DefaultModelLoader(BaseModelLoader):
    def get_all_weights(self, vllm_config: VllmConfig) -> Iterable[tuple[str, torch.Tensor]]:
        folder = vllm_config.model_config.model
        if folder.endswith(".gguf") and os.path.isfile(folder):
            from ..layers.quant import get_quant_config
            from .gguf import GGUFPackedConfig, gguf_weights_iterator
            qc = get_quant_config(vllm_config)
            return gguf_weights_iterator(folder, vllm_config.model_config.dtype,
                                         qc if isinstance(qc, GGUFPackedConfig) else None)
        fmt = self.load_config.load_format
        streamed = fmt in ("fastsafetensors", "runai_streamer")
        files, is_st = find_weight_files(folder, "safetensors" if streamed else fmt, self.load_config.ignore_patterns)
        if streamed:
            xc = self.load_config.model_loader_extra_config or {}
            return streamed_safetensors_iterator(
                files, int(xc.get("concurrency", 16)), int(xc.get("memory_limit", 1 << 30)),
                pin=vllm_config.device_config.device != "cpu")
        return safetensors_iterator(files) if is_st else pt_iterator(files)

    def load_weights(self, model: nn.Module, vllm_config: VllmConfig) -> None:
        from .dequant import LoadTimeDequant
        weights = self.get_all_weights(vllm_config)
        qc = getattr(vllm_config, "quant_config", None)
        if isinstance(qc, LoadTimeDequant):
            weights = qc.wrap(weights, vllm_config.model_config.dtype)
        loaded = model.load_weights(weights)
        if loaded is not None and (vllm_config.model_config.quantization is None or isinstance(qc, LoadTimeDequant)):
            # strict: every parameter must have come from the checkpoint
            missing = {n for n, _ in model.named_parameters()} - set(loaded)
            if missing:
                raise ValueError(f"these parameters were not initialised from the checkpoint: {sorted(missing)[:8]}"
                                 f"{' ...' if len(missing) > 8 else ''}")


class DummyModelLoader(BaseModelLoader):
    def load_weights(self, model, vllm_config):
        initialize_dummy_weights(model, seed=1234 + vllm_config.model_config.seed)


class ShardedStateLoader(BaseModelLoader):
    """Each rank reads back exactly the tensors it wrote with :meth:`save_model`
    (already TP-sharded and post-processed): no re-slicing at load time."""
    PATTERN = "model-rank-{rank}-part-{part}.safetensors"

    def load_weights(self, model, vllm_config):
        from safetensors.torch import load_file
        from ..parallel.state import get_world_group
        rank = get_world_group().rank
        pat = (self.load_config.model_loader_extra_config or {}).get("pattern", self.PATTERN)
        files = sorted(glob.glob(os.path.join(vllm_config.model_config.model, pat.format(rank=rank, part="*"))))
        if not files:
            raise FileNotFoundError(f"no sharded-state files for rank {rank} in {vllm_config.model_config.model}")
        state = model.state_dict()
        for f in files:
            for k, t in load_file(f).items():
                if k not in state:
                    raise KeyError(f"unexpected tensor {k} in {f}")
                state[k].copy_(t)
                state.pop(k)
        if state:
            raise ValueError(f"tensors missing from the sharded checkpoint: {sorted(state)[:8]}")

    @staticmethod
    def save_model(model: nn.Module, path: str, pattern: Optional[str] = None, max_size: Optional[int] = None) -> None:
        from safetensors.torch import save_file
        from ..parallel.state import get_world_group
        rank = get_world_group().rank
        pattern = pattern or ShardedStateLoader.PATTERN
        os.makedirs(path, exist_ok=True)
        part, size, cur = 0, 0, {}
        seen_ptrs = set()
        for k, t in model.state_dict().items():
            if t.data_ptr() in seen_ptrs:      # tied weights are stored once
                continue
            seen_ptrs.add(t.data_ptr())
            n = t.numel() * t.element_size()
            if max_size is not None and size + n > max_size and cur:
                save_file(cur, os.path.join(path, pattern.format(rank=rank, part=part)))
                part, size, cur = part + 1, 0, {}
            cur[k] = t.detach().cpu().contiguous()
            size += n
        if cur:
            save_file(cur, os.path.join(path, pattern.format(rank=rank, part=part)))


class TensorizerLoader(BaseModelLoader):
    """One file per rank holding the model exactly as it sits in memory after loading (tensor-parallel shards cut,
    fused projections stacked, quantised weights processed): start-up is a straight streamed read by the native
    range reader into the parameters, with no per-tensor routing and no re-slicing, and an optional SHA-256 check
    of every tensor.  ``tensorizer_uri`` may contain ``%03d`` (or ``{rank}``) for the rank; default
    ``<model dir>/model-rank-%03d.tensors``.  The container is safetensors' (8-byte header length, JSON table, raw
    data), so other tools can read the files too.  Layers whose post-load processing changes parameter shapes
    (some quantisation methods) keep the default loader: the shapes in the file would not match a fresh model.
    Reference: vllm/model_executor/model_loader/tensorizer.py (TensorizerConfig :163-260, serialize_vllm_model
    :625-660, deserialisation :491-560) and tensorizer_loader.py; encryption (libsodium) is not available."""

    def _uri(self, vllm_config, rank: int) -> str:
        extra = self.load_config.model_loader_extra_config or {}
        if extra.get("encryption_keyfile"):
            raise ValueError("encrypted tensorizer files are not supported in this build")
        uri = extra.get("tensorizer_uri") or os.path.join(vllm_config.model_config.model, "model-rank-%03d.tensors")
        return uri.replace("{rank}", str(rank)) % rank if "%" in uri else uri.replace("{rank}", str(rank))

    def load_weights(self, model, vllm_config):
        from ..parallel.state import get_world_group
        extra = self.load_config.model_loader_extra_config or {}
        path = self._uri(vllm_config, get_world_group().rank)
        if not os.path.isfile(path):
            raise FileNotFoundError(f"{path}: write it with TensorizerLoader.tensorize_model(model, path)")
        want = {}
        if extra.get("verify_hash"):
            with open(path, "rb") as f:
                want = json.loads(f.read(int.from_bytes(f.read(8), "little"))).get("__metadata__") or {}
            if not want:
                raise ValueError(f"{path} carries no tensor hashes to verify")
        state = model.state_dict()
        for name, t in streamed_safetensors_iterator([path], concurrency=int(extra.get("num_readers", 16))):
            if name not in state:
                raise KeyError(f"unexpected tensor {name} in {path}")
            if want:
                got = hashlib.sha256(t.contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()
                if want.get(name) != got:
                    raise ValueError(f"{path}: tensor {name} does not match its recorded hash")
            state.pop(name).copy_(t)
        ptrs = {}
        for k, p in model.state_dict().items():         # tied tensors were written once
            ptrs.setdefault(p.data_ptr(), k)
        missing = [k for k in state if ptrs.get(model.state_dict()[k].data_ptr()) == k]
        if missing:
            raise ValueError(f"tensors missing from {path}: {sorted(missing)[:8]}")

    @staticmethod
    def tensorize_model(model: nn.Module, path: str) -> None:
        """Write this rank's model state to ``path`` (with per-tensor hashes in the file's metadata)."""
        from safetensors.torch import save_file
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        out, meta, seen = {}, {}, set()
        for k, t in model.state_dict().items():
            if t.data_ptr() in seen:
                continue
            seen.add(t.data_ptr())
            c = t.detach().cpu().contiguous()
            out[k] = c
            meta[k] = hashlib.sha256(c.view(torch.uint8).numpy().tobytes()).hexdigest()
        save_file(out, path, metadata=meta)


def get_model_loader(load_config) -> BaseModelLoader:
    fmt = load_config.load_format
    if fmt == "tensorizer":
        return TensorizerLoader(load_config)
    if fmt == "dummy":
        return DummyModelLoader(load_config)
    if fmt == "sharded_state":
        return ShardedStateLoader(load_config)
    return DefaultModelLoader(load_config)


def get_model(vllm_config: VllmConfig, device=None) -> nn.Module:
    # rotary modules are shared between the layers of ONE model; a second model in this process (a draft
    # model, a CPU engine beside a GPU one) must not pick up buffers that the first one moved to its device
    from ..layers import rotary
    rotary._CACHE.clear()
    return get_model_loader(vllm_config.load_config).load_model(vllm_config, device)
