"""Multi-LoRA: many adapters share one base model, each request picks one.

Every adapted linear layer keeps stacked device buffers
    A [max_loras, r_max, in_shard]        B [max_loras, out_shard, r_max]   (alpha/r folded into B)
indexed by adapter *slot*.  Each step the runner gives the slot of every token; a layer then adds
``(x_s A_s^T) B_s^T`` for the tokens of each active slot s (tokens are grouped by slot once per
step, so a layer does two skinny GEMMs per active adapter and nothing when no adapter is active).
Tensor parallel: column-parallel layers shard B by output rows, row-parallel layers shard A by input
columns and add their partial product before the layer's all-reduce.  Adapters are loaded from PEFT
checkpoints (adapter_config.json + adapter_model.safetensors), kept in a host LRU
(``max_cpu_loras``) and paged into device slots (``max_loras``) on demand.
The token embedding and the output head are adapted too: an embedding adapter (``lora_embedding_A``
[r, vocab], ``lora_embedding_B`` [hidden, r]) adds ``A^T[ids] B^T`` to the looked-up rows (A is kept
whole on every rank: a row gather needs no sharding), and the head takes an ordinary low-rank update
with B sharded by vocabulary rows; the rows fed to the head are a subset of the step's tokens, so the
runner names them (``head_rows``) and the manager regroups them by adapter.
Reference: vllm/lora/{models.py (LoRAModelManager :312-720, LRUCache), layers.py, worker_manager.py,
punica_wrapper/}, vllm/v1/worker/lora_model_runner_mixin.py.
"""
from __future__ import annotations

import contextlib
import json
import math
import os
import re
from collections import OrderedDict
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..layers.basic import ParallelLMHead, VocabParallelEmbedding
from ..layers.linear import (ColumnParallelLinear, LinearBase, MergedColumnParallelLinear, QKVParallelLinear,
                             RowParallelLinear)
from .. import envs, ops
from ..logger import init_logger

logger = init_logger(__name__)


class LoRAModel:
    """One adapter on the host: {module path: (A [r, in], B [out, r])} with scaling folded into B."""

    def __init__(self, lora_id: int, rank: int, weights: dict[str, tuple[torch.Tensor, torch.Tensor]]):
        self.id, self.rank, self.weights = lora_id, rank, weights

    @classmethod
    def from_checkpoint(cls, path: str, lora_id: int, max_rank: int, dtype: torch.dtype,
                        expected_modules: Optional[set[str]] = None) -> "LoRAModel":
        cfg_path = os.path.join(path, "adapter_config.json")
        if not os.path.isfile(cfg_path):
            raise ValueError(f"{path} has no adapter_config.json")
        with open(cfg_path) as f:
            cfg = json.load(f)
        r, alpha = int(cfg["r"]), float(cfg.get("lora_alpha", cfg["r"]))
        if r > max_rank:
            raise ValueError(f"LoRA rank {r} exceeds max_lora_rank {max_rank}")
        if cfg.get("use_dora") or cfg.get("modules_to_save"):
            raise ValueError("DoRA adapters and modules_to_save are not supported")
        if any(os.path.isfile(os.path.join(path, f)) for f in ("new_embeddings.safetensors", "new_embeddings.bin")):
            raise ValueError("adapters that extend the vocabulary (new_embeddings) are not supported")
        scale = alpha / (math.sqrt(r) if cfg.get("use_rslora") else r)
        st = os.path.join(path, "adapter_model.safetensors")
        if os.path.isfile(st):
            from safetensors.torch import load_file
            tensors = load_file(st)
        else:
            binp = os.path.join(path, "adapter_model.bin")
            if not os.path.isfile(binp):
                raise ValueError(f"{path} has no adapter_model.safetensors / adapter_model.bin")
            tensors = torch.load(binp, map_location="cpu", weights_only=True)
        pairs: dict[str, dict[str, torch.Tensor]] = {}
        for name, t in tensors.items():
            m = re.match(r"(?:base_model\.model\.)?(.*)\.lora_([AB])(?:\.default)?\.weight$", name) or \
                re.match(r"(?:base_model\.model\.)?(.*)\.lora_embedding_([AB])(?:\.default)?$", name)
            if not m:
                continue
            pairs.setdefault(m.group(1), {})[m.group(2)] = t
        weights = {}
        for mod, ab in pairs.items():
            if "A" not in ab or "B" not in ab:
                raise ValueError(f"adapter module {mod} lacks lora_A or lora_B")
            leaf = mod.rsplit(".", 1)[-1]
            if expected_modules is not None and leaf not in expected_modules:
                raise ValueError(f"adapter targets module {leaf!r} which this model does not expose for LoRA")
            weights[mod] = (ab["A"].to(dtype).contiguous(), (ab["B"].float() * scale).to(dtype).contiguous())
        return cls(lora_id, r, weights)


class _LoRALinearMethod:
    """Wraps a layer's quant method: base GEMM + the active adapters' low-rank update."""

    def __init__(self, base, manager: "LoRAManager"):
        self.base, self.manager = base, manager

    def __getattr__(self, k):
        return getattr(self.base, k)

    def apply(self, layer: nn.Module, x: torch.Tensor, bias=None) -> torch.Tensor:
        y = self.base.apply(layer, x, bias)
        groups = self.manager.active_groups
        if not groups or not any(layer.lora_has[slot] for slot, _ in groups):
            return y
        A, B = layer.lora_A, layer.lora_B
        x2 = x.reshape(-1, x.shape[-1])
        y2 = y.view(-1, y.shape[-1])
        ts = self.manager.token_slots
        if ts is not None and x2.shape[0] <= envs.VDA_LORA_KERNEL_MAX_TOKENS and ops.lora_update_ok(x2, A, B) \
                and ts.shape[0] == x2.shape[0]:
            # decode-sized batch: one shrink + one expand launch for all adapters (slots without weights
            # for this layer hold zeros)
            ops.lora_update(y2, x2 if x2.stride(1) == 1 else x2.contiguous(), A, B, ts)
            return y
        for slot, idx in groups:
            if not layer.lora_has[slot]:
                continue
            xs = x2 if idx is None else x2.index_select(0, idx)
            d = torch.mm(torch.mm(xs, A[slot].t()), B[slot].t())
            if idx is None:
                y2 += d
            else:
                y2.index_add_(0, idx, d)
        return y


class LoRAManager:
    def __init__(self, model: nn.Module, vllm_config, device: torch.device):
        lc = vllm_config.lora_config
        self.model, self.device = model, device
        self.max_loras, self.max_cpu, self.r_max = lc.max_loras, lc.max_cpu_loras, lc.max_lora_rank
        self.dtype = vllm_config.model_config.dtype if lc.lora_dtype == "auto" else lc.lora_dtype
        if not getattr(model, "supports_lora", False):
            raise ValueError(f"{type(model).__name__} does not support LoRA")
        self.packed = dict(getattr(model, "packed_modules_mapping", {}))
        self.cpu_cache: "OrderedDict[int, LoRAModel]" = OrderedDict()
        self.pinned: set[int] = set()
        self.slot_of: "OrderedDict[int, int]" = OrderedDict()        # adapter id -> device slot (LRU order)
        self.free_slots = list(range(self.max_loras))
        self.active_groups: list[tuple[int, Optional[torch.Tensor]]] = []
        self.token_slots: Optional[torch.Tensor] = None       # int32 [T] adapter slot of every token row (-1: none)
        self.step_slots: Optional[np.ndarray] = None          # host copy of the step's token slots (head_rows)
        self.layers: dict[str, LinearBase] = {}
        self.embeddings: dict[str, VocabParallelEmbedding] = {}
        self.heads: dict[str, ParallelLMHead] = {}
        for name, m in model.named_modules():
            if isinstance(m, LinearBase) and "lm_head" not in name and isinstance(
                    m, (ColumnParallelLinear, RowParallelLinear)):
                self._wrap(name, m)
            elif isinstance(m, ParallelLMHead):
                self._wrap_head(name, m)
            elif isinstance(m, VocabParallelEmbedding):
                self._wrap_embedding(name, m)
        self.expected = set()
        for name in self.layers:
            leaf = name.rsplit(".", 1)[-1]
            self.expected.update(self.packed.get(leaf, [leaf]))
        self.expected.update(n.rsplit(".", 1)[-1] for n in list(self.embeddings) + list(self.heads))
        logger.info("LoRA enabled on %d layers: %d device slots, rank <= %d", len(self.layers), self.max_loras,
                    self.r_max)

    def _wrap(self, name: str, m: LinearBase) -> None:
        w = m.weight
        out_shard, in_shard = w.shape
        m.register_buffer("lora_A", torch.zeros(self.max_loras, self.r_max, in_shard, dtype=self.dtype,
                                                device=self.device), persistent=False)
        m.register_buffer("lora_B", torch.zeros(self.max_loras, out_shard, self.r_max, dtype=self.dtype,
                                                device=self.device), persistent=False)
        m.lora_has = [False] * self.max_loras
        m.quant_method = _LoRALinearMethod(m.quant_method, self)
        self.layers[name] = m

    def _wrap_head(self, name: str, m: ParallelLMHead) -> None:
        """Output head: A whole [r, hidden], B sharded like the head's vocabulary rows."""
        m.register_buffer("lora_A", torch.zeros(self.max_loras, self.r_max, m.embedding_dim, dtype=self.dtype,
                                                device=self.device), persistent=False)
        m.register_buffer("lora_B", torch.zeros(self.max_loras, m.per_rank, self.r_max, dtype=self.dtype,
                                                device=self.device), persistent=False)
        m.lora_has = [False] * self.max_loras
        m.quant_method = _LoRALinearMethod(m.quant_method, self)
        self.heads[name] = m

    def _wrap_embedding(self, name: str, m: VocabParallelEmbedding) -> None:
        """Token embedding: A^T whole [vocab, r] (gathered by token id), B [hidden, r]."""
        m.register_buffer("lora_A", torch.zeros(self.max_loras, m.num_embeddings, self.r_max, dtype=self.dtype,
                                                device=self.device), persistent=False)
        m.register_buffer("lora_B", torch.zeros(self.max_loras, m.embedding_dim, self.r_max, dtype=self.dtype,
                                                device=self.device), persistent=False)
        m.lora_has = [False] * self.max_loras

        def add_update(mod, args, out):
            groups = self.active_groups
            if not groups:
                return out
            ids = args[0].reshape(-1)
            o2 = out.view(-1, out.shape[-1])
            if ids.shape[0] != o2.shape[0]:
                return out
            for slot, idx in groups:
                if not mod.lora_has[slot]:
                    continue
                sel = ids if idx is None else ids.index_select(0, idx)
                d = torch.mm(F.embedding(sel, mod.lora_A[slot]), mod.lora_B[slot].t()).to(o2.dtype)
                if idx is None:
                    o2 += d
                else:
                    o2.index_add_(0, idx, d)
            return out
        m.register_forward_hook(add_update)
        self.embeddings[name] = m

    def _place_vocab_layers(self, lora: LoRAModel, slot: int) -> None:
        for name, m in self.embeddings.items():
            ab = lora.weights.get(name)
            m.lora_has[slot] = ab is not None
            m.lora_A[slot].zero_()
            m.lora_B[slot].zero_()
            if ab is not None:
                A, B = ab                                   # [r, vocab], [hidden, r]
                r, v = A.shape[0], min(A.shape[1], m.num_embeddings)
                m.lora_A[slot][:v, :r].copy_(A[:, :v].t())
                m.lora_B[slot][:, :r].copy_(B)
        for name, m in self.heads.items():
            ab = lora.weights.get(name)
            m.lora_has[slot] = ab is not None
            m.lora_A[slot].zero_()
            m.lora_B[slot].zero_()
            if ab is not None:
                A, B = ab                                   # [r, hidden], [vocab, r]
                r = A.shape[0]
                lo, hi = m.vocab_start, min(m.vocab_end, B.shape[0])
                m.lora_A[slot][:r].copy_(A)
                if hi > lo:
                    m.lora_B[slot][: hi - lo, :r].copy_(B[lo:hi])

    # ---- adapter placement ------------------------------------------------------------------------
    def _shard_into(self, layer: LinearBase, slot: int, parts: list[Optional[tuple[torch.Tensor, torch.Tensor]]]) -> None:
        """``parts``: one (A, B) per logical sub-projection of the layer (None = not adapted)."""
        A_buf, B_buf = layer.lora_A[slot], layer.lora_B[slot]
        A_buf.zero_()
        B_buf.zero_()
        tp, rank = getattr(layer, "tp_size", 1), getattr(layer, "tp_rank", 0)
        if isinstance(layer, RowParallelLinear):
            A, B = parts[0]
            r = A.shape[0]
            n = layer.input_size_per_partition
            A_buf[:r].copy_(A[:, rank * n:(rank + 1) * n])
            B_buf[:, :r].copy_(B)
            return
        # column-parallel family: rows of B follow the layer's output layout; several sub-projections
        # share the rank dimension by stacking their A matrices (block structure in B)
        if isinstance(layer, QKVParallelLinear):
            offs = layer._offsets()
            slices = []
            for sid in ("q", "k", "v"):
                off, size = offs[sid]
                src_rank = rank if sid == "q" else rank // layer.num_kv_head_replicas
                slices.append((off, size, src_rank * size))
        elif isinstance(layer, MergedColumnParallelLinear):
            slices, off = [], 0
            for s in layer.output_sizes:
                size = s // tp
                slices.append((off, size, rank * size))
                off += size
        else:
            size = layer.output_size_per_partition
            slices = [(0, size, rank * size)]
        r_off = 0
        for part, (off, size, src) in zip(parts, slices):
            if part is None:
                continue
            A, B = part
            r = A.shape[0]
            if r_off + r > self.r_max:
                raise ValueError("packed LoRA sub-modules exceed max_lora_rank; raise --max-lora-rank to the sum of "
                                 "their ranks")
            A_buf[r_off:r_off + r].copy_(A)
            B_buf[off:off + size, r_off:r_off + r].copy_(B[src:src + size])
            r_off += r

    def _activate(self, lora: LoRAModel, busy: set[int]) -> int:
        if lora.id in self.slot_of:
            self.slot_of.move_to_end(lora.id)
            return self.slot_of[lora.id]
        if self.free_slots:
            slot = self.free_slots.pop(0)
        else:
            victim = next((i for i in self.slot_of if i not in busy and i not in self.pinned), None)
            if victim is None:
                raise RuntimeError("no free LoRA slot: more distinct adapters in one batch than max_loras")
            slot = self.slot_of.pop(victim)
        for name, layer in self.layers.items():
            leaf = name.rsplit(".", 1)[-1]
            prefix = name[: -len(leaf)]
            subs = self.packed.get(leaf, [leaf])
            parts = [lora.weights.get(prefix + s) for s in subs]
            layer.lora_has[slot] = any(p is not None for p in parts)
            if layer.lora_has[slot]:
                self._shard_into(layer, slot, parts)
            else:                                   # the fused kernel reads every slot: no stale weights
                layer.lora_A[slot].zero_()
                layer.lora_B[slot].zero_()
        self._place_vocab_layers(lora, slot)
        self.slot_of[lora.id] = slot
        return slot

    # ---- public API (worker RPC targets) ---------------------------------------------------------------
    def add_adapter(self, req) -> bool:
        if req.lora_int_id in self.cpu_cache:
            self.cpu_cache.move_to_end(req.lora_int_id)
            return False
        lora = LoRAModel.from_checkpoint(req.lora_path, req.lora_int_id, self.r_max, self.dtype, self.expected)
        # the sub-projections of a fused layer share the rank dimension of its buffers
        for subs in self.packed.values():
            n = sum(1 for sub in subs if any(k.endswith("." + sub) for k in lora.weights))
            if n * lora.rank > self.r_max:
                raise ValueError(f"adapter rank {lora.rank} on {n} fused projections needs max_lora_rank >= "
                                 f"{n * lora.rank}")
        while len(self.cpu_cache) >= self.max_cpu:
            victim = next((i for i in self.cpu_cache if i not in self.pinned), None)
            if victim is None:
                raise RuntimeError("all cached adapters are pinned")
            self.remove_adapter(victim)
        self.cpu_cache[lora.id] = lora
        return True

    def remove_adapter(self, lora_id: int) -> bool:
        found = self.cpu_cache.pop(lora_id, None) is not None
        slot = self.slot_of.pop(lora_id, None)
        if slot is not None:
            for layer in list(self.layers.values()) + list(self.embeddings.values()) + list(self.heads.values()):
                layer.lora_has[slot] = False
            self.free_slots.append(slot)
        self.pinned.discard(lora_id)
        return found

    def pin_adapter(self, lora_id: int) -> bool:
        if lora_id not in self.cpu_cache:
            raise ValueError(f"adapter {lora_id} is not loaded")
        self.pinned.add(lora_id)
        return True

    def list_adapters(self) -> set[int]:
        return set(self.cpu_cache)

    # ---- per step ------------------------------------------------------------------------------------
    def set_active(self, req_ids: list[str], rows: np.ndarray, n_sched: np.ndarray, ib) -> None:
        ids = ib.lora_id[rows]
        self.active_groups = []
        self.token_slots = None
        self.step_slots = None
        if not (ids > 0).any():
            return
        busy = set(int(i) for i in np.unique(ids) if i > 0)
        slot_by_id = {}
        for rid, lid in zip(req_ids, ids):
            lid = int(lid)
            if lid <= 0 or lid in slot_by_id:
                continue
            if lid not in self.cpu_cache:
                self.add_adapter(ib.reqs[rid].lora_request)
            slot_by_id[lid] = self._activate(self.cpu_cache[lid], busy)
        tok_slot = np.repeat(np.array([slot_by_id.get(int(i), -1) for i in ids], dtype=np.int64), n_sched)
        self.step_slots = tok_slot
        self._group(tok_slot)

    def _group(self, tok_slot: np.ndarray) -> None:
        self.active_groups = []
        self.token_slots = torch.from_numpy(tok_slot.astype(np.int32)).to(self.device, non_blocking=True)
        for slot in sorted(set(int(s) for s in np.unique(tok_slot) if s >= 0)):
            idx = np.nonzero(tok_slot == slot)[0]
            if len(idx) == len(tok_slot):
                self.active_groups.append((slot, None))
            else:
                self.active_groups.append((slot, torch.from_numpy(idx).to(self.device, non_blocking=True)))

    def clear_active(self) -> None:
        """End of the forward pass: nothing is adapted until the next ``set_active`` / ``head_rows``."""
        self.active_groups = []
        self.token_slots = None

    @contextlib.contextmanager
    def head_rows(self, index: np.ndarray):
        """The rows ``index`` of this step's tokens go through the output head inside this block."""
        if self.step_slots is None or not self.heads:
            yield
            return
        self._group(self.step_slots[np.asarray(index, dtype=np.int64)])
        try:
            yield
        finally:
            self.clear_active()
