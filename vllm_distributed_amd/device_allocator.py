"""Device memory services outside the PyTorch caching allocator.

* IPC-shareable regions (``alloc_shareable`` / ``open_shared``): plain ``hipMalloc`` memory with
  an IPC handle, wrapped as torch tensors through ``__cuda_array_interface__``.  Used for the
  KV arena of instances that export KV to peers over xGMI.
* Sleep mode (``offload_module`` / ``restore_module`` / ``drop_tensor`` / ``restore_tensor``):
  weights are parked in pinned host memory (level 1) or dropped (level 2) and the device
  memory goes back to the driver, so that another job (RLHF trainer) can use the GPU.
Reference: vllm/device_allocator/cumem.py (CuMemAllocator sleep/wake_up), csrc/cumem_allocator.cpp.
"""
from __future__ import annotations

import contextlib
import ctypes
import gc
from typing import Optional

import torch
import torch.nn as nn

from .logger import init_logger

logger = init_logger(__name__)
HANDLE_BYTES = 64


class _Foreign:
    """Minimal ``__cuda_array_interface__`` provider over a raw device pointer."""

    def __init__(self, ptr: int, nbytes: int, owner=None):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
        self._owner = owner


class SharedRegion:
    """Keeps an IPC region alive; ``tensor`` is a uint8 view of it."""

    def __init__(self, ptr: int, nbytes: int, device: torch.device, handle: bytes, owner: bool):
        self.ptr, self.nbytes, self.device, self.handle, self.owner = ptr, nbytes, device, handle, owner
        self.tensor = torch.as_tensor(_Foreign(ptr, nbytes, self), device=device)

    def close(self) -> None:
        if self.ptr:
            lib = _lib()
            (lib.vda_mem_free if self.owner else lib.vda_mem_close)(ctypes.c_void_p(self.ptr))
            self.ptr = 0


def _lib():
    from .parallel.custom_allreduce import comm_library
    lib = comm_library()
    if not getattr(lib, "_vda_mem_ready", False):
        vp = ctypes.c_void_p
        lib.vda_mem_alloc.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(vp), ctypes.c_char_p]
        lib.vda_mem_open.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(vp)]
        lib.vda_mem_free.argtypes = [vp]
        lib.vda_mem_close.argtypes = [vp]
        lib._vda_mem_ready = True
    return lib


def ipc_safe_size(nbytes: int) -> int:
    """Smallest size >= nbytes that another process can map.  On this ROCm stack
    hipIpcOpenMemHandle never returns for a region whose size has bit 31 set (size mod 4 GiB >=
    2 GiB: 2, 3, 6, 7, 14 GiB stall; 1, 4, 5, 8, 12, 13, 40, 100 GiB map in ~1 ms), so such sizes are
    rounded up to the next multiple of 4 GiB."""
    if nbytes & 0x80000000:
        return (nbytes + 0xFFFFFFFF) & ~0xFFFFFFFF
    return nbytes


def alloc_shareable(nbytes: int, device: torch.device) -> SharedRegion:
    """Zero-filled device region with an IPC handle (GPU only); may be larger than asked
    (see ``ipc_safe_size``), ``tensor`` covers the whole region."""
    nbytes = ipc_safe_size(nbytes)
    ptr = ctypes.c_void_p()
    handle = ctypes.create_string_buffer(HANDLE_BYTES)
    rc = _lib().vda_mem_alloc(nbytes, device.index or 0, ctypes.byref(ptr), handle)
    if rc != 0:
        raise RuntimeError(f"hipMalloc/hipIpcGetMemHandle of {nbytes} bytes failed with HIP error {rc}")
    return SharedRegion(ptr.value, nbytes, device, handle.raw, owner=True)


def open_shared(handle: bytes, nbytes: int, device: torch.device) -> SharedRegion:
    ptr = ctypes.c_void_p()
    rc = _lib().vda_mem_open(handle, device.index or 0, ctypes.byref(ptr))
    if rc != 0:
        raise RuntimeError(f"hipIpcOpenMemHandle failed with HIP error {rc} (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)")
    return SharedRegion(ptr.value, nbytes, device, handle, owner=False)


# ---- sleep mode, address-stable (virtual memory management) ---------------------------------------
def _immortal(obj) -> None:
    ctypes.pythonapi.Py_IncRef(ctypes.py_object(obj))


class SleepAllocator:
    """Tagged memory pools on the native VMM allocator (parallel/csrc/device_memory.hip).

        alloc = SleepAllocator.get()
        with alloc.use_memory_pool("weights"):   model = load(...)
        with alloc.use_memory_pool("kv_cache"):  kv = torch.zeros(...)
        alloc.sleep(offload_tags=("weights",))   # weights -> pinned host, KV pages discarded
        alloc.wake_up()                          # same addresses again

    Every allocation made inside ``use_memory_pool`` goes through PyTorch's pluggable-allocator
    hook into ``vda_vmm_malloc``; the native side keeps the block table, so sleeping and waking are
    one call each and tensors, views, graph-captured pointers all stay valid.
    Reference: vllm/device_allocator/cumem.py:113-281."""
    TAGS = {"weights": 1, "kv_cache": 2}
    _instance: Optional["SleepAllocator"] = None

    @classmethod
    def get(cls) -> "SleepAllocator":
        if cls._instance is None:
            cls._instance = cls()
        return cls._instance

    def __init__(self):
        from . import envs
        import os
        lib = _lib()
        ll = ctypes.c_longlong
        lib.vda_vmm_set_tag.argtypes = [ctypes.c_int]
        lib.vda_vmm_sleep.argtypes = [ctypes.c_int, ctypes.c_int]
        lib.vda_vmm_sleep.restype = ll
        lib.vda_vmm_wake.argtypes = [ctypes.c_int]
        lib.vda_vmm_wake.restype = ll
        lib.vda_vmm_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
        lib.vda_vmm_bytes.restype = ll
        self.lib = lib
        path = envs.VDA_COMM_LIBRARY or os.path.join(os.path.dirname(os.path.abspath(__file__)), "parallel",
                                                     "libvda_comm.so")
        self._pluggable = torch.cuda.memory.CUDAPluggableAllocator(path, "vda_vmm_malloc", "vda_vmm_free")
        # A MemPool keeps only a raw pointer to its allocator, and interpreter shutdown destroys objects
        # in no useful order: the allocator and the pools are made immortal, the driver reclaims the
        # memory when the process ends.
        _immortal(self._pluggable)
        self._pools: dict[str, "torch.cuda.MemPool"] = {}
        self.asleep: set[str] = set()

    @contextlib.contextmanager
    def use_memory_pool(self, tag: str):
        if tag not in self.TAGS:
            raise ValueError(f"unknown memory tag {tag!r}")
        pool = self._pools.get(tag)
        if pool is None:
            pool = self._pools[tag] = torch.cuda.MemPool(self._pluggable.allocator())
            _immortal(pool)
        self.lib.vda_vmm_set_tag(self.TAGS[tag])
        try:
            with torch.cuda.use_mem_pool(pool):
                yield
        finally:
            self.lib.vda_vmm_set_tag(0)

    def bytes(self, tag: str, which: int = 0) -> int:
        return int(self.lib.vda_vmm_bytes(self.TAGS[tag], which))

    def sleep(self, offload_tags: tuple = ("weights",)) -> int:
        """Release the physical memory of every pool; pools named in ``offload_tags`` keep their
        contents in pinned host memory.  Returns bytes released."""
        torch.cuda.synchronize()
        total = 0
        for tag, tid in self.TAGS.items():
            if tag in self.asleep:
                continue
            n = int(self.lib.vda_vmm_sleep(tid, int(tag in offload_tags)))
            if n < 0:
                raise RuntimeError(f"releasing the {tag} pool failed with HIP error {self.lib.vda_vmm_last_error()}")
            total += n
            self.asleep.add(tag)
        return total

    def wake_up(self, tags: Optional[list] = None) -> int:
        total = 0
        for tag in (tags or list(self.TAGS)):
            if tag not in self.asleep:
                continue
            n = int(self.lib.vda_vmm_wake(self.TAGS[tag]))
            if n < 0:
                raise RuntimeError(f"mapping the {tag} pool failed with HIP error {self.lib.vda_vmm_last_error()}")
            total += n
            self.asleep.discard(tag)
        return total


# ---- sleep mode without VMM (free and re-create) ------------------------------------------------
def offload_module(module: nn.Module, keep_copy: bool = True) -> dict[str, Optional[torch.Tensor]]:
    """Release the device memory of every parameter and buffer; with ``keep_copy`` their values are
    kept in pinned host memory.  Tied tensors are stored once."""
    saved: dict[str, Optional[torch.Tensor]] = {}
    seen: dict[int, str] = {}
    for name, t in list(module.named_parameters()) + list(module.named_buffers()):
        if not t.is_cuda:
            continue
        key = t.data_ptr()
        if key in seen:
            saved[name] = saved[seen[key]]
            t.data = torch.empty(0, dtype=t.dtype, device=t.device)
            continue
        seen[key] = name
        if keep_copy:
            host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            host.copy_(t.data, non_blocking=True)
            saved[name] = host
        else:
            saved[name] = None
        module_shape = tuple(t.shape)
        saved[name + "\0shape"] = torch.tensor(module_shape, dtype=torch.int64)
    torch.cuda.synchronize()
    for name, t in list(module.named_parameters()) + list(module.named_buffers()):
        if t.is_cuda:
            t.data = torch.empty(0, dtype=t.dtype, device=t.device)
    gc.collect()
    torch.cuda.empty_cache()
    return saved


def restore_module(module: nn.Module, saved: dict[str, Optional[torch.Tensor]], device: torch.device) -> None:
    done: dict[int, torch.Tensor] = {}
    for name, t in list(module.named_parameters()) + list(module.named_buffers()):
        if name not in saved:
            continue
        host = saved[name]
        if host is None:
            shape = tuple(saved[name + "\0shape"].tolist())
            t.data = torch.empty(shape, dtype=t.dtype, device=device)      # level 2: caller reloads weights
            continue
        if id(host) in done:
            t.data = done[id(host)]
            continue
        dev_t = host.to(device, non_blocking=True)
        done[id(host)] = dev_t
        t.data = dev_t
    torch.cuda.synchronize()


def drop_tensor(t: torch.Tensor) -> None:
    t.data = torch.empty(0, dtype=t.dtype, device=t.device)


def restore_tensor(t: torch.Tensor, shape, device: torch.device) -> None:
    t.data = torch.zeros(tuple(shape), dtype=t.dtype, device=device)
