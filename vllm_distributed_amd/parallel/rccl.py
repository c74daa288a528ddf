"""Stream-explicit RCCL communicator (ctypes on librccl.so).

Why not only ``torch.distributed``: collectives issued here go on the *current*
HIP stream with no watchdog thread or work object in between, so they can be
recorded into a hipGraph and they order naturally with the surrounding kernels.
Used for everything the peer-memory kernels do not cover (large all-reduce,
all-gather, reduce-scatter, send/recv between pipeline stages, multi-node).
Reference: vllm/distributed/device_communicators/pynccl.py:21-218,
pynccl_wrapper.py (same role; on ROCm "nccl" is RCCL).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch
import torch.distributed as dist

from .. import envs
from ..logger import init_logger

logger = init_logger(__name__)

ncclResult_t = ctypes.c_int
ncclComm_t = ctypes.c_void_p
hipStream_t = ctypes.c_void_p
buffer_t = ctypes.c_void_p


class ncclUniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * 128)]


_DTYPES = {torch.int8: 0, torch.uint8: 1, torch.int32: 2, torch.int64: 4, torch.float16: 6, torch.float32: 7,
           torch.float64: 8, torch.bfloat16: 9}
_OPS = {"sum": 0, "prod": 1, "max": 2, "min": 3, "avg": 4}


def _find_library() -> str:
    if envs.VDA_RCCL_SO_PATH:
        return envs.VDA_RCCL_SO_PATH
    for p in (os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so",
              "librccl.so.1", "librccl.so"):
        if os.path.sep not in p or os.path.exists(p):
            return p
    return "librccl.so"


class RcclLibrary:
    _SIGS = {
        "ncclGetErrorString": (ctypes.c_char_p, [ncclResult_t]),
        "ncclGetUniqueId": (ncclResult_t, [ctypes.POINTER(ncclUniqueId)]),
        "ncclCommInitRank": (ncclResult_t, [ctypes.POINTER(ncclComm_t), ctypes.c_int, ncclUniqueId, ctypes.c_int]),
        "ncclAllReduce": (ncclResult_t, [buffer_t, buffer_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ncclComm_t,
                                         hipStream_t]),
        "ncclAllGather": (ncclResult_t, [buffer_t, buffer_t, ctypes.c_size_t, ctypes.c_int, ncclComm_t, hipStream_t]),
        "ncclReduceScatter": (ncclResult_t, [buffer_t, buffer_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                             ncclComm_t, hipStream_t]),
        "ncclBroadcast": (ncclResult_t, [buffer_t, buffer_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ncclComm_t,
                                         hipStream_t]),
        "ncclSend": (ncclResult_t, [buffer_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ncclComm_t, hipStream_t]),
        "ncclRecv": (ncclResult_t, [buffer_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ncclComm_t, hipStream_t]),
        "ncclGroupStart": (ncclResult_t, []),
        "ncclGroupEnd": (ncclResult_t, []),
        "ncclCommDestroy": (ncclResult_t, [ncclComm_t]),
    }
    _cache: dict[str, "RcclLibrary"] = {}

    def __init__(self, path: Optional[str] = None):
        path = path or _find_library()
        self.lib = ctypes.CDLL(path)
        for name, (res, args) in self._SIGS.items():
            f = getattr(self.lib, name)
            f.restype, f.argtypes = res, args
        self.path = path

    @classmethod
    def get(cls, path: Optional[str] = None) -> "RcclLibrary":
        key = path or _find_library()
        if key not in cls._cache:
            cls._cache[key] = cls(key)
        return cls._cache[key]

    def check(self, rc: int) -> None:
        if rc != 0:
            raise RuntimeError(f"RCCL error {rc}: {self.lib.ncclGetErrorString(rc).decode()}")


class RcclCommunicator:
    def __init__(self, cpu_group, device: torch.device, library_path: Optional[str] = None):
        """``cpu_group`` is a Gloo group (or a StatelessProcessGroup) used once to share the unique id."""
        self.rccl = RcclLibrary.get(library_path)
        self.device = device
        if hasattr(cpu_group, "broadcast_obj"):            # StatelessProcessGroup
            self.rank, self.world_size = cpu_group.rank, cpu_group.world_size
            uid = ncclUniqueId()
            if self.rank == 0:
                self.rccl.check(self.rccl.lib.ncclGetUniqueId(ctypes.byref(uid)))
            raw = cpu_group.broadcast_obj(bytes(uid.internal) if self.rank == 0 else None, src=0)
        else:
            self.rank, self.world_size = dist.get_rank(cpu_group), dist.get_world_size(cpu_group)
            uid = ncclUniqueId()
            if self.rank == 0:
                self.rccl.check(self.rccl.lib.ncclGetUniqueId(ctypes.byref(uid)))
            box = [bytes(uid.internal) if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(cpu_group, 0), group=cpu_group)
            raw = box[0]
        ctypes.memmove(ctypes.byref(uid), raw, 128)
        self.comm = ncclComm_t()
        with torch.cuda.device(device):
            self.rccl.check(self.rccl.lib.ncclCommInitRank(ctypes.byref(self.comm), self.world_size, uid, self.rank))
            # first collective builds the rings/trees: do it now, outside any graph capture
            warm = torch.zeros(1, device=device)
            self.all_reduce(warm)
            torch.cuda.current_stream(device).synchronize()
        self.disabled = False

    @staticmethod
    def _s() -> int:
        return torch.cuda.current_stream().cuda_stream

    def all_reduce(self, x: torch.Tensor, op: str = "sum", out: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = x if out is None else out
        self.rccl.check(self.rccl.lib.ncclAllReduce(x.data_ptr(), out.data_ptr(), x.numel(), _DTYPES[x.dtype],
                                                    _OPS[op], self.comm, self._s()))
        return out

    def all_gather(self, out: torch.Tensor, x: torch.Tensor) -> None:
        self.rccl.check(self.rccl.lib.ncclAllGather(x.data_ptr(), out.data_ptr(), x.numel(), _DTYPES[x.dtype],
                                                    self.comm, self._s()))

    def reduce_scatter(self, out: torch.Tensor, x: torch.Tensor, op: str = "sum") -> None:
        self.rccl.check(self.rccl.lib.ncclReduceScatter(x.data_ptr(), out.data_ptr(), out.numel(), _DTYPES[x.dtype],
                                                        _OPS[op], self.comm, self._s()))

    def broadcast(self, x: torch.Tensor, src: int) -> None:
        self.rccl.check(self.rccl.lib.ncclBroadcast(x.data_ptr(), x.data_ptr(), x.numel(), _DTYPES[x.dtype], src,
                                                    self.comm, self._s()))

    def send(self, x: torch.Tensor, dst: int) -> None:
        self.rccl.check(self.rccl.lib.ncclSend(x.data_ptr(), x.numel(), _DTYPES[x.dtype], dst, self.comm, self._s()))

    def recv(self, x: torch.Tensor, src: int) -> None:
        self.rccl.check(self.rccl.lib.ncclRecv(x.data_ptr(), x.numel(), _DTYPES[x.dtype], src, self.comm, self._s()))

    def group_start(self) -> None:
        self.rccl.check(self.rccl.lib.ncclGroupStart())

    def group_end(self) -> None:
        self.rccl.check(self.rccl.lib.ncclGroupEnd())

    def close(self) -> None:
        if getattr(self, "comm", None):
            try:
                self.rccl.lib.ncclCommDestroy(self.comm)
            except Exception:
                pass
            self.comm = None
