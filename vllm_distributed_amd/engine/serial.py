"""msgpack codec for the engine wire structs.

Dataclasses registered with :func:`wire_struct` travel as ``[type_id, [field
values...]]`` arrays (array-like, no field names); tensors and numpy arrays are
msgpack ext types whose payload is the raw buffer, so large token / embedding
arrays are not copied element by element.  Nothing on the wire can execute
code on the receiver, unlike pickle.
Reference: vllm/v1/serial_utils.py (msgspec MsgpackEncoder/Decoder with aux buffers).
"""
from __future__ import annotations

import dataclasses
import enum
import struct
from typing import Any, Callable

import msgpack
import numpy as np
import torch

_EXT_NDARRAY, _EXT_TENSOR, _EXT_STRUCT, _EXT_SET, _EXT_TUPLE, _EXT_ENUM, _EXT_BYTES_ENUM, _EXT_DTYPE = \
    1, 2, 3, 4, 5, 6, 7, 8
_BY_ID: dict[int, type] = {}
_BY_CLS: dict[type, int] = {}
_ENUMS: dict[int, type] = {}
_ENUM_IDS: dict[type, int] = {}
_CUSTOM: dict[type, tuple[int, Callable, Callable]] = {}


def wire_struct(type_id: int):
    def deco(cls):
        if type_id in _BY_ID and _BY_ID[type_id] is not cls:
            raise ValueError(f"wire id {type_id} already used by {_BY_ID[type_id].__name__}")
        _BY_ID[type_id], _BY_CLS[cls] = cls, type_id
        return cls
    return deco


def wire_enum(type_id: int, cls: type) -> None:
    _ENUMS[type_id], _ENUM_IDS[cls] = cls, type_id


def wire_custom(type_id: int, cls: type, to_wire: Callable, from_wire: Callable) -> None:
    """Classes with their own (dict based) wire form, e.g. SamplingParams."""
    _BY_ID[type_id] = cls
    _CUSTOM[cls] = (type_id, to_wire, from_wire)


_TORCH_DT = {str(d).replace("torch.", ""): d for d in (
    torch.float32, torch.float16, torch.bfloat16, torch.float64, torch.int8, torch.uint8, torch.int16, torch.int32,
    torch.int64, torch.bool, torch.float8_e4m3fn)}


def _default(obj: Any):
    cls = type(obj)
    if cls in _CUSTOM:
        tid, to_wire, _ = _CUSTOM[cls]
        return msgpack.ExtType(_EXT_STRUCT, _pack([tid, to_wire(obj)]))
    if cls in _BY_CLS:
        vals = [getattr(obj, f.name) for f in dataclasses.fields(obj)]
        return msgpack.ExtType(_EXT_STRUCT, _pack([_BY_CLS[cls], vals]))
    if isinstance(obj, torch.Tensor):
        t = obj.detach().cpu().contiguous()
        name = str(t.dtype).replace("torch.", "")
        raw = t.view(torch.uint8).numpy().tobytes() if t.numel() else b""
        return msgpack.ExtType(_EXT_TENSOR, _pack([name, list(t.shape), raw]))
    if isinstance(obj, torch.dtype):
        return msgpack.ExtType(_EXT_DTYPE, str(obj).replace("torch.", "").encode())
    if isinstance(obj, np.ndarray):
        a = np.ascontiguousarray(obj)
        return msgpack.ExtType(_EXT_NDARRAY, _pack([a.dtype.str, list(a.shape), a.tobytes()]))
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if isinstance(obj, (np.floating,)):
        return float(obj)
    if isinstance(obj, (set, frozenset)):
        return msgpack.ExtType(_EXT_SET, _pack(sorted(obj, key=repr)))
    if isinstance(obj, enum.Enum):
        if cls not in _ENUM_IDS:
            raise TypeError(f"enum {cls.__name__} is not registered for the wire")
        v = obj.value
        if isinstance(v, bytes):
            return msgpack.ExtType(_EXT_BYTES_ENUM, _pack([_ENUM_IDS[cls], v]))
        return msgpack.ExtType(_EXT_ENUM, _pack([_ENUM_IDS[cls], v]))
    raise TypeError(f"cannot serialise {cls.__name__}")


def _ext_hook(code: int, data: bytes):
    if code == _EXT_STRUCT:
        tid, vals = _unpack(data)
        cls = _BY_ID[tid]
        if cls in _CUSTOM:
            return _CUSTOM[cls][2](vals)
        return cls(*vals)
    if code == _EXT_TENSOR:
        name, shape, raw = _unpack(data)
        dt = _TORCH_DT[name]
        if not raw:
            return torch.empty(shape, dtype=dt)
        return torch.frombuffer(bytearray(raw), dtype=torch.uint8).view(dt).reshape(shape)
    if code == _EXT_NDARRAY:
        dts, shape, raw = _unpack(data)
        return np.frombuffer(raw, dtype=np.dtype(dts)).reshape(shape).copy()
    if code == _EXT_DTYPE:
        return getattr(torch, data.decode())
    if code == _EXT_SET:
        return set(_unpack(data))
    if code == _EXT_TUPLE:
        return tuple(_unpack(data))
    if code in (_EXT_ENUM, _EXT_BYTES_ENUM):
        eid, v = _unpack(data)
        return _ENUMS[eid](v)
    return msgpack.ExtType(code, data)


def _unpack(data: bytes) -> Any:
    return msgpack.unpackb(data, ext_hook=_ext_hook, raw=False, strict_map_key=False)


def _default_with_tuples(obj):
    # msgpack would turn tuples into lists; strict_types=True routes them here, where they are
    # tagged so that block-id tuples etc. come back as tuples
    if isinstance(obj, tuple):
        return msgpack.ExtType(_EXT_TUPLE, _pack(list(obj)))
    return _default(obj)


def _pack(obj: Any) -> bytes:
    return msgpack.packb(obj, default=_default_with_tuples, use_bin_type=True, strict_types=True)


def encode(obj: Any) -> bytes:
    return _pack(obj)


def decode(data: bytes) -> Any:
    return _unpack(data)


def _register_engine_types() -> None:
    from .. import request as R
    from ..metrics import stats as S
    from ..sampling_params import GuidedDecodingParams, PoolingParams, RequestOutputKind, SamplingParams
    from ..sched import output as O
    for i, cls in enumerate((R.EngineCoreRequest, R.EngineCoreOutput, R.EngineCoreOutputs, R.UtilityOutput,
                             R.LogprobsLists, R.EngineCoreEvent, R.LoRARequest, R.PlaceholderRange,
                             S.SchedulerStats, S.PrefixCacheStats, S.SpecDecodingStats,
                             O.NewRequestData, O.CachedRequestData, O.SchedulerOutput, O.ModelRunnerOutput,
                             O.TokenParallelAllocation, GuidedDecodingParams), start=1):
        wire_struct(i)(cls)
    from ..sched import kv_cache_interface as K
    for i, cls in enumerate((K.FullAttentionSpec, K.SlidingWindowSpec, K.MambaSpec, K.KVCacheTensor,
                             K.KVCacheGroupSpec, K.KVCacheConfig, K.MLASpec), start=40):
        wire_struct(i)(cls)
    wire_custom(100, SamplingParams, SamplingParams.to_wire, SamplingParams.from_wire)
    wire_custom(101, PoolingParams, PoolingParams.to_wire, PoolingParams.from_wire)
    for i, cls in enumerate((R.FinishReason, R.EngineCoreEventType, R.EngineCoreRequestType, RequestOutputKind,
                             R.RequestStatus), start=1):
        wire_enum(i, cls)


_register_engine_types()
