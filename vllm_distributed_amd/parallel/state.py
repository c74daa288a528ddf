"""Process groups and the rank mesh.

Rank grid (outermost to innermost): ``[external DP, DP | TKNP, PP, TP]``; TP is
the innermost, contiguous axis so that a TP group sits on xGMI-adjacent GPUs of
one node.  EP is the DP(or TKNP) x TP plane flattened.
Reference: vllm/distributed/parallel_state.py (GroupCoordinator :176-803, mesh
:1107-1203, tensor-dict transport :629-749).

Each :class:`GroupCoordinator` owns a device process group (RCCL through
``torch.distributed`` backend "nccl") and a Gloo CPU group, plus an optional
peer-memory communicator (parallel/custom_allreduce.py) that takes small and
medium all-reduces off RCCL: on the fully connected xGMI mesh a direct
one-shot/two-shot exchange uses all seven links at once where a ring uses one.

With world size 1 nothing from ``torch.distributed`` is initialised: the groups
are trivial and every collective is the identity.
"""
from __future__ import annotations

import contextlib
import pickle
from dataclasses import dataclass
from typing import Any, Optional, Sequence, Union

import torch
import torch.distributed as dist

from .. import envs
from ..logger import init_logger

logger = init_logger(__name__)


@dataclass
class GraphCaptureContext:
    stream: Any


class AsyncWork:
    """Result of ``GroupCoordinator.all_reduce_async``: ``wait()`` orders the caller's current stream
    after the collective and hands out the reduced tensor."""

    def __init__(self, out: torch.Tensor, done):
        self.out, self.done = out, done

    def wait(self) -> torch.Tensor:
        if self.done is not None:
            cur = torch.cuda.current_stream(self.out.device)
            cur.wait_event(self.done)
            self.out.record_stream(cur)    # allocated on the comm stream, consumed (and freed) on this one
            self.done = None
        return self.out


class GroupCoordinator:
    def __init__(self, ranks: Sequence[int], rank: int, local_rank: int, backend: str, name: str,
                 use_device_comm: bool = False, use_message_queue: bool = False):
        self.name = name
        self.ranks = list(ranks)
        self.world_size = len(self.ranks)
        self.rank = rank                         # global rank
        self.local_rank = local_rank
        self.rank_in_group = self.ranks.index(rank) if rank in self.ranks else 0
        self.device_group = None
        self.cpu_group = None
        self.backend = backend
        # "gloo+hip": rehearsal of a multi-rank GPU run on ONE device (every rank uses cuda:0; RCCL refuses
        # two ranks per device, so process groups are gloo, device tensors are staged through the host and
        # only the peer-memory all-reduce runs on the device)
        self.staged = backend == "gloo+hip"
        self.device = torch.device(f"cuda:{local_rank}") if backend == "nccl" else \
            (torch.device("cuda:0") if self.staged else torch.device("cpu"))
        self.ca_comm = None       # peer-memory all-reduce
        self.far_comm = None      # peer-memory all-reduce fused with residual add + RMSNorm
        self.ca_comm_async = None # second context (own arena, own call counters) for the communication stream
        self.qr_comm = None       # quantised peer-memory all-reduce (opt-in)
        self.pipe = None          # peer-memory hand-off between neighbouring pipeline stages (enable_pipe)
        self._pipe_layout = None  # tensor dictionary layout agreed on the first hop: later hops carry no metadata
        self.hop_counts = {"pipe": 0, "metadata": 0}
        # all-reduce calls issued from the host by path (a call captured into a graph counts once, at capture)
        self.path_counts = {"quick": 0, "peer_memory": 0, "fused_norm": 0, "rccl": 0, "torch": 0}
        self.rccl_comm = None     # stream-explicit RCCL (graph capturable)
        self.mq_broadcaster = None
        self.all2all = None
        self._comm_stream = None  # side stream of all_reduce_async
        self._use_device_comm = use_device_comm and backend in ("nccl", "gloo+hip")
        self._use_mq = use_message_queue

    def _attach(self, device_group, cpu_group) -> None:
        self.device_group, self.cpu_group = device_group, cpu_group
        if self.world_size > 1 and self._use_device_comm:
            self._init_device_comms()
        if self.world_size > 1 and self._use_mq:
            from ..runtime.shm_ring import MessageQueue
            self.mq_broadcaster = MessageQueue.create_from_process_group(self.cpu_group, 1 << 22, 6)

    def _init_device_comms(self) -> None:
        from ..platforms import current_platform
        if not self.staged:
            try:
                from .rccl import RcclCommunicator
                self.rccl_comm = RcclCommunicator(self.cpu_group, self.device)
            except Exception as e:   # library missing or init failure: torch.distributed still works
                logger.warning("stream-explicit RCCL communicator unavailable for group %s: %s", self.name, e)
                self.rccl_comm = None
        if "tp" in self.name and not envs.VLLM_DISABLE_CUSTOM_ALL_REDUCE \
                and current_platform.use_custom_allreduce():
            try:
                from .custom_allreduce import CustomAllReduce
                self.ca_comm = CustomAllReduce.maybe_create(self.cpu_group, self.device)
            except Exception as e:
                logger.warning("peer-memory all-reduce disabled for group %s: %s", self.name, e)
                self.ca_comm = None
            if self.ca_comm is not None:
                try:
                    from .custom_allreduce import FusedAllReduceNorm
                    self.far_comm = FusedAllReduceNorm.maybe_create(self.cpu_group, self.device)
                except Exception as e:
                    logger.warning("fused all-reduce + norm disabled for group %s: %s", self.name, e)
                    self.far_comm = None
            if envs.VLLM_ROCM_QUICK_REDUCE_QUANTIZATION != "NONE":
                try:
                    from .custom_allreduce import QuickAllReduce
                    self.qr_comm = QuickAllReduce.maybe_create(self.cpu_group, self.device)
                except Exception as e:
                    logger.warning("quantised all-reduce disabled for group %s: %s", self.name, e)
                    self.qr_comm = None

    def enable_pipe(self, slot_bytes: int) -> bool:
        """Collective over the group: build the peer-memory pipe for messages of up to ``slot_bytes`` between
        neighbouring ranks (on-node GPU groups; every rank answers alike)."""
        if self.world_size < 2 or not self._use_device_comm or not envs.VDA_PP_PEER_PIPE or self.pipe is not None:
            return self.pipe is not None
        from ..platforms import current_platform
        if not current_platform.use_custom_allreduce():
            return False
        try:
            from .custom_allreduce import PeerPipe
            self.pipe = PeerPipe.maybe_create(self.cpu_group, self.device, slot_bytes)
        except Exception as e:   # noqa: BLE001
            logger.warning("peer-memory pipeline hand-off disabled for group %s: %s", self.name, e)
            self.pipe = None
        return self.pipe is not None

    # ---- properties --------------------------------------------------------
    @property
    def first_rank(self) -> int:
        return self.ranks[0]

    @property
    def last_rank(self) -> int:
        return self.ranks[-1]

    @property
    def is_first_rank(self) -> bool:
        return self.rank == self.first_rank

    @property
    def is_last_rank(self) -> bool:
        return self.rank == self.last_rank

    @property
    def next_rank(self) -> int:
        return self.ranks[(self.rank_in_group + 1) % self.world_size]

    @property
    def prev_rank(self) -> int:
        return self.ranks[(self.rank_in_group - 1) % self.world_size]

    # ---- collectives -------------------------------------------------------
    def _pg(self, x: torch.Tensor):
        return self.device_group if (x.is_cuda and not self.staged) else self.cpu_group

    def _down(self, x: torch.Tensor) -> torch.Tensor:
        """Host copy of a device tensor in staged mode, the tensor itself otherwise."""
        return x.cpu() if (self.staged and x.is_cuda) else x

    def all_reduce(self, x: torch.Tensor, _side_stream: bool = False) -> torch.Tensor:
        """Out of place for the peer-memory path, in place otherwise; always use the return value."""
        if self.world_size == 1:
            return x
        if x.is_cuda:
            # A peer-memory context keeps per-workgroup call counters and parity-selected staging halves:
            # two kernels of ONE context in flight on two streams would corrupt each other's parity, so
            # calls issued on the communication stream use a context of their own.
            ca = self.ca_comm_async if _side_stream else self.ca_comm
            if self.qr_comm is not None and not _side_stream:          # opt-in, lossy, large messages only
                out = self.qr_comm.all_reduce(x)
                if out is not None:
                    self.path_counts["quick"] += 1
                    return out
            if ca is not None and (self.rccl_comm is None or envs.VDA_CUSTOM_AR_CHUNKED
                                   or x.numel() * x.element_size() <= ca.max_bytes):
                # latency-bound sizes (decode) take the peer-memory kernel; messages beyond one arena
                # (prefill chunks, 100+ MiB) are bandwidth bound and go to RCCL
                out = ca.all_reduce(x)
                if out is not None:
                    self.path_counts["peer_memory"] += 1
                    return out
            if self.rccl_comm is not None:
                self.path_counts["rccl"] += 1
                return self.rccl_comm.all_reduce(x)
        self.path_counts["torch"] += 1
        h = self._down(x)
        dist.all_reduce(h, group=self._pg(x))
        if h is not x:
            x.copy_(h)
        return x

    def can_fuse_reduce_norm(self, x: torch.Tensor) -> bool:
        """Whether ``all_reduce_rms_norm`` would take the one-kernel path for this tensor (a pure function
        of shape and dtype, so every rank answers alike)."""
        return self.world_size > 1 and x.is_cuda and self.far_comm is not None and self.far_comm.fits(x)

    def all_reduce_rms_norm(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                            weight_bias: float = 0.0) -> tuple[torch.Tensor, torch.Tensor]:
        """``residual += all_reduce(x)`` (in place) and ``rmsnorm(residual)``: one peer-memory kernel where
        the shape fits, all-reduce followed by the fused add + norm kernel otherwise (same bits)."""
        if self.can_fuse_reduce_norm(x):
            out = self.far_comm.run(x, residual, weight, eps, weight_bias)
            if out is not None:
                self.path_counts["fused_norm"] += 1
                return out, residual
        from .. import ops
        x = self.all_reduce(x)
        return ops.rms_norm(x, weight, eps, residual, weight_bias), residual

    def all_reduce_async(self, x: torch.Tensor) -> "AsyncWork":
        """Start the all-reduce of ``x`` on this group's communication stream and return at once; the
        caller keeps issuing independent kernels on its own stream and calls ``wait()`` where the sum is
        needed.  All ranks must start their collectives in the same order.  On the host path (and for
        groups of one) the work is finished before this returns."""
        if self.world_size == 1 or not x.is_cuda:
            return AsyncWork(self.all_reduce(x), None)
        cur = torch.cuda.current_stream(x.device)
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=x.device)
            if self.ca_comm is not None:
                # first asynchronous call (every rank makes it at the same point): second context
                from .custom_allreduce import CustomAllReduce
                torch.cuda.synchronize(x.device)
                self.ca_comm_async = CustomAllReduce.maybe_create(self.cpu_group, self.device)
        comm = self._comm_stream
        ready = torch.cuda.Event()
        ready.record(cur)
        comm.wait_event(ready)
        x.record_stream(comm)              # produced on the caller's stream, consumed on the comm stream
        with torch.cuda.stream(comm):
            out = self.all_reduce(x, _side_stream=True)
            done = torch.cuda.Event()
            done.record(comm)
        return AsyncWork(out, done)

    def comm_info(self) -> dict:
        """What this group's collectives really ran on: the ranks the RCCL communicator and the peer-memory arena
        were built over (0 = that layer is absent) and the all-reduce calls per path so far."""
        def ranks(c):
            if c is None or getattr(c, "disabled", False):
                return 0
            return int(getattr(c, "world_size", None) or getattr(c, "world", None) or self.world_size)
        return {"group": self.name, "world_size": self.world_size,
                "rccl_ranks": ranks(self.rccl_comm), "peer_arena_ranks": ranks(self.ca_comm),
                "fused_norm_ranks": ranks(self.far_comm), "allreduce_path": dict(self.path_counts),
                "pipe_ranks": ranks(self.pipe), "hops": dict(self.hop_counts)}

    def check_comm_errors(self) -> None:
        """Raise if a peer-memory collective of this group timed out since the last check.  A kernel
        whose barrier timed out skipped its reads, so the step that contained it produced garbage: the
        worker must die (the engine then reports ENGINE_CORE_DEAD) rather than serve it.  Reads pinned
        host words only; call after the stream has been synchronised (the sampled-token read-back)."""
        for name in ("ca_comm", "ca_comm_async", "qr_comm", "far_comm", "pipe"):
            c = getattr(self, name, None)
            if c is not None and not c.disabled:
                e = c.check_error()
                if e:
                    raise RuntimeError(f"peer-memory collective of group {self.name} timed out waiting for rank "
                                       f"{e - 1} of the group ({name}); the step's results are invalid")

    def all_gather(self, x: torch.Tensor, dim: int = -1) -> torch.Tensor:
        if self.world_size == 1:
            return x
        if dim < 0:
            dim += x.dim()
        x = x.contiguous()
        if x.dim() == 0:
            x = x.reshape(1)
        flat = None
        if x.is_cuda and self.ca_comm is not None and not self.ca_comm.disabled:
            flat = self.ca_comm.all_gather(x)              # peer-memory kernel for on-node groups
            if flat is not None:
                self.path_counts["peer_memory"] += 1
        if flat is not None:
            pass
        elif x.is_cuda and self.rccl_comm is not None:
            flat = torch.empty((self.world_size * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            self.rccl_comm.all_gather(flat, x)
        else:
            flat = torch.empty((self.world_size * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            hf, hx = self._down(flat), self._down(x)
            dist.all_gather_into_tensor(hf, hx, group=self._pg(x))
            if hf is not flat:
                flat.copy_(hf)
        out = flat.view((self.world_size,) + tuple(x.shape))
        if dim == 0:
            return out.reshape((self.world_size * x.shape[0],) + tuple(x.shape[1:]))
        out = out.movedim(0, dim)
        return out.reshape(tuple(x.shape[:dim]) + (self.world_size * x.shape[dim],) + tuple(x.shape[dim + 1:]))

    def reduce_scatter(self, x: torch.Tensor, dim: int = 0) -> torch.Tensor:
        if self.world_size == 1:
            return x
        if dim < 0:
            dim += x.dim()
        xt = x.movedim(dim, 0).contiguous()
        assert xt.shape[0] % self.world_size == 0
        if x.is_cuda and self.ca_comm is not None and not self.ca_comm.disabled:
            got = self.ca_comm.reduce_scatter(xt)           # peer-memory kernel for on-node groups
            if got is not None:
                self.path_counts["peer_memory"] += 1
                return got if dim == 0 else got.movedim(0, dim)
        out = torch.empty((xt.shape[0] // self.world_size,) + tuple(xt.shape[1:]), dtype=x.dtype, device=x.device)
        if x.is_cuda and self.rccl_comm is not None:
            self.rccl_comm.reduce_scatter(out, xt)
        else:
            ho, hx = self._down(out), self._down(xt)
            if self.staged and x.is_cuda:     # gloo has no reduce-scatter: all-reduce and slice
                dist.all_reduce(hx, group=self.cpu_group)
                n = ho.shape[0]
                ho.copy_(hx[self.rank_in_group * n:(self.rank_in_group + 1) * n])
            else:
                dist.reduce_scatter_tensor(ho, hx, group=self._pg(x))
            if ho is not out:
                out.copy_(ho)
        return out.movedim(0, dim)

    def gather(self, x: torch.Tensor, dst: int = 0, dim: int = -1) -> Optional[torch.Tensor]:
        """Concatenate along ``dim`` on group rank ``dst``; None elsewhere."""
        if self.world_size == 1:
            return x
        if dim < 0:
            dim += x.dim()
        x = x.contiguous()
        hx = self._down(x)
        lst = [torch.empty_like(hx) for _ in range(self.world_size)] if self.rank_in_group == dst else None
        dist.gather(hx, lst, dst=self.ranks[dst], group=self._pg(x))
        return torch.cat(lst, dim=dim).to(x.device) if lst is not None else None

    def broadcast(self, x: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.world_size == 1:
            return x
        h = self._down(x)
        dist.broadcast(h, src=self.ranks[src], group=self._pg(x))
        if h is not x:
            x.copy_(h)
        return x

    def all_to_all_single(self, out: torch.Tensor, inp: torch.Tensor, out_splits=None, in_splits=None) -> torch.Tensor:
        if self.world_size == 1:
            out.copy_(inp)
            return out
        ho, hi = self._down(out), self._down(inp)
        dist.all_to_all_single(ho, hi, out_splits, in_splits, group=self._pg(inp))
        if ho is not out:
            out.copy_(ho)
        return out

    def broadcast_object(self, obj: Any = None, src: int = 0) -> Any:
        if self.world_size == 1:
            return obj
        if self.mq_broadcaster is not None and src == 0:
            return self.mq_broadcaster.broadcast_object(obj)
        box = [obj]
        dist.broadcast_object_list(box, src=self.ranks[src], group=self.cpu_group)
        return box[0]

    def all_gather_object(self, obj: Any) -> list:
        if self.world_size == 1:
            return [obj]
        out = [None] * self.world_size
        dist.all_gather_object(out, obj, group=self.cpu_group)
        return out

    def send_object(self, obj: Any, dst: int) -> None:
        data = torch.frombuffer(bytearray(pickle.dumps(obj)), dtype=torch.uint8)
        size = torch.tensor([data.numel()], dtype=torch.long)
        dist.send(size, dst=self.ranks[dst], group=self.cpu_group)
        dist.send(data, dst=self.ranks[dst], group=self.cpu_group)

    def recv_object(self, src: int) -> Any:
        size = torch.empty(1, dtype=torch.long)
        dist.recv(size, src=self.ranks[src], group=self.cpu_group)
        data = torch.empty(int(size.item()), dtype=torch.uint8)
        dist.recv(data, src=self.ranks[src], group=self.cpu_group)
        return pickle.loads(data.numpy().tobytes())

    def send(self, x: torch.Tensor, dst: Optional[int] = None) -> None:
        dst = (self.rank_in_group + 1) % self.world_size if dst is None else dst
        if x.is_cuda and self.rccl_comm is not None:
            self.rccl_comm.send(x, dst)
        else:
            dist.send(self._down(x), self.ranks[dst], group=self._pg(x))

    def recv(self, size, dtype: torch.dtype, src: Optional[int] = None, device=None) -> torch.Tensor:
        src = (self.rank_in_group - 1) % self.world_size if src is None else src
        x = torch.empty(size, dtype=dtype, device=device or self.device)
        if x.is_cuda and self.rccl_comm is not None:
            self.rccl_comm.recv(x, src)
        else:
            h = self._down(x)
            dist.recv(h, self.ranks[src], group=self._pg(x))
            if h is not x:
                x.copy_(h)
        return x

    # ---- tensor dicts ----------------------------------------------------------
    @staticmethod
    def _split_dict(d: dict) -> tuple[list, list[torch.Tensor]]:
        meta, tensors = [], []
        for k, v in d.items():
            if isinstance(v, torch.Tensor):
                meta.append((k, ("T", v.device.type, v.dtype, tuple(v.shape))))
                tensors.append(v)
            else:
                meta.append((k, ("O", v)))
        return meta, tensors

    def broadcast_tensor_dict(self, d: Optional[dict] = None, src: int = 0) -> Optional[dict]:
        if self.world_size == 1:
            return d
        if self.rank_in_group == src:
            meta, tensors = self._split_dict(d)
            self.broadcast_object(meta, src)
            for t in tensors:
                if t.numel():
                    self.broadcast(t, src)
            return d
        meta = self.broadcast_object(None, src)
        out = {}
        for k, m in meta:
            if m[0] == "T":
                _, devt, dtype, shape = m
                t = torch.empty(shape, dtype=dtype, device=self.device if devt == "cuda" else "cpu")
                if t.numel():
                    self.broadcast(t, src)
                out[k] = t
            else:
                out[k] = m[1]
        return out

    # Pipe form of a hop (``rows`` given, pipe enabled, next / previous rank): the first hop of a group travels with
    # its metadata as before and fixes the LAYOUT of the dictionary: per key the dtype and trailing shape of a tensor
    # whose leading dimension is the step's row count, an empty tensor, or a constant.  Every later hop sends the
    # first ``rows`` rows of each tensor through the peer-memory pipe and nothing else: both sides know ``rows``
    # from the scheduler output they share, so no host message accompanies the activations.
    @staticmethod
    def _layout_of(d: dict, rows: int) -> list:
        out = []
        for k, v in d.items():
            if isinstance(v, torch.Tensor) and v.numel() == 0:
                out.append((k, "empty", v.dtype, tuple(v.shape)))
            elif isinstance(v, torch.Tensor) and v.is_cuda and v.dim() >= 1 and v.shape[0] >= rows:
                out.append((k, "rows", v.dtype, tuple(v.shape[1:])))
            else:
                return []                         # something the pipe does not carry: keep the metadata form
        return out

    def _pipe_fits(self, layout: list, rows: int, n: int) -> bool:
        for _, kind, dtype, trail in layout:
            if kind != "rows":
                continue
            numel = rows
            for s_ in trail:
                numel *= s_
            if numel == 0 or numel % n:
                return False
            nb = numel // n * torch.empty((), dtype=dtype).element_size()
            if nb % 16 or nb > self.pipe.slot_bytes:
                return False
        return True

    def send_tensor_dict(self, d: dict, dst: Optional[int] = None, all_gather_group: "GroupCoordinator" = None,
                         rows: Optional[int] = None) -> None:
        """Metadata over Gloo, payload over the device link.  With ``all_gather_group``
        (the TP group) each rank ships only its 1/TP slice and the receiver
        all-gathers, so a PP hop costs M/TP per xGMI link instead of M.  With ``rows`` (the number of leading rows
        that matter, known to both sides) and an enabled pipe, hops after the first carry no metadata at all."""
        if self.world_size == 1:
            return
        dst = (self.rank_in_group + 1) % self.world_size if dst is None else dst
        ag = all_gather_group
        n, r = (ag.world_size, ag.rank_in_group) if ag is not None else (1, 0)
        piped = rows is not None and self.pipe is not None and not self.pipe.disabled \
            and dst == (self.rank_in_group + 1) % self.world_size
        if piped and self._pipe_layout is None:
            # first hop: the ordinary form, preceded by the layout the later hops will follow
            self._pipe_layout = self._layout_of(d, rows)
            self.send_object(("layout", self._pipe_layout), dst)
        elif piped and self._pipe_layout and self._pipe_fits(self._pipe_layout, rows, n):
            if self._layout_of(d, rows) != self._pipe_layout:
                raise RuntimeError(f"group {self.name}: the tensors handed to the next stage changed their layout")
            for k, kind, _, _ in self._pipe_layout:
                if kind != "rows":
                    continue
                t = d[k][:rows]
                t = t.reshape(n, -1)[r] if n > 1 else t
                self.pipe.send(t.contiguous())
            self.hop_counts["pipe"] += 1
            return
        self.hop_counts["metadata"] += 1
        meta, tensors = self._split_dict(d)
        self.send_object(meta, dst)
        for t in tensors:
            if not t.numel():
                continue
            if n > 1 and t.numel() % n == 0:
                t = t.reshape(n, -1)[r]
            self.send(t.contiguous(), dst)

    def recv_tensor_dict(self, src: Optional[int] = None, all_gather_group: "GroupCoordinator" = None,
                         rows: Optional[int] = None) -> Optional[dict]:
        if self.world_size == 1:
            return None
        src = (self.rank_in_group - 1) % self.world_size if src is None else src
        ag = all_gather_group
        n = ag.world_size if ag is not None else 1
        piped = rows is not None and self.pipe is not None and not self.pipe.disabled \
            and src == (self.rank_in_group - 1) % self.world_size
        if piped and self._pipe_layout is None:
            tag, layout = self.recv_object(src)
            assert tag == "layout"
            self._pipe_layout = [tuple(e) for e in layout]
        elif piped and self._pipe_layout and self._pipe_fits(self._pipe_layout, rows, n):
            out = {}
            for k, kind, dtype, trail in self._pipe_layout:
                if kind == "empty":
                    out[k] = torch.empty(trail, dtype=dtype, device=self.device)
                    continue
                numel = rows
                for s_ in trail:
                    numel *= s_
                part = self.pipe.recv((numel // n,), dtype)
                out[k] = (ag.all_gather(part, dim=0) if n > 1 else part).reshape((rows,) + tuple(trail))
            self.hop_counts["pipe"] += 1
            return out
        self.hop_counts["metadata"] += 1
        meta = self.recv_object(src)
        out = {}
        for k, m in meta:
            if m[0] != "T":
                out[k] = m[1]
                continue
            _, devt, dtype, shape = m
            dev = self.device if devt == "cuda" else torch.device("cpu")
            numel = 1
            for s in shape:
                numel *= s
            if numel == 0:
                out[k] = torch.empty(shape, dtype=dtype, device=dev)
            elif n > 1 and numel % n == 0:
                part = self.recv((numel // n,), dtype, src, dev)
                out[k] = ag.all_gather(part, dim=0).reshape(shape)
            else:
                out[k] = self.recv(shape, dtype, src, dev)
        return out

    def barrier(self) -> None:
        if self.world_size > 1:
            dist.barrier(group=self.cpu_group)

    @contextlib.contextmanager
    def graph_capture(self, ctx: Optional[GraphCaptureContext] = None):
        """Collectives issued inside are recorded into the hipGraph being captured."""
        if ctx is None:
            ctx = GraphCaptureContext(torch.cuda.Stream())
        cm = self.ca_comm.capture() if self.ca_comm is not None else contextlib.nullcontext()
        with cm:
            yield ctx

    def destroy(self) -> None:
        if self.qr_comm is not None:
            self.qr_comm.close()
            self.qr_comm = None
        if self.ca_comm is not None:
            self.ca_comm.close()
            self.ca_comm = None
        if self.ca_comm_async is not None:        # created lazily by the first all_reduce_async: same IPC mappings
            self.ca_comm_async.close()
            self.ca_comm_async = None
        if self.far_comm is not None:
            self.far_comm.close()
            self.far_comm = None
        if self.pipe is not None:
            self.pipe.close()
            self.pipe = None
        if self.rccl_comm is not None:
            self.rccl_comm.close()
            self.rccl_comm = None
        if self.mq_broadcaster is not None:
            self.mq_broadcaster = None
        for g in (self.device_group, self.cpu_group):
            if g is not None:
                try:
                    dist.destroy_process_group(g)
                except Exception:
                    pass
        self.device_group = self.cpu_group = None


# ---------------------------------------------------------------------------
_WORLD: Optional[GroupCoordinator] = None
_TP: Optional[GroupCoordinator] = None
_PP: Optional[GroupCoordinator] = None
_DP: Optional[GroupCoordinator] = None
_EP: Optional[GroupCoordinator] = None
_TKNP: Optional[GroupCoordinator] = None
_BACKEND = "gloo"
_LOCAL_RANK = 0


def _trivial(name: str) -> GroupCoordinator:
    rank = dist.get_rank() if dist.is_initialized() else 0
    return GroupCoordinator([rank], rank, _LOCAL_RANK, _BACKEND, name)


def init_distributed_environment(world_size: int = 1, rank: int = 0, distributed_init_method: str = "env://",
                                 local_rank: int = 0, backend: str = "nccl") -> None:
    global _WORLD, _BACKEND, _LOCAL_RANK
    _BACKEND, _LOCAL_RANK = backend, local_rank
    if world_size > 1 and not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device(f"cuda:{local_rank}")
        dist.init_process_group(backend="gloo" if backend == "gloo+hip" else backend,
                                init_method=distributed_init_method, world_size=world_size, rank=rank, **kw)
    if _WORLD is None:
        if world_size > 1:
            ranks = list(range(dist.get_world_size()))
            _WORLD = _new_group([ranks], rank, local_rank, backend, "world")
        else:
            _WORLD = _trivial("world")


def _new_group(all_ranks: list[list[int]], rank: int, local_rank: int, backend: str, name: str,
               use_device_comm: bool = False, use_message_queue: bool = False) -> GroupCoordinator:
    """Every process must create every subgroup (torch.distributed rule); keep ours."""
    mine = None
    for ranks in all_ranks:
        dg = dist.new_group(ranks, backend="gloo" if backend == "gloo+hip" else backend)
        cg = dist.new_group(ranks, backend="gloo")
        if rank in ranks:
            mine = GroupCoordinator(ranks, rank, local_rank, backend, name, use_device_comm, use_message_queue)
            mine._attach(dg, cg)
    assert mine is not None, f"rank {rank} is in no {name} group"
    return mine


def build_mesh(world_size: int, tp: int, pp: int, mid: int) -> torch.Tensor:
    """Rank grid [ext_dp, mid (DP or TKNP), pp, tp]."""
    assert world_size % (tp * pp * mid) == 0, (world_size, tp, pp, mid)
    return torch.arange(world_size).reshape(-1, mid, pp, tp)


def group_ranks(mesh: torch.Tensor, axis: str) -> list[list[int]]:
    if axis == "tp":
        return [x.tolist() for x in mesh.reshape(-1, mesh.shape[-1]).unbind(0)]
    if axis == "pp":
        return [x.tolist() for x in mesh.transpose(2, 3).reshape(-1, mesh.shape[2]).unbind(0)]
    if axis == "mid":
        return [x.tolist() for x in mesh.transpose(1, 3).reshape(-1, mesh.shape[1]).unbind(0)]
    if axis == "ep":
        return [x.tolist() for x in mesh.transpose(1, 2).reshape(-1, mesh.shape[1] * mesh.shape[3]).unbind(0)]
    raise ValueError(axis)


def initialize_model_parallel(tensor_parallel_size: int = 1, pipeline_parallel_size: int = 1,
                              data_parallel_size: int = 1, token_parallel_size: int = 1,
                              backend: Optional[str] = None) -> None:
    global _TP, _PP, _DP, _EP, _TKNP
    if _WORLD is None:
        init_distributed_environment()
    world = dist.get_world_size() if dist.is_initialized() else 1
    if token_parallel_size > 1 and data_parallel_size > 1:
        raise ValueError("token parallelism and in-group data parallelism are mutually exclusive")
    if world == 1:
        _TP, _PP, _DP, _EP, _TKNP = (_trivial(n) for n in ("tp", "pp", "dp", "ep", "tknp"))
        return
    backend = backend or _BACKEND
    rank = dist.get_rank()
    mid = token_parallel_size if token_parallel_size > 1 else data_parallel_size
    mesh = build_mesh(world, tensor_parallel_size, pipeline_parallel_size, mid)
    _TP = _new_group(group_ranks(mesh, "tp"), rank, _LOCAL_RANK, backend, "tp", use_device_comm=True,
                     use_message_queue=False)
    _PP = _new_group(group_ranks(mesh, "pp"), rank, _LOCAL_RANK, backend, "pp", use_device_comm=True)
    mid_groups = group_ranks(mesh, "mid")
    if token_parallel_size > 1:
        _TKNP = _new_group(mid_groups, rank, _LOCAL_RANK, backend, "tknp", use_device_comm=True)
        _DP = _trivial("dp")
    else:
        _DP = _new_group(mid_groups, rank, _LOCAL_RANK, backend, "dp", use_device_comm=True)
        _TKNP = _trivial("tknp")
    _EP = _new_group(group_ranks(mesh, "ep"), rank, _LOCAL_RANK, backend, "ep", use_device_comm=True)
    logger.info("rank %d: TP %d/%d PP %d/%d DP %d/%d TKNP %d/%d EP %d/%d", rank, _TP.rank_in_group,
                _TP.world_size, _PP.rank_in_group, _PP.world_size, _DP.rank_in_group, _DP.world_size,
                _TKNP.rank_in_group, _TKNP.world_size, _EP.rank_in_group, _EP.world_size)


def ensure_model_parallel_initialized(tp: int, pp: int, dp: int = 1, tknp: int = 1) -> None:
    if _TP is None:
        initialize_model_parallel(tp, pp, dp, tknp)
        return
    assert _TP.world_size == tp and _PP.world_size == pp, "model-parallel groups already exist with other sizes"


def model_parallel_is_initialized() -> bool:
    return _TP is not None


def _get(g: Optional[GroupCoordinator], name: str) -> GroupCoordinator:
    if g is None:
        # single-process convenience: layers can be built without any init call
        if not dist.is_initialized():
            initialize_model_parallel()
            return globals()[{"tp": "_TP", "pp": "_PP", "dp": "_DP", "ep": "_EP", "tknp": "_TKNP",
                              "world": "_WORLD"}[name]]
        raise RuntimeError(f"{name} group is not initialised")
    return g


def get_world_group() -> GroupCoordinator: return _get(_WORLD, "world")
def get_tp_group() -> GroupCoordinator: return _get(_TP, "tp")
def get_pp_group() -> GroupCoordinator: return _get(_PP, "pp")
def get_dp_group() -> GroupCoordinator: return _get(_DP, "dp")
def get_ep_group() -> GroupCoordinator: return _get(_EP, "ep")
def get_tknp_group() -> GroupCoordinator: return _get(_TKNP, "tknp")


def get_tensor_model_parallel_world_size() -> int: return get_tp_group().world_size
def get_tensor_model_parallel_rank() -> int: return get_tp_group().rank_in_group
def get_tknp_world_size() -> int: return get_tknp_group().world_size
def get_tknp_rank() -> int: return get_tknp_group().rank_in_group
def is_tknp_root() -> bool: return get_tknp_group().rank_in_group == 0


def tensor_model_parallel_all_reduce(x: torch.Tensor) -> torch.Tensor:
    return get_tp_group().all_reduce(x)


def tensor_model_parallel_all_gather(x: torch.Tensor, dim: int = -1) -> torch.Tensor:
    return get_tp_group().all_gather(x, dim)


def tensor_model_parallel_reduce_scatter(x: torch.Tensor, dim: int = 0) -> torch.Tensor:
    return get_tp_group().reduce_scatter(x, dim)


def tensor_model_parallel_gather(x: torch.Tensor, dst: int = 0, dim: int = -1):
    return get_tp_group().gather(x, dst, dim)


@contextlib.contextmanager
def graph_capture(device=None):
    """Enter the capture contexts of the TP and PP communicators."""
    ctx = GraphCaptureContext(torch.cuda.Stream(device=device))
    with get_tp_group().graph_capture(ctx), get_pp_group().graph_capture(ctx):
        yield ctx


def destroy_model_parallel() -> None:
    global _TP, _PP, _DP, _EP, _TKNP
    for g in (_TP, _PP, _DP, _EP, _TKNP):
        if g is not None and g.world_size > 1:
            g.destroy()
    _TP = _PP = _DP = _EP = _TKNP = None


def destroy_distributed_environment() -> None:
    global _WORLD
    if _WORLD is not None and _WORLD.world_size > 1:
        _WORLD.destroy()
    _WORLD = None
    if dist.is_initialized():
        dist.destroy_process_group()


def cleanup_dist_env_and_memory() -> None:
    destroy_model_parallel()
    destroy_distributed_environment()
    import gc
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
