"""Python side of the peer-memory all-reduce (parallel/csrc/custom_allreduce.hip).

Set-up: every rank allocates its arena + signal block, IPC handles are
all-gathered over Gloo, peers are mapped, and a self-test against a Gloo
all-reduce decides (collectively) whether the fast path is used at all.
Because inputs are copied into the arena inside the kernel, there is nothing to
register when hipGraphs are captured.
Reference: vllm/distributed/device_communicators/custom_all_reduce.py:49-304,
quick_all_reduce.py:46-278 (same role: take TP all-reduces off the ring).
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import socket
from typing import Optional

import torch
import torch.distributed as dist

from .. import envs
from ..logger import init_logger

logger = init_logger(__name__)
_HERE = os.path.dirname(os.path.abspath(__file__))
_DT = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}
_LIB = None


def comm_library() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        path = envs.VDA_COMM_LIBRARY or os.path.join(_HERE, "libvda_comm.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `python -m vllm_distributed_amd.ops.build`")
        lib = ctypes.CDLL(path)
        vp, sz, i32, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_ulonglong
        lib.vda_car_alloc.argtypes = [sz, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.c_char_p, ctypes.c_char_p]
        lib.vda_car_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
        lib.vda_car_close.argtypes = [vp]
        lib.vda_car_free.argtypes = [vp, vp]
        lib.vda_car_create.argtypes = [i32, i32, ctypes.POINTER(vp), ctypes.POINTER(vp), sz, u64]
        lib.vda_car_create.restype = vp
        lib.vda_car_destroy.argtypes = [vp]
        lib.vda_car_max_bytes.argtypes = [vp]
        lib.vda_car_max_bytes.restype = sz
        lib.vda_car_error.argtypes = [vp]
        lib.vda_car_allreduce.argtypes = [vp, vp, vp, sz, i32, i32, sz, vp]
        lib.vda_car_slice.argtypes = [vp, vp, vp, sz, i32, i32, vp]
        lib.vda_pipe_create.argtypes = [i32, i32, ctypes.POINTER(vp), ctypes.POINTER(vp), sz, u64]
        lib.vda_pipe_create.restype = vp
        lib.vda_pipe_destroy.argtypes = [vp]
        lib.vda_pipe_error.argtypes = [vp]
        lib.vda_pipe_send.argtypes = [vp, vp, sz, vp]
        lib.vda_pipe_recv.argtypes = [vp, vp, sz, vp]
        lib.vda_qr_half_bytes.argtypes = [sz, i32]
        lib.vda_qr_half_bytes.restype = sz
        lib.vda_qr_create.argtypes = [i32, i32, ctypes.POINTER(vp), ctypes.POINTER(vp), sz, i32, u64]
        lib.vda_qr_create.restype = vp
        lib.vda_qr_allreduce.argtypes = [vp, vp, vp, sz, i32, i32, vp]
        lib.vda_far_create.argtypes = [i32, i32, ctypes.POINTER(vp), ctypes.POINTER(vp), sz, u64]
        lib.vda_far_create.restype = vp
        lib.vda_far_run.argtypes = [vp, vp, vp, vp, vp, i32, i32, ctypes.c_float, ctypes.c_float, i32, vp]
        lib.vda_a2a_half_bytes.argtypes = [i32, i32, i32]
        lib.vda_a2a_half_bytes.restype = sz
        lib.vda_a2a_create.argtypes = [i32, i32, ctypes.POINTER(vp), ctypes.POINTER(vp), i32, i32, u64]
        lib.vda_a2a_create.restype = vp
        lib.vda_a2a_destroy.argtypes = [vp]
        lib.vda_a2a_error.argtypes = [vp]
        lib.vda_a2a_exchange.argtypes = [vp, vp, ctypes.c_long, vp, vp, i32, vp, vp, ctypes.c_long, i32, vp, vp, i32, i32,
                                         i32, i32, vp]
        lib.vda_a2a_combine.argtypes = [vp, vp, ctypes.c_long, vp, vp, ctypes.c_long, i32, vp, vp, vp, vp, ctypes.c_long,
                                        i32, i32, i32, i32, vp]
        _LIB = lib
    return _LIB


class CustomAllReduce:
    SUPPORTED_WORLD = (2, 3, 4, 5, 6, 7, 8)
    SELF_TEST_SIZES = (1024, 300_000, 3_000_000)

    def __init__(self, cpu_group, device: torch.device, arena_mb: int = 0):
        self.lib = comm_library()
        self.group = cpu_group
        self.device = device
        self.rank = dist.get_rank(cpu_group)
        self.world = dist.get_world_size(cpu_group)
        self.disabled = True
        self.oneshot_max = envs.VDA_CUSTOM_AR_ONESHOT_BYTES
        want = envs.VDA_CUSTOM_AR_MAX_BYTES
        # one staging half = input area + result area (1/world of it), two halves in the arena
        sweep = 64 * 512 * 16
        in_b = -(-want // sweep) * sweep
        half = in_b + (-(-(in_b // sweep) // self.world)) * sweep
        self.half_bytes = half
        pd, ps = self._exchange_arena(2 * half)
        # ~100 ns per spin: bound the wait by the configured time-out
        spins = int(envs.VDA_COMM_SPIN_TIMEOUT_MS * 1e4)
        self.ctx = self.lib.vda_car_create(self.rank, self.world, pd, ps, half, spins)
        if not self.ctx:
            raise RuntimeError("could not create the all-reduce context")
        self.max_bytes = int(self.lib.vda_car_max_bytes(self.ctx))
        dist.barrier(group=cpu_group)
        self.disabled = False

    def _exchange_arena(self, arena_bytes: int):
        """Allocate this rank's arena + signal block, trade IPC handles, map the peers'."""
        hs = self.lib.vda_car_handle_size()
        data, sig = ctypes.c_void_p(), ctypes.c_void_p()
        hd, hsg = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
        with torch.cuda.device(self.device):
            rc = self.lib.vda_car_alloc(arena_bytes, ctypes.byref(data), ctypes.byref(sig), hd, hsg)
        if rc:
            raise RuntimeError(f"arena allocation failed with HIP error {rc}")
        self._own = (data, sig)
        handles = [None] * self.world
        dist.all_gather_object(handles, (hd.raw[:hs], hsg.raw[:hs], socket.gethostname(), os.getpid()),
                               group=self.group)
        if len({h[2] for h in handles}) != 1:
            raise RuntimeError("ranks are on different nodes")
        self._opened = []
        pd = (ctypes.c_void_p * self.world)()
        ps = (ctypes.c_void_p * self.world)()
        with torch.cuda.device(self.device):
            for r, (h_d, h_s, _, _) in enumerate(handles):
                if r == self.rank:
                    pd[r], ps[r] = data, sig
                    continue
                for h, arr in ((h_d, pd), (h_s, ps)):
                    p = ctypes.c_void_p()
                    rc = self.lib.vda_car_open(h.ljust(64, b"\0"), ctypes.byref(p))
                    if rc:
                        raise RuntimeError(f"hipIpcOpenMemHandle failed with {rc} (peer {r})")
                    arr[r] = p
                    self._opened.append(p)
        return pd, ps

    @classmethod
    def maybe_create(cls, cpu_group, device: torch.device) -> Optional["CustomAllReduce"]:
        world = dist.get_world_size(cpu_group)
        if world not in cls.SUPPORTED_WORLD:
            return None
        ok, obj, err = 1, None, ""
        try:
            obj = cls(cpu_group, device)
            ok = int(obj._self_test())
        except Exception as e:  # noqa: BLE001 - any failure just disables the fast path
            ok, err = 0, str(e)
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=cpu_group)
        if int(flag.item()) == 0:
            if dist.get_rank(cpu_group) == 0:
                logger.warning("peer-memory all-reduce disabled (%s); using RCCL", err or "self-test failed on a peer")
            if obj is not None:
                obj.disabled = True
            return None
        if dist.get_rank(cpu_group) == 0:
            logger.info("peer-memory all-reduce enabled: %d ranks, up to %.0f MiB per launch, one-shot <= %d KiB",
                        world, obj.max_bytes / 2**20, obj.oneshot_max // 1024)
        return obj

    def _self_test(self) -> bool:
        for n in self.SELF_TEST_SIZES:
            x = ((torch.arange(n, device=self.device) % 7) + self.rank).to(torch.bfloat16)
            exp = ((torch.arange(n, device=self.device) % 7) * self.world
                   + sum(range(self.world))).to(torch.bfloat16)
            for _ in range(3):          # consecutive calls exercise both staging halves
                out = self._launch(x)
                torch.cuda.synchronize(self.device)
                if self.lib.vda_car_error(self.ctx) != 0 or not torch.equal(out, exp):
                    return False
            # the sliced forms between all-reduces: same counters, same halves
            m = n - n % (self.world * 64)
            part = self.reduce_scatter(x[:m].view(self.world * 8, -1))
            back = None if part is None else self.all_gather(part)
            torch.cuda.synchronize(self.device)
            if back is None or self.lib.vda_car_error(self.ctx) != 0 or not torch.equal(back.view(-1), exp[:m]):
                return False
        return True

    def should_use(self, x: torch.Tensor) -> bool:
        if self.disabled or x.dtype not in _DT or not x.is_contiguous():
            return False
        nb = x.numel() * x.element_size()
        return nb % 16 == 0 and 0 < nb

    def _launch(self, x: torch.Tensor) -> torch.Tensor:
        out = torch.empty_like(x)
        nb = x.numel() * x.element_size()
        st = torch.cuda.current_stream().cuda_stream
        off = 0
        # messages larger than the arena go through it in pieces (each piece keeps all links busy)
        while off < nb:
            n = min(self.max_bytes, nb - off)
            rc = self.lib.vda_car_allreduce(self.ctx, x.data_ptr() + off, out.data_ptr() + off, n, _DT[x.dtype], 0,
                                            self.oneshot_max, st)
            if rc:
                raise RuntimeError(f"peer-memory all-reduce launch failed with {rc}")
            off += n
        return out

    def all_reduce(self, x: torch.Tensor) -> Optional[torch.Tensor]:
        """Out-of-place sum over the group, or None when this tensor should take the RCCL path."""
        if not self.should_use(x):
            return None
        return self._launch(x)

    def _slice_ok(self, x: torch.Tensor, full_bytes: int) -> bool:
        return (not self.disabled and x.dtype in _DT and x.is_contiguous() and full_bytes > 0
                and full_bytes % (16 * self.world) == 0 and full_bytes <= self.max_bytes)

    def reduce_scatter(self, x: torch.Tensor) -> Optional[torch.Tensor]:
        """Sum over the group of x [R, ...], this rank keeping rows [rank R / world, (rank + 1) R / world): one
        kernel, one barrier (the first half of a two-shot all-reduce with contiguous slices).  None when the
        tensor should take the RCCL path."""
        nb = x.numel() * x.element_size()
        if x.dim() == 0 or x.shape[0] % self.world or not self._slice_ok(x, nb):
            return None
        if not getattr(self, "_slice_logged", False):
            self._slice_logged = True
            logger.info("peer-memory reduce-scatter / all-gather in use (%d ranks)", self.world)
        out = torch.empty((x.shape[0] // self.world,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        rc = self.lib.vda_car_slice(self.ctx, x.data_ptr(), out.data_ptr(), nb, _DT[x.dtype], 0,
                                    torch.cuda.current_stream().cuda_stream)
        if rc:
            raise RuntimeError(f"peer-memory reduce-scatter launch failed with {rc}")
        return out

    def all_gather(self, x: torch.Tensor) -> Optional[torch.Tensor]:
        """Concatenation over the group along dimension 0 of x [r, ...] (the same shape on every rank), or None
        when the tensor should take the RCCL path."""
        nb = x.numel() * x.element_size() * self.world
        if x.dim() == 0 or not self._slice_ok(x, nb):
            return None
        out = torch.empty((x.shape[0] * self.world,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        rc = self.lib.vda_car_slice(self.ctx, x.data_ptr(), out.data_ptr(), nb, _DT[x.dtype], 1,
                                    torch.cuda.current_stream().cuda_stream)
        if rc:
            raise RuntimeError(f"peer-memory all-gather launch failed with {rc}")
        return out

    def check_error(self) -> int:
        return int(self.lib.vda_car_error(self.ctx))

    @contextlib.contextmanager
    def capture(self):
        yield     # nothing to register: inputs are staged inside the kernel

    def close(self) -> None:
        if getattr(self, "ctx", None):
            self.lib.vda_car_destroy(self.ctx)
            self.ctx = None
            for p in self._opened:
                self.lib.vda_car_close(p)
            self.lib.vda_car_free(*self._own)
        self.disabled = True


class FusedAllReduceNorm(CustomAllReduce):
    """All-reduce + residual add + RMSNorm in one peer-memory kernel (``far_kernel``) for decode-sized
    batches: every TP all-reduce of a dense decoder layer is followed by exactly that, so the fusion
    saves one launch and one round trip of the [tokens, hidden] activations per reduction.  Rounding is
    the unfused path's, bit for bit.  Own arena and counters: its workgroups own whole rows, the plain
    kernel's own strided elements.  Reference role: vllm/compilation/collective_fusion.py
    (AllReduceFusedAddRMSNormPattern)."""
    SUPPORTED_WORLD = (2, 3, 4, 5, 6, 7, 8)

    def __init__(self, cpu_group, device: torch.device):
        self.lib = comm_library()
        self.group = cpu_group
        self.device = device
        self.rank = dist.get_rank(cpu_group)
        self.world = dist.get_world_size(cpu_group)
        self.disabled = True
        self.max_tokens = envs.VDA_FUSED_AR_NORM_MAX_TOKENS
        # 64 workgroup regions, each holding ceil(max_tokens / 64) rows of up to 16384 16-bit values
        region = -(-self.max_tokens // 64) * 16384 * 2
        self.half_bytes = 64 * region
        self.region_bytes = region
        pd, ps = self._exchange_arena(2 * self.half_bytes)
        spins = int(envs.VDA_COMM_SPIN_TIMEOUT_MS * 1e4)
        self.ctx = self.lib.vda_far_create(self.rank, self.world, pd, ps, self.half_bytes, spins)
        if not self.ctx:
            raise RuntimeError("could not create the fused all-reduce + norm context")
        self.max_bytes = self.half_bytes
        dist.barrier(group=cpu_group)
        self.disabled = False

    def fits(self, x: torch.Tensor) -> bool:
        """Pure function of shape and dtype: every rank answers alike."""
        if self.disabled or x.dim() != 2 or x.dtype not in (torch.bfloat16, torch.float16):
            return False
        M, H = x.shape
        return 0 < M <= self.max_tokens and H % 8 == 0 and H <= 16384 and -(-M // 64) * H * 2 <= self.region_bytes

    def run(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
            weight_bias: float = 0.0) -> Optional[torch.Tensor]:
        """residual += sum over ranks of x (in place); returns rmsnorm(residual) * (weight + weight_bias), or
        None when the shape is outside the kernel (nothing has been touched then)."""
        if not self.fits(x) or not (x.is_contiguous() and residual.is_contiguous() and weight.is_contiguous()) \
                or residual.shape != x.shape or residual.dtype != x.dtype or weight.dtype != x.dtype:
            return None
        out = torch.empty_like(x)
        rc = self.lib.vda_far_run(self.ctx, x.data_ptr(), residual.data_ptr(), weight.data_ptr(), out.data_ptr(),
                                  x.shape[0], x.shape[1], float(eps), float(weight_bias), _DT[x.dtype],
                                  torch.cuda.current_stream().cuda_stream)
        if rc:
            raise RuntimeError(f"fused all-reduce + norm launch failed with {rc}")
        return out

    def _self_test(self) -> bool:
        from .. import ops
        for M, H in ((1, 4096), (70, 8192), (130, 1024)):
            g = torch.Generator(device="cpu").manual_seed(1234 + self.rank)
            x = torch.randn(M, H, generator=g).to(self.device, torch.bfloat16)
            res0 = torch.randn(M, H, generator=torch.Generator(device="cpu").manual_seed(99)).to(self.device, torch.bfloat16)
            w = (torch.randn(H, generator=torch.Generator(device="cpu").manual_seed(7)) * 0.1 + 1).to(self.device, torch.bfloat16)
            tot = x.float().cpu()
            dist.all_reduce(tot, group=self.group)
            ref_res = res0.clone()
            ref_out = ops.rms_norm(tot.to(self.device, torch.bfloat16), w, 1e-5, ref_res)
            for _ in range(3):          # consecutive calls exercise both staging halves
                res = res0.clone()
                out = self.run(x, res, w, 1e-5)
                torch.cuda.synchronize(self.device)
                if out is None or self.lib.vda_car_error(self.ctx) != 0:
                    return False
                # gloo sums in another order than rank 0..n-1: allow one rounding step of the sum
                if not (torch.allclose(res.float(), ref_res.float(), atol=0.07, rtol=0.02)
                        and torch.allclose(out.float(), ref_out.float(), atol=0.07, rtol=0.02)):
                    return False
        return True

    @classmethod
    def maybe_create(cls, cpu_group, device: torch.device) -> Optional["FusedAllReduceNorm"]:
        world = dist.get_world_size(cpu_group)
        if world not in cls.SUPPORTED_WORLD or envs.VDA_FUSED_AR_NORM_MAX_TOKENS <= 0:
            return None
        ok, obj, err = 1, None, ""
        try:
            obj = cls(cpu_group, device)
            ok = int(obj._self_test())
        except Exception as e:  # noqa: BLE001 - any failure just disables this path
            ok, err = 0, str(e)
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=cpu_group)
        if int(flag.item()) == 0:
            if dist.get_rank(cpu_group) == 0:
                logger.warning("fused all-reduce + norm disabled (%s)", err or "self-test failed on a peer")
            if obj is not None:
                obj.disabled = True
            return None
        if dist.get_rank(cpu_group) == 0:
            logger.info("fused all-reduce + RMSNorm enabled: %d ranks, batches of up to %d tokens", world,
                        obj.max_tokens)
        return obj


class QuickAllReduce(CustomAllReduce):
    """Quantised two-shot all-reduce for bandwidth-bound messages: int8, int6 or int4 values with one fp32
    scale per 32 values cross xGMI (1.8x / 2.3x / 3.2x fewer bytes than bf16), sums are formed in fp32; the FP
    codec sends bf16 values through the same schedule.  Lossy by design; off unless
    ``VLLM_ROCM_QUICK_REDUCE_QUANTIZATION`` is INT8, INT6, INT4 or FP.  Used for messages of at least
    ``min_bytes`` (the reference's table, quick_all_reduce.py:50-59, has 1-4 MiB for fp16).
    Reference: vllm/distributed/device_communicators/quick_all_reduce.py:46-278."""
    SUPPORTED_WORLD = (2, 4, 8)
    MIN_BYTES = {2: 1 << 20, 4: 1 << 20, 8: 2 << 20}

    def __init__(self, cpu_group, device: torch.device, bits: int = 8, max_bytes: Optional[int] = None):
        self.lib = comm_library()
        self.group = cpu_group
        self.device = device
        self.rank = dist.get_rank(cpu_group)
        self.world = dist.get_world_size(cpu_group)
        self.bits = bits
        self.disabled = True
        mb = envs.VLLM_ROCM_QUICK_REDUCE_MAX_SIZE_BYTES_MB
        self.max_bytes = max_bytes or (mb * (1 << 20) if mb else 64 << 20)
        self.max_bytes -= self.max_bytes % 64
        self.min_bytes = self.MIN_BYTES.get(self.world, 1 << 20)
        half = int(self.lib.vda_qr_half_bytes(self.max_bytes, bits))
        pd, ps = self._exchange_arena(2 * half)
        spins = int(envs.VDA_COMM_SPIN_TIMEOUT_MS * 1e4)
        self.ctx = self.lib.vda_qr_create(self.rank, self.world, pd, ps, self.max_bytes, bits, spins)
        if not self.ctx:
            raise RuntimeError("could not create the quantised all-reduce context")
        dist.barrier(group=cpu_group)
        self.disabled = False

    @classmethod
    def maybe_create(cls, cpu_group, device: torch.device) -> Optional["QuickAllReduce"]:
        mode = envs.VLLM_ROCM_QUICK_REDUCE_QUANTIZATION
        bits = {"INT8": 8, "INT6": 6, "INT4": 4, "FP": 16}.get(mode)
        world = dist.get_world_size(cpu_group)
        if bits is None or world not in cls.SUPPORTED_WORLD:
            return None
        ok, obj, err = 1, None, ""
        try:
            obj = cls(cpu_group, device, bits)
            ok = int(obj._self_test())
        except Exception as e:  # noqa: BLE001 - any failure just disables this path
            ok, err = 0, str(e)
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=cpu_group)
        if int(flag.item()) == 0:
            if dist.get_rank(cpu_group) == 0:
                logger.warning("quantised all-reduce disabled (%s)", err or "self-test failed on a peer")
            if obj is not None:
                obj.disabled = True
            return None
        if dist.get_rank(cpu_group) == 0:
            logger.info("quantised all-reduce enabled: %d bits, %d ranks, messages of %d KiB .. %d MiB", bits, world,
                        obj.min_bytes // 1024, obj.max_bytes >> 20)
        return obj

    def _self_test(self) -> bool:
        g = torch.Generator(device="cpu").manual_seed(1234)       # same values on every rank
        for n in (32 * 1024, 1_000_000 // 32 * 32):
            base = torch.randn(n, generator=g).to(self.device)
            x = (base * (1 + self.rank)).to(torch.bfloat16)
            exp = base * sum(1 + r for r in range(self.world))
            for _ in range(3):
                out = self._launch(x).float()
                torch.cuda.synchronize(self.device)
                if self.lib.vda_car_error(self.ctx) != 0:
                    return False
                # two rounding steps, each at most half a quantisation step of the group maximum
                tol = exp.abs().max().item() * (2.2 / {16: 100, 8: 127, 6: 31, 4: 7}[self.bits])
                if not bool(((out - exp).abs() <= tol).all()):
                    return False
        return True

    def should_use(self, x: torch.Tensor) -> bool:
        if self.disabled or x.dtype not in (torch.bfloat16, torch.float16) or not x.is_contiguous():
            return False
        nb = x.numel() * x.element_size()
        return nb % 64 == 0 and nb >= self.min_bytes

    def _launch(self, x: torch.Tensor) -> torch.Tensor:
        out = torch.empty_like(x)
        nb = x.numel() * x.element_size()
        st = torch.cuda.current_stream().cuda_stream
        off = 0
        while off < nb:
            n = min(self.max_bytes, nb - off)
            rc = self.lib.vda_qr_allreduce(self.ctx, x.data_ptr() + off, out.data_ptr() + off, n, _DT[x.dtype],
                                           self.bits, st)
            if rc:
                raise RuntimeError(f"quantised all-reduce launch failed with {rc}")
            off += n
        return out


class PeerAllToAll(CustomAllReduce):
    """All-to-all-v of rows through peer memory (parallel/csrc/all2all.hip): every rank packs the rows it
    sends, grouped by destination, into its own IPC-exported arena and pulls its region of every peer's
    arena over all xGMI links at once.  ``cap`` rows per (source, destination) pair and ``row_bytes`` per
    row are fixed at construction; counts live on the device, so a call never syncs with the host and is
    hipGraph-capturable.  Results sit at fixed positions: ``recv[src * cap + j]`` is the j-th row rank
    ``src`` sent here.
    Reference role: vllm/distributed/device_communicators/all2all.py:77-264 (pplx / DeepEP managers)."""
    SUPPORTED_WORLD = (2, 3, 4, 5, 6, 7, 8)

    def __init__(self, cpu_group, device: torch.device, cap: int, row_bytes: int):
        self.lib = comm_library()
        self.group = cpu_group
        self.device = device
        self.rank = dist.get_rank(cpu_group)
        self.world = dist.get_world_size(cpu_group)
        self.cap, self.row_bytes = int(cap), int(row_bytes)
        self.disabled = True
        half = int(self.lib.vda_a2a_half_bytes(self.world, self.cap, self.row_bytes))
        pd, ps = self._exchange_arena(2 * half)
        spins = int(envs.VDA_COMM_SPIN_TIMEOUT_MS * 1e4)
        self.ctx = self.lib.vda_a2a_create(self.rank, self.world, pd, ps, self.cap, self.row_bytes, spins)
        if not self.ctx:
            raise RuntimeError("could not create the all-to-all context")
        dist.barrier(group=cpu_group)
        self.disabled = False

    @classmethod
    def maybe_create(cls, cpu_group, device: torch.device, cap: int, row_bytes: int) -> Optional["PeerAllToAll"]:
        world = dist.get_world_size(cpu_group)
        if world not in cls.SUPPORTED_WORLD:
            return None
        ok, obj, err = 1, None, ""
        try:
            obj = cls(cpu_group, device, cap, row_bytes)
            ok = int(obj._self_test())
        except Exception as e:  # noqa: BLE001 - any failure just disables the fast path
            ok, err = 0, str(e)
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=cpu_group)
        if int(flag.item()) == 0:
            if dist.get_rank(cpu_group) == 0:
                logger.warning("peer-memory all-to-all disabled (%s); using torch.distributed", err or "self-test failed on a peer")
            if obj is not None:
                obj.disabled = True
            return None
        if dist.get_rank(cpu_group) == 0:
            logger.info("peer-memory all-to-all enabled: %d ranks, %d rows of %d bytes per pair", world, cap, row_bytes)
        return obj

    def _self_test(self) -> bool:
        n = min(self.cap, 37)
        w = self.row_bytes // 2
        for it in range(3):                 # consecutive calls exercise both arena halves
            # rank r sends (d + 1 + it) % (n + 1) rows to rank d, row j filled with r * 1000 + d * 10 + j
            counts = [(d + 1 + it + self.rank) % (n + 1) for d in range(self.world)]
            rows, dest = [], []
            for d, c in enumerate(counts):
                for j in range(c):
                    rows.append(torch.full((w,), float((self.rank * 100 + d * 10 + j) % 251), dtype=torch.bfloat16))
                    dest.append(d)
            x = torch.stack(rows).to(self.device) if rows else torch.zeros((1, w), dtype=torch.bfloat16, device=self.device)
            order = torch.arange(len(dest), dtype=torch.int32, device=self.device)
            off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32, device=self.device)
            recv, _, rc = self.exchange(x, order, off)
            torch.cuda.synchronize(self.device)
            if self.lib.vda_a2a_error(self.ctx) != 0:
                return False
            rc = rc.cpu().tolist()
            for s in range(self.world):
                want = (self.rank + 1 + it + s) % (n + 1)
                if rc[s] != want:
                    return False
                for j in range(want):
                    if float(recv[s * self.cap + j, 0]) != float((s * 100 + self.rank * 10 + j) % 251):
                        return False
        return True

    def exchange(self, x: torch.Tensor, sorted_ids: torch.Tensor, offsets: torch.Tensor, k_div: int = 1,
                 meta: Optional[torch.Tensor] = None, cap: int = 0, meta_sub: int = 0, fill_meta: bool = False):
        """Send row ``sorted_ids[i] // k_div`` of ``x`` to rank d for offsets[d] <= i < offsets[d + 1].
        ``cap``: rows per (source, destination) pair of this call (default: the arena's capacity; the same
        on every rank).  The received meta values have ``meta_sub`` subtracted; with ``fill_meta`` the slots past
        a source's count read -1.  Returns (recv [world * cap, W], recv_meta int32 [world * cap] or None,
        recv_counts int32 [world])."""
        return self._run(x, sorted_ids, offsets, k_div, meta, 1, cap, meta_sub, fill_meta)

    def combine(self, y: torch.Tensor, counts: torch.Tensor, dest: torch.Tensor, slot: torch.Tensor,
                topk_w: torch.Tensor, cap: int = 0) -> torch.Tensor:
        """The way back of a dispatch with the weighting fused in (one launch): ``y`` [world * cap, W] returns to the
        senders like ``exchange_regions`` and out[t] = sum_j topk_w[t, j] * returned row of pair (t, j), where
        ``dest`` / ``slot`` int32 [T * k] name the owner rank (-1: none) and the pair's position in its region."""
        cap = int(cap) or self.cap
        T, k = topk_w.shape
        wb = y.shape[1] * y.element_size()
        if y.dtype not in (torch.bfloat16, torch.float16) or wb % 16 or wb > self.row_bytes or cap > self.cap \
                or y.shape[0] < self.world * cap or y.stride(1) != 1:
            raise ValueError("combine: rows do not fit the all-to-all arena")
        ret = torch.empty((self.world * cap, y.shape[1]), dtype=y.dtype, device=y.device)
        out = torch.empty((T, y.shape[1]), dtype=y.dtype, device=y.device)
        w = topk_w.contiguous().float()
        rcode = self.lib.vda_a2a_combine(
            self.ctx, y.data_ptr(), y.stride(0) * y.element_size(), counts.data_ptr(), ret.data_ptr(),
            ret.stride(0) * ret.element_size(), wb, dest.data_ptr(), slot.data_ptr(), w.data_ptr(), out.data_ptr(),
            out.stride(0) * out.element_size(), T, k, 0 if y.dtype == torch.bfloat16 else 1, cap,
            torch.cuda.current_stream().cuda_stream)
        if rcode:
            raise RuntimeError(f"peer-memory combine launch failed with {rcode}")
        return out

    def exchange_regions(self, y: torch.Tensor, counts: torch.Tensor, cap: int = 0):
        """``y`` [world * cap, W]: rows [d * cap, d * cap + counts[d]) go to rank d (the way back of an
        earlier ``exchange``: pass its recv_counts).  Returns recv [world * cap, W] and its counts."""
        recv, _, rc = self._run(y, None, counts, 1, None, 0, cap)
        return recv, rc

    def _run(self, x, sorted_ids, offs, k_div, meta, gather, cap=0, meta_sub=0, fill_meta=False):
        assert x.dim() == 2 and x.stride(1) == 1
        cap = int(cap) or self.cap
        wb = x.shape[1] * x.element_size()
        if wb % 16 or wb > self.row_bytes or (x.stride(0) * x.element_size()) % 16 or cap > self.cap:
            raise ValueError(f"{cap} rows of {wb} bytes do not fit the all-to-all arena ({self.cap} rows of "
                             f"{self.row_bytes} bytes per pair)")
        if not gather and x.shape[0] < self.world * cap:
            raise ValueError("region form needs world * cap rows")
        recv = torch.empty((self.world * cap, x.shape[1]), dtype=x.dtype, device=x.device)
        rmeta = torch.empty(self.world * cap, dtype=torch.int32, device=x.device) if meta is not None else None
        rc = torch.empty(self.world, dtype=torch.int32, device=x.device)
        rcode = self.lib.vda_a2a_exchange(
            self.ctx, x.data_ptr(), x.stride(0) * x.element_size(), sorted_ids.data_ptr() if sorted_ids is not None else None,
            offs.data_ptr(), int(k_div), meta.data_ptr() if meta is not None else None, recv.data_ptr(),
            recv.stride(0) * recv.element_size(), wb, rmeta.data_ptr() if rmeta is not None else None, rc.data_ptr(),
            gather, cap, int(meta_sub), int(bool(fill_meta)), torch.cuda.current_stream().cuda_stream)
        if rcode:
            raise RuntimeError(f"peer-memory all-to-all launch failed with {rcode}")
        return recv, rmeta, rc

    def check_error(self) -> int:
        return int(self.lib.vda_a2a_error(self.ctx))

    def close(self) -> None:
        if getattr(self, "ctx", None):
            self.lib.vda_a2a_destroy(self.ctx)
            self.ctx = None
            for p in self._opened:
                self.lib.vda_car_close(p)
            self.lib.vda_car_free(*self._own)
        self.disabled = True


class PeerPipe(CustomAllReduce):
    """Activations between neighbouring pipeline stages through peer memory (parallel/csrc/peer_pipe.hip): the sender's
    kernel writes into a two-slot ring in the receiver's arena and raises a flag, the receiver's kernel waits for
    it, copies the slot out and acknowledges.  Counters live on the device: a hop involves no host round trip, no
    pickled metadata and no RCCL call.  ``slot_bytes`` bounds one message.
    Reference role: vllm/distributed/parallel_state.py:629-749 (send_tensor_dict / recv_tensor_dict)."""

    def __init__(self, cpu_group, device: torch.device, slot_bytes: int):
        self.lib = comm_library()
        self.group = cpu_group
        self.device = device
        self.rank = dist.get_rank(cpu_group)
        self.world = dist.get_world_size(cpu_group)
        self.slot_bytes = max(-(-int(slot_bytes) // 4096) * 4096, 1 << 16)
        self.disabled = True
        pd, ps = self._exchange_arena(2 * self.slot_bytes)
        spins = int(envs.VDA_COMM_SPIN_TIMEOUT_MS * 1e4)
        self.ctx = self.lib.vda_pipe_create(self.rank, self.world, pd, ps, self.slot_bytes, spins)
        if not self.ctx:
            raise RuntimeError("could not create the pipe context")
        dist.barrier(group=cpu_group)
        self.disabled = False

    @classmethod
    def maybe_create(cls, cpu_group, device: torch.device, slot_bytes: int) -> Optional["PeerPipe"]:
        world = dist.get_world_size(cpu_group)
        if world not in cls.SUPPORTED_WORLD:
            return None
        ok, obj, err = 1, None, ""
        try:
            obj = cls(cpu_group, device, slot_bytes)
            ok = int(obj._self_test())
        except Exception as e:  # noqa: BLE001 - any failure just disables the fast path
            ok, err = 0, str(e)
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=cpu_group)
        if int(flag.item()) == 0:
            if dist.get_rank(cpu_group) == 0:
                logger.warning("peer-memory pipeline hand-off disabled (%s); using RCCL send / recv",
                               err or "self-test failed on a peer")
            if obj is not None:
                obj.disabled = True
            return None
        if dist.get_rank(cpu_group) == 0:
            logger.info("peer-memory pipeline hand-off enabled: %d stages, %.1f MiB per message", world,
                        obj.slot_bytes / 2**20)
        return obj

    def fits(self, x: torch.Tensor) -> bool:
        nb = x.numel() * x.element_size()
        return not self.disabled and x.is_contiguous() and nb % 16 == 0 and 0 < nb <= self.slot_bytes

    def send(self, x: torch.Tensor) -> None:
        """To the next rank of the ring."""
        rc = self.lib.vda_pipe_send(self.ctx, x.data_ptr(), x.numel() * x.element_size(),
                                    torch.cuda.current_stream().cuda_stream)
        if rc:
            raise RuntimeError(f"pipe send launch failed with {rc}")

    def recv(self, shape, dtype: torch.dtype) -> torch.Tensor:
        """From the previous rank of the ring."""
        out = torch.empty(shape, dtype=dtype, device=self.device)
        rc = self.lib.vda_pipe_recv(self.ctx, out.data_ptr(), out.numel() * out.element_size(),
                                    torch.cuda.current_stream().cuda_stream)
        if rc:
            raise RuntimeError(f"pipe receive launch failed with {rc}")
        return out

    def _self_test(self) -> bool:
        # a ring of messages of three sizes, five rounds: both slots, every workgroup count, slot reuse
        for it in range(5):
            for n in (512, 40_000, 600_000):
                n = min(n, self.slot_bytes // 2)            # elements of two bytes: at most the slot
                n -= n % 8
                x = ((torch.arange(n, device=self.device) % 13) + self.rank + it).to(torch.bfloat16)
                self.send(x)
                got = self.recv((n,), torch.bfloat16)
                torch.cuda.synchronize(self.device)
                prev = (self.rank - 1) % self.world
                exp = ((torch.arange(n, device=self.device) % 13) + prev + it).to(torch.bfloat16)
                if self.check_error() or not torch.equal(got, exp):
                    return False
        return True

    def check_error(self) -> int:
        return int(self.lib.vda_pipe_error(self.ctx))

    def close(self) -> None:
        if getattr(self, "ctx", None):
            self.lib.vda_pipe_destroy(self.ctx)
            self.ctx = None
            for p in self._opened:
                self.lib.vda_car_close(p)
            self.lib.vda_car_free(*self._own)
        self.disabled = True
