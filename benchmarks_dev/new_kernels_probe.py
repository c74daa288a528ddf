"""Achieved bandwidth of the W4A16 skinny GEMM, the MLA decode kernel, the state-space kernels and the
multi-LoRA update at serving shapes.
    python benchmarks_dev/new_kernels_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd import ops  # noqa: E402

DEV, DT = "cuda", torch.bfloat16


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


print("== W4A16 skinny GEMM (Llama-3-70B shapes, group 128) vs bf16 skinny / library GEMM")
for M in (1, 8, 16, 32):
    for N, K in ((10240, 8192), (57344, 8192), (8192, 28672)):
        qw = torch.randint(-2**31, 2**31 - 1, (N, K // 8), device=DEV, dtype=torch.int32)
        sc = (torch.rand(N, K // 128, device=DEV) * 0.01).to(DT)
        z = torch.full((N, K // 128), 8.0, device=DEV, dtype=DT)
        x = torch.randn(M, K, device=DEV, dtype=DT)
        w = torch.randn(N, K, device=DEV, dtype=DT)
        t4 = timeit(lambda: ops.w4_linear(x, qw, sc, z, 128))
        t16 = timeit(lambda: ops.linear(x, w))
        b4 = N * K / 2 + N * (K // 128) * 4
        print(f"M={M:3d} N={N:6d} K={K:6d}: w4 {t4 * 1e6:7.1f} us ({b4 / t4 / 1e12:5.2f} TB/s of packed bytes)   "
              f"bf16 {t16 * 1e6:7.1f} us ({N * K * 2 / t16 / 1e12:5.2f} TB/s)   speed-up {t16 / t4:4.2f}x")
        del qw, sc, z, x, w

print("== MLA decode (DK 576 / DV 512), block 16")
for H, S, L in ((16, 64, 4096), (128, 32, 4096), (16, 256, 1024), (128, 8, 16384)):
    BS = 16
    nb_seq = (L + BS - 1) // BS
    nblocks = S * nb_seq + 1
    kc = torch.randn(nblocks, BS, 576, device=DEV, dtype=DT)
    vt = torch.randn(nblocks, 512, BS, device=DEV, dtype=DT)
    bt = (torch.arange(S * nb_seq, device=DEV, dtype=torch.int32) + 1).view(S, nb_seq)
    seq = torch.full((S,), L, device=DEV, dtype=torch.int32)
    qsl = torch.arange(S + 1, device=DEV, dtype=torch.int32)
    q = torch.randn(S, H, 576, device=DEV, dtype=DT)
    out = torch.empty(S, H, 512, device=DEV, dtype=DT)
    for ns in (1, ops.decode_num_splits(S * max(H // 16, 1), 1, L)):
        t = timeit(lambda: ops.mla_attention(out, q, kc, vt, bt, seq, qsl, 0.04, 1, ns))
        byt = S * L * (576 + 512) * 2
        fl = 2 * S * H * L * (576 + 512)
        print(f"H={H:3d} S={S:3d} L={L:5d} splits={ns:2d}: {t * 1e6:8.1f} us  {byt / t / 1e12:5.2f} TB/s of latent cache, "
              f"{fl / t / 1e12:6.1f} TFLOP/s")
    del kc, vt

print("== state-space kernels (Mamba-1 2.8b: d_inner 5120, N 16; Mamba-2: 128 heads x 64, N 128)")
for S, Lq in ((256, 1), (4, 2048)):
    T = S * Lq
    qsl = (torch.arange(S + 1, device=DEV, dtype=torch.int32) * Lq)
    seq = torch.full((S,), Lq + 7, device=DEV, dtype=torch.int32)
    bt = torch.arange(S, device=DEV, dtype=torch.int32).view(S, 1)
    C = 5120
    x = torch.randn(T, C, device=DEV, dtype=DT)
    w = torch.randn(C, 4, device=DEV, dtype=DT)
    cs = torch.zeros(S, C, 3, device=DEV, dtype=DT)
    t = timeit(lambda: ops.causal_conv1d(x, w, None, cs, qsl, seq, bt, Lq))
    print(f"conv1d  S={S:3d} L={Lq:4d}: {t * 1e6:8.1f} us  ({2 * T * C * 2 / t / 1e12:5.2f} TB/s in+out)")
    A = -torch.rand(C, 16, device=DEV)
    dl = torch.randn(T, C, device=DEV, dtype=DT)
    B = torch.randn(T, 16, device=DEV, dtype=DT)
    Cm = torch.randn(T, 16, device=DEV, dtype=DT)
    ss = torch.zeros(S, C, 16, device=DEV)
    t = timeit(lambda: ops.selective_scan(x, dl, A, B, Cm, None, x, None, ss, qsl, seq, bt, max_query_len=Lq))
    print(f"scan-1  S={S:3d} L={Lq:4d}: {t * 1e6:8.1f} us  ({(4 * T * C * 2 + 2 * S * C * 16 * 4) / t / 1e12:5.2f} TB/s)")
    C2, Hh, N2 = 8192, 128, 128
    x2 = torch.randn(T, C2, device=DEV, dtype=DT)
    d2 = torch.randn(T, Hh, device=DEV, dtype=DT)
    A2 = -torch.rand(Hh, device=DEV)
    B2 = torch.randn(T, N2, device=DEV, dtype=DT)
    ss2 = torch.zeros(S, C2, N2, device=DEV)
    t = timeit(lambda: ops.selective_scan(x2, d2, A2, B2, B2, None, x2, None, ss2, qsl, seq, bt, max_query_len=Lq))
    print(f"scan-2  S={S:3d} L={Lq:4d}: {t * 1e6:8.1f} us  ({(3 * T * C2 * 2 + 2 * S * C2 * N2 * 4) / t / 1e12:5.2f} TB/s)")

print("== multi-LoRA update, rank 16, 4 adapters")
for T, K, N in ((64, 8192, 10240), (256, 8192, 8192)):
    x = torch.randn(T, K, device=DEV, dtype=DT)
    y = torch.randn(T, N, device=DEV, dtype=DT)
    A = torch.randn(4, 16, K, device=DEV, dtype=DT)
    B = torch.randn(4, N, 16, device=DEV, dtype=DT)
    ts = torch.randint(0, 4, (T,), device=DEV, dtype=torch.int32)
    t = timeit(lambda: ops.lora_update(y, x, A, B, ts))
    print(f"T={T:3d} K={K} N={N}: {t * 1e6:7.1f} us")
