"""Prefill GEMM: does the library run faster with the weight stored [K, N] instead of [N, K]?"""
import time

import torch

dev = "cuda"


def bench(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


for M, K, N in [(16384, 8192, 57344), (16384, 28672, 8192), (16384, 8192, 10240), (16384, 8192, 8192), (8192, 8192, 57344),
                (32768, 8192, 57344)]:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16) * 0.1
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    wt = w.t().contiguous()
    fl = 2.0 * M * K * N
    a = bench(lambda: torch.nn.functional.linear(x, w))
    b = bench(lambda: torch.matmul(x, wt))
    print(f"M={M} K={K} N={N}: W[N,K] {a * 1e3:7.3f} ms ({fl / a / 1e15:.2f} PF/s) | W[K,N] {b * 1e3:7.3f} ms ({fl / b / 1e15:.2f} PF/s)",
          flush=True)
    del x, w, wt
