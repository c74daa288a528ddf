"""Prefill-GEMM probe: hipBLASLt default heuristic vs TunableOp at M = 16384 (Llama-3-70B shapes, TP 1 and 8)."""
import os
import sys
import time

import torch

dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
shapes = [(8192, 10240), (8192, 8192), (8192, 57344), (28672, 8192),          # TP1
          (8192, 1280), (1024, 8192), (8192, 7168), (3584, 8192)]             # TP8


def bench(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters


res = {}
xs, ws = {}, {}
for (K, N) in shapes:
    xs[K] = xs.get(K) if K in xs else torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    ws[(K, N)] = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    res[(K, N, "d")] = bench(lambda: torch.nn.functional.linear(xs[K], ws[(K, N)]))
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(300)
torch.cuda.tunable.set_max_tuning_iterations(10)
torch.cuda.tunable.set_filename("gpurun_out/tunableop_prefill.csv")
for (K, N) in shapes:
    t = time.time()
    torch.nn.functional.linear(xs[K], ws[(K, N)])
    torch.cuda.synchronize()
    res[(K, N, "ts")] = time.time() - t
    res[(K, N, "t")] = bench(lambda: torch.nn.functional.linear(xs[K], ws[(K, N)]))
for (K, N) in shapes:
    fl = 2.0 * M * K * N
    print(f"M={M} K={K:6d} N={N:6d} default {res[(K, N, 'd')] * 1e3:7.3f} ms ({fl / res[(K, N, 'd')] / 1e15:.2f} PF/s) | "
          f"tuned {res[(K, N, 't')] * 1e3:7.3f} ms ({fl / res[(K, N, 't')] / 1e15:.2f} PF/s), tuning {res[(K, N, 'ts')]:.1f} s",
          flush=True)
torch.cuda.tunable.write_file()
