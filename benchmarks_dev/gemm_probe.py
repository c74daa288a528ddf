"""Decode-GEMM probe: hipBLASLt default heuristic vs PyTorch TunableOp vs the skinny MFMA kernel."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vllm_distributed_amd import ops

dev = "cuda"
shapes = [(8192, 10240), (8192, 8192), (8192, 57344), (28672, 8192)]      # (K, N) of Llama-3-70B TP1
Ms = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,64,128,256").split(",")]


def bench(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e6


ws = {s: torch.randn(s[1], s[0], device=dev, dtype=torch.bfloat16) * 0.02 for s in shapes}
# rotate through copies so the weights do not sit in the 256 MB infinity cache
res = {}
for M in Ms:
    for (K, N) in shapes:
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = ws[(K, N)]
        t0 = bench(lambda: torch.nn.functional.linear(x, w))
        res[(M, K, N, "default")] = t0
        if M <= 64:
            res[(M, K, N, "skinny")] = bench(lambda: ops.skinny_gemm(x, w))
if os.environ.get("TUNE", "1") == "1":
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.tuning_enable(True)
    torch.cuda.tunable.set_max_tuning_duration(50)
    torch.cuda.tunable.set_max_tuning_iterations(20)
    for M in Ms:
        for (K, N) in shapes:
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            w = ws[(K, N)]
            t = time.time()
            torch.nn.functional.linear(x, w)
            torch.cuda.synchronize()
            tune_s = time.time() - t
            res[(M, K, N, "tuned")] = bench(lambda: torch.nn.functional.linear(x, w))
            res[(M, K, N, "tune_s")] = tune_s
for M in Ms:
    for (K, N) in shapes:
        floor = K * N * 2 / 6.3e12 * 1e6
        row = f"M={M:4d} K={K:6d} N={N:6d} floor {floor:6.1f} us | default {res[(M, K, N, 'default')]:7.1f}"
        if (M, K, N, "skinny") in res:
            row += f" | skinny {res[(M, K, N, 'skinny')]:7.1f}"
        if (M, K, N, "tuned") in res:
            row += f" | tuned {res[(M, K, N, 'tuned')]:7.1f} (tuning took {res[(M, K, N, 'tune_s')]:.1f} s)"
        print(row, flush=True)
