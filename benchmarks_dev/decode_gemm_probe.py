"""Decode-batch GEMMs (16 <= M <= 256) of Llama-3-70B / Llama-3-8B layers: the weight-streaming split-K
MFMA kernel (ops/csrc/gemm_decode.hip) against the library GEMM.  Every call uses another copy of the
weight matrix, enough copies to exceed the 256 MB Infinity Cache, so that weights come from HBM as they
do inside a model.  Prints microseconds per call, the weight stream in TB/s and the layer totals.

    python benchmarks_dev/decode_gemm_probe.py [70b|8b|all]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd import ops  # noqa: E402

DEV = "cuda"
SHAPES = {
    "70b": [("qkv", 10240, 8192), ("o", 8192, 8192), ("gate_up", 57344, 8192), ("down", 8192, 28672)],
    "8b": [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336)],
}


def timeit(fn, n_copies, reps=24, warm=4):
    for i in range(warm):
        fn(i % n_copies)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn(i % n_copies)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def run(model: str) -> None:
    print(f"\n{model}: us per call  lib / decode_gemm   (weight stream of decode_gemm, max |err| vs fp32)")
    for M in (16, 32, 64, 128, 192, 256):
        tot = [0.0, 0.0]
        for name, N_, K in SHAPES[model]:
            nbytes = N_ * K * 2
            copies = max(2, -(-(600 << 20) // nbytes))
            ws = [torch.randn(N_, K, device=DEV, dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
            x = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
            ref = x.float() @ ws[0].float().t()
            got = ops.decode_gemm(x, ws[0]).float()
            err = (got - ref).abs().max().item()
            lib_err = (torch.nn.functional.linear(x, ws[0]).float() - ref).abs().max().item()
            t0 = timeit(lambda i: torch.nn.functional.linear(x, ws[i]), copies)
            t1 = timeit(lambda i: ops.decode_gemm(x, ws[i]), copies)
            print(f"M={M:>4} {name:>8}: {t0 * 1e6:>7.1f} {t1 * 1e6:>7.1f}   {nbytes / t1 / 1e12:.2f} TB/s  "
                  f"err {err:.3f} (lib {lib_err:.3f})", flush=True)
            tot[0] += t0
            tot[1] += t1
            del ws
        print(f"M={M:>4}    layer: {tot[0] * 1e6:>7.1f} {tot[1] * 1e6:>7.1f}", flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    for m in ("70b", "8b"):
        if which in ("all", m):
            run(m)
