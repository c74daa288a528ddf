"""Throughput of the LDS-tiled MFMA GEMM (ops/csrc/gemm_tile.hip): dense fp8 against hipBLASLt's
``torch._scaled_mm`` and against the bf16 library GEMM, and the Mixtral-8x7B expert layer at prefill size
(tiled fp8 / tiled bf16 / per-expert library GEMMs).  Random operands (zero-filled ones read high).

    python benchmarks_dev/tile_gemm_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd import ops  # noqa: E402
from vllm_distributed_amd.ops import reference as R  # noqa: E402

FP8 = torch.float8_e4m3fn
DEV = "cuda"


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def dense():
    print("dense fp8 GEMM  y[M,N] = x[M,K] w[N,K]^T   (TFLOP/s)")
    print(f"{'M':>6} {'N':>6} {'K':>6} {'tile fp8':>9} {'scaled_mm':>10} {'bf16 lib':>9}")
    for M, N_, K in [(4096, 4096, 4096), (8192, 8192, 8192), (16384, 28672, 8192), (16384, 8192, 28672),
                     (16384, 10240, 8192), (512, 8192, 8192), (256, 28672, 8192)]:
        x = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
        w = torch.randn(N_, K, device=DEV, dtype=torch.bfloat16) * 0.05
        ws = (w.abs().max().float() / 448).reshape(1)
        wq = (w.float() / ws).to(FP8)
        xq, xs = ops.per_token_fp8_quant(x)
        fl = 2.0 * M * N_ * K
        t1 = timeit(lambda: ops.fp8_gemm(xq, wq, xs, ws, torch.bfloat16))
        sb = ws.reshape(1, 1).expand(1, N_).contiguous()
        try:
            t2 = timeit(lambda: torch._scaled_mm(xq, wq.t(), scale_a=xs.view(M, 1), scale_b=sb, out_dtype=torch.bfloat16))
        except Exception as ex:   # noqa: BLE001
            print("scaled_mm failed:", str(ex)[:80])
            t2 = float("nan")
        t3 = timeit(lambda: torch.nn.functional.linear(x, w))
        print(f"{M:>6} {N_:>6} {K:>6} {fl / t1 / 1e12:>9.0f} {fl / t2 / 1e12:>10.0f} {fl / t3 / 1e12:>9.0f}", flush=True)


def moe():
    print("\nMixtral-8x7B expert layer (E=8, top-2, H=4096, I=14336): ms per layer")
    E, k, H, I = 8, 2, 4096, 14336
    w13 = torch.randn(E, 2 * I, H, device=DEV, dtype=torch.bfloat16) * 0.03
    w2 = torch.randn(E, H, I, device=DEV, dtype=torch.bfloat16) * 0.03
    s13 = w13.float().abs().amax(dim=(1, 2)) / 448
    s2 = w2.float().abs().amax(dim=(1, 2)) / 448
    q13 = (w13.float() / s13.view(E, 1, 1)).to(FP8)
    q2 = (w2.float() / s2.view(E, 1, 1)).to(FP8)
    os.environ["VDA_MOE_GROUPED_MAX_PAIRS"] = "0"
    print(f"{'tokens':>7} {'tile bf16':>10} {'TF/s':>6} {'tile fp8':>9} {'TF/s':>6} {'per-expert lib':>15} {'TF/s':>6}")
    for T in (512, 2048, 8192, 16384):
        x = torch.randn(T, H, device=DEV, dtype=torch.bfloat16)
        tw, ids = R.moe_topk_softmax(torch.randn(T, E, device=DEV), k, True)
        fl = 2.0 * T * k * H * I * 3
        t1 = timeit(lambda: ops.fused_moe(x, w13, w2, tw, ids), reps=5, warm=2)
        t2 = timeit(lambda: ops.fused_moe(x, q13, q2, tw, ids, s13, s2), reps=5, warm=2)

        def lib():
            flat = ids.reshape(-1)
            order = torch.argsort(flat, stable=True)
            counts = torch.bincount(flat, minlength=E).tolist()
            xg = x[order // k]
            s = 0
            for e, c in enumerate(counts):
                if c:
                    h = ops.act_and_mul(torch.mm(xg[s:s + c], w13[e].t()))
                    torch.mm(h, w2[e].t())
                s += c
        t3 = timeit(lib, reps=5, warm=2)
        print(f"{T:>7} {t1 * 1e3:>10.2f} {fl / t1 / 1e12:>6.0f} {t2 * 1e3:>9.2f} {fl / t2 / 1e12:>6.0f} "
              f"{t3 * 1e3:>15.2f} {fl / t3 / 1e12:>6.0f}", flush=True)


def moe_decode():
    """Mixtral-8x7B expert layer with fp8 weights at decode batch sizes: the row-split streaming kernel
    (weights widened in registers, 16-bit MFMA) against the tiled kernel (activations quantised, fp8 MFMA)."""
    print("\nMixtral-8x7B expert layer, fp8 weights, decode batches: us per layer (grouped rows W8A16 / tiled fp8)")
    E, k, H, I = 8, 2, 4096, 14336
    copies = 3                                   # 1.4 GB of weights each: never resident in the Infinity Cache
    q13 = [(torch.randn(E, 2 * I, H, device=DEV, dtype=torch.bfloat16) * 0.03).to(FP8) for _ in range(copies)]
    q2 = [(torch.randn(E, H, I, device=DEV, dtype=torch.bfloat16) * 0.03).to(FP8) for _ in range(copies)]
    s13 = torch.full((E,), 0.01, device=DEV)
    s2 = torch.full((E,), 0.01, device=DEV)
    for T in (16, 32, 64, 128, 256, 512, 1024):
        x = torch.randn(T, H, device=DEV, dtype=torch.bfloat16)
        tw, ids = R.moe_topk_softmax(torch.randn(T, E, device=DEV), k, True)
        it = [0]

        def run():
            i = it[0] = (it[0] + 1) % copies
            ops.fused_moe(x, q13[i], q2[i], tw, ids, s13, s2)
        os.environ["VDA_MOE_GROUPED_MAX_PAIRS"] = os.environ["VDA_MOE_GROUPED_MAX_PAIRS_FP8"] = "1000000"
        t1 = timeit(run, reps=12, warm=3)
        os.environ["VDA_MOE_GROUPED_MAX_PAIRS"] = "0"
        t2 = timeit(run, reps=12, warm=3)
        nbytes = 3.0 * E * H * I
        print(f"T={T:>5}: {t1 * 1e6:>8.1f} {t2 * 1e6:>8.1f}   best {nbytes / min(t1, t2) / 1e12:.2f} TB/s", flush=True)


def decode():
    """Llama-3-70B projections at decode batch sizes: library GEMM against the tiled 16-bit kernel."""
    print("\nLlama-3-70B projections, bf16, us per call (lib / tile128 / tile64) and weight TB/s of the best")
    shapes = [("qkv", 10240, 8192), ("o", 8192, 8192), ("gate_up", 57344, 8192), ("down", 8192, 28672)]
    for M in (64, 128, 256):
        tot = [0.0, 0.0, 0.0]
        for name, N_, K in shapes:
            x = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
            w = torch.randn(N_, K, device=DEV, dtype=torch.bfloat16) * 0.02
            t0 = timeit(lambda: torch.nn.functional.linear(x, w), reps=20)
            t1 = timeit(lambda: ops.tile_linear(x, w, None, 128), reps=20)
            t2 = timeit(lambda: ops.tile_linear(x, w, None, 64), reps=20)
            best = min(t0, t1, t2)
            print(f"M={M:>4} {name:>8}: {t0 * 1e6:>7.1f} {t1 * 1e6:>7.1f} {t2 * 1e6:>7.1f}   {N_ * K * 2 / best / 1e12:.2f} TB/s", flush=True)
            for i, t in enumerate((t0, t1, t2)):
                tot[i] += t
        print(f"M={M:>4}    layer: {tot[0] * 1e6:>7.1f} {tot[1] * 1e6:>7.1f} {tot[2] * 1e6:>7.1f}", flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "dense"):
        dense()
    if which in ("all", "moe"):
        moe()
    if which in ("all", "moe_decode"):
        moe_decode()
    if which in ("all", "decode"):
        decode()
