"""Ping-pong MFMA GEMM (ops/csrc/gemm_dense.hip) against the library GEMM.

    python benchmarks_dev/dense_gemm_probe.py check            correctness over tile shapes, edges, epilogues, K splits
    python benchmarks_dev/dense_gemm_probe.py prefill [70b|8b] M = 2304 / 4096 / 16384, PFLOP/s of both
    python benchmarks_dev/dense_gemm_probe.py decode [70b|8b]  M = 16 .. 256, microseconds and weight stream

Timed calls rotate over enough weight copies to exceed the 256 MB Infinity Cache (decode) and use uniform random
operands in [-1, 1) scaled to keep outputs O(1) (zero or constant data would raise the clock and flatter both).
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


def rnd(*shape, dtype=torch.bfloat16, scale=1.0):
    return ((torch.rand(*shape, device=DEV) * 2 - 1) * scale).to(dtype)


def timeit(fn, n_copies, reps=20, warm=3):
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


def check() -> int:
    bad = 0
    cases = []
    for M in (1, 16, 31, 33, 64, 100, 128, 129, 200, 256, 257, 300, 513, 1000):
        for N_, K in ((256, 256), (512, 1024), (264, 96), (1000, 512), (4096, 4096)):
            cases.append((M, N_, K))
    for dtype in (torch.bfloat16, torch.float16):
        for (M, N_, K) in cases:
            x = rnd(M, K, dtype=dtype)
            w = rnd(N_, K, dtype=dtype, scale=K ** -0.5)
            ref = x.float() @ w.float().t()
            for wt, variant in ((2, 1), (4, 1), (4, 2)):
                for splits in (1, 0, 3):
                    got = ops.dense_gemm(x, w, wt=wt, splits=splits, variant=variant).float()
                    err = (got - ref).abs().max().item()
                    tol = 0.02 + 0.01 * ref.abs().max().item()
                    if not err < tol:
                        bad += 1
                        print(f"FAIL plain {dtype} M={M} N={N_} K={K} wt={wt} splits={splits} variant={variant}: "
                              f"err {err:.4f}", flush=True)
            if dtype == torch.bfloat16 and N_ % 8 == 0:
                b = rnd(N_, dtype=dtype)
                res = rnd(M, N_, dtype=dtype)
                for splits in (1, 2):
                    got = ops.dense_gemm(x, w, bias=b, residual=res, splits=splits).float()
                    err = (got - (ref + b.float() + res.float())).abs().max().item()
                    if not err < 0.05:
                        bad += 1
                        print(f"FAIL bias+res M={M} N={N_} K={K} splits={splits}: err {err:.4f}", flush=True)
                i_ = N_ // 2
                gref = torch.nn.functional.silu(ref[:, :i_]) * ref[:, i_:]
                for wt, variant in ((2, 1), (4, 1), (4, 2)):
                    for splits in (1, 2):
                        got = ops.dense_gemm(x, w, epilogue="gate_up", wt=wt, splits=splits, variant=variant).float()
                        err = (got - gref).abs().max().item()
                        if not err < 0.03 + 0.01 * gref.abs().max().item():
                            bad += 1
                            print(f"FAIL gate_up M={M} N={N_} K={K} wt={wt} splits={splits} variant={variant}: "
                                  f"err {err:.4f}", flush=True)
        print(f"checked {dtype}: failures so far {bad}", flush=True)
    # strided activations (a slice of a wider buffer) and a strided output
    x_full = rnd(300, 1024 + 64)
    w = rnd(512, 1024, scale=1 / 32)
    xs = x_full[:, 64:]
    out_full = torch.zeros(300, 768, device=DEV, dtype=torch.bfloat16)
    ops.dense_gemm(xs, w, out=out_full[:, 128:640])
    err = (out_full[:, 128:640].float() - xs.float() @ w.float().t()).abs().max().item()
    if not err < 0.05 or out_full[:, :128].abs().max().item() != 0 or out_full[:, 640:].abs().max().item() != 0:
        bad += 1
        print(f"FAIL strided: err {err:.4f}", flush=True)
    # exact integer data with an asymmetric pattern: catches transposed or permuted fragments
    x = torch.zeros(64, 64, device=DEV, dtype=torch.bfloat16)
    x[torch.arange(64), torch.arange(64)] = 1
    w = (torch.arange(128 * 64, device=DEV).reshape(128, 64) % 251).to(torch.bfloat16)
    got = ops.dense_gemm(x, w).float()
    if not torch.equal(got, w.float().t()):
        bad += 1
        print("FAIL identity", flush=True)
    print("check:", "OK" if bad == 0 else f"{bad} failures", flush=True)
    return bad


def prefill(model: str) -> None:
    print(f"\n{model} prefill: PFLOP/s  lib / dense_gemm 32-unit ring / 64-unit form / gate_up fused (64-unit)  "
          f"[lib + act_and_mul us]")
    for M in (2304, 4096, 16384):
        tl = td = 0.0
        for name, N_, K in SHAPES[model]:
            x = rnd(M, K)
            ws = [rnd(N_, K, scale=K ** -0.5) for _ in range(2)]
            fl = 2.0 * M * N_ * K
            reps = 6 if M >= 16384 else 12
            t0 = timeit(lambda i: torch.nn.functional.linear(x, ws[i]), 2, reps=reps)
            t1 = timeit(lambda i: ops.dense_gemm(x, ws[i], variant=1), 2, reps=reps)
            t1b = timeit(lambda i: ops.dense_gemm(x, ws[i], variant=2), 2, reps=reps)
            line = f"M={M:>6} {name:>8}: {fl / t0 / 1e15:6.3f} {fl / t1 / 1e15:6.3f} {fl / t1b / 1e15:6.3f}"
            t1 = min(t1, t1b)
            if name == "gate_up":
                t2 = timeit(lambda i: ops.dense_gemm(x, ws[i], epilogue="gate_up", variant=2), 2, reps=reps)
                t3 = timeit(lambda i: ops.silu_and_mul(torch.nn.functional.linear(x, ws[i])), 2, reps=reps)
                line += f" {fl / t2 / 1e15:6.3f}   [{t3 * 1e6:.0f} us vs fused {t2 * 1e6:.0f} us]"
                tl += t3
                td += t2
            else:
                tl += t0
                td += t1
            print(line, flush=True)
            del ws
        print(f"M={M:>6}    layer: lib {tl * 1e6:8.0f} us   dense {td * 1e6:8.0f} us", flush=True)


def _tuned_library(on: bool) -> None:
    """Library GEMM with the committed algorithm selections (what the engine runs) or with its own heuristic."""
    import torch.cuda.tunable as tn
    if on:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vllm_distributed_amd", "ops",
                            "tuned", "gfx950_gemm.csv")
        tn.enable(True)
        tn.tuning_enable(False)
        tn.record_untuned_enable(False)
        if not tn.read_file(path):
            print("selection table rejected (library versions differ)", flush=True)
            tn.enable(False)
    else:
        tn.enable(False)


def decode(model: str) -> None:
    print(f"\n{model} decode: us per call  library (heuristic) / library (selection table) / in-tree wt4 / wt2   "
          f"(weight stream of the in-tree best); gate_up rows: library + act_and_mul kernel against the fused epilogue")
    for M in (1, 4, 8, 16, 32, 64, 128, 192, 256):
        tot = [0.0, 0.0, 0.0]
        for name, N_, K in SHAPES[model]:
            nbytes = N_ * K * 2
            copies = max(2, -(-(600 << 20) // nbytes))
            ws = [rnd(N_, K, scale=K ** -0.5) for _ in range(copies)]
            x = rnd(M, K)
            epi = "gate_up" if name == "gate_up" else "plain"
            if epi == "gate_up":
                lib = lambda i: ops.silu_and_mul(torch.nn.functional.linear(x, ws[i]))   # noqa: E731
            else:
                lib = lambda i: torch.nn.functional.linear(x, ws[i])                    # noqa: E731
            t0 = timeit(lib, copies)
            _tuned_library(True)
            t0t = timeit(lib, copies)
            _tuned_library(False)
            t4 = timeit(lambda i: ops.dense_gemm(x, ws[i], epilogue=epi, wt=4), copies)
            t2 = timeit(lambda i: ops.dense_gemm(x, ws[i], epilogue=epi, wt=2), copies)
            tb = min(t4, t2)
            print(f"M={M:>4} {name:>8}: {t0 * 1e6:>7.1f} {t0t * 1e6:>7.1f} {t4 * 1e6:>7.1f} {t2 * 1e6:>7.1f}   "
                  f"{nbytes / tb / 1e12:.2f} TB/s", flush=True)
            tot[0] += t0
            tot[1] += t0t
            tot[2] += tb
            del ws
        print(f"M={M:>4}    layer: {tot[0] * 1e6:>7.1f} {tot[1] * 1e6:>7.1f} {tot[2] * 1e6:>7.1f}", flush=True)


def chain(model: str) -> None:
    """One decoder layer's four GEMMs back to back with real data dependencies, over enough layers (different
    weights each) to exceed the Infinity Cache, inside a captured graph as the engine runs them: microseconds per layer."""
    (_, nq, k), (_, no, _), (_, ng, _), (_, nd, kd) = SHAPES[model]
    layers = max(2, -(-(700 << 20) // ((nq + no + ng) * k * 2 + nd * kd * 2)))
    W = [[rnd(nq, k, scale=k ** -0.5), rnd(no, k, scale=k ** -0.5), rnd(ng, k, scale=k ** -0.5),
          rnd(nd, kd, scale=kd ** -0.5)] for _ in range(layers)]
    print(f"\n{model} layer chain ({layers} layers of weights): us per layer  library (selection table) / in-tree / "
          f"in-tree with the library's gate_up / in-tree without K splits")

    def run(x, mode):
        for wq, wo, wg, wd in W:
            if mode == "lib":
                q = torch.nn.functional.linear(x, wq)
                o = torch.nn.functional.linear(q[:, :k], wo)
                h = ops.silu_and_mul(torch.nn.functional.linear(o, wg))
                x = torch.nn.functional.linear(h, wd)
            else:
                sp = 1 if mode == "nosplit" else 0
                q = ops.dense_gemm(x, wq, splits=sp)
                o = ops.dense_gemm(q[:, :k], wo, splits=sp)
                if mode == "mixed":
                    h = ops.silu_and_mul(torch.nn.functional.linear(o, wg))
                else:
                    h = ops.dense_gemm(o, wg, epilogue="gate_up", splits=sp)
                x = ops.dense_gemm(h, wd, splits=sp)
        return x
    _tuned_library(True)
    for M in (8, 64, 128, 192, 256):
        res = []
        for mode in ("lib", "tree", "mixed", "nosplit"):
            x = rnd(M, k)
            run(x, mode)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                y = run(x, mode)
            for _ in range(2):
                g.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                g.replay()
            b.record()
            torch.cuda.synchronize()
            res.append(a.elapsed_time(b) / 5 / layers * 1e3)
            del g, y
        print(f"M={M:>4}: " + "  ".join(f"{t:7.1f}" for t in res), flush=True)
    _tuned_library(False)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    which = sys.argv[2] if len(sys.argv) > 2 else "70b"
    if mode == "check":
        sys.exit(1 if check() else 0)
    elif mode == "prefill":
        prefill(which)
    elif mode == "chain":
        chain(which)
    else:
        decode(which)
