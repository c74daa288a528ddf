"""Packed-weight GEMM (ops.wq_gemm) against the two alternatives it replaces: expanding the weights on every call
and running the library GEMM, and the library GEMM on weights expanded once (4x / 2x the memory).
python benchmarks_dev/wq_gemm_probe.py [bits]"""
import sys

import torch

from vllm_distributed_amd import ops


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    bits_list = [int(sys.argv[1])] if len(sys.argv) > 1 else [4, 8]
    G = 128
    print(f"{'bits':>4} {'M':>5} {'N':>6} {'K':>6} {'packed us':>10} {'TF/s':>7} {'expand+lib':>10} {'lib 16-bit':>10} {'GB/s wts':>9}")
    for bits in bits_list:
        for (N, K) in ((10240, 8192), (8192, 8192), (57344, 8192), (8192, 28672), (6144, 4096), (28672, 4096)):
            top = (1 << bits) - 1
            q = torch.randint(0, top + 1, (N, K), device="cuda")
            qw = ops.w4_pack(q) if bits == 4 else q.to(torch.uint8)
            del q
            scales = (torch.rand(N, K // G, device="cuda") * 0.01).bfloat16()
            zeros = torch.full((N, K // G), float(top // 2 + 1), device="cuda").bfloat16()
            sz = torch.stack([scales.t(), zeros.t()], dim=-1).contiguous()
            w16 = ops.wq_dequant(qw, scales, zeros, G, bits, torch.bfloat16)
            for M in (1, 16, 32, 64, 128, 256, 512, 1024, 4096):
                x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
                t = bench(lambda: ops.wq_gemm(x, qw, sz, G, bits))
                if M <= 128:
                    tt = bench(lambda: ops.wq_gemm(x, qw, sz, G, bits, variant=1))
                    t5 = bench(lambda: ops.wq_gemm(x, qw, sz, G, bits, variant=3))
                t2 = bench(lambda: torch.nn.functional.linear(x, ops.wq_dequant(qw, scales, zeros, G, bits, torch.bfloat16)))
                t3 = bench(lambda: torch.nn.functional.linear(x, w16))
                if bits == 4 and M <= 64:
                    t4 = bench(lambda: ops.w4_linear(x, qw, scales, zeros, G))
                    extra = f"  skinny {t4:8.1f}"
                else:
                    extra = ""
                if M <= 128:
                    extra += f"  tiled {tt:8.1f}  per-element {t5:8.1f}"
                print(f"{bits:>4} {M:>5} {N:>6} {K:>6} {t:>10.1f} {2.0 * M * N * K / t / 1e6:>7.0f} {t2:>10.1f} {t3:>10.1f} "
                      f"{N * K * bits / 8 / t / 1e3:>9.0f}{extra}", flush=True)
            del qw, w16, sz


if __name__ == "__main__":
    main()
