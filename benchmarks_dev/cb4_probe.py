"""4-bit code-book GEMM (ops/csrc/cb4.hip) on the Llama-3-70B projections: the skinny kernel at 1 .. 64 rows and the
expand + library path above, against the bf16 library GEMM of the same layer.  Usage: python benchmarks_dev/cb4_probe.py"""
import torch

from vllm_distributed_amd import ops
from vllm_distributed_amd.layers.quant import NF4_CODE
from vllm_distributed_amd.models.dequant import _E2M1


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    import sys
    shapes = {"qkv": (10240, 8192), "o": (8192, 8192), "gate_up": (57344, 8192), "down": (8192, 28672)}
    if len(sys.argv) > 1 and sys.argv[1] == "8b":
        shapes = {"qkv": (6144, 4096), "o": (4096, 4096), "gate_up": (28672, 4096), "down": (4096, 14336)}
    for fmt, G, hi, table in (("nf4 g64", 64, True, NF4_CODE), ("nvfp4 g16", 16, False, _E2M1)):
        tab = torch.tensor(table, dtype=torch.float32, device="cuda")
        for name, (N, K) in shapes.items():
            # rotate over copies so that no call finds its weights in the 256 MiB last-level cache
            copies = max(2, int(600e6 // (N * K // 2)))
            codes = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device="cuda") for _ in range(copies)]
            scales = torch.rand(N, K // G, device="cuda") + 0.5
            chunked = ops.cb4_chunk_layout_ok(K, G)
            if chunked:
                scales = scales.t().contiguous()        # random codes: only the scales' layout matters for timing
            w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
            for M in (1, 8, 16, 32, 64, 256):
                x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
                i = [0]

                def run():
                    i[0] += 1
                    return ops.cb4_linear(x, codes[i[0] % copies], scales, tab, G, hi, chunked=chunked)
                t = timed(run)
                t_lib = timed(lambda: torch.nn.functional.linear(x, w))
                bytes_ = N * K / 2 + N * (K // G) * 4
                extra = ""
                if chunked and M <= 64:
                    def run1():
                        i[0] += 1
                        return ops.cb4_linear(x, codes[i[0] % copies], scales, tab, G, hi, chunked=True, splits=1)
                    extra = f"   without K splits {timed(run1):8.1f} us"
                print(f"{fmt:10s} {name:8s} M={M:4d}  code-book {t:8.1f} us ({bytes_ / t / 1e6:5.2f} TB/s of packed bytes)"
                      f"   bf16 library {t_lib:8.1f} us{extra}", flush=True)
            del codes, w


if __name__ == "__main__":
    main()
