"""Microscaling fp4 weights in their two in-memory forms: code-book kernel (ops/csrc/cb4.hip) against the exact 8-bit\nre-coding through the W8A16 streaming kernel (ops/csrc/gemm_wq.hip).  Usage: PYTHONPATH=. python benchmarks_dev/mxfp4_probe.py"""
import torch
from vllm_distributed_amd import ops
from vllm_distributed_amd.models.dequant import _E2M1
from benchmarks_dev.cb4_probe import timed
shapes = {"qkv": (10240, 8192), "o": (8192, 8192), "gate_up": (57344, 8192), "down": (8192, 28672)}
tab = torch.tensor(_E2M1, dtype=torch.float32, device="cuda")
G = 32
for name, (N, K) in shapes.items():
    copies = max(2, int(600e6 // (N * K // 2)))
    codes = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device="cuda") for _ in range(copies)]
    st = (torch.rand(K // G, N, device="cuda") + 0.5)
    c8 = max(2, copies // 2)
    q8 = [torch.randint(0, 256, (N, K), dtype=torch.uint8, device="cuda") for _ in range(c8)]
    sz = torch.stack([torch.rand(K // G, N, device="cuda") + 0.5, torch.full((K // G, N), 128.0, device="cuda")], -1).to(torch.bfloat16).contiguous()
    for M in (1, 16, 32, 64):
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        i = [0]
        def run():
            i[0] += 1
            return ops.cb4_linear(x, codes[i[0] % copies], st, tab, G, False, chunked=True)
        def run8():
            i[0] += 1
            return ops.wq_gemm(x, q8[i[0] % c8], sz, G, 8)
        print(f"mxfp4 g32 {name:8s} M={M:3d}  code-book {timed(run):8.1f} us   as 8-bit integers {timed(run8):8.1f} us", flush=True)
    del codes, q8
