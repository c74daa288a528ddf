"""A few launches of the 4-bit code-book GEMM kernels (ops/csrc/cb4.hip) on the Llama-3-70B gate/up and down shapes at
one and 16 rows, NF4 in the chunk layout and NVFP4 in the per-weight form, for a rocprofv3 counter pass:

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS \
        SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d out -- python3 benchmarks_dev/cb4_pmc.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd import ops  # noqa: E402
from vllm_distributed_amd.layers.quant import NF4_CODE  # noqa: E402
from vllm_distributed_amd.models.dequant import _E2M1  # noqa: E402

for N_, K in ((57344, 8192), (8192, 28672)):
    codes = torch.randint(0, 256, (N_, K // 2), dtype=torch.uint8, device="cuda")
    for G, hi, table, chunked in ((64, True, NF4_CODE, True), (16, False, _E2M1, False)):
        tab = torch.tensor(table, dtype=torch.float32, device="cuda")
        scales = torch.rand((K // G, N_) if chunked else (N_, K // G), device="cuda") + 0.5
        for M in (1, 16):
            x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
            for _ in range(2):
                ops.cb4_linear(x, codes, scales, tab, G, hi, chunked=chunked)
torch.cuda.synchronize()
print("done")
