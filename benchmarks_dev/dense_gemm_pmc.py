"""A few launches of the ping-pong MFMA GEMM (both kernel forms) and of the library GEMM on one prefill shape,
for a rocprofv3 counter pass:

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
        SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv \
        -d out -- python3 benchmarks_dev/dense_gemm_pmc.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd import ops  # noqa: E402

M, N_, K = (int(os.environ.get(k, d)) for k, d in (("M", "8192"), ("N", "8192"), ("K", "8192")))
x = ((torch.rand(M, K, device="cuda") * 2 - 1)).to(torch.bfloat16)
w = ((torch.rand(N_, K, device="cuda") * 2 - 1) * K ** -0.5).to(torch.bfloat16)
for _ in range(int(os.environ.get("REPS", "3"))):
    ops.dense_gemm(x, w, variant=1)
    ops.dense_gemm(x, w, variant=2)
    if os.environ.get("LIB", "1") == "1":
        torch.nn.functional.linear(x, w)
torch.cuda.synchronize()
print("done")
