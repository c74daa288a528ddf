"""A few launches of the tiled MFMA expert GEMMs (Mixtral-8x7B layer, 8192 tokens, fp8 and bf16 operands)
for a rocprofv3 counter pass:

    rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_WAVE_CYCLES \
        SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace -d out -- python3 benchmarks_dev/tile_gemm_pmc.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd import ops  # noqa: E402
from vllm_distributed_amd.ops import reference as R  # noqa: E402

E, k, H, I, T = 8, 2, 4096, 14336, int(os.environ.get("TOKENS", "8192"))
os.environ["VDA_MOE_GROUPED_MAX_PAIRS"] = "0"
w13 = torch.randn(E, 2 * I, H, device="cuda", dtype=torch.bfloat16) * 0.03
w2 = torch.randn(E, H, I, device="cuda", dtype=torch.bfloat16) * 0.03
s13 = w13.float().abs().amax(dim=(1, 2)) / 448
s2 = w2.float().abs().amax(dim=(1, 2)) / 448
q13 = (w13.float() / s13.view(E, 1, 1)).to(torch.float8_e4m3fn)
q2 = (w2.float() / s2.view(E, 1, 1)).to(torch.float8_e4m3fn)
x = torch.randn(T, H, device="cuda", dtype=torch.bfloat16)
tw, ids = R.moe_topk_softmax(torch.randn(T, E, device="cuda"), k, True)
for _ in range(int(os.environ.get("REPS", "2"))):
    ops.fused_moe(x, q13, q2, tw, ids, s13, s2)
    ops.fused_moe(x, w13, w2, tw, ids)
torch.cuda.synchronize()
print("done")
