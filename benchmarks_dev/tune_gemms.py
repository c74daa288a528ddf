"""Offline GEMM algorithm selection for decode-sized batches (PyTorch TunableOp over hipBLASLt + rocBLAS).
Writes the selections to vllm_distributed_amd/ops/tuned/gfx950_gemm.csv, which workers load at start-up
with tuning disabled (pure lookup).  usage: tune_gemms.py <out.csv> <model preset> [tp list] [M list]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vllm_distributed_amd.config import load_hf_config

out, model = sys.argv[1], sys.argv[2]
tps = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,8").split(",")]
Ms = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "64,128,256").split(",")]
hf = load_hf_config(model)
H, I = hf.hidden_size, hf.intermediate_size
nh, nkv = hf.num_attention_heads, hf.num_key_value_heads
hd = getattr(hf, "head_dim", None) or H // nh
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_max_tuning_duration(60)
torch.cuda.tunable.set_max_tuning_iterations(60)
torch.cuda.tunable.set_rotating_buffer_size(512)          # weights must come from HBM, not the infinity cache
torch.cuda.tunable.set_filename(out)
if os.path.exists(out):
    torch.cuda.tunable.read_file(out)
seen = set()
t0 = time.time()
for tp in tps:
    shapes = [(H, (nh // tp + 2 * max(nkv // tp, 1)) * hd), (nh * hd // tp, H), (H, 2 * I // tp), (I // tp, H)]
    for (K, N) in shapes:
        for M in Ms:
            if (M, K, N) in seen:
                continue
            seen.add((M, K, N))
            x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
            w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
            torch.nn.functional.linear(x, w)
            torch.cuda.synchronize()
            print(f"[{time.time() - t0:6.1f}s] tp={tp} M={M} K={K} N={N}", flush=True)
            del x, w
print("done", len(seen), "shapes")
