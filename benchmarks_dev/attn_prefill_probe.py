"""Prefill attention: LDS-staged kernel vs the direct-to-register form."""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vllm_distributed_amd import ops
from vllm_distributed_amd.ops import native as N

dev = "cuda"


def problem(S, L, Hq, Hkv, D, bs, dt, ctx=0):
    nb = (L + ctx + bs - 1) // bs
    nblk = S * nb + 1
    kc = torch.randn(nblk, Hkv, bs, D, device=dev, dtype=dt)
    vc = torch.randn(nblk, Hkv, D, bs, device=dev, dtype=dt)
    bt = (torch.arange(S * nb, device=dev, dtype=torch.int32) + 1).view(S, nb)
    q = torch.randn(S * L, Hq, D, device=dev, dtype=dt)
    qsl = torch.arange(S + 1, device=dev, dtype=torch.int32) * L
    sl = torch.full((S,), L + ctx, device=dev, dtype=torch.int32)
    return q, kc, vc, bt, sl, qsl


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters


CASES = [(16, 1024, 32, 8, 128, 16, 0), (16, 1024, 64, 8, 128, 16, 0), (4, 4096, 32, 8, 128, 16, 0),
                                    (2, 2048, 32, 8, 128, 16, 6144), (16, 1024, 32, 32, 128, 16, 0), (16, 1024, 12, 12, 64, 16, 0),
                                    (16, 1024, 64, 8, 128, 32, 0)]
ONLY = os.environ.get("CASE")
REPS = int(os.environ.get("REPS", "20"))
for (S, L, Hq, Hkv, D, bs, ctx) in (CASES if ONLY is None else [CASES[int(ONLY)]]):
    q, kc, vc, bt, sl, qsl = problem(S, L, Hq, Hkv, D, bs, torch.bfloat16, ctx)
    out = torch.empty_like(q)
    scale = 1 / math.sqrt(D)
    flops = 4.0 * S * Hq * D * (L * (L + 1) / 2 + L * ctx)
    res = []
    outs = []
    for v in ((0, 1) if os.environ.get("VARIANT") is None else (int(os.environ["VARIANT"]),) * 2):
        N.call("vda_set_prefill_variant", v)
        t = bench(lambda: ops.paged_attention_prefill(out, q, kc, vc, bt, sl, qsl, scale), REPS)
        res.append(t)
        outs.append(out.clone())
    err = (outs[0].float() - outs[1].float()).abs().max().item()
    print(f"S={S} L={L} ctx={ctx} Hq={Hq} Hkv={Hkv} D={D} bs={bs}: direct {res[0] * 1e3:.3f} ms ({flops / res[0] / 1e12:.0f} TF/s) | "
          f"lds {res[1] * 1e3:.3f} ms ({flops / res[1] / 1e12:.0f} TF/s) | max diff {err:.4f}", flush=True)
