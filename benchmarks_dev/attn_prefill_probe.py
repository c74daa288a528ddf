"""Prefill attention: direct-to-register form (0) vs 4-wave LDS-staged form (2) vs wide 8-wave form (3)."""
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
                                    (16, 1024, 64, 8, 128, 32, 0), (1, 8192, 32, 8, 128, 16, 24576), (8, 2048, 64, 8, 128, 16, 0)]
ONLY = os.environ.get("CASE")
REPS = int(os.environ.get("REPS", "20"))
MAXQ = int(os.environ.get("MAXQ", "1"))      # 0: do not tell the kernel the longest query run (linear tile order)
for (S, L, Hq, Hkv, D, bs, ctx) in (CASES if ONLY is None else [CASES[int(ONLY)]]):
    q, kc, vc, bt, sl, qsl = problem(S, L, Hq, Hkv, D, bs, torch.bfloat16, ctx)
    out = torch.empty_like(q)
    scale = 1 / math.sqrt(D)
    flops = 4.0 * S * Hq * D * (L * (L + 1) / 2 + L * ctx)
    res = []
    outs = []
    variants = (0, 2, 3) if os.environ.get("VARIANT") is None else tuple(int(v) for v in os.environ["VARIANT"].split(","))
    for v in variants:
        N.call("vda_set_prefill_variant", v)
        t = bench(lambda: ops.paged_attention_prefill(out, q, kc, vc, bt, sl, qsl, scale, max_q_len=MAXQ * L), REPS)
        res.append(t)
        outs.append(out.clone())
    N.call("vda_set_prefill_variant", 1)
    err = max((outs[0].float() - o.float()).abs().max().item() for o in outs)
    names = {0: "direct", 2: "lds", 3: "wide"}
    print(f"S={S} L={L} ctx={ctx} Hq={Hq} Hkv={Hkv} D={D} bs={bs}: " + " | ".join(
        f"{names[v]} {t * 1e3:.3f} ms ({flops / t / 1e12:.0f} TF/s)" for v, t in zip(variants, res)) +
        f" | max diff {err:.4f}", flush=True)
