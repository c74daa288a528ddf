"""Sparse (vertical-slash) against dense paged prefill on one long request: 28 query / 4 KV heads of 128, one chunk.
Prints the time of each form and the fraction of 32-key steps the sparse form visits."""
import math
import sys
import time

import torch

from vllm_distributed_amd import ops

dev = "cuda"
S = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
NV, NS = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1000, 6096)
Hq, Hkv, D, bs, dt = 28, 4, 128, 16, torch.bfloat16
nb = S // bs
kc = torch.randn(nb + 1, Hkv, bs, D, device=dev, dtype=dt)
vc = torch.randn(nb + 1, Hkv, D, bs, device=dev, dtype=dt)
bt = (torch.randperm(nb, device=dev) + 1).int()[None].contiguous()
q = torch.randn(S, Hq, D, device=dev, dtype=dt)
sl = torch.tensor([S], dtype=torch.int32, device=dev)
qsl = torch.tensor([0, S], dtype=torch.int32, device=dev)
g = torch.Generator(device="cpu").manual_seed(0)
MODE = sys.argv[4] if len(sys.argv) > 4 else "random"
if MODE == "random":        # worst case: every far line lands in steps of its own
    vert = torch.randint(0, S, (1, Hq, NV), generator=g)
    near = torch.randint(0, 4096, (1, Hq, NS // 2), generator=g)
    far = torch.randint(0, S, (1, Hq, NS - NS // 2), generator=g)
    slash = torch.cat([near, far], -1)
else:                       # what trained heads show: the first keys, the nearest distances, a few runs of neighbours
    def runs(total, head_run, run):
        n = (total - head_run) // run
        starts = torch.randint(head_run, S - run, (1, Hq, n), generator=g)
        far = (starts[..., None] + torch.arange(run)).flatten(2)
        first = torch.arange(total - far.shape[-1]).expand(1, Hq, -1)
        return torch.cat([first, far], -1)
    vert, slash = runs(NV, NV // 4, 25), runs(NS, NS // 3, 127)
vert, slash = vert.int().contiguous().to(dev), slash.int().contiguous().to(dev)
vcount = torch.full((1, Hq), NV, dtype=torch.int32, device=dev)
scount = torch.full((1, Hq), NS, dtype=torch.int32, device=dev)
zero = torch.zeros(1, dtype=torch.int32, device=dev)
out = torch.empty_like(q)
scale = 1 / math.sqrt(D)


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


td = timed(lambda: ops.paged_attention_prefill(out, q, kc, vc, bt, sl, qsl, scale))
ts = timed(lambda: ops.paged_attention_prefill_sparse(out, q, kc, vc, bt, sl, qsl, scale, vert, vcount, slash, scount,
                                                      zero, zero))
flops = 4.0 * S * S / 2 * Hq * D
print(f"S={S} lines v={NV} s={NS} ({MODE}): dense {td:.1f} ms ({flops / td / 1e9:.0f} TF/s), sparse {ts:.1f} ms "
      f"-> {td / ts:.2f}x", flush=True)
