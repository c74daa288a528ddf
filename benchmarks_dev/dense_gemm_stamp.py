"""Where a loop step of the ping-pong GEMM spends its cycles: runs the stamped diagnostic kernels (both forms) on
one prefill shape and prints, per wave group, the mean shader cycles per step in each part of the step, and per
round of workgroups (a CU holds one) the start / loop / end times of the workgroups on the 100 MHz clock.

    python benchmarks_dev/dense_gemm_stamp.py [M N K]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_distributed_amd.ops import native as N  # noqa: E402

M, N_, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8192, 8192, 8192)
x = ((torch.rand(M, K, device="cuda") * 2 - 1)).to(torch.bfloat16)
w = ((torch.rand(N_, K, device="cuda") * 2 - 1) * K ** -0.5).to(torch.bfloat16)
names = ["issue+reads", "wait copies", "barrier 1", "MFMAs", "barrier 2"]
pm, pn = -(-(-(-M // 256)) // 8), -(-(-(-N_ // 256)) // 8)
nblocks = -(-(pm * pn) // 8) * 8 * 64
for variant in (1, 2):
    dbg = torch.zeros(64 * 8 * 6 + nblocks * 4, dtype=torch.int64, device="cuda")
    for _ in range(3):
        N.call("vda_dense_gemm_stamp", dbg.data_ptr(), x.data_ptr(), w.data_ptr(), M, N_, K, variant, N.stream())
    torch.cuda.synchronize()
    d = dbg[:64 * 8 * 6].view(64, 8, 6).double()
    steps = d[..., 5:6]
    per = d[..., :5] / steps
    print(f"variant {variant}: cycles per step of 32 MFMAs per wave (mean over 64 workgroups; ideal MFMA part = 512)")
    for grp in (0, 1):
        v = per[:, 4 * grp:4 * grp + 4].mean(dim=(0, 1))
        print(f"  group {grp}: " + "  ".join(f"{n} {c:7.1f}" for n, c in zip(names, v.tolist())) +
              f"   total {v.sum().item():7.1f}")
    b = dbg[64 * 8 * 6:].view(nblocks, 4).cpu()
    b = b[b[:, 0] > 0]
    t0 = b[:, 0].min().item()
    start, loop0, end = ((b[:, i] - t0).double() / 100 for i in range(3))      # microseconds
    order = torch.argsort(start)
    n = len(order)
    print(f"  {n} workgroups; kernel span {end.max().item():.1f} us; loop time per workgroup: mean "
          f"{(end - loop0).mean().item():.1f} us, min {(end - loop0).min().item():.1f}, max {(end - loop0).max().item():.1f}; "
          f"prologue (start -> loop) mean {(loop0 - start).mean().item():.2f} us")
    for q in range(0, n, 256):
        idx = order[q:q + 256]
        print(f"    workgroups {q:5d}..{q + len(idx) - 1:5d} by start: start {start[idx].min().item():8.1f}..{start[idx].max().item():8.1f} us  "
              f"loop {(end[idx] - loop0[idx]).mean().item():7.1f} us  end {end[idx].min().item():8.1f}..{end[idx].max().item():8.1f}")
    xc = torch.bincount(b[:, 3].int(), minlength=8).tolist()
    print(f"    workgroups per XCC id: {xc}")
