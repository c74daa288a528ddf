"""One-token lightning attention step (MiniMax-Text-01 layer: 64 heads of 128) at decode batch sizes: the fused HIP
kernel against the same step in PyTorch ops.  The state (fp32 [S, 64, 128, 128] = 4 MB per sequence) is read and
written once per step, so the floor is 2 x |state| at the HBM rate.

    python benchmarks_dev/lightning_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vllm_distributed_amd import ops


def torch_step(qkv, state, slope, seq_lens, table, H, D):
    S = qkv.shape[0]
    q, k, v = qkv.view(S, H, 3 * D).float().split(D, dim=-1)
    slots = table[:S, 0].long()
    fresh = (seq_lens[:S] <= 1).view(S, 1, 1, 1)
    s = state[slots] * torch.where(fresh, 0.0, 1.0) * torch.exp(-slope).view(1, H, 1, 1)
    s = s + k.unsqueeze(-1) * v.unsqueeze(-2)
    state[slots] = s
    return torch.matmul(q.unsqueeze(-2), s).squeeze(-2)


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    H, D = 64, 128
    dev = "cuda"
    print(f"{'seqs':>6} {'state GB':>9} {'kernel us':>10} {'TB/s':>6} {'torch us':>10} {'speed-up':>8}")
    for S in (16, 64, 256, 512):
        qkv = (torch.randn(S, H * 3 * D, device=dev) * 0.3).to(torch.bfloat16)
        state = torch.randn(S + 1, H, D, D, device=dev) * 0.01
        slope = torch.rand(H, device=dev) * 0.3 + 0.01
        seq_lens = torch.full((S,), 100, dtype=torch.int32, device=dev)
        table = (torch.arange(S, dtype=torch.int32, device=dev) + 1).view(S, 1)
        tk = bench(lambda: ops.lightning_attention_decode(qkv, state, slope, seq_lens, table, H, D))
        tt = bench(lambda: torch_step(qkv, state, slope, seq_lens, table, H, D), n=5)
        gb = S * H * D * D * 4 / 1e9
        print(f"{S:>6} {gb:>9.2f} {tk * 1e6:>10.1f} {2 * gb / tk / 1e3:>6.2f} {tt * 1e6:>10.1f} {tt / tk:>8.1f}")


if __name__ == "__main__":
    main()
