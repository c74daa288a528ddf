"""Rank program: fused all-reduce + residual add + RMSNorm between processes that share GPU 0, against the
unfused pair (peer-memory all-reduce, then the fused add + norm kernel): the residual bit-exact (two ranks;
within a rounding step for more, where the summation orders differ), the
normalised rows within one rounding step (the two kernels add the squares in different orders), eagerly
and under hipGraph replay, for row counts below, at and above the number of workgroups."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

from vllm_distributed_amd import ops
from vllm_distributed_amd.parallel.custom_allreduce import CustomAllReduce, FusedAllReduceNorm


ULP = {torch.bfloat16: dict(rtol=1.6e-2, atol=1e-3), torch.float16: dict(rtol=2.5e-3, atol=1e-4)}     # two roundings


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ca = CustomAllReduce.maybe_create(dist.group.WORLD, dev)
    far = FusedAllReduceNorm.maybe_create(dist.group.WORLD, dev)
    assert ca is not None and far is not None, "self-test failed"
    for dtype in (torch.bfloat16, torch.float16):
        for M, H, wb in ((1, 4096, 0.0), (7, 8192, 0.0), (64, 8192, 0.0), (65, 5120, 1.0), (300, 8192, 0.0),
                         (512, 16384, 0.0), (33, 1024, 0.0)):
            torch.manual_seed(1000 * rank + M)
            x = torch.randn(M, H, device=dev).to(dtype)
            torch.manual_seed(M)                       # the same residual and weight on every rank
            res0 = torch.randn(M, H, device=dev).to(dtype)
            w = (torch.randn(H, device=dev) * 0.1 + 1).to(dtype)
            ref_res = res0.clone()
            ref_out = ops.rms_norm(ca.all_reduce(x), w, 1e-5, ref_res, wb)
            for _ in range(3):
                res = res0.clone()
                out = far.run(x, res, w, 1e-5, wb)
                assert out is not None, (M, H)
                torch.cuda.synchronize()
                if world == 2:       # two addends: every summation order gives the same bits
                    assert torch.equal(res, ref_res), (dtype, M, H, rank, (res.float() - ref_res.float()).abs().max())
                else:                # the two-shot form of the plain kernel adds the ranks in a rotated order
                    torch.testing.assert_close(res.float(), ref_res.float(), **ULP[dtype])
                torch.testing.assert_close(out.float(), ref_out.float(), **ULP[dtype])
    assert far.run(torch.zeros(600, 8192, device=dev, dtype=torch.bfloat16), torch.zeros(600, 8192, device=dev,
                   dtype=torch.bfloat16), torch.ones(8192, device=dev, dtype=torch.bfloat16), 1e-5) is None
    # graph capture + replay with changing inputs, two chained calls inside one graph
    M, H = 48, 8192
    x = torch.zeros(M, H, device=dev, dtype=torch.bfloat16)
    res = torch.zeros(M, H, device=dev, dtype=torch.bfloat16)
    w = torch.ones(H, device=dev, dtype=torch.bfloat16)
    far.run(x, res, w, 1e-5)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            y = far.run(x, res, w, 1e-5)
            z = far.run(y, res, w, 1e-5)
    for it in range(4):
        torch.manual_seed(it * 10 + rank)
        x.copy_(torch.randn(M, H, device=dev))
        res.zero_()
        dist.barrier()
        g.replay()
        torch.cuda.synchronize()
        r2 = torch.zeros_like(res)
        y2 = ops.rms_norm(ca.all_reduce(x), w, 1e-5, r2)
        z2 = ops.rms_norm(ca.all_reduce(y2), w, 1e-5, r2)
        torch.cuda.synchronize()
        torch.testing.assert_close(y.float(), y2.float(), **ULP[torch.bfloat16])
        torch.testing.assert_close(z.float(), z2.float(), rtol=3e-2, atol=3e-2)     # a rounding step of y moves z's sums
        torch.testing.assert_close(res.float(), r2.float(), rtol=3e-2, atol=3e-2)
    assert far.check_error() == 0
    for M in (16, 64, 256):
        x = torch.randn(M, 8192, device=dev).to(torch.bfloat16)
        res = torch.zeros_like(x)
        w = torch.ones(8192, device=dev, dtype=torch.bfloat16)
        for fused in (False, True):
            fn = (lambda: far.run(x, res, w, 1e-5)) if fused else (lambda: ops.rms_norm(ca.all_reduce(x), w, 1e-5, res))
            for _ in range(3):
                fn()
            torch.cuda.synchronize(); dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            if rank == 0:
                print(f"M={M} H=8192 {'fused' if fused else 'all-reduce + norm'}: {e0.elapsed_time(e1) / 20 * 1000:.1f} us "
                      f"(same-GPU peers)", flush=True)
    dist.barrier()
    far.close(); ca.close()
    dist.destroy_process_group()
    print(f"rank {rank} ok", flush=True)


if __name__ == "__main__":
    main()
