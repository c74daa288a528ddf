"""Rank program: peer-memory all-reduce between processes that share GPU 0
(the 1-GPU box cannot host RCCL ranks on one device, but IPC peer mapping works)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

from vllm_distributed_amd.parallel.custom_allreduce import CustomAllReduce


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ca = CustomAllReduce.maybe_create(dist.group.WORLD, dev)
    assert ca is not None, "self-test failed"
    torch.manual_seed(rank)
    for dtype in (torch.bfloat16, torch.float16, torch.float32):
        for n in (8, 4096, 70_000, 1_000_000, 5_000_000, 20_000_000):
            x = (torch.randint(-8, 8, (n,), device=dev)).to(dtype)     # integers: the sum is exact
            ref = x.float().cpu()
            dist.all_reduce(ref)
            for _ in range(2):
                out = ca.all_reduce(x)
                assert out is not None
                torch.cuda.synchronize()
                assert torch.equal(out.float().cpu(), ref), (dtype, n, rank)
    # graph capture + replay with changing inputs
    x = torch.zeros(64 * 8192, device=dev, dtype=torch.bfloat16)
    ca.all_reduce(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            y = ca.all_reduce(x)
            z = ca.all_reduce(y * 0 + x)     # two chained calls inside one graph
    for it in range(5):
        x.fill_(float(it + rank))
        dist.barrier()
        g.replay()
        torch.cuda.synchronize()
        exp = float(sum(it + r for r in range(world)))
        assert torch.all(y == exp) and torch.all(z == exp), (it, rank, y[:4], z[:4])
    assert ca.check_error() == 0
    # timing (informative)
    for n in (16 * 8192, 256 * 8192, 4096 * 8192):
        x = torch.ones(n, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            ca.all_reduce(x)
        torch.cuda.synchronize(); dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ca.all_reduce(x)
        e1.record(); torch.cuda.synchronize()
        if rank == 0:
            print(f"allreduce {n * 2 / 1024:.0f} KiB: {e0.elapsed_time(e1) / 20 * 1000:.1f} us (same-GPU peers)", flush=True)
    dist.barrier()
    ca.close()
    dist.destroy_process_group()
    print(f"rank {rank} ok", flush=True)


if __name__ == "__main__":
    main()
