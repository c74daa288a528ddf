"""Rank program: one rank withholds a collective call; the peers' kernels must time out, leave (no hang)
and report through the error word instead of handing out a result silently."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

from vllm_distributed_amd.parallel.state import GroupCoordinator


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    g = GroupCoordinator(list(range(world)), rank, 0, "gloo+hip", "tp_test", use_device_comm=True)
    g._attach(dist.group.WORLD, dist.group.WORLD)
    assert g.ca_comm is not None
    x = torch.ones(4096, device=dev, dtype=torch.bfloat16)
    out = g.all_reduce(x)
    torch.cuda.synchronize()
    assert torch.all(out == world)
    g.check_comm_errors()                      # healthy: no complaint
    dist.barrier()
    if rank != world - 1:
        t0 = time.time()
        g.all_reduce(x)                        # the last rank never makes this call
        torch.cuda.synchronize()
        assert time.time() - t0 < 60, "the kernel must leave on its own"
        try:
            g.check_comm_errors()
        except RuntimeError as e:
            assert "timed out" in str(e), e
            print(f"rank {rank} saw: {e}", flush=True)
        else:
            raise AssertionError("a withheld call went unnoticed")
    dist.barrier()
    print(f"rank {rank} ok", flush=True)
    sys.stdout.flush()
    os._exit(0)        # the contexts are out of step on purpose: skip their teardown


if __name__ == "__main__":
    main()
