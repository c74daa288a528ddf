"""Rank program: quantised peer-memory all-reduce between processes that share GPU 0."""
import faulthandler
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
faulthandler.dump_traceback_later(150, exit=True)
import torch
import torch.distributed as dist

from vllm_distributed_amd.parallel.custom_allreduce import QuickAllReduce


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    # FP sends bf16 values: its "step" is bf16's relative rounding (2^-8 of the value), bounded here like a 256-level code
    for mode, qmax in (("INT8", 127.0), ("INT6", 31.0), ("INT4", 7.0), ("FP", 256.0)):
        os.environ["VLLM_ROCM_QUICK_REDUCE_QUANTIZATION"] = mode
        qr = QuickAllReduce.maybe_create(dist.group.WORLD, dev)
        assert qr is not None, f"{mode}: self-test failed"
        torch.manual_seed(100 + rank)
        for dtype in (torch.bfloat16, torch.float16):
            for n in (32, 64 * 1024, 1_000_000 // 32 * 32, 40_000_000 // 32 * 32):
                x = torch.randn(n, device=dev).to(dtype)
                ref = x.float().cpu()
                dist.all_reduce(ref)
                for _ in range(2):
                    out = qr._launch(x)
                    torch.cuda.synchronize()
                    assert qr.check_error() == 0
                    err = (out.float().cpu() - ref).abs()
                    # per 32-value group: input rounding (world terms) + result rounding, each half a step
                    bound = (x.float().abs().max().item() * world * 0.5 + ref.abs().max().item() * 0.5) / qmax * 1.1 \
                        + ref.abs().max().item() * 2 ** -7
                    assert float(err.max()) <= bound, (mode, dtype, n, float(err.max()), bound)
                    rel = float(err.norm() / ref.norm())
                    assert rel < {"FP": 0.01, "INT8": 0.02, "INT6": 0.06, "INT4": 0.25}[mode], (mode, dtype, n, rel)
        assert not qr.should_use(torch.zeros(1024, device=dev, dtype=torch.bfloat16))     # below the threshold
        assert qr.should_use(torch.zeros(4 << 20, device=dev, dtype=torch.bfloat16))
        # graph capture + replay
        x = torch.zeros(256 * 8192, device=dev, dtype=torch.bfloat16)
        qr._launch(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                y = qr._launch(x)
        for it in range(4):
            x.fill_(float(it + rank + 1))
            dist.barrier()
            g.replay()
            torch.cuda.synchronize()
            exp = float(sum(it + r + 1 for r in range(world)))
            assert torch.allclose(y.float(), torch.full_like(y, exp).float(), rtol={"INT4": 0.3, "INT6": 0.07}.get(mode, 0.02)), \
                (mode, it, rank, y[:4], exp)
        for n in (256 * 8192, 4096 * 8192):
            x = torch.randn(n, device=dev).to(torch.bfloat16)
            for _ in range(3):
                qr._launch(x)
            torch.cuda.synchronize(); dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                qr._launch(x)
            e1.record(); torch.cuda.synchronize()
            if rank == 0:
                print(f"{mode} allreduce {n * 2 / 1024:.0f} KiB: {e0.elapsed_time(e1) / 10 * 1000:.1f} us "
                      f"(same-GPU peers)", flush=True)
        dist.barrier()
        qr.close()
    dist.destroy_process_group()
    print(f"rank {rank} ok", flush=True)


if __name__ == "__main__":
    main()
