"""Decode-sized fused MoE: row-split grouped kernel (activations shared through LDS) vs the K-split form."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vllm_distributed_amd import ops
from vllm_distributed_amd.ops import native as N
from vllm_distributed_amd.ops import reference as R

dev = "cuda"


def bench(fn, iters=int(os.environ.get("REPS", "20"))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def main():
    torch.manual_seed(0)
    # (name, experts on this GPU, hidden, intermediate, top_k, weight dtype)
    shapes = [("mixtral-8x7b tp1", 8, 4096, 14336, 2, torch.bfloat16), ("mixtral-8x7b fp8", 8, 4096, 14336, 2, "fp8"),
              ("qwen3-30b-a3b", 128, 2048, 768, 8, torch.bfloat16), ("mixtral ep8 (1 expert)", 1, 4096, 14336, 1, torch.bfloat16)]
    if os.environ.get("SHAPE") is not None:          # counter passes: one shape, one batch size, few repetitions
        shapes = [shapes[int(os.environ["SHAPE"])]]
    batch_sizes = [int(x) for x in os.environ.get("TOKENS", "8,32,128,256,512").split(",")]
    for name, E, H, I, k, wdt in shapes:
        w13 = torch.randn(E, 2 * I, H, device=dev, dtype=torch.bfloat16) * 0.02
        w2 = torch.randn(E, H, I, device=dev, dtype=torch.bfloat16) * 0.02
        s13 = s2 = None
        if wdt == "fp8":
            s13 = torch.full((E,), 0.01, device=dev)
            s2 = torch.full((E,), 0.01, device=dev)
            w13 = (w13.float() / 0.01).to(torch.float8_e4m3fn)
            w2 = (w2.float() / 0.01).to(torch.float8_e4m3fn)
        wbytes = (w13.numel() * w13.element_size() + w2.numel() * w2.element_size())
        for T in batch_sizes:
            x = torch.randn(T, H, device=dev, dtype=torch.bfloat16)
            logits = torch.randn(T, E, device=dev)
            tw, ids = torch.topk(torch.softmax(logits, -1), k)
            ids = ids.int()
            res = {}
            for form in (0, 1, 2):
                os.environ["VDA_MOE_ROWS_FORM"] = str(form)
                out = ops.fused_moe(x, w13, w2, tw, ids, s13, s2)
                res[form] = (bench(lambda: ops.fused_moe(x, w13, w2, tw, ids, s13, s2)), out)
            os.environ["VDA_MOE_ROWS_FORM"] = "-1"
            auto = bench(lambda: ops.fused_moe(x, w13, w2, tw, ids, s13, s2))
            diff = (res[0][1].float() - res[1][1].float()).abs().max().item()
            touched = min(E, T * k) / E
            print(f"{name:24s} T={T:4d}: k-split {res[0][0]:8.1f} us | row-split {res[1][0]:8.1f} / wide {res[2][0]:8.1f} us "
                  f"({wbytes * touched / res[1][0] / 1e6:5.2f} TB/s of touched weights) | by shape {auto:8.1f} us | max diff {diff:.4f}", flush=True)


if __name__ == "__main__":
    main()
