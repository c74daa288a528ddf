"""Token-parallel micro-benchmark: batch x sequence-length sweep reporting prefill time, decode-only
time per step, system decode tokens/s and per-user tokens/s.  Launch with torchrun, one rank per GPU:

    python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 -m \
        vllm_distributed_amd.benchmarks.tknp --model llama-3.2-1b --token-parallel-size 4

Every rank runs the same engine on the same synthetic prompts (external-launcher mode).
Reference: examples/offline_inference/TKNP/tknp_inference_benchmarks.py:68-166.
"""
from __future__ import annotations

import argparse
import json
import os
import time

import numpy as np


def main(argv=None) -> list[dict]:
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3.2-1b")
    ap.add_argument("--token-parallel-size", type=int, default=int(os.environ.get("WORLD_SIZE", "1")))
    ap.add_argument("--tensor-parallel-size", type=int, default=1)
    ap.add_argument("--batch-sizes", default="8,32,128")
    ap.add_argument("--seq-lens", default="512,2048")
    ap.add_argument("--decode-tokens", type=int, default=32)
    ap.add_argument("--load-format", default="dummy")
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--gpu-memory-utilization", type=float, default=0.85)
    ap.add_argument("--output-json", default=None)
    ap.add_argument("--enforce-eager", action="store_true",
                    help="no captured decode graphs (what the reference's benchmark runs: its token-parallel "
                         "path cannot be captured)")
    a = ap.parse_args(argv)
    from .. import LLM, SamplingParams
    batches = [int(x) for x in a.batch_sizes.split(",")]
    lens = [int(x) for x in a.seq_lens.split(",")]
    tknp = a.token_parallel_size
    llm = LLM(model=a.model, load_format=a.load_format, dtype=a.dtype, device=a.device, skip_tokenizer_init=True,
              enforce_eager=a.enforce_eager, enable_prefix_caching=False, max_model_len=max(lens) + a.decode_tokens + 8,
              max_num_seqs=max(batches), max_num_batched_tokens=max(8192, max(lens)),
              gpu_memory_utilization=a.gpu_memory_utilization,
              distributed_executor_backend="external_launcher" if "RANK" in os.environ else None,
              enable_token_parallel=tknp > 1, token_parallel_size=tknp, tensor_parallel_size=a.tensor_parallel_size)
    vocab = llm.llm_engine.model_config.get_vocab_size()
    rank = int(os.environ.get("RANK", "0"))
    rows = []
    for B in batches:
        for L in lens:
            g = np.random.default_rng(B * 131 + L)
            prompts = [{"prompt_token_ids": g.integers(0, vocab, size=L).tolist()} for _ in range(B)]

            def run(n_out: int) -> float:
                sp = SamplingParams(temperature=0, max_tokens=n_out, ignore_eos=True, detokenize=False)
                t = time.perf_counter()
                llm.generate(prompts, sp, use_tqdm=False)
                return time.perf_counter() - t
            run(1)                                   # warm-up
            t_prefill = run(1)
            t_full = run(1 + a.decode_tokens)
            step = (t_full - t_prefill) / a.decode_tokens
            row = {"batch": B, "seq_len": L, "tknp": tknp, "tp": a.tensor_parallel_size,
                   "prefill_ms": round(t_prefill * 1e3, 2), "decode_step_ms": round(step * 1e3, 3),
                   "system_decode_tok_s": round(B / step, 1), "per_user_tok_s": round(1.0 / step, 2)}
            rows.append(row)
            if rank == 0:
                print(json.dumps(row), flush=True)
    if rank == 0 and a.output_json:
        with open(a.output_json, "w") as f:
            json.dump(rows, f, indent=2)
    llm.shutdown()
    return rows


if __name__ == "__main__":
    main()
