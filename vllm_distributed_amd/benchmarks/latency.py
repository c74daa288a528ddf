"""Fixed-batch end-to-end latency (reference: benchmarks/benchmark_latency.py, vllm/benchmarks/latency.py)."""
from __future__ import annotations

import argparse
import json
import time

import numpy as np


def add_cli_args(p: argparse.ArgumentParser) -> None:
    p.add_argument("--model", type=str, required=True)
    from ..engine.arg_utils import EngineArgs
    p.add_argument("--input-len", type=int, default=32)
    p.add_argument("--output-len", type=int, default=128)
    p.add_argument("--batch-size", type=int, default=8)
    p.add_argument("--n", type=int, default=1)
    p.add_argument("--num-iters-warmup", type=int, default=3)
    p.add_argument("--num-iters", type=int, default=10)
    p.add_argument("--profile", action="store_true")
    p.add_argument("--output-json", default=None)
    EngineArgs.add_cli_args(p)


def main(args: argparse.Namespace) -> dict:
    from ..engine.arg_utils import EngineArgs
    from ..entrypoints.llm import LLM
    from ..sampling_params import SamplingParams
    llm = LLM.from_engine_args(EngineArgs.from_cli_args(args))
    sp = SamplingParams(n=args.n, temperature=1.0, top_p=1.0, ignore_eos=True, max_tokens=args.output_len,
                        detokenize=False)
    vocab = llm.llm_engine.model_config.get_vocab_size()
    ids = np.random.default_rng(0).integers(0, vocab, size=(args.batch_size, args.input_len))
    prompts = [{"prompt_token_ids": r.tolist()} for r in ids]

    def run() -> float:
        t = time.perf_counter()
        llm.generate(prompts, sp, use_tqdm=False)
        return time.perf_counter() - t

    for _ in range(args.num_iters_warmup):
        run()
    if args.profile:
        llm.start_profile()
        run()
        llm.stop_profile()
        return {}
    lat = np.array([run() for _ in range(args.num_iters)])
    pct = [10, 25, 50, 75, 90, 99]
    res = {"avg_latency": float(lat.mean()), "latencies": lat.tolist(),
           "percentiles": {str(p): float(v) for p, v in zip(pct, np.percentile(lat, pct))}}
    print(f"Avg latency: {res['avg_latency']:.4f} s")
    for p, v in res["percentiles"].items():
        print(f"{p}% percentile latency: {v:.4f} s")
    if args.output_json:
        with open(args.output_json, "w") as f:
            json.dump(res, f, indent=2)
    return res
