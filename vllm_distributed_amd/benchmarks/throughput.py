"""Offline throughput (reference: benchmarks/benchmark_throughput.py, vllm/benchmarks/throughput.py)."""
from __future__ import annotations

import argparse
import asyncio
import json
import time

from .datasets import add_dataset_args, get_samples


def add_cli_args(p: argparse.ArgumentParser) -> None:
    p.add_argument("--model", type=str, required=True)
    from ..engine.arg_utils import AsyncEngineArgs
    add_dataset_args(p, with_seed=False)
    p.add_argument("--n", type=int, default=1)
    p.add_argument("--async-engine", action="store_true")
    p.add_argument("--output-json", default=None)
    AsyncEngineArgs.add_cli_args(p)


def _sampling(r, n):
    from ..sampling_params import SamplingParams
    return SamplingParams(n=n, temperature=1.0, top_p=1.0, ignore_eos=True, max_tokens=r.expected_output_len,
                          detokenize=False)


def run_sync(args, reqs) -> float:
    from ..engine.arg_utils import EngineArgs
    from ..entrypoints.llm import LLM
    llm = LLM.from_engine_args(EngineArgs.from_cli_args(args))
    t = time.perf_counter()
    llm.generate([r.prompt for r in reqs], [_sampling(r, args.n) for r in reqs], use_tqdm=False)
    return time.perf_counter() - t


async def run_async(args, reqs) -> float:
    from ..engine.arg_utils import AsyncEngineArgs
    from ..engine.async_llm import AsyncLLM
    eng = AsyncLLM.from_engine_args(AsyncEngineArgs.from_cli_args(args))

    async def one(i, r):
        async for _ in eng.generate(r.prompt, _sampling(r, args.n), f"tp-{i}"):
            pass
    try:
        t = time.perf_counter()
        await asyncio.gather(*(one(i, r) for i, r in enumerate(reqs)))
        return time.perf_counter() - t
    finally:
        eng.shutdown()


def main(args: argparse.Namespace) -> dict:
    from ..config import load_hf_config
    tok = None
    if args.dataset_name != "random":
        from ..engine.processor import get_tokenizer
        tok = get_tokenizer(args.tokenizer or args.model)
    vocab = getattr(load_hf_config(args.model), "vocab_size", 32000)
    reqs = get_samples(args, tok, vocab)
    el = asyncio.run(run_async(args, reqs)) if args.async_engine else run_sync(args, reqs)
    tin = sum(r.prompt_len for r in reqs)
    tout = sum(r.expected_output_len for r in reqs) * args.n
    res = {"elapsed_time": el, "num_requests": len(reqs), "total_num_tokens": tin + tout,
           "requests_per_second": len(reqs) / el, "tokens_per_second": (tin + tout) / el,
           "output_tokens_per_second": tout / el}
    print(f"Throughput: {res['requests_per_second']:.2f} requests/s, {res['tokens_per_second']:.2f} total tokens/s, "
          f"{res['output_tokens_per_second']:.2f} output tokens/s")
    if args.output_json:
        with open(args.output_json, "w") as f:
            json.dump(res, f, indent=2)
    return res
