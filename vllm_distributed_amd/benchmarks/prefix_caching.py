"""Prefix-caching benchmark: the same set of prompts (a long shared prefix + distinct tails, or fully
repeated prompts) run with and without prefix caching; reports time, speed-up and the cache hit rate.
Reference: benchmarks/benchmark_prefix_caching.py."""
from __future__ import annotations

import argparse
import json
import time

import numpy as np


def add_cli_args(p: argparse.ArgumentParser) -> None:
    from ..engine.arg_utils import EngineArgs
    p.add_argument("--model", type=str, required=True)
    p.add_argument("--num-prompts", type=int, default=64)
    p.add_argument("--prefix-len", type=int, default=1024, help="tokens shared by every prompt")
    p.add_argument("--suffix-len", type=int, default=64, help="distinct tokens behind the shared prefix")
    p.add_argument("--output-len", type=int, default=16)
    p.add_argument("--repeat-count", type=int, default=1, help="send every prompt this many times")
    p.add_argument("--output-json", default=None)
    EngineArgs.add_cli_args(p)


def _run(args, enable: bool, prompts) -> tuple[float, dict]:
    from ..engine.arg_utils import EngineArgs
    from ..entrypoints.llm import LLM
    from ..sampling_params import SamplingParams
    ea = EngineArgs.from_cli_args(args)
    ea.enable_prefix_caching = enable
    llm = LLM.from_engine_args(ea)
    sp = SamplingParams(temperature=0.0, max_tokens=args.output_len, ignore_eos=True, detokenize=False)
    llm.generate(prompts[:1], sp, use_tqdm=False)              # warm-up (also seeds the cache with the prefix)
    t0 = time.perf_counter()
    outs = llm.generate(prompts, sp, use_tqdm=False)
    dt = time.perf_counter() - t0
    cached = sum(int(getattr(o, "num_cached_tokens", 0) or 0) for o in outs)
    total = sum(len(o.prompt_token_ids) for o in outs)
    llm.shutdown()
    return dt, {"cached_tokens": cached, "prompt_tokens": total}


def main(args: argparse.Namespace) -> dict:
    from ..config import load_hf_config
    vocab = int(load_hf_config(args.model).vocab_size)
    g = np.random.default_rng(getattr(args, "seed", 0) or 0)
    prefix = g.integers(10, vocab - 10, args.prefix_len).tolist()
    prompts = [{"prompt_token_ids": prefix + g.integers(10, vocab - 10, args.suffix_len).tolist()}
               for _ in range(args.num_prompts)] * max(args.repeat_count, 1)
    t_off, _ = _run(args, False, prompts)
    t_on, st = _run(args, True, prompts)
    res = {"num_prompts": len(prompts), "prefix_len": args.prefix_len, "suffix_len": args.suffix_len,
           "seconds_without_cache": round(t_off, 3), "seconds_with_cache": round(t_on, 3),
           "speedup": round(t_off / t_on, 2), "hit_rate": round(st["cached_tokens"] / max(st["prompt_tokens"], 1), 3)}
    print(json.dumps(res))
    if args.output_json:
        with open(args.output_json, "w") as f:
            json.dump(res, f, indent=2)
    return res
