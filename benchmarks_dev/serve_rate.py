"""Online-serving load test without the HTTP layer: requests arrive as a Poisson process at a fixed rate
and stream through AsyncLLM (engine core in its own process, as under the API server).  Synthetic token-id
prompts, random-init weights.  Reports what the reference's serving benchmark reports: output tokens/s and
TTFT / ITL percentiles (reference workload: Llama-3.1-8B, 1024-in / 1024-out, request rate 3,
docs/design/v1/p2p_nccl_connector.md:309-341).

    python benchmarks_dev/serve_rate.py --model llama-3.1-8b --rate 3 --num-prompts 600
"""
import argparse
import asyncio
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VLLM_LOGGING_LEVEL", "WARNING")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np


async def main(a) -> None:
    from vllm_distributed_amd import SamplingParams
    from vllm_distributed_amd.engine.arg_utils import AsyncEngineArgs
    from vllm_distributed_amd.engine.async_llm import AsyncLLM
    from vllm_distributed_amd.inputs import TokensPrompt
    from vllm_distributed_amd.sampling_params import RequestOutputKind
    args = AsyncEngineArgs(model=a.model, load_format="dummy", skip_tokenizer_init=True, dtype="bfloat16",
                           max_model_len=a.input_len + a.output_len + 16, max_num_seqs=a.max_num_seqs,
                           enable_prefix_caching=False, gpu_memory_utilization=0.9, seed=0)
    llm = AsyncLLM.from_engine_args(args)
    rng = np.random.default_rng(0)
    vocab = 32000
    sp = SamplingParams(temperature=0.0, max_tokens=a.output_len, ignore_eos=True, output_kind=RequestOutputKind.DELTA)
    ttft, itl, done_tok, lat = [], [], [0], []

    async def one(i: int):
        ids = rng.integers(10, vocab - 10, a.input_len).tolist()
        t0 = last = time.perf_counter()
        first = True
        n = 0
        async for out in llm.generate(TokensPrompt(prompt_token_ids=ids), sp, f"r{i}"):
            now = time.perf_counter()
            k = len(out.outputs[0].token_ids)
            if k == 0:
                continue
            if first:
                ttft.append(now - t0)
                first = False
                if k > 1:
                    itl.extend([0.0] * (k - 1))
            else:
                itl.extend([(now - last) / k] * k)
            last = now
            n += k
        done_tok[0] += n
        lat.append(time.perf_counter() - t0)

    # warm-up: a few requests outside the measurement
    await asyncio.gather(*(one(-1 - j) for j in range(4)))
    ttft.clear(); itl.clear(); lat.clear(); done_tok[0] = 0
    gaps = rng.exponential(1.0 / a.rate, a.num_prompts) if a.rate > 0 else np.zeros(a.num_prompts)
    t_start = time.perf_counter()
    tasks = []
    for i in range(a.num_prompts):
        tasks.append(asyncio.create_task(one(i)))
        await asyncio.sleep(float(gaps[i]))
        if i % 100 == 99:
            print(f"[{time.perf_counter() - t_start:7.1f} s] sent {i + 1}, finished {len(lat)}", flush=True)
    await asyncio.gather(*tasks)
    dur = time.perf_counter() - t_start
    pct = lambda xs, p: float(np.percentile(np.asarray(xs) * 1000, p))          # noqa: E731
    res = {"model": a.model, "request_rate": a.rate, "num_prompts": a.num_prompts, "input_len": a.input_len,
           "output_len": a.output_len, "duration_s": round(dur, 1),
           "request_throughput": round(a.num_prompts / dur, 3), "output_tok_per_s": round(done_tok[0] / dur, 1),
           "total_tok_per_s": round((done_tok[0] + a.num_prompts * a.input_len) / dur, 1),
           "ttft_ms": {"mean": round(float(np.mean(ttft)) * 1000, 2), "p50": round(pct(ttft, 50), 2),
                       "p99": round(pct(ttft, 99), 2)},
           "itl_ms": {"mean": round(float(np.mean(itl)) * 1000, 3), "p50": round(pct(itl, 50), 3),
                      "p99": round(pct(itl, 99), 3)},
           "e2el_ms": {"p50": round(pct(lat, 50), 1), "p99": round(pct(lat, 99), 1)},
           "data": "synthetic token prompts, random-init weights, bf16, one MI355X"}
    print(json.dumps(res), flush=True)
    llm.shutdown()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama-3.1-8b")
    ap.add_argument("--rate", type=float, default=3.0)
    ap.add_argument("--num-prompts", type=int, default=600)
    ap.add_argument("--input-len", type=int, default=1024)
    ap.add_argument("--output-len", type=int, default=1024)
    ap.add_argument("--max-num-seqs", type=int, default=256)
    asyncio.run(main(ap.parse_args()))
