"""Online serving benchmark against an OpenAI-compatible endpoint: Poisson/gamma arrivals, ramp-up,
max-concurrency cap, streaming TTFT / TPOT / ITL / E2EL percentiles, goodput SLOs.
Reference: benchmarks/benchmark_serving.py, backend_request_func.py, vllm/benchmarks/serve.py."""
from __future__ import annotations

import argparse
import asyncio
import json
import time
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from .datasets import SampleRequest, add_dataset_args, get_samples


@dataclass
class RequestResult:
    success: bool = False
    latency: float = 0.0
    ttft: float = 0.0
    itl: list[float] = field(default_factory=list)
    prompt_len: int = 0
    output_tokens: int = 0
    error: str = ""
    text: str = ""


def add_cli_args(p: argparse.ArgumentParser) -> None:
    add_dataset_args(p)
    p.add_argument("--base-url", default=None)
    p.add_argument("--host", default="127.0.0.1")
    p.add_argument("--port", type=int, default=8000)
    p.add_argument("--endpoint", default="/v1/completions")
    p.add_argument("--backend", default="openai", choices=["openai", "openai-chat"])
    p.add_argument("--model", required=True)
    p.add_argument("--served-model-name", default=None)
    p.add_argument("--tokenizer", default=None)
    p.add_argument("--request-rate", type=float, default=float("inf"))
    p.add_argument("--burstiness", type=float, default=1.0)
    p.add_argument("--ramp-up-strategy", default=None, choices=["linear", "exponential"],
                   help="sweep the request rate from --ramp-up-start-rps to --ramp-up-end-rps over the run")
    p.add_argument("--ramp-up-start-rps", type=float, default=1.0)
    p.add_argument("--ramp-up-end-rps", type=float, default=1.0)
    p.add_argument("--max-concurrency", type=int, default=None)
    p.add_argument("--ignore-eos", action="store_true")
    p.add_argument("--percentile-metrics", default="ttft,tpot,itl")
    p.add_argument("--metric-percentiles", default="99")
    p.add_argument("--goodput", nargs="*", default=None, help="SLOs like ttft:200 tpot:50 e2el:5000 (ms)")
    p.add_argument("--save-result", action="store_true")
    p.add_argument("--result-filename", default=None)
    p.add_argument("--profile", action="store_true")


async def _one(client, url: str, model: str, r: SampleRequest, chat: bool, ignore_eos: bool) -> RequestResult:
    res = RequestResult(prompt_len=r.prompt_len)
    if chat:
        body = {"model": model, "messages": [{"role": "user", "content": r.prompt}],
                "max_completion_tokens": r.expected_output_len}
    else:
        prompt = r.prompt["prompt_token_ids"] if isinstance(r.prompt, dict) else r.prompt
        body = {"model": model, "prompt": prompt, "max_tokens": r.expected_output_len}
    body.update(stream=True, temperature=0.0, ignore_eos=ignore_eos, stream_options={"include_usage": True})
    t0 = time.perf_counter()
    last = t0
    try:
        async with client.stream("POST", url, json=body) as resp:
            if resp.status_code != 200:
                res.error = f"HTTP {resp.status_code}: {(await resp.aread())[:200]!r}"
                return res
            async for line in resp.aiter_lines():
                if not line.startswith("data: "):
                    continue
                data = line[6:]
                if data == "[DONE]":
                    break
                d = json.loads(data)
                now = time.perf_counter()
                ch = d.get("choices") or []
                if ch:
                    piece = (ch[0].get("delta", {}).get("content") if chat else ch[0].get("text")) or ""
                    if res.ttft == 0.0:
                        res.ttft = now - t0
                    else:
                        res.itl.append(now - last)
                    last = now
                    res.text += piece
                if d.get("usage"):
                    res.output_tokens = d["usage"].get("completion_tokens", 0)
        res.latency = last - t0
        res.success = True
        if not res.output_tokens:
            res.output_tokens = len(res.itl) + 1
    except Exception as e:   # noqa: BLE001
        res.error = repr(e)
    return res


def _rate_at(i: int, n: int, rate: float, ramp: Optional[str], start: float, end: float) -> float:
    """Request rate in force when request i of n is sent: constant, or ramping from ``start`` to ``end``
    (linear in the request index, or exponential) as in the reference's --ramp-up-strategy."""
    if not ramp:
        return rate
    f = i / max(n - 1, 1)
    if ramp == "linear":
        return start + (end - start) * f
    if ramp == "exponential":
        return start * (end / start) ** f
    raise ValueError(f"unknown ramp-up strategy {ramp!r} (linear, exponential)")


async def _arrivals(reqs, rate: float, burstiness: float, seed: int, ramp: Optional[str] = None,
                    ramp_start: float = 1.0, ramp_end: float = 1.0):
    g = np.random.default_rng(seed)
    n = len(reqs)
    for i, r in enumerate(reqs):
        yield r
        cur = _rate_at(i, n, rate, ramp, ramp_start, ramp_end)
        if cur != float("inf"):
            await asyncio.sleep(float(g.gamma(shape=burstiness, scale=1.0 / (cur * burstiness))))


def summarize(results: list[RequestResult], duration: float, pcts: list[float], goodput: Optional[dict]) -> dict:
    ok = [r for r in results if r.success]
    ttft = np.array([r.ttft for r in ok]) * 1e3
    tpot = np.array([(r.latency - r.ttft) / (r.output_tokens - 1) for r in ok if r.output_tokens > 1]) * 1e3
    itl = np.array([x for r in ok for x in r.itl]) * 1e3
    e2el = np.array([r.latency for r in ok]) * 1e3
    out = {"duration": duration, "completed": len(ok), "failed": len(results) - len(ok),
           "total_input_tokens": int(sum(r.prompt_len for r in ok)),
           "total_output_tokens": int(sum(r.output_tokens for r in ok)),
           "request_throughput": len(ok) / duration,
           "output_throughput": sum(r.output_tokens for r in ok) / duration,
           "total_token_throughput": sum(r.output_tokens + r.prompt_len for r in ok) / duration}
    for name, arr in (("ttft", ttft), ("tpot", tpot), ("itl", itl), ("e2el", e2el)):
        if arr.size == 0:
            arr = np.zeros(1)
        out[f"mean_{name}_ms"], out[f"median_{name}_ms"], out[f"std_{name}_ms"] = \
            float(arr.mean()), float(np.median(arr)), float(arr.std())
        for p in pcts:
            out[f"p{p:g}_{name}_ms"] = float(np.percentile(arr, p))
    if goodput:
        good = 0
        for r in ok:
            vals = {"ttft": r.ttft * 1e3, "e2el": r.latency * 1e3,
                    "tpot": (r.latency - r.ttft) / max(r.output_tokens - 1, 1) * 1e3}
            good += all(vals[k] <= v for k, v in goodput.items())
        out["request_goodput"] = good / duration
    return out


async def benchmark(args, reqs: list[SampleRequest], transport=None) -> dict:
    import httpx
    base = args.base_url or f"http://{args.host}:{args.port}"
    model = args.served_model_name or args.model
    chat = args.backend == "openai-chat"
    sem = asyncio.Semaphore(args.max_concurrency) if args.max_concurrency else None
    async with httpx.AsyncClient(base_url=base, timeout=None, transport=transport,
                                 limits=httpx.Limits(max_connections=None)) as client:
        probe = await _one(client, args.endpoint, model, reqs[0], chat, args.ignore_eos)
        if not probe.success:
            raise RuntimeError(f"initial test request failed: {probe.error}")
        if args.profile:
            await client.post("/start_profile")

        async def limited(r):
            if sem is None:
                return await _one(client, args.endpoint, model, r, chat, args.ignore_eos)
            async with sem:
                return await _one(client, args.endpoint, model, r, chat, args.ignore_eos)
        t0 = time.perf_counter()
        tasks = []
        async for r in _arrivals(reqs, args.request_rate, args.burstiness, args.seed,
                                 getattr(args, "ramp_up_strategy", None), getattr(args, "ramp_up_start_rps", 1.0),
                                 getattr(args, "ramp_up_end_rps", 1.0)):
            tasks.append(asyncio.create_task(limited(r)))
        results = await asyncio.gather(*tasks)
        dur = time.perf_counter() - t0
        if args.profile:
            await client.post("/stop_profile")
    goodput = None
    if args.goodput:
        goodput = {k: float(v) for k, v in (x.split(":") for x in args.goodput)}
    pcts = [float(x) for x in args.metric_percentiles.split(",")]
    return summarize(list(results), dur, pcts, goodput)


def main(args: argparse.Namespace) -> dict:
    from ..config import load_hf_config
    tok = None
    if args.dataset_name != "random" or args.backend == "openai-chat":
        from ..engine.processor import get_tokenizer
        tok = get_tokenizer(args.tokenizer or args.model)
    vocab = getattr(load_hf_config(args.model), "vocab_size", 32000)
    reqs = get_samples(args, tok, vocab)
    if args.backend == "openai-chat":
        for r in reqs:
            if isinstance(r.prompt, dict):
                r.prompt = tok.decode(r.prompt["prompt_token_ids"])
    res = asyncio.run(benchmark(args, reqs))
    print("{s:{c}^{n}}".format(s=" Serving Benchmark Result ", n=50, c="="))
    for k, v in res.items():
        print(f"{k:<40} {v:<10.2f}" if isinstance(v, float) else f"{k:<40} {v}")
    if args.save_result:
        fn = args.result_filename or f"serve-{int(time.time())}.json"
        with open(fn, "w") as f:
            json.dump(res, f, indent=2)
    return res
