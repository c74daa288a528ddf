"""Disaggregated prefill / decode benchmark: x prefill engines + y decode engines on x + y (groups of) GPUs of
one node, requests arriving as a Poisson process, the KV cache of every prompt moving prefill -> decode
GPU through the XgmiConnector (peer memory over xGMI).

The harness is the router a deployment's proxy would be (entrypoints/disagg_proxy.py), without HTTP in the
way: every request goes to a prefill engine with ``max_tokens=1`` and ``do_remote_decode``, the
``kv_transfer_params`` of its answer go with the same prompt to a decode engine, and the decode engine's
token stream is what the client sees.  Every engine is an AsyncLLM whose engine core runs in a process of
its own, started with a device list that puts the engine's GPUs first and leaves the others visible (peer
mappings need both ends visible).

Reported: output tokens/s of the deployment and per GPU, TTFT and ITL percentiles as seen after the
router (TTFT = arrival -> first token streamed by the decode engine).
Workload and metric definitions: the reference's P2pNccl study (docs/design/v1/p2p_nccl_connector.md:309-354,
1024-token prompts, 200 output tokens) and benchmarks/disagg_benchmarks/disagg_performance_benchmark.sh:88-155.

    python -m vllm_distributed_amd.benchmarks.disagg --model llama-3-8b --load-format dummy \\
        --prefill 1 --decode 1 --num-prompts 200 --request-rate 3
    # all engines on one GPU (functional rehearsal of the whole path, not a performance number):
    python -m vllm_distributed_amd.benchmarks.disagg --rehearse-on-one-gpu ...
"""
from __future__ import annotations

import argparse
import asyncio
import json
import os
import socket
import sys
import time
from typing import Optional

import numpy as np


def add_cli_args(p: argparse.ArgumentParser) -> None:
    p.add_argument("--model", default="llama-3-8b")
    p.add_argument("--load-format", default="dummy")
    p.add_argument("--dtype", default="bfloat16")
    p.add_argument("--device", default="auto", help="auto | cuda | cpu (cpu: plumbing test over the socket data plane)")
    p.add_argument("--vocab-size", type=int, default=30000, help="prompt token ids are drawn below this")
    p.add_argument("--prefill", type=int, default=1, help="number of prefill engines")
    p.add_argument("--decode", type=int, default=1, help="number of decode engines")
    p.add_argument("--tensor-parallel-size", type=int, default=1, help="GPUs per engine")
    p.add_argument("--input-len", type=int, default=1024)
    p.add_argument("--output-len", type=int, default=200)
    p.add_argument("--num-prompts", type=int, default=200)
    p.add_argument("--request-rate", type=float, default=float("inf"), help="requests/s (Poisson); inf = all at once")
    p.add_argument("--max-num-seqs", type=int, default=256)
    p.add_argument("--max-num-batched-tokens", type=int, default=8192)
    p.add_argument("--gpu-memory-utilization", type=float, default=0.9)
    p.add_argument("--hf-overrides", type=json.loads, default=None)
    p.add_argument("--enforce-eager", action="store_true")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--rehearse-on-one-gpu", action="store_true",
                   help="every engine on GPU 0 with a share of its memory: exercises the path, measures nothing")
    p.add_argument("--kv-push", action="store_true",
                   help="push mode: prefill engines write blocks into the decode engine's landing pool while they "
                        "compute (XgmiPushConnector) instead of the decode engine pulling them afterwards")
    p.add_argument("--baseline-single", action="store_true",
                   help="also run the same workload on ONE ordinary engine (the study's 'single instance' row)")
    p.add_argument("--output-json", default=None)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _device_list(first: list[int], total: int) -> str:
    rest = [d for d in range(total) if d not in first]
    return ",".join(str(d) for d in first + rest)


def _make_engine(args, role: Optional[str], gpus: list[int], n_gpus_total: int, share: float):
    """One AsyncLLM with its engine core in a child process.  The child inherits the device list set here."""
    from ..engine.arg_utils import AsyncEngineArgs
    from ..engine.async_llm import AsyncLLM
    if not args.rehearse_on_one_gpu and args.device != "cpu":
        os.environ["HIP_VISIBLE_DEVICES"] = _device_list(gpus, n_gpus_total)
    kt = None
    port = _free_port()
    if role is not None:
        kt = dict(kv_connector="XgmiPushConnector" if args.kv_push else "XgmiConnector", kv_role=role, kv_port=port)
    ea = AsyncEngineArgs(
        model=args.model, load_format=args.load_format, dtype=args.dtype, skip_tokenizer_init=True,
        tensor_parallel_size=args.tensor_parallel_size, max_model_len=args.input_len + args.output_len + 16,
        max_num_seqs=args.max_num_seqs, max_num_batched_tokens=args.max_num_batched_tokens,
        gpu_memory_utilization=share, hf_overrides=args.hf_overrides, enforce_eager=args.enforce_eager,
        enable_prefix_caching=False, kv_transfer_config=kt, seed=args.seed, disable_log_stats=True,
        device=args.device)
    eng = AsyncLLM.from_engine_args(ea)
    eng.kv_port = port
    return eng


async def _drive(args, route, n_gpus: int, label: str) -> dict:
    """Send the workload through ``route(i, prompt) -> async iterator of (n_tokens_so_far, timestamp)``."""
    rng = np.random.default_rng(args.seed)
    prompts = [rng.integers(10, args.vocab_size, size=args.input_len).tolist() for _ in range(args.num_prompts)]
    gaps = (rng.exponential(1.0 / args.request_rate, size=args.num_prompts)
            if np.isfinite(args.request_rate) else np.zeros(args.num_prompts))
    ttft, itl, done_tokens = [], [], [0]
    errors: list[str] = []

    async def one(i: int, t_arrival: float):
        try:
            prev_n, prev_t = 0, t_arrival
            async for n, t in route(i, prompts[i]):
                if n <= prev_n:
                    continue
                if prev_n == 0:
                    ttft.append(t - t_arrival)
                    if n > 1:           # several tokens in the first message
                        itl.extend([0.0] * (n - 1))
                else:
                    itl.extend([(t - prev_t) / (n - prev_n)] * (n - prev_n))
                prev_n, prev_t = n, t
            done_tokens[0] += prev_n
        except Exception as e:          # noqa: BLE001 - reported with the result
            errors.append(f"request {i}: {type(e).__name__}: {e}")

    t0 = time.perf_counter()
    tasks = []
    for i in range(args.num_prompts):
        if gaps[i] > 0:
            await asyncio.sleep(float(gaps[i]))
        tasks.append(asyncio.create_task(one(i, time.perf_counter())))
    await asyncio.gather(*tasks)
    wall = time.perf_counter() - t0

    def pct(xs, q):
        return float(np.percentile(xs, q)) * 1e3 if xs else float("nan")
    return {
        "config": label, "n_gpus": n_gpus, "num_prompts": args.num_prompts, "completed": len(ttft),
        "input_len": args.input_len, "output_len": args.output_len,
        "request_rate": None if not np.isfinite(args.request_rate) else args.request_rate,
        "duration_s": wall, "output_tok_s": done_tokens[0] / wall, "output_tok_s_per_gpu": done_tokens[0] / wall / n_gpus,
        "ttft_ms": {"mean": float(np.mean(ttft)) * 1e3 if ttft else float("nan"), "p50": pct(ttft, 50), "p99": pct(ttft, 99)},
        "itl_ms": {"mean": float(np.mean(itl)) * 1e3 if itl else float("nan"), "p50": pct(itl, 50), "p99": pct(itl, 99)},
        "errors": errors[:5], "num_errors": len(errors),
    }


async def run_disagg(args) -> dict:
    from ..sampling_params import SamplingParams
    tp = args.tensor_parallel_size
    n_inst = args.prefill + args.decode
    n_gpus = n_inst * tp
    share = args.gpu_memory_utilization / n_inst if args.rehearse_on_one_gpu else args.gpu_memory_utilization
    engines = []
    try:
        for k in range(n_inst):
            role = "kv_producer" if k < args.prefill else "kv_consumer"
            gpus = list(range(k * tp, (k + 1) * tp))
            engines.append(_make_engine(args, role, gpus, n_gpus, share))
        P, D = engines[:args.prefill], engines[args.prefill:]

        async def route(i: int, prompt: list[int]):
            p, d = P[i % len(P)], D[i % len(D)]
            rid = f"bench-{i}"
            ktp = {"do_remote_decode": True}
            if args.kv_push:                    # the router picks the decode engine first: that is where blocks go
                ktp.update(push_host="127.0.0.1", push_port=d.kv_port)
            sp_p = SamplingParams(temperature=0.0, max_tokens=1, ignore_eos=True, extra_args={"kv_transfer_params": ktp})
            last = None
            async for out in p.generate({"prompt_token_ids": prompt}, sp_p, rid):
                last = out
            params = getattr(last, "kv_transfer_params", None)
            extra = {"kv_transfer_params": params} if params else None    # none: the decode engine prefills itself
            sp_d = SamplingParams(temperature=0.0, max_tokens=args.output_len, ignore_eos=True, extra_args=extra)
            async for out in d.generate({"prompt_token_ids": prompt}, sp_d, rid):
                yield len(out.outputs[0].token_ids), time.perf_counter()

        # one short request through every (prefill, decode) pair first: handshakes and IPC mappings
        warm_args = argparse.Namespace(**{**vars(args), "num_prompts": len(P) * len(D), "request_rate": float("inf"),
                                          "output_len": min(args.output_len, 4)})
        await _drive(warm_args, route, n_gpus, "warm-up")
        res = await _drive(args, route, n_gpus, f"{args.prefill}p{args.decode}d")
    finally:
        for e in engines:
            e.shutdown()
    return res


async def run_single(args) -> dict:
    from ..sampling_params import SamplingParams
    eng = _make_engine(args, None, list(range(args.tensor_parallel_size)), args.tensor_parallel_size,
                       args.gpu_memory_utilization / (2 if args.rehearse_on_one_gpu else 1))
    try:
        async def route(i: int, prompt: list[int]):
            sp = SamplingParams(temperature=0.0, max_tokens=args.output_len, ignore_eos=True)
            async for out in eng.generate({"prompt_token_ids": prompt}, sp, f"single-{i}"):
                yield len(out.outputs[0].token_ids), time.perf_counter()
        return await _drive(args, route, args.tensor_parallel_size, "single")
    finally:
        eng.shutdown()


def main(argv: Optional[list[str]] = None) -> dict:
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    add_cli_args(p)
    args = p.parse_args(argv)
    os.environ["VLLM_ENABLE_V1_MULTIPROCESSING"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = {"disagg": asyncio.run(run_disagg(args))}
    if args.baseline_single:
        out["single"] = asyncio.run(run_single(args))
        a, b = out["disagg"], out["single"]
        out["per_gpu_vs_single"] = a["output_tok_s_per_gpu"] / b["output_tok_s_per_gpu"] if b["output_tok_s_per_gpu"] else None
    out["rehearsal"] = bool(args.rehearse_on_one_gpu)
    print(json.dumps(out), flush=True)
    if args.output_json:
        with open(args.output_json, "w") as f:
            json.dump(out, f, indent=1)
    return out


if __name__ == "__main__":
    res = main()
    sys.exit(1 if res["disagg"]["num_errors"] else 0)
