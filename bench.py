"""Flagship serving benchmark: Llama-3-70B, tensor parallel over the N GPUs of one
node, synthetic 1024-token prompts, 128 generated tokens per request, bf16,
random-init weights.  One "step" is one full serving round: a fixed global batch
of requests is admitted at once and run to completion through the whole engine
(scheduler, chunked prefill, paged KV cache, hipGraph decode, sampling).  The
reported value is generated tokens per second over the timed rounds, aggregate
over all GPUs; p50 TTFT / ITL of the timed rounds are reported alongside.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29511 bench.py --gpus 8 --steps 3 --warmup 1
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("VLLM_ENABLE_V1_MULTIPROCESSING", "0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("VLLM_LOGGING_LEVEL", "WARNING")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="llama-3-70b")
    ap.add_argument("--batch", type=int, default=256, help="requests per round (global, fixed as N grows)")
    ap.add_argument("--input-len", type=int, default=1024)
    ap.add_argument("--output-len", type=int, default=128)
    ap.add_argument("--max-num-batched-tokens", type=int, default=16384)
    ap.add_argument("--enforce-eager", action="store_true")
    ap.add_argument("--no-async-tp", action="store_true", help="keep TP all-reduces on the compute stream")
    ap.add_argument("--gpu-memory-utilization", type=float, default=0.92)
    ap.add_argument("--quantization", default=None, help="e.g. fp8 (weights quantised while loading); default: none")
    ap.add_argument("--expert-parallel", action="store_true", help="MoE models: shard experts, not expert matrices")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from vllm_distributed_amd import LLM, SamplingParams
    from vllm_distributed_amd.inputs import TokensPrompt
    from vllm_distributed_amd.sampling_params import RequestOutputKind

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {a.gpus} does not match WORLD_SIZE {world}")
    kw = {}
    if a.quantization:
        kw["quantization"] = a.quantization
    if a.expert_parallel:
        kw["enable_expert_parallel"] = True
    if world > 1:
        kw["distributed_executor_backend"] = "external_launcher"
        if not a.no_async_tp:
            # prefill chunks run as two micro-batches whose TP all-reduces overlap with each other's GEMMs
            kw["compilation_config"] = {"enable_async_tp": True}
    llm = LLM(model=a.model, load_format="dummy", skip_tokenizer_init=True, dtype="bfloat16",
              tensor_parallel_size=a.gpus, max_model_len=a.input_len + a.output_len + 16,
              max_num_seqs=max(a.batch, 8), max_num_batched_tokens=a.max_num_batched_tokens,
              enable_prefix_caching=False, gpu_memory_utilization=a.gpu_memory_utilization,
              enforce_eager=a.enforce_eager, seed=0, **kw)
    vocab = llm.llm_engine.model_config.get_vocab_size()
    eng = llm.llm_engine
    sp = SamplingParams(temperature=0.0, max_tokens=a.output_len, ignore_eos=True,
                        output_kind=RequestOutputKind.CUMULATIVE)

    def prompts(round_idx: int):
        # distinct synthetic prompts every round (no reuse between rounds)
        g = torch.Generator().manual_seed(1000 + round_idx)
        ids = torch.randint(10, vocab - 10, (a.batch, a.input_len), generator=g)
        return [TokensPrompt(prompt_token_ids=row.tolist()) for row in ids]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_round(idx: int):
        ps = prompts(idx)
        t_add = time.perf_counter()
        for i, p in enumerate(ps):
            eng.add_request(f"r{idx}-{i}", p, sp)
        first, last, gaps, ntok = {}, {}, [], {}
        while eng.has_unfinished_requests():
            outs = eng.step()
            now = time.perf_counter()
            for o in outs:
                n = len(o.outputs[0].token_ids)
                rid = o.request_id
                if rid not in first:
                    first[rid] = now - t_add
                elif n > ntok.get(rid, 0):
                    gaps.append((now - last[rid]) / (n - ntok[rid]))
                last[rid], ntok[rid] = now, n
        assert len(ntok) == a.batch and all(v == a.output_len for v in ntok.values()), "round did not complete"
        return list(first.values()), gaps

    for w in range(a.warmup):
        one_round(-1 - w)
    sync()
    t0 = time.perf_counter()
    ttfts, itls = [], []
    for k in range(a.steps):
        f, g = one_round(k)
        ttfts += f
        itls += g
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    total_out = a.batch * a.output_len * a.steps
    if rank == 0:
        print(json.dumps({
            "metric": "output tokens/s per node (Llama-3-70B TP=N, 1024-in/128-out)" if "70b" in a.model
            else f"output tokens/s per node ({a.model}, {a.input_len}-in/{a.output_len}-out)",
            "value": round(total_out / dt, 2), "unit": "tokens/s", "n_gpus": a.gpus, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1000, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16" if not a.quantization else f"bf16 activations, {a.quantization} weights",
            "data": "synthetic random token prompts, random-init weights",
            "p50_ttft_ms": round(statistics.median(ttfts) * 1000, 2),
            "p50_itl_ms": round(statistics.median(itls) * 1000, 3) if itls else None,
            "total_tokens_per_s": round(a.batch * (a.input_len + a.output_len) * a.steps / dt, 1),
            "config": {"model": a.model, "global_batch": a.batch, "seq_len": a.input_len + a.output_len,
                       "input_len": a.input_len, "output_len": a.output_len, "parallelism": f"tp{a.gpus}"}}),
            flush=True)
    llm.shutdown()
    if world > 1 and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    # the measurement is out; leave without running interpreter-exit destructors of the communication
    # libraries (a helper thread still alive there aborts the process and would turn a finished run into a
    # failed one for the launcher)
    sys.stdout.flush()
    sys.stderr.flush()
    if os.environ.get("VDA_BENCH_SOFT_EXIT") == "1":
        return          # under a profiler: its at-exit handlers write the trace files
    os._exit(0)


if __name__ == "__main__":
    main()
