"""Flagship serving benchmark: Llama-3-70B, tensor parallel over the N GPUs of one
node, synthetic 1024-token prompts, 128 generated tokens per request, bf16,
random-init weights.  One "step" is one full serving round: a fixed global batch
of requests is admitted at once and run to completion through the whole engine
(scheduler, chunked prefill, paged KV cache, hipGraph decode, sampling).  The
reported value is generated tokens per second over the timed rounds, aggregate
over all GPUs; p50 TTFT / ITL of the timed rounds are reported alongside.

A round of the 70B model takes tens of seconds, so the process works against a
wall-clock budget (--time-budget, start-up included): warm-up and timed rounds
that would not fit are dropped, at least one warm-up and three timed rounds
always run, and a complete JSON line is printed (flushed) after every timed
round with the number of rounds really timed in "steps", so the last line of
stdout is the result whenever the process ends.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus 8 --steps 3 --warmup 1          # starts its own 8 ranks (see below)
    python bench.py --gpus 8 --pp 2                         # PP 2 x TP 4 through the multi-process executor
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29511 bench.py --gpus 8 --steps 3 --warmup 1

Started with --gpus N > 1 and no WORLD_SIZE in the environment, the process (which has not touched the GPU)
starts ``torch.distributed.run`` with N ranks of this same file as a child, relays the child's stdout (rank 0's
JSON lines) and exits with the child's code.  Under ``--pp P`` one engine drives all N GPUs through the
multi-process executor (P pipeline stages of N / P tensor-parallel ranks); if that is started under
torch.distributed.run, rank 0 runs the engine and the other ranks wait.
"""
import argparse
import json
import os
import signal
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("VLLM_ENABLE_V1_MULTIPROCESSING", "0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("VLLM_LOGGING_LEVEL", "WARNING")
T_PROCESS_START = time.time()


def note(msg: str) -> None:
    """Progress lines go to stderr: stdout carries only the JSON result lines."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.time() - T_PROCESS_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def run_disagg(a) -> None:
    """xPyD deployment on one node through the peer-memory KV connector; one JSON line in this file's format."""
    import re
    m = re.fullmatch(r"(\d+)p(\d+)d", a.disagg.lower())
    if not m:
        raise SystemExit("--disagg takes xPyD, e.g. 1p1d or 4p4d")
    x, y = int(m.group(1)), int(m.group(2))
    from vllm_distributed_amd.benchmarks import disagg
    model = "llama-3-8b" if a.model == "llama-3-70b" else a.model
    out_len = 200 if a.output_len == 128 else a.output_len
    argv = ["--model", model, "--prefill", str(x), "--decode", str(y), "--input-len", str(a.input_len),
            "--output-len", str(out_len), "--num-prompts", str(a.num_prompts),
            "--request-rate", str(a.request_rate), "--gpu-memory-utilization", str(min(a.gpu_memory_utilization, 0.9)),
            "--max-num-batched-tokens", str(a.max_num_batched_tokens)]
    if a.enforce_eager:
        argv.append("--enforce-eager")
    if a.rehearse_on_one_gpu:
        argv.append("--rehearse-on-one-gpu")
    if a.kv_push:
        argv.append("--kv-push")
    res = disagg.main(argv)["disagg"]
    print(json.dumps({
        "metric": "disagg_output_tokens_per_s_per_gpu", "value": res["output_tok_s_per_gpu"], "unit": "tokens/s/GPU",
        "n_gpus": res["n_gpus"], "steps": res["completed"], "warmup": x * y, "ms_per_step": None,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": res["output_tok_s_per_gpu"] / 738.0,
        "dtype": "bf16", "data": "synthetic", "rehearsal_on_one_gpu": bool(a.rehearse_on_one_gpu),
        "ttft_ms": res["ttft_ms"], "itl_ms": res["itl_ms"], "output_tok_s": res["output_tok_s"],
        "num_errors": res["num_errors"],
        "config": {"model": model, "deployment": a.disagg.lower(), "input_len": a.input_len, "output_len": out_len,
                   "num_prompts": a.num_prompts, "request_rate": res["request_rate"],
                   "connector": "XgmiPushConnector" if a.kv_push else "XgmiConnector"},
    }), flush=True)


def launch_ranks(n: int) -> int:
    """Run this file as ``n`` ranks under torch.distributed.run (a child process: this one has not initialised the
    GPU and stays that way), pass the child's stdout through and return its exit code."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    note(f"starting {n} ranks: {' '.join(cmd[1:9])} ...")
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, bufsize=1, start_new_session=True)

    def forward(sig, _frame):                      # the driver's time limit must reach every rank
        try:
            os.killpg(child.pid, sig)
        except ProcessLookupError:
            pass
    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, forward)
    for line in child.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return child.wait()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--pp", type=int, default=1,
                    help="pipeline stages (tensor parallel degree = gpus / pp); runs through the multi-process "
                         "executor of one engine")
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
    ap.add_argument("--time-budget", type=float, default=float(os.environ.get("VDA_BENCH_BUDGET_S", "420")),
                    help="wall-clock seconds for the whole process (start-up included); warm-up and timed rounds "
                         "that would not fit are dropped (never below one warm-up and --min-steps timed rounds) "
                         "and the JSON line reports the rounds really run; 0 = run exactly --steps/--warmup")
    ap.add_argument("--min-steps", type=int, default=3, help="timed rounds that run whatever the budget says")
    ap.add_argument("--sync-scheduling", action="store_true",
                    help="schedule step N + 1 only after step N's results are on the host (default: run ahead)")
    ap.add_argument("--hf-overrides", type=json.loads, default=None,
                    help="JSON overrides of the model config (rehearsals of the N-rank launch with a small model)")
    ap.add_argument("--disagg", default=None, metavar="xPyD",
                    help="disaggregated prefill/decode instead of the flagship run: e.g. 1p1d or 4p4d, Llama-3-8B, "
                         "1024-token prompts, 200 generated tokens (benchmarks/disagg.py); needs x + y GPUs unless "
                         "--rehearse-on-one-gpu")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="--disagg: every engine on GPU 0")
    ap.add_argument("--kv-push", action="store_true", help="--disagg: push blocks while prefilling (landing pool)")
    ap.add_argument("--num-prompts", type=int, default=200, help="--disagg: requests")
    ap.add_argument("--request-rate", type=float, default=float("inf"), help="--disagg: Poisson arrivals per second")
    a = ap.parse_args()
    if a.disagg:
        return run_disagg(a)
    if a.pp < 1 or a.gpus % a.pp:
        raise SystemExit(f"--pp {a.pp} does not divide --gpus {a.gpus}")
    launched = "WORLD_SIZE" in os.environ
    if a.gpus > 1 and a.pp == 1 and not launched:
        sys.exit(launch_ranks(a.gpus))             # nothing above has touched the GPU

    import torch
    import torch.distributed as dist
    from vllm_distributed_amd import LLM, SamplingParams
    from vllm_distributed_amd.inputs import TokensPrompt
    from vllm_distributed_amd.sampling_params import RequestOutputKind

    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    if a.pp > 1:
        # one engine, multi-process executor: its own worker processes, one per GPU.  Under a launcher rank 0 runs
        # it and the others only keep the launcher's rendezvous alive.
        if world > 1:
            import datetime
            dist.init_process_group("gloo", timeout=datetime.timedelta(hours=3))
            if rank != 0:
                dist.barrier()
                dist.destroy_process_group()
                return
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                      "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
                os.environ.pop(k, None)            # the executor's workers make their own rendezvous
        world_sync = world
        world = 1
    elif world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} does not match WORLD_SIZE {world}")
    else:
        world_sync = 1
    tp = a.gpus // a.pp
    kw = {}
    if a.quantization:
        kw["quantization"] = a.quantization
    if a.expert_parallel:
        kw["enable_expert_parallel"] = True
    if a.pp > 1:
        kw["distributed_executor_backend"] = "mp"
        kw["pipeline_parallel_size"] = a.pp
    elif world > 1:
        kw["distributed_executor_backend"] = "external_launcher"
        if not a.no_async_tp:
            # prefill chunks run as two micro-batches whose TP all-reduces overlap with each other's GEMMs
            kw["compilation_config"] = {"enable_async_tp": True}
    on_gpu_here = a.pp == 1                        # with the multi-process executor this process never uses the GPU
    llm = LLM(model=a.model, load_format="dummy", skip_tokenizer_init=True, dtype="bfloat16",
              tensor_parallel_size=tp, max_model_len=a.input_len + a.output_len + 16,
              max_num_seqs=max(a.batch, 8), max_num_batched_tokens=a.max_num_batched_tokens,
              enable_prefix_caching=False, gpu_memory_utilization=a.gpu_memory_utilization,
              enforce_eager=a.enforce_eager, seed=0, hf_overrides=a.hf_overrides,
              async_scheduling=not a.sync_scheduling, **kw)
    vocab = llm.llm_engine.model_config.get_vocab_size()
    eng = llm.llm_engine
    sp = SamplingParams(temperature=0.0, max_tokens=a.output_len, ignore_eos=True,
                        output_kind=RequestOutputKind.CUMULATIVE)
    note(f"engine ready after {time.time() - T_PROCESS_START:.1f} s")

    def prompts(round_idx: int):
        # distinct synthetic prompts every round (no reuse between rounds)
        g = torch.Generator().manual_seed(1000 + round_idx)
        ids = torch.randint(10, vocab - 10, (a.batch, a.input_len), generator=g)
        return [TokensPrompt(prompt_token_ids=row.tolist()) for row in ids]

    def sync():
        if world > 1:
            dist.barrier()
        if on_gpu_here:
            torch.cuda.synchronize()
        # multi-process executor: a round ends when every request's last token has come back from the workers,
        # which read it off the device themselves; nothing is left in flight to wait for

    def elapsed() -> float:
        """Seconds since process start, the same number on every rank (the slowest rank's clock)."""
        e = time.time() - T_PROCESS_START
        if world > 1:
            t = torch.tensor([e], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e = float(t.item())
        return e

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

    budget = a.time_budget
    # ---- warm-up: at least one full round; more only while min_steps timed rounds still fit behind them ----
    warm_done, round_s = 0, 0.0
    for w in range(max(a.warmup, 1)):
        if w > 0 and budget > 0 and elapsed() + round_s * (1 + max(a.min_steps, 1)) * 1.1 > budget:
            break
        sync()
        t_w = time.perf_counter()
        one_round(-1 - w)
        sync()
        round_s = time.perf_counter() - t_w
        warm_done += 1
        note(f"warm-up round {warm_done}: {round_s:.1f} s")
    if a.warmup == 0:
        warm_done = 1      # one round always runs untimed: the first round pays one-time allocations

    # ---- timed rounds: each bracketed by barrier + device sync; a full JSON line after every round ----
    dt, steps_done, ttfts, itls = 0.0, 0, [], []

    def comm_fields() -> dict:
        """What the collectives really ran on (rank 0's tensor-parallel group): ranks under the RCCL communicator,
        ranks of the peer-memory arena, all-reduce calls per path."""
        try:
            info = llm.collective_rpc("comm_info")[0] or {}
            tpi = info.get("tp") or {}
            ppi = info.get("pp") or {}
            return {"rccl_ranks": tpi.get("rccl_ranks", 0), "peer_arena_ranks": tpi.get("peer_arena_ranks", 0),
                    "allreduce_path": tpi.get("allreduce_path", {}), "tp_world": tpi.get("world_size", 1),
                    "pp_world": ppi.get("world_size", 1)}
        except Exception as e:   # noqa: BLE001 - the result line must still be printed
            return {"comm_info_error": repr(e)}

    def report():
        d = dt
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        comm = comm_fields()                       # a collective_rpc: every rank makes it
        if rank != 0:
            return
        total_out = a.batch * a.output_len * steps_done
        print(json.dumps({
            "metric": "output tokens/s per node (Llama-3-70B TP=N, 1024-in/128-out)" if "70b" in a.model
            else f"output tokens/s per node ({a.model}, {a.input_len}-in/{a.output_len}-out)",
            "value": round(total_out / d, 2), "unit": "tokens/s", "n_gpus": a.gpus, "steps": steps_done,
            "warmup": warm_done, "ms_per_step": round(d / steps_done * 1000, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16" if not a.quantization else f"bf16 activations, {a.quantization} weights",
            "data": "synthetic random token prompts, random-init weights",
            "p50_ttft_ms": round(statistics.median(ttfts) * 1000, 2),
            "p50_itl_ms": round(statistics.median(itls) * 1000, 3) if itls else None,
            "total_tokens_per_s": round(a.batch * (a.input_len + a.output_len) * steps_done / d, 1),
            "requested_steps": a.steps, "requested_warmup": a.warmup, "time_budget_s": budget,
            "step_definition": f"one serving round: {a.batch} requests admitted at once and run to completion",
            **comm,
            "config": {"model": a.model, "global_batch": a.batch, "seq_len": a.input_len + a.output_len,
                       "input_len": a.input_len, "output_len": a.output_len,
                       "parallelism": f"tp{tp}" if a.pp == 1 else f"pp{a.pp}xtp{tp}"}}),
            flush=True)

    for k in range(a.steps):
        if budget > 0 and k >= a.min_steps and elapsed() + round_s * 1.1 > budget:
            note(f"time budget: stopping after {k} of {a.steps} timed rounds")
            break
        sync()
        t0 = time.perf_counter()
        f, g = one_round(k)
        sync()
        r = time.perf_counter() - t0
        dt += r
        round_s = max(round_s * 0.5, r)
        steps_done += 1
        ttfts += f
        itls += g
        report()
    llm.shutdown()
    if a.pp > 1 and world_sync > 1:
        dist.barrier()                             # releases the ranks that only waited
        dist.destroy_process_group()
    if world > 1 and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    sys.stderr.flush()
    if os.environ.get("VDA_BENCH_HARD_EXIT") == "1":
        os._exit(0)


if __name__ == "__main__":
    main()
