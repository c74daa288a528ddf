"""Multi-rank GPU paths rehearsed on ONE device: two tensor-parallel ranks share cuda:0
(``VDA_REHEARSE_ON_ONE_GPU``), so weight sharding on the GPU, the peer-memory all-reduce inside captured
hipGraphs, vocab-parallel logits and the external-launcher engine all run for real; and the ctypes RCCL
communicator is exercised with a one-rank communicator (signatures, explicit stream, graph capture)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, model, out, kw, extra_env, forced=None):
    """Run ``world`` ranks; with ``forced`` (one continuation per prompt) they score those tokens instead of
    generating."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
        cmd = [sys.executable, os.path.join(HERE, "scripts", "run_parallel_llm.py"), model, out, json.dumps(kw)]
        if forced is not None:
            cmd.append(json.dumps(forced))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=200)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for p, l in zip(procs, logs):
        assert p.returncode == 0, l[-3000:]
    return [json.load(open(f"{out}.{r}")) for r in range(world)], logs


def _teacher_forced_gap(world, model, out, kw, env, single_out):
    """Largest |difference| between the log-probabilities the single-rank run reported for its own tokens and the
    ones ``world`` ranks give the same tokens in the same contexts: a wrong shard of any projection shows up here
    even when greedy decoding happens to pick the same tokens."""
    tokens = json.load(open(f"{single_out}.0"))
    ref = json.load(open(f"{single_out}.0.lp"))
    _launch(world, model, out, kw, env, forced=tokens)
    worst = 0.0
    for r in range(world):
        got = json.load(open(f"{out}.{r}.lp"))
        worst = max(worst, max(abs(a - b) for x, y in zip(got, ref) for a, b in zip(x, y)))
    return worst


@pytest.mark.parametrize("eager,async_tp", [(True, False), (False, False), (False, True)])
def test_tp2_on_one_gpu(tmp_path, eager, async_tp):
    from tests.test_e2e_cpu import _save_tiny
    model, _ = _save_tiny(str(tmp_path), "llama")
    kw = dict(device="cuda", dtype="bfloat16", gpu_memory_utilization=0.03, enforce_eager=eager, max_num_seqs=8)
    single, _ = _launch(1, model, str(tmp_path / "one"), kw, {})
    env = {"VDA_REHEARSE_ON_ONE_GPU": "1", "VLLM_LOGGING_LEVEL": "INFO"}
    if async_tp:
        # prefill steps run as two micro-batches with their all-reduces on the communication stream,
        # decode steps replay graphs
        kw = dict(kw, compilation_config={"enable_async_tp": True})
        env["VDA_ASYNC_TP_MIN_TOKENS"] = "8"
    got, logs = _launch(2, model, str(tmp_path / "two"), dict(kw, tensor_parallel_size=2), env)
    assert got[0] == got[1], "tensor-parallel ranks disagree"
    assert any("peer-memory all-reduce enabled" in l for l in logs), logs[0][-2000:]
    # bf16 sums are reassociated across ranks: greedy picks may flip at near ties, log-probabilities may not move
    first = [a[0] == b[0] for a, b in zip(got[0], single[0])]
    assert sum(first) >= len(first) - 1
    gap = _teacher_forced_gap(2, model, str(tmp_path / "tf"), dict(kw, tensor_parallel_size=2), env,
                              str(tmp_path / "one"))
    assert gap < 0.1, gap


@pytest.mark.parametrize("kw2,eager", [({"pipeline_parallel_size": 2}, True), ({"pipeline_parallel_size": 2}, False),
                                       ({"pipeline_parallel_size": 2, "tensor_parallel_size": 2}, False)])
def test_pipeline_parallel_on_one_gpu(tmp_path, kw2, eager):
    """Two pipeline stages on one device: after the first hop (which fixes the layout) the activations travel through
    the peer-memory pipe with no metadata message, also when the second stage replays decode graphs; the result
    equals the single-rank run."""
    from tests.test_e2e_cpu import _save_tiny
    model, _ = _save_tiny(str(tmp_path), "llama")
    kw = dict(device="cuda", dtype="bfloat16", gpu_memory_utilization=0.03, enforce_eager=eager, max_num_seqs=8)
    single, _ = _launch(1, model, str(tmp_path / "one"), kw, {})
    env = {"VDA_REHEARSE_ON_ONE_GPU": "1", "VLLM_LOGGING_LEVEL": "INFO"}
    world = 2 * kw2.get("tensor_parallel_size", 1)
    got, logs = _launch(world, model, str(tmp_path / "pp"), dict(kw, **kw2), env)
    assert all(g == got[0] for g in got)
    assert any("peer-memory pipeline hand-off enabled" in l for l in logs), logs[0][-2000:]
    for r in range(world):
        hops = json.load(open(f"{tmp_path / 'pp'}.{r}.comm"))["pp"]["hops"]
        assert hops["metadata"] == 1 and hops["pipe"] >= 9, hops            # one layout hop, then the pipe only
    if world == 2:
        # the same kernels on the same rows, only on another rank: tokens equal the single-rank run
        assert got[0] == single[0]
    gap = _teacher_forced_gap(world, model, str(tmp_path / "tf"), dict(kw, **kw2), env, str(tmp_path / "one"))
    assert gap < (0.05 if world == 2 else 0.1), gap


@pytest.mark.parametrize("eager", [True, False])
def test_sequence_parallel_on_one_gpu(tmp_path, eager):
    """Sequence parallelism over two ranks: the row-parallel layers end in the peer-memory reduce-scatter and the
    next column-parallel layer starts from the peer-memory all-gather (no RCCL, no gloo on that path)."""
    from tests.test_e2e_cpu import _save_tiny
    model, _ = _save_tiny(str(tmp_path), "llama")
    kw = dict(device="cuda", dtype="bfloat16", gpu_memory_utilization=0.03, enforce_eager=eager, max_num_seqs=8)
    single, _ = _launch(1, model, str(tmp_path / "one"), kw, {})
    env = {"VDA_REHEARSE_ON_ONE_GPU": "1", "VLLM_LOGGING_LEVEL": "INFO"}
    kw2 = dict(kw, tensor_parallel_size=2, compilation_config={"enable_sequence_parallelism": True})
    got, logs = _launch(2, model, str(tmp_path / "two"), kw2, env)
    assert got[0] == got[1], "tensor-parallel ranks disagree"
    assert any("peer-memory reduce-scatter / all-gather in use" in l for l in logs), logs[0][-2000:]
    first = [a[0] == b[0] for a, b in zip(got[0], single[0])]
    assert sum(first) >= len(first) - 1
    gap = _teacher_forced_gap(2, model, str(tmp_path / "tf"), kw2, env, str(tmp_path / "one"))
    assert gap < 0.1, gap


@pytest.mark.parametrize("world,eager,quant", [(2, False, None), (2, True, None), (2, False, "fp8")])
def test_expert_parallel_on_one_gpu(tmp_path, world, eager, quant):
    """Expert parallelism with token routing: every rank sends its slice of the tokens to the expert
    owners through the peer-memory all-to-all-v, inside captured decode graphs when eager is off.
    With e4m3 experts that run the fp8 tiles (forced here for every batch size) the rows travel as e4m3 + one scale
    each: the owner skips its own quantisation pass and gets the same bytes it would have produced."""
    from tests.test_e2e_cpu import _save_tiny
    model, _ = _save_tiny(str(tmp_path), "mixtral")
    kw = dict(device="cuda", dtype="bfloat16", gpu_memory_utilization=0.03, enforce_eager=eager, max_num_seqs=8)
    base = {}
    if quant:
        kw["quantization"] = quant
        base = {"VDA_MOE_GROUPED_MAX_PAIRS_FP8": "0"}
    single, _ = _launch(1, model, str(tmp_path / "one"), kw, base)
    env = dict(base, VDA_REHEARSE_ON_ONE_GPU="1", VLLM_LOGGING_LEVEL="INFO")
    got, logs = _launch(world, model, str(tmp_path / "ep"),
                        dict(kw, tensor_parallel_size=world, enable_expert_parallel=True), env)
    assert all(g == got[0] for g in got), "expert-parallel ranks disagree"
    assert any("peer-memory all-to-all enabled" in l for l in logs), logs[0][-2000:]
    assert any("e4m3 rows on the wire" in l for l in logs) == bool(quant), logs[0][-2000:]
    first = [a[0] == b[0] for a, b in zip(got[0], single[0])]
    assert sum(first) >= len(first) - 1
    gap = _teacher_forced_gap(world, model, str(tmp_path / "tf"),
                              dict(kw, tensor_parallel_size=world, enable_expert_parallel=True), env,
                              str(tmp_path / "one"))
    assert gap < 0.1, gap


@pytest.mark.parametrize("world,eager", [(2, False), (3, False), (2, True)])
def test_token_parallel_on_one_gpu(tmp_path, world, eager):
    """Token parallelism through the peer-memory row exchange: scatter / gather of attention rows and the
    sampled-id row, inside captured decode graphs when eager is off; tokens must equal the single-rank run
    (every row is computed by the same kernels, only on another rank)."""
    from tests.test_e2e_cpu import _save_tiny
    model, _ = _save_tiny(str(tmp_path), "llama")
    kw = dict(device="cuda", dtype="bfloat16", gpu_memory_utilization=0.08, enforce_eager=eager, max_num_seqs=8,
              enable_prefix_caching=False)
    single, _ = _launch(1, model, str(tmp_path / "one"), kw, {})
    env = {"VDA_REHEARSE_ON_ONE_GPU": "1", "VLLM_LOGGING_LEVEL": "INFO"}
    got, logs = _launch(world, model, str(tmp_path / "tk"),
                        dict(kw, enable_token_parallel=True, token_parallel_size=world), env)
    assert any("peer-memory all-to-all enabled" in l for l in logs), logs[0][-2000:]
    if not eager:
        assert any("captured" in l and "decode graphs" in l for l in logs), logs[0][-2000:]
    for r in range(world):
        assert got[r] == single[0], f"rank {r} diverged"


def test_rccl_wrapper_single_rank_and_graph_capture():
    import torch.distributed as dist
    from vllm_distributed_amd.parallel.rccl import RcclCommunicator
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1, init_method=f"tcp://127.0.0.1:{_free_port()}")
    dev = torch.device("cuda:0")
    comm = RcclCommunicator(dist.group.WORLD, dev)
    x = torch.arange(1024, device=dev, dtype=torch.bfloat16)
    y = comm.all_reduce(x.clone())
    g = torch.empty(1024, device=dev, dtype=torch.bfloat16)
    comm.all_gather(g, x)
    rs = torch.empty(1024, device=dev, dtype=torch.bfloat16)
    comm.reduce_scatter(rs, x)
    b = x.clone()
    comm.broadcast(b, 0)
    torch.cuda.synchronize()
    for t in (y, g, rs, b):
        assert torch.equal(t, x)
    # collectives recorded into a hipGraph on the capture stream
    inp = torch.zeros(4096, device=dev, dtype=torch.float32)
    out = torch.empty_like(inp)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        comm.all_reduce(inp, out=out)
        out.mul_(2.0)
    inp.fill_(3.0)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.all(out == 6.0)
    comm.close()


def _bench(args, timeout=420):
    env = dict(os.environ, VDA_REHEARSE_ON_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    over = {"num_hidden_layers": 2, "hidden_size": 1024, "intermediate_size": 2048, "num_attention_heads": 16,
            "num_key_value_heads": 8, "vocab_size": 32000}
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), *args, "--steps", "2", "--warmup", "1",
           "--model", "llama-3-8b", "--hf-overrides", json.dumps(over), "--batch", "16", "--input-len", "128",
           "--output-len", "16", "--max-num-batched-tokens", "1024", "--gpu-memory-utilization", "0.04",
           "--time-budget", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) >= 2, r.stdout[-1500:]
    return lines[-1]


def test_bench_eight_ranks_on_one_gpu(tmp_path):
    """``python bench.py --gpus 8`` exactly as typed, no launcher around it: the process must start its own 8 ranks
    (external launcher, async TP, peer-memory all-reduce inside decode graphs; rehearsed on ONE device with a
    shrunken Llama), relay a JSON line per timed round and say in the last one what the collectives ran on."""
    last = _bench(["--gpus", "8"])
    assert last["n_gpus"] == 8 and last["steps"] == 2 and last["value"] > 0
    assert last["config"]["parallelism"] == "tp8"
    assert last["tp_world"] == 8 and last["peer_arena_ranks"] == 8
    assert last["allreduce_path"]["peer_memory"] + last["allreduce_path"]["fused_norm"] > 0


def test_bench_pipeline_parallel_on_one_gpu(tmp_path):
    """``--gpus 2 --pp 2``: two pipeline stages through the multi-process executor of one engine."""
    last = _bench(["--gpus", "2", "--pp", "2"])
    assert last["n_gpus"] == 2 and last["steps"] == 2 and last["value"] > 0
    assert last["config"]["parallelism"] == "pp2xtp1" and last["pp_world"] == 2
