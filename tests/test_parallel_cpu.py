"""Tensor / pipeline parallel correctness on CPU (gloo, 2 processes, torchrun-style
external launcher): every rank must produce the tokens of the single-process run
(modelled on the reference's tests/distributed/test_torchrun_example.py and
test_pipeline_parallel.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from tests.test_e2e_cpu import _save_tiny
from vllm_distributed_amd import LLM, SamplingParams

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, model, out, kw):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "scripts", "run_parallel_llm.py"), model,
                                       out, json.dumps(kw)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for p, l in zip(procs, logs):
        assert p.returncode == 0, l[-3000:]
    return [json.load(open(f"{out}.{r}")) for r in range(world)]


def _reference(model):
    prompts = [[(i * 7 + j) % 300 for j in range(20 + 5 * i)] for i in range(5)]
    llm = LLM(model=model, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256,
              max_num_seqs=8, max_num_batched_tokens=64)
    outs = llm.generate(prompt_token_ids=prompts,
                        sampling_params=SamplingParams(temperature=0, max_tokens=10, ignore_eos=True))
    llm.shutdown()
    return [list(o.outputs[0].token_ids) for o in outs]


@pytest.mark.parametrize("kind,kw", [("llama", {"tensor_parallel_size": 2}),
                                     ("opt", {"tensor_parallel_size": 2}),
                                     ("bloom", {"tensor_parallel_size": 2}),          # ALiBi slopes per rank
                                     ("falcon_new", {"tensor_parallel_size": 2}),     # one KV head, replicated
                                     ("olmo2", {"tensor_parallel_size": 2}),          # q/k norm over all heads
                                     ("jamba", {"tensor_parallel_size": 2}),          # Mamba channels + experts sharded
                                     ("llama", {"pipeline_parallel_size": 2}),
                                     ("llama", {"tensor_parallel_size": 2, "compilation_config":
                                                {"enable_sequence_parallelism": True}}),
                                     ("llama", {"tensor_parallel_size": 2, "compilation_config":
                                                {"enable_async_tp": True}}),
                                     ("llama", {"tensor_parallel_size": 2, "pipeline_parallel_size": 1,
                                                "max_num_batched_tokens": 256, "compilation_config":
                                                {"enable_async_tp": True}})])
def test_two_rank_matches_single(tmp_path, kind, kw, monkeypatch):
    # async TP: every step of at least 8 tokens runs as two micro-batches (mixed decode + chunked prefill)
    monkeypatch.setenv("VDA_ASYNC_TP_MIN_TOKENS", "8")
    model, _ = _save_tiny(str(tmp_path), kind)
    ref = _reference(model)
    got = _launch(2, model, str(tmp_path / "out"), kw)
    for r, toks in enumerate(got):
        assert toks == ref, f"rank {r} diverged with {kw}"


@pytest.mark.parametrize("kw", [{"tensor_parallel_size": 2}, {"pipeline_parallel_size": 2}])
def test_multiproc_executor_matches_single(tmp_path, kw):
    """Worker processes behind the shared-memory RPC ring (the default executor for TP/PP>1);
    PP=2 also exercises the batch queue with two batches in flight."""
    model, _ = _save_tiny(str(tmp_path), "llama")
    ref = _reference(model)
    prompts = [[(i * 7 + j) % 300 for j in range(20 + 5 * i)] for i in range(5)]
    llm = LLM(model=model, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256,
              max_num_seqs=8, max_num_batched_tokens=64, distributed_executor_backend="mp", **kw)
    try:
        outs = llm.generate(prompt_token_ids=prompts,
                            sampling_params=SamplingParams(temperature=0, max_tokens=10, ignore_eos=True))
        assert [list(o.outputs[0].token_ids) for o in outs] == ref
        ex = llm.llm_engine.engine_core.engine_core.model_executor
        assert ex.collective_rpc("check_health") == [True, True]
    finally:
        llm.shutdown()


def test_multiproc_executor_reports_worker_death(tmp_path):
    import signal
    import threading
    model, _ = _save_tiny(str(tmp_path), "llama")
    llm = LLM(model=model, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128,
              max_num_seqs=4, max_num_batched_tokens=64, distributed_executor_backend="mp", tensor_parallel_size=2)
    ex = llm.llm_engine.engine_core.engine_core.model_executor
    fired = threading.Event()
    ex.register_failure_callback(fired.set)
    os.kill(ex.workers[1].proc.pid, signal.SIGKILL)
    assert fired.wait(20)
    assert ex.is_failed
    with pytest.raises(RuntimeError):
        ex.collective_rpc("check_health")
    llm.shutdown()


TKNP = {"enable_token_parallel": True, "token_parallel_size": 2}


@pytest.mark.parametrize("world,kw", [
    (2, dict(TKNP)),
    (2, dict(TKNP, additional_config={"tknp_policy": "blockwise"})),
    (3, {"enable_token_parallel": True, "token_parallel_size": 3}),
    (4, dict(TKNP, tensor_parallel_size=2)),
])
def test_token_parallel_matches_single(tmp_path, world, kw):
    """Token parallelism: requests (and their KV) are spread over the TKNP ranks, the root holds
    the weights; 5 requests with chunked prefill (64-token budget) and continuous batching make the
    per-rank rows non-contiguous, which the explicit row permutation must handle."""
    model, _ = _save_tiny(str(tmp_path), "llama")
    ref = _reference(model)
    got = _launch(world, model, str(tmp_path / "out"), kw)
    for r, toks in enumerate(got):
        assert toks == ref, f"rank {r} diverged with {kw}"


def test_expert_parallel_with_eplb_matches_single(tmp_path):
    """Mixtral with experts spread over two ranks (EP), two redundant slots and a rebalance every 3 steps:
    replicas and moved weights must not change a single token."""
    model, _ = _save_tiny(str(tmp_path), "mixtral")
    ref = _reference(model)
    kw = dict(tensor_parallel_size=2, enable_expert_parallel=True, enable_eplb=True, num_redundant_experts=2,
              eplb_window_size=8, eplb_step_interval=3)
    got = _launch(2, model, str(tmp_path / "out"), kw)
    for r, toks in enumerate(got):
        assert toks == ref, f"rank {r} diverged"
    plain = _launch(2, model, str(tmp_path / "out2"), dict(tensor_parallel_size=2, enable_expert_parallel=True))
    assert plain[0] == ref
