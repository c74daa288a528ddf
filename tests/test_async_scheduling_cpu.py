"""Run-ahead scheduling: the engine schedules and launches step N + 1 before the ids of step N are on the host.
Whatever the mode, every request must produce exactly the tokens of the plain engine (modelled on
tests/v1/core/test_async_scheduler.py and tests/v1/e2e/test_async_sched_and_preempt.py)."""
import numpy as np
import pytest

from tests.test_e2e_cpu import _save_tiny
from vllm_distributed_amd import LLM, SamplingParams

KW = dict(skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=192, max_num_batched_tokens=48)


def _workload():
    rng = np.random.default_rng(0)
    prompts = [rng.integers(3, 300, size=int(n)).tolist() for n in rng.integers(4, 90, size=14)]
    sps = []
    for i in range(len(prompts)):
        kw = dict(temperature=0.0, max_tokens=int(rng.integers(1, 40)), ignore_eos=(i % 3 != 0))
        if i % 4 == 1:
            kw.update(temperature=0.9, seed=100 + i, top_k=20)
        if i % 5 == 2:
            kw.update(stop_token_ids=[int(rng.integers(3, 300)) for _ in range(40)])      # stops early most of the time
        if i == 6:
            kw.update(presence_penalty=0.7, repetition_penalty=1.2)                       # collected before the next step
        if i == 9:
            kw.update(min_tokens=5, logprobs=2)
        if i == 11:
            kw.update(logprobs=3)
        sps.append(SamplingParams(**kw))
    return prompts, sps


def _run(path, **kw):
    prompts, sps = _workload()
    llm = LLM(model=path, **KW, **kw)
    try:
        core = llm.llm_engine.engine_core.engine_core
        sch = core.scheduler
        free0 = sch.kv_cache_manager.block_pool.get_num_free_blocks()
        outs = llm.generate(prompt_token_ids=prompts, sampling_params=sps)
        assert core._inflight is None and not core._ready
        assert not sch.running and not len(sch.waiting) and not sch.requests
        assert sch.kv_cache_manager.block_pool.get_num_free_blocks() == free0          # nothing leaked
        return [(list(o.outputs[0].token_ids), o.outputs[0].finish_reason,
                 None if o.outputs[0].logprobs is None else [sorted(d) for d in o.outputs[0].logprobs]) for o in outs]
    finally:
        llm.shutdown()


@pytest.mark.parametrize("kind", ["llama", "mamba2"])
def test_run_ahead_matches_the_plain_engine(tmp_path, kind):
    path, _ = _save_tiny(str(tmp_path), kind)
    ref = _run(path, max_num_seqs=6, async_scheduling=False)
    got = _run(path, max_num_seqs=6, async_scheduling=True)
    assert got == ref


def test_run_ahead_under_preemption(tmp_path):
    """A KV cache too small for the batch: requests are preempted and resumed while steps are in flight."""
    path, _ = _save_tiny(str(tmp_path), "llama")
    ref = _run(path, max_num_seqs=8, num_gpu_blocks_override=14, async_scheduling=False)
    got = _run(path, max_num_seqs=8, num_gpu_blocks_override=14, async_scheduling=True)
    assert got == ref


def test_run_ahead_really_runs_ahead(tmp_path):
    """The step in flight is collected after the next one has been launched, and only for plain requests."""
    path, _ = _save_tiny(str(tmp_path), "llama")
    llm = LLM(model=path, max_num_seqs=4, async_scheduling=True, **KW)
    try:
        core = llm.llm_engine.engine_core.engine_core
        runner = core.model_executor.worker.model_runner
        deferred = []
        orig = runner._defer_output

        def spy(*a, **k):
            out = orig(*a, **k)
            deferred.append(out)
            return out
        runner._defer_output = spy
        sp = SamplingParams(temperature=0, max_tokens=12, ignore_eos=True)
        llm.generate(prompt_token_ids=[[5, 6, 7, 8], [9, 10, 11]], sampling_params=sp)
        assert len(deferred) >= 10
        n0 = len(deferred)
        llm.generate(prompt_token_ids=[[5, 6, 7, 8]], sampling_params=SamplingParams(temperature=0, max_tokens=6,
                                                                                     frequency_penalty=0.5))
        assert len(deferred) == n0                       # penalties read the token history: no hand-off on the device
    finally:
        llm.shutdown()


def test_config_turns_run_ahead_off_where_it_cannot_work(tmp_path):
    from vllm_distributed_amd.engine.arg_utils import EngineArgs
    path, _ = _save_tiny(str(tmp_path), "llama")
    base = dict(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", async_scheduling=True)
    assert EngineArgs(**base).create_engine_config().scheduler_config.async_scheduling
    cfg = EngineArgs(**base, kv_transfer_config=dict(kv_connector="XgmiConnector", kv_role="kv_producer")
                     ).create_engine_config()
    assert not cfg.scheduler_config.async_scheduling
    cfg = EngineArgs(**base, tensor_parallel_size=2, distributed_executor_backend="mp").create_engine_config()
    assert not cfg.scheduler_config.async_scheduling
