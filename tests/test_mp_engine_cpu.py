"""Engine core in a child process behind the socket transport: same tokens as the
in-process engine, utility RPC round trip, and fail-fast when the engine dies
(modelled on the reference's tests/v1/engine/test_engine_core_client.py and tests/v1/shutdown)."""
import os
import signal
import time

import pytest

from tests.test_e2e_cpu import _save_tiny
from vllm_distributed_amd import LLM, SamplingParams
from vllm_distributed_amd.engine.core_client import EngineDeadError


def _llm(path, monkeypatch, mp):
    monkeypatch.setenv("VLLM_ENABLE_V1_MULTIPROCESSING", "1" if mp else "0")
    return LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256,
               max_num_seqs=8, max_num_batched_tokens=64)


def test_multiprocess_engine_matches_inproc(tmp_path, monkeypatch):
    path, _ = _save_tiny(str(tmp_path), "llama")
    prompts = [[5, 17, 99, 3], [7] * 30, [1, 2, 3, 4, 5]]
    sp = SamplingParams(temperature=0, max_tokens=8, ignore_eos=True)
    a = _llm(path, monkeypatch, False)
    ref = [list(o.outputs[0].token_ids) for o in a.generate(prompt_token_ids=prompts, sampling_params=sp)]
    a.shutdown()
    b = _llm(path, monkeypatch, True)
    from vllm_distributed_amd.engine.mp_client import SyncMPClient
    assert isinstance(b.llm_engine.engine_core, SyncMPClient)
    got = [list(o.outputs[0].token_ids) for o in b.generate(prompt_token_ids=prompts, sampling_params=sp)]
    assert got == ref
    assert b.reset_prefix_cache() is True
    assert b.llm_engine.engine_core.call_utility("get_request_counts") in ((0, 0), [0, 0])
    b.shutdown()


def test_engine_death_is_reported(tmp_path, monkeypatch):
    path, _ = _save_tiny(str(tmp_path), "llama")
    llm = _llm(path, monkeypatch, True)
    pid = llm.llm_engine.engine_core.manager.procs[0].pid
    llm.llm_engine.add_request("x", {"prompt_token_ids": [1, 2, 3]}, SamplingParams(max_tokens=200, ignore_eos=True))
    os.kill(pid, signal.SIGKILL)
    with pytest.raises(EngineDeadError):
        for _ in range(500):
            llm.llm_engine.step()
    llm.shutdown()


def test_data_parallel_two_engines(tmp_path, monkeypatch):
    """DP=2 behind AsyncLLM: requests are spread over both engine cores, every request gets the
    tokens of the single-engine run, aborts reach the engine that owns the request."""
    import asyncio
    from vllm_distributed_amd.engine.arg_utils import AsyncEngineArgs
    from vllm_distributed_amd.engine.async_llm import AsyncLLM
    from vllm_distributed_amd.engine.mp_client import DPLBAsyncMPClient
    path, _ = _save_tiny(str(tmp_path), "llama")
    prompts = [[(3 * i + j) % 300 for j in range(6 + i)] for i in range(8)]
    sp = SamplingParams(temperature=0, max_tokens=6, ignore_eos=True)
    a = _llm(path, monkeypatch, False)
    ref = [list(o.outputs[0].token_ids) for o in a.generate(prompt_token_ids=prompts, sampling_params=sp)]
    a.shutdown()
    monkeypatch.setenv("VLLM_ENABLE_V1_MULTIPROCESSING", "1")
    eng = AsyncLLM.from_engine_args(AsyncEngineArgs(
        model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
        max_num_batched_tokens=64, data_parallel_size=2))
    client = eng.engine_core
    assert isinstance(client, DPLBAsyncMPClient) and len(client.conns) == 2
    used = set()

    async def one(i):
        toks = []
        async for out in eng.generate({"prompt_token_ids": prompts[i]}, sp, f"r{i}"):
            used.add(client.req_to_engine.get(f"r{i}", -1))
            toks = list(out.outputs[0].token_ids)
        return toks

    async def aborted():
        n = 0
        long = SamplingParams(temperature=0, max_tokens=200, ignore_eos=True)
        gen = eng.generate({"prompt_token_ids": [1, 2, 3]}, long, "gone")
        async for out in gen:
            n += 1
            if n == 2:
                break
        await gen.aclose()                # closing the stream early aborts the request
        return n

    async def go():
        res = await asyncio.gather(*(one(i) for i in range(len(prompts))))
        await aborted()
        await asyncio.sleep(0.3)
        counts = await client.call_all_async("get_request_counts")
        return res, counts

    try:
        got, counts = asyncio.run(asyncio.wait_for(go(), 180))
    finally:
        eng.shutdown()
    assert got == ref
    assert {0, 1} <= used
    assert all(tuple(c) == (0, 0) for c in counts), counts
