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


def test_data_parallel_attention_with_expert_parallel(tmp_path, monkeypatch):
    """DP=2 with the experts split over the two engine cores: every replica routes its own tokens to the expert
    owners through the all-to-all-v (torch.distributed form on the CPU) instead of all-gathering everybody's tokens;
    replicas step in lock step (an idle one runs dummy batches).  Tokens equal the single-engine run."""
    import asyncio
    from vllm_distributed_amd.engine.arg_utils import AsyncEngineArgs
    from vllm_distributed_amd.engine.async_llm import AsyncLLM
    path, _ = _save_tiny(str(tmp_path), "mixtral")
    prompts = [[(3 * i + j) % 300 for j in range(6 + 3 * i)] for i in range(7)]
    sp = SamplingParams(temperature=0, max_tokens=6, ignore_eos=True)
    a = _llm(path, monkeypatch, False)
    ref = [list(o.outputs[0].token_ids) for o in a.generate(prompt_token_ids=prompts, sampling_params=sp)]
    a.shutdown()
    monkeypatch.setenv("VLLM_ENABLE_V1_MULTIPROCESSING", "1")
    eng = AsyncLLM.from_engine_args(AsyncEngineArgs(
        model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
        max_num_batched_tokens=64, data_parallel_size=2, enable_expert_parallel=True))

    async def one(i):
        toks = []
        async for out in eng.generate({"prompt_token_ids": prompts[i]}, sp, f"r{i}"):
            toks = list(out.outputs[0].token_ids)
        return toks

    async def go():
        return await asyncio.gather(*(one(i) for i in range(len(prompts))))

    try:
        got = asyncio.run(asyncio.wait_for(go(), 240))
    finally:
        eng.shutdown()
    assert got == ref


def _async_args(path, **kw):
    from vllm_distributed_amd.engine.arg_utils import AsyncEngineArgs
    return AsyncEngineArgs(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256,
                           max_num_seqs=8, max_num_batched_tokens=64, **kw)


def test_two_front_ends_share_one_engine_core(tmp_path, monkeypatch):
    """Two AsyncLLM front ends (client_index 0 and 1) attached to ONE engine core: each gets exactly
    its own requests' outputs, and the core keeps running when one front end goes away."""
    import asyncio
    import threading
    from vllm_distributed_amd.engine.async_llm import AsyncLLM
    from vllm_distributed_amd.engine.headless import shutdown_procs
    from vllm_distributed_amd.engine.mp_client import launch_engine_cores
    from vllm_distributed_amd.engine.transport import make_ipc_address
    from vllm_distributed_amd.executor.base import Executor
    path, _ = _save_tiny(str(tmp_path), "llama")
    prompts = [[(5 * i + j) % 300 for j in range(5 + i)] for i in range(6)]
    sp = SamplingParams(temperature=0, max_tokens=5, ignore_eos=True)
    a = _llm(path, monkeypatch, False)
    ref = [list(o.outputs[0].token_ids) for o in a.generate(prompt_token_ids=prompts, sampling_params=sp)]
    a.shutdown()
    monkeypatch.setenv("VLLM_ENABLE_V1_MULTIPROCESSING", "1")
    cfg = _async_args(path).create_engine_config()
    addrs = [make_ipc_address("fe0"), make_ipc_address("fe1")]
    fes: list = [None, None]

    def build(i):
        fes[i] = AsyncLLM.from_engine_args(_async_args(path), client_index=i,
                                           attach={"address": addrs[i], "num_engines": 1, "coordinator_address": None})

    ts = [threading.Thread(target=build, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    procs = launch_engine_cores(cfg, Executor.get_class(cfg), False, addrs, 1)
    try:
        for t in ts:
            t.join(120)
        assert all(fes), "front ends did not attach"

        async def one(fe, i):
            toks = []
            async for out in fe.generate({"prompt_token_ids": prompts[i]}, sp, f"fe{fe.client_index}-r{i}"):
                toks = list(out.outputs[0].token_ids)
            return toks

        async def go():
            res = await asyncio.gather(*(one(fes[i % 2], i) for i in range(len(prompts))))
            # both front ends used the same request-id suffixes on purpose: routing is by client index
            fes[1].shutdown()
            await asyncio.sleep(0.3)
            again = await one(fes[0], 0)
            return res, again

        got, again = asyncio.run(asyncio.wait_for(go(), 180))
        assert got == ref and again == ref[0]
        assert procs[0].is_alive()
    finally:
        for fe in fes:
            if fe is not None:
                fe.shutdown()
        shutdown_procs(procs)


def test_headless_node_joins_front_end(tmp_path, monkeypatch):
    """DP=2 with one local engine core and one started by the headless entry point (as on a second
    node): both serve requests of the front end."""
    import argparse
    import asyncio
    import threading
    from vllm_distributed_amd.engine.async_llm import AsyncLLM
    from vllm_distributed_amd.engine.headless import run_headless, shutdown_procs
    from vllm_distributed_amd.entrypoints.openai.api_server import make_arg_parser
    from vllm_distributed_amd.parallel.utils import get_open_port
    path, _ = _save_tiny(str(tmp_path), "llama")
    prompts = [[(7 * i + j) % 300 for j in range(5 + i)] for i in range(6)]
    sp = SamplingParams(temperature=0, max_tokens=5, ignore_eos=True)
    a = _llm(path, monkeypatch, False)
    ref = [list(o.outputs[0].token_ids) for o in a.generate(prompt_token_ids=prompts, sampling_params=sp)]
    a.shutdown()
    monkeypatch.setenv("VLLM_ENABLE_V1_MULTIPROCESSING", "1")
    port = get_open_port()
    common = dict(data_parallel_size=2, data_parallel_size_local=1, data_parallel_address="127.0.0.1",
                  data_parallel_rpc_port=port)
    head: list = [None]

    def build():
        head[0] = AsyncLLM.from_engine_args(_async_args(path, **common))

    t = threading.Thread(target=build)
    t.start()
    cli = make_arg_parser().parse_args([
        "--model", path, "--skip-tokenizer-init", "--dtype", "float32", "--device", "cpu", "--max-model-len", "256",
        "--max-num-seqs", "8", "--max-num-batched-tokens", "64", "--data-parallel-size", "2",
        "--data-parallel-size-local", "1", "--data-parallel-address", "127.0.0.1", "--data-parallel-rpc-port",
        str(port), "--headless", "--data-parallel-start-rank", "1"])
    procs = run_headless(cli, block=False)
    try:
        t.join(180)
        eng = head[0]
        assert eng is not None, "front end did not come up"
        client = eng.engine_core
        assert len(client.conns) == 2 and len(client.manager.procs) == 1
        used = set()

        async def one(i):
            toks = []
            async for out in eng.generate({"prompt_token_ids": prompts[i]}, sp, f"r{i}"):
                used.add(client.req_to_engine.get(f"r{i}", -1))
                toks = list(out.outputs[0].token_ids)
            return toks

        async def go():
            return await asyncio.gather(*(one(i) for i in range(len(prompts))))

        got = asyncio.run(asyncio.wait_for(go(), 180))
        assert got == ref
        assert {0, 1} <= used
    finally:
        if head[0] is not None:
            head[0].shutdown()
        shutdown_procs(procs)


def test_multi_api_server_over_http(tmp_path, monkeypatch):
    """`--api-server-count 2`: two HTTP server processes share one port and one engine core."""
    import httpx
    from vllm_distributed_amd.entrypoints.openai.api_server import make_arg_parser
    from vllm_distributed_amd.entrypoints.openai.multi_api_server import MultiAPIServer
    from tests.test_openai_server_cpu import _add_tokenizer
    path, _ = _save_tiny(str(tmp_path), "llama")
    _add_tokenizer(path)
    prompts = [[(11 * i + j) % 300 for j in range(5 + i)] for i in range(8)]
    sp = SamplingParams(temperature=0, max_tokens=5, ignore_eos=True)
    a = _llm(path, monkeypatch, False)
    ref = [list(o.outputs[0].token_ids) for o in a.generate(prompt_token_ids=prompts, sampling_params=sp)]
    a.shutdown()
    monkeypatch.setenv("VLLM_ENABLE_V1_MULTIPROCESSING", "1")
    cli = make_arg_parser().parse_args([
        "--model", path, "--dtype", "float32", "--device", "cpu", "--max-model-len", "256",
        "--max-num-seqs", "8", "--max-num-batched-tokens", "64", "--api-server-count", "2", "--host", "127.0.0.1",
        "--port", "0", "--uvicorn-log-level", "warning", "--return-tokens-as-token-ids"])
    m = MultiAPIServer(cli)
    try:
        port = m.start(timeout=240)
        got = []
        for p in prompts:               # a fresh connection per request, so the kernel can pick either server
            r = httpx.post(f"http://127.0.0.1:{port}/v1/completions", timeout=120,
                           json={"model": path, "prompt": p, "max_tokens": 5, "temperature": 0, "ignore_eos": True,
                                 "logprobs": 0})
            assert r.status_code == 200, r.text
            toks = r.json()["choices"][0]["logprobs"]["tokens"]
            got.append([int(t.split(":")[1]) for t in toks])
        assert got == ref
        assert all(p.is_alive() for p in m.api_procs + m.engine_procs)
    finally:
        m.close()
