"""Disaggregated prefill/decode on the GPU: two processes, the decode one maps the prefill one's
KV arena through HIP IPC and pulls blocks with the transfer kernel (the xGMI data plane; on a
one-GPU box both processes share the device, the code path is the same)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_prefill_decode_over_peer_memory(tmp_path):
    from tests.test_e2e_cpu import _save_tiny
    path, _ = _save_tiny(str(tmp_path), "llama")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "scripts", "disagg_worker.py"), role, path,
                               str(port), str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for role in ("producer", "consumer")]
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(out.decode(errors="replace"))
    for p, l in zip(procs, logs):
        assert p.returncode == 0, l[-3000:]
    c = json.load(open(tmp_path / "consumer.json"))
    assert c["peer_memory"], "the consumer fell back to the socket data plane"
    assert c["got"] == c["ref"]
    assert all(n >= 96 for n in c["cached"])
    assert json.load(open(tmp_path / "producer.json"))["freed"] > 0


def test_push_into_the_landing_pool_over_peer_memory(tmp_path):
    """PUT_ASYNC on the GPU: the producer maps the consumer's landing arena, copies blocks into it layer by layer
    on its push stream (24 device slots, the other 14 blocks go to the pinned host pool), and has its blocks
    back before the consumer decodes; the consumer moves slots into blocks and pulls nothing."""
    from tests.test_e2e_cpu import _save_tiny
    path, _ = _save_tiny(str(tmp_path), "llama")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "scripts", "disagg_worker.py"), role, path,
                               str(port), str(tmp_path), "push"], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for role in ("producer", "consumer")]
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(out.decode(errors="replace"))
    for p, l in zip(procs, logs):
        assert p.returncode == 0, l[-3000:]
    c = json.load(open(tmp_path / "consumer.json"))
    pr = json.load(open(tmp_path / "producer.json"))
    assert pr["landing_mapped"], "the producer pushed through the socket instead of the mapped landing arena"
    assert c["got"] == c["ref"]
    assert all(n >= 96 for n in c["cached"])
    assert c["pulled"] == 0, "the consumer fell back to pulling"
    assert c["pool_free_after"][0] == 24


def test_host_offload_connector_on_gpu(tmp_path):
    """Prompt blocks go to the pinned host pool after their prefill and come back with host -> device copies (the HBM
    prefix cache is off, so every repeat is served from the host pool); tokens equal the plain engine's, graphs on."""
    os.environ.setdefault("VLLM_ENABLE_V1_MULTIPROCESSING", "0")
    from tests.test_e2e_cpu import _save_tiny
    from vllm_distributed_amd import LLM, SamplingParams
    path, _ = _save_tiny(str(tmp_path), "llama")
    prompts = [[(7 * i + 3 * j) % 300 + 1 for j in range(40 + 23 * i)] for i in range(4)]
    sp = SamplingParams(temperature=0, max_tokens=8, ignore_eos=True)
    kw = dict(skip_tokenizer_init=True, dtype="bfloat16", max_model_len=256, max_num_seqs=8, gpu_memory_utilization=0.1,
              enable_prefix_caching=False)
    base = LLM(model=path, **kw)
    ref = [list(o.outputs[0].token_ids) for o in base.generate(prompt_token_ids=prompts, sampling_params=sp)]
    base.shutdown()
    llm = LLM(model=path, kv_transfer_config=dict(kv_connector="HostOffloadConnector", kv_role="kv_both",
                                                  kv_connector_extra_config={"host_blocks": 64}), **kw)
    first = llm.generate(prompt_token_ids=prompts, sampling_params=sp)
    again = llm.generate(prompt_token_ids=prompts, sampling_params=sp)
    llm.shutdown()
    assert [list(o.outputs[0].token_ids) for o in first] == ref
    assert [list(o.outputs[0].token_ids) for o in again] == ref
    assert all(o.num_cached_tokens == (len(p) - 1) // 16 * 16 for o, p in zip(again, prompts))
