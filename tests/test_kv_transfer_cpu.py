"""KV connectors end to end on CPU: a prefill instance exports the KV of a prompt, a decode
instance imports it and must continue with exactly the tokens of a single-instance run
(modelled on tests/v1/kv_connector/{unit,nixl_integration})."""
import os
import socket

import time

import pytest

from tests.test_e2e_cpu import _save_tiny
from vllm_distributed_amd import LLM, SamplingParams

KW = dict(skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
          max_num_batched_tokens=64)
PROMPTS = [[(7 * i + 3 * j) % 300 for j in range(37 + 9 * i)] for i in range(3)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ref(path):
    llm = LLM(model=path, **KW)
    outs = llm.generate(prompt_token_ids=PROMPTS, sampling_params=SamplingParams(temperature=0, max_tokens=8,
                                                                                 ignore_eos=True))
    toks = [list(o.outputs[0].token_ids) for o in outs]
    llm.shutdown()
    return toks


def test_xgmi_connector_prefill_decode_split(tmp_path):
    path, _ = _save_tiny(str(tmp_path), "llama")
    ref = _ref(path)
    port = _free_port()
    kt = dict(kv_connector="XgmiConnector", kv_port=port)
    prefill = LLM(model=path, kv_transfer_config=dict(kt, kv_role="kv_producer"), **KW)
    decode = LLM(model=path, kv_transfer_config=dict(kt, kv_role="kv_consumer"), **KW)
    try:
        sp1 = SamplingParams(temperature=0, max_tokens=1, ignore_eos=True,
                             extra_args={"kv_transfer_params": {"do_remote_decode": True}})
        first = prefill.generate(prompt_token_ids=PROMPTS, sampling_params=sp1)
        params = [o.kv_transfer_params for o in first]
        assert all(p and p["do_remote_prefill"] and p["remote_block_ids"] for p in params)
        sched_p = prefill.llm_engine.engine_core.engine_core.scheduler
        free_before = sched_p.kv_cache_manager.block_pool.get_num_free_blocks()
        sps = [SamplingParams(temperature=0, max_tokens=8, ignore_eos=True, extra_args={"kv_transfer_params": p})
               for p in params]
        outs = decode.generate(prompt_token_ids=PROMPTS, sampling_params=sps)
        assert [list(o.outputs[0].token_ids) for o in outs] == ref
        # the decode side computed only the last prompt token of each request itself
        sched_d = decode.llm_engine.engine_core.engine_core.scheduler
        assert all(o.num_cached_tokens >= len(p) - 1 - 16 for o, p in zip(outs, PROMPTS))
        # DONE notifications let the prefill side release the exported blocks on its next step
        prefill.generate(prompt_token_ids=[[1, 2, 3]], sampling_params=SamplingParams(max_tokens=2))
        assert sched_p.kv_cache_manager.block_pool.get_num_free_blocks() > free_before
        del sched_d
    finally:
        prefill.shutdown()
        decode.shutdown()


def test_shared_storage_connector(tmp_path):
    path, _ = _save_tiny(str(tmp_path), "llama")
    ref = _ref(path)
    kt = dict(kv_connector="SharedStorageConnector",
              kv_connector_extra_config={"shared_storage_path": str(tmp_path / "kv")})
    sp = SamplingParams(temperature=0, max_tokens=8, ignore_eos=True)
    a = LLM(model=path, kv_transfer_config=dict(kt, kv_role="kv_both"), enable_prefix_caching=False, **KW)
    out_a = a.generate(prompt_token_ids=PROMPTS, sampling_params=sp)
    a.shutdown()
    assert [list(o.outputs[0].token_ids) for o in out_a] == ref
    b = LLM(model=path, kv_transfer_config=dict(kt, kv_role="kv_both"), enable_prefix_caching=False, **KW)
    out_b = b.generate(prompt_token_ids=PROMPTS, sampling_params=sp)
    b.shutdown()
    assert [list(o.outputs[0].token_ids) for o in out_b] == ref
    # second instance found the prompts on disk: block-aligned prefixes were not recomputed
    assert all(o.num_cached_tokens == (len(p) - 1) // 16 * 16 for o, p in zip(out_b, PROMPTS))


# ---------------------------------------------------------------------------------------------------------
# heterogeneous tensor parallelism, parameter validation, lifecycle while WAITING_FOR_REMOTE_KVS
# (modelled on the reference's tests/v1/kv_connector/unit/test_remote_prefill_lifecycle.py and
# test_nixl_connector.py, with fake runners in place of GPUs)
# ---------------------------------------------------------------------------------------------------------
import time
from types import SimpleNamespace

import torch

from vllm_distributed_amd.parallel.kv_transfer.xgmi import RecvSpec, XgmiConnector, XgmiMetadata, head_range, \
    head_segments
from vllm_distributed_amd.parallel.kv_transfer.base import KVConnectorRole


@pytest.mark.parametrize("heads,ptp,dtp", [(8, 1, 2), (8, 2, 8), (8, 4, 2), (8, 8, 1), (2, 2, 8), (2, 8, 2), (4, 4, 4),
                                           (1, 4, 2), (8, 2, 2)])
def test_head_segments_cover_every_consumer_head_exactly(heads, ptp, dtp):
    for r in range(dtp):
        lo, hi = head_range(r, dtp, heads)
        got = []
        for p, src_h, dst_h, n in head_segments(r, dtp, ptp, heads):
            plo, phi = head_range(p, ptp, heads)
            assert 0 <= src_h and src_h + n <= phi - plo and 0 <= p < ptp
            got += [(plo + src_h + i, dst_h + i) for i in range(n)]
        assert got == [(lo + i, i) for i in range(hi - lo)]


def _fake_side(tp, rank, role, port, heads_total=4, nb=12, bs=4, d=8, layers=("l0", "l1"), engine_id="E", key="k"):
    """A connector with a fake runner: per layer K [nb, h, bs, d] and V [nb, h, d, bs] as uint8 rows, plus a
    head-less (replicated) tensor per block."""
    h = max(1, heads_total // tp)
    per = h * bs * d * 2
    extra = 32
    chunk = torch.zeros(len(layers) * nb * (2 * per + extra), dtype=torch.uint8)
    parts, off = {}, 0
    for n in layers:
        parts[n] = [(0, off, nb * per, h), (0, off + nb * per, nb * per, h), (0, off + 2 * nb * per, nb * extra, 0)]
        off += nb * (2 * per + extra)
    runner = SimpleNamespace(is_gpu=False, device=torch.device("cpu"), kv_chunks=[chunk], kv_regions=[],
                             kv_layer_parts=parts, kv_cache_config=SimpleNamespace(num_blocks=nb))
    cfg = SimpleNamespace(
        kv_transfer_config=SimpleNamespace(engine_id=engine_id, kv_port=port, is_kv_producer=role == "p",
                                           kv_connector_extra_config={"auth_key": key, "side_channel_host": "127.0.0.1"}),
        cache_config=SimpleNamespace(block_size=bs), parallel_config=SimpleNamespace(tensor_parallel_size=tp),
        model_config=SimpleNamespace(get_total_num_kv_heads=lambda: heads_total))
    c = XgmiConnector.__new__(XgmiConnector)
    XgmiConnector.__init__(c, cfg, KVConnectorRole.WORKER)
    c.runner, c.tp_rank, c.total_kv_heads = runner, rank, heads_total
    if role == "p":
        from vllm_distributed_amd.engine.transport import Listener
        import threading
        c.listener = Listener(f"tcp://127.0.0.1:{port + rank}")
        threading.Thread(target=c._serve, daemon=True).start()
    return c, runner, (h, bs, d, per, extra)


def _wait_done(conn, rid, timeout=10.0):
    t0 = time.time()
    while time.time() - t0 < timeout:
        with conn._lock:
            if rid in conn._done_recving:
                return True
        time.sleep(0.01)
    return False


@pytest.mark.parametrize("ptp,dtp", [(1, 2), (2, 1), (2, 4), (2, 2)])
def test_pull_with_another_tensor_parallel_degree(ptp, dtp):
    """Every consumer rank must end up with exactly its heads of the exported blocks, whatever the
    producer's sharding; head-less tensors arrive whole."""
    port = _free_port()
    H, nb, bs, d = 4, 12, 4, 8
    full = {n: (torch.randint(0, 255, (nb, H, bs * d * 2), dtype=torch.uint8),
                torch.randint(0, 255, (nb, H, bs * d * 2), dtype=torch.uint8),
                torch.randint(0, 255, (nb, 32), dtype=torch.uint8)) for n in ("l0", "l1")}
    prods = []
    for r in range(ptp):
        c, run, (h, _, _, per, extra) = _fake_side(ptp, r, "p", port)
        lo, hi = head_range(r, ptp, H)
        for n, views in c._my_views().items():
            views[0][0].copy_(full[n][0][:, lo:hi].reshape(nb, -1))
            views[1][0].copy_(full[n][1][:, lo:hi].reshape(nb, -1))
            views[2][0].copy_(full[n][2])
        prods.append(c)
    remote = [7, 2, 9]
    local = [1, 5, 3]
    token = prods[0]._sign("n0", int(time.time()) + 600, remote)
    try:
        for r in range(dtp):
            c, run, _ = _fake_side(dtp, r, "d", port)
            spec = RecvSpec(local, remote, "127.0.0.1", port, "E", token, remote, 0)
            c._start_pull("req", spec)
            assert _wait_done(c, "req") and not c._failed_recving
            lo, hi = head_range(r, dtp, H)
            for n, views in c._my_views().items():
                for part in range(2):
                    want = full[n][part][remote][:, lo:hi].reshape(len(remote), -1)
                    assert torch.equal(views[part][0][local], want), (n, part, r)
                assert torch.equal(views[2][0][local], full[n][2][remote])
                untouched = [b for b in range(nb) if b not in local]
                assert int(views[0][0][untouched].sum()) == 0
            c.shutdown()
        # every producer rank saw as many DONE notices as consumers read from it: the export is released
        time.sleep(0.2)
        for p in prods:
            with p._lock:
                assert "req" in p._done_sending and not p._done_votes
    finally:
        for p in prods:
            p.shutdown()


def test_forged_transfer_parameters_are_refused():
    """Block ids the producer did not export (or out of range, or lists of unequal length) never reach a
    copy: the pull fails, the request is reported as failed and nothing is written."""
    port = _free_port()
    p, _, _ = _fake_side(1, 0, "p", port)
    for n, views in p._my_views().items():
        views[0][0].fill_(7)
    d, _, _ = _fake_side(1, 0, "d", port)
    good = [3, 4]
    token = p._sign("n1", int(time.time()) + 600, good)
    expired = p._sign("n2", int(time.time()) - 5, good)
    with p._lock:
        p._nonce_of["freed"] = "n3"
        p._revoke("freed")                      # the export "n3" was released: its token is dead
    revoked = p._sign("n3", int(time.time()) + 600, good)
    try:
        cases = [
            RecvSpec([1, 2], good, "127.0.0.1", port, "E", expired, good, 0),          # token past its expiry
            RecvSpec([1, 2], good, "127.0.0.1", port, "E", revoked, good, 0),          # blocks already released
            RecvSpec([1, 2], [5, 6], "127.0.0.1", port, "E", token, [5, 6], 0),        # other blocks, stolen token
            RecvSpec([1, 2], good, "127.0.0.1", port, "E", "0" * 64, good, 0),         # no valid token
            RecvSpec([1, 2], [3, 99], "127.0.0.1", port, "E", token, [3, 99], 0),      # remote id out of range
            RecvSpec([1, 99], good, "127.0.0.1", port, "E", token, good, 0),           # local id out of range
            RecvSpec([1], good, "127.0.0.1", port, "E", token, good, 0),               # lengths differ
        ]
        for i, spec in enumerate(cases):
            d._meta = XgmiMetadata({f"bad{i}": spec}, {})
            d.start_load_kv()
            assert _wait_done(d, f"bad{i}"), i
        fin = d.get_finished(set())
        assert fin[1] == {f"bad{i}" for i in range(len(cases))}
        assert d.failed_recving() == fin[1]
        for n, views in d._my_views().items():
            assert int(views[0][0].sum()) == 0
        # and the genuine request still works
        d._meta = XgmiMetadata({"ok": RecvSpec([1, 2], good, "127.0.0.1", port, "E", token, good, 0)}, {})
        d.start_load_kv()
        assert _wait_done(d, "ok")
        assert d.get_finished(set())[1] == {"ok"} and not d.failed_recving()
        assert int(d._my_views()["l0"][0][0][[1, 2]].min()) == 7
    finally:
        p.shutdown()
        d.shutdown()


def test_layout_mismatch_is_refused_at_handshake():
    port = _free_port()
    p, _, _ = _fake_side(1, 0, "p", port, d=8)
    d, _, _ = _fake_side(1, 0, "d", port, d=16)          # another head size
    try:
        with pytest.raises(ValueError, match="differs in dtype, head size"):
            d._peer("127.0.0.1", port, 0)
    finally:
        p.shutdown()
        d.shutdown()


@pytest.mark.parametrize("mode", ["pull", "push"])
def test_disagg_benchmark_harness_on_cpu(tmp_path, mode):
    """benchmarks/disagg.py end to end (1 prefill + 1 decode engine process, socket data plane): every
    request completes through the router path and the report carries the metrics of the study."""
    import json
    import subprocess
    import sys
    from tests.test_e2e_cpu import _save_tiny
    path, _ = _save_tiny(str(tmp_path), "llama")
    out = tmp_path / "res.json"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "vllm_distributed_amd.benchmarks.disagg", "--model", path,
                        "--load-format", "auto", "--dtype", "float32", "--device", "cpu", "--prefill", "1",
                        "--decode", "1", "--input-len", "64", "--output-len", "6", "--num-prompts", "5",
                        "--vocab-size", "300", "--max-num-seqs", "8", "--max-num-batched-tokens", "256",
                        "--request-rate", "50", "--output-json", str(out)] + (["--kv-push"] if mode == "push" else []),
                       cwd=root, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.load(open(out))["disagg"]
    assert res["completed"] == 5 and res["num_errors"] == 0
    assert res["n_gpus"] == 2 and res["output_tok_s"] > 0
    assert res["ttft_ms"]["p99"] >= res["ttft_ms"]["p50"] > 0 and res["itl_ms"]["p50"] > 0


def _engine_parts(llm):
    core = llm.llm_engine.engine_core.engine_core
    return core.scheduler, core.model_executor.worker.model_runner.kv_connector


@pytest.mark.parametrize("send_type", ["PUT", "PUT_ASYNC"])
def test_push_connector_lands_blocks_before_the_decode_request_arrives(tmp_path, send_type):
    """Push mode: the prefill instance writes the blocks into the decode instance's landing pool step by step
    (chunked prefill: 64-token steps), frees its own blocks without waiting for the consumer, and the decode
    instance moves the slots into the request's blocks when the request shows up."""
    path, _ = _save_tiny(str(tmp_path), "llama")
    ref = _ref(path)
    pport, dport = _free_port(), _free_port()
    xc = {"send_type": send_type, "mem_pool_size_gb": 0.01}
    prefill = LLM(model=path, kv_transfer_config=dict(kv_connector="XgmiPushConnector", kv_port=pport,
                                                      kv_role="kv_producer", kv_connector_extra_config=dict(xc)), **KW)
    decode = LLM(model=path, kv_transfer_config=dict(kv_connector="P2pNcclConnector", kv_port=dport,
                                                     kv_role="kv_consumer", kv_connector_extra_config=dict(xc)), **KW)
    try:
        sched_p, _ = _engine_parts(prefill)
        _, conn_d = _engine_parts(decode)
        assert type(conn_d).__name__ == "XgmiPushConnector"
        free0 = sched_p.kv_cache_manager.block_pool.get_num_free_blocks()
        host_free0 = conn_d.pool.num_free()[1]
        sp1 = SamplingParams(temperature=0, max_tokens=1, ignore_eos=True, extra_args={"kv_transfer_params": {
            "do_remote_decode": True, "push_host": "127.0.0.1", "push_port": dport}})
        first = prefill.generate(prompt_token_ids=PROMPTS, sampling_params=sp1)
        params = [o.kv_transfer_params for o in first]
        assert all(p["pushed"] and p["transfer_id"] and p["remote_block_ids"] for p in params)
        # every transfer is complete on the consumer although no decode request exists yet
        import time
        t0 = time.time()
        while time.time() - t0 < 5 and not all(conn_d._rx.get(p["transfer_id"], {}).get("complete") for p in params):
            time.sleep(0.01)
        assert all(conn_d._rx[p["transfer_id"]]["complete"] for p in params)
        assert conn_d.pool.num_free()[1] == host_free0 - sum(len(p["remote_block_ids"]) for p in params)
        # one more producer step: the blocks are free again, nobody sent DONE
        prefill.generate(prompt_token_ids=[[1, 2, 3]], sampling_params=SamplingParams(max_tokens=2))
        assert sched_p.kv_cache_manager.block_pool.get_num_free_blocks() == free0
        sps = [SamplingParams(temperature=0, max_tokens=8, ignore_eos=True, extra_args={"kv_transfer_params": p})
               for p in params]
        outs = decode.generate(prompt_token_ids=PROMPTS, sampling_params=sps)
        assert [list(o.outputs[0].token_ids) for o in outs] == ref
        assert all(o.num_cached_tokens >= len(p) - 1 - 16 for o, p in zip(outs, PROMPTS))
        decode.generate(prompt_token_ids=[[1, 2, 3]], sampling_params=SamplingParams(max_tokens=2))
        assert conn_d.pool.num_free()[1] == host_free0 and not conn_d._rx
    finally:
        prefill.shutdown()
        decode.shutdown()


def test_push_falls_back_to_pull_when_the_landing_pool_is_full(tmp_path):
    path, _ = _save_tiny(str(tmp_path), "llama")
    ref = _ref(path)
    pport, dport = _free_port(), _free_port()
    prefill = LLM(model=path, kv_transfer_config=dict(kv_connector="XgmiPushConnector", kv_port=pport,
                                                      kv_role="kv_producer"), **KW)
    decode = LLM(model=path, kv_transfer_config=dict(kv_connector="XgmiPushConnector", kv_port=dport, kv_role="kv_consumer",
                                                     kv_connector_extra_config={"mem_pool_size_gb": 0}), **KW)
    try:
        _, conn_d = _engine_parts(decode)
        assert conn_d.pool.num_free() == (0, 0)
        sp1 = SamplingParams(temperature=0, max_tokens=1, ignore_eos=True, extra_args={"kv_transfer_params": {
            "do_remote_decode": True, "push_host": "127.0.0.1", "push_port": dport}})
        params = [o.kv_transfer_params for o in prefill.generate(prompt_token_ids=PROMPTS, sampling_params=sp1)]
        sps = [SamplingParams(temperature=0, max_tokens=8, ignore_eos=True, extra_args={"kv_transfer_params": p})
               for p in params]
        outs = decode.generate(prompt_token_ids=PROMPTS, sampling_params=sps)
        assert [list(o.outputs[0].token_ids) for o in outs] == ref
        assert all(o.num_cached_tokens >= len(p) - 1 - 16 for o, p in zip(outs, PROMPTS))      # pulled, not recomputed
    finally:
        prefill.shutdown()
        decode.shutdown()


def test_push_metadata_survives_the_wire():
    from vllm_distributed_amd.engine import serial
    from vllm_distributed_amd.parallel.kv_transfer.push import LandSpec, PushMetadata, PushSpec
    from vllm_distributed_amd.parallel.kv_transfer.xgmi import RecvSpec
    m = PushMetadata(pushes={"a": PushSpec("t1", "h", 5, [3, 4], 1, 2, True)},
                     lands={"b": LandSpec("t2", RecvSpec([1], [9], "h", 7, "E", "tok", [8, 9], 1))}, aborted=["t3"])
    back = serial.decode(serial.encode(m))
    assert back == m


# ---------------------------------------------------------------------------------------------------------
# KV offloading to host memory
# ---------------------------------------------------------------------------------------------------------
def test_host_offload_connector_serves_evicted_prefixes(tmp_path):
    """Prompts are copied to the host pool when their prefill completes.  With the HBM prefix cache off every later
    request is served from the host pool: same tokens, block-aligned prefixes (also of LONGER prompts that merely
    share a prefix) counted as cached."""
    path, _ = _save_tiny(str(tmp_path), "llama")
    ref = _ref(path)
    kt = dict(kv_connector="HostOffloadConnector", kv_role="kv_both", kv_connector_extra_config={"host_blocks": 64})
    sp = SamplingParams(temperature=0, max_tokens=8, ignore_eos=True)
    llm = LLM(model=path, kv_transfer_config=kt, enable_prefix_caching=False, **KW)
    first = llm.generate(prompt_token_ids=PROMPTS, sampling_params=sp)
    assert [list(o.outputs[0].token_ids) for o in first] == ref
    assert all(o.num_cached_tokens == 0 for o in first)
    again = llm.generate(prompt_token_ids=PROMPTS, sampling_params=sp)
    assert [list(o.outputs[0].token_ids) for o in again] == ref
    assert all(o.num_cached_tokens == (len(p) - 1) // 16 * 16 for o, p in zip(again, PROMPTS))
    # a longer prompt that shares a prefix: the shared full blocks come from the host pool
    longest = max(PROMPTS, key=len)
    ext = longest + [11, 12, 13, 14, 15, 16, 17, 18, 19, 20]
    base = LLM(model=path, enable_prefix_caching=False, **KW)
    want = list(base.generate(prompt_token_ids=[ext], sampling_params=sp)[0].outputs[0].token_ids)
    base.shutdown()
    got = llm.generate(prompt_token_ids=[ext], sampling_params=sp)[0]
    assert list(got.outputs[0].token_ids) == want
    assert got.num_cached_tokens == (len(longest) - 1) // 16 * 16 > 0
    llm.shutdown()


def test_host_offload_index_evicts_least_recently_used():
    from vllm_distributed_amd.parallel.kv_transfer.host_offload import HostOffloadConnector
    cfg = SimpleNamespace(kv_transfer_config=SimpleNamespace(kv_connector_extra_config={"host_blocks": 3}),
                          cache_config=SimpleNamespace(block_size=4, prefix_caching_hash_algo="builtin"))
    c = HostOffloadConnector(cfg, KVConnectorRole.SCHEDULER)

    def step(new=(), cached=(), finished=()):
        so = SimpleNamespace(
            scheduled_new_reqs=[SimpleNamespace(req_id=r, prompt_token_ids=t, block_ids=b, num_computed_tokens=0)
                                for r, t, b in new],
            scheduled_cached_reqs=SimpleNamespace(req_ids=[], resumed_from_preemption=[], new_block_ids=[],
                                                  num_computed_tokens=[]),
            num_scheduled_tokens={r: len(t) for r, t, _ in new}, finished_req_ids=set(finished))
        return c.build_connector_meta(so)
    a, b = list(range(100, 109)), list(range(200, 209))        # two full blocks each (the ninth token is computed)
    m = step(new=[("a", a, [7, 8, 9])])
    assert [blk for _, blk in m.stores] == [7, 8] and len(c.index) == 2
    m = step(new=[("b", b, [3, 4, 5])])
    assert len(m.stores) == 2 and len(c.index) == 3 and c.stats["evicted_blocks"] == 1
    # a's first block was the oldest: gone, so nothing of a matches any more (the chain starts at block 0)
    req = SimpleNamespace(request_id="a2", prompt_token_ids=a, mm_inputs=None)
    assert c.get_num_new_matched_tokens(req, 0) == (0, False)
    req = SimpleNamespace(request_id="b2", prompt_token_ids=b, mm_inputs=None)
    assert c.get_num_new_matched_tokens(req, 0) == (8, False)
    c.update_state_after_alloc(req, [11, 12, 13], 8)
    m = step()
    assert [blk for _, blk in m.loads] == [11, 12] and not m.stores


def test_host_offload_keys_blocks_by_adapter_and_salt():
    """Equal prompts computed under different LoRA adapters, or carrying different cache salts, must not share host
    blocks: the index hashes the adapter id and the salt with the tokens, like the device prefix cache does."""
    from vllm_distributed_amd.parallel.kv_transfer.host_offload import HostOffloadConnector
    cfg = SimpleNamespace(kv_transfer_config=SimpleNamespace(kv_connector_extra_config={"host_blocks": 16}),
                          cache_config=SimpleNamespace(block_size=4, prefix_caching_hash_algo="builtin"))
    c = HostOffloadConnector(cfg, KVConnectorRole.SCHEDULER)
    toks = list(range(100, 109))

    def req(rid, lora=None, salt=None):
        return SimpleNamespace(request_id=rid, prompt_token_ids=toks, mm_inputs=None, cache_salt=salt,
                               lora_request=SimpleNamespace(lora_int_id=lora) if lora else None)

    def prefill(r, blocks):
        assert c.get_num_new_matched_tokens(r, 0)[0] == 0
        so = SimpleNamespace(
            scheduled_new_reqs=[SimpleNamespace(req_id=r.request_id, prompt_token_ids=toks, block_ids=blocks,
                                                num_computed_tokens=0)],
            scheduled_cached_reqs=SimpleNamespace(req_ids=[], resumed_from_preemption=[], new_block_ids=[],
                                                  num_computed_tokens=[]),
            num_scheduled_tokens={r.request_id: len(toks)}, finished_req_ids=set())
        return c.build_connector_meta(so)
    assert len(prefill(req("plain"), [1, 2, 3]).stores) == 2
    assert len(prefill(req("lora1", lora=1), [4, 5, 6]).stores) == 2          # not served by the plain blocks
    assert len(prefill(req("lora2", lora=2), [7, 8, 9]).stores) == 2
    assert len(prefill(req("salted", salt="s1"), [10, 11, 12]).stores) == 2
    assert len(c.index) == 8
    # the same identity does hit
    assert c.get_num_new_matched_tokens(req("lora1-again", lora=1), 0) == (8, False)
    assert c.get_num_new_matched_tokens(req("salted-again", salt="s1"), 0) == (8, False)
    assert c.get_num_new_matched_tokens(req("other-salt", salt="s2"), 0) == (0, False)
    assert c.get_num_new_matched_tokens(req("plain-again"), 0) == (8, False)


def test_push_begin_needs_the_deployment_key_and_owns_its_transfer():
    """PUT_BEGIN is refused without a MAC made with the deployment's auth_key; PUT_DATA / PUT_END / PUT_ABORT for a
    transfer are honoured only on the connection that began it, and a second connection cannot take over its id."""
    import hashlib
    import hmac as _hmac

    from vllm_distributed_amd.parallel.kv_transfer.push import XgmiPushConnector

    class Conn:
        def __init__(self, msgs):
            self.msgs, self.sent = list(msgs), []

        def recv(self):
            if not self.msgs:
                raise EOFError
            return self.msgs.pop(0)

        def send(self, x):
            self.sent.append(x)

    class Pool:
        def __init__(self):
            self.freed = []

        def alloc(self, n):
            return [("host", i) for i in range(n)]

        def free(self, slots):
            self.freed.append(slots)
    import threading
    c = XgmiPushConnector.__new__(XgmiPushConnector)
    c._push_key, c._lock, c._rx, c.pool = b"secret", threading.Lock(), {}, Pool()
    mac = _hmac.new(b"secret", b"put|t1", hashlib.sha256).hexdigest()
    bad = Conn([("PUT_BEGIN", "t1", 2, "0" * 64), ("PUT_BEGIN", "t1", 2)])
    c._serve_conn(bad)
    assert bad.sent == [None, None] and not c._rx
    owner = Conn([("PUT_BEGIN", "t1", 2, mac)])
    c._serve_conn(owner)
    assert owner.sent == [[("host", 0), ("host", 1)]] and "t1" in c._rx
    thief = Conn([("PUT_BEGIN", "t1", 2, mac), ("PUT_END", "t1"), ("PUT_ABORT", "t1"), ("PUT_DATA", "t1", 0, {})])
    c._serve_conn(thief)
    assert thief.sent == [None, False]                    # begin refused (id in use), data refused
    assert c._rx["t1"]["complete"] is False and not c.pool.freed
