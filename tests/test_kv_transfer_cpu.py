"""KV connectors end to end on CPU: a prefill instance exports the KV of a prompt, a decode
instance imports it and must continue with exactly the tokens of a single-instance run
(modelled on tests/v1/kv_connector/{unit,nixl_integration})."""
import socket

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
