"""fp8 quantisation plumbing on CPU: online quantisation of a 16-bit checkpoint, loading an
fp8-serialised checkpoint with per-projection scales (fused-layer requantisation), MoE scales."""
import json
import os

import torch
from safetensors.torch import load_file, save_file

from tests.test_e2e_cpu import _save_tiny
from vllm_distributed_amd import LLM, SamplingParams

KW = dict(skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
          max_num_batched_tokens=64)
PROMPTS = [[5, 17, 99, 3, 250, 11], [7] * 20, [1, 2, 3]]
SP = SamplingParams(temperature=0, max_tokens=6, ignore_eos=True, logprobs=1)


def _first_logprobs(llm):
    outs = llm.generate(prompt_token_ids=PROMPTS, sampling_params=SP)
    return [[list(lp.values())[0].logprob for lp in o.outputs[0].logprobs] for o in outs], \
        [list(o.outputs[0].token_ids) for o in outs]


def _to_fp8_checkpoint(path, static_act):
    """Rewrite a tiny HF checkpoint as per-tensor fp8 with one scale per original projection."""
    st = load_file(os.path.join(path, "model.safetensors"))
    out = {}
    for k, v in st.items():
        if k.endswith("proj.weight") or ".experts." in k:
            s = v.abs().max().float() / 448.0
            out[k] = (v.float() / s).to(torch.float8_e4m3fn)
            out[k + "_scale"] = s.reshape(())
            if static_act:
                out[k.replace(".weight", ".input_scale")] = torch.tensor(0.05)
        else:
            out[k] = v
    save_file(out, os.path.join(path, "model.safetensors"))
    cfg = json.load(open(os.path.join(path, "config.json")))
    cfg["quantization_config"] = {"quant_method": "fp8", "activation_scheme": "static" if static_act else "dynamic"}
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))


def test_online_fp8_close_to_fp32(tmp_path):
    path, _ = _save_tiny(str(tmp_path), "llama")
    a = LLM(model=path, **KW)
    ref_lp, ref_tok = _first_logprobs(a)
    a.shutdown()
    b = LLM(model=path, quantization="fp8", **KW)
    m = b.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
    assert m.model.layers[0].mlp.gate_up_proj.weight.dtype == torch.float8_e4m3fn
    assert m.lm_head.weight.dtype == torch.float32
    lp, tok = _first_logprobs(b)
    b.shutdown()
    # the first generated token's log-probability moves only by quantisation noise
    for x, y in zip(lp, ref_lp):
        assert abs(x[0] - y[0]) < 0.05


def test_serialized_fp8_checkpoint_matches_dequantised_reference(tmp_path):
    for kind in ("llama", "mixtral"):
        path, _ = _save_tiny(str(tmp_path), kind)
        _to_fp8_checkpoint(path, static_act=(kind == "llama"))
        q = LLM(model=path, **KW)
        lp, tok = _first_logprobs(q)
        q.shutdown()
        # reference: the same fp8 values dequantised back into an ordinary checkpoint
        st = load_file(os.path.join(path, "model.safetensors"))
        deq = {}
        for k, v in st.items():
            if k.endswith("_scale"):
                continue
            deq[k] = v.float() * st[k + "_scale"] if v.dtype == torch.float8_e4m3fn else v
        save_file(deq, os.path.join(path, "model.safetensors"))
        cfg = json.load(open(os.path.join(path, "config.json")))
        cfg.pop("quantization_config")
        json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
        r = LLM(model=path, **KW)
        ref_lp, ref_tok = _first_logprobs(r)
        r.shutdown()
        for x, y in zip(lp, ref_lp):
            assert abs(x[0] - y[0]) < 0.03, kind      # only the fused-shard requantisation differs


def test_fp8_kv_cache_close_to_16bit(tmp_path):
    path, _ = _save_tiny(str(tmp_path), "llama")
    a = LLM(model=path, **KW)
    ref_lp, ref_tok = _first_logprobs(a)
    a.shutdown()
    b = LLM(model=path, kv_cache_dtype="fp8", **KW)
    layer = next(iter(b.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.attn_layers.values()))
    assert layer.k_cache.dtype == torch.float8_e4m3fn
    lp, tok = _first_logprobs(b)
    b.shutdown()
    for x, y in zip(lp, ref_lp):
        assert abs(x[0] - y[0]) < 0.05
