"""Multi-LoRA: an adapter applied at run time must reproduce the model with the adapter merged into
its weights; requests with and without adapters share a batch; slots are recycled (modelled on
tests/lora/test_llama_tp.py, test_lora_manager.py)."""
import json
import os

import torch
from safetensors.torch import save_file

from tests.test_e2e_cpu import _save_tiny
from vllm_distributed_amd import LLM, SamplingParams
from vllm_distributed_amd.request import LoRARequest

PROMPTS = [[5, 17, 99, 3, 250, 11], [7] * 20, [1, 2, 3]]
SP = SamplingParams(temperature=0, max_tokens=8, ignore_eos=True)


def _make_adapter(tmp, name, hf, seed, targets=("q_proj", "v_proj", "gate_proj", "down_proj"), r=4, alpha=8.0):
    g = torch.Generator().manual_seed(seed)
    d = os.path.join(tmp, name)
    os.makedirs(d)
    w = {}
    merged = {k: v.clone() for k, v in hf.state_dict().items()}
    for k, v in hf.state_dict().items():
        leaf = k.split(".")[-2] if k.endswith(".weight") else ""
        if leaf in targets:
            out, inp = v.shape
            A = torch.randn(r, inp, generator=g) * 0.2
            B = torch.randn(out, r, generator=g) * 0.2
            mod = k[: -len(".weight")]
            w[f"base_model.model.{mod}.lora_A.weight"] = A
            w[f"base_model.model.{mod}.lora_B.weight"] = B
            merged[k] = v + (alpha / r) * (B @ A)
    save_file(w, os.path.join(d, "adapter_model.safetensors"))
    json.dump({"r": r, "lora_alpha": alpha, "target_modules": list(targets), "peft_type": "LORA"},
              open(os.path.join(d, "adapter_config.json"), "w"))
    return d, merged


def _hf_tokens(hf, state, prompts):
    import copy
    m = copy.deepcopy(hf)
    m.load_state_dict(state)
    out = []
    with torch.no_grad():
        for p in prompts:
            out.append(m.generate(torch.tensor([p]), max_new_tokens=8, do_sample=False, pad_token_id=0,
                                  eos_token_id=None)[0, len(p):].tolist())
    return out


def test_lora_matches_merged_weights_and_mixes_in_a_batch(tmp_path):
    path, hf = _save_tiny(str(tmp_path), "llama")
    a1, merged1 = _make_adapter(str(tmp_path), "ad1", hf, 1)
    a2, merged2 = _make_adapter(str(tmp_path), "ad2", hf, 2, targets=("k_proj", "o_proj", "up_proj"))
    base = _hf_tokens(hf, hf.state_dict(), PROMPTS)
    ref1 = _hf_tokens(hf, merged1, PROMPTS)
    ref2 = _hf_tokens(hf, merged2, PROMPTS)
    assert ref1 != base and ref2 != base
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
              max_num_batched_tokens=64, enable_lora=True, max_loras=1, max_cpu_loras=2, max_lora_rank=8)
    try:
        r1, r2 = LoRARequest("ad1", 1, a1), LoRARequest("ad2", 2, a2)
        got1 = llm.generate(prompt_token_ids=PROMPTS, sampling_params=SP, lora_request=r1)
        assert [list(o.outputs[0].token_ids) for o in got1] == ref1
        # one slot only: adapter 2 evicts adapter 1 from the device, both stay in the host cache
        got2 = llm.generate(prompt_token_ids=PROMPTS, sampling_params=SP, lora_request=r2)
        assert [list(o.outputs[0].token_ids) for o in got2] == ref2
        # mixed batch: base, adapter 1, base (max_loras=1 forces adapter batches to alternate)
        mixed = llm.generate(prompt_token_ids=PROMPTS, sampling_params=SP, lora_request=[None, r1, None])
        toks = [list(o.outputs[0].token_ids) for o in mixed]
        assert toks == [base[0], ref1[1], base[2]]
        both = llm.generate(prompt_token_ids=PROMPTS[:2], sampling_params=SP, lora_request=[r1, r2])
        assert [list(o.outputs[0].token_ids) for o in both] == [ref1[0], ref2[1]]
    finally:
        llm.shutdown()


def _make_vocab_adapter(tmp, name, hf, seed, r=4, alpha=8.0):
    """PEFT layout for an adapter on the token embedding (lora_embedding_A [r, vocab], lora_embedding_B
    [hidden, r]), the output head and one projection."""
    g = torch.Generator().manual_seed(seed)
    d = os.path.join(tmp, name)
    os.makedirs(d)
    sd = hf.state_dict()
    merged = {k: v.clone() for k, v in sd.items()}
    w = {}
    E = sd["model.embed_tokens.weight"]
    A = torch.randn(r, E.shape[0], generator=g) * 0.2
    B = torch.randn(E.shape[1], r, generator=g) * 0.2
    w["base_model.model.model.embed_tokens.lora_embedding_A"] = A
    w["base_model.model.model.embed_tokens.lora_embedding_B"] = B
    merged["model.embed_tokens.weight"] = E + (alpha / r) * (B @ A).t()
    for k in ("lm_head.weight", "model.layers.0.self_attn.q_proj.weight"):
        out, inp = sd[k].shape
        A = torch.randn(r, inp, generator=g) * 0.2
        B = torch.randn(out, r, generator=g) * 0.2
        mod = k[: -len(".weight")]
        w[f"base_model.model.{mod}.lora_A.weight"] = A
        w[f"base_model.model.{mod}.lora_B.weight"] = B
        merged[k] = sd[k] + (alpha / r) * (B @ A)
    save_file(w, os.path.join(d, "adapter_model.safetensors"))
    json.dump({"r": r, "lora_alpha": alpha, "target_modules": ["embed_tokens", "lm_head", "q_proj"], "peft_type": "LORA"},
              open(os.path.join(d, "adapter_config.json"), "w"))
    return d, merged


def test_lora_on_embedding_and_output_head(tmp_path):
    """Adapters on embed_tokens and lm_head: equal to the merged model, also when a batch mixes them with
    base requests and with an adapter that leaves both alone, and for prompt logprobs (head rows of a
    prefill chunk)."""
    path, hf = _save_tiny(str(tmp_path), "llama")
    av, merged_v = _make_vocab_adapter(str(tmp_path), "adv", hf, 3)
    a1, merged1 = _make_adapter(str(tmp_path), "ad1", hf, 1)
    base = _hf_tokens(hf, hf.state_dict(), PROMPTS)
    refv = _hf_tokens(hf, merged_v, PROMPTS)
    ref1 = _hf_tokens(hf, merged1, PROMPTS)
    assert refv != base
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
              max_num_batched_tokens=16, enable_lora=True, max_loras=2, max_cpu_loras=2, max_lora_rank=8)
    try:
        rv, r1 = LoRARequest("adv", 1, av), LoRARequest("ad1", 2, a1)
        got = llm.generate(prompt_token_ids=PROMPTS, sampling_params=SP, lora_request=rv)
        assert [list(o.outputs[0].token_ids) for o in got] == refv
        mixed = llm.generate(prompt_token_ids=PROMPTS, sampling_params=SP, lora_request=[r1, rv, None])
        assert [list(o.outputs[0].token_ids) for o in mixed] == [ref1[0], refv[1], base[2]]
        # prompt logprobs go through the head for every prompt position
        import copy
        m = copy.deepcopy(hf)
        m.load_state_dict(merged_v)
        with torch.no_grad():
            lp = torch.log_softmax(m(torch.tensor([PROMPTS[1]])).logits[0].float(), -1)
        out = llm.generate(prompt_token_ids=[PROMPTS[1]], lora_request=rv,
                           sampling_params=SamplingParams(temperature=0, max_tokens=1, prompt_logprobs=0))[0]
        for pos in range(1, len(PROMPTS[1])):
            tok = PROMPTS[1][pos]
            assert abs(out.prompt_logprobs[pos][tok].logprob - float(lp[pos - 1, tok])) < 1e-3
    finally:
        llm.shutdown()
