"""Speculative decoding: greedy outputs must not depend on the proposer (n-gram, EAGLE, Medusa with
random draft weights), drafts must actually be accepted for repetitive text, and the rejection
sampler must preserve the target distribution (modelled on tests/v1/spec_decode, tests/v1/sample)."""
import json
import os

import numpy as np
import torch
from safetensors.torch import save_file

from tests.test_e2e_cpu import _save_tiny
from vllm_distributed_amd import LLM, SamplingParams

KW = dict(skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
          max_num_batched_tokens=64)
PROMPTS = [[5, 6, 7, 8] * 6, [(3 * j) % 200 for j in range(30)], [9, 9, 9, 9, 9, 9]]
SP = SamplingParams(temperature=0, max_tokens=24, ignore_eos=True)


def _gen(path, **kw):
    llm = LLM(model=path, **KW, **kw)
    outs = llm.generate(prompt_token_ids=PROMPTS, sampling_params=SP)
    toks = [list(o.outputs[0].token_ids) for o in outs]
    stats = llm.llm_engine.engine_core.engine_core.scheduler.make_stats()
    llm.shutdown()
    return toks, stats


def test_native_ngram_matcher():
    import ctypes
    from vllm_distributed_amd.runtime.shm_ring import host_library
    lib = host_library()
    lib.vda_ngram_propose.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p]
    toks = np.array([1, 2, 3, 4, 5, 9, 2, 3, 4, 7, 7, 2, 3, 4], dtype=np.int32)
    out = np.zeros(3, dtype=np.int32)
    n = lib.vda_ngram_propose(toks.ctypes.data, len(toks), 2, 3, 3, out.ctypes.data)
    assert n == 3 and out.tolist() == [7, 7, 2]          # latest earlier "2 3 4" is followed by 7 7 2
    n = lib.vda_ngram_propose(toks.ctypes.data, 5, 2, 3, 3, out.ctypes.data)
    assert n == 0


def test_ngram_same_tokens_and_accepts(tmp_path):
    path, _ = _save_tiny(str(tmp_path), "llama")
    ref, _ = _gen(path)
    got, stats = _gen(path, speculative_config={"method": "ngram", "num_speculative_tokens": 3,
                                                "prompt_lookup_max": 3, "prompt_lookup_min": 1})
    assert got == ref


def _save_draft(tmp, kind, target_path):
    cfg = json.load(open(os.path.join(target_path, "config.json")))
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    d = os.path.join(tmp, kind)
    os.makedirs(d)
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g) * 0.05      # noqa: E731
    if kind == "eagle":
        cfg.update(architectures=["EagleLlamaForCausalLM"], num_hidden_layers=1)
        hd, nh, nkv = cfg["head_dim"], cfg["num_attention_heads"], cfg["num_key_value_heads"]
        w = {"fc.weight": r(H, 2 * H),
             "layers.0.self_attn.q_proj.weight": r(nh * hd, H), "layers.0.self_attn.k_proj.weight": r(nkv * hd, H),
             "layers.0.self_attn.v_proj.weight": r(nkv * hd, H), "layers.0.self_attn.o_proj.weight": r(H, nh * hd),
             "layers.0.mlp.gate_proj.weight": r(I, H), "layers.0.mlp.up_proj.weight": r(I, H),
             "layers.0.mlp.down_proj.weight": r(H, I), "layers.0.post_attention_layernorm.weight": torch.ones(H)}
    elif kind == "eagle_minicpm":
        cfg.update(architectures=["EagleMiniCPMForCausalLM"], num_hidden_layers=1, scale_emb=4.0, scale_depth=1.4,
                   mup_denominator=8, dim_model_base=H // 2, tie_word_embeddings=False)
        hd, nh, nkv = cfg["head_dim"], cfg["num_attention_heads"], cfg["num_key_value_heads"]
        p_ = "model.eagle_layers.0."
        w = {"model.fc.weight": r(H, 2 * H), "model.input_norm1.weight": torch.ones(H),
             "model.input_norm2.weight": torch.ones(H), "model.embed_tokens.weight": r(V, H), "lm_head.weight": r(V, H),
             p_ + "self_attn.q_proj.weight": r(nh * hd, H), p_ + "self_attn.k_proj.weight": r(nkv * hd, H),
             p_ + "self_attn.v_proj.weight": r(nkv * hd, H), p_ + "self_attn.o_proj.weight": r(H, nh * hd),
             p_ + "mlp.gate_proj.weight": r(I, H), p_ + "mlp.up_proj.weight": r(I, H),
             p_ + "mlp.down_proj.weight": r(H, I), p_ + "input_layernorm.weight": torch.ones(H),
             p_ + "post_attention_layernorm.weight": torch.ones(H)}
    else:
        cfg = {"architectures": ["MedusaModel"], "hidden_size": H, "vocab_size": V, "num_heads": 3,
               "num_hidden_layers": 1, "model_type": "medusa"}
        w = {}
        for h in range(3):
            w[f"{h}.0.linear.weight"], w[f"{h}.0.linear.bias"] = r(H, H), r(H)
            w[f"{h}.1.weight"] = r(V, H)
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    save_file(w, os.path.join(d, "model.safetensors"))
    return d


def test_eagle_and_medusa_same_tokens(tmp_path):
    path, _ = _save_tiny(str(tmp_path), "llama")
    ref, _ = _gen(path)
    for kind in ("eagle", "medusa", "eagle_minicpm"):
        draft = _save_draft(str(tmp_path), kind, path)
        got, n = _proposals(path, speculative_config={"method": kind.split("_")[0], "model": draft,
                                                      "num_speculative_tokens": 3})
        assert got == ref and n > 0, kind        # the target's own tokens, and drafts were verified


def test_rejection_sampler_greedy_and_distribution():
    from vllm_distributed_amd.layers.rejection_sampler import RejectionSampler
    from vllm_distributed_amd.layers.sampler import Sampler, SamplingMetadata
    rs, sampler = RejectionSampler(), Sampler()
    V = 6
    lg = torch.full((5, V), -10.0)
    for r, t in enumerate([2, 3, 1, 4, 0]):
        lg[r, t] = 10.0
    md = SamplingMetadata(None, True, False)
    # request 0: drafts [2, 5] -> first accepted, second rejected (target 3); request 1: draft [4] accepted + bonus 0
    assert rs(lg, [[2, 5], [4]], md, sampler) == [[2, 3], [4, 0]]
    # sampled: the emitted first token must follow the target distribution whatever the draft is
    torch.manual_seed(0)
    p = torch.tensor([0.5, 0.3, 0.1, 0.05, 0.03, 0.02])
    N = 4000
    logits = p.log().repeat(2 * N, 1)
    md = SamplingMetadata(torch.ones(N), False, True)
    outs = rs(logits, [[1]] * N, md, sampler)
    first = torch.tensor([o[0] for o in outs])
    freq = torch.bincount(first, minlength=V).float() / N
    assert torch.allclose(freq, p, atol=0.03), freq
    acc = sum(len(o) == 2 for o in outs) / N
    assert abs(acc - 0.3) < 0.03          # P(accept draft token 1) = p(1) for a one-hot draft


def _proposals(path, **kw):
    """(tokens, number of drafted tokens the engine verified)."""
    llm = LLM(model=path, **KW, **kw)
    n = [0]
    sch = llm.llm_engine.engine_core.engine_core.scheduler
    orig = sch.update_from_output

    def spy(so, mro):
        n[0] += sum(len(v) for v in so.scheduled_spec_decode_tokens.values())
        return orig(so, mro)
    sch.update_from_output = spy
    outs = llm.generate(prompt_token_ids=PROMPTS, sampling_params=SP)
    llm.shutdown()
    return [list(o.outputs[0].token_ids) for o in outs], n[0]


def test_mlp_speculator_same_tokens(tmp_path):
    path, _ = _save_tiny(str(tmp_path), "llama")
    ref, _ = _gen(path)
    cfg = json.load(open(os.path.join(path, "config.json")))
    H, V, inner, k = cfg["hidden_size"], cfg["vocab_size"], 64, 3
    d = os.path.join(str(tmp_path), "speculator")
    os.makedirs(d)
    g = torch.Generator().manual_seed(1)
    w = {}
    for i in range(k):
        w[f"speculator.emb.{i}.weight"] = torch.randn(V, inner, generator=g) * 0.05
        w[f"speculator.proj.{i}.weight"] = torch.randn(inner, H if i == 0 else inner, generator=g) * 0.05
        w[f"speculator.head.{i}.weight"] = torch.randn(V, inner, generator=g) * 0.05
        w[f"speculator.ln.{i}.weight"], w[f"speculator.ln.{i}.bias"] = torch.ones(inner), torch.zeros(inner)
    json.dump({"architectures": ["MLPSpeculatorPreTrainedModel"], "model_type": "mlp_speculator", "vocab_size": V,
               "emb_dim": H, "inner_dim": inner, "n_predict": k, "scale_input": True, "tie_weights": False,
               "hidden_size": H}, open(os.path.join(d, "config.json"), "w"))
    save_file(w, os.path.join(d, "model.safetensors"))
    got, n = _proposals(path, speculative_config={"method": "mlp_speculator", "model": d, "num_speculative_tokens": k})
    assert got == ref and n > 0


def test_mimo_multi_token_prediction_layer_drafts(tmp_path):
    """A MiMo checkpoint (Qwen2 weights + ``model.mtp_layers.0``) drafts with its own prediction layer."""
    from safetensors.torch import load_file
    path, _ = _save_tiny(str(tmp_path), "qwen2")
    cfg = json.load(open(os.path.join(path, "config.json")))
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    hd = H // nh
    w = load_file(os.path.join(path, "model.safetensors"))
    g = torch.Generator().manual_seed(2)
    r = lambda *s: torch.randn(*s, generator=g) * 0.05      # noqa: E731
    p = "model.mtp_layers.0."
    w.update({p + "token_layernorm.weight": torch.ones(H), p + "hidden_layernorm.weight": torch.ones(H),
              p + "input_proj.weight": r(H, 2 * H), p + "final_layernorm.weight": torch.ones(H),
              p + "input_layernorm.weight": torch.ones(H), p + "post_attention_layernorm.weight": torch.ones(H),
              p + "self_attn.q_proj.weight": r(nh * hd, H), p + "self_attn.q_proj.bias": r(nh * hd),
              p + "self_attn.k_proj.weight": r(nkv * hd, H), p + "self_attn.k_proj.bias": r(nkv * hd),
              p + "self_attn.v_proj.weight": r(nkv * hd, H), p + "self_attn.v_proj.bias": r(nkv * hd),
              p + "self_attn.o_proj.weight": r(H, nh * hd), p + "mlp.gate_proj.weight": r(I, H),
              p + "mlp.up_proj.weight": r(I, H), p + "mlp.down_proj.weight": r(H, I)})
    d = os.path.join(str(tmp_path), "mimo")
    os.makedirs(d)
    cfg.update(architectures=["MiMoForCausalLM"], model_type="mimo", num_nextn_predict_layers=1)
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    save_file(w, os.path.join(d, "model.safetensors"))
    ref, _ = _gen(d)
    got, n = _proposals(d, speculative_config={"method": "mtp", "num_speculative_tokens": 2})
    assert got == ref and n > 0


def test_deepseek_v3_multi_token_prediction_layer_drafts(tmp_path):
    """``model.layers.<num_hidden_layers>`` of a DeepSeek-V3 checkpoint (enorm / hnorm / eh_proj / shared_head +
    a decoder layer with experts) drafts for its own model."""
    from safetensors.torch import load_file
    path, _ = _save_tiny(str(tmp_path), "deepseek_v3")
    cfg = json.load(open(os.path.join(path, "config.json")))
    w = load_file(os.path.join(path, "model.safetensors"))
    H, V, L = cfg["hidden_size"], cfg["vocab_size"], cfg["num_hidden_layers"]
    g = torch.Generator().manual_seed(3)
    src, dst = f"model.layers.{L - 1}.", f"model.layers.{L}."
    extra = {dst + k[len(src):]: v.clone() + torch.randn(v.shape, generator=g) * 0.01
             for k, v in w.items() if k.startswith(src)}
    extra.update({dst + "enorm.weight": torch.ones(H), dst + "hnorm.weight": torch.ones(H),
                  dst + "eh_proj.weight": torch.randn(H, 2 * H, generator=g) * 0.05,
                  dst + "shared_head.norm.weight": torch.ones(H),
                  dst + "shared_head.head.weight": torch.randn(V, H, generator=g) * 0.05,
                  dst + "embed_tokens.weight": w["model.embed_tokens.weight"].clone()})
    w.update(extra)
    d = os.path.join(str(tmp_path), "dsv3_mtp")
    os.makedirs(d)
    cfg.update(num_nextn_predict_layers=1)
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    save_file(w, os.path.join(d, "model.safetensors"))
    ref, _ = _gen(d)
    got, n = _proposals(d, speculative_config={"method": "mtp", "num_speculative_tokens": 2})
    assert got == ref and n > 0


def test_eagle3_draft_same_tokens_and_layer_math(tmp_path):
    """EAGLE-3 draft over a tiny Llama target: (a) speculative greedy decoding emits the target's own tokens and does
    verify drafted tokens; (b) the draft network's forward (three-depth residual stream -> fc, [normed embedding |
    normed hidden] -> attention, reduced vocabulary mapped through d2t) equals the block written out by hand.
    No transformers class exists for the draft (parity unpinned); names follow the published EAGLE-3 checkpoints."""
    path, hf = _save_tiny(str(tmp_path), "llama")
    cfg = json.load(open(os.path.join(path, "config.json")))
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    hd, nh, nkv = cfg["head_dim"], cfg["num_attention_heads"], cfg["num_key_value_heads"]
    DV = 200
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g) * 0.05      # noqa: E731
    ids = torch.sort(torch.randperm(V, generator=g)[:DV]).values
    w = {"fc.weight": r(H, 3 * H), "midlayer.self_attn.q_proj.weight": r(nh * hd, 2 * H),
         "midlayer.self_attn.k_proj.weight": r(nkv * hd, 2 * H), "midlayer.self_attn.v_proj.weight": r(nkv * hd, 2 * H),
         "midlayer.self_attn.o_proj.weight": r(H, nh * hd), "midlayer.mlp.gate_proj.weight": r(I, H),
         "midlayer.mlp.up_proj.weight": r(I, H), "midlayer.mlp.down_proj.weight": r(H, I),
         "midlayer.input_layernorm.weight": 1 + r(H), "midlayer.hidden_norm.weight": 1 + r(H),
         "midlayer.post_attention_layernorm.weight": 1 + r(H), "norm.weight": 1 + r(H), "lm_head.weight": r(DV, H),
         "d2t": ids - torch.arange(DV), "t2d": torch.zeros(V, dtype=torch.bool)}
    d = os.path.join(str(tmp_path), "eagle3")
    os.makedirs(d)
    cfg.update(architectures=["LlamaForCausalLMEagle3"], num_hidden_layers=1, draft_vocab_size=DV)
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    save_file({k: v.contiguous() for k, v in w.items()}, os.path.join(d, "model.safetensors"))
    ref, _ = _gen(path)
    got, n = _proposals(path, speculative_config={"method": "eagle3", "model": d, "num_speculative_tokens": 3})
    assert got == ref and n > 0
    # (b) one draft forward on a fresh engine's model object, positions 0..5, against the block by hand
    llm = LLM(model=path, **KW, speculative_config={"method": "eagle3", "model": d, "num_speculative_tokens": 2})
    mr = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner
    draft = mr.drafter.model
    assert len(mr.model.model.aux_layers) == 3 and mr.model.model.aux_hidden.shape[1] == 3 * H
    T = 6
    tok = torch.arange(3, 3 + T)
    aux = torch.randn(T, 3 * H, generator=g)
    rms = lambda x, wt: x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + cfg["rms_norm_eps"]) * wt   # noqa: E731
    with torch.no_grad():
        e = hf.model.embed_tokens(tok)
        hdn = aux @ w["fc.weight"].T
        x = torch.cat((rms(e, w["midlayer.input_layernorm.weight"]), rms(hdn, w["midlayer.hidden_norm.weight"])), -1)
        q = (x @ w["midlayer.self_attn.q_proj.weight"].T).view(T, nh, hd)
        k = (x @ w["midlayer.self_attn.k_proj.weight"].T).view(T, nkv, hd)
        v = (x @ w["midlayer.self_attn.v_proj.weight"].T).view(T, nkv, hd)
        theta = cfg.get("rope_theta") or cfg["rope_parameters"]["rope_theta"]
        inv = 1.0 / (theta ** (torch.arange(0, hd, 2).float() / hd))
        ang = torch.arange(T)[:, None].float() * inv[None]

        def rope(t):
            a, b = t[..., : hd // 2], t[..., hd // 2:]
            c, s_ = ang.cos()[:, None], ang.sin()[:, None]
            return torch.cat((a * c - b * s_, b * c + a * s_), -1)
        q, k = rope(q), rope(k).repeat_interleave(nh // nkv, 1)
        v = v.repeat_interleave(nh // nkv, 1)
        o = torch.nn.functional.scaled_dot_product_attention(q.transpose(0, 1)[None], k.transpose(0, 1)[None],
                                                             v.transpose(0, 1)[None], is_causal=True)[0]
        a = o.transpose(0, 1).reshape(T, nh * hd) @ w["midlayer.self_attn.o_proj.weight"].T
        res = hdn + a
        y = rms(res, w["midlayer.post_attention_layernorm.weight"])
        mlp = (torch.nn.functional.silu(y @ w["midlayer.mlp.gate_proj.weight"].T) * (y @ w["midlayer.mlp.up_proj.weight"].T)) \
            @ w["midlayer.mlp.down_proj.weight"].T
        want_h = mlp + res
        want_lg = rms(want_h, w["norm.weight"]) @ w["lm_head.weight"].T
        # run the draft through the engine's own plumbing: a prefill-form pass over one fresh request row
        import numpy as np
        ib = mr.input_batch
        from vllm_distributed_amd.forward_context import set_forward_context
        rows = np.array([0], dtype=np.int64)
        ib.block_tables[0][0, 0] = 5
        got_h = mr.drafter._run(mr, tok.numpy(), aux, rows, np.array([T], dtype=np.int32), np.array([0], dtype=np.int64),
                                decode=False)
        torch.testing.assert_close(got_h, want_h, atol=2e-4, rtol=1e-3)
        lg = draft.compute_logits(got_h)
        assert lg.shape == (T, V)
        torch.testing.assert_close(lg[:, ids], want_lg, atol=2e-4, rtol=1e-3)
        rest = torch.ones(V, dtype=torch.bool)
        rest[ids] = False
        assert torch.isinf(lg[:, rest]).all()
    llm.shutdown()
