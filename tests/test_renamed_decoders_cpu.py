"""Families that are the Llama (or GLM) block under other checkpoint names: a tiny transformers model is written out
in the family's own layout (names, fused tensors, config keys) and must decode to the tokens transformers produces.
transformers itself carries none of these architectures (they ship as remote code), so the layouts follow the
reference's loaders: vllm/model_executor/models/{exaone,telechat2,teleflm,orion,qwen,chatglm}.py."""
import json
import os

import pytest
import torch
from safetensors.torch import save_file

from vllm_distributed_amd import LLM, SamplingParams

PROMPTS = [[5, 17, 99, 3, 250, 11], [7] * 20, [1, 2, 3], list(range(30, 75))]


def _llama(**kw):
    import transformers as tf
    torch.manual_seed(0)
    base = dict(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                num_key_value_heads=1, max_position_embeddings=256, rms_norm_eps=1e-5, tie_word_embeddings=False,
                head_dim=64)
    base.update(kw)
    m = tf.LlamaForCausalLM(tf.LlamaConfig(**base)).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith(".bias"):
                p.normal_(0, 0.05)
    return m


def _hf_tokens(m, n=8):
    out = []
    with torch.no_grad():
        for p in PROMPTS:
            out.append(m.generate(torch.tensor([p]), max_new_tokens=n, do_sample=False, pad_token_id=0,
                                  eos_token_id=None)[0, len(p):].tolist())
    return out


def _ours(path, n=8):
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128, max_num_seqs=4)
    try:
        outs = llm.generate(prompt_token_ids=PROMPTS, sampling_params=SamplingParams(temperature=0, max_tokens=n,
                                                                                     ignore_eos=True))
        return [list(o.outputs[0].token_ids) for o in outs]
    finally:
        llm.shutdown()


def _write(tmp, name, config, tensors):
    d = os.path.join(str(tmp), name)
    os.makedirs(d)
    json.dump(config, open(os.path.join(d, "config.json"), "w"))
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(d, "model.safetensors"))
    return d


def _layers(sd, n_layers):
    for i in range(n_layers):
        yield i, {k[len(f"model.layers.{i}."):]: v for k, v in sd.items() if k.startswith(f"model.layers.{i}.")}


def test_exaone_layout(tmp_path):
    m = _llama()
    sd = m.state_dict()
    t = {"transformer.wte.weight": sd["model.embed_tokens.weight"], "transformer.ln_f.weight": sd["model.norm.weight"],
         "lm_head.weight": sd["lm_head.weight"]}
    for i, L in _layers(sd, 2):
        p = f"transformer.h.{i}."
        for a, b in (("q_proj", "q_proj"), ("k_proj", "k_proj"), ("v_proj", "v_proj"), ("o_proj", "out_proj")):
            t[p + f"attn.attention.{b}.weight"] = L[f"self_attn.{a}.weight"]
        t[p + "ln_1.weight"], t[p + "ln_2.weight"] = L["input_layernorm.weight"], L["post_attention_layernorm.weight"]
        t[p + "mlp.c_fc_0.weight"], t[p + "mlp.c_fc_1.weight"] = L["mlp.gate_proj.weight"], L["mlp.up_proj.weight"]
        t[p + "mlp.c_proj.weight"] = L["mlp.down_proj.weight"]
    cfg = dict(model_type="exaone", architectures=["ExaoneForCausalLM"], vocab_size=320, hidden_size=128,
               intermediate_size=256, num_layers=2, num_attention_heads=2, num_key_value_heads=1, head_dim=64,
               max_position_embeddings=256, layer_norm_epsilon=1e-5, activation_function="silu", rope_theta=10000.0,
               tie_word_embeddings=False, torch_dtype="float32")
    assert _ours(_write(tmp_path, "exaone", cfg, t)) == _hf_tokens(m)


def test_telechat2_layout(tmp_path):
    m = _llama(attention_bias=True, mlp_bias=True, num_key_value_heads=2)
    with torch.no_grad():
        for n, p in m.named_parameters():          # TeleChat2 has biases on the output projections only
            if n.endswith(("q_proj.bias", "k_proj.bias", "v_proj.bias", "gate_proj.bias", "up_proj.bias")):
                p.zero_()
    sd = m.state_dict()
    t = {"transformer.word_embeddings.weight": sd["model.embed_tokens.weight"],
         "transformer.ln_f.weight": sd["model.norm.weight"], "lm_head.weight": sd["lm_head.weight"]}
    for i, L in _layers(sd, 2):
        p = f"transformer.h.{i}."
        t[p + "self_attention.query.weight"] = L["self_attn.q_proj.weight"]
        k, v = L["self_attn.k_proj.weight"].view(2, 64, 128), L["self_attn.v_proj.weight"].view(2, 64, 128)
        t[p + "self_attention.key_value.weight"] = torch.stack([k, v], dim=1).reshape(2 * 2 * 64, 128)
        t[p + "self_attention.dense.weight"], t[p + "self_attention.dense.bias"] = \
            L["self_attn.o_proj.weight"], L["self_attn.o_proj.bias"]
        for w in ("input_layernorm.weight", "post_attention_layernorm.weight", "mlp.gate_proj.weight",
                  "mlp.up_proj.weight", "mlp.down_proj.weight", "mlp.down_proj.bias"):
            t[p + w] = L[w]
    cfg = dict(model_type="telechat", architectures=["TeleChat2ForCausalLM"], vocab_size=320, hidden_size=128,
               ffn_hidden_size=256, n_layer=2, n_head=2, layer_norm_epsilon=1e-5, training_seqlen=256,
               tie_word_embeddings=False, torch_dtype="float32")
    assert _ours(_write(tmp_path, "telechat", cfg, t)) == _hf_tokens(m)


def test_teleflm_mup_scaling(tmp_path):
    m = _llama()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    cfg = dict(m.config.to_dict(), model_type="llama", architectures=["TeleFLMForCausalLM"], use_mup=True,
               input_mult=2.5, output_mult=3.0, mup_scale_factor=4.0, torch_dtype="float32")
    path = _write(tmp_path, "teleflm", cfg, sd)
    with torch.no_grad():                           # the same scaling folded into the transformers model
        m.model.embed_tokens.weight.mul_(2.5)
        m.lm_head.weight.mul_(3.0 / 4.0)
    assert _ours(path) == _hf_tokens(m)


def test_orion_layernorm_blocks(tmp_path):
    m = _llama()
    torch.manual_seed(1)
    for layer in m.model.layers:
        layer.input_layernorm = torch.nn.LayerNorm(128, eps=1e-5)
        layer.post_attention_layernorm = torch.nn.LayerNorm(128, eps=1e-5)
    m.model.norm = torch.nn.LayerNorm(128, eps=1e-5)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "norm" in n:
                p.add_(torch.randn_like(p) * 0.1)
    cfg = dict(m.config.to_dict(), model_type="orion", architectures=["OrionForCausalLM"], torch_dtype="float32")
    assert _ours(_write(tmp_path, "orion", cfg, m.state_dict())) == _hf_tokens(m)


def test_qwen_v1_layout(tmp_path):
    m = _llama(attention_bias=True, num_key_value_heads=2)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("o_proj.bias"):
                p.zero_()
    sd = m.state_dict()
    t = {"transformer.wte.weight": sd["model.embed_tokens.weight"], "transformer.ln_f.weight": sd["model.norm.weight"],
         "lm_head.weight": sd["lm_head.weight"]}
    for i, L in _layers(sd, 2):
        p = f"transformer.h.{i}."
        t[p + "attn.c_attn.weight"] = torch.cat([L[f"self_attn.{x}_proj.weight"] for x in "qkv"])
        t[p + "attn.c_attn.bias"] = torch.cat([L[f"self_attn.{x}_proj.bias"] for x in "qkv"])
        t[p + "attn.c_proj.weight"] = L["self_attn.o_proj.weight"]
        t[p + "ln_1.weight"], t[p + "ln_2.weight"] = L["input_layernorm.weight"], L["post_attention_layernorm.weight"]
        t[p + "mlp.w2.weight"], t[p + "mlp.w1.weight"] = L["mlp.gate_proj.weight"], L["mlp.up_proj.weight"]
        t[p + "mlp.c_proj.weight"] = L["mlp.down_proj.weight"]
    cfg = dict(model_type="qwen", architectures=["QWenLMHeadModel"], vocab_size=320, hidden_size=128,
               intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, kv_channels=64, seq_length=256,
               layer_norm_epsilon=1e-5, rotary_emb_base=10000, tie_word_embeddings=False, torch_dtype="float32")
    assert _ours(_write(tmp_path, "qwen", cfg, t)) == _hf_tokens(m)


@pytest.mark.parametrize("qkv_bias", [False, True])
def test_chatglm_layout(tmp_path, qkv_bias):
    import transformers as tf
    torch.manual_seed(0)
    m = tf.GlmForCausalLM(tf.GlmConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                       num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256,
                                       head_dim=64, pad_token_id=0, attention_bias=qkv_bias)).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith(".bias"):
                p.normal_(0, 0.05)
    sd = m.state_dict()
    # the original layout pads the vocabulary: 320 real rows, 384 stored
    pad = torch.zeros(64, 128)
    t = {"transformer.embedding.word_embeddings.weight": torch.cat([sd["model.embed_tokens.weight"], pad]),
         "transformer.encoder.final_layernorm.weight": sd["model.norm.weight"],
         "transformer.output_layer.weight": torch.cat([sd["lm_head.weight"], pad + 5.0]),
         "transformer.rotary_pos_emb.inv_freq": torch.ones(16)}
    for i, L in _layers(sd, 2):
        p = f"transformer.encoder.layers.{i}."
        t[p + "self_attention.query_key_value.weight"] = torch.cat([L[f"self_attn.{x}_proj.weight"] for x in "qkv"])
        if qkv_bias:
            t[p + "self_attention.query_key_value.bias"] = torch.cat([L[f"self_attn.{x}_proj.bias"] for x in "qkv"])
        t[p + "self_attention.dense.weight"] = L["self_attn.o_proj.weight"]
        t[p + "input_layernorm.weight"] = L["input_layernorm.weight"]
        t[p + "post_attention_layernorm.weight"] = L["post_attention_layernorm.weight"]
        t[p + "mlp.dense_h_to_4h.weight"] = L["mlp.gate_up_proj.weight"]
        t[p + "mlp.dense_4h_to_h.weight"] = L["mlp.down_proj.weight"]
    cfg = dict(model_type="chatglm", architectures=["ChatGLMModel"], vocab_size=320, padded_vocab_size=384,
               hidden_size=128, ffn_hidden_size=256, num_layers=2, num_attention_heads=2, kv_channels=64,
               multi_query_attention=True, multi_query_group_num=1, seq_length=256, layernorm_epsilon=m.config.rms_norm_eps,
               rmsnorm=True, add_bias_linear=False, add_qkv_bias=qkv_bias, post_layer_norm=True,
               apply_residual_connection_post_layernorm=False, original_rope=True, torch_dtype="float32")
    # the padded head rows carry large logits: they must never be sampled
    assert _ours(_write(tmp_path, "chatglm", cfg, t)) == _hf_tokens(m)


def test_minicpm3_latent_attention_and_scaling(tmp_path):
    """MiniCPM3 = DeepSeek-V2's latent attention with half-split rotary pairs + MiniCPM's scaled branches.  A tiny
    transformers DeepSeek-V2 (dense) is turned into the equivalent MiniCPM3 checkpoint: rotary rows de-interleaved,
    and the branch / embedding / logit scales divided out of the weights that the MiniCPM3 config then re-applies."""
    import transformers as tf
    torch.manual_seed(0)
    cfg = tf.DeepseekV2Config(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                              num_attention_heads=4, num_key_value_heads=4, n_routed_experts=4,
                              first_k_dense_replace=2, moe_intermediate_size=64, kv_lora_rank=32, q_lora_rank=48, qk_rope_head_dim=16,
                              qk_nope_head_dim=32, v_head_dim=32, max_position_embeddings=256, tie_word_embeddings=False)
    m = tf.DeepseekV2ForCausalLM(cfg).eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    H, nope, rope, R = 4, 32, 16, 32
    perm = torch.cat([torch.arange(0, rope, 2), torch.arange(1, rope, 2)])
    scale_emb, scale_depth, base = 4.0, 2.8, 32
    branch = scale_depth / (2 ** 0.5)
    for i in range(2):
        p = f"model.layers.{i}.self_attn."
        w = sd[p + "q_b_proj.weight"].view(H, nope + rope, -1)
        w[:, nope:] = w[:, nope:][:, perm]
        sd[p + "q_b_proj.weight"] = w.reshape(H * (nope + rope), -1)
        w = sd[p + "kv_a_proj_with_mqa.weight"]
        w[R:] = w[R:][perm]
        sd[p + "o_proj.weight"] /= branch
        sd[f"model.layers.{i}.mlp.down_proj.weight"] /= branch
    sd["model.embed_tokens.weight"] /= scale_emb
    sd["lm_head.weight"] /= base / 128
    out = dict({k: v for k, v in cfg.to_dict().items() if k not in ("n_routed_experts", "n_shared_experts")},
               model_type="minicpm3", architectures=["MiniCPM3ForCausalLM"], scale_emb=scale_emb,
               scale_depth=scale_depth, dim_model_base=base, torch_dtype="float32")
    assert _ours(_write(tmp_path, "minicpm3", out, sd)) == _hf_tokens(m)


def test_jais_matches_dense_form(tmp_path):
    """JAIS (GPT-2 layout, ALiBi, SwiGLU over two up projections, muP scales).  The upstream model is remote code
    (no transformers class; parity unpinned): the reference is the network written out in fp32."""
    import json
    import math
    from safetensors.torch import save_file
    from vllm_distributed_amd import LLM, SamplingParams
    torch.manual_seed(0)
    V, d, H, I, Lr = 320, 128, 4, 192, 2
    D = d // H
    g = lambda *s: torch.randn(*s) * 0.08      # noqa: E731
    sd = {"transformer.wte.weight": g(V, d), "transformer.ln_f.weight": 1 + g(d), "transformer.ln_f.bias": g(d),
          "transformer.relative_pe.slopes": torch.zeros(H, 1, 1)}
    for i in range(Lr):
        p = f"transformer.h.{i}."
        sd.update({p + "ln_1.weight": 1 + g(d), p + "ln_1.bias": g(d), p + "ln_2.weight": 1 + g(d), p + "ln_2.bias": g(d),
                   p + "attn.c_attn.weight": g(d, 3 * d), p + "attn.c_attn.bias": g(3 * d),
                   p + "attn.c_proj.weight": g(d, d), p + "attn.c_proj.bias": g(d),
                   p + "mlp.c_fc.weight": g(d, I), p + "mlp.c_fc.bias": g(I), p + "mlp.c_fc2.weight": g(d, I),
                   p + "mlp.c_fc2.bias": g(I), p + "mlp.c_proj.weight": g(I, d), p + "mlp.c_proj.bias": g(d)})
    path = os.path.join(tmp_path, "jais")
    os.makedirs(path)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    cfg = {"architectures": ["JAISLMHeadModel"], "model_type": "jais", "vocab_size": V, "n_embd": d, "n_head": H,
           "n_layer": Lr, "n_inner": I, "n_positions": 256, "layer_norm_epsilon": 1e-5, "activation_function": "swiglu",
           "position_embedding_type": "alibi", "mup_embeddings_scale": 9.1, "mup_output_alpha": 1.1,
           "mup_width_scale": 0.2, "mup_scale_qk_dot_by_d": True, "torch_dtype": "float32",
           "scale_attn_by_inverse_layer_idx": False}
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    prompts = [[5, 17, 99, 3, 250, 11], [7, 8, 9] * 7]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128, max_num_seqs=4)
    outs = llm.generate([{"prompt_token_ids": p} for p in prompts],
                        SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
    llm.shutdown()
    F = torch.nn.functional

    def slopes(n):
        start = 2 ** (-(2 ** -(math.log2(n) - 3)))
        return torch.tensor([start * start ** i for i in range(n)])

    def logits(ids):
        T = len(ids)
        x = sd["transformer.wte.weight"][ids] * 9.1
        pos = torch.arange(T)
        bias = -(pos[:, None] - pos[None, :]).float()[None] * slopes(H)[:, None, None]       # [H, T, T]
        mask = pos[None, :] > pos[:, None]
        for i in range(Lr):
            p = f"transformer.h.{i}."
            h = F.layer_norm(x, (d,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], 1e-5)
            q, k, v = (h @ sd[p + "attn.c_attn.weight"] + sd[p + "attn.c_attn.bias"]).view(T, 3, H, D).unbind(1)
            s = torch.einsum("ihd,jhd->hij", q, k) / D + bias
            s = s.masked_fill(mask[None], float("-inf")).softmax(-1)
            a = torch.einsum("hij,jhd->ihd", s, v).reshape(T, d)
            x = x + a @ sd[p + "attn.c_proj.weight"] + sd[p + "attn.c_proj.bias"]
            h = F.layer_norm(x, (d,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], 1e-5)
            m = (h @ sd[p + "mlp.c_fc.weight"] + sd[p + "mlp.c_fc.bias"]) * \
                F.silu(h @ sd[p + "mlp.c_fc2.weight"] + sd[p + "mlp.c_fc2.bias"])
            x = x + m @ sd[p + "mlp.c_proj.weight"] + sd[p + "mlp.c_proj.bias"]
        x = F.layer_norm(x, (d,), sd["transformer.ln_f.weight"], sd["transformer.ln_f.bias"], 1e-5)
        return (x[-1] @ sd["transformer.wte.weight"].T) * (1.1 * 0.2)

    for p, o in zip(prompts, outs):
        ids = list(p)
        for tok in o.outputs[0].token_ids:
            lg = logits(ids)
            assert tok == int(lg.argmax()) or float(lg.max() - lg[tok]) < 1e-4
            ids.append(tok)


def test_arctic_matches_dense_form(tmp_path):
    """Snowflake Arctic: dense layers and expert layers alternate; an expert layer also runs a small dense MLP on the
    post-attention stream while its experts read the normalised layer input.  No transformers class (remote code
    upstream, parity unpinned): the reference is the network written out in fp32 from the checkpoint's own names."""
    torch.manual_seed(0)
    V, d, H, KV, I, E, Lr = 320, 128, 2, 1, 192, 4, 4
    D = d // H
    g = lambda *s: torch.randn(*s) * 0.06      # noqa: E731
    sd = {"model.embed_tokens.weight": g(V, d), "model.norm.weight": 1 + g(d), "lm_head.weight": g(V, d)}
    for i in range(Lr):
        p = f"model.layers.{i}."
        sd.update({p + "self_attn.q_proj.weight": g(H * D, d), p + "self_attn.k_proj.weight": g(KV * D, d),
                   p + "self_attn.v_proj.weight": g(KV * D, d), p + "self_attn.o_proj.weight": g(d, H * D),
                   p + "input_layernorm.weight": 1 + g(d), p + "post_attention_layernorm.weight": 1 + g(d)})
        if (i + 1) % 2 == 0:
            sd.update({p + "residual_layernorm.weight": 1 + g(d), p + "residual_mlp.w1.weight": g(d, d),
                       p + "residual_mlp.w3.weight": g(d, d), p + "residual_mlp.w2.weight": g(d, d),
                       p + "block_sparse_moe.gate.weight": g(E, d)})
            for e in range(E):
                q = p + f"block_sparse_moe.experts.{e}."
                sd.update({q + "w1.weight": g(I, d), q + "w3.weight": g(I, d), q + "w2.weight": g(d, I)})
        else:
            q = p + "block_sparse_moe.mlp."
            sd.update({q + "w1.weight": g(I, d), q + "w3.weight": g(I, d), q + "w2.weight": g(d, I)})
    path = os.path.join(tmp_path, "arctic")
    os.makedirs(path)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    json.dump({"architectures": ["ArcticForCausalLM"], "model_type": "arctic", "vocab_size": V, "hidden_size": d,
               "num_attention_heads": H, "num_key_value_heads": KV, "intermediate_size": I, "num_hidden_layers": Lr,
               "num_local_experts": E, "num_experts_per_tok": 2, "moe_layer_frequency": 2, "use_residual": True,
               "rms_norm_eps": 1e-5, "rope_theta": 10000.0, "max_position_embeddings": 256, "hidden_act": "silu",
               "tie_word_embeddings": False, "torch_dtype": "float32"}, open(os.path.join(path, "config.json"), "w"))
    prompts = [[5, 17, 99, 3, 250, 11], [7, 8, 9] * 7]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128, max_num_seqs=4)
    outs = llm.generate([{"prompt_token_ids": p} for p in prompts],
                        SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
    llm.shutdown()
    F = torch.nn.functional
    rms = lambda x, w: x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * w      # noqa: E731
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))

    def glu(x, q):
        return (F.silu(x @ sd[q + "w1.weight"].T) * (x @ sd[q + "w3.weight"].T)) @ sd[q + "w2.weight"].T

    def logits(ids):
        T = len(ids)
        ang = torch.arange(T)[:, None].float() * inv[None]

        def rope(t):
            a, b = t[..., : D // 2], t[..., D // 2:]
            c, s_ = ang.cos()[:, None], ang.sin()[:, None]
            return torch.cat((a * c - b * s_, b * c + a * s_), -1)
        x = sd["model.embed_tokens.weight"][ids]
        for i in range(Lr):
            p = f"model.layers.{i}."
            h = rms(x, sd[p + "input_layernorm.weight"])
            q = rope((h @ sd[p + "self_attn.q_proj.weight"].T).view(T, H, D))
            k = rope((h @ sd[p + "self_attn.k_proj.weight"].T).view(T, KV, D)).repeat_interleave(H // KV, 1)
            v = (h @ sd[p + "self_attn.v_proj.weight"].T).view(T, KV, D).repeat_interleave(H // KV, 1)
            o = F.scaled_dot_product_attention(q.transpose(0, 1)[None], k.transpose(0, 1)[None], v.transpose(0, 1)[None],
                                               is_causal=True)[0].transpose(0, 1).reshape(T, H * D)
            attn = x + o @ sd[p + "self_attn.o_proj.weight"].T
            if (i + 1) % 2 == 0:
                res = glu(rms(attn, sd[p + "residual_layernorm.weight"]), p + "residual_mlp.")
                m_in = rms(x, sd[p + "post_attention_layernorm.weight"])
                w, idx = (m_in @ sd[p + "block_sparse_moe.gate.weight"].T).softmax(-1).topk(2, -1)
                w = w / w.sum(-1, keepdim=True)
                moe = torch.zeros_like(m_in)
                for t in range(T):
                    for j in range(2):
                        moe[t] += w[t, j] * glu(m_in[t:t + 1], p + f"block_sparse_moe.experts.{int(idx[t, j])}.")[0]
                x = attn + res + moe
            else:
                x = attn + glu(rms(attn, sd[p + "post_attention_layernorm.weight"]), p + "block_sparse_moe.mlp.")
        return rms(x[-1], sd["model.norm.weight"]) @ sd["lm_head.weight"].T

    for p, o in zip(prompts, outs):
        ids = list(p)
        for tok in o.outputs[0].token_ids:
            lg = logits(ids)
            assert tok == int(lg.argmax()) or float(lg.max() - lg[tok]) < 1e-4
            ids.append(tok)


def test_glm4v_chat_layout_with_eva_tower(tmp_path):
    """GLM-4V in the ChatGLM layout: the text path must still equal transformers' GLM on the re-laid-out weights, one
    EVA2-CLIP block is checked against its arithmetic written out (norms BEHIND attention and MLP), and an image
    request fills begin marker + placeholders + end marker from the tower (parity of the tower: unpinned)."""
    import numpy as np
    import transformers as tf
    from vllm_distributed_amd.models.glm4v_chat import EVA2CLIPModel, _EvaLayer
    torch.manual_seed(0)
    m = tf.GlmForCausalLM(tf.GlmConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                       num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256,
                                       head_dim=64, pad_token_id=0, attention_bias=True)).eval()
    sd = m.state_dict()
    t = {"transformer.embedding.word_embeddings.weight": sd["model.embed_tokens.weight"],
         "transformer.encoder.final_layernorm.weight": sd["model.norm.weight"],
         "transformer.output_layer.weight": sd["lm_head.weight"]}
    for i, L in _layers(sd, 2):
        p = f"transformer.encoder.layers.{i}."
        t[p + "self_attention.query_key_value.weight"] = torch.cat([L[f"self_attn.{x}_proj.weight"] for x in "qkv"])
        t[p + "self_attention.query_key_value.bias"] = torch.cat([L[f"self_attn.{x}_proj.bias"] for x in "qkv"])
        t[p + "self_attention.dense.weight"] = L["self_attn.o_proj.weight"]
        t[p + "input_layernorm.weight"] = L["input_layernorm.weight"]
        t[p + "post_attention_layernorm.weight"] = L["post_attention_layernorm.weight"]
        t[p + "mlp.dense_h_to_4h.weight"] = L["mlp.gate_up_proj.weight"]
        t[p + "mlp.dense_4h_to_h.weight"] = L["mlp.down_proj.weight"]
    vc = dict(hidden_size=64, num_heads=2, intermediate_size=128, num_hidden_layers=2, image_size=56, patch_size=14,
              in_channels=3, layer_norm_eps=1e-6, hidden_act="gelu", scaling_factor=8.0, dropout_prob=0.0)
    cfg = dict(model_type="chatglm", architectures=["GLM4VForCausalLM"], vocab_size=320, padded_vocab_size=320,
               hidden_size=128, ffn_hidden_size=256, num_layers=2, num_attention_heads=2, kv_channels=64,
               multi_query_attention=True, multi_query_group_num=1, seq_length=256, layernorm_epsilon=m.config.rms_norm_eps,
               rmsnorm=True, add_bias_linear=False, add_qkv_bias=True, post_layer_norm=True,
               apply_residual_connection_post_layernorm=False, original_rope=True, torch_dtype="float32",
               vision_config=vc, boi_token_id=301, eoi_token_id=302, pad_token_id=300)
    from types import SimpleNamespace
    tower = EVA2CLIPModel(SimpleNamespace(vision_config=vc, hidden_size=128, ffn_hidden_size=256)).eval()
    with torch.no_grad():
        for p_ in tower.parameters():
            p_.normal_(0, 0.2)
    t.update({"transformer.vision." + n: v.detach().clone() for n, v in tower.state_dict().items()})
    # one block, written out
    lay = _EvaLayer(vc).eval()
    x = torch.randn(1, 5, 64)
    with torch.no_grad():
        qkv = lay.attention.query_key_value(x)[0]
        heads = []
        for h in range(2):
            q, k, v = (qkv[:, j * 64 + h * 32: j * 64 + (h + 1) * 32] for j in range(3))
            heads.append(torch.softmax(q @ k.t() / 32 ** 0.5, -1) @ v)
        y = x + lay.input_layernorm(lay.attention.dense(torch.cat(heads, -1)))[None]
        y = y + lay.post_attention_layernorm(lay.mlp.fc2(torch.nn.functional.gelu(lay.mlp.fc1(y))))
        torch.testing.assert_close(lay(x), y, rtol=1e-4, atol=1e-5)
    path = _write(tmp_path, "glm4v", cfg, t)
    assert _ours(path) == _hf_tokens(m)
    img = (np.random.default_rng(3).random((40, 52, 3)) * 255).astype(np.uint8)
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128, max_num_seqs=4)
    try:
        outs = llm.generate([{"prompt_token_ids": [5, 301, 300, 302, 9], "multi_modal_data": {"image": img}},
                             {"prompt_token_ids": [5, 300, 9], "multi_modal_data": {"image": img}}],
                            SamplingParams(temperature=0, max_tokens=4, ignore_eos=True))
        want = [5, 301] + [300] * 4 + [302, 9]
        assert list(outs[0].prompt_token_ids) == want and list(outs[1].prompt_token_ids) == want
        assert list(outs[0].outputs[0].token_ids) == list(outs[1].outputs[0].token_ids)
        model = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner.model
        from vllm_distributed_amd.multimodal import process_multimodal
        _, inputs, _, pos = process_multimodal(llm.llm_engine.vllm_config, [5, 300, 9], {"image": img})
        assert pos[0].offset == 1 and pos[0].length == 6
        with torch.no_grad():
            dev = next(model.parameters()).device          # these tests also run with the engine on the GPU
            feats = model.encode_images(inputs[0][None].to(dev)).cpu()
            torch.testing.assert_close(feats, tower(inputs[0][None]), atol=1e-4, rtol=1e-4)
        assert feats.shape == (1, 6, 128)
    finally:
        llm.shutdown()
