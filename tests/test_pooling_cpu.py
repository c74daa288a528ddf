"""Pooling models: last-token embeddings and sequence classification equal the HF reference
(modelled on tests/models/language/pooling)."""
import os

import pytest

import torch

from vllm_distributed_amd import LLM

PROMPTS = [[5, 17, 99, 3, 250, 11], [7] * 30, [1, 2, 3]]
KW = dict(skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=256, max_num_seqs=8,
          max_num_batched_tokens=256)


def test_embedding_and_classification_match_hf(tmp_path):
    import transformers as tf
    torch.manual_seed(0)
    cfg = tf.LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                         num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256, head_dim=64,
                         num_labels=3, pad_token_id=0)
    emb_hf = tf.LlamaModel(cfg).float().eval()
    p1 = os.path.join(tmp_path, "emb")
    emb_hf.save_pretrained(p1, safe_serialization=True)
    llm = LLM(model=p1, task="embed", **KW)
    outs = llm.embed([{"prompt_token_ids": p} for p in PROMPTS])
    llm.shutdown()
    for p, o in zip(PROMPTS, outs):
        with torch.no_grad():
            h = emb_hf(torch.tensor([p])).last_hidden_state[0, -1]
        ref = torch.nn.functional.normalize(h, dim=-1)
        torch.testing.assert_close(torch.tensor(o.outputs.embedding), ref, atol=1e-4, rtol=1e-3)

    cls_hf = tf.LlamaForSequenceClassification(cfg).float().eval()
    p2 = os.path.join(tmp_path, "cls")
    cls_hf.save_pretrained(p2, safe_serialization=True)
    llm = LLM(model=p2, task="classify", **KW)
    outs = llm.classify([{"prompt_token_ids": p} for p in PROMPTS])
    llm.shutdown()
    for p, o in zip(PROMPTS, outs):
        with torch.no_grad():
            ref = torch.softmax(cls_hf(torch.tensor([p])).logits[0], dim=-1)
        torch.testing.assert_close(torch.tensor(o.outputs.probs), ref, atol=1e-4, rtol=1e-3)


def test_pooling_rerank_and_classify_endpoints(tmp_path):
    """/pooling, /rerank and /classify through the FastAPI app (in-process engine, ASGI transport)."""
    import asyncio

    import httpx
    import transformers as tf

    from tests.test_openai_server_cpu import _add_tokenizer
    from vllm_distributed_amd.engine.arg_utils import AsyncEngineArgs
    from vllm_distributed_amd.engine.async_llm import AsyncLLM
    from vllm_distributed_amd.entrypoints.openai.api_server import build_app, make_arg_parser
    torch.manual_seed(0)
    cfg = tf.LlamaConfig(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                         num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256, head_dim=64,
                         num_labels=3, pad_token_id=0, id2label={0: "neg", 1: "neu", 2: "pos"},
                         label2id={"neg": 0, "neu": 1, "pos": 2})
    results = {}
    for task, cls in (("embed", tf.LlamaModel), ("classify", tf.LlamaForSequenceClassification)):
        path = os.path.join(tmp_path, task)
        cls(cfg).float().eval().save_pretrained(path, safe_serialization=True)
        _add_tokenizer(path)
        args = make_arg_parser().parse_args(["--model", path, "--dtype", "float32", "--device", "cpu", "--task", task,
                                             "--max-model-len", "256", "--max-num-seqs", "8",
                                             "--max-num-batched-tokens", "256", "--served-model-name", "tiny"])
        engine = AsyncLLM.from_engine_args(AsyncEngineArgs.from_cli_args(args))
        app = build_app(args, engine)
        loop = asyncio.new_event_loop()

        async def go():
            async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://t") as c:
                if task == "embed":
                    pool = (await c.post("/pooling", json={"model": "tiny", "input": ["w1 w2 w3", "w4"]})).json()
                    emb = (await c.post("/v1/embeddings", json={"input": ["w1 w2 w3", "w4"]})).json()
                    rr = (await c.post("/v1/rerank", json={"query": "w1 w2 w3", "documents": ["w9 w8", "w1 w2 w3", "w4"],
                                                           "top_n": 2})).json()
                    return pool, emb, rr
                return (await c.post("/classify", json={"input": ["w1 w2 w3", "w4 w5"]})).json()
        try:
            results[task] = loop.run_until_complete(asyncio.wait_for(go(), timeout=120))
        finally:
            engine.shutdown()
            loop.close()
    pool, emb, rr = results["embed"]
    assert len(pool["data"]) == 2 and len(pool["data"][0]["data"]) == 128
    torch.testing.assert_close(torch.tensor(pool["data"][0]["data"]), torch.tensor(emb["data"][0]["embedding"]),
                               atol=1e-5, rtol=1e-4)
    assert len(rr["results"]) == 2 and rr["results"][0]["index"] == 1          # the identical document ranks first
    assert rr["results"][0]["relevance_score"] > 0.999 >= rr["results"][1]["relevance_score"]
    cl = results["classify"]
    assert len(cl["data"]) == 2 and cl["data"][0]["num_classes"] == 3
    assert abs(sum(cl["data"][0]["probs"]) - 1.0) < 1e-3 and cl["data"][0]["label"] in ("neg", "neu", "pos")


def _tiny_encoder(tmp, kind):
    import transformers as tf
    torch.manual_seed(0)
    kw = dict(vocab_size=320, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
              max_position_embeddings=128)
    if kind == "bert":
        m = tf.BertModel(tf.BertConfig(type_vocab_size=2, **kw))
    elif kind == "roberta":
        m = tf.RobertaModel(tf.RobertaConfig(type_vocab_size=1, pad_token_id=1, **kw))
    elif kind == "bert_cls":
        m = tf.BertForSequenceClassification(tf.BertConfig(type_vocab_size=2, num_labels=1, **kw))
    else:
        m = tf.RobertaForSequenceClassification(tf.RobertaConfig(type_vocab_size=1, pad_token_id=1, num_labels=3,
                                                                 **kw))
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() >= 2:
                p.mul_(3.0)
    m = m.float().eval()
    path = os.path.join(tmp, kind)
    m.save_pretrained(path, safe_serialization=True)
    return path, m


@pytest.mark.parametrize("kind", ["bert", "roberta"])
def test_encoder_only_embeddings_match_hf(tmp_path, kind):
    """Bidirectional attention through the paged cache: [CLS] embeddings equal transformers'."""
    path, hf = _tiny_encoder(str(tmp_path), kind)
    prompts = [[5, 17, 99, 3, 250, 11], [7, 8, 9] * 11, [2, 3]]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128,
              max_num_seqs=8, block_size=16)
    outs = llm.embed(None, prompt_token_ids=prompts)
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf(torch.tensor([p])).last_hidden_state[0, 0]
        ref = torch.nn.functional.normalize(ref, dim=-1)
        torch.testing.assert_close(torch.as_tensor(o.outputs.embedding), ref, atol=2e-4, rtol=1e-3)
    llm.shutdown()


@pytest.mark.parametrize("kind", ["bert_cls", "roberta_cls"])
def test_encoder_only_classifiers_match_hf(tmp_path, kind):
    from vllm_distributed_amd.sampling_params import PoolingParams
    path, hf = _tiny_encoder(str(tmp_path), kind)
    prompts = [[5, 17, 99, 3, 250, 11, 4, 4], [7, 8, 9] * 7]
    types = [[0, 0, 0, 0, 1, 1, 1, 1], [0] * 10 + [1] * 11]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128,
              max_num_seqs=8, block_size=16)
    for p, t in zip(prompts, types):
        pp = PoolingParams(token_type_ids=t) if kind == "bert_cls" else PoolingParams()
        out = llm.encode(None, pooling_params=pp, prompt_token_ids=[p])[0]
        with torch.no_grad():
            kw = {"token_type_ids": torch.tensor([t])} if kind == "bert_cls" else {}
            logits = hf(torch.tensor([p]), **kw).logits[0]
        ref = torch.sigmoid(logits) if logits.numel() == 1 else torch.softmax(logits, -1)
        torch.testing.assert_close(out.outputs.data.float().flatten(), ref, atol=2e-4, rtol=1e-3)
    llm.shutdown()


@pytest.mark.parametrize("arch", ["Qwen2ForRewardModel", "Qwen2ForProcessRewardModel"])
def test_qwen2_reward_heads(tmp_path, arch):
    """Reward heads over every prompt token.  These classes are not in transformers (parity with the original
    remote code unpinned): a tiny HF Qwen2 backbone plus random head weights, checked against the head applied
    in plain PyTorch to transformers' own hidden states."""
    import json

    import transformers as tf
    from safetensors.torch import save_file
    torch.manual_seed(0)
    cfg = tf.Qwen2Config(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                         num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=256)
    hf = tf.Qwen2Model(cfg).float().eval()
    n = 1 if arch == "Qwen2ForRewardModel" else 2
    head = {"score.0.weight": torch.randn(128, 128) * 0.1, "score.0.bias": torch.randn(128) * 0.1,
            "score.2.weight": torch.randn(n, 128) * 0.1, "score.2.bias": torch.randn(n) * 0.1}
    path = str(tmp_path / "rm")
    os.makedirs(path)
    save_file({**{"model." + k: v.contiguous() for k, v in hf.state_dict().items()}, **head},
              os.path.join(path, "model.safetensors"))
    c = cfg.to_dict()
    c["architectures"] = [arch]
    json.dump(c, open(os.path.join(path, "config.json"), "w"))
    prompts = [[5, 17, 99, 3, 250, 11], [7, 8, 9] * 6]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128, max_num_seqs=4,
              task="reward")
    outs = llm.encode(None, prompt_token_ids=prompts)
    llm.shutdown()
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            h = hf(torch.tensor([p])).last_hidden_state[0]
            ref = torch.relu(h @ head["score.0.weight"].T + head["score.0.bias"]) @ head["score.2.weight"].T \
                + head["score.2.bias"]
            if n == 2:
                ref = torch.softmax(ref, -1)
        torch.testing.assert_close(o.outputs.data.float().view(len(p), n), ref, atol=2e-4, rtol=1e-3)


@pytest.mark.parametrize("kind", ["gpt2", "gemma", "jamba"])
def test_sequence_classification_adapters_match_hf(tmp_path, kind):
    """Classification heads on decoders other than Llama (GPT-2, Gemma, the hybrid Jamba)."""
    import transformers as tf
    torch.manual_seed(0)
    if kind == "gpt2":
        hf = tf.GPT2ForSequenceClassification(tf.GPT2Config(vocab_size=320, n_embd=128, n_layer=2, n_head=2,
                                                            n_positions=256, num_labels=3, pad_token_id=0))
    elif kind == "gemma":
        hf = tf.GemmaForSequenceClassification(tf.GemmaConfig(
            vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
            num_key_value_heads=1, head_dim=64, max_position_embeddings=256, num_labels=4, pad_token_id=0))
    else:
        hf = tf.JambaForSequenceClassification(tf.JambaConfig(
            vocab_size=320, hidden_size=128, intermediate_size=128, num_hidden_layers=4, num_attention_heads=2,
            num_key_value_heads=1, attn_layer_period=2, attn_layer_offset=1, expert_layer_period=2,
            expert_layer_offset=1, num_experts=4, num_experts_per_tok=2, mamba_d_state=8, mamba_d_conv=4,
            mamba_expand=2, mamba_dt_rank=8, use_mamba_kernels=False, num_labels=2, pad_token_id=0))
    hf = hf.float().eval()
    path = os.path.join(tmp_path, kind)
    hf.save_pretrained(path, safe_serialization=True)
    llm = LLM(model=path, task="classify", **KW)
    outs = llm.classify([{"prompt_token_ids": p} for p in PROMPTS])
    llm.shutdown()
    for p, o in zip(PROMPTS, outs):
        with torch.no_grad():
            ref = torch.softmax(hf(torch.tensor([p])).logits[0], dim=-1)
        torch.testing.assert_close(torch.tensor(o.outputs.probs), ref, atol=2e-4, rtol=2e-3)


def test_nomic_bert_rotary_encoder_matches_hf(tmp_path):
    """Rotary BERT with a gated MLP: mean-pooled, normalised sentence embeddings equal transformers' hidden states
    pooled the same way (bidirectional attention with the fused rotary + cache write in front)."""
    import transformers as tf
    torch.manual_seed(0)
    cfg = tf.NomicBertConfig(vocab_size=320, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                             intermediate_size=256, max_position_embeddings=128, head_dim=64)
    hf = tf.NomicBertModel(cfg).float().eval()
    path = os.path.join(tmp_path, "nomic")
    hf.save_pretrained(path, safe_serialization=True)
    prompts = [[5, 17, 99, 3, 250, 11], [7, 8, 9] * 11, [2, 3]]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128,
              max_num_seqs=8, block_size=16)
    outs = llm.embed(None, prompt_token_ids=prompts)
    llm.shutdown()
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf(torch.tensor([p])).last_hidden_state[0].mean(0)
        ref = torch.nn.functional.normalize(ref, dim=-1)
        torch.testing.assert_close(torch.as_tensor(o.outputs.embedding), ref, atol=2e-4, rtol=1e-3)


@pytest.mark.parametrize("arch", ["GteNewModel", "GteModel"])
def test_gte_new_encoder_matches_dense_form(tmp_path, arch):
    """Alibaba-NLP "new" encoder (rotary BERT, biased projections, fused up_gate_proj with the up rows first, GELU
    gate, fixed-NTK rope, CLS pooling) from a checkpoint in its own naming.  transformers has no class for it (remote
    code upstream, parity unpinned): the reference is the block written out in fp32 below."""
    import json
    from safetensors.torch import save_file
    torch.manual_seed(0)
    V, d, H, I, Lr, maxp, theta, factor = 320, 128, 2, 256, 2, 256, 10000.0, 4.0
    D = d // H
    g = lambda *s: torch.randn(*s) * 0.05
    pre = "new." if arch == "GteNewModel" else ""
    sd = {pre + "embeddings.word_embeddings.weight": g(V, d), pre + "embeddings.token_type_embeddings.weight": g(2, d),
          pre + "embeddings.LayerNorm.weight": 1 + g(d), pre + "embeddings.LayerNorm.bias": g(d)}
    for i in range(Lr):
        p = f"{pre}encoder.layer.{i}."
        sd.update({p + "attention.qkv_proj.weight": g(3 * d, d), p + "attention.qkv_proj.bias": g(3 * d),
                   p + "attention.o_proj.weight": g(d, d), p + "attention.o_proj.bias": g(d),
                   p + "attn_ln.weight": 1 + g(d), p + "attn_ln.bias": g(d),
                   p + "mlp.up_gate_proj.weight": g(2 * I, d), p + "mlp.down_proj.weight": g(d, I),
                   p + "mlp.down_proj.bias": g(d), p + "mlp_ln.weight": 1 + g(d), p + "mlp_ln.bias": g(d)})
    sd[pre + "pooler.dense.weight"] = g(d, d)
    path = os.path.join(tmp_path, "gte")
    os.makedirs(path)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    json.dump({"architectures": [arch], "model_type": "new", "vocab_size": V, "hidden_size": d, "num_attention_heads": H,
               "intermediate_size": I, "num_hidden_layers": Lr, "max_position_embeddings": maxp, "rope_theta": theta,
               "rope_scaling": {"type": "ntk", "factor": factor}, "hidden_act": "gelu", "layer_norm_eps": 1e-12,
               "type_vocab_size": 2, "position_embedding_type": "rope", "torch_dtype": "float32"},
              open(os.path.join(path, "config.json"), "w"))
    prompts = [[5, 17, 99, 3, 250, 11], [7, 8, 9] * 11, [2, 3]]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128,
              max_num_seqs=8, block_size=16)
    outs = llm.embed(None, prompt_token_ids=prompts)
    llm.shutdown()
    F = torch.nn.functional
    inv = 1.0 / ((theta * factor) ** (torch.arange(0, D, 2).float() / D)) / factor ** (2 / D)

    def rope(x):                                         # [T, H, D]
        ang = torch.arange(x.shape[0])[:, None].float() * inv[None]
        c, s_ = ang.cos()[:, None], ang.sin()[:, None]
        a, b = x[..., :D // 2], x[..., D // 2:]
        return torch.cat((a * c - b * s_, b * c + a * s_), -1)

    for p_ids, o in zip(prompts, outs):
        T = len(p_ids)
        x = sd[pre + "embeddings.word_embeddings.weight"][p_ids] + sd[pre + "embeddings.token_type_embeddings.weight"][0]
        x = F.layer_norm(x, (d,), sd[pre + "embeddings.LayerNorm.weight"], sd[pre + "embeddings.LayerNorm.bias"], 1e-12)
        for i in range(Lr):
            p = f"{pre}encoder.layer.{i}."
            q, k, v = (x @ sd[p + "attention.qkv_proj.weight"].T + sd[p + "attention.qkv_proj.bias"]).chunk(3, -1)
            q, k, v = rope(q.view(T, H, D)), rope(k.view(T, H, D)), v.view(T, H, D)
            a = F.scaled_dot_product_attention(q.transpose(0, 1)[None], k.transpose(0, 1)[None], v.transpose(0, 1)[None])
            a = a[0].transpose(0, 1).reshape(T, d) @ sd[p + "attention.o_proj.weight"].T + sd[p + "attention.o_proj.bias"]
            x = F.layer_norm(x + a, (d,), sd[p + "attn_ln.weight"], sd[p + "attn_ln.bias"], 1e-12)
            up, gate = (x @ sd[p + "mlp.up_gate_proj.weight"].T).chunk(2, -1)
            m = (F.gelu(gate) * up) @ sd[p + "mlp.down_proj.weight"].T + sd[p + "mlp.down_proj.bias"]
            x = F.layer_norm(x + m, (d,), sd[p + "mlp_ln.weight"], sd[p + "mlp_ln.bias"], 1e-12)
        ref = F.normalize(x[0], dim=-1)
        torch.testing.assert_close(torch.as_tensor(o.outputs.embedding), ref, atol=2e-4, rtol=1e-3)


def test_modernbert_matches_hf(tmp_path):
    """ModernBERT: alternating global / local (|i - j| <= 4 here) bidirectional attention with two rope bases, GeGLU
    MLP, no norm before the first attention.  Embeddings (CLS, normalised) and classification probabilities against
    transformers; one prompt is longer than the local window so the band matters."""
    import transformers as tf
    torch.manual_seed(0)
    cfg = tf.ModernBertConfig(vocab_size=320, hidden_size=128, num_hidden_layers=4, num_attention_heads=2,
                              intermediate_size=192, max_position_embeddings=128, local_attention=8,
                              global_attn_every_n_layers=3, num_labels=3, pad_token_id=0, bos_token_id=1,
                              eos_token_id=2, cls_token_id=1, sep_token_id=2)
    hf = tf.ModernBertForSequenceClassification(cfg).float().eval()
    with torch.no_grad():
        for n, p_ in hf.named_parameters():
            if "norm" in n and n.endswith("weight"):
                p_.add_(torch.randn_like(p_) * 0.1)
    path = os.path.join(tmp_path, "mb")
    hf.save_pretrained(path, safe_serialization=True)
    prompts = [[5, 17, 99, 3, 250, 11], [7, 8, 9] * 11, [2, 3]]
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128,
              max_num_seqs=8, block_size=16)
    outs = llm.classify([{"prompt_token_ids": p} for p in prompts])
    llm.shutdown()
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = hf(torch.tensor([p])).logits[0].softmax(-1)
        torch.testing.assert_close(torch.as_tensor(o.outputs.probs), ref, atol=2e-4, rtol=1e-3)
    import json
    c = json.load(open(os.path.join(path, "config.json")))
    c["architectures"] = ["ModernBertModel"]
    json.dump(c, open(os.path.join(path, "config.json"), "w"))
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=128,
              max_num_seqs=8, block_size=16)
    outs = llm.embed(None, prompt_token_ids=prompts)
    llm.shutdown()
    for p, o in zip(prompts, outs):
        with torch.no_grad():
            ref = torch.nn.functional.normalize(hf.model(torch.tensor([p])).last_hidden_state[0, 0], dim=-1)
        torch.testing.assert_close(torch.as_tensor(o.outputs.embedding), ref, atol=2e-4, rtol=1e-3)
