"""Dual-chunk attention (Qwen2.5-1M): the engine's three-pass form over paged keys against a dense fp32
re-computation of the definition (one softmax over scores whose query rotation depends on how many chunks lie
between query and key).  transformers has no implementation of it; the dense form below is written from the
reference's rule (vllm/attention/backends/dual_chunk_flash_attn.py:705-1145, rotary_embedding.py:1679-1845)."""
import math
import os

import pytest
import torch

from vllm_distributed_amd import LLM, SamplingParams


def _rot(x, pos, inv_freq):
    """Neox-style rotary of x [T, H, D] at integer positions pos [T]."""
    ang = pos[:, None].float() * inv_freq[None, :]
    cos, sin = ang.cos()[:, None, :], ang.sin()[:, None, :]
    h = x.shape[-1] // 2
    a, b = x[..., :h], x[..., h:]
    return torch.cat((a * cos - b * sin, b * cos + a * sin), dim=-1)


def dense_dual_chunk_logits(hf, ids, chunk_size, local_size, orig_max, prompt_len):
    """Logits at the last position of ``ids`` with dual-chunk attention in every layer."""
    m, cfg = hf.model, hf.config
    L = chunk_size - local_size
    T = len(ids)
    H, Hkv = cfg.num_attention_heads, cfg.num_key_value_heads
    D = cfg.hidden_size // H
    theta = getattr(cfg, "rope_theta", None) or cfg.rope_parameters["rope_theta"]
    inv = 1.0 / (theta ** (torch.arange(0, D, 2).float() / D))
    pos = torch.arange(T)
    p_in = pos % L
    f = torch.ones(T)
    if orig_max:
        known = torch.maximum(pos + 1, torch.tensor(prompt_len))
        f = (0.1 * torch.log(known.float() / orig_max) + 1.0).clamp(min=1.0)
    x = m.embed_tokens(torch.tensor(ids))
    ci = (pos // L)[:, None]
    cj = (pos // L)[None, :]
    causal = pos[None, :] <= pos[:, None]
    for layer in m.layers:
        a = layer.self_attn
        h = layer.input_layernorm(x)
        q = a.q_proj(h).view(T, H, D) * f[:, None, None]
        k = a.k_proj(h).view(T, Hkv, D) * f[:, None, None]
        v = a.v_proj(h).view(T, Hkv, D)
        k = _rot(k, p_in, inv).repeat_interleave(H // Hkv, dim=1)
        v = v.repeat_interleave(H // Hkv, dim=1)
        q0 = _rot(q, p_in, inv)
        q1 = _rot(q, (p_in + L).clamp(max=chunk_size), inv)
        q2 = _rot(q, torch.full_like(p_in, min(2 * L - 1, chunk_size)), inv)
        s0, s1, s2 = (torch.einsum("ihd,jhd->hij", qq, k) for qq in (q0, q1, q2))
        s = torch.where(cj == ci, s0, torch.where(cj == ci - 1, s1, s2)) / math.sqrt(D)
        s = s.masked_fill(~causal, float("-inf"))
        o = torch.einsum("hij,jhd->ihd", s.softmax(-1), v).reshape(T, H * D)
        x = x + a.o_proj(o)
        x = x + layer.mlp(layer.post_attention_layernorm(x))
    return hf.lm_head(m.norm(x[-1]))


def save_tiny_dual_chunk(tmp_path, orig_max, head_dim=64):
    import json
    import transformers as tf
    torch.manual_seed(0)
    cfg = tf.Qwen2Config(vocab_size=300, hidden_size=2 * head_dim, intermediate_size=256, num_hidden_layers=2,
                         num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=4096,
                         tie_word_embeddings=False)
    hf = tf.Qwen2ForCausalLM(cfg).float().eval()
    path = os.path.join(tmp_path, "q")
    hf.save_pretrained(path, safe_serialization=True)
    c = json.load(open(os.path.join(path, "config.json")))
    c["dual_chunk_attention_config"] = {"chunk_size": 64, "local_size": 16, "original_max_position_embeddings": orig_max}
    json.dump(c, open(os.path.join(path, "config.json"), "w"))
    return path, hf


@pytest.mark.parametrize("orig_max,chunked", [(0, False), (0, True), (40, False)])
def test_dual_chunk_attention_matches_dense_definition(tmp_path, orig_max, chunked):
    path, hf = save_tiny_dual_chunk(tmp_path, orig_max)
    g = torch.Generator().manual_seed(1)
    prompts = [torch.randint(3, 300, (n,), generator=g).tolist() for n in (150, 37, 101)]
    kw = dict(enable_chunked_prefill=True, max_num_batched_tokens=64) if chunked else {}
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=512, max_num_seqs=4,
              enable_prefix_caching=False, **kw)
    from vllm_distributed_amd.layers.attention import DualChunkAttention
    mr = llm.llm_engine.engine_core.engine_core.model_executor.worker.model_runner
    assert mr.dual_chunk and all(isinstance(m, DualChunkAttention) for m in mr.attn_layers.values())
    outs = llm.generate([{"prompt_token_ids": p} for p in prompts],
                        SamplingParams(temperature=0, max_tokens=14, ignore_eos=True))
    for p, o in zip(prompts, outs):
        ids = list(p)
        with torch.no_grad():
            for tok in o.outputs[0].token_ids:
                lg = dense_dual_chunk_logits(hf, ids, 64, 16, orig_max, len(p))
                top = lg.topk(2)
                # a near tie in the dense form may legitimately fall the other way
                assert tok == int(top.indices[0]) or float(top.values[0] - lg[tok]) < 1e-3, (len(ids), tok, top)
                ids.append(tok)
    llm.shutdown()
