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


def _sparse_visibility(q, k, L, chunk_size, inv, P, budget):
    """bool [H, P, P]: which keys the prompt rows see under vertical-slash selection when the whole prompt of P tokens
    is prefilled in one step.  Written from the definition, row by row: per chunk of query rows, the last ``last_q``
    rows are scored against every key before them (rotation by how many chunks lie between), the softmax is summed
    per key and per distance, the best lines are kept (first 30 keys / nearest 100 distances always), and a tile of
    128 rows (counted from the chunk's first row) sees the 32-key steps -- counted from the first key of the own
    chunk, of the chunk before, or of the sequence for anything older -- that a kept line touches."""
    vsize, ssize, last_q = budget
    H, D = q.shape[1], q.shape[2]
    vis = torch.zeros((H, P, P), dtype=torch.bool)
    for c in range((P - 1) // L + 1):
        r0, r1 = c * L, min((c + 1) * L, P)
        probe = list(range(max(r0, r1 - last_q), r1))
        for h in range(H):
            vert, slash = torch.zeros(r1), torch.zeros(r1)
            for i in probe:
                gap = c - torch.arange(r1) // L
                sc = torch.full((r1,), float("-inf"))
                for kind, shift in ((0, 0), (1, L), (2, chunk_size)):
                    qi = _rot(q[i:i + 1], torch.tensor([i % L + shift]), inv)[0, h]
                    hit = ((gap == kind) if kind < 2 else (gap >= 2)) & (torch.arange(r1) <= i)
                    sc[hit] = (k[:r1, h][hit] @ qi) / math.sqrt(D)
                pr = sc.softmax(-1)
                vert += pr
                for j in range(i + 1):
                    slash[i - j] += pr[j]
            vert[:30] = float("inf")
            slash[:100] = float("inf")
            vk = vert.topk(min(vsize, r1)).indices.tolist()
            sk = slash.topk(min(ssize, r1)).indices.tolist()
            for t0 in range(r0, r1, 128):
                t1 = min(t0 + 128, r1) - 1
                for base, lo, hi in ((c * L, c * L, r1), ((c - 1) * L, (c - 1) * L, c * L), (0, 0, (c - 1) * L)):
                    if hi <= max(lo, 0):
                        continue
                    steps = {(v - base) // 32 for v in vk if lo <= v < hi}
                    for d in sk + ([0] if base == c * L else []):
                        a, b = max(t0 - d, lo), min(t1 - d, hi - 1)
                        steps |= set(range((a - base) // 32, (b - base) // 32 + 1)) if a <= b else set()
                    for j in range(lo, hi):
                        if (j - base) // 32 in steps:
                            vis[h, t0:t1 + 1, j] = True
    return vis


def dense_dual_chunk_logits(hf, ids, chunk_size, local_size, orig_max, prompt_len, sparse_budget=None):
    """Logits at the last position of ``ids`` with dual-chunk attention in every layer.  ``sparse_budget``
    (vertical lines, slash lines, probe rows): the prompt rows see only their vertical-slash selection."""
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
        if sparse_budget is not None:
            vis = _sparse_visibility(q, k, L, chunk_size, inv, prompt_len, sparse_budget)
            s[:, :prompt_len, :prompt_len] = s[:, :prompt_len, :prompt_len].masked_fill(~vis, float("-inf"))
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


def _with_sparse(path, vertical, slash, sidecar):
    """Add a vertical-slash budget for every head of both layers (as the checkpoint's side file, or inline)."""
    import json
    rows = [{str(h): ["vertical_and_slash", vertical, slash, 0.9] for h in range(2)} for _ in range(2)]
    c = json.load(open(os.path.join(path, "config.json")))
    c["dual_chunk_attention_config"].update(sparse_attention_threshold=40, sparse_attention_last_q=8)
    if sidecar:
        json.dump(rows, open(os.path.join(path, "sparse_attention_config.json"), "w"))
    else:
        c["dual_chunk_attention_config"]["sparse_attention_config"] = rows
    json.dump(c, open(os.path.join(path, "config.json"), "w"))


@pytest.mark.parametrize("chunked,sidecar", [(False, True), (True, False)])
def test_sparse_prefill_with_full_budget_equals_dense(tmp_path, chunked, sidecar):
    """Budgets that cover every key: the sparse passes (estimate, line lists, step selection, three merges) must
    reproduce dense dual-chunk attention exactly.  One request per step, or the step stays dense."""
    path, hf = save_tiny_dual_chunk(tmp_path, 0)
    _with_sparse(path, 4096, 4096, sidecar)
    g = torch.Generator().manual_seed(2)
    prompts = [torch.randint(3, 300, (n,), generator=g).tolist() for n in (150, 230)]
    kw = dict(enable_chunked_prefill=True, max_num_batched_tokens=64) if chunked else {}
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=512, max_num_seqs=1,
              enable_prefix_caching=False, **kw)
    from vllm_distributed_amd import ops
    calls = []
    real = ops.paged_attention_prefill_sparse
    ops.paged_attention_prefill_sparse = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        outs = llm.generate([{"prompt_token_ids": p} for p in prompts],
                            SamplingParams(temperature=0, max_tokens=6, ignore_eos=True))
    finally:
        ops.paged_attention_prefill_sparse = real
    assert calls, "the sparse passes never ran"
    for p, o in zip(prompts, outs):
        ids = list(p)
        with torch.no_grad():
            for tok in o.outputs[0].token_ids:
                lg = dense_dual_chunk_logits(hf, ids, 64, 16, 0, len(p))
                top = lg.topk(2)
                assert tok == int(top.indices[0]) or float(top.values[0] - lg[tok]) < 1e-3, (len(ids), tok, top)
                ids.append(tok)
    llm.shutdown()


def test_step_selection_by_hand():
    """One head, 300 keys, 200 query rows at positions 100..299: vertical key 70 -> step 2 for both tiles; slash
    distance 150 -> for tile 0 (rows 100..227) keys -50..77 -> steps 0..2, for tile 1 (rows 228..299) keys 78..149
    -> steps 2..4; the causal diagonal: tile 0 keys 100..227 -> steps 3..7, tile 1 keys 228..299 -> steps 7..9."""
    from vllm_distributed_amd.ops.reference import sparse_step_selection
    i32 = lambda *v: torch.tensor(v, dtype=torch.int32)                       # noqa: E731
    sel = sparse_step_selection(i32(70, 0).view(1, 2), i32(1), i32(150, 0).view(1, 2), i32(1), 0, 100, 200, 300, True)
    assert sel.shape == (1, 2, 10)
    assert sel[0, 0].nonzero().flatten().tolist() == [0, 1, 2, 3, 4, 5, 6, 7]
    assert sel[0, 1].nonzero().flatten().tolist() == [2, 3, 4, 7, 8, 9]
    # a table that starts at absolute key 64 (the chunk before): the same lines, shifted by two steps and clipped
    sel = sparse_step_selection(i32(70, 0).view(1, 2), i32(1), i32(150, 0).view(1, 2), i32(1), 64, 100, 200, 96, False)
    assert sel[0, 0].nonzero().flatten().tolist() == [0]                      # key 70 -> 6; keys -50..77 -> ..13
    assert sel[0, 1].nonzero().flatten().tolist() == [0, 1, 2]                # keys 78..149 -> 14..85


def test_sparse_prefill_with_small_budget_matches_written_out_selection(tmp_path):
    """34 vertical and 104 slash lines (30 / 100 of them forced) on a 500-token prompt prefilled in one step: the
    engine's tokens against dense attention under the visibility pattern computed row by row from the definition."""
    path, hf = save_tiny_dual_chunk(tmp_path, 0)
    _with_sparse(path, 34, 104, False)
    g = torch.Generator().manual_seed(3)
    p = torch.randint(3, 300, (500,), generator=g).tolist()
    llm = LLM(model=path, skip_tokenizer_init=True, dtype="float32", device="cpu", max_model_len=512, max_num_seqs=1,
              enable_prefix_caching=False, max_num_batched_tokens=512)
    out = llm.generate([{"prompt_token_ids": p}], SamplingParams(temperature=0, max_tokens=3, ignore_eos=True))[0]
    llm.shutdown()
    ids, differs = list(p), False
    with torch.no_grad():
        for tok in out.outputs[0].token_ids:
            lg = dense_dual_chunk_logits(hf, ids, 64, 16, 0, len(p), sparse_budget=(34, 104, 8))
            top = lg.topk(2)
            assert tok == int(top.indices[0]) or float(top.values[0] - lg[tok]) < 1e-3, (len(ids), tok, top)
            differs |= float((lg - dense_dual_chunk_logits(hf, ids, 64, 16, 0, len(p))).abs().max()) > 1e-4
            ids.append(tok)
    assert differs, "the budget left nothing out: the test would not see a wrong selection"
