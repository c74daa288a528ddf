"""HIP kernels vs the plain PyTorch fp32 references (ops/reference.py), in the
style of the reference's tests/kernels/*: dtype x shape grids, fixed seeds."""
import math

import pytest
import torch

from vllm_distributed_amd import ops
from vllm_distributed_amd.ops import reference as R

pytestmark = pytest.mark.gpu
DEV = "cuda"
DTYPES = [torch.bfloat16, torch.float16]


def tol(dt):
    return dict(atol=2e-2, rtol=2e-2) if dt == torch.bfloat16 else dict(atol=4e-3, rtol=4e-3)


@pytest.fixture(autouse=True)
def _seed():
    torch.manual_seed(0)


def test_native_library_is_loaded():
    from vllm_distributed_amd.ops import native
    assert native.is_available(), "libvda_ops.so must be built in-tree"
    native.lib()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("T,H", [(1, 4096), (7, 8192), (83, 768), (5, 5120), (3, 16384)])
@pytest.mark.parametrize("add", [False, True])
def test_rms_norm(dt, T, H, add):
    x = torch.randn(T, H, device=DEV, dtype=dt)
    w = (torch.randn(H, device=DEV, dtype=dt) * 0.1 + 1)
    res = torch.randn(T, H, device=DEV, dtype=dt) if add else None
    res_ref = res.clone() if add else None
    y = ops.rms_norm(x, w, 1e-5, res)
    y_ref = R.rms_norm(x, w, 1e-5, res_ref)
    torch.testing.assert_close(y.float(), y_ref.float(), **tol(dt))
    if add:
        torch.testing.assert_close(res.float(), res_ref.float(), **tol(dt))


@pytest.mark.parametrize("dt", DTYPES)
def test_rms_norm_fp8_dynamic_and_static(dt):
    x = torch.randn(33, 4096, device=DEV, dtype=dt)
    w = torch.ones(4096, device=DEV, dtype=dt)
    q, s = ops.rms_norm_fp8(x, w, 1e-6)
    y = R.rms_norm(x, w, 1e-6)
    qr, sr = R.per_token_fp8_quant(y)
    torch.testing.assert_close(s.view(-1), sr.view(-1), rtol=1e-3, atol=1e-6)
    torch.testing.assert_close(q.float() * s, qr.float() * sr, atol=0.15, rtol=0.07)
    st = torch.tensor([0.02], device=DEV)
    q2, _ = ops.rms_norm_fp8(x, w, 1e-6, static_scale=st)
    torch.testing.assert_close(q2.float() * st, y.float().clamp(-448 * 0.02, 448 * 0.02), atol=0.1, rtol=0.07)


@pytest.mark.parametrize("dt", DTYPES)
def test_layer_norm(dt):
    x = torch.randn(19, 768, device=DEV, dtype=dt)
    w, b = torch.randn(768, device=DEV, dtype=dt), torch.randn(768, device=DEV, dtype=dt)
    torch.testing.assert_close(ops.layer_norm(x, w, b, 1e-5).float(), R.layer_norm(x, w, b, 1e-5).float(), **tol(dt))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("kind", ["silu", "gelu", "gelu_tanh", "fatrelu"])
def test_act_and_mul(dt, kind):
    x = torch.randn(37, 2 * 1536, device=DEV, dtype=dt)
    torch.testing.assert_close(ops.act_and_mul(x, kind, threshold=0.1).float(),
                               R.act_and_mul(x, kind, threshold=0.1).float(), **tol(dt))
    torch.testing.assert_close(ops.act_and_mul(x, "silu", act_first=False).float(),
                               R.act_and_mul(x, "silu", act_first=False).float(), **tol(dt))


@pytest.mark.parametrize("kind", ["gelu_new", "gelu_fast", "quick_gelu", "relu", "relu2"])
def test_plain_activation(kind):
    x = torch.randn(64, 3072, device=DEV, dtype=torch.float16)
    torch.testing.assert_close(ops.activation(x, kind).float(), R.activation(x, kind).float(), atol=4e-3, rtol=4e-3)


def _cos_sin(max_pos, rot, base=10000.0):
    inv = 1.0 / (base ** (torch.arange(0, rot, 2, dtype=torch.float32) / rot))
    f = torch.outer(torch.arange(max_pos, dtype=torch.float32), inv)
    return torch.cat((f.cos(), f.sin()), dim=-1).to(DEV)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("neox", [True, False])
@pytest.mark.parametrize("D,rot", [(128, 128), (64, 64), (128, 64)])
def test_rotary(dt, neox, D, rot):
    T, Hq, Hkv = 45, 8, 2
    cs = _cos_sin(512, rot)
    pos = torch.randint(0, 512, (T,), device=DEV)
    q = torch.randn(T, Hq * D, device=DEV, dtype=dt)
    k = torch.randn(T, Hkv * D, device=DEV, dtype=dt)
    qr, kr = q.clone(), k.clone()
    ops.rotary_embedding(pos, q, k, cs, rot, D, neox)
    R.rotary_embedding(pos, qr, kr, cs, rot, D, neox)
    torch.testing.assert_close(q.float(), qr.float(), **tol(dt))
    torch.testing.assert_close(k.float(), kr.float(), **tol(dt))


def _make_cache(nblk, Hkv, bs, D, dt):
    kc = torch.zeros(nblk, Hkv, bs, D, device=DEV, dtype=dt)
    vc = torch.zeros(nblk, Hkv, D, bs, device=DEV, dtype=dt)
    return kc, vc


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("bs", [16, 32, 64])
def test_reshape_and_cache_and_gather(dt, bs):
    T, Hkv, D, nblk = 70, 4, 128, 20
    kc, vc = _make_cache(nblk, Hkv, bs, D, dt)
    kr, vr = kc.clone(), vc.clone()
    k = torch.randn(T, Hkv, D, device=DEV, dtype=dt)
    v = torch.randn(T, Hkv, D, device=DEV, dtype=dt)
    slots = torch.randperm(nblk * bs, device=DEV)[:T]
    slots[3] = -1
    ops.reshape_and_cache(k, v, kc, vc, slots)
    R.reshape_and_cache(k, v, kr, vr, slots)
    assert torch.equal(kc, kr) and torch.equal(vc, vr)
    good = slots[slots >= 0]
    kg, vg = ops.gather_cache(kc, vc, good)
    assert torch.equal(kg, k[slots >= 0]) and torch.equal(vg, v[slots >= 0])


@pytest.mark.parametrize("dt", DTYPES)
def test_rope_and_cache_fused(dt):
    T, Hq, Hkv, D, bs, nblk = 50, 8, 2, 128, 16, 8
    cs = _cos_sin(256, D)
    pos = torch.randint(0, 256, (T,), device=DEV)
    qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device=DEV, dtype=dt)
    ref = qkv.clone()
    kc, vc = _make_cache(nblk, Hkv, bs, D, dt)
    kr, vr = kc.clone(), vc.clone()
    slots = torch.randperm(nblk * bs, device=DEV)[:T]
    ops.rope_and_cache(pos, qkv, cs, kc, vc, slots, D, D, Hq, Hkv, True)
    q, k, v = ref[:, :Hq * D], ref[:, Hq * D:(Hq + Hkv) * D], ref[:, (Hq + Hkv) * D:]
    R.rotary_embedding(pos, q, k, cs, D, D, True)
    R.reshape_and_cache(k.reshape(T, Hkv, D), v.reshape(T, Hkv, D), kr, vr, slots)
    torch.testing.assert_close(qkv[:, :Hq * D].float(), q.float(), **tol(dt))
    torch.testing.assert_close(kc.float(), kr.float(), **tol(dt))
    assert torch.equal(vc, vr)


def _paged_problem(seq_lens, q_lens, Hq, Hkv, D, bs, dt, scale_in=1.0):
    S = len(seq_lens)
    maxb = max((s + bs - 1) // bs for s in seq_lens) + 1
    nblk = sum((s + bs - 1) // bs for s in seq_lens) + 4
    kc = (torch.randn(nblk, Hkv, bs, D, device=DEV, dtype=dt) * scale_in)
    vc = torch.randn(nblk, Hkv, D, bs, device=DEV, dtype=dt)
    perm = torch.randperm(nblk - 1, device=DEV) + 1
    bt = torch.zeros(S, maxb, dtype=torch.int32, device=DEV)
    o = 0
    for i, s in enumerate(seq_lens):
        n = (s + bs - 1) // bs
        bt[i, :n] = perm[o:o + n].int()
        o += n
    T = sum(q_lens)
    q = torch.randn(T, Hq, D, device=DEV, dtype=dt) * scale_in
    qsl = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0)), dtype=torch.int32, device=DEV)
    sl = torch.tensor(seq_lens, dtype=torch.int32, device=DEV)
    return q, kc, vc, bt, sl, qsl


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (8, 1, 128), (12, 12, 64), (28, 4, 128), (16, 2, 256), (32, 32, 96),
                                       (12, 4, 96), (16, 2, 192)])
@pytest.mark.parametrize("bs", [16, 32, 64])
@pytest.mark.parametrize("nsplit", [1, 4])
def test_paged_attention_decode(dt, Hq, Hkv, D, bs, nsplit):
    seq_lens = [1, 17, 300, 1025, 64, 33]
    q, kc, vc, bt, sl, qsl = _paged_problem(seq_lens, [1] * len(seq_lens), Hq, Hkv, D, bs, dt)
    scale = 1.0 / math.sqrt(D)
    out = torch.empty_like(q)
    ops.paged_attention_decode(out, q, kc, vc, bt, sl, qsl, scale, num_splits=nsplit)
    ref = R.paged_attention(q, kc, vc, bt, sl, qsl, scale)
    torch.testing.assert_close(out.float(), ref.float(), **tol(dt))


@pytest.mark.parametrize("dt", DTYPES)
def test_paged_attention_decode_window_softcap_alibi_multiq(dt):
    Hq, Hkv, D, bs = 8, 2, 128, 16
    seq_lens, q_lens = [40, 700, 130], [3, 4, 1]
    q, kc, vc, bt, sl, qsl = _paged_problem(seq_lens, q_lens, Hq, Hkv, D, bs, dt)
    scale = 1.0 / math.sqrt(D)
    alibi = torch.tensor([2 ** (-(i + 1)) for i in range(Hq)], device=DEV, dtype=torch.float32)
    for kw in (dict(window=100), dict(softcap=30.0), dict(alibi=alibi), dict()):
        out = torch.empty_like(q)
        ops.paged_attention_decode(out, q, kc, vc, bt, sl, qsl, scale, num_splits=2, max_q_len=4, **kw)
        ref = R.paged_attention(q, kc, vc, bt, sl, qsl, scale, **kw)
        torch.testing.assert_close(out.float(), ref.float(), **tol(dt))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (8, 1, 128), (12, 12, 64), (28, 4, 128), (32, 32, 96), (12, 4, 96),
                                       (16, 2, 192)])
@pytest.mark.parametrize("bs", [16, 32])
def test_paged_attention_prefill(dt, Hq, Hkv, D, bs):
    # mix of full prefills, chunked prefills with cached context, and single-token rows
    seq_lens = [5, 130, 300, 77, 1, 257]
    q_lens = [5, 130, 44, 77, 1, 100]
    q, kc, vc, bt, sl, qsl = _paged_problem(seq_lens, q_lens, Hq, Hkv, D, bs, dt)
    scale = 1.0 / math.sqrt(D)
    out = torch.full_like(q, float("nan"))
    ops.paged_attention_prefill(out, q, kc, vc, bt, sl, qsl, scale)
    ref = R.paged_attention(q, kc, vc, bt, sl, qsl, scale)
    torch.testing.assert_close(out.float(), ref.float(), **tol(dt))


def test_paged_attention_prefill_window():
    dt, Hq, Hkv, D, bs = torch.bfloat16, 8, 2, 128, 16
    q, kc, vc, bt, sl, qsl = _paged_problem([400, 90], [400, 30], Hq, Hkv, D, bs, dt)
    out = torch.empty_like(q)
    ops.paged_attention_prefill(out, q, kc, vc, bt, sl, qsl, 0.09, window=64)
    ref = R.paged_attention(q, kc, vc, bt, sl, qsl, 0.09, window=64)
    torch.testing.assert_close(out.float(), ref.float(), **tol(dt))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("Hq,Hkv,D", [(8, 2, 128), (12, 12, 64), (4, 1, 256), (8, 2, 96), (4, 1, 192)])
def test_paged_attention_prefill_bidirectional_and_lse(dt, Hq, Hkv, D):
    seq_lens = [5, 130, 300, 33]
    q, kc, vc, bt, sl, qsl = _paged_problem(seq_lens, seq_lens, Hq, Hkv, D, 16, dt)
    scale = 1.0 / math.sqrt(D)
    for causal in (False, True):
        out = torch.full_like(q, float("nan"))
        _, lse = ops.paged_attention_prefill(out, q, kc, vc, bt, sl, qsl, scale, causal=causal, return_lse=True)
        ref, lref = R.paged_attention(q, kc, vc, bt, sl, qsl, scale, return_lse=True, causal=causal)
        torch.testing.assert_close(out.float(), ref.float(), **tol(dt))
        torch.testing.assert_close(lse, lref, atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("Hq,Hkv,D,bs", [(8, 2, 128, 16), (4, 4, 64, 32), (14, 2, 128, 64)])
@pytest.mark.parametrize("causal", [True, False])
def test_paged_attention_prefill_sparse(dt, Hq, Hkv, D, bs, causal):
    """Vertical-slash sparse prefill: LDS bitmap of 32-key steps per (128-row tile, head) against the masked fp32
    form; sequences whose tables start at different absolute keys, partial tiles, heads with empty lists."""
    seq_lens, q_lens = [700, 300, 1200, 40], [700, 130, 257, 40]
    q, kc, vc, bt, sl, qsl = _paged_problem(seq_lens, q_lens, Hq, Hkv, D, bs, dt)
    S, NV, NS = len(seq_lens), 24, 12
    g = torch.Generator(device="cpu").manual_seed(5)
    key_base = torch.tensor([0, 64, 0, 512], dtype=torch.int32)
    ctx = torch.tensor([s - ql for s, ql in zip(seq_lens, q_lens)], dtype=torch.int32)
    q_base = key_base + ctx if causal else key_base + torch.tensor(seq_lens, dtype=torch.int32) + 100
    vert = torch.randint(0, 1400, (S, Hq, NV), generator=g).int()
    slash = torch.randint(0, 1400, (S, Hq, NS), generator=g).int()
    vcount = torch.randint(0, NV + 1, (S, Hq), generator=g).int()
    scount = torch.randint(0, NS + 1, (S, Hq), generator=g).int()
    vcount[0, 0] = scount[0, 0] = 0                     # a head that sees only its diagonal (or nothing)
    args = [t.to(DEV) for t in (vert, vcount, slash, scount, key_base, q_base)]
    scale = 1.0 / math.sqrt(D)
    out = torch.full_like(q, float("nan"))
    _, lse = ops.paged_attention_prefill_sparse(out, q, kc, vc, bt, sl, qsl, scale, *args, causal=causal)
    ref, lref = R.paged_attention_sparse(q, kc, vc, bt, sl, qsl, scale, *args, causal=causal)
    torch.testing.assert_close(out.float(), ref.float(), **tol(dt))
    seen = ~torch.isinf(lref)
    assert bool(seen.any()) and torch.equal(seen, ~torch.isinf(lse))
    torch.testing.assert_close(lse[seen], lref[seen], atol=2e-2, rtol=2e-2)
    if causal:                                          # the selection really leaves keys out
        dense = R.paged_attention(q, kc, vc, bt, sl, qsl, scale)
        assert float((dense.float() - ref.float()).abs().max()) > 0.05


def test_paged_attention_prefill_sparse_fp8_cache():
    dt, Hq, Hkv, D, bs = torch.bfloat16, 8, 2, 128, 16
    seq_lens, q_lens = [900, 300], [900, 140]
    q, kc, vc, bt, sl, qsl = _paged_problem(seq_lens, q_lens, Hq, Hkv, D, bs, dt)
    kc, vc = (kc / 0.05).to(torch.float8_e4m3fn), (vc / 0.02).to(torch.float8_e4m3fn)
    g = torch.Generator(device="cpu").manual_seed(6)
    vert = torch.randint(0, 900, (2, Hq, 16), generator=g).int().to(DEV)
    slash = torch.randint(0, 900, (2, Hq, 8), generator=g).int().to(DEV)
    vcount = torch.full((2, Hq), 16, dtype=torch.int32, device=DEV)
    scount = torch.full((2, Hq), 8, dtype=torch.int32, device=DEV)
    key_base = torch.zeros(2, dtype=torch.int32, device=DEV)
    q_base = torch.tensor([0, 160], dtype=torch.int32, device=DEV)
    args = (vert, vcount, slash, scount, key_base, q_base)
    out = torch.full_like(q, float("nan"))
    ops.paged_attention_prefill_sparse(out, q, kc, vc, bt, sl, qsl, 0.09, *args, k_scale=0.05, v_scale=0.02)
    ref, _ = R.paged_attention_sparse(q, kc, vc, bt, sl, qsl, 0.09, *args, k_scale=0.05, v_scale=0.02)
    torch.testing.assert_close(out.float(), ref.float(), **tol(dt))


def test_cascade_composition_equals_plain_attention():
    """Shared prefix of P keys attended once by every query token (one bidirectional pseudo-sequence),
    per-sequence suffixes through a block table that starts behind the prefix, merged by log-sum-exp:
    the result must equal ordinary causal attention over prefix + suffix."""
    dt, Hq, Hkv, D, bs = torch.bfloat16, 8, 2, 128, 16
    P = 10 * bs
    seq_lens, q_lens = [P + 40, P + 1, P + 300, P + 17], [40, 1, 100, 1]
    q, kc, vc, bt, sl, qsl = _paged_problem(seq_lens, q_lens, Hq, Hkv, D, bs, dt)
    bt[:, :P // bs] = bt[0, :P // bs]            # every sequence shares the first P // bs blocks
    scale = 1.0 / math.sqrt(D)
    ref = R.paged_attention(q, kc, vc, bt, sl, qsl, scale)
    T = q.shape[0]
    pre = torch.empty_like(q)
    one = torch.tensor([0, T], dtype=torch.int32, device=DEV)
    _, pre_lse = ops.paged_attention_prefill(pre, q, kc, vc, bt[:1], torch.tensor([P], dtype=torch.int32, device=DEV),
                                             one, scale, causal=False, return_lse=True)
    suf = torch.empty_like(q)
    _, suf_lse = ops.paged_attention_prefill(suf, q, kc, vc, bt[:, P // bs:], sl - P, qsl, scale, return_lse=True)
    out = ops.merge_attn_states(pre, pre_lse, suf, suf_lse)
    torch.testing.assert_close(out.float(), ref.float(), **tol(dt))


def test_attention_online_softmax_rescale_path():
    """One key far larger than the rest late in the sequence forces the running-max rescale."""
    dt, Hq, Hkv, D, bs = torch.bfloat16, 8, 2, 128, 16
    q, kc, vc, bt, sl, qsl = _paged_problem([513], [1], Hq, Hkv, D, bs, dt)
    blk, off = int(bt[0, 480 // bs]), 480 % bs
    kc[blk, :, off, :] = q[0, ::4, :] * 6.0
    out = torch.empty_like(q)
    ops.paged_attention_decode(out, q, kc, vc, bt, sl, qsl, 0.1)
    ref = R.paged_attention(q, kc, vc, bt, sl, qsl, 0.1)
    torch.testing.assert_close(out.float(), ref.float(), **tol(dt))


def test_merge_attn_states():
    T, H, D = 33, 8, 128
    a = torch.randn(T, H, D, device=DEV, dtype=torch.bfloat16)
    b = torch.randn(T, H, D, device=DEV, dtype=torch.bfloat16)
    la, lb = torch.randn(H, T, device=DEV), torch.randn(H, T, device=DEV)
    lb[0, 0] = float("-inf")
    o, l = ops.merge_attn_states(a, la, b, lb, return_lse=True)
    orf, lrf = R.merge_attn_states(a, la, b, lb)
    torch.testing.assert_close(o.float(), orf.float(), atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(l, lrf, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_argmax(dt):
    x = torch.randn(9, 128256, device=DEV, dtype=dt)
    assert torch.equal(ops.argmax(x), x.float().argmax(-1)) or dt != torch.float32
    y = x.float()
    assert torch.equal(y.gather(1, ops.argmax(x)[:, None]), y.max(-1, keepdim=True).values)


def test_sample_greedy_rows_and_filters():
    S, V = 6, 32000
    logits = torch.randn(S, V, device=DEV) * 3
    temp = torch.tensor([0.0, 1.0, 0.7, 1.0, 1.5, 0.0], device=DEV)
    topk = torch.tensor([0, 1, 5, 0, 50, 0], dtype=torch.int32, device=DEV)
    topp = torch.tensor([1.0, 1.0, 1.0, 0.3, 0.9, 1.0], device=DEV)
    seeds = torch.arange(S, device=DEV)
    tok = ops.sample(logits, temp, topk, topp, None, seeds)
    am = logits.argmax(-1)
    assert tok[0] == am[0] and tok[5] == am[5] and tok[1] == am[1]   # greedy rows and top_k=1
    assert tok[2] in logits[2].topk(5).indices
    # row 3: token must be in the top-p=0.3 nucleus
    p = torch.softmax(logits[3], -1)
    sp, si = p.sort(descending=True)
    nucleus = si[: int(((sp.cumsum(0) - sp) < 0.3).sum())]
    assert tok[3] in nucleus
    # same seed -> same draw
    assert torch.equal(tok, ops.sample(logits, temp, topk, topp, None, seeds))


def test_sample_distribution_matches_softmax():
    V, n = 64, 40000
    logits = torch.randn(1, V, device=DEV).repeat(n, 1)
    seeds = torch.arange(n, device=DEV) * 7919 + 13
    tok = ops.sample(logits, torch.ones(n, device=DEV), None, None, None, seeds)
    emp = torch.bincount(tok, minlength=V).float() / n
    torch.testing.assert_close(emp, torch.softmax(logits[0], -1), atol=0.012, rtol=0.0)


def test_topk_topp_mask_matches_sort_reference():
    S, V = 8, 50000
    logits = torch.randn(S, V, device=DEV) * 4
    k = torch.tensor([0, 1, 10, 100, 0, 40, 7, 0], dtype=torch.int32, device=DEV)
    p = torch.tensor([1.0, 1.0, 0.5, 0.9, 0.2, 1.0, 0.99, 0.7], device=DEV)
    mp = torch.tensor([0, 0, 0, 0, 0, 0.05, 0, 0.01], device=DEV)
    a = ops.topk_topp_mask(logits.clone(), k, p, mp)
    b = R.topk_topp_mask(logits.clone(), k, p, mp)
    keep_a, keep_b = ~torch.isinf(a), ~torch.isinf(b)
    # float summation order may move the top-p boundary by one token
    assert (keep_a != keep_b).sum(dim=1).max() <= 1
    assert torch.equal(a[keep_a & keep_b], logits[keep_a & keep_b])


def test_top_p_threshold_is_repeatable():
    """Flat distributions with many near-boundary tokens: the kept set and the seeded draw must be the same
    on every run (the histogram is summed in fixed point, independent of atomic arrival order)."""
    S, V = 16, 128256
    logits = torch.randn(S, V, device=DEV) * 0.3
    p = torch.linspace(0.05, 0.95, S, device=DEV)
    k = torch.zeros(S, dtype=torch.int32, device=DEV)
    first = ops.topk_topp_mask(logits.clone(), k, p, None)
    seeds = torch.arange(S, device=DEV) * 31 + 5
    temp = torch.ones(S, device=DEV)
    tok = ops.sample(logits, temp, k, p, None, seeds)
    for _ in range(20):
        assert torch.equal(torch.isinf(ops.topk_topp_mask(logits.clone(), k, p, None)), torch.isinf(first))
        assert torch.equal(ops.sample(logits, temp, k, p, None, seeds), tok)
    # and the kept mass is the requested fraction (within one boundary token)
    probs = torch.softmax(logits, -1)
    kept = (probs * (~torch.isinf(first))).sum(-1)
    assert bool(((kept >= p - 1e-4) & (kept <= p + probs.max(-1).values + 1e-4)).all()), (kept, p)


def test_penalties():
    S, V = 4, 1000
    logits = torch.randn(S, V, device=DEV)
    pm = torch.rand(S, V, device=DEV) < 0.05
    oc = (torch.rand(S, V, device=DEV) < 0.03).int() * torch.randint(1, 4, (S, V), device=DEV).int()
    rep = torch.tensor([1.0, 1.3, 0.8, 1.1], device=DEV)
    fr = torch.tensor([0.0, 0.5, 0.0, -0.2], device=DEV)
    pr = torch.tensor([0.0, 0.0, 1.0, 0.3], device=DEV)
    a = ops.apply_penalties(logits.clone(), pm, oc, rep, fr, pr)
    b = R.apply_penalties(logits.clone(), pm, oc, rep, fr, pr)
    torch.testing.assert_close(a, b, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("dt", DTYPES)
def test_fp8_quant_roundtrip(dt):
    x = torch.randn(64, 4096, device=DEV, dtype=dt) * 3
    q, s = ops.scaled_fp8_quant(x)
    qr, sr = R.scaled_fp8_quant(x)
    torch.testing.assert_close(s, sr, rtol=1e-3, atol=0)
    torch.testing.assert_close(q.float(), qr.float(), atol=0, rtol=0.13)
    torch.testing.assert_close(ops.fp8_dequant(q, s, dt).float(), x.float(), atol=0.2, rtol=0.07)
    q2, s2 = ops.per_token_fp8_quant(x)
    q2r, s2r = R.per_token_fp8_quant(x)
    torch.testing.assert_close(s2, s2r, rtol=1e-3, atol=0)
    torch.testing.assert_close(q2.float() * s2, x.float(), atol=0.2, rtol=0.07)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M", [1, 3, 8, 16, 17, 48, 64])
@pytest.mark.parametrize("N,K", [(4096, 4096), (1024, 14336), (6144, 4096), (256, 768)])
def test_skinny_gemm(dt, M, N, K):
    x = torch.randn(M, K, device=DEV, dtype=dt) * 0.5
    w = torch.randn(N, K, device=DEV, dtype=dt) * 0.05
    b = torch.randn(N, device=DEV, dtype=dt)
    y = ops.skinny_gemm(x, w, b)
    ref = (x.float() @ w.float().T + b.float())
    torch.testing.assert_close(y.float(), ref, atol=3e-2 if dt == torch.bfloat16 else 8e-3, rtol=2e-2)


def test_copy_and_transfer_blocks():
    c = torch.randn(10, 4, 16, 128, device=DEV, dtype=torch.bfloat16)
    ref = c.clone()
    pairs = torch.tensor([[1, 5], [2, 7]], device=DEV)
    ops.copy_blocks(c, pairs)
    ref[5], ref[7] = ref[1].clone(), ref[2].clone()
    assert torch.equal(c, ref)
    dst = torch.zeros_like(c)
    ops.transfer_blocks(dst, c, torch.tensor([0, 3], device=DEV), torch.tensor([5, 9], device=DEV))
    assert torch.equal(dst[0], c[5]) and torch.equal(dst[3], c[9]) and dst[1].abs().sum() == 0


# ---- mixture of experts -------------------------------------------------------------------
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("T,E,k", [(1, 8, 2), (33, 8, 2), (257, 128, 8), (5, 60, 4), (64, 256, 6)])
@pytest.mark.parametrize("renorm", [True, False])
def test_moe_topk_softmax(dt, T, E, k, renorm):
    lg = torch.randn(T, E, device=DEV, dtype=dt)
    w, ids = ops.moe_topk_softmax(lg, k, renorm)
    w_ref, ids_ref = R.moe_topk_softmax(lg, k, renorm)
    assert torch.equal(ids.sort(dim=-1).values, ids_ref.sort(dim=-1).values)
    # same expert -> same weight, whatever the order inside the top-k
    dense = torch.zeros(T, E, device=DEV).scatter_(1, ids.long(), w)
    dense_ref = torch.zeros(T, E, device=DEV).scatter_(1, ids_ref.long(), w_ref)
    torch.testing.assert_close(dense, dense_ref, atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("T,E,k,H,I", [(1, 8, 2, 512, 256), (17, 8, 2, 1024, 512), (130, 8, 2, 512, 384),
                                       (64, 64, 6, 256, 128), (300, 4, 2, 256, 128)])
def test_fused_moe_grouped(dt, T, E, k, H, I):
    x = torch.randn(T, H, device=DEV, dtype=dt) * 0.5
    w13 = torch.randn(E, 2 * I, H, device=DEV, dtype=dt) * (H ** -0.5)
    w2 = torch.randn(E, H, I, device=DEV, dtype=dt) * (I ** -0.5)
    tw, ids = R.moe_topk_softmax(torch.randn(T, E, device=DEV), k, True)
    out = ops.fused_moe(x, w13, w2, tw, ids)
    ref = R.fused_moe(x, w13, w2, tw, ids)
    torch.testing.assert_close(out.float(), ref.float(), **tol(dt))


@pytest.mark.parametrize("form", [0, 1, 2])
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("T,E,k,H,I", [(5, 8, 2, 512, 256), (130, 8, 2, 512, 384), (300, 4, 2, 256, 128),
                                       (200, 2, 1, 384, 192)])
def test_fused_moe_every_kernel_form(monkeypatch, form, fp8, T, E, k, H, I):
    """K-split kernel, row-split kernel with 64 and with 128 tokens per pass (several passes per expert
    in the larger cases), 16-bit and e4m3 weights, rows that are and are not a multiple of 64."""
    monkeypatch.setenv("VDA_MOE_ROWS_FORM", str(form))
    dt = torch.bfloat16
    x = torch.randn(T, H, device=DEV, dtype=dt) * 0.5
    w13 = torch.randn(E, 2 * I, H, device=DEV, dtype=dt) * (H ** -0.5)
    w2 = torch.randn(E, H, I, device=DEV, dtype=dt) * (I ** -0.5)
    s13 = s2 = None
    if fp8:
        s13 = w13.float().abs().amax(dim=(1, 2)) / 448.0
        s2 = w2.float().abs().amax(dim=(1, 2)) / 448.0
        w13 = (w13.float() / s13.view(E, 1, 1)).to(torch.float8_e4m3fn)
        w2 = (w2.float() / s2.view(E, 1, 1)).to(torch.float8_e4m3fn)
    tw, ids = R.moe_topk_softmax(torch.randn(T, E, device=DEV), k, True)
    out = ops.fused_moe(x, w13, w2, tw, ids, s13, s2)
    ref = R.fused_moe(x, w13, w2, tw, ids, s13, s2)
    torch.testing.assert_close(out.float(), ref.float(), atol=3e-2, rtol=3e-2)


@pytest.mark.parametrize("activation", ["gelu", "gelu_tanh", "relu", "relu2", "silu"])
@pytest.mark.parametrize("T,tiled", [(6, False), (150, False), (700, True)])
def test_fused_moe_gate_activations(monkeypatch, activation, T, tiled):
    """The gate activation is applied inside the first expert GEMM's epilogue in every kernel form: the small-batch
    grouped kernels (K split and row split) and the LDS-tiled GEMM."""
    if tiled:
        monkeypatch.setenv("VDA_MOE_GROUPED_MAX_PAIRS", "64")
    E, k, H, I = 8, 2, 512, 256
    dt = torch.bfloat16
    x = torch.randn(T, H, device=DEV, dtype=dt) * 0.5
    w13 = torch.randn(E, 2 * I, H, device=DEV, dtype=dt) * (H ** -0.5) * 3
    w2 = torch.randn(E, H, I, device=DEV, dtype=dt) * (I ** -0.5)
    tw, ids = R.moe_topk_softmax(torch.randn(T, E, device=DEV), k, True)
    out = ops.fused_moe(x, w13, w2, tw, ids, activation=activation)
    ref = R.fused_moe(x, w13, w2, tw, ids, activation=activation)
    torch.testing.assert_close(out.float(), ref.float(), atol=3e-2, rtol=3e-2)
    if activation != "silu":          # and it is not silently the default
        silu = R.fused_moe(x, w13, w2, tw, ids)
        assert (ref.float() - silu.float()).abs().max().item() > 0.05


def test_fused_moe_expert_parallel_mask_and_large_path(monkeypatch):
    dt = torch.bfloat16
    T, E, k, H, I = 96, 8, 2, 512, 256
    x = torch.randn(T, H, device=DEV, dtype=dt) * 0.5
    w13 = torch.randn(E, 2 * I, H, device=DEV, dtype=dt) * (H ** -0.5)
    w2 = torch.randn(E, H, I, device=DEV, dtype=dt) * (I ** -0.5)
    tw, ids = R.moe_topk_softmax(torch.randn(T, E, device=DEV), k, True)
    # this "rank" owns experts 2..5 only: global -> local map with -1 elsewhere
    emap = torch.full((E,), -1, dtype=torch.int32, device=DEV)
    emap[2:6] = torch.arange(4, dtype=torch.int32, device=DEV)
    local = emap[ids.long()]
    ref = R.fused_moe(x, w13[2:6], w2[2:6], tw, local)
    out = ops.fused_moe(x, w13[2:6].contiguous(), w2[2:6].contiguous(), tw, local)
    torch.testing.assert_close(out.float(), ref.float(), **tol(dt))
    monkeypatch.setenv("VDA_MOE_GROUPED_MAX_PAIRS", "16")        # force the sorted hipBLASLt path
    out2 = ops.fused_moe(x, w13[2:6].contiguous(), w2[2:6].contiguous(), tw, local)
    torch.testing.assert_close(out2.float(), ref.float(), **tol(dt))


def test_fused_moe_fp8_weights():
    dt = torch.bfloat16
    T, E, k, H, I = 40, 8, 2, 512, 256
    x = torch.randn(T, H, device=DEV, dtype=dt) * 0.5
    w13 = torch.randn(E, 2 * I, H, device=DEV, dtype=dt) * (H ** -0.5)
    w2 = torch.randn(E, H, I, device=DEV, dtype=dt) * (I ** -0.5)
    s13 = w13.float().abs().amax(dim=(1, 2)) / 448.0
    s2 = w2.float().abs().amax(dim=(1, 2)) / 448.0
    q13 = (w13.float() / s13.view(E, 1, 1)).to(torch.float8_e4m3fn)
    q2 = (w2.float() / s2.view(E, 1, 1)).to(torch.float8_e4m3fn)
    tw, ids = R.moe_topk_softmax(torch.randn(T, E, device=DEV), k, True)
    out = ops.fused_moe(x, q13, q2, tw, ids, s13, s2)
    ref = R.fused_moe(x, q13, q2, tw, ids, s13, s2)
    torch.testing.assert_close(out.float(), ref.float(), atol=3e-2, rtol=3e-2)


def test_fused_moe_graph_capture():
    dt = torch.bfloat16
    T, E, k, H, I = 16, 8, 2, 512, 256
    x = torch.randn(T, H, device=DEV, dtype=dt) * 0.5
    w13 = torch.randn(E, 2 * I, H, device=DEV, dtype=dt) * (H ** -0.5)
    w2 = torch.randn(E, H, I, device=DEV, dtype=dt) * (I ** -0.5)
    lg = torch.randn(T, E, device=DEV, dtype=dt)

    def run():
        tw, ids = ops.moe_topk_softmax(lg, k, True)
        return ops.fused_moe(x, w13, w2, tw, ids)
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = run()
    lg.copy_(torch.randn(T, E, device=DEV, dtype=dt))      # new routing, same graph
    g.replay()
    torch.cuda.synchronize()
    tw, ids = R.moe_topk_softmax(lg, k, True)
    torch.testing.assert_close(out.float(), R.fused_moe(x, w13, w2, tw, ids).float(), **tol(dt))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (8, 1024, 8192), (33, 512, 1024), (64, 6144, 4096)])
def test_skinny_gemm_fp8_weights(dt, M, N, K):
    x = torch.randn(M, K, device=DEV, dtype=dt)
    w = torch.randn(N, K, device=DEV, dtype=dt) * (K ** -0.5)
    s = (w.abs().amax().float() / 448.0).reshape(1)
    q = (w.float() / s).to(torch.float8_e4m3fn)
    b = torch.randn(N, device=DEV, dtype=dt)
    y = ops.skinny_gemm(x, q, b, s)
    torch.testing.assert_close(y.float(), R.skinny_gemm(x, q, b, s).float(), **tol(dt))


@pytest.mark.parametrize("M", [4, 48, 300])
@pytest.mark.parametrize("static", [False, True])
def test_fp8_linear_paths(M, static):
    from vllm_distributed_amd.layers.quant import apply_fp8_linear
    dt = torch.bfloat16
    N, K = 1024, 2048
    x = torch.randn(M, K, device=DEV, dtype=dt)
    w = torch.randn(N, K, device=DEV, dtype=dt) * (K ** -0.5)
    s = (w.abs().amax().float() / 448.0).reshape(())
    q = (w.float() / s).to(torch.float8_e4m3fn)
    ins = (x.abs().amax().float() / 448.0).reshape(()) if static else None
    y = apply_fp8_linear(x, q, s, ins, None)
    ref = x.float() @ (q.float() * s).t()
    # M=300 goes through activation quantisation + fp8 MFMA: looser
    torch.testing.assert_close(y.float(), ref, atol=0.15 if M > 64 else 3e-2, rtol=5e-2)


# ---- fp8 KV cache ---------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("Hq,Hkv,D,bs", [(32, 8, 128, 16), (12, 12, 64, 32), (8, 1, 128, 64)])
def test_fp8_kv_cache_write_and_attention(dt, Hq, Hkv, D, bs):
    """Write K/V through the fused RoPE+cache kernel into e4m3 caches, then decode / prefill attention
    must equal the reference evaluated on the SAME quantised caches (same bytes, same scales)."""
    f8 = torch.float8_e4m3fn
    seq_lens, q_lens = [33, 300, 129, 64], [33, 300, 129, 64]
    T = sum(q_lens)
    nblk = sum((s + bs - 1) // bs for s in seq_lens) + 2
    kc = torch.zeros(nblk, Hkv, bs, D, device=DEV, dtype=f8)
    vc = torch.zeros(nblk, Hkv, D, bs, device=DEV, dtype=f8)
    kr, vr = kc.clone(), vc.clone()
    ks, vs = 0.05, 0.02
    cs = _cos_sin(512, D)
    qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device=DEV, dtype=dt)
    ref = qkv.clone()
    # slots: sequence i occupies consecutive blocks
    maxb = max((s + bs - 1) // bs for s in seq_lens)
    bt = torch.zeros(len(seq_lens), maxb, dtype=torch.int32, device=DEV)
    slots, pos, b = [], [], 1
    for i, s in enumerate(seq_lens):
        n = (s + bs - 1) // bs
        bt[i, :n] = torch.arange(b, b + n, dtype=torch.int32)
        slots += [b * bs + j for j in range(s)]
        pos += list(range(s))
        b += n
    slots = torch.tensor(slots, device=DEV)
    pos = torch.tensor(pos, device=DEV)
    ops.rope_and_cache(pos, qkv, cs, kc, vc, slots, D, D, Hq, Hkv, True, ks, vs)
    q, k, v = ref[:, :Hq * D], ref[:, Hq * D:(Hq + Hkv) * D], ref[:, (Hq + Hkv) * D:]
    R.rotary_embedding(pos, q, k, cs, D, D, True)
    R.reshape_and_cache(k.reshape(T, Hkv, D), v.reshape(T, Hkv, D), kr, vr, slots, ks, vs)
    # quantisation may differ by one e4m3 step where rounding is on the edge
    dk = (kc.float() - kr.float()).abs()
    assert (dk <= kr.float().abs() * 0.13 + 1e-3).all()
    assert (dk > 0).float().mean() < 0.02
    scale = 1.0 / math.sqrt(D)
    qsl = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0)), dtype=torch.int32, device=DEV)
    sl = torch.tensor(seq_lens, dtype=torch.int32, device=DEV)
    qq = qkv[:, :Hq * D].reshape(T, Hq, D).contiguous()
    out = torch.empty_like(qq)
    ops.paged_attention_prefill(out, qq, kc, vc, bt, sl, qsl, scale, k_scale=ks, v_scale=vs)
    want = R.paged_attention(qq, kc, vc, bt, sl, qsl, scale, k_scale=ks, v_scale=vs)
    torch.testing.assert_close(out.float(), want.float(), **tol(dt))
    # decode: last token of every sequence as the single query
    last = (qsl[1:] - 1).long()
    qd = qq[last].contiguous()
    qsl1 = torch.arange(len(seq_lens) + 1, dtype=torch.int32, device=DEV)
    for nsplit in (1, 3):
        od = torch.empty_like(qd)
        ops.paged_attention_decode(od, qd, kc, vc, bt, sl, qsl1, scale, num_splits=nsplit, k_scale=ks, v_scale=vs)
        wd = R.paged_attention(qd, kc, vc, bt, sl, qsl1, scale, k_scale=ks, v_scale=vs)
        torch.testing.assert_close(od.float(), wd.float(), **tol(dt))


# ---- int8 W8A8 --------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("H", [128, 4096, 8192, 28672])
def test_int8_quant_dynamic_and_static(dt, H):
    x = (torch.randn(37, H, device=DEV) * 3).to(dt)
    q, s = ops.int8_quant(x)
    qr, sr = R.int8_quant(x)
    torch.testing.assert_close(s, sr, rtol=1e-6, atol=0)
    assert int((q.int() - qr.int()).abs().max()) <= 1          # rounding of values that sit on .5 after the divide
    assert float((q != qr).float().mean()) < 1e-3
    st = torch.tensor([0.05], device=DEV)
    q2, _ = ops.int8_quant(x, st)
    q2r, _ = R.int8_quant(x, st)
    assert int((q2.int() - q2r.int()).abs().max()) <= 1


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("H", [128, 4096, 11008])
def test_int8_quant_asymmetric(dt, H):
    """Zero-point int8 (reference: tests/kernels/quantization/test_int8_quant.py, the azp cases)."""
    x = (torch.randn(37, H, device=DEV) * 3 + 1.5).to(dt)
    q, s, z = ops.int8_quant_azp(x)
    qr, sr, zr = R.int8_quant_azp(x)
    torch.testing.assert_close(s, sr, rtol=1e-6, atol=0)
    assert int((z - zr).abs().max()) <= 1
    assert int((q.int() - qr.int()).abs().max()) <= 1 and float((q != qr).float().mean()) < 2e-3
    # the quantised rows reproduce the input to half a step
    back = (q.float() - z.float()) * s
    assert float((back - x.float()).abs().max()) <= float(s.max()) * 0.51 + 2e-2
    st, sz = torch.tensor([0.04], device=DEV), torch.tensor([-9], device=DEV, dtype=torch.int32)
    q2, _, _ = ops.int8_quant_azp(x, st, sz)
    q2r, _, _ = R.int8_quant_azp(x, st, sz)
    assert int((q2.int() - q2r.int()).abs().max()) <= 1


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("rows,cols", [(1, 128), (33, 4096), (7, 1000)])
def test_permute_cols(dt, rows, cols):
    x = torch.randn(rows, cols, device=DEV).to(dt)
    perm = torch.randperm(cols, device=DEV).int()
    assert torch.equal(ops.permute_cols(x, perm), x[:, perm.long()])


@pytest.mark.parametrize("dt", DTYPES)
def test_convert_fp8_both_directions(dt):
    x = torch.randn(40, 8, 128, device=DEV).to(dt)
    q = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device=DEV)
    ops.convert_fp8(q, x, 0.05)
    torch.testing.assert_close(q.float(), (x.float() / 0.05).clamp(-448, 448).to(torch.float8_e4m3fn).float(),
                               rtol=0.07, atol=1e-3)
    back = torch.empty_like(x)
    ops.convert_fp8(back, q, 0.05)
    torch.testing.assert_close(back.float(), q.float() * 0.05, rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize("M", [1, 17, 300])
def test_int8_linear_matches_integer_reference(M):
    from vllm_distributed_amd.layers.quant import _int8_matmul
    xq = torch.randint(-127, 128, (M, 1024), device=DEV, dtype=torch.int8)
    w = torch.randint(-127, 128, (512, 1024), device=DEV, dtype=torch.int8)
    ref = (xq.cpu().long() @ w.cpu().long().t()).float()
    torch.testing.assert_close(_int8_matmul(xq, w).cpu(), ref, rtol=1e-6, atol=0.5)


# ---- state-space kernels ---------------------------------------------------------------------------
def _ragged(lens, computed):
    """Batch description shared by the state-space kernels: qsl, seq_lens, block table (slot in column 0)."""
    S = len(lens)
    qsl = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)
    seq = torch.tensor([l + c if l else 0 for l, c in zip(lens, computed)], dtype=torch.int32, device=DEV)
    bt = torch.zeros((S, 4), dtype=torch.int32, device=DEV)
    bt[:, 0] = torch.randperm(S + 3, device=DEV)[:S].int()
    return qsl, seq, bt


@pytest.mark.parametrize("dt", DTYPES + [torch.float32])
@pytest.mark.parametrize("W", [2, 3, 4])
@pytest.mark.parametrize("lens,computed", [([1, 1, 1, 1], [5, 0, 9, 1]), ([7, 150, 1, 2], [0, 0, 3, 2]),
                                           ([200], [64]), ([3, 1, 70], [1, 0, 0])])
def test_causal_conv1d(dt, W, lens, computed):
    C = 328
    T = sum(lens)
    qsl, seq, bt = _ragged(lens, computed)
    x = torch.randn(T, 2 * C, device=DEV, dtype=dt)[:, :C]          # strided rows, like a slice of in_proj's output
    w = torch.randn(C, W, device=DEV, dtype=dt) * 0.5
    b = torch.randn(C, device=DEV, dtype=dt)
    st = torch.randn(len(lens) + 3, C, W - 1, device=DEV, dtype=dt)
    st_ref = st.clone()
    y = ops.causal_conv1d(x, w, b, st, qsl, seq, bt, max(lens))
    y_ref = R.causal_conv1d(x, w, b, st_ref, qsl, seq, bt)
    torch.testing.assert_close(y.float(), y_ref.float(), **tol(dt if dt != torch.float32 else torch.float16))
    torch.testing.assert_close(st.float(), st_ref.float(), atol=1e-6, rtol=0)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("variant", ["mamba1", "mamba2"])
@pytest.mark.parametrize("lens,computed", [([1, 1, 1], [4, 0, 30]), ([9, 33, 1], [0, 5, 2]), ([64], [0]),
                                           ([600, 3, 260], [0, 5, 40])])      # last: chunk-parallel passes
def test_selective_scan(dt, variant, lens, computed):
    T = sum(lens)
    qsl, seq, bt = _ragged(lens, computed)
    if variant == "mamba1":
        Ch, N, E, G = 192, 16, 192, 1
        A = -torch.rand(Ch, N, device=DEV) * 2
    else:
        Ch, N, E, G = 256, 128, 8, 2           # 8 heads of 32 channels, 2 B/C groups
        A = -torch.rand(E, device=DEV) * 2
    u = torch.randn(T, Ch, device=DEV, dtype=dt)
    z = torch.randn(T, Ch, device=DEV, dtype=dt)
    delta = torch.randn(T, E, device=DEV, dtype=dt) * 0.5
    bc = torch.randn(T, 2 * G * N + 5, device=DEV, dtype=dt) * 0.3
    B, C = bc[:, :G * N], bc[:, G * N:2 * G * N]
    D = torch.randn(E, device=DEV)
    db = torch.randn(E, device=DEV) * 0.2
    st = torch.randn(len(lens) + 3, Ch, N, device=DEV)
    st_ref = st.clone()
    y = ops.selective_scan(u, delta, A, B, C, D, z, db, st, qsl, seq, bt, num_groups=G, max_query_len=max(lens))
    y_ref = R.selective_scan(u, delta, A, B, C, D, z, db, st_ref, qsl, seq, bt, num_groups=G)
    torch.testing.assert_close(y.float(), y_ref.float(), atol=6e-2 if dt == torch.bfloat16 else 2e-2, rtol=3e-2)
    torch.testing.assert_close(st, st_ref, atol=2e-3, rtol=2e-3)


def test_state_space_kernels_skip_padded_rows():
    """Rows with seq_len 0 (graph padding) neither read nor write state."""
    qsl = torch.tensor([0, 1, 2], dtype=torch.int32, device=DEV)
    seq = torch.tensor([3, 0], dtype=torch.int32, device=DEV)
    bt = torch.tensor([[1], [2]], dtype=torch.int32, device=DEV)
    C, W, N = 64, 4, 16
    x = torch.randn(2, C, device=DEV, dtype=torch.bfloat16)
    cs = torch.randn(4, C, W - 1, device=DEV, dtype=torch.bfloat16)
    ss = torch.randn(4, C, N, device=DEV)
    cs0, ss0 = cs.clone(), ss.clone()
    ops.causal_conv1d(x, torch.randn(C, W, device=DEV, dtype=torch.bfloat16), None, cs, qsl, seq, bt, 1)
    ops.selective_scan(x, x, -torch.rand(C, N, device=DEV), torch.randn(2, N, device=DEV, dtype=torch.bfloat16),
                       torch.randn(2, N, device=DEV, dtype=torch.bfloat16), None, None, None, ss, qsl, seq, bt)
    assert torch.equal(cs[2], cs0[2]) and torch.equal(ss[2], ss0[2])
    assert not torch.equal(cs[1], cs0[1]) and not torch.equal(ss[1], ss0[1])


# ---- 4-bit weights ---------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K,G", [(1, 4096, 4096, 128), (7, 1024, 8192, 64), (33, 512, 1024, 32),
                                     (64, 6144, 4096, 128), (200, 256, 512, 128)])
def test_w4_linear(dt, M, N, K, G):
    q = torch.randint(0, 16, (N, K), device=DEV)
    qw = ops.w4_pack(q)
    scales = (torch.rand(N, K // G, device=DEV) * 0.02 + 0.005).to(dt)
    zeros = torch.randint(0, 17, (N, K // G), device=DEV).to(dt)
    x = torch.randn(M, K, device=DEV, dtype=dt)
    b = torch.randn(N, device=DEV, dtype=dt)
    w_ref = R.w4_dequant(qw, scales, zeros, G, torch.float32)
    torch.testing.assert_close(ops.w4_dequant(qw, scales, zeros, G, dt).float(), w_ref, atol=2e-3, rtol=1e-2)
    y = ops.w4_linear(x, qw, scales, zeros, G, b)
    y_ref = x.float() @ w_ref.t() + b.float()
    torch.testing.assert_close(y.float(), y_ref, atol=0.15 if dt == torch.bfloat16 else 0.03, rtol=2e-2)


# ---- multi-LoRA ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("T,K,N,Rk", [(1, 4096, 4096, 16), (37, 1024, 3072, 8), (256, 2048, 512, 64)])
def test_lora_update(dt, T, K, N, Rk):
    slots = 4
    x = torch.randn(T, K, device=DEV, dtype=dt)
    y = torch.randn(T, N, device=DEV, dtype=dt)
    A = torch.randn(slots, Rk, K, device=DEV, dtype=dt) * (K ** -0.5)
    B = torch.randn(slots, N, Rk, device=DEV, dtype=dt) * 0.3
    ts = torch.randint(-1, slots, (T,), device=DEV, dtype=torch.int32)
    y_ref = y.clone()
    R.lora_update(y_ref, x, A, B, ts)
    ops.lora_update(y, x, A, B, ts)
    torch.testing.assert_close(y.float(), y_ref.float(), **tol(dt))


# ---- multi-head latent attention -------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("DK,DV", [(96, 64), (576, 512), (160, 128)])
@pytest.mark.parametrize("H,lens,computed,nsplit", [(16, [1, 1, 1], [37, 700, 5], 1), (16, [1, 1], [1500, 90], 4),
                                                    (4, [9, 40, 1], [0, 13, 100], 1), (128, [2, 1], [300, 64], 2),
                                                    (8, [130], [0], 1)])
@pytest.mark.parametrize("fp8", [False, True])
def test_mla_attention(dt, DK, DV, H, lens, computed, nsplit, fp8):
    BS, S = 16, len(lens)
    cdt, kv_scale = (torch.float8_e4m3fn, 0.5) if fp8 else (dt, 1.0)
    T = sum(lens)
    total = [l + c for l, c in zip(lens, computed)]
    nblocks = sum((t + BS - 1) // BS for t in total) + 4
    kc = torch.zeros(nblocks, BS, DK, device=DEV, dtype=cdt)
    vt = torch.zeros(nblocks, DV, BS, device=DEV, dtype=cdt)
    kc_ref, vt_ref = kc.clone(), vt.clone()
    perm = torch.randperm(nblocks - 1, device=DEV) + 1
    maxb = max((t + BS - 1) // BS for t in total)
    bt = torch.zeros((S, maxb), dtype=torch.int32, device=DEV)
    used = 0
    for s, t in enumerate(total):
        nb = (t + BS - 1) // BS
        bt[s, :nb] = perm[used:used + nb].int()
        used += nb
    # fill the caches through the write kernel, one call with every token of every sequence
    all_c = torch.randn(sum(total), DV, device=DEV, dtype=dt)
    all_pe = torch.randn(sum(total), DK - DV, device=DEV, dtype=dt)
    slots, o = [], 0
    for s, t in enumerate(total):
        pos = torch.arange(t, device=DEV)
        slots.append(bt[s, pos // BS].long() * BS + pos % BS)
    slots = torch.cat(slots)
    slots[3] = -1                                   # skipped token
    ops.mla_cache_write(all_c, all_pe, kc, vt, slots, kv_scale)
    R.mla_cache_write(all_c, all_pe, kc_ref, vt_ref, slots, kv_scale)
    if fp8:          # the same e4m3 bytes up to the rounding of a tie
        assert (kc.float() - kc_ref.float()).abs().max().item() <= 0.13 * kc_ref.float().abs().max().item()
        assert (kc.view(torch.uint8) != kc_ref.view(torch.uint8)).float().mean().item() < 0.01
        kc_ref, vt_ref = kc, vt
    else:
        assert torch.equal(kc, kc_ref) and torch.equal(vt, vt_ref)
    qsl = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)
    seq = torch.tensor(total, dtype=torch.int32, device=DEV)
    q = torch.randn(T, H, DK, device=DEV, dtype=dt) * 0.5
    out = torch.empty(T, H, DV, device=DEV, dtype=dt)
    ops.mla_attention(out, q, kc, vt, bt, seq, qsl, DK ** -0.5, max(lens), nsplit, kv_scale)
    ref = R.mla_attention(q, kc_ref, vt_ref, bt, seq, qsl, DK ** -0.5, kv_scale)
    torch.testing.assert_close(out.float(), ref.float(), **tol(dt))


@pytest.mark.parametrize("D,dtype", [(128, torch.bfloat16), (64, torch.float16), (32, torch.float32)])
def test_lightning_attention_decode_matches_reference(D, dtype):
    """One-token lightning attention step (state update + q.S in one pass) against the same step in plain PyTorch
    fp32; rows: ongoing sequences, a sequence at its first token (starts from zero) and a padded row (untouched)."""
    from vllm_distributed_amd import ops
    torch.manual_seed(0)
    S, H, slots = 6, 5, 9
    dev = "cuda"
    qkv = (torch.randn(S, H * 3 * D, device=dev) * 0.5).to(dtype)
    state = torch.randn(slots, H, D, D, device=dev)
    slope = torch.rand(H, device=dev) * 0.5 + 0.01
    seq_lens = torch.tensor([7, 1, 30, 0, 2, 1], dtype=torch.int32, device=dev)
    table = torch.tensor([[3], [1], [8], [0], [5], [2]], dtype=torch.int32, device=dev)
    ref_state = state.clone()
    q, k, v = qkv.view(S, H, 3 * D).float().split(D, dim=-1)
    want = torch.zeros(S, H, D, device=dev)
    for s in range(S):
        if int(seq_lens[s]) <= 0:
            continue
        slot = int(table[s, 0])
        prev = ref_state[slot] * (0.0 if int(seq_lens[s]) <= 1 else 1.0) * torch.exp(-slope).view(H, 1, 1)
        new = prev + k[s].unsqueeze(-1) * v[s].unsqueeze(-2)
        ref_state[slot] = new
        want[s] = torch.matmul(q[s].unsqueeze(-2), new).squeeze(-2)
    got = ops.lightning_attention_decode(qkv, state, slope, seq_lens, table, H, D)
    live = seq_lens > 0
    torch.testing.assert_close(got[live], want[live], atol=1e-3, rtol=1e-4)
    torch.testing.assert_close(state, ref_state, atol=1e-4, rtol=1e-5)
