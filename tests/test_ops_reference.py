"""CPU checks of the reference ops (they are the oracle for the HIP kernels and
the CPU plumbing path, so they are validated against independent formulas)."""
import math

import torch

from vllm_distributed_amd import ops
from vllm_distributed_amd.ops import reference as R


def test_paged_attention_reference_matches_dense():
    torch.manual_seed(0)
    Hq, Hkv, D, bs = 4, 2, 32, 16
    seq_lens, q_lens = [37, 5], [6, 5]
    nblk = 8
    kc = torch.randn(nblk, Hkv, bs, D)
    vc = torch.randn(nblk, Hkv, D, bs)
    bt = torch.tensor([[3, 1, 6], [2, 0, 0]], dtype=torch.int32)
    q = torch.randn(sum(q_lens), Hq, D)
    qsl = torch.tensor([0, 6, 11], dtype=torch.int32)
    sl = torch.tensor(seq_lens, dtype=torch.int32)
    out = torch.empty_like(q)
    ops.paged_attention_prefill(out, q, kc, vc, bt, sl, qsl, 1 / math.sqrt(D))
    # dense check for sequence 0
    ks = torch.cat([kc[b] for b in (3, 1, 6)], dim=1)[:, :37]          # [Hkv, 37, D]
    vs = torch.cat([vc[b].transpose(1, 2) for b in (3, 1, 6)], dim=1)[:, :37]
    for h in range(Hq):
        for i in range(6):
            pos = 37 - 6 + i
            s = (ks[h // 2, :pos + 1] @ q[i, h]) / math.sqrt(D)
            o = torch.softmax(s, 0) @ vs[h // 2, :pos + 1]
            torch.testing.assert_close(out[i, h], o, atol=1e-5, rtol=1e-5)


def test_reshape_and_cache_then_gather_roundtrip():
    kc, vc = torch.zeros(4, 2, 16, 8), torch.zeros(4, 2, 8, 16)
    k, v = torch.randn(5, 2, 8), torch.randn(5, 2, 8)
    slots = torch.tensor([3, 17, 40, -1, 63])
    ops.reshape_and_cache(k, v, kc, vc, slots)
    kg, vg = ops.gather_cache(kc, vc, slots[slots >= 0])
    assert torch.equal(kg, k[slots >= 0]) and torch.equal(vg, v[slots >= 0])


def test_rotary_neox_matches_complex_rotation():
    D, T = 16, 7
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    pos = torch.arange(T)
    f = torch.outer(torch.arange(32).float(), inv)
    cs = torch.cat((f.cos(), f.sin()), -1)
    q = torch.randn(T, 2 * D)
    ref = q.clone().view(T, 2, D)
    ops.rotary_embedding(pos, q, None, cs, D, D, True)
    z = torch.complex(ref[..., : D // 2], ref[..., D // 2:]) * torch.polar(torch.ones(T, D // 2), f[:T])[:, None]
    torch.testing.assert_close(q.view(T, 2, D), torch.cat((z.real, z.imag), -1), atol=1e-5, rtol=1e-5)


def test_topk_topp_reference():
    logits = torch.tensor([[1.0, 2.0, 3.0, 4.0]])
    out = R.topk_topp_mask(logits.clone(), top_k=torch.tensor([2]))
    assert torch.isinf(out[0, :2]).all() and not torch.isinf(out[0, 2:]).any()
    out = R.topk_topp_mask(logits.clone(), top_p=torch.tensor([0.6]))
    assert (~torch.isinf(out)).sum() == 1      # p(4.0) = 0.64 >= 0.6
    out = R.topk_topp_mask(logits.clone(), top_p=torch.tensor([0.7]))
    assert (~torch.isinf(out)).sum() == 2


def test_merge_attn_states_equals_joint_softmax():
    torch.manual_seed(1)
    D = 8
    q = torch.randn(D)
    k, v = torch.randn(10, D), torch.randn(10, D)
    s = k @ q
    full = torch.softmax(s, 0) @ v
    a = torch.softmax(s[:4], 0) @ v[:4]
    b = torch.softmax(s[4:], 0) @ v[4:]
    la, lb = torch.logsumexp(s[:4], 0), torch.logsumexp(s[4:], 0)
    o = ops.merge_attn_states(a.view(1, 1, D), la.view(1, 1), b.view(1, 1, D), lb.view(1, 1))
    torch.testing.assert_close(o.view(D), full, atol=1e-5, rtol=1e-5)
