"""Wide prefill attention kernel (ops/csrc/attention_prefill.hip: 8 waves, 32x32x16 MFMA, 64-key tiles)
against the plain PyTorch fp32 reference, forced on for shapes that the automatic choice would give to
the 4-wave form.  Grids in the style of the reference's tests/kernels/attention/test_prefix_prefill.py."""
import math

import pytest
import torch

from vllm_distributed_amd import ops
from vllm_distributed_amd.ops import native as N
from vllm_distributed_amd.ops import reference as R
from test_ops_gpu import _paged_problem, tol

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _force_wide():
    torch.manual_seed(0)
    N.call("vda_set_prefill_variant", 3)
    yield
    N.call("vda_set_prefill_variant", 1)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Hq,Hkv", [(32, 8), (8, 1), (16, 16), (64, 8), (4, 2)])
@pytest.mark.parametrize("bs", [16, 32, 64])
def test_wide_prefill_mixed_batch(dt, Hq, Hkv, bs):
    # full prefills, chunked prefills behind cached context, single-token rows, lengths around tile edges
    seq_lens = [5, 130, 300, 77, 1, 257, 64, 640]
    q_lens = [5, 130, 44, 77, 1, 100, 64, 129]
    q, kc, vc, bt, sl, qsl = _paged_problem(seq_lens, q_lens, Hq, Hkv, 128, bs, dt)
    scale = 1.0 / math.sqrt(128)
    out = torch.full_like(q, float("nan"))
    ops.paged_attention_prefill(out, q, kc, vc, bt, sl, qsl, scale)
    ref = R.paged_attention(q, kc, vc, bt, sl, qsl, scale)
    torch.testing.assert_close(out.float(), ref.float(), **tol(dt))


@pytest.mark.parametrize("q_lens,seq_lens", [([300, 300, 300], [300, 300, 300]), ([5, 130, 44, 77, 1, 100], [5, 130, 300, 77, 1, 257]),
                                             ([700, 3, 64], [900, 3, 128])])
def test_wide_prefill_level_order(q_lens, seq_lens):
    """Longest query run known on the host: workgroups walk (level from the end, sequence); every tile of
    every sequence must still be produced, also when the bound is loose or the batch is ragged."""
    dt, Hq, Hkv, bs = torch.bfloat16, 16, 4, 16
    q, kc, vc, bt, sl, qsl = _paged_problem(seq_lens, q_lens, Hq, Hkv, 128, bs, dt)
    scale = 1.0 / math.sqrt(128)
    ref = R.paged_attention(q, kc, vc, bt, sl, qsl, scale)
    for bound in (max(q_lens), max(q_lens) + 100):
        out = torch.full_like(q, float("nan"))
        ops.paged_attention_prefill(out, q, kc, vc, bt, sl, qsl, scale, max_q_len=bound)
        torch.testing.assert_close(out.float(), ref.float(), **tol(dt))


def test_wide_equals_four_wave_form_on_long_prompts():
    dt, Hq, Hkv, bs = torch.bfloat16, 32, 8, 16
    q, kc, vc, bt, sl, qsl = _paged_problem([2048, 1500 + 700], [2048, 700], Hq, Hkv, 128, bs, dt)
    scale = 1.0 / math.sqrt(128)
    wide = torch.empty_like(q)
    ops.paged_attention_prefill(wide, q, kc, vc, bt, sl, qsl, scale)
    N.call("vda_set_prefill_variant", 2)
    narrow = torch.empty_like(q)
    ops.paged_attention_prefill(narrow, q, kc, vc, bt, sl, qsl, scale)
    ref = R.paged_attention(q, kc, vc, bt, sl, qsl, scale)
    torch.testing.assert_close(wide.float(), ref.float(), **tol(dt))
    torch.testing.assert_close(wide.float(), narrow.float(), atol=1e-2, rtol=1e-2)


@pytest.mark.parametrize("kw", ["window", "softcap", "alibi", "window_long"])
def test_wide_prefill_window_softcap_alibi(kw):
    dt, Hq, Hkv, bs = torch.bfloat16, 8, 2, 16
    q, kc, vc, bt, sl, qsl = _paged_problem([400, 90, 700], [400, 30, 333], Hq, Hkv, 128, bs, dt)
    args = {"window": dict(window=64), "window_long": dict(window=300), "softcap": dict(softcap=30.0),
            "alibi": dict(alibi=torch.tensor([2 ** (-(i + 1)) for i in range(Hq)], device=DEV,
                                             dtype=torch.float32))}[kw]
    out = torch.full_like(q, float("nan"))
    ops.paged_attention_prefill(out, q, kc, vc, bt, sl, qsl, 0.09, **args)
    ref = R.paged_attention(q, kc, vc, bt, sl, qsl, 0.09, **args)
    torch.testing.assert_close(out.float(), ref.float(), **tol(dt))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_wide_prefill_bidirectional_and_lse(dt):
    Hq, Hkv = 8, 2
    seq_lens = [5, 130, 300, 33, 64]
    q, kc, vc, bt, sl, qsl = _paged_problem(seq_lens, seq_lens, Hq, Hkv, 128, 16, dt)
    scale = 1.0 / math.sqrt(128)
    for causal in (False, True):
        out = torch.full_like(q, float("nan"))
        _, lse = ops.paged_attention_prefill(out, q, kc, vc, bt, sl, qsl, scale, causal=causal, return_lse=True)
        ref, lref = R.paged_attention(q, kc, vc, bt, sl, qsl, scale, return_lse=True, causal=causal)
        torch.testing.assert_close(out.float(), ref.float(), **tol(dt))
        torch.testing.assert_close(lse, lref, atol=2e-2, rtol=2e-2)


def test_wide_prefill_rescale_path():
    """Keys that dwarf everything before them, late in the sequence and in the middle of a tile, force the
    running-maximum rescale of the output accumulators for some columns of a wave and not for others."""
    dt, Hq, Hkv, bs = torch.bfloat16, 8, 2, 16
    q, kc, vc, bt, sl, qsl = _paged_problem([600], [600], Hq, Hkv, 128, bs, dt)
    for pos in (70, 333, 480):
        blk, off = int(bt[0, pos // bs]), pos % bs
        kc[blk, :, off, :] = q[pos + 7, ::4, :] * 5.0
    out = torch.empty_like(q)
    ops.paged_attention_prefill(out, q, kc, vc, bt, sl, qsl, 0.1)
    ref = R.paged_attention(q, kc, vc, bt, sl, qsl, 0.1)
    torch.testing.assert_close(out.float(), ref.float(), atol=3e-2, rtol=3e-2)


@pytest.mark.parametrize("bs", [16, 64])
def test_wide_prefill_fp8_cache(bs):
    dt, Hq, Hkv, D = torch.bfloat16, 16, 4, 128
    seq_lens, q_lens = [300, 129, 64], [300, 40, 64]
    q, kc, vc, bt, sl, qsl = _paged_problem(seq_lens, q_lens, Hq, Hkv, D, bs, dt)
    ks, vs = 0.05, 0.04
    kc8 = (kc.float() / ks).clamp(-448, 448).to(torch.float8_e4m3fn)
    vc8 = (vc.float() / vs).clamp(-448, 448).to(torch.float8_e4m3fn)
    scale = 1.0 / math.sqrt(D)
    out = torch.empty_like(q)
    ops.paged_attention_prefill(out, q, kc8, vc8, bt, sl, qsl, scale, k_scale=ks, v_scale=vs)
    ref = R.paged_attention(q, kc8, vc8, bt, sl, qsl, scale, k_scale=ks, v_scale=vs)
    torch.testing.assert_close(out.float(), ref.float(), **tol(dt))
