"""LDS-tiled MFMA GEMM family (ops/csrc/gemm_tile.hip) against fp32 PyTorch references: dense fp8 GEMM
with per-token and with block scales, grouped MoE GEMMs (16-bit and e4m3) over expert-sorted pairs, the
multi-workgroup counting sort and the per-group quantiser.  Modelled on the reference's
tests/kernels/moe/test_moe.py, test_moe_align_block_size.py and tests/kernels/quantization/test_block_fp8.py."""
import pytest
import torch

from vllm_distributed_amd import ops
from vllm_distributed_amd.ops import reference as R

pytestmark = pytest.mark.gpu
DEV = "cuda"
FP8 = torch.float8_e4m3fn


@pytest.fixture(autouse=True)
def _seed():
    torch.manual_seed(0)


def _rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp(min=1e-9))


@pytest.mark.parametrize("rows,K,group", [(5, 512, 128), (130, 7168, 128), (33, 1024, 64)])
def test_per_group_fp8_quant(rows, K, group):
    x = torch.randn(rows, K, device=DEV, dtype=torch.bfloat16) * 3
    q, s = ops.per_group_fp8_quant(x, group)
    qr, sr = R.per_group_fp8_quant(x, group)
    torch.testing.assert_close(s, sr, rtol=1e-5, atol=1e-8)
    deq = q.float().view(rows, K // group, group) * s.unsqueeze(-1)
    deq_r = qr.float().view(rows, K // group, group) * sr.unsqueeze(-1)
    torch.testing.assert_close(deq, deq_r, atol=0.3, rtol=0.07)      # at most one e4m3 step apart


@pytest.mark.parametrize("M,N,K", [(1, 256, 512), (77, 384, 1024), (128, 4096, 4096), (300, 1000, 2048),
                                    (1024, 6144, 4096)])
@pytest.mark.parametrize("out_dt", [torch.bfloat16, torch.float16])
def test_fp8_gemm_per_token(M, N, K, out_dt):
    x = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=DEV, dtype=torch.float32) * 0.05
    ws = (w.abs().max() / 448).reshape(1)
    wq = (w / ws).to(FP8)
    xq, xs = ops.per_token_fp8_quant(x)
    bias = torch.randn(N, device=DEV, dtype=out_dt) if M % 2 else None
    y = ops.fp8_gemm(xq, wq, xs, ws, out_dt, bias)
    ref = R.fp8_gemm(xq, wq, xs, ws, torch.float32, bias)
    assert _rel(y, ref) < 6e-3, _rel(y, ref)


@pytest.mark.parametrize("M,N,K", [(4, 256, 512), (200, 1536, 7168), (512, 4096, 2048), (129, 576, 1024)])
def test_fp8_gemm_block_scales(M, N, K):
    x = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
    wq = (torch.randn(N, K, device=DEV) * 100).clamp(-448, 448).to(FP8)
    ws = torch.rand((N + 127) // 128, K // 128, device=DEV) * 1e-3 + 1e-4
    xq, xs = ops.per_group_fp8_quant(x)
    y = ops.fp8_gemm(xq, wq, xs, ws, torch.bfloat16, block=True)
    ref = R.fp8_gemm(xq, wq, xs, ws, torch.float32, block=True)
    assert _rel(y, ref) < 6e-3, _rel(y, ref)


def test_fp8_gemm_exact_integers():
    """Small integers are exact in e4m3 and in the fp32 accumulator: any operand-layout mistake shows."""
    M, N, K = 96, 256, 384
    x = torch.randint(-3, 4, (M, K), device=DEV).float()
    w = torch.randint(-3, 4, (N, K), device=DEV).float()
    one = torch.ones(1, device=DEV)
    y = ops.fp8_gemm(x.to(FP8), w.to(FP8), one.expand(M).contiguous(), one, torch.float16)
    torch.testing.assert_close(y.float(), x @ w.t(), atol=0, rtol=0)


@pytest.mark.parametrize("n,E", [(1, 8), (5000, 8), (40000, 64), (70000, 256), (9000, 1000)])
def test_moe_align_tiles(n, E):
    ids = torch.randint(-1, E, (n,), device=DEV, dtype=torch.int32)
    for rows in (64, 128):
        sorted_ids, offsets, tiles, max_tiles = ops.moe_align_tiles(ids.view(-1, 1), E, rows)
        off = offsets.cpu().tolist()
        cnt = torch.bincount(ids[ids >= 0].long(), minlength=E).cpu().tolist()
        assert [off[e + 1] - off[e] for e in range(E)] == cnt
        srt = sorted_ids.cpu()
        idc = ids.cpu()
        for e in range(0, E, max(E // 16, 1)):
            seg = srt[off[e]:off[e + 1]]
            assert bool((idc[seg.long()] == e).all()) and seg.unique().numel() == seg.numel()
        te, tb, tend, nt = (tiles[:max_tiles].cpu().tolist(), tiles[max_tiles:2 * max_tiles].cpu().tolist(),
                            tiles[2 * max_tiles:3 * max_tiles].cpu().tolist(), int(tiles[3 * max_tiles]))
        want = [(e, off[e] + i * rows, off[e + 1]) for e in range(E) for i in range(-(-cnt[e] // rows))]
        assert nt == len(want) <= max_tiles
        assert list(zip(te[:nt], tb[:nt], tend[:nt])) == want


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,E,k,H,I", [(4096, 8, 2, 1024, 2048), (3000, 64, 6, 512, 384), (1, 8, 2, 512, 256),
                                        (16384, 8, 2, 512, 512), (2500, 256, 8, 256, 128)])
def test_fused_moe_tiled_16bit(monkeypatch, dt, T, E, k, H, I):
    monkeypatch.setenv("VDA_MOE_GROUPED_MAX_PAIRS", "0")
    x = torch.randn(T, H, device=DEV, dtype=dt) * 0.5
    w13 = torch.randn(E, 2 * I, H, device=DEV, dtype=dt) * 0.05
    w2 = torch.randn(E, H, I, device=DEV, dtype=dt) * 0.05
    tw, ids = R.moe_topk_softmax(torch.randn(T, E, device=DEV), k, True)
    out = ops.fused_moe(x, w13, w2, tw, ids)
    ref = R.fused_moe(x, w13, w2, tw, ids)
    assert _rel(out, ref) < 1.5e-2, _rel(out, ref)


@pytest.mark.parametrize("T,E,k,H,I", [(4096, 8, 2, 1024, 2048), (3000, 64, 6, 512, 384), (3, 8, 2, 512, 256),
                                        (2500, 256, 8, 256, 128)])
def test_fused_moe_tiled_fp8(monkeypatch, T, E, k, H, I):
    """A8W8: compared with a reference that quantises the activations the same way (exact up to the
    accumulation order), and loosely with the unquantised result."""
    monkeypatch.setenv("VDA_MOE_GROUPED_MAX_PAIRS", "0")
    dt = torch.bfloat16
    x = torch.randn(T, H, device=DEV, dtype=dt) * 0.5
    w13 = torch.randn(E, 2 * I, H, device=DEV) * 0.05
    w2 = torch.randn(E, H, I, device=DEV) * 0.05
    s13 = w13.abs().amax(dim=(1, 2)) / 448
    s2 = w2.abs().amax(dim=(1, 2)) / 448
    q13 = (w13 / s13.view(E, 1, 1)).to(FP8)
    q2 = (w2 / s2.view(E, 1, 1)).to(FP8)
    tw, ids = R.moe_topk_softmax(torch.randn(T, E, device=DEV), k, True)
    ids[::7, 0] = -1                                   # some pairs live on another rank
    out = ops.fused_moe(x, q13, q2, tw, ids, s13, s2)
    # reference with the same activation quantisation
    xq, xs = R.per_token_fp8_quant(x)
    xd = xq.float() * xs
    ref = torch.zeros(T, H, device=DEV)
    for e in range(E):
        tok, slot = torch.where(ids == e)
        if tok.numel() == 0:
            continue
        h = xd[tok] @ (q13[e].float() * s13[e]).t()
        act = (torch.nn.functional.silu(h[:, :I]) * h[:, I:]).to(dt)
        aq, a_s = R.per_token_fp8_quant(act)
        ref.index_add_(0, tok, ((aq.float() * a_s) @ (q2[e].float() * s2[e]).t()).to(dt).float() * tw[tok, slot].unsqueeze(1))
    assert _rel(out, ref) < 2e-2, _rel(out, ref)
    loose = R.fused_moe(x, q13, q2, tw, ids, s13, s2)
    assert _rel(out, loose) < 8e-2, _rel(out, loose)


def test_fused_moe_block_scaled_fp8():
    T, E, k, H, I = 700, 16, 4, 512, 256
    dt = torch.bfloat16
    x = torch.randn(T, H, device=DEV, dtype=dt) * 0.5
    q13 = (torch.randn(E, 2 * I, H, device=DEV) * 60).clamp(-448, 448).to(FP8)
    q2 = (torch.randn(E, H, I, device=DEV) * 60).clamp(-448, 448).to(FP8)
    s13 = torch.rand(E, 2 * I // 128, H // 128, device=DEV) * 1e-3 + 3e-4
    s2 = torch.rand(E, H // 128, I // 128, device=DEV) * 1e-3 + 3e-4
    tw, ids = R.moe_topk_softmax(torch.randn(T, E, device=DEV), k, True)
    out = ops.fused_moe(x, q13, q2, tw, ids, s13, s2, block_scales=True)
    ref = R.fused_moe(x, q13, q2, tw, ids, s13, s2, block_scales=True)
    assert _rel(out, ref) < 6e-2, _rel(out, ref)


def test_fused_moe_tiled_in_graph(monkeypatch):
    """Device-side work distribution: the layer can be captured and replayed with other routing."""
    monkeypatch.setenv("VDA_MOE_GROUPED_MAX_PAIRS", "0")
    T, E, k, H, I = 600, 8, 2, 512, 256
    dt = torch.bfloat16
    x = torch.randn(T, H, device=DEV, dtype=dt)
    w13 = torch.randn(E, 2 * I, H, device=DEV, dtype=dt) * 0.05
    w2 = torch.randn(E, H, I, device=DEV, dtype=dt) * 0.05
    lg = torch.randn(T, E, device=DEV)
    tw, ids = R.moe_topk_softmax(lg, k, True)
    tw_s, ids_s = tw.clone(), ids.clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.fused_moe(x, w13, w2, tw_s, ids_s)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ops.fused_moe(x, w13, w2, tw_s, ids_s)
    tw2, ids2 = R.moe_topk_softmax(torch.randn(T, E, device=DEV) * 3, k, True)
    tw_s.copy_(tw2)
    ids_s.copy_(ids2)
    g.replay()
    torch.cuda.synchronize()
    ref = R.fused_moe(x, w13, w2, tw2, ids2)
    assert _rel(out, ref) < 1.5e-2
