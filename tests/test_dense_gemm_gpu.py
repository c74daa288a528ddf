"""Ping-pong MFMA dense GEMM (ops/csrc/gemm_dense.hip) against plain fp32 PyTorch products: both kernel forms,
both tile heights, every activation-row count class, K splits, edges, fused epilogues, strides, graph capture and the
default routing of ops.linear / the gated-MLP layer."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rnd(*shape, dtype=torch.bfloat16, scale=1.0):
    return ((torch.rand(*shape, device="cuda") * 2 - 1) * scale).to(dtype)


def _tol(ref):
    return 2e-2 * ref.abs().max().item() + 1e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 16, 31, 33, 64, 100, 128, 129, 200, 256, 257, 513, 1000])
@pytest.mark.parametrize("N,K", [(256, 256), (264, 96), (1000, 512), (4096, 4096), (6144, 1024)])
def test_dense_gemm_matches_fp32(dtype, M, N, K):
    from vllm_distributed_amd import ops
    torch.manual_seed(M * 13 + N)
    x = _rnd(M, K, dtype=dtype)
    w = _rnd(N, K, dtype=dtype, scale=K ** -0.5)
    ref = x.float() @ w.float().t()
    lib_err = (torch.nn.functional.linear(x, w).float() - ref).abs().max().item()
    for wt, variant in ((0, 0), (2, 1), (4, 1), (4, 2)):
        for splits in (0, 1, 3):
            got = ops.dense_gemm(x, w, wt=wt, splits=splits, variant=variant).float()
            err = (got - ref).abs().max().item()
            assert err < _tol(ref), (wt, variant, splits, err)
            # fp32 accumulation in both: the error must be of the library's size
            assert err <= 2.0 * lib_err + 1e-2, (wt, variant, splits, err, lib_err)


@pytest.mark.parametrize("M", [8, 64, 128, 256, 300, 1024])
@pytest.mark.parametrize("I,K", [(128, 256), (500, 512), (14336, 4096)])
def test_dense_gemm_gate_up_epilogue(M, I, K):
    """w = [gate rows; up rows]: silu(x gate^T) * (x up^T) leaves the GEMM directly."""
    from vllm_distributed_amd import ops
    x = _rnd(M, K)
    w = _rnd(2 * I, K, scale=K ** -0.5)
    y = x.float() @ w.float().t()
    ref = torch.nn.functional.silu(y[:, :I]) * y[:, I:]
    for wt, variant in ((0, 0), (2, 1), (4, 1), (4, 2)):
        for splits in (0, 1, 2):
            got = ops.dense_gemm(x, w, epilogue="gate_up", wt=wt, splits=splits, variant=variant).float()
            assert got.shape == (M, I)
            assert (got - ref).abs().max().item() < _tol(ref) + 1e-2, (wt, variant, splits)


@pytest.mark.parametrize("M", [5, 130, 256, 600])
def test_dense_gemm_bias_and_residual(M):
    from vllm_distributed_amd import ops
    N, K = 1032, 768
    x, w = _rnd(M, K), _rnd(N, K, scale=K ** -0.5)
    b, res = _rnd(N), _rnd(M, N)
    ref = x.float() @ w.float().t()
    for splits in (1, 2):
        got = ops.dense_gemm(x, w, bias=b, residual=res, splits=splits).float()
        assert (got - (ref + b.float() + res.float())).abs().max().item() < 5e-2
        got = ops.dense_gemm(x, w, bias=b, splits=splits).float()
        assert (got - (ref + b.float())).abs().max().item() < 5e-2


def test_dense_gemm_strides_and_exact_layout():
    from vllm_distributed_amd import ops
    x_full = _rnd(300, 1024 + 64)
    w = _rnd(512, 1024, scale=1 / 32)
    xs = x_full[:, 64:]
    out_full = torch.zeros(300, 768, device="cuda", dtype=torch.bfloat16)
    ops.dense_gemm(xs, w, out=out_full[:, 128:640])
    ref = xs.float() @ w.float().t()
    assert (out_full[:, 128:640].float() - ref).abs().max().item() < 5e-2
    assert out_full[:, :128].abs().max().item() == 0 and out_full[:, 640:].abs().max().item() == 0
    # exact integer data, identity activations, asymmetric weights: a transposed or permuted fragment cannot pass
    for M, variant in ((64, 1), (512, 1), (512, 2)):
        x = torch.zeros(M, 64, device="cuda", dtype=torch.bfloat16)
        idx = torch.arange(M, device="cuda")
        x[idx, idx % 64] = 1
        wq = (torch.arange(320 * 64, device="cuda").reshape(320, 64) % 251).to(torch.bfloat16)
        got = ops.dense_gemm(x, wq, variant=variant).float()
        assert torch.equal(got, wq.float().t()[idx % 64]), (M, variant)


def test_dense_gemm_in_a_graph():
    from vllm_distributed_amd import ops
    x = _rnd(96, 4096)
    w = _rnd(6144, 4096, scale=1 / 64)
    out = torch.empty(96, 3072, device="cuda", dtype=torch.bfloat16)
    ops.dense_gemm(x, w, epilogue="gate_up", out=out)          # allocates the workspace outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.dense_gemm(x, w, epilogue="gate_up", out=out)
    x.copy_(_rnd(96, 4096))
    g.replay()
    torch.cuda.synchronize()
    y = x.float() @ w.float().t()
    ref = torch.nn.functional.silu(y[:, :3072]) * y[:, 3072:]
    assert (out.float() - ref).abs().max().item() < _tol(ref) + 1e-2


def test_linear_and_gated_mlp_take_the_in_tree_gemm_by_default(monkeypatch):
    """A decode-sized 16-bit linear runs the in-tree kernel (not the library) and the Llama MLP fuses SiLU * up."""
    from vllm_distributed_amd import ops
    monkeypatch.setenv("VDA_DENSE_GATE_UP_MAX_M", "128")
    calls = []
    real = ops.dense_gemm
    monkeypatch.setattr(ops, "dense_gemm", lambda *a, **k: (calls.append(k.get("epilogue", "plain")), real(*a, **k))[1])
    x = _rnd(96, 4096)
    w = _rnd(6144, 4096, scale=1 / 64)
    y = ops.linear(x, w)
    assert calls == ["plain"]
    ref = x.float() @ w.float().t()
    assert (y.float() - ref).abs().max().item() < _tol(ref)
    h = ops.gate_up_silu_mul(x, w)
    assert calls == ["plain", "gate_up"] and h is not None and h.shape == (96, 3072)
    assert ops.gate_up_silu_mul(_rnd(4096, 4096), w) is None            # prefill batches stay on the library
    big = ops.linear(_rnd(2048, 4096), w)
    assert calls == ["plain", "gate_up"] and big.shape == (2048, 6144)
    wide = ops.linear(x, _rnd(28672, 4096, scale=1 / 64))               # the wide gate/up matrix: library
    small = ops.linear(_rnd(8, 4096), w)                                # tiny batches: library
    assert calls == ["plain", "gate_up"] and wide.shape == (96, 28672) and small.shape == (8, 6144)
