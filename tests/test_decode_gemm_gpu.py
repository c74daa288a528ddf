"""Weight-streaming split-K decode GEMM (ops/csrc/gemm_decode.hip) against a plain fp32 PyTorch product."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(x, w, b):
    y = x.float() @ w.float().t()
    return y if b is None else y + b.float()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 15, 16, 17, 32, 33, 64, 100, 128, 129, 200, 256])
@pytest.mark.parametrize("N,K", [(512, 256), (8192, 8192), (1000, 1024), (4100, 2048), (6144, 4096), (4096, 14336),
                                 (28672, 512)])
def test_decode_gemm_matches_fp32(dtype, M, N, K):
    from vllm_distributed_amd import ops
    torch.manual_seed(M * 7 + N)
    x = torch.randn(M, K, device="cuda", dtype=dtype)
    w = (torch.randn(N, K, device="cuda", dtype=dtype) * 0.05)
    got = ops.decode_gemm(x, w).float()
    ref = _ref(x, w, None)
    tol = 2e-2 * ref.abs().max().item() + 1e-2
    assert (got - ref).abs().max().item() < tol
    # against the library GEMM the error must be of the same size (fp32 accumulation in both)
    lib = torch.nn.functional.linear(x, w).float()
    assert (got - ref).abs().max().item() <= 2.0 * (lib - ref).abs().max().item() + 1e-2


@pytest.mark.parametrize("M", [48, 160])
def test_decode_gemm_bias_strided_input_and_output(M):
    from vllm_distributed_amd import ops
    N, K = 2048, 1024
    big = torch.randn(M, K + 512, device="cuda", dtype=torch.bfloat16)
    x = big[:, 256:256 + K]                       # row stride K + 512
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
    b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    out_big = torch.zeros(M, N + 64, device="cuda", dtype=torch.bfloat16)
    out = out_big[:, :N]
    ops.decode_gemm(x, w, b, out=out)
    ref = _ref(x, w, b)
    assert (out.float() - ref).abs().max().item() < 2e-2 * ref.abs().max().item() + 1e-2
    assert out_big[:, N:].abs().max().item() == 0


def test_linear_routes_decode_batches_through_the_kernel(monkeypatch):
    """ops.linear must give the same result as the library for a decode batch, and must have called the kernel."""
    from vllm_distributed_amd import ops
    monkeypatch.setenv("VDA_DECODE_GEMM_MAX_M", "256")
    calls = []
    real = ops.decode_gemm
    monkeypatch.setattr(ops, "decode_gemm", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    x = torch.randn(96, 4096, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(6144, 4096, device="cuda", dtype=torch.bfloat16) * 0.03
    y = ops.linear(x, w)
    assert calls, "decode batch did not reach the decode GEMM"
    ref = _ref(x, w, None)
    assert (y.float() - ref).abs().max().item() < 2e-2 * ref.abs().max().item() + 1e-2


def test_decode_gemm_in_a_graph():
    from vllm_distributed_amd import ops
    x = torch.randn(64, 8192, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16) * 0.02
    out = torch.empty(64, 8192, device="cuda", dtype=torch.bfloat16)
    ops.decode_gemm(x, w, out=out)            # allocates the workspace outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.decode_gemm(x, w, out=out)
    x.copy_(torch.randn_like(x))
    g.replay()
    torch.cuda.synchronize()
    ref = _ref(x, w, None)
    assert (out.float() - ref).abs().max().item() < 2e-2 * ref.abs().max().item() + 1e-2
