"""Weight-only quantised GEMM (ops/csrc/gemm_wq.hip) against a plain fp32 product with the dequantised weights:
4- and 8-bit weights, every row-count class, group sizes, K splits, edges, bias, strides, graph capture, and the
linear method of GPTQ / AWQ layers (packed memory, activation reordering)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _quantised(N, K, G, bits, dtype):
    from vllm_distributed_amd import ops
    top = (1 << bits) - 1
    q = torch.randint(0, top + 1, (N, K), device="cuda")
    scales = ((torch.rand(N, K // G, device="cuda") + 0.5) * (K ** -0.5) * (16.0 / (top + 1))).to(dtype)
    zeros = torch.randint(0, top + 1, (N, K // G), device="cuda").to(dtype)
    g = torch.arange(K, device="cuda") // G
    deq = (q.float() - zeros.float()[:, g]) * scales.float()[:, g]
    qw = ops.w4_pack(q) if bits == 4 else q.to(torch.uint8)
    sz = torch.stack([scales.t(), zeros.t()], dim=-1).contiguous()
    return qw, scales, zeros, sz, deq


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("M", [1, 7, 32, 33, 64, 100, 128, 129, 256, 300, 1024, 4096])
@pytest.mark.parametrize("N,K,G", [(256, 256, 32), (1000, 512, 128), (264, 96, 32), (512, 1024, 1024), (4096, 4096, 128)])
def test_wq_gemm_matches_fp32(dtype, bits, M, N, K, G):
    from vllm_distributed_amd import ops
    torch.manual_seed(M * 7 + N + bits)
    x = ((torch.rand(M, K, device="cuda") * 2 - 1)).to(dtype)
    qw, scales, zeros, sz, deq = _quantised(N, K, G, bits, dtype)
    ref = x.float() @ deq.t()
    # the kernel rounds each widened weight to the activation type, as a dequantise-then-GEMM path does
    ref16 = x.float() @ deq.to(dtype).float().t()
    tol = 2e-2 * ref.abs().max().item() + 2e-2
    variants = (1, 2, 3) if M <= 128 and K % 128 == 0 else (1,)         # tiled everywhere, both streaming forms where they apply
    for variant in (0,) + variants:
        for splits in (0, 1, 3):
            got = ops.wq_gemm(x, qw, sz, G, bits, splits=splits, variant=variant).float()
            assert (got - ref).abs().max().item() < tol, (variant, splits, (got - ref).abs().max().item())
            assert (got - ref16).abs().max().item() < 0.5 * tol, (variant, splits)


@pytest.mark.parametrize("bits", [4, 8])
def test_wq_gemm_exact_layout(bits):
    """Identity activations return the dequantised rows exactly: a misplaced nibble, byte, scale or zero cannot pass."""
    from vllm_distributed_amd import ops
    N, K, G = 320, 128, 32
    top = (1 << bits) - 1
    q = (torch.arange(N * K, device="cuda").reshape(N, K) * 7 % (top + 1))
    scales = (2.0 ** torch.randint(-3, 2, (N, K // G), device="cuda")).to(torch.bfloat16)
    zeros = torch.randint(0, min(top, 63) + 1, (N, K // G), device="cuda").to(torch.bfloat16)
    g = torch.arange(K, device="cuda") // G
    deq = (q.float() - zeros.float()[:, g]) * scales.float()[:, g]
    qw = ops.w4_pack(q) if bits == 4 else q.to(torch.uint8)
    sz = torch.stack([scales.t(), zeros.t()], dim=-1).contiguous()
    for M, variant in ((16, 1), (16, 2), (16, 3), (40, 2), (40, 3), (128, 1), (128, 2), (512, 1), (1100, 1)):
        idx = torch.arange(M, device="cuda")
        x = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16)
        x[idx, idx % K] = 1
        got = ops.wq_gemm(x, qw, sz, G, bits, splits=1, variant=variant).float()
        assert torch.equal(got, deq.t()[idx % K]), (bits, M, variant)


def test_wq_gemm_bias_strides_and_graph():
    from vllm_distributed_amd import ops
    N, K, G = 1032, 768, 128
    qw, scales, zeros, sz, deq = _quantised(N, K, G, 4, torch.bfloat16)
    b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    full = torch.randn(200, K + 64, device="cuda", dtype=torch.bfloat16)
    x = full[:, 64:]
    ref = x.float() @ deq.t() + b.float()
    for splits in (1, 2):
        got = ops.wq_gemm(x, qw, sz, G, 4, bias=b, splits=splits).float()
        assert (got - ref).abs().max().item() < 2e-2 * ref.abs().max().item() + 2e-2
    xs = torch.randn(48, K, device="cuda", dtype=torch.bfloat16)
    out = ops.wq_gemm(xs, qw, sz, G, 4)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ops.wq_gemm(xs, qw, sz, G, 4)
    xs.copy_(torch.randn(48, K, device="cuda", dtype=torch.bfloat16))
    g.replay()
    torch.cuda.synchronize()
    ref = xs.float() @ deq.t()
    assert (out.float() - ref).abs().max().item() < 2e-2 * ref.abs().max().item() + 2e-2


@pytest.mark.parametrize("bits,act_order", [(4, False), (8, False), (4, True), (8, True)])
def test_gptq_layers_keep_their_weights_packed(bits, act_order):
    """GPTQ tensors -> the packed linear method: the layer's memory is the checkpoint's, decode and mid-sized
    batches run a packed-weight kernel (no expanded copy), large prefill batches a transient expansion plus the library
    GEMM, and every result equals the dequantised product."""
    from tests.test_quant_cpu import _pack_cols, _pack_rows
    from vllm_distributed_amd import ops
    from vllm_distributed_amd.models.dequant import PackedIntLinearMethod, dequantize_gptq, gptq_to_packed
    N, K, G = 768, 1024, 128
    top = (1 << bits) - 1
    q = torch.randint(0, top + 1, (K, N))
    zeros = torch.randint(1, top + 1, (K // G, N))
    scales = ((torch.rand(K // G, N) + 0.5) / (top + 1) / 8).half()
    g_idx = (torch.randperm(K) // G if act_order else torch.arange(K) // G).to(torch.int32)
    ck = {"qweight": _pack_rows(q, bits), "qzeros": _pack_cols(zeros - 1, bits), "scales": scales, "g_idx": g_idx}
    deq = dequantize_gptq(ck["qweight"], ck["qzeros"], scales, g_idx, bits, 1, torch.float32)        # [N, K]
    assert torch.equal(deq, ((q.float() - zeros.float()[g_idx.long()]) * scales.float()[g_idx.long()]).t())
    t = gptq_to_packed(ck["qweight"], ck["qzeros"], scales, g_idx, bits, 1, torch.float16)
    assert ("perm" in t) == act_order

    class Layer(torch.nn.Module):
        input_size = K
    layer = Layer()
    method = PackedIntLinearMethod(G, bits, act_order)
    with torch.device("cuda"):
        method.create_weights(layer, K, N, torch.float16)
    for name, v in t.items():
        getattr(layer, name).data.copy_(v)
    method.process_weights_after_loading(layer)
    assert sum(p.numel() * p.element_size() for p in layer.parameters()) <= N * K * bits // 8 + 4 * N * (K // G) + 4 * K
    seen = []
    real_gemm, real_deq = ops.wq_gemm, ops.wq_dequant
    ops.wq_gemm = lambda *a, **k: (seen.append("gemm"), real_gemm(*a, **k))[1]
    ops.wq_dequant = lambda *a, **k: (seen.append("dequant"), real_deq(*a, **k))[1]
    try:
        for M in (1, 16, 48, 200, 512, 2048):
            x = torch.randn(3, M, K, device="cuda", dtype=torch.float16)[1]
            y = method.apply(layer, x)
            ref = x.float() @ deq.cuda().t()
            assert y.shape == (M, N)
            assert (y.float() - ref).abs().max().item() < 2e-2 * ref.abs().max().item() + 2e-2, M
    finally:
        ops.wq_gemm, ops.wq_dequant = real_gemm, real_deq
    assert seen == ["gemm"] * 5 + ["dequant"]


@pytest.mark.parametrize("tid", [2, 12, 13, 8])
def test_gguf_blocks_run_packed(tid):
    """ggml blocks (Q4_0, Q4_K, Q5_K, Q8_0) -> packed codes with one (scale, zero) pair per 32 weights -> the
    packed-weight kernels, against the float dequantiser of the block type."""
    import numpy as np

    from vllm_distributed_amd import ops
    from vllm_distributed_amd.models import gguf as G
    name, per, nbytes = G.GGML_TYPES[tid]
    rng = np.random.default_rng(tid)
    N, K = 512, 1024
    blocks = rng.integers(0, 256, size=(N * K // per, nbytes), dtype=np.uint8)
    for o in ((0,) if name in ("Q4_0", "Q8_0") else (0, 2)):
        vals = rng.uniform(0.001, 0.01, blocks.shape[0]).astype(np.float16)
        blocks[:, o:o + 2] = np.frombuffer(vals.tobytes(), dtype=np.uint8).reshape(-1, 2)
    deq = G.dequantize(blocks.reshape(-1), tid, (N, K)).cuda()
    bits = 4 if G._BLOCKS[tid][0] == 4 else 8
    t = {k: v.cuda() for k, v in G.packed_tensors(blocks.reshape(-1), tid, (N, K), bits, torch.float16).items()}
    sz = torch.stack([t["scales"].t(), t["qzeros"].t()], dim=-1).contiguous()
    for M in (1, 24, 100, 300, 1024):
        x = torch.randn(M, K, device="cuda", dtype=torch.float16)
        ref = x.float() @ deq.t()
        y = ops.wq_linear(x, t["qweight"], t["scales"], t["qzeros"], sz, 32, bits)
        assert (y.float() - ref).abs().max().item() < 2e-2 * ref.abs().max().item() + 2e-2, (name, M)


def test_mxfp4_rows_come_back_exactly():
    """Microscaling fp4 re-coded as 8-bit integers with power-of-two group scales (models/dequant.py::mxfp4_to_packed8):
    identity activations must return exactly the rows that decoding the fp4 codes gives, in every kernel form."""
    from vllm_distributed_amd import ops
    from vllm_distributed_amd.models.dequant import dequantize_mxfp4, mxfp4_to_packed8
    g = torch.Generator().manual_seed(0)
    N, K = 320, 256
    packed = torch.randint(0, 256, (N, K // 2), generator=g, dtype=torch.int32).to(torch.uint8)
    scale = torch.randint(112, 131, (N, K // 32), generator=g, dtype=torch.int32).to(torch.uint8)     # 2^-15 .. 2^3
    w = mxfp4_to_packed8(packed, scale, torch.bfloat16)
    deq = dequantize_mxfp4(packed, scale, torch.float32).cuda()
    qw, sc, z = (w[k].cuda() for k in ("qweight", "scales", "qzeros"))
    sz = torch.stack([sc.t(), z.t()], dim=-1).contiguous()
    for M, variant in ((16, 1), (16, 2), (16, 3), (128, 2), (512, 1)):
        idx = torch.arange(M, device="cuda")
        x = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16)
        x[idx, idx % K] = 1
        got = ops.wq_gemm(x, qw, sz, 32, 8, splits=1, variant=variant).float()
        assert torch.equal(got, deq.t()[idx % K]), (M, variant)


def _codebook(N, K, G, table):
    g = torch.Generator().manual_seed(N + K + G)
    codes = torch.randint(0, 256, (N, K // 2), generator=g, dtype=torch.int32).to(torch.uint8)
    scales = (torch.rand(N, K // G, generator=g) + 0.5) * K ** -0.5
    return codes.cuda(), scales.cuda(), torch.tensor(table, dtype=torch.float32).cuda()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("G,hi_first,table", [(64, True, "nf4"), (16, False, "e2m1"), (32, False, "e2m1"),
                                              (128, True, "nf4"), (256, True, "e2m1")])
def test_codebook4_kernels_match_fp32(dtype, G, hi_first, table):
    """4-bit code-book weights (ops/csrc/cb4.hip): the expansion kernel equals table[code] * scale rounded once, the
    skinny kernel (1 .. 64 rows) and the expand + library path (more rows) equal the fp32 product, bias and strided
    activations included."""
    from vllm_distributed_amd import ops
    from vllm_distributed_amd.layers.quant import NF4_CODE
    from vllm_distributed_amd.models.dequant import _E2M1
    from vllm_distributed_amd.ops import reference as R
    N, K = 528, 1024
    codes, scales, tab = _codebook(N, K, G, NF4_CODE if table == "nf4" else _E2M1)
    deq = R.cb4_dequant(codes, scales, tab, G, hi_first, torch.float32)
    got = ops.cb4_dequant(codes, scales, tab, G, hi_first, dtype)
    if dtype == torch.bfloat16:
        assert torch.equal(got, deq.to(dtype))
    else:       # fp16: the kernel's fused multiply-convert rounds the exact product once, the reference twice
        torch.testing.assert_close(got.float(), deq, rtol=2 ** -11, atol=2 ** -24)
    bias = torch.randn(N, device="cuda", dtype=dtype)
    forms = [(codes, scales, False)]
    if ops.cb4_chunk_layout_ok(K, G):        # the layout whose kernel scales accumulators instead of weights
        forms.append((*ops.cb4_to_chunk_layout(codes, scales), True))
        got = ops.cb4_dequant(forms[1][0], forms[1][1], tab, G, hi_first, dtype, chunked=True)
        assert torch.equal(got, ops.cb4_dequant(codes, scales, tab, G, hi_first, dtype))
    else:
        assert G == 16
    seen = []
    real = ops.cb4_dequant
    ops.cb4_dequant = lambda *a, **k: (seen.append(1), real(*a, **k))[1]
    try:
        for c, sc, chunked in forms:
            for M in (1, 7, 16, 17, 33, 48, 64, 100):
                x = torch.randn(M, K + 32, device="cuda", dtype=dtype)[:, 32:]
                ref = x.float() @ deq.t()
                tol = 2e-2 * ref.abs().max().item() + 2e-2
                got = ops.cb4_linear(x, c, sc, tab, G, hi_first, chunked=chunked)
                assert got.shape == (M, N) and (got.float() - ref).abs().max().item() < tol, (M, chunked)
                got = ops.cb4_linear(x, c, sc, tab, G, hi_first, bias, chunked=chunked)
                assert (got.float() - ref - bias.float()).abs().max().item() < tol, (M, chunked)
    finally:
        ops.cb4_dequant = real
    assert len(seen) == 2 * len(forms)       # only the 100-row batches expanded the weights


@pytest.mark.parametrize("G,hi_first", [(16, False), (64, True)])
def test_codebook4_rows_come_back_exactly(G, hi_first):
    """Identity activations return exactly the expanded rows: a misplaced nibble, byte or scale cannot pass."""
    from vllm_distributed_amd import ops
    from vllm_distributed_amd.layers.quant import NF4_CODE
    from vllm_distributed_amd.ops import reference as R
    N, K = 320, 256
    codes, scales, tab = _codebook(N, K, G, NF4_CODE)
    scales = torch.exp2(torch.randint(-6, 3, scales.shape, device="cuda").float())
    deq = R.cb4_dequant(codes, scales, tab, G, hi_first, torch.bfloat16).float()
    chunked = ops.cb4_chunk_layout_ok(K, G)
    if chunked:         # power-of-two scales: scaling the 16-bit table value equals rounding the scaled value
        for M in (16, 64):
            idx = torch.arange(M, device="cuda")
            x = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16)
            x[idx, (idx * 5) % K] = 1
            got = ops.cb4_linear(x, *ops.cb4_to_chunk_layout(codes, scales), tab, G, hi_first, chunked=True).float()
            assert torch.equal(got, deq.t()[(idx * 5) % K]), M
    for M in (16, 40, 64):
        idx = torch.arange(M, device="cuda")
        x = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16)
        x[idx, (idx * 5) % K] = 1
        got = ops.cb4_linear(x, codes, scales, tab, G, hi_first).float()
        assert torch.equal(got, deq.t()[(idx * 5) % K]), M
    if chunked:
        codes, scales = ops.cb4_to_chunk_layout(codes, scales)
    xs = torch.randn(24, K, device="cuda", dtype=torch.bfloat16)
    out = ops.cb4_linear(xs, codes, scales, tab, G, hi_first, chunked=chunked)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ops.cb4_linear(xs, codes, scales, tab, G, hi_first, chunked=chunked)
    xs.copy_(torch.randn(24, K, device="cuda", dtype=torch.bfloat16))
    g.replay()
    torch.cuda.synchronize()
    ref = xs.float() @ deq.t()
    assert (out.float() - ref).abs().max().item() < 2e-2 * ref.abs().max().item() + 2e-2


@pytest.mark.parametrize("G", [32, 64, 128])
def test_codebook4_wide_layers_share_activation_registers(G):
    """Layers wide enough for 512 workgroups of 32 / 64 rows take the forms whose activation registers serve two / four
    row sets (ops/csrc/cb4.hip, RW): same results as the fp32 product, bias included."""
    from vllm_distributed_amd import ops
    from vllm_distributed_amd.layers.quant import NF4_CODE
    from vllm_distributed_amd.ops import reference as R
    N, K = 32768, 384
    codes, scales, tab = _codebook(N, K, G, NF4_CODE)
    deq = R.cb4_dequant(codes, scales, tab, G, True, torch.float32)
    c, st = ops.cb4_to_chunk_layout(codes, scales)
    bias = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    for M in (1, 8, 16, 20, 32, 40):                     # four row sets up to 16 rows, two up to 32, one beyond
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        ref = x.float() @ deq.t() + bias.float()
        got = ops.cb4_linear(x, c, st, tab, G, True, bias, chunked=True)
        assert (got.float() - ref).abs().max().item() < 2e-2 * ref.abs().max().item() + 2e-2, M
        got = ops.cb4_linear(x, codes, scales, tab, G, True, bias)          # the per-weight form has row sets too
        assert (got.float() - ref).abs().max().item() < 2e-2 * ref.abs().max().item() + 2e-2, M
    codes16, scales16, _ = _codebook(N, K, 16, NF4_CODE)
    deq = R.cb4_dequant(codes16, scales16, tab, 16, False, torch.float32)
    for M in (3, 16, 30, 50):
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        ref = x.float() @ deq.t()
        got = ops.cb4_linear(x, codes16, scales16, tab, 16, False)
        assert (got.float() - ref).abs().max().item() < 2e-2 * ref.abs().max().item() + 2e-2, M


@pytest.mark.parametrize("N,K,G", [(528, 1024, 64), (8192, 4096, 64), (1024, 8192, 128), (16384, 2048, 32)])
def test_codebook4_k_splits(N, K, G):
    """K splits of the chunk-layout kernel: every split writes a slab of fp32 partial sums, a second kernel adds the
    slabs and the bias.  Forced and automatic splits, calls back to back, inside a captured graph too."""
    from vllm_distributed_amd import ops
    from vllm_distributed_amd.layers.quant import NF4_CODE
    from vllm_distributed_amd.ops import reference as R
    codes, scales, tab = _codebook(N, K, G, NF4_CODE)
    deq = R.cb4_dequant(codes, scales, tab, G, True, torch.float32)
    c, st = ops.cb4_to_chunk_layout(codes, scales)
    bias = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    for M in (1, 16, 24, 64):
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        ref = x.float() @ deq.t() + bias.float()
        tol = 2e-2 * ref.abs().max().item() + 2e-2
        for splits in (0, 2, 3, 8, 0):
            got = ops.cb4_linear(x, c, st, tab, G, True, bias, chunked=True, splits=splits)
            assert (got.float() - ref).abs().max().item() < tol, (M, splits)
    x = torch.randn(16, K, device="cuda", dtype=torch.bfloat16)
    out = ops.cb4_linear(x, c, st, tab, G, True, chunked=True, splits=4)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ops.cb4_linear(x, c, st, tab, G, True, chunked=True, splits=4)
    for _ in range(3):
        x.copy_(torch.randn(16, K, device="cuda", dtype=torch.bfloat16))
        g.replay()
        torch.cuda.synchronize()
        ref = x.float() @ deq.t()
        assert (out.float() - ref).abs().max().item() < 2e-2 * ref.abs().max().item() + 2e-2
