"""``-m gpu``: the MFMA product sg_gemm_nt (csrc/gemm_mfma.hip) -- the dense per-vertex feature x weight GEMM of
ChebConv.forward [3P] (util/networks.py:42,49) and its input gradient -- against fp32 matmul on the same bf16 inputs,
bit-exact on integer data, plus the BatchNorm tile moments it leaves behind and the layer-level wiring."""
import os

import numpy as np
import pytest
import torch

import golden_util as GU
from semigcn_amd import capi, functional as F_sg, nn as sgnn, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(a, b, bias=None):
    r = a.float() @ b.float().t()
    return r if bias is None else r + bias


SHAPES = [  # (M, N, K)
    (1, 16, 8), (127, 16, 48), (128, 24, 16), (129, 32, 96), (1000, 48, 32), (4097, 64, 72), (300, 96, 64),
    (513, 128, 192), (2050, 192, 128), (777, 256, 384), (640, 384, 256), (300, 512, 768), (260, 768, 512),
]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_nt_integer_data_is_bit_exact(M, N, K):
    """Small-integer operands: every product and partial sum is exact in fp32, so the result must EQUAL the
    reference (rounded once to bf16) bit for bit -- catches any fragment / accumulator layout or K-tail mistake.
    B is asymmetric (A = I with a symmetric B would hide a transposed C write: CDNA guide section 3)."""
    g = torch.Generator(device=DEV).manual_seed(M * 7 + N * 3 + K)
    lim = 2 if K > 64 else 3
    a = torch.randint(-lim, lim + 1, (M, K), device=DEV, generator=g).to(torch.bfloat16)
    b = torch.randint(-1, 2, (N, K), device=DEV, generator=g).to(torch.bfloat16)
    b[:, 0] = torch.arange(N, device=DEV).remainder(5).to(torch.bfloat16) - 2       # column pattern that is not symmetric
    ref = _ref(a, b)                                          # exact integers in fp32
    out = capi.gemm_nt(a, b)
    assert out.dtype == torch.bfloat16 and torch.equal(out, ref.to(torch.bfloat16))
    if N > 64:                                                # the 64 x 256 tile variant behind the tuning knob
        capi.tuning_set(capi.TUNE_GEMM_TILE, 2)
        try:
            out2, mom2 = capi.gemm_nt(a, b, moments=True)
        finally:
            capi.tuning_set(capi.TUNE_GEMM_TILE, 0)
        assert torch.equal(out2, out) and mom2.shape[0] == (M + 63) // 64
        assert torch.allclose(mom2[0, 0], out[:64].float().mean(0), atol=1e-4)
    # identity rows pick out columns of B: out[i] == B[:, i]
    if M >= K:
        eye = torch.zeros(M, K, device=DEV, dtype=torch.bfloat16)
        eye[torch.arange(K), torch.arange(K)] = 1
        assert torch.equal(capi.gemm_nt(eye, b)[:K].float(), b.float().t())


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_nt_random_data_bias_strides_and_moments(M, N, K):
    g = torch.Generator(device=DEV).manual_seed(N + K)
    wide = torch.randn(M, K + 24, device=DEV, generator=g).to(torch.bfloat16)
    a = wide[:, 8:8 + K]                                       # a column block of a wider buffer (row stride K + 24)
    b = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    ref = _ref(a, b, bias)
    outw = torch.full((M, N + 16), 7.0, device=DEV, dtype=torch.bfloat16)
    out, mom = capi.gemm_nt(a, b, bias, out=outw[:, 8:8 + N], moments=True)
    # one rounding of the fp32 result to bf16 (2^-9 relative) + fp32 summation-order noise
    err = (out.float() - ref).abs().max() / ref.abs().max()
    assert float(err) < 2.0 ** -8, float(err)
    assert GU.rel_l2(out.float().cpu(), ref.cpu()) < 2.0 ** -8
    assert bool((outw[:, :8] == 7).all()) and bool((outw[:, 8 + N:] == 7).all())      # neighbours untouched
    # tile moments == moments of the ROUNDED output over rows [R t, R t + R)
    R = capi.gemm_tile_rows(N)
    assert R == 128 and mom.shape == ((M + R - 1) // R, 2, N)
    o = out.float()
    for t in range(mom.shape[0]):
        blk = o[t * R:(t + 1) * R]
        mean = blk.mean(0)
        m2 = ((blk - mean) ** 2).sum(0)
        assert float((mom[t, 0] - mean).abs().max()) <= 1e-5 * max(float(mean.abs().max()), 1.0)
        assert float((mom[t, 1] - m2).abs().max()) <= 1e-4 * max(float(m2.max()), 1e-3)
    # merged with the tile finalizer == BatchNorm statistics of the output
    if M > 1:
        gam, bet = torch.rand(N, device=DEV) + 0.5, torch.randn(N, device=DEV)
        fin = capi.bn_stats_finalize_tiles(mom, R, M, gam, bet, None, None, 0.1, 1e-5)
        var, mean = torch.var_mean(o.double(), dim=0, unbiased=False)
        assert float((fin[0].double() - mean).abs().max()) < 1e-5
        assert float((fin[1].double() - (var + 1e-5).rsqrt()).abs().max() / (var + 1e-5).rsqrt().abs().max()) < 1e-4


def test_gemm_nt_rejects_what_it_cannot_take():
    a = torch.zeros(64, 12, device=DEV, dtype=torch.bfloat16)
    b = torch.zeros(16, 12, device=DEV, dtype=torch.bfloat16)
    assert not capi.gemm_nt_supported(a, b, 16)                 # K = 12 (the 4 -> 16 layer's [V, 12] operand)
    with pytest.raises(capi.SemigcnLibraryError, match="multiples of 8"):
        capi.gemm_nt(a, b)
    a32 = torch.zeros(64, 16, device=DEV)
    assert not capi.gemm_nt_supported(a32, a32, 16)
    # the host wrapper routes such products to the BLAS library instead
    y = F_sg.dense_nt(a, b)
    assert y.shape == (64, 16) and y.dtype == torch.bfloat16


@pytest.mark.parametrize("cin,cout", [(32, 64), (64, 16), (16, 32), (256, 128)])
def test_chebconv_bf16_layer_same_result_on_mfma_and_blas_paths(cin, cout):
    """One ChebConv + BatchNorm + activation block, bf16 features, forward and backward: the MFMA path (own GEMM, BatchNorm
    moments from its epilogue) against the BLAS path (hipBLASLt + separate moments pass) -- same math, a few bf16
    roundings apart.  The activation slope is 1 (identity) so that no LeakyReLU kink amplifies a rounding difference;
    the fused BatchNorm+activation kernels run all the same."""
    m = synth.torus_mesh(40, 24)
    ei = torch.from_numpy(m.edge_index).to(DEV)
    seq = sgnn.Sequential("x, edge_index", [(sgnn.ChebConv(cin, cout, K=3), "x, edge_index -> x"), torch.nn.BatchNorm1d(cout),
                                            torch.nn.LeakyReLU(negative_slope=1.0)])
    GU.fill_state(seq, seed=cin + cout)
    seq.to(DEV).train()
    x0 = torch.randn(m.num_vertices, cin, device=DEV).to(torch.bfloat16)
    r = torch.randn(m.num_vertices, cout, device=DEV).to(torch.bfloat16)
    res = {}
    limit = F_sg.MFMA_MAX_WEIGHT_ELEMS
    for mfma in (True, False):
        F_sg.USE_MFMA_GEMM = mfma
        F_sg.MFMA_MAX_WEIGHT_ELEMS = 1 << 30                 # every product of the layer on the kernel under test
        try:
            for mod in seq.modules():
                if isinstance(mod, sgnn.ChebConv):
                    mod.invalidate_weight_cache()
            seq.module_1.reset_running_stats()
            seq.zero_grad()
            x = x0.clone().requires_grad_(True)
            y = seq(x, ei)
            (y.float() * r.float()).sum().backward()
            # (module_0.bias is left out: BatchNorm removes constant shifts, so its true gradient is exactly zero and
            #  what autograd returns is the bf16 rounding noise of the column sums, ~2^-9 sqrt(V) |dH|, on both paths)
            res[mfma] = (y.detach().float(), x.grad.float(),
                         [p.grad.clone() for n, p in seq.named_parameters() if n != "module_0.bias"],
                         seq.module_1.running_mean.clone(), seq.module_1.running_var.clone())
        finally:
            F_sg.USE_MFMA_GEMM = True
            F_sg.MFMA_MAX_WEIGHT_ELEMS = limit
    a, b = res[True], res[False]
    assert GU.rel_l2(a[0].cpu(), b[0].cpu()) < 2e-2 and GU.rel_l2(a[1].cpu(), b[1].cpu()) < 3e-2
    for pa, pb in zip(a[2], b[2]):
        scale = max(float(pb.norm()), 1e-3 * float(max(q.abs().max() for q in b[2])) * pb.numel() ** 0.5)
        assert float((pa - pb).norm()) <= 3e-2 * scale
    assert GU.rel_l2(a[3].cpu(), b[3].cpu()) < 1e-2 and GU.rel_l2(a[4].cpu(), b[4].cpu()) < 1e-2


TN_SHAPES = [  # (M, N, Kp): dW[N, Kp] = A[M, N]^T B[M, Kp]
    (1, 16, 8), (63, 16, 48), (64, 24, 16), (65, 32, 96), (1000, 48, 32), (8191, 64, 72), (8193, 96, 64),
    (20000, 128, 192), (16385, 192, 128), (9000, 256, 384), (3000, 384, 256), (2500, 512, 768), (1300, 768, 512),
]


@pytest.mark.parametrize("M,N,Kp", TN_SHAPES)
def test_gemm_tn_integer_data_is_bit_exact(M, N, Kp):
    """Weight-gradient product on the transposing-LDS-read kernel, small-integer operands (every partial sum exact in
    fp32): must EQUAL the fp32 reference bit for bit; both operands asymmetric, several slabs, ragged M / N / Kp."""
    g = torch.Generator(device=DEV).manual_seed(M + 3 * N + 7 * Kp)
    a = torch.randint(-2, 3, (M, N), device=DEV, generator=g).to(torch.bfloat16)
    b = torch.randint(-2, 3, (M, Kp), device=DEV, generator=g).to(torch.bfloat16)
    a[:, 0] = 1
    b[:, -1] = torch.arange(M, device=DEV).remainder(3).to(torch.bfloat16)
    ref = a.float().t() @ b.float()
    out = capi.gemm_tn(a, b)
    assert out.dtype == torch.float32 and out.shape == (N, Kp) and torch.equal(out, ref)


@pytest.mark.parametrize("M,N,Kp", TN_SHAPES[3:])
def test_gemm_tn_random_data_and_strides(M, N, Kp):
    g = torch.Generator(device=DEV).manual_seed(N + Kp)
    wa = torch.randn(M, N + 8, device=DEV, generator=g).to(torch.bfloat16)
    wb = torch.randn(M, Kp + 16, device=DEV, generator=g).to(torch.bfloat16)
    a, b = wa[:, 8:], wb[:, :Kp]                                  # column blocks of wider buffers
    ref = a.double().t() @ b.double()
    out = capi.gemm_tn(a, b)
    assert GU.rel_l2(out.double().cpu(), ref.cpu()) < 1e-5          # fp32 accumulation of exact bf16 products
    assert torch.equal(out, capi.gemm_tn(a, b))                    # deterministic
    assert GU.rel_l2(F_sg.weight_grad(a.contiguous(), b.contiguous()).double().cpu(), ref.cpu()) < 1e-5     # own kernel or BLAS slabs, by shape


# ---------------------------------------------------------------------------------------------------------------------
# the persistent 256 x 256-tile kernel (csrc/gemm_mfma256.hip): the compute-bound products
# ---------------------------------------------------------------------------------------------------------------------
BIG_SHAPES = [  # (M, N, K): one partial tile .. many tiles per workgroup stream, 2 .. 12 K steps, 1 .. 3 column tiles
    (256, 256, 128), (300, 256, 256), (1000, 512, 128), (4097, 256, 768), (70001, 512, 768), (70001, 768, 256),
    (131072, 256, 384), (200003, 768, 512), (1_000_000, 512, 768),
]


def _with_tile(tile, fn):
    capi.tuning_set(capi.TUNE_GEMM_TILE, tile)
    try:
        return fn()
    finally:
        capi.tuning_set(capi.TUNE_GEMM_TILE, 0)


@pytest.mark.parametrize("M,N,K", BIG_SHAPES)
def test_big_tile_gemm_integer_data_is_bit_exact(M, N, K):
    """Exact integers (see test_gemm_nt_integer_data_is_bit_exact): every LDS-DMA chunk, swizzled fragment read, swapped
    MFMA operand and 8-byte store lands where it must, on every workgroup stream, across tile boundaries of the persistent
    loop, with rows past M clamped on the way in and masked on the way out.  Repeated: the counted-vmcnt / raw-barrier
    pipeline is a race screen as much as a layout check (guide 5: "screen it for races over many runs at several sizes")."""
    assert capi.gemm_nt_takes_big_tile(M, N, K, K, K, N) == (K * N > 90_000 and M >= 16384)
    g = torch.Generator(device=DEV).manual_seed(M + 3 * N + K)
    lim = 2 if K > 64 else 3
    a = torch.randint(-lim, lim + 1, (M, K), device=DEV, generator=g).to(torch.bfloat16)
    b = torch.randint(-1, 2, (N, K), device=DEV, generator=g).to(torch.bfloat16)
    b[:, 0] = torch.arange(N, device=DEV).remainder(5).to(torch.bfloat16) - 2
    a[:, 1] = torch.arange(M, device=DEV).remainder(3).to(torch.bfloat16) - 1          # rows differ: a shifted tile shows
    ref = _with_tile(1, lambda: capi.gemm_nt(a, b))            # the 128-row kernel (itself pinned to fp32 matmul above)
    if M <= 300_000:
        assert torch.equal(ref, _ref(a, b).to(torch.bfloat16))
    for rep in range(4 if M <= 300_000 else 2):
        out = torch.full((M + 3, N), 9.0, device=DEV, dtype=torch.bfloat16)
        _with_tile(3, lambda: capi.gemm_nt(a, b, out=out[:M]))
        assert torch.equal(out[:M], ref), (rep, int((out[:M] != ref).sum()))
        assert bool((out[M:] == 9).all())                      # nothing written past row M


@pytest.mark.parametrize("M,N,K", [(4097, 256, 768), (70001, 512, 768), (200003, 768, 512)])
def test_big_tile_gemm_random_data_bias_and_strides(M, N, K):
    g = torch.Generator(device=DEV).manual_seed(N + K + 1)
    wide = torch.randn(M, K + 24, device=DEV, generator=g).to(torch.bfloat16)
    a = wide[:, 8:8 + K]
    b = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    ref = _ref(a, b, bias)
    outw = torch.full((M, N + 16), 7.0, device=DEV, dtype=torch.bfloat16)
    out = _with_tile(3, lambda: capi.gemm_nt(a, b, bias, out=outw[:, 8:8 + N]))
    assert float((out.float() - ref).abs().max() / ref.abs().max()) < 2.0 ** -8
    assert bool((outw[:, :8] == 7).all()) and bool((outw[:, 8 + N:] == 7).all())
    # the same values as the 128-row kernel up to fp32 summation order (same bf16 rounding of nearly the same sums)
    out128 = _with_tile(1, lambda: capi.gemm_nt(a, b, bias))
    assert float((out.float() - out128.float()).abs().max() / ref.abs().max()) < 2.0 ** -7
    assert float((out != out128).float().mean()) < 0.01


def test_big_tile_kernel_serves_the_wide_layers_of_the_model():
    """functional.dense_nt hands the K x N > 100 K products to the 256 x 256 kernel and nothing goes to the BLAS library:
    a 256 -> 512 ChebConv layer (forward [V,768] x [768,512], input gradient [V,512] x [512,768]) on 70 K vertices against the
    same layer with the kernel switched off."""
    m = synth.torus_mesh(280, 250, masks=False)
    ei = torch.from_numpy(m.edge_index).to(DEV)
    conv = sgnn.ChebConv(256, 512, K=3)
    GU.fill_state(conv, seed=8)
    conv.to(DEV)
    x = torch.randn(m.num_vertices, 256, device=DEV).bfloat16().requires_grad_(True)
    r = torch.randn(m.num_vertices, 512, device=DEV).bfloat16()
    timer = capi.LaunchTimer()
    F_sg.set_gemm_timer(timer)
    try:
        y = conv(x, ei)
        (y * r).sum().backward()
    finally:
        F_sg.set_gemm_timer(None)
    torch.cuda.synchronize()
    engines = {(k[0], k[2], k[3]): k[5] for k in timer.results()}
    assert engines[("nt", 512, 768)] == "mfma" and engines[("nt", 768, 512)] == "mfma", engines
    gx, gw = x.grad.clone(), conv.lins[1].weight.grad.clone()
    x.grad = None
    conv.zero_grad()
    old = F_sg.USE_MFMA_BIG_TILE
    F_sg.USE_MFMA_BIG_TILE = False
    try:
        y2 = conv(x, ei)
        (y2 * r).sum().backward()
    finally:
        F_sg.USE_MFMA_BIG_TILE = old
    assert GU.rel_l2(y.detach().float().cpu(), y2.detach().float().cpu()) < 2.0 ** -8
    assert GU.rel_l2(gx.float().cpu(), x.grad.float().cpu()) < 2.0 ** -7
    assert GU.rel_l2(gw.cpu(), conv.lins[1].weight.grad.cpu()) < 2.0 ** -7


BIG_TN_SHAPES = [  # (M, N, Kp): fewer 64-row steps than slabs .. ~190 steps per slab, a tail of 0 .. 63 rows, 1 .. 9 output tiles
    (4096, 256, 256), (4113, 256, 256), (70001, 256, 768), (131077, 512, 768), (200003, 768, 512), (1_000_000, 256, 768),
    (300_011, 768, 768),
]


@pytest.mark.parametrize("M,N,Kp", BIG_TN_SHAPES)
def test_big_tile_weight_gradient_integer_data_is_bit_exact(M, N, Kp):
    """dW = A^T B on the 256 x 256 ring (csrc/gemm_mfma256.hip, gemm_tn_256): exact integers, so the fp32 result must
    EQUAL the 128 x 128 kernel's (and fp32 matmul's) whatever the slab cut and summation order -- every DMA chunk, source
    swizzle, transposing fragment read and partial-tile store in its place; rows past the last full 64-row step go
    through the 128 x 128 kernel's slab.  Repeated as a race screen."""
    assert capi.gemm_tn_takes_big_tile(M, N, Kp, N, Kp) == (M >= 16384)
    g = torch.Generator(device=DEV).manual_seed(M + N + 7 * Kp)
    a = torch.randint(-2, 3, (M, N), device=DEV, generator=g).to(torch.bfloat16)
    b = torch.randint(-1, 2, (M, Kp), device=DEV, generator=g).to(torch.bfloat16)
    a[:, 1] = torch.arange(M, device=DEV).remainder(3).to(torch.bfloat16) - 1
    b[:, 0] = torch.arange(M, device=DEV).remainder(5).to(torch.bfloat16) - 2
    ref = _with_tile(1, lambda: capi.gemm_tn(a, b))
    if M <= 200_003:
        assert torch.equal(ref, a.float().t() @ b.float())
    for rep in range(3):
        out = _with_tile(3, lambda: capi.gemm_tn(a, b))
        assert out.dtype == torch.float32 and torch.equal(out, ref), (rep, int((out != ref).sum()))


def test_big_tile_weight_gradient_random_data_and_strides():
    M, N, Kp = 70001, 512, 768
    g = torch.Generator(device=DEV).manual_seed(11)
    wa = torch.randn(M, N + 16, device=DEV, generator=g).to(torch.bfloat16)
    wb = torch.randn(M, Kp + 8, device=DEV, generator=g).to(torch.bfloat16)
    a, b = wa[:, 8:8 + N], wb[:, :Kp]
    ref = a.double().t() @ b.double()
    out = _with_tile(3, lambda: capi.gemm_tn(a, b))
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 1e-5       # fp32 accumulation of exact bf16 products
    out2 = _with_tile(3, lambda: capi.gemm_tn(a, b))
    assert torch.equal(out, out2)                                                 # deterministic: fixed slab order


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("V,N,K", [(1, 1, 1), (257, 16, 12), (70001, 12, 16), (100000, 3, 16), (4099, 16, 3), (5000, 16, 16), (333, 7, 5),
                                   (50000, 3, 32), (18001, 32, 3), (777, 8, 32), (777, 32, 8), (300, 8, 17), (300, 5, 29)])
def test_thin_products_match_float64(V, N, K, dtype):
    """sg_thin_nt / sg_thin_tn (weight matrices of at most 256 entries -- K <= 8 with N <= 32, N, K <= 16, K <= 32 with N <= 8:
    the 4 -> 16 input layer, the 16 -> 3 output layer, the 32 -> 3 heads of the MGCN and their autograd) against float64: fp32 accumulation error only, on strided operands, with and without bias; the
    weight gradient deterministic run to run."""
    from semigcn_amd import capi
    assert capi.thin_shape(N, K) and not capi.thin_shape(17, 17) and not capi.thin_shape(9, 32) and not capi.thin_shape(33, 8) \
        and not capi.thin_shape(4, 33)
    g = torch.Generator(device=DEV).manual_seed(V + 31 * N + K)
    wide = torch.randn(V, K + 5, device=DEV, generator=g).to(dtype)
    x = wide[:, 2:2 + K]                                           # row stride K + 5, unit column stride
    w = torch.randn(N, K, device=DEV, generator=g)
    b = torch.randn(N, device=DEV, generator=g)
    out_wide = torch.full((V, N + 3), 7.0, device=DEV, dtype=dtype)
    y = capi.thin_nt(x, w, b, out=out_wide[:, :N])
    assert y.data_ptr() == out_wide.data_ptr() and bool((out_wide[:, N:] == 7.0).all())
    want = x.double() @ w.double().t() + b.double()
    tol = 2e-6 if dtype == torch.float32 else 2.0 ** -8
    assert float((y.double() - want).abs().max()) <= tol * float(want.abs().max() + 1.0)
    y0 = capi.thin_nt(x, w)
    assert float((y0.double() - x.double() @ w.double().t()).abs().max()) <= tol * float(want.abs().max() + 1.0)
    a = torch.randn(V, N, device=DEV, generator=g).to(dtype)
    dw = capi.thin_tn(a, x)
    want_w = a.double().t() @ x.double()
    assert dw.dtype == torch.float32 and dw.shape == (N, K)
    assert float((dw.double() - want_w).abs().max()) <= 2e-6 * float((a.double().abs().t() @ x.double().abs()).max() + 1.0)
    assert torch.equal(dw, capi.thin_tn(a, x))


def test_bf16_training_iteration_uses_no_blas_product():
    """With bf16 features every dense product of an SGCN training iteration is served by the library's own kernels (MFMA
    or thin): the per-call records of functional's timer name no "blas" engine."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from semigcn_amd import networks, train
    import bench
    mesh = bench.make_mesh(160, 128, "survey")           # >= 16 K vertices: the wide layers take the 256 x 256 ring
    batch = bench.build_mesh_batch(mesh, DEV, 5)
    torch.manual_seed(0)
    net = networks.SingleScaleGCN(DEV).to(DEV).set_feature_dtype(torch.bfloat16)
    tr = train.SGCNTrainer(net, batch)
    tr.iteration_step()
    timer = capi.LaunchTimer()
    F_sg.set_gemm_timer(timer)
    try:
        tr.iteration_step()
        torch.cuda.synchronize()
    finally:
        F_sg.set_gemm_timer(None)
    keys = list(timer.results())
    assert keys and not [k for k in keys if k[-1] == "blas"], [k for k in keys if k[-1] == "blas"]
    assert {"mfma", "thin"} <= {k[-1] for k in keys}
