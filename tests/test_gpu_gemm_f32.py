"""``-m gpu``: the dense products at the reference's OWN precision -- float32 features, float32 parameters
(util/networks.py:40-53; BASELINE configs c2 / c3; [3P] ChebConv ``lins[k]`` and their autograd) -- on the bf16 matrix cores
(csrc/gemm_split.hip: every operand as three bf16 pieces, six piece products).  Small-integer operands must come out bit
for bit (every fragment / piece / ring-slot mistake shows); on random data the error against a float64 product must be
no larger than the BLAS library's own float32 MFMA product on the same operands; and the layer-level wiring must put
the products of a float32 training iteration on these kernels."""
import pytest
import torch

from semigcn_amd import capi, functional as F_sg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (M, N, K): 256- and 128-column tiles, ragged M / N, one and many K blocks, a tile stream with several tiles per workgroup
# (below 16 384 rows: 128 x 128 tiles on four wavefronts, round 6 -- the reference's own mesh sizes; from there on 256-row tiles)
NT_SHAPES = [(256, 64, 64), (1000, 256, 64), (4133, 384, 96), (2048, 192, 128), (777, 512, 256), (5000, 64, 768),
             (70001, 256, 768), (66000, 768, 512), (300, 100, 160), (33000, 128, 192), (130, 64, 64), (5000, 768, 512),
             (10800, 256, 768), (16383, 192, 128)]


def _ints(shape, lim, g):
    return torch.randint(-lim, lim + 1, shape, device=DEV, generator=g).float()


@pytest.fixture(params=[0, 32], ids=["rows-pick-the-tile", "256-row-tiles-only"])
def tile_rule(request):
    """SG_TUNE_F32_ENGINE bit 5 switches the 128-row variant off: both kernels see the small shapes."""
    capi.tuning_set(capi.TUNE_F32_ENGINE, request.param)
    yield request.param
    capi.tuning_set(capi.TUNE_F32_ENGINE, 0)


@pytest.mark.parametrize("M,N,K", NT_SHAPES)
def test_split_nt_integer_data_is_bit_exact(M, N, K, tile_rule):
    if tile_rule and M < 256:
        pytest.skip("the 256-row kernel needs 256 rows")
    g = torch.Generator(device=DEV).manual_seed(M + 3 * N + 7 * K)
    a, w, bias = _ints((M, K), 8, g), _ints((N, K), 8, g), _ints((N,), 8, g)
    w[:, 0] = torch.arange(N, device=DEV).remainder(5).float() - 2           # asymmetric: a transposed write would show
    ref = (a.double() @ w.double().t() + bias.double()).float()
    assert capi.gemm_nt_f32_supported(a, N)
    out = capi.gemm_nt_f32(a, w, bias)
    assert torch.equal(out, ref)
    # the input-gradient form: the same matrix stored [K, N], no transposed copy
    assert torch.equal(capi.gemm_nt_f32(a, w.t().contiguous(), bias, w_is_kn=True), ref)
    # race screen: the ring's counted waits leave three K blocks in flight -- repeat and compare
    for _ in range(3):
        assert torch.equal(capi.gemm_nt_f32(a, w, bias), out)
    # values that need all three pieces (24 significant bits), exact products by construction: x * 2^k
    a2 = (torch.randint(1, 1 << 24, (M, K), device=DEV, generator=g).float() * 2.0 ** -24)
    e = torch.zeros((N, K), device=DEV)
    e[torch.arange(N, device=DEV), torch.arange(N, device=DEV) % K] = 4.0    # one non-zero per row of W: out[i, n] = 4 a[i, n % K]
    assert torch.equal(capi.gemm_nt_f32(a2, e), 4.0 * a2[:, torch.arange(N, device=DEV) % K])


@pytest.mark.parametrize("M,N,K", NT_SHAPES[:8] + NT_SHAPES[11:13])
def test_split_nt_random_data_strides_and_error_vs_float64(M, N, K):
    g = torch.Generator(device=DEV).manual_seed(N + K)
    wide = torch.randn((M, K + 24), device=DEV, generator=g)
    a = wide[:, 8:8 + K]                                      # a column block of a wider buffer (16-byte aligned, stride K + 24)
    w = torch.randn((N, K), device=DEV, generator=g) * 0.1
    bias = torch.randn((N,), device=DEV, generator=g)
    outw = torch.zeros((M, N + 12), device=DEV)
    out = outw[:, 4:4 + N]
    assert capi.gemm_nt_f32_supported(a, N, out.stride(0))
    capi.gemm_nt_f32(a, w, bias, out=out)
    assert torch.all(outw[:, :4] == 0) and torch.all(outw[:, 4 + N:] == 0)      # nothing written outside the block
    r64 = a.double() @ w.double().t() + bias.double()
    den = a.double().abs() @ w.double().abs().t() + bias.double().abs()
    e_own = ((out.double() - r64).abs() / den).max().item()
    e_lib = (((a.contiguous() @ w.t() + bias).double() - r64).abs() / den).max().item()
    # float32-equivalent: no worse than the library's float32 MFMA product (plus a rounding of slack), and absolutely small
    assert e_own <= 1.25 * e_lib + 3e-8 and e_own < 6e-7, (e_own, e_lib)


TN_SHAPES = [(4096, 256, 128), (10000, 64, 192), (33333, 512, 384), (8191, 128, 96), (70000, 256, 768), (50000, 768, 512)]


@pytest.mark.parametrize("M,N,Kp", TN_SHAPES)
def test_split_tn_integer_data_is_bit_exact(M, N, Kp):
    g = torch.Generator(device=DEV).manual_seed(M + N + Kp)
    a, b = _ints((M, N), 4, g), _ints((M, Kp), 4, g)
    ref = (a.double().t() @ b.double()).float()
    assert capi.gemm_tn_f32_supported(a, b)
    out = capi.gemm_tn_f32(a, b)
    assert torch.equal(out, ref)
    for _ in range(2):
        assert torch.equal(capi.gemm_tn_f32(a, b), out)         # deterministic (slab partials added in slab order)


@pytest.mark.parametrize("M,N,Kp", TN_SHAPES[:5])
def test_split_tn_random_data_strides_and_error_vs_float64(M, N, Kp):
    g = torch.Generator(device=DEV).manual_seed(N * 3 + Kp)
    aw = torch.randn((M, N + 8), device=DEV, generator=g)
    bw = torch.randn((M, Kp + 16), device=DEV, generator=g)
    a, b = aw[:, 4:4 + N], bw[:, 12:12 + Kp]
    assert capi.gemm_tn_f32_supported(a, b)
    out = capi.gemm_tn_f32(a, b)
    r64 = a.double().t() @ b.double()
    den = a.double().abs().t() @ b.double().abs()
    e_own = ((out.double() - r64).abs() / den).max().item()
    e_lib = (((a.t() @ b).double() - r64).abs() / den).max().item()
    assert e_own <= 1.25 * e_lib + 3e-8 and e_own < 6e-7, (e_own, e_lib)


MID_SHAPES = [(50000, 32, 48), (50000, 48, 32), (1000, 16, 48), (777, 48, 48), (4099, 8, 20), (256, 4, 4), (33333, 24, 36)]


@pytest.mark.parametrize("M,N,K", MID_SHAPES)
def test_small_weight_products_on_the_vector_alus(M, N, K):
    """csrc/gemm_mid.hip behind the same entry points: weight matrices of 4 .. 48 rows and columns (the 16 -> 32 and 32 -> 16
    layers, util/networks.py:40-53) as plain float32 FMA chains.  Small integers bit for bit in all three forms (forward,
    input gradient from the [K, N] storage, weight gradient); random data against float64 no worse than the BLAS library."""
    g = torch.Generator(device=DEV).manual_seed(M + 5 * N + 11 * K)
    a, w, bias = _ints((M, K), 8, g), _ints((N, K), 8, g), _ints((N,), 8, g)
    ref = (a.double() @ w.double().t() + bias.double()).float()
    assert capi.gemm_nt_f32_supported(a, N)
    assert torch.equal(capi.gemm_nt_f32(a, w, bias), ref)
    assert torch.equal(capi.gemm_nt_f32(a, w.t().contiguous(), bias, w_is_kn=True), ref)
    wide = torch.zeros((M, N + 8), device=DEV)
    capi.gemm_nt_f32(a, w, bias, out=wide[:, 4:4 + N])
    assert torch.equal(wide[:, 4:4 + N], ref) and torch.all(wide[:, :4] == 0) and torch.all(wide[:, 4 + N:] == 0)
    b = _ints((M, N), 4, g)
    a4 = _ints((M, K), 4, g)
    assert capi.gemm_tn_f32_supported(b, a4)
    dw = capi.gemm_tn_f32(b, a4)
    assert torch.equal(dw, (b.double().t() @ a4.double()).float())
    assert torch.equal(capi.gemm_tn_f32(b, a4), dw)
    ar, wr = torch.randn((M, K), device=DEV, generator=g), torch.randn((N, K), device=DEV, generator=g) * 0.1
    r64 = ar.double() @ wr.double().t()
    den = ar.double().abs() @ wr.double().abs().t()
    e_own = ((capi.gemm_nt_f32(ar, wr).double() - r64).abs() / den).max().item()
    e_lib = (((ar @ wr.t()).double() - r64).abs() / den).max().item()
    assert e_own <= 1.25 * e_lib + 3e-8 and e_own < 6e-7, (e_own, e_lib)
    br = torch.randn((M, N), device=DEV, generator=g)
    r64 = br.double().t() @ ar.double()
    den = br.double().abs().t() @ ar.double().abs()
    e_own = ((capi.gemm_tn_f32(br, ar).double() - r64).abs() / den).max().item()
    e_lib = (((br.t() @ ar).double() - r64).abs() / den).max().item()
    assert e_own <= 1.25 * e_lib + 3e-8 and e_own < 6e-7, (e_own, e_lib)


def test_split_products_reject_what_they_cannot_take():
    a = torch.zeros((1000, 80), device=DEV)
    assert not capi.gemm_nt_f32_supported(a, 64)                     # K = 80 is no multiple of 32 (and no small weight matrix)
    assert not capi.gemm_nt_f32_supported(torch.zeros((100, 64), device=DEV), 64)      # fewer rows than a tile
    assert not capi.gemm_nt_f32_supported(torch.zeros((1000, 64), device=DEV, dtype=torch.bfloat16), 64)
    with pytest.raises(capi.SemigcnLibraryError, match="unsupported shape"):
        capi.gemm_nt_f32(a, torch.zeros((64, 80), device=DEV))
    assert not capi.gemm_tn_f32_supported(torch.zeros((1000, 64), device=DEV), torch.zeros((1000, 64), device=DEV))
    lib = capi.load()
    assert lib.sg_gemm_nt_f32(a.data_ptr(), 80, a.data_ptr(), 80, 1, None, a.data_ptr(), 80, 1000, 64, 64, None, 0, None) == -1
    assert b"workspace" in lib.sg_last_error()


@pytest.mark.parametrize("nu,nv", [(160, 128), (100, 50)])
@pytest.mark.parametrize("cin,cout", [(64, 128), (256, 128)])
def test_float32_block_runs_on_the_split_kernels_and_matches_the_blas_engine(cin, cout, nu, nv):
    """One [ChebConv -> BatchNorm -> LeakyReLU] block of float32 features, forward and backward, with the products on the
    split kernels (default) and on the BLAS library (SG_TUNE_F32_ENGINE bit 0): the launch trace names the engine of every
    product, and the two results agree to float32 rounding (both are float32-equivalent products of the same operands)."""
    from semigcn_amd import synth
    from semigcn_amd.graph import MeshGraph
    from test_gpu_blocks import _block_module, _run
    # 20 480 rows: 256-row tiles; 5 000 rows (BASELINE configs[0], the reference's own mesh size): 128-row tiles -- no product of a
    # float32 block is left with the BLAS library at either size (round 5: below 16 384 rows the forward / input-gradient ones were)
    m = synth.torus_mesh(nu, nv)
    g = MeshGraph.from_edge_index(torch.from_numpy(m.edge_index).to(DEV), m.num_vertices)
    seq = _block_module(cin, cout)
    gen = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn((m.num_vertices, cin), device=DEV, generator=gen)
    r = torch.randn((m.num_vertices, cout), device=DEV, generator=gen)

    def run(engine):
        capi.tuning_set(capi.TUNE_F32_ENGINE, engine)
        try:
            seq[1].reset_running_stats()
            with capi.LaunchTrace(256, kinds=("nt", "tn")) as tr:
                res = _run(seq, g, x, r)
                torch.cuda.synchronize()
                recs = tr.records()
            return res, recs
        finally:
            capi.tuning_set(capi.TUNE_F32_ENGINE, 0)

    res_s, rec_s = run(0)
    res_b, rec_b = run(1)
    assert len(rec_s) == len(rec_b) == 3                       # forward, input gradient, weight gradient
    # every product on the split kernels -- except where the library's own rule (sg_gemm_nt_f32_pays: fewer than 128 work items over
    # K >= 384, the A/B of profiles/r06_gemm_f32split_5k.json) hands a forward / input-gradient product of a 5 K-row block back
    want = ["split" if (t["kind"] == "tn" or capi.gemm_nt_f32_pays(t["a"], t["b"], t["c"])) else "blas" for t in rec_s]
    assert [t["engine"] for t in rec_s] == want, rec_s
    assert "split" in want[:2] or nu * nv >= 16384 or cin == 256, want
    assert all(t["engine"] == "blas" for t in rec_b), rec_b
    names = ["y", "dx"] + [n for n, _ in seq.named_parameters()] + ["running_mean", "running_var"]
    for n, a, b in zip(names, res_s[:-1], res_b[:-1]):
        if n.endswith("0.bias"):          # the conv bias in front of a BatchNorm: zero in exact arithmetic, rounding noise on both sides
            continue
        assert float((a - b).norm() / b.norm()) < 5e-6, n


@pytest.mark.parametrize("nu,nv", [(160, 128), (100, 50)])
@pytest.mark.parametrize("cin,cout", [(64, 128), (256, 128), (128, 128)])
def test_cached_split_images_give_the_bits_of_per_product_packing_and_follow_the_weights(cin, cout, nu, nv):
    """Round 6: a float32 block keeps the split-bf16 images of its weight matrix (sg_block::wsplit / wsplit_t) and rebuilds them
    only when the weights changed.  Same bits as a block that splits its weights inside every product (forward, input gradient,
    parameter gradients, running statistics) -- before AND after an in-place weight update, at both tile shapes."""
    import copy
    from semigcn_amd import synth
    from semigcn_amd.graph import MeshGraph
    from test_gpu_blocks import _block_module, _run
    m = synth.torus_mesh(nu, nv)
    g = MeshGraph.from_edge_index(torch.from_numpy(m.edge_index).to(DEV), m.num_vertices)
    seq_a = _block_module(cin, cout)
    seq_b = copy.deepcopy(seq_a)
    gen = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn((m.num_vertices, cin), device=DEV, generator=gen)
    r = torch.randn((m.num_vertices, cout), device=DEV, generator=gen)

    def run(seq, images):
        old = F_sg.USE_SPLIT_IMAGES
        F_sg.USE_SPLIT_IMAGES = images
        try:
            return _run(seq, g, x, r)
        finally:
            F_sg.USE_SPLIT_IMAGES = old

    for step in range(3):
        res_a, res_b = run(seq_a, True), run(seq_b, False)
        for a, b in zip(res_a[:-1], res_b[:-1]):
            assert torch.equal(a, b), step
        with torch.no_grad():                      # an optimiser step's worth of change: the version counters move, the images follow
            for p_a, p_b in zip(seq_a.parameters(), seq_b.parameters()):
                d = 0.01 * torch.randn(p_a.shape, device=DEV, generator=gen)
                p_a.add_(d)
                p_b.add_(d)
    # and the outputs did change with the weights (the images were not left stale)
    assert not torch.equal(run(seq_a, True)[0], res_a[0])


def test_one_module_on_meshes_of_both_size_classes_keeps_an_image_per_tile_variant():
    """The split image's layout follows the tile variant (128-row tiles below 16 K rows, 256-row tiles from there on).  ONE block
    applied to a 5 K-vertex and a 20 K-vertex mesh inside one autograd graph (shared weights): the second forward must not
    overwrite the image the first mesh's backward still reads -- an image pair per variant (functional.BlockPlan.bind).  Bits
    against the per-product packing."""
    import copy
    from semigcn_amd import synth
    from semigcn_amd.graph import MeshGraph
    from test_gpu_blocks import _block_module
    meshes = [synth.torus_mesh(100, 50), synth.torus_mesh(160, 128)]
    graphs = [MeshGraph.from_edge_index(torch.from_numpy(m.edge_index).to(DEV), m.num_vertices) for m in meshes]
    assert capi.gemm_nt_f32_variant(meshes[0].num_vertices) == 1 and capi.gemm_nt_f32_variant(meshes[1].num_vertices) == 0
    seq_a = _block_module(128, 128)
    seq_b = copy.deepcopy(seq_a)
    gen = torch.Generator(device=DEV).manual_seed(21)
    xs = [torch.randn((m.num_vertices, 128), device=DEV, generator=gen) for m in meshes]

    def run(seq, images):
        old = F_sg.USE_SPLIT_IMAGES
        F_sg.USE_SPLIT_IMAGES = images
        try:
            seq.train()
            seq.zero_grad()
            ins = [x.clone().requires_grad_(True) for x in xs]
            ys = [seq(x, g) for x, g in zip(ins, graphs)]          # small mesh first, then the large one: its refresh comes second
            (ys[0].square().mean() + ys[1].square().mean()).backward()
            return [y.detach().clone() for y in ys] + [x.grad.clone() for x in ins] + [p.grad.clone() for p in seq.parameters()]
        finally:
            F_sg.USE_SPLIT_IMAGES = old

    for step in range(2):
        for a, b in zip(run(seq_a, True), run(seq_b, False)):
            assert torch.equal(a, b), step
        with torch.no_grad():
            for p_a, p_b in zip(seq_a.parameters(), seq_b.parameters()):
                d = 0.01 * torch.randn(p_a.shape, device=DEV, generator=gen)
                p_a.add_(d)
                p_b.add_(d)
