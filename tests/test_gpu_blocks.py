"""sg_block_forward / sg_block_backward (include/semigcn.h, csrc/block.hip): one [ChebConv -> pool? -> BatchNorm1d ->
LeakyReLU] block per foreign call -- the unit of util/networks.py:40-46,83-101 and util/meshnet.py:39-62,105-128,223-245.

  * the C entry points against the per-operator entry points they orchestrate (bit-exact: same kernels, same arguments),
    every variant: evaluation order 0 / 1, MeshPool / MeshUnpool between conv and BatchNorm, eval mode, bf16 and fp32;
  * the networks on the block path against the same networks on the per-module path (A/B in one process);
  * golden g2 (the reference's own SingleScaleGCN) driven through the block calls, with the call count checked.
"""
import numpy as np
import pytest
import torch

import golden_util as GU
from semigcn_amd import capi, functional as F_sg, nn as sgnn, synth
from semigcn_amd.graph import MeshGraph
from semigcn_amd.networks import SingleScaleGCN
from test_gpu_config_parity import _batch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _per_module(fn):
    """Run ``fn()`` with nn.Sequential executing every module on its own (the path before the block calls)."""
    old = F_sg.USE_BLOCK_CALLS
    F_sg.USE_BLOCK_CALLS = False
    try:
        return fn()
    finally:
        F_sg.USE_BLOCK_CALLS = old


def _block_module(cin, cout, pool=None, slope=0.01, seed=0, K=3):
    torch.manual_seed(seed)
    layers = [(sgnn.ChebConv(cin, cout, K=K), "x, edge_index -> x")]
    if pool is not None:
        layers.append((pool, "x -> x"))
    layers += [(torch.nn.BatchNorm1d(cout), "x -> x"), (torch.nn.LeakyReLU(slope) if slope else torch.nn.ReLU(), "x -> x")]
    seq = sgnn.Sequential("x, edge_index", layers).to(DEV)
    with torch.no_grad():
        bn = seq[len(layers) - 2]
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
        seq[0].bias.uniform_(-0.2, 0.2)
    return seq


def _run(seq, g, x, r, train=True):
    seq.train(train)
    seq.zero_grad()
    x = x.clone().requires_grad_(True)
    y = seq(x, g)
    (y.float() * r).sum().backward()
    bn = [m for m in seq.modules() if isinstance(m, torch.nn.BatchNorm1d)][0]
    return [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in seq.parameters()] + \
        [bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("cin,cout", [(4, 16), (16, 32), (64, 128), (128, 256), (256, 128), (32, 16), (4, 32)])
@pytest.mark.parametrize("train", [True, False])
def test_block_call_equals_the_per_module_path(dtype, cin, cout, train, fixture_meshes):
    """One block through sg_block_* against ChebConv, BatchNorm + activation run as modules of their own: the same kernels
    with the same arguments.  bf16 features: bit-identical wherever both paths use the library's own kernels; fp32 (and the
    bf16 4 -> 32 layer, K = 12 columns): the products go to the BLAS library by two routes (torch's call / the library's
    own hipBLASLt call), equal to rounding."""
    m = synth.torus_mesh(40, 30)
    g = MeshGraph.from_edge_index(torch.from_numpy(m.edge_index).to(DEV), m.num_vertices)
    seq = _block_module(cin, cout)
    gen = torch.Generator().manual_seed(cin * 1000 + cout)
    x = torch.randn(m.num_vertices, cin, generator=gen).to(DEV).to(dtype)
    r = torch.randn(m.num_vertices, cout, generator=gen).to(DEV)
    state = {k: v.clone() for k, v in seq.state_dict().items()}
    before = list(F_sg.block_calls)
    got = _run(seq, g, x, r, train)
    assert F_sg.block_calls == [before[0] + 1, before[1] + 1], "the block path did not serve this block"
    seq.load_state_dict(state)
    want = _per_module(lambda: _run(seq, g, x, r, train))
    assert F_sg.block_calls == [before[0] + 1, before[1] + 1]
    # (eval mode: the per-module path takes scale / shift and the two BatchNorm gradient sums with ATen ops)
    own_kernels = dtype == torch.bfloat16 and (cin, cout) != (4, 32) and train
    _compare(got, want, own_kernels, 2e-6 if dtype == torch.float32 else 2e-2, train)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("K,cin,cout", [(1, 32, 64), (2, 16, 32), (2, 64, 32), (1, 64, 16)])
def test_block_call_with_fewer_chebyshev_terms(dtype, K, cin, cout):
    """K = 1 (no aggregation at all: a per-vertex Linear + BatchNorm + activation) and K = 2 (one hop; two planes for a
    narrow layer) through the same entry points -- the reference only instantiates K = 3 (util/networks.py:11), [3P]
    ChebConv takes any K >= 1 -- against the per-module path."""
    m = synth.torus_mesh(40, 30)
    g = MeshGraph.from_edge_index(torch.from_numpy(m.edge_index).to(DEV), m.num_vertices)
    seq = _block_module(cin, cout, K=K, seed=K)
    gen = torch.Generator().manual_seed(cin * 100 + cout + K)
    x = torch.randn(m.num_vertices, cin, generator=gen).to(DEV).to(dtype)
    r = torch.randn(m.num_vertices, cout, generator=gen).to(DEV)
    state = {k: v.clone() for k, v in seq.state_dict().items()}
    before = list(F_sg.block_calls)
    got = _run(seq, g, x, r)
    assert F_sg.block_calls == [before[0] + 1, before[1] + 1], "the block path did not serve this block"
    seq.load_state_dict(state)
    want = _per_module(lambda: _run(seq, g, x, r))
    scale = max(float(w.abs().max()) for w in want[3:3 + K])
    for i, (a, b) in enumerate(zip(got, want)):
        assert a.shape == b.shape and a.dtype == b.dtype
        if i == 2:            # the conv bias gradient: zero in exact arithmetic behind a BatchNorm
            assert float(a.abs().max()) <= 4e-3 * scale and float(b.abs().max()) <= 4e-3 * scale
        else:
            assert GU.rel_l2(a.float().cpu(), b.float().cpu()) < (2e-6 if dtype == torch.float32 else 2e-2), (i, K)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("cin,cout", [(16, 32), (64, 16), (128, 256)])
def test_block_call_on_an_asymmetric_graph_with_isolated_vertices(dtype, cin, cout):
    """An edge_index that is NOT symmetric (a third of the reverse edges dropped: the backward pass aggregates over the
    transposed CSR), with repeated edges, self-loops (dropped by [3P] ChebConv.__norm__) and a few vertices without any edge
    (deg^-1/2 = inf -> 0): the block entry points against the per-module path, both evaluation orders, planes included."""
    m = synth.torus_mesh(40, 30)
    V = m.num_vertices
    ei = torch.from_numpy(m.edge_index)
    gen = torch.Generator().manual_seed(cin + cout)
    keep = torch.rand(ei.shape[1], generator=gen) > 0.33
    keep |= ei[0] < ei[1]                                           # every undirected edge keeps at least one direction
    ei = ei[:, keep]
    lone = torch.randperm(V, generator=gen)[:7]                     # cut seven vertices out completely
    ei = ei[:, ~(torch.isin(ei[0], lone) | torch.isin(ei[1], lone))]
    ei = torch.cat([ei, ei[:, :50], torch.arange(20).repeat(2, 1)], dim=1)       # repeated edges and self-loops
    g = MeshGraph.from_edge_index(ei.to(DEV), V)
    assert not g.symmetric
    seq = _block_module(cin, cout, seed=3)
    x = torch.randn(V, cin, generator=gen).to(DEV).to(dtype)
    r = torch.randn(V, cout, generator=gen).to(DEV)
    state = {k: v.clone() for k, v in seq.state_dict().items()}
    before = list(F_sg.block_calls)
    got = _run(seq, g, x, r)
    assert F_sg.block_calls == [before[0] + 1, before[1] + 1]
    seq.load_state_dict(state)
    want = _per_module(lambda: _run(seq, g, x, r))
    assert all(bool(torch.isfinite(t.float()).all()) for t in got)
    _compare(got, want, dtype == torch.bfloat16, 2e-6, True)


def _compare(got, want, exact: bool, tol: float, train: bool = True):
    """Entries: y, dx, d conv-bias, dW_0..2, d gamma, d beta, running_mean, running_var, num_batches_tracked.  In training
    mode the conv bias gradient is zero in exact arithmetic (BatchNorm removes the column mean): both paths return
    rounding noise there, compared on the scale of the weight gradients."""
    scale = max(float(w.abs().max()) for w in want[3:6])
    for i, (a, b) in enumerate(zip(got, want)):
        assert a.shape == b.shape and a.dtype == b.dtype
        if exact:
            assert torch.equal(a, b), i
        elif i == 2 and train:
            assert float(a.abs().max()) <= 4e-3 * scale and float(b.abs().max()) <= 4e-3 * scale, i
        else:
            assert GU.rel_l2(a.float().cpu(), b.float().cpu()) < tol, i


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("mode", ["pool", "unpool"])
@pytest.mark.parametrize("cin,cout", [(32, 32), (256, 128)])
def test_block_call_with_a_pool_between_conv_and_batchnorm(dtype, mode, cin, cout):
    """DownConv.model1's conv -> MeshPool -> BatchNorm -> act (util/meshnet.py:44-47) and UpConv.model1's conv -> MeshUnpool
    -> BatchNorm -> act (:106-109) as one call each.  Against the per-module path; the conv's bias gradient is taken as the
    column sums of the pooled gradient there and of the BatchNorm-input gradient here (equal in exact arithmetic)."""
    from semigcn_amd.meshnet import MeshPool, MeshUnpool, pool_hash_to_mask, unpool_hash_to_mask
    m = synth.torus_mesh(40, 30)
    ph, ei_c, Vc = synth.greedy_pool_hierarchy(m.edge_index, m.num_vertices, seed=7)
    if mode == "pool":
        op, ei, V = MeshPool(pool_hash_to_mask(ph)), m.edge_index, m.num_vertices
    else:
        op, ei, V = MeshUnpool(unpool_hash_to_mask(ph)), ei_c, Vc
    g = MeshGraph.from_edge_index(torch.from_numpy(ei).to(DEV), V)
    seq = _block_module(cin, cout, pool=op)
    gen = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(V, cin, generator=gen).to(DEV).to(dtype)
    V_out = Vc if mode == "pool" else m.num_vertices
    r = torch.randn(V_out, cout, generator=gen).to(DEV)
    state = {k: v.clone() for k, v in seq.state_dict().items()}
    before = list(F_sg.block_calls)
    got = _run(seq, g, x, r)
    assert F_sg.block_calls == [before[0] + 1, before[1] + 1]
    seq.load_state_dict(state)
    want = _per_module(lambda: _run(seq, g, x, r))
    _compare(got, want, False, 2e-6 if dtype == torch.float32 else 1.5e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sgcn_iteration_on_block_calls_equals_the_per_module_path(dtype):
    """A whole SGCN training iteration (13 blocks forward and backward, gradient sinks on) by block calls and by modules:
    two foreign calls for the 13 blocks, and the same losses / gradients / BatchNorm buffers."""
    from semigcn_amd import train

    def run(blocks: bool):
        old = F_sg.USE_BLOCK_CALLS
        F_sg.USE_BLOCK_CALLS = blocks
        try:
            m = synth.torus_mesh(48, 32)
            torch.manual_seed(11)
            net = SingleScaleGCN(DEV).to(DEV)
            if dtype != torch.float32:
                net.set_feature_dtype(dtype)
            tr = train.SGCNTrainer(net, _batch(m, 5))
            losses = [float(tr.iteration_step()) for _ in range(3)]
            return losses, [p.grad.clone() for p in net.parameters()], [b.clone() for b in net.buffers()]
        finally:
            F_sg.USE_BLOCK_CALLS = old
    before, chains = list(F_sg.block_calls), list(F_sg.chain_calls)
    la, ga, ba = run(True)
    assert F_sg.block_calls == [before[0] + 39, before[1] + 39]
    assert F_sg.chain_calls == [chains[0] + 3, chains[1] + 3], "13 blocks = ONE foreign call per direction and iteration"
    lb, gb, bb = run(False)
    exact = dtype == torch.bfloat16
    for a, b in zip(la, lb):
        assert (a == b) if exact else abs(a - b) <= 1e-6 * abs(b)
    for a, b in zip(ga + ba, gb + bb):
        if exact:
            assert torch.equal(a, b)
        else:
            assert GU.rel_l2(a.float().cpu(), b.float().cpu()) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mgcn_iteration_on_block_chains_equals_the_per_module_path(dtype):
    """MGCN (33 blocks: 6 stages of five, MeshPool / MeshUnpool inside the stages' first Sequential, 3 heads;
    util/meshnet.py:295-312) on chains of blocks against the per-module path, in eval mode (no dropout draws): four
    outputs, input gradient and every parameter gradient; 9 foreign calls per direction for the 33 blocks."""
    from test_host_logic import _mgcn_from_golden
    g3 = GU.load("g3_mgcn.npz")
    net = _mgcn_from_golden(DEV, g3)
    net.to(DEV).eval()
    if dtype != torch.float32:
        net.set_feature_dtype(dtype)

    def run():
        net.zero_grad()

        class D:
            z1 = torch.from_numpy(g3["z1"]).to(DEV).requires_grad_(True)
            x_pos = None
        outs = net(D, None)
        sum(((o - s) ** 2).mean() for o, s in zip(outs, net.smposs_list)).backward()
        return [o.detach().clone() for o in outs] + [D.z1.grad.clone()] + [p.grad.clone() for p in net.parameters() if p.grad is not None]
    before, chains = list(F_sg.block_calls), list(F_sg.chain_calls)
    got = run()
    assert [F_sg.block_calls[0] - before[0], F_sg.block_calls[1] - before[1]] == [33, 33]
    assert [F_sg.chain_calls[0] - chains[0], F_sg.chain_calls[1] - chains[1]] == [9, 9]
    want = _per_module(run)
    assert len(got) == len(want)
    tol = 2e-5 if dtype == torch.float32 else 6e-2
    for i, (a, b) in enumerate(zip(got, want)):
        assert GU.rel_l2(a.float().cpu(), b.float().cpu()) < tol or float((a - b).abs().max()) < 1e-6, i


def test_chains_and_single_block_calls_are_the_same_arithmetic():
    """SEMIGCN_NO_BLOCK_CHAINS (every block a call and an autograd node of its own, what runs above CHAIN_MAX_ROWS
    vertices) against the chained run: bit-identical."""
    from semigcn_amd import train

    def run(chains: bool):
        old = F_sg.USE_BLOCK_CHAINS
        F_sg.USE_BLOCK_CHAINS = chains
        try:
            m = synth.torus_mesh(48, 32)
            torch.manual_seed(11)
            net = SingleScaleGCN(DEV).to(DEV)
            net.set_feature_dtype(torch.bfloat16)
            tr = train.SGCNTrainer(net, _batch(m, 5))
            losses = [float(tr.iteration_step()) for _ in range(2)]
            return losses, [p.grad.clone() for p in net.parameters()]
        finally:
            F_sg.USE_BLOCK_CHAINS = old
    c0 = list(F_sg.chain_calls)
    la, ga = run(True)
    c1 = list(F_sg.chain_calls)
    lb, gb = run(False)
    c2 = list(F_sg.chain_calls)
    assert [c1[0] - c0[0], c2[0] - c1[0]] == [2, 26]
    assert la == lb and all(torch.equal(a, b) for a, b in zip(ga, gb))


@pytest.mark.parametrize("chains", [True, False])
def test_narrow_layers_as_planes_are_the_same_arithmetic(chains):
    """sg_block_planar: the narrow bf16 layers (16 / 32 / 64 channels) keep [Tx0 | Tx1 | Tx2] (and the library its dT / Z / G
    scratch) as K dense [V, C] planes instead of column blocks of a [V, K*C] buffer -- addresses only: two SGCN training
    iterations with and without the planes (SG_TUNE_BLOCK_PLANES / functional.USE_PLANES) are bit-identical, in one chain
    (the planes sit in the chain's arena) and block by block (the producer hands plane 0 of a registered [K, V, C] buffer to
    the block that adopts it)."""
    from semigcn_amd import train

    def run(planes: bool):
        old, oldc = F_sg.USE_PLANES, F_sg.USE_BLOCK_CHAINS
        F_sg.USE_PLANES, F_sg.USE_BLOCK_CHAINS = planes, chains
        capi.tuning_set(capi.TUNE_BLOCK_PLANES, 1 if planes else 0)
        try:
            m = synth.torus_mesh(48, 32)
            torch.manual_seed(11)
            net = SingleScaleGCN(DEV).to(DEV)
            net.set_feature_dtype(torch.bfloat16)
            tr = train.SGCNTrainer(net, _batch(m, 5))
            losses = [float(tr.iteration_step()) for _ in range(2)]
            used = [bool(pl) for ch in F_sg._chains.values() for lay in ch._layouts.values() for pl in lay.planar
                    if any(p.conv is net.blocks[i].module_0 for p in ch.plans for i in range(13))]
            return losses, [p.grad.clone() for p in net.parameters()], used
        finally:
            F_sg.USE_PLANES, F_sg.USE_BLOCK_CHAINS = old, oldc
            capi.tuning_set(capi.TUNE_BLOCK_PLANES, 1)
    la, ga, ua = run(True)
    lb, gb, ub = run(False)
    # 16 -> 32, 32 -> 64 and 64 -> 128 aggregate first on 16 / 32 / 64 channels: their T is planes (the decoder's narrow layers
    # aggregate after the product: planes inside the library's scratch)
    assert sum(ua) == 3 and not any(ub), (ua, ub)
    assert la == lb and all(torch.equal(a, b) for a, b in zip(ga, gb))


def test_planes_of_a_single_narrow_block_against_column_blocks():
    """One order-0 block (16 -> 32) and one order-1 block (64 -> 16) at an odd row count, planes against column blocks: outputs,
    input gradient, every parameter gradient and the BatchNorm buffers bit-identical; a shape whose products leave the
    128-row kernels (16 -> 1024) answers 0 to sg_block_planar and runs on column blocks."""
    m = synth.torus_mesh(37, 29)
    g = MeshGraph.from_edge_index(torch.from_numpy(m.edge_index).to(DEV), m.num_vertices)
    for cin, cout in ((16, 32), (64, 16), (8, 64)):
        seq = _block_module(cin, cout, seed=cin)
        gen = torch.Generator().manual_seed(cin + cout)
        x = torch.randn(m.num_vertices, cin, generator=gen).to(DEV).bfloat16()
        r = torch.randn(m.num_vertices, cout, generator=gen).to(DEV)
        state = {k: v.clone() for k, v in seq.state_dict().items()}
        got = _run(seq, g, x, r)
        capi.tuning_set(capi.TUNE_BLOCK_PLANES, 0)
        F_sg.USE_PLANES = False
        try:
            seq.load_state_dict(state)
            want = _run(seq, g, x, r)
        finally:
            capi.tuning_set(capi.TUNE_BLOCK_PLANES, 1)
            F_sg.USE_PLANES = True
        for i, (a, b) in enumerate(zip(got, want)):
            assert torch.equal(a, b), (cin, cout, i)
    blk = capi.sg_block()
    blk.graph, blk.dtype, blk.K, blk.V, blk.V_out = g.handle._h, capi._DTYPES[torch.bfloat16], 3, m.num_vertices, m.num_vertices
    for cin, cout, want in ((16, 32, True), (64, 128, True), (16, 1024, True), (32, 2048, False), (128, 256, False), (4, 16, False)):
        blk.Cin, blk.Cout, blk.order = cin, cout, 0
        assert capi.block_planar(blk) == want, (cin, cout)
    blk.dtype = capi._DTYPES[torch.float32]
    blk.Cin, blk.Cout = 16, 32
    assert capi.block_planar(blk) is False


def test_golden_sgcn_through_the_block_calls(fixture_meshes):
    """Golden g2 (outputs, BatchNorm statistics and parameter gradients of the reference's own SingleScaleGCN, frozen by
    oracle/make_golden.py) reproduced with every block served by sg_block_forward / sg_block_backward: the golden test of
    test_gpu_parity.py run here with the calls counted -- three eval forwards and one training forward + backward."""
    import test_gpu_parity as TP
    for skip in (False, True):
        before = list(F_sg.block_calls)
        TP.test_sgcn_vs_reference_golden("sphere", skip, fixture_meshes)
        assert [F_sg.block_calls[0] - before[0], F_sg.block_calls[1] - before[1]] == [4 * 13, 13]


# --------------------------------------------------------------------------------------
# the C entry points themselves, against the per-operator entry points
# --------------------------------------------------------------------------------------
def test_block_descriptor_validation():
    lib = capi.load()
    blk = capi.sg_block()
    assert lib.sg_block_forward(None, None) == -1 and b"null block" in lib.sg_last_error()
    assert lib.sg_block_forward(blk, None) == -1 and b"null graph" in lib.sg_last_error()
    g = MeshGraph.from_edge_index(torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]], device=DEV), 3)
    blk.graph, blk.dtype, blk.K, blk.V, blk.V_out, blk.Cin, blk.Cout = g.handle._h, 0, 4, 3, 3, 8, 8
    assert lib.sg_block_workspace(blk, 0) == -1 and b"K = 4" in lib.sg_last_error()
    blk.K, blk.Cout = 3, 6
    assert lib.sg_block_workspace(blk, 0) == -3 and b"Cout = 6" in lib.sg_last_error()
    blk.Cout = 8
    n = lib.sg_block_workspace(blk, 0)
    assert n > 0 and lib.sg_block_workspace(blk, 1) > 0
    assert lib.sg_block_forward(blk, None) == -1 and b"null pointer" in lib.sg_last_error()


def test_launch_trace_sees_the_kernels_of_a_block():
    """sg_trace_*: event pairs around the aggregations and products a block call launches (what bench.py's roofline
    figures are taken from now that no Python-side timer can bracket them)."""
    m = synth.torus_mesh(40, 30)
    g = MeshGraph.from_edge_index(torch.from_numpy(m.edge_index).to(DEV), m.num_vertices)
    seq = _block_module(64, 128)
    x = torch.randn(m.num_vertices, 64, device=DEV).bfloat16()
    r = torch.randn(m.num_vertices, 128, device=DEV)
    _run(seq, g, x, r)            # warm
    with capi.LaunchTrace(256) as tr:
        _run(seq, g, x, r)
        torch.cuda.synchronize()
        recs = tr.records()
    kinds = [(t["kind"], t["engine"]) for t in recs]
    assert kinds.count(("agg", "agg")) == 4 and kinds.count(("nt", "mfma")) == 2 and kinds.count(("tn", "mfma")) == 1
    assert all(t["ms"] > 0 for t in recs)
    fwd = [t for t in recs if t["kind"] == "nt"][0]
    assert (fwd["a"], fwd["b"], fwd["c"]) == (m.num_vertices, 128, 192)


def test_block_plan_follows_replaced_buffers_and_changed_hyperparameters_and_the_model_pickles():
    """the descriptors of a block are bound once and re-bound when the plan's fingerprint moves.  The fingerprint
    covers every tensor the descriptors point at and the BatchNorm's momentum / eps: replacing ``bn.running_mean`` by a new
    tensor, or changing ``bn.momentum``, after the first call is seen by the next one (the per-module path re-reads the module
    on every call; the two must agree).  And a model that has run (plans cached on its modules, weak references, packed
    weights on the device) can be pickled and deep-copied: the copy rebuilds its caches and computes the same."""
    import copy
    import pickle
    m = synth.torus_mesh(40, 30)
    g = MeshGraph.from_edge_index(torch.from_numpy(m.edge_index).to(DEV), m.num_vertices)
    seq = _block_module(16, 32)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(m.num_vertices, 16, generator=gen).to(DEV)
    r = torch.randn(m.num_vertices, 32, generator=gen).to(DEV)
    _run(seq, g, x, r)                                       # plans and descriptors exist now
    bn = seq[1]
    old_mean = bn.running_mean
    bn.running_mean = torch.zeros_like(old_mean)             # a NEW tensor object
    bn.momentum = 0.5
    kept = old_mean.clone()
    var_before = bn.running_var.clone()
    _run(seq, g, x, r)
    assert torch.equal(old_mean, kept)                       # the replaced tensor is no longer written ...
    # ... the new one is, with the new momentum -- what the per-module path computes from the same state
    ref_seq = _block_module(16, 32)
    ref_seq.load_state_dict(seq.state_dict())
    ref_seq[1].running_mean.zero_()
    ref_seq[1].running_var.copy_(var_before)
    ref_seq[1].momentum = 0.5
    old = F_sg.USE_BLOCK_CALLS
    F_sg.USE_BLOCK_CALLS = False
    try:
        _run(ref_seq, g, x, r)
    finally:
        F_sg.USE_BLOCK_CALLS = old
    assert float(bn.running_mean.abs().max()) > 0
    assert float((bn.running_mean - ref_seq[1].running_mean).abs().max()) < 1e-6
    assert float((bn.running_var - ref_seq[1].running_var).abs().max()) < 1e-6
    # pickling and deep copies after a forward
    blob = pickle.dumps(seq)
    for clone in (pickle.loads(blob), copy.deepcopy(seq)):
        assert "_block_plans" not in clone.__dict__ and "_block_plans_of" not in clone[0].__dict__
        clone.load_state_dict(seq.state_dict())
        a, b = _run(clone, g, x, r), _run(seq, g, x, r)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs a second device")
def test_block_calls_on_a_device_that_is_not_the_current_one():
    """a model whose tensors live on cuda:1 while cuda:0 is the process's current device -- the block
    calls take the device of their tensors (capi._on_device around sg_block_chain_* / sg_block_run), as every per-operator
    wrapper does; the results equal those of the same block on cuda:0."""
    m = synth.torus_mesh(40, 30)
    ei = torch.from_numpy(m.edge_index)
    outs = []
    for dev in ("cuda:0", "cuda:1"):
        torch.cuda.set_device(0)
        g = MeshGraph.from_edge_index(ei.to(dev), m.num_vertices)
        seq = _block_module(16, 32).to(dev)
        gen = torch.Generator().manual_seed(5)
        x = torch.randn(m.num_vertices, 16, generator=gen).to(dev)
        r = torch.randn(m.num_vertices, 32, generator=gen).to(dev)
        outs.append([t.cpu() for t in _run(seq, g, x, r)])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
