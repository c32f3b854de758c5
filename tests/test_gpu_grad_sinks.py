"""``-m gpu``: the launch-saving plumbing of the training iteration -- parameter gradients added into their ``.grad``
accumulators by the library's own kernels (functional.sink_param_grads, sg_multi_add, sg_bn_bwd_coeffs' accumulators),
nn.BatchNorm1d.num_batches_tracked counted inside the statistics kernel -- against the plain autograd route, bit for bit
(the reference accumulates five backward passes per optimiser step: sgcn.py:123-146)."""
import copy

import pytest
import torch

from semigcn_amd import capi, functional as F_sg, nn as sgnn, synth, train
from semigcn_amd.networks import SingleScaleGCN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _Data:
    def __init__(self, m):
        self.z1 = torch.from_numpy(m.z1).to(DEV).requires_grad_(True)
        self.x_pos = torch.from_numpy(m.x_pos).to(DEV)
        self.edge_index = torch.from_numpy(m.edge_index).to(DEV)


def test_multi_add_matches_torch_on_blocks_vectors_and_chunks():
    g = torch.Generator(device=DEV).manual_seed(5)
    wide = torch.randn(48, 3 * 20, device=DEV, generator=g)
    tall = torch.randn(3 * 16, 24, device=DEV, generator=g)
    srcs = [wide[:, k * 20:(k + 1) * 20] for k in range(3)] + [tall[k * 16:(k + 1) * 16] for k in range(3)]
    srcs += [torch.randn(48, device=DEV, generator=g), torch.randn(1, 7, device=DEV, generator=g),
             torch.randn(5, 1, device=DEV, generator=g), torch.randn(0, 4, device=DEV), torch.randn(300, 257, device=DEV, generator=g)]
    dsts = [torch.randn(s.shape, device=DEV, generator=g) for s in srcs]           # 11 pairs: two launches
    want = [d + s for s, d in zip(srcs, dsts)]
    capi.multi_add(srcs, dsts)
    for w, d in zip(want, dsts):
        assert torch.equal(w, d)
    with pytest.raises(capi.SemigcnLibraryError, match="one shape"):
        capi.multi_add([srcs[0]], [torch.zeros(48, 21, device=DEV)])
    with pytest.raises(capi.SemigcnLibraryError, match="float32"):
        capi.multi_add([srcs[0].double()], [dsts[0].double()])


def test_bn_coefficient_kernel_accumulates_and_statistics_kernel_counts():
    V, C = 5000, 24
    x = torch.randn(V, C, device=DEV)
    dy = torch.randn(V, C, device=DEV)
    gam, bet = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
    count = torch.tensor(41, device=DEV)
    fin = capi.bn_stats_finalize(capi.col_moments(x), V, gam, bet, None, None, 0.1, 1e-5, count)
    assert int(count) == 42
    mom = capi.gemm_nt(x.to(torch.bfloat16), torch.eye(C, device=DEV, dtype=torch.bfloat16), moments=True)[1]
    capi.bn_stats_finalize_tiles(mom, capi.gemm_tile_rows(C), V, gam, bet, None, None, 0.1, 1e-5, count)
    assert int(count) == 43
    with pytest.raises(capi.SemigcnLibraryError, match="int64 scalar"):
        capi.bn_stats_finalize(capi.col_moments(x), V, gam, bet, None, None, 0.1, 1e-5, torch.zeros(1, device=DEV))
    part = capi.bn_act_bwd_reduce(dy, x, fin[2], fin[3], fin[0], fin[1], 0.01)
    plain = capi.bn_bwd_coeffs(part, float(V), gam, fin[1])
    aw, ab = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
    aw0, ab0 = aw.clone(), ab.clone()
    co = capi.bn_bwd_coeffs(part, float(V), gam, fin[1], aw, ab)
    assert torch.equal(co, plain)
    assert torch.equal(aw, aw0 + plain[1]) and torch.equal(ab, ab0 + plain[0])


def test_partition_statistics_row_also_where_the_block_count_is_capped():
    """bn_local_stats on col_moments' partials at row counts where sg_col_blocks caps the block count (V > 262 144: a few
    empty blocks at the end) -- the row a rank of a 2-way partition of the 1 M mesh contributes to the statistics all-gather."""
    for V in (300_000, 501_264, 5_000):
        C = 8
        x = torch.randn(V, C, device=DEV) * 2 + 1
        local = torch.empty(1, 2 * C + 1, device=DEV)
        capi.bn_local_stats(capi.col_moments(x), 0, V, local)
        var, mean = torch.var_mean(x.double(), dim=0, unbiased=False)
        assert float((local[0, :C].double() - mean).abs().max()) < 1e-5
        assert float((local[0, C:2 * C].double() / V - var).abs().max() / var.max()) < 1e-5
        assert float(local[0, 2 * C]) == float(V)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sunk_parameter_gradients_equal_autograd_accumulation(dtype):
    """Three accumulated forward/backward passes of the SGCN: gradients sunk into preallocated accumulators by the
    kernels == gradients accumulated by autograd's AccumulateGrad, for every parameter, bit for bit; the BatchNorm
    counters and running statistics agree as well."""
    m = synth.torus_mesh(24, 16)
    D = _Data(m)
    dm = torch.from_numpy(synth.make_dummy_masks(m.edge_index, m.num_vertices, dm_size=1, k=4, p=0.014, seed=3)).to(DEV)
    torch.manual_seed(3)
    net_a = SingleScaleGCN(DEV).to(DEV).train()
    if dtype != torch.float32:
        net_a.set_feature_dtype(dtype)
    net_b = copy.deepcopy(net_a)
    for p in net_b.parameters():
        p.grad = torch.zeros_like(p)
    for it in range(3):
        (net_a(D, dm) ** 2).mean().backward()
        with F_sg.sink_param_grads():
            (net_b(D, dm) ** 2).mean().backward()
    used = 0
    for (n, pa), (_, pb) in zip(net_a.named_parameters(), net_b.named_parameters()):
        if pa.grad is None:          # a parameter the forward pass does not use (the skip blocks of a skip=False net)
            assert not bool(pb.grad.any()), n
            continue
        used += 1
        assert torch.equal(pa.grad, pb.grad), n
    assert used >= 13 * 6
    for (n, ba), (_, bb) in zip(net_a.named_buffers(), net_b.named_buffers()):
        assert torch.equal(ba, bb), n
    assert all(int(b) == 3 for n, b in net_b.named_buffers() if n.endswith("num_batches_tracked"))


def test_sinking_is_off_outside_the_context_and_for_missing_accumulators():
    m = synth.torus_mesh(16, 12)
    ei = torch.from_numpy(m.edge_index).to(DEV)
    conv = sgnn.ChebConv(8, 16, K=3).to(DEV)
    x = torch.randn(m.num_vertices, 8, device=DEV)
    conv(x, ei).sum().backward()
    ref = [p.grad.clone() for p in conv.parameters()]
    conv.zero_grad(set_to_none=True)
    with F_sg.sink_param_grads():                       # no accumulators yet: gradients arrive through autograd
        conv(x, ei).sum().backward()
    assert all(torch.equal(p.grad, r) for p, r in zip(conv.parameters(), ref))
    got = torch.autograd.grad(conv(x, ei).sum(), list(conv.parameters()))          # outside the context: untouched route
    assert all(torch.equal(g, r) for g, r in zip(got, ref))
    assert all(torch.equal(p.grad, r) for p, r in zip(conv.parameters(), ref))     # .grad not written by autograd.grad


def test_trainer_gradient_buffer_views_and_optimizer_steps():
    """SGCNTrainer keeps every gradient as a view of one flat buffer; ten iterations (two optimiser steps) give the same
    parameters as the same loop with per-parameter gradients and autograd accumulation."""
    m = synth.torus_mesh(24, 16)
    import bench
    batch = bench.build_mesh_batch(m, torch.device(DEV), 5)
    torch.manual_seed(11)
    net = SingleScaleGCN(DEV).to(DEV)
    ref = copy.deepcopy(net)
    tr = train.SGCNTrainer(net, batch)
    assert tr.grads.whole() is not None and tr.grads.flat.numel() == sum(p.numel() for p in net.parameters())
    assert all(p.grad.data_ptr() >= tr.grads.flat.data_ptr() for p in net.parameters())
    opt = torch.optim.Adam(ref.parameters(), lr=0.01)
    helper = train.SGCNTrainer.__new__(train.SGCNTrainer)
    helper.mesh, helper.k1, helper.k2 = batch, 4.0, 0.0
    ref.train()
    for it in range(10):
        loss_a = tr.iteration_step()
        dm = batch.v_keep * batch.dummy_masks[:, it % 5:it % 5 + 1]
        loss_b = train.SGCNTrainer.loss(helper, ref(batch.data, dm))
        loss_b.backward()
        if (it + 1) % 5 == 0:
            opt.step()
            opt.zero_grad(set_to_none=True)
        assert torch.equal(loss_a, loss_b.detach()), it
    for (n, pa), (_, pb) in zip(net.named_parameters(), ref.named_parameters()):
        assert torch.equal(pa, pb), n


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("permuted", [False, True])
def test_fused_input_step_equals_the_reference_composition(dtype, permuted):
    """sg_input_prep / sg_input_prep_bwd against the op-by-op composition of util/networks.py:67-79 (prepare_input ->
    rows into processing order -> feature dtype): forward bit for bit, dz1 (direct part + the part routed through the
    bounding box's arg-extreme vertices) to fp32 rounding."""
    from semigcn_amd.networks import _column_min_max, prepare_input
    g = torch.Generator(device=DEV).manual_seed(12)
    V = 20000
    z = (torch.randn(V, 3, device=DEV, generator=g) * torch.tensor([1.0, 2.5, 0.7], device=DEV)).contiguous()
    dm = (torch.rand(V, 1, device=DEV, generator=g) > 0.1).float()
    order = torch.randperm(V, device=DEV, generator=g) if permuted else None
    rank = torch.empty_like(order) if permuted else None
    if permuted:
        rank[order] = torch.arange(V, device=DEV)
    w = torch.randn(V, 4, device=DEV, generator=g)

    za = z.clone().requires_grad_(True)
    xa = prepare_input(za, dm)
    if permuted:
        xa = xa.index_select(0, order)
    xa = xa.to(dtype)
    (xa.float() * w).sum().backward()

    zb = z.clone().requires_grad_(True)
    lo, hi = _column_min_max(zb)
    xb = F_sg.input_prep(zb, lo, hi, dm, order, rank, dtype)
    assert xb.dtype == dtype and torch.equal(xa.detach(), xb.detach())
    (xb.float() * w).sum().backward()
    scale = float(za.grad.abs().max())
    assert float((za.grad - zb.grad).abs().max()) <= 2e-6 * scale, float((za.grad - zb.grad).abs().max()) / scale
    # the bounding-box gradient is really there: with detached bounds only the (at most six) arg-extreme rows differ
    zd = z.clone().requires_grad_(True)
    lo_d, hi_d = _column_min_max(zd.detach())
    (F_sg.input_prep(zd, lo_d, hi_d, dm, order, rank, dtype).float() * w).sum().backward()
    moved = int(((zd.grad - zb.grad).abs().sum(1) > 0).sum())
    assert 1 <= moved <= 6, moved

    # no mask, no order: dm = None means all ones
    zc = z.clone().requires_grad_(True)
    xc = F_sg.input_prep(zc, *_column_min_max(zc), None, None, None, torch.float32)
    assert torch.equal(xc.detach(), prepare_input(z, torch.ones(V, 1, device=DEV)))


def test_output_step_gathers_both_ways():
    g = torch.Generator(device=DEV).manual_seed(2)
    V = 5000
    x = torch.randn(V, 3, device=DEV, generator=g, requires_grad=True)
    base = torch.randn(V, 3, device=DEV, generator=g)
    order = torch.randperm(V, device=DEV, generator=g)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(V, device=DEV)
    w = torch.randn(V, 3, device=DEV, generator=g)
    y = F_sg.output_in_caller_order(x, base, rank, order)
    assert torch.equal(y.detach(), base + x.detach()[rank])
    (y * w).sum().backward()
    ref = torch.zeros_like(w)
    ref[rank] = w                                      # adjoint of the gather by rank
    assert torch.equal(x.grad, ref)
