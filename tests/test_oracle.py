"""Pins the oracle: against the golden vectors made by the reference's own module code
(oracle/make_golden.py), against the reference's own dense D^-1/2 A D^-1/2 matrices,
and -- when /root/reference is present -- against the reference classes directly."""
import numpy as np
import pytest
import torch

import golden_util as GU
from oracle import dense, models as OM, pyg_restatement as P, ref_shim
from semigcn_amd import synth


# thread-count dependent summation order inside ATen (BN statistics, GEMM) already moves a
# 13-layer fp32 forward by ~2e-6 between two runs of the same code
MODEL_TOL = 1e-5
MGCN_TOL = 5e-5  # 33 convs, 4 resolutions; relative L2
# ... and its gradients (13 train-mode BatchNorms over a 258-vertex "batch", LeakyReLU kinks,
# random weights) by ~1e-3: the reference's own fp32 noise floor on these fixtures
GRAD_TOL = 2e-3


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("name", ["sphere", "torus"])
def test_edge_layout_matches_reference_mesh(name, fixture_meshes):
    g0 = GU.load("g0_mesh_layout.npz")
    m = fixture_meshes[name]
    assert np.array_equal(g0[f"{name}/faces"], m.faces)
    # synth.edges_from_faces reproduces util/mesh.py:60-100,229-230 (face-discovery order)
    assert np.array_equal(g0[f"{name}/edge_index"], m.edge_index)


@pytest.mark.parametrize("name", ["sphere", "torus"])
def test_dense_lhat_matches_reference_matrices(name):
    g0 = GU.load("g0_mesh_layout.npz")
    ei = g0[f"{name}/edge_index"]
    L = dense.dense_lhat(ei, int(ei.max()) + 1)
    assert np.abs(L - g0[f"{name}/lhat_dense_ref"]).max() < 5e-7  # reference matrices are fp32


@pytest.mark.parametrize("name,cin,cout", [("sphere", 4, 16), ("sphere", 32, 64), ("sphere", 256, 512),
                                           ("torus", 4, 16), ("torus", 32, 64), ("torus", 3, 5)])
def test_chebconv_restatement_vs_dense_fp64_and_golden(name, cin, cout, fixture_meshes):
    g1 = GU.load("g1_chebconv.npz")
    tag = f"{name}/{cin}x{cout}"
    m = fixture_meshes[name]
    conv = P.ChebConv(cin, cout, K=3)
    GU.fill_state(conv, seed=11)
    x = torch.from_numpy(g1[tag + "/x"]).requires_grad_(True)
    y = conv(x, torch.from_numpy(m.edge_index))
    assert rel(y.detach().numpy(), g1[tag + "/out"]) < 1e-6
    y64 = dense.cheb_conv_dense(g1[tag + "/x"], m.edge_index, [l.weight.detach().numpy() for l in conv.lins],
                                conv.bias.detach().numpy())
    assert rel(y.detach().numpy(), y64) < 2e-6
    (y * torch.from_numpy(g1[tag + "/r"])).sum().backward()
    assert rel(x.grad.numpy(), g1[tag + "/dx"]) < 1e-6
    # dx in fp64: dL/dx = sum_k T_k(L)^T r W_k
    L = dense.dense_lhat(m.edge_index, m.num_vertices)
    r = g1[tag + "/r"].astype(np.float64)
    W = [l.weight.detach().numpy().astype(np.float64) for l in conv.lins]
    T1, T2 = L, 2 * L @ L - np.eye(L.shape[0])
    dx64 = r @ W[0] + T1.T @ (r @ W[1]) + T2.T @ (r @ W[2])
    assert rel(x.grad.numpy(), dx64) < 5e-6


@pytest.mark.parametrize("name", ["sphere", "torus"])
@pytest.mark.parametrize("skip", [False, True])
def test_sgcn_oracle_vs_reference_golden(name, skip, fixture_meshes):
    g2 = GU.load("g2_sgcn.npz")
    m = fixture_meshes[name]
    tag = f"{name}/skip{int(skip)}"
    net = OM.SGCNOracle(skip=skip)
    assert list(net.state_dict().keys()) == list(g2[f"{name}/state_dict_keys"])
    GU.fill_state(net, seed=314)
    z1, x_pos, ei = torch.from_numpy(m.z1), torch.from_numpy(m.x_pos), torch.from_numpy(m.edge_index)
    dm = g2[f"{name}/dm"]
    net.eval()
    with torch.no_grad():
        assert rel(net(z1, x_pos, ei, torch.from_numpy(dm)).numpy(), g2[tag + "/eval_dm_tensor"]) < MODEL_TOL
        assert rel(net(z1, x_pos, ei, dm).numpy(), g2[tag + "/eval_dm_ndarray"]) < MODEL_TOL
        assert rel(net(z1, x_pos, ei, None).numpy(), g2[tag + "/eval_dm_none"]) < MODEL_TOL
    net.train()
    z1g = z1.clone().requires_grad_(True)
    pos = net(z1g, x_pos, ei, torch.from_numpy(dm))
    assert rel(pos.detach().numpy(), g2[tag + "/train_out"]) < MODEL_TOL
    r = torch.from_numpy(GU.probe(tag + "/r", (m.num_vertices, 3)))
    (pos * r).sum().backward(retain_graph=True)
    assert GU.rel_l2(z1g.grad.numpy(), g2[tag + "/dz1"]) < GRAD_TOL
    golden = {k[len(tag + "/grad/"):]: g2[k] for k in g2.files if k.startswith(tag + "/grad/")}
    GU.check_grad_summary([(n, p.grad) for n, p in net.named_parameters() if p.grad is not None], golden, GRAD_TOL, tag)
    z1g.grad = None
    g0 = GU.load("g0_mesh_layout.npz")
    v_mask = g2[f"{name}/v_mask"]
    f_mask = v_mask[m.faces].all(1)
    lp = OM.mask_pos_rec_loss(pos, torch.from_numpy(m.vs.astype(np.float32)), v_mask)
    ln = OM.mask_norm_rec_loss(OM.compute_fn(pos, m.faces), torch.from_numpy(g0[f"{name}/fn"]), f_mask)
    loss = lp + 4.0 * ln
    assert np.allclose([lp.item(), ln.item(), loss.item()], g2[tag + "/loss"], rtol=1e-5)
    loss.backward()
    assert GU.rel_l2(z1g.grad.numpy(), g2[tag + "/loss_dz1"]) < 2e-2  # L1-on-normals kinks
    sd = net.state_dict()
    for k in g2.files:
        if k.startswith(tag + "/bn/"):
            assert rel(sd[k[len(tag + "/bn/"):]].numpy(), g2[k]) < 1e-5


def _mgcn_oracle(g3):
    eis = [torch.from_numpy(g3[f"edge_index/{l}"]) for l in range(4)]
    phs = [g3[f"pool_hash/{l}"] for l in range(3)]
    sms = [torch.from_numpy(g3[f"smposs/{l}"]) for l in range(4)]
    net = OM.MGCNOracle(eis, phs, sms, drop=(0.0, 0.0, 0.0))
    GU.fill_state(net, seed=2718)
    return net


def test_pool_unpool_vs_reference_classes():
    g3 = GU.load("g3_mgcn.npz")
    x = torch.from_numpy(g3["pool/x"])
    px = OM.pool_mean(g3["pool_hash/0"], x)
    assert rel(px.numpy(), g3["pool/out"]) < 1e-6
    assert rel(OM.unpool_gather(g3["pool_hash/0"], px).numpy(), g3["unpool/out"]) < 1e-6


def test_mgcn_oracle_vs_reference_golden():
    g3 = GU.load("g3_mgcn.npz")
    net = _mgcn_oracle(g3)
    ref_keys = [k for k in g3["state_dict_keys"] if not k.endswith("pool_hash")]
    assert list(net.state_dict().keys()) == ref_keys
    z1, dm = torch.from_numpy(g3["z1"]), g3["dm"]
    net.eval()
    with torch.no_grad():
        for key, d in (("eval_dm_ndarray", dm), ("eval_dm_tensor", torch.from_numpy(dm)), ("eval_dm_none", None)):
            for l, p in enumerate(net(z1, d)):
                assert GU.rel_l2(p.numpy(), g3[f"{key}/{l}"]) < MGCN_TOL, (key, l)
    # quirk App. C-1: a Tensor dm is ignored by MGCN
    for l in range(4):
        assert np.array_equal(g3[f"eval_dm_tensor/{l}"], g3[f"eval_dm_none/{l}"])
        assert not np.array_equal(g3[f"eval_dm_ndarray/{l}"], g3[f"eval_dm_none/{l}"])
    net.train()
    z1g = z1.clone().requires_grad_(True)
    poss = net(z1g, dm)
    for l, p in enumerate(poss):
        assert GU.rel_l2(p.detach().numpy(), g3[f"train_out/{l}"]) < MGCN_TOL
    w = [0.35, 0.3, 0.2, 0.15]
    loss = sum(wi * (p * torch.from_numpy(GU.probe(f"mgcn/r{l}", p.shape))).sum() for l, (wi, p) in enumerate(zip(w, poss)))
    assert abs(loss.item() - float(g3["train_loss"])) < 2e-3 * abs(float(g3["train_loss"]))  # cancelling sum
    loss.backward()
    assert GU.rel_l2(z1g.grad.numpy(), g3["dz1"]) < GRAD_TOL
    golden = {k[len("grad/"):]: g3[k] for k in g3.files if k.startswith("grad/")}
    GU.check_grad_summary([(n, p.grad) for n, p in net.named_parameters() if p.grad is not None], golden, GRAD_TOL)


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present (GPU box)")
def test_oracle_vs_live_reference_sgcn(fixture_meshes):
    ref = ref_shim.load()
    m = fixture_meshes["torus"]
    rnet = ref.networks.SingleScaleGCN("cpu", skip=True)
    GU.fill_state(rnet, seed=99)
    onet = OM.SGCNOracle(skip=True)
    onet.load_state_dict(rnet.state_dict())

    class D:
        z1, x_pos, edge_index = torch.from_numpy(m.z1), torch.from_numpy(m.x_pos), torch.from_numpy(m.edge_index)

    rnet.train(), onet.train()
    a = rnet(D, None)
    b = onet(D.z1, D.x_pos, D.edge_index, None)
    assert torch.equal(a, b)


# --------------------------------------------------------------------------------------
# mesh connectivity + dummy masks: oracle/meshprep.py vs the reference's own
# Mesh / make_dummy_mask / vmask_to_fmask outputs (tests/golden/g4_meshprep.npz)
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["sphere", "torus", "open"])
def test_meshprep_oracle_vs_reference_golden(name):
    from oracle import meshprep as MP
    g = GU.load("g4_meshprep.npz")
    faces, V = g[f"{name}/faces"], int(g[f"{name}/num_vertices"])
    edges = MP.edges_first_meeting(faces)
    assert np.array_equal(edges, g[f"{name}/edges"])
    assert np.array_equal(MP.edge_index(edges), g[f"{name}/edge_index"])
    assert np.array_equal(np.sort(MP.face_ring(faces, V), 1), np.sort(g[f"{name}/f2f"], 1))
    rs_state = np.random.get_state()
    try:
        np.random.seed(317)            # the seed make_golden gave the reference
        vm, fm = MP.make_dummy_mask(faces, edges, V, dm_size=3, kn=[1, 2, 3])
    finally:
        np.random.set_state(rs_state)
    assert np.array_equal(vm, g[f"{name}/vmask_dummy"])
    assert np.array_equal(fm, g[f"{name}/fmask_dummy"])
    assert np.array_equal(MP.vmask_to_fmask(faces, g[f"{name}/vm"]), g[f"{name}/fm"])
    # the vectorised builder the synthetic meshes use agrees with the literal scan
    from semigcn_amd import synth
    assert np.array_equal(synth.edges_from_faces(faces, V), edges)


@pytest.mark.parametrize("name", ["sphere", "torus"])
def test_refine_oracle_vs_reference_golden(name):
    """oracle/refine.py (sparse float64) against the reference's own dense float32 Mesh.mesh_merge."""
    from oracle import refine as R
    g = GU.load("g5_refine.npz")
    for tag in ("w1", "w03", "wb"):
        w, wb = g[f"{name}/{tag}/w"]
        x = R.mesh_merge(g[f"{name}/edge_index"], g[f"{name}/org_pos"], g[f"{name}/new_pos"], g[f"{name}/preserve"], w, wb)
        ref = g[f"{name}/{tag}/ref_pos"]
        assert np.abs(x - ref).max() / np.abs(ref).max() < 1e-5      # the reference's fp32 dense solve: ~1e-6..4e-6
        assert np.abs(ref - g[f"{name}/new_pos"]).max() > 1e-2      # the solve moves vertices (not a trivial fixture)


# ---- bf16-storage oracle and the prescribed-pattern activation (the checkers of tests/test_gpu_config_parity.py) ----
@pytest.mark.parametrize("skip", [False, True])
@pytest.mark.parametrize("post", [False, True])
def test_bf16_oracle_without_rounding_is_the_fp32_oracle(skip, post, fixture_meshes, monkeypatch):
    """With its rounding switched off, the bf16-storage oracle -- a hand-written backward of the ChebConv recurrence in
    either evaluation order -- must BE the fp32 oracle: same forward, same autograd gradients."""
    from oracle import bf16 as OB
    monkeypatch.setattr(OB, "rb", lambda t: t)
    m = fixture_meshes["torus"]
    a, b = OM.SGCNOracle(skip=skip), OB.SGCNOracleBf16(skip=skip, post_when_narrowing=post)
    GU.fill_state(a, seed=11)
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
    b.load_state_dict(a.state_dict())
    a.train(), b.train()
    ei, xp = torch.from_numpy(m.edge_index), torch.from_numpy(m.x_pos)
    za, zb = (torch.from_numpy(m.z1).requires_grad_(True) for _ in range(2))
    r = torch.from_numpy(GU.probe("bf16-oracle", (m.num_vertices, 3)))
    ra, rb_ = GU.ActivationMasks(a), GU.ActivationMasks(b)
    pa, pb = a(za, xp, ei, None), b(zb, xp, ei, None)
    ra.close(), rb_.close()
    flips = ra.flips_against(rb_)
    (pa * r).sum().backward()
    (pb * r).sum().backward()
    assert GU.rel_l2(pb.detach(), pa.detach()) < 1e-5
    tol = GU.grad_tolerance(flips, 2e-4)
    assert GU.rel_l2(zb.grad, za.grad) < tol
    ga = dict(a.named_parameters())
    floor = 1e-2 * max(float(p.grad.abs().max()) for p in ga.values() if p.grad is not None)
    for n, p in b.named_parameters():
        if p.grad is None:
            continue
        ref = ga[n].grad
        assert float((p.grad - ref).norm()) <= tol * max(float(ref.norm()), floor * np.sqrt(ref.numel())), n


def test_bf16_oracle_stores_bf16_values(fixture_meshes):
    from oracle import bf16 as OB
    m = fixture_meshes["sphere"]
    net = OB.SGCNOracleBf16().train()
    GU.fill_state(net, seed=12)
    seen = []
    hooks = [blk.module_0.register_forward_hook(lambda mod, i, o: seen.append((i[0], o))) for blk in net.blocks]
    z = torch.from_numpy(m.z1).requires_grad_(True)
    out = net(z, torch.from_numpy(m.x_pos), torch.from_numpy(m.edge_index), None)
    for h in hooks:
        h.remove()
    assert len(seen) == 13
    for x_in, y in seen:            # what a conv reads and what it writes are bf16-representable
        assert torch.equal(x_in, OB.rb(x_in)) and torch.equal(y, OB.rb(y))
    assert out.dtype == torch.float32
    (out ** 2).mean().backward()
    assert bool(torch.isfinite(z.grad).all())


def test_prescribed_leaky_relu_with_its_own_pattern_is_leaky_relu(fixture_meshes):
    m = fixture_meshes["torus"]
    ref = OM.SGCNOracle()
    GU.fill_state(ref, seed=13)
    ref.train()
    rec = GU.ActivationMasks(ref)
    z1, xp, ei = torch.from_numpy(m.z1), torch.from_numpy(m.x_pos), torch.from_numpy(m.edge_index)
    p0 = ref(z1, xp, ei, None)
    rec.close()
    act = GU.PrescribedLeakyReLU(rec.masks)
    net = OM.SGCNOracle(act=act)
    GU.fill_state(net, seed=13)
    net.train()
    p1 = net(z1, xp, ei, None)
    assert torch.equal(p0, p1) and act.flips == 0 and act.calls == 13
    # an overridden element is counted (and those it moves across their kinks downstream) and changes the output
    masks = [mk.clone() for mk in rec.masks]
    masks[5][7, 3] = ~masks[5][7, 3]
    act.reset(masks)
    p2 = net(z1, xp, ei, None)
    assert act.flips >= 1 and act.max_flip_z > 0 and not torch.equal(p0, p2)
