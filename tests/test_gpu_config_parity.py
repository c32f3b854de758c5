"""Model-level parity AT CONFIGURATION SIZE and for the headline storage type.

The fixtures of test_gpu_parity.py pin the path on 240-960 vertices.  Here the reference's forward
(util/networks.py:63-103: batch-statistics BatchNorm over ALL V vertices), its loss (sgcn.py:129-138) and the whole
backward are compared with the oracle where BASELINE.json's configurations live:

  c2  SGCN, 250 x 200 torus (V = 50 K), fp32, train mode             test_c2_*
  c4  SGCN, 1000 x 1000 (V = 1 M): one full-size oracle iteration     test_c4_*   (~3 min of host time)
  c4's storage type: bf16 features against the bf16-STORAGE oracle    test_bf16_*  (block by block, and end to end
                                                                      relative to what bf16 storage itself costs)
  the training loop of sgcn.py:118-147 as a 10-iteration trajectory   test_training_trajectory_*

Gradient bounds are FLAT: the oracle runs with the LeakyReLU sign pattern the HIP forward produced
(golden_util.PrescribedLeakyReLU), so both sides differentiate the same piecewise-linear branch; the test asserts that
the pattern was overridden only at rounding-level kink crossings (a handful of elements with |z| < 1e-4 rms).
"""
import os

import numpy as np
import pytest
import torch

import golden_util as GU
from oracle import bf16 as OB, models as OM
from semigcn_amd import functional as F_sg, synth, train
from semigcn_amd.networks import CHANNELS, SingleScaleGCN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


#: faces whose predicted normal is within this distance of the target normal in some component are left out of the loss on both
#: sides of a gradient comparison (the L1 normal term has a kink there: see _one_iteration_both_sides)
KINK_TAU = 1e-4


# --------------------------------------------------------------------------------------
def _batch(m, n_masks, device=DEV):
    V = m.num_vertices
    faces = torch.from_numpy(m.faces).to(device)
    target = torch.from_numpy(m.vs.astype(np.float32)).to(device)
    v_keep = torch.from_numpy(m.v_mask.astype(np.float32)).view(-1, 1).to(device)
    f_keep = v_keep[faces[:, 0]] * v_keep[faces[:, 1]] * v_keep[faces[:, 2]]
    dms = torch.from_numpy(synth.make_dummy_masks(m.edge_index, V, dm_size=n_masks, k=4, p=0.014, seed=317)).to(device)

    class Data:
        z1 = torch.from_numpy(m.z1).to(device).requires_grad_(True)
        x_pos = torch.from_numpy(m.x_pos).to(device)
        edge_index = torch.from_numpy(m.edge_index).to(device)

    return train.MeshBatch(Data, faces, target, train.face_normals(target, faces), v_keep, f_keep, dms)


class _OracleSide:
    """The oracle's inputs and loss for one mesh (sgcn.py:126-138), on the host."""

    def __init__(self, m, batch):
        self.z1 = torch.from_numpy(m.z1).requires_grad_(True)
        self.x_pos, self.ei = torch.from_numpy(m.x_pos), torch.from_numpy(m.edge_index)
        self.faces = torch.from_numpy(m.faces)
        self.tgt = torch.from_numpy(m.vs.astype(np.float32))
        self.tfn = OM.compute_fn(self.tgt, self.faces)
        self.v_mask = m.v_mask
        self.f_mask = m.v_mask[m.faces].all(1)
        self.dms = batch.dummy_masks.cpu()
        self.rm = torch.from_numpy(m.v_mask.astype(np.float32)).view(-1, 1)

    def dm(self, k):
        return self.rm * self.dms[:, k:k + 1]

    def loss(self, pos):
        return OM.mask_pos_rec_loss(pos, self.tgt, self.v_mask) + 4.0 * OM.mask_norm_rec_loss(
            OM.compute_fn(pos, self.faces), self.tfn, self.f_mask)


def _worst_param_grad_error(net, ora, floor_frac=1e-2):
    """max over parameters of |g_hip - g_ora|_2 / max(|g_ora|_2, floor sqrt(n)), floor = floor_frac x the largest
    gradient entry of the model (the ChebConv biases in front of a BatchNorm have a true gradient of exactly zero:
    what either side returns there is rounding noise)."""
    og = {n: p.grad for n, p in ora.named_parameters() if p.grad is not None}
    floor = floor_frac * max(float(g.abs().max()) for g in og.values())
    worst, where = 0.0, None
    for n, p in net.named_parameters():
        if p.grad is None or n not in og:          # unused on both sides (the skip Linears of a skip=False model)
            assert n not in og or float(og[n].abs().max()) == 0.0, n
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        a, b = p.grad.detach().cpu().double(), og[n].double()
        e = float((a - b).norm()) / max(float(b.norm()), floor * np.sqrt(b.numel()))
        if e > worst:
            worst, where = e, n
    return worst, where


def _bn_state_error(net, ora):
    so, worst = ora.state_dict(), 0.0
    for k, v in net.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            a, b = v.detach().cpu().double(), so[k].double()
            worst = max(worst, float((a - b).abs().max() / max(float(b.abs().max()), 1e-30)))
        elif k.endswith("num_batches_tracked"):
            assert int(v) == int(so[k]), k
    return worst


def _oracle_run(ora, side, m, dtype, threads):
    """forward + loss + backward of the oracle in ``dtype``; returns (positions, loss, dz1, {name: grad}, bn state)."""
    ora = ora.to(dtype).train()
    z1 = side.z1.detach().to(dtype).requires_grad_(True)
    tgt = side.tgt.to(dtype)
    old_threads = torch.get_num_threads()
    if threads:
        torch.set_num_threads(threads)
    try:
        pos = ora(z1, side.x_pos.to(dtype), side.ei, side.dm(0).to(dtype))
        loss = OM.mask_pos_rec_loss(pos, tgt, side.v_mask) + 4.0 * OM.mask_norm_rec_loss(
            OM.compute_fn(pos, side.faces), OM.compute_fn(tgt, side.faces), side.f_mask)
        loss.backward()
    finally:
        torch.set_num_threads(old_threads)
    grads = {n: p.grad.detach().double() for n, p in ora.named_parameters() if p.grad is not None}
    bn = {k: v.detach().double() for k, v in ora.state_dict().items() if "running" in k}
    return pos.detach().double(), float(loss.detach()), z1.grad.detach().double(), grads, bn


def _errors(a, b, xp):
    """a, b = (pos, loss, dz1, grads, bn) of two runs: relative errors of a against b."""
    floor = 1e-2 * max(float(g.abs().max()) for g in b[3].values())
    worst, where = 0.0, None
    for n, g in b[3].items():
        if n not in a[3]:
            continue
        e = float((a[3][n] - g).norm()) / max(float(g.norm()), floor * np.sqrt(g.numel()))
        if e > worst:
            worst, where = e, n
    bn = max(float((a[4][k] - v).abs().max() / max(float(v.abs().max()), 1e-30)) for k, v in b[4].items())
    return {"pos": GU.rel_l2(a[0], b[0]), "offset": GU.rel_l2(a[0] - xp, b[0] - xp), "loss": abs(a[1] - b[1]) / abs(b[1]),
            "dz1": GU.rel_l2(a[2], b[2]), "param_grad": worst, "param_grad_where": where, "bn": bn}


def _one_iteration_both_sides(m, *, feature_dtype, seed, skip=False, post=True, threads=None, arbiter=False, fp32_oracle=True):
    """One training iteration (forward, loss, backward) on the HIP path and on the oracle run with the HIP forward's
    LeakyReLU sign pattern.  Returns {"hip_vs_oracle": errors, ...}; with ``arbiter`` the oracle is also evaluated in
    float64 (same pattern): "hip_vs_fp64" and "oracle_vs_fp64" -- the fp32 oracle's OWN distance from exact arithmetic
    is the yardstick for everything fp32 rounding moves (the gradients of a 13-layer BatchNorm stack above all: at 5 000
    vertices the fp32 oracle's dz1 is 2e-3 from the fp64 one with identical activation patterns)."""
    batch = _batch(m, n_masks=1)
    net = SingleScaleGCN(DEV, skip=skip)
    GU.fill_state(net, seed=seed)
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    net.to(DEV).train()
    if feature_dtype == torch.bfloat16:
        net.set_feature_dtype(torch.bfloat16)
    tr = train.SGCNTrainer(net, batch)
    data = batch.data
    dm = batch.v_keep * batch.dummy_masks[:, :1]
    rank = net._layout(data)[2]
    # The normal term of the loss is an L1 norm (sgcn.py:131 -> util/loss.py, mask_norm_rec_loss): its gradient is
    # sign(fn_pred - fn_target), discontinuous where a component crosses zero -- the loss has kinks exactly like the
    # network's LeakyReLUs, and a face within rounding of one sends a gradient of either sign to its three vertices depending
    # on the last bits of the forward (measured, c2 mesh: 3 of 300 K components change sign between two fp32-equivalent dense
    # engines, and those 9 vertices then carry 8e-3 of the whole |dz1| -- the comparison would measure the dice, not the
    # arithmetic).  So both sides evaluate the loss on the faces that are NOT within KINK_TAU of such a kink in the path's own
    # forward (a preliminary pass; ~0.1 % of the faces drop out; parameters and BatchNorm buffers are restored after it).
    with torch.no_grad():
        pos0 = net(data, dm).double()
    net.load_state_dict(state0)
    fn0 = train.face_normals(pos0, batch.faces)
    near = ((fn0 - batch.target_fn.double()).abs() < KINK_TAU).any(1) & (batch.f_keep.view(-1) > 0)
    kink_faces = int(near.sum())
    assert kink_faces <= 0.02 * batch.n_f_keep, (kink_faces, batch.n_f_keep)
    batch.f_keep = batch.f_keep.clone()
    batch.f_keep[near] = 0.0
    batch.n_f_keep = int(batch.f_keep.sum().item())
    near_host = near.cpu().numpy()
    del pos0, fn0
    with GU.FusedActivationMasks(rank) as rec:
        pos = net(data, dm)
    loss = tr.loss(pos)
    loss.backward()
    torch.cuda.synchronize()
    hip = (pos.detach().cpu().double(), float(loss.detach()), data.z1.grad.detach().cpu().double(),
           {n: p.grad.detach().cpu().double() for n, p in net.named_parameters() if p.grad is not None},
           {k: v.detach().cpu().double() for k, v in net.state_dict().items() if "running" in k})
    for k, v in net.state_dict().items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(state0[k]) + 1, k

    act = GU.PrescribedLeakyReLU(rec.cpu())
    rec.masks = []

    def make(dtype):
        act.reset()
        if feature_dtype == torch.bfloat16:
            blas = _blas_layers(m.num_vertices, post)
            ora = OB.SGCNOracleBf16(skip=skip, post_when_narrowing=post, act=act, bias_bf16_layers=blas)
        else:
            ora = OM.SGCNOracle(skip=skip, act=act)
        ora.load_state_dict(state0)
        return ora
    side = _OracleSide(m, batch)
    side.f_mask = side.f_mask & ~near_host          # the same face set on the oracle's side
    assert int(side.f_mask.sum()) == batch.n_f_keep
    xp = side.x_pos.double()
    res = {"kink_faces": kink_faces}
    if fp32_oracle:
        o32 = _oracle_run(make(torch.float32), side, m, torch.float32, threads)
        res["hip_vs_oracle"] = _errors(hip, o32, xp)
        res["flips"], res["flip_frac"], res["max_flip_z"] = act.flips, act.flips / max(act.elements, 1), act.max_flip_z
    if arbiter:
        o64 = _oracle_run(make(torch.float64), side, m, torch.float64, threads)
        res["hip_vs_fp64"] = _errors(hip, o64, xp)
        if fp32_oracle:
            res["oracle_vs_fp64"] = _errors(o32, o64, xp)
        else:
            res["flips"], res["flip_frac"], res["max_flip_z"] = act.flips, act.flips / max(act.elements, 1), act.max_flip_z
    res["loss_value"] = hip[1]
    for k, v in res.items():
        print(k, {kk: (f"{vv:.3e}" if isinstance(vv, float) else vv) for kk, vv in v.items()} if isinstance(v, dict) else v)
    return res


def _assert_pattern_only_overridden_at_kinks(res, frac=2e-5, z=1e-3):
    assert res["flip_frac"] <= frac, res
    assert res["max_flip_z"] <= z, res


def _assert_fp32_parity(res, factor=2.0):
    """Forward quantities flat (north_star: 1e-5 relative fp32); everything else no further from exact arithmetic than
    ``factor`` x the fp32 oracle is."""
    h, o = res["hip_vs_fp64"], res["oracle_vs_fp64"]
    assert h["pos"] < 1e-5 and res["hip_vs_oracle"]["pos"] < 1e-5, res
    assert h["loss"] < max(1e-5, factor * o["loss"]), res
    for key, floor in (("offset", 1e-5), ("bn", 1e-5), ("dz1", 2e-4), ("param_grad", 2e-4)):
        assert h[key] < max(floor, factor * o[key]), (key, res)


# --------------------------------------------------------------------------------------
# c2: SGCN on the 50 K-vertex mesh, fp32, train mode (BASELINE configs[1])
# --------------------------------------------------------------------------------------
def test_c2_sgcn_train_iteration_vs_oracle():
    res = _one_iteration_both_sides(synth.torus_mesh(250, 200), feature_dtype=torch.float32, seed=50, arbiter=True)
    _assert_pattern_only_overridden_at_kinks(res)
    _assert_fp32_parity(res)


def test_c2_sgcn_with_skip_connections_vs_oracle():
    res = _one_iteration_both_sides(synth.torus_mesh(100, 50), feature_dtype=torch.float32, seed=51, skip=True, arbiter=True)
    _assert_pattern_only_overridden_at_kinks(res, frac=1e-4)
    _assert_fp32_parity(res)


# --------------------------------------------------------------------------------------
# c4: ONE full-size oracle iteration at V = 1 M (BASELINE.md section 3's CPU run) against the HIP fp32 iteration
# --------------------------------------------------------------------------------------
@pytest.mark.skipif(os.environ.get("SEMIGCN_SKIP_FULL_SIZE_ORACLE") == "1", reason="switched off by the environment")
def test_c4_full_size_train_iteration_vs_oracle():
    """V = 1 M.  At this size the fp32 ORACLE is the noisier side by two orders of magnitude (ATen's CPU BatchNorm and
    scatter_add sum a million rows in fp32 chunk by chunk).  Measured with both oracles, profiles/
    r03_c4_parity_fp32_fp64.json -- distance from the float64 evaluation, HIP fp32 path / fp32 oracle: network offsets
    8.1e-6 / 7.6e-4, loss 7e-9 / 1.0e-5, BatchNorm running statistics 1.6e-7 / 1.0e-4, dz1 3.5e-4 / 1.8e-2, parameter
    gradients 7.5e-4 / 2.7e-2.  So the comparison that means something is against the oracle in float64: one full-size
    float64 iteration (~4 min, ~110 GB of host memory), with flat bounds a factor 2-3 above the measured HIP figures.
    SEMIGCN_C4_BOTH_ORACLES=1 runs the fp32 oracle as well and applies the relative criterion of the c2 test."""
    import psutil
    both = os.environ.get("SEMIGCN_C4_BOTH_ORACLES") == "1"
    if psutil.virtual_memory().available < (200e9 if both else 140e9):
        pytest.skip("the full-size float64 oracle iteration needs ~110 GB of host memory")
    threads = min(16, os.cpu_count() or 1)       # ATen's index/scatter kernels stop scaling there (bench.py's probe)
    res = _one_iteration_both_sides(synth.torus_mesh(1000, 1000), feature_dtype=torch.float32, seed=52, threads=threads,
                                    arbiter=True, fp32_oracle=both)
    if os.environ.get("SEMIGCN_PARITY_JSON"):
        import json
        json.dump(res, open(os.environ["SEMIGCN_PARITY_JSON"], "w"), indent=1)
    _assert_pattern_only_overridden_at_kinks(res, frac=2e-4, z=1e-2)
    h = res["hip_vs_fp64"]
    assert h["pos"] < 1e-5 and h["loss"] < 1e-5 and h["bn"] < 1e-5, res
    if both:
        _assert_fp32_parity(res)
    else:
        assert h["offset"] < 2e-5 and h["dz1"] < 1.5e-3 and h["param_grad"] < 2.5e-3, res


# --------------------------------------------------------------------------------------
# bf16 feature storage (BASELINE configs[3]) against the oracle that rounds at the same storage points
# --------------------------------------------------------------------------------------
# Two bf16 evaluations of this network that differ ONLY in the order of their fp32 additions do not stay together end to
# end: an fp32 sum that lands on the other side of a bf16 rounding boundary moves one stored value by 2^-8, the next
# layer's products spread that over ~19 rows x Cout outputs of which one in sqrt(3 Cin) crosses a boundary of its own,
# and BatchNorm's mean carries it to every row -- after three layers every element is an independent draw of the same
# rounding noise (measured: 2 % of the elements differ after the first conv, 25 % after the second block, 79 % after the
# fifth; tools/bf16_probe.py).  So "HIP bf16 == bf16 oracle to 5e-3 end to end" is not a property ANY pair of
# implementations has.  What the storage oracle can and does pin:
#   (a) every block on its own, fed the oracle's input and the oracle's output gradient (teacher forcing): forward,
#       input gradient and parameter gradients of the HIP block against the oracle block -- same storage points or not;
#   (b) end to end, the HIP bf16 path is no further from the fp32 oracle than the bf16-storage oracle is (outputs, loss,
#       direction of the gradients): bf16 storage costs what it must and nothing more.
# 1.5 x the measured maxima (round 4, block path: 2.2e-4, 1.8e-3, 2.7e-3 -- blocks 6 / 7, the 512-channel layers).  Rounds 2-3
# measured 8.3e-3 / 1.1e-2 for dx / dw: that was torch.addmm rounding the conv bias to bf16 on the BLAS-served layers, which
# the oracle then had to be told to imitate; every engine behind sg_block_forward adds the fp32 bias.
BF16_BLOCK_TOL = {"out": 3.3e-4, "dx": 2.7e-3, "dw": 4.1e-3}


def _blas_layers(V, post):
    """Blocks whose conv bias enters the product rounded to bf16: those torch.addmm serves on the per-module path (it takes
    the bias in the operand type).  None on the block path: every engine behind sg_block_forward -- MFMA, thin, the library's
    own hipBLASLt call -- adds the fp32 parameter in its epilogue."""
    if F_sg.blocks_enabled():
        return []
    blas = []
    for i in range(13):
        cin, cout = CHANNELS[i], CHANNELS[i + 1]
        wshape = (3 * cout, cin) if (post and cout < cin) else (cout, 3 * cin)
        a = torch.empty((V, wshape[1]), dtype=torch.bfloat16, device=DEV)
        if not F_sg._mfma_ok(a, torch.empty(wshape, dtype=torch.bfloat16, device=DEV), wshape[0]) and not F_sg._thin_ok(a, wshape[0], wshape[1]):
            blas.append(i)      # (bias rounded to bf16 by addmm; the library's own kernels add the fp32 bias)
    return blas


@pytest.mark.parametrize("post", [True, False])
def test_bf16_blocks_teacher_forced_vs_bf16_storage_oracle(post, monkeypatch):
    monkeypatch.setattr(F_sg, "AGGREGATE_AFTER_GEMM_WHEN_NARROWING", post)
    m = synth.torus_mesh(100, 50)
    V = m.num_vertices
    net = SingleScaleGCN(DEV, reorder=False)
    GU.fill_state(net, seed=61)
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    net.to(DEV).train()
    net.set_feature_dtype(torch.bfloat16)
    ora = OB.SGCNOracleBf16(post_when_narrowing=post, bias_bf16_layers=_blas_layers(V, post))
    ora.load_state_dict(state0)
    ora.train()
    # the oracle end to end, every block's input / output and their gradients kept
    xs, ys, dys, dxs = [None] * 13, [None] * 13, [None] * 13, [None] * 13

    def pre(i):
        def hook(mod, args):
            x = args[0]
            if not x.requires_grad:                 # block 0's input: the rounded network input
                x = x.detach().requires_grad_(True)
            xs[i] = x
            x.register_hook(lambda g: dxs.__setitem__(i, g.detach().clone()))
            return (x,) + tuple(args[1:])
        return hook

    def post_hook(i):
        def hook(mod, args, out):
            ys[i] = out
            out.register_hook(lambda g: dys.__setitem__(i, g.detach().clone()))
        return hook
    for i, blk in enumerate(ora.blocks):
        blk.register_forward_pre_hook(pre(i))
        blk.register_forward_hook(post_hook(i))
    z1, xp, ei = torch.from_numpy(m.z1), torch.from_numpy(m.x_pos), torch.from_numpy(m.edge_index)
    r = torch.from_numpy(GU.probe("bf16-blocks", (V, 3)))
    (ora(z1, xp, ei, None) * r).sum().backward()
    og = {n: p.grad for n, p in ora.named_parameters() if p.grad is not None}

    class D:
        z1 = torch.from_numpy(m.z1).to(DEV)
        x_pos = torch.from_numpy(m.x_pos).to(DEV)
        edge_index = torch.from_numpy(m.edge_index).to(DEV)
    graph = net.graph(D)
    worst = {"out": (0.0, None), "dx": (0.0, None), "dw": (0.0, None)}
    for i, blk in enumerate(net.blocks):
        net.zero_grad(set_to_none=True)
        x = xs[i].detach().to(DEV).to(torch.bfloat16).requires_grad_(True)
        y = blk(x, graph)
        assert y.dtype == (torch.float32 if i == 12 else torch.bfloat16)
        e_out = GU.rel_l2(y.detach().float().cpu(), ys[i].detach())
        y.backward(dys[i].to(DEV).to(y.dtype))
        e_dx = GU.rel_l2(x.grad.float().cpu(), dxs[i])
        e_dw = 0.0
        for n, p in blk.named_parameters():
            ref = og[f"blocks.{i}.{n}"]
            if n == "module_0.bias":                # in front of a BatchNorm: exactly zero in exact arithmetic
                continue
            e_dw = max(e_dw, GU.rel_l2(p.grad.float().cpu(), ref))
        print(f"block {i:2d} {CHANNELS[i]:3d}->{CHANNELS[i + 1]:3d}  out {e_out:.2e}  dx {e_dx:.2e}  dw {e_dw:.2e}")
        for k, e in (("out", e_out), ("dx", e_dx), ("dw", e_dw)):
            if e > worst[k][0]:
                worst[k] = (e, i)
    print("worst", worst)
    for k, (e, i) in worst.items():
        assert e < BF16_BLOCK_TOL[k], (k, i, e)


@pytest.mark.skipif(os.environ.get("SEMIGCN_SKIP_FULL_SIZE_ORACLE") == "1", reason="switched off by the environment")
def test_c4_bf16_blocks_teacher_forced_full_size():
    """BASELINE configs[3] in ITS OWN storage type at ITS OWN size: blocks 4-7 of the SGCN (128 -> 256 -> 256 -> 512 -> 256
    channels: the launches the headline is timed on -- spmm_ring at V = 1 M, gemm_nt_256 / gemm_tn_256, BatchNorm tile
    moments merged over 7 813 tiles) one block at a time on the 1000 x 1000 mesh against the bf16-STORAGE oracle's block
    (oracle/bf16.py: ChebConv with fp32 sums and bf16 rounding at the stored rows), both fed the same stored input rows and
    the same stored output gradient.  The oracle's BatchNorm is evaluated in float64: ATen's fp32 CPU BatchNorm is the noisy
    side at a million rows (1e-4, see test_c4_full_size_train_iteration_vs_oracle).  Bounds: BF16_BLOCK_TOL, the bounds of
    the 5 K-vertex test; the running statistics against the float64 moments of the oracle's conv output."""
    import psutil
    if psutil.virtual_memory().available < 80e9:
        pytest.skip("the oracle's [E + 2V, C] message tensors need ~40 GB of host memory at V = 1 M")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    m = synth.torus_mesh(1000, 1000, edge_flips=0.15)
    V = m.num_vertices
    net = SingleScaleGCN(DEV, reorder=False)
    GU.fill_state(net, seed=63)
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    net.to(DEV).train()
    net.set_feature_dtype(torch.bfloat16)
    ora = OB.SGCNOracleBf16(post_when_narrowing=True, bias_bf16_layers=_blas_layers(V, True))
    ora.load_state_dict(state0)
    ora.train()
    ei = torch.from_numpy(m.edge_index)

    class D:
        z1 = torch.from_numpy(m.z1).to(DEV)
        x_pos = torch.from_numpy(m.x_pos).to(DEV)
        edge_index = torch.from_numpy(m.edge_index).to(DEV)
    graph = net.graph(D)
    gen = torch.Generator().manual_seed(64)
    worst = {"out": (0.0, None), "dx": (0.0, None), "dw": (0.0, None), "bn": (0.0, None)}
    for i in (4, 5, 6, 7):
        cin, cout = CHANNELS[i], CHANNELS[i + 1]
        # stored rows: what a BatchNorm + LeakyReLU in front leaves, and a gradient of the size a mean loss sends back
        x = OB.rb(torch.nn.functional.leaky_relu(torch.randn((V, cin), generator=gen)))
        dy = OB.rb(torch.randn((V, cout), generator=gen) * (1.0 / V))
        conv, bn = ora.blocks[i].module_0, ora.blocks[i].module_1
        xo = x.clone().requires_grad_(True)
        h = conv(xo, ei)                                         # fp32 sums, bf16-stored rows
        h64 = h.double()
        mean, var = h64.mean(0), h64.var(0, unbiased=False)
        y64 = (h64 - mean) * torch.rsqrt(var + bn.eps) * bn.weight.double() + bn.bias.double()
        yo = ora.act(y64.float())
        yo.backward(dy)
        og = {n: p.grad.clone() for n, p in ora.blocks[i].named_parameters() if p.grad is not None}
        ora.zero_grad(set_to_none=True)
        rm_ref = ((1 - bn.momentum) * state0[f"blocks.{i}.module_1.running_mean"].double() + bn.momentum * mean).detach()
        rv_ref = ((1 - bn.momentum) * state0[f"blocks.{i}.module_1.running_var"].double()
                  + bn.momentum * var * (V / (V - 1.0))).detach()
        del h64, y64

        net.zero_grad(set_to_none=True)
        blk = net.blocks[i]
        xh = x.to(DEV).to(torch.bfloat16).requires_grad_(True)
        yh = blk(xh, graph)
        yh.backward(dy.to(DEV).to(yh.dtype))
        e_out = GU.rel_l2(yh.detach().float().cpu(), yo.detach())
        e_dx = GU.rel_l2(xh.grad.float().cpu(), xo.grad)
        e_dw = 0.0
        for n, p in blk.named_parameters():
            if n == "module_0.bias":                             # in front of a BatchNorm: zero in exact arithmetic
                continue
            e_dw = max(e_dw, GU.rel_l2(p.grad.float().cpu(), og[n]))
        hbn = blk.module_1
        e_bn = max(GU.rel_l2(hbn.running_mean.cpu().double(), rm_ref), GU.rel_l2(hbn.running_var.cpu().double(), rv_ref))
        print(f"block {i:2d} {cin:3d}->{cout:3d} V={V}  out {e_out:.2e}  dx {e_dx:.2e}  dw {e_dw:.2e}  bn {e_bn:.2e}")
        for k, e in (("out", e_out), ("dx", e_dx), ("dw", e_dw), ("bn", e_bn)):
            if e > worst[k][0]:
                worst[k] = (e, i)
        del xo, h, yo, xh, yh
    print("worst", worst)
    for k in ("out", "dx", "dw"):
        assert worst[k][0] < BF16_BLOCK_TOL[k], (k, worst[k])
    assert worst["bn"][0] < 1e-4, worst["bn"]


def _cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a * b).sum() / (a.norm() * b.norm()).clamp_min(1e-300))


@pytest.mark.parametrize("name", ["torus", "c1", "c2"])
def test_bf16_end_to_end_no_further_from_fp32_than_the_storage_oracle(name, fixture_meshes):
    m = {"torus": lambda: fixture_meshes["torus"], "c1": lambda: synth.torus_mesh(100, 50),
         "c2": lambda: synth.torus_mesh(250, 200)}[name]()
    V = m.num_vertices
    batch = _batch(m, n_masks=1)
    net = SingleScaleGCN(DEV)
    GU.fill_state(net, seed=62)
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    net.to(DEV).train()
    net.set_feature_dtype(torch.bfloat16)
    tr = train.SGCNTrainer(net, batch)
    pos = net(batch.data, batch.v_keep * batch.dummy_masks[:, :1])
    loss = tr.loss(pos)
    loss.backward()
    side = _OracleSide(m, batch)
    xp = side.x_pos.double()
    hip = (pos.detach().cpu().double() - xp, float(loss.detach()), batch.data.z1.grad.cpu().double(),
           {n: p.grad.detach().cpu().double() for n, p in net.named_parameters() if p.grad is not None})
    runs = {}
    for key, ora in (("fp32", OM.SGCNOracle()), ("bf16", OB.SGCNOracleBf16(bias_bf16_layers=_blas_layers(V, True)))):
        ora.load_state_dict(state0)
        o = _oracle_run(ora, side, m, torch.float32, None)
        runs[key] = (o[0] - xp, o[1], o[2], o[3])
    ref = runs["fp32"]
    names = [n for n in ref[3] if not n.endswith("module_0.bias")]

    def dist(a):
        flat = lambda g: torch.cat([g[n].reshape(-1) for n in names])
        return {"offset": GU.rel_l2(a[0], ref[0]), "loss": abs(a[1] - ref[1]) / abs(ref[1]),
                "cos_dz1": _cos(a[2], ref[2]), "cos_params": _cos(flat(a[3]), flat(ref[3]))}
    d_hip, d_ora = dist(hip), dist(runs["bf16"])
    print(name, "HIP bf16 vs fp32 oracle", {k: f"{v:.3e}" for k, v in d_hip.items()})
    print(name, "bf16-storage oracle vs fp32 oracle", {k: f"{v:.3e}" for k, v in d_ora.items()})
    # (the loss bound has a floor at a tenth of the oracle's OFFSET distance: the storage oracle's own loss distance is one draw
    #  of a chaotic quantity -- the same code measured 7.0e-4 inside the whole suite and 2.9e-3 alone on the torus fixture, the
    #  host's thread partition decides its summation order -- while the HIP side read 3.65e-3 both times)
    assert d_hip["offset"] < 1.5 * d_ora["offset"] + 1e-3
    assert d_hip["loss"] < max(1.5 * d_ora["loss"] + 1e-3, 0.1 * d_ora["offset"])
    assert d_hip["cos_dz1"] > d_ora["cos_dz1"] - 0.1 and d_hip["cos_params"] > d_ora["cos_params"] - 0.05
    assert d_ora["offset"] > 5e-3          # bf16 storage moves the result by much more than fp32 rounding does


# --------------------------------------------------------------------------------------
# the training loop of sgcn.py:118-147 as a trajectory: 10 iterations, 2 Adam steps
# --------------------------------------------------------------------------------------
def _trajectory_setup(size):
    m = synth.torus_mesh(100, 50) if size == "c1" else synth.torus_mesh(250, 200)
    batch = _batch(m, n_masks=10)
    torch.manual_seed(314)                               # sgcn.py:19-25,76: the reference's seed
    net = SingleScaleGCN(DEV)
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    net.to(DEV)
    return m, batch, net, state0, _OracleSide(m, batch)


@pytest.mark.parametrize("size", ["c1", "c2"])
def test_training_trajectory_resynchronised_at_the_optimiser_steps(size):
    """Every one of the 10 losses at 1e-5, both Adam steps entry by entry (golden_util.synchronised_trajectory)."""
    m, batch, net, state0, side = _trajectory_setup(size)
    tr = train.SGCNTrainer(net, batch, lr=0.01, k1=4.0, accumulate=5)
    ora = OM.SGCNOracle()
    ora.load_state_dict(state0)

    def oracle_iteration(k):
        loss = side.loss(ora(side.z1, side.x_pos, side.ei, side.dm(k)))
        loss.backward()
        return float(loss.detach())
    n_steps = 2 if size == "c1" else 1           # (c2: five 50 K-vertex oracle iterations per step, ~15 s each)
    errs, dev = GU.synchronised_trajectory(tr, net, ora, oracle_iteration, batch.dummy_masks, n_steps=n_steps)
    print("trajectory (re-synchronised)", size, [f"{e:.1e}" for e in errs], dev)
    assert len(errs) == 5 * n_steps and max(errs) < 1e-5


def test_training_trajectory_free_running_vs_the_oracles_own_spread():
    """The same loop free-running (train.SGCNTrainer.iteration_step against oracle.models.sgcn_training_loop, no
    re-synchronisation).  Before the first Adam step the losses agree at 1e-5.  After it Adam's sign-like first step has
    moved every noise-level gradient entry by +-lr, differently on every run of ANY implementation: the yardstick is the
    oracle against ITSELF at two thread counts (1e-3 .. 9e-3 on the 240- and 5 000-vertex meshes), and the HIP path has
    to stay within 3 x that spread."""
    m, batch, net, state0, side = _trajectory_setup("c1")
    tr = train.SGCNTrainer(net, batch, lr=0.01, k1=4.0, accumulate=5)
    hip = [float(tr.iteration_step(mask_index=k)) for k in range(10)]
    runs = []
    old = torch.get_num_threads()
    try:
        for threads in (max(1, old // 2), old):
            torch.set_num_threads(threads)
            ora = OM.SGCNOracle()
            ora.load_state_dict(state0)
            z1 = side.z1.detach().clone().requires_grad_(True)
            runs.append(OM.sgcn_training_loop(ora, z1, side.x_pos, side.ei, side.faces, side.tgt, side.tfn, side.v_mask,
                                              side.f_mask, side.dms, range(10), batch=5, lr=0.01, k1=4.0))
    finally:
        torch.set_num_threads(old)
    ref = runs[1]
    err = [abs(a - b) / abs(b) for a, b in zip(hip, ref)]
    own = [abs(a - b) / abs(b) for a, b in zip(runs[0], ref)]
    print("trajectory (free-running) hip vs oracle", [f"{e:.1e}" for e in err], "oracle vs itself", [f"{e:.1e}" for e in own])
    assert max(err[:5]) < 1e-5, err
    assert max(err[5:]) < max(3.0 * max(own[5:]), 1e-2), (err, own)


# --------------------------------------------------------------------------------------
# the ZERO-LINE integration on hardware: the reference's model composition over the operator
# tier (what `from torch_geometric.nn import ChebConv, Sequential` resolves to after compat.install()), host-resident
# data moved to the device on every forward as util/networks.py:65 does
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("container", ["compat.Data", "plain tensors"])
@pytest.mark.parametrize("name,skip", [("sphere", False), ("torus", True)])
def test_reference_composition_on_the_operator_tier(name, skip, container, fixture_meshes, monkeypatch):
    from semigcn_amd import capi, compat, graph, nn as sgnn
    g2 = GU.load("g2_sgcn.npz")
    m = fixture_meshes[name]
    tag = f"{name}/skip{int(skip)}"
    net = OM.SGCNComposition(torch.device(DEV), sgnn, skip=skip)
    assert list(net.state_dict().keys()) == list(g2[f"{name}/state_dict_keys"])
    GU.fill_state(net, seed=314)
    net.to(DEV)
    z1 = torch.from_numpy(m.z1).requires_grad_(True)
    if container == "compat.Data":      # what util/datamaker.py:105 builds once compat.install() is in effect
        data = compat.Data(x=z1, z1=z1, x_pos=torch.from_numpy(m.x_pos), edge_index=torch.from_numpy(m.edge_index))
    else:                               # any other holder of plain host tensors

        class data:
            pass
        data.z1, data.x_pos, data.edge_index = z1, torch.from_numpy(m.x_pos), torch.from_numpy(m.edge_index)
    graph.clear_graph_cache()
    builds, syncs = [], []
    real_build, real_fp = capi.GraphHandle.from_edge_index, graph._fingerprint
    monkeypatch.setattr(capi.GraphHandle, "from_edge_index", classmethod(lambda cls, e, n: (builds.append(1), real_build(e, n))[1]))
    monkeypatch.setattr(graph, "_fingerprint", lambda e: (syncs.append(1), real_fp(e))[1])
    dm = g2[f"{name}/dm"]
    net.eval()
    with torch.no_grad():
        assert GU.rel_l2(net(data, dm).cpu(), g2[tag + "/eval_dm_ndarray"]) < 1e-5
        assert GU.rel_l2(net(data, torch.from_numpy(dm)).cpu(), g2[tag + "/eval_dm_tensor"]) < 1e-5
        assert GU.rel_l2(net(data, None).cpu(), g2[tag + "/eval_dm_none"]) < 1e-5
        for _ in range(7):
            net(data, dm)
    assert len(builds) == 1, builds                      # ten forwards, ONE sg_graph_create
    if container == "compat.Data":
        assert len(syncs) == 1                           # the same device tensor comes back from every .to(device)
        assert data.edge_index.to(DEV) is data.edge_index.to(DEV) and data.edge_index.to(DEV).is_cuda
    else:
        assert len(syncs) == 10                          # a fresh device tensor per forward: found by content each time
    net.train()
    pos = net(data, torch.from_numpy(dm))
    assert GU.rel_l2(pos.detach().cpu(), g2[tag + "/train_out"]) < 1e-5
    (pos * torch.from_numpy(GU.probe(tag + "/r", (m.num_vertices, 3))).to(DEV)).sum().backward()
    assert GU.rel_l2(z1.grad, g2[tag + "/dz1"]) < GU.grad_tolerance(3, 2e-3)     # (flip-aware bound: test_sgcn_vs_reference_golden)
    assert len(builds) == 1
