"""The network's input and loss steps as single launches (round 4): sg_input_bounds + sg_input_prep_bwd_routed against the
torch.min / torch.max composition of util/networks.py:67-79, sg_mesh_loss_finalize against the scalar arithmetic of
sgcn.py:130-138 / mgcn.py:138-143 on the sums of sg_mesh_loss_fwd."""
import numpy as np
import pytest
import torch

import golden_util as GU
from semigcn_amd import functional as F_sg, synth, train
from semigcn_amd.networks import _column_min_max

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("V", [5, 1000, 70001])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_input_prep_takes_the_bounds_itself(V, dtype):
    gen = torch.Generator().manual_seed(V)
    z = (torch.randn(V, 3, generator=gen) * torch.tensor([0.05, 0.2, 0.01])).to(DEV)
    dm = (torch.rand(V, 1, generator=gen) > 0.3).float().to(DEV)
    order = torch.randperm(V, generator=gen).to(DEV)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(V, device=DEV)
    r = torch.randn(V, 4, generator=gen).to(DEV)
    za = z.clone().requires_grad_(True)
    lo, hi = _column_min_max(za)
    xa = F_sg.input_prep(za, lo, hi, dm, order, rank, dtype)
    (xa.float() * r).sum().backward()
    zb = z.clone().requires_grad_(True)
    xb = F_sg.input_prep(zb, None, None, dm, order, rank, dtype)
    (xb.float() * r).sum().backward()
    assert torch.equal(xa, xb)
    scale = float(za.grad.abs().max())
    assert float((za.grad - zb.grad).abs().max()) <= 2e-6 * scale
    # the bounds' gradients went to one vertex per bound (the arg-extremes), as autograd routes them through torch.min / max
    bounds, arg = F_sg.capi.input_bounds(z)
    assert torch.equal(bounds[:3], z.min(0)[0]) and torch.equal(bounds[3:], z.max(0)[0])
    assert torch.equal(z[arg[:3], torch.arange(3)], bounds[:3]) and torch.equal(z[arg[3:], torch.arange(3)], bounds[3:])


def test_input_bounds_ties_go_to_the_lowest_vertex():
    z = torch.zeros(3000, 3, device=DEV)
    z[[7, 2000], 0] = -1.0
    z[[11, 1500, 2999], 2] = 4.0
    bounds, arg = F_sg.capi.input_bounds(z)
    assert arg.tolist() == [7, 0, 0, 0, 0, 11] and bounds.tolist() == [-1.0, 0.0, 0.0, 0.0, 0.0, 4.0]


def test_mesh_loss_scalar_equals_the_composition(fixture_meshes):
    m = synth.torus_mesh(60, 40)
    faces = torch.from_numpy(m.faces).to(DEV)
    target = torch.from_numpy(m.vs.astype(np.float32)).to(DEV)
    tfn = train.face_normals(target, faces)
    v_keep = torch.from_numpy(m.v_mask.astype(np.float32)).to(DEV)
    f_keep = v_keep[faces[:, 0]] * v_keep[faces[:, 1]] * v_keep[faces[:, 2]]
    n_v, n_f = float(v_keep.sum()), float(f_keep.sum())
    pos0 = target + 0.05 * torch.randn_like(target)
    for w, k1 in ((1.0, 4.0), (0.35, 4.0)):
        pa = pos0.clone().requires_grad_(True)
        s = F_sg.mesh_loss_sums(pa, faces, target, v_keep, tfn, f_keep)
        la = w * torch.sqrt(s[0] / n_v + 1.0e-6) + k1 * (s[1] / n_f)
        (3.0 * la).backward()
        pb = pos0.clone().requires_grad_(True)
        lb = F_sg.mesh_loss(pb, faces, target, v_keep, tfn, f_keep, n_v, n_f, w, k1)
        (3.0 * lb).backward()
        fa, fb = float(la.detach()), float(lb.detach())
        assert lb.dim() == 0 and abs(fa - fb) <= 2e-7 * abs(fa)
        assert GU.rel_l2(pb.grad.cpu(), pa.grad.cpu()) < 1e-6
    # one resolution's position term (mgcn.py:138-143): no faces
    pa = pos0.clone().requires_grad_(True)
    la = 0.3 * train.masked_position_rmse(pa, target, v_keep.view(-1, 1), n_v)
    la.backward()
    pb = pos0.clone().requires_grad_(True)
    lb = F_sg.mesh_loss(pb, None, target, v_keep, None, None, n_v, 0.0, 0.3)
    lb.backward()
    assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(la)) and GU.rel_l2(pb.grad.cpu(), pa.grad.cpu()) < 1e-6
