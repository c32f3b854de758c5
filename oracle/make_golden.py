"""Generates tests/golden/*.npz by running the REFERENCE'S OWN module code
(/root/reference/util/{mesh,networks,meshnet,loss,models}.py, imported in place
through oracle/ref_shim.py -- nothing is copied) on small seeded inputs.

    python -m oracle.make_golden [--only g4]     # run in the authoring container only

TEST INFRASTRUCTURE ONLY.  The [3P] torch-geometric operators underneath are
oracle/pyg_restatement.py (the real package is not installable here); what these
vectors pin is therefore (a) the reference's model composition, input
normalisation, masking, skip wiring, pooling matrices and losses exactly, and
(b) the restated operator semantics as cross-checked against the reference's
own dense D^-1/2 A D^-1/2 matrices (G0 below).
"""
from __future__ import annotations

import os
import sys
import tempfile
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ref_shim  # noqa: E402
from semigcn_amd import synth  # noqa: E402
import golden_util as GU  # noqa: E402

OUT = GU.GOLDEN_DIR


class _Data:
    def __init__(self, m):
        self.z1 = torch.from_numpy(m.z1).clone().requires_grad_(True)
        self.x_pos = torch.from_numpy(m.x_pos)
        self.edge_index = torch.from_numpy(m.edge_index)


def _ref_mesh(ref, m, tmp, name, **kw):
    path = os.path.join(tmp, name + ".obj")
    synth.write_obj(path, m.vs, m.faces)
    return ref.mesh.Mesh(path, **kw)


def g0_layout_and_dense(ref, meshes, tmp):
    """G0/G5: the reference Mesh's edge_index layout and its own adjacency / degree
    matrices (util/mesh.py:229-230,276-285) on the fixture meshes."""
    out = {}
    for name, m in meshes.items():
        rm = _ref_mesh(ref, m, tmp, name)
        rm.build_adj_mat()
        out[f"{name}/faces"] = m.faces
        out[f"{name}/vs"] = m.vs
        out[f"{name}/edge_index"] = rm.edge_index.numpy()
        L = -(rm.D_minus_half.to_dense().double() @ rm.Adj.to_dense().double() @ rm.D_minus_half.to_dense().double())
        out[f"{name}/lhat_dense_ref"] = L.numpy().astype(np.float32)
        out[f"{name}/fn"] = rm.fn.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "g0_mesh_layout.npz"), **out)
    return out


def g1_chebconv(meshes):
    """G1: single ChebConv layers, forward + gradients of L = sum(out * r)."""
    from oracle.pyg_restatement import ChebConv
    out = {}
    for name, m in meshes.items():
        ei = torch.from_numpy(m.edge_index)
        V = m.num_vertices
        shapes = [(4, 16), (32, 64)] + ([(256, 512)] if name == "sphere" else [(3, 5)])
        for cin, cout in shapes:
            tag = f"{name}/{cin}x{cout}"
            conv = ChebConv(cin, cout, K=3)
            GU.fill_state(conv, seed=11)
            rs = np.random.RandomState(1234 + cin)
            x = torch.from_numpy(rs.standard_normal((V, cin)).astype(np.float32)).requires_grad_(True)
            r = torch.from_numpy(rs.standard_normal((V, cout)).astype(np.float32))
            y = conv(x, ei)
            (y * r).sum().backward()
            out[tag + "/x"] = x.detach().numpy()
            out[tag + "/r"] = r.numpy()
            out[tag + "/out"] = y.detach().numpy()
            out[tag + "/dx"] = x.grad.numpy()
            for k, v in GU.grad_summary((n, p.grad) for n, p in conv.named_parameters()).items():
                out[tag + "/grad/" + k] = v
    np.savez_compressed(os.path.join(OUT, "g1_chebconv.npz"), **out)


def g2_sgcn(ref, meshes, ref_meshes):
    """G2: the reference's own SingleScaleGCN (util/networks.py) -- train/eval forwards with
    dm as Tensor / ndarray / None, skip on/off, BN running stats, and parameter gradients of
    the reference's real loss (util/loss.py:14-34,78-107 with k1 = 4, sgcn.py:137)."""
    out = {}
    for name, m in meshes.items():
        rm = ref_meshes[name]
        V = m.num_vertices
        dm_np = synth.make_dummy_masks(m.edge_index, V, dm_size=2, k=2, p=0.03, seed=317)[:, :1].copy()
        v_mask = m.v_mask
        f_mask = v_mask[m.faces].all(1)
        out[f"{name}/dm"] = dm_np
        out[f"{name}/v_mask"] = v_mask
        out[f"{name}/z1"] = m.z1
        out[f"{name}/x_pos"] = m.x_pos
        for skip in (False, True):
            tag = f"{name}/skip{int(skip)}"
            net = ref.networks.SingleScaleGCN("cpu", skip=skip)
            GU.fill_state(net, seed=314)
            data = _Data(m)
            net.eval()
            with torch.no_grad():
                out[tag + "/eval_dm_tensor"] = net(data, torch.from_numpy(dm_np)).numpy()
                out[tag + "/eval_dm_ndarray"] = net(data, dm_np).numpy()
                out[tag + "/eval_dm_none"] = net(data, None).numpy()
            net.train()
            pos = net(data, torch.from_numpy(dm_np))
            out[tag + "/train_out"] = pos.detach().numpy()
            # (a) smooth probe loss L = sum(pos * r): tight gradient parity
            r = torch.from_numpy(GU.probe(tag + "/r", (V, 3)))
            (pos * r).sum().backward(retain_graph=True)
            out[tag + "/dz1"] = data.z1.grad.numpy().copy()
            named = [(n, p.grad) for n, p in net.named_parameters() if p.grad is not None]
            for k, v in GU.grad_summary(named).items():
                out[tag + "/grad/" + k] = v
            # (b) the reference's real loss (L1 on unit normals has kinks -> looser gradient check)
            data.z1.grad = None
            net.zero_grad()
            norm = ref.models.compute_fn(pos, m.faces)
            loss_p = ref.loss.mask_pos_rec_loss(pos, rm.vs.astype(np.float32), v_mask)
            loss_n = ref.loss.mask_norm_rec_loss(norm, rm.fn.astype(np.float32), f_mask)
            loss = loss_p + 4.0 * loss_n
            loss.backward()
            out[tag + "/loss"] = np.array([loss_p.item(), loss_n.item(), loss.item()], np.float64)
            out[tag + "/loss_dz1"] = data.z1.grad.numpy().copy()
            sd = net.state_dict()
            for k in ("blocks.0.module_1.running_mean", "blocks.0.module_1.running_var",
                      "blocks.7.module_1.running_mean", "blocks.12.module_1.running_var"):
                out[tag + "/bn/" + k] = sd[k].numpy()
            if not skip:
                out[f"{name}/state_dict_keys"] = np.array(list(sd.keys()))
                out[f"{name}/state_dict_shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
    np.savez_compressed(os.path.join(OUT, "g2_sgcn.npz"), **out)


def g3_mgcn(ref, m, rm, tmp):
    """G3/G4: the reference's own MGCN (util/meshnet.py) on the 258-vertex sphere: hierarchy from
    its own QEM simplification, eval forward, train forward with dropout forced to 0, and the
    Tensor-dm == no-dm quirk (util/meshnet.py:287-290)."""
    out = {}
    smo = _ref_mesh(ref, type("M", (), {"vs": m.x_pos.astype(np.float64), "faces": m.faces})(), tmp, "sphere_smooth")
    v_mask = torch.from_numpy(m.v_mask)
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        net = ref.meshnet.MGCN("cpu", smo, rm, v_mask)
    finally:
        os.chdir(cwd)
    GU.fill_state(net, seed=2718)
    for mod in net.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    for l in range(3):
        out[f"pool_hash/{l}"] = np.array(net.meshes[l + 1].pool_hash, dtype=np.int64)
    for l in range(4):
        out[f"edge_index/{l}"] = net.edge_inds[l].numpy()
        out[f"smposs/{l}"] = net.smposs_list[l].numpy()
        out[f"poss/{l}"] = net.poss_list[l].numpy()
        out[f"v_masks/{l}"] = net.v_masks_list[l].numpy()
    out["nvs"] = np.array(net.nvs)
    out["z1"] = m.z1
    dm_np = synth.make_dummy_masks(m.edge_index, m.num_vertices, dm_size=2, k=2, p=0.03, seed=317)[:, :1].copy()
    out["dm"] = dm_np
    data = _Data(m)
    net.eval()
    with torch.no_grad():
        for key, dm in (("eval_dm_ndarray", dm_np), ("eval_dm_tensor", torch.from_numpy(dm_np)), ("eval_dm_none", None)):
            for l, p in enumerate(net(data, dm)):
                out[f"{key}/{l}"] = p.numpy()
    net.train()
    poss = net(data, dm_np)
    w = [0.35, 0.3, 0.2, 0.15]  # mgcn.py:82
    loss = sum(wi * (p * torch.from_numpy(GU.probe(f"mgcn/r{l}", p.shape))).sum()
               for l, (wi, p) in enumerate(zip(w, poss)))
    loss.backward()
    for l, p in enumerate(poss):
        out[f"train_out/{l}"] = p.detach().numpy()
    out["train_loss"] = np.float64(loss.item())
    out["dz1"] = data.z1.grad.numpy()
    named = [(n, p.grad) for n, p in net.named_parameters() if p.grad is not None]
    for k, v in GU.grad_summary(named).items():
        out["grad/" + k] = v
    sd = net.state_dict()
    out["state_dict_keys"] = np.array(list(sd.keys()))
    # MeshPool / MeshUnpool as the reference classes compute them (util/meshnet.py:9-27)
    rs = np.random.RandomState(5)
    x = torch.from_numpy(rs.standard_normal((m.num_vertices, 6)).astype(np.float32))
    pool = ref.meshnet.MeshPool(net.p_hashes[0])
    unpool = ref.meshnet.MeshUnpool(net.up_hashes[0])
    px = pool(x)
    out["pool/x"], out["pool/out"], out["unpool/out"] = x.numpy(), px.numpy(), unpool(px).numpy()
    np.savez_compressed(os.path.join(OUT, "g3_mgcn.npz"), **out)


def g4_meshprep(ref, meshes, tmp):
    """G4: the reference Mesh's own connectivity (util/mesh.py:60-100 edges, :214-227 f2f) and
    its own ``make_dummy_mask`` / ``vmask_to_fmask`` (util/datamaker.py:110-159) with numpy's
    global generator seeded, on the closed fixtures and on an open mesh (boundary edges)."""
    out = {}
    sph = meshes["sphere"]
    keep = np.ones(len(sph.faces), bool)
    keep[[3, 4, 5, 100, 101, 250]] = False
    cases = {"sphere": (sph.vs, sph.faces), "torus": (meshes["torus"].vs, meshes["torus"].faces),
             "open": (sph.vs, sph.faces[keep])}
    for name, (vs, faces) in cases.items():
        holder = type("M", (), {"vs": vs, "faces": faces})()
        # the reference's cotangent Laplacian (util/mesh.py:289-317) needs two faces per edge
        rm = _ref_mesh(ref, holder, tmp, "g4_" + name, build_mat=(name != "open"))
        os.makedirs(os.path.join(tmp, "dummy_mask"), exist_ok=True)
        out[f"{name}/faces"] = np.asarray(rm.faces, dtype=np.int64)
        out[f"{name}/num_vertices"] = np.int64(len(rm.vs))
        out[f"{name}/edges"] = rm.edges
        out[f"{name}/edge_index"] = rm.edge_index.numpy()
        out[f"{name}/f2f"] = rm.f2f.astype(np.int64)
        np.random.seed(317)
        vmask, fmask = ref.datamaker.make_dummy_mask(rm, dm_size=3, kn=[1, 2, 3],
                                                     exist_face=np.ones(len(rm.faces)))
        out[f"{name}/vmask_dummy"] = vmask.numpy().astype(np.uint8)
        out[f"{name}/fmask_dummy"] = fmask.numpy().astype(np.uint8)
        rs = np.random.RandomState(9)
        vm = rs.random(len(rm.vs)) > 0.1
        out[f"{name}/vm"] = vm
        out[f"{name}/fm"] = ref.datamaker.vmask_to_fmask(rm, vm.astype(np.float32)).numpy()
    np.savez_compressed(os.path.join(OUT, "g4_meshprep.npz"), **out)


def g5_refine(ref, meshes, ref_meshes):
    """G5: the reference's own ``Mesh.mesh_merge`` (util/mesh.py:678-698, dense float32 solve) as
    sgcn.py:189 calls it: Lap and AdjI from the reference Mesh, a perturbed position field as the
    network output, the fixture's v_mask as ``preserve``, w = mu in {1.0, 0.3}, and one case with a
    boundary weight."""
    out = {}
    for name, m in meshes.items():
        rm = ref_meshes[name]
        rs = np.random.RandomState(41)
        new_pos = torch.from_numpy((m.x_pos + 0.02 * rs.standard_normal(m.x_pos.shape)).astype(np.float32))
        keep = torch.from_numpy(m.v_mask)
        out[f"{name}/edge_index"] = rm.edge_index.numpy()
        out[f"{name}/org_pos"] = rm.vs.astype(np.float32)
        out[f"{name}/new_pos"] = new_pos.numpy()
        out[f"{name}/preserve"] = m.v_mask
        for tag, (w, wb) in {"w1": (1.0, 0.0), "w03": (0.3, 0.0), "wb": (1.0, 0.5)}.items():
            ref_pos = ref.mesh.Mesh.mesh_merge(rm.Lap, rm, new_pos, keep, w=w, w_b=wb)
            out[f"{name}/{tag}/ref_pos"] = ref_pos.numpy()
            out[f"{name}/{tag}/w"] = np.array([w, wb])
    np.savez_compressed(os.path.join(OUT, "g5_refine.npz"), **out)


def g6_bnf(ref, meshes, tmp):
    """G6: the reference's ``fn_bnf_detach_loss`` (util/loss.py:196-253; the -CAD term of sgcn.py:133-135)
    with its own ``Mesh.f2f``, on the closed sphere and on the open mesh (rows of f2f padded with -1):
    loss, filtered normals, and d loss / d pos through ``compute_fn``."""
    out = {}
    sph = meshes["sphere"]
    keep = np.ones(len(sph.faces), bool)
    keep[[3, 4, 5, 100, 101, 250]] = False
    for name, faces in (("sphere", sph.faces), ("open", sph.faces[keep])):
        holder = type("M", (), {"vs": sph.vs, "faces": faces})()
        rm = _ref_mesh(ref, holder, tmp, "g6_" + name, build_mat=(name != "open"))
        rs = np.random.RandomState(77)
        pos = torch.from_numpy((sph.vs + 0.03 * rs.standard_normal(sph.vs.shape)).astype(np.float32)).requires_grad_(True)
        fn = ref.models.compute_fn(pos, rm.faces)
        loss, new_fn = ref.loss.fn_bnf_detach_loss(pos, fn, rm, loop=5)
        loss.backward()
        out[f"{name}/faces"] = np.asarray(rm.faces, dtype=np.int64)
        out[f"{name}/f2f"] = rm.f2f.astype(np.int64)
        out[f"{name}/pos"] = pos.detach().numpy()
        out[f"{name}/loss"] = np.float64(loss.item())
        out[f"{name}/new_fn"] = new_fn.numpy()
        out[f"{name}/dpos"] = pos.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g6_bnf.npz"), **out)


def main():
    only = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else None
    warnings.simplefilter("ignore")
    torch.manual_seed(0)
    torch.set_num_threads(4)
    os.makedirs(OUT, exist_ok=True)
    ref = ref_shim.load()
    meshes = {"sphere": synth.octahedron_sphere(3), "torus": synth.torus_mesh(20, 12)}
    with tempfile.TemporaryDirectory() as tmp:
        if only is None or "g0" in only:
            g0_layout_and_dense(ref, meshes, tmp)
        ref_meshes = {n: _ref_mesh(ref, m, tmp, n) for n, m in meshes.items()}
        if only is None or "g1" in only:
            g1_chebconv(meshes)
        if only is None or "g2" in only:
            g2_sgcn(ref, meshes, ref_meshes)
        if only is None or "g3" in only:
            g3_mgcn(ref, meshes["sphere"], ref_meshes["sphere"], tmp)
        if only is None or "g4" in only:
            g4_meshprep(ref, meshes, tmp)
        if only is None or "g5" in only:
            g5_refine(ref, meshes, ref_meshes)
        if only is None or "g6" in only:
            g6_bnf(ref, meshes, tmp)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
