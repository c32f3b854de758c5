"""Host logic on CPU: autograd wiring of ChebConv, Sequential semantics, state-dict layout,
model composition -- with the HIP handles swapped for oracle-backed doubles (conftest)."""
import os

import numpy as np
import pytest
import torch

import golden_util as GU
from oracle import models as OM, pyg_restatement as P
from semigcn_amd import compat, nn as sgnn
from semigcn_amd.networks import SingleScaleGCN


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def nasty_graph(V=40, E=300, seed=0):
    """Asymmetric, with self-loops, duplicate edges and an isolated vertex."""
    rs = np.random.RandomState(seed)
    ei = rs.randint(0, V - 1, size=(2, E))       # vertex V-1 stays isolated
    ei[:, :10] = ei[:, 10:20]                    # duplicates
    ei[1, 20:30] = ei[0, 20:30]                  # self-loops
    return torch.from_numpy(ei).long()


@pytest.mark.parametrize("K", [1, 2, 3, 4])
@pytest.mark.parametrize("graph", ["torus", "nasty"])
@pytest.mark.parametrize("cin,cout", [(6, 10), (10, 6)])      # widening: aggregate-then-GEMM; narrowing: GEMM-then-aggregate
def test_chebconv_autograd_matches_oracle(cpu_kernels, fixture_meshes, K, graph, cin, cout):
    if graph == "torus":
        ei = torch.from_numpy(fixture_meshes["torus"].edge_index)
        V = fixture_meshes["torus"].num_vertices
    else:
        ei, V = nasty_graph(), 40
    mine, ora = sgnn.ChebConv(cin, cout, K=K), P.ChebConv(cin, cout, K=K)
    GU.fill_state(ora, seed=3)
    mine.load_state_dict(ora.state_dict())
    rs = np.random.RandomState(1)
    x = torch.from_numpy(rs.standard_normal((V, cin)).astype(np.float32))
    r = torch.from_numpy(rs.standard_normal((V, cout)).astype(np.float32))
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = mine(xa, ei), ora(xb, ei)
    assert rel(ya.detach(), yb.detach()) < 2e-6
    (ya * r).sum().backward()
    (yb * r).sum().backward()
    assert rel(xa.grad, xb.grad) < 5e-6
    for (n, p), (_, q) in zip(mine.named_parameters(), ora.named_parameters()):
        assert rel(p.grad, q.grad) < 5e-6, n


def test_chebconv_parameter_layout_and_init():
    torch.manual_seed(0)
    c = sgnn.ChebConv(32, 64, K=3)
    assert [n for n, _ in c.named_parameters()] == ["bias", "lins.0.weight", "lins.1.weight", "lins.2.weight"]
    assert c.lins[0].weight.shape == (64, 32) and c.bias.shape == (64,)
    bound = (6.0 / 96) ** 0.5
    assert float(c.lins[1].weight.abs().max()) <= bound and float(c.bias.abs().max()) == 0.0
    with pytest.raises(NotImplementedError):
        sgnn.GCNConv(4, 4)
    with pytest.raises(NotImplementedError):
        sgnn.ChebConv(4, 4, K=3, normalization="rw")


def test_sequential_threads_named_values():
    class Add(torch.nn.Module):
        def forward(self, a, b):
            return a + b

    seq = sgnn.Sequential("x, y", [(Add(), "x, y -> x"), torch.nn.ReLU(), (Add(), "x, y -> z"),
                                   (torch.nn.Identity(), "z -> z")])
    assert [n for n, _ in seq.named_children()] == ["module_0", "module_1", "module_2", "module_3"]
    x, y = torch.tensor([-3.0, 1.0]), torch.tensor([1.0, 1.0])
    assert torch.equal(seq(x, y), torch.relu(x + y) + y)
    with pytest.raises(ValueError):
        sgnn.Sequential("x", [torch.nn.ReLU()])


def test_compat_install_aliases_torch_geometric():
    import sys
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k.startswith("torch_geometric")}
    try:
        assert compat.install(force=True)
        from torch_geometric.nn import ChebConv, GCNConv, Sequential  # noqa: F401
        from torch_geometric.data import Data
        assert ChebConv is sgnn.ChebConv
        d = Data(x=torch.zeros(3, 2), edge_index=torch.tensor([[0, 1], [1, 0]]))
        assert d.num_nodes == 3 and d.num_edges == 2 and d.num_node_features == 2
        assert d.has_isolated_nodes() and not d.has_self_loops() and d["x"].shape == (3, 2)
    finally:
        for k in [k for k in sys.modules if k.startswith("torch_geometric")]:
            del sys.modules[k]
        sys.modules.update(saved)


class _Data:
    def __init__(self, m):
        self.z1 = torch.from_numpy(m.z1).clone().requires_grad_(True)
        self.x_pos = torch.from_numpy(m.x_pos)
        self.edge_index = torch.from_numpy(m.edge_index)


@pytest.mark.parametrize("skip", [False, True])
def test_sgcn_composition_vs_reference_golden(cpu_kernels, fixture_meshes, skip):
    g2 = GU.load("g2_sgcn.npz")
    m = fixture_meshes["torus"]
    tag = f"torus/skip{int(skip)}"
    net = SingleScaleGCN("cpu", skip=skip)
    sd = net.state_dict()
    assert list(sd.keys()) == list(g2["torus/state_dict_keys"])
    assert [",".join(map(str, v.shape)) for v in sd.values()] == list(g2["torus/state_dict_shapes"])
    assert sum(p.numel() for p in net.parameters()) == 1753475
    GU.fill_state(net, seed=314)
    data, dm = _Data(m), g2["torus/dm"]
    net.eval()
    with torch.no_grad():
        assert GU.rel_l2(net(data, torch.from_numpy(dm)), g2[tag + "/eval_dm_tensor"]) < 1e-5
        assert GU.rel_l2(net(data, dm), g2[tag + "/eval_dm_ndarray"]) < 1e-5
        assert GU.rel_l2(net(data, None), g2[tag + "/eval_dm_none"]) < 1e-5
    net.train()
    ora = OM.SGCNOracle(skip=skip)
    ora.load_state_dict(net.state_dict())
    ora.train()
    rec_o, rec_h = GU.ActivationMasks(ora), GU.ActivationMasks(net, fused=True)
    ora(torch.from_numpy(m.z1), torch.from_numpy(m.x_pos), torch.from_numpy(m.edge_index), torch.from_numpy(dm))
    pos = net(data, torch.from_numpy(dm))
    flips = rec_h.flips_against(rec_o, [net._layout(data)[2]] * len(rec_h.masks))
    rec_o.close(), rec_h.close()
    assert GU.rel_l2(pos.detach(), g2[tag + "/train_out"]) < 1e-5
    r = torch.from_numpy(GU.probe(tag + "/r", (m.num_vertices, 3)))
    (pos * r).sum().backward()
    gtol = GU.grad_tolerance(flips, 2e-3)
    assert GU.rel_l2(data.z1.grad, g2[tag + "/dz1"]) < gtol, flips
    golden = {k[len(tag + "/grad/"):]: g2[k] for k in g2.files if k.startswith(tag + "/grad/")}
    GU.check_grad_summary([(n, p.grad) for n, p in net.named_parameters() if p.grad is not None], golden, gtol, tag)
    for k in g2.files:
        if k.startswith(tag + "/bn/"):
            assert rel(net.state_dict()[k[len(tag + "/bn/"):]], g2[k]) < 1e-5


def test_graph_cache_reuses_and_invalidates(cpu_kernels, fixture_meshes):
    from semigcn_amd.graph import graph_for
    ei = torch.from_numpy(fixture_meshes["torus"].edge_index).clone()
    g1 = graph_for(ei, 240)
    assert graph_for(ei, 240) is g1                 # level 1: same tensor object
    builds = []
    from semigcn_amd import capi
    real = capi.GraphHandle.from_edge_index
    capi.GraphHandle.from_edge_index = classmethod(lambda cls, e, n: (builds.append(1), real(e, n))[1])
    try:
        # level 2: ANOTHER tensor object with the same edges (what `data.edge_index.to(device)` per forward produces,
        # util/networks.py:65) finds the graph by content, wherever the allocator put it; so does a column permutation
        assert graph_for(ei.clone(), 240) is g1
        assert graph_for(ei[:, torch.randperm(ei.shape[1])].contiguous(), 240) is g1
        assert not builds
        assert graph_for(ei, 241) is not g1             # same edges, another vertex count: a different operator
        ei[0, 0] = (ei[0, 0] + 1) % 240                 # in-place edit bumps the version counter
        assert graph_for(ei, 240) is not g1
        assert len(builds) == 2
    finally:
        capi.GraphHandle.from_edge_index = real


def test_resident_tensor_of_compat_data():
    """compat.Data keeps ``edge_index`` as a tensor whose ``.to(device)`` returns one cached device copy (a CPU target
    and every other operation see a plain tensor; the CUDA branch is exercised in tests/test_gpu_config_parity.py)."""
    import copy
    import pickle
    ei = torch.randint(0, 10, (2, 30))
    d = compat.Data(x=torch.randn(10, 3), edge_index=ei)
    r = d["edge_index"]
    assert isinstance(r, compat.ResidentTensor) and d.edge_index is r and d.num_edges == 30
    assert type(r.to("cpu")) is torch.Tensor and type(r + 1) is torch.Tensor and type(r[0]) is torch.Tensor
    assert torch.equal(torch.cat([r, r[[1, 0], :]], dim=1)[:, 30:], ei[[1, 0], :])
    for c in (copy.deepcopy(d)["edge_index"], pickle.loads(pickle.dumps(r))):
        assert isinstance(c, compat.ResidentTensor) and torch.equal(c, ei) and c.data_ptr() != r.data_ptr()
    assert type(r.to(torch.int32)) is torch.Tensor and r.to(torch.int32).dtype == torch.int32


def _mgcn_from_golden(device, g3, skip=False):
    from semigcn_amd.meshnet import MGCN
    eis = [torch.from_numpy(g3[f"edge_index/{l}"]) for l in range(4)]
    phs = [g3[f"pool_hash/{l}"] for l in range(3)]
    sms = [torch.from_numpy(g3[f"smposs/{l}"]) for l in range(4)]
    v_mask = torch.from_numpy(g3["v_masks/0"][:, 0] > 0)
    return MGCN.from_hierarchy(device, eis, phs, sms, ini_pos=torch.from_numpy(g3["poss/0"]), v_mask=v_mask, skip=skip)


def check_mgcn_against_golden(net, g3, device, out_tol, grad_tight):
    """Shared by the CPU (double) and GPU (HIP) MGCN tests."""
    assert list(net.state_dict().keys()) == list(g3["state_dict_keys"])
    for l in range(4):
        assert np.array_equal(net.v_masks_list[l].numpy(), g3[f"v_masks/{l}"]), l
        assert rel(net.poss_list[l].cpu(), g3[f"poss/{l}"]) < 1e-6, l
    GU.fill_state(net, seed=2718)
    for mod in net.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0

    class D:
        z1 = torch.from_numpy(g3["z1"]).to(device).requires_grad_(True)
        x_pos = torch.from_numpy(g3["smposs/0"]).to(device)
    dm = g3["dm"]
    net.eval()
    with torch.no_grad():
        for key, d in (("eval_dm_ndarray", dm), ("eval_dm_tensor", torch.from_numpy(dm).to(device)), ("eval_dm_none", None)):
            for l, p in enumerate(net(D, d)):
                assert GU.rel_l2(p.cpu(), g3[f"{key}/{l}"]) < out_tol, (key, l)
    net.train()
    ora = OM.MGCNOracle([torch.from_numpy(g3[f"edge_index/{l}"]) for l in range(4)], [g3[f"pool_hash/{l}"] for l in range(3)],
                        [torch.from_numpy(g3[f"smposs/{l}"]) for l in range(4)], drop=(0.0, 0.0, 0.0))
    ora.load_state_dict({k: v.cpu() for k, v in net.state_dict().items() if not k.endswith("pool_hash")})
    ora.train()
    rec_o, rec_h = GU.ActivationMasks(ora), GU.ActivationMasks(net, fused=True)
    ora(torch.from_numpy(g3["z1"]), dm)
    poss = net(D, dm)
    row_maps = None
    if getattr(net, "_orders", None) is not None:      # each level is processed in its own Morton order
        by_size = {rank.numel(): rank.cpu() for _, rank in net._orders}
        row_maps = [by_size[mk.shape[0]] for mk in rec_h.masks]
    flips = rec_h.flips_against(rec_o, row_maps)
    rec_o.close(), rec_h.close()
    for l, p in enumerate(poss):
        assert GU.rel_l2(p.detach().cpu(), g3[f"train_out/{l}"]) < out_tol, l
    w = [0.35, 0.3, 0.2, 0.15]
    loss = sum(wi * (p * torch.from_numpy(GU.probe(f"mgcn/r{l}", p.shape)).to(device)).sum()
               for l, (wi, p) in enumerate(zip(w, poss)))
    loss.backward()
    gtol = GU.grad_tolerance(flips, grad_tight)
    assert GU.rel_l2(D.z1.grad.cpu(), g3["dz1"]) < gtol, flips
    golden = {k[len("grad/"):]: g3[k] for k in g3.files if k.startswith("grad/")}
    GU.check_grad_summary([(n, p.grad) for n, p in net.named_parameters() if p.grad is not None], golden, gtol)


def test_mgcn_composition_vs_reference_golden(cpu_kernels):
    g3 = GU.load("g3_mgcn.npz")
    net = _mgcn_from_golden("cpu", g3)
    assert sum(p.numel() for p in net.parameters()) == 1514828
    check_mgcn_against_golden(net, g3, "cpu", 5e-5, 2e-3)


def test_mesh_pool_unpool_modules_vs_reference(cpu_kernels):
    from semigcn_amd.meshnet import MeshPool, MeshUnpool, pool_hash_to_mask, unpool_hash_to_mask
    g3 = GU.load("g3_mgcn.npz")
    ph = g3["pool_hash/0"]
    pool, unpool = MeshPool(pool_hash_to_mask(ph)), MeshUnpool(unpool_hash_to_mask(ph))
    assert list(pool.state_dict()) == ["pool_hash"] and list(unpool.state_dict()) == ["unpool_hash"]
    x = torch.from_numpy(g3["pool/x"]).requires_grad_(True)
    px = pool(x)
    assert rel(px.detach(), g3["pool/out"]) < 1e-6
    up = unpool(px)
    assert rel(up.detach(), g3["unpool/out"]) < 1e-6
    up.sum().backward()
    assert rel(x.grad, np.ones_like(g3["pool/x"])) < 1e-6   # mean then broadcast back: every fine vertex gets 1


class _FakeMesh:
    """Duck-typed stand-in for the reference's Mesh: what MGCN.__init__ touches."""

    def __init__(self, vs, edge_index, levels, path="/tmp/none/x.obj"):
        self.vs, self.edge_index, self._levels, self.path = vs, edge_index, levels, path
        self.pool_hash = None

    def simplification(self, target_v):
        ph, ei, vs = self._levels[0]
        m = _FakeMesh(vs, ei, self._levels[1:])
        m.pool_hash = [tuple(r) for r in ph.tolist()]
        return m


def test_mgcn_reference_constructor_path(cpu_kernels):
    """MGCN(device, smo_mesh, ini_mesh, v_mask): calls mesh.simplification like util/meshnet.py:182-201."""
    from semigcn_amd.meshnet import MGCN
    g3 = GU.load("g3_mgcn.npz")
    levels = [(g3[f"pool_hash/{l}"], torch.from_numpy(g3[f"edge_index/{l + 1}"]), g3[f"smposs/{l + 1}"]) for l in range(3)]
    smo = _FakeMesh(g3["smposs/0"], torch.from_numpy(g3["edge_index/0"]), levels)
    ini = _FakeMesh(g3["poss/0"], torch.from_numpy(g3["edge_index/0"]), levels)
    net = MGCN("cpu", smo, ini, torch.from_numpy(g3["v_masks/0"][:, 0] > 0))
    assert net.nvs == list(g3["nvs"]) and len(net.meshes) == 4
    assert [tuple(p.shape) for p in net.p_hashes] == [(154, 258), (92, 154), (55, 92)]
    check_mgcn_against_golden(net, g3, "cpu", 5e-5, 2e-3)


@pytest.mark.skipif(not __import__("oracle.ref_shim", fromlist=["x"]).available(), reason="/root/reference not present")
def test_reference_own_model_classes_run_on_this_operator_tier(cpu_kernels, fixture_meshes):
    """with compat.install() the reference's OWN SingleScaleGCN
    (util/networks.py, imported in place) runs on semigcn_amd.nn and reproduces its golden output."""
    import importlib
    import sys
    import types
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k.startswith(("torch_geometric", "util"))}
    had_turtle = "turtle" in sys.modules
    try:
        compat.install(force=True)
        if not had_turtle:
            sys.modules["turtle"] = types.SimpleNamespace(pd=None)      # util/mesh.py:1
        sys.path.insert(0, "/root/reference")
        sys.dont_write_bytecode = True
        ref_networks = importlib.import_module("util.networks")
        assert ref_networks.ChebConv is sgnn.ChebConv
        g2 = GU.load("g2_sgcn.npz")
        m = fixture_meshes["sphere"]
        net = ref_networks.SingleScaleGCN("cpu", skip=True)
        GU.fill_state(net, seed=314)
        net.eval()
        with torch.no_grad():
            out = net(_Data(m), g2["sphere/dm"])
        assert GU.rel_l2(out, g2["sphere/skip1/eval_dm_ndarray"]) < 1e-5
    finally:
        sys.path.remove("/root/reference")
        for k in [k for k in sys.modules if k.startswith(("torch_geometric", "util"))]:
            del sys.modules[k]
        if not had_turtle:
            sys.modules.pop("turtle", None)
        sys.modules.update(saved)


def test_weight_grad_split_reduction_matches_plain_product():
    from semigcn_amd.functional import weight_grad
    g = torch.Generator().manual_seed(0)
    for V in (100, 9001, 20480):
        dout, T = torch.randn(V, 7, generator=g), torch.randn(V, 12, generator=g)
        want = dout.double().t() @ T.double()
        assert rel(weight_grad(dout, T), want) < 1e-5
        assert rel(weight_grad(dout.bfloat16(), T.bfloat16()), dout.bfloat16().double().t() @ T.bfloat16().double()) < 2e-2


# --------------------------------------------------------------------------------------
# mesh preparation host logic (semigcn_amd/meshprep.py): bit packing runs anywhere, the kernels do not
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 40, 64, 65, 130])
def test_mask_bit_packing_round_trip(n):
    from semigcn_amd import meshprep
    rs = np.random.RandomState(n)
    m = torch.from_numpy((rs.random((37, n)) < 0.4).astype(np.float32))
    bits = meshprep.pack_bits(m)
    assert bits.dtype == torch.int64 and bits.shape == (37, (n + 63) // 64)
    assert torch.equal(meshprep.unpack_bits(bits, n), m != 0)
    if n >= 64:     # bit 63 lands on the sign bit, not in the next word
        one = torch.zeros(1, n)
        one[0, 63] = 1
        assert meshprep.pack_bits(one)[0, 0].item() == -(1 << 63)


def test_meshprep_has_no_cpu_path():
    from semigcn_amd import capi, meshprep
    faces = torch.tensor([[0, 1, 2], [0, 2, 3]])
    with pytest.raises(capi.SemigcnLibraryError, match="HIP device only"):
        meshprep.MeshTopology(faces, 4, device="cpu")


@pytest.mark.parametrize("name", ["sphere", "open"])
def test_bilateral_normal_loss_vs_reference_golden(name):
    """train.bilateral_normal_loss (plain torch ops) == the reference's fn_bnf_detach_loss (util/loss.py:196-253)
    with the reference's own f2f: loss value, filtered normals, d loss / d pos (golden g6)."""
    from semigcn_amd import train
    g = GU.load("g6_bnf.npz")
    pos = torch.from_numpy(g[f"{name}/pos"]).requires_grad_(True)
    faces, f2f = torch.from_numpy(g[f"{name}/faces"]), torch.from_numpy(g[f"{name}/f2f"])
    loss, new_fn = train.bilateral_normal_loss(pos, train.face_normals(pos, faces), faces, f2f, loop=5)
    loss.backward()
    assert abs(float(loss.detach()) - float(g[f"{name}/loss"])) < 1e-6 * float(g[f"{name}/loss"])
    assert np.abs(new_fn.numpy() - g[f"{name}/new_fn"]).max() < 1e-6 and not new_fn.requires_grad
    assert rel(pos.grad, g[f"{name}/dpos"]) < 1e-5
    if name == "open":
        assert (g[f"{name}/f2f"] < 0).sum() > 0            # the fixture does exercise the -1 padding


def test_wide_buffer_adoption_needs_a_registered_buffer(cpu_kernels):
    """a tensor that merely LOOKS like block 0 of a [V, K*C] buffer (a column slice of the caller's tensor,
    the gradient view torch.cat hands to a backward) must not be adopted -- its neighbouring columns are live data."""
    from semigcn_amd import functional as F_sg, nn as sgnn, synth
    m = synth.torus_mesh(12, 8)
    ei = torch.from_numpy(m.edge_index)
    V = m.num_vertices
    torch.manual_seed(0)
    # forward: a user slice big[:, :C] of a [V, 3C] tensor keeps its other columns
    conv = sgnn.ChebConv(6, 10, K=3)
    big = torch.randn(V, 18)
    keep = big.clone()
    assert F_sg._adopt_wide(big[:, :6], 3) is None
    y = conv(big[:, :6], ei)
    assert torch.equal(big, keep)
    assert torch.allclose(y, conv(keep[:, :6].contiguous(), ei), atol=1e-6)
    # backward: cat([conv_out, b, c]).backward() hands the narrowing conv a (3C,1)-strided gradient view
    conv2 = sgnn.ChebConv(10, 6, K=3)                     # Cout < Cin -> _ChebConvPostFn, which adopts its dout
    x = torch.randn(V, 10, requires_grad=True)
    b = torch.randn(V, 6, requires_grad=True)
    c = torch.randn(V, 6, requires_grad=True)
    r = torch.randn(V, 18)
    (torch.cat([conv2(x, ei), b, c], 1) * r).sum().backward()
    assert torch.equal(b.grad, r[:, 6:12]) and torch.equal(c.grad, r[:, 12:])
    x2 = x.detach().clone().requires_grad_(True)
    (conv2(x2, ei) * r[:, :6]).sum().backward()
    assert torch.allclose(x.grad, x2.grad, atol=1e-6)
    # the registered buffer IS adopted, exactly once per buffer shape
    wide = F_sg._new_wide(V, V, 6, 3, torch.float32, "cpu")
    assert F_sg._adopt_wide(wide, 3) is not None and F_sg._adopt_wide(wide, 2) is None
    assert F_sg._adopt_wide(wide.clone(), 3) is None


def test_plane_buffer_adoption_needs_a_registered_buffer():
    """functional._new_planes / _adopt_planes (the narrow layers' [K, V, C] recurrence buffers, include/semigcn.h
    sg_block_planar): plane 0 is an ordinary contiguous tensor for everybody but the block that adopts it, and only the
    registered buffer of the right shape and dtype is adopted."""
    from semigcn_amd import functional as F_sg
    x = F_sg._new_planes(40, 16, 3, torch.bfloat16, "cpu")
    assert x.shape == (40, 16) and x.is_contiguous() and x.storage_offset() == 0
    base = F_sg._adopt_planes(x, 3)
    assert base is not None and base.shape == (3, 40, 16) and base[0].data_ptr() == x.data_ptr()
    assert F_sg._adopt_planes(x, 2) is None                      # another K
    assert F_sg._adopt_planes(x.clone(), 3) is None              # a copy owns no neighbouring planes
    assert F_sg._adopt_planes(x[:20], 3) is None                 # fewer rows than the planes were made for
    assert F_sg._adopt_planes(base[1], 3) is None                # not plane 0
    assert F_sg._adopt_planes(torch.zeros(3, 40, 16, dtype=torch.bfloat16)[0], 3) is None      # looks alike, not registered
    assert F_sg._adopt_wide(x, 3) is None                        # and the column-block registry does not know it
    # the rule a producer applies: bf16, 8 .. 64 channels a power of two, K > 1
    assert F_sg._planes_wanted(16, 3, torch.bfloat16) and F_sg._planes_wanted(64, 3, torch.bfloat16)
    assert not F_sg._planes_wanted(16, 3, torch.float32) and not F_sg._planes_wanted(128, 3, torch.bfloat16)
    assert not F_sg._planes_wanted(16, 1, torch.bfloat16) and not F_sg._planes_wanted(4, 3, torch.bfloat16)
    del base, x


def test_weight_cache_invalidation(cpu_kernels):
    """writes through .data do not bump the version counter; invalidate_weight_cache / load_state_dict do."""
    from semigcn_amd import nn as sgnn, synth
    m = synth.torus_mesh(12, 8)
    ei = torch.from_numpy(m.edge_index)
    conv = sgnn.ChebConv(5, 7, K=3)
    x = torch.randn(m.num_vertices, 5)
    y0 = conv(x, ei).detach()
    with torch.no_grad():
        conv.lins[0].weight.mul_(2.0)                      # autograd-visible in-place update: picked up by itself
    y1 = conv(x, ei).detach()
    assert not torch.allclose(y0, y1)
    conv.lins[0].weight.data.mul_(0.5)                     # invisible to the version counter
    conv.invalidate_weight_cache()
    assert torch.allclose(conv(x, ei).detach(), y0, atol=1e-6)
    sd = {k: v.clone() for k, v in conv.state_dict().items()}
    sd["lins.1.weight"] = sd["lins.1.weight"] * 3.0
    conv.load_state_dict(sd)
    assert "_weight_cache" not in conv.__dict__
    assert not torch.allclose(conv(x, ei).detach(), y0)


def test_bench_refuses_to_run_fewer_ranks_than_asked():
    """`python bench.py --gpus 2` where fewer than 2 HIP devices are visible must exit non-zero before touching a GPU
    (it used to fall through to a single-rank run that reported n_gpus = 1)."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices visible: the refusal path cannot trigger")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SEMIGCN_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr and not r.stdout.strip()
    # and a launcher whose WORLD_SIZE disagrees with --gpus is an error too, not a silent resize
    env["WORLD_SIZE"], env["RANK"], env["LOCAL_RANK"] = "1", "0", "0"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_trainer_trajectory_equals_the_oracle_loop(cpu_kernels, fixture_meshes):
    """train.SGCNTrainer (5 accumulated forward + loss + backward passes, Adam, in-place gradient buffer) against the
    oracle's restatement of sgcn.py:118-147 over 10 iterations / 2 Adam steps -- the host logic of the trajectory tests
    that run on the device in tests/test_gpu_config_parity.py (see golden_util.synchronised_trajectory for why the two
    sides are re-synchronised at the optimiser steps)."""
    from semigcn_amd import synth, train
    m = fixture_meshes["torus"]
    V = m.num_vertices
    faces = torch.from_numpy(m.faces)
    target = torch.from_numpy(m.vs.astype(np.float32))
    v_keep = torch.from_numpy(m.v_mask.astype(np.float32)).view(-1, 1)
    f_keep = v_keep[faces[:, 0]] * v_keep[faces[:, 1]] * v_keep[faces[:, 2]]
    dms = torch.from_numpy(synth.make_dummy_masks(m.edge_index, V, dm_size=10, k=1, p=0.05))
    data = _Data(m)
    batch = train.MeshBatch(data, faces, target, train.face_normals(target, faces), v_keep, f_keep, dms)
    torch.manual_seed(5)
    net = SingleScaleGCN("cpu", reorder=False)
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    # a stale gradient from a pass made BEFORE the trainer exists must not reach the first Adam step (sgcn.py:121)
    (net(data, None) ** 2).mean().backward()
    tr = train.SGCNTrainer(net, batch, lr=0.01, k1=4.0, accumulate=5)
    assert all(float(p.grad.abs().max()) == 0.0 for p in net.parameters() if p.grad is not None)
    net.load_state_dict(state0)
    # (a) the trainer's own stepping against the oracle's loop, free-running: exact agreement before the first step
    mine = [float(tr.iteration_step(mask_index=k)) for k in range(10)]
    ora = OM.SGCNOracle()
    ora.load_state_dict(state0)
    tfn, f_mask = OM.compute_fn(target, faces), m.v_mask[m.faces].all(1)
    z1o, xpo, eio = torch.from_numpy(m.z1).requires_grad_(True), torch.from_numpy(m.x_pos), torch.from_numpy(m.edge_index)
    ref = OM.sgcn_training_loop(ora, z1o, xpo, eio, faces, target, tfn, m.v_mask, f_mask, dms, range(10), batch=5, lr=0.01, k1=4.0)
    err = [abs(a - b) / abs(b) for a, b in zip(mine, ref)]
    assert max(err[:5]) < 1e-5 and max(err) < 5e-2, err     # after a step: Adam's sign-step on noise-level entries
    # (b) re-synchronised at the optimiser steps: every loss tight, every above-noise parameter takes the same step
    net.load_state_dict(state0)
    ora.load_state_dict(state0)
    tr = train.SGCNTrainer(net, batch, lr=0.01, k1=4.0, accumulate=5)

    def oracle_iteration(k):
        dm = v_keep * dms[:, k:k + 1]
        pos = ora(z1o, xpo, eio, dm)
        loss = OM.mask_pos_rec_loss(pos, target, m.v_mask) + 4.0 * OM.mask_norm_rec_loss(OM.compute_fn(pos, faces), tfn, f_mask)
        loss.backward()
        return float(loss)
    # (240 vertices: ONE LeakyReLU sign that differs moves a gradient by ~0.5 % of its rms, hence the noise floor here)
    # and one sign that differs in the five passes after the first step moves the second step by a sizeable part of lr:
    # the second step is asserted at configuration size on the device only)
    errs, dev = GU.synchronised_trajectory(tr, net, ora, oracle_iteration, dms, noise=5e-2, step_tol=(0.02, 2.0))
    assert len(errs) == 10


def test_graph_cache_verifies_a_fingerprint_hit(cpu_kernels, fixture_meshes, monkeypatch):
    """A level-2 hit is accepted only after the edges have been compared with the ones the cached graph was built from
    (three wrapping sums are not the content): with the fingerprint forced to collide, another graph of the
    same shape gets its OWN CSR."""
    from semigcn_amd import graph
    graph.clear_graph_cache()
    monkeypatch.setattr(graph, "_fingerprint", lambda ei: (1, 2, 3))
    ei = torch.from_numpy(fixture_meshes["torus"].edge_index).clone()
    g1 = graph.graph_for(ei, 240)
    other = ei.clone()
    other[1, :6] = (other[1, :6] + 7) % 240            # same shape, same (forced) fingerprint, different edges
    g2 = graph.graph_for(other, 240)
    assert g2 is not g1
    x = torch.randn(240, 3)
    y1, y2 = torch.empty(240, 3), torch.empty(240, 3)
    g1.aggregate(x, y1), g2.aggregate(x, y2)
    assert not torch.equal(y1, y2)
    assert graph.graph_for(other.clone(), 240) is g2                                        # verified hit: elementwise equal
    assert graph.graph_for(other[:, torch.randperm(other.shape[1])].contiguous(), 240) is g2  # ... or equal as a multiset


def test_trainer_epochs_step_the_scheduler_and_redraw_the_batches(cpu_kernels, fixture_meshes):
    """Epoch-level semantics of sgcn.py:110-148: a fresh torch.randperm of ALL dummy masks per epoch, ``accumulate`` of them
    per optimiser step, ``scheduler.step()`` at the end of the epoch (the StepLR was built and never
    stepped)."""
    from semigcn_amd import synth, train
    m = fixture_meshes["torus"]
    faces = torch.from_numpy(m.faces)
    target = torch.from_numpy(m.vs.astype(np.float32))
    v_keep = torch.from_numpy(m.v_mask.astype(np.float32)).view(-1, 1)
    f_keep = v_keep[faces[:, 0]] * v_keep[faces[:, 1]] * v_keep[faces[:, 2]]
    dms = torch.from_numpy(synth.make_dummy_masks(m.edge_index, m.num_vertices, dm_size=4, k=1, p=0.05))
    batch = train.MeshBatch(_Data(m), faces, target, train.face_normals(target, faces), v_keep, f_keep, dms)
    torch.manual_seed(3)
    net = SingleScaleGCN("cpu", reorder=False)
    tr = train.SGCNTrainer(net, batch, lr=0.01, accumulate=2)
    tr.sched = torch.optim.lr_scheduler.StepLR(tr.opt, step_size=2, gamma=0.5)     # (the reference's period is 50 epochs)
    seen, real = [], tr.iteration_step
    tr.iteration_step = lambda k=None: (seen.append(k), real(k))[1]
    steps = []
    tr.opt.register_step_post_hook(lambda *a: steps.append(tr.iteration))
    torch.manual_seed(17)
    want = [torch.randperm(4).tolist() for _ in range(3)]
    torch.manual_seed(17)
    lrs = []
    for _ in range(3):
        loss = tr.train_epoch()
        assert loss.dim() == 0 and bool(torch.isfinite(loss))
        lrs.append(tr.opt.param_groups[0]["lr"])
    assert seen == sum(want, []) and want[0] != want[1]                       # every mask once per epoch, order redrawn
    assert steps == [2, 4, 6, 8, 10, 12]                                       # n_data / batch optimiser steps per epoch
    assert lrs == [0.01, 0.005, 0.005] and tr.epoch == 3                       # StepLR(2, 0.5) stepped once per epoch
    with pytest.raises(ValueError, match="cannot be cut"):
        train.SGCNTrainer(net, batch, accumulate=3).train_epoch()


def test_replay_check_does_not_spin_without_room_for_it():
    """train.replay_matches_eager with accumulate = 1 (`(iteration + 1) % 1 == 0` held forever and the loop ran
    optimiser steps while it span): it now declines -- the trainer goes eager -- after no iteration at all."""
    from semigcn_amd import train

    class Rep:
        graph = object()

    class Tr:
        accumulate, iteration, _graphed, _segmented = 1, 0, Rep(), None

        def iteration_step(self, k=None):
            raise AssertionError("no iteration may run")
    tr = Tr()
    assert train.replay_matches_eager(tr) is False and tr._graphed is None
