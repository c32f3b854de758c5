"""Host logic on CPU: autograd wiring of ChebConv, Sequential semantics, state-dict layout,
model composition -- with the HIP handles swapped for oracle-backed doubles (conftest)."""
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
def test_chebconv_autograd_matches_oracle(cpu_kernels, fixture_meshes, K, graph):
    if graph == "torus":
        ei = torch.from_numpy(fixture_meshes["torus"].edge_index)
        V = fixture_meshes["torus"].num_vertices
    else:
        ei, V = nasty_graph(), 40
    mine, ora = sgnn.ChebConv(6, 10, K=K), P.ChebConv(6, 10, K=K)
    GU.fill_state(ora, seed=3)
    mine.load_state_dict(ora.state_dict())
    rs = np.random.RandomState(1)
    x = torch.from_numpy(rs.standard_normal((V, 6)).astype(np.float32))
    r = torch.from_numpy(rs.standard_normal((V, 10)).astype(np.float32))
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
    pos = net(data, torch.from_numpy(dm))
    assert GU.rel_l2(pos.detach(), g2[tag + "/train_out"]) < 1e-5
    r = torch.from_numpy(GU.probe(tag + "/r", (m.num_vertices, 3)))
    (pos * r).sum().backward()
    assert GU.rel_l2(data.z1.grad, g2[tag + "/dz1"]) < 2e-3
    golden = {k[len(tag + "/grad/"):]: g2[k] for k in g2.files if k.startswith(tag + "/grad/")}
    GU.check_grad_summary([(n, p.grad) for n, p in net.named_parameters() if p.grad is not None], golden, 2e-3, tag)
    for k in g2.files:
        if k.startswith(tag + "/bn/"):
            assert rel(net.state_dict()[k[len(tag + "/bn/"):]], g2[k]) < 1e-5


def test_graph_cache_reuses_and_invalidates(cpu_kernels, fixture_meshes):
    from semigcn_amd.graph import graph_for
    ei = torch.from_numpy(fixture_meshes["torus"].edge_index).clone()
    g1 = graph_for(ei, 240)
    assert graph_for(ei, 240) is g1                 # level 1: same tensor object
    assert graph_for(ei.clone(), 240) is not g1     # different storage -> rebuilt
    ei[0, 0] = (ei[0, 0] + 1) % 240                 # in-place edit bumps the version counter
    assert graph_for(ei, 240) is not g1
