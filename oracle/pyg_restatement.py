"""Plain-torch CPU restatement of the torch-geometric operators on the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference calls ``torch_geometric.nn.{ChebConv, GCNConv, Sequential}``
(/root/reference/util/networks.py:4,42,49; util/meshnet.py:6,40-58,106-124,
224-240) and ``torch_geometric.data.Data`` (util/datamaker.py:9,105).  That
dependency -- torch-geometric==2.2.0 on torch-scatter==2.1.0
(/root/reference/requirements.txt:15,19) -- is not vendored, not installed and
not fetchable here, so its published algorithm is restated below with the very
ATen ops it dispatches to on CPU (``index_select`` -> broadcast multiply ->
``scatter_add_``; three separate bias-free ``linear`` calls).  Parity of THIS
file is unpinned by the reference's tests (it has none); tests/test_oracle.py
pins it against the reference's own dense ``D^-1/2 A D^-1/2`` matrices.

Nothing here is tuned: it is deliberately the literal op sequence, because it
doubles as the "reference CPU path" timed by bench.py's ``cpu_baseline`` leg.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple, Union

import torch
import torch.nn as nn
from torch import Tensor


# --------------------------------------------------------------------------- #
# torch_geometric.utils pieces used by ChebConv.__norm__ (PyG 2.2.0)
# --------------------------------------------------------------------------- #
def remove_self_loops(edge_index: Tensor, edge_weight=None):
    keep = edge_index[0] != edge_index[1]
    edge_index = edge_index[:, keep]
    if edge_weight is not None:
        edge_weight = edge_weight[keep]
    return edge_index, edge_weight


def add_self_loops(edge_index: Tensor, edge_weight: Tensor, fill_value: float, num_nodes: int):
    loop = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    loop = loop.unsqueeze(0).repeat(2, 1)
    loop_w = edge_weight.new_full((num_nodes,), fill_value)
    return torch.cat([edge_index, loop], dim=1), torch.cat([edge_weight, loop_w], dim=0)


def scatter_sum(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """torch_scatter.scatter(src, index, dim=0, dim_size=.., reduce='sum'):
    broadcast ``index`` to ``src``'s shape, then ``zeros.scatter_add_``."""
    if src.dim() == 1:
        out = src.new_zeros(dim_size)
        return out.scatter_add_(0, index, src)
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    out = src.new_zeros((dim_size,) + tuple(src.shape[1:]))
    return out.scatter_add_(0, idx, src)


def get_laplacian_sym(edge_index: Tensor, dtype, num_nodes: int):
    """get_laplacian(edge_index, None, 'sym', dtype, N): L = I - D^-1/2 A D^-1/2
    as COO with the E off-diagonal entries first and N unit diagonal entries after."""
    edge_index, _ = remove_self_loops(edge_index)
    w = torch.ones(edge_index.size(1), dtype=dtype, device=edge_index.device)
    row, col = edge_index[0], edge_index[1]
    deg = scatter_sum(w, row, num_nodes)
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float("inf"), 0)
    w = dis[row] * w * dis[col]
    return add_self_loops(edge_index, -w, 1.0, num_nodes)


class PygLinear(nn.Module):
    """torch_geometric.nn.dense.linear.Linear(in, out, bias=False,
    weight_initializer='glorot'): weight [out, in] ~ U(-a, a), a = sqrt(6/(in+out))."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        self.reset_parameters()

    def reset_parameters(self):
        a = math.sqrt(6.0 / (self.weight.size(-2) + self.weight.size(-1)))
        with torch.no_grad():
            self.weight.uniform_(-a, a)

    def forward(self, x: Tensor) -> Tensor:
        return torch.nn.functional.linear(x, self.weight, None)


class ChebConv(nn.Module):
    """torch_geometric.nn.ChebConv(in, out, K, normalization='sym', bias=True).

    forward(x[V,Cin], edge_index[2,E]) with lambda_max=None -> 2.0:
      Tx0 = x; Tx1 = L^ x; Tx_k = 2 L^ Tx_{k-1} - Tx_{k-2}; out = sum_k lins[k](Tx_k) + bias
    where L^ = 2 L / lambda_max - I is materialised as a COO list of
    E + N (+1 diag) + N (-1 diag) weighted entries and applied by
    gather -> multiply -> scatter_add (MessagePassing.propagate, aggr='add',
    flow source_to_target: gather at edge_index[0], reduce at edge_index[1]).
    """

    def __init__(self, in_channels: int, out_channels: int, K: int,
                 normalization: str = "sym", bias: bool = True):
        super().__init__()
        assert K > 0 and normalization == "sym"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.normalization = normalization
        # PyG 2.2.0: each Linear initialises itself on construction ...
        self.lins = nn.ModuleList([PygLinear(in_channels, out_channels) for _ in range(K)])
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        # ... and ChebConv.__init__ ends with reset_parameters(), drawing them again.
        self.reset_parameters()

    def reset_parameters(self):
        for lin in self.lins:
            lin.reset_parameters()
        if self.bias is not None:
            with torch.no_grad():
                self.bias.zero_()

    @staticmethod
    def norm(edge_index: Tensor, num_nodes: int, dtype, lambda_max: float = 2.0):
        edge_index, _ = remove_self_loops(edge_index)
        edge_index, w = get_laplacian_sym(edge_index, dtype, num_nodes)
        w = (2.0 * w) / lambda_max
        w.masked_fill_(w == float("inf"), 0)
        edge_index, w = add_self_loops(edge_index, w, -1.0, num_nodes)
        return edge_index, w

    @staticmethod
    def propagate(edge_index: Tensor, x: Tensor, norm: Tensor) -> Tensor:
        x_j = x.index_select(0, edge_index[0])
        msg = norm.view(-1, 1) * x_j
        return scatter_sum(msg, edge_index[1], x.size(0))

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        edge_index, norm = self.norm(edge_index, x.size(0), x.dtype)
        Tx_0 = x
        Tx_1 = x
        out = self.lins[0](Tx_0)
        if len(self.lins) > 1:
            Tx_1 = self.propagate(edge_index, x, norm)
            out = out + self.lins[1](Tx_1)
        for lin in self.lins[2:]:
            Tx_2 = self.propagate(edge_index, Tx_1, norm)
            Tx_2 = 2.0 * Tx_2 - Tx_0
            out = out + lin(Tx_2)
            Tx_0, Tx_1 = Tx_1, Tx_2
        if self.bias is not None:
            out = out + self.bias
        return out


class GCNConv(nn.Module):
    """Name-only placeholder: the reference imports GCNConv but its
    ``conv == "gcnconv"`` branches are dead (util/networks.py:13, meshnet.py:36)."""

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("GCNConv is never instantiated by the reference (dead branch)")


def _parse_desc(desc: str) -> Tuple[List[str], List[str]]:
    ins, outs = desc.split("->")
    return [s.strip() for s in ins.split(",")], [s.strip() for s in outs.split(",")]


class Sequential(nn.Module):
    """torch_geometric.nn.Sequential(input_args, modules).

    Children are registered as ``module_{i}``; an entry given as a bare module
    (no "a, b -> c" string) consumes and produces the previous entry's output
    names (util/networks.py:43-45).  forward returns the last entry's output."""

    def __init__(self, input_args: str, modules: Sequence[Union[nn.Module, Tuple[nn.Module, str]]]):
        super().__init__()
        self._input_args = [s.strip() for s in input_args.split(",")]
        self._calls: List[Tuple[str, List[str], List[str]]] = []
        prev_out = None
        for i, entry in enumerate(modules):
            if isinstance(entry, (tuple, list)):
                mod, desc = entry
                ins, outs = _parse_desc(desc)
            else:
                mod = entry
                if prev_out is None:
                    raise ValueError("first Sequential entry needs an explicit signature")
                ins, outs = list(prev_out), list(prev_out)
            name = f"module_{i}"
            if isinstance(mod, nn.Module):
                self.add_module(name, mod)
            else:  # plain callable
                setattr(self, name, mod)
            self._calls.append((name, ins, outs))
            prev_out = outs

    def forward(self, *args):
        env = dict(zip(self._input_args, args))
        out = None
        for name, ins, outs in self._calls:
            out = getattr(self, name)(*[env[k] for k in ins])
            if len(outs) == 1:
                env[outs[0]] = out
            else:
                for k, v in zip(outs, out):
                    env[k] = v
        return out


class Data:
    """Keyed container offering what Dataset.__init__ touches
    (/root/reference/util/datamaker.py:15-25)."""

    def __init__(self, **kw):
        self._store = dict(kw)

    @property
    def keys(self):
        return list(self._store.keys())

    def __getitem__(self, k):
        return self._store[k]

    @property
    def num_nodes(self):
        x = self._store.get("x")
        return None if x is None else x.size(0)

    @property
    def num_edges(self):
        return self._store["edge_index"].size(1)

    @property
    def num_node_features(self):
        x = self._store.get("x")
        return 0 if x is None else (1 if x.dim() == 1 else x.size(1))

    def has_isolated_nodes(self):
        ei = self._store["edge_index"]
        return bool(torch.unique(ei).numel() < self.num_nodes)

    def has_self_loops(self):
        ei = self._store["edge_index"]
        return bool((ei[0] == ei[1]).any())
