"""Makes ``import torch_geometric`` resolve to this package for the names the
reference uses, so ``sgcn.py`` / ``mgcn.py`` / ``util/*.py`` run unmodified:

    import semigcn_amd.compat; semigcn_amd.compat.install()
    # from torch_geometric.nn import GCNConv, ChebConv, Sequential   (util/networks.py:4)
    # from torch_geometric.data import Data                           (util/datamaker.py:9)

A real torch_geometric installation is never shadowed unless ``force=True``.
"""
from __future__ import annotations

import importlib.util
import sys
import types


class Data:
    """The keyed container util/datamaker.py:105-106 builds and ``Dataset.__init__``
    (util/datamaker.py:13-25) reads back."""

    def __init__(self, **fields):
        self.__dict__["_fields"] = dict(fields)

    def __getitem__(self, key):
        return self._fields[key]

    def __getattr__(self, key):
        try:
            return self.__dict__["_fields"][key]
        except KeyError:
            raise AttributeError(key) from None

    @property
    def keys(self):
        return list(self._fields)

    @property
    def num_nodes(self):
        x = self._fields.get("x")
        return None if x is None else x.shape[0]

    @property
    def num_edges(self):
        return self._fields["edge_index"].shape[1]

    @property
    def num_node_features(self):
        x = self._fields.get("x")
        return 0 if x is None else (1 if x.dim() == 1 else x.shape[1])

    def has_isolated_nodes(self):
        import torch
        return bool(torch.unique(self._fields["edge_index"]).numel() < self.num_nodes)

    def has_self_loops(self):
        ei = self._fields["edge_index"]
        return bool((ei[0] == ei[1]).any())


def install(force: bool = False) -> bool:
    """Register ``torch_geometric``, ``torch_geometric.nn`` and ``torch_geometric.data``
    aliases.  Returns True when installed, False when a real torch_geometric exists."""
    if not force and "torch_geometric" not in sys.modules and importlib.util.find_spec("torch_geometric"):
        return False
    from . import nn as sg_nn

    root = types.ModuleType("torch_geometric")
    m_nn = types.ModuleType("torch_geometric.nn")
    m_data = types.ModuleType("torch_geometric.data")
    m_nn.ChebConv, m_nn.GCNConv, m_nn.Sequential = sg_nn.ChebConv, sg_nn.GCNConv, sg_nn.Sequential
    m_data.Data = Data
    root.nn, root.data = m_nn, m_data
    root.__semigcn_amd__ = True
    sys.modules.update({"torch_geometric": root, "torch_geometric.nn": m_nn, "torch_geometric.data": m_data})
    return True
