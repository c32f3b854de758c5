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

import torch


class ResidentTensor(torch.Tensor):
    """A per-mesh CONSTANT held on the host whose ``.to(device)`` hands back ONE cached device copy instead of a fresh
    upload per call.  The reference moves its graph to the device on every forward (util/networks.py:65:
    ``data.edge_index.to(self.device)``, 96 MB at V = 1 M) and its ChebConv then re-derives the Laplacian from it 13
    times; with the same device tensor OBJECT coming back each time, the graph prepared for it is found on the object
    (graph._Cache level 1): no copy, no hash, no synchronisation per forward.  The copy is refreshed when the host
    tensor is modified in place (version counter).  Every other operation sees a plain tensor."""

    __torch_function__ = torch._C._disabled_torch_function_impl

    @staticmethod
    def wrap(t: torch.Tensor) -> "ResidentTensor":
        if isinstance(t, ResidentTensor):
            return t
        out = torch.Tensor._make_subclass(ResidentTensor, t.detach(), False)
        out.__dict__["_resident"] = {}
        return out

    def to(self, *args, **kwargs):
        target = None
        if len(args) == 1 and not kwargs and isinstance(args[0], (str, torch.device, int)):
            target = torch.device("cuda", args[0]) if isinstance(args[0], int) else torch.device(args[0])
        elif not args and set(kwargs) == {"device"}:
            target = torch.device(kwargs["device"])
        plain = self.as_subclass(torch.Tensor)
        if target is None or target.type == "cpu":
            return plain.to(*args, **kwargs)
        if target.index is None and target.type == "cuda":
            target = torch.device("cuda", torch.cuda.current_device())
        cache = self.__dict__.setdefault("_resident", {})
        ent = cache.get(target)
        if ent is None or ent[0] != self._version:
            ent = cache[target] = (self._version, plain.to(target))
        return ent[1]

    def __deepcopy__(self, memo):       # sgcn.py:66 deep-copies the dataset: the copy gets its own (empty) cache
        return ResidentTensor.wrap(self.as_subclass(torch.Tensor).clone())

    def __reduce_ex__(self, proto):
        return (ResidentTensor.wrap, (self.as_subclass(torch.Tensor).clone(),))


class Data:
    """The keyed container util/datamaker.py:105-106 builds and ``Dataset.__init__``
    (util/datamaker.py:13-25) reads back.  ``edge_index`` is kept as a ResidentTensor (above)."""

    def __init__(self, **fields):
        f = dict(fields)
        ei = f.get("edge_index")
        if isinstance(ei, torch.Tensor) and not ei.is_cuda and not ei.requires_grad:
            f["edge_index"] = ResidentTensor.wrap(ei)
        self.__dict__["_fields"] = f

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
