"""Per-edge_index graph state: built once, reused by every conv call.

The reference's ChebConv re-derives the scaled-Laplacian edge weights inside
every forward ([3P] ChebConv.__norm__, reached 13x per SGCN forward from
util/networks.py:42,49 and 33x per MGCN forward from util/meshnet.py:40-240).
Here an ``edge_index`` tensor is turned into a device CSR (``capi.GraphHandle``)
the first time it is seen and looked up afterwards.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import capi

_ATTR = "_semigcn_graph"


class MeshGraph:
    """Scaled Laplacian ``L^ = -D^-1/2 A D^-1/2`` of one mesh ``edge_index``
    ([2, E] int64, reference layout util/mesh.py:229-230) as a device CSR."""

    def __init__(self, handle: capi.GraphHandle, num_vertices: int, num_edges: int):
        self.handle = handle
        self.num_vertices = num_vertices
        self.num_edges = num_edges  # directed, as given (self-loops included)
        if type(self).aggregate is MeshGraph.aggregate:
            self.aggregate = handle.spmm       # the bound method itself: one Python frame less per launch

    @classmethod
    def from_edge_index(cls, edge_index: torch.Tensor, num_vertices: int) -> "MeshGraph":
        h = capi.GraphHandle.from_edge_index(edge_index, num_vertices)
        return cls(h, int(num_vertices), int(edge_index.shape[1]))

    @property
    def device(self) -> torch.device:
        return self.handle.device

    @property
    def symmetric(self) -> bool:
        return self.handle.symmetric

    def aggregate(self, X, Y, **kw):
        return self.handle.spmm(X, Y, **kw)

    def new_like(self, X: torch.Tensor, cols: Optional[int] = None) -> torch.Tensor:
        return torch.empty((self.handle.num_rows, X.shape[1] if cols is None else cols),
                           dtype=X.dtype, device=X.device)


def _fingerprint(edge_index: torch.Tensor) -> Tuple[int, int, int]:
    """Order-independent content hash of the edge multiset (the graph does not depend on the order of the columns):
    three wrapping int64 sums, ONE transfer to the host."""
    r, c = edge_index[0], edge_index[1]
    k = r * 1000003 + c
    fp = torch.stack([k.sum(), (k * k).sum(), (r ^ (c * 8191)).sum()]).tolist()
    return int(fp[0]), int(fp[1]), int(fp[2])


def _same_edges(a: torch.Tensor, b: torch.Tensor, num_vertices: int) -> bool:
    """True iff the two [2, E] tensors hold the same edge MULTISET: elementwise equal (the usual case -- the same data
    uploaded again: one comparison kernel), or equal after sorting the (source, target) keys."""
    if a.shape != b.shape:
        return False
    if torch.equal(a, b):
        return True
    n = max(int(num_vertices), 1)
    return torch.equal(torch.sort(a[0] * n + a[1])[0], torch.sort(b[0] * n + b[1])[0])


class _Cache:
    """Level 1: the graph rides on the edge_index tensor object itself (valid while the object lives and its version
    counter is unchanged): no device work at all.  Level 2, for a tensor OBJECT not seen before: keyed by (device, shape,
    V, content fingerprint) and VERIFIED against a copy of the edges the cached graph was built from (a fingerprint
    collision must not hand back another graph's CSR) -- never by address, which the caching allocator may or may not hand out again -- so a caller
    that re-creates the device tensor on every forward (util/networks.py:65: ``data.edge_index.to(self.device)`` on
    CPU-resident data) pays three E-sized reductions and one host synchronisation per forward, but ``sg_graph_create``
    runs once per distinct graph.  ``compat.Data`` removes even that: its ``edge_index`` hands the SAME device tensor back
    from every ``.to(device)``, which then hits level 1."""

    def __init__(self, capacity: int = 16):
        self.capacity = capacity
        self._lvl2: Dict[tuple, tuple] = {}

    def get(self, edge_index: torch.Tensor, num_vertices: int) -> MeshGraph:
        hit = getattr(edge_index, _ATTR, None)
        if hit is not None:
            ver, nv, g = hit
            if ver == edge_index._version and nv == num_vertices:
                return g
        key = (str(edge_index.device), tuple(edge_index.shape), num_vertices, _fingerprint(edge_index))
        ent = self._lvl2.get(key)
        if ent is not None and not _same_edges(edge_index, ent[1], num_vertices):
            ent = None              # three equal 64-bit sums over DIFFERENT edges: a collision, not a hit
        if ent is None:
            g = MeshGraph.from_edge_index(edge_index, num_vertices)
            if len(self._lvl2) >= self.capacity:
                self._lvl2.pop(next(iter(self._lvl2)))
            # (the edges the graph was built from are kept beside it: a later fingerprint hit is verified against them)
            self._lvl2[key] = (g, edge_index.detach().clone())
        else:
            g = ent[0]
        try:
            setattr(edge_index, _ATTR, (edge_index._version, num_vertices, g))
        except Exception:  # pragma: no cover
            pass
        return g

    def clear(self):
        self._lvl2.clear()


_cache = _Cache()


def graph_for(edge_index: torch.Tensor, num_vertices: int) -> MeshGraph:
    """The (cached) MeshGraph of ``edge_index`` for ``num_vertices`` vertices."""
    return _cache.get(edge_index, num_vertices)


def clear_graph_cache():
    _cache.clear()
