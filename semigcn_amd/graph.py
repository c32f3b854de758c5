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


def _fingerprint(edge_index: torch.Tensor) -> Tuple[int, int]:
    r, c = edge_index[0], edge_index[1]
    a = (r * 1000003 + c).sum()
    b = (r ^ (c * 8191)).sum()
    ab = torch.stack([a, b]).tolist()
    return int(ab[0]), int(ab[1])


class _Cache:
    """Level 1: the graph rides on the edge_index tensor object itself (valid while
    the object lives and its version counter is unchanged).  Level 2: keyed by
    (device, data_ptr, E, V) and verified by a content fingerprint, for callers that
    re-create the device tensor every forward (util/networks.py:65 does
    ``data.edge_index.to(device)`` per call)."""

    def __init__(self, capacity: int = 16):
        self.capacity = capacity
        self._lvl2: Dict[tuple, Tuple[Tuple[int, int], MeshGraph]] = {}

    def get(self, edge_index: torch.Tensor, num_vertices: int) -> MeshGraph:
        hit = getattr(edge_index, _ATTR, None)
        if hit is not None:
            ver, nv, g = hit
            if ver == edge_index._version and nv == num_vertices:
                return g
        key = (str(edge_index.device), edge_index.data_ptr(), tuple(edge_index.shape), num_vertices)
        fp = _fingerprint(edge_index)
        ent = self._lvl2.get(key)
        if ent is not None and ent[0] == fp:
            g = ent[1]
        else:
            g = MeshGraph.from_edge_index(edge_index, num_vertices)
            if len(self._lvl2) >= self.capacity:
                self._lvl2.pop(next(iter(self._lvl2)))
            self._lvl2[key] = (fp, g)
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
