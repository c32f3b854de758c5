"""Mesh connectivity and hole masks built on the device.

Host mirror of the pieces of the reference's ``Mesh`` / ``Datamaker`` that feed the
graph convolutions:

  ============================  =================================================
  here                          reference
  ============================  =================================================
  ``MeshTopology.edges``        ``Mesh.edges``       util/mesh.py:60-100
  ``MeshTopology.edge_index``   ``Mesh.edge_index``  util/mesh.py:229-230
  ``MeshTopology.f2f``          ``Mesh.f2f``         util/mesh.py:214-227
  ``make_dummy_mask``           util/datamaker.py:110-153
  ``vmask_to_fmask``            util/datamaker.py:156-159, util/meshnet.py:179,196
  ============================  =================================================

The reference does these with Python loops over faces and dense [V, V] matrices
(``Mesh.AdjI`` is built from ``torch.eye(V)``, util/mesh.py:267-274), which caps it at a
few 10 K vertices; here faces -> edges is a radix sort of half-edges and a mask ring is
a bitwise OR over CSR neighbours of bit-packed masks (semigcn_amd/csrc/mesh_prep.hip),
so the 1 M - 4 M vertex configurations can be prepared from a face list in milliseconds.
HIP device only, like the rest of the package.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from . import capi
from .graph import MeshGraph

# util/datamaker.py:118 -- seed density per ring count k (index = k)
P_LIST = (0.014, 0.014, 0.014, 0.014, 0.014, 0.014, 0.014, 0.0014, 0.014)


def _as_faces(faces, device) -> torch.Tensor:
    f = torch.as_tensor(np.asarray(faces) if not isinstance(faces, torch.Tensor) else faces)
    return f.to(device=device, dtype=torch.int64).contiguous()


class MeshTopology:
    """Connectivity of a triangle list, in the reference's layouts, as device tensors."""

    def __init__(self, faces, num_vertices: int, device="cuda", with_f2f: bool = True):
        device = torch.device(device)
        self.faces = _as_faces(faces, device)
        self.num_vertices = int(num_vertices)
        self.edges, self.f2f, self.manifold = capi.mesh_edges(self.faces, self.num_vertices, with_f2f)
        e = self.edges.t()
        self.edge_index = torch.cat([e, e.flip(0)], dim=1).contiguous()   # [edges.T | edges.T[[1,0]]]
        self._graph: Optional[MeshGraph] = None

    @property
    def device(self) -> torch.device:
        return self.faces.device

    @property
    def graph(self) -> MeshGraph:
        if self._graph is None:
            self._graph = MeshGraph.from_edge_index(self.edge_index, self.num_vertices)
        return self._graph

    def __len__(self):
        return self.num_vertices


def pack_bits(m: torch.Tensor) -> torch.Tensor:
    """[N, n] (non-zero = set) -> int64 [N, ceil(n/64)], bit b of word w = column 64 w + b."""
    n = m.shape[1]
    W = (n + 63) // 64
    b = (m != 0).to(torch.int64)
    if W * 64 != n:
        b = torch.cat([b, b.new_zeros(b.shape[0], W * 64 - n)], dim=1)
    sh = torch.arange(64, device=m.device, dtype=torch.int64)
    return (b.view(-1, W, 64) << sh).sum(dim=2)      # disjoint bits: the sum is the OR (bit 63 wraps to the sign)


def unpack_bits(bits: torch.Tensor, n: int) -> torch.Tensor:
    """int64 [N, W] -> bool [N, n]."""
    sh = torch.arange(64, device=bits.device, dtype=torch.int64)
    return (((bits.unsqueeze(2) >> sh) & 1) != 0).reshape(bits.shape[0], -1)[:, :n]


def dilate(topology_or_graph, seeds: torch.Tensor, rings: int) -> torch.Tensor:
    """``rings`` times ``M <- (AdjI @ M) > 0`` (util/datamaker.py:125-128) for all columns of
    ``seeds`` [V, n] at once; returns bool [V, n]."""
    g = topology_or_graph.graph if isinstance(topology_or_graph, MeshTopology) else topology_or_graph
    bits = pack_bits(seeds)
    for _ in range(int(rings)):
        bits = g.handle.dilate_bits(bits)
    return unpack_bits(bits, seeds.shape[1])


def vmask_to_fmask(mesh: MeshTopology, vmask) -> torch.Tensor:
    """A face is kept iff all three of its vertices are (util/datamaker.py:156-159).
    ``vmask`` [V] or [V, n] (non-zero = kept); returns bool [F] or [F, n]."""
    vm = torch.as_tensor(vmask).to(mesh.device)
    one = vm.dim() == 1
    vm = vm.reshape(vm.shape[0], -1)
    out = unpack_bits(capi.face_mask_bits(mesh.faces, pack_bits(vm)), vm.shape[1])
    return out.reshape(-1) if one else out


def make_dummy_mask(mesh: MeshTopology, dm_size: int = 40, kn: Sequence[int] = (3, 4, 5), exist_face=None,
                    p_list: Sequence[float] = P_LIST, rng=None):
    """Synthetic-hole masks (util/datamaker.py:110-153): for each ring count ``k`` in ``kn``,
    ``dm_size`` Bernoulli(p_list[k]) seed sets dilated ``k`` rings and complemented.

    Returns ``(vmask float32 [V, dm_size*len(kn)], fmask float32 [F, dm_size*len(kn)])``, 1 = kept.
    The seeds come from ``rng.binomial`` exactly as the reference draws them -- ``rng`` defaults
    to numpy's global generator, so ``np.random.seed(s)`` reproduces the reference's masks bit
    for bit; pass a ``torch.Generator`` on the mesh's device to draw them on the GPU instead.
    ``exist_face`` only colours the reference's debug PLY files and is ignored."""
    V = mesh.num_vertices
    cols = []
    for k in kn:
        p = float(np.float32(p_list[k]))     # the reference indexes a float32 tensor (util/datamaker.py:118,123)
        if isinstance(rng, torch.Generator):
            seeds = torch.rand((V, dm_size), device=mesh.device, generator=rng) < p
        else:
            src = np.random if rng is None else rng
            seeds = torch.from_numpy(src.binomial(1, p, size=[V, dm_size]).astype(np.uint8))
            seeds = seeds.to(mesh.device)
        cols.append(dilate(mesh, seeds, k))
    hole = torch.cat(cols, dim=1)
    kept = ~hole
    fmask = vmask_to_fmask(mesh, kept)
    return kept.float(), fmask.float()


# --------------------------------------------------------------------------------------
# Pooling hierarchy for MGCN at scale
# --------------------------------------------------------------------------------------
def contract_matching(edges: torch.Tensor, priority: torch.Tensor, num_vertices: int, target_v: int):
    """Choose ``num_vertices - target_v`` vertex-disjoint edges, lowest ``priority`` first, by
    rounds of locally-dominant matching (an edge is taken when it is the best remaining edge
    at BOTH of its ends), and contract them.  Returns ``coarse_of`` int64 [V] (cluster id of
    every vertex; ids ascend with the smallest member, as np.unique numbers them) and V_coarse.
    Runs on the tensors' device; every round is a handful of scatter-min / gather passes."""
    dev = edges.device
    need = int(num_vertices) - int(target_v)
    rank = torch.empty_like(priority, dtype=torch.int64)
    rank[torch.argsort(priority, stable=True)] = torch.arange(priority.numel(), device=dev)
    cand = torch.arange(edges.shape[0], device=dev)
    used = torch.zeros(num_vertices, dtype=torch.bool, device=dev)
    parent = torch.arange(num_vertices, device=dev)
    big = torch.iinfo(torch.int64).max
    while need > 0 and cand.numel():
        a, b, r = edges[cand, 0], edges[cand, 1], rank[cand]
        best = torch.full((num_vertices,), big, dtype=torch.int64, device=dev)
        best.scatter_reduce_(0, a, r, "amin")
        best.scatter_reduce_(0, b, r, "amin")
        win = (best[a] == r) & (best[b] == r)
        w = cand[win]
        if w.numel() == 0:
            break
        if w.numel() > need:
            w = w[torch.argsort(rank[w])[:need]]
        lo, hi = edges[w, 0], edges[w, 1]
        parent[hi] = lo
        used[lo] = True
        used[hi] = True
        need -= int(w.numel())
        cand = cand[~win]
        cand = cand[~(used[edges[cand, 0]] | used[edges[cand, 1]])]
    is_root = parent == torch.arange(num_vertices, device=dev)
    root_id = torch.cumsum(is_root.to(torch.int64), 0) - 1
    return root_id[parent], int(is_root.sum())


class DeviceMesh:
    """What ``MGCN(device, smo_mesh, ini_mesh, v_mask)`` reads from the reference's ``Mesh``
    (util/meshnet.py:170-199: ``.vs .faces .edge_index .path .simplification(target_v)`` and, on
    the result, ``.pool_hash``), held on the device so that a 1 M-vertex hierarchy is built in
    milliseconds.  ``simplification`` is NOT the reference's QEM edge collapse
    (util/mesh.py:394-482: Python heap loop, minutes at 50 K vertices): it contracts a
    shortest-edge-first matching, which yields the same artefacts -- ``pool_hash`` rows
    (fine_i, coarse_i) covering every fine vertex with clusters of one or two, positions at the
    cluster mean, the surviving triangles, and the quotient graph as ``edge_index``."""

    def __init__(self, vs, faces, device="cuda", path: str = "./mesh.obj", pool_hash=None, edge_index=None):
        device = torch.device(device)
        self.vs = torch.as_tensor(vs).to(device=device, dtype=torch.float32)
        self.topology = MeshTopology(faces, self.vs.shape[0], device, with_f2f=False)
        self.faces = self.topology.faces
        self.edge_index = self.topology.edge_index if edge_index is None else edge_index
        self.pool_hash = pool_hash
        self.path = path

    def simplification(self, target_v: int) -> "DeviceMesh":
        V = self.vs.shape[0]
        half = self.edge_index.shape[1] // 2
        und = self.edge_index[:, :half].t().contiguous()
        d = self.vs[und[:, 0]] - self.vs[und[:, 1]]
        coarse_of, Vc = contract_matching(und, (d * d).sum(1), V, target_v)
        fine = torch.arange(V, device=self.vs.device)
        pool = capi.PoolHandle(fine, coarse_of, V, Vc)
        cvs = pool.pool_mean(self.vs.contiguous())
        cf = coarse_of[self.faces]
        alive = (cf[:, 0] != cf[:, 1]) & (cf[:, 1] != cf[:, 2]) & (cf[:, 2] != cf[:, 0])
        ce = coarse_of[und]
        ce = ce[ce[:, 0] != ce[:, 1]]
        key = torch.unique(torch.minimum(ce[:, 0], ce[:, 1]) * Vc + torch.maximum(ce[:, 0], ce[:, 1]))
        e = torch.stack([key // Vc, key % Vc])
        out = DeviceMesh(cvs, cf[alive], self.vs.device, self.path,
                         pool_hash=torch.stack([fine, coarse_of], 1).cpu().numpy(),
                         edge_index=torch.cat([e, e.flip(0)], dim=1).contiguous())
        return out
