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


#: util/mesh.py:10-11
OPTIM_VALENCE, VALENCE_WEIGHT = 6, 1


def _vertex_quadrics(vs: torch.Tensor, faces: torch.Tensor) -> torch.Tensor:
    """Q_v = sum over the faces around v of [n | d]^T [n | d] (unit normal n, d = -n . centroid): util/mesh.py:397-406,
    as [V, 4, 4] float64 on the device (an index_add over the 3 F face corners instead of a Python loop over V)."""
    p = vs.double()
    a, b, c = p[faces[:, 0]], p[faces[:, 1]], p[faces[:, 2]]
    n = torch.linalg.cross(b - a, c - a, dim=1)
    n = n / n.norm(dim=1, keepdim=True).clamp_min(1e-30)
    d = -(n * ((a + b + c) / 3.0)).sum(1, keepdim=True)
    abcd = torch.cat([n, d], dim=1)                                   # [F, 4]
    q = abcd.unsqueeze(2) * abcd.unsqueeze(1)                         # [F, 4, 4]
    Q = torch.zeros((vs.shape[0], 4, 4), dtype=torch.float64, device=vs.device)
    for k in range(3):
        Q.index_add_(0, faces[:, k], q)
    return Q


def qem_contract(vs: torch.Tensor, faces: torch.Tensor, target_v: int, max_cluster: int = 5, rounds: int = 8):
    """The reference's simplification criterion (util/mesh.py:394-482: quadric error of the edge MIDPOINT under the sum of
    the two end vertices' quadrics, times the valence penalty |valence_new - 6| + 1, x 1e5 for a valence of 3), applied
    in parallel: instead of one heap-ordered collapse at a time, ``rounds`` rounds of a locally-dominant matching on the
    CURRENT (already contracted) mesh -- an edge is collapsed when it is the cheapest remaining edge at both of its ends
    -- each round taking its share of the vertices still to remove.  Like the reference, the surviving vertex (the
    lower id) moves to the midpoint and KEEPS its own quadric (util/mesh.py:564-571), a vertex that has been merged
    into can be merged again in a later round (clusters of 1 .. ``max_cluster``; the reference's reach 5), and an edge
    with other than two faces, or whose ends share other than two neighbours, is not collapsed (util/mesh.py:455-464: the
    boundary test and the link condition -- it would fold the surface).  Within a round no edge on the one-ring of a
    collapse is taken: everything a round uses (costs, valences, link tests) is computed on the mesh the round began with.
    Returns ``coarse_of`` int64 [V] (ids ascend with the cluster's smallest member) and V_coarse."""
    dev = vs.device
    V = vs.shape[0]
    need = V - int(target_v)
    Q = _vertex_quadrics(vs, faces)
    pos = vs.double().clone()
    rep = torch.arange(V, device=dev)                 # representative (surviving vertex) of every original vertex
    size = torch.ones(V, dtype=torch.int64, device=dev)
    f = faces.clone()
    big = torch.iinfo(torch.int64).max
    for r in range(rounds):
        if need <= 0:
            break
        # current mesh: faces over representatives, degenerate ones dropped
        f = rep[f]
        f = f[(f[:, 0] != f[:, 1]) & (f[:, 1] != f[:, 2]) & (f[:, 2] != f[:, 0])]
        he = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], dim=0)
        lo, hi = torch.minimum(he[:, 0], he[:, 1]), torch.maximum(he[:, 0], he[:, 1])
        key, cnt = torch.unique(lo * V + hi, return_counts=True)
        ea, eb = key // V, key % V                      # undirected edges, ea < eb
        nf = torch.bincount(f.reshape(-1), minlength=V)                     # faces around a vertex (= its valence)
        # the reference's two tests (util/mesh.py:455-464): two faces on the edge (not a boundary), and the LINK condition --
        # the two ends share exactly two neighbours.  An edge next to a valence-3 vertex has two faces but three shared
        # neighbours; collapsing it folds the surface onto itself (duplicate / opposite faces in the coarse level).
        nb_key = torch.cat([key, eb * V + ea])                              # directed adjacency (u, n) as u * V + n, ...
        nb_key = torch.sort(nb_key)[0]                                      # ... sorted: u's neighbours are one run
        deg = torch.bincount(nb_key // V, minlength=V)
        start = torch.cumsum(deg, 0) - deg
        e_id = torch.repeat_interleave(torch.arange(key.numel(), device=dev), deg[ea])      # every (edge, neighbour of ea)
        n_of_a = nb_key[start[ea][e_id] + (torch.arange(e_id.numel(), device=dev) - torch.repeat_interleave(
            torch.cumsum(deg[ea], 0) - deg[ea], deg[ea]))] % V
        probe = eb[e_id] * V + n_of_a                                       # is that neighbour adjacent to eb as well?
        at = torch.searchsorted(nb_key, probe).clamp_(max=nb_key.numel() - 1)
        shared = torch.bincount(e_id[nb_key[at] == probe], minlength=key.numel())
        ok = (cnt == 2) & (shared == 2) & (size[ea] + size[eb] <= max_cluster)
        val_new = nf[ea] + nf[eb] - 4
        pen = (val_new - OPTIM_VALENCE).abs().double() * VALENCE_WEIGHT + 1.0
        pen = torch.where(val_new == 3, pen * 100000.0, pen)
        mid = 0.5 * (pos[ea] + pos[eb])
        v4 = torch.cat([mid, torch.ones((mid.shape[0], 1), dtype=torch.float64, device=dev)], dim=1)
        Qs = Q[ea] + Q[eb]
        err = torch.einsum("ei,eij,ej->e", v4, Qs, v4) * pen
        err = torch.where(ok, err, torch.full_like(err, float("inf")))
        # a round's quota: its share of what is left (the last round takes whatever it can)
        quota = need if r == rounds - 1 else -(-need // (rounds - r))
        rank = torch.empty_like(key)
        rank[torch.argsort(err, stable=True)] = torch.arange(key.numel(), device=dev)
        cand = torch.nonzero(ok, as_tuple=False).reshape(-1)
        used = torch.zeros(V, dtype=torch.bool, device=dev)
        taken = 0
        while taken < quota and cand.numel():
            a, b, rk = ea[cand], eb[cand], rank[cand]
            best = torch.full((V,), big, dtype=torch.int64, device=dev)
            best.scatter_reduce_(0, a, rk, "amin")
            best.scatter_reduce_(0, b, rk, "amin")
            win = (best[a] == rk) & (best[b] == rk)
            w = cand[win]
            if w.numel() == 0:
                break
            if w.numel() > quota - taken:
                w = w[torch.argsort(rank[w])[: quota - taken]]
            a, b = ea[w], eb[w]
            # the two triangles on a collapsing edge vanish; their third vertices must not collapse in the same round
            # towards each other through this pair -- vertex-disjointness of the matching already guarantees a, b are fresh
            pos[a] = 0.5 * (pos[a] + pos[b])
            size[a] = size[a] + size[b]
            rep_b = torch.arange(V, device=dev)
            rep_b[b] = a
            rep = rep_b[rep]
            used[a] = True
            used[b] = True
            # costs, valences and link tests of this round were taken on the mesh as it was when the round began: an edge
            # with an end on the one-ring of a collapse no longer has the valence / shared neighbours it was priced with
            # -- it waits for the next round, which prices everything again
            ring = nb_key[torch.repeat_interleave(start[torch.cat([a, b])], deg[torch.cat([a, b])]) + (
                torch.arange(int(deg[torch.cat([a, b])].sum()), device=dev) - torch.repeat_interleave(
                    torch.cumsum(deg[torch.cat([a, b])], 0) - deg[torch.cat([a, b])], deg[torch.cat([a, b])]))] % V
            used[ring] = True
            taken += int(w.numel())
            cand = cand[~win]
            cand = cand[~(used[ea[cand]] | used[eb[cand]])]
        need -= taken
        if taken == 0:
            break
    is_root = rep == torch.arange(V, device=dev)
    root_id = torch.cumsum(is_root.to(torch.int64), 0) - 1
    return root_id[rep], int(is_root.sum()), pos[is_root].float()


class DeviceMesh:
    """What ``MGCN(device, smo_mesh, ini_mesh, v_mask)`` reads from the reference's ``Mesh``
    (util/meshnet.py:170-199: ``.vs .faces .edge_index .path .simplification(target_v)`` and, on
    the result, ``.pool_hash``), held on the device so that a 1 M-vertex hierarchy is built in
    a fraction of a second.  ``simplification`` applies the reference's criterion -- quadric error of the edge midpoint
    with its valence penalty (util/mesh.py:394-482) -- as rounds of a parallel matching instead of one heap-ordered
    collapse at a time (``qem_contract``; ``criterion="length"`` is the shortest-edge matching of rounds 1-2: clusters of
    one or two).  The artefacts are the reference's: ``pool_hash`` rows (fine_i, coarse_i) covering every fine vertex
    (clusters of 1 .. 5), positions (midpoints of the collapses, as util/mesh.py:564), the surviving triangles, and the
    quotient graph as ``edge_index``.  The collapse ORDER differs from a sequential heap, so the clusters are not the
    reference's clusters; tests/test_gpu_parity.py compares the two hierarchies by what they are for: geometric error of
    the coarse meshes, and the loss an MGCN trained on either reaches."""

    def __init__(self, vs, faces, device="cuda", path: str = "./mesh.obj", pool_hash=None, edge_index=None):
        device = torch.device(device)
        self.vs = torch.as_tensor(vs).to(device=device, dtype=torch.float32)
        self.topology = MeshTopology(faces, self.vs.shape[0], device, with_f2f=False)
        self.faces = self.topology.faces
        self.edge_index = self.topology.edge_index if edge_index is None else edge_index
        self.pool_hash = pool_hash
        self.path = path

    def simplification(self, target_v: int, criterion: str = "qem") -> "DeviceMesh":
        V = self.vs.shape[0]
        half = self.edge_index.shape[1] // 2
        und = self.edge_index[:, :half].t().contiguous()
        fine = torch.arange(V, device=self.vs.device)
        if criterion == "qem":
            coarse_of, Vc, cvs = qem_contract(self.vs, self.faces, target_v)
        elif criterion == "length":
            d = self.vs[und[:, 0]] - self.vs[und[:, 1]]
            coarse_of, Vc = contract_matching(und, (d * d).sum(1), V, target_v)
            cvs = capi.PoolHandle(fine, coarse_of, V, Vc).pool_mean(self.vs.contiguous())
        else:
            raise ValueError("criterion must be 'qem' or 'length'")
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
