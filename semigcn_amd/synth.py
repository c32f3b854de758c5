"""Synthetic closed manifold meshes with the reference's input layouts.

The reference's sample data is absent (/root/reference/.MISSING_LARGE_BLOBS:1)
and its ``Mesh`` class builds dense V x V matrices (util/mesh.py:267-274), so
meshes of benchmark size are generated here instead:
a closed ``nu x nv`` torus triangulation (V = nu*nv, F = 2V, E_directed = 6V
exactly) whose quad diagonals are randomly flipped so the valence spreads over
4..8 like an isotropically remeshed scan, with unit mean edge length
(preprocess/prepare.py:48-52 scales real inputs the same way).

Layouts follow the reference exactly:
  * ``edge_index`` int64 [2, E]: unique (lo, hi) pairs in face-discovery order,
    then the mirrored (hi, lo) block (util/mesh.py:60-100,229-230;
    util/datamaker.py:76-77),
  * ``z1 = initial - smooth`` float32 [V, 3], ``x_pos = smooth`` (util/datamaker.py:70-73),
  * dummy masks: Bernoulli(p) seeds dilated ``k`` rings through (I + A), complemented
    (util/datamaker.py:110-136).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np


@dataclass
class SynthMesh:
    nu: int
    nv: int
    vs: np.ndarray            # [V,3] float64 "initial" positions
    faces: np.ndarray         # [F,3] int64
    edges: np.ndarray         # [E/2,2] int64, lo<hi, face-discovery order
    edge_index: np.ndarray    # [2,E] int64 reference layout
    z1: np.ndarray            # [V,3] float32 displacement initial - smooth
    x_pos: np.ndarray         # [V,3] float32 smooth positions
    v_mask: np.ndarray        # [V] bool, True = original (kept) vertex
    perm: Optional[np.ndarray] = None   # new_id = perm[old_id] if permuted
    extra: dict = field(default_factory=dict)

    @property
    def num_vertices(self) -> int:
        return self.vs.shape[0]

    @property
    def num_edges(self) -> int:
        return self.edge_index.shape[1]


def _torus_faces(nu: int, nv: int, flip_frac: float, rng: np.random.Generator) -> np.ndarray:
    u, v = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    u, v = u.ravel(), v.ravel()
    up, vp = (u + 1) % nu, (v + 1) % nv
    a = u * nv + v
    b = up * nv + v
    c = up * nv + vp
    d = u * nv + vp
    flip = rng.random(a.shape[0]) < flip_frac
    # default diagonal a-c: (a,b,c),(a,c,d); flipped diagonal b-d: (a,b,d),(b,c,d)
    t0 = np.where(flip[:, None], np.stack([a, b, d], 1), np.stack([a, b, c], 1))
    t1 = np.where(flip[:, None], np.stack([b, c, d], 1), np.stack([a, c, d], 1))
    faces = np.empty((2 * a.shape[0], 3), dtype=np.int64)
    faces[0::2] = t0
    faces[1::2] = t1
    return faces


def random_edge_flips(faces: np.ndarray, vs: Optional[np.ndarray], n_flips: int, rng: np.random.Generator,
                      vmin: int = 4, vmax: int = 9, max_rounds: int = 60) -> np.ndarray:
    """``n_flips`` random VALID edge flips on a closed manifold triangle mesh ("0.15 * E_und
    random valid edge flips; keeps V, E; valence 4..9").  A flip replaces the edge a-b shared by the triangles (a,b,c)
    and (b,a,d) by c-d: triangles (a,d,c), (d,b,c).  Valid: a and b keep valence >= ``vmin``, c and d stay <= ``vmax``,
    c-d is not an edge yet, and (with positions ``vs``) neither new triangle folds over (its normal keeps the side of
    the two old ones).  Vectorised in rounds: a random subset of the edges is drawn, every drawn edge claims its four
    vertices with a random priority, and the edges that hold all four of their claims (no other flip of the round
    touches those vertices) are flipped together; rounds repeat until ``n_flips`` flips are done."""
    faces = np.ascontiguousarray(faces, dtype=np.int64).copy()
    V = int(faces.max()) + 1
    done = 0
    for _ in range(max_rounds):
        need = n_flips - done
        if need <= 0:
            break
        F = faces.shape[0]
        src, dst, opp = faces.ravel(), faces[:, [1, 2, 0]].ravel(), faces[:, [2, 0, 1]].ravel()
        he_face = np.repeat(np.arange(F), 3)
        key = np.minimum(src, dst) * np.int64(V) + np.maximum(src, dst)
        order = np.argsort(key, kind="stable")
        ks = key[order]
        if not (ks.shape[0] % 2 == 0 and np.array_equal(ks[0::2], ks[1::2]) and (ks.shape[0] < 4 or np.all(ks[2::2] > ks[0:-2:2]))):
            raise ValueError("random_edge_flips needs a closed manifold mesh (every edge in exactly two triangles)")
        h1, h2 = order[0::2], order[1::2]
        ekeys = ks[0::2]                                   # sorted unique undirected edge keys
        val = np.bincount(src, minlength=V)
        n_e = h1.shape[0]
        cand = rng.permutation(n_e)[: min(n_e, max(4 * need, 1024))]
        a, b, c, d = src[h1[cand]], dst[h1[cand]], opp[h1[cand]], opp[h2[cand]]
        ok = (val[a] - 1 >= vmin) & (val[b] - 1 >= vmin) & (val[c] + 1 <= vmax) & (val[d] + 1 <= vmax) & (c != d)
        nk = np.minimum(c, d) * np.int64(V) + np.maximum(c, d)
        pos = np.searchsorted(ekeys, nk)
        ok &= ~((pos < n_e) & (ekeys[np.minimum(pos, n_e - 1)] == nk))
        if vs is not None:
            pa, pb, pc, pd = vs[a], vs[b], vs[c], vs[d]
            n_old = np.cross(pb - pa, pc - pa) + np.cross(pa - pb, pd - pb)
            n1, n2 = np.cross(pd - pa, pc - pa), np.cross(pb - pd, pc - pd)
            ok &= ((n1 * n_old).sum(1) > 0) & ((n2 * n_old).sum(1) > 0)
        cand, a, b, c, d = cand[ok], a[ok], b[ok], c[ok], d[ok]
        if cand.shape[0] == 0:
            continue
        prio = rng.permutation(cand.shape[0]).astype(np.int64) + 1
        claim = np.zeros(V, np.int64)
        for v in (a, b, c, d):
            np.maximum.at(claim, v, prio)
        win = (claim[a] == prio) & (claim[b] == prio) & (claim[c] == prio) & (claim[d] == prio)
        w = np.flatnonzero(win)[:need]
        if w.shape[0] == 0:
            continue
        f1, f2 = he_face[h1[cand[w]]], he_face[h2[cand[w]]]
        faces[f1] = np.stack([a[w], d[w], c[w]], 1)
        faces[f2] = np.stack([d[w], b[w], c[w]], 1)
        done += w.shape[0]
    if done < n_flips:
        raise RuntimeError(f"only {done} of {n_flips} valid edge flips found")
    return faces


def edges_from_faces(faces: np.ndarray, num_vertices: int) -> np.ndarray:
    """Unique undirected edges (lo, hi) in the order a face-by-face scan first
    meets them -- the order util/mesh.py:60-100 (``build_gemm``) produces."""
    f = faces.astype(np.int64)
    pairs = np.stack([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=1).reshape(-1, 2)
    lo = pairs.min(1)
    hi = pairs.max(1)
    key = lo * np.int64(num_vertices) + hi
    _, first = np.unique(key, return_index=True)
    first.sort()
    return np.stack([lo[first], hi[first]], axis=1)


def edge_index_from_edges(edges: np.ndarray) -> np.ndarray:
    """[edges.T | edges.T[[1,0]]]  (util/mesh.py:229-230)."""
    e = edges.T
    return np.ascontiguousarray(np.concatenate([e, e[[1, 0]]], axis=1))


def adjacency_plus_identity(edge_index: np.ndarray, num_vertices: int):
    import scipy.sparse as sp

    E = edge_index.shape[1]
    A = sp.coo_matrix((np.ones(E, np.float32), (edge_index[1], edge_index[0])),
                      shape=(num_vertices, num_vertices)).tocsr()
    return A + sp.identity(num_vertices, dtype=np.float32, format="csr")


def make_dummy_masks(edge_index: np.ndarray, num_vertices: int, dm_size: int = 40,
                     k: int = 4, p: float = 0.014, seed: int = 317) -> np.ndarray:
    """float32 [V, dm_size]; 0 inside synthetic holes, 1 elsewhere."""
    rng = np.random.default_rng(seed)
    AI = adjacency_plus_identity(edge_index, num_vertices)
    M = (rng.random((num_vertices, dm_size)) < p).astype(np.float32)
    for _ in range(k):
        M = (AI @ M > 0).astype(np.float32)
    return 1.0 - M


def make_v_mask(edge_index: np.ndarray, num_vertices: int, n_holes: int = 5,
                frac: float = 0.05, seed: int = 316) -> np.ndarray:
    """True except ``n_holes`` graph-geodesic discs totalling about ``frac`` of V."""
    rng = np.random.default_rng(seed)
    AI = adjacency_plus_identity(edge_index, num_vertices)
    target = frac * num_vertices / n_holes
    rings = max(1, int(round((np.sqrt(12 * target - 3) - 3) / 6)))  # 1+3r(r+1) vertices per disc
    hole = np.zeros((num_vertices, 1), np.float32)
    hole[rng.choice(num_vertices, n_holes, replace=False)] = 1
    for _ in range(rings):
        hole = (AI @ hole > 0).astype(np.float32)
    return hole[:, 0] == 0


def torus_mesh(nu: int, nv: int, *, flip_frac: float = 0.45, jitter: float = 0.05,
               permute: bool = False, seed: int = 314, masks: bool = True, edge_flips: Optional[float] = None) -> SynthMesh:
    """Closed torus triangulation with V = nu*nv vertices (nu, nv >= 4).  Irregularity: by default every quad's
    diagonal is flipped independently with probability ``flip_frac`` (valence 4..8); ``edge_flips = 0.15`` is SURVEY.md
    section 8(d)'s recipe instead -- the regular triangulation followed by ``edge_flips * E_und`` random valid edge
    flips (``random_edge_flips``: valence 4..9), which bench.py uses."""
    assert nu >= 4 and nv >= 4
    rng = np.random.default_rng(seed)
    faces = _torus_faces(nu, nv, 0.0 if edge_flips is not None else flip_frac, rng)
    V = nu * nv
    u, v = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    th = 2 * np.pi * u.ravel() / nu
    ph = 2 * np.pi * v.ravel() / nv
    R = max(nu / (2 * np.pi), 2.0 * nv / (2 * np.pi))
    r = nv / (2 * np.pi)
    vs = np.stack([(R + r * np.cos(ph)) * np.cos(th), (R + r * np.cos(ph)) * np.sin(th), r * np.sin(ph)], 1)
    vs = vs + rng.normal(0.0, jitter, vs.shape)
    if edge_flips is not None:
        faces = random_edge_flips(faces, vs, int(round(edge_flips * 3 * V)), rng)
    perm = None
    if permute:
        perm = np.random.default_rng(seed + 4).permutation(V)
        inv = np.empty_like(perm)
        inv[perm] = np.arange(V)
        faces = perm[faces]
        vs = vs[inv]
    edges = edges_from_faces(faces, V)
    edge_index = edge_index_from_edges(edges)
    z1 = np.random.default_rng(seed + 1).normal(0.0, 0.05, (V, 3)).astype(np.float32)
    x_pos = (vs - z1).astype(np.float32)
    v_mask = make_v_mask(edge_index, V, seed=seed + 2) if masks else np.ones(V, bool)
    return SynthMesh(nu, nv, vs, faces, edges, edge_index, z1, x_pos, v_mask, perm)


def octahedron_sphere(level: int, seed: int = 314) -> SynthMesh:
    """``level`` x subdivided octahedron projected to the sphere (V = 4^level*4+2);
    a second mesh family with valence-4 poles for the parity fixtures."""
    vs = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], float)
    faces = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4],
                      [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], np.int64)
    for _ in range(level):
        V = vs.shape[0]
        e = np.sort(np.stack([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 1).reshape(-1, 2), 1)
        key, inv = np.unique(e[:, 0] * V + e[:, 1], return_inverse=True)
        mid = 0.5 * (vs[key // V] + vs[key % V])
        m = V + inv.reshape(-1, 3)
        vs = np.concatenate([vs, mid], 0)
        a, b, c = faces[:, 0], faces[:, 1], faces[:, 2]
        ab, bc, ca = m[:, 0], m[:, 1], m[:, 2]
        faces = np.concatenate([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1),
                                np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)], 0)
    vs = vs / np.linalg.norm(vs, axis=1, keepdims=True)
    mean_len = np.mean(np.linalg.norm(vs[faces[:, 0]] - vs[faces[:, 1]], axis=1))
    vs = vs / mean_len
    V = vs.shape[0]
    edges = edges_from_faces(faces, V)
    edge_index = edge_index_from_edges(edges)
    z1 = np.random.default_rng(seed + 1).normal(0.0, 0.05, (V, 3)).astype(np.float32)
    x_pos = (vs - z1).astype(np.float32)
    v_mask = make_v_mask(edge_index, V, n_holes=2, seed=seed + 2)
    return SynthMesh(0, 0, vs, faces, edges, edge_index, z1, x_pos, v_mask)


def write_obj(path: str, vs: np.ndarray, faces: np.ndarray) -> None:
    """Minimal OBJ in the dialect util/mesh.py:35-58 parses (1-based ``f a b c``)."""
    with open(path, "w") as f:
        for p in vs:
            f.write("v %.9g %.9g %.9g\n" % (p[0], p[1], p[2]))
        for t in faces:
            f.write("f %d %d %d\n" % (t[0] + 1, t[1] + 1, t[2] + 1))


def greedy_pool_hierarchy(edge_index: np.ndarray, num_vertices: int, ratio: float = 0.6,
                          seed: int = 319):
    """A cheap stand-in for the reference's QEM simplification
    (util/mesh.py:394-482, out of scope: Python heap loops, minutes at 50 K) that
    yields the same *artefacts* MGCN consumes: ``pool_hash`` = sorted
    (fine_i, coarse_i) pairs covering every fine vertex (util/mesh.py:653-676)
    and the coarse ``edge_index``.  Collapses a random maximal matching of edges
    until ``int(V*ratio)`` vertices remain; the coarse graph is the quotient graph.
    Clusters have 1..2 vertices per round (the reference's QEM gives 1..5)."""
    rng = np.random.default_rng(seed)
    target = int(num_vertices * ratio)
    half = edge_index.shape[1] // 2
    und = edge_index[:, :half].T
    order = rng.permutation(und.shape[0])
    cluster = np.arange(num_vertices)
    used = np.zeros(num_vertices, bool)
    need = num_vertices - target
    # vectorised greedy rounds: an edge is taken if both ends are free and it is
    # the first (in random order) to claim each of them
    taken = 0
    cand = und[order]
    while taken < need and cand.shape[0]:
        free = ~used[cand[:, 0]] & ~used[cand[:, 1]]
        cand = cand[free]
        if cand.shape[0] == 0:
            break
        idx = np.arange(cand.shape[0])
        first0 = np.full(num_vertices, cand.shape[0], np.int64)
        np.minimum.at(first0, cand[:, 0], idx)
        np.minimum.at(first0, cand[:, 1], idx)
        win = (first0[cand[:, 0]] == idx) & (first0[cand[:, 1]] == idx)
        w = cand[win][: need - taken]
        cluster[w[:, 1]] = w[:, 0]
        used[w[:, 0]] = True
        used[w[:, 1]] = True
        taken += w.shape[0]
        cand = cand[~win]
    roots, coarse_of = np.unique(cluster, return_inverse=True)
    Vc = roots.shape[0]
    pool_hash = np.stack([np.arange(num_vertices), coarse_of], 1).astype(np.int64)
    ce = coarse_of[edge_index]
    ce = ce[:, ce[0] != ce[1]]
    lo, hi = ce.min(0), ce.max(0)
    key = np.unique(lo * np.int64(Vc) + hi)
    cedges = np.stack([key // Vc, key % Vc], 1)
    return pool_hash, edge_index_from_edges(cedges), Vc
