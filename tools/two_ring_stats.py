#!/usr/bin/env python3
"""Two-ring tile records of the benchmark mesh: how large the rings of a row tile are (the cost side of a fused two-hop
aggregation), with the record builder such a kernel would read and a self-check of what the records mean.

    python tools/two_ring_stats.py [--mesh 1000x1000] [--json out.json]        (CPU; seconds)

A ChebConv pass is two dependent aggregations ([3P] ChebConv.forward: ``Tx1 = L^ x``, ``Tx2 = 2 L^ Tx1 - x``; reached
from util/networks.py:42 -- and, unwound, the same pair in its backward pass).  For rows of 32 .. 128 bytes both hops of
a TILE of 64 consecutive rows fit into the LDS of one workgroup: stage the x rows of the tile's two-ring once, compute the
inner hop for the tile's ring-1 rows into LDS (rounded as the stored tensor would be), the outer hop for the tile's own
rows from there.  ``build`` lists, per tile, what that needs -- plain ``torch`` index arithmetic from a CSR.  NOT part of the
product: measured on the benchmark mesh the rings are too large for the fusion to pay (see ``main``).

Per tile t (rows ``[t T, min((t + 1) T, V))``, T = 64), in LOCAL numbering:

    local ids 0 .. n_own - 1         the tile's own rows, in row order
              n_own .. n1 - 1        the other ring-1 rows (neighbours of own rows), ascending global id
              n1 .. n2 - 1           the other ring-2 rows (neighbours of those), ascending global id

    ids2[off2[t] + l]                global row id of local id l                                   (l < n2 <= 255)
    eptr[off1[t] + r .. + r + 1]     the edges of ring-1 row r (r < n1), in CSR order (= ascending global source id: the
                                     summation order of the sequential kernels, so the fused pass is bit-identical)
    eidx[e]                          local id of the edge's source (uint8)

A neighbour of an own row is a ring-1 row, so the same edge lists serve the outer hop (rows 0 .. n_own - 1, sources < n1).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

TILE_ROWS = 64
#: limits a kernel with uint8 local ids and one workgroup's LDS would have; a graph with a tile beyond them gets no records
MAX_RING2, MAX_RING1, MAX_EDGES = 255, 192, 1792


@dataclass
class TwoRing:
    n_tiles: int
    off2: torch.Tensor      # int32 [n_tiles + 1]
    off1: torch.Tensor      # int32 [n_tiles + 1]
    ids2: torch.Tensor      # int32 [off2[-1]]
    eptr: torch.Tensor      # int32 [off1[-1] + 1]
    eidx: torch.Tensor      # uint8 [eptr[-1]]
    max_n1: int
    max_n2: int
    max_edges: int


def _segment_rank(seg_of: torch.Tensor, n_seg: int):
    """For a sorted segment id per element: (rank of each element inside its segment, elements per segment, segment starts)."""
    cnt = torch.bincount(seg_of, minlength=n_seg)
    start = torch.cumsum(cnt, 0) - cnt
    return torch.arange(seg_of.numel(), device=seg_of.device) - start[seg_of], cnt, start


def build(rowptr: torch.Tensor, colidx: torch.Tensor, num_rows: int, tile: int = TILE_ROWS) -> Optional[TwoRing]:
    """Records of a square CSR (rows = targets, ``colidx`` ascending inside a row).  None when a tile does not fit."""
    dev = rowptr.device
    V, T = int(num_rows), int(tile)
    if V == 0:
        return None
    rp, ci = rowptr.long(), colidx.long()
    nt = (V + T - 1) // T
    deg = rp[1:] - rp[:-1]
    ar = lambda n: torch.arange(n, device=dev)
    tile_of_edge = torch.repeat_interleave(ar(V), deg) // T
    own_cnt = (V - ar(nt) * T).clamp(max=T)
    # ring 1 beside the own rows: (tile, neighbour) pairs, unique, ascending by tile then vertex
    far = (ci // T) != tile_of_edge
    key1 = torch.unique(tile_of_edge[far] * V + ci[far])
    t1, v1 = key1 // V, key1 % V
    rank1, cnt1, start1 = _segment_rank(t1, nt)
    n1 = own_cnt + cnt1
    # ring 2 beside ring 1: neighbours of those rows that are neither own rows nor ring-1 rows of the tile
    d1 = deg[v1]
    owner = torch.repeat_interleave(ar(v1.numel()), d1)
    e_of = rp[v1][owner] + (ar(int(d1.sum())) - (torch.cumsum(d1, 0) - d1)[owner])
    w, tw = ci[e_of], t1[owner]
    key2 = torch.unique((tw * V + w)[(w // T) != tw])
    if key1.numel():
        p = torch.searchsorted(key1, key2).clamp(max=key1.numel() - 1)
        key2 = key2[key1[p] != key2]
    t2, v2 = key2 // V, key2 % V
    rank2, cnt2, start2 = _segment_rank(t2, nt)
    n2 = n1 + cnt2
    if int(n2.max()) > MAX_RING2 or int(n1.max()) > MAX_RING1:
        return None
    off2 = torch.zeros(nt + 1, dtype=torch.long, device=dev)
    off2[1:] = torch.cumsum(n2, 0)
    off1 = torch.zeros(nt + 1, dtype=torch.long, device=dev)
    off1[1:] = torch.cumsum(n1, 0)
    ids2 = torch.empty(int(off2[-1]), dtype=torch.long, device=dev)
    rows = ar(V)
    ids2[off2[rows // T] + rows % T] = rows
    ids2[off2[t1] + own_cnt[t1] + rank1] = v1
    ids2[off2[t2] + n1[t2] + rank2] = v2
    # the ring-1 rows of all tiles in local order, and their edges
    rows1 = torch.empty(int(off1[-1]), dtype=torch.long, device=dev)
    rows1[off1[rows // T] + rows % T] = rows
    rows1[off1[t1] + own_cnt[t1] + rank1] = v1
    tile1 = torch.repeat_interleave(ar(nt), n1)
    dr = deg[rows1]
    eptr = torch.zeros(rows1.numel() + 1, dtype=torch.long, device=dev)
    eptr[1:] = torch.cumsum(dr, 0)
    per_tile_edges = torch.zeros(nt, dtype=torch.long, device=dev).index_add_(0, tile1, dr)
    if int(per_tile_edges.max()) > MAX_EDGES:
        return None
    owner = torch.repeat_interleave(ar(rows1.numel()), dr)
    e_of = rp[rows1][owner] + (ar(int(eptr[-1])) - eptr[:-1][owner])
    src, te = ci[e_of], tile1[owner]
    # local id of (tile, source): own row / ring-1 row / ring-2 row
    key = te * V + src
    loc = src - te * T                                            # valid where the source is an own row of the tile
    if key1.numel():
        p1 = torch.searchsorted(key1, key).clamp(max=key1.numel() - 1)
        in1 = key1[p1] == key
        loc = torch.where(in1, own_cnt[te] + (p1 - start1[te]), loc)
    else:
        in1 = torch.zeros_like(key, dtype=torch.bool)
    own = (src // T) == te
    rest = ~(own | in1)
    if bool(rest.any()):
        p2 = torch.searchsorted(key2, key).clamp(max=max(key2.numel() - 1, 0))
        if key2.numel() == 0 or not bool((key2[p2][rest] == key[rest]).all()):
            raise RuntimeError("two-ring records: a source of a ring-1 row is missing from the tile's ring 2")
        loc = torch.where(rest, n1[te] + (p2 - start2[te]), loc)
    return TwoRing(nt, off2.to(torch.int32), off1.to(torch.int32), ids2.to(torch.int32), eptr.to(torch.int32),
                   loc.to(torch.uint8), int(n1.max()), int(n2.max()), int(per_tile_edges.max()))


def reference_two_hop(rec: TwoRing, dis: torch.Tensor, X: torch.Tensor, a1: float, a2: float, tile: int = TILE_ROWS):
    """What the records mean, in plain index arithmetic (tests): ``U = a1 L^ X`` on every tile's ring-1 rows and
    ``Y = a2 L^ U`` on its own rows, both read through the records only.  ``L^[i, j] = -dis[i] dis[j]`` per edge j -> i.
    Returns (U on the own rows [V, C], Y [V, C])."""
    V = X.shape[0]
    off1, off2, eptr = rec.off1.long(), rec.off2.long(), rec.eptr.long()
    n1 = off1[1:] - off1[:-1]
    tile1 = torch.repeat_interleave(torch.arange(rec.n_tiles), n1)
    deg = eptr[1:] - eptr[:-1]
    row_of_edge = torch.repeat_interleave(torch.arange(deg.numel()), deg)
    te = tile1[row_of_edge]
    src_global = rec.ids2.long()[off2[te] + rec.eidx.long()]
    rows1 = rec.ids2.long()[(off2[tile1] + (torch.arange(tile1.numel()) - off1[tile1]))]
    acc = torch.zeros((deg.numel(), X.shape[1]), dtype=X.dtype).index_add_(0, row_of_edge, dis[src_global].view(-1, 1) * X[src_global])
    U1 = -a1 * dis[rows1].view(-1, 1) * acc                       # on all ring-1 rows of all tiles
    # outer hop: the own rows are the first rows of every tile's ring-1 list; their sources are ring-1 rows (local id < n1)
    local_row = torch.arange(tile1.numel()) - off1[tile1]
    own_cnt = (V - torch.arange(rec.n_tiles) * tile).clamp(max=tile)
    is_own_edge = local_row[row_of_edge] < own_cnt[te]
    e = torch.nonzero(is_own_edge).flatten()
    src_local = off1[te[e]] + rec.eidx.long()[e]
    assert bool((rec.eidx.long()[e] < n1[te[e]]).all())
    acc2 = torch.zeros((deg.numel(), X.shape[1]), dtype=X.dtype).index_add_(0, row_of_edge[e], dis[src_global[e]].view(-1, 1) * U1[src_local])
    Y1 = -a2 * dis[rows1].view(-1, 1) * acc2
    own_rows = torch.nonzero(local_row < own_cnt[tile1]).flatten()
    U = torch.zeros_like(X)
    Y = torch.zeros_like(X)
    U[rows1[own_rows]] = U1[own_rows]
    Y[rows1[own_rows]] = Y1[own_rows]
    return U, Y


def main():
    import argparse
    import json
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from semigcn_amd import reorder
    global MAX_RING2, MAX_RING1, MAX_EDGES
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", default="1000x1000")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    nu, nv = map(int, a.mesh.split("x"))
    m = bench.make_mesh(nu, nv, "survey")
    V = m.num_vertices
    ei = torch.from_numpy(m.edge_index)
    order, rank = reorder.morton_order(torch.from_numpy(m.x_pos))            # the processing order of SingleScaleGCN(reorder=True)
    ei = reorder.permute_edge_index(ei, rank)
    key = torch.sort(ei[0] * V + ei[1])[0]
    r, c = key // V, key % V
    rowptr = torch.zeros(V + 1, dtype=torch.long)
    rowptr[1:] = torch.cumsum(torch.bincount(r, minlength=V), 0)
    # self-check on a small mesh: the records reproduce L^ X and L^ (L^ X)
    ms = bench.make_mesh(37, 29, "survey")
    es = torch.from_numpy(ms.edge_index)
    ks = torch.sort(es[0] * ms.num_vertices + es[1])[0]
    rs, cs = ks // ms.num_vertices, ks % ms.num_vertices
    rps = torch.zeros(ms.num_vertices + 1, dtype=torch.long)
    rps[1:] = torch.cumsum(torch.bincount(rs, minlength=ms.num_vertices), 0)
    lim = (MAX_RING2, MAX_RING1, MAX_EDGES)
    MAX_RING2 = MAX_RING1 = MAX_EDGES = 10 ** 9
    small = build(rps.int(), cs.int(), ms.num_vertices)
    dis = (rps[1:] - rps[:-1]).double().clamp(min=1).pow(-0.5)
    X = torch.randn(ms.num_vertices, 5, dtype=torch.float64)
    L = torch.zeros(ms.num_vertices, ms.num_vertices, dtype=torch.float64).index_put_((rs, cs), -(dis[rs] * dis[cs]), accumulate=True)
    if int((small.off2[1:] - small.off2[:-1]).max()) <= 255:
        U, Y = reference_two_hop(small, dis, X, 1.0, 2.0)
        assert float((U - L @ X).abs().max()) < 1e-12 and float((Y - 2 * L @ (L @ X)).abs().max()) < 1e-12
    out = {"mesh": a.mesh, "V": V, "order": "Morton (SingleScaleGCN reorder=True)", "tiles": []}
    for T in (64, 32, 16):
        rec = build(rowptr.int(), c.int(), V, tile=T)
        n1 = (rec.off1[1:] - rec.off1[:-1]).double()
        n2 = (rec.off2[1:] - rec.off2[:-1]).double()
        row = {"tile_rows": T, "tiles": rec.n_tiles, "ring1_rows_mean": round(float(n1.mean()), 1), "ring1_rows_max": int(n1.max()),
               "ring2_rows_mean": round(float(n2.mean()), 1), "ring2_rows_p99": int(torch.quantile(n2, 0.99)), "ring2_rows_max": int(n2.max()),
               "inner_hop_rows_per_own_row": round(float(n1.mean()) / T, 2), "rows_staged_per_own_row": round(float(n2.mean()) / T, 2),
               "tiles_over_255_ring2_rows": int((n2 > 255).sum()), "edges_per_tile_max": rec.max_edges}
        out["tiles"].append(row)
        print(json.dumps(row))
    MAX_RING2, MAX_RING1, MAX_EDGES = lim
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
