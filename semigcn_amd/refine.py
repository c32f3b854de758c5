"""The refinement step that closes the reference's scripts (sgcn.py:185-189, mgcn.py:205-209,
refinement.py:36): ``Mesh.mesh_merge`` (util/mesh.py:678-698), at any mesh size.

The reference stacks  A = [Lap ; w I_S ; w_b I_B]  densely ([V + |S| + |B|, V] floats from
``torch.eye(V)``), forms A^T A and calls a dense ``torch.linalg.solve`` -- O(V^2) memory and
O(V^3) time, i.e. unusable past a few 10 K vertices.  The same least-squares problem

    minimise  |Lap x - b_mix|^2 + w^2 |x_S - org_S|^2 + w_b^2 |x_B - org_B|^2

is solved here by conjugate gradients on its normal equations with the HIP aggregation kernel as
the only sparse operator:  Lap = I - D^-1 A  (uniform Laplacian, util/mesh.py:262-274)  is
``x + dis * L^(x / dis)``  and  Lap^T y = y + L^(dis * y) / dis  with  L^ = -D^-1/2 A D^-1/2  the
operator ``sg_spmm`` applies and ``dis = D^-1/2`` the scale the graph handle already holds.
S = preserved vertices whose whole 1-ring is preserved (one ``sg_mask_dilate`` ring),
B = the remaining preserved vertices.  HIP device only.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import capi
from .graph import MeshGraph


def _graph_of(lap, org_mesh, num_vertices: int, device) -> MeshGraph:
    g = getattr(org_mesh, "graph", None)
    if isinstance(g, MeshGraph):
        return g
    topo = getattr(org_mesh, "topology", None)
    if topo is not None:
        return topo.graph
    if isinstance(lap, MeshGraph):
        return lap
    ei = getattr(org_mesh, "edge_index", None)
    if ei is None:                               # the reference's sparse Lap: off-diagonal pattern = adjacency
        idx = lap.coalesce().indices()
        ei = idx[:, idx[0] != idx[1]]
    return MeshGraph.from_edge_index(torch.as_tensor(ei).long().to(device), num_vertices)


def mesh_merge(lap, org_mesh, new_pos, preserve, w: float = 1.0, w_b: float = 0.0, *, device=None,
               tol: float = 1e-7, max_iter: int = 5000, return_info: bool = False):
    """Drop-in for ``Mesh.mesh_merge(lap, org_mesh, new_pos, preserve, w=1, w_b=0)``.

    ``lap``: the reference's sparse ``Mesh.Lap`` (only its sparsity pattern is read), a MeshGraph, or
    None when ``org_mesh`` carries ``.graph`` / ``.topology`` / ``.edge_index``.  ``org_mesh.vs`` [V,3]
    original positions; ``new_pos`` [V,3] network output; ``preserve`` [V] bool (the reference's
    v_mask).  Returns the refined positions, float32 [V,3] on the compute device.  CG stops when every
    column's residual is below ``tol`` times its right-hand side (fp32 vectors, fp64 dot products)."""
    vs = org_mesh.vs
    org = (vs if torch.is_tensor(vs) else torch.from_numpy(np.asarray(vs))).float()
    if device is None:
        device = org.device if org.is_cuda else (new_pos.device if torch.is_tensor(new_pos) and new_pos.is_cuda
                                                 else torch.device("cuda"))
    device = torch.device(device)
    if device.type != "cuda":
        raise capi.SemigcnLibraryError("mesh_merge runs on a HIP device only (there is no CPU path)")
    org = org.to(device)
    new = torch.as_tensor(new_pos).float().to(device)
    V = org.shape[0]
    keep = torch.as_tensor(preserve).reshape(-1).to(device) != 0
    g = _graph_of(lap, org_mesh, V, device)
    dis = g.handle.arrays()[2]
    if bool((dis == 0).any()):
        raise ValueError("mesh_merge: isolated vertex (the reference's D^-1 is infinite there)")
    dis = dis.view(-1, 1)
    inv_dis = 1.0 / dis

    # S: preserved and no dropped vertex in the closed 1-ring (util/mesh.py:682); B = preserved \ S
    dropped = (~keep).to(torch.int64).view(-1, 1)
    inner = g.handle.dilate_bits(dropped).view(-1) == 0
    border = keep ^ inner
    diag = (float(w) ** 2) * inner.float().view(-1, 1) + (float(w_b) ** 2) * border.float().view(-1, 1)

    buf = torch.empty_like(org)

    def lap_mul(x):                      # (I - D^-1 A) x
        g.aggregate((x * inv_dis).contiguous(), buf)
        return x + dis * buf

    def lap_t_mul(y):                    # (I - A D^-1) y
        g.aggregate((y * dis).contiguous(), buf)
        return y + inv_dis * buf

    def normal_mul(x):
        return lap_t_mul(lap_mul(x)) + diag * x

    b_mix = torch.where(inner.view(-1, 1), lap_mul(org), lap_mul(new))       # util/mesh.py:689-690
    rhs = lap_t_mul(b_mix) + diag * org

    def dot(a, b):
        return (a.double() * b.double()).sum(0)

    x = torch.where(keep.view(-1, 1), org, new)
    r = rhs - normal_mul(x)
    p = r.clone()
    rs = dot(r, r)
    stop = (float(tol) ** 2) * dot(rhs, rhs).clamp_min(1e-300)
    it = 0
    while it < max_iter and bool((rs > stop).any()):
        q = normal_mul(p)
        alpha = rs / dot(p, q).clamp_min(1e-300)
        x = x + alpha.float() * p
        r = r - alpha.float() * q
        rs_new = dot(r, r)
        p = r + (rs_new / rs.clamp_min(1e-300)).float() * p
        rs = rs_new
        it += 1
    if return_info:
        return x, {"iterations": it, "relative_residual": float((rs / dot(rhs, rhs).clamp_min(1e-300)).sqrt().max())}
    return x
