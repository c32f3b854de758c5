"""CPU restatement of the reference's refinement solve, ``Mesh.mesh_merge`` (util/mesh.py:678-698).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Same stacked least-squares system as the
reference builds, assembled sparsely and solved in float64 through the normal equations
(scipy ``spsolve``); pinned against tests/golden/g5_refine.npz, the output of the reference's own
dense float32 ``torch.linalg.solve`` on the fixture meshes.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def uniform_laplacian(edge_index: np.ndarray, num_vertices: int):
    """Lap = I - D^-1 A  (util/mesh.py:262-274, build_v2v)."""
    A = sp.coo_matrix((np.ones(edge_index.shape[1]), (edge_index[0], edge_index[1])),
                      shape=(num_vertices, num_vertices)).tocsr()
    A.data[:] = 1.0
    deg = np.asarray(A.sum(1)).reshape(-1)
    return sp.identity(num_vertices, format="csr") - sp.diags(1.0 / deg) @ A, A


def mesh_merge(edge_index, org_pos, new_pos, preserve, w=1.0, w_b=0.0):
    V = org_pos.shape[0]
    org, new = np.asarray(org_pos, np.float64), np.asarray(new_pos, np.float64)
    keep = np.asarray(preserve).astype(bool)
    L, A = uniform_laplacian(np.asarray(edge_index), V)
    AI = A + sp.identity(V, format="csr")
    inner = np.asarray(AI @ (1.0 - keep.astype(np.float64))).reshape(-1) == 0     # util/mesh.py:682
    border = np.logical_xor(keep, inner)
    eye = sp.identity(V, format="csr")
    stack = sp.vstack([L, eye[inner] * w, eye[border] * w_b]).tocsr()
    b_mix = L @ new
    b_mix[inner] = (L @ org)[inner]
    b = np.concatenate([b_mix, org[inner] * w, org[border] * w_b], axis=0)
    AtA = (stack.T @ stack).tocsc()
    return spla.spsolve(AtA, stack.T @ b)
