"""CPU restatement of the reference's mesh connectivity and dummy-mask producers.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned against
tests/golden/g4_meshprep.npz, which holds the outputs of the reference's own
``Mesh`` (util/mesh.py) and ``make_dummy_mask`` / ``vmask_to_fmask``
(util/datamaker.py) on three small meshes.

  edges_first_meeting   util/mesh.py:60-100   (build_gemm -> self.edges)
  edge_index            util/mesh.py:229-230
  face_ring             util/mesh.py:214-227  (f2f; rows compared as sets, the
                                               reference's order is Python-set order)
  make_dummy_mask       util/datamaker.py:110-136
  vmask_to_fmask        util/datamaker.py:156-159
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

P_LIST = (0.014, 0.014, 0.014, 0.014, 0.014, 0.014, 0.014, 0.0014, 0.014)  # util/datamaker.py:118


def edges_first_meeting(faces: np.ndarray) -> np.ndarray:
    """The dict-based scan of util/mesh.py:69-88, literally."""
    seen = {}
    edges = []
    for f in np.asarray(faces):
        for i in range(3):
            e = tuple(sorted((int(f[i]), int(f[(i + 1) % 3]))))
            if e not in seen:
                seen[e] = len(edges)
                edges.append(e)
    return np.array(edges, dtype=np.int64).reshape(-1, 2)


def edge_index(edges: np.ndarray) -> np.ndarray:
    e = edges.T
    return np.concatenate([e, e[[1, 0]]], axis=1)


def face_ring(faces: np.ndarray, num_vertices: int) -> np.ndarray:
    """Faces sharing exactly two vertices with each face (util/mesh.py:216-224), ascending,
    -1 padded to three columns."""
    faces = np.asarray(faces)
    F = len(faces)
    inc = sp.coo_matrix((np.ones(3 * F, np.int32), (np.repeat(np.arange(F), 3), faces.reshape(-1))),
                        shape=(F, num_vertices)).tocsr()
    shared = (inc @ inc.T).tocoo()
    out = -np.ones((F, 3), np.int64)
    fill = np.zeros(F, np.int64)
    order = np.lexsort((shared.col, shared.row))
    for r, c, v in zip(shared.row[order], shared.col[order], shared.data[order]):
        if v == 2 and r != c:
            out[r, fill[r]] = c
            fill[r] += 1
    return out


def adjacency_plus_identity(edges: np.ndarray, num_vertices: int):
    ei = edge_index(edges)
    A = sp.coo_matrix((np.ones(ei.shape[1], np.float32), (ei[0], ei[1])), shape=(num_vertices, num_vertices))
    return (A + sp.identity(num_vertices, dtype=np.float32)).tocsr()


def vmask_to_fmask(faces: np.ndarray, vmask: np.ndarray) -> np.ndarray:
    """(f2v_mat @ (1 - vmask)) == 0: no vertex of the face is dropped."""
    vm = np.asarray(vmask) != 0
    return vm[np.asarray(faces)].all(axis=1)


def make_dummy_mask(faces, edges, num_vertices, dm_size=40, kn=(3, 4, 5), p_list=P_LIST, rng=np.random):
    AI = adjacency_plus_identity(edges, num_vertices)
    cols = []
    for k in kn:
        p = float(np.float32(p_list[k]))    # the reference indexes a float32 torch tensor (util/datamaker.py:118,123)
        M = rng.binomial(1, p, size=[num_vertices, dm_size]).astype(np.float32)
        for _ in range(k):
            M = ((AI @ M) > 0).astype(np.float32)
        cols.append(1.0 - M)
    vmask = np.concatenate(cols, axis=1)
    return vmask, vmask_to_fmask(faces, vmask).astype(np.float32)
