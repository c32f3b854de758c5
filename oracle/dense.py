"""fp64 dense evaluation of the scaled Laplacian -- the accuracy arbiter.

TEST INFRASTRUCTURE ONLY.  ``L^ = -D^-1/2 A D^-1/2`` exactly as the reference's
own ``Mesh.build_adj_mat`` forms ``Adj`` and ``D_minus_half``
(/root/reference/util/mesh.py:276-285), evaluated densely in float64; and the
Chebyshev recurrence ``T0 = I, T1 = L^, Tk = 2 L^ T(k-1) - T(k-2)`` that the
reference itself spells out in ``Mesh.get_chebconv_coef`` (util/mesh.py:352-366).
Small meshes only (O(V^2) memory).
"""
from __future__ import annotations

import numpy as np


def dense_lhat(edge_index: np.ndarray, num_vertices: int) -> np.ndarray:
    ei = np.asarray(edge_index)
    keep = ei[0] != ei[1]
    src, dst = ei[0][keep], ei[1][keep]
    A = np.zeros((num_vertices, num_vertices), np.float64)
    np.add.at(A, (dst, src), 1.0)          # out[dst] += x[src]
    deg = np.zeros(num_vertices, np.float64)
    np.add.at(deg, src, 1.0)               # PyG degree is taken over edge_index[0]
    with np.errstate(divide="ignore"):
        dis = np.where(deg > 0, deg ** -0.5, 0.0)
    return -(dis[:, None] * A * dis[None, :])


def cheb_conv_dense(x: np.ndarray, edge_index: np.ndarray, weights, bias=None) -> np.ndarray:
    """out = sum_k T_k(L^) x W_k^T + b, float64."""
    x = np.asarray(x, np.float64)
    L = dense_lhat(edge_index, x.shape[0])
    Tx = [x]
    if len(weights) > 1:
        Tx.append(L @ x)
    for _ in range(2, len(weights)):
        Tx.append(2.0 * (L @ Tx[-1]) - Tx[-2])
    out = sum(t @ np.asarray(w, np.float64).T for t, w in zip(Tx, weights))
    if bias is not None:
        out = out + np.asarray(bias, np.float64)
    return out
