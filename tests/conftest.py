"""pytest configuration.

Markers
-------
``gpu``  needs a real MI355X; these are the parity tests proper and call the HIP
         kernels through the C ABI.  Everything else runs on CPU.

CPU test double
---------------
Host logic (autograd wiring, module composition, state-dict layout, partitioning,
halo exchange) is exercised on CPU by the ``cpu_kernels`` fixture, which swaps the
ctypes handles in ``semigcn_amd.capi`` for doubles built on the ORACLE
(oracle/pyg_restatement.py).  That swap exists only inside tests: the product has
no CPU path, and ``test_capi.py`` checks that it raises without a device.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu on the GPU box")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _seed_every_test(request):
    """Tests that draw inputs without an explicit generator still see the SAME inputs on every run (a seed derived from
    the test id): a rounding-level kink hit on one box is then reproducible instead of a 1-in-N flake."""
    import zlib
    seed = zlib.crc32(request.node.nodeid.encode()) & 0x7FFFFFFF
    torch.manual_seed(seed)
    np.random.seed(seed % (2 ** 32))
    yield


class OracleGraphDouble:
    """CPU stand-in for capi.GraphHandle: same methods, oracle arithmetic."""

    def __init__(self, dst, src, n_rows, n_cols, dis_dst, dis_src, symmetric, square=True):
        self.dst, self.src = dst.long(), src.long()
        self.num_rows, self.num_cols = int(n_rows), int(n_cols)
        self.dis_dst, self.dis_src = dis_dst, dis_src
        self.nnz = int(dst.numel())
        self.symmetric, self.square = symmetric, square
        self.device = dst.device
        self.max_degree = int(torch.bincount(self.dst, minlength=1).max()) if self.nnz else 0

    @classmethod
    def from_edge_index(cls, edge_index, num_vertices):
        from oracle.pyg_restatement import remove_self_loops, scatter_sum
        ei, _ = remove_self_loops(edge_index)
        row, col = ei[0], ei[1]
        deg = scatter_sum(torch.ones(row.numel()), row, num_vertices)
        dis = deg.pow(-0.5)
        dis[dis == float("inf")] = 0
        a = torch.sort(row * num_vertices + col)[0]
        b = torch.sort(col * num_vertices + row)[0]
        return cls(col, row, num_vertices, num_vertices, dis, dis, bool(torch.equal(a, b)))

    @classmethod
    def from_partition(cls, dst, src, n_owned, n_ext, dis_ext):
        return cls(dst, src, n_owned, n_ext, dis_ext[:n_owned].float(), dis_ext.float(), True, square=False)

    @classmethod
    def from_rows(cls, dst_pos, src, row_id, out_rows, n_ext, dis_rows, dis_ext):
        """Row subset (sg_graph_create_rows): processed row p writes row row_id[p] of Y and leaves the others alone."""
        h = cls(row_id.long()[dst_pos.long()], src, out_rows, n_ext, None, dis_ext.float(), True, square=False)
        dd = torch.zeros(out_rows)
        dd[row_id.long()] = dis_rows.float()
        h.dis_dst = dd
        h.rows_written = row_id.long()
        return h

    def arrays(self):
        order = torch.argsort(self.dst * self.num_cols + self.src)
        rowptr = torch.zeros(self.num_rows + 1, dtype=torch.int32)
        rowptr[1:] = torch.cumsum(torch.bincount(self.dst, minlength=self.num_rows), 0).int()
        return rowptr, self.src[order].int(), self.dis_src.clone()

    def spmm(self, X, Y, *, alpha=1.0, X0=None, beta=0.0, X1=None, gamma=0.0, transpose=False):
        from oracle.pyg_restatement import scatter_sum
        if transpose:
            assert self.square
            gather, reduce_at = self.dst, self.src
        else:
            gather, reduce_at = self.src, self.dst
        w = -(self.dis_src[self.src] * self.dis_dst[self.dst]).to(torch.float32)
        msg = w.view(-1, 1) * X.float().index_select(0, gather)
        out = alpha * scatter_sum(msg, reduce_at, Y.shape[0])
        if X0 is not None:
            out = out + beta * X0.float()
        if X1 is not None:
            out = out + gamma * X1.float()
        rows = getattr(self, "rows_written", None)
        if rows is None:
            Y.copy_(out.to(Y.dtype))
        else:
            Y[rows] = out[rows].to(Y.dtype)
        return Y

    def close(self):
        pass


class OraclePoolDouble:
    def __init__(self, fine, coarse, n_fine, n_coarse):
        self.h = np.stack([fine.cpu().numpy(), coarse.cpu().numpy()], 1)
        self.n_fine, self.n_coarse, self.device = int(n_fine), int(n_coarse), fine.device
        self.cnt = torch.bincount(coarse.long(), minlength=n_coarse).float().clamp(min=1).view(-1, 1)

    def pool_mean(self, X):
        from oracle.models import pool_mean
        return pool_mean(self.h, X, self.n_coarse)

    def unpool(self, X):
        from oracle.models import unpool_gather
        return unpool_gather(self.h, X, self.n_fine)

    def pool_mean_bwd(self, dY):
        from oracle.models import unpool_gather
        return unpool_gather(self.h, dY / self.cnt, self.n_fine)

    def unpool_bwd(self, dY):
        f, c = torch.as_tensor(self.h[:, 0]), torch.as_tensor(self.h[:, 1])
        return dY.new_zeros(self.n_coarse, dY.shape[1]).index_add_(0, c, dY[f])

    def close(self):
        pass


@pytest.fixture
def cpu_kernels(monkeypatch):
    """Swap the HIP handles for oracle-backed doubles so host logic runs on CPU."""
    from semigcn_amd import capi, graph
    monkeypatch.setattr(capi, "GraphHandle", OracleGraphDouble)
    monkeypatch.setattr(capi, "PoolHandle", OraclePoolDouble)
    monkeypatch.setattr(capi, "gather_rows", lambda rows, X, out=None: X.index_select(0, rows.long()))
    graph.clear_graph_cache()
    yield
    graph.clear_graph_cache()


@pytest.fixture(scope="session")
def fixture_meshes():
    from semigcn_amd import synth
    return {"sphere": synth.octahedron_sphere(3), "torus": synth.torus_mesh(20, 12)}
