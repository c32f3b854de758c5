#!/usr/bin/env python3
"""Host cost of ONE block call (forward, backward) by feature dtype and size: the enqueue time of sg_block_chain_forward /
_backward for a [ChebConv(K=3) -> BatchNorm -> LeakyReLU] block, the autograd node around it included.
    python tools/block_host_cost.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semigcn_amd import capi, functional as F_sg, nn as sgnn, synth  # noqa: E402
from semigcn_amd.graph import MeshGraph  # noqa: E402

DEV = "cuda:0"


def main():
    m = synth.torus_mesh(100, 50)
    g = MeshGraph.from_edge_index(torch.from_numpy(m.edge_index).to(DEV), m.num_vertices)
    for dtype in (torch.float32, torch.bfloat16):
        for cin, cout in ((32, 64), (64, 128), (256, 256), (256, 128)):
            seq = sgnn.Sequential("x, edge_index", [(sgnn.ChebConv(cin, cout, K=3), "x, edge_index -> x"),
                                                    (torch.nn.BatchNorm1d(cout), "x -> x"), (torch.nn.LeakyReLU(), "x -> x")]).to(DEV)
            x = torch.randn(m.num_vertices, cin, device=DEV).to(dtype).requires_grad_(True)
            r = torch.randn(m.num_vertices, cout, device=DEV).to(dtype)
            for _ in range(5):
                seq(x, g).backward(r)
            torch.cuda.synchronize()
            n = 60
            tf = tb = 0.0
            for _ in range(n):
                t0 = time.perf_counter()
                y = seq(x, g)
                t1 = time.perf_counter()
                y.backward(r)
                t2 = time.perf_counter()
                tf += t1 - t0
                tb += t2 - t1
                torch.cuda.synchronize()
            print(f"{str(dtype):16s} {cin:4d} -> {cout:4d}: forward {tf / n * 1e6:7.1f} us   backward {tb / n * 1e6:7.1f} us (host, per call)")


if __name__ == "__main__":
    main()
