#!/usr/bin/env python3
"""Can a weight-gradient product run UNDER the aggregation launches of the same layer's backward pass?  Times, at V = 1 M
with bf16 features: (a) two fused aggregations (C = 256), (b) one weight-gradient product dW = dOut^T T, (c) both issued
back to back on one stream, (d) the product on a side stream while the aggregations run on the main one.

    python tools/overlap_probe.py [--C 256] [--N 256] [--Kp 768]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semigcn_amd import capi, functional as F_sg, synth  # noqa: E402
from semigcn_amd.graph import MeshGraph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--C", type=int, default=256)
    ap.add_argument("--N", type=int, default=256)
    ap.add_argument("--Kp", type=int, default=768)
    ap.add_argument("--rounds", type=int, default=7)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    m = synth.torus_mesh(1000, 1000, masks=False)
    V = m.num_vertices
    g = MeshGraph.from_edge_index(torch.from_numpy(m.edge_index).to(dev), V)
    T = torch.randn(V, 3 * a.C, device=dev).to(torch.bfloat16)
    blk = [T[:, k * a.C:(k + 1) * a.C] for k in range(3)]
    dx = torch.empty(V, a.C, device=dev, dtype=torch.bfloat16)
    dout = torch.randn(V, a.N, device=dev).to(torch.bfloat16)
    Tw = torch.randn(V, a.Kp, device=dev).to(torch.bfloat16)
    side = torch.cuda.Stream(dev)

    def aggs():
        g.aggregate(blk[2], blk[1], alpha=2.0, X0=blk[1], beta=1.0)
        g.aggregate(blk[1], dx, alpha=1.0, X0=blk[0], beta=1.0, X1=blk[2], gamma=-1.0)

    def wgrad():
        return F_sg.weight_grad(dout, Tw)

    def serial():
        wgrad()
        aggs()

    def forked():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            wgrad()
        aggs()
        cur.wait_stream(side)

    variants = {"aggregations": aggs, "weight_grad": wgrad, "serial": serial, "forked": forked}
    times = {k: [] for k in variants}
    for rnd in range(a.rounds + 1):
        for k, fn in variants.items():
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[k].append(e0.elapsed_time(e1) / 3)
    print(f"C={a.C} dW [{a.N} x V] x [V x {a.Kp}]: " + "  ".join(f"{k} {np.median(v):.3f} ms" for k, v in times.items()))


if __name__ == "__main__":
    main()
