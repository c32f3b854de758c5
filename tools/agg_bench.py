#!/usr/bin/env python3
"""Kernel-level benchmark of the edge-aggregation kernel (sg_spmm) on the benchmark mesh:
per (C, dtype, epilogue operands) the mean launch time over interleaved rounds, algorithmic
GB/s (SURVEY 8(d) byte count) and fraction of the 8 TB/s HBM peak.  Variants (tuning knobs)
are interleaved in ONE process as the CDNA guide's rule 24 asks.

    python tools/agg_bench.py [--mesh 1000x1000] [--permute] [--variants ch=32,flags=1 ch=16,flags=3 ...]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semigcn_amd import capi, synth  # noqa: E402
from semigcn_amd.graph import MeshGraph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", default="1000x1000")
    ap.add_argument("--permute", action="store_true")
    ap.add_argument("--order", default="asis", choices=["asis", "morton"])
    ap.add_argument("--channels", default="4,16,32,64,128,256,512")
    ap.add_argument("--dtypes", default="fp32,bf16")
    ap.add_argument("--epilogue", default="0,1")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--variants", nargs="*", default=["ch=0,flags=1,unroll=0"])
    ap.add_argument("--json", default=None)
    ap.add_argument("--blocks", type=int, default=1, help="operands are column blocks of [V, blocks*C] buffers (the in-model layout: 3)")
    ap.add_argument("--reorder", type=int, default=0, help="SG_TUNE_GRAPH_REORDER while the graph is created: 0 auto, 1 never, 2 always")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    nu, nv = map(int, a.mesh.split("x"))
    m = synth.torus_mesh(nu, nv, permute=a.permute, masks=False)
    V, E = m.num_vertices, m.num_edges
    ei = torch.from_numpy(m.edge_index).to(dev)
    if a.order == "morton":
        from semigcn_amd import reorder
        order, rank = reorder.morton_order(torch.from_numpy(m.vs).to(dev))
        ei = reorder.permute_edge_index(ei, rank)
    capi.tuning_set(capi.TUNE_GRAPH_REORDER, a.reorder)
    if any("flags=" in v and int(dict(kv.split("=") for kv in v.split(","))["flags"]) & 128 for v in a.variants):
        capi.tuning_set(capi.TUNE_FLAGS, 129)      # experimental LDS tile lists are only built on request
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    g = MeshGraph.from_edge_index(ei, V)
    torch.cuda.synchronize()
    print(f"graph created in {(time.perf_counter() - t0) * 1e3:.1f} ms, locality view: {g.handle.reordered}", flush=True)
    capi.tuning_set(capi.TUNE_GRAPH_REORDER, 0)
    capi.tuning_set(capi.TUNE_FLAGS, 1)
    variants = []
    for v in a.variants:
        d = dict(kv.split("=") for kv in v.split(","))
        variants.append((v, int(d.get("ch", 0)), int(d.get("flags", 1)), int(d.get("unroll", 0)), int(d.get("slab", 0)), int(d.get("tmin", 1024))))
    out = []
    for dt in a.dtypes.split(","):
        dtype = torch.float32 if dt == "fp32" else torch.bfloat16
        es = 4 if dt == "fp32" else 2
        for C in map(int, a.channels.split(",")):
            if a.blocks > 1:        # [Tx0|Tx1|Tx2]-style: X = block 0, Y = block 1 of one buffer, X0 block 0 of another
                wide = torch.randn(V, a.blocks * C, device=dev).to(dtype)
                wide0 = torch.randn(V, a.blocks * C, device=dev).to(dtype)
                x, y, x0 = wide[:, :C], wide[:, C:2 * C], wide0[:, :C]
            else:
                x = torch.randn(V, C, device=dev).to(dtype)
                x0 = torch.randn(V, C, device=dev).to(dtype)
                y = torch.empty_like(x)
            for nepi in map(int, a.epilogue.split(",")):
                times = {v[0]: [] for v in variants}
                for rnd in range(a.rounds + 1):
                    for name, ch, flags, unroll, slab, tmin in variants:
                        capi.tuning_set(capi.TUNE_CHUNK_ROWS, ch)
                        capi.tuning_set(capi.TUNE_FLAGS, flags)
                        capi.tuning_set(capi.TUNE_UNROLL, unroll)
                        capi.tuning_set(capi.TUNE_SLAB, slab)
                        capi.tuning_set(capi.TUNE_TILED_MIN_ROW_BYTES, tmin)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(a.reps):
                            if nepi == 2:
                                g.aggregate(x, y, alpha=1.0, X0=x0, beta=1.0, X1=x0, gamma=-1.0)
                            elif nepi:
                                g.aggregate(x, y, alpha=2.0, X0=x0, beta=-1.0)
                            else:
                                g.aggregate(x, y)
                        e1.record()
                        torch.cuda.synchronize()
                        if rnd:
                            times[name].append(e0.elapsed_time(e1) / a.reps)
                B = (2 + nepi) * V * C * es + 4.0 * E + 4.0 * (V + 1) + 4.0 * V
                for name in times:
                    med, mn = float(np.median(times[name])), float(np.min(times[name]))
                    rec = {"dtype": dt, "C": C, "epi": nepi, "variant": name, "median_ms": round(med, 4),
                           "min_ms": round(mn, 4), "GBs": round(B / med / 1e6, 1), "frac": round(B / med / 1e6 / 8000, 4)}
                    out.append(rec)
                    print(f"{dt:5s} C={C:4d} epi={nepi} {name:28s} median {med:8.4f} ms  min {mn:8.4f} ms  "
                          f"{rec['GBs']:8.1f} GB/s  {100 * rec['frac']:5.1f}% of 8 TB/s", flush=True)
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        json.dump({"mesh": a.mesh, "permute": a.permute, "order": a.order, "reorder_knob": a.reorder,
                   "locality_view": bool(g.handle.reordered), "V": V, "E": E, "results": out}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
