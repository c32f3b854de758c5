#!/usr/bin/env python3
"""Host-side cost per call of the ctypes binding (tiny graph, GPU time negligible)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semigcn_amd import capi, synth
m = synth.torus_mesh(16, 16, masks=False)
h = capi.GraphHandle.from_edge_index(torch.from_numpy(m.edge_index).cuda(), m.num_vertices)
x = torch.randn(m.num_vertices, 64, device="cuda"); y = torch.empty_like(x)
for name, fn in (("spmm", lambda: h.spmm(x, y)), ("spmm+epi", lambda: h.spmm(x, y, alpha=2.0, X0=x, beta=-1.0)),
                 ("col_moments", lambda: capi.col_moments(x)), ("torch.add", lambda: torch.add(x, x, out=y))):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(2000): fn()
    dt = (time.perf_counter() - t) / 2000; torch.cuda.synchronize()
    print(f"{name:12s} {dt * 1e6:6.1f} us per call (host)")
