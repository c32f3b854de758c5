"""spmm_ring_f32 against spmm_rows (and the shared-gather kernel for plain 1 KB rows) on the benchmark mesh, float32 column
blocks of [V, 3 C] buffers as in the model: fraction of outputs whose bits differ, time per launch with the two variants
interleaved (A/B by SG_TUNE_FLAGS bit 13), GB/s on SURVEY 8(d)'s algorithmic bytes.

    python tools/ring_f32_probe.py [1000x1000]
"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semigcn_amd import capi, synth, reorder  # noqa: E402

dev = "cuda:0"
nu, nv = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1000x1000").split("x"))
m = synth.torus_mesh(nu, nv, masks=False)
V = m.num_vertices
ei = torch.from_numpy(m.edge_index).to(dev)
ei = reorder.permute_edge_index(ei, reorder.morton_order(torch.from_numpy(m.x_pos).to(dev))[1])
g = capi.GraphHandle.from_edge_index(ei, V)
for C in (128, 256):
    wide = torch.randn(V, 3 * C, device=dev)
    x, x0, x1 = wide[:, :C], wide[:, C:2 * C], wide[:, 2 * C:]
    for nepi, kw in ((0, {}), (1, {"alpha": 2.0, "X0": x0, "beta": -1.0}), (2, {"alpha": 1.0, "X0": x0, "beta": 1.0, "X1": x1, "gamma": -1.0})):
        res = {}
        variants = (8193, 1)
        ys = {f: torch.empty((V, C), device=dev) for f in variants}
        best = {f: 1e9 for f in variants}
        for rnd in range(6):                       # variants interleaved: a launch's time moves with what ran before it
            for flags in variants:
                capi.tuning_set(capi.TUNE_FLAGS, flags)
                g.spmm(x, ys[flags], **kw)
                evs = []
                for _ in range(5):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    g.spmm(x, ys[flags], **kw)
                    b.record()
                    evs.append((a, b))
                torch.cuda.synchronize()
                ts = sorted(p.elapsed_time(q) for p, q in evs)
                if rnd > 0:
                    best[flags] = min(best[flags], ts[len(ts) // 2])
        for f in variants:
            res[f] = (ys[f], best[f])
        capi.tuning_set(capi.TUNE_FLAGS, 1)
        d = (res[1][0] - res[8193][0]).abs()
        nbytes = (3 + nepi) * V * C * 4 + 4 * m.num_edges + 4 * (V + 1) + 4 * V
        nbytes = (2 + nepi) * V * C * 4 + 4 * m.num_edges + 4 * (V + 1) + 4 * V
        print(f"C={C} nepi={nepi}: rows {res[8193][1]:.4f} ms ({nbytes / res[8193][1] / 1e6:.0f} GB/s)  ring {res[1][1]:.4f} ms ({nbytes / res[1][1] / 1e6:.0f} GB/s)  "
              f"differing {float((d > 0).float().mean()):.4f} max rel {float((d / res[8193][0].abs().clamp(min=1e-3)).max()):.2e}", flush=True)
