"""Times the device mesh preparation (semigcn_amd/meshprep.py) at benchmark sizes, next to the
numpy/scipy restatement of the reference's algorithm on the host (the reference's own dense
V x V formulation cannot be run at these sizes at all).

    python tools/prep_bench.py [--mesh 1000x1000] [--no-cpu]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semigcn_amd import meshprep, synth  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", default="1000x1000")
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    nu, nv = map(int, a.mesh.split("x"))
    m = synth.torus_mesh(nu, nv, masks=False)
    V, F = m.num_vertices, len(m.faces)
    faces = torch.from_numpy(m.faces).cuda()
    res = {"mesh": a.mesh, "V": V, "F": F}
    res["topology_ms"], topo = timed(lambda: meshprep.MeshTopology(faces, V, "cuda"))
    res["edges_only_ms"], _ = timed(lambda: meshprep.MeshTopology(faces, V, "cuda", with_f2f=False))
    res["csr_ms"], _ = timed(lambda: meshprep.MeshGraph.from_edge_index(topo.edge_index, V))
    gen = torch.Generator(device="cuda").manual_seed(1)
    res["dummy_mask_120_gpu_rng_ms"], _ = timed(lambda: meshprep.make_dummy_mask(topo, 40, (3, 4, 5), rng=gen), reps=3)
    seeds = torch.rand((V, 40), device="cuda") < 0.014
    bits = meshprep.pack_bits(seeds)
    res["one_ring_40_masks_ms"], _ = timed(lambda: topo.graph.handle.dilate_bits(bits), reps=20)
    # algorithmic bytes of one ring: read + write one word per vertex, int32 index per edge, row pointers
    ring_bytes = 16 * V + 4 * topo.edge_index.shape[1] + 4 * (V + 1)
    res["one_ring_GBps"] = ring_bytes / res["one_ring_40_masks_ms"] / 1e6
    t = time.perf_counter()
    meshprep.make_dummy_mask(topo, 40, (3, 4, 5), rng=np.random.RandomState(0))
    torch.cuda.synchronize()
    res["dummy_mask_120_numpy_rng_ms"] = (time.perf_counter() - t) * 1e3
    # refinement solve (Mesh.mesh_merge) by CG on the aggregation kernel
    from semigcn_amd import refine
    mm = synth.torus_mesh(nu, nv)
    org = torch.from_numpy(mm.vs.astype(np.float32)).cuda()
    new = torch.from_numpy(mm.x_pos).cuda() + 0.05 * torch.randn(V, 3, device="cuda")

    class M:
        vs, topology = org, topo
    torch.cuda.synchronize()
    t = time.perf_counter()
    _, info = refine.mesh_merge(None, M, new, torch.from_numpy(mm.v_mask).cuda(), return_info=True)
    torch.cuda.synchronize()
    res["mesh_merge_ms"] = (time.perf_counter() - t) * 1e3
    res["mesh_merge_cg_iterations"] = info["iterations"]
    if not a.no_cpu:
        from oracle import meshprep as MP   # checker used as the CPU comparison, as in bench.py's cpu_baseline
        t = time.perf_counter()
        e = synth.edges_from_faces(m.faces, V)
        res["cpu_edges_vectorised_ms"] = (time.perf_counter() - t) * 1e3
        t = time.perf_counter()
        MP.make_dummy_mask(m.faces, e, V, 40, (3, 4, 5), rng=np.random.RandomState(0))
        res["cpu_dummy_mask_120_ms"] = (time.perf_counter() - t) * 1e3
    print(json.dumps(res))


if __name__ == "__main__":
    main()
