#!/usr/bin/env python3
"""Where the HOST time of one eager SGCN training iteration goes on a small mesh (the reference's own sizes are host-bound
in eager mode): cProfile over N iterations, top functions by own time.
    python tools/host_profile.py [--mesh 100x50] [--dtype fp32] [--iters 30]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from semigcn_amd import networks, train  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", default="100x50")
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--model", default="sgcn", choices=["sgcn", "mgcn"])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    nu, nv = map(int, a.mesh.split("x"))
    mesh = bench.make_mesh(nu, nv, "survey")
    batch = bench.build_mesh_batch(mesh, dev, 5)
    torch.manual_seed(0)
    if a.model == "mgcn":
        from semigcn_amd import meshprep
        from semigcn_amd.meshnet import MGCN
        import numpy as np
        smo = meshprep.DeviceMesh(mesh.x_pos, mesh.faces, dev)
        ini = meshprep.DeviceMesh(mesh.vs.astype(np.float32), mesh.faces, dev)
        net = MGCN(dev, smo, ini, torch.from_numpy(mesh.v_mask)).to(dev)
        if a.dtype == "bf16":
            net.set_feature_dtype(torch.bfloat16)
        tr = train.MGCNTrainer(net, batch)
    else:
        net = networks.SingleScaleGCN(dev).to(dev)
        if a.dtype == "bf16":
            net.set_feature_dtype(torch.bfloat16)
        tr = train.SGCNTrainer(net, batch)
    for _ in range(5):
        tr.iteration_step()
    torch.cuda.synchronize()
    from semigcn_amd import capi
    c0 = list(capi.chain_host_seconds)
    t0 = time.perf_counter()
    for _ in range(a.iters):
        tr.iteration_step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / a.iters * 1e3:.2f} ms per eager iteration (host enqueue {(t1 - t0) / a.iters * 1e3:.2f} ms)")
    if os.environ.get("SEMIGCN_PROFILE_CHAINS") == "1":
        print(f"  of it inside sg_block_chain_forward {(capi.chain_host_seconds[0] - c0[0]) / a.iters * 1e3:.2f} ms, "
              f"sg_block_chain_backward {(capi.chain_host_seconds[1] - c0[1]) / a.iters * 1e3:.2f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.iters):
        tr.iteration_step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
