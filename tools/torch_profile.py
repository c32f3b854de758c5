#!/usr/bin/env python3
"""torch.profiler over a few eager training iterations (CPU side: which operators and autograd nodes the host spends its
time in, backward thread included).   python tools/torch_profile.py [--model mgcn] [--mesh 250x200] [--dtype fp32]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from semigcn_amd import networks, train  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", default="250x200")
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--model", default="mgcn", choices=["sgcn", "mgcn"])
    ap.add_argument("--rows", type=int, default=45)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    nu, nv = map(int, a.mesh.split("x"))
    mesh = bench.make_mesh(nu, nv, "survey")
    batch = bench.build_mesh_batch(mesh, dev, 5)
    torch.manual_seed(0)
    if a.model == "mgcn":
        from semigcn_amd import meshprep
        from semigcn_amd.meshnet import MGCN
        smo = meshprep.DeviceMesh(mesh.x_pos, mesh.faces, dev)
        ini = meshprep.DeviceMesh(mesh.vs.astype(np.float32), mesh.faces, dev)
        net = MGCN(dev, smo, ini, torch.from_numpy(mesh.v_mask)).to(dev)
        tr = train.MGCNTrainer(net, batch)
    else:
        net = networks.SingleScaleGCN(dev).to(dev)
        tr = train.SGCNTrainer(net, batch)
    if a.dtype == "bf16":
        net.set_feature_dtype(torch.bfloat16)
    for _ in range(6):
        tr.iteration_step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        for _ in range(a.iters):
            tr.iteration_step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=a.rows, max_name_column_width=60))


if __name__ == "__main__":
    main()
