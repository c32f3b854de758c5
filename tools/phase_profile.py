#!/usr/bin/env python3
"""Host time of one eager training iteration by PHASE (mask, network forward, loss, backward, optimiser) on a small mesh,
where the host is the bound: perf_counter around the phases of SGCNTrainer / MGCNTrainer's iteration (no synchronisation in
between, so a phase's time is its enqueue time as long as the GPU keeps up).
    python tools/phase_profile.py [--mesh 100x50] [--model sgcn|mgcn] [--dtype fp32|bf16]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from semigcn_amd import capi, functional as F_sg, networks, train  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", default="100x50")
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--iters", type=int, default=40)
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
    ph = np.zeros(6)
    b = batch
    t_all = time.perf_counter()
    for it in range(a.iters):
        t0 = time.perf_counter()
        dm = b.v_keep * b.dummy_masks[:, it % 5:it % 5 + 1]
        t1 = time.perf_counter()
        out = net(tr._data, dm)
        t2 = time.perf_counter()
        if a.model == "mgcn":
            s0 = F_sg.mesh_loss_sums(out[0], b.faces, net.poss_list[0], tr.keeps[0], b.target_fn, b.f_keep)
            loss = tr.weights[0] * torch.sqrt(s0[0] / tr.counts[0] + 1.0e-6) + tr.k1 * (s0[1] / b.n_f_keep)
            for w, p, t, keep, n in list(zip(tr.weights, out, net.poss_list, tr.keeps, tr.counts))[1:]:
                loss = loss + w * train.masked_position_rmse(p, t, keep, n)
        else:
            loss = tr.loss(out)
        t3 = time.perf_counter()
        with F_sg.sink_param_grads():
            loss.backward()
        t4 = time.perf_counter()
        tr.loss_sum += loss.detach()
        tr.iteration += 1
        if tr.iteration % tr.accumulate == 0:
            tr.opt.step()
            tr.grads.zero()
        t5 = time.perf_counter()
        ph += np.array([t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0])
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t_all) / a.iters * 1e3
    ph = ph / a.iters * 1e3
    print(f"{a.model} {a.mesh} {a.dtype}: wall {wall:.2f} ms / iteration; host: mask {ph[0]:.3f}  forward {ph[1]:.3f}  loss {ph[2]:.3f}  "
          f"backward {ph[3]:.3f}  optimiser (1 step in {tr.accumulate}) {ph[4]:.3f}  sum {ph[5]:.3f} ms")
    print(f"  inside the chain calls: forward {capi.chain_host_seconds[0] / (a.iters + 6) * 1e3:.3f} ms, backward "
          f"{capi.chain_host_seconds[1] / (a.iters + 6) * 1e3:.3f} ms per iteration")


if __name__ == "__main__":
    main()
