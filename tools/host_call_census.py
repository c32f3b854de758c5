#!/usr/bin/env python3
"""Host-side calls per eager training iteration: foreign calls into libsemigcn_hip.so (every ctypes entry point counted
through a proxy) and top-level ATen operator dispatches (torch.profiler, backward thread included; operators that another
ATen operator called are not counted again), next to the kernel launches they produce.
    python tools/host_call_census.py [--mesh 250x200] [--dtype fp32|bf16] [--model sgcn|mgcn] [--iters 10]"""
import argparse
import collections
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from semigcn_amd import capi, functional as F_sg, networks, train  # noqa: E402


class CountingLib:
    def __init__(self, lib):
        self._lib, self.calls = lib, collections.Counter()

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        calls = self.calls

        def counted(*a):
            calls[name] += 1
            return fn(*a)
        return counted


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", default="250x200")
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--model", default="sgcn", choices=["sgcn", "mgcn"])
    ap.add_argument("--json", default=None)
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
    for _ in range(10):
        tr.iteration_step()
    torch.cuda.synchronize()
    proxy = CountingLib(capi._lib)
    capi._lib = proxy
    b0 = list(F_sg.block_calls)
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(a.iters):
            tr.iteration_step()
        torch.cuda.synchronize()
    capi._lib = proxy._lib
    aten, kernels = collections.Counter(), 0
    for ev in prof.events():
        if str(ev.device_type).endswith("CUDA"):
            kernels += 1
            continue
        if ev.name.startswith("aten::"):
            par = ev.cpu_parent
            if par is None or not par.name.startswith("aten::"):
                aten[ev.name] += 1
    n = float(a.iters)
    foreign = sum(proxy.calls.values()) / n
    ops = sum(aten.values()) / n
    # allocations and views launch nothing: reported, but not part of the call count that costs a launch or a library entry
    free = sum(v for k, v in aten.items() if k in ("aten::empty", "aten::empty_strided", "aten::view", "aten::as_strided", "aten::slice",
                                                   "aten::select", "aten::t", "aten::transpose", "aten::detach", "aten::alias",
                                                   "aten::_unsafe_view", "aten::reshape", "aten::expand", "aten::unsqueeze",
                                                   "aten::squeeze", "aten::permute", "aten::result_type", "aten::item",
                                                   "aten::_local_scalar_dense", "aten::empty_like", "aten::lift_fresh")) / n
    out = {"workload": f"{a.model} {a.mesh} {a.dtype}", "iterations": a.iters,
           "foreign_calls_per_iteration": round(foreign, 1), "aten_dispatches_per_iteration": round(ops, 1),
           "of_them_allocations_and_views": round(free, 1),
           "host_side_calls_per_iteration": round(foreign + ops - free, 1),
           "device_kernels_and_copies_per_iteration": round(kernels / n, 1),
           "blocks_per_iteration": [(F_sg.block_calls[0] - b0[0]) / n, (F_sg.block_calls[1] - b0[1]) / n],
           "foreign_calls": {k: round(v / n, 1) for k, v in proxy.calls.most_common()},
           "aten": {k: round(v / n, 1) for k, v in aten.most_common(40)}}
    print(json.dumps(out, indent=1))
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
