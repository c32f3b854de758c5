#!/usr/bin/env python3
"""Experiment behind DESIGN.md section 8 / 3.10: why hipGraph replay of the training iteration is opt-in and tied to
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 on this ROCm 7.2 / PyTorch 2.10 image.

An MGCN training iteration (forward + loss + backward, ~1280 kernel nodes, captured as ONE linear chain:
hipGraphGetEdges shows 1 root, 1 leaf, no fan-out) is replayed five times and the accumulated parameter
gradients are compared with five eager iterations.

    python tools/graph_replay_check.py

    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tools/graph_replay_check.py      # all four cases exact

Observed on MI355X with the runtime's defaults:
  * replays launched back to back, or preceded only by device-to-device memcpys:      identical to eager
  * an unrelated eager elementwise kernel on the same stream + an idle GPU before the
    replay (device or stream synchronize between iterations):                         gradients wrong
    (relative L2 20, deterministic), fixed by AMD_SERIALIZE_KERNEL=3
SGCN iterations captured the same way replayed correctly in every pattern tried."""
import gc
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from semigcn_amd import meshprep, synth, train  # noqa: E402
from semigcn_amd.meshnet import MGCN  # noqa: E402

DEV = "cuda:0"
m = synth.torus_mesh(250, 200)
batch = bench.build_mesh_batch(m, torch.device(DEV), n_masks=3)
smo = meshprep.DeviceMesh(m.x_pos, m.faces, DEV)
ini = meshprep.DeviceMesh(m.vs.astype(np.float32), m.faces, DEV)


def fresh():
    torch.manual_seed(314)
    net = MGCN(DEV, smo, ini, torch.from_numpy(m.v_mask)).to(DEV).train()
    for mod in net.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return net


def grads(graphed: bool, eager_kernel_between: bool, sync_between: bool, n: int = 5):
    net = fresh()
    tr = train.MGCNTrainer(net, batch, accumulate=1000)

    def body():
        poss = net(batch.data, None)
        loss = sum(w * train.masked_position_rmse(p, t, k, c)
                   for w, p, t, k, c in zip(tr.weights, poss, net.poss_list, tr.keeps, tr.counts))
        loss.backward()
        return loss.detach()

    stream, graph = torch.cuda.Stream(), None
    for i in range(n):
        if eager_kernel_between:
            _ = batch.v_keep * 2.0                       # unrelated elementwise kernel on the current stream
        if not graphed:
            body()
        elif i < 3:                                      # warm-up on the capture stream
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                body()
            torch.cuda.current_stream().wait_stream(stream)
        else:
            if graph is None:
                for p in net.parameters():
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream):
                    body()
            graph.replay()
        if sync_between:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    out = [p.grad.detach().clone() for p in net.parameters() if p.grad is not None]
    del graph
    gc.collect()
    return out


def rel(a, b):
    return float((sum(((x.double() - y.double()) ** 2).sum() for x, y in zip(a, b))
                  / sum((x.double() ** 2).sum() for x in a)) ** 0.5)


def main():
    ref = grads(False, False, False)
    for kernel in (False, True):
        for sync in (False, True):
            print(f"eager kernel between replays: {kernel!s:5}  synchronize between replays: {sync!s:5}  "
                  f"relative L2 of the gradients vs eager: {rel(ref, grads(True, kernel, sync)):.3e}", flush=True)


if __name__ == "__main__":
    main()
