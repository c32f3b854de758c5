#!/usr/bin/env python3
"""Experiment: capture one SGCN training iteration into a hipGraph (torch.cuda.CUDAGraph) and replay it.
Usage: graph_probe.py NUxNV [flags...]   flags: noreorder nosplit nowiden nopost nofusedloss eagerloss"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv_backup = list(sys.argv)
mesh_arg, flags = sys.argv[1], set(sys.argv[2:])
sys.argv = ["bench.py"]
import bench
from semigcn_amd import functional as F_sg, synth, train
from semigcn_amd.networks import SingleScaleGCN

dev = torch.device("cuda:0")
nu, nv = map(int, mesh_arg.split("x"))
mesh = synth.torus_mesh(nu, nv)
batch = bench.build_mesh_batch(mesh, dev, 5)
if "nosplit" in flags:
    F_sg.weight_grad = lambda dout, T: F_sg._mm_f32_out(dout.t(), T)
if "nopost" in flags:
    F_sg.AGGREGATE_AFTER_GEMM_WHEN_NARROWING = False
if "nowiden" in flags:
    F_sg._adopt_wide = lambda x, K: None
torch.manual_seed(314)
model = SingleScaleGCN(dev, reorder="noreorder" not in flags).to(dev)
tr = train.SGCNTrainer(model, batch)
static_dm = torch.ones_like(batch.v_keep)

def step():
    model.train()
    pos = model(batch.data, static_dm)
    if "eagerloss" in flags:
        loss = (pos ** 2).mean()
    else:
        loss = tr.loss(pos)
    loss.backward()
    return loss.detach()

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        eager = step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("eager loss", float(eager), flush=True)
for p in model.parameters():
    if p.grad is not None:
        p.grad.zero_()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = step()
torch.cuda.synchronize()
print("captured", flush=True)
for i in range(3):
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, "loss", float(static_loss), flush=True)
t = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print(f"graph replay {1e3 * (time.perf_counter() - t) / 10:.2f} ms/iteration", flush=True)
t = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print(f"eager        {1e3 * (time.perf_counter() - t) / 10:.2f} ms/iteration", flush=True)
