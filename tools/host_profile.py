#!/usr/bin/env python3
"""cProfile of the host side of SGCN training iterations on a small mesh (CPU-launch-bound regime)."""
import cProfile, os, pstats, sys, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MESH = sys.argv[1] if len(sys.argv) > 1 else "250x200"
sys.argv = ["bench.py", "--mesh", MESH]
import bench
from semigcn_amd import synth, train
from semigcn_amd.networks import SingleScaleGCN
dev = torch.device("cuda:0")
mesh = synth.torus_mesh(*map(int, MESH.split("x")))
batch = bench.build_mesh_batch(mesh, dev, 5)
model = SingleScaleGCN(dev).to(dev)
tr = train.SGCNTrainer(model, batch)
for _ in range(5): tr.iteration_step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): tr.iteration_step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
