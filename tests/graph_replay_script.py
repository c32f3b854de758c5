"""Run by tests/test_gpu_parity.py::test_graphed_training_matches_eager in a SUBPROCESS whose environment carries
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 from the start (the flag is read when the HIP runtime initialises).  Eight training
iterations (three eager warm-ups, the capture, four replays; Adam after the 5th) with a device synchronize and an
unrelated eager kernel between iterations -- the pattern that corrupts replays under ROCm 7.2's default -- must give
the losses, parameters and BatchNorm statistics of eight eager iterations."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import golden_util as GU  # noqa: E402
from semigcn_amd import meshprep, synth, train  # noqa: E402
from semigcn_amd.meshnet import MGCN  # noqa: E402
from semigcn_amd.networks import SingleScaleGCN  # noqa: E402

DEV = "cuda:0"
kind = sys.argv[1]
m = synth.torus_mesh(60, 40)
batch = bench.build_mesh_batch(m, torch.device(DEV), n_masks=3)


def build():
    if kind == "sgcn":
        net = SingleScaleGCN(DEV)
    else:
        smo = meshprep.DeviceMesh(m.x_pos, m.faces, DEV)
        net = MGCN(DEV, smo, meshprep.DeviceMesh(m.vs.astype(np.float32), m.faces, DEV), torch.from_numpy(m.v_mask))
        for mod in net.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
    GU.fill_state(net, seed=21)
    return net.to(DEV)


def run(capture):
    cls = train.SGCNTrainer if kind == "sgcn" else train.MGCNTrainer
    tr = cls(build(), batch, capture=capture)
    losses = []
    for _ in range(8):
        _ = batch.v_keep * 2.0                 # an unrelated eager kernel ...
        losses.append(float(tr.iteration_step()))
        torch.cuda.synchronize()               # ... and an idle GPU before the next replay
    assert (tr._graphed is not None and tr._graphed.graph is not None) == capture
    return losses, {k: v.clone() for k, v in tr.model.state_dict().items() if v.is_floating_point() and not v.is_sparse}


le, se = run(False)
lg, sg = run(True)
assert lg == le, (lg, le)                      # no atomics anywhere: bit-identical
for k in se:
    assert torch.equal(se[k], sg[k]), k
print("GRAPH_REPLAY_OK", kind, le[-1])
