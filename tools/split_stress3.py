"""Determinism of the fp32 SGCN forward + backward under contention (several processes on one GPU): debug aid."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import golden_util as GU
from semigcn_amd import capi, synth, train
from semigcn_amd.networks import SingleScaleGCN
import bench
ENG, REPS = int(sys.argv[1]), int(sys.argv[2])
mesh = synth.torus_mesh(500, 400)
dev = torch.device("cuda:0")
capi.tuning_set(capi.TUNE_F32_ENGINE, ENG)
ref = SingleScaleGCN(dev)
GU.fill_state(ref, seed=77)
ref.to(dev).train()
batch = bench.build_mesh_batch(mesh, dev, n_masks=2)
rt = train.SGCNTrainer(ref, batch, accumulate=1000)
dm = batch.v_keep * batch.dummy_masks[:, :1]
first = None
bad_f = bad_b = 0
for i in range(REPS):
    ref.zero_grad(set_to_none=True)
    pos = ref(batch.data, dm)
    loss = rt.loss(pos)
    loss.backward()
    torch.cuda.synchronize()
    g = torch.cat([p.grad.reshape(-1) for p in ref.parameters() if p.grad is not None])
    if first is None:
        first = (pos.detach().clone(), g.clone())
    else:
        bad_f += 0 if torch.equal(pos.detach(), first[0]) else 1
        bad_b += 0 if torch.equal(g, first[1]) else 1
print(ENG, "forward mismatches", bad_f, "backward mismatches", bad_b, "of", REPS - 1)
