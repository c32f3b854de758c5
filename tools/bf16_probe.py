#!/usr/bin/env python3
"""Where do the bf16-feature HIP path and the bf16-storage oracle part ways?  Per block: conv output and block output,
relative L2 and the fraction of elements that are not bit-equal (diagnostic for tests/test_gpu_config_parity.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import golden_util as GU  # noqa: E402
from oracle import bf16 as OB  # noqa: E402
from semigcn_amd import functional as F_sg, synth  # noqa: E402
from semigcn_amd.networks import CHANNELS, SingleScaleGCN  # noqa: E402

DEV = "cuda:0"
post = "--pre" not in sys.argv
F_sg.AGGREGATE_AFTER_GEMM_WHEN_NARROWING = post
m = synth.octahedron_sphere(3) if "--torus" not in sys.argv else synth.torus_mesh(100, 50)
net = SingleScaleGCN(DEV)
GU.fill_state(net, seed=60)
state0 = {k: v.clone() for k, v in net.state_dict().items()}
net.to(DEV).train()
net.set_feature_dtype(torch.bfloat16)


class D:
    z1 = torch.from_numpy(m.z1).to(DEV)
    x_pos = torch.from_numpy(m.x_pos).to(DEV)
    edge_index = torch.from_numpy(m.edge_index).to(DEV)


rank = net._layout(D)[2]
V = m.num_vertices
blas = []
for i in range(13):
    cin, cout = CHANNELS[i], CHANNELS[i + 1]
    wshape = (3 * cout, cin) if (post and cout < cin) else (cout, 3 * cin)
    a = torch.empty((V, wshape[1]), dtype=torch.bfloat16, device=DEV)
    if not F_sg._mfma_ok(a, torch.empty(wshape, dtype=torch.bfloat16, device=DEV), wshape[0]) and not F_sg._thin_ok(a, wshape[0], wshape[1]):
        blas.append(i)
print("layers whose product goes to the BLAS library:", blas)
ora = OB.SGCNOracleBf16(post_when_narrowing=post, bias_bf16_layers=blas)
ora.load_state_dict(state0)
ora.train()
hc, hb, oc, ob, hin, oin = [], [], [], [], [], []
def keep(into_in, into_out, unpermute):
    def hook(mod, i, o):
        f = (lambda t: t.detach().float().index_select(0, rank).cpu()) if unpermute else (lambda t: t.detach())
        if into_in is not None:
            into_in.append(f(i[0]))
        into_out.append(f(o))
    return hook


for b in net.blocks:
    b.module_0.register_forward_hook(keep(hin, hc, True))
    b.register_forward_hook(keep(None, hb, True))
for b in ora.blocks:
    b.module_0.register_forward_hook(keep(oin, oc, False))
    b.register_forward_hook(keep(None, ob, False))
with torch.no_grad():
    ph = net(D, None)
    po = ora(torch.from_numpy(m.z1), torch.from_numpy(m.x_pos), torch.from_numpy(m.edge_index), None)
for i in range(13):
    print(f"block {i:2d} {CHANNELS[i]:3d}->{CHANNELS[i+1]:3d}  in {GU.rel_l2(hin[i], oin[i]):.2e} ({float((hin[i] != oin[i]).float().mean()):.4f})"
          f"  conv {GU.rel_l2(hc[i], oc[i]):.2e} ({float((hc[i] != oc[i]).float().mean()):.4f})"
          f"  block {GU.rel_l2(hb[i], ob[i]):.2e} ({float((hb[i] != ob[i]).float().mean()):.4f})")
xp = torch.from_numpy(m.x_pos)
print("offsets", GU.rel_l2(ph.cpu() - xp, po - xp))
# block 0 in isolation on the oracle's own input: conv pieces
