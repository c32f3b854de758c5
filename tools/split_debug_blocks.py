"""Where does the backward of the fp32 SGCN diverge between dense engines?  Per block: output and gradient-at-output under
SG_TUNE_F32_ENGINE = a against = b (debug aid)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import golden_util as GU
from semigcn_amd import capi, synth, train, functional as F_sg
F_sg.CHAIN_MAX_ROWS = 0      # every block its own call: module hooks see the blocks
from semigcn_amd.networks import SingleScaleGCN
from test_gpu_config_parity import _batch
ea, eb = int(sys.argv[1]), int(sys.argv[2])
m = synth.torus_mesh(250, 200)
batch = _batch(m, n_masks=1)

def run(engine):
    capi.tuning_set(capi.TUNE_F32_ENGINE, engine)
    net = SingleScaleGCN("cuda:0")
    GU.fill_state(net, seed=50)
    net.to("cuda:0").train()
    tr = train.SGCNTrainer(net, batch)
    ys, dys = {}, {}
    hooks = []
    for i, blk in enumerate(net.blocks):
        def fh(mod, args, out, i=i):
            ys[i] = out.detach().float().clone()
            if out.requires_grad:
                out.register_hook(lambda g, i=i: dys.__setitem__(i, g.detach().float().clone()))
        hooks.append(blk.register_forward_hook(fh))
    batch.data.z1.grad = None
    pos = net(batch.data, batch.v_keep * batch.dummy_masks[:, :1])
    loss = tr.loss(pos)
    loss.backward()
    torch.cuda.synchronize()
    g = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    return ys, dys, batch.data.z1.grad.detach().clone(), g

A = run(ea)
B = run(eb)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))
for i in range(13):
    print(f"block {i:2d}: out {rel(A[0][i], B[0][i]):.2e}   grad-at-output {rel(A[1][i], B[1][i]) if i in A[1] and i in B[1] else float('nan'):.2e}")
print("dz1", rel(A[2], B[2]))
for n in A[3]:
    if "module_1.bias" in n or "lins.0" in n:
        print(n, f"{rel(A[3][n], B[3][n]):.2e}")
print("---- per-row differences of the block outputs")
for i in (2, 6, 12):
    d = (A[0][i].double() - B[0][i].double()).abs().amax(1)
    ref = B[0][i].double().abs().amax()
    top = torch.topk(d, 8)
    print(f"block {i}: median row err {float(d.median() / ref):.2e}  max {float(d.max() / ref):.2e}  rows {top.indices.tolist()}  vals {[f'{float(v / ref):.1e}' for v in top.values]}")
gd = (A[1][12].double() - B[1][12].double()).abs().amax(1)
gr = B[1][12].double().abs().amax()
top = torch.topk(gd, 8)
print(f"grad at output 12: median row err {float(gd.median() / gr):.2e} max {float(gd.max() / gr):.2e} rows {top.indices.tolist()} vals {[f'{float(v / gr):.1e}' for v in top.values]}")
print("hist of row err / max grad:", torch.histc((gd / gr).log10().clamp(-12, 0).float(), bins=12, min=-12, max=0).tolist())
print("---- the faces at the outlier rows: L1 kink or thin triangle?")
net0 = SingleScaleGCN("cuda:0")
rank = None
faces = batch.faces
tfn = batch.target_fn
# positions of both runs: recompute from block-12 outputs is not possible here (processing order); rerun forward quickly
def positions(engine):
    capi.tuning_set(capi.TUNE_F32_ENGINE, engine)
    net = SingleScaleGCN("cuda:0"); GU.fill_state(net, seed=50); net.to("cuda:0").train()
    with torch.no_grad():
        return net(batch.data, batch.v_keep * batch.dummy_masks[:, :1]).double()
pa, pb = positions(ea), positions(eb)
def fnorm(p):
    a, b, c = p[faces[:, 0]], p[faces[:, 1]], p[faces[:, 2]]
    n = torch.linalg.cross(b - a, c - a, dim=1)
    l = n.norm(dim=1, keepdim=True)
    e = torch.stack([(b - a).norm(dim=1), (c - a).norm(dim=1), (c - b).norm(dim=1)], 1)
    return n / l, (l.squeeze(1) / (e.amax(1) ** 2))      # unit normal, sin-like thinness measure
na, tha = fnorm(pa); nb, thb = fnorm(pb)
d_a = (na - tfn.double()); d_b = (nb - tfn.double())
flips = ((d_a > 0) != (d_b > 0)) & (batch.f_keep.view(-1, 1) > 0)
print("faces whose L1 sign pattern differs between the two forwards:", int(flips.any(1).sum()), "components", int(flips.sum()))
fl = flips.any(1).nonzero().flatten()
for f in fl[:10].tolist():
    print(" face", f, "verts", faces[f].tolist(), "pred-target", [f"{float(v):+.2e}" for v in d_a[f]], "/", [f"{float(v):+.2e}" for v in d_b[f]], "thinness", f"{float(tha[f]):.2e}")
print("thinnest kept faces:", [f"{float(v):.1e}" for v in torch.topk(-tha[batch.f_keep.flatten() > 0], 5).values.neg()])
