#!/usr/bin/env python3
"""Times the three dense GEMM shapes of a ChebConv layer at V = 1 M (hipBLASLt through torch) and
split-K formulations of the weight-gradient product dW = dOut^T T (a reduction over all V)."""
import torch

V = 1_000_000
dev = "cuda:0"


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def dw_split(dout, T, S):
    Vs = (dout.shape[0] // S) * S
    a = dout[:Vs].view(S, Vs // S, -1).transpose(1, 2)
    b = T[:Vs].view(S, Vs // S, -1)
    if dout.dtype == torch.float32:
        out = torch.bmm(a, b).sum(0)
    else:
        out = torch.bmm(a, b, out_dtype=torch.float32).sum(0)
    if Vs < dout.shape[0]:
        out = out + (dout[Vs:].t().float() @ T[Vs:].float())
    return out


for dtype in (torch.float32, torch.bfloat16):
    for cin, cout in ((256, 256), (512, 256), (256, 512), (128, 256), (64, 128), (32, 64)):
        T = torch.randn(V, 3 * cin, device=dev).to(dtype)
        dout = torch.randn(V, cout, device=dev).to(dtype)
        W = torch.randn(cout, 3 * cin, device=dev).to(dtype)
        fl = 2.0 * V * 3 * cin * cout
        f_fwd = t(lambda: T @ W.t())
        f_dx = t(lambda: dout @ W)
        if dtype == torch.float32:
            f_dw = t(lambda: dout.t() @ T)
        else:
            f_dw = t(lambda: torch.mm(dout.t(), T, out_dtype=torch.float32))
        line = f"{str(dtype)[6:]:9s} {cin:4d}->{cout:4d}  fwd {f_fwd:6.2f} ms ({fl/f_fwd/1e9:6.0f} TF)  dX {f_dx:6.2f} ({fl/f_dx/1e9:6.0f} TF)  dW {f_dw:6.2f} ({fl/f_dw/1e9:6.0f} TF) |"
        ref = (dout.t().float() @ T.float()) if dtype != torch.float32 else dout.t() @ T
        for S in (8, 32, 128, 512):
            ms = t(lambda: dw_split(dout, T, S))
            err = float((dw_split(dout, T, S) - ref).abs().max() / ref.abs().max())
            line += f" S={S}: {ms:5.2f} ({fl/ms/1e9:5.0f} TF, {err:.0e})"
        print(line, flush=True)
        del T, dout, W
