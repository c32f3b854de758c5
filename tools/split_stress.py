"""Determinism + error sweep of the split-bf16 fp32 products on the layer shapes of one mesh size (debug aid)."""
import sys, os, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semigcn_amd import capi
V = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = "cuda:0"
CH = (4, 16, 32, 64, 128, 256, 256, 512, 256, 256, 128, 64, 32, 16)
g = torch.Generator(device=dev).manual_seed(0)
bad = 0
for i in range(13):
    ci, co = CH[i], CH[i + 1]
    post = co < ci
    shapes = [(co, 3 * ci)] if not post else [(3 * co, ci)]
    for (N, K) in shapes:
        A = torch.randn((V, K), device=dev, generator=g)
        W = torch.randn((N, K), device=dev, generator=g) * 0.05
        dH = torch.randn((V, N), device=dev, generator=g) * 1e-5
        for name, fn, ref in (
            ("nt", lambda: capi.gemm_nt_f32(A, W) if capi.gemm_nt_f32_supported(A, N) else None, lambda: A.double() @ W.double().t()),
            ("nn", lambda: capi.gemm_nt_f32(dH, W, w_is_kn=True) if capi.gemm_nt_f32_supported(dH, K) else None, lambda: dH.double() @ W.double()),
            ("tn", lambda: capi.gemm_tn_f32(dH, A) if capi.gemm_tn_f32_supported(dH, A) else None, lambda: dH.double().t() @ A.double()),
        ):
            o = fn()
            if o is None:
                print(f"L{i} {name} N={N} K={K}: not served"); continue
            r = ref()
            rel = float((o.double() - r).norm() / r.norm())
            rowerr = ((o.double() - r).abs().amax(1) / r.abs().amax().clamp_min(1e-300))
            det = all(torch.equal(fn(), o) for _ in range(REPS))
            flag = "" if (rel < 1e-6 and det) else "   <<<<<<"
            bad += flag != ""
            print(f"L{i} {name} V={V} N={N} K={K}: rel-L2 {rel:.2e} worst-row {float(rowerr.max()):.2e} at {int(rowerr.argmax())} deterministic={det}{flag}")
print("bad", bad)
