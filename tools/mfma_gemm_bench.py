#!/usr/bin/env python3
"""Per-layer timing of the dense products of an SGCN iteration at V vertices, bf16 features:
sg_gemm_nt (csrc/gemm_mfma.hip, with and without the BatchNorm-moments epilogue) against the BLAS library
(hipBLASLt through torch.addmm / torch.mm), variants interleaved in ONE process (CDNA guide rule 24).

    python tools/mfma_gemm_bench.py [--V 1000000] [--rounds 5] [--json out.json]

Per product: algorithmic bytes = (M*K + M*N) * 2 (A read once, C written once; B is L2-resident), flops = 2*M*N*K;
`hbm_frac` = bytes / time / 8 TB/s, `mfma_frac` = flops / time / 2.5 PFLOP/s (dense bf16 peak) -- whichever is
larger bounds that shape."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semigcn_amd import capi  # noqa: E402

SGCN = (4, 16, 32, 64, 128, 256, 256, 512, 256, 256, 128, 64, 32, 16)


def products():
    """(name, N, K) of every dense product of one SGCN iteration that the MFMA kernel can take (M = V)."""
    out = []
    for i in range(13):
        cin, cout = SGCN[i], SGCN[i + 1]
        if cout >= cin:      # aggregate, then [V,3Cin] x [3Cin,Cout]
            out.append((f"L{i} fwd  [V,{3*cin}]x[{3*cin},{cout}]", cout, 3 * cin))
            out.append((f"L{i} dT   [V,{cout}]x[{cout},{3*cin}]", 3 * cin, cout))
        else:                # [V,Cin] x [Cin,3Cout], then Clenshaw aggregation
            out.append((f"L{i} fwd  [V,{cin}]x[{cin},{3*cout}]", 3 * cout, cin))
            out.append((f"L{i} dx   [V,{3*cout}]x[{3*cout},{cin}]", cin, 3 * cout))
    return [(n, N, K) for n, N, K in out if N % 8 == 0 and K % 8 == 0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--V", type=int, default=1_000_000)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M = a.V
    res = []
    tot = {"mfma": 0.0, "mfma128": 0.0, "mfma64x256": 0.0, "mfma+moments": 0.0, "blas": 0.0, "best": 0.0, "ideal": 0.0}
    for name, N, K in products():
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        B = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        bias16 = bias.to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        def own(tile, moments=False):
            def run():
                capi.tuning_set(capi.TUNE_GEMM_TILE, tile)
                capi.gemm_nt(A, B, bias, out=out, moments=moments)
                capi.tuning_set(capi.TUNE_GEMM_TILE, 0)
            return run
        variants = {
            "mfma": own(0),                       # the shipped tile choice
            "mfma128": own(1),                    # 128-row tiles always
            "mfma64x256": own(2),                 # 64 x 256 tiles wherever N > 64
            "mfma+moments": own(0, True),
            "blas": lambda: torch.addmm(bias16, A, B.t(), out=out),
        }
        times = {k: [] for k in variants}
        for rnd in range(a.rounds + 1):
            for k, fn in variants.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[k].append(e0.elapsed_time(e1) / a.reps)
        ref = torch.addmm(bias16, A, B.t()).float()
        err = float((capi.gemm_nt(A, B, bias).float() - ref).abs().max() / ref.abs().max())
        byts, flops = (M * K + M * N) * 2.0, 2.0 * M * N * K
        ideal = max(byts / 8e12, flops / 2.5e15) * 1e3
        rec = {"product": name, "M": M, "N": N, "K": K, "bytes": byts, "flops": flops, "ideal_ms": round(ideal, 4),
               "max_rel_diff_vs_blas": err}
        line = f"{name:34s}"
        for k in variants:
            med = float(np.median(times[k]))
            rec[k + "_ms"] = round(med, 4)
            rec[k + "_hbm_frac"] = round(byts / med / 1e-3 / 8e12, 3)
            rec[k + "_mfma_frac"] = round(flops / med / 1e-3 / 2.5e15, 3)
            tot[k] += med
            line += f"  {k} {med:7.3f} ms (hbm {rec[k + '_hbm_frac']:.2f} mfma {rec[k + '_mfma_frac']:.2f})"
        tot["ideal"] += ideal
        tot["best"] += min(rec[k + "_ms"] for k in ("mfma", "blas"))
        res.append(rec)
        print(line + f"  diff {err:.1e}", flush=True)
        del A, B, out
    print("totals (one forward + input-gradient pass of every supported layer): " +
          "  ".join(f"{k} {v:.2f} ms" for k, v in tot.items()), flush=True)
    # ---- weight gradients dW = dOut^T T: own transposing-read kernel against the slab-batched BLAS formulation
    from semigcn_amd import functional as F_sg
    tot_dw = {"mfma_tn": 0.0, "blas_slabs": 0.0}
    dw = []
    for i in range(13):
        cin, cout = SGCN[i], SGCN[i + 1]
        N, Kp = (cout, 3 * cin) if cout >= cin else (3 * cout, cin)
        if N % 8 or Kp % 8:
            continue
        A = torch.randn(M, N, device=dev).to(torch.bfloat16)
        B = torch.randn(M, Kp, device=dev).to(torch.bfloat16)

        def blas():
            F_sg.USE_MFMA_GEMM = False
            try:
                return F_sg.weight_grad(A, B)
            finally:
                F_sg.USE_MFMA_GEMM = True
        variants = {"mfma_tn": lambda: capi.gemm_tn(A, B), "blas_slabs": blas}
        times = {k: [] for k in variants}
        for rnd in range(a.rounds + 1):
            for k, fn in variants.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    times[k].append(e0.elapsed_time(e1) / a.reps)
        ref = blas()
        err = float((capi.gemm_tn(A, B) - ref).abs().max() / ref.abs().max())
        byts, flops = (M * N + M * Kp) * 2.0, 2.0 * M * N * Kp
        rec = {"product": f"L{i} dW [{N},V]x[V,{Kp}]", "N": N, "Kp": Kp, "bytes": byts, "flops": flops, "max_rel_diff": err}
        line = f"{rec['product']:30s}"
        for k in variants:
            med = float(np.median(times[k]))
            rec[k + "_ms"] = round(med, 4)
            rec[k + "_hbm_frac"] = round(byts / med / 1e-3 / 8e12, 3)
            rec[k + "_mfma_frac"] = round(flops / med / 1e-3 / 2.5e15, 3)
            tot_dw[k] += med
            line += f"  {k} {med:7.3f} ms (hbm {rec[k + '_hbm_frac']:.2f} mfma {rec[k + '_mfma_frac']:.2f})"
        dw.append(rec)
        print(line + f"  diff {err:.1e}", flush=True)
        del A, B
    print("weight-gradient totals: " + "  ".join(f"{k} {v:.2f} ms" for k, v in tot_dw.items()), flush=True)
    if a.json:
        json.dump({"V": M, "totals_ms": tot, "products": res, "weight_gradient_totals_ms": tot_dw, "weight_gradients": dw},
                  open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
