#!/usr/bin/env python3
"""A/B of the compute-bound feature x weight products of an SGCN iteration (K x N > 100 K weight elements) at V rows,
bf16: the persistent 256 x 256-tile kernel (csrc/gemm_mfma256.hip) against the 128-row-tile kernel (csrc/gemm_mfma.hip)
and the BLAS library (hipBLASLt through torch.addmm), variants interleaved in ONE process (CDNA guide rule 24), random
operands (rule 25: zero-filled operands read high).

    python tools/gemm256_bench.py [--V 1000000] [--rounds 5] [--json out.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semigcn_amd import capi  # noqa: E402

SHAPES = [  # (what, N, K)
    ("L5/L8 fwd  [V,768]x[768,256]", 256, 768),
    ("L6 fwd / L7 dx  [V,768]x[768,512]", 512, 768),
    ("L5/L8 dT  [V,256]x[256,768]", 768, 256),
    ("L6 dT / L7 fwd  [V,512]x[512,768]", 768, 512),
    ("L4 fwd  [V,384]x[384,256]", 256, 384),
    ("L9 dx  [V,384]x[384,256] (as L4)", 256, 384),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--V", type=int, default=1_000_000)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--json", default=None)
    ap.add_argument("--nt-only", action="store_true")
    ap.add_argument("--only", default=None, help="comma-separated variants to run (tile256,tile128,blas)")
    ap.add_argument("--shapes", default=None, help="comma-separated indices into SHAPES")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M = a.V
    res = []
    for name, N, K in ([SHAPES[int(i)] for i in a.shapes.split(",")] if a.shapes else SHAPES[:5]):
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        B = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        bias16 = bias.to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

        def own(tile):
            def run():
                capi.tuning_set(capi.TUNE_GEMM_TILE, tile)
                capi.gemm_nt(A, B, bias, out=out)
                capi.tuning_set(capi.TUNE_GEMM_TILE, 0)
            return run
        variants = {"tile256": own(3), "tile128": own(1), "blas": lambda: torch.addmm(bias16, A, B.t(), out=out)}
        if a.only:
            variants = {k: v for k, v in variants.items() if k in a.only.split(",")}
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
        (own(3) if "tile256" in variants else list(variants.values())[0])()
        ref = A[:4096].float() @ B.float().t() + bias
        err = float((out[:4096].float() - ref).abs().max() / ref.abs().max())
        flops, byts = 2.0 * M * N * K, (M * K + M * N) * 2.0
        row = {"product": name, "M": M, "N": N, "K": K, "rel_err_256": err}
        for k in variants:
            ms = float(np.median(times[k]))
            row[k] = {"ms": round(ms, 4), "TFLOPs": round(flops / ms / 1e9, 1), "mfma_frac": round(flops / ms / 1e9 / 2500.0, 4),
                      "hbm_frac": round(byts / ms / 1e6 / 8000.0, 4)}
        res.append(row)
        print(f"{name:40s} " + "  ".join(f"{k}: {row[k]['ms']:.3f} ms ({row[k]['mfma_frac']:.3f} mfma, {row[k]['hbm_frac']:.3f} hbm)"
                                         for k in variants) + f"  err {err:.1e}", flush=True)
        del A, out
    # the weight gradients of the same layers: out[N, Kp] = A[M, N]^T B[M, Kp]
    from semigcn_amd import functional as F_sg
    for name, N, Kp in () if a.nt_only else (("L5/L8 dW  [256,V]x[V,768]", 256, 768), ("L6 dW  [512,V]x[V,768]", 512, 768), ("L7 dW  [768,V]x[V,512]", 768, 512)):
        A = torch.randn(M, N, device=dev).to(torch.bfloat16)
        B = torch.randn(M, Kp, device=dev).to(torch.bfloat16)

        def own(tile):
            def run():
                capi.tuning_set(capi.TUNE_GEMM_TILE, tile)
                r = capi.gemm_tn(A, B)
                capi.tuning_set(capi.TUNE_GEMM_TILE, 0)
                return r
            return run

        def blas():
            old = F_sg.USE_MFMA_GEMM
            F_sg.USE_MFMA_GEMM = False
            try:
                return F_sg._weight_grad(A, B)
            finally:
                F_sg.USE_MFMA_GEMM = old
        variants = {"tile256": own(3), "tile128": own(1), "blas": blas}
        if a.only:
            variants = {k: v for k, v in variants.items() if k in a.only.split(",")}
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
        err = float((own(3)() - blas()).abs().max() / blas().abs().max())
        for k in ("tile256", "tile128", "blas"):
            row_fill = k in variants
        flops = 2.0 * M * N * Kp
        row = {"product": name, "M": M, "N": N, "K": Kp, "rel_diff_256_vs_blas": err}
        for k in variants:
            ms = float(np.median(times[k]))
            row[k] = {"ms": round(ms, 4), "TFLOPs": round(flops / ms / 1e9, 1), "mfma_frac": round(flops / ms / 1e9 / 2500.0, 4)}
        res.append(row)
        print(f"{name:40s} " + "  ".join(f"{k}: {row[k]['ms']:.3f} ms ({row[k]['mfma_frac']:.3f} mfma)" for k in variants)
              + f"  diff {err:.1e}", flush=True)
        del A, B
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        json.dump({"V": M, "rounds": a.rounds, "reps": a.reps, "products": res}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
