#!/usr/bin/env python3
"""The thin products of an SGCN iteration at V rows (csrc/thin_gemm.hip): ms per launch and GB/s of the bytes they move.
    python tools/thin_bench.py [--V 1000000] [--json out.json]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semigcn_amd import capi  # noqa: E402

SHAPES = [("4 -> 16 forward [V,12]x[12,16]", 16, 12, torch.bfloat16), ("4 -> 16 input gradient [V,16]x[16,12]", 12, 16, torch.bfloat16),
          ("Linear(16, 3) forward [V,16]x[16,3]", 3, 16, torch.float32), ("Linear(16, 3) input gradient [V,3]x[3,16]", 16, 3, torch.float32),
          ("4 -> 16 forward, fp32 rows", 16, 12, torch.float32)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--V", type=int, default=1_000_000)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    res = []
    for name, N, K, dt in SHAPES:
        X = torch.randn(a.V, K, device=dev).to(dt)
        W = torch.randn(N, K, device=dev)
        b = torch.randn(N, device=dev)
        out = torch.empty(a.V, N, device=dev, dtype=dt)
        ts = []
        for rnd in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                capi.thin_nt(X, W, b, out=out)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                ts.append(e0.elapsed_time(e1) / 10)
        ms = float(np.median(ts))
        ref = X.float() @ W.t() + b
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        es = 4 if dt == torch.float32 else 2
        row = {"product": name, "V": a.V, "N": N, "K": K, "dtype": str(dt), "ms": round(ms, 4),
               "GBs": round(a.V * (N + K) * es / ms / 1e6, 1), "max_rel_err": err}
        res.append(row)
        print(json.dumps(row))
    for name, N, K, dt in (("4 -> 16 weight gradient [V,16]^T [V,12]", 16, 12, torch.bfloat16), ("Linear(16, 3) weight gradient [V,3]^T [V,16]", 3, 16, torch.float32)):
        A = torch.randn(a.V, N, device=dev).to(dt)
        B = torch.randn(a.V, K, device=dev).to(dt)
        ts = []
        for rnd in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                out = capi.thin_tn(A, B)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                ts.append(e0.elapsed_time(e1) / 10)
        ms = float(np.median(ts))
        ref = A.double().t() @ B.double()
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        es = 4 if dt == torch.float32 else 2
        row = {"product": name, "V": a.V, "N": N, "K": K, "dtype": str(dt), "ms (both launches + the allocation)": round(ms, 4),
               "GBs": round(a.V * (N + K) * es / ms / 1e6, 1), "max_rel_err": err}
        res.append(row)
        print(json.dumps(row))
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
