#!/usr/bin/env python3
"""A/B of the HBM-bound products with N = 192 / 384 columns at V rows (bf16): 128 x 192 output tiles (the shipped choice,
SG_TUNE_GEMM_TILE = 0) against 128 x 128 tiles with a part-empty last column tile (= 4; N = 384: = 5 forces the 192-column tiles, which lost), interleaved in one process, random
operands.    python tools/gemm192_bench.py [--V 1000000] [--json out.json]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semigcn_amd import capi  # noqa: E402

SHAPES = [("[V,128]x[128,192]", 192, 128), ("[V,256]x[256,384]", 384, 256), ("[V,64]x[64,192]", 192, 64), ("[V,128]x[128,384]", 384, 128)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--V", type=int, default=1_000_000)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M, res = a.V, []
    for name, N, K in SHAPES:
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        B = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        outs = {}
        times = {"tile192": [], "tile128": []}
        for rnd in range(a.rounds + 1):
            for key, tile in (("tile192", 5), ("tile128", 4)):
                out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
                capi.tuning_set(capi.TUNE_GEMM_TILE, tile)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    capi.gemm_nt(A, B, bias, out=out)
                e1.record()
                torch.cuda.synchronize()
                capi.tuning_set(capi.TUNE_GEMM_TILE, 0)
                if rnd:
                    times[key].append(e0.elapsed_time(e1) / a.reps)
                outs[key] = out
        same = bool(torch.equal(outs["tile192"], outs["tile128"]))
        byts = 2.0 * M * (K + N)
        row = {"product": name, "M": M, "N": N, "K": K, "bit_identical": same}
        for k, v in times.items():
            ms = float(np.median(v))
            row[k] = {"ms": round(ms, 4), "TBs": round(byts / ms / 1e9, 3)}
        res.append(row)
        print(json.dumps(row))
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        json.dump({"V": M, "products": res}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
