"""Per-kernel streaming rate of the fused BatchNorm+LeakyReLU kernels (csrc/bn_act.hip) at V = 1 M,
HIP-event timed, next to a torch 2-read/1-write add of the same size.

    python tools/bn_bench.py [--V 1000000]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semigcn_amd import capi  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--V", type=int, default=1_000_000)
    ap.add_argument("--rows", type=int, default=-1, help="SG_TUNE_BN_ROWS (0 by shape, 1 contiguous row ranges, 2 strided); -1: the library's default")
    a = ap.parse_args()
    if a.rows >= 0:
        capi.tuning_set(capi.TUNE_BN_ROWS, a.rows)
    V = a.V
    rows = []
    for dt in (torch.float32, torch.bfloat16):
        s = 4 if dt == torch.float32 else 2
        for C in (16, 32, 64, 128, 256, 512):
            X = torch.randn(V, C, device="cuda").to(dt)
            dA = torch.randn(V, C, device="cuda").to(dt)
            vec = [torch.rand(C, device="cuda") + 0.5 for _ in range(7)]
            out = torch.empty_like(X)
            nb = V * C * s
            r = {"dtype": str(dt).replace("torch.", ""), "C": C}
            r["moments_GBps"] = nb / timed(lambda: capi.col_moments(X)) / 1e6
            r["apply_GBps"] = 2 * nb / timed(lambda: capi.scale_shift_act(X, vec[0], vec[1], 0.01, out=out)) / 1e6
            r["bwd_reduce_GBps"] = 2 * nb / timed(lambda: capi.bn_act_bwd_reduce(dA, X, *vec[:4], 0.01)) / 1e6
            r["bwd_apply_GBps"] = 3 * nb / timed(lambda: capi.bn_act_bwd_apply(dA, X, *vec, 0.01)) / 1e6
            r["torch_add_GBps"] = 3 * nb / timed(lambda: torch.add(X, dA, out=out)) / 1e6
            r["torch_copy_GBps"] = 2 * nb / timed(lambda: out.copy_(X)) / 1e6
            rows.append({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()})
            print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
