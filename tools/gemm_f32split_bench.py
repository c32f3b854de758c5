"""fp32 products on the bf16 matrix cores (csrc/gemm_split.hip) against the BLAS library (torch.mm -> hipBLASLt):
correctness (small integers bit for bit; random data against float64, beside the library's own error) and time per shape
at V vertices, interleaved in one process.  Writes one JSON object per line.

    python tools/gemm_f32split_bench.py [--V 1000000] [--out gpurun_out/gemm_f32split_bench.json] [--quick]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semigcn_amd import capi  # noqa: E402


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in evs)
    return ts[len(ts) // 2]


def check(dev):
    g = torch.Generator(device="cpu").manual_seed(5)
    rows = []
    for (M, N, K) in [(1000, 256, 64), (4133, 384, 96), (2048, 192, 128), (777, 512, 256), (5000, 64, 768)]:
        A = torch.randint(-8, 9, (M, K), generator=g).float().to(dev)
        W = torch.randint(-8, 9, (N, K), generator=g).float().to(dev)
        bias = torch.randint(-8, 9, (N,), generator=g).float().to(dev)
        ref = (A.double() @ W.double().t() + bias.double()).float()
        if not capi.gemm_nt_f32_supported(A, N):
            rows.append({"check": "nt int", "shape": [M, N, K], "supported": False})
            continue
        out = capi.gemm_nt_f32(A, W, bias)
        out2 = capi.gemm_nt_f32(A, W.t().contiguous(), bias, w_is_kn=True)
        rows.append({"check": "nt int", "shape": [M, N, K], "exact": bool(torch.equal(out, ref)), "exact_kn": bool(torch.equal(out2, ref))})
        Ar = torch.randn((M, K), generator=g).to(dev)
        Wr = (torch.randn((N, K), generator=g) * 0.1).to(dev)
        r64 = Ar.double() @ Wr.double().t()
        den = (Ar.double().abs() @ Wr.double().abs().t())
        e_own = ((capi.gemm_nt_f32(Ar, Wr).double() - r64).abs() / den).max().item()
        e_lib = (((Ar @ Wr.t()).double() - r64).abs() / den).max().item()
        rows.append({"check": "nt random", "shape": [M, N, K], "err_own": e_own, "err_blas": e_lib})
    for (M, N, Kp) in [(4096, 256, 128), (10000, 64, 192), (33333, 512, 384), (8191, 128, 96)]:
        A = torch.randint(-4, 5, (M, N), generator=g).float().to(dev)
        B = torch.randint(-4, 5, (M, Kp), generator=g).float().to(dev)
        ref = (A.double().t() @ B.double()).float()
        if not capi.gemm_tn_f32_supported(A, B):
            rows.append({"check": "tn int", "shape": [M, N, Kp], "supported": False})
            continue
        out = capi.gemm_tn_f32(A, B)
        rows.append({"check": "tn int", "shape": [M, N, Kp], "exact": bool(torch.equal(out, ref))})
        Ar = torch.randn((M, N), generator=g).to(dev)
        Br = torch.randn((M, Kp), generator=g).to(dev)
        r64 = Ar.double().t() @ Br.double()
        den = Ar.double().abs().t() @ Br.double().abs()
        e_own = ((capi.gemm_tn_f32(Ar, Br).double() - r64).abs() / den).max().item()
        e_lib = (((Ar.t() @ Br).double() - r64).abs() / den).max().item()
        rows.append({"check": "tn random", "shape": [M, N, Kp], "err_own": e_own, "err_blas": e_lib})
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--V", type=int, default=1_000_000)
    ap.add_argument("--out", default="gpurun_out/gemm_f32split_bench.json")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--engine", type=int, default=0, help="SG_TUNE_F32_ENGINE value (kernel variants; see include/semigcn.h)")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--only", type=int, default=-1, help="run only product number N of the sequence (0 = nt of the first layer, 1 = nn, 2 = tn, 3 = ...)")
    ap.add_argument("--no-blas", action="store_true", help="skip the BLAS library's timing")
    a = ap.parse_args()
    capi.tuning_set(capi.TUNE_F32_ENGINE, a.engine)
    dev = torch.device("cuda:0")
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    out = open(a.out, "w")

    def emit(r):
        print(json.dumps(r), flush=True)
        out.write(json.dumps(r) + "\n")
        out.flush()

    if not a.no_check:
        for r in check(dev):
            emit(r)
    V = a.V
    # (Cin, Cout) of the SGCN's ChebConv layers (K = 3); aggregate-first layers multiply [V, 3 Cin] x [3 Cin, Cout]
    layers = [(256, 512), (256, 256), (128, 256), (64, 128)] if a.quick else [(256, 512), (256, 256), (128, 256), (64, 128), (32, 64), (16, 32)]
    seq = 0
    for (ci, co) in layers:
        K, N = 3 * ci, co
        T = torch.randn((V, K), device=dev)
        Wc = torch.randn((N, K), device=dev) * 0.05
        H = torch.empty((V, N), device=dev)
        dH = torch.randn((V, N), device=dev)
        dT = torch.empty((V, K), device=dev)
        flop = 2.0 * V * N * K
        for name, own, lib, sup in (
            ("nt [V,%d]x[%d,%d]" % (K, K, N), lambda: capi.gemm_nt_f32(T, Wc, None, out=H), lambda: torch.mm(T, Wc.t(), out=H),
             capi.gemm_nt_f32_supported(T, N)),
            ("nn [V,%d]x[%d,%d]" % (N, N, K), lambda: capi.gemm_nt_f32(dH, Wc, None, out=dT, w_is_kn=True), lambda: torch.mm(dH, Wc, out=dT),
             capi.gemm_nt_f32_supported(dH, K)),
            ("tn [%d,V]x[V,%d]" % (N, K), lambda: capi.gemm_tn_f32(dH, T), lambda: torch.mm(dH.t(), T), capi.gemm_tn_f32_supported(dH, T)),
        ):
            seq += 1
            if a.only >= 0 and a.only != seq - 1:
                continue
            row = {"product": name, "V": V, "gflop": flop / 1e9, "engine": a.engine}
            t_lib = float("nan")
            if not a.no_blas:
                t_lib = timed(lib, a.reps)
                row["blas_ms"] = t_lib
                row["blas_tflops"] = flop / t_lib / 1e9
            if sup:
                t_own = timed(own, a.reps)
                row["own_ms"] = t_own
                row["own_tflops_fp32_equiv"] = flop / t_own / 1e9
                row["own_bf16_mfma_frac"] = 6 * flop / t_own / 1e9 / 2500.0
                row["speedup"] = t_lib / t_own
            else:
                row["supported"] = False
            emit(row)
        del T, H, dH, dT
    out.close()


if __name__ == "__main__":
    main()
