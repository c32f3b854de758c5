"""Determinism under contention: nt variants (stagger on / off) and the bf16 256-tile kernel, many repetitions (debug aid)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semigcn_amd import capi
V, REPS, ENG = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
capi.tuning_set(capi.TUNE_F32_ENGINE, ENG)
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
res = {}
for (N, K) in [(192, 128), (128, 192), (256, 384), (384, 256), (256, 768), (96, 64), (64, 96)]:
    A = torch.randn((V, K), device=dev, generator=g)
    W = torch.randn((N, K), device=dev, generator=g) * 0.05
    o = capi.gemm_nt_f32(A, W)
    bad = sum(0 if torch.equal(capi.gemm_nt_f32(A, W), o) else 1 for _ in range(REPS))
    res[f"f32 nt N={N} K={K}"] = bad
for (N, K) in [(192, 128), (256, 384), (512, 768), (96, 64), (768, 512)]:
    A = torch.randn((V, N), device=dev, generator=g)
    B = torch.randn((V, K), device=dev, generator=g)
    if not capi.gemm_tn_f32_supported(A, B):
        continue
    o = capi.gemm_tn_f32(A, B)
    bad = sum(0 if torch.equal(capi.gemm_tn_f32(A, B), o) else 1 for _ in range(REPS))
    res[f"f32 tn N={N} Kp={K}"] = bad
for (N, K) in [(256, 768), (512, 768), (768, 256)]:
    A = torch.randn((V, K), device=dev, generator=g).bfloat16()
    W = (torch.randn((N, K), device=dev, generator=g) * 0.05).bfloat16()
    o = capi.gemm_nt(A, W)
    bad = sum(0 if torch.equal(capi.gemm_nt(A, W), o) else 1 for _ in range(REPS))
    res[f"bf16 nt256 N={N} K={K} big={capi.gemm_nt_takes_big_tile(V, N, K, K, K, N)}"] = bad
print(ENG, res)
