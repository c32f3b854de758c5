// In-kernel shader-clock stamps of the persistent 256 x 256 product (one workgroup, wavefronts 0 and 4, K steps 16..31):
// where does a phase spend its cycles -- issue of the LDS reads / DMA, the counted vmcnt wait, the barriers, the MFMAs?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Isemigcn_amd/csrc -DSG_GEMM256_STAMPS tools/gemm256_stamps.hip -o tools/gemm256_stamps
#include <cstdio>
#include <vector>
#include "../semigcn_amd/csrc/gemm_mfma256.hip"
namespace sg {
thread_local char g_err[512];
void set_error(const char* fmt, ...) {}
}
int main(int argc, char** argv) {
  using namespace sg;
  const int64_t M = argc > 1 ? atoll(argv[1]) : 1000000, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 768;
  uint16_t *A, *B, *C;
  unsigned long long* st;
  hipMalloc(&A, M * K * 2); hipMalloc(&B, N * K * 2); hipMalloc(&C, M * N * 2); hipMalloc(&st, 2 * 16 * 4 * 6 * 8);
  hipMemset(A, 0x3c, M * K * 2); hipMemset(B, 0x3c, N * K * 2); hipMemset(st, 0, 2 * 16 * 4 * 6 * 8);
  Big g;
  g.A = A; g.lda = K; g.B = B; g.ldb = K; g.bias = nullptr; g.C = C; g.ldc = N; g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.n_col_tiles = (int)(N / 256); g.n_row_tiles = (int)((M + 255) / 256);
  int streams = (256 / g.n_col_tiles) / 8 * 8;
  g.streams = streams; g.stamps = st;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    gemm_nt_256<<<streams * g.n_col_tiles, kThreads256>>>(g);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("M=%lld N=%lld K=%lld: %.3f ms  %.1f TFLOP/s\n", (long long)M, (long long)N, (long long)K, ms, 2.0 * M * N * K / ms / 1e9);
  }
  std::vector<unsigned long long> h(2 * 16 * 4 * 6);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  const char* names[6] = {"start", "issued", "vmwait", "barrier1", "mfma_done", "barrier2"};
  for (int w = 0; w < 2; ++w) {
    double acc[4][6] = {};
    for (int s = 1; s < 15; ++s)
      for (int p = 0; p < 4; ++p)
        for (int k = 0; k < 6; ++k) {
          unsigned long long prev = k ? h[((w * 16 + s) * 4 + p) * 6 + k - 1] : (p ? h[((w * 16 + s) * 4 + p - 1) * 6 + 5] : h[((w * 16 + s - 1) * 4 + 3) * 6 + 5]);
          acc[p][k] += (double)(h[((w * 16 + s) * 4 + p) * 6 + k] - prev) / 14.0;
        }
    printf("wavefront %d (group %d): mean cycles between stamps over 14 K steps (shader clock)\n", w * 4, w);
    for (int p = 0; p < 4; ++p) {
      printf("  phase %d:", p);
      double t = 0;
      for (int k = 0; k < 6; ++k) { printf(" %s %.0f", names[k], acc[p][k]); t += acc[p][k]; }
      printf("  | total %.0f\n", t);
    }
    printf("  K step: %llu cycles (steps 17..30 mean)\n", (h[((w * 16 + 14) * 4 + 3) * 6 + 5] - h[((w * 16 + 0) * 4 + 3) * 6 + 5]) / 14);
  }
  return 0;
}
