// Dense products with a tiny weight matrix (K, N <= 16): the 4 -> 16 input layer of the SGCN (`lins[k]` on the 12-wide
// [Tx0|Tx1|Tx2] of 4-channel features, util/networks.py:42 via [3P] ChebConv.forward), its input and weight gradients,
// and the 16 -> 3 output layer (`nn.Linear(16, 3)`, util/networks.py:36,55) with its autograd.  K = 12 is no multiple of
// the 16- / 32-deep MFMA steps and the work is a few FLOPs per byte: one thread per row, the weights in LDS, fp32
// accumulation, rows streamed once.  These were the last products of the training iteration on the BLAS library
// (five launches, 0.05 - 0.12 ms each at V = 1 M for 24 - 64 MB of traffic).
//
//   thin_nt:  Y[v, n] = sum_k X[v, k] * W[n, k] (+ bias[n])        X, Y fp32 or bf16 (same type), W and bias fp32
//   thin_tn:  out[n, k] = sum_v A[v, n] * B[v, k]  (fp32 out)       per-block partial sums in block order: deterministic
#include "sg_common.h"

namespace sg {
namespace {

constexpr int kThinMax = 16;
constexpr int kThinBlock = 256;

template <typename T> __device__ __forceinline__ float thin_load(const T* p);
template <> __device__ __forceinline__ float thin_load<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float thin_load<uint16_t>(const uint16_t* p) { return __uint_as_float((uint32_t)*p << 16); }
template <typename T> __device__ __forceinline__ void thin_store(T* p, float v);
template <> __device__ __forceinline__ void thin_store<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void thin_store<uint16_t>(uint16_t* p, float v) { *p = __builtin_bit_cast(uint16_t, (__bf16)v); }

template <typename T>
__global__ __launch_bounds__(kThinBlock) void thin_nt(const T* __restrict__ X, int64_t ldx, const float* __restrict__ W, int64_t ldw,
                                                      const float* __restrict__ bias, T* __restrict__ Y, int64_t ldy, int64_t V,
                                                      int N, int K) {
  __shared__ float s_w[kThinMax * kThinMax];
  __shared__ float s_b[kThinMax];
  for (int i = threadIdx.x; i < N * K; i += kThinBlock) s_w[i] = W[(int64_t)(i / K) * ldw + i % K];
  if (threadIdx.x < N) s_b[threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
  __syncthreads();
  for (int64_t v = (int64_t)blockIdx.x * kThinBlock + threadIdx.x; v < V; v += (int64_t)gridDim.x * kThinBlock) {
    float x[kThinMax];
#pragma unroll
    for (int k = 0; k < kThinMax; ++k) x[k] = k < K ? thin_load<T>(X + v * ldx + k) : 0.f;
    for (int n = 0; n < N; ++n) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < kThinMax; ++k)
        if (k < K) acc = fmaf(x[k], s_w[n * K + k], acc);
      thin_store<T>(Y + v * ldy + n, acc + s_b[n]);
    }
  }
}

// one thread per output element (n, k) of the block's partial; the block's rows pass through LDS in chunks
constexpr int kThinChunk = 128;
template <typename T>
__global__ __launch_bounds__(kThinBlock) void thin_tn_partial(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb,
                                                              int64_t V, int N, int K, int64_t rows_per_block,
                                                              float* __restrict__ part) {
  __shared__ float s_a[kThinChunk * kThinMax];
  __shared__ float s_b[kThinChunk * kThinMax];
  const int n = threadIdx.x / kThinMax, k = threadIdx.x % kThinMax;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  r1 = r1 > V ? V : r1;
  float acc = 0.f;
  for (int64_t c0 = r0; c0 < r1; c0 += kThinChunk) {
    const int rows = (int)(r1 - c0 < kThinChunk ? r1 - c0 : kThinChunk);
    __syncthreads();
    for (int i = threadIdx.x; i < rows * N; i += kThinBlock) s_a[i] = thin_load<T>(A + (c0 + i / N) * lda + i % N);
    for (int i = threadIdx.x; i < rows * K; i += kThinBlock) s_b[i] = thin_load<T>(B + (c0 + i / K) * ldb + i % K);
    __syncthreads();
    if (n < N && k < K)
      for (int r = 0; r < rows; ++r) acc = fmaf(s_a[r * N + n], s_b[r * K + k], acc);
  }
  part[(int64_t)blockIdx.x * (kThinMax * kThinMax) + threadIdx.x] = acc;
}

__global__ __launch_bounds__(kThinBlock) void thin_tn_reduce(const float* __restrict__ part, int nblocks, int N, int K,
                                                             float* __restrict__ out, int64_t ldo) {
  const int n = threadIdx.x / kThinMax, k = threadIdx.x % kThinMax;
  if (n >= N || k >= K) return;
  float acc = 0.f;
  for (int b = 0; b < nblocks; ++b) acc += part[(int64_t)b * (kThinMax * kThinMax) + threadIdx.x];      // fixed order
  out[(int64_t)n * ldo + k] = acc;
}

}  // namespace

bool thin_shape(int64_t N, int64_t K) { return N >= 1 && K >= 1 && N <= kThinMax && K <= kThinMax; }

int64_t thin_tn_blocks(int64_t V) {
  int64_t nb = (V + 4095) / 4096;
  return nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
}

int launch_thin_nt(const void* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, void* Y, int64_t ldy, int64_t V,
                   int64_t N, int64_t K, int dtype, hipStream_t stream) {
  if (V == 0) return SG_OK;
  int64_t nb = (V + kThinBlock - 1) / kThinBlock;
  nb = nb > 256 * 16 ? 256 * 16 : nb;
  if (dtype == SG_F32)
    thin_nt<float><<<(int)nb, kThinBlock, 0, stream>>>((const float*)X, ldx, W, ldw, bias, (float*)Y, ldy, V, (int)N, (int)K);
  else if (dtype == SG_BF16)
    thin_nt<uint16_t><<<(int)nb, kThinBlock, 0, stream>>>((const uint16_t*)X, ldx, W, ldw, bias, (uint16_t*)Y, ldy, V, (int)N, (int)K);
  else {
    set_error("sg_thin_nt: unsupported dtype %d", dtype);
    return SG_ERR_UNSUPPORTED;
  }
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

int launch_thin_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t V, int64_t N, int64_t K, int dtype,
                   float* workspace, float* out, int64_t ldo, hipStream_t stream) {
  const int64_t nb = thin_tn_blocks(V);
  const int64_t rpb = (V + nb - 1) / nb;
  if (dtype == SG_F32)
    thin_tn_partial<float><<<(int)nb, kThinBlock, 0, stream>>>((const float*)A, lda, (const float*)B, ldb, V, (int)N, (int)K, rpb, workspace);
  else if (dtype == SG_BF16)
    thin_tn_partial<uint16_t><<<(int)nb, kThinBlock, 0, stream>>>((const uint16_t*)A, lda, (const uint16_t*)B, ldb, V, (int)N, (int)K, rpb, workspace);
  else {
    set_error("sg_thin_tn: unsupported dtype %d", dtype);
    return SG_ERR_UNSUPPORTED;
  }
  SG_HIP_TRY(hipGetLastError());
  thin_tn_reduce<<<1, kThinBlock, 0, stream>>>(workspace, (int)nb, (int)N, (int)K, out, ldo);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

}  // namespace sg
