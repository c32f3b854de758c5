// Dense products with a tiny weight matrix (N x K <= 256 entries: K <= 8 with N <= 32, K, N <= 16, or K <= 32 with N <= 8 --
// the last two shapes are the 32 -> 3 heads of the MGCN, util/meshnet.py:228,236,244, and their gradients): the 4 -> 16 input layer of the SGCN (`lins[k]` on the 12-wide
// [Tx0|Tx1|Tx2] of 4-channel features, util/networks.py:42 via [3P] ChebConv.forward), its input and weight gradients,
// and the 16 -> 3 output layer (`nn.Linear(16, 3)`, util/networks.py:36,55) with its autograd.  K = 12 is no multiple of
// the 16- / 32-deep MFMA steps and the work is a few FLOPs per byte: one thread per row, the weights in LDS, fp32
// accumulation, rows streamed once.  These were the last products of the training iteration on the BLAS library
// (five launches, 0.05 - 0.12 ms each at V = 1 M for 24 - 64 MB of traffic).
//
//   thin_nt:  Y[v, n] = sum_k X[v, k] * W[n, k] (+ bias[n])        X, Y fp32 or bf16 (same type), W and bias fp32
//   thin_tn:  out[n, k] = sum_v A[v, n] * B[v, k]  (fp32 out)       per-block partial sums added in a fixed tree: deterministic
#include "sg_common.h"

namespace sg {
namespace {

constexpr int kThinMax = 32;           // largest N or K
constexpr int kThinElems = 256;        // N x (K rounded up to 8 / 16 / 32) <= this: one thread per entry of the weight gradient
constexpr int kThinBlock = 256;
__host__ __device__ inline int thin_pitch(int K) { return K <= 8 ? 8 : (K <= 16 ? 16 : 32); }

template <typename T> __device__ __forceinline__ float thin_load(const T* p);
template <> __device__ __forceinline__ float thin_load<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float thin_load<uint16_t>(const uint16_t* p) { return __uint_as_float((uint32_t)*p << 16); }
template <typename T> __device__ __forceinline__ void thin_store(T* p, float v);
template <> __device__ __forceinline__ void thin_store<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void thin_store<uint16_t>(uint16_t* p, float v) { *p = __builtin_bit_cast(uint16_t, (__bf16)v); }

// rows x W elements from global memory (row stride ld) into LDS (row stride lds_ld floats), converted to fp32, with the
// widest loads the shape allows: VEC elements per load (16 bytes, 8 bytes, or one element)
template <typename T, int VEC>
__device__ __forceinline__ void thin_stage_in(const T* __restrict__ src, int64_t ld, int rows, int W, float* __restrict__ dst, int lds_ld) {
  const int per_row = W / VEC;
  for (int i = threadIdx.x; i < rows * per_row; i += kThinBlock) {
    const int r = i / per_row, c = (i - r * per_row) * VEC;
    const T* p = src + (int64_t)r * ld + c;
    float* q = dst + r * lds_ld + c;
    if constexpr (VEC == 1) {
      q[0] = thin_load<T>(p);
    } else if constexpr (sizeof(T) == 4) {
      if constexpr (VEC == 4) { const float4 v = *(const float4*)p; q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w; }
      else { const float2 v = *(const float2*)p; q[0] = v.x; q[1] = v.y; }
    } else {
      uint32_t w[VEC / 2];
      if constexpr (VEC == 8) { const uint4 v = *(const uint4*)p; w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
      else if constexpr (VEC == 4) { const uint2 v = *(const uint2*)p; w[0] = v.x; w[1] = v.y; }
      else { w[0] = *(const uint32_t*)p; }
#pragma unroll
      for (int j = 0; j < VEC / 2; ++j) {
        q[2 * j] = __uint_as_float(w[j] << 16);
        q[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
      }
    }
  }
}
template <typename T>
__device__ __forceinline__ void thin_stage_rows(const T* __restrict__ src, int64_t ld, int rows, int W, float* __restrict__ dst, int lds_ld) {
  constexpr int V16 = 16 / (int)sizeof(T), V8 = 8 / (int)sizeof(T), V4 = 4 / (int)sizeof(T);
  const uintptr_t a = (uintptr_t)src;
  if (W % V16 == 0 && ld % V16 == 0 && a % 16 == 0) thin_stage_in<T, V16>(src, ld, rows, W, dst, lds_ld);
  else if (W % V8 == 0 && ld % V8 == 0 && a % 8 == 0) thin_stage_in<T, V8>(src, ld, rows, W, dst, lds_ld);
  else if (V4 > 1 && W % V4 == 0 && ld % V4 == 0 && a % 4 == 0) thin_stage_in<T, (V4 > 1 ? V4 : 1)>(src, ld, rows, W, dst, lds_ld);
  else thin_stage_in<T, 1>(src, ld, rows, W, dst, lds_ld);
}

// One block = 256 consecutive rows.  Rows pass through LDS in both directions so that consecutive lanes touch consecutive
// addresses (a row is 6 - 64 bytes: one thread per row straight from global memory runs at ~1 TB/s).  P = K rounded up to
// 8 / 16 / 32: a thread keeps its row in P registers; the weights are read with wave-uniform addresses straight from
// global memory (scalar loads through the constant cache -- the first version staged them in LDS and issued one LDS read
// per multiply) and the LDS is sized at the launch (5 - 7 blocks per CU instead of 3): [V,12] x [12,16] bf16 0.076 -> 0.048 ms at
// V = 1 M, [V,16] x [16,3] fp32 0.049 -> 0.021 (tools/thin_bench.py).  Same fma chain per output: bit-identical to that version.
template <typename T, int P>
__global__ __launch_bounds__(kThinBlock) void thin_nt(const T* __restrict__ X, int64_t ldx, const float* __restrict__ W, int64_t ldw,
                                                      const float* __restrict__ bias, T* __restrict__ Y, int64_t ldy, int64_t V,
                                                      int N, int K) {
  extern __shared__ float s_io[];      // x rows (pitch K + 1) | y rows (pitch N + 1): sized at the launch, so that 5 - 7 blocks share a CU
  const int px = K + 1, py = N + 1;
  float* const s_x = s_io;
  float* const s_y = s_io + kThinBlock * px;
  const int64_t row0 = (int64_t)blockIdx.x * kThinBlock;
  const int rows = (int)(V - row0 < kThinBlock ? V - row0 : kThinBlock);
  thin_stage_rows<T>(X + row0 * ldx, ldx, rows, K, s_x, px);
  __syncthreads();
  if ((int)threadIdx.x < rows) {
    float x[P];
#pragma unroll
    for (int k = 0; k < P; ++k) x[k] = k < K ? s_x[threadIdx.x * px + k] : 0.f;
    for (int n = 0; n < N; ++n) {
      const float* __restrict__ wn = W + (int64_t)n * ldw;       // wave-uniform: scalar loads
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < P; ++k)
        if (k < K) acc = fmaf(x[k], wn[k], acc);
      s_y[threadIdx.x * py + n] = acc + (bias ? bias[n] : 0.f);
    }
  }
  __syncthreads();
  constexpr int V8 = 8 / (int)sizeof(T);               // 8-byte stores where the rows allow (4 bf16 / 2 fp32)
  T* const Y0 = Y + row0 * ldy;
  if (N % V8 == 0 && ldy % V8 == 0 && (uintptr_t)Y0 % 8 == 0) {
    const int per_row = N / V8;
    for (int i = threadIdx.x; i < rows * per_row; i += kThinBlock) {
      const int r = i / per_row, c = (i - r * per_row) * V8;
      const float* q = s_y + r * py + c;
      if constexpr (sizeof(T) == 4) {
        *(float2*)(Y0 + (int64_t)r * ldy + c) = make_float2(q[0], q[1]);
      } else {
        uint16_t h[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = __builtin_bit_cast(uint16_t, (__bf16)q[j]);
        *(uint2*)(Y0 + (int64_t)r * ldy + c) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
      }
    }
  } else {
    for (int i = threadIdx.x; i < rows * N; i += kThinBlock) {
      const int r = i / N, n = i - r * N;
      thin_store<T>(Y0 + (int64_t)r * ldy + n, s_y[r * py + n]);
    }
  }
}

// one thread per output element (n, k) of the block's partial; the block's rows pass through LDS in chunks
constexpr int kThinChunk = 128;
constexpr int kThinTnRows = 512;      // rows per block: >= 8 blocks per CU at V = 1 M
template <typename T>
__global__ __launch_bounds__(kThinBlock) void thin_tn_partial(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb,
                                                              int64_t V, int N, int K, int64_t rows_per_block,
                                                              float* __restrict__ part) {
  extern __shared__ float s_io[];                  // [chunk][N] | [chunk][K], sized at the launch
  float* const s_a = s_io;
  float* const s_b = s_io + kThinChunk * N;
  const int P = thin_pitch(K);
  const int n = threadIdx.x / P, k = threadIdx.x % P;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  r1 = r1 > V ? V : r1;
  float acc = 0.f;
  for (int64_t c0 = r0; c0 < r1; c0 += kThinChunk) {
    const int rows = (int)(r1 - c0 < kThinChunk ? r1 - c0 : kThinChunk);
    __syncthreads();
    thin_stage_rows<T>(A + c0 * lda, lda, rows, N, s_a, N);
    thin_stage_rows<T>(B + c0 * ldb, ldb, rows, K, s_b, K);
    __syncthreads();
    if (n < N && k < K) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;         // four independent chains; added in a fixed order
      int r = 0;
      for (; r + 4 <= rows; r += 4) {
        a0 = fmaf(s_a[(r + 0) * N + n], s_b[(r + 0) * K + k], a0);
        a1 = fmaf(s_a[(r + 1) * N + n], s_b[(r + 1) * K + k], a1);
        a2 = fmaf(s_a[(r + 2) * N + n], s_b[(r + 2) * K + k], a2);
        a3 = fmaf(s_a[(r + 3) * N + n], s_b[(r + 3) * K + k], a3);
      }
      for (; r < rows; ++r) a0 = fmaf(s_a[r * N + n], s_b[r * K + k], a0);
      acc += (a0 + a1) + (a2 + a3);
    }
  }
  part[(int64_t)blockIdx.x * kThinElems + threadIdx.x] = acc;
}

// one block per output element: 256 threads add the blocks' partials (thread t: blocks t, t + 256, ..), then a fixed tree
__global__ __launch_bounds__(kThinBlock) void thin_tn_reduce(const float* __restrict__ part, int nblocks, int N, int K,
                                                             float* __restrict__ out, int64_t ldo, const GradSink sink) {
  __shared__ float s_p[kThinBlock];
  if ((int)blockIdx.x >= kThinElems) {                         // the rider's workgroups (GradSink::cs_*)
    colsum_ride(sink, (int)blockIdx.x - kThinElems);
    return;
  }
  const int P = thin_pitch(K);
  const int slot = blockIdx.x, n = slot / P, k = slot % P;
  if (n >= N || k >= K) return;
  float acc = 0.f;
  for (int b = threadIdx.x; b < nblocks; b += kThinBlock) acc += part[(int64_t)b * kThinElems + slot];
  s_p[threadIdx.x] = acc;
  __syncthreads();
  for (int off = kThinBlock / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s_p[threadIdx.x] += s_p[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[(int64_t)n * ldo + k] = s_p[0];
    if (sink.mode) *sink_ptr(sink, n, k) += s_p[0];
  }
}

}  // namespace

bool thin_shape(int64_t N, int64_t K) { return N >= 1 && K >= 1 && K <= kThinMax && N * thin_pitch((int)K) <= kThinElems; }

int64_t thin_tn_blocks(int64_t V) {
  int64_t nb = (V + kThinTnRows - 1) / kThinTnRows;
  return nb < 1 ? 1 : (nb > 4096 ? 4096 : nb);
}

int launch_thin_nt(const void* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, void* Y, int64_t ldy, int64_t V,
                   int64_t N, int64_t K, int dtype, hipStream_t stream) {
  if (V == 0) return SG_OK;
  SG_REQUIRE(V < ((int64_t)1 << 31) * kThinBlock, "sg_thin_nt: too many rows");
  const int64_t nb = (V + kThinBlock - 1) / kThinBlock;
  const int P = thin_pitch((int)K);
  const size_t lds = (size_t)kThinBlock * (size_t)(K + 1 + N + 1) * sizeof(float);
#define SG_THIN_NT(TT, PP) thin_nt<TT, PP><<<(int)nb, kThinBlock, lds, stream>>>((const TT*)X, ldx, W, ldw, bias, (TT*)Y, ldy, V, (int)N, (int)K)
  if (dtype == SG_F32) {
    if (P == 8) SG_THIN_NT(float, 8); else if (P == 16) SG_THIN_NT(float, 16); else SG_THIN_NT(float, 32);
  } else if (dtype == SG_BF16) {
    if (P == 8) SG_THIN_NT(uint16_t, 8); else if (P == 16) SG_THIN_NT(uint16_t, 16); else SG_THIN_NT(uint16_t, 32);
  } else {
    set_error("sg_thin_nt: unsupported dtype %d", dtype);
    return SG_ERR_UNSUPPORTED;
  }
#undef SG_THIN_NT
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

int launch_thin_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t V, int64_t N, int64_t K, int dtype,
                   float* workspace, float* out, int64_t ldo, hipStream_t stream, const GradSink* sink) {
  const int64_t nb = thin_tn_blocks(V);
  const int64_t rpb = (V + nb - 1) / nb;
  const size_t lds = (size_t)kThinChunk * (size_t)(N + K) * sizeof(float);
  if (dtype == SG_F32)
    thin_tn_partial<float><<<(int)nb, kThinBlock, lds, stream>>>((const float*)A, lda, (const float*)B, ldb, V, (int)N, (int)K, rpb, workspace);
  else if (dtype == SG_BF16)
    thin_tn_partial<uint16_t><<<(int)nb, kThinBlock, lds, stream>>>((const uint16_t*)A, lda, (const uint16_t*)B, ldb, V, (int)N, (int)K, rpb, workspace);
  else {
    set_error("sg_thin_tn: unsupported dtype %d", dtype);
    return SG_ERR_UNSUPPORTED;
  }
  SG_HIP_TRY(hipGetLastError());
  thin_tn_reduce<<<kThinElems + ((sink && sink->cs_partial) ? sink->cs_C : 0), kThinBlock, 0, stream>>>(workspace, (int)nb, (int)N, (int)K, out, ldo,
                                                                                                    sink ? *sink : GradSink{});
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

}  // namespace sg
