// float32 products with a SMALL weight matrix (4 .. 48 rows and columns, multiples of 4) that neither the thin kernels
// (<= 256 weight entries, thin_gemm.hip) nor the split-bf16 matrix-core kernels (N, K >= 64, K a multiple of 32,
// gemm_split.hip) take: the 16 -> 32 and 32 -> 16 layers of the SGCN (`lins[k]` on [V, 48] x [48, 32] and
// [V, 32] x [32, 48], util/networks.py:40-53 via [3P] ChebConv.forward) with their input and weight gradients, and the narrow
// stages of the MGCN (util/meshnet.py:92-95,157-160) -- the last products of a float32 iteration that went to the BLAS library
// (0.24 ms of c2's 5.5 ms at V = 50 K: 16 - 78 us per launch for 16 - 20 MB of traffic).
// They are HBM-bound (2 N K / (4 (N + K)) = 10 - 16 flop per byte): plain float32 FMAs on the vector ALUs, rows streamed
// once, the weights in LDS.
//
//   mid_nt:  Y[v, n] = sum_k X[v, k] w(n, k) (+ bias[n])     w(n, k) = W[n * w_rs + k * w_cs]: [N, K] or [K, N] storage
//            one thread per row (its K values in registers), 256 rows per block passing through LDS both ways so that
//            consecutive lanes touch consecutive addresses; every weight is one broadcast LDS read per 4 FMAs
//   mid_tn:  out[n, k] = sum_v A[v, n] B[v, k]                 each wavefront holds the WHOLE [N, Kp] result as 8 x 8
//            register tiles of TN x TK and takes every fourth row of a 128-row chunk staged in LDS; the block's partial
//            (wavefronts added in order) goes to the workspace, the blocks' partials are summed by gemm_split.hip's
//            reduce kernel in a fixed tree (+= into the .grad accumulators when a sink is given): deterministic
#include "sg_common.h"

namespace sg {
namespace {

constexpr int kMidBlock = 256;
constexpr int kMidChunk = 128;
constexpr int kMidRowsPerBlock = 512;

template <int P>      // K rounded up to 16 / 32 / 48
__global__ __launch_bounds__(kMidBlock) void mid_nt(const float* __restrict__ X, int64_t ldx, const float* __restrict__ W, int64_t w_rs,
                                                    int64_t w_cs, const float* __restrict__ bias, float* __restrict__ Y, int64_t ldy,
                                                    int64_t V, int N, int K) {
  extern __shared__ float s_mid[];                   // weights [N][P] (zero beyond K) | rows: x [256][K + 1], then y [256][N + 1]
  float* const s_w = s_mid;
  float* const s_io = s_mid + N * P;
  const int tid = threadIdx.x;
  for (int i = tid; i < N * P; i += kMidBlock) {
    const int n = i / P, k = i - n * P;
    s_w[i] = k < K ? W[(int64_t)n * w_rs + (int64_t)k * w_cs] : 0.f;
  }
  const int64_t row0 = (int64_t)blockIdx.x * kMidBlock;
  const int rows = (int)(V - row0 < kMidBlock ? V - row0 : kMidBlock);
  const int px = K + 1, py = N + 1;                  // odd pitches: a thread per row reads without bank conflicts
  {
    const int per_row = K >> 2;
    const float* const X0 = X + row0 * ldx;
    for (int i = tid; i < rows * per_row; i += kMidBlock) {
      const int r = i / per_row, c = (i - r * per_row) << 2;
      const float4 v = *(const float4*)(X0 + (int64_t)r * ldx + c);
      float* q = s_io + r * px + c;
      q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
    }
  }
  __syncthreads();
  float x[P];
#pragma unroll
  for (int k = 0; k < P; ++k) x[k] = (k < K && tid < rows) ? s_io[tid * px + k] : 0.f;
  __syncthreads();                                   // the staging area is free: the results go back through it
  for (int n = 0; n < N; ++n) {
    const float4* const wn = (const float4*)(s_w + n * P);
    float acc = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < P / 4; ++k4) {
      const float4 w = wn[k4];                       // one address for the whole wavefront: a broadcast read
      acc = fmaf(x[4 * k4 + 0], w.x, acc);
      acc = fmaf(x[4 * k4 + 1], w.y, acc);
      acc = fmaf(x[4 * k4 + 2], w.z, acc);
      acc = fmaf(x[4 * k4 + 3], w.w, acc);
    }
    s_io[tid * py + n] = acc + (bias ? bias[n] : 0.f);
  }
  __syncthreads();
  {
    const int per_row = N >> 2;
    float* const Y0 = Y + row0 * ldy;
    for (int i = tid; i < rows * per_row; i += kMidBlock) {
      const int r = i / per_row, c = (i - r * per_row) << 2;
      const float* q = s_io + r * py + c;
      *(float4*)(Y0 + (int64_t)r * ldy + c) = make_float4(q[0], q[1], q[2], q[3]);
    }
  }
}

template <int TN, int TK>
__global__ __launch_bounds__(kMidBlock) void mid_tn_partial(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                            int64_t V, int N, int Kp, float* __restrict__ part) {
  constexpr int PA = 8 * TN, PB = 8 * TK;            // LDS row pitches: columns beyond N / Kp are zeros
  __shared__ __attribute__((aligned(16))) float s_all[kMidChunk * (PA + PB)];
  float* const s_a = s_all;
  float* const s_b = s_all + kMidChunk * PA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tn = lane >> 3, tk = lane & 7;
  const int64_t r0 = (int64_t)blockIdx.x * kMidRowsPerBlock;
  int64_t r1 = r0 + kMidRowsPerBlock;
  r1 = r1 > V ? V : r1;
  float acc[TN][TK];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j) acc[i][j] = 0.f;
  for (int64_t c0 = r0; c0 < r1; c0 += kMidChunk) {
    const int rows = (int)(r1 - c0 < kMidChunk ? r1 - c0 : kMidChunk);
    __syncthreads();
    for (int i = tid; i < kMidChunk * (PA / 4); i += kMidBlock) {
      const int r = i / (PA / 4), c = (i - r * (PA / 4)) << 2;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < rows && c < N) v = *(const float4*)(A + (c0 + r) * lda + c);
      *(float4*)(s_a + r * PA + c) = v;
    }
    for (int i = tid; i < kMidChunk * (PB / 4); i += kMidBlock) {
      const int r = i / (PB / 4), c = (i - r * (PB / 4)) << 2;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < rows && c < Kp) v = *(const float4*)(B + (c0 + r) * ldb + c);
      *(float4*)(s_b + r * PB + c) = v;
    }
    __syncthreads();
#pragma unroll 2
    for (int r = wave; r < kMidChunk; r += 4) {      // (rows past the end of the block are zeros)
      float a[TN], b[TK];
#pragma unroll
      for (int i = 0; i < TN; i += 2) {
        const float2 t = *(const float2*)(s_a + r * PA + tn * TN + i);
        a[i] = t.x;
        a[i + 1] = t.y;
      }
#pragma unroll
      for (int j = 0; j < TK; j += 2) {
        const float2 t = *(const float2*)(s_b + r * PB + tk * TK + j);
        b[j] = t.x;
        b[j + 1] = t.y;
      }
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
  }
  // the four wavefronts' sums, added in wavefront order
  __syncthreads();
  float* const s_red = s_all;                        // [4][TN * TK][64]
  static_assert(4 * 64 * TN * TK <= kMidChunk * (PA + PB), "the reduction area fits into the staging arrays");
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j) s_red[(wave * TN * TK + i * TK + j) * 64 + lane] = acc[i][j];
  __syncthreads();
  float* const out = part + (int64_t)blockIdx.x * N * Kp;
  for (int e = tid; e < 64 * TN * TK; e += kMidBlock) {
    const int l = e & 63, ij = e >> 6;
    const int i = ij / TK, j = ij - i * TK;
    const int n = (l >> 3) * TN + i, k = (l & 7) * TK + j;
    if (n < N && k < Kp) {
      float v = s_red[(0 * TN * TK + ij) * 64 + l];
      v += s_red[(1 * TN * TK + ij) * 64 + l];
      v += s_red[(2 * TN * TK + ij) * 64 + l];
      v += s_red[(3 * TN * TK + ij) * 64 + l];
      out[(int64_t)n * Kp + k] = v;
    }
  }
}

inline int mid_t(int64_t n) { return n <= 16 ? 2 : (n <= 32 ? 4 : 6); }

}  // namespace

// (48: the rows of a block and the weights fit into 64 KB of LDS)
bool mid_shape(int64_t N, int64_t K) { return N >= 4 && N <= 48 && K >= 4 && K <= 48 && N % 4 == 0 && K % 4 == 0; }

static int64_t mid_tn_blocks(int64_t M) { return (M + kMidRowsPerBlock - 1) / kMidRowsPerBlock; }
int64_t mid_tn_workspace(int64_t M, int64_t N, int64_t Kp) { return mid_tn_blocks(M) * N * Kp; }      // float32 elements

int launch_mid_nt(const float* X, int64_t ldx, const float* W, int64_t w_rs, int64_t w_cs, const float* bias, float* Y, int64_t ldy,
                  int64_t M, int64_t N, int64_t K, hipStream_t stream) {
  SG_REQUIRE(mid_shape(N, K) && ldx % 4 == 0 && ldy % 4 == 0 && (((uintptr_t)X | (uintptr_t)Y) & 15) == 0,
             "mid_nt: unsupported shape or alignment (N=%lld K=%lld)", (long long)N, (long long)K);
  if (M == 0) return SG_OK;
  const int64_t nb = (M + kMidBlock - 1) / kMidBlock;
  SG_REQUIRE(nb < ((int64_t)1 << 31), "mid_nt: too many rows");
  const int P = (int)((K + 15) / 16 * 16);
  const int64_t wide = (K > N ? K : N) + 1;
  const size_t lds = (size_t)(N * P + kMidBlock * wide) * sizeof(float);
#define SG_MID_NT(PP) mid_nt<PP><<<(int)nb, kMidBlock, lds, stream>>>(X, ldx, W, w_rs, w_cs, bias, Y, ldy, M, (int)N, (int)K)
  if (P == 16) SG_MID_NT(16);
  else if (P == 32) SG_MID_NT(32);
  else SG_MID_NT(48);
#undef SG_MID_NT
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

int launch_mid_tn(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int64_t Kp, float* ws, float* out,
                  int64_t ldo, hipStream_t stream, const GradSink* sink) {
  SG_REQUIRE(mid_shape(N, Kp) && lda % 4 == 0 && ldb % 4 == 0 && ldo % 4 == 0 && ws &&
                 (((uintptr_t)A | (uintptr_t)B | (uintptr_t)out | (uintptr_t)ws) & 15) == 0,
             "mid_tn: unsupported shape or alignment (N=%lld Kp=%lld)", (long long)N, (long long)Kp);
  SG_REQUIRE(M > 0 && mid_tn_blocks(M) < (1 << 30), "mid_tn: bad row count");
  const int nb = (int)mid_tn_blocks(M);
  const int tn = mid_t(N), tk = mid_t(Kp);
#define SG_MID_TN(TN, TK) mid_tn_partial<TN, TK><<<nb, kMidBlock, 0, stream>>>(A, lda, B, ldb, M, (int)N, (int)Kp, ws)
#define SG_MID_TN_ROW(TN)                                   \
  do {                                                      \
    if (tk == 2) SG_MID_TN(TN, 2);                          \
    else if (tk == 4) SG_MID_TN(TN, 4);                     \
    else SG_MID_TN(TN, 6);                                  \
  } while (0)
  if (tn == 2) SG_MID_TN_ROW(2);
  else if (tn == 4) SG_MID_TN_ROW(4);
  else SG_MID_TN_ROW(6);
#undef SG_MID_TN_ROW
#undef SG_MID_TN
  SG_HIP_TRY(hipGetLastError());
  return launch_split_tn_reduce(ws, nb, N, Kp, out, ldo, sink, stream);
}

}  // namespace sg
