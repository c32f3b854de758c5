// The network's input step and its autograd, fused (SingleScaleGCN.forward, util/networks.py:65-79):
//
//   mid = (lo + hi) / 2,  extent = max_k (hi_k - lo_k)                  -- bounding box of z1 (one scale, per-axis centre)
//   X[p] = ( dm[v] (z1[v] - mid) / extent ,  dm[v] ),   v = order[p]    -- masked, mask appended as 4th channel,
//                                                                          rows in PROCESSING order, feature dtype
//
// The reference does this with ~8 elementwise / cat / index launches over [V, 3..4] tensors; its autograd adds two
// column reductions over all V rows (the gradient of the broadcast centre and of the scalar extent) that ATen runs at
// ~150 us each at V = 1 M.  Here: one launch forward; backward one launch (dz1 in caller order, per-block partial sums
// of the centre / extent gradients, fixed order: deterministic) and one tiny finalize that turns the partials into the
// gradients of lo and hi (which autograd then routes to the arg-extreme vertices through torch.min / torch.max).
#include "sg_common.h"

namespace sg {
namespace {

constexpr int kBlock = 256;
constexpr int kRowsPerBlock = 1024;

__device__ __forceinline__ void box(const float* __restrict__ lo, const float* __restrict__ hi, float* mid, float& extent,
                                    int& kmax) {
  extent = hi[0] - lo[0];
  kmax = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    mid[k] = (lo[k] + hi[k]) * 0.5f;
    const float e = hi[k] - lo[k];
    if (e > extent) { extent = e; kmax = k; }     // (first maximal axis wins, as torch.max does)
  }
}

template <typename OUT>
__global__ __launch_bounds__(kBlock) void input_prep_fwd(const float* __restrict__ z1, const float* __restrict__ dm,
                                                         const int64_t* __restrict__ order, const float* __restrict__ lo,
                                                         const float* __restrict__ hi, OUT* __restrict__ X, int64_t ldx,
                                                         int64_t V) {
  float mid[3], extent;
  int kmax;
  box(lo, hi, mid, extent, kmax);
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < V; p += (int64_t)gridDim.x * kBlock) {
    const int64_t v = order ? order[p] : p;
    const float m = dm ? dm[v] : 1.0f;
    OUT* row = X + p * ldx;
#pragma unroll
    for (int k = 0; k < 3; ++k) row[k] = (OUT)(m * ((z1[v * 3 + k] - mid[k]) / extent));
    row[3] = (OUT)m;
  }
}

// partial[b] = { sum_v g_c[v][0..2] / extent ,  sum_v sum_k g_c[v][k] (z1[v][k] - mid[k]) / extent^2 }  over the block's
// vertices, g_c = dm * gX[rank[v]][0..2];  dz1[v] = g_c[v] / extent
template <typename IN>
__global__ __launch_bounds__(kBlock) void input_prep_bwd(const IN* __restrict__ gX, int64_t ldg, const float* __restrict__ z1,
                                                         const float* __restrict__ dm, const int64_t* __restrict__ rank,
                                                         const float* __restrict__ lo, const float* __restrict__ hi,
                                                         float* __restrict__ dz1, float* __restrict__ partial, int64_t V) {
  __shared__ float s_red[4][kBlock / 64];
  float mid[3], extent;
  int kmax;
  box(lo, hi, mid, extent, kmax);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t v0 = (int64_t)blockIdx.x * kRowsPerBlock;
  const int64_t v1 = v0 + kRowsPerBlock < V ? v0 + kRowsPerBlock : V;
  for (int64_t v = v0 + threadIdx.x; v < v1; v += kBlock) {
    const int64_t p = rank ? rank[v] : v;
    const float m = dm ? dm[v] : 1.0f;
    const IN* g = gX + p * ldg;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float gc = m * (float)g[k];
      const float d = gc / extent;
      if (dz1) dz1[v * 3 + k] = d;
      acc[k] += d;
      acc[3] += d * ((z1[v * 3 + k] - mid[k]) / extent);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float a = acc[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if ((threadIdx.x & 63) == 0) s_red[q][threadIdx.x >> 6] = a;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int q = threadIdx.x;
    partial[(int64_t)blockIdx.x * 4 + q] = (s_red[q][0] + s_red[q][1]) + (s_red[q][2] + s_red[q][3]);
  }
}

// d_lo[k] = d_hi[k] = -0.5 S_k;  the extent's axis also gets  d_hi += -T,  d_lo -= -T   (extent = hi_k* - lo_k*)
__global__ __launch_bounds__(256) void input_prep_bwd_finalize(const float* __restrict__ partial, int64_t nb,
                                                               const float* __restrict__ lo, const float* __restrict__ hi,
                                                               float* __restrict__ d_lo, float* __restrict__ d_hi) {
  __shared__ double s_w[4][4];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t b = threadIdx.x; b < nb; b += 256) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] += partial[b * 4 + q];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    double a = acc[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[q][threadIdx.x >> 6] = a;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float mid[3], extent;
    int kmax;
    box(lo, hi, mid, extent, kmax);
    const double T = (s_w[3][0] + s_w[3][1]) + (s_w[3][2] + s_w[3][3]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double S = (s_w[k][0] + s_w[k][1]) + (s_w[k][2] + s_w[k][3]);
      double gl = -0.5 * S, gh = -0.5 * S;
      if (k == kmax) { gh += -T; gl -= -T; }
      d_lo[k] = (float)gl;
      d_hi[k] = (float)gh;
    }
  }
}

}  // namespace

int64_t input_prep_blocks(int64_t V) { return V <= 0 ? 0 : (V + kRowsPerBlock - 1) / kRowsPerBlock; }

int launch_input_prep(const float* z1, const float* dm, const int64_t* order, const float* lo, const float* hi, void* X,
                      int64_t ldx, int64_t V, int dtype, hipStream_t stream) {
  if (V == 0) return SG_OK;
  int64_t nb = (V + kBlock - 1) / kBlock;
  if (nb > 256 * 32) nb = 256 * 32;
  if (dtype == SG_F32) input_prep_fwd<float><<<(int)nb, kBlock, 0, stream>>>(z1, dm, order, lo, hi, (float*)X, ldx, V);
  else input_prep_fwd<__bf16><<<(int)nb, kBlock, 0, stream>>>(z1, dm, order, lo, hi, (__bf16*)X, ldx, V);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

int launch_input_prep_bwd(const void* gX, int64_t ldg, const float* z1, const float* dm, const int64_t* rank, const float* lo,
                          const float* hi, float* dz1, float* partial, float* d_lo, float* d_hi, int64_t V, int dtype,
                          hipStream_t stream) {
  if (V == 0) return SG_OK;
  const int64_t nb = input_prep_blocks(V);
  if (dtype == SG_F32)
    input_prep_bwd<float><<<(int)nb, kBlock, 0, stream>>>((const float*)gX, ldg, z1, dm, rank, lo, hi, dz1, partial, V);
  else
    input_prep_bwd<__bf16><<<(int)nb, kBlock, 0, stream>>>((const __bf16*)gX, ldg, z1, dm, rank, lo, hi, dz1, partial, V);
  SG_HIP_TRY(hipGetLastError());
  input_prep_bwd_finalize<<<1, 256, 0, stream>>>(partial, nb, lo, hi, d_lo, d_hi);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

}  // namespace sg
