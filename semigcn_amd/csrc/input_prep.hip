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

// ---- bounding box of z1 with the arg-extreme vertices (util/networks.py:67: torch.min / torch.max over dim 0) ----------------
// Two launches instead of the transposed copy + four ATen reductions, and the vertex each bound came from is kept: the
// backward pass routes the gradients of lo / hi to those vertices itself (what autograd does through torch.min / torch.max).
// A tie goes to the lowest vertex id.
struct Ext { float lo, hi; int64_t ilo, ihi; };
__device__ __forceinline__ void ext_merge(Ext& a, const Ext& b) {
  if (b.lo < a.lo || (b.lo == a.lo && b.ilo < a.ilo)) { a.lo = b.lo; a.ilo = b.ilo; }
  if (b.hi > a.hi || (b.hi == a.hi && b.ihi < a.ihi)) { a.hi = b.hi; a.ihi = b.ihi; }
}
__device__ __forceinline__ Ext ext_shfl_down(const Ext& e, int off) {
  Ext o;
  o.lo = __shfl_down(e.lo, off, 64);
  o.hi = __shfl_down(e.hi, off, 64);
  o.ilo = __shfl_down((long long)e.ilo, off, 64);
  o.ihi = __shfl_down((long long)e.ihi, off, 64);
  return o;
}
// block sums of one stage: `n` candidates per column read through get(i, k); result by thread 0
template <typename Get>
__device__ __forceinline__ void ext_block(int64_t begin, int64_t end, Get get, float* vals /*[6]*/, int64_t* idx /*[6]*/, bool write) {
  __shared__ Ext s_e[3][kBlock / 64];
  constexpr int64_t kNone = INT64_MAX;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    Ext e{INFINITY, -INFINITY, kNone, kNone};
    for (int64_t i = begin + threadIdx.x; i < end; i += kBlock) ext_merge(e, get(i, k));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const Ext o = ext_shfl_down(e, off);
      ext_merge(e, o);
    }
    if ((threadIdx.x & 63) == 0) s_e[k][threadIdx.x >> 6] = e;
  }
  __syncthreads();
  if (threadIdx.x == 0 && write) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      Ext e = s_e[k][0];
      for (int w = 1; w < kBlock / 64; ++w) ext_merge(e, s_e[k][w]);
      vals[k] = e.lo; vals[3 + k] = e.hi; idx[k] = e.ilo; idx[3 + k] = e.ihi;
    }
  }
}

__global__ __launch_bounds__(kBlock) void bounds_partial(const float* __restrict__ z1, int64_t V, float* __restrict__ pv,
                                                         int64_t* __restrict__ pi) {
  const int64_t v0 = (int64_t)blockIdx.x * kRowsPerBlock;
  const int64_t v1 = v0 + kRowsPerBlock < V ? v0 + kRowsPerBlock : V;
  ext_block(v0, v1, [&](int64_t v, int k) { const float x = z1[v * 3 + k]; return Ext{x, x, v, v}; },
            pv + (int64_t)blockIdx.x * 6, pi + (int64_t)blockIdx.x * 6, true);
}

__global__ __launch_bounds__(kBlock) void bounds_finalize(const float* __restrict__ pv, const int64_t* __restrict__ pi, int64_t nb,
                                                          float* __restrict__ bounds, int64_t* __restrict__ arg) {
  ext_block(0, nb, [&](int64_t b, int k) { return Ext{pv[b * 6 + k], pv[b * 6 + 3 + k], pi[b * 6 + k], pi[b * 6 + 3 + k]}; },
            bounds, arg, true);
}

// the gradients of the bounds go to the vertices the bounds came from (one thread: six adds)
__global__ void bounds_route(const float* __restrict__ d_lo, const float* __restrict__ d_hi, const int64_t* __restrict__ arg,
                             float* __restrict__ dz1) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    dz1[arg[k] * 3 + k] += d_lo[k];
    dz1[arg[3 + k] * 3 + k] += d_hi[k];
  }
}

}  // namespace

int64_t input_prep_blocks(int64_t V) { return V <= 0 ? 0 : (V + kRowsPerBlock - 1) / kRowsPerBlock; }

int launch_input_bounds(const float* z1, int64_t V, float* pv, int64_t* pi, float* bounds, int64_t* arg, hipStream_t stream) {
  const int64_t nb = input_prep_blocks(V);
  bounds_partial<<<(int)nb, kBlock, 0, stream>>>(z1, V, pv, pi);
  SG_HIP_TRY(hipGetLastError());
  bounds_finalize<<<1, kBlock, 0, stream>>>(pv, pi, nb, bounds, arg);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

int launch_bounds_route(const float* d_lo, const float* d_hi, const int64_t* arg, float* dz1, hipStream_t stream) {
  bounds_route<<<1, 1, 0, stream>>>(d_lo, d_hi, arg, dz1);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

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
