// BatchNorm over the VERTEX axis fused with LeakyReLU, forward and backward, for gfx950.
//
// Replaces the nn.BatchNorm1d + nn.LeakyReLU pair that follows every ChebConv of the reference
// (util/networks.py:43-45,50-51; util/meshnet.py:41-62,107-128,226-243): batch = all V vertices of
// the mesh, eps 1e-5, biased variance for normalisation.  All four kernels are pure HBM streams
// over [V, C] row-major features (16 B per lane, whole rows per wavefront), so the pair costs
//   forward : 1 read (moments) + 1 read + 1 write (normalise+activate, optionally straight into a
//             column block of the next layer's [V, 3C] buffer)
//   backward: 2 reads (reduce) + 2 reads + 1 write (apply)
// against 5 + 8 passes for the separate ATen kernels plus the copy into the next layer's buffer.
// The LeakyReLU mask and x-hat are recomputed from the saved conv output; nothing extra is stored.
#include <type_traits>

#include "sg_common.h"

namespace sg {
namespace {

constexpr int kBlock = 256;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int DT> struct Io;
template <> struct Io<SG_F32> {
  static constexpr int VEC = 4;
  using raw = f32x4;
  using elem = float;
  static __device__ __forceinline__ void unpack(const raw& v, float* f) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
  static __device__ __forceinline__ raw pack(const float* f) { return raw{f[0], f[1], f[2], f[3]}; }
  static __device__ __forceinline__ float load1(const elem* p) { return *p; }
  static __device__ __forceinline__ void store1(elem* p, float v) { *p = v; }
};
template <> struct Io<SG_BF16> {
  static constexpr int VEC = 8;
  using raw = u32x4;
  using elem = uint16_t;
  static __device__ __forceinline__ void unpack(const raw& v, float* f) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ uint32_t cvt2(float lo, float hi) {
    return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)lo) | ((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)hi) << 16);
  }
  static __device__ __forceinline__ raw pack(const float* f) {
    return raw{cvt2(f[0], f[1]), cvt2(f[2], f[3]), cvt2(f[4], f[5]), cvt2(f[6], f[7])};
  }
  static __device__ __forceinline__ float load1(const elem* p) { return __uint_as_float((uint32_t)*p << 16); }
  static __device__ __forceinline__ void store1(elem* p, float v) { *p = __builtin_bit_cast(uint16_t, (__bf16)v); }
};

__device__ __forceinline__ float act_slope(float z, float slope) { return z > 0.f ? 1.0f : slope; }

// ---- column reductions --------------------------------------------------------------------
// Block b owns rows [b*rpb, (b+1)*rpb).  A row's C channels are spread over `tpr` threads
// (VEC channels each); the block's 256/tpr row groups are merged through LDS.
// MODE 0: out[b][0][c] = mean_b, out[b][1][c] = M2_b  of X
// MODE 1: out[b][0][c] = sum dz, out[b][1][c] = sum dz*xhat; dz = dA*act'(scale*h+shift), xhat=(h-mean)*invstd
template <int DT, int MODE, int VECW>
__global__ __launch_bounds__(kBlock) void col_reduce(const void* A_, int64_t lda, const void* H_, int64_t ldh,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     float slope, float* __restrict__ out, int64_t V, int C, int rpb,
                                                     int tpr) {
  using IO = Io<DT>;
  using elem_t = typename IO::elem;
  using raw_t = typename IO::raw;
  constexpr int VEC = VECW;  // IO::VEC (vector path) or 1 (scalar path)
  constexpr bool LEAN = DT == SG_BF16 && VECW > 1;
  __shared__ float s_red[2][kBlock][VECW];
  const elem_t* A = (const elem_t*)A_;
  const elem_t* H = (const elem_t*)H_;
  const int groups = kBlock / tpr;
  const int tg = threadIdx.x / tpr;   // row group of this thread
  const int tc = threadIdx.x % tpr;   // column slot
  const int64_t r_begin = (int64_t)blockIdx.x * rpb;
  int64_t r_end = r_begin + rpb;
  if (r_end > V) r_end = V;
  const float n_rows = (float)(r_end - r_begin);
  const int ncol = (C + VEC - 1) / VEC;   // column slots per row

  for (int c0 = 0; c0 < ncol; c0 += tpr) {      // one pass per 256-thread-wide column chunk (C > tpr*VEC)
    const int cs = c0 + tc;
    const bool cok = cs < ncol;
    float a0[VEC], a1[VEC];
    float sc[VEC], sh[VEC], mu[VEC], is[VEC];
    // MODE 0 accumulates deviations from the block's FIRST row: sum (x-K), sum (x-K)^2.  With K within a
    // few sigma of the mean the subtraction sum2 - sum^2/n no longer cancels (plain sums lose
    // eps*(mean^2/var) and showed up as 6x the reference's forward error).
    float kshift[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) kshift[k] = 0.f;
    if (MODE == 0 && cok && r_begin < r_end) {
      if (VEC == 1) kshift[0] = IO::load1(A + r_begin * lda + cs);
      else IO::unpack(*(const raw_t*)(A + r_begin * lda + (int64_t)cs * VEC), kshift);
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      a0[k] = 0.f; a1[k] = 0.f;
      const int c = cs * VEC + k;
      const bool ok = cok && c < C;
      if (MODE == 1) {
        sc[k] = ok ? scale[c] : 0.f; sh[k] = ok ? shift[c] : 0.f;
        mu[k] = ok ? mean[c] : 0.f; is[k] = ok ? invstd[c] : 0.f;
      }
    }
    if (cok) {
      // U rows in flight per thread (row ids past the end are clamped for the load and masked in the sum):
      // one 16-B load per thread per trip leaves the CU with too few bytes in flight to cover HBM latency
      constexpr int U = 4;
      for (int64_t r = r_begin + tg; r < r_end; r += (int64_t)groups * U) {
        float x[U][VEC], h[U][VEC];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int64_t ru = r + (int64_t)u * groups;
          live[u] = ru < r_end;
          ru = live[u] ? ru : r;
          if (VEC == 1) {
            x[u][0] = IO::load1(A + ru * lda + cs);
            if (MODE == 1) h[u][0] = IO::load1(H + ru * ldh + cs);
          } else {
            IO::unpack(*(const raw_t*)(A + ru * lda + (int64_t)cs * VEC), x[u]);
            if (MODE == 1) IO::unpack(*(const raw_t*)(H + ru * ldh + (int64_t)cs * VEC), h[u]);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float on = live[u] ? 1.f : 0.f;
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            if (MODE == 0) {
              const float d = (x[u][k] - kshift[k]) * on;
              a0[k] += d;
              a1[k] = fmaf(d, d, a1[k]);
            } else if (LEAN) {
              // bf16 rows carry twice the values per byte and this loop is VALU-bound there: the row mask is folded into
              // the operand (a dead row's dA is 0), invstd is applied once at the end
              const float xm = live[u] ? x[u][k] : 0.f;
              const float dz = fmaf(sc[k], h[u][k], sh[k]) > 0.f ? xm : xm * slope;
              a0[k] += dz;
              a1[k] = fmaf(dz, h[u][k] - mu[k], a1[k]);
            } else {
              const float dz = x[u][k] * act_slope(fmaf(sc[k], h[u][k], sh[k]), slope) * on;
              a0[k] += dz;
              a1[k] = fmaf(dz, (h[u][k] - mu[k]) * is[k], a1[k]);
            }
          }
        }
      }
    }
    if (MODE == 1 && LEAN) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) a1[k] *= is[k];
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) { s_red[0][threadIdx.x][k] = a0[k]; s_red[1][threadIdx.x][k] = a1[k]; }
    __syncthreads();   // (every row group of a column slot used the same shift K: the block's first row)
    for (int off = groups >> 1; off > 0; off >>= 1) {      // tree over the row groups (groups is a power of two)
      if (tg < off) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          s_red[0][threadIdx.x][k] += s_red[0][threadIdx.x + off * tpr][k];
          s_red[1][threadIdx.x][k] += s_red[1][threadIdx.x + off * tpr][k];
        }
      }
      __syncthreads();
    }
    if (tg == 0 && cok) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float s0 = s_red[0][tc][k], s1 = s_red[1][tc][k];
        const int c = cs * VEC + k;
        if (c < C) {
          if (MODE == 0) {
            const float m = s0 / n_rows;                                          // mean of (x - K)
            out[((int64_t)blockIdx.x * 2 + 0) * C + c] = kshift[k] + m;
            out[((int64_t)blockIdx.x * 2 + 1) * C + c] = fmaxf(s1 - s0 * m, 0.f);   // sum (x-mean)^2 over <= rpb rows
          } else {
            out[((int64_t)blockIdx.x * 2 + 0) * C + c] = s0;
            out[((int64_t)blockIdx.x * 2 + 1) * C + c] = s1;
          }
        }
      }
    }
    __syncthreads();
  }
}

// ---- elementwise --------------------------------------------------------------------------
// MODE 0: Y = act(scale*X + shift)
// MODE 1: dH = k*(dz - c1 - xhat*c2), dz = dA*act'(scale*H+shift), xhat = (H-mean)*invstd
template <int DT, int MODE, int VECW>
__global__ __launch_bounds__(kBlock) void col_apply(const void* A_, int64_t lda, const void* H_, int64_t ldh,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                    const float* __restrict__ kk, const float* __restrict__ c1,
                                                    const float* __restrict__ c2, float slope, void* Y_, int64_t ldy,
                                                    int64_t V, int C) {
  using IO = Io<DT>;
  using elem_t = typename IO::elem;
  using raw_t = typename IO::raw;
  constexpr int VEC = VECW;
  const elem_t* A = (const elem_t*)A_;
  const elem_t* H = (const elem_t*)H_;
  elem_t* Y = (elem_t*)Y_;
  const int ncol = (C + VEC - 1) / VEC;
  const int64_t total = V * ncol;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    const int64_t r = t / ncol;
    const int cs = (int)(t - r * ncol);
    float x[VEC], h[VEC], y[VEC];
    if (VEC == 1) {
      x[0] = IO::load1(A + r * lda + cs);
      if (MODE == 1) h[0] = IO::load1(H + r * ldh + cs);
    } else {
      IO::unpack(*(const raw_t*)(A + r * lda + (int64_t)cs * VEC), x);
      if (MODE == 1) IO::unpack(*(const raw_t*)(H + r * ldh + (int64_t)cs * VEC), h);
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int c = cs * VEC + k;
      if (MODE == 0) {
        const float z = fmaf(scale[c], x[k], shift[c]);
        y[k] = z > 0.f ? z : z * slope;
      } else {
        const float dz = x[k] * act_slope(fmaf(scale[c], h[k], shift[c]), slope);
        y[k] = kk[c] * (dz - c1[c] - (h[k] - mean[c]) * invstd[c] * c2[c]);
      }
    }
    if (VEC == 1) IO::store1(Y + r * ldy + cs, y[0]);
    else *(raw_t*)(Y + r * ldy + (int64_t)cs * VEC) = IO::pack(y);
  }
}

// Same arithmetic as col_apply, for C / VEC <= 256 column slots: a thread keeps ONE column slot for the
// whole launch, so the per-channel parameters live in registers (col_apply re-reads 2 (MODE 0) or 7
// (MODE 1) vectors and does a 64-bit division per element group), and it has U rows in flight.
// COLSUM (MODE 1 only): also leave colsum[blockIdx][c] = this workgroup's column sums of the ROUNDED values it wrote --
// dH is the output gradient of the ChebConv in front of the BatchNorm, and that layer's bias gradient is exactly these
// column sums: taking them here saves the separate pass over dH (and two launches) per layer.
template <int DT, int MODE, bool COLSUM = false>
__global__ __launch_bounds__(kBlock) void col_apply_rows(const void* A_, int64_t lda, const void* H_, int64_t ldh,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ kk, const float* __restrict__ c1,
                                                         const float* __restrict__ c2, float slope, void* Y_, int64_t ldy,
                                                         int64_t V, int C, int tpr_log2, float* __restrict__ colsum = nullptr,
                                                         int64_t rows_per_block = 0) {
  using IO = Io<DT>;
  using elem_t = typename IO::elem;
  using raw_t = typename IO::raw;
  constexpr int VEC = IO::VEC;
  constexpr int U = (MODE == 0 && DT == SG_F32) ? 4 : 2;   // rows in flight per thread (bf16 rows carry 8 values per vector)
  const elem_t* A = (const elem_t*)A_;
  const elem_t* H = (const elem_t*)H_;
  elem_t* Y = (elem_t*)Y_;
  const int tc = threadIdx.x & ((1 << tpr_log2) - 1);
  const int tg = threadIdx.x >> tpr_log2;
  const int groups = kBlock >> tpr_log2;
  __shared__ float s_sum[COLSUM ? kBlock : 1][COLSUM ? VEC : 1];
  float csum[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) csum[k] = 0.f;
  const bool col_on = tc < C / VEC;
  if (!COLSUM && !col_on) return;
  const int c0 = (col_on ? tc : 0) * VEC;
  // bf16 (LEAN): dH = k (dz - c1 - xhat c2) regrouped per channel as  x t - A - B h  with t = k or k slope by the sign of
  // the BatchNorm output, B = k c2 invstd, A = k c1 - B mean: 5 VALU operations per value instead of 8 (this pass is close
  // to VALU-bound with 8 values per 16-byte vector); the result is rounded to bf16 (2^-9) right after
  constexpr bool LEAN = MODE == 1 && DT == SG_BF16;
  float sc[VEC], sh[VEC], mu[VEC], is[VEC], kc[VEC], k1[VEC], k2[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    sc[k] = scale[c0 + k];
    sh[k] = shift[c0 + k];
    if (MODE == 1) {
      mu[k] = mean[c0 + k];
      is[k] = invstd[c0 + k];
      kc[k] = kk[c0 + k];
      k1[k] = c1[c0 + k];
      k2[k] = c2[c0 + k] * is[k];
      if (LEAN) {
        const float B = kc[k] * k2[k];
        k2[k] = -B;                          // -B
        k1[k] = fmaf(B, mu[k], -kc[k] * k1[k]);   // -A = B mean - k c1
        is[k] = kc[k] * slope;               // t for a negative BatchNorm output
      }
    }
  }
  // rows of a workgroup: rows_per_block == 0: row groups strided over the grid (block b: b, b + gridDim, ..); > 0: ONE
  // contiguous range per workgroup, walked front to back (what a plain copy kernel does: every workgroup is a sequential stream)
  const int64_t step = rows_per_block ? groups : (int64_t)gridDim.x * groups;
  const int64_t r_first = rows_per_block ? (int64_t)blockIdx.x * rows_per_block : (int64_t)blockIdx.x * groups;
  int64_t r_end = rows_per_block ? r_first + rows_per_block : V;
  r_end = r_end < V ? r_end : V;
  for (int64_t r = r_first + tg; col_on && r < r_end; r += step * U) {
    raw_t xa[U], xh[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t ru = r + u * step;
      live[u] = ru < r_end;
      ru = live[u] ? ru : r;
      xa[u] = *(const raw_t*)(A + ru * lda + c0);
      if (MODE == 1) xh[u] = *(const raw_t*)(H + ru * ldh + c0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float x[VEC], h[VEC], y[VEC];
      IO::unpack(xa[u], x);
      if (MODE == 1) IO::unpack(xh[u], h);
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if (MODE == 0) {
          const float z = fmaf(sc[k], x[k], sh[k]);
          y[k] = z > 0.f ? z : z * slope;
        } else if (LEAN) {
          const float t = fmaf(sc[k], h[k], sh[k]) > 0.f ? kc[k] : is[k];
          y[k] = fmaf(x[k], t, fmaf(k2[k], h[k], k1[k]));
        } else {
          const float dz = x[k] * act_slope(fmaf(sc[k], h[k], sh[k]), slope);
          y[k] = kc[k] * (dz - k1[k] - (h[k] - mu[k]) * k2[k]);
        }
      }
      if (live[u]) {
        const raw_t packed = IO::pack(y);
        *(raw_t*)(Y + (r + u * step) * ldy + c0) = packed;
        if (COLSUM) {
          float yr[VEC];
          IO::unpack(packed, yr);                      // the values as stored (bf16: rounded)
#pragma unroll
          for (int k = 0; k < VEC; ++k) csum[k] += yr[k];
        }
      }
    }
  }
  if (COLSUM) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) s_sum[threadIdx.x][k] = csum[k];
    __syncthreads();
    if (tg == 0 && col_on) {                         // fixed order over the row groups: deterministic
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        float t = 0.f;
        for (int gq = 0; gq < groups; ++gq) t += s_sum[(gq << tpr_log2) + tc][k];
        colsum[(int64_t)blockIdx.x * C + c0 + k] = t;
      }
    }
  }
}

// Sum of per-block partial values of ONE channel over nb blocks by a whole workgroup (256 threads), in double and in a
// fixed order (thread t takes blocks t, t + 256, ..; a shuffle tree per wavefront; thread 0 adds the four wavefront
// sums): valid in thread 0.  One wavefront per channel walked up to 4096 strided partials in 64 dependent steps -- 17-28 us
// per launch, 26 launches per training iteration.
__device__ inline double block_sum_256(double v, double* s_w /*[4]*/) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// out[c] = sum over the nb workgroup partials of col_apply_rows<.., COLSUM>; one workgroup per channel
__global__ __launch_bounds__(256) void colsum_finalize(const float* __restrict__ partial, int64_t nb, int C,
                                                       float* __restrict__ out, float* __restrict__ acc) {
  __shared__ double s_w[4];
  const int c = blockIdx.x;
  double s0 = 0.0;
  for (int64_t b = threadIdx.x; b < nb; b += 256) s0 += partial[b * C + c];
  s0 = block_sum_256(s0, s_w);
  if (threadIdx.x == 0) {
    out[c] = (float)s0;
    if (acc) acc[c] += (float)s0;       // (the conv bias' .grad accumulator: no add launch of its own)
  }
}

// stats[0][c] = mean, stats[1][c] = M2 over all V rows, from the per-block partials (Chan et al.,
// in double).  One wavefront per channel: each lane merges every 64th block, then a shuffle tree merges lanes.
__global__ __launch_bounds__(256) void bn_merge(const float* __restrict__ partial, int64_t nb, int64_t V, int C,
                                                int rpb, float* __restrict__ stats, float* count_out) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  double n = 0.0, mean = 0.0, m2 = 0.0;
  if (c < C) {
    for (int64_t b = lane; b < nb; b += 64) {
      int64_t rows = V - b * rpb;
      rows = rows > rpb ? rpb : rows;
      if (rows <= 0) break;
      const double nbk = (double)rows, mb = partial[(b * 2 + 0) * C + c], qb = partial[(b * 2 + 1) * C + c];
      const double tot = n + nbk, delta = mb - mean;
      mean += delta * (nbk / tot);
      m2 += qb + delta * delta * (n * nbk / tot);
      n = tot;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {       // lanes l and l+off sit in the same 32-lane half of the wavefront
    const double n2 = __shfl_down(n, off, 64), mean2 = __shfl_down(mean, off, 64), m22 = __shfl_down(m2, off, 64);
    const double tot = n + n2;
    if (tot > 0.0) {
      const double delta = mean2 - mean;
      mean += delta * (n2 / tot);
      m2 += m22 + delta * delta * (n * n2 / tot);
      n = tot;
    }
  }
  if (c < C && lane == 0) {
    stats[c] = (float)mean;
    stats[C + c] = (float)m2;
    if (count_out && c == 0) count_out[0] = (float)V;      // the row count rides along (a partition's all-gather row)
  }
}

// From (mean, M2, N): invstd, scale = gamma*invstd, shift = beta - mean*scale, and the running
// statistics update of nn.BatchNorm1d (unbiased variance), one thread per channel.
__global__ void bn_finalize(const float* __restrict__ stats, double N, int C, const float* __restrict__ gamma,
                            const float* __restrict__ beta, float* running_mean, float* running_var,
                            float momentum, float eps, float* __restrict__ out /*[4][C]: mean invstd scale shift*/) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mean = stats[c];
  const double m2 = stats[C + c];
  const float var = (float)(m2 / N);
  const float invstd = rsqrtf(var + eps);
  const float scale = gamma[c] * invstd;
  out[c] = mean;
  out[C + c] = invstd;
  out[2 * C + c] = scale;
  out[3 * C + c] = beta[c] - mean * scale;
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(m2 / (N - 1.0));
  }
}

// bn_merge + bn_finalize in one launch for the single-device case (N = V), and the per-channel coefficients of the
// backward pass from the partial sums of col_reduce<MODE 1>: two tiny kernels instead of six launches per
// BatchNorm (merge, finalize; sum over blocks, /N, gamma*invstd) -- they matter where an iteration is launch-bound.
// One WORKGROUP per channel (the MFMA product hands over one partial per 128-row tile: 7813 of them at V = 1 M, which a
// single wavefront per channel took 36 us to walk): every thread merges every 256th block, a shuffle tree merges the
// lanes of a wavefront, thread 0 merges the four wavefronts -- a fixed order, so the result is deterministic.
__global__ __launch_bounds__(256) void bn_stats_finalize(const float* __restrict__ partial, int64_t nb, int64_t V, int C,
                                                         int rpb, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* running_mean,
                                                         float* running_var, float momentum, float eps,
                                                         float* __restrict__ out /*[4][C]*/,
                                                         long long* batches_tracked) {
  __shared__ double s_n[4], s_mean[4], s_m2[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x;
  double n = 0.0, mean = 0.0, m2 = 0.0;
  for (int64_t b = threadIdx.x; b < nb; b += 256) {
    int64_t rows = V - b * rpb;
    rows = rows > rpb ? rpb : rows;
    if (rows <= 0) break;
    const double nbk = (double)rows, mb = partial[(b * 2 + 0) * C + c], qb = partial[(b * 2 + 1) * C + c];
    const double tot = n + nbk, delta = mb - mean;
    mean += delta * (nbk / tot);
    m2 += qb + delta * delta * (n * nbk / tot);
    n = tot;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double n2 = __shfl_down(n, off, 64), mean2 = __shfl_down(mean, off, 64), m22 = __shfl_down(m2, off, 64);
    const double tot = n + n2;
    if (tot > 0.0) {
      const double delta = mean2 - mean;
      mean += delta * (n2 / tot);
      m2 += m22 + delta * delta * (n * n2 / tot);
      n = tot;
    }
  }
  if (lane == 0) { s_n[wave] = n; s_mean[wave] = mean; s_m2[wave] = m2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      const double n2 = s_n[w], tot = n + n2;
      if (tot > 0.0 && n2 > 0.0) {
        const double delta = s_mean[w] - mean;
        mean += delta * (n2 / tot);
        m2 += s_m2[w] + delta * delta * (n * n2 / tot);
        n = tot;
      }
    }
    // through float, exactly as bn_merge hands (mean, M2) to bn_finalize
    const float meanf = (float)mean;
    const double m2d = (double)(float)m2;
    const double N = (double)V;
    const float invstd = rsqrtf((float)(m2d / N) + eps);
    const float scale = gamma[c] * invstd;
    out[c] = meanf;
    out[C + c] = invstd;
    out[2 * C + c] = scale;
    out[3 * C + c] = beta[c] - meanf * scale;
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(m2d / (N - 1.0));
    }
    if (batches_tracked && c == 0) *batches_tracked += 1;      // nn.BatchNorm1d.num_batches_tracked, without a launch of its own
  }
}

// out[0] = sum dz, out[1] = sum dz*xhat (the bias / weight gradients), out[2] = out[0]/N, out[3] = out[1]/N, out[4] = gamma*invstd;
// acc_dweight / acc_dbias (optional): the parameters' gradient accumulators, += out[1] / out[0]
__global__ __launch_bounds__(256) void bn_bwd_coeffs(const float* __restrict__ partial, int64_t nb, int C, double N,
                                                     const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                     float* __restrict__ out /*[5][C]*/, float* acc_dweight,
                                                     float* acc_dbias, const float* __restrict__ count_dev) {
  __shared__ double s_w0[4], s_w1[4];
  if (count_dev) N = (double)count_dev[0];       // the mesh-wide row count of a partition, kept on the device
  const int c = blockIdx.x;                      // one workgroup per channel
  double s0 = 0.0, s1 = 0.0;
  for (int64_t b = threadIdx.x; b < nb; b += 256) {
    s0 += partial[(b * 2 + 0) * C + c];
    s1 += partial[(b * 2 + 1) * C + c];
  }
  s0 = block_sum_256(s0, s_w0);
  s1 = block_sum_256(s1, s_w1);
  if (threadIdx.x == 0) {
    const float f0 = (float)s0, f1 = (float)s1;
    out[c] = f0;
    out[C + c] = f1;
    out[2 * C + c] = (float)((double)f0 / N);
    out[3 * C + c] = (float)((double)f1 / N);
    out[4 * C + c] = gamma[c] * invstd[c];
    if (acc_dweight) acc_dweight[c] += f1;
    if (acc_dbias) acc_dbias[c] += f0;
  }
}

// Mesh-wide statistics on a vertex partition: every rank's (mean[C], M2[C], row count) row of `all` [world, 2C+1]
// (after an all-gather) is merged per channel (Chan et al., in double) and finalised like bn_finalize, with the
// total row count N taken from the gathered counts ON THE DEVICE (out_n[0] = N: the backward pass divides by it
// as a device scalar too), so the partitioned BatchNorm needs no host round trip.  One thread per channel.
__global__ void bn_finalize_ranks(const float* __restrict__ all, int world, int C, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, float* running_mean, float* running_var,
                                  float momentum, float eps, float* __restrict__ out /*[4][C]*/,
                                  float* __restrict__ out_n, long long* batches_tracked) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double n = 0.0, mean = 0.0, m2 = 0.0;
  for (int r = 0; r < world; ++r) {
    const float* row = all + (int64_t)r * (2 * C + 1);
    const double nr = row[2 * C];
    if (nr <= 0.0) continue;
    const double tot = n + nr, delta = (double)row[c] - mean;
    mean += delta * (nr / tot);
    m2 += (double)row[C + c] + delta * delta * (n * nr / tot);
    n = tot;
  }
  const float meanf = (float)mean;
  const double m2d = (double)(float)m2;
  const float invstd = rsqrtf((float)(m2d / n) + eps);
  const float scale = gamma[c] * invstd;
  out[c] = meanf;
  out[C + c] = invstd;
  out[2 * C + c] = scale;
  out[3 * C + c] = beta[c] - meanf * scale;
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(m2d / (n - 1.0));
  }
  if (c == 0) {
    out_n[0] = (float)n;
    if (batches_tracked) *batches_tracked += 1;
  }
}

// dst_s[r][c] += src_s[r * ld_s + c] for up to kMultiAddMax small fp32 matrices in one launch: a layer's parameter
// gradients (column / row blocks of dWcat, the bias sums) added into the parameters' .grad accumulators
struct MultiAddArgs {
  const float* src[kMultiAddMax];
  float* dst[kMultiAddMax];
  int64_t ld[kMultiAddMax];
  int cols[kMultiAddMax];
  int64_t start[kMultiAddMax + 1];   // element offsets of the segments in the flattened index space
  int n;
};

__global__ __launch_bounds__(256) void multi_add(MultiAddArgs a) {
  const int64_t total = a.start[a.n];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int s = 0;
#pragma unroll
    for (int k = 1; k < kMultiAddMax; ++k) s += (k < a.n && i >= a.start[k]) ? 1 : 0;
    const int64_t e = i - a.start[s];
    const int cols = a.cols[s];
    const int64_t r = e / cols;
    const int c = (int)(e - r * cols);
    a.dst[s][e] += a.src[s][r * a.ld[s] + c];
  }
}

inline bool a16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int threads_per_row(int ncol) {
  int t = 1;
  while (t < ncol && t < kBlock) t <<= 1;
  return t;
}

}  // namespace

// 128 rows per block up to 2048 blocks: a 125 K-row block of a partitioned mesh still gets ~1000 workgroups (with 512
// rows per block it got 245 -- less than one per CU -- and the moment passes of a 50 K mesh ran at a seventh of the
// streaming rate); at V = 1 M the cap decides either way
int64_t col_blocks(int64_t V) {
  int64_t nb = (V + 127) / 128;
  if (nb < 1) nb = 1;
  if (nb > 2048) nb = 2048;
  return nb;
}

int launch_bn_merge(const float* partial, int64_t nb, int64_t V, int64_t C, float* stats, hipStream_t stream) {
  if (C == 0) return SG_OK;
  SG_REQUIRE(nb == col_blocks(V), "partial buffer must have sg_col_blocks(V) blocks");
  const int rpb = (int)((V + nb - 1) / nb);
  bn_merge<<<(int)((C + 3) / 4), 256, 0, stream>>>(partial, nb, V, (int)C, rpb, stats, nullptr);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

int launch_bn_merge_tiles(const float* partial, int64_t nb, int64_t rpb, int64_t V, int64_t C, float* stats,
                          float* count_out, hipStream_t stream) {
  if (C == 0) return SG_OK;
  // (at least: sg_col_moments' capped block count can leave a few empty blocks at the end, which the kernel skips)
  SG_REQUIRE(rpb > 0 && rpb <= INT32_MAX && nb >= (V + rpb - 1) / rpb, "partial buffer must have >= ceil(V / rows_per_tile) tiles");
  bn_merge<<<(int)((C + 3) / 4), 256, 0, stream>>>(partial, nb, V, (int)C, (int)rpb, stats, count_out);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

int launch_bn_stats_finalize(const float* partial, int64_t nb, int64_t V, int64_t C, const float* gamma,
                             const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                             float* out, int64_t* batches_tracked, hipStream_t stream) {
  if (C == 0) return SG_OK;
  SG_REQUIRE(nb == col_blocks(V), "partial buffer must have sg_col_blocks(V) blocks");
  const int rpb = (int)((V + nb - 1) / nb);
  bn_stats_finalize<<<(int)C, 256, 0, stream>>>(partial, nb, V, (int)C, rpb, gamma, beta, running_mean, running_var,
                                                momentum, eps, out, (long long*)batches_tracked);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

int launch_bn_stats_finalize_tiles(const float* partial, int64_t nb, int64_t rpb, int64_t V, int64_t C, const float* gamma,
                                   const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                   float* out, int64_t* batches_tracked, hipStream_t stream) {
  if (C == 0) return SG_OK;
  SG_REQUIRE(rpb > 0 && rpb <= INT32_MAX && nb == (V + rpb - 1) / rpb, "partial buffer must have ceil(V / rows_per_tile) tiles");
  bn_stats_finalize<<<(int)C, 256, 0, stream>>>(partial, nb, V, (int)C, (int)rpb, gamma, beta, running_mean, running_var,
                                                momentum, eps, out, (long long*)batches_tracked);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

int launch_multi_add(int n, const float* const* srcs, const int64_t* src_ld, const int64_t* rows, const int64_t* cols,
                     float* const* dsts, hipStream_t stream) {
  MultiAddArgs a{};
  a.n = n;
  int64_t at = 0;
  for (int s = 0; s < n; ++s) {
    a.src[s] = srcs[s];
    a.dst[s] = dsts[s];
    a.ld[s] = src_ld[s];
    a.cols[s] = (int)(cols[s] > 0 ? cols[s] : 1);
    a.start[s] = at;
    at += rows[s] * cols[s];
  }
  a.start[n] = at;
  if (at == 0) return SG_OK;
  int64_t blocks = (at + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  multi_add<<<(int)blocks, 256, 0, stream>>>(a);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

int launch_bn_bwd_coeffs(const float* partial, int64_t nb, int64_t C, double N, const float* gamma,
                         const float* invstd, float* out, float* acc_dweight, float* acc_dbias, const float* count_dev,
                         hipStream_t stream) {
  if (C == 0) return SG_OK;
  bn_bwd_coeffs<<<(int)C, 256, 0, stream>>>(partial, nb, (int)C, N, gamma, invstd, out, acc_dweight, acc_dbias,
                                                        count_dev);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

int launch_bn_finalize_ranks(const float* all, int64_t world, int64_t C, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float momentum, float eps, float* out,
                             float* out_n, int64_t* batches_tracked, hipStream_t stream) {
  if (C == 0) return SG_OK;
  bn_finalize_ranks<<<(int)((C + 127) / 128), 128, 0, stream>>>(all, (int)world, (int)C, gamma, beta, running_mean,
                                                               running_var, momentum, eps, out, out_n,
                                                               (long long*)batches_tracked);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

int launch_bn_finalize(const float* stats, double N, int64_t C, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float momentum, float eps, float* out,
                       hipStream_t stream) {
  if (C == 0) return SG_OK;
  bn_finalize<<<(int)((C + 127) / 128), 128, 0, stream>>>(stats, N, (int)C, gamma, beta, running_mean, running_var,
                                                         momentum, eps, out);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

int launch_col_reduce(int mode, const void* A, int64_t lda, const void* H, int64_t ldh, const float* scale,
                      const float* shift, const float* mean, const float* invstd, float slope, float* out,
                      int64_t nblk, int64_t V, int64_t C, int dtype, hipStream_t stream) {
  if (V == 0 || C == 0) return SG_OK;
  SG_REQUIRE(nblk == col_blocks(V), "partial buffer must have sg_col_blocks(V) = %lld blocks", (long long)col_blocks(V));
  const int rpb = (int)((V + nblk - 1) / nblk);
  auto run = [&](auto dt_tag) -> int {
    constexpr int DT = decltype(dt_tag)::value;
    constexpr int VEC = Io<DT>::VEC;
    const bool vec = C % VEC == 0 && lda % VEC == 0 && a16(A) && (mode == 0 || (ldh % VEC == 0 && a16(H)));
    const int ncol = vec ? (int)(C / VEC) : (int)C;
    const int tpr = threads_per_row(ncol);
#define SG_LAUNCH(MODE, VW) col_reduce<DT, MODE, VW><<<(int)nblk, kBlock, 0, stream>>>(A, lda, H, ldh, scale, shift, mean, invstd, slope, out, V, (int)C, rpb, tpr)
    if (mode == 0) { if (vec) SG_LAUNCH(0, VEC); else SG_LAUNCH(0, 1); }
    else { if (vec) SG_LAUNCH(1, VEC); else SG_LAUNCH(1, 1); }
#undef SG_LAUNCH
    SG_HIP_TRY(hipGetLastError());
    return SG_OK;
  };
  if (dtype == SG_F32) return run(std::integral_constant<int, SG_F32>{});
  if (dtype == SG_BF16) return run(std::integral_constant<int, SG_BF16>{});
  set_error("unsupported dtype %d", dtype);
  return SG_ERR_UNSUPPORTED;
}

int64_t col_apply_blocks(int64_t V, int64_t C, int dtype) {
  const int VEC = dtype == SG_F32 ? 4 : 8;
  if (C % VEC || C / VEC > kBlock || V <= 0) return 0;     // the row-owning kernel does not serve this shape
  const int ncol = (int)(C / VEC);
  int lg = 0;
  while ((1 << lg) < ncol) ++lg;
  const int groups = kBlock >> lg;
  int64_t nbr = (V + (int64_t)groups * 4 - 1) / ((int64_t)groups * 4);
  if (nbr > 256 * 16) nbr = 256 * 16;
  return nbr < 1 ? 1 : nbr;
}

int launch_colsum_finalize(const float* partial, int64_t nb, int64_t C, float* out, hipStream_t stream, float* acc) {
  if (C == 0) return SG_OK;
  colsum_finalize<<<(int)C, 256, 0, stream>>>(partial, nb, (int)C, out, acc);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

// SG_TUNE_BN_ROWS: 0 = by shape (below), 1 = every workgroup of col_apply_rows walks one contiguous range of rows, 2 = row
// groups strided over the grid (A/B)
int g_bn_rows_contiguous = 0;
int set_bn_rows_tuning(int value) {
  g_bn_rows_contiguous = value;
  return SG_OK;
}

int launch_col_apply(int mode, const void* A, int64_t lda, const void* H, int64_t ldh, const float* scale,
                     const float* shift, const float* mean, const float* invstd, const float* kk, const float* c1,
                     const float* c2, float slope, void* Y, int64_t ldy, int64_t V, int64_t C, int dtype,
                     hipStream_t stream, float* colsum) {
  if (V == 0 || C == 0) return SG_OK;
  auto run = [&](auto dt_tag) -> int {
    constexpr int DT = decltype(dt_tag)::value;
    constexpr int VEC = Io<DT>::VEC;
    const bool vec = C % VEC == 0 && lda % VEC == 0 && ldy % VEC == 0 && a16(A) && a16(Y) &&
                     (mode == 0 || (ldh % VEC == 0 && a16(H)));
    if (vec && C / VEC <= kBlock) {
      const int ncol = (int)(C / VEC);
      int lg = 0;
      while ((1 << lg) < ncol) ++lg;
      const int groups = kBlock >> lg;
      int64_t nbr = (V + (int64_t)groups * 4 - 1) / ((int64_t)groups * 4);
      if (nbr > 256 * 16) nbr = 256 * 16;
      if (nbr < 1) nbr = 1;
      if (colsum) SG_REQUIRE(mode == 1 && nbr == col_apply_blocks(V, C, dtype), "column sums: wrong partial buffer size");
      // rows of a workgroup: one contiguous range where it pays (measured at V = 1 M, tools/bn_bench.py --rows 0 / 1, GB/s:
      // fp32 C = 256 apply 4 540 -> 5 290, backward 4 810 -> 5 300; fp32 C = 512 4 530 -> 5 230 / 4 600 -> 5 300; bf16 C = 512
      // backward 4 820 -> 5 400; bf16 C = 256 forward 5 470 -> 5 345: not there; 128-byte rows level), else strided over the grid
      const int64_t row_bytes = C * (int64_t)sizeof(typename Io<DT>::elem);
      const bool contiguous = g_bn_rows_contiguous == 1 || (g_bn_rows_contiguous == 0 && (row_bytes >= 1024 || (mode == 1 && row_bytes >= 512)));
      int64_t rpb = 0;
      if (contiguous) rpb = ((V + nbr - 1) / nbr + groups - 1) / groups * groups;
      if (mode == 0) col_apply_rows<DT, 0><<<(int)nbr, kBlock, 0, stream>>>(A, lda, H, ldh, scale, shift, mean, invstd, kk, c1, c2, slope, Y, ldy, V, (int)C, lg, nullptr, rpb);
      else if (colsum) col_apply_rows<DT, 1, true><<<(int)nbr, kBlock, 0, stream>>>(A, lda, H, ldh, scale, shift, mean, invstd, kk, c1, c2, slope, Y, ldy, V, (int)C, lg, colsum, rpb);
      else col_apply_rows<DT, 1><<<(int)nbr, kBlock, 0, stream>>>(A, lda, H, ldh, scale, shift, mean, invstd, kk, c1, c2, slope, Y, ldy, V, (int)C, lg, nullptr, rpb);
      SG_HIP_TRY(hipGetLastError());
      return SG_OK;
    }
    SG_REQUIRE(colsum == nullptr, "column sums are only taken by the row-owning kernel (sg_col_apply_blocks() > 0)");
    const int64_t total = V * (vec ? C / VEC : C);
    int64_t nb = (total + kBlock - 1) / kBlock;
    if (nb > 256 * 16) nb = 256 * 16;
#define SG_LAUNCH(MODE, VW) col_apply<DT, MODE, VW><<<(int)nb, kBlock, 0, stream>>>(A, lda, H, ldh, scale, shift, mean, invstd, kk, c1, c2, slope, Y, ldy, V, (int)C)
    if (mode == 0) { if (vec) SG_LAUNCH(0, VEC); else SG_LAUNCH(0, 1); }
    else { if (vec) SG_LAUNCH(1, VEC); else SG_LAUNCH(1, 1); }
#undef SG_LAUNCH
    SG_HIP_TRY(hipGetLastError());
    return SG_OK;
  };
  if (dtype == SG_F32) return run(std::integral_constant<int, SG_F32>{});
  if (dtype == SG_BF16) return run(std::integral_constant<int, SG_BF16>{});
  set_error("unsupported dtype %d", dtype);
  return SG_ERR_UNSUPPORTED;
}

}  // namespace sg
