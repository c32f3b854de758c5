// Per-iteration geometry + losses of the training step, fused.
//
// Replaces, for the position tensor the network just produced:
//   Models.compute_fn          util/models.py:121-126   unit face normals  n = (b-a)x(c-a) / |.|
//   Loss.mask_pos_rec_loss     util/loss.py:14-34       ltype 'rmse' : sqrt(sum_kept |p - t|^2 / n_v + 1e-6)
//   Loss.mask_norm_rec_loss    util/loss.py:78-107      ltype 'l1mae': sum_kept |n - n_t|_1 / n_f
// called right after every forward (sgcn.py:130-132).  The kernels produce the two masked SUMS
// (S_p, S_n) and their gradient w.r.t. the positions; the scalar tail (sqrt, division by the
// counts, k1 weighting, cross-rank all-reduce) stays in the host code.  ~15 small ATen kernels and
// an index_put-with-accumulate backward become three launches over [V,3] / [F,3].
#include "sg_common.h"

namespace sg {
namespace {

constexpr int kBlock = 256;

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 ld3(const float* p, int64_t i) { return F3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
__device__ __forceinline__ F3 sub(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 cross(F3 a, F3 b) { return F3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float sgn(float v) { return (float)((v > 0.f) - (v < 0.f)); }

__device__ __forceinline__ float block_sum(float v, float* s_buf) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) s_buf[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0) for (int w = 0; w < kBlock / 64; ++w) t += s_buf[w];
  __syncthreads();
  return t;
}

// partial[b][0] = sum over this block's vertices of keep*|p - t|^2, partial[b][1] = faces: keep*|n - n_t|_1
__global__ __launch_bounds__(kBlock) void mesh_loss_fwd(const float* __restrict__ pos, const int64_t* __restrict__ faces,
                                                        const float* __restrict__ tpos, const float* __restrict__ vkeep,
                                                        const float* __restrict__ tfn, const float* __restrict__ fkeep,
                                                        int64_t V, int64_t F, float* __restrict__ partial) {
  __shared__ float s_buf[kBlock / 64];
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  float sp = 0.f, sn = 0.f;
  if (i < V) {
    const float k = vkeep[i];
    if (k != 0.f) {
      const F3 d = sub(ld3(tpos, i), ld3(pos, i));
      sp = k * dot(d, d);
    }
  }
  if (i < F) {
    const float k = fkeep[i];
    if (k != 0.f) {
      const F3 a = ld3(pos, faces[3 * i]), b = ld3(pos, faces[3 * i + 1]), c = ld3(pos, faces[3 * i + 2]);
      const F3 cr = cross(sub(b, a), sub(c, a));
      const float inv = 1.0f / sqrtf(dot(cr, cr));
      const F3 t = ld3(tfn, i);
      sn = k * (fabsf(cr.x * inv - t.x) + fabsf(cr.y * inv - t.y) + fabsf(cr.z * inv - t.z));
    }
  }
  const float a = block_sum(sp, s_buf);
  const float b = block_sum(sn, s_buf);
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = a; partial[2 * blockIdx.x + 1] = b; }
}

// grad[v] = g[0] * 2 * keep * (p - t) for owned vertices, 0 for halo rows (V <= v < V_ext)
__global__ __launch_bounds__(kBlock) void mesh_loss_bwd_vertex(const float* __restrict__ pos, const float* __restrict__ tpos,
                                                               const float* __restrict__ vkeep, const float* __restrict__ g,
                                                               int64_t V, int64_t V_ext, float* __restrict__ grad) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= V_ext) return;
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (i < V) {
    const float k = 2.0f * g[0] * vkeep[i];
    if (k != 0.f) {
      const F3 d = sub(ld3(pos, i), ld3(tpos, i));
      gx = k * d.x; gy = k * d.y; gz = k * d.z;
    }
  }
  grad[3 * i] = gx; grad[3 * i + 1] = gy; grad[3 * i + 2] = gz;
}

// grad[a,b,c] += d(g[1] * keep * |n - n_t|_1)/d(a,b,c)   (float atomics: 9 per kept face)
__global__ __launch_bounds__(kBlock) void mesh_loss_bwd_face(const float* __restrict__ pos, const int64_t* __restrict__ faces,
                                                             const float* __restrict__ tfn, const float* __restrict__ fkeep,
                                                             const float* __restrict__ g, int64_t F, float* __restrict__ grad) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= F) return;
  const float k = g[1] * fkeep[i];
  if (k == 0.f) return;
  const int64_t ia = faces[3 * i], ib = faces[3 * i + 1], ic = faces[3 * i + 2];
  const F3 a = ld3(pos, ia), b = ld3(pos, ib), c = ld3(pos, ic);
  const F3 e1 = sub(b, a), e2 = sub(c, a);
  const F3 cr = cross(e1, e2);
  const float inv = 1.0f / sqrtf(dot(cr, cr));
  const F3 n = F3{cr.x * inv, cr.y * inv, cr.z * inv};
  const F3 t = ld3(tfn, i);
  const F3 s = F3{k * sgn(n.x - t.x), k * sgn(n.y - t.y), k * sgn(n.z - t.z)};   // dL/dn
  const float ns = dot(n, s);
  const F3 gc = F3{(s.x - n.x * ns) * inv, (s.y - n.y * ns) * inv, (s.z - n.z * ns) * inv};   // dL/d(cross)
  const F3 g1 = cross(e2, gc);   // dL/de1
  const F3 g2 = cross(gc, e1);   // dL/de2
  atomicAdd(&grad[3 * ib], g1.x); atomicAdd(&grad[3 * ib + 1], g1.y); atomicAdd(&grad[3 * ib + 2], g1.z);
  atomicAdd(&grad[3 * ic], g2.x); atomicAdd(&grad[3 * ic + 1], g2.y); atomicAdd(&grad[3 * ic + 2], g2.z);
  atomicAdd(&grad[3 * ia], -(g1.x + g2.x)); atomicAdd(&grad[3 * ia + 1], -(g1.y + g2.y));
  atomicAdd(&grad[3 * ia + 2], -(g1.z + g2.z));
}

// Deterministic variant of the face term: the three corner gradients of every face go to corner[3f + i] (zeros
// for a masked face); a CSR gather over the vertex -> corner incidence (sg_unpool_bwd of the incidence handle)
// then sums them per vertex in a fixed order, and mesh_loss_bwd_vertex_add puts the vertex term on top.
__global__ __launch_bounds__(kBlock) void mesh_loss_bwd_corners(const float* __restrict__ pos, const int64_t* __restrict__ faces,
                                                                const float* __restrict__ tfn, const float* __restrict__ fkeep,
                                                                const float* __restrict__ g, int64_t F,
                                                                float* __restrict__ corner) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= F) return;
  F3 ga = F3{0.f, 0.f, 0.f}, g1 = ga, g2 = ga;
  const float k = g[1] * fkeep[i];
  if (k != 0.f) {
    const F3 a = ld3(pos, faces[3 * i]), b = ld3(pos, faces[3 * i + 1]), c = ld3(pos, faces[3 * i + 2]);
    const F3 e1 = sub(b, a), e2 = sub(c, a);
    const F3 cr = cross(e1, e2);
    const float inv = 1.0f / sqrtf(dot(cr, cr));
    const F3 n = F3{cr.x * inv, cr.y * inv, cr.z * inv};
    const F3 t = ld3(tfn, i);
    const F3 s = F3{k * sgn(n.x - t.x), k * sgn(n.y - t.y), k * sgn(n.z - t.z)};
    const float ns = dot(n, s);
    const F3 gc = F3{(s.x - n.x * ns) * inv, (s.y - n.y * ns) * inv, (s.z - n.z * ns) * inv};
    g1 = cross(e2, gc);
    g2 = cross(gc, e1);
    ga = F3{-(g1.x + g2.x), -(g1.y + g2.y), -(g1.z + g2.z)};
  }
  float* o = corner + 9 * i;
  o[0] = ga.x; o[1] = ga.y; o[2] = ga.z;
  o[3] = g1.x; o[4] = g1.y; o[5] = g1.z;
  o[6] = g2.x; o[7] = g2.y; o[8] = g2.z;
}

__global__ __launch_bounds__(kBlock) void mesh_loss_bwd_vertex_add(const float* __restrict__ pos, const float* __restrict__ tpos,
                                                                   const float* __restrict__ vkeep, const float* __restrict__ g,
                                                                   int64_t V, float* __restrict__ grad) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= V) return;
  const float k = 2.0f * g[0] * vkeep[i];
  if (k == 0.f) return;
  const F3 d = sub(ld3(pos, i), ld3(tpos, i));
  grad[3 * i] += k * d.x; grad[3 * i + 1] += k * d.y; grad[3 * i + 2] += k * d.z;
}

// loss = w_pos sqrt(S_p / n_v + 1e-6) + k1 S_n / n_f from the block partials (sgcn.py:130-138 / mgcn.py:138-143 on the
// sums of mesh_loss_fwd), with the two derivatives the backward kernels are scaled by: out = (loss, dloss/dS_p, dloss/dS_n).
// One workgroup; the partials are added in double in a fixed order.
__global__ __launch_bounds__(256) void mesh_loss_finalize(const float* __restrict__ partial, int64_t nb, float n_v, float n_f,
                                                          float w_pos, float k1, float* __restrict__ out) {
  __shared__ double s_w[2][4];
  double a0 = 0.0, a1 = 0.0;
  for (int64_t b = threadIdx.x; b < nb; b += 256) {
    a0 += partial[b * 2];
    a1 += partial[b * 2 + 1];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a0 += __shfl_down(a0, off, 64);
    a1 += __shfl_down(a1, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_w[0][threadIdx.x >> 6] = a0;
    s_w[1][threadIdx.x >> 6] = a1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float sp = (float)((s_w[0][0] + s_w[0][1]) + (s_w[0][2] + s_w[0][3]));
    const float sn = (float)((s_w[1][0] + s_w[1][1]) + (s_w[1][2] + s_w[1][3]));
    const float root = sqrtf(__fadd_rn(__fdiv_rn(sp, n_v), 1.0e-6f));
    const float normal = n_f > 0.f ? __fmul_rn(k1, __fdiv_rn(sn, n_f)) : 0.f;
    out[0] = __fadd_rn(__fmul_rn(w_pos, root), normal);
    out[1] = w_pos * 0.5f / (root * n_v);
    out[2] = n_f > 0.f ? k1 / n_f : 0.f;
  }
}

}  // namespace

int launch_mesh_loss_finalize(const float* partial, int64_t nb, float n_v, float n_f, float w_pos, float k1, float* out,
                              hipStream_t stream) {
  mesh_loss_finalize<<<1, 256, 0, stream>>>(partial, nb, n_v, n_f, w_pos, k1, out);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

int launch_mesh_loss_bwd_corners(const float* pos, const int64_t* faces, const float* tfn, const float* fkeep, const float* g,
                                 int64_t F, float* corner, hipStream_t stream) {
  if (F > 0) {
    mesh_loss_bwd_corners<<<(int)((F + kBlock - 1) / kBlock), kBlock, 0, stream>>>(pos, faces, tfn, fkeep, g, F, corner);
    SG_HIP_TRY(hipGetLastError());
  }
  return SG_OK;
}

int launch_mesh_loss_bwd_vertex_add(const float* pos, const float* tpos, const float* vkeep, const float* g, int64_t V,
                                    float* grad, hipStream_t stream) {
  if (V > 0) {
    mesh_loss_bwd_vertex_add<<<(int)((V + kBlock - 1) / kBlock), kBlock, 0, stream>>>(pos, tpos, vkeep, g, V, grad);
    SG_HIP_TRY(hipGetLastError());
  }
  return SG_OK;
}

int64_t mesh_loss_blocks(int64_t V, int64_t F) {
  const int64_t n = V > F ? V : F;
  return n > 0 ? (n + kBlock - 1) / kBlock : 1;
}

int launch_mesh_loss_fwd(const float* pos, const int64_t* faces, const float* tpos, const float* vkeep, const float* tfn,
                         const float* fkeep, int64_t V, int64_t F, float* partial, hipStream_t stream) {
  mesh_loss_fwd<<<(int)mesh_loss_blocks(V, F), kBlock, 0, stream>>>(pos, faces, tpos, vkeep, tfn, fkeep, V, F, partial);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

int launch_mesh_loss_bwd(const float* pos, const int64_t* faces, const float* tpos, const float* vkeep, const float* tfn,
                         const float* fkeep, const float* g, int64_t V, int64_t V_ext, int64_t F, float* grad,
                         hipStream_t stream) {
  if (V_ext > 0) {
    mesh_loss_bwd_vertex<<<(int)((V_ext + kBlock - 1) / kBlock), kBlock, 0, stream>>>(pos, tpos, vkeep, g, V, V_ext, grad);
    SG_HIP_TRY(hipGetLastError());
  }
  if (F > 0) {
    mesh_loss_bwd_face<<<(int)((F + kBlock - 1) / kBlock), kBlock, 0, stream>>>(pos, faces, tfn, fkeep, g, F, grad);
    SG_HIP_TRY(hipGetLastError());
  }
  return SG_OK;
}

}  // namespace sg
