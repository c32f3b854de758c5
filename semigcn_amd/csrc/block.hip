// sg_block_forward / sg_block_backward: one [ChebConv -> (MeshPool | MeshUnpool)? -> BatchNorm1d -> LeakyReLU] block of
// the reference's networks per foreign call (semigcn.h; util/networks.py:40-46,83-101; util/meshnet.py:39-62,105-128,
// 223-245,295-312).  Nothing here computes: the functions below launch, in the order the per-operator host code used to,
// the kernels of spmm.hip (aggregation), gemm_mfma*.hip / thin_gemm.hip / the BLAS library (products, dense.hip) and
// bn_act.hip (BatchNorm + activation), on buffers the caller owns.  What moves below the ABI is the HOST work between
// those launches -- ~20 ctypes calls, three autograd nodes and a dozen small allocations per block and direction -- which is
// what bounds an iteration on the reference's own mesh sizes (5 K - 50 K vertices).
#include <math.h>

#include "sg_common.h"

namespace sg {
namespace {

bool g_block_planes = true;      // SG_TUNE_BLOCK_PLANES (A/B switch; sg_block_planar then answers 0)

inline int64_t esize(int dtype) { return dtype == SG_F32 ? 4 : 2; }
inline int64_t align_up(int64_t v) { return (v + 255) & ~(int64_t)255; }

// bump allocator over the caller's scratch; with base == nullptr it only counts
struct Carver {
  char* base;
  int64_t cap, at = 0;
  bool overflow = false;
  Carver(void* b, int64_t c) : base((char*)b), cap(c) {}
  void* take(int64_t bytes) {
    const int64_t off = at;
    at += align_up(bytes > 0 ? bytes : 1);
    if (!base) return nullptr;
    if (at > cap) {
      overflow = true;
      return nullptr;
    }
    return base + off;
  }
};

// ---- weights: K fp32 [Cout, Cin] matrices -> the concatenated / stacked copies the products read ---------------------------
struct PackArgs {
  const float* W[3];
  const float* bias;
  void* wpack;
  void* wpack_t;
  float* wpack32;
  float* wpack32_t;
  float* bias_k;
  int K, Cin, Cout, order, dtype;
};

__device__ __forceinline__ void put(void* p, int dtype, int64_t i, float v) {
  if (dtype == SG_F32) ((float*)p)[i] = v;
  else ((uint16_t*)p)[i] = __builtin_bit_cast(uint16_t, (__bf16)v);
}

__global__ __launch_bounds__(256) void pack_weights(const PackArgs a) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n_el = (int64_t)a.K * a.Cin * a.Cout;
  if (e < n_el) {
    int64_t r, c, rows, cols;     // position in wpack [rows, cols]
    float v;
    if (a.order == 0) {           // Wcat [Cout, K*Cin]: block k of row n = W_k[n, :]
      rows = a.Cout; cols = (int64_t)a.K * a.Cin;
      r = e / cols; c = e % cols;
      v = a.W[c / a.Cin][r * a.Cin + c % a.Cin];
    } else {                      // Wstack [K*Cout, Cin]: rows k*Cout .. of it = W_k
      rows = (int64_t)a.K * a.Cout; cols = a.Cin;
      r = e / cols; c = e % cols;
      v = a.W[r / a.Cout][(r % a.Cout) * a.Cin + c];
    }
    put(a.wpack, a.dtype, e, v);
    if (a.wpack_t) put(a.wpack_t, a.dtype, c * rows + r, v);
    // (the float32 copies for the thin kernels hold the weights AS STORED for this feature dtype: rounded to bf16 for bf16
    // features, like the MFMA kernels' operands -- one rounding rule for a layer whatever kernel serves it)
    const float v32 = a.dtype == SG_BF16 ? (float)(__bf16)v : v;
    if (a.wpack32) a.wpack32[e] = v32;
    if (a.wpack32_t) a.wpack32_t[c * rows + r] = v32;
  }
  if (a.bias_k && e < (int64_t)a.K * a.Cout) a.bias_k[e] = (a.bias && e < a.Cout) ? a.bias[e] : 0.f;
}

// eval-mode BatchNorm: (mean, invstd, scale, shift) from the running statistics, with nn.BatchNorm1d's own operation order
__global__ void bn_eval_coeffs(const float* __restrict__ rm, const float* __restrict__ rv, const float* __restrict__ gamma,
                               const float* __restrict__ beta, float eps, int C, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mean = rm[c];
  const float invstd = rsqrtf(__fadd_rn(rv[c], eps));
  const float scale = __fmul_rn(gamma[c], invstd);
  out[c] = mean;
  out[C + c] = invstd;
  out[2 * C + c] = scale;
  out[3 * C + c] = __fsub_rn(beta[c], __fmul_rn(mean, scale));
}

int copy_rows(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, int dtype, hipStream_t s) {
  const int64_t e = esize(dtype);
  SG_HIP_TRY(hipMemcpy2DAsync(dst, (size_t)(ldd * e), src, (size_t)(lds * e), (size_t)(cols * e), (size_t)rows,
                              hipMemcpyDeviceToDevice, s));
  return SG_OK;
}

struct Shape {
  int64_t V, Vo, Ci, Co, KCi, KCo, e;
  int K;
  bool thin, blas_fwd, blas_dx, blas_dw;
  // narrow rows as planes (sg_common.h Planes): the K column blocks of the recurrence buffers -- T and dT of a layer that
  // aggregates first (Cin wide), Z and G of one that aggregates after the product (Cout wide) -- as K dense [V, C] tensors
  bool planes_ok;        // the shape qualifies (order 0: the caller chooses it by passing ldt == Cin)
  Planes pl;             // of 2^log2 = C columns, V * C elements apart
  int64_t nb;            // sg_col_blocks(V_out)
  int64_t mom_floats;    // the larger of the tile-moment / block-moment buffers
  int64_t tn_floats;
};

int shape_of(const sg_block& b, Shape* s) {
  SG_REQUIRE(b.graph != nullptr, "sg_block: null graph");
  SG_REQUIRE(b.dtype == SG_F32 || b.dtype == SG_BF16, "sg_block: unknown dtype %d", b.dtype);
  SG_REQUIRE(b.K >= 1 && b.K <= 3, "sg_block: K = %d (1..3 are implemented)", b.K);
  SG_REQUIRE(b.order == 0 || (b.order == 1 && b.K >= 2), "sg_block: order %d with K = %d", b.order, b.K);
  SG_REQUIRE(b.V > 0 && b.V_out > 0 && b.Cin > 0 && b.Cout > 0, "sg_block: empty shape");
  SG_REQUIRE(b.graph->square && b.graph->fwd.n_rows == b.V, "sg_block: the graph has %lld rows, V = %lld (square graphs only)",
             (long long)b.graph->fwd.n_rows, (long long)b.V);
  if (b.pool_mode == 0) {
    SG_REQUIRE(b.pool == nullptr && b.V_out == b.V, "sg_block: V_out != V without a pool");
  } else if (b.pool_mode == 1) {
    SG_REQUIRE(b.pool && b.pool->by_coarse.n_cols == b.V && b.pool->by_coarse.n_rows == b.V_out,
               "sg_block: MeshPool handle does not map %lld fine rows onto %lld coarse rows", (long long)b.V, (long long)b.V_out);
  } else if (b.pool_mode == 2) {
    SG_REQUIRE(b.pool && b.pool->by_fine.n_cols == b.V && b.pool->by_fine.n_rows == b.V_out,
               "sg_block: MeshUnpool handle does not map %lld coarse rows onto %lld fine rows", (long long)b.V, (long long)b.V_out);
  } else {
    set_error("sg_block: pool_mode %d", b.pool_mode);
    return SG_ERR_INVALID;
  }
  if (b.training) SG_REQUIRE(b.V_out > 1, "sg_block: BatchNorm in training mode needs more than one row");
  s->V = b.V; s->Vo = b.V_out; s->Ci = b.Cin; s->Co = b.Cout; s->K = b.K;
  s->KCi = b.K * b.Cin; s->KCo = b.K * b.Cout; s->e = esize(b.dtype);
  // the fused column sums of dH (= the conv's bias gradient) come from the row-owning apply kernel
  if (col_apply_blocks(b.V_out, b.Cout, b.dtype) == 0) {
    set_error("sg_block: Cout = %lld is not served (whole 16-byte vectors, at most 256 of them per row)", (long long)b.Cout);
    return SG_ERR_UNSUPPORTED;
  }
  if (b.order == 0) {
    s->thin = thin_shape(s->Co, s->KCi);
    s->blas_fwd = !s->thin && !dense_nt_own(b.dtype, s->V, s->Co, s->KCi, s->KCi, s->KCi, s->Co);
    s->blas_dx = !s->thin && !dense_nt_own(b.dtype, s->V, s->KCi, s->Co, s->Co, s->Co, s->KCi);
    s->blas_dw = !s->thin && !dense_tn_own(b.dtype, s->V, s->Co, s->KCi, s->Co, s->KCi);
    s->tn_floats = dense_tn_workspace(b.dtype, s->V, s->Co, s->KCi);
  } else {
    s->thin = false;
    s->blas_fwd = !dense_nt_own(b.dtype, s->V, s->KCo, s->Ci, s->Ci, s->Ci, s->KCo);
    s->blas_dx = !dense_nt_own(b.dtype, s->V, s->Ci, s->KCo, s->KCo, s->KCo, s->Ci);
    s->blas_dw = !dense_tn_own(b.dtype, s->V, s->KCo, s->Ci, s->KCo, s->Ci) || thin_shape(s->KCo, s->Ci);
    s->tn_floats = thin_shape(s->KCo, s->Ci) ? 0 : dense_tn_workspace(b.dtype, s->V, s->KCo, s->Ci);
    if (thin_shape(s->KCo, s->Ci)) {      // (never a layer of the reference: order 1 has K*Cout >= 2 and Cin > Cout)
      set_error("sg_block: order 1 with a %lld x %lld weight stack is not served", (long long)s->KCo, (long long)s->Ci);
      return SG_ERR_UNSUPPORTED;
    }
  }
  {
    const int64_t C = b.order == 0 ? s->Ci : s->Co;
    int lg = 0;
    while ((1 << lg) < C) ++lg;
    bool ok = b.dtype == SG_BF16 && b.K > 1 && (1 << lg) == C && C >= 8 && C <= 64 && g_block_planes;
    if (ok && b.order == 0)
      ok = dense_planes_ok_nt(b.dtype, s->V, s->Co, s->KCi) && dense_planes_ok_nt(b.dtype, s->V, s->KCi, s->Co) &&
           dense_planes_ok_tn(b.dtype, s->V, s->Co, s->KCi);
    else if (ok)
      ok = dense_planes_ok_nt(b.dtype, s->V, s->KCo, s->Ci) && dense_planes_ok_nt(b.dtype, s->V, s->Ci, s->KCo) &&
           dense_planes_ok_tn(b.dtype, s->V, s->KCo, s->Ci);
    s->planes_ok = ok;
    s->pl = Planes{};
    if (ok) {
      s->pl.log2 = lg;
      s->pl.stride = s->V * C;
    }
  }
  s->nb = col_blocks(s->Vo);
  const int64_t R = gemm_tile_rows(s->Co);
  const int64_t tiles = (s->V + R - 1) / R;
  s->mom_floats = (tiles > s->nb ? tiles : s->nb) * 2 * s->Co;
  return SG_OK;
}

// scratch layout of the two passes (one routine walks it for sizing and for the pointers)
struct FwdWs {
  float* moments;
  void* Z;        // order 1: [V, K*Cout]
  void* Hc;       // with a pool: [V, Cout]
  void* blas;
};
void carve_fwd(const sg_block& b, const Shape& s, Carver& c, FwdWs* w) {
  w->moments = (float*)c.take(s.mom_floats * 4);
  w->Z = b.order == 1 ? c.take(s.V * s.KCo * s.e) : nullptr;
  w->Hc = b.pool_mode ? c.take(s.V * s.Co * s.e) : nullptr;
  w->blas = s.blas_fwd ? c.take((int64_t)kBlasWorkspace) : nullptr;
}

struct BwdWs {
  float* part;      // [nb, 2, Cout]
  float* colsum;    // [apply blocks, Cout]
  void* dHp;        // [V_out, Cout]: gradient of the BatchNorm input (order 0, or with a pool)
  void* dHc;        // order 0 with a pool: [V, Cout]
  void* G;          // order 0: dT [V, K*Cin]; order 1: [V, K*Cout]
  float* tn;
  void* blas;
};
void carve_bwd(const sg_block& b, const Shape& s, Carver& c, BwdWs* w) {
  w->part = (float*)c.take(s.nb * 2 * s.Co * 4);
  w->colsum = (float*)c.take(col_apply_blocks(s.Vo, s.Co, b.dtype) * s.Co * 4);
  const bool sep = b.order == 0 || b.pool_mode != 0;
  w->dHp = sep ? c.take(s.Vo * s.Co * s.e) : nullptr;
  w->dHc = (b.order == 0 && b.pool_mode) ? c.take(s.V * s.Co * s.e) : nullptr;
  w->G = c.take(s.V * (b.order == 0 ? s.KCi : s.KCo) * s.e);     // (sized without looking at need_dx: one size per shape)
  w->tn = (float*)c.take(s.tn_floats * 4);
  w->blas = (s.blas_dx || s.blas_dw) ? c.take((int64_t)kBlasWorkspace) : nullptr;
}

int spmm(const sg_block& b, bool transpose, const void* X, int64_t ldx, const void* X0, int64_t ldx0, const void* X1,
         int64_t ldx1, void* Y, int64_t ldy, int64_t C, float alpha, float beta, float gamma, hipStream_t s) {
  return sg_spmm(b.graph, transpose ? 1 : 0, X, ldx, X0, ldx0, X1, ldx1, Y, ldy, C, b.dtype, alpha, beta, gamma, (void*)s);
}

inline char* col(void* base, int64_t cols, int64_t e) { return (char*)base + cols * e; }

// shapes launch_pack_split builds an image for (what gemm_nt_f32s_supported asks of N and K)
inline bool split_image_shape(int64_t N, int64_t K) { return N >= 64 && N % 4 == 0 && K >= 64 && K % 32 == 0; }
inline const void* split_image(const sg_block& b, bool transposed, int64_t N, int64_t K) {
  return (b.dtype == SG_F32 && split_image_shape(N, K)) ? (transposed ? b.wsplit_t : b.wsplit) : nullptr;
}

int pack(const sg_block& b, hipStream_t stream) {
  SG_REQUIRE(b.wpack != nullptr, "sg_block: wpack is null");
  PackArgs a;
  for (int k = 0; k < 3; ++k) a.W[k] = k < b.K ? b.W[k] : nullptr;
  for (int k = 0; k < b.K; ++k) SG_REQUIRE(b.W[k] != nullptr, "sg_block: W[%d] is null", k);
  a.bias = b.bias;
  a.wpack = b.wpack; a.wpack_t = b.wpack_t; a.wpack32 = b.wpack32; a.wpack32_t = b.wpack32_t;
  a.bias_k = (b.order == 1 && b.bias) ? b.bias_k : nullptr;
  a.K = b.K; a.Cin = (int)b.Cin; a.Cout = (int)b.Cout; a.order = b.order; a.dtype = b.dtype;
  const int64_t n = (int64_t)b.K * b.Cin * b.Cout;
  pack_weights<<<(int)((n + 255) / 256), 256, 0, stream>>>(a);
  SG_HIP_TRY(hipGetLastError());
  // float32 features: the split-bf16 images of the weight matrix, once per weight update instead of once per product
  // (sg_block::wsplit / wsplit_t; a shape the split kernels do not take gets no image and its products pack nothing either)
  if (b.dtype == SG_F32 && (b.wsplit || b.wsplit_t)) {
    const int64_t na = b.order == 0 ? b.Cout : (int64_t)b.K * b.Cout, ka = b.order == 0 ? (int64_t)b.K * b.Cin : b.Cin;
    const float* const w32 = (const float*)b.wpack;                   // [na, ka] row-major in the feature dtype = float32
    if (b.wsplit && split_image_shape(na, ka)) {
      int rc = launch_pack_split(w32, ka, 1, b.V, na, ka, b.wsplit, stream);
      if (rc != SG_OK) return rc;
    }
    if (b.wsplit_t && split_image_shape(ka, na)) {                      // the input-gradient product reads the same storage as [ka, na]^T
      int rc = launch_pack_split(w32, 1, ka, b.V, ka, na, b.wsplit_t, stream);
      if (rc != SG_OK) return rc;
    }
  }
  return SG_OK;
}

// The layout of the caller's T of a layer that aggregates first: ldt >= K*Cin = K column blocks of one [V, K*Cin] buffer;
// ldt == Cin (K > 1) = K planes [K][V][Cin], for the shapes sg_block_planar() answers 1 for.
int t_layout(const sg_block& b, const Shape& s, const char* who, Planes* pt) {
  *pt = Planes{};
  SG_REQUIRE(b.T != nullptr, "%s: order 0 needs the [V, K*Cin] buffer T", who);
  if (b.K > 1 && b.ldt == s.Ci) {
    SG_REQUIRE(s.planes_ok, "%s: T given as K planes (ldt == Cin) for a shape that does not take them (sg_block_planar)", who);
    *pt = s.pl;
    return SG_OK;
  }
  SG_REQUIRE(b.ldt >= s.KCi, "%s: row stride of T shorter than K*Cin", who);
  return SG_OK;
}

int forward(const sg_block& b, hipStream_t stream) {
  Shape s;
  int rc = shape_of(b, &s);
  if (rc != SG_OK) return rc;
  SG_REQUIRE(b.X && b.H && b.Y && b.stats && b.gamma && b.beta && b.wpack, "sg_block_forward: null pointer");
  SG_REQUIRE(b.ldy >= s.Co && b.ldx >= s.Ci, "sg_block_forward: row stride shorter than the row");
  SG_REQUIRE((b.running_mean == nullptr) == (b.running_var == nullptr), "sg_block_forward: give both running buffers or none");
  SG_REQUIRE(b.training || b.running_mean, "sg_block_forward: eval mode needs the running statistics");
  if (s.thin) SG_REQUIRE(b.wpack32 && b.wpack32_t, "sg_block_forward: a tiny weight matrix needs wpack32 / wpack32_t");
  if (b.order == 1 && b.bias) SG_REQUIRE(b.bias_k != nullptr, "sg_block_forward: order 1 with a bias needs bias_k");
  Carver c(b.ws, b.ws_bytes);
  FwdWs w;
  carve_fwd(b, s, c, &w);
  SG_REQUIRE(b.ws != nullptr && !c.overflow, "sg_block_forward: scratch too small (%lld bytes given, %lld needed)",
             (long long)b.ws_bytes, (long long)c.at);
  if (b.refresh_weights && (rc = pack(b, stream)) != SG_OK) return rc;

  void* const Hc = b.pool_mode ? w.Hc : b.H;
  bool tile_moments = false;
  if (b.order == 0) {
    Planes pt;
    if ((rc = t_layout(b, s, "sg_block_forward", &pt)) != SG_OK) return rc;
    auto t = [&](int k) { return pt.on() ? col(b.T, k * pt.stride, s.e) : col(b.T, k * s.Ci, s.e); };
    if (b.X != b.T && (rc = copy_rows(b.X, b.ldx, b.T, b.ldt, s.V, s.Ci, b.dtype, stream)) != SG_OK) return rc;
    for (int k = 1; k < b.K; ++k) {     // Tx1 = L^ Tx0;  Txk = 2 L^ Tx(k-1) - Tx(k-2)
      rc = spmm(b, false, t(k - 1), b.ldt, k >= 2 ? t(k - 2) : nullptr, b.ldt, nullptr, 0, t(k), b.ldt, s.Ci, k == 1 ? 1.f : 2.f,
                -1.f, 0.f, stream);
      if (rc != SG_OK) return rc;
    }
    const bool want = b.training && b.pool_mode == 0;      // (pooled rows have other statistics than the tiles of Hc)
    rc = dense_nt(b.T, b.ldt, b.wpack, b.wpack32, s.KCi, b.bias, Hc, s.Co, s.V, s.Co, s.KCi, b.dtype, want ? w.moments : nullptr,
                  &tile_moments, w.blas, kBlasWorkspace, stream, pt, Planes{}, split_image(b, false, s.Co, s.KCi), b.V);
    if (rc != SG_OK) return rc;
  } else {
    // Z = X Wstack^T (+ bias on Z_0); Clenshaw in place: b_k = Z_k + 2 L^ b_(k+1) - b_(k+2); out = Z_0 + L^ b_1 - b_2
    const Planes pz = s.planes_ok ? s.pl : Planes{};
    const int64_t ldz = pz.on() ? s.Co : s.KCo;
    rc = dense_nt(b.X, b.ldx, b.wpack, nullptr, s.Ci, b.bias ? b.bias_k : nullptr, w.Z, ldz, s.V, s.KCo, s.Ci, b.dtype, nullptr,
                  nullptr, w.blas, kBlasWorkspace, stream, Planes{}, pz, split_image(b, false, s.KCo, s.Ci), b.V);
    if (rc != SG_OK) return rc;
    auto z = [&](int k) { return pz.on() ? col(w.Z, k * pz.stride, s.e) : col(w.Z, k * s.Co, s.e); };
    for (int k = b.K - 2; k >= 1; --k) {
      rc = spmm(b, false, z(k + 1), ldz, z(k), ldz, k + 2 <= b.K - 1 ? z(k + 2) : nullptr, ldz, z(k), ldz, s.Co, 2.f, 1.f, -1.f,
                stream);
      if (rc != SG_OK) return rc;
    }
    rc = spmm(b, false, z(1), ldz, z(0), ldz, b.K >= 3 ? z(2) : nullptr, ldz, Hc, s.Co, s.Co, 1.f, 1.f, -1.f, stream);
    if (rc != SG_OK) return rc;
  }
  if (b.pool_mode == 1) rc = sg_pool_mean(b.pool, Hc, s.Co, b.H, s.Co, s.Co, b.dtype, (void*)stream);
  else if (b.pool_mode == 2) rc = sg_unpool(b.pool, Hc, s.Co, b.H, s.Co, s.Co, b.dtype, (void*)stream);
  if (rc != SG_OK) return rc;

  if (b.training) {
    if (tile_moments) {
      const int64_t R = gemm_tile_rows(s.Co);
      rc = launch_bn_stats_finalize_tiles(w.moments, (s.V + R - 1) / R, R, s.Vo, s.Co, b.gamma, b.beta, b.running_mean,
                                          b.running_var, b.momentum, b.eps, b.stats, b.batches_tracked, stream);
    } else {
      rc = launch_col_reduce(0, b.H, s.Co, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0.f, w.moments, s.nb, s.Vo, s.Co,
                             b.dtype, stream);
      if (rc != SG_OK) return rc;
      rc = launch_bn_stats_finalize(w.moments, s.nb, s.Vo, s.Co, b.gamma, b.beta, b.running_mean, b.running_var, b.momentum,
                                    b.eps, b.stats, b.batches_tracked, stream);
    }
    if (rc != SG_OK) return rc;
  } else {
    bn_eval_coeffs<<<(int)((s.Co + 127) / 128), 128, 0, stream>>>(b.running_mean, b.running_var, b.gamma, b.beta, b.eps,
                                                                 (int)s.Co, b.stats);
    SG_HIP_TRY(hipGetLastError());
  }
  return launch_col_apply(0, b.H, s.Co, nullptr, 0, b.stats + 2 * s.Co, b.stats + 3 * s.Co, nullptr, nullptr, nullptr, nullptr,
                          nullptr, b.slope, b.Y, b.ldy, s.Vo, s.Co, b.dtype, stream);
}

int backward(const sg_block& b, hipStream_t stream) {
  Shape s;
  int rc = shape_of(b, &s);
  if (rc != SG_OK) return rc;
  SG_REQUIRE(b.dY && b.H && b.stats && b.gamma && b.dW && b.dvec && b.wpack, "sg_block_backward: null pointer");
  SG_REQUIRE(b.lddy >= s.Co, "sg_block_backward: row stride of dY shorter than the row");
  SG_REQUIRE(!b.need_dx || (b.dX && b.lddx >= s.Ci), "sg_block_backward: need_dx without dX");
  SG_REQUIRE((b.acc_gamma == nullptr) == (b.acc_beta == nullptr), "sg_block_backward: give both BatchNorm accumulators or none");
  const bool sink_w = b.acc_W[0] != nullptr;
  for (int k = 0; k < b.K; ++k)
    SG_REQUIRE((b.acc_W[k] != nullptr) == sink_w, "sg_block_backward: give all K weight accumulators or none");
  if (s.thin) SG_REQUIRE(b.wpack32 && b.wpack32_t, "sg_block_backward: a tiny weight matrix needs wpack32 / wpack32_t");
  Carver c(b.ws, b.ws_bytes);
  BwdWs w;
  carve_bwd(b, s, c, &w);
  SG_REQUIRE(b.ws != nullptr && !c.overflow, "sg_block_backward: scratch too small (%lld bytes given, %lld needed)",
             (long long)b.ws_bytes, (long long)c.at);
  const float *mean = b.stats, *invstd = b.stats + s.Co, *scale = b.stats + 2 * s.Co, *shift = b.stats + 3 * s.Co;
  float* const co = b.dvec;                  // [5, Cout]: sum dz, sum dz xhat, c1, c2, k
  float* const db = b.dvec + 5 * s.Co;       // conv bias gradient

  // BatchNorm + activation: sums -> coefficients (eval mode: N = inf makes c1 = c2 = 0, k = gamma invstd = scale) -> dH
  rc = launch_col_reduce(1, b.dY, b.lddy, b.H, s.Co, scale, shift, mean, invstd, b.slope, w.part, s.nb, s.Vo, s.Co, b.dtype, stream);
  if (rc != SG_OK) return rc;
  rc = launch_bn_bwd_coeffs(w.part, s.nb, s.Co, b.training ? (double)s.Vo : (double)INFINITY, b.gamma, invstd, co, b.acc_gamma,
                            b.acc_beta, nullptr, stream);
  if (rc != SG_OK) return rc;
  // where dH goes: order 1 runs its recurrence on the gradient, dH is block 0 of G; a pool sits in between in either order
  const Planes pg = (b.order == 1 && s.planes_ok) ? s.pl : Planes{};       // order 1: G [V, K*Cout] as planes
  const int64_t ldg = pg.on() ? s.Co : s.KCo;
  void* dHp = w.dHp;
  int64_t lddh = s.Co;
  if (b.order == 1 && b.pool_mode == 0) {
    dHp = w.G;
    lddh = ldg;
  }
  rc = launch_col_apply(1, b.dY, b.lddy, b.H, s.Co, scale, shift, mean, invstd, co + 4 * s.Co, co + 2 * s.Co, co + 3 * s.Co, b.slope,
                        dHp, lddh, s.Vo, s.Co, b.dtype, stream, w.colsum);
  if (rc != SG_OK) return rc;
  // d bias = column sums of the conv output's gradient; through a pool they equal the column sums of dH (every cluster's
  // members share its gradient / count, an unpooled row's gradient goes to one parent).  Nothing in this pass reads them, so
  // with gradient accumulators they RIDE in the launch that finishes the weight gradient (GradSink::cs_*: the same sums in
  // the same order, one launch fewer per block); a product whose engine has no accumulating reduce gets the launch of its own
  GradSink sink;                         // the K weight .grad accumulators: += in the kernel that finishes the reduction
  if (sink_w) {
    for (int k = 0; k < b.K; ++k) sink.dst[k] = b.acc_W[k];
    sink.mode = b.order == 0 ? 1 : 2;
    sink.Cin = (int)s.Ci;
    sink.Cout = (int)s.Co;
    sink.cs_partial = w.colsum;
    sink.cs_nb = col_apply_blocks(s.Vo, s.Co, b.dtype);
    sink.cs_C = (int)s.Co;
    sink.cs_out = db;
    sink.cs_acc = b.acc_bias;
  }
  bool sunk = false;
  void* dHc = dHp;          // gradient of the conv output [V, Cout]
  int64_t lddc = lddh;
  if (b.pool_mode) {
    dHc = b.order == 0 ? w.dHc : w.G;
    lddc = b.order == 0 ? s.Co : ldg;
    if (b.pool_mode == 1) rc = sg_pool_mean_bwd(b.pool, dHp, s.Co, dHc, lddc, s.Co, b.dtype, (void*)stream);
    else rc = sg_unpool_bwd(b.pool, dHp, s.Co, dHc, lddc, s.Co, b.dtype, (void*)stream);
    if (rc != SG_OK) return rc;
  }
  const bool tr = !b.graph->symmetric;

  if (b.order == 0) {
    Planes pt;
    if ((rc = t_layout(b, s, "sg_block_backward", &pt)) != SG_OK) return rc;
    rc = dense_tn(dHc, lddc, b.T, b.ldt, s.V, s.Co, s.KCi, b.dtype, w.tn, b.dW, s.KCi, w.blas, kBlasWorkspace, stream,
                  sink_w ? &sink : nullptr, &sunk, Planes{}, pt);
    if (rc != SG_OK) return rc;
    if (b.need_dx) {
      // dT = dH Wcat: block k = dL/dTx_k before the recurrence is unwound (planes when T is)
      void* const dT = b.K == 1 ? b.dX : w.G;
      const int64_t ldt = b.K == 1 ? b.lddx : (pt.on() ? s.Ci : s.KCi);
      rc = dense_nn(dHc, lddc, b.wpack, s.KCi, b.wpack_t, s.Co, b.wpack32_t, dT, ldt, s.V, s.KCi, s.Co, b.dtype, w.blas,
                    kBlasWorkspace, stream, Planes{}, pt, split_image(b, true, s.KCi, s.Co), b.V);
      if (rc != SG_OK) return rc;
      if (b.K > 1) {
        auto g = [&](int k) { return pt.on() ? col(dT, k * pt.stride, s.e) : col(dT, k * s.Ci, s.e); };
        for (int k = b.K - 2; k >= 1; --k) {     // g_k += 2 L^T g_(k+1) - g_(k+2), in place
          rc = spmm(b, tr, g(k + 1), ldt, g(k), ldt, k + 2 <= b.K - 1 ? g(k + 2) : nullptr, ldt, g(k), ldt, s.Ci, 2.f, 1.f, -1.f, stream);
          if (rc != SG_OK) return rc;
        }
        rc = spmm(b, tr, g(1), ldt, g(0), ldt, b.K >= 3 ? g(2) : nullptr, ldt, b.dX, b.lddx, s.Ci, 1.f, 1.f, -1.f, stream);
        if (rc != SG_OK) return rc;
      }
    }
  } else {
    SG_REQUIRE(b.X && b.ldx >= s.Ci, "sg_block_backward: order 1 needs the saved input X");
    auto g = [&](int k) { return pg.on() ? col(w.G, k * pg.stride, s.e) : col(w.G, k * s.Co, s.e); };
    // G = [T_0 | T_1 | ..](L^T) dH: the forward Chebyshev recurrence applied to the gradient
    rc = spmm(b, tr, g(0), ldg, nullptr, 0, nullptr, 0, g(1), ldg, s.Co, 1.f, 0.f, 0.f, stream);
    if (rc != SG_OK) return rc;
    for (int k = 2; k < b.K; ++k) {
      rc = spmm(b, tr, g(k - 1), ldg, g(k - 2), ldg, nullptr, 0, g(k), ldg, s.Co, 2.f, -1.f, 0.f, stream);
      if (rc != SG_OK) return rc;
    }
    if (b.need_dx) {
      rc = dense_nn(w.G, ldg, b.wpack, s.Ci, b.wpack_t, s.KCo, nullptr, b.dX, b.lddx, s.V, s.Ci, s.KCo, b.dtype, w.blas,
                    kBlasWorkspace, stream, pg, Planes{}, split_image(b, true, s.Ci, s.KCo), b.V);
      if (rc != SG_OK) return rc;
    }
    rc = dense_tn(w.G, ldg, b.X, b.ldx, s.V, s.KCo, s.Ci, b.dtype, w.tn, b.dW, s.Ci, w.blas, kBlasWorkspace, stream,
                  sink_w ? &sink : nullptr, &sunk, pg);
    if (rc != SG_OK) return rc;
  }

  if (!(sink_w && sunk)) {    // the bias sums did not ride (no accumulators, or an engine without the accumulating reduce)
    rc = launch_colsum_finalize(w.colsum, col_apply_blocks(s.Vo, s.Co, b.dtype), s.Co, db, stream, b.acc_bias);
    if (rc != SG_OK) return rc;
  }
  if (sink_w && !sunk) {      // (an engine without the accumulating epilogue ran -- a single BLAS product: one add launch)
    const float* srcs[3];
    float* dsts[3];
    int64_t ld[3], rows[3], cols[3];
    for (int k = 0; k < b.K; ++k) {
      srcs[k] = b.order == 0 ? b.dW + k * s.Ci : b.dW + k * s.Co * s.Ci;
      ld[k] = b.order == 0 ? s.KCi : s.Ci;
      dsts[k] = b.acc_W[k]; rows[k] = s.Co; cols[k] = s.Ci;
    }
    rc = launch_multi_add(b.K, srcs, ld, rows, cols, dsts, stream);
    if (rc != SG_OK) return rc;
  }
  return SG_OK;
}

// =====================================================================================================================
// One block of a VERTEX PARTITION, phase by phase (sg_block_run; SURVEY 8(e): the reference is single-device).
//
// The rank owns V rows; every feature buffer has V_ext rows, [owned | per peer: its halo rows, kPadRows pad rows].  The halo
// exchange that follows SG_PHASE_CONV moves rows of H -- the conv output BEFORE BatchNorm -- and, in the pad rows of every
// peer's segment, this rank's BatchNorm statistics of H: the all-gather of the statistics rides in the exchange, and
// SG_PHASE_BN applies BatchNorm + activation on all V_ext rows, halo rows included (they are forward-only copies: the
// backward pass is owner-computes on exchanged GRADIENT rows, the global L^ being symmetric).
constexpr int kPadRows = 5;      // (2 C + 1) floats fit into 5 rows of C bf16 (or fp32) values for every C >= 2

// dst[r] = src[idx[r]] (width elements of esz bytes); idx[r] = -1 - j: pad row j, bytes [j, j + 1) * width * esz of `blob`
// (blob_bytes long, zero beyond; blob == nullptr: zeros)
__global__ __launch_bounds__(256) void pack_rows(const char* __restrict__ src, int64_t ld_bytes, const int32_t* __restrict__ idx,
                                                 int64_t n, int row_bytes, const char* __restrict__ blob, int blob_bytes,
                                                 char* __restrict__ dst) {
  const int per_row = row_bytes / 4;                      // 4-byte words (row_bytes % 4 == 0: C % 2 == 0 for bf16)
  for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < n * per_row; w += (int64_t)gridDim.x * 256) {
    const int64_t r = w / per_row;
    const int c = (int)(w - r * per_row) * 4;
    const int i = idx[r];
    uint32_t v = 0;
    if (i >= 0) {
      v = *(const uint32_t*)(src + (int64_t)i * ld_bytes + c);
    } else if (blob) {
      const int off = (-1 - i) * row_bytes + c;
      if (off + 4 <= blob_bytes) v = *(const uint32_t*)(blob + off);
    }
    *(uint32_t*)(dst + r * row_bytes + c) = v;
  }
}

// gathered[q] = rank q's (mean[C], M2[C], rows): this rank's from `local`, a peer's from the pad rows of its segment in H
// The pad rows are then cleared (an eval-mode block and a gradient exchange send zeros there already: pack_rows): they held
// statistics bytes that would read as arbitrary bf16 / fp32 FEATURES, NaN and Inf included, and SG_PHASE_BN applies
// BatchNorm + activation to all V_ext rows -- no pad row is ever an aggregation source and every reduction stops at the V
// owned rows, but a row of finite values costs nothing and cannot poison a later V_ext-wide pass.
__global__ void gather_stats(char* __restrict__ H, int64_t ld_bytes, const int64_t* __restrict__ stats_rows,
                             const float* __restrict__ local, int world, int n, float* __restrict__ gathered) {
  const int q = blockIdx.x;
  const int64_t row = stats_rows[q];
  const float* src = row < 0 ? local : (const float*)(H + row * ld_bytes);
  for (int i = threadIdx.x; i < n; i += blockDim.x) gathered[(int64_t)q * n + i] = src[i];
  if (row < 0) return;
  __syncthreads();
  uint32_t* pad = (uint32_t*)(H + row * ld_bytes);
  for (int64_t i = threadIdx.x; i < kPadRows * ld_bytes / 4; i += blockDim.x) pad[i] = 0u;
}

int part_check(const sg_block& b) {
  SG_REQUIRE(b.graph && b.graph_wide, "sg_block_run: a partition block needs graph and graph_wide");
  SG_REQUIRE(b.dtype == SG_F32 || b.dtype == SG_BF16, "sg_block_run: unknown dtype %d", b.dtype);
  SG_REQUIRE(b.K == 3 && (b.order == 0 || b.order == 1), "sg_block_run: K = 3 only (the reference's), order 0 / 1");
  SG_REQUIRE(b.V > 1 && b.V_ext >= b.V && b.Cin > 0 && b.Cout > 0 && b.world >= 1, "sg_block_run: bad shape");
  SG_REQUIRE(b.graph->fwd.n_rows == b.V && b.graph->fwd.n_cols == b.V_ext && b.graph_wide->fwd.n_rows == b.V_ext &&
                 b.graph_wide->fwd.n_cols == b.V_ext,
             "sg_block_run: graph must be V x V_ext and graph_wide V_ext x V_ext");
  SG_REQUIRE(b.pool == nullptr && b.pool_mode == 0, "sg_block_run: no pool inside a partition block");
  SG_REQUIRE(b.ldh == b.Cout, "sg_block_run: H must be a contiguous [V_ext, Cout] buffer (ldh = Cout)");
  SG_REQUIRE((b.Cout * esize(b.dtype)) % 4 == 0 && kPadRows * b.Cout * esize(b.dtype) >= (2 * b.Cout + 1) * 4,
             "sg_block_run: Cout = %lld cannot carry the statistics in its pad rows", (long long)b.Cout);
  if (col_apply_blocks(b.V, b.Cout, b.dtype) == 0) {
    set_error("sg_block_run: Cout = %lld is not served", (long long)b.Cout);
    return SG_ERR_UNSUPPORTED;
  }
  return SG_OK;
}

struct PartWs {
  float* moments;   // tile / block moments of the conv output
  void* Z;          // order 1 forward: [V_ext, K*Cout]
  float* part;      // BatchNorm backward partials [nb, 2, Cout]
  float* co;        // [5, Cout]
  float* colsum;
  void* dH;         // order 0: [V, Cout]
  float* tn;
  void* blas;
};
void carve_part(const sg_block& b, Carver& c, PartWs* w) {
  const int64_t e = esize(b.dtype), KCi = b.K * b.Cin, KCo = b.K * b.Cout;
  const int64_t nb = col_blocks(b.V), R = gemm_tile_rows(b.Cout), tiles = (b.V + R - 1) / R;
  w->moments = (float*)c.take((tiles > nb ? tiles : nb) * 2 * b.Cout * 4);
  w->Z = b.order == 1 ? c.take(b.V_ext * KCo * e) : nullptr;
  w->part = (float*)c.take(nb * 2 * b.Cout * 4);
  w->co = (float*)c.take(5 * b.Cout * 4);
  w->colsum = (float*)c.take(col_apply_blocks(b.V, b.Cout, b.dtype) * b.Cout * 4);
  w->dH = b.order == 0 ? c.take(b.V * b.Cout * e) : nullptr;
  const int64_t tn = b.order == 0 ? dense_tn_workspace(b.dtype, b.V, b.Cout, KCi) : dense_tn_workspace(b.dtype, b.V, KCo, b.Cin);
  w->tn = (float*)c.take(tn * 4);
  w->blas = c.take((int64_t)kBlasWorkspace);
}

int pack_to(const sg_block& b, const void* src, int64_t ld, int64_t width, const float* blob, int64_t blob_floats, hipStream_t stream) {
  if (b.n_send == 0) return SG_OK;
  SG_REQUIRE(b.send && b.send_index, "sg_block_run: no send buffer / index");
  const int64_t e = esize(b.dtype);
  const int64_t words = b.n_send * (width * e / 4);
  int64_t blocks = (words + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  pack_rows<<<(int)blocks, 256, 0, stream>>>((const char*)src, ld * e, b.send_index, b.n_send, (int)(width * e), (const char*)blob,
                                            (int)(blob_floats * 4), (char*)b.send);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

int part_run(const sg_block& b, hipStream_t stream) {
  int rc = part_check(b);
  if (rc != SG_OK) return rc;
  const int64_t e = esize(b.dtype), Ci = b.Cin, Co = b.Cout, KCi = b.K * Ci, KCo = b.K * Co, V = b.V, Ve = b.V_ext;
  SG_REQUIRE(b.H != nullptr, "sg_block_run: H is null");
  Carver c(b.ws, b.ws_bytes);
  PartWs w;
  carve_part(b, c, &w);
  SG_REQUIRE(b.ws != nullptr && !c.overflow, "sg_block_run: scratch too small (%lld bytes given, %lld needed)",
             (long long)b.ws_bytes, (long long)c.at);
  const bool thin = b.order == 0 && thin_shape(Co, KCi);
  if (thin) SG_REQUIRE(b.wpack32 && b.wpack32_t, "sg_block_run: a tiny weight matrix needs wpack32 / wpack32_t");
  const float *mean = b.stats, *invstd = b.stats + Co, *scale = b.stats + 2 * Co, *shift = b.stats + 3 * Co;
  auto agg = [&](const sg_graph* g, const void* X, int64_t ldx, const void* X0, int64_t ld0, const void* X1, int64_t ld1, void* Y,
                 int64_t ldy, int64_t C, float a, float be, float ga) {
    return sg_spmm(g, 0, X, ldx, X0, ld0, X1, ld1, Y, ldy, C, b.dtype, a, be, ga, (void*)stream);
  };

  if (b.phase & SG_PHASE_CONV) {
    SG_REQUIRE(b.local && b.wpack, "sg_block_run (conv): null pointer");
    if (b.refresh_weights && (rc = pack(b, stream)) != SG_OK) return rc;
    bool tile_moments = false;
    if (b.order == 0) {
      SG_REQUIRE(b.T && b.ldt >= KCi, "sg_block_run (conv): order 0 needs the [V_ext, K*Cin] buffer T");
      // Tx1 on the owned and ring-1 rows (one exchange serves both aggregations), Tx2 on the owned rows
      if ((rc = agg(b.graph_wide, b.T, b.ldt, nullptr, 0, nullptr, 0, col(b.T, Ci, e), b.ldt, Ci, 1.f, 0.f, 0.f)) != SG_OK) return rc;
      if ((rc = agg(b.graph, col(b.T, Ci, e), b.ldt, b.T, b.ldt, nullptr, 0, col(b.T, 2 * Ci, e), b.ldt, Ci, 2.f, -1.f, 0.f)) != SG_OK) return rc;
      rc = dense_nt(b.T, b.ldt, b.wpack, b.wpack32, KCi, b.bias, b.H, b.ldh, V, Co, KCi, b.dtype, b.training ? w.moments : nullptr,
                    &tile_moments, w.blas, kBlasWorkspace, stream, Planes{}, Planes{}, split_image(b, false, Co, KCi), b.V);
      if (rc != SG_OK) return rc;
    } else {
      SG_REQUIRE(b.X && b.ldx >= Ci, "sg_block_run (conv): order 1 needs the [V_ext, Cin] input X");
      if (b.bias) SG_REQUIRE(b.bias_k != nullptr, "sg_block_run (conv): order 1 with a bias needs bias_k");
      // the product on ALL V_ext rows (the halo rows' share is a few per cent): no exchange between product and aggregation
      rc = dense_nt(b.X, b.ldx, b.wpack, nullptr, Ci, b.bias ? b.bias_k : nullptr, w.Z, KCo, Ve, KCo, Ci, b.dtype, nullptr, nullptr,
                    w.blas, kBlasWorkspace, stream, Planes{}, Planes{}, split_image(b, false, KCo, Ci), b.V);
      if (rc != SG_OK) return rc;
      char* z0 = (char*)w.Z;
      char* z1 = col(w.Z, Co, e);
      char* z2 = col(w.Z, 2 * Co, e);
      if ((rc = agg(b.graph_wide, z2, KCo, z1, KCo, nullptr, 0, z1, KCo, Co, 2.f, 1.f, 0.f)) != SG_OK) return rc;
      if ((rc = agg(b.graph, z1, KCo, z0, KCo, z2, KCo, b.H, b.ldh, Co, 1.f, 1.f, -1.f)) != SG_OK) return rc;
    }
    if (b.training) {      // this rank's (mean, M2, rows) of its owned rows
      if (tile_moments) {
        const int64_t R = gemm_tile_rows(Co);
        rc = launch_bn_merge_tiles(w.moments, (V + R - 1) / R, R, V, Co, b.local, b.local + 2 * Co, stream);
      } else {
        const int64_t nb = col_blocks(V);
        rc = launch_col_reduce(0, b.H, b.ldh, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0.f, w.moments, nb, V, Co, b.dtype, stream);
        if (rc != SG_OK) return rc;
        rc = launch_bn_merge_tiles(w.moments, nb, (V + nb - 1) / nb, V, Co, b.local, b.local + 2 * Co, stream);
      }
      if (rc != SG_OK) return rc;
    }
    if (b.n_send > 0 && (rc = pack_to(b, b.H, b.ldh, Co, b.training ? b.local : nullptr, 2 * Co + 1, stream)) != SG_OK) return rc;
  }

  if (b.phase & SG_PHASE_BN) {
    SG_REQUIRE(b.stats && b.gamma && b.beta && b.Y && b.ldy >= Co && b.V_out >= V && b.V_out <= Ve, "sg_block_run (bn): bad argument");
    if (b.training) {
      SG_REQUIRE(b.gathered && b.count && (b.gathered_ready || (b.stats_rows && b.local)), "sg_block_run (bn): no statistics");
      if (!b.gathered_ready) {
        gather_stats<<<b.world, 128, 0, stream>>>((char*)b.H, b.ldh * e, b.stats_rows, b.local, b.world, (int)(2 * Co + 1), b.gathered);
        SG_HIP_TRY(hipGetLastError());
      }
      rc = launch_bn_finalize_ranks(b.gathered, b.world, Co, b.gamma, b.beta, b.running_mean, b.running_var, b.momentum, b.eps,
                                    b.stats, b.count, b.batches_tracked, stream);
      if (rc != SG_OK) return rc;
    } else {
      SG_REQUIRE(b.running_mean && b.running_var, "sg_block_run (bn): eval mode needs the running statistics");
      bn_eval_coeffs<<<(int)((Co + 127) / 128), 128, 0, stream>>>(b.running_mean, b.running_var, b.gamma, b.beta, b.eps, (int)Co, b.stats);
      SG_HIP_TRY(hipGetLastError());
    }
    rc = launch_col_apply(0, b.H, b.ldh, nullptr, 0, scale, shift, nullptr, nullptr, nullptr, nullptr, nullptr, b.slope, b.Y, b.ldy,
                          b.V_out, Co, b.dtype, stream);
    if (rc != SG_OK) return rc;
  }

  if (b.phase & (SG_PHASE_BWD_REDUCE | SG_PHASE_BWD_A | SG_PHASE_BWD_B))
    SG_REQUIRE(b.training, "sg_block_run: the backward phases of a partition block are for training mode");

  if (b.phase & SG_PHASE_BWD_REDUCE) {
    SG_REQUIRE(b.dY && b.lddy >= Co && b.dvec && b.stats, "sg_block_run (bwd reduce): null pointer");
    const int64_t nb = col_blocks(V);
    rc = launch_col_reduce(1, b.dY, b.lddy, b.H, b.ldh, scale, shift, mean, invstd, b.slope, w.part, nb, V, Co, b.dtype, stream);
    if (rc != SG_OK) return rc;
    // rows 0, 1 of dvec: this rank's sums (the caller all-reduces them in place); += into the BatchNorm .grad accumulators
    rc = launch_bn_bwd_coeffs(w.part, nb, Co, 1.0, b.gamma, invstd, b.dvec, b.acc_gamma, b.acc_beta, nullptr, stream);
    if (rc != SG_OK) return rc;
  }

  if (b.phase & SG_PHASE_BWD_A) {
    SG_REQUIRE(b.dY && b.dvec && b.dW && b.G && b.count && b.wpack, "sg_block_run (bwd a): null pointer");
    const bool sink_w = b.acc_W[0] != nullptr;
    // c1, c2, k of the whole mesh from the all-reduced sums and the device-resident row count
    rc = launch_bn_bwd_coeffs(b.dvec, 1, Co, 0.0, b.gamma, invstd, w.co, nullptr, nullptr, b.count, stream);
    if (rc != SG_OK) return rc;
    void* dH = b.order == 0 ? w.dH : b.G;
    const int64_t lddh = b.order == 0 ? Co : KCo;
    rc = launch_col_apply(1, b.dY, b.lddy, b.H, b.ldh, scale, shift, mean, invstd, w.co + 4 * Co, w.co + 2 * Co, w.co + 3 * Co, b.slope,
                          dH, lddh, V, Co, b.dtype, stream, w.colsum);
    if (rc != SG_OK) return rc;
    if (b.order != 0) {        // (the weight gradient of an order-1 block is finished in a later phase: the sums get their own launch)
      rc = launch_colsum_finalize(w.colsum, col_apply_blocks(V, Co, b.dtype), Co, b.dvec + 5 * Co, stream, b.acc_bias);
      if (rc != SG_OK) return rc;
    }
    if (b.order == 0) {
      GradSink sink;
      if (sink_w) {
        for (int k = 0; k < b.K; ++k) sink.dst[k] = b.acc_W[k];
        sink.mode = 1; sink.Cin = (int)Ci; sink.Cout = (int)Co;
        sink.cs_partial = w.colsum;                 // the bias sums ride in the weight gradient's reduce (see backward())
        sink.cs_nb = col_apply_blocks(V, Co, b.dtype);
        sink.cs_C = (int)Co;
        sink.cs_out = b.dvec + 5 * Co;
        sink.cs_acc = b.acc_bias;
      }
      bool sunk = false;
      rc = dense_tn(dH, lddh, b.T, b.ldt, V, Co, KCi, b.dtype, w.tn, b.dW, KCi, w.blas, kBlasWorkspace, stream, sink_w ? &sink : nullptr, &sunk);
      if (rc != SG_OK) return rc;
      if (!(sink_w && sunk)) {
        rc = launch_colsum_finalize(w.colsum, col_apply_blocks(V, Co, b.dtype), Co, b.dvec + 5 * Co, stream, b.acc_bias);
        if (rc != SG_OK) return rc;
      }
      if (sink_w && !sunk) {
        const float* srcs[3]; float* dsts[3]; int64_t ld[3], rows[3], cols[3];
        for (int k = 0; k < b.K; ++k) { srcs[k] = b.dW + k * Ci; ld[k] = KCi; dsts[k] = b.acc_W[k]; rows[k] = Co; cols[k] = Ci; }
        if ((rc = launch_multi_add(b.K, srcs, ld, rows, cols, dsts, stream)) != SG_OK) return rc;
      }
      // the K gradient blocks of the owned rows; blocks 1, 2 of the boundary rows go to the peers
      rc = dense_nn(dH, lddh, b.wpack, KCi, b.wpack_t, Co, b.wpack32_t, b.G, KCi, V, KCi, Co, b.dtype, w.blas, kBlasWorkspace, stream,
                    Planes{}, Planes{}, split_image(b, true, KCi, Co), b.V);
      if (rc != SG_OK) return rc;
      if ((rc = pack_to(b, col(b.G, Ci, e), KCi, 2 * Ci, nullptr, 0, stream)) != SG_OK) return rc;
    } else {
      if ((rc = pack_to(b, b.G, KCo, Co, nullptr, 0, stream)) != SG_OK) return rc;     // dH is block 0 of G
    }
  }

  if (b.phase & SG_PHASE_BWD_B) {
    SG_REQUIRE(b.G && (!b.need_dx || (b.dX && b.lddx >= Ci)) && (Ve == V || b.recv), "sg_block_run (bwd b): null pointer");
    const bool sink_w = b.acc_W[0] != nullptr;
    if (b.order == 0) {
      if (Ve > V && (rc = copy_rows(b.recv, 2 * Ci, (char*)b.G + (V * KCi + Ci) * e, KCi, Ve - V, 2 * Ci, b.dtype, stream)) != SG_OK) return rc;
      char* g0 = (char*)b.G;
      char* g1 = col(b.G, Ci, e);
      char* g2 = col(b.G, 2 * Ci, e);
      if ((rc = agg(b.graph_wide, g2, KCi, g1, KCi, nullptr, 0, g1, KCi, Ci, 2.f, 1.f, 0.f)) != SG_OK) return rc;
      if (b.need_dx && (rc = agg(b.graph, g1, KCi, g0, KCi, g2, KCi, b.dX, b.lddx, Ci, 1.f, 1.f, -1.f)) != SG_OK) return rc;
    } else {
      SG_REQUIRE(b.X && b.dW, "sg_block_run (bwd b): order 1 needs the saved input X and dW");
      if (Ve > V && (rc = copy_rows(b.recv, Co, (char*)b.G + V * KCo * e, KCo, Ve - V, Co, b.dtype, stream)) != SG_OK) return rc;
      char* g0 = (char*)b.G;
      char* g1 = col(b.G, Co, e);
      char* g2 = col(b.G, 2 * Co, e);
      if ((rc = agg(b.graph_wide, g0, KCo, nullptr, 0, nullptr, 0, g1, KCo, Co, 1.f, 0.f, 0.f)) != SG_OK) return rc;
      if ((rc = agg(b.graph, g1, KCo, g0, KCo, nullptr, 0, g2, KCo, Co, 2.f, -1.f, 0.f)) != SG_OK) return rc;
      if (b.need_dx) {
        rc = dense_nn(b.G, KCo, b.wpack, Ci, b.wpack_t, KCo, nullptr, b.dX, b.lddx, V, Ci, KCo, b.dtype, w.blas, kBlasWorkspace, stream,
                      Planes{}, Planes{}, split_image(b, true, Ci, KCo), b.V);
        if (rc != SG_OK) return rc;
      }
      GradSink sink;
      if (sink_w) {
        for (int k = 0; k < b.K; ++k) sink.dst[k] = b.acc_W[k];
        sink.mode = 2; sink.Cin = (int)Ci; sink.Cout = (int)Co;
      }
      bool sunk = false;
      rc = dense_tn(b.G, KCo, b.X, b.ldx, V, KCo, Ci, b.dtype, w.tn, b.dW, Ci, w.blas, kBlasWorkspace, stream, sink_w ? &sink : nullptr, &sunk);
      if (rc != SG_OK) return rc;
      if (sink_w && !sunk) {
        const float* srcs[3]; float* dsts[3]; int64_t ld[3], rows[3], cols[3];
        for (int k = 0; k < b.K; ++k) { srcs[k] = b.dW + k * Co * Ci; ld[k] = Ci; dsts[k] = b.acc_W[k]; rows[k] = Co; cols[k] = Ci; }
        if ((rc = launch_multi_add(b.K, srcs, ld, rows, cols, dsts, stream)) != SG_OK) return rc;
      }
    }
  }
  return SG_OK;
}

}  // namespace

int set_block_planes(int value) {
  g_block_planes = value != 0;
  return SG_OK;
}

}  // namespace sg

using namespace sg;

extern "C" {

SG_API int sg_block_run(const sg_block* blks, int64_t n, void* stream) {
  SG_REQUIRE(n >= 0 && (n == 0 || blks != nullptr), "sg_block_run: bad argument");
  for (int64_t i = 0; i < n; ++i) {
    SG_REQUIRE(blks[i].phase != 0, "sg_block_run: block %lld has no phase", (long long)i);
    const int rc = part_run(blks[i], (hipStream_t)stream);
    if (rc != SG_OK) return rc;
  }
  return SG_OK;
}

SG_API int64_t sg_block_sizeof(void) { return (int64_t)sizeof(sg_block); }

SG_API int sg_block_planar(const sg_block* blk) {
  if (!blk) {
    set_error("sg_block_planar: null block");
    return SG_ERR_INVALID;
  }
  Shape s;
  const int rc = shape_of(*blk, &s);
  if (rc != SG_OK) return rc;
  return (blk->order == 0 && s.planes_ok) ? 1 : 0;
}

SG_API int64_t sg_block_workspace(const sg_block* blk, int backward_pass) {
  if (!blk) {
    set_error("sg_block_workspace: null block");
    return SG_ERR_INVALID;
  }
  if (backward_pass == 2) {          // a partition block: one size for all of its phases
    const int rc = part_check(*blk);
    if (rc != SG_OK) return rc;
    Carver c(nullptr, 0);
    PartWs w;
    carve_part(*blk, c, &w);
    return c.at;
  }
  Shape s;
  const int rc = shape_of(*blk, &s);
  if (rc != SG_OK) return rc;
  Carver c(nullptr, 0);
  if (backward_pass) {
    BwdWs w;
    carve_bwd(*blk, s, c, &w);
  } else {
    FwdWs w;
    carve_fwd(*blk, s, c, &w);
  }
  return c.at;
}

SG_API int sg_block_forward(const sg_block* blk, void* stream) {
  SG_REQUIRE(blk != nullptr, "sg_block_forward: null block");
  return forward(*blk, (hipStream_t)stream);
}

SG_API int sg_block_backward(const sg_block* blk, void* stream) {
  SG_REQUIRE(blk != nullptr, "sg_block_backward: null block");
  return backward(*blk, (hipStream_t)stream);
}

SG_API int sg_block_chain_forward(const sg_block* blks, int64_t n, void* stream) {
  SG_REQUIRE(n >= 0 && (n == 0 || blks != nullptr), "sg_block_chain_forward: bad argument");
  for (int64_t i = 0; i < n; ++i) {
    const int rc = forward(blks[i], (hipStream_t)stream);
    if (rc != SG_OK) return rc;
  }
  return SG_OK;
}

SG_API int sg_block_chain_backward(const sg_block* blks, int64_t n, void* stream) {
  SG_REQUIRE(n >= 0 && (n == 0 || blks != nullptr), "sg_block_chain_backward: bad argument");
  for (int64_t i = n - 1; i >= 0; --i) {
    const int rc = backward(blks[i], (hipStream_t)stream);
    if (rc != SG_OK) return rc;
  }
  return SG_OK;
}

}  // extern "C"
