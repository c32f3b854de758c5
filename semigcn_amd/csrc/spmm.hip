// Edge aggregation for gfx950: Y[r] = alpha*sd[r] * sum_{k in row r} ss[idx[k]] * X[idx[k]]
//                                       + beta*X0[r] + gamma*X1[r]
//
// Replaces MessagePassing.propagate -> torch_scatter.scatter(sum) -> ATen scatter_add_ [3P]
// (the two `propagate` calls per ChebConv.forward reached from util/networks.py:42,49 and
// util/meshnet.py:40-58,106-124,224-240), its autograd transposes, and the sparse mm of
// MeshPool / MeshUnpool (util/meshnet.py:14-17,25-27).
//
// Design (HBM-bound; no MFMA on purpose):
//  * CSR rows, one owner per output row -> no atomics, deterministic, Y written once.  The
//    reference materialises an [E+2V, C] message tensor per call; here X is gathered straight
//    into registers and reduced there.
//  * A row of C features is spread over G = C/VEC lanes (16-byte vectors: 4 f32 / 8 bf16 per
//    lane), so a 64-lane wavefront owns 64/G rows at a time and every gather instruction
//    moves whole 16-B..1-KiB row segments.  C > 64*VEC uses R vectors per lane.
//  * Each wavefront owns a chunk of `ch` consecutive rows.  Its row pointers and its whole
//    neighbour list (index + deg^-1/2 of the neighbour) are staged ONCE into LDS with
//    coalesced loads, so the inner loop's only global traffic is the feature gather; the
//    dependent rowptr -> idx -> scale chain is paid once per chunk, not once per row.
//  * Up to 8 gathers per row are issued back to back, branch-free, before the first FMA (up to
//    256 B per lane in flight); the batch size (2/4/6/8) follows the longest row in flight.
//  * Chunks are dealt to workgroups so that the workgroups sharing an XCD (blockIdx % 8)
//    sweep one contiguous range of rows: neighbouring rows' gathers then hit that XCD's L2.
#include "sg_common.h"

namespace sg {
namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kChMax = 64;   // rows per wavefront chunk (upper bound)
constexpr int kCap = 512;    // staged neighbour slots per wavefront

struct bf16_tag {};
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

enum : int { kFlagXcdMap = 1, kFlagNoTiles = 2 };

template <typename T> struct Vt;
template <> struct Vt<float> {
  static constexpr int VEC = 4;
  using raw = f32x4;
  using elem = float;
  static __device__ __forceinline__ void unpack(const raw& v, float* f) {
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
  static __device__ __forceinline__ raw pack(const float* f) { return raw{f[0], f[1], f[2], f[3]}; }
  static __device__ __forceinline__ float load1(const void* p, int64_t i) { return ((const float*)p)[i]; }
  static __device__ __forceinline__ void store1(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
};
template <> struct Vt<bf16_tag> {
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
    // plain casts: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
    uint16_t a = __builtin_bit_cast(uint16_t, (__bf16)lo);
    uint16_t b = __builtin_bit_cast(uint16_t, (__bf16)hi);
    return (uint32_t)a | ((uint32_t)b << 16);
  }
  static __device__ __forceinline__ raw pack(const float* f) {
    return raw{cvt2(f[0], f[1]), cvt2(f[2], f[3]), cvt2(f[4], f[5]), cvt2(f[6], f[7])};
  }
  static __device__ __forceinline__ float load1(const void* p, int64_t i) {
    return __uint_as_float((uint32_t)((const uint16_t*)p)[i] << 16);
  }
  static __device__ __forceinline__ void store1(void* p, int64_t i, float v) {
    ((uint16_t*)p)[i] = __builtin_bit_cast(uint16_t, (__bf16)v);
  }
};

// Workgroups b and b+8 share an XCD (round-robin dispatch; speed only, never correctness).
// Give each XCD label one contiguous run of tiles; bijective for every nblocks.
__device__ __forceinline__ int xcd_contiguous(int b, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = b & 7, slot = b >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// One batch of N neighbour gathers for the row(s) this wavefront is working on.  Every load is
// UNCONDITIONAL (slots past the end of a row re-read the row's last neighbour, an L1 hit, and are
// dropped by a select): hipcc puts `s_waitcnt vmcnt(0)` in front of every load that sits behind
// a branch, which serialises the gathers -- the batch must be branch-free to keep N*R*16 B per
// lane in flight.
template <typename T, int R, int N>
__device__ __forceinline__ void gather_batch(const int2* __restrict__ edges, int k, int ke,
                                             const typename Vt<T>::elem* __restrict__ X, int64_t ldx,
                                             const int (&voff)[R], float (&acc)[R][Vt<T>::VEC]) {
  using V = Vt<T>;
  using raw_t = typename V::raw;
  constexpr int VEC = V::VEC;
  int2 e[N];
#pragma unroll
  for (int u = 0; u < N; ++u) {
    int kc = k + u < ke ? k + u : ke - 1;
    kc = kc < 0 ? 0 : kc;
    e[u] = edges[kc];
  }
  raw_t xv[N][R];
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const typename V::elem* src = X + (int64_t)e[u].x * ldx;
#pragma unroll
    for (int r = 0; r < R; ++r) xv[u][r] = *(const raw_t*)(src + voff[r]);
  }
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const bool valid = k + u < ke;
    const float w = __int_as_float(e[u].y);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float f[VEC];
      V::unpack(xv[u][r], f);
#pragma unroll
      for (int c = 0; c < VEC; ++c) acc[r][c] = valid ? fmaf(w, f[c], acc[r][c]) : acc[r][c];
    }
  }
}

template <typename T, int G, int R, int NEPI>
__global__ __launch_bounds__(kBlock) void spmm_rows(const SpmmArgs a, const int ch, const int nblocks,
                                                     const int flags) {
  using V = Vt<T>;
  constexpr int VEC = V::VEC;
  constexpr int RPW = 64 / G;  // rows a wavefront works on simultaneously
  using raw_t = typename V::raw;
  using elem_t = typename V::elem;

  __shared__ int32_t s_rp[kWaves][kChMax + 1];
  __shared__ float s_sd[kWaves][kChMax];
  __shared__ int2 s_e[kWaves][kCap];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tile = (flags & kFlagXcdMap) ? xcd_contiguous(blockIdx.x, nblocks) : (int)blockIdx.x;
  const int r0 = (tile * kWaves + wave) * ch;
  int nrows = a.n_rows - r0;
  nrows = nrows < 0 ? 0 : (nrows > ch ? ch : nrows);

  // ---- stage this chunk's row pointers, row scales and neighbour list in LDS (coalesced) ----
  if (nrows > 0) {
    for (int l = lane; l <= nrows; l += 64) s_rp[wave][l] = a.rowptr[r0 + l];
    for (int l = lane; l < nrows; l += 64) s_sd[wave][l] = a.scale_dst ? a.scale_dst[r0 + l] : 1.0f;
  }
  __syncthreads();
  int e0 = 0, ne = 0;
  if (nrows > 0) {
    e0 = s_rp[wave][0];
    ne = s_rp[wave][nrows] - e0;
  }
  const bool staged = ne <= kCap;  // wave-uniform; a chunk holding a huge row takes the slow path
  if (staged) {
    for (int k = lane; k < ne; k += 64) {
      const int j = a.idx[e0 + k];
      const float w = a.scale_src ? a.scale_src[j] : 1.0f;
      s_e[wave][k] = make_int2(j, __float_as_int(w));
    }
  }
  __syncthreads();

  const int g = lane / G;   // which of the RPW simultaneous rows
  const int gl = lane % G;  // lane inside the row group
  const int nvec = a.C / VEC;
  const elem_t* __restrict__ X = (const elem_t*)a.X;
  const elem_t* __restrict__ X0 = (const elem_t*)a.X0;
  const elem_t* __restrict__ X1 = (const elem_t*)a.X1;
  elem_t* __restrict__ Y = (elem_t*)a.Y;
  const int2* __restrict__ edges = s_e[wave];

  int voff[R];     // element offset of this lane's r-th vector inside a row (clamped: loads stay in bounds)
  bool vok[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int vi = gl + r * G;
    vok[r] = vi < nvec;
    voff[r] = (vok[r] ? vi : nvec - 1) * VEC;
  }

  for (int it = 0; it * RPW < nrows; ++it) {
    const int lr = it * RPW + g;
    const bool rvalid = lr < nrows;
    const int lrc = rvalid ? lr : 0;
    const int row = r0 + lrc;
    const int ks = s_rp[wave][lrc] - e0;
    const int ke = rvalid ? s_rp[wave][lrc + 1] - e0 : ks;
    const float sdst = a.alpha * s_sd[wave][lrc];
    // epilogue operands are issued first so they fly together with the gathers
    raw_t x0v[R], x1v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (NEPI >= 1) x0v[r] = *(const raw_t*)(X0 + (int64_t)row * a.ldx0 + voff[r]);
      if (NEPI >= 2) x1v[r] = *(const raw_t*)(X1 + (int64_t)row * a.ldx1 + voff[r]);
    }
    float acc[R][VEC];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < VEC; ++c) acc[r][c] = 0.f;

    if (staged) {
      int k = ks;
      while (__any(k < ke)) {                 // wave-uniform trip count: the longest row decides
        const int rem = ke - k;
        if (R * 8 <= 16 && __any(rem > 6)) { gather_batch<T, R, 8>(edges, k, ke, X, a.ldx, voff, acc); k += 8; }
        else if (R * 6 <= 16 && __any(rem > 4)) { gather_batch<T, R, 6>(edges, k, ke, X, a.ldx, voff, acc); k += 6; }
        else if (R * 4 <= 16 && __any(rem > 2)) { gather_batch<T, R, 4>(edges, k, ke, X, a.ldx, voff, acc); k += 4; }
        else { gather_batch<T, R, 2>(edges, k, ke, X, a.ldx, voff, acc); k += 2; }
      }
    } else {
      for (int k = ks; k < ke; ++k) {         // rare: more than kCap neighbours in one chunk
        const int j = a.idx[e0 + k];
        const float w = a.scale_src ? a.scale_src[j] : 1.0f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float f[VEC];
          V::unpack(*(const raw_t*)(X + (int64_t)j * a.ldx + voff[r]), f);
#pragma unroll
          for (int c = 0; c < VEC; ++c) acc[r][c] = fmaf(w, f[c], acc[r][c]);
        }
      }
    }

#pragma unroll
    for (int r = 0; r < R; ++r) {
      float y[VEC];
#pragma unroll
      for (int c = 0; c < VEC; ++c) y[c] = sdst * acc[r][c];
      if (NEPI >= 1) {
        float f[VEC];
        V::unpack(x0v[r], f);
#pragma unroll
        for (int c = 0; c < VEC; ++c) y[c] = fmaf(a.beta, f[c], y[c]);
      }
      if (NEPI >= 2) {
        float f[VEC];
        V::unpack(x1v[r], f);
#pragma unroll
        for (int c = 0; c < VEC; ++c) y[c] = fmaf(a.gamma, f[c], y[c]);
      }
      if (rvalid && vok[r]) *(raw_t*)(Y + (int64_t)row * a.ldy + voff[r]) = V::pack(y);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-tiled variant for wide rows -- EXPERIMENTAL, off by default (SG_TUNE_TILED_MIN_ROW_BYTES).
// Measured on the 1 M-vertex mesh it is bit-identical to spmm_rows but 5-25 % SLOWER except at
// C = 512 fp32 (-6 %): with 69 KB of LDS only two workgroups fit a CU and every tile pays its
// metadata chain, the DMA wait and the epilogue-operand latency in sequence.  It needs a persistent,
// double-buffered pipeline across tiles to pay; kept as the starting point for that.
// With the vertices in a locality order (Morton), the 32 rows
// of a tile reference only ~70 DISTINCT source rows for their ~192 edges.  The workgroup pulls each
// distinct row ONCE, straight into LDS with direct-to-LDS loads (global_load_lds_dwordx4: no VGPRs
// held while in flight, 2 rows x 512 B per wavefront instruction), then every output row is
// reduced out of LDS (ds_read_b128 per edge).  L2 -> CU traffic drops from 6 to ~2.2 rows per
// output row.  Rows wider than 512 B are processed in 512-B column segments over the same tile
// metadata, so the staging buffer stays at kTileSlots x 512 B = 64 KB (two workgroups per CU: one
// loads while the other reduces).
// ---------------------------------------------------------------------------------------------
constexpr int kSegBytes = 512;

template <typename T, int NEPI>
__global__ __launch_bounds__(kBlock) void spmm_tiled(const SpmmArgs a, const int ntiles, const int flags) {
  using V = Vt<T>;
  constexpr int VEC = V::VEC;
  using raw_t = typename V::raw;
  using elem_t = typename V::elem;
  constexpr int kSegElems = kSegBytes / (int)sizeof(elem_t);

  __shared__ __attribute__((aligned(16))) unsigned char s_rows[kTileSlots * kSegBytes];
  __shared__ int32_t s_uniq[kTileSlots];
  __shared__ int2 s_edge[kTileEdges];          // (byte offset of the slot in s_rows, weight bits)
  __shared__ int32_t s_rp[kTileRows + 1];
  __shared__ float s_sd[kTileRows];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, hl = lane & 31;
  const int tile = (flags & kFlagXcdMap) ? xcd_contiguous(blockIdx.x, ntiles) : (int)blockIdx.x;
  const int r0 = tile * kTileRows;
  int nrows = a.n_rows - r0;
  nrows = nrows > kTileRows ? kTileRows : nrows;

  // ---- tile metadata -> LDS ----
  for (int l = tid; l <= nrows; l += kBlock) s_rp[l] = a.rowptr[r0 + l];
  for (int l = tid; l < nrows; l += kBlock) s_sd[l] = a.scale_dst ? a.scale_dst[r0 + l] : 1.0f;
  const int u0 = a.tile_uptr[tile], nu = a.tile_uptr[tile + 1] - u0;
  for (int l = tid; l < nu; l += kBlock) s_uniq[l] = a.tile_uniq[u0 + l];
  __syncthreads();
  const int e0 = s_rp[0], ne = s_rp[nrows] - e0;
  for (int k = tid; k < ne; k += kBlock) {
    const int slot = a.tile_eloc[e0 + k];
    const float w = a.scale_src ? a.scale_src[s_uniq[slot]] : 1.0f;
    s_edge[k] = make_int2(slot * kSegBytes, __float_as_int(w));
  }
  __syncthreads();

  const elem_t* __restrict__ X = (const elem_t*)a.X;
  const elem_t* __restrict__ X0 = (const elem_t*)a.X0;
  const elem_t* __restrict__ X1 = (const elem_t*)a.X1;
  elem_t* __restrict__ Y = (elem_t*)a.Y;
  const int nseg = (a.C * (int)sizeof(elem_t) + kSegBytes - 1) / kSegBytes;

  for (int seg = 0; seg < nseg; ++seg) {
    const int c0 = seg * kSegElems;                              // first channel of this segment
    int cvec = (a.C - c0) / VEC;                                 // 16-B vectors in this segment
    cvec = cvec > 32 ? 32 : cvec;
    const int voff = c0 + (hl < cvec ? hl : cvec - 1) * VEC;     // clamped: loads stay inside the row
    // ---- stage the distinct source rows of this tile (this segment) in LDS, 2 rows per instruction ----
    for (int q = wave; 2 * q < nu; q += kWaves) {
      int slot = 2 * q + half;
      slot = slot < nu ? slot : nu - 1;
      const elem_t* src = X + (int64_t)s_uniq[slot] * a.ldx + voff;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(s_rows + 2 * q * kSegBytes), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- reduce: each half-wavefront owns one output row at a time ----
    for (int lr = wave * 2 + half; lr < nrows; lr += 2 * kWaves) {
      const int row = r0 + lr;
      const int ks = s_rp[lr] - e0, ke = s_rp[lr + 1] - e0;
      raw_t x0v, x1v;
      if (NEPI >= 1) x0v = *(const raw_t*)(X0 + (int64_t)row * a.ldx0 + voff);
      if (NEPI >= 2) x1v = *(const raw_t*)(X1 + (int64_t)row * a.ldx1 + voff);
      float acc[VEC];
#pragma unroll
      for (int c = 0; c < VEC; ++c) acc[c] = 0.f;
      int k = ks;
      for (; k + 4 <= ke; k += 4) {
        int2 e[4];
        raw_t xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = s_edge[k + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = *(const raw_t*)(s_rows + e[u].x + hl * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float f[VEC];
          V::unpack(xv[u], f);
          const float w = __int_as_float(e[u].y);
#pragma unroll
          for (int c = 0; c < VEC; ++c) acc[c] = fmaf(w, f[c], acc[c]);
        }
      }
      for (; k < ke; ++k) {
        const int2 e = s_edge[k];
        float f[VEC];
        V::unpack(*(const raw_t*)(s_rows + e.x + hl * 16), f);
        const float w = __int_as_float(e.y);
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[c] = fmaf(w, f[c], acc[c]);
      }
      const float sdst = a.alpha * s_sd[lr];
      float y[VEC];
#pragma unroll
      for (int c = 0; c < VEC; ++c) y[c] = sdst * acc[c];
      if (NEPI >= 1) {
        float f[VEC];
        V::unpack(x0v, f);
#pragma unroll
        for (int c = 0; c < VEC; ++c) y[c] = fmaf(a.beta, f[c], y[c]);
      }
      if (NEPI >= 2) {
        float f[VEC];
        V::unpack(x1v, f);
#pragma unroll
        for (int c = 0; c < VEC; ++c) y[c] = fmaf(a.gamma, f[c], y[c]);
      }
      if (hl < cvec) *(raw_t*)(Y + (int64_t)row * a.ldy + voff) = V::pack(y);
    }
    __syncthreads();   // all reads of this segment's rows are done before the next segment lands
  }
}

// Any C, any stride, any alignment: one thread per output element, lanes along the channel.
template <typename T>
__global__ __launch_bounds__(kBlock) void spmm_scalar(const SpmmArgs a) {
  using V = Vt<T>;
  const int64_t total = (int64_t)a.n_rows * a.C;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * kBlock) {
    const int row = (int)(t / a.C);
    const int c = (int)(t - (int64_t)row * a.C);
    float acc = 0.f;
    for (int k = a.rowptr[row]; k < a.rowptr[row + 1]; ++k) {
      const int j = a.idx[k];
      const float w = a.scale_src ? a.scale_src[j] : 1.0f;
      acc = fmaf(w, V::load1(a.X, (int64_t)j * a.ldx + c), acc);
    }
    float y = a.alpha * (a.scale_dst ? a.scale_dst[row] : 1.0f) * acc;
    if (a.X0) y = fmaf(a.beta, V::load1(a.X0, (int64_t)row * a.ldx0 + c), y);
    if (a.X1) y = fmaf(a.gamma, V::load1(a.X1, (int64_t)row * a.ldx1 + c), y);
    V::store1(a.Y, (int64_t)row * a.ldy + c, y);
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void gather_rows_vec(const int32_t* __restrict__ rows, int64_t n,
                                                          const void* X, int64_t ldx, void* Y,
                                                          int64_t ldy, int nvec) {
  using V = Vt<T>;
  using raw_t = typename V::raw;
  using elem_t = typename V::elem;
  const int64_t total = n * nvec;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * kBlock) {
    const int64_t i = t / nvec;
    const int vi = (int)(t - i * nvec);
    const raw_t v = *(const raw_t*)((const elem_t*)X + (int64_t)rows[i] * ldx + vi * V::VEC);
    *(raw_t*)((elem_t*)Y + i * ldy + vi * V::VEC) = v;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void gather_rows_scalar(const int32_t* __restrict__ rows, int64_t n,
                                                             const void* X, int64_t ldx, void* Y,
                                                             int64_t ldy, int C) {
  using V = Vt<T>;
  const int64_t total = n * C;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * kBlock) {
    const int64_t i = t / C;
    const int c = (int)(t - i * C);
    V::store1(Y, i * ldy + c, V::load1(X, (int64_t)rows[i] * ldx + c));
  }
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

struct Tuning {
  int chunk_rows = 0;             // 0 = automatic
  int flags = kFlagXcdMap;        // kFlag* bits
  int unroll = 0;                 // 0 = default per shape
  int slab = 0;                   // channels per column slab; 0 = whole rows
  int tiled_min_row_bytes = 0;    // > 0: rows at least this wide take the LDS-tiled kernel (experimental, off)
};
Tuning g_tuning;

template <typename T, int G, int R, int NEPI>
int launch_rows_epi(const SpmmArgs& a, hipStream_t stream) {
  constexpr int RPW = 64 / G;
  // rows per wavefront chunk: large enough to amortise staging, small enough to keep the band
  // of rows that are in flight at once (the sweep front) thin, and >> 256 workgroups in flight
  int ch = g_tuning.chunk_rows;
  if (ch <= 0) {
    // measured on the 1 M-vertex mesh (tools/agg_bench.py): wide rows want a thin sweep front
    const int row_bytes = a.C * (int)sizeof(typename Vt<T>::elem);
    ch = row_bytes >= 2048 ? 4 : row_bytes >= 1024 ? (sizeof(typename Vt<T>::elem) == 4 ? 4 : 8) : 16;
    while (ch > RPW && (int64_t)a.n_rows / ch < 4096) ch >>= 1;
  }
  if (ch < RPW) ch = RPW;
  if (ch > kChMax) ch = kChMax;
  ch = (ch / RPW) * RPW;
  const int64_t chunks = ((int64_t)a.n_rows + ch - 1) / ch;
  const int nblocks = (int)((chunks + kWaves - 1) / kWaves);
  spmm_rows<T, G, R, NEPI><<<nblocks, kBlock, 0, stream>>>(a, ch, nblocks, g_tuning.flags);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

template <typename T>
int launch_tiled(const SpmmArgs& a, hipStream_t stream) {
  const int ntiles = (a.n_rows + kTileRows - 1) / kTileRows;
  if (a.X0 && a.X1) spmm_tiled<T, 2><<<ntiles, kBlock, 0, stream>>>(a, ntiles, g_tuning.flags);
  else if (a.X0) spmm_tiled<T, 1><<<ntiles, kBlock, 0, stream>>>(a, ntiles, g_tuning.flags);
  else if (a.X1) {
    SpmmArgs b = a;
    b.X0 = a.X1; b.ldx0 = a.ldx1; b.beta = a.gamma; b.X1 = nullptr; b.ldx1 = 0; b.gamma = 0.f;
    spmm_tiled<T, 1><<<ntiles, kBlock, 0, stream>>>(b, ntiles, g_tuning.flags);
  } else spmm_tiled<T, 0><<<ntiles, kBlock, 0, stream>>>(a, ntiles, g_tuning.flags);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

template <typename T, int G, int R>
int launch_rows(const SpmmArgs& a, hipStream_t stream) {
  if (a.X0 && a.X1) return launch_rows_epi<T, G, R, 2>(a, stream);
  if (a.X0) return launch_rows_epi<T, G, R, 1>(a, stream);
  if (a.X1) {  // only X1 given: treat it as the first operand
    SpmmArgs b = a;
    b.X0 = a.X1; b.ldx0 = a.ldx1; b.beta = a.gamma; b.X1 = nullptr; b.ldx1 = 0; b.gamma = 0.f;
    return launch_rows_epi<T, G, R, 1>(b, stream);
  }
  return launch_rows_epi<T, G, R, 0>(a, stream);
}

template <typename T>
int launch_typed_one(const SpmmArgs& a, hipStream_t stream) {
  constexpr int VEC = Vt<T>::VEC;
  constexpr int esz = sizeof(typename Vt<T>::elem);
  const bool vec_ok = a.C % VEC == 0 && a.ldx % VEC == 0 && a.ldy % VEC == 0 && aligned16(a.X) &&
                      aligned16(a.Y) && (!a.X0 || (a.ldx0 % VEC == 0 && aligned16(a.X0))) &&
                      (!a.X1 || (a.ldx1 % VEC == 0 && aligned16(a.X1))) && a.C / VEC <= 256;
  (void)esz;
  if (!vec_ok) {
    const int64_t total = (int64_t)a.n_rows * a.C;
    int64_t nb = (total + kBlock - 1) / kBlock;
    if (nb > 256 * 32) nb = 256 * 32;
    spmm_scalar<T><<<(int)nb, kBlock, 0, stream>>>(a);
    SG_HIP_TRY(hipGetLastError());
    return SG_OK;
  }
  const int nvec = a.C / VEC;
  // wide rows of a tileable graph: stage each distinct source row once in LDS
  const int row_bytes = a.C * (int)sizeof(typename Vt<T>::elem);
  if (a.tile_uptr && g_tuning.tiled_min_row_bytes > 0 && row_bytes >= g_tuning.tiled_min_row_bytes &&
      !(g_tuning.flags & kFlagNoTiles))
    return launch_tiled<T>(a, stream);
  if (nvec <= 1) return launch_rows<T, 1, 1>(a, stream);
  if (nvec <= 2) return launch_rows<T, 2, 1>(a, stream);
  if (nvec <= 4) return launch_rows<T, 4, 1>(a, stream);
  if (nvec <= 8) return launch_rows<T, 8, 1>(a, stream);
  if (nvec <= 16) return launch_rows<T, 16, 1>(a, stream);
  if (nvec <= 32) return launch_rows<T, 32, 1>(a, stream);
  if (nvec <= 64) return launch_rows<T, 64, 1>(a, stream);
  if (nvec <= 128) return launch_rows<T, 64, 2>(a, stream);
  return launch_rows<T, 64, 4>(a, stream);
}

// Column slabs: sweep all rows once per slab of `slab` channels, so that the rows a sweep
// keeps re-gathering (the neighbouring "lines" of the mesh) shrink to slab-wide segments
// that stay L2-resident.
template <typename T>
int launch_typed(const SpmmArgs& a, hipStream_t stream) {
  constexpr int VEC = Vt<T>::VEC;
  const int slab = g_tuning.slab;
  if (slab <= 0 || a.C <= slab || slab % VEC != 0) return launch_typed_one<T>(a, stream);
  using elem_t = typename Vt<T>::elem;
  for (int c0 = 0; c0 < a.C; c0 += slab) {
    SpmmArgs s = a;
    s.C = a.C - c0 < slab ? a.C - c0 : slab;
    s.X = (const elem_t*)a.X + c0;
    s.Y = (elem_t*)a.Y + c0;
    if (a.X0) s.X0 = (const elem_t*)a.X0 + c0;
    if (a.X1) s.X1 = (const elem_t*)a.X1 + c0;
    int rc = launch_typed_one<T>(s, stream);
    if (rc != SG_OK) return rc;
  }
  return SG_OK;
}

}  // namespace

bool tiles_enabled() { return g_tuning.tiled_min_row_bytes > 0; }

int set_tuning(int knob, int value) {
  switch (knob) {
    case SG_TUNE_CHUNK_ROWS: g_tuning.chunk_rows = value; return SG_OK;
    case SG_TUNE_FLAGS: g_tuning.flags = value; return SG_OK;
    case SG_TUNE_UNROLL: g_tuning.unroll = value; return SG_OK;
    case SG_TUNE_SLAB: g_tuning.slab = value; return SG_OK;
    case SG_TUNE_TILED_MIN_ROW_BYTES: g_tuning.tiled_min_row_bytes = value; return SG_OK;
    default: set_error("unknown tuning knob %d", knob); return SG_ERR_INVALID;
  }
}

int launch_spmm(const SpmmArgs& a, int dtype, hipStream_t stream) {
  if (a.n_rows == 0 || a.C == 0) return SG_OK;
  switch (dtype) {
    case SG_F32: return launch_typed<float>(a, stream);
    case SG_BF16: return launch_typed<bf16_tag>(a, stream);
    default: set_error("unsupported dtype %d", dtype); return SG_ERR_UNSUPPORTED;
  }
}

int launch_gather_rows(const int32_t* rows, int64_t n, const void* X, int64_t ldx, void* Y,
                       int64_t ldy, int64_t C, int dtype, hipStream_t stream) {
  if (n == 0 || C == 0) return SG_OK;
  auto run = [&](auto tag) -> int {
    using T = decltype(tag);
    constexpr int VEC = Vt<T>::VEC;
    const bool vec_ok = C % VEC == 0 && ldx % VEC == 0 && ldy % VEC == 0 && aligned16(X) && aligned16(Y);
    const int64_t total = vec_ok ? n * (C / VEC) : n * C;
    int64_t nb = (total + kBlock - 1) / kBlock;
    if (nb > 256 * 16) nb = 256 * 16;
    if (vec_ok)
      gather_rows_vec<T><<<(int)nb, kBlock, 0, stream>>>(rows, n, X, ldx, Y, ldy, (int)(C / VEC));
    else
      gather_rows_scalar<T><<<(int)nb, kBlock, 0, stream>>>(rows, n, X, ldx, Y, ldy, (int)C);
    SG_HIP_TRY(hipGetLastError());
    return SG_OK;
  };
  switch (dtype) {
    case SG_F32: return run(float{});
    case SG_BF16: return run(bf16_tag{});
    default: set_error("unsupported dtype %d", dtype); return SG_ERR_UNSUPPORTED;
  }
}

}  // namespace sg
