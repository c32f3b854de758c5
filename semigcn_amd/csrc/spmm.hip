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
//  * bf16 rows of 128 / 256 channels (the launches that dominate the iteration) take spmm_ring further down: 16-row tiles
//    whose distinct source rows are staged ONCE in LDS by a pipelined LDS-DMA ring and reduced on the matrix cores.
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

enum : int { kFlagXcdMap = 1, kFlagNoTiles = 2, kFlagNoPackedScale = 4, kFlagBlockBarrier = 8, kFlagWideAddr = 16, kFlagNoExact = 32,
             kFlagStreamEpilogue = 64, kFlagLdsTiles = 128, kFlagNoQuad = 256, kFlagThreadRows = 512, kFlagNoRing = 2048, kFlagRingDirectStores = 4096,
             kFlagNoRingF32 = 8192 };

// Every wavefront stages ITS OWN chunk in its own LDS slice, so nothing crosses wavefronts: LDS operations of
// one wavefront complete in issue order, and a compiler-level wave barrier keeps the reads behind the writes.
// A workgroup barrier here would march the four wavefronts of a block through the metadata / gather / store
// phases in lock-step (kFlagBlockBarrier restores that for A/B timing).
__device__ __forceinline__ void wave_sync(int flags) {
  if (flags & kFlagBlockBarrier) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

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
typedef float f32x2 __attribute__((ext_vector_type(2)));

// acc += w * f over VEC values as VEC/2 packed v_pk_fma_f32 (two fp32 FMAs per issued instruction)
template <int VEC>
__device__ __forceinline__ void axpy(float w, const float (&f)[VEC], float (&acc)[VEC]) {
  const f32x2 w2 = {w, w};
#pragma unroll
  for (int c = 0; c < VEC; c += 2) {
    const f32x2 v = {f[c], f[c + 1]};
    f32x2 a = {acc[c], acc[c + 1]};
    a = __builtin_elementwise_fma(w2, v, a);
    acc[c] = a.x;
    acc[c + 1] = a.y;
  }
}

template <typename T, int R, int N, bool NARROW, bool EXACT = false>
__device__ __forceinline__ void gather_batch(const int2* __restrict__ edges, int k, int ke, int klast,
                                             const typename Vt<T>::elem* __restrict__ X, int64_t ldx,
                                             const int (&voff)[R], float (&acc)[R][Vt<T>::VEC]) {
  using V = Vt<T>;
  using raw_t = typename V::raw;
  using elem_t = typename V::elem;
  constexpr int VEC = V::VEC;
  // Slots past the end of this lane group's row re-read the row's LAST neighbour (klast; an L1 hit) and are
  // switched off through the weight: one select per slot instead of one per accumulated value.  An empty
  // row has klast = its start, i.e. the next row's first neighbour or the sentinel behind the list (row 0,
  // weight 0).  Consequence: a non-finite value in such a re-read row surfaces as NaN (0 * inf).
  // EXACT (one row per wavefront, so the caller can size the batch to the row): every slot is live -- no clamp,
  // no select, and the N list entries are read at immediate offsets from one LDS address.
  int2 e[N];
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const int kc = EXACT ? k + u : (k + u < klast ? k + u : klast);
    e[u] = edges[kc];
  }
  raw_t xv[N][R];
#pragma unroll
  for (int u = 0; u < N; ++u) {
    if (NARROW) {
      // byte offsets fit 32 bits, row ids and the row pitch fit 24: one full-rate v_mad_u32_u24 per gather
      // instead of a 64-bit multiply-add chain (quarter rate) -- this kernel is VALU-issue-bound
      const uint32_t base = __umul24((uint32_t)e[u].x, (uint32_t)ldx * (uint32_t)sizeof(elem_t));
#pragma unroll
      for (int r = 0; r < R; ++r)
        xv[u][r] = *(const raw_t*)((const char*)X + (base + (uint32_t)voff[r] * (uint32_t)sizeof(elem_t)));
    } else {
      const elem_t* src = X + (int64_t)e[u].x * ldx;
#pragma unroll
      for (int r = 0; r < R; ++r) xv[u][r] = *(const raw_t*)(src + voff[r]);
    }
  }
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const float w = (EXACT || k + u < ke) ? __int_as_float(e[u].y) : 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float f[VEC];
      V::unpack(xv[u][r], f);
      axpy<VEC>(w, f, acc[r]);
    }
  }
}

// CAP = most gathers a lane group issues back to back (8, 6 or 4): the loaded vectors of a batch are live together, so
// CAP sets the VGPR budget and with it the wavefronts per SIMD (bf16 C=256: 106 / 84 / 74 VGPRs = 4 / 5 / 6 waves).
// rocprofv3 PMC (profiles/r02_pmc_issue_spmm.json): wavefronts of this kernel sit in s_waitcnt 70 % of their
// lifetime at 3.6 resident waves per SIMD -- it is bound by memory latency x resident waves, not by instruction issue.
template <typename T, int G, int R, int NEPI, bool NARROW, int CAP>
__global__ __launch_bounds__(kBlock) void spmm_rows(const SpmmArgs a, const int ch, const int nblocks,
                                                     const int flags) {
  using V = Vt<T>;
  constexpr int VEC = V::VEC;
  constexpr int RPW = 64 / G;  // rows a wavefront works on simultaneously
  using raw_t = typename V::raw;
  using elem_t = typename V::elem;

  __shared__ int32_t s_rp[kWaves][kChMax + 1];
  __shared__ float s_sd[kWaves][kChMax];
  __shared__ int32_t s_row[kWaves][kChMax];   // caller's row id of each processed row (Csr::row_id)
  __shared__ int2 s_e[kWaves][kCap + 1];   // + the sentinel slot of gather_batch

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tile = (flags & kFlagXcdMap) ? xcd_contiguous(blockIdx.x, nblocks) : (int)blockIdx.x;
  const int r0 = (tile * kWaves + wave) * ch;
  int nrows = a.n_rows - r0;
  nrows = nrows < 0 ? 0 : (nrows > ch ? ch : nrows);

  // ---- stage this chunk's row pointers, row scales, row ids and neighbour list in LDS (coalesced) ----
  if (nrows > 0) {
    for (int l = lane; l <= nrows; l += 64) s_rp[wave][l] = a.rowptr[r0 + l];
    for (int l = lane; l < nrows; l += 64) {
      s_sd[wave][l] = a.scale_dst ? a.scale_dst[r0 + l] : 1.0f;
      s_row[wave][l] = a.row_id ? a.row_id[r0 + l] : r0 + l;
    }
  }
  wave_sync(flags);
  int e0 = 0, ne = 0;
  if (nrows > 0) {
    e0 = s_rp[wave][0];
    ne = s_rp[wave][nrows] - e0;
  }
  const bool staged = ne <= kCap;  // wave-uniform; a chunk holding a huge row takes the slow path
  if (staged) {
    if (a.idx_w) {
      for (int k = lane; k < ne; k += 64) s_e[wave][k] = a.idx_w[e0 + k];
    } else {
      for (int k = lane; k < ne; k += 64) {
        const int j = a.idx[e0 + k];
        const float w = a.scale_src ? a.scale_src[j] : 1.0f;
        s_e[wave][k] = make_int2(j, __float_as_int(w));
      }
    }
    if (lane == 0) s_e[wave][ne] = make_int2(0, 0);
  }
  wave_sync(flags);

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
    const int row = nrows > 0 ? s_row[wave][lrc] : 0;
    const int ks = s_rp[wave][lrc] - e0;
    const int ke = rvalid ? s_rp[wave][lrc + 1] - e0 : ks;
    const float sdst = a.alpha * s_sd[wave][lrc];
    // epilogue operands are issued first so they fly together with the gathers
    raw_t x0v[R], x1v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      // the epilogue operands and Y are touched once per launch: with kFlagStreamEpilogue they go around the caches'
      // retention (nontemporal), leaving L2 to the gathered rows, which ARE re-used
      if (flags & kFlagStreamEpilogue) {
        if (NEPI >= 1) x0v[r] = __builtin_nontemporal_load((const raw_t*)(X0 + (int64_t)row * a.ldx0 + voff[r]));
        if (NEPI >= 2) x1v[r] = __builtin_nontemporal_load((const raw_t*)(X1 + (int64_t)row * a.ldx1 + voff[r]));
      } else {
        if (NEPI >= 1) x0v[r] = *(const raw_t*)(X0 + (int64_t)row * a.ldx0 + voff[r]);
        if (NEPI >= 2) x1v[r] = *(const raw_t*)(X1 + (int64_t)row * a.ldx1 + voff[r]);
      }
    }
    float acc[R][VEC];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int c = 0; c < VEC; ++c) acc[r][c] = 0.f;

    if (staged) {
      int k = ks;
      bool uniform = RPW == 1;
      if (RPW == 2) {   // two rows side by side: when they happen to have the same length the exact path serves both
        const int d = ke - ks;
        uniform = __builtin_amdgcn_readlane(d, 0) == __builtin_amdgcn_readlane(d, 32);
      }
      if (uniform && !(flags & kFlagNoExact)) {
        // every lane group of the wavefront has the same row length: it is wave-uniform, so the batches are cut
        // to fit exactly (always the case with one row per wavefront)
        constexpr int NMAX = (R * 8 <= 16 && CAP >= 8) ? 8 : (R * 4 <= 16 ? (CAP >= 6 && R * 6 <= 16 ? 6 : 4) : 2);
        int rem = __builtin_amdgcn_readfirstlane(ke - ks);
        while (rem > 0) {
          const int n = rem < NMAX ? rem : NMAX;
          switch (n) {
            case 8: if (NMAX >= 8) gather_batch<T, R, 8, NARROW, true>(edges, k, ke, 0, X, a.ldx, voff, acc); break;
            case 7: if (NMAX >= 8) gather_batch<T, R, 7, NARROW, true>(edges, k, ke, 0, X, a.ldx, voff, acc); break;
            case 6: if (NMAX >= 6) gather_batch<T, R, 6, NARROW, true>(edges, k, ke, 0, X, a.ldx, voff, acc); break;
            case 5: if (NMAX >= 6) gather_batch<T, R, 5, NARROW, true>(edges, k, ke, 0, X, a.ldx, voff, acc); break;
            case 4: if (NMAX >= 4) gather_batch<T, R, 4, NARROW, true>(edges, k, ke, 0, X, a.ldx, voff, acc); break;
            case 3: if (NMAX >= 4) gather_batch<T, R, 3, NARROW, true>(edges, k, ke, 0, X, a.ldx, voff, acc); break;
            case 2: gather_batch<T, R, 2, NARROW, true>(edges, k, ke, 0, X, a.ldx, voff, acc); break;
            default: gather_batch<T, R, 1, NARROW, true>(edges, k, ke, 0, X, a.ldx, voff, acc); break;
          }
          k += n;
          rem -= n;
        }
      } else {
      // narrow rows: slots past the row's end run on into the next rows' neighbours (a free prefetch of what this
      // wavefront gathers next, measured faster up to 256-B rows); wide rows: re-read the row's last neighbour
      const int klast = a.C * (int)sizeof(elem_t) <= 256 ? ne : (ke - 1 > ks ? ke - 1 : ks);
      while (__any(k < ke)) {                 // wave-uniform trip count: the longest row decides
        const int rem = ke - k;
        if (CAP >= 8 && R * 8 <= 16 && __any(rem > 6)) { gather_batch<T, R, 8, NARROW>(edges, k, ke, klast, X, a.ldx, voff, acc); k += 8; }
        else if (CAP >= 6 && R * 6 <= 16 && __any(rem > 4)) { gather_batch<T, R, 6, NARROW>(edges, k, ke, klast, X, a.ldx, voff, acc); k += 6; }
        else if (R * 4 <= 16 && __any(rem > 2)) { gather_batch<T, R, 4, NARROW>(edges, k, ke, klast, X, a.ldx, voff, acc); k += 4; }
        else { gather_batch<T, R, 2, NARROW>(edges, k, ke, klast, X, a.ldx, voff, acc); k += 2; }
      }
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
      if (rvalid && vok[r]) {
        if (flags & kFlagStreamEpilogue) __builtin_nontemporal_store(V::pack(y), (raw_t*)(Y + (int64_t)row * a.ldy + voff[r]));
        else *(raw_t*)(Y + (int64_t)row * a.ldy + voff[r]) = V::pack(y);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Shared-gather variant for wide rows.  With the vertices in a locality order (Morton), the 4 rows
// of a mini-tile reference ~17 DISTINCT source rows for their ~24 edges.  A lane group gathers each
// distinct row ONCE into registers and feeds up to 4 row accumulators from it (a 4-bit adjacency
// mask per distinct row says which), so the L2 -> CU gather traffic drops from 6.0 to ~4.2 rows per
// output row.  Per row the neighbours are still visited in ascending-id (CSR) order with the same
// fma chain, so the result is bit-identical to spmm_rows.  Metadata: Csr::tile_* (csr_build.hip).
// ---------------------------------------------------------------------------------------------
constexpr int kShTilesPerGroup = 2;                       // mini-tiles a lane group works through per wavefront

template <typename T, int R, int N, bool NARROW, bool EXACT = false>
__device__ __forceinline__ void shared_batch(const int2* __restrict__ su, const uint32_t* __restrict__ sm, int k, int nu,
                                             const typename Vt<T>::elem* __restrict__ X, int64_t ldx,
                                             const int (&voff)[R], float (&acc)[kTileRows][R][Vt<T>::VEC]) {
  using V = Vt<T>;
  using raw_t = typename V::raw;
  using elem_t = typename V::elem;
  constexpr int VEC = V::VEC;
  int2 e[N];
  uint32_t m[N];
#pragma unroll
  for (int u = 0; u < N; ++u) {
    if (EXACT) {        // the caller cut the batch to the tile's source list: every slot is live
      e[u] = su[k + u];
      m[u] = sm[k + u];
    } else {
      int kc = k + u < nu ? k + u : nu - 1;
      kc = kc < 0 ? 0 : kc;
      e[u] = su[kc];
      m[u] = k + u < nu ? sm[kc] : 0u;
    }
  }
  raw_t xv[N][R];
#pragma unroll
  for (int u = 0; u < N; ++u) {
    if (NARROW) {
      const uint32_t base = __umul24((uint32_t)e[u].x, (uint32_t)ldx * (uint32_t)sizeof(elem_t));
#pragma unroll
      for (int r = 0; r < R; ++r)
        xv[u][r] = *(const raw_t*)((const char*)X + (base + (uint32_t)voff[r] * (uint32_t)sizeof(elem_t)));
    } else {
      const elem_t* src = X + (int64_t)e[u].x * ldx;
#pragma unroll
      for (int r = 0; r < R; ++r) xv[u][r] = *(const raw_t*)(src + voff[r]);
    }
  }
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const float w = __int_as_float(e[u].y);
    float wq[kTileRows];      // the source's weight for each of the tile's rows, 0 where it is no neighbour
#pragma unroll
    for (int q = 0; q < kTileRows; ++q) wq[q] = (m[u] >> q) & 1u ? w : 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float f[VEC];
      V::unpack(xv[u][r], f);
#pragma unroll
      for (int q = 0; q < kTileRows; ++q) axpy<VEC>(wq[q], f, acc[q][r]);
    }
  }
}

template <typename T, int G, int R, int NEPI, bool NARROW>
__global__ __launch_bounds__(kBlock) void spmm_shared(const SpmmArgs a, const int nblocks, const int flags) {
  using V = Vt<T>;
  constexpr int VEC = V::VEC;
  constexpr int RPW = 64 / G;                              // mini-tiles in flight per wavefront
  constexpr int MT = RPW * kShTilesPerGroup;               // mini-tiles per wavefront
  constexpr int ROWS = MT * kTileRows;
  constexpr int CAPW = MT * kTileSlots;
  using raw_t = typename V::raw;
  using elem_t = typename V::elem;

  __shared__ int32_t s_up[kWaves][MT + 1];
  __shared__ int32_t s_rp[kWaves][ROWS + 1];
  __shared__ float s_sd[kWaves][ROWS];
  __shared__ int32_t s_row[kWaves][ROWS];
  __shared__ int2 s_u[kWaves][CAPW];
  __shared__ uint32_t s_m[kWaves][CAPW];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = (flags & kFlagXcdMap) ? xcd_contiguous(blockIdx.x, nblocks) : (int)blockIdx.x;
  const int n_mt = (a.n_rows + kTileRows - 1) / kTileRows;
  const int m0 = (blk * kWaves + wave) * MT;
  int nm = n_mt - m0;
  nm = nm < 0 ? 0 : (nm > MT ? MT : nm);
  const int r0 = m0 * kTileRows;
  int nrows = a.n_rows - r0;
  nrows = nrows < 0 ? 0 : (nrows > ROWS ? ROWS : nrows);

  // ---- stage mini-tile metadata: distinct-source lists (+ scale), row pointers, adjacency masks ----
  if (nm > 0) {
    for (int l = lane; l <= nm; l += 64) s_up[wave][l] = a.tile_uptr[m0 + l];
    for (int l = lane; l <= nrows; l += 64) s_rp[wave][l] = a.rowptr[r0 + l];
    for (int l = lane; l < nrows; l += 64) {
      s_sd[wave][l] = a.scale_dst ? a.scale_dst[r0 + l] : 1.0f;
      s_row[wave][l] = a.row_id ? a.row_id[r0 + l] : r0 + l;
    }
  }
  wave_sync(flags);
  int ub = 0, nut = 0, e0 = 0, ne = 0;
  if (nm > 0) {
    ub = s_up[wave][0];
    nut = s_up[wave][nm] - ub;
    e0 = s_rp[wave][0];
    ne = s_rp[wave][nrows] - e0;
  }
  for (int k = lane; k < nut; k += 64) {
    if (a.tile_uniq_w) {
      s_u[wave][k] = a.tile_uniq_w[ub + k];
    } else {
      const int j = a.tile_uniq[ub + k];
      s_u[wave][k] = make_int2(j, __float_as_int(a.scale_src ? a.scale_src[j] : 1.0f));
    }
    s_m[wave][k] = 0u;
  }
  wave_sync(flags);
  for (int k = lane; k < ne; k += 64) {
    int lo = 0, hi = nrows;                                 // row of edge e0+k: last lr with s_rp[lr] <= e0+k
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_rp[wave][mid] - e0 <= k) lo = mid; else hi = mid;
    }
    const int mt = lo / kTileRows;
    atomicOr(&s_m[wave][s_up[wave][mt] - ub + a.tile_eloc[e0 + k]], 1u << (lo % kTileRows));
  }
  wave_sync(flags);

  const int g = lane / G, gl = lane % G;
  const int nvec = a.C / VEC;
  const elem_t* __restrict__ X = (const elem_t*)a.X;
  const elem_t* __restrict__ X0 = (const elem_t*)a.X0;
  const elem_t* __restrict__ X1 = (const elem_t*)a.X1;
  elem_t* __restrict__ Y = (elem_t*)a.Y;
  int voff[R];
  bool vok[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int vi = gl + r * G;
    vok[r] = vi < nvec;
    voff[r] = (vok[r] ? vi : nvec - 1) * VEC;
  }

  for (int it = 0; it * RPW < nm; ++it) {
    const int mt = it * RPW + g;
    const bool tvalid = mt < nm;
    const int mtc = tvalid ? mt : 0;
    const int u0 = s_up[wave][mtc] - ub;
    const int nu = tvalid ? s_up[wave][mtc + 1] - s_up[wave][mtc] : 0;
    const int lr0 = mtc * kTileRows;
    float acc[kTileRows][R][VEC];
#pragma unroll
    for (int q = 0; q < kTileRows; ++q)
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[q][r][c] = 0.f;

    const int2* su = s_u[wave] + u0;
    const uint32_t* sm = s_m[wave] + u0;
    int k = 0;
    if (RPW == 1 && !(flags & kFlagNoExact)) {     // one mini-tile per wavefront at a time: its source count is wave-uniform -> exact batches
      constexpr int NMAX = R * 8 <= 8 ? 8 : 4;
      int rem = __builtin_amdgcn_readfirstlane(nu);
      while (rem > 0) {
        const int n = rem < NMAX ? rem : NMAX;
        switch (n) {
          case 8: if (NMAX >= 8) shared_batch<T, R, 8, NARROW, true>(su, sm, k, nu, X, a.ldx, voff, acc); break;
          case 7: if (NMAX >= 8) shared_batch<T, R, 7, NARROW, true>(su, sm, k, nu, X, a.ldx, voff, acc); break;
          case 6: if (NMAX >= 8) shared_batch<T, R, 6, NARROW, true>(su, sm, k, nu, X, a.ldx, voff, acc); break;
          case 5: if (NMAX >= 8) shared_batch<T, R, 5, NARROW, true>(su, sm, k, nu, X, a.ldx, voff, acc); break;
          case 4: shared_batch<T, R, 4, NARROW, true>(su, sm, k, nu, X, a.ldx, voff, acc); break;
          case 3: shared_batch<T, R, 3, NARROW, true>(su, sm, k, nu, X, a.ldx, voff, acc); break;
          case 2: shared_batch<T, R, 2, NARROW, true>(su, sm, k, nu, X, a.ldx, voff, acc); break;
          default: shared_batch<T, R, 1, NARROW, true>(su, sm, k, nu, X, a.ldx, voff, acc); break;
        }
        k += n;
        rem -= n;
      }
    } else {
      while (__any(k < nu)) {
        const int rem = nu - k;
        if (R * 8 <= 8 && __any(rem > 4)) { shared_batch<T, R, 8, NARROW>(su, sm, k, nu, X, a.ldx, voff, acc); k += 8; }
        else if (__any(rem > 2)) { shared_batch<T, R, 4, NARROW>(su, sm, k, nu, X, a.ldx, voff, acc); k += 4; }
        else { shared_batch<T, R, 2, NARROW>(su, sm, k, nu, X, a.ldx, voff, acc); k += 2; }
      }
    }
    // epilogue operands AFTER the gathers: holding 4 rows x NEPI vectors across the gather loop costs a whole
    // occupancy step (130 -> ~100 VGPRs at C = 256 fp32); other wavefronts cover this one extra latency
    raw_t x0v[kTileRows][R], x1v[kTileRows][R];
#pragma unroll
    for (int q = 0; q < kTileRows; ++q) {
      const int lq = lr0 + q < nrows ? lr0 + q : (nrows > 0 ? nrows - 1 : 0);
      const int row = nrows > 0 ? s_row[wave][lq] : 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (NEPI >= 1) x0v[q][r] = *(const raw_t*)(X0 + (int64_t)row * a.ldx0 + voff[r]);
        if (NEPI >= 2) x1v[q][r] = *(const raw_t*)(X1 + (int64_t)row * a.ldx1 + voff[r]);
      }
    }

#pragma unroll
    for (int q = 0; q < kTileRows; ++q) {
      const int lr = lr0 + q;
      const bool rvalid = tvalid && lr < nrows;
      const int row = rvalid ? s_row[wave][lr] : 0;
      const float sdst = a.alpha * s_sd[wave][rvalid ? lr : 0];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float y[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) y[c] = sdst * acc[q][r][c];
        if (NEPI >= 1) {
          float f[VEC];
          V::unpack(x0v[q][r], f);
#pragma unroll
          for (int c = 0; c < VEC; ++c) y[c] = fmaf(a.beta, f[c], y[c]);
        }
        if (NEPI >= 2) {
          float f[VEC];
          V::unpack(x1v[q][r], f);
#pragma unroll
          for (int c = 0; c < VEC; ++c) y[c] = fmaf(a.gamma, f[c], y[c]);
        }
        if (rvalid && vok[r]) *(raw_t*)(Y + (int64_t)row * a.ldy + voff[r]) = V::pack(y);
      }
    }
  }
}

struct Tuning {
  int chunk_rows = 0;             // 0 = automatic
  int flags = kFlagXcdMap;        // kFlag* bits
  int unroll = 0;                 // 0 = default per shape
  int slab = 0;                   // channels per column slab; 0 = whole rows
  int tiled_min_row_bytes = 1024; // shared-gather kernel: see launch_typed_one; 0 = never (and build no mini-tiles)
};
Tuning g_tuning;

// ---------------------------------------------------------------------------------------------
// LDS-tile variant (north_star: "neighbour feature tiles staged in LDS") -- EXPERIMENTAL, opt-in (SG_TUNE_FLAGS bit 7 when
// the graph is created and when it is applied), measured SLOWER than spmm_rows: one workgroup (2 wavefronts) owns a tile
// of kLdsRows = 16 consecutive rows.  The tile's DISTINCT source rows (~36 for 16 rows in Morton order, 96 edges) are
// brought into LDS ONCE with global_load_lds_dwordx4 (LDS-DMA: no VGPR round trip), then every output row sums its
// neighbours out of LDS in CSR order with the same fma chain as spmm_rows -- bit-identical results, and the L2 -> L1
// request stream drops from 6 rows per output row to ~2.2.  ~25 KB of LDS per workgroup at 512-B rows -> 6 tiles in
// flight per CU.  Measured on the Morton-ordered 1 M-vertex mesh: bf16 C=256 + epilogue 0.56 ms
// against 0.36 ms, fp32 C=256 1.48 against 0.70 -- every tile pays its metadata chain, the DMA round trip and the
// reduction one after the other, and 12 wavefronts per CU do not cover that; kept as the measured answer to "why not LDS
// tiles", and as a working LDS-DMA gather to build a pipelined version on.
// ---------------------------------------------------------------------------------------------
constexpr int kLdsBlock = 128;

template <typename T, int G, int NEPI>
__global__ __launch_bounds__(kLdsBlock) void spmm_lds(const SpmmArgs a, const int nblocks, const int flags) {
  using V = Vt<T>;
  constexpr int VEC = V::VEC;
  constexpr int RPW = 64 / G;                      // rows (and DMA slots) one wavefront instruction covers
  constexpr int ROWB = G * 16;                     // bytes of one feature row
  using raw_t = typename V::raw;
  using elem_t = typename V::elem;

  __shared__ __attribute__((aligned(16))) uint8_t s_data[kLdsSlots * ROWB];
  __shared__ int2 s_u[kLdsSlots];
  __shared__ int32_t s_rp[kLdsRows + 1];
  __shared__ float s_sd[kLdsRows];
  __shared__ int32_t s_row[kLdsRows];
  __shared__ uint8_t s_el[kLdsEdges];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = (flags & kFlagXcdMap) ? xcd_contiguous(blockIdx.x, nblocks) : (int)blockIdx.x;
  const int r0 = t * kLdsRows;
  int nrows = a.n_rows - r0;
  nrows = nrows > kLdsRows ? kLdsRows : nrows;

  // ---- tile metadata: row pointers / scales / row ids, then the distinct-source list and every edge's slot ----
  if (tid <= nrows) s_rp[tid] = a.rowptr[r0 + tid];
  if (tid < nrows) {
    s_sd[tid] = a.scale_dst ? a.scale_dst[r0 + tid] : 1.0f;
    s_row[tid] = a.row_id ? a.row_id[r0 + tid] : r0 + tid;
  }
  const int ub = a.lt_uptr[t];
  const int nu = a.lt_uptr[t + 1] - ub;
  if (tid < nu) s_u[tid] = a.lt_uniq_w[ub + tid];
  __syncthreads();
  const int e0 = s_rp[0], ne = s_rp[nrows] - e0;
  // a tile without a source list although it has edges did not fit the LDS budget (Morton seam, hub vertex, duplicate
  // edges): its rows gather from global memory, edge by edge, in the same order
  const bool in_lds = nu > 0;
  if (in_lds)
    for (int k = tid; k < ne; k += kLdsBlock) s_el[k] = a.lt_eloc[e0 + k];

  // ---- LDS-DMA of the source rows: one wavefront instruction = RPW slots of ROWB bytes, 16 B per lane ----
  const elem_t* __restrict__ X = (const elem_t*)a.X;
  const int g = lane / G, gl = lane % G;
  const int n_inst = (nu + RPW - 1) / RPW;
  for (int i = wave; i < n_inst; i += kLdsBlock / 64) {
    int slot = i * RPW + g;
    slot = slot < nu ? slot : nu - 1;               // the tail re-reads the last source row (its LDS slot is unused)
    const elem_t* src = X + (int64_t)s_u[slot].x * a.ldx + gl * VEC;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)(s_data + i * (RPW * ROWB)), 16, 0, 0);
  }

  const elem_t* __restrict__ X0 = (const elem_t*)a.X0;
  const elem_t* __restrict__ X1 = (const elem_t*)a.X1;
  elem_t* __restrict__ Y = (elem_t*)a.Y;
  constexpr int ROWS_PER_IT = RPW * (kLdsBlock / 64);
  // the epilogue operands of the first row go out together with the DMA
  raw_t x0v, x1v;
  {
    const int lr = wave * RPW + g;
    const int row = s_row[lr < nrows ? lr : 0];
    if (NEPI >= 1) x0v = *(const raw_t*)(X0 + (int64_t)row * a.ldx0 + gl * VEC);
    if (NEPI >= 2) x1v = *(const raw_t*)(X1 + (int64_t)row * a.ldx1 + gl * VEC);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int it = 0; it * ROWS_PER_IT < nrows; ++it) {
    const int lr = it * ROWS_PER_IT + wave * RPW + g;
    const bool rvalid = lr < nrows;
    const int lrc = rvalid ? lr : 0;
    const int row = s_row[lrc];
    const int ks = s_rp[lrc] - e0;
    const int ke = rvalid ? s_rp[lrc + 1] - e0 : ks;
    raw_t nx0, nx1;                                 // next row's epilogue operands, in flight under this row's sums
    {
      const int nl = lr + ROWS_PER_IT;
      const int nrow = s_row[nl < nrows ? nl : 0];
      if (NEPI >= 1) nx0 = *(const raw_t*)(X0 + (int64_t)nrow * a.ldx0 + gl * VEC);
      if (NEPI >= 2) nx1 = *(const raw_t*)(X1 + (int64_t)nrow * a.ldx1 + gl * VEC);
    }
    float acc[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[c] = 0.f;
    if (in_lds) {
      for (int k = ks; __any(k < ke); k += 4) {     // four neighbours per step out of LDS; dead slots carry weight 0
        raw_t xv[4];
        float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kc = k + u < ke ? k + u : (ke > ks ? ke - 1 : 0);
          const int slot = ke > ks ? s_el[kc] : 0;
          w[u] = (k + u < ke) ? __int_as_float(s_u[slot].y) : 0.f;
          xv[u] = *(const raw_t*)(s_data + slot * ROWB + gl * 16);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float f[VEC];
          V::unpack(xv[u], f);
          axpy<VEC>(w[u], f, acc);
        }
      }
    } else {
      for (int k = ks; k < ke; ++k) {               // rare fallback: straight from global memory
        const int2 e = a.lt_idx_w[e0 + k];
        float f[VEC];
        V::unpack(*(const raw_t*)(X + (int64_t)e.x * a.ldx + gl * VEC), f);
        axpy<VEC>(__int_as_float(e.y), f, acc);
      }
    }
    const float sdst = a.alpha * s_sd[lrc];
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
    if (rvalid) *(raw_t*)(Y + (int64_t)row * a.ldy + gl * VEC) = V::pack(y);
    if (NEPI >= 1) x0v = nx0;
    if (NEPI >= 2) x1v = nx1;
  }
}

template <typename T, int G>
int launch_lds(const SpmmArgs& a, hipStream_t stream) {
  const int64_t nt = ((int64_t)a.n_rows + kLdsRows - 1) / kLdsRows;
  const int nblocks = (int)nt;
  SpmmArgs b = a;
  if (!a.X0 && a.X1) { b.X0 = a.X1; b.ldx0 = a.ldx1; b.beta = a.gamma; b.X1 = nullptr; b.ldx1 = 0; b.gamma = 0.f; }
  if (b.X0 && b.X1) spmm_lds<T, G, 2><<<nblocks, kLdsBlock, 0, stream>>>(b, nblocks, g_tuning.flags);
  else if (b.X0) spmm_lds<T, G, 1><<<nblocks, kLdsBlock, 0, stream>>>(b, nblocks, g_tuning.flags);
  else spmm_lds<T, G, 0><<<nblocks, kLdsBlock, 0, stream>>>(b, nblocks, g_tuning.flags);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

// ---------------------------------------------------------------------------------------------
// spmm_ring: the LDS-tile aggregation as a SOFTWARE PIPELINE with the reduction on the matrix cores (bf16 rows of 128 /
// 256 channels on graphs that carry tile records; SG_TUNE_FLAGS bit 11 switches it off at graph creation or at launch).
//
// spmm_lds above pays, per tile and strictly one after the other, a metadata chain, the LDS-DMA round trip and the
// reduction.  Here ONE persistent workgroup per CU (256 channels: 8 consumer + 4 producer wavefronts; 128 channels: 4 + 4,
// two workgroups per CU) walks a stream of tiles (<= 16 rows, <= 56 distinct sources: csr_build.hip::build_ring_records)
// through a ring of D = 3 stages in LDS.  Iteration i, behind ONE s_barrier:
//
//   producers   tile i+D-1:  LDS-DMA of its distinct source rows and of its rows of the epilogue operands (the addresses
//                            were read from its record one iteration earlier); its 16 x 64 weight matrix, split into three
//                            bf16 pieces, scattered into LDS
//               tile i+2D-1: LDS-DMA of its 896-byte record (sources, slot of every edge, scales, row ids)
//   consumers   tile i:      Y_tile = W_tile (16 x nu) * X_sources (nu x C) on v_mfma_f32_16x16x{32,16}_bf16, epilogue, into LDS
//               tile i-1:    leaves LDS as full rows (16-byte stores)
//
// Memory parallelism comes from the ring depth, not from occupancy: D - 1 tiles (~22 + 8 KB each at 256 channels) are in
// flight per CU under the tile being reduced, which is what the latency-bound spmm_rows cannot reach with registers.
// Why MFMA for an HBM-bound kernel: with the sources in LDS the VALU reduction of spmm_rows (8 unpack + 4 packed-fma
// instructions per 16 bytes and neighbour, plus slot / weight addressing) became the bound (measured: 0.46 ms with, 0.21 ms
// without the reduction at C = 256).  The tile's weights are a dense 16 x 56 matrix with ~6 nonzeros per row; multiplying
// it costs 12-18 MFMA instructions per wavefront and NO unpacking: the bf16 source rows are MFMA operands as they lie in
// LDS (ds_read_b64_tr_b16 transposes them on the way).  The fp32 weights enter as hi + mid + lo bf16 pieces (3 x 8 mantissa
// bits: exact), products are exact in fp32, the accumulation is the matrix core's -- so the result differs from
// spmm_rows' sequential fma chain in the last bits (same error bound; deterministic; tested against it to one ulp of the
// output type and against the float64 oracle).
//
// EVERY global read is an LDS-DMA (no VGPR result), so hipcc has no load to put `s_waitcnt vmcnt(0)` in front of; only the
// producers wait on the vector-memory queue, at the top of an iteration, for everything but the DMA instructions they
// issued in the last D - 2 iterations -- a runtime count (the number of source rows varies), served by a jump table of
// s_waitcnt immediates; the tail of the stream re-loads its last tile instead of issuing nothing.  Source rows and operand
// rows are stored XOR-swizzled (applied to the per-lane SOURCE address of the DMA) so that the transposing reads, the
// epilogue reads and the row-wise read-back are bank-conflict-free.  Tiles that do not fit (record.nu == 0) are gathered
// from global memory inside the loop with the fma chain of spmm_rows.
// ---------------------------------------------------------------------------------------------
constexpr int kRingARow = 144;                       // bytes of one row of a weight piece: 64 bf16 + pad (conflict-free b128 reads)
constexpr int kRingAPiece = kLdsRows * kRingARow;

typedef __attribute__((ext_vector_type(8))) __bf16 ring_bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 ring_bf16x4;
typedef __attribute__((ext_vector_type(4))) short ring_s16x4;

// MODE: how the finished tile leaves.  0: straight from the MFMA layout (16 rows x 64 bytes per store instruction: half a cache
// line per row and request).  2: through one of two LDS buffers as FULL ROWS; the stores are issued behind the NEXT tile's
// barrier (no barrier of their own).  Measured at 256 channels: 2 is 1.5 - 2.5 % faster with <= 1 epilogue operand; with
// two there is no LDS left for it.  (A variant with one more barrier per tile and the tile staged in place of its first
// epilogue operand was level with 2; two instead of three weight pieces bought nothing: the MFMAs are not the bound.)
constexpr int kRingPieces = 3;                       // the fp32 weights enter the MFMA as hi + mid + lo bf16 pieces: exact
template <int ROWB, int NEPI, int D, int MODE>
struct RingLds {
  static constexpr int NP = kRingPieces;
  static constexpr int NREC = 2 * D;
  static constexpr int kSrc = kRingSlots * ROWB;         // source rows of one stage
  static constexpr int kA = NP * kRingAPiece;            // the weight pieces of one stage; directly BEHIND the source rows:
                                                         // the last MFMA step reads 64 - kRingSlots "rows" of it (finite, weight 0)
  static constexpr int kEpi = kLdsRows * ROWB;           // one epilogue operand of one stage
  static constexpr int kStage = kSrc + kA + NEPI * kEpi;
  static constexpr int oA = kSrc;                        // inside a stage
  static constexpr int oEpi = kSrc + kA;
  static constexpr int oRec = D * kStage;                // [NREC][kRecBytes]
  static constexpr int oStg = oRec + NREC * kRecBytes;   // output tiles on their way to full-row stores
  static constexpr int total = oStg + (MODE == 2 ? 2 : 0) * kLdsRows * ROWB;
  static_assert((64 - kRingSlots) * ROWB <= kA, "the rows the last MFMA step reads behind the source slots lie in the weight pieces");
};

__device__ __forceinline__ void ring_dma16(const void* src, uint8_t* lds) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                   (void __attribute__((address_space(3)))*)lds, 16, 0, 0);
}
__device__ __forceinline__ int ring_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
template <int N> __device__ __forceinline__ void ring_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// s_waitcnt takes an immediate: the allowed number of outstanding DMA instructions is a (wave-uniform) runtime value
__device__ __forceinline__ void ring_wait_vm_n(int n) {
  switch (n) {
#define SG_RING_CASE(k) case k: ring_wait_vm<k>(); break;
    SG_RING_CASE(1) SG_RING_CASE(2) SG_RING_CASE(3) SG_RING_CASE(4) SG_RING_CASE(5) SG_RING_CASE(6) SG_RING_CASE(7) SG_RING_CASE(8)
    SG_RING_CASE(9) SG_RING_CASE(10) SG_RING_CASE(11) SG_RING_CASE(12) SG_RING_CASE(13) SG_RING_CASE(14) SG_RING_CASE(15) SG_RING_CASE(16)
    SG_RING_CASE(17) SG_RING_CASE(18) SG_RING_CASE(19) SG_RING_CASE(20) SG_RING_CASE(21) SG_RING_CASE(22) SG_RING_CASE(23) SG_RING_CASE(24)
    SG_RING_CASE(25) SG_RING_CASE(26) SG_RING_CASE(27) SG_RING_CASE(28) SG_RING_CASE(29) SG_RING_CASE(30) SG_RING_CASE(31) SG_RING_CASE(32)
#undef SG_RING_CASE
    default: ring_wait_vm<0>(); break;
  }
}
#define SG_RING_BARRIER()                                   \
  do {                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
    __builtin_amdgcn_s_barrier();                           \
    asm volatile("" ::: "memory");                          \
  } while (0)

template <int NBW, int NEPI, int D, int MODE>
__global__ __launch_bounds__(128 * NBW + 256) void spmm_ring(const SpmmArgs a, const int nt) {
  using V = Vt<bf16_tag>;
  constexpr int VEC = 8;
  constexpr int CW = 2 * NBW;                       // consumer wavefronts: each owns two 16-channel blocks
  constexpr int PW = 4;                             // producer wavefronts: 256 threads = one per (row, neighbour) of a tile
  constexpr int C = 64 * NBW;
  constexpr int ROWB = 2 * C;                       // bytes of one feature row
  constexpr int G = ROWB / 16;                      // lanes per row in a DMA instruction
  constexpr int RPW = 64 / G;                       // rows one DMA instruction covers
  constexpr int SRC_MAX = (kRingSlots / RPW + PW - 1) / PW;        // source-row DMA instructions per producer, at most
  constexpr int EPI_PER_WAVE = kLdsRows / RPW / PW;
  constexpr int FIXED = NEPI * EPI_PER_WAVE + 1;    // operand rows + record share: DMA instructions per producer and tile
  static_assert(NBW == 2 || NBW == 4, "128 or 256 channels");
  static_assert(EPI_PER_WAVE * RPW * PW == kLdsRows && CW * RPW == kLdsRows, "equal shares");
  static_assert((D - 2) * (SRC_MAX + FIXED) <= 32 && D >= 2 && D <= 5, "vmcnt switch range");
  constexpr int NP = kRingPieces;
  static_assert(MODE == 0 || MODE == 2, "store modes");
  using L = RingLds<ROWB, NEPI, D, MODE>;
  using raw_t = typename V::raw;
  using elem_t = typename V::elem;

  __shared__ __attribute__((aligned(16))) uint8_t smem[L::total];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // this workgroup's tiles: the workgroups of one XCD (block ids b, b + 8, ..) share ONE contiguous run of tiles and take
  // them round-robin, so the rows in flight on an XCD are one thin front (its L2 serves the re-gathered neighbours)
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int nx = ((int)gridDim.x - xcd + 7) >> 3;
  const int q = nt >> 3, rem = nt & 7;
  const int tile_lo = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
  const int tile_cnt = q + (xcd < rem ? 1 : 0);
  const int n_my = jb < tile_cnt ? (tile_cnt - jb + nx - 1) / nx : 0;
  if (n_my == 0) return;
  const int t_first = tile_lo + jb;

  const elem_t* __restrict__ X = (const elem_t*)a.X;

  // the weight pieces start as zeros (only the nonzeros are ever written and taken back); so do the source slots
  for (int o = tid * 16; o < D * L::kStage; o += (CW + PW) * 64 * 16) *(u32x4*)(smem + o) = u32x4{0u, 0u, 0u, 0u};
  SG_RING_BARRIER();

  if (wave >= CW) {
    // =============================== producers: DMA + weight matrices, D - 1 tiles ahead ===============================
    const int pw = wave - CW, ptid = tid - CW * 64;
    __builtin_amdgcn_s_setprio(3);                  // the DMA issue goes ahead of the consumers' arithmetic (2 - 5 % measured)
    const int g = lane / G, gl = lane % G;          // DMA: row of the instruction, 16-byte chunk of the row
    const elem_t* __restrict__ X0 = (const elem_t*)a.X0;
    const elem_t* __restrict__ X1 = (const elem_t*)a.X1;
    auto tile_of = [&](int k) { return t_first + (k < n_my ? k : n_my - 1) * nx; };   // past the end: the last tile again
    auto issue_meta = [&](int tile, int slot) {     // 1 KB record, an equal share per producer
      constexpr int LPW = kRecBytes / 16 / PW;
      if (lane < LPW)
        ring_dma16(a.lt_rec + (int64_t)tile * kRecBytes + (pw * LPW + lane) * 16, smem + L::oRec + slot * kRecBytes + pw * LPW * 16);
    };
    // what the DMA of a tile needs from its record, fetched ONE ITERATION EARLY so that the DMA starts right behind the barrier
    int f_nu, f_src[SRC_MAX], f_erow[EPI_PER_WAVE];
    auto fetch = [&](int slot) {
      const uint8_t* rec = smem + L::oRec + slot * kRecBytes;
      f_nu = *(const int32_t*)(rec + kRecNu);
#pragma unroll
      for (int j = 0; j < SRC_MAX; ++j) {
        const int sl = (pw + PW * j) * RPW + g;
        f_src[j] = ((const int32_t*)(rec + kRecSrc))[sl < kRingSlots ? sl : kRingSlots - 1];   // padded with the last source
      }
#pragma unroll
      for (int j = 0; j < EPI_PER_WAVE; ++j) f_erow[j] = ((const int32_t*)(rec + kRecRow))[(pw + PW * j) * RPW + g];
    };
    uint32_t a_slots = 0xffffffffu;                 // 6 bits per stage: the slot this thread's weight went to (63: none)
    // everything the reduction of the tile in record `slot` (fetched) reads -> stage st; returns the DMA instructions issued
    auto issue_tile = [&](int slot, int st) -> int {
      const uint8_t* rec = smem + L::oRec + slot * kRecBytes;
      uint8_t* stage = smem + st * L::kStage;
      const int nu = __builtin_amdgcn_readfirstlane(f_nu);
      const int n_inst = (nu + RPW - 1) / RPW;
      int issued = 0;
#pragma unroll
      for (int j = 0; j < SRC_MAX; ++j) {
        const int i = pw + PW * j;
        if (i < n_inst) {                           // wave-uniform
          const int sl = i * RPW + g;
          const int chunk = ((((gl >> 1) ^ ring_swz(sl)) << 1) | (gl & 1));   // 32-byte segments XOR-swizzled by the slot
          ring_dma16(X + (int64_t)f_src[j] * a.ldx + chunk * VEC, stage + i * (RPW * ROWB));
          ++issued;
        }
      }
      if (NEPI >= 1) {
#pragma unroll
        for (int j = 0; j < EPI_PER_WAVE; ++j) {
          const int i = pw + PW * j;
          const int lr = i * RPW + g;
          const int chunk = gl ^ (lr & 15);                                 // 16-byte chunks XOR-swizzled by the row
          ring_dma16(X0 + (int64_t)f_erow[j] * a.ldx0 + chunk * VEC, stage + L::oEpi + i * (RPW * ROWB));
          if (NEPI >= 2) ring_dma16(X1 + (int64_t)f_erow[j] * a.ldx1 + chunk * VEC, stage + L::oEpi + L::kEpi + i * (RPW * ROWB));
        }
        issued += NEPI * EPI_PER_WAVE;
      }
      // the slots of the last MFMA step behind the list: zeros (stale rows of an earlier tile must not meet a zero weight
      // as Inf / NaN far from where they came from)
      const int kend = nu > 48 ? kRingSlots : nu > 32 ? 48 : nu > 0 ? 32 : 0;
      for (int o = ((nu + RPW - 1) / RPW) * RPW * ROWB + ptid * 16; o < kend * ROWB; o += PW * 64 * 16) *(u32x4*)(stage + o) = u32x4{0u, 0u, 0u, 0u};
      // weight matrix: thread (row, u) owns the u-th neighbour of the row; fp32 weight = hi + mid (+ lo) in bf16
      const int ar = ptid >> 4, au = ptid & 15;
      const int deg = rec[kRecDeg + ar];
      const int sl_new = rec[kRecSlot + ptid] & 63;
      const float w = ((const float*)(rec + kRecW))[sl_new < kRingSlots ? sl_new : 0];
      uint8_t* Ab = stage + L::oA;
      const int old = (a_slots >> (6 * st)) & 63;
      if (old != 63) {
#pragma unroll
        for (int p = 0; p < NP; ++p) *(uint16_t*)(Ab + p * kRingAPiece + ar * kRingARow + old * 2) = 0;
      }
      int sl = 63;
      if (au < deg) {
        sl = sl_new;
        const uint32_t hi = __float_as_uint(w) & 0xffff0000u;
        const float r1 = w - __uint_as_float(hi);
        const uint32_t mid = __float_as_uint(r1) & 0xffff0000u;
        const int off = ar * kRingARow + sl * 2;
        *(uint16_t*)(Ab + 0 * kRingAPiece + off) = (uint16_t)(hi >> 16);
        *(uint16_t*)(Ab + 1 * kRingAPiece + off) = (uint16_t)(mid >> 16);
        const float r2 = r1 - __uint_as_float(mid);
        *(uint16_t*)(Ab + 2 * kRingAPiece + off) = (uint16_t)(__float_as_uint(r2) >> 16);
      }
      a_slots = (a_slots & ~(63u << (6 * st))) | ((uint32_t)sl << (6 * st));
      return issued;
    };

    // prologue: the records of the first 2D - 1 tiles, then the rows of the first D - 1
    for (int k = 0; k < 2 * D - 1; ++k) issue_meta(tile_of(k), k);
    ring_wait_vm<0>();
    SG_RING_BARRIER();                              // (P0) the consumers wait at it too
    int inflight[D];                                // DMA instructions of the iterations i - 1, i - 2, ..
#pragma unroll
    for (int k = 0; k < D; ++k) inflight[k] = 0;
    for (int k = 0; k < D - 1; ++k) {
      fetch(k);
      (void)issue_tile(k, k);
    }
    fetch(D - 1);
    asm volatile("" ::: "memory");
    int st = 0, rs = 0;                             // stage and record slot of tile i
    for (int i = 0; i < n_my; ++i) {
      // the rows, operands and weights of tile i and the record of tile i + D have landed (this wavefront's share):
      // everything issued up to iteration i - D + 1; the DMA instructions of the D - 2 iterations after it may fly
      if (i == 0) {
        ring_wait_vm<0>();
      } else {
        int allowed = 0;
#pragma unroll
        for (int k = 0; k < D - 2; ++k) allowed += inflight[k];
        ring_wait_vm_n(allowed);
      }
      SG_RING_BARRIER();                            // tile i is ready for the consumers; they are done with tile i - 1
      const int st_n = st == 0 ? D - 1 : st - 1;                  // (i + D - 1) % D
      const int rs_n = rs + D - 1 >= L::NREC ? rs + D - 1 - L::NREC : rs + D - 1;
      const int rs_f = rs + D >= L::NREC ? rs + D - L::NREC : rs + D;
      const int rs_m = rs == 0 ? L::NREC - 1 : rs - 1;            // (i + 2D - 1) % 2D
      const int cnt = issue_tile(rs_n, st_n) + 1;
      issue_meta(tile_of(i + 2 * D - 1), rs_m);
      fetch(rs_f);
#pragma unroll
      for (int k = D - 2; k > 0; --k) inflight[k] = inflight[k - 1];       // [0] = the youngest iteration
      inflight[0] = cnt;
      asm volatile("" ::: "memory");
      st = st + 1 == D ? 0 : st + 1;
      rs = rs + 1 == L::NREC ? 0 : rs + 1;
    }
    if (MODE == 2) SG_RING_BARRIER();               // the last output tile is complete in LDS
    ring_wait_vm<0>();                              // no LDS-DMA may be in flight when the workgroup ends
    return;
  }

  // ================================= consumers: MFMA reduction, epilogue, store =================================
  elem_t* __restrict__ Y = (elem_t*)a.Y;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)smem;
  const int m = lane & 15, fg = lane >> 4;          // MFMA: tile row (operand / result column), k group
  const int fq = (lane >> 2) & 3, fp = lane & 3;    // transposing reads: row and 8-byte column quad a lane SUPPLIES
  const int col = wave * 32 + (fg & 1) * 16 + (fg >> 1) * 8;     // this lane's 8 consecutive channels of row m
  const int olr = wave * RPW + lane / G, ogl = lane % G;          // full-row stores: row of the tile, 16-byte chunk
  SG_RING_BARRIER();                                // (P0): the first records are there
  // the record fields of a tile are read one iteration early
  int c_nu, c_nrows, c_row, c_orow;
  float c_sd;
  auto fetch = [&](int slot) {
    const uint8_t* rec = smem + L::oRec + slot * kRecBytes;
    c_nu = *(const int32_t*)(rec + kRecNu);
    c_nrows = *(const int32_t*)(rec + kRecNrows);
    c_sd = ((const float*)(rec + kRecSd))[m];
    c_row = ((const int32_t*)(rec + kRecRow))[m];
    c_orow = ((const int32_t*)(rec + kRecRow))[olr];
  };
  fetch(0);
  int st = 0, rs = 0;
  int p_nrows = 0, p_orow = 0;                      // MODE 2: the tile whose stores are still to be issued
  for (int i = 0; i < n_my; ++i) {
    SG_RING_BARRIER();
    const uint8_t* rec = smem + L::oRec + rs * kRecBytes;
    uint8_t* stage = smem + st * L::kStage;
    const int nu = __builtin_amdgcn_readfirstlane(c_nu);
    const int nrows = __builtin_amdgcn_readfirstlane(c_nrows);
    const float sd = c_sd;
    const int row = c_row, orow = c_orow;
    // ---- every LDS read of this iteration is issued here, in one batch ----
    raw_t out_prev = raw_t{0u, 0u, 0u, 0u};
    if (MODE == 2) out_prev = *(const raw_t*)(smem + L::oStg + ((i + 1) & 1) * (kLdsRows * ROWB) + olr * ROWB + ((ogl ^ (olr & 15)) << 4));
    const uint8_t* Ab = stage + L::oA;
    ring_bf16x8 wf0[NP];
    ring_s16x4 wf1[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      wf0[p] = *(const ring_bf16x8*)(Ab + p * kRingAPiece + m * kRingARow + fg * 16);
      wf1[p] = *(const ring_s16x4*)(Ab + p * kRingAPiece + m * kRingARow + 64 + fg * 8);
    }
    raw_t x0v = raw_t{0u, 0u, 0u, 0u}, x1v = raw_t{0u, 0u, 0u, 0u};
    if (NEPI >= 1) x0v = *(const raw_t*)(stage + L::oEpi + m * ROWB + (((col >> 3) ^ m) << 4));
    if (NEPI >= 2) x1v = *(const raw_t*)(stage + L::oEpi + L::kEpi + m * ROWB + (((col >> 3) ^ m) << 4));
    fetch(rs + 1 == L::NREC ? 0 : rs + 1);          // (past the end: a copy of the last record)
    const uint32_t xs = lds0 + st * L::kStage;
    ring_bf16x4 h0[2][2];
    ring_s16x4 h1[2];
    {
      const int sw = fq | ((fg & 1) << 2);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const uint32_t addr = xs + (uint32_t)((8 * fg + 4 * hh + fq) * ROWB + (((wave * 2 + nb) ^ sw) << 5) + fp * 8);
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(h0[nb][hh]) : "v"(addr) : "memory");
        }
      const int sw1 = fq | (((fg >> 1) & 1) << 2);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const uint32_t addr = xs + (uint32_t)((32 + 4 * fg + fq) * ROWB + (((wave * 2 + nb) ^ sw1) << 5) + fp * 8);
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(h1[nb]) : "v"(addr) : "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0[0][0]), "+v"(h0[0][1]), "+v"(h0[1][0]), "+v"(h0[1][1]), "+v"(h1[0]), "+v"(h1[1])::"memory");
    if (MODE == 2) {                                // the previous tile leaves: full rows, 16 bytes per lane
      if (olr < p_nrows) *(raw_t*)(Y + (int64_t)p_orow * a.ldy + ogl * VEC) = out_prev;
      p_nrows = nrows;
      p_orow = orow;
    }
    float v[8];
    if (nu > 0) {
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int p = 0; p < NP; ++p)                  // slots 0 .. 31
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const ring_bf16x8 xf = {h0[nb][0][0], h0[nb][0][1], h0[nb][0][2], h0[nb][0][3], h0[nb][1][0], h0[nb][1][1], h0[nb][1][2], h0[nb][1][3]};
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, wf0[p], acc[nb], 0, 0, 0);
        }
      if (nu > 32) {                                // slots 32 .. 47
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(h1[nb], wf1[p], acc[nb], 0, 0, 0);
      }
      if (nu > 48) {   // slots 48 .. 63 (the rows behind slot kRingSlots - 1 are weight pieces: finite, weight 0)
        ring_s16x4 wf[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) wf[p] = *(const ring_s16x4*)(Ab + p * kRingAPiece + m * kRingARow + 96 + fg * 8);
        ring_s16x4 h[2];
        const int sw = fq | (((fg >> 1) & 1) << 2);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const uint32_t addr = xs + (uint32_t)((48 + 4 * fg + fq) * ROWB + (((wave * 2 + nb) ^ sw) << 5) + fp * 8);
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(h[nb]) : "v"(addr) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h[0]), "+v"(h[1])::"memory");
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(h[nb], wf[p], acc[nb], 0, 0, 0);
      }
      // a lane holds 4 channels of row m per block; one exchange per value with the neighbouring 16-lane row -> 8
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[0][jj]), __float_as_uint(acc[1][jj]), false, false);
        v[jj] = __uint_as_float(r[0]);
        v[4 + jj] = __uint_as_float(r[1]);
      }
    } else {          // rare: the tile's sources do not fit; its rows are gathered from global memory (sequential fma chain)
      const int r0 = *(const int32_t*)(rec + kRecR0);
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = 0.f;
      if (m < nrows) {
        const int gs = a.rowptr[r0 + m], ge = a.rowptr[r0 + m + 1];
        for (int k = gs; k < ge; ++k) {
          const int2 e = a.lt_idx_w[k];
          float f[VEC];
          V::unpack(*(const raw_t*)(X + (int64_t)e.x * a.ldx + col), f);
          axpy<VEC>(__int_as_float(e.y), f, v);
        }
      }
    }
    const float sdst = a.alpha * sd;
    float y[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) y[c] = sdst * v[c];
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
    if (MODE == 0) {
      if (m < nrows) *(raw_t*)(Y + (int64_t)row * a.ldy + col) = V::pack(y);
    } else {
      *(raw_t*)(smem + L::oStg + (i & 1) * (kLdsRows * ROWB) + m * ROWB + (((col >> 3) ^ m) << 4)) = V::pack(y);
    }
    st = st + 1 == D ? 0 : st + 1;
    rs = rs + 1 == L::NREC ? 0 : rs + 1;
  }
  if (MODE == 2) {
    SG_RING_BARRIER();
    const raw_t out = *(const raw_t*)(smem + L::oStg + ((n_my + 1) & 1) * (kLdsRows * ROWB) + olr * ROWB + ((ogl ^ (olr & 15)) << 4));
    if (olr < p_nrows) *(raw_t*)(Y + (int64_t)p_orow * a.ldy + ogl * VEC) = out;
  }
}

template <int NBW, int NEPI, int D, int MODE>
int launch_ring_k(const SpmmArgs& b, hipStream_t stream) {
  constexpr int lds = RingLds<128 * NBW, NEPI, D, MODE>::total;
  static_assert(lds <= 160 * 1024, "LDS budget");
  int per_cu = (160 * 1024) / lds;
  per_cu = per_cu > 4 ? 4 : per_cu;
  int64_t nb = (int64_t)per_cu * 256;
  if (nb > b.lt_nrec) nb = b.lt_nrec;
  nb = (nb + 7) / 8 * 8;
  spmm_ring<NBW, NEPI, D, MODE><<<(int)nb, 128 * NBW + 256, 0, stream>>>(b, b.lt_nrec);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

// Ring depth and store mode per shape (tools/agg_bench.py on the 1 M-vertex mesh; LDS decides most of it):
//   256 channels: one workgroup (8 + 4 wavefronts) per CU, 3 stages of 35 / 43 / 51 KB; full-row stores while LDS lasts
//   128 channels: two workgroups (4 + 4 wavefronts) per CU: 3 stages (2 with two epilogue operands), 76 / 80 / 61 KB
template <int NBW>
int launch_ring(const SpmmArgs& a, hipStream_t stream) {
  SpmmArgs b = a;
  if (!a.X0 && a.X1) { b.X0 = a.X1; b.ldx0 = a.ldx1; b.beta = a.gamma; b.X1 = nullptr; b.ldx1 = 0; b.gamma = 0.f; }
  const int nepi = (b.X0 ? 1 : 0) + (b.X1 ? 1 : 0);
  const bool direct = (g_tuning.flags & kFlagRingDirectStores) != 0;       // A/B
  if constexpr (NBW == 4) {
    if (nepi == 2) return launch_ring_k<NBW, 2, 3, 0>(b, stream);
    if (nepi == 1) return direct ? launch_ring_k<NBW, 1, 3, 0>(b, stream) : launch_ring_k<NBW, 1, 3, 2>(b, stream);
    return direct ? launch_ring_k<NBW, 0, 3, 0>(b, stream) : launch_ring_k<NBW, 0, 3, 2>(b, stream);
  } else {
    if (nepi == 2) return launch_ring_k<NBW, 2, 2, 0>(b, stream);
    if (nepi == 1) return launch_ring_k<NBW, 1, 3, 0>(b, stream);
    return direct ? launch_ring_k<NBW, 0, 3, 0>(b, stream) : launch_ring_k<NBW, 0, 3, 2>(b, stream);
  }
}

// ---------------------------------------------------------------------------------------------
// spmm_ring_f32: the same pipeline for FLOAT32 rows of 128 / 256 channels -- the reference's own precision
// (util/networks.py:40-53 via [3P] ChebConv.propagate; BASELINE configs c2 / c3), where spmm_rows stood at 0.55 of the HBM peak.
// Producers, records, ring and barriers are those of spmm_ring; what changes is the reduction:
//   Y_tile^T (16 channels x 16 rows) += X^T (16 channels x 4 slots) * W^T (4 slots x 16 rows) on v_mfma_f32_16x16x4_f32 --
//   float32 operands, float32 products and sums, no pieces: a lane supplies ONE source value per step (a plain ds_read_b32
//   of the row as it lies in LDS, no transposing read) and one weight, and ends up with 4 consecutive channels of its row:
//   the epilogue operands come in and the result goes out as float4.  The nonzero terms of a row are added in ascending
//   source order, as spmm_rows does; the matrix core's 4-term inner sum rounds differently: equal to a few ulp, tested.
//   * source rows are 512 B / 1 KB: the 64-byte segments of a row are XOR-swizzled by (slot & 3) -- the four slots of a
//     step then sit in four different bank groups (ds_read_b32: conflict-free);
//   * the weights of a tile are ONE float32 matrix [16 rows][4 k-lanes][16 steps] (row pitch 272 B: conflict-free b128 reads),
//     4.25 KB per stage instead of three bf16 pieces;
//   * the kernel takes 128 channels (512-byte rows, D = 3 stages of 32 - 48 KB); rows of 256 channels run as their two halves
//     (launch_ring_f32 below: a 1 KB row leaves room for two stages only);
//   * finished tiles leave straight from the MFMA layout (16 rows x 64 B per store, two stores per 128-byte line back to back).
// ---------------------------------------------------------------------------------------------
constexpr int kRingFARow = 272;                      // bytes of one row of the float32 weight matrix: 4 x 16 floats + 16
template <int ROWB, int NEPI, int D>
struct RingLdsF {
  static constexpr int NREC = 2 * D;
  static constexpr int kSrc = kRingSlots * ROWB;
  static constexpr int kA = kLdsRows * kRingFARow;
  static constexpr int kEpi = kLdsRows * ROWB;
  static constexpr int kStage = kSrc + kA + NEPI * kEpi;
  static constexpr int oA = kSrc;
  static constexpr int oEpi = kSrc + kA;
  static constexpr int oRec = D * kStage;
  static constexpr int total = oRec + NREC * kRecBytes;
};

template <int NBW, int NEPI, int D, int BPW>
__global__ __launch_bounds__(64 * (4 * NBW / BPW) + 256) void spmm_ring_f32(const SpmmArgs a, const int nt) {
  constexpr int CW = 4 * NBW / BPW;                 // consumer wavefronts: each owns BPW 16-channel blocks
  constexpr int PW = 4;
  constexpr int C = 64 * NBW;
  constexpr int ROWB = 4 * C;
  constexpr int G = ROWB / 16;
  constexpr int RPW = 64 / G;
  constexpr int SRC_MAX = (kRingSlots / RPW + PW - 1) / PW;
  constexpr int EPI_PER_WAVE = kLdsRows / RPW / PW;
  constexpr int FIXED = NEPI * EPI_PER_WAVE + 1;
  static_assert(NBW == 2 || NBW == 4, "128 or 256 channels");
  static_assert(EPI_PER_WAVE * RPW * PW == kLdsRows, "equal shares");
  static_assert((D - 2) * (SRC_MAX + FIXED) <= 32 && D >= 2 && D <= 5, "vmcnt switch range");
  using L = RingLdsF<ROWB, NEPI, D>;
  static_assert(L::total <= 160 * 1024, "LDS budget");

  __shared__ __attribute__((aligned(16))) uint8_t smem[L::total];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int nx = ((int)gridDim.x - xcd + 7) >> 3;
  const int q = nt >> 3, rem = nt & 7;
  const int tile_lo = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
  const int tile_cnt = q + (xcd < rem ? 1 : 0);
  const int n_my = jb < tile_cnt ? (tile_cnt - jb + nx - 1) / nx : 0;
  if (n_my == 0) return;
  const int t_first = tile_lo + jb;

  const float* __restrict__ X = (const float*)a.X;

  // weights and source slots start as zeros (only the nonzero weights are ever written and taken back)
  for (int o = tid * 16; o < D * L::kStage; o += (CW + PW) * 64 * 16) *(u32x4*)(smem + o) = u32x4{0u, 0u, 0u, 0u};
  SG_RING_BARRIER();

  if (wave >= CW) {
    // =============================== producers: DMA + weight matrices, D - 1 tiles ahead ===============================
    const int pw = wave - CW, ptid = tid - CW * 64;
    __builtin_amdgcn_s_setprio(3);
    const int g = lane / G, gl = lane % G;
    const float* __restrict__ X0 = (const float*)a.X0;
    const float* __restrict__ X1 = (const float*)a.X1;
    auto tile_of = [&](int k) { return t_first + (k < n_my ? k : n_my - 1) * nx; };
    auto issue_meta = [&](int tile, int slot) {
      constexpr int LPW = kRecBytes / 16 / PW;
      if (lane < LPW)
        ring_dma16(a.lt_rec + (int64_t)tile * kRecBytes + (pw * LPW + lane) * 16, smem + L::oRec + slot * kRecBytes + pw * LPW * 16);
    };
    int f_nu, f_src[SRC_MAX], f_erow[EPI_PER_WAVE];
    auto fetch = [&](int slot) {
      const uint8_t* rec = smem + L::oRec + slot * kRecBytes;
      f_nu = *(const int32_t*)(rec + kRecNu);
#pragma unroll
      for (int j = 0; j < SRC_MAX; ++j) {
        const int sl = (pw + PW * j) * RPW + g;
        f_src[j] = ((const int32_t*)(rec + kRecSrc))[sl < kRingSlots ? sl : kRingSlots - 1];
      }
#pragma unroll
      for (int j = 0; j < EPI_PER_WAVE; ++j) f_erow[j] = ((const int32_t*)(rec + kRecRow))[(pw + PW * j) * RPW + g];
    };
    uint32_t a_slots = 0xffffffffu;                 // 6 bits per stage: the slot this thread's weight went to (63: none)
    auto issue_tile = [&](int slot, int st) -> int {
      const uint8_t* rec = smem + L::oRec + slot * kRecBytes;
      uint8_t* stage = smem + st * L::kStage;
      const int nu = __builtin_amdgcn_readfirstlane(f_nu);
      const int n_inst = (nu + RPW - 1) / RPW;
      int issued = 0;
#pragma unroll
      for (int j = 0; j < SRC_MAX; ++j) {
        const int i = pw + PW * j;
        if (i < n_inst) {                           // wave-uniform
          const int sl = i * RPW + g;
          const int chunk = gl ^ ((sl & 3) << 2);   // 64-byte segments XOR-swizzled by the slot
          ring_dma16(X + (int64_t)f_src[j] * a.ldx + chunk * 4, stage + i * (RPW * ROWB));
          ++issued;
        }
      }
      if (NEPI >= 1) {
#pragma unroll
        for (int j = 0; j < EPI_PER_WAVE; ++j) {
          const int i = pw + PW * j;
          const int lr = i * RPW + g;
          const int chunk = gl ^ (lr & 15);         // 16-byte chunks XOR-swizzled by the row
          ring_dma16(X0 + (int64_t)f_erow[j] * a.ldx0 + chunk * 4, stage + L::oEpi + i * (RPW * ROWB));
          if (NEPI >= 2) ring_dma16(X1 + (int64_t)f_erow[j] * a.ldx1 + chunk * 4, stage + L::oEpi + L::kEpi + i * (RPW * ROWB));
        }
        issued += NEPI * EPI_PER_WAVE;
      }
      // the slots behind the list up to the end of their group of four steps: zeros (a stale row of an earlier tile must not
      // meet a zero weight as Inf / NaN far from where it came from)
      int kend = (nu + 15) & ~15;
      kend = kend > kRingSlots ? kRingSlots : kend;
      for (int o = n_inst * RPW * ROWB + ptid * 16; o < kend * ROWB; o += PW * 64 * 16) *(u32x4*)(stage + o) = u32x4{0u, 0u, 0u, 0u};
      // weight matrix: thread (row, u) owns the u-th neighbour of the row
      const int ar = ptid >> 4, au = ptid & 15;
      const int deg = rec[kRecDeg + ar];
      const int sl_new = rec[kRecSlot + ptid] & 63;
      const float w = ((const float*)(rec + kRecW))[sl_new < kRingSlots ? sl_new : 0];
      uint8_t* Ab = stage + L::oA + ar * kRingFARow;
      const int old = (a_slots >> (6 * st)) & 63;
      if (old != 63) *(float*)(Ab + (old & 3) * 64 + (old >> 2) * 4) = 0.f;
      int sl = 63;
      if (au < deg) {
        sl = sl_new;
        *(float*)(Ab + (sl & 3) * 64 + (sl >> 2) * 4) = w;
      }
      a_slots = (a_slots & ~(63u << (6 * st))) | ((uint32_t)sl << (6 * st));
      return issued;
    };

    for (int k = 0; k < 2 * D - 1; ++k) issue_meta(tile_of(k), k);
    ring_wait_vm<0>();
    SG_RING_BARRIER();                              // (P0)
    int inflight[D];
#pragma unroll
    for (int k = 0; k < D; ++k) inflight[k] = 0;
    for (int k = 0; k < D - 1; ++k) {
      fetch(k);
      (void)issue_tile(k, k);
    }
    fetch(D - 1);
    asm volatile("" ::: "memory");
    int st = 0, rs = 0;
    for (int i = 0; i < n_my; ++i) {
      if (i == 0) {
        ring_wait_vm<0>();
      } else {
        int allowed = 0;
#pragma unroll
        for (int k = 0; k < D - 2; ++k) allowed += inflight[k];
        ring_wait_vm_n(allowed);
      }
      SG_RING_BARRIER();                            // tile i is ready for the consumers; they are done with tile i - 1
      const int st_n = st == 0 ? D - 1 : st - 1;
      const int rs_n = rs + D - 1 >= L::NREC ? rs + D - 1 - L::NREC : rs + D - 1;
      const int rs_f = rs + D >= L::NREC ? rs + D - L::NREC : rs + D;
      const int rs_m = rs == 0 ? L::NREC - 1 : rs - 1;
      const int cnt = issue_tile(rs_n, st_n) + 1;
      issue_meta(tile_of(i + 2 * D - 1), rs_m);
      fetch(rs_f);
#pragma unroll
      for (int k = D - 2; k > 0; --k) inflight[k] = inflight[k - 1];
      if (D > 2) inflight[0] = cnt;
      asm volatile("" ::: "memory");
      st = st + 1 == D ? 0 : st + 1;
      rs = rs + 1 == L::NREC ? 0 : rs + 1;
    }
    ring_wait_vm<0>();                              // no LDS-DMA may be in flight when the workgroup ends
    return;
  }

  // ================================= consumers: MFMA reduction, epilogue, store =================================
  float* __restrict__ Y = (float*)a.Y;
  const int m = lane & 15, fg = lane >> 4;          // MFMA: tile row (B / result column), k lane; the result: channels 4 fg .. + 3
  int xo[BPW], eo[BPW];
#pragma unroll
  for (int nb = 0; nb < BPW; ++nb) {
    const int blk = wave * BPW + nb;
    xo[nb] = fg * ROWB + ((((blk * 4 + (m >> 2)) ^ (fg << 2)) << 4) | ((m & 3) << 2));      // channel blk * 16 + m of slot 4 s + fg
    eo[nb] = m * ROWB + (((blk * 4 + fg) ^ m) << 4);                                        // channels blk * 16 + 4 fg .. of row m
  }
  SG_RING_BARRIER();                                // (P0): the first records are there
  int c_nu, c_nrows, c_row;
  float c_sd;
  auto fetch = [&](int slot) {
    const uint8_t* rec = smem + L::oRec + slot * kRecBytes;
    c_nu = *(const int32_t*)(rec + kRecNu);
    c_nrows = *(const int32_t*)(rec + kRecNrows);
    c_sd = ((const float*)(rec + kRecSd))[m];
    c_row = ((const int32_t*)(rec + kRecRow))[m];
  };
  fetch(0);
  int st = 0, rs = 0;
  for (int i = 0; i < n_my; ++i) {
    SG_RING_BARRIER();
    const uint8_t* rec = smem + L::oRec + rs * kRecBytes;
    const uint8_t* stage = smem + st * L::kStage;
    const int nu = __builtin_amdgcn_readfirstlane(c_nu);
    const int nrows = __builtin_amdgcn_readfirstlane(c_nrows);
    const float sd = c_sd;
    const int row = c_row;
    // ---- the LDS reads of this iteration, in one batch ----
    const uint8_t* Ab = stage + L::oA + m * kRingFARow + fg * 64;
    f32x4 wq[4];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) wq[qq] = *(const f32x4*)(Ab + qq * 16);       // the weights of steps 4 qq .. 4 qq + 3
    f32x4 x0v[BPW], x1v[BPW];
#pragma unroll
    for (int nb = 0; nb < BPW; ++nb) {
      x0v[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      x1v[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (NEPI >= 1) x0v[nb] = *(const f32x4*)(stage + L::oEpi + eo[nb]);
      if (NEPI >= 2) x1v[nb] = *(const f32x4*)(stage + L::oEpi + L::kEpi + eo[nb]);
    }
    fetch(rs + 1 == L::NREC ? 0 : rs + 1);          // (past the end: a copy of the last record)
    float xs[14][BPW];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq)
      if (nu > 16 * qq) {                           // wave-uniform: a group of four steps (the last group: two)
#pragma unroll
        for (int s = 4 * qq; s < 4 * qq + 4 && s < 14; ++s)
#pragma unroll
          for (int nb = 0; nb < BPW; ++nb) xs[s][nb] = *(const float*)(stage + 4 * s * ROWB + xo[nb]);
      }
    f32x4 acc[BPW];
#pragma unroll
    for (int nb = 0; nb < BPW; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nu > 0) {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
        if (nu > 16 * qq) {
#pragma unroll
          for (int s = 4 * qq; s < 4 * qq + 4 && s < 14; ++s)
#pragma unroll
            for (int nb = 0; nb < BPW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[s][nb], wq[qq][s & 3], acc[nb], 0, 0, 0);
        }
    } else {          // rare: the tile's sources do not fit; its rows are gathered from global memory (sequential fma chain)
      const int r0 = *(const int32_t*)(rec + kRecR0);
      if (m < nrows) {
        const int gs = a.rowptr[r0 + m], ge = a.rowptr[r0 + m + 1];
        for (int k = gs; k < ge; ++k) {
          const int2 e = a.lt_idx_w[k];
          const float w = __int_as_float(e.y);
#pragma unroll
          for (int nb = 0; nb < BPW; ++nb) {
            const f32x4 xv = *(const f32x4*)(X + (int64_t)e.x * a.ldx + (wave * BPW + nb) * 16 + 4 * fg);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[nb][c] = fmaf(w, xv[c], acc[nb][c]);
          }
        }
      }
    }
    const float sdst = a.alpha * sd;
#pragma unroll
    for (int nb = 0; nb < BPW; ++nb) {
      f32x4 y;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float t = sdst * acc[nb][c];
        if (NEPI >= 1) t = fmaf(a.beta, x0v[nb][c], t);
        if (NEPI >= 2) t = fmaf(a.gamma, x1v[nb][c], t);
        y[c] = t;
      }
      if (m < nrows) *(f32x4*)(Y + (int64_t)row * a.ldy + (wave * BPW + nb) * 16 + 4 * fg) = y;
    }
    st = st + 1 == D ? 0 : st + 1;
    rs = rs + 1 == L::NREC ? 0 : rs + 1;
  }
}

template <int NBW, int NEPI, int D, int BPW>
int launch_ring_f32_k(const SpmmArgs& b, hipStream_t stream) {
  constexpr int lds = RingLdsF<256 * NBW, NEPI, D>::total;
  int per_cu = (160 * 1024) / lds;
  per_cu = per_cu > 2 ? 2 : per_cu;
  int64_t nb = (int64_t)per_cu * 256;
  if (nb > b.lt_nrec) nb = b.lt_nrec;
  nb = (nb + 7) / 8 * 8;
  spmm_ring_f32<NBW, NEPI, D, BPW><<<(int)nb, 64 * (4 * NBW / BPW) + 256, 0, stream>>>(b, b.lt_nrec);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

// 128 channels: one launch.  256 channels: the two 128-channel halves of the rows, one launch each -- a 1 KB row leaves room
// for only two ring stages (one tile in flight: 0.59 of the peak, measured), two passes over 512-byte half rows at three
// stages reach 0.65 (tools/ring_f32_probe.py, profiles/r05_ring_f32_probe.txt) although the tile records are read twice.
// false: the shape stays with the other kernels (plain 256-channel rows: the shared-gather kernel is level with two passes).
bool launch_ring_f32(const SpmmArgs& a, hipStream_t stream, int* rc) {
  SpmmArgs b = a;
  if (!a.X0 && a.X1) { b.X0 = a.X1; b.ldx0 = a.ldx1; b.beta = a.gamma; b.X1 = nullptr; b.ldx1 = 0; b.gamma = 0.f; }
  const int nepi = (b.X0 ? 1 : 0) + (b.X1 ? 1 : 0);
  if (a.C == 256 && nepi == 0) return false;
  const int halves = a.C / 128;
  b.C = 128;
  for (int h = 0; h < halves; ++h) {
    *rc = nepi == 2 ? launch_ring_f32_k<2, 2, 3, 2>(b, stream) : nepi == 1 ? launch_ring_f32_k<2, 1, 3, 2>(b, stream) : launch_ring_f32_k<2, 0, 3, 2>(b, stream);
    if (*rc != SG_OK) return true;
    b.X = (const float*)b.X + 128;
    b.Y = (float*)b.Y + 128;
    if (b.X0) b.X0 = (const float*)b.X0 + 128;
    if (b.X1) b.X1 = (const float*)b.X1 + 128;
  }
  return true;
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
    const int64_t orow = a.row_id ? a.row_id[row] : row;       // the caller's row of processing position `row`
    if (a.X0) y = fmaf(a.beta, V::load1(a.X0, orow * a.ldx0 + c), y);
    if (a.X1) y = fmaf(a.gamma, V::load1(a.X1, orow * a.ldx1 + c), y);
    V::store1(a.Y, orow * a.ldy + c, y);
  }
}

// bf16 rows of FOUR channels (8 bytes: the network's input layer, util/networks.py:15 h[0] = 4) -- too short for the
// 16-byte vectors of spmm_rows, and one thread per ELEMENT (spmm_scalar) spends four threads' worth of row-pointer and
// index loads per row.  One thread per ROW: 8-byte gathers, up to four in flight (slots past the end of the row re-read
// its last neighbour with weight zero, like spmm_rows' batches), the same fma chain in CSR order as spmm_scalar, so the
// result is bit-identical to it.
__global__ __launch_bounds__(kBlock) void spmm_quad_bf16(const SpmmArgs a) {
  const uint16_t* __restrict__ X = (const uint16_t*)a.X;
  const uint16_t* __restrict__ X0 = (const uint16_t*)a.X0;
  const uint16_t* __restrict__ X1 = (const uint16_t*)a.X1;
  uint16_t* __restrict__ Y = (uint16_t*)a.Y;
  auto widen = [](uint2 v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  };
  for (int row = blockIdx.x * kBlock + threadIdx.x; row < a.n_rows; row += gridDim.x * kBlock) {
    const int k0 = a.rowptr[row], k1 = a.rowptr[row + 1];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = k0; k < k1; k += 4) {
      uint2 v[4];
      float w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool on = k + u < k1;
        const int kk = on ? k + u : k1 - 1;
        int j;
        float ws;
        if (a.idx_w) {
          const int2 e = a.idx_w[kk];
          j = e.x;
          ws = __int_as_float(e.y);
        } else {
          j = a.idx[kk];
          ws = a.scale_src ? a.scale_src[j] : 1.0f;
        }
        w[u] = on ? ws : 0.f;
        v[u] = *(const uint2*)(X + (int64_t)j * a.ldx);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float x[4];
        widen(v[u], x);
        if (k + u < k1) {                     // (a skipped slot must not touch the sum: 0 * inf would)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = fmaf(w[u], x[c], acc[c]);
        }
      }
    }
    const float sd = a.alpha * (a.scale_dst ? a.scale_dst[row] : 1.0f);
    const int64_t orow = a.row_id ? a.row_id[row] : row;
    float y[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) y[c] = sd * acc[c];
    if (X0) {
      float x[4];
      widen(*(const uint2*)(X0 + orow * a.ldx0), x);
#pragma unroll
      for (int c = 0; c < 4; ++c) y[c] = fmaf(a.beta, x[c], y[c]);
    }
    if (X1) {
      float x[4];
      widen(*(const uint2*)(X1 + orow * a.ldx1), x);
#pragma unroll
      for (int c = 0; c < 4; ++c) y[c] = fmaf(a.gamma, x[c], y[c]);
    }
    uint2 out;
    out.x = Vt<bf16_tag>::cvt2(y[0], y[1]);
    out.y = Vt<bf16_tag>::cvt2(y[2], y[3]);
    *(uint2*)(Y + orow * a.ldy) = out;
  }
}

// The same idea for bf16 rows of 8 / 16 / 32 channels (NV = 1 / 2 / 4 vectors of 16 bytes): one thread owns a whole ROW,
// four neighbours in flight (NV loads each), CSR-order fma chain -- no LDS staging of the neighbour lists, no lane
// groups.  Behind SG_TUNE_FLAGS bit 9 while it is being measured against spmm_rows<.., G = NV, ..>.
template <int NV>
__global__ __launch_bounds__(kBlock) void spmm_thread_rows_bf16(const SpmmArgs a) {
  using V = Vt<bf16_tag>;
  const uint16_t* __restrict__ X = (const uint16_t*)a.X;
  const uint16_t* __restrict__ X0 = (const uint16_t*)a.X0;
  const uint16_t* __restrict__ X1 = (const uint16_t*)a.X1;
  uint16_t* __restrict__ Y = (uint16_t*)a.Y;
  for (int row = blockIdx.x * kBlock + threadIdx.x; row < a.n_rows; row += gridDim.x * kBlock) {
    const int k0 = a.rowptr[row], k1 = a.rowptr[row + 1];
    float acc[NV * 8];
#pragma unroll
    for (int c = 0; c < NV * 8; ++c) acc[c] = 0.f;
    for (int k = k0; k < k1; k += 4) {
      u32x4 v[4][NV];
      float w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kk = k + u < k1 ? k + u : k1 - 1;
        int j;
        if (a.idx_w) {
          const int2 e = a.idx_w[kk];
          j = e.x;
          w[u] = __int_as_float(e.y);
        } else {
          j = a.idx[kk];
          w[u] = a.scale_src ? a.scale_src[j] : 1.0f;
        }
#pragma unroll
        for (int q = 0; q < NV; ++q) v[u][q] = *(const u32x4*)(X + (int64_t)j * a.ldx + q * 8);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (k + u < k1) {
#pragma unroll
          for (int q = 0; q < NV; ++q) {
            float x[8];
            V::unpack(v[u][q], x);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[q * 8 + c] = fmaf(w[u], x[c], acc[q * 8 + c]);
          }
        }
    }
    const float sd = a.alpha * (a.scale_dst ? a.scale_dst[row] : 1.0f);
    const int64_t orow = a.row_id ? a.row_id[row] : row;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      float y[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) y[c] = sd * acc[q * 8 + c];
      if (X0) {
        float x[8];
        V::unpack(*(const u32x4*)(X0 + orow * a.ldx0 + q * 8), x);
#pragma unroll
        for (int c = 0; c < 8; ++c) y[c] = fmaf(a.beta, x[c], y[c]);
      }
      if (X1) {
        float x[8];
        V::unpack(*(const u32x4*)(X1 + orow * a.ldx1 + q * 8), x);
#pragma unroll
        for (int c = 0; c < 8; ++c) y[c] = fmaf(a.gamma, x[c], y[c]);
      }
      *(u32x4*)(Y + orow * a.ldy + q * 8) = V::pack(y);
    }
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



// Gathers per batch by shape (see spmm_rows, CAP), measured on the 1 M-vertex mesh (tools/agg_bench.py --variants
// unroll=8|6|4, profiles/r02_agg_cap_sweep.json): 4 gathers per batch (6 resident wavefronts per SIMD) win where a
// lane group is narrow (C <= 32 fp32 / C <= 64 bf16: -7 .. -19 %) and for bf16 C = 256 (-6 .. -7 %); 8 per batch stays
// ahead for the shapes in between (fp32 C = 64: 4 is 11 % slower) and for one-row-per-wavefront shapes.
template <typename T, int G, int R>
constexpr int default_cap() {
  return (G <= 8 || (sizeof(typename Vt<T>::elem) == 2 && G == 32 && R == 1)) ? 4 : 8;
}

template <typename T, int G, int R, int NEPI>
int launch_rows_epi(const SpmmArgs& a, hipStream_t stream) {
  constexpr int RPW = 64 / G;
  // rows per wavefront chunk: large enough to amortise staging, small enough to keep the band
  // of rows that are in flight at once (the sweep front) thin, and >> 256 workgroups in flight
  int ch = g_tuning.chunk_rows;
  if (ch <= 0) {
    // measured on the 1 M-vertex mesh (tools/agg_bench.py): wide rows want a thin sweep front
    const int row_bytes = a.C * (int)sizeof(typename Vt<T>::elem);
    ch = row_bytes >= 2048 ? 4 : row_bytes >= 1024 ? (sizeof(typename Vt<T>::elem) == 4 ? 4 : 8) : row_bytes >= 512 ? 8 : 16;
    // (round 4, dense 64-byte bf16 rows -- the planes of a 32-channel layer: 32 rows per chunk, 0.040 / 0.049 / 0.050 ms ->
    // 0.037 / 0.044 / 0.044 ms with 0 / 1 / 2 epilogue operands; 32- and 128-byte rows measured level or slower)
    if (sizeof(typename Vt<T>::elem) == 2 && row_bytes == 64) ch = 32;
    while (ch > RPW && (int64_t)a.n_rows / ch < 4096) ch >>= 1;
  }
  if (ch < RPW) ch = RPW;
  if (ch > kChMax) ch = kChMax;
  ch = (ch / RPW) * RPW;
  const int64_t chunks = ((int64_t)a.n_rows + ch - 1) / ch;
  const int nblocks = (int)((chunks + kWaves - 1) / kWaves);
  // 24-bit row ids / row pitch and 32-bit byte offsets into X: the cheap gather addressing (see gather_batch)
  constexpr int64_t esz = sizeof(typename Vt<T>::elem);
  const bool narrow = !(g_tuning.flags & kFlagWideAddr) && a.n_cols > 0 && a.n_cols < (1 << 24) &&
                      a.ldx * esz < (1 << 24) && a.n_cols * a.ldx * esz < ((int64_t)1 << 32);
  const int cap = g_tuning.unroll == 4 || g_tuning.unroll == 6 || g_tuning.unroll == 8 ? g_tuning.unroll : default_cap<T, G, R>();
#define SG_ROWS(NW, CP) spmm_rows<T, G, R, NEPI, NW, CP><<<nblocks, kBlock, 0, stream>>>(a, ch, nblocks, g_tuning.flags)
  if (!narrow) SG_ROWS(false, 8);          // > 4 GiB operands / > 16 M rows: rare, one variant
  else if (cap == 4) SG_ROWS(true, 4);
  else if (cap == 6) SG_ROWS(true, 6);
  else SG_ROWS(true, 8);
#undef SG_ROWS
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

template <typename T, int G, int R>
int launch_shared(const SpmmArgs& a, hipStream_t stream) {
  constexpr int MT = (64 / G) * kShTilesPerGroup;
  const int64_t n_mt = ((int64_t)a.n_rows + kTileRows - 1) / kTileRows;
  const int nblocks = (int)((n_mt + (int64_t)MT * kWaves - 1) / ((int64_t)MT * kWaves));
  SpmmArgs b = a;
  if (!a.X0 && a.X1) { b.X0 = a.X1; b.ldx0 = a.ldx1; b.beta = a.gamma; b.X1 = nullptr; b.ldx1 = 0; b.gamma = 0.f; }
  constexpr int64_t esz = sizeof(typename Vt<T>::elem);
  const bool narrow = !(g_tuning.flags & kFlagWideAddr) && a.n_cols > 0 && a.n_cols < (1 << 24) &&
                      a.ldx * esz < (1 << 24) && a.n_cols * a.ldx * esz < ((int64_t)1 << 32);
#define SG_SHARED(NE, NW) spmm_shared<T, G, R, NE, NW><<<nblocks, kBlock, 0, stream>>>(b, nblocks, g_tuning.flags)
  if (b.X0 && b.X1) { if (narrow) SG_SHARED(2, true); else SG_SHARED(2, false); }
  else if (b.X0) { if (narrow) SG_SHARED(1, true); else SG_SHARED(1, false); }
  else { if (narrow) SG_SHARED(0, true); else SG_SHARED(0, false); }
#undef SG_SHARED
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
  if (esz == 2 && a.C == 4 && a.ldx % 4 == 0 && a.ldy % 4 == 0 && ((uintptr_t)a.X & 7) == 0 && ((uintptr_t)a.Y & 7) == 0 &&
      (!a.X0 || (a.ldx0 % 4 == 0 && ((uintptr_t)a.X0 & 7) == 0)) && (!a.X1 || (a.ldx1 % 4 == 0 && ((uintptr_t)a.X1 & 7) == 0)) &&
      !(g_tuning.flags & kFlagNoQuad)) {
    int nb = (a.n_rows + kBlock - 1) / kBlock;
    nb = nb > 256 * 64 ? 256 * 64 : nb;
    spmm_quad_bf16<<<nb, kBlock, 0, stream>>>(a);
    SG_HIP_TRY(hipGetLastError());
    return SG_OK;
  }
  if (!vec_ok) {
    const int64_t total = (int64_t)a.n_rows * a.C;
    int64_t nb = (total + kBlock - 1) / kBlock;
    if (nb > 256 * 32) nb = 256 * 32;
    spmm_scalar<T><<<(int)nb, kBlock, 0, stream>>>(a);
    SG_HIP_TRY(hipGetLastError());
    return SG_OK;
  }
  const int nvec = a.C / VEC;
  if (esz == 2 && (g_tuning.flags & kFlagThreadRows) && (nvec == 1 || nvec == 2 || nvec == 4)) {
    int nb = (a.n_rows + kBlock - 1) / kBlock;
    nb = nb > 256 * 64 ? 256 * 64 : nb;
    if (nvec == 1) spmm_thread_rows_bf16<1><<<nb, kBlock, 0, stream>>>(a);
    else if (nvec == 2) spmm_thread_rows_bf16<2><<<nb, kBlock, 0, stream>>>(a);
    else spmm_thread_rows_bf16<4><<<nb, kBlock, 0, stream>>>(a);
    SG_HIP_TRY(hipGetLastError());
    return SG_OK;
  }
  // Wide rows of a graph that carries mini-tiles: gather each distinct source row of 4 rows once.
  // Measured on the 1 M-vertex Morton-ordered mesh (tools/agg_bench.py, variants interleaved in one process):
  // it pays for fp32 rows of 2 KiB (C=512: 1.13 -> 0.97 ms plain, 1.40 -> 1.31 ms with an epilogue operand) and
  // for plain 1 KiB rows (C=256: 0.571 -> 0.515 ms); with an epilogue operand the 1 KiB case is a wash
  // (0.687 vs 0.708 ms) and bf16 loses (4 x 8 accumulators per lane cost an occupancy step: 0.31 -> 0.43 ms at
  // C=256), so those keep spmm_rows unless SG_TUNE_TILED_MIN_ROW_BYTES forces them (negative value = force).
  const int row_bytes = a.C * (int)sizeof(typename Vt<T>::elem);
  const int tmin = g_tuning.tiled_min_row_bytes;
  const bool forced = tmin < 0 && row_bytes >= -tmin;
  const bool pays = tmin > 0 && sizeof(typename Vt<T>::elem) == 4 && row_bytes >= tmin &&
                    (row_bytes >= 2 * tmin || !(a.X0 || a.X1));
  // LDS-tile kernel (opt-in while it is being measured: SG_TUNE_FLAGS bit 7): whole-row lane groups of 16 / 32 / 64 lanes
  if (!(g_tuning.flags & kFlagNoRing) && esz == 2 && a.lt_rec && a.lt_nrec > 0 && a.n_cols < ((int64_t)1 << 31)) {
    if (a.C == 128) return launch_ring<2>(a, stream);
    if (a.C == 256) return launch_ring<4>(a, stream);
  }
  if (!(g_tuning.flags & (kFlagNoRing | kFlagNoRingF32)) && esz == 4 && a.lt_rec && a.lt_nrec > 0 && a.n_cols < ((int64_t)1 << 31)) {
    int rc = SG_OK;
    if ((a.C == 128 || a.C == 256) && launch_ring_f32(a, stream, &rc)) return rc;
  }
  if ((g_tuning.flags & kFlagLdsTiles) && a.lt_uptr && a.ldx % VEC == 0 && a.n_cols < ((int64_t)1 << 31)) {
    if (nvec == 16) return launch_lds<T, 16>(a, stream);
    if (nvec == 32) return launch_lds<T, 32>(a, stream);
    if (nvec == 64) return launch_lds<T, 64>(a, stream);
  }
  if (a.tile_uptr && (forced || pays) && !(g_tuning.flags & kFlagNoTiles)) {
    if (nvec > 16 && nvec <= 32) return launch_shared<T, 32, 1>(a, stream);
    if (nvec > 32 && nvec <= 64) return launch_shared<T, 64, 1>(a, stream);
    if (nvec > 64 && nvec <= 128) return launch_shared<T, 64, 2>(a, stream);
  }
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

bool tiles_enabled() { return g_tuning.tiled_min_row_bytes != 0; }
bool lds_tiles_enabled() { return (g_tuning.flags & kFlagLdsTiles) != 0; }
bool ring_enabled() { return (g_tuning.flags & kFlagNoRing) == 0; }
bool ring_f32_enabled() { return (g_tuning.flags & (kFlagNoRing | kFlagNoRingF32)) == 0; }

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

int launch_spmm(const SpmmArgs& a_in, int dtype, hipStream_t stream) {
  if (a_in.n_rows == 0 || a_in.C == 0) return SG_OK;
  SpmmArgs a = a_in;
  if (g_tuning.flags & kFlagNoPackedScale) {     // A/B switch: chase scale_src[idx[k]] as before
    a.idx_w = nullptr;
    a.tile_uniq_w = nullptr;
  }
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
