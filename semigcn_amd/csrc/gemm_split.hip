// The reference's OWN precision (fp32 features: BASELINE configs c2 / c3, util/networks.py:40-53, [3P] ChebConv `lins[k]`)
// on the bf16 matrix cores: every fp32 operand is split EXACTLY into three bf16 pieces
//
//     x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)      (round to nearest even; 3 x 8
//                                                                                          significand bits >= fp32's 24)
//
// and a product a * b is accumulated in fp32 from the six piece products whose weight is >= 2^-16 of it:
//     hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi          (dropped: mid*lo, lo*mid, lo*lo <= 2^-25 |a b| each)
// gfx950's bf16 MFMA runs at 16 x the rate of its fp32 MFMA (v_mfma_f32_16x16x4_f32 = the fp32 VALU rate), so six bf16
// instructions per fp32 one leave a 2.67 x higher ceiling than an fp32-MFMA kernel (the BLAS library's, which sits at
// 0.85 of that peak) at fp32-equivalent error; small-integer operands are reproduced bit for bit (mid = lo = 0).
//
//   sg::gemm_nt_f32s   C[M, N] = A[M, K] op(W) (+ bias)       forward product and input gradient of a ChebConv layer
//   sg::gemm_tn_f32s   out[N, Kp] = A[M, N]^T B[M, Kp]        weight gradient (a reduction over all M vertices)
//
// ---- nt ---------------------------------------------------------------------------------------------------------------
// The weights are split ONCE per call by sg::pack_split (a few microseconds: <= 768 x 768 entries) into the exact byte
// image the kernel's LDS ring holds: per (256- or 128-column tile, 32-deep K block) three planes x NF fragments x 1 KB,
// each fragment lane-linear (lane l's 16 bytes = the eight k values the MFMA takes from that lane).  So the weight stream
// is a plain linear copy by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip), and every fragment read is a
// conflict-free ds_read_b128 at `base + 16 * lane`.
// A is NOT staged through LDS: one workgroup = 8 wavefronts x 32 rows, a wavefront owns its 32 rows x all NT columns, so
// no other wavefront ever needs its A rows -- they go global -> VGPR (two 16-byte loads per lane and 16 x 32 fragment, 64
// contiguous bytes per row and instruction), are split in registers (11 VALU per pair of values, v_cvt_pk_bf16_f32 does
// the rounding and the packing) two K blocks ahead of their use, and 192 MFMAs (NT = 256) run on each K block: 0.5 VALU
// per MFMA, inside the issue slots a 16-cycle MFMA leaves free.  The k order inside a 32-block is permuted (lane group q
// holds k = 4q..4q+3 and 16+4q..16+4q+3) so that each A load is one dwordx4; the weight image uses the same order.
// Persistent: a workgroup keeps its column tile and walks row tiles; the DMA ring and the A prefetch run across tile
// boundaries.  D = C^T fragments (weights as the MFMA's first operand): a lane ends with four consecutive columns of one
// row -- the tile leaves as 16-byte stores, 64 contiguous bytes per row.
#include "sg_common.h"

namespace sg {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kSplitThreads = 512;

#define SGS_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__device__ __forceinline__ void glds16(const void* src, uint8_t* lds_dst) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                   (void __attribute__((address_space(3)))*)lds_dst, 16, 0, 0);
}

// position of k-slot e (0..7) of lane group q inside a 32-deep K block
__host__ __device__ __forceinline__ int kmap(int q, int e) { return e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4); }

// x[0..7] -> three bf16x8 pieces, exact: x = h + m + l
struct Pieces {
  bf16x8 h, m, l;
};
__device__ __forceinline__ uint32_t pk(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ Pieces split8(const f32x4 a, const f32x4 b) {
  const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float x0 = x[2 * p], x1 = x[2 * p + 1];
    h[p] = pk(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, h[p] << 16);
    const float r1 = x1 - __builtin_bit_cast(float, h[p] & 0xffff0000u);
    m[p] = pk(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, m[p] << 16);
    const float s1 = r1 - __builtin_bit_cast(float, m[p] & 0xffff0000u);
    l[p] = pk(s0, s1);
  }
  Pieces o;
  o.h = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
  o.m = __builtin_bit_cast(bf16x8, u32x4{m[0], m[1], m[2], m[3]});
  o.l = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
  return o;
}

// ---- weights -> the LDS image -----------------------------------------------------------------------------------------------
// W element (n, k) at W[n * rs + k * cs] (rs / cs in elements: [N, K] row-major is (K', 1), [K, N] row-major is (1, N')).
// out: [col tile][K block][plane 3][fragment NF][lane 64][8] bf16; columns >= N are zeros.
struct PackSplit {
  const float* W;
  int64_t rs, cs;
  int N, K, NF, n_col_tiles;
  uint8_t* out;
};
__global__ __launch_bounds__(256) void pack_split(const PackSplit p) {
  const int nkb = p.K >> 5;
  const int64_t frag = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // (ct, kb, j)
  const int64_t n_frag = (int64_t)p.n_col_tiles * nkb * p.NF;
  if (frag >= n_frag) return;
  const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
  const int j = (int)(frag % p.NF);
  const int kb = (int)((frag / p.NF) % nkb);
  const int ct = (int)(frag / ((int64_t)p.NF * nkb));
  const int n = (ct * p.NF + j) * 16 + fr;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = n < p.N ? p.W[(int64_t)n * p.rs + (int64_t)(kb * 32 + kmap(fq, e)) * p.cs] : 0.f;
  const Pieces s = split8(f32x4{x[0], x[1], x[2], x[3]}, f32x4{x[4], x[5], x[6], x[7]});
  uint8_t* const chunk = p.out + ((int64_t)ct * nkb + kb) * (3 * p.NF * 1024);
  *(bf16x8*)(chunk + (0 * p.NF + j) * 1024 + lane * 16) = s.h;
  *(bf16x8*)(chunk + (1 * p.NF + j) * 1024 + lane * 16) = s.m;
  *(bf16x8*)(chunk + (2 * p.NF + j) * 1024 + lane * 16) = s.l;
}

struct SplitNt {
  const float* A; int64_t lda;
  const uint8_t* Bp;
  const float* bias;                    // nullable
  float* C; int64_t ldc;
  int M, N, K;
  int n_col_tiles, n_row_tiles, streams;
};

template <int NF>
__global__ __launch_bounds__(kSplitThreads, 1) void gemm_nt_f32s(const SplitNt g) {
  constexpr int SLOT = 3 * NF * 1024;              // one K block of the column tile: 48 KB (NT = 256) / 24 KB (NT = 128)
  constexpr int RING = NF == 16 ? 3 : 4;
  constexpr int D = RING - 1;                      // K blocks the DMA runs ahead
  constexpr int P = SLOT / 1024 / 8;               // DMA instructions per wavefront and K block
  constexpr int NSTORE = 2 * NF;                   // stores per wavefront and finished tile
  constexpr int NT = NF * 16;
  __shared__ __attribute__((aligned(1024))) uint8_t lds[RING * SLOT + NT * 4];      // the ONLY LDS object
  float* const bias_s = (float*)(lds + RING * SLOT);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;

  const int b = blockIdx.x;
  const int slot = b >> 3;
  const int ct = slot % g.n_col_tiles;
  const int stream = (b & 7) + 8 * (slot / g.n_col_tiles);
  const int my_tiles = stream < g.n_row_tiles ? (g.n_row_tiles - stream + g.streams - 1) / g.streams : 0;
  if (my_tiles == 0) return;
  const int nkb = g.K >> 5;
  const int total = my_tiles * nkb;
  const bool exact_stores = (ct + 1) * NT <= g.N;         // every store instruction of a full tile has an active lane

  if (tid < NT) {
    const int col = ct * NT + tid;
    bias_s[tid] = (g.bias && col < g.N) ? g.bias[col] : 0.f;
  }
  // (an ordinary load + ds_write: hipcc waits for it right here, before the first DMA is counted)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // ---- the weight stream: K block kb of this column tile = SLOT contiguous bytes; wavefront w copies pieces w*P .. w*P+P-1
  const uint8_t* const Bt = g.Bp + (int64_t)ct * nkb * SLOT + (wave * P) * 1024 + lane * 16;
  int dk = 0, dslot = 0;                                  // DMA cursor: K block (mod nkb) and ring slot
  auto dma = [&]() {
    const uint8_t* src = Bt + (int64_t)dk * SLOT;
    uint8_t* dst = lds + dslot * SLOT + (wave * P) * 1024;
#pragma unroll
    for (int i = 0; i < P; ++i) glds16(src + i * 1024, dst + i * 1024);
    dk = dk + 1 == nkb ? 0 : dk + 1;
    dslot = dslot + 1 == RING ? 0 : dslot + 1;
  };

  // ---- the A stream: lane (fr, fq) of row fragment i reads row row0 + 32 wave + 16 i + fr, k = 4 fq .. and 16 + 4 fq ..
  const float* pa[2];
  int rt = 0, rk = 0;                                     // A cursor: tile, K block (clamped to the stream's last step)
  auto set_tile = [&](int t) {
    const int row0 = (stream + t * g.streams) * 256 + wave * 32 + fr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int r = row0 + 16 * i;
      r = r < g.M - 1 ? r : g.M - 1;
      pa[i] = g.A + (int64_t)r * g.lda + 4 * fq;
    }
  };
  struct Raw {
    f32x4 v[2][2];
  };
  auto load_a = [&]() {
    Raw r;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      r.v[i][0] = *(const f32x4*)(pa[i] + rk * 32);
      r.v[i][1] = *(const f32x4*)(pa[i] + rk * 32 + 16);
    }
    if (rk + 1 < nkb) {
      ++rk;
    } else if (rt + 1 < my_tiles) {
      ++rt;
      rk = 0;
      set_tile(rt);
    }
    return r;
  };

  f32x4 acc[2][NF];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue ------------------------------------------------------------------------------------------------------------
  set_tile(0);
#pragma unroll
  for (int d = 0; d < D; ++d) dma();
  Pieces cur[2], nxt[2];
  {
    const Raw r0 = load_a();
#pragma unroll
    for (int i = 0; i < 2; ++i) cur[i] = split8(r0.v[i][0], r0.v[i][1]);
  }
  Raw rawc = load_a();                                    // K block 1

  int since = D;                                          // K blocks since a tile was stored, capped at D
  int rslot = 0, ck = 0, ctile = 0;
  constexpr int kBase = 4 + (D - 1) * (P + 4);            // vector-memory operations younger than the DMA of the block read next
  for (int s = 0; s < total; ++s) {
    // the DMA of this K block has landed (this wavefront's share), then everybody's; the slot read one block ago is free
    if (since < D && exact_stores) {
      if constexpr (NF == 16) SGS_WAIT_VM(46); else SGS_WAIT_VM(34);
    } else {
      if constexpr (NF == 16) SGS_WAIT_VM(14); else SGS_WAIT_VM(18);
    }
    static_assert(kBase == (NF == 16 ? 14 : 18) && kBase + NSTORE == (NF == 16 ? 46 : 34), "counted waits");
    __builtin_amdgcn_s_barrier();
    dma();                                                // K block s + D
    __builtin_amdgcn_sched_barrier(0);                    // (the counted waits assume the DMA is the step's FIRST vector-memory work)
    const Raw rawn = load_a();                            // A of K block s + 2
#pragma unroll
    for (int i = 0; i < 2; ++i) nxt[i] = split8(rawc.v[i][0], rawc.v[i][1]);      // pieces of K block s + 1

    const uint8_t* const sb = lds + rslot * SLOT + lane * 16;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const bf16x8 bh = *(const bf16x8*)(sb + (0 * NF + j) * 1024);
      const bf16x8 bm = *(const bf16x8*)(sb + (1 * NF + j) * 1024);
      const bf16x8 bl = *(const bf16x8*)(sb + (2 * NF + j) * 1024);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x4 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, cur[i].h, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, cur[i].l, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, cur[i].m, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, cur[i].h, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, cur[i].m, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, cur[i].h, c, 0, 0, 0);
        acc[i][j] = c;
      }
    }
    rslot = rslot + 1 == RING ? 0 : rslot + 1;
    since = since < D ? since + 1 : D;

    if (++ck == nkb) {                                    // the tile is complete: + bias, store, restart the sums
      const int row0 = (stream + ctile * g.streams) * 256 + wave * 32 + fr;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = row0 + 16 * i;
        float* const crow = g.C + (int64_t)row * g.ldc + ct * NT + 4 * fq;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          const f32x4 bv = *(const f32x4*)(bias_s + 16 * j + 4 * fq);
          const f32x4 o = acc[i][j] + bv;
          if (row < g.M && ct * NT + 16 * j + 4 * fq < g.N) *(f32x4*)(crow + 16 * j) = o;
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      ck = 0;
      ++ctile;
      since = 0;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) cur[i] = nxt[i];
    rawc = rawn;
  }
  SGS_WAIT_VM(0);                                         // no LDS-DMA may be in flight when the workgroup ends
}


// ---- tn: the weight gradient ---------------------------------------------------------------------------------------------------
// out[N, Kp] = A[M, N]^T B[M, Kp] (dWcat = dOut^T [Tx0|Tx1|Tx2]; autograd of the `lins[k]` calls, util/networks.py:42,49).
// The reduction index m is the ROW of both operands and both stream from HBM, so both are split on the way into LDS: one
// workgroup = one 256 (n) x 128 (k') tile of the result for one slab of rows, walked in steps of 32 rows.  Per step every
// thread loads 6 float4 (rows of A: 1 KB per wavefront instruction), splits them in registers (no redundancy: each value
// is split once per tile it feeds) and writes the three bf16 planes [32 m][256 | 128] into LDS (ds_write_b64; 32-byte
// segments XOR-swizzled by the row, as sg::gemm_tn_bf16); the MFMA fragments -- 8 consecutive m of one column -- are read
// with the transposing ds_read_b64_tr_b16.  Two buffers of 72 KB; the loads of step s + 1 are issued before the MFMAs of
// step s and stashed after them.  96 MFMAs per wavefront and step (4 x 4 fragments x 6 piece products), 132 VALU.
// Slab partials go to the workspace and are summed in slab order by sg::split_tn_reduce: deterministic.
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct SplitTn {
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  float* W;                                 // [slabs][N][Kp]
  int M, N, Kp;
  int tiles_k, n_tiles, slabs, steps;       // steps = ceil(M / 32)
};

__device__ __forceinline__ int tn_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

struct Pieces4 {
  u32x2 h, m, l;
};
__device__ __forceinline__ Pieces4 split4(const f32x4 a) {
  uint32_t h[2], m[2], l[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float x0 = a[2 * p], x1 = a[2 * p + 1];
    h[p] = pk(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, h[p] << 16);
    const float r1 = x1 - __builtin_bit_cast(float, h[p] & 0xffff0000u);
    m[p] = pk(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, m[p] << 16);
    const float s1 = r1 - __builtin_bit_cast(float, m[p] & 0xffff0000u);
    l[p] = pk(s0, s1);
  }
  return Pieces4{u32x2{h[0], h[1]}, u32x2{m[0], m[1]}, u32x2{l[0], l[1]}};
}

__global__ __launch_bounds__(kSplitThreads, 1) void gemm_tn_f32s(const SplitTn g) {
  constexpr int PA = 32 * 512, PB = 32 * 256;          // one plane of A ([32 m][256 n] bf16) / of B ([32 m][128 k'])
  constexpr int BUF = 3 * PA + 3 * PB;                 // 72 KB
  __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int b = blockIdx.x;
  const int slot = b >> 3;
  const int tile = slot % g.n_tiles;
  const int slab = (b & 7) + 8 * (slot / g.n_tiles);
  const int n0 = (tile / g.tiles_k) * 256, k0 = (tile % g.tiles_k) * 128;
  const int q = g.steps / g.slabs, rem = g.steps % g.slabs;
  const int first = slab * q + (slab < rem ? slab : rem);
  const int count = q + (slab < rem ? 1 : 0);

  // ---- staging map: A float4 f = tid + 512 i -> row (tid >> 6) + 8 i, columns 4 (tid & 63); B: row (tid >> 5) + 16 i, 4 (tid & 31)
  const int ca = tid & 63, cb = tid & 31;
  const bool a_ok = n0 + 4 * ca < g.N, b_ok = k0 + 4 * cb < g.Kp;           // N, Kp % 4 == 0
  const float* const a_src = g.A + (a_ok ? n0 + 4 * ca : 0);
  const float* const b_src = g.B + (b_ok ? k0 + 4 * cb : 0);
  int dst_a[4], dst_b[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (tid >> 6) + 8 * i;
    dst_a[i] = r * 512 + ((((ca >> 2) ^ tn_swz(r)) << 5) | ((ca & 3) << 3));
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (tid >> 5) + 16 * i;
    dst_b[i] = 3 * PA + r * 256 + ((((cb >> 2) ^ tn_swz(r)) << 5) | ((cb & 3) << 3));
  }
  f32x4 ra[4], rb[2];
  auto fetch = [&](int step) {
    const int m0 = (first + step) * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = m0 + (tid >> 6) + 8 * i;
      const bool in = m < g.M;
      m = in ? m : g.M - 1;
      const f32x4 v = *(const f32x4*)(a_src + (int64_t)m * g.lda);
      ra[i] = (in && a_ok) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int m = m0 + (tid >> 5) + 16 * i;
      const bool in = m < g.M;
      m = in ? m : g.M - 1;
      const f32x4 v = *(const f32x4*)(b_src + (int64_t)m * g.ldb);
      rb[i] = (in && b_ok) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto stash = [&](int buf) {
    uint8_t* const base = lds + buf * BUF;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const Pieces4 s = split4(ra[i]);
      *(u32x2*)(base + dst_a[i]) = s.h;
      *(u32x2*)(base + PA + dst_a[i]) = s.m;
      *(u32x2*)(base + 2 * PA + dst_a[i]) = s.l;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const Pieces4 s = split4(rb[i]);
      *(u32x2*)(base + dst_b[i]) = s.h;
      *(u32x2*)(base + PB + dst_b[i]) = s.m;
      *(u32x2*)(base + 2 * PB + dst_b[i]) = s.l;
    }
  };

  // ---- transposing fragment reads: lane = 16 fg + 4 fq + fp supplies row 8 fg + 4 hh + fq, columns 4 fp .. + 3 of a 16-column
  //      segment; lane 16 fg + i receives column i of those 4 rows (hh = 0 / 1: the two halves of the lane's 8 m)
  const int fg = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
  int row_a[2], row_b[2], swz[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int row = 8 * fg + 4 * hh + fq;
    row_a[hh] = row * 512 + fp * 8;
    row_b[hh] = row * 256 + fp * 8;
    swz[hh] = tn_swz(row);
  }
  auto frag = [&](const uint8_t* plane, const int (&row_off)[2], int seg) -> bf16x8 {
    bf16x4 h[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const uint8_t* p = plane + row_off[hh] + ((seg ^ swz[hh]) << 5);
      h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p);
    }
    return bf16x8{h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (count > 0) {
    fetch(0);
    stash(0);
  }
  for (int s = 0; s < count; ++s) {
    const int buf = s & 1;
    __syncthreads();                 // step s is in LDS (every wavefront's share); the other buffer's readers are done
    const bool more = s + 1 < count;
    if (more) fetch(s + 1);
    const uint8_t* const base = lds + buf * BUF;
    bf16x8 ah[4], am[4], al[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = frag(base, row_a, wn * 4 + i);
      am[i] = frag(base + PA, row_a, wn * 4 + i);
      al[i] = frag(base + 2 * PA, row_a, wn * 4 + i);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16x8 bh = frag(base + 3 * PA, row_b, wk * 4 + j);
      const bf16x8 bm = frag(base + 3 * PA + PB, row_b, wk * 4 + j);
      const bf16x8 bl = frag(base + 3 * PA + 2 * PB, row_b, wk * 4 + j);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[i], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[i], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, am[i], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, ah[i], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, am[i], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[i], c, 0, 0, 0);
        acc[i][j] = c;
      }
    }
    if (more) stash(buf ^ 1);
  }

  // ---- the slab's partial tile: D = (B fragment) x (A fragment): lane holds k' = 4 (lane >> 4) .. + 3 of column n = lane & 15
  float* const W = g.W + (int64_t)slab * g.N * g.Kp;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + (wn * 4 + i) * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = k0 + (wk * 4 + j) * 16 + 4 * (lane >> 4);
      if (n < g.N && kk < g.Kp) *(f32x4*)(W + (int64_t)n * g.Kp + kk) = acc[i][j];
    }
  }
}

// out[n][k] = the slabs' partial tiles added in slab order (64 x 16 threads: 64 consecutive float4, 16 slab groups, then the
// group sums in order: a fixed tree); += into the layer's weight .grad accumulators when a sink is given
__global__ __launch_bounds__(1024) void split_tn_reduce(const float* __restrict__ W, int n_slabs, int64_t elems, int Kp,
                                                        float* __restrict__ out, int64_t ldo, const GradSink sink) {
  __shared__ f32x4 s_part[16][64];
  const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
  const int64_t e = ((int64_t)blockIdx.x * 64 + x) * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (e < elems)
    for (int t = y; t < n_slabs; t += 16) acc += *(const f32x4*)(W + (int64_t)t * elems + e);
  s_part[y][x] = acc;
  __syncthreads();
  if (y == 0 && e < elems) {
#pragma unroll
    for (int k = 1; k < 16; ++k) acc += s_part[k][x];
    const int64_t n = e / Kp, kk = e - n * Kp;
    *(f32x4*)(out + n * ldo + kk) = acc;
    if (sink.mode) {
      float* const d = sink_ptr(sink, n, kk);
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] += acc[q];
    }
  }
}

}  // namespace

// ---- host side --------------------------------------------------------------------------------------------------------------
static inline int split_nf(int64_t N) { return N % 256 == 0 || N > 640 ? 16 : (N % 128 == 0 || N <= 128 ? 8 : 16); }

bool gemm_nt_f32s_supported(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc) {
  return M >= 256 && M < ((int64_t)1 << 31) - 512 && N >= 64 && N % 4 == 0 && K >= 64 && K % 32 == 0 && lda % 4 == 0 &&
         ldc % 4 == 0 && N < (1 << 20) && K < (1 << 20);
}

int64_t gemm_nt_f32s_workspace(int64_t N, int64_t K) {      // bytes of the split weight image
  const int nf = split_nf(N);
  const int64_t nt = nf * 16;
  return ((N + nt - 1) / nt) * (K / 32) * (3 * nf * 1024);
}

// C[M, N] = A[M, K] op(W) (+ bias): W element (n, k) at W[n * w_rs + k * w_cs]
int launch_gemm_nt_f32s(const float* A, int64_t lda, const float* W, int64_t w_rs, int64_t w_cs, const float* bias, float* C,
                        int64_t ldc, int64_t M, int64_t N, int64_t K, void* ws, int64_t ws_bytes, hipStream_t stream) {
  SG_REQUIRE(gemm_nt_f32s_supported(M, N, K, lda, ldc), "sg_gemm_nt_f32: unsupported shape (M=%lld N=%lld K=%lld)", (long long)M,
             (long long)N, (long long)K);
  SG_REQUIRE((((uintptr_t)A | (uintptr_t)C | (uintptr_t)ws) & 15) == 0, "sg_gemm_nt_f32: misaligned operand");
  SG_REQUIRE(ws && ws_bytes >= gemm_nt_f32s_workspace(N, K), "sg_gemm_nt_f32: workspace too small (%lld bytes given, %lld needed)",
             (long long)ws_bytes, (long long)gemm_nt_f32s_workspace(N, K));
  const int nf = split_nf(N);
  const int nt = nf * 16;
  PackSplit p;
  p.W = W; p.rs = w_rs; p.cs = w_cs;
  p.N = (int)N; p.K = (int)K; p.NF = nf;
  p.n_col_tiles = (int)((N + nt - 1) / nt);
  p.out = (uint8_t*)ws;
  const int64_t n_frag = (int64_t)p.n_col_tiles * (K / 32) * nf;
  pack_split<<<(int)((n_frag + 3) / 4), 256, 0, stream>>>(p);
  SG_HIP_TRY(hipGetLastError());

  SplitNt g;
  g.A = A; g.lda = lda;
  g.Bp = (const uint8_t*)ws;
  g.bias = bias;
  g.C = C; g.ldc = ldc;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.n_col_tiles = p.n_col_tiles;
  g.n_row_tiles = (int)((M + 255) / 256);
  int dev = 0, cus = 256;
  SG_HIP_TRY(hipGetDevice(&dev));
  SG_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  int streams = (cus / g.n_col_tiles) / 8 * 8;
  streams = streams < 8 ? 8 : streams;
  const int need = (g.n_row_tiles + 7) / 8 * 8;
  streams = streams > need ? need : streams;
  g.streams = streams;
  if (nf == 16) gemm_nt_f32s<16><<<streams * g.n_col_tiles, kSplitThreads, 0, stream>>>(g);
  else gemm_nt_f32s<8><<<streams * g.n_col_tiles, kSplitThreads, 0, stream>>>(g);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

bool gemm_tn_f32s_supported(int64_t M, int64_t N, int64_t Kp, int64_t lda, int64_t ldb) {
  return M >= 4096 && M < ((int64_t)1 << 31) - 64 && N >= 64 && N % 4 == 0 && Kp >= 64 && Kp % 4 == 0 && lda % 4 == 0 &&
         ldb % 4 == 0 && N < (1 << 20) && Kp < (1 << 20);
}

static int split_tn_slabs(int64_t M, int64_t N, int64_t Kp) {
  const int64_t n_tiles = ((N + 255) / 256) * ((Kp + 127) / 128);
  int64_t slabs = (256 / n_tiles) / 8 * 8;
  slabs = slabs < 8 ? 8 : slabs;
  const int64_t steps = (M + 31) / 32;
  while (slabs > 8 && steps < 4 * slabs) slabs -= 8;
  return (int)slabs;
}

int64_t gemm_tn_f32s_workspace(int64_t M, int64_t N, int64_t Kp) { return (int64_t)split_tn_slabs(M, N, Kp) * N * Kp * 4; }

int launch_gemm_tn_f32s(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int64_t Kp, void* ws,
                        int64_t ws_bytes, float* out, int64_t ldo, hipStream_t stream, const GradSink* sink_) {
  SG_REQUIRE(gemm_tn_f32s_supported(M, N, Kp, lda, ldb) && ldo % 4 == 0, "sg_gemm_tn_f32: unsupported shape (M=%lld N=%lld Kp=%lld)",
             (long long)M, (long long)N, (long long)Kp);
  SG_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)out | (uintptr_t)ws) & 15) == 0, "sg_gemm_tn_f32: misaligned operand");
  SG_REQUIRE(ws && ws_bytes >= gemm_tn_f32s_workspace(M, N, Kp), "sg_gemm_tn_f32: workspace too small (%lld bytes given, %lld needed)",
             (long long)ws_bytes, (long long)gemm_tn_f32s_workspace(M, N, Kp));
  const GradSink sink = sink_ ? *sink_ : GradSink{};
  SG_REQUIRE(sink.mode == 0 || sink.Cin % 4 == 0, "sg_gemm_tn_f32: a gradient sink needs Cin to be a multiple of 4");
  SplitTn g;
  g.A = A; g.lda = lda;
  g.B = B; g.ldb = ldb;
  g.W = (float*)ws;
  g.M = (int)M; g.N = (int)N; g.Kp = (int)Kp;
  g.tiles_k = (int)((Kp + 127) / 128);
  g.n_tiles = (int)((N + 255) / 256) * g.tiles_k;
  g.slabs = split_tn_slabs(M, N, Kp);
  g.steps = (int)((M + 31) / 32);
  gemm_tn_f32s<<<g.slabs * g.n_tiles, kSplitThreads, 0, stream>>>(g);
  SG_HIP_TRY(hipGetLastError());
  const int64_t elems = N * Kp;
  split_tn_reduce<<<(int)((elems / 4 + 63) / 64), 1024, 0, stream>>>(g.W, g.slabs, elems, (int)Kp, out, ldo, sink);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

}  // namespace sg
