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
// the rounding and the packing) one K block ahead of their use, and 192 MFMAs (NT = 256) run on each K block: 0.5 VALU
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

// (A variant with wavefronts 4-7 half a K block behind wavefronts 0-3 -- two workgroup barriers per block, the DMA issued by
// each half right after the barrier that frees the slot -- measured 1-2 % faster alone on the GPU and NOT deterministic with
// four processes sharing it: 1-6 of 60 repetitions of a product differed, tools/split_stress2.py; the cause was not found and
// the variant was removed)
//
// WAVES = 4 (round 6): the same kernel on 128-row tiles -- four wavefronts, a ring of three 128-column K blocks (72 KB: two
// workgroups per CU).  A product with few rows (the reference's own mesh sizes: 5 K vertices, the coarse levels of an MGCN) is
// a handful of 256-row tiles on 256 CUs; 128 x 128 tiles make four times as many work items, and the BLAS library leaves the
// float32 iteration at every size (launch_gemm_nt_f32s picks the variant by the row count).
template <int NF, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES == 8 ? 1 : 2) void gemm_nt_f32s(const SplitNt g) {
  constexpr int SLOT = 3 * NF * 1024;              // one K block of the column tile: 48 KB (NT = 256) / 24 KB (NT = 128)
  constexpr int RING = WAVES == 4 ? 3 : (NF == 16 ? 3 : 4);
  constexpr int D = RING - 1;                      // K blocks the DMA runs ahead
  constexpr int P = SLOT / 1024 / WAVES;           // DMA instructions per wavefront and K block
  constexpr int RT = 32 * WAVES;                   // rows of a tile
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

  for (int i = tid; i < NT; i += 64 * WAVES) {
    const int col = ct * NT + i;
    bias_s[i] = (g.bias && col < g.N) ? g.bias[col] : 0.f;
  }
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
    const int row0 = (stream + t * g.streams) * RT + wave * 32 + fr;
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
  Pieces cur[2];
  {
    const Raw r0 = load_a();
#pragma unroll
    for (int i = 0; i < 2; ++i) cur[i] = split8(r0.v[i][0], r0.v[i][1]);      // (hipcc waits for r0 here: the DMAs before it have landed)
  }

  // counted wait at the top of K block s (vector-memory operations of THIS wavefront younger than its pieces of the DMA of
  // block s; the stores of a finished tile sit in the same in-order queue).  Per K block, in program order:
  // [DMA P][A loads 4] .. [stores]; the DMA of block s was issued at the top of block s - D:  4 + (D - 1)(P + 4)
  constexpr int kTopPlain = 4 + (D - 1) * (P + 4);
  static_assert(kTopPlain + NSTORE < 64, "vmcnt is a 6-bit field");
#define SGS_WAIT2(base, with_stores)                                        \
  do {                                                                       \
    if (with_stores) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((base) + NSTORE) : "memory"); \
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(base) : "memory");        \
  } while (0)

  int since = D;                                          // completed K blocks since a tile was stored, capped at D
  int rslot = 0, ck = 0, ctile = 0;
  // One half of a K block.  The weight fragments of two column fragments (a GROUP: 6 x ds_read_b128) are read one group
  // ahead of the 24 MFMAs that use them: all eight wavefronts run this code in step, and a group that waited for its own reads
  // left both wavefronts of a SIMD parked on lgkmcnt together with the matrix pipe idle (0.59 busy, PMC, first version).
  // The reads are inline asm with hand-counted `s_waitcnt lgkmcnt(6)` TIED to the fragments they make valid: hipcc's own
  // bookkeeping waits for lgkmcnt(0) in front of the third group, i.e. for the reads it has just issued.
  const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)lds + lane * 16;
  uint32_t sbaddr = lds0;
  bf16x8 f[2][2][3];
#define SGS_RD1(dst, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(sbaddr), "n"(OFF))
#define SGS_RD6(B, H, GI)                                                       \
  SGS_RD1(f[B][0][0], (0 * NF + (H) * (NF / 2) + 2 * (GI) + 0) * 1024);         \
  SGS_RD1(f[B][0][1], (1 * NF + (H) * (NF / 2) + 2 * (GI) + 0) * 1024);         \
  SGS_RD1(f[B][0][2], (2 * NF + (H) * (NF / 2) + 2 * (GI) + 0) * 1024);         \
  SGS_RD1(f[B][1][0], (0 * NF + (H) * (NF / 2) + 2 * (GI) + 1) * 1024);         \
  SGS_RD1(f[B][1][1], (1 * NF + (H) * (NF / 2) + 2 * (GI) + 1) * 1024);         \
  SGS_RD1(f[B][1][2], (2 * NF + (H) * (NF / 2) + 2 * (GI) + 1) * 1024)
#define SGS_READY6(B, N)                                                                                         \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                       \
               : "+v"(f[B][0][0]), "+v"(f[B][0][1]), "+v"(f[B][0][2]), "+v"(f[B][1][0]), "+v"(f[B][1][1]), "+v"(f[B][1][2])::"memory")
#define SGS_MM(B, H, GI)                                                                  \
  _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                         \
    const bf16x8 bh = f[B][u][0], bm = f[B][u][1], bl = f[B][u][2];                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                       \
      f32x4 c = acc[i][(H) * (NF / 2) + 2 * (GI) + u];                                    \
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, cur[i].h, c, 0, 0, 0);              \
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, cur[i].l, c, 0, 0, 0);              \
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, cur[i].m, c, 0, 0, 0);              \
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, cur[i].h, c, 0, 0, 0);              \
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, cur[i].m, c, 0, 0, 0);              \
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, cur[i].h, c, 0, 0, 0);              \
      acc[i][(H) * (NF / 2) + 2 * (GI) + u] = c;                                          \
    }                                                                                     \
  }                                                                                       \
  __builtin_amdgcn_sched_barrier(0)
#define SGS_HALF(H)                    \
  do {                                 \
    SGS_RD6(0, H, 0);                  \
    SGS_RD6(1, H, 1);                  \
    SGS_READY6(0, 6);                  \
    SGS_MM(0, H, 0);                   \
    if constexpr (NF == 16) {          \
      SGS_RD6(0, H, 2);                \
      SGS_READY6(1, 6);                \
      SGS_MM(1, H, 1);                 \
      SGS_RD6(1, H, 3);                \
      SGS_READY6(0, 6);                \
      SGS_MM(0, H, 2);                 \
      SGS_READY6(1, 0);                \
      SGS_MM(1, H, 3);                 \
    } else {                           \
      SGS_READY6(1, 0);                \
      SGS_MM(1, H, 1);                 \
    }                                  \
  } while (0)

  for (int s = 0; s < total; ++s) {
    // ---- top of the K block
    SGS_WAIT2(kTopPlain, since < D && exact_stores);
    __builtin_amdgcn_s_barrier();                         // block s is in LDS (everybody's pieces); the slot of block s - 1 is free
    dma();                                                // K block s + D
    __builtin_amdgcn_sched_barrier(0);                    // (the counted waits assume this order of the vector-memory work)
    const Raw rawn = load_a();                            // A of K block s + 1: a whole block of MFMAs to arrive
    sbaddr = lds0 + rslot * SLOT;
    SGS_HALF(0);
    SGS_HALF(1);
    rslot = rslot + 1 == RING ? 0 : rslot + 1;
    since = since < D ? since + 1 : D;

    if (++ck == nkb) {                                    // the tile is complete: + bias, store, restart the sums
      const int row0 = (stream + ctile * g.streams) * RT + wave * 32 + fr;
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
    // the pieces of the next K block, in place (this block's MFMAs have been issued; beside the SIMD partner's MFMAs)
#pragma unroll
    for (int i = 0; i < 2; ++i) cur[i] = split8(rawn.v[i][0], rawn.v[i][1]);
  }
  SGS_WAIT_VM(0);                                         // no LDS-DMA may be in flight when the workgroup ends
#undef SGS_WAIT2
#undef SGS_HALF
#undef SGS_MM
#undef SGS_READY6
#undef SGS_RD6
#undef SGS_RD1
}

// ---- tn: the weight gradient ---------------------------------------------------------------------------------------------------
// out[N, Kp] = A[M, N]^T B[M, Kp] (dWcat = dOut^T [Tx0|Tx1|Tx2]; autograd of the `lins[k]` calls, util/networks.py:42,49).
// The reduction index m is the ROW of both operands and both stream from HBM, so both are split on the way into LDS: one
// workgroup = one 256 (n) x 128 (k') tile of the result for one slab of rows, walked in steps of 32 rows.  Per step every
// thread loads 6 float4 (rows of A: 1 KB per wavefront instruction) TWO steps ahead, splits them in registers (each value
// once per tile it feeds) and writes the three bf16 planes [32 m][256 | 128] into the other LDS buffer (ds_write_b64;
// 32-byte segments XOR-swizzled by the row, as sg::gemm_tn_bf16); the MFMA fragments -- 8 consecutive m of one column -- are
// read with the transposing ds_read_b64_tr_b16.  A wavefront owns 32 n x all 128 k': its two A fragments stay in registers
// for the step, the eight B fragments stream one ahead of the 12 MFMAs that use them (inline asm reads, hand-counted
// lgkmcnt: hipcc's own bookkeeping waits for every read in flight).  ONE instruction stream per wavefront, nothing waits
// for work it has just started: the six split-and-stash slices of the next step sit between the MFMA groups (pinned by empty
// asm statements: hipcc otherwise sinks them behind the last MFMA), the step's only workgroup barrier behind its last
// MFMA group.  96 MFMAs per wavefront and step, 132 VALU.
// Slab partials go to the workspace and are summed in slab order by sg::split_tn_reduce: deterministic.
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
  // ---- work items: (slab, tile) pairs; the workgroups with equal blockIdx % 8 (one XCD, for speed only) walk the items of the
  //      slabs x, x + 8, .. in order, tile fastest, so the tiles of a slab run side by side and share its rows through one L2
  const int b = blockIdx.x;
  const int xg = b & 7, yg = b >> 3, per_group = gridDim.x >> 3;
  const int items = (g.slabs >> 3) * g.n_tiles;        // per group
  const int q = g.steps / g.slabs, rem = g.steps % g.slabs;
  int n0 = 0, k0 = 0, first = 0, count = 0, slab = 0;

  // ---- staging map: A float4 f = tid + 512 i -> row (tid >> 6) + 8 i, columns 4 (tid & 63); B: row (tid >> 5) + 16 i, 4 (tid & 31)
  const int ca = tid & 63, cb = tid & 31;
  bool a_ok = false, b_ok = false;                     // N, Kp % 4 == 0
  const float* a_src = g.A;
  const float* b_src = g.B;
  uint32_t offa[4], offb[2];                           // byte offset of the thread's float4s inside a 32-row step of its operand
  const int64_t step_a = (int64_t)32 * g.lda * 4, step_b = (int64_t)32 * g.ldb * 4;
  const int full_steps = g.M >> 5;                     // steps whose 32 rows all exist
  auto set_item = [&](int item) {
    const int tile = item % g.n_tiles;
    slab = xg + 8 * (item / g.n_tiles);
    n0 = (tile / g.tiles_k) * 256;
    k0 = (tile % g.tiles_k) * 128;
    first = slab * q + (slab < rem ? slab : rem);
    count = q + (slab < rem ? 1 : 0);
    a_ok = n0 + 4 * ca < g.N;
    b_ok = k0 + 4 * cb < g.Kp;
    // (columns past N / Kp of the last tile: the thread reads column 0 instead -- what it stashes only reaches rows / columns
    //  of the result that are never stored, so it needs no zeroing; rows past M do: see fetch)
    const int acol = a_ok ? n0 + 4 * ca : 0, bcol = b_ok ? k0 + 4 * cb : 0;
    a_src = g.A + acol;
    b_src = g.B + bcol;
#pragma unroll
    for (int i = 0; i < 4; ++i) offa[i] = (uint32_t)((((tid >> 6) + 8 * i) * g.lda + acol) * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) offb[i] = (uint32_t)((((tid >> 5) + 16 * i) * g.ldb + bcol) * 4);
  };
  int dst[6];                                          // byte offset of the thread's float4 q in plane 0 of its operand
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (tid >> 6) + 8 * i;
    dst[i] = r * 512 + ((((ca >> 2) ^ tn_swz(r)) << 5) | ((ca & 3) << 3));
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (tid >> 5) + 16 * i;
    dst[4 + i] = 3 * PA + r * 256 + ((((cb >> 2) ^ tn_swz(r)) << 5) | ((cb & 3) << 3));
  }
  f32x4 raw[2][6];                                     // two steps of this thread's rows in flight
  // A step's rows come from a wave-uniform base (SGPRs, advanced by the scalar unit) + the thread's constant 32-bit offset: no
  // vector address arithmetic per load (the first version computed row * ld per load: 54 of its 228 VALU instructions per
  // step, beside 24 selects of the validity masks -- matrix pipe busy 0.40, PMC).  Only a step that holds rows past M (the
  // last one when M % 32 != 0, and the prefetches behind it) takes the path with per-thread clamps and zeroed rows.
  auto fetch = [&](f32x4 (&r)[6], int step) {
    const int st = first + step;                       // wave-uniform
    if (st < full_steps) {
      const char* const pa = (const char*)g.A + st * step_a;
      const char* const pb = (const char*)g.B + st * step_b;
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = *(const f32x4*)(pa + offa[i]);
#pragma unroll
      for (int i = 0; i < 2; ++i) r[4 + i] = *(const f32x4*)(pb + offb[i]);
    } else {
      const int m0 = st * 32;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int m = m0 + (tid >> 6) + 8 * i;
        const bool in = m < g.M;
        m = in ? m : g.M - 1;
        const f32x4 v = *(const f32x4*)(a_src + (int64_t)m * g.lda);
        r[i] = in ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int m = m0 + (tid >> 5) + 16 * i;
        const bool in = m < g.M;
        m = in ? m : g.M - 1;
        const f32x4 v = *(const f32x4*)(b_src + (int64_t)m * g.ldb);
        r[4 + i] = in ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  // one float4 of the thread -> its 3 x 8 bytes in the planes of buffer `buf` (the empty asm pins the arithmetic here)
  auto slice = [&](const f32x4 v, int qq, int buf) {
    Pieces4 p = split4(v);
    asm volatile("" : "+v"(p.h), "+v"(p.m), "+v"(p.l));
    uint8_t* const d = lds + buf * BUF + dst[qq];
    const int plane = qq < 4 ? PA : PB;
    *(u32x2*)(d) = p.h;
    *(u32x2*)(d + plane) = p.m;
    *(u32x2*)(d + 2 * plane) = p.l;
  };

  // ---- transposing fragment reads: lane = 16 fg + 4 fq + fp supplies row 8 fg + 4 hh + fq, columns 4 fp .. + 3 of a 16-column
  //      segment; lane 16 fg + i receives column i of those 4 rows (hh = 0 / 1: the two halves of the lane's 8 m)
  const int fg = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)lds;
  uint32_t adr_a[2][2], base_b[2];                     // [fragment][hh]: address in plane 0 of buffer 0; B: without the segment
  uint32_t swz_b[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int row = 8 * fg + 4 * hh + fq;
#pragma unroll
    for (int i = 0; i < 2; ++i) adr_a[i][hh] = lds0 + row * 512 + fp * 8 + (((wave * 2 + i) ^ tn_swz(row)) << 5);
    base_b[hh] = lds0 + 3 * PA + row * 256 + fp * 8;
    swz_b[hh] = (uint32_t)tn_swz(row);
  }
  // (the 16 B addresses are two VALU each where they are used, not 16 registers: the kernel sits at the 256-register limit)
#define TN_ADRB(J, H) (base_b[H] + ((((uint32_t)(J)) ^ swz_b[H]) << 5))
  u32x2 af[2][3][2];                                   // [i][plane][hh]
  u32x2 bf[2][3][2];                                   // [buffer of the stream][plane][hh]
#define TN_RD1(dst, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(ADDR), "n"(OFF))
#define TN_RDB(S, J, BOFF)                                      \
  TN_RD1(bf[S][0][0], TN_ADRB(J, 0) + (BOFF), 0);               \
  TN_RD1(bf[S][0][1], TN_ADRB(J, 1) + (BOFF), 0);               \
  TN_RD1(bf[S][1][0], TN_ADRB(J, 0) + (BOFF), PB);              \
  TN_RD1(bf[S][1][1], TN_ADRB(J, 1) + (BOFF), PB);              \
  TN_RD1(bf[S][2][0], TN_ADRB(J, 0) + (BOFF), 2 * PB);          \
  TN_RD1(bf[S][2][1], TN_ADRB(J, 1) + (BOFF), 2 * PB)
#define TN_RDA(I, BOFF)                                         \
  TN_RD1(af[I][0][0], adr_a[I][0] + (BOFF), 0);                 \
  TN_RD1(af[I][0][1], adr_a[I][1] + (BOFF), 0);                 \
  TN_RD1(af[I][1][0], adr_a[I][0] + (BOFF), PA);                \
  TN_RD1(af[I][1][1], adr_a[I][1] + (BOFF), PA);                \
  TN_RD1(af[I][2][0], adr_a[I][0] + (BOFF), 2 * PA);            \
  TN_RD1(af[I][2][1], adr_a[I][1] + (BOFF), 2 * PA)
#define TN_READYB(S, N)                                                                                          \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                       \
               : "+v"(bf[S][0][0]), "+v"(bf[S][0][1]), "+v"(bf[S][1][0]), "+v"(bf[S][1][1]), "+v"(bf[S][2][0]), "+v"(bf[S][2][1])::"memory")
#define TN_READYA(I)                                                                                             \
  asm volatile("" : "+v"(af[I][0][0]), "+v"(af[I][0][1]), "+v"(af[I][1][0]), "+v"(af[I][1][1]), "+v"(af[I][2][0]), "+v"(af[I][2][1]))
#define TN_TIE6(x) "+v"((x)[0][0]), "+v"((x)[0][1]), "+v"((x)[1][0]), "+v"((x)[1][1]), "+v"((x)[2][0]), "+v"((x)[2][1])
#define TN_OP(x) __builtin_bit_cast(bf16x8, u32x4{(x)[0][0], (x)[0][1], (x)[1][0], (x)[1][1]})
#define TN_MM(S, J)                                                            \
  {                                                                            \
    const bf16x8 bh = TN_OP(bf[S][0]), bm = TN_OP(bf[S][1]), bl = TN_OP(bf[S][2]); \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                            \
      const bf16x8 ah = TN_OP(af[i][0]), am = TN_OP(af[i][1]), al = TN_OP(af[i][2]);             \
      f32x4 c = acc[i][J];                                                     \
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, c, 0, 0, 0);         \
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, c, 0, 0, 0);         \
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, am, c, 0, 0, 0);         \
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, ah, c, 0, 0, 0);         \
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, am, c, 0, 0, 0);         \
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, c, 0, 0, 0);         \
      acc[i][J] = c;                                                           \
    }                                                                          \
  }
  // fragment J of the step: read J + 1, multiply J, then a slice of the side work
#define TN_J(J, BOFF, SIDE)                       \
  TN_RDB(((J) + 1) & 1, (J) + 1, BOFF);           \
  TN_READYB((J) & 1, 6);                          \
  TN_MM((J) & 1, J)                               \
  SIDE;                                           \
  __builtin_amdgcn_sched_barrier(0)
  // one step with literal parity PAR (= s & 1): multiplies buffer PAR, stashes step s + 1 (raw set PAR ^ 1) into buffer
  // PAR ^ 1, loads step s + 2 into raw set PAR.  The step's own A fragments and first B fragment are read at its top -- NOT
  // ahead, behind the previous step's barrier: behind the LAST step nothing would use such a read, and to hipcc a read whose
  // result nothing uses is a register that is free again right behind the asm statement; it put the addresses of the
  // partial-tile stores there, and the data, landing late, overwrote them (seen only with several processes sharing the GPU:
  // longer LDS latencies; a branch around the read-ahead, a peeled last step and registers tied to a later wait all made
  // hipcc spill asm outputs).  Every asm read here is consumed inside the step that issues it.
#define TN_STEP(PAR)                                                                      \
  do {                                                                                    \
    constexpr uint32_t boff = (PAR) * BUF;                                                \
    fetch(raw[PAR], s + 2);                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    TN_RDA(0, boff);                                                                      \
    TN_RDA(1, boff);                                                                      \
    TN_RDB(0, 0, boff);                                                                   \
    TN_READYA(0);                                                                         \
    TN_READYA(1);                                                                         \
    TN_J(0, boff, (void)0);                                                               \
    TN_J(1, boff, slice(raw[(PAR) ^ 1][0], 0, (PAR) ^ 1));               \
    TN_J(2, boff, slice(raw[(PAR) ^ 1][1], 1, (PAR) ^ 1));               \
    TN_J(3, boff, slice(raw[(PAR) ^ 1][2], 2, (PAR) ^ 1));               \
    TN_J(4, boff, slice(raw[(PAR) ^ 1][3], 3, (PAR) ^ 1));               \
    TN_J(5, boff, slice(raw[(PAR) ^ 1][4], 4, (PAR) ^ 1));               \
    TN_J(6, boff, slice(raw[(PAR) ^ 1][5], 5, (PAR) ^ 1));               \
    TN_READYB(1, 0);      /* fragment 7, and every LDS write of this wavefront's stash */ \
    TN_MM(1, 7)                                                                           \
    __builtin_amdgcn_s_barrier();      /* step s + 1 is in LDS; buffer PAR is free */     \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)

  for (int item = yg; item < items; item += per_group) {
  set_item(item);
  f32x4 acc[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (count > 0) {
    // ---- prologue: step 0 stashed into buffer 0 and visible, step 1 in raw set 1
    fetch(raw[0], 0);
#pragma unroll
    for (int qq = 0; qq < 6; ++qq) slice(raw[0][qq], qq, 0);
    fetch(raw[1], 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < count; s += 2) {
      TN_STEP(0);
      if (s + 1 < count) {
        ++s;
        TN_STEP(1);
        --s;
      } else {
        break;
      }
    }
  }
#undef TN_STEP
#undef TN_J
#undef TN_MM
#undef TN_OP
#undef TN_TIE6
#undef TN_READYA
#undef TN_READYB
#undef TN_RDA
#undef TN_RDB
#undef TN_RD1
#undef TN_ADRB

  // ---- the slab's partial tile: D = (B fragment) x (A fragment): lane holds k' = 4 (lane >> 4) .. + 3 of column n = lane & 15
  float* const W = g.W + (int64_t)slab * g.N * g.Kp;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = n0 + (wave * 2 + i) * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kk = k0 + j * 16 + 4 * (lane >> 4);
      if (n < g.N && kk < g.Kp) *(f32x4*)(W + (int64_t)n * g.Kp + kk) = acc[i][j];
    }
  }
  }      // items
}

// out[n][k] = the slabs' partial tiles added in slab order (64 x 16 threads: 64 consecutive float4, 16 slab groups, then the
// group sums in order: a fixed tree); += into the layer's weight .grad accumulators when a sink is given.  TR: the partials
// are [Kp][N] (the product was computed with its operands exchanged because that tiles with less padding): element (r, c)
// of a partial goes to out[c][r].
template <bool TR>
__global__ __launch_bounds__(1024) void split_tn_reduce(const float* __restrict__ W, int n_slabs, int64_t elems, int cols,
                                                        float* __restrict__ out, int64_t ldo, const GradSink sink) {
  __shared__ f32x4 s_part[16][64];
  const int main_blocks = (int)((elems / 4 + 63) / 64);
  if ((int)blockIdx.x >= main_blocks) {                        // the rider's workgroups (GradSink::cs_*)
    colsum_ride(sink, (int)blockIdx.x - main_blocks);
    return;
  }
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
    const int64_t r = e / cols, c = e - r * cols;          // cols % 4 == 0: the four values lie in one row of the partial
    if (!TR) {
      *(f32x4*)(out + r * ldo + c) = acc;
      if (sink.mode) {
        float* const d = sink_ptr(sink, r, c);
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] += acc[q];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        out[(c + q) * ldo + r] = acc[q];
        if (sink.mode) *sink_ptr(sink, c + q, r) += acc[q];
      }
    }
  }
}

}  // namespace

// ---- host side --------------------------------------------------------------------------------------------------------------
// SG_TUNE_F32_ENGINE: 0 = shipped; bit 0: the BLAS library serves every fp32 product (A/B runs); bits 1 / 2 / 3: only the
// forward / input-gradient / weight-gradient products go to the library (bisecting aid)
static int g_split_variant = 0;
int set_split_tuning(int value) {
  g_split_variant = value;
  return SG_OK;
}
// (round 6: with the 128-row variant the split kernels take a float32 product at every row count; bit 5 switches that variant
//  off -- then, as in round 5, products below kSplitNtPaysRows rows go to the BLAS library unless bit 4 is set)
// Where the library still wins (A/B at 5 K rows, profiles/r06_gemm_f32split_5k.json): a product that makes fewer than 128 of the
// 128 x 128 work items (half the CUs: one wavefront per SIMD, nothing to overlap its waits with) over a long K loop -- at 5 K
// rows [V,768]x[768,256] 0.043 ms against 0.030, [V,384]x[384,256] 0.027 against 0.025; every other shape of the SGCN is level
// (1.0-1.1 x) or ahead (1.3-1.9 x on the narrow layers).
bool split_nt_pays(int64_t M, int64_t N, int64_t K) {
  if (M >= kSplitNtPaysRows || (g_split_variant & 16)) return true;
  if (g_split_variant & 32) return false;
  const int64_t items = ((M + 127) / 128) * ((N + 127) / 128);
  return !(items < 128 && K >= 384);
}
bool split_engine_enabled(int kind) { return !(g_split_variant & 1) && !(g_split_variant & (2 << kind)); }      // kind 0 nt, 1 nn, 2 tn

static inline int split_nf(int64_t N) { return N % 256 == 0 || N > 640 ? 16 : (N % 128 == 0 || N <= 128 ? 8 : 16); }
// few rows: 128 x 128 tiles on four wavefronts (the 128-column weight image is never larger than the 256-column one, so the
// workspace size -- a function of N and K alone -- covers both)
static inline bool split_small_rows(int64_t M) { return M < kSplitNtPaysRows && !(g_split_variant & 32); }
int gemm_nt_f32s_variant(int64_t M) { return split_small_rows(M) ? 1 : 0; }

bool gemm_nt_f32s_supported(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc) {
  return M >= 128 && M < ((int64_t)1 << 31) - 512 && N >= 64 && N % 4 == 0 && K >= 64 && K % 32 == 0 && lda % 4 == 0 &&
         ldc % 4 == 0 && N < (1 << 20) && K < (1 << 20);
}

int64_t gemm_nt_f32s_workspace(int64_t N, int64_t K) {      // bytes of the split weight image
  const int nf = split_nf(N);
  const int64_t nt = nf * 16;
  return ((N + nt - 1) / nt) * (K / 32) * (3 * nf * 1024);
}

// The split image of W for a product whose tile shape is picked by `variant_rows` (the row count the image is packed for: a
// block packs once per weight update for ITS row count and runs products of V or V_ext rows on it), into `out`
// (gemm_nt_f32s_workspace(N, K) bytes).  W element (n, k) at W[n * w_rs + k * w_cs].
static int pack_split_into(const float* W, int64_t w_rs, int64_t w_cs, int64_t variant_rows, int64_t N, int64_t K, void* out,
                           hipStream_t stream, int* nf_out, bool* small_out) {
  const bool small = split_small_rows(variant_rows);
  const int nf = small ? 8 : split_nf(N);
  const int nt = nf * 16;
  PackSplit p;
  p.W = W; p.rs = w_rs; p.cs = w_cs;
  p.N = (int)N; p.K = (int)K; p.NF = nf;
  p.n_col_tiles = (int)((N + nt - 1) / nt);
  p.out = (uint8_t*)out;
  const int64_t n_frag = (int64_t)p.n_col_tiles * (K / 32) * nf;
  pack_split<<<(int)((n_frag + 3) / 4), 256, 0, stream>>>(p);
  SG_HIP_TRY(hipGetLastError());
  if (nf_out) *nf_out = nf;
  if (small_out) *small_out = small;
  return SG_OK;
}

int launch_pack_split(const float* W, int64_t w_rs, int64_t w_cs, int64_t variant_rows, int64_t N, int64_t K, void* out,
                      hipStream_t stream) {
  SG_REQUIRE(W && out && ((uintptr_t)out & 15) == 0 && N >= 64 && K >= 64 && K % 32 == 0, "pack_split: bad argument");
  return pack_split_into(W, w_rs, w_cs, variant_rows, N, K, out, stream, nullptr, nullptr);
}

// C[M, N] = A[M, K] op(W) (+ bias): W element (n, k) at W[n * w_rs + k * w_cs].  `prepacked` (nullable): the image of W that
// launch_pack_split built for `variant_rows` rows -- no packing launch, ws is not touched (round 6: a block packs when its
// weights change, i.e. every fifth iteration of the reference's loop, not in each of its two products per iteration).
int launch_gemm_nt_f32s(const float* A, int64_t lda, const float* W, int64_t w_rs, int64_t w_cs, const float* bias, float* C,
                        int64_t ldc, int64_t M, int64_t N, int64_t K, void* ws, int64_t ws_bytes, hipStream_t stream,
                        const void* prepacked, int64_t variant_rows) {
  SG_REQUIRE(gemm_nt_f32s_supported(M, N, K, lda, ldc), "sg_gemm_nt_f32: unsupported shape (M=%lld N=%lld K=%lld)", (long long)M,
             (long long)N, (long long)K);
  SG_REQUIRE((((uintptr_t)A | (uintptr_t)C | (uintptr_t)ws | (uintptr_t)prepacked) & 15) == 0, "sg_gemm_nt_f32: misaligned operand");
  bool small = false;
  int nf = 0;
  if (prepacked) {
    small = split_small_rows(variant_rows);
    nf = small ? 8 : split_nf(N);
  } else {
    SG_REQUIRE(ws && ws_bytes >= gemm_nt_f32s_workspace(N, K), "sg_gemm_nt_f32: workspace too small (%lld bytes given, %lld needed)",
               (long long)ws_bytes, (long long)gemm_nt_f32s_workspace(N, K));
    int rc = pack_split_into(W, w_rs, w_cs, M, N, K, ws, stream, &nf, &small);
    if (rc != SG_OK) return rc;
  }
  const int nt = nf * 16;
  const int rt = small ? 128 : 256;

  SplitNt g;
  g.A = A; g.lda = lda;
  g.Bp = (const uint8_t*)(prepacked ? prepacked : ws);
  g.bias = bias;
  g.C = C; g.ldc = ldc;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.n_col_tiles = (int)((N + nt - 1) / nt);
  g.n_row_tiles = (int)((M + rt - 1) / rt);
  int dev = 0, cus = 256;
  SG_HIP_TRY(hipGetDevice(&dev));
  SG_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  int streams = ((small ? 2 * cus : cus) / g.n_col_tiles) / 8 * 8;      // (the 128-row variant: two workgroups per CU)
  streams = streams < 8 ? 8 : streams;
  const int need = (g.n_row_tiles + 7) / 8 * 8;
  streams = streams > need ? need : streams;
  g.streams = streams;
  const int grid = streams * g.n_col_tiles;
  if (small) gemm_nt_f32s<8, 4><<<grid, 256, 0, stream>>>(g);
  else if (nf == 16) gemm_nt_f32s<16, 8><<<grid, kSplitThreads, 0, stream>>>(g);
  else gemm_nt_f32s<8, 8><<<grid, kSplitThreads, 0, stream>>>(g);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

bool gemm_tn_f32s_supported(int64_t M, int64_t N, int64_t Kp, int64_t lda, int64_t ldb) {
  return M >= 4096 && M < ((int64_t)1 << 31) - 64 && N >= 64 && N % 4 == 0 && Kp >= 64 && Kp % 4 == 0 && lda % 4 == 0 &&
         ldb % 4 == 0 && N < (1 << 20) && Kp < (1 << 20);
}

// slabs of rows: 8 x s (one set of s per XCD group of 32 workgroups).  s by a small cost model, in units of one 32-row step
// of one tile (~2.5 us): the items of a group (s x tiles) run in ceil(items / 32) rounds of (steps per slab + ~4 steps of
// pipeline fill and partial-tile store) each, and the slab partials cost the reduce kernel a read (~10 MB per step unit).
// A 1 M-row product of 12 tiles takes s = 8 (3 rounds of 488 steps); a 50 K-row product of ONE tile takes s = 32 (one round
// of 7 steps) where a fixed 64-step minimum per slab left it on 16 workgroups for 98 steps (0.24 ms for a 0.6 GFLOP product).
static int split_tn_slabs(int64_t M, int64_t N, int64_t Kp) {
  int64_t n_tiles = ((N + 255) / 256) * ((Kp + 127) / 128);
  const int64_t n_tiles_t = ((Kp + 255) / 256) * ((N + 127) / 128);
  if (n_tiles_t * 256 * 128 < n_tiles * 256 * 128) n_tiles = n_tiles_t;         // (the orientation launch_gemm_tn_f32s takes)
  const int64_t steps = (M + 31) / 32;
  int best = 1;
  double best_cost = 1e30;
  for (int s = 1; s <= 32; ++s) {
    const int64_t per_slab = (steps + 8 * s - 1) / (8 * s);
    if (s > 1 && (per_slab < 4 || (double)(8 * s) * N * Kp * 4 > 192e6)) break;
    const int64_t items = s * n_tiles, rounds = (items + 31) / 32;
    const double cost = (double)rounds * (per_slab + 4) + (double)(8 * s) * N * Kp * 4 / 10e6;
    if (cost < best_cost) {
      best_cost = cost;
      best = s;
    }
  }
  return 8 * best;
}

int64_t gemm_tn_f32s_workspace(int64_t M, int64_t N, int64_t Kp) { return (int64_t)split_tn_slabs(M, N, Kp) * N * Kp * 4; }

static inline int64_t split_tn_padded(int64_t N, int64_t Kp) { return ((N + 255) / 256 * 256) * ((Kp + 127) / 128 * 128); }

int launch_gemm_tn_f32s(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int64_t Kp, void* ws,
                        int64_t ws_bytes, float* out, int64_t ldo, hipStream_t stream, const GradSink* sink_) {
  SG_REQUIRE(gemm_tn_f32s_supported(M, N, Kp, lda, ldb) && ldo % 4 == 0, "sg_gemm_tn_f32: unsupported shape (M=%lld N=%lld Kp=%lld)",
             (long long)M, (long long)N, (long long)Kp);
  SG_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)out | (uintptr_t)ws) & 15) == 0, "sg_gemm_tn_f32: misaligned operand");
  SG_REQUIRE(ws && ws_bytes >= gemm_tn_f32s_workspace(M, N, Kp), "sg_gemm_tn_f32: workspace too small (%lld bytes given, %lld needed)",
             (long long)ws_bytes, (long long)gemm_tn_f32s_workspace(M, N, Kp));
  const GradSink sink = sink_ ? *sink_ : GradSink{};
  SG_REQUIRE(sink.mode == 0 || sink.Cin % 4 == 0, "sg_gemm_tn_f32: a gradient sink needs Cin to be a multiple of 4");
  // the tile is 256 (rows of the result) x 128 (columns): compute the transposed result where that pads less (384 x 256 as
  // 256 x 384: 2 x 2 tiles with a quarter empty -> 1 x 3 full ones)
  const bool tr = split_tn_padded(Kp, N) < split_tn_padded(N, Kp);
  SplitTn g;
  g.A = tr ? B : A; g.lda = tr ? ldb : lda;
  g.B = tr ? A : B; g.ldb = tr ? lda : ldb;
  g.W = (float*)ws;
  g.M = (int)M; g.N = (int)(tr ? Kp : N); g.Kp = (int)(tr ? N : Kp);
  g.tiles_k = (g.Kp + 127) / 128;
  g.n_tiles = ((g.N + 255) / 256) * g.tiles_k;
  g.slabs = split_tn_slabs(M, N, Kp);
  g.steps = (int)((M + 31) / 32);
  const int items = (g.slabs / 8) * g.n_tiles;         // per XCD group
  gemm_tn_f32s<<<8 * (items < 32 ? items : 32), kSplitThreads, 0, stream>>>(g);
  SG_HIP_TRY(hipGetLastError());
  const int64_t elems = N * Kp;
  const int blocks = (int)((elems / 4 + 63) / 64) + (sink.cs_partial ? sink.cs_C : 0);      // (+ the rider's workgroups)
  if (tr) split_tn_reduce<true><<<blocks, 1024, 0, stream>>>(g.W, g.slabs, elems, g.Kp, out, ldo, sink);
  else split_tn_reduce<false><<<blocks, 1024, 0, stream>>>(g.W, g.slabs, elems, g.Kp, out, ldo, sink);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

// out = the n_slabs partials [N, Kp] of `ws` added in the fixed tree of split_tn_reduce (+= into the sink's accumulators):
// the reduce step of the weight-gradient kernels, also used by gemm_mid.hip
int launch_split_tn_reduce(const float* ws, int n_slabs, int64_t N, int64_t Kp, float* out, int64_t ldo, const GradSink* sink,
                           hipStream_t stream) {
  SG_REQUIRE(Kp % 4 == 0 && ldo % 4 == 0, "split_tn_reduce: the column count and the row stride must be multiples of 4");
  const int64_t elems = N * Kp;
  const int blocks = (int)((elems / 4 + 63) / 64) + ((sink && sink->cs_partial) ? sink->cs_C : 0);
  split_tn_reduce<false><<<blocks, 1024, 0, stream>>>(ws, n_slabs, elems, (int)Kp, out, ldo, sink ? *sink : GradSink{});
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

}  // namespace sg
