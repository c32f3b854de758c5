// The COMPUTE-BOUND feature x weight products (the 256- and 512-channel layers of the SGCN, K x N > 100 K) on a
// 256 x 256 output tile with eight wavefronts:
//
//     C[M, N] = A[M, K] * B[N, K]^T (+ bias[N]),  bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16)
//
// Same operator as gemm_mfma.hip's sg::gemm_nt_bf16 (ChebConv's `lins[k]` [3P torch_geometric 2.2.0], call sites
// util/networks.py:42,49, and their input gradient), for the shapes where that kernel's structure -- a 128-row tile, two
// workgroup barriers and a full drain of the loads per 64-deep K step -- tops out at ~0.3 of the MFMA peak: M = V rows,
// K = 256 .. 768, N = 256 .. 768.  The CDNA4 guide's 256^2 schedule, rebuilt for SHORT K and a PERSISTENT workgroup:
//
//  * one workgroup (512 threads = 8 wavefronts, two per SIMD) per CU owns ONE column tile and walks a stream of row
//    tiles; the eight wavefronts are 2 (M) x 4 (N), each accumulating 128 x 64 outputs (128 accumulator VGPRs);
//  * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip); the LDS image is lane-linear per
//    wavefront instruction, so the bank-conflict-free XOR swizzle of the 16-byte chunks (chunk ^ (row & 7)) is applied to
//    the per-lane SOURCE address; fragments are read with ds_read_b128;
//  * a K step (64 deep) is FOUR phases; a phase = {read one register sub-tile from LDS, start ONE half-tile (128 rows x
//    64 k, 16 KB) of a later K step, counted s_waitcnt vmcnt -- never 0 --, s_barrier, 16 MFMAs (one quadrant of the
//    wavefront's tile over the whole K step), s_barrier}.  Four half-tiles (64 KB) stay in flight across the barriers;
//    a half-tile is started five phases before its first read and at least two phases after the last read of the bytes
//    it overwrites (two LDS buffers of 64 KB);
//  * wavefronts 4-7 run half a phase behind wavefronts 0-3 (one extra barrier at the start): on every SIMD one wavefront
//    issues its MFMAs while its partner reads LDS and starts the DMA;
//  * the load stream does not stop at a tile boundary: while the last K steps of one output tile are multiplied the
//    first K steps of the workgroup's NEXT row tile are already arriving, so the pipeline is filled once per launch, not
//    once per tile (K is only 4 .. 12 steps long: a per-tile prologue would cost 15 .. 40 % of the tile);
//  * the MFMA operands are swapped (D = B-fragment x A-fragment = a tile of C^T), so a lane holds FOUR CONSECUTIVE
//    COLUMNS of one row of C, eight after one v_permlane16_swap with its neighbour: the tile leaves the registers as
//    16-byte stores (64 contiguous bytes per row and instruction) with no LDS transpose and no barrier.  The
//    stores sit in the same in-order VMEM queue as the DMA, so the four phases after an epilogue wait for
//    vmcnt(8 + 16) instead of vmcnt(8): exactly 16 stores are issued per wavefront per full tile (a partial tile is the
//    last of its stream: nothing follows it).
//
// Round 6, measured and not kept (profiles/r06_gemm256_two_phase_ab.txt, r06_gemm_stream256_ab.txt, r06_pmc_gemm256.json):
// TWO phases per K step (four barriers instead of eight, 32-MFMA segments; identical bits) ran 1-2 % slower at N <= 512 and
// 12-14 % slower at N = 768; A streamed global -> VGPR as MFMA fragments with only B through an LDS ring (one barrier per K
// block) ran 1.5-1.8 x slower -- a fragment load is 64 separate 16-byte requests to the address unit.  The counters say why
// neither helps: matrix pipe busy 0.41-0.47 at 1.81-1.98 GHz here, 0.52 at 1.67 GHz in gemm_tn_256, 0.51 at 1.74 GHz in
// hipBLASLt's kernel for the same product -- busy x clock = 0.81-0.88 GHz in all three (0.34-0.37 of the nominal peak): a bf16
// product at this flop : byte mix runs into the chip's power envelope whatever the schedule; the split kernel (six MFMAs per
// operand byte moved) reaches 1.4.
//
// Column tiles of one row tile are walked by sibling workgroups on ONE XCD (block ids b, b + 8, ..) in step, so A leaves
// HBM once and is re-read through that XCD's L2.  Needs K % 64 == 0 and N % 256 == 0; everything else stays with
// sg::gemm_nt_bf16.
#include "sg_common.h"

namespace sg {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads256 = 512;
constexpr int kHalf = 16384;            // one half-tile: 128 rows x 64 bf16
constexpr int kBuf = 4 * kHalf;         // A rows 0-127 | A rows 128-255 | B rows 0-127 | B rows 128-255 of one K step
constexpr int AH0 = 0, AH1 = 1, BH0 = 2, BH1 = 3;

struct Big {
  const uint16_t* A; int64_t lda;
  const uint16_t* B; int64_t ldb;
  const float* bias;                    // nullable
  uint16_t* C; int64_t ldc;
  int M, N, K;
  int n_col_tiles, n_row_tiles, streams;
#ifdef SG_GEMM256_STAMPS
  unsigned long long* stamps;           // tools/gemm256_stamps.hip: [2 wavefronts][K steps][4 phases][6] shader-clock stamps
#endif
};

#if defined(SG_GEMM256_STAMPS) && !defined(SG_G256_NOSTAMP)
// (kept in LDS behind the staging buffers and flushed at the end: a global store per stamp would sit in the VMEM queue
//  that the counted vmcnt waits watch)
#define SG_STAMP(ph, k)                                                                                         \
  if (blockIdx.x == 8 && (wave == 0 || wave == 4) && lane == 0 && s >= 16 && s < 32)                              \
    ((unsigned long long*)(lds + 2 * kBuf))[(((wave >> 2) * 16 + (s - 16)) * 4 + (ph)) * 6 + (k)] = __builtin_readcyclecounter();
#else
#define SG_STAMP(ph, k)
#endif

#define SG_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define SG_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#ifdef SG_G256_NOBARRIER
#define SG_BARRIER()
#else
#define SG_BARRIER() __builtin_amdgcn_s_barrier()
#endif
#ifdef SG_G256_NTSTORE
#define SG_STORE16(p, v) __builtin_nontemporal_store(v, (u32x4*)(p))
#else
#define SG_STORE16(p, v) (*(u32x4*)(p) = (v))
#endif
#ifdef SG_G256_ONEBAR
#define SG_BARRIER2()
#else
#define SG_BARRIER2() SG_BARRIER()
#endif
#ifdef SG_G256_NOREAD
#define SG_LDS_FRAG(ptr) sg_fake_frag(ptr)
#else
#define SG_LDS_FRAG(ptr) (*(const bf16x8*)(ptr))
#endif
#ifdef SG_G256_NOMFMA
#define SG_MFMA(a, b, c, x, y, z) ((c) + f32x4{(float)(a)[0], (float)(b)[1], 0.f, 0.f})
#else
#define SG_MFMA(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z)
#endif
#ifdef SG_G256_NOPRIO
#define SG_PRIO(p)
#else
#define SG_PRIO(p) __builtin_amdgcn_s_setprio(p)
#endif

[[maybe_unused]] __device__ __forceinline__ bf16x8 sg_fake_frag(const uint8_t* p) {
  bf16x8 v;
  asm volatile("" : "=v"(v) : "v"(p));
  return v;
}

__device__ __forceinline__ void glds16(const void* src, uint8_t* lds_dst) {
#ifdef SG_G256_NOLOAD
  asm volatile("" ::"v"(src), "s"(lds_dst));
  return;
#endif
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                   (void __attribute__((address_space(3)))*)lds_dst, 16, 0, 0);
}

__global__ __launch_bounds__(kThreads256, 1) void gemm_nt_256(const Big g) {
#ifdef SG_GEMM256_STAMPS
  __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * kBuf + 2 * 16 * 4 * 6 * 8];
#else
  __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * kBuf];      // the ONLY LDS object (guide 5, trap 4a)
#endif

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;

  // ---- which tiles: column tile fixed, row tiles stream, stream + streams, .. ---------------------------------------
  const int b = blockIdx.x;
  const int slot = b >> 3;
  const int ct = slot % g.n_col_tiles;
  const int stream = (b & 7) + 8 * (slot / g.n_col_tiles);
  const int my_tiles = stream < g.n_row_tiles ? (g.n_row_tiles - stream + g.streams - 1) / g.streams : 0;
  if (my_tiles == 0) return;
  const int col0 = ct * 256;
  const int nk = g.K >> 6;
  const int total = my_tiles * nk;                     // K steps of this workgroup's whole stream

  // ---- LDS-DMA: a half-tile is 16 wavefront instructions of 1 KB (8 rows x 128 B); wavefront w issues blocks 2w, 2w+1.
  //      lane -> row (lane >> 3) of the block, LDS slot (lane & 7); the slot holds source chunk slot ^ (row & 7).
  const int s_row = lane >> 3, s_chunk = (lane & 7) ^ (lane >> 3);
  int r_half[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) r_half[i] = 16 * wave + 8 * i + s_row;        // row of the half-tile
  uint32_t offB[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) offB[h][i] = (uint32_t)((h * 128 + r_half[i]) * g.ldb * 2 + s_chunk * 16);
  const char* const Bw = (const char*)g.B + (int64_t)col0 * g.ldb * 2;

  // the load cursor: K step `ls` of the stream (tile lt, step lk), clamped to the stream's last step
  int lt = 0, lk = 0;
  uint32_t offA[2][2];                                  // of the cursor's tile (rows past M are clamped to M - 1)
  const char* a_base;                                   // A + row0 * lda + lk * 64 (bytes)
  auto set_tile = [&](int t) {
    const int row0 = (stream + t * g.streams) * 256;
    a_base = (const char*)g.A + (int64_t)row0 * g.lda * 2;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int r = h * 128 + r_half[i];
        const int lim = g.M - 1 - row0;
        r = r < lim ? r : lim;
        offA[h][i] = (uint32_t)(r * g.lda * 2 + s_chunk * 16);
      }
  };
  auto advance = [&]() {
    if (lk + 1 < nk) {
      ++lk;
    } else if (lt + 1 < my_tiles) {
      ++lt;
      lk = 0;
      set_tile(lt);
    }                                                   // else: stay on the last step (harmless re-load)
  };
  // half-tile loads of the cursor's K step into buffer `buf`
  auto load_a = [&](int h, int buf) {
    uint8_t* dst = lds + buf * kBuf + h * kHalf + wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(a_base + lk * 128 + offA[h][i], dst + i * 1024);
  };
  auto load_b = [&](int h, int buf) {
    uint8_t* dst = lds + buf * kBuf + (2 + h) * kHalf + wave * 2048;
    const char* const b_base = Bw + lk * 128;        // (a scalar base + a 32-bit lane offset: the saddr form of the DMA)
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(b_base + offB[h][i], dst + i * 1024);
  };

  // ---- fragment read addresses -------------------------------------------------------------------------------------
  // wavefront (wr, wc): rows wr*64 .. +64 of BOTH A halves, rows (= columns of C) wc*32 .. +32 of BOTH B halves
  int a_rd[2], b_rd[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int sw = (((ks << 2) | fq) ^ (fr & 7)) << 4;
    a_rd[ks] = (wr * 64 + fr) * 128 + sw;
    b_rd[ks] = (wc * 32 + fr) * 128 + sw;
  }

  float bias_r[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bias_r[j][q] = g.bias ? g.bias[col0 + (j >> 1) * 128 + wc * 32 + (j & 1) * 16 + fq * 4 + q] : 0.f;
  // the bias loads must have RETURNED before the first LDS-DMA is issued: hipcc places the wait for an ordinary load at
  // its first use, which would be the epilogue inside the main loop -- a vmcnt(0) there drains the DMA pipeline
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(bias_r[j][q]));

  f32x4 bias_v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bias_v[j] = f32x4{bias_r[j][0], bias_r[j][1], bias_r[j][2], bias_r[j][3]};
  f32x4 acc[8][4];                                      // start from the bias: the epilogue is convert + store only
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = bias_v[j];

  // ---- prologue: K step 0 whole, A-h0 / B-h0 of K step 1 --------------------------------------------------------------
  set_tile(0);
  load_a(0, 0);
  load_b(0, 0);
  load_b(1, 0);
  load_a(1, 0);
  advance();                                            // cursor = K step 1
  load_a(0, 1);
  load_b(0, 1);
  // cursor convention from here on: at the top of K step s the cursor is at step s + 1; after phase 1 it moves to s + 2
  SG_WAIT_VM(8);                                        // A-h0, B-h0 of step 0 have landed (this wavefront's share)
  SG_BARRIER();
  if (wr == 1) SG_BARRIER2();            // wavefronts 4-7 run half a phase behind

  // ---- epilogue, one quadrant of the wavefront's tile at a time -------------------------------------------------------------
  // Lane (row fr, group fq) holds columns 4 fq .. + 3 of each 16-column fragment; v_permlane16_swap trades the odd 16-lane
  // rows of one register with the even rows of another, so fragments j and j + 1 swap halves between lane groups (0, 1)
  // and (2, 3) and a lane owns EIGHT consecutive columns: 16-byte stores, 64 contiguous bytes per row and instruction.  The
  // stores are request-bound (8-byte stores -- twice the write requests -- cost 10-20 % of the whole product).  Tried and
  // not kept: spreading the four quadrants over the four phases of the next K step (the stores then sit between the DMA
  // loads in the in-order VMEM queue and hold them up: +8 .. +20 % time), non-temporal stores (+5 .. +20 %).
  int since = 2;                                        // K steps since a tile was stored (0: the step right after), capped at 2
  int pend_row0 = 0;                                    // first row of the tile being stored
  auto store_quadrant = [&](int q) {
    const int i0 = q >= 2 ? 4 : 0, jp = (q == 1 || q == 2) ? 1 : 0;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const int i = i0 + ii;
      const int row = pend_row0 + (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + fr;
      uint16_t* const cptr = g.C + (int64_t)row * g.ldc + col0 + wc * 32 + (fq & 1) * 16 + (fq >> 1) * 8 + jp * 128;
      uint32_t lo[2], hi[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * jp + jj;
        lo[jj] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{acc[i][j][0], acc[i][j][1]}, bf16x2));
        hi[jj] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{acc[i][j][2], acc[i][j][3]}, bf16x2));
        acc[i][j] = bias_v[j];
      }
      const auto ra = __builtin_amdgcn_permlane16_swap(lo[0], lo[1], false, false);
      const auto rb = __builtin_amdgcn_permlane16_swap(hi[0], hi[1], false, false);
      if (row < g.M) SG_STORE16(cptr, (u32x4{ra[0], rb[0], ra[1], rb[1]}));
    }
  };
  // counted waits: 8 = four half-tiles of DMA younger than the one needed next; the 16 stores of a tile sit in the same
  // in-order queue, which raises the count to 24 for the four phases of the K step after them
#define SG_PHASE_WAIT(n0, n1)                 \
  if (since == 0) SG_WAIT_VM(n0);             \
  else SG_WAIT_VM(n1);

  bf16x8 af[4][2], bf0[2][2], bf1[2][2];
  int ct_k = 0, ct_t = 0;                               // compute position: K step ct_k of tile ct_t
  for (int s = 0; s < total; ++s) {
    const int buf = s & 1;
    const uint8_t* const base = lds + buf * kBuf;
    // ---------------- phase 0: A0 x B0 ----------------------------------------------------------------------------------
    SG_STAMP(0, 0)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bf0[j][ks] = SG_LDS_FRAG(base + BH0 * kHalf + b_rd[ks] + j * 2048);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[i][ks] = SG_LDS_FRAG(base + AH0 * kHalf + a_rd[ks] + i * 2048);
    load_b(1, buf ^ 1);                                 // B-h1 of step s + 1
    SG_STAMP(0, 1)
    SG_PHASE_WAIT(24, 8)
    SG_STAMP(0, 2)
    SG_BARRIER();
    SG_STAMP(0, 3)
    SG_WAIT_LGKM0();
    SG_PRIO(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = SG_MFMA(bf0[j][ks], af[i][ks], acc[i][j], 0, 0, 0);
    SG_PRIO(0);
    SG_STAMP(0, 4)
    SG_BARRIER2();
    SG_STAMP(0, 5)
    // ---------------- phase 1: A0 x B1 ----------------------------------------------------------------------------------
    SG_STAMP(1, 0)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bf1[j][ks] = SG_LDS_FRAG(base + BH1 * kHalf + b_rd[ks] + j * 2048);
    load_a(1, buf ^ 1);                                 // A-h1 of step s + 1
    advance();                                          // cursor = step s + 2
    SG_STAMP(1, 1)
    SG_PHASE_WAIT(24, 8)
    SG_STAMP(1, 2)
    SG_BARRIER();
    SG_STAMP(1, 3)
    SG_WAIT_LGKM0();
    SG_PRIO(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][2 + j] = SG_MFMA(bf1[j][ks], af[i][ks], acc[i][2 + j], 0, 0, 0);
    SG_PRIO(0);
    SG_STAMP(1, 4)
    SG_BARRIER2();
    SG_STAMP(1, 5)
    // ---------------- phase 2: A1 x B1 ----------------------------------------------------------------------------------
    SG_STAMP(2, 0)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[i][ks] = SG_LDS_FRAG(base + AH1 * kHalf + a_rd[ks] + i * 2048);
    load_a(0, buf);                                     // A-h0 of step s + 2 (this buffer's A-h0 was last read in phase 0)
    SG_STAMP(2, 1)
    SG_PHASE_WAIT(24, 8)
    SG_STAMP(2, 2)
    SG_BARRIER();
    SG_STAMP(2, 3)
    SG_WAIT_LGKM0();
    SG_PRIO(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[4 + i][2 + j] = SG_MFMA(bf1[j][ks], af[i][ks], acc[4 + i][2 + j], 0, 0, 0);
    SG_PRIO(0);
    SG_STAMP(2, 4)
    SG_BARRIER2();
    SG_STAMP(2, 5)
    // ---------------- phase 3: A1 x B0 (both still in registers) --------------------------------------------------------
    SG_STAMP(3, 0)
    load_b(0, buf);                                     // B-h0 of step s + 2
    SG_STAMP(3, 1)
    SG_PHASE_WAIT(24, 8)
    SG_STAMP(3, 2)
    SG_BARRIER();
    SG_STAMP(3, 3)
    SG_PRIO(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[4 + i][j] = SG_MFMA(bf0[j][ks], af[i][ks], acc[4 + i][j], 0, 0, 0);
    SG_PRIO(0);
    SG_STAMP(3, 4)
    SG_BARRIER2();
    SG_STAMP(3, 5)

    // ---------------- end of a tile: convert, store (16 stores per wavefront), restart the sums from the bias ---------------
    since = since < 2 ? since + 1 : 2;
    if (++ct_k == nk) {
      pend_row0 = (stream + ct_t * g.streams) * 256;
#pragma unroll
      for (int q = 0; q < 4; ++q) store_quadrant(q);
      since = 0;
      ct_k = 0;
      ++ct_t;
    }
  }
  if (wr == 0) SG_BARRIER2();            // pairs with the extra barrier of wavefronts 4-7
  SG_WAIT_VM(0);                                        // no LDS-DMA may be in flight when the workgroup ends
#ifdef SG_GEMM256_STAMPS
  __syncthreads();
  if (blockIdx.x == 8)
    for (int i = tid; i < 2 * 16 * 4 * 6; i += kThreads256) g.stamps[i] = ((unsigned long long*)(lds + 2 * kBuf))[i];
#endif
}


// =====================================================================================================================
// Weight gradient of the compute-bound layers:  W[N, Kp] = A[M, N]^T * B[M, Kp]  (dOut^T [Tx0|Tx1|Tx2]; autograd of the
// `lins[k]` calls, util/networks.py:42,49) on the same eight-wavefront ring.  The reduction index m is the ROW of both
// operands, so a K step is 64 ROWS of A and B (two 64 x 256 tiles = 4 half-tiles of 64 rows x 128 columns, 16 KB each) and
// the MFMA fragments are columns of the LDS tiles, read with ds_read_b64_tr_b16 (as sg::gemm_tn_bf16: 32-byte segments
// XOR-swizzled by the row so that the eight rows a 32-lane half reads hit disjoint banks -- here the swizzle is applied
// to the per-lane SOURCE address of the DMA).  One workgroup = one 256 x 256 tile of W for one SLAB of rows, walked with
// the pipeline of gemm_nt_256 (four phases per K step, counted vmcnt(8), staggered halves); there are no stores inside
// the loop.  Slab partials go to the fp32 workspace and are summed in slab order by sg::tn_reduce: deterministic.
struct BigTn {
  const uint16_t* A; int64_t lda;
  const uint16_t* B; int64_t ldb;
  float* W;                                // [slabs][N][Kp]
  int N, Kp;
  int tiles_k, n_tiles, slabs, steps;      // steps = full 64-row steps of the whole product
};

__device__ __forceinline__ int tn_swz256(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

__global__ __launch_bounds__(kThreads256, 1) void gemm_tn_256(const BigTn g) {
  __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * kBuf];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int b = blockIdx.x;
  const int slot = b >> 3;
  const int tile = slot % g.n_tiles;
  const int slab = (b & 7) + 8 * (slot / g.n_tiles);
  const int n0 = (tile / g.tiles_k) * 256, k0 = (tile % g.tiles_k) * 256;
  const int q = g.steps / g.slabs, rem = g.steps % g.slabs;
  const int first = slab * q + (slab < rem ? slab : rem);
  const int total = q + (slab < rem ? 1 : 0);
  float* const W = g.W + (int64_t)slab * g.N * g.Kp;

  // ---- LDS-DMA: a half-tile = 64 rows x 256 B = 16 instructions of 1 KB (4 rows); wavefront w issues blocks 2w, 2w + 1
  const int s_slot = lane & 15;
  uint32_t offA[2], offB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * wave + 4 * i + (lane >> 4);
    const int src_chunk = (((s_slot >> 1) ^ tn_swz256(row)) << 1) | (s_slot & 1);
    offA[i] = (uint32_t)(row * g.lda * 2 + src_chunk * 16);
    offB[i] = (uint32_t)(row * g.ldb * 2 + src_chunk * 16);
  }
  const char* const Abase = (const char*)g.A + (int64_t)n0 * 2;
  const char* const Bbase = (const char*)g.B + (int64_t)k0 * 2;
  int ls = 0;                                           // load cursor: step of this slab, clamped to the last one
  auto a_ptr = [&]() { return Abase + (int64_t)(first + ls) * 64 * g.lda * 2; };
  auto b_ptr = [&]() { return Bbase + (int64_t)(first + ls) * 64 * g.ldb * 2; };
  auto advance = [&]() { if (ls + 1 < total) ++ls; };
  auto load_a = [&](int h, int buf) {
    uint8_t* dst = lds + buf * kBuf + h * kHalf + wave * 2048;
    const char* src = a_ptr() + h * 256;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(src + offA[i], dst + i * 1024);
  };
  auto load_b = [&](int h, int buf) {
    uint8_t* dst = lds + buf * kBuf + (2 + h) * kHalf + wave * 2048;
    const char* src = b_ptr() + h * 256;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(src + offB[i], dst + i * 1024);
  };

  // ---- transposing fragment reads (see sg::gemm_tn_bf16): lane = 16 fg + 4 fq + fp supplies row 32 ms + 8 fg + 4 hh + fq,
  //      columns 4 fp .. + 3 of a 16-column segment; lane 16 fg + i receives column i of those 4 rows.
  //      Issued as inline asm: in front of the ds_read_tr BUILTIN hipcc puts an `s_waitcnt vmcnt(0)` in every phase (it
  //      cannot tell the read from the LDS-DMA in flight), which drains the ring; the asm form is invisible to that pass,
  //      so the lgkmcnt wait that makes the fragments valid is written by hand and TIED to them (SG_TN_READY).
  const int fg = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)lds;
  uint32_t frow[2][2];                                  // byte offset of row (ms, hh) + fp * 8, and its swizzle
  int fsw[2][2];
#pragma unroll
  for (int ms = 0; ms < 2; ++ms)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int row = 32 * ms + 8 * fg + 4 * hh + fq;
      frow[ms][hh] = (uint32_t)(row * 256 + fp * 8);
      fsw[ms][hh] = tn_swz256(row);
    }
  auto frag = [&](uint32_t half_off, int seg, int ms) -> bf16x8 {
    bf16x4 h[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const uint32_t addr = lds0 + half_off + frow[ms][hh] + (uint32_t)((seg ^ fsw[ms][hh]) << 5);
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(h[hh]) : "v"(addr) : "memory");
    }
    return bf16x8{h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
  };
#define SG_TN_READY4(a, b, c, d) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory")

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (total > 0) {
    load_a(0, 0);
    load_b(0, 0);
    load_b(1, 0);
    load_a(1, 0);
    advance();
    load_a(0, 1);
    load_b(0, 1);
    SG_WAIT_VM(8);
    SG_BARRIER();
    if (wr == 1) SG_BARRIER2();

    bf16x8 af[4][2], bf0[2][2], bf1[2][2];
    for (int s = 0; s < total; ++s) {
      const int buf = s & 1;
      const uint32_t boff = (uint32_t)(buf * kBuf);
      // phase 0: A0 x B0
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) bf0[j][ms] = frag(boff + BH0 * kHalf, wc * 2 + j, ms);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) af[i][ms] = frag(boff + AH0 * kHalf, wr * 4 + i, ms);
      load_b(1, buf ^ 1);
      SG_WAIT_VM(8);
      SG_BARRIER();
      SG_TN_READY4(bf0[0][0], bf0[0][1], bf0[1][0], bf0[1][1]);
      SG_TN_READY4(af[0][0], af[0][1], af[1][0], af[1][1]);
      SG_TN_READY4(af[2][0], af[2][1], af[3][0], af[3][1]);
      SG_PRIO(1);
#pragma unroll
      for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = SG_MFMA(af[i][ms], bf0[j][ms], acc[i][j], 0, 0, 0);
      SG_PRIO(0);
      SG_BARRIER2();
      // phase 1: A0 x B1
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) bf1[j][ms] = frag(boff + BH1 * kHalf, wc * 2 + j, ms);
      load_a(1, buf ^ 1);
      advance();
      SG_WAIT_VM(8);
      SG_BARRIER();
      SG_TN_READY4(bf1[0][0], bf1[0][1], bf1[1][0], bf1[1][1]);
      SG_PRIO(1);
#pragma unroll
      for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][2 + j] = SG_MFMA(af[i][ms], bf1[j][ms], acc[i][2 + j], 0, 0, 0);
      SG_PRIO(0);
      SG_BARRIER2();
      // phase 2: A1 x B1
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) af[i][ms] = frag(boff + AH1 * kHalf, wr * 4 + i, ms);
      load_a(0, buf);
      SG_WAIT_VM(8);
      SG_BARRIER();
      SG_TN_READY4(af[0][0], af[0][1], af[1][0], af[1][1]);
      SG_TN_READY4(af[2][0], af[2][1], af[3][0], af[3][1]);
      SG_PRIO(1);
#pragma unroll
      for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = SG_MFMA(af[i][ms], bf1[j][ms], acc[4 + i][2 + j], 0, 0, 0);
      SG_PRIO(0);
      SG_BARRIER2();
      // phase 3: A1 x B0
      load_b(0, buf);
      SG_WAIT_VM(8);
      SG_BARRIER();
      SG_PRIO(1);
#pragma unroll
      for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[4 + i][j] = SG_MFMA(af[i][ms], bf0[j][ms], acc[4 + i][j], 0, 0, 0);
      SG_PRIO(0);
      SG_BARRIER2();
    }
    if (wr == 0) SG_BARRIER2();
    SG_WAIT_VM(0);
  }
  // ---- the slab's partial tile: D row = n (4 (lane >> 4) + reg), column = k' (lane & 15) --------------------------------
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = k0 + (j >> 1) * 128 + wc * 32 + (j & 1) * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + fg * 4 + r;
        W[(int64_t)n * g.Kp + kk] = acc[i][j][r];
      }
    }
}

}  // namespace

bool gemm_nt_256_supported(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc) {
  return K >= 128 && K % 64 == 0 && N >= 256 && N % 256 == 0 && M >= 256 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 &&
         256 * lda * 2 < (int64_t)1 << 31 && 256 * ldb * 2 < (int64_t)1 << 31;
}

int launch_gemm_nt_256(const void* A, int64_t lda, const void* B, int64_t ldb, const float* bias, void* C, int64_t ldc,
                       int64_t M, int64_t N, int64_t K, hipStream_t stream) {
  SG_REQUIRE(gemm_nt_256_supported(M, N, K, lda, ldb, ldc), "sg_gemm_nt (256-tile kernel): unsupported shape");
  SG_REQUIRE((((uintptr_t)A | (uintptr_t)B) & 15) == 0 && ((uintptr_t)C & 15) == 0, "sg_gemm_nt (256-tile kernel): misaligned operand");
  Big g;
  g.A = (const uint16_t*)A; g.lda = lda;
  g.B = (const uint16_t*)B; g.ldb = ldb;
  g.bias = bias;
  g.C = (uint16_t*)C; g.ldc = ldc;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.n_col_tiles = (int)(N / 256);
  g.n_row_tiles = (int)((M + 255) / 256);
  int dev = 0, cus = 256;
  SG_HIP_TRY(hipGetDevice(&dev));
  SG_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  // one workgroup per CU: streams = the multiple of 8 with streams * n_col_tiles <= CUs, never more than the row tiles need
  int streams = (cus / g.n_col_tiles) / 8 * 8;
  streams = streams < 8 ? 8 : streams;
  const int need = (g.n_row_tiles + 7) / 8 * 8;
  streams = streams > need ? need : streams;
  g.streams = streams;
  gemm_nt_256<<<streams * g.n_col_tiles, kThreads256, 0, stream>>>(g);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

bool gemm_tn_256_supported(int64_t M, int64_t N, int64_t Kp, int64_t lda, int64_t ldb) {
  return M >= 64 * 64 && N >= 256 && N % 256 == 0 && Kp >= 256 && Kp % 256 == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
         64 * lda * 2 < (int64_t)1 << 31 && 64 * ldb * 2 < (int64_t)1 << 31;
}

// slabs the 256 x 256 kernel cuts the FULL 64-row steps of M into (one workgroup per CU: slabs x tiles <= CUs, slabs a
// multiple of 8 so that the tiles of one slab share an XCD)
int gemm_tn_256_slabs(int64_t M, int64_t N, int64_t Kp) {
  const int n_tiles = (int)((N / 256) * (Kp / 256));
  int slabs = (256 / n_tiles) / 8 * 8;
  slabs = slabs < 8 ? 8 : slabs;
  const int64_t steps = M / 64;
  while (slabs > 8 && steps < slabs) slabs -= 8;
  return slabs;
}

int launch_gemm_tn_256(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t Kp,
                       float* workspace, hipStream_t stream) {
  SG_REQUIRE(gemm_tn_256_supported(M, N, Kp, lda, ldb), "sg_gemm_tn (256-tile kernel): unsupported shape");
  BigTn g;
  g.A = (const uint16_t*)A; g.lda = lda;
  g.B = (const uint16_t*)B; g.ldb = ldb;
  g.W = workspace;
  g.N = (int)N; g.Kp = (int)Kp;
  g.tiles_k = (int)(Kp / 256);
  g.n_tiles = (int)((N / 256) * g.tiles_k);
  g.slabs = gemm_tn_256_slabs(M, N, Kp);
  g.steps = (int)(M / 64);
  gemm_tn_256<<<g.slabs * g.n_tiles, kThreads256, 0, stream>>>(g);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

}  // namespace sg
