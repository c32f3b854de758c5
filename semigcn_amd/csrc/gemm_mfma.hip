// Dense per-vertex feature x weight product on the gfx950 matrix cores:
//
//     C[M, N] = A[M, K] * B[N, K]^T (+ bias[N]),  bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16)
//
// Replaces the three `lins[k](Tx_k)` of ChebConv.forward [3P torch_geometric 2.2.0] (reached from
// util/networks.py:42,49 and util/meshnet.py:40-240) -- A = [Tx0|Tx1|Tx2] (one [V, 3 Cin] buffer written by the
// aggregation kernel), B = [W0|W1|W2] -- and, with the transposed weights as B, their input gradient
// dT = dOut * Wcat.  M is the vertex count (1 M .. 4 M), N and K are channel counts (16 .. 768): the product is a
// STREAM over A and C with a small L2-resident B, so for most layers the bound is HBM, not the MFMA rate
// (DESIGN.md section 3.3 has the per-layer byte / flop counts).
//
// Structure (CDNA4 guide, section 5: the 128-row tile, two barriers per K step, ~3 workgroups per CU):
//  * one workgroup = 4 wavefronts = one 128 x BN tile of C (BN = 16/32/64/128 by N); K is walked in steps of
//    64 bf16 = 128-byte rows of the LDS tiles;
//  * global -> registers -> LDS staging (16 B per lane, whole 128-B lines per 8 lanes): the loads of step t+1 are
//    issued BEFORE the MFMAs of step t and written to LDS after them, so HBM latency hides behind the matrix work
//    and behind the other workgroups of the CU; register staging (not LDS-DMA) because the K tail and the M / N
//    edges are zero-filled / clamped per lane;
//  * LDS rows are XOR-swizzled in 16-B chunks (chunk ^ (row & 7)): the ds_read_b128 fragment reads (16 rows x one
//    chunk per 16 lanes) and the ds_write_b128 fills are both bank-conflict-free;
//  * epilogue: accumulators (+ bias) are rounded to bf16 into an LDS image of the C tile and leave as 16-byte row
//    segments (the MFMA result layout holds one COLUMN per lane -- stored directly it would write 2-byte pieces);
//    optionally the tile's per-column (mean, M2) go out beside it, so BatchNorm needs no separate moments pass;
//  * workgroups that share an XCD (blockIdx % 8) take consecutive tiles, column tile fastest: the column tiles of
//    one row tile run side by side on one L2, so A leaves HBM once.
#include "sg_common.h"

namespace sg {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA A/B fragment: 8 consecutive k (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kBK = 64;           // bf16 per K step
constexpr int kRowBytes = 128;    // kBK * 2

struct GemmArgs {
  const uint16_t* A; int64_t lda;
  const uint16_t* B; int64_t ldb;
  const float* bias;              // nullable
  uint16_t* C; int64_t ldc;
  float* moments;                 // nullable: [row tiles][2][N] = per-tile column (mean, M2) of the ROUNDED output
  int M, N, K;
  int n_col_tiles, n_tiles;
  int a_plog; int64_t a_pstride;  // A / C kept as planes of 2^plog columns (sg_common.h Planes; 31: plain row-major)
  int c_plog; int64_t c_pstride;
};

__device__ __forceinline__ int xcd_run(int b, int nblocks) {   // blocks b, b+8, .. (one XCD) get one contiguous run
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = b & 7, slot = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

template <int WAVES_M, int WAVES_N, int MT, int NT, bool MOMENTS>
__global__ __launch_bounds__(kThreads, (WAVES_N * NT > 8 && WAVES_M * MT >= 8) ? 2 : 3) void gemm_nt_bf16(const GemmArgs g) {
  static_assert(WAVES_M * WAVES_N == 4, "four wavefronts per workgroup");
  constexpr int BM = WAVES_M * MT * 16, BN = WAVES_N * NT * 16;
  constexpr int A_BYTES = BM * kRowBytes, B_BYTES = BN * kRowBytes;
  constexpr int C_PITCH = BN * 2 + 16;                         // padded: the column-per-lane writes spread over banks
  constexpr int C_BYTES = BM * C_PITCH;
  constexpr int RED_BYTES = MOMENTS ? (WAVES_M * BN * 2 + 4) * 4 : 0;   // per-wavefront column (mean, M2) + row counts
  constexpr int LDS_BYTES = (A_BYTES + B_BYTES > C_BYTES ? A_BYTES + B_BYTES : C_BYTES) + RED_BYTES;
  constexpr int A_CH = BM * 8 / kThreads;                      // 16-B chunks of the A tile per thread (4)
  constexpr int B_CH = (BN * 8 + kThreads - 1) / kThreads;     // of the B tile (4 / 2 / 1 / 1)
  __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_BYTES];
  uint8_t* const ldsA = lds;
  uint8_t* const ldsB = lds + A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int tile = xcd_run(blockIdx.x, g.n_tiles);
  const int row0 = (tile / g.n_col_tiles) * BM;
  const int col0 = (tile % g.n_col_tiles) * BN;

  // ---- staging addresses: chunk q = tid + 256 i -> tile row q / 8, 16-B chunk q % 8 -------------------------------
  const int s_c = tid & 7;
  const uint16_t* a_src[A_CH];
  int a_dst[A_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int r = (tid >> 3) + 32 * i;
    int gr = row0 + r;
    gr = gr < g.M ? gr : g.M - 1;                              // clamped: in bounds, masked at the store
    a_src[i] = g.A + (int64_t)gr * g.lda;
    a_dst[i] = r * kRowBytes + ((s_c ^ (r & 7)) << 4);
  }
  const uint16_t* b_src[B_CH];
  int b_dst[B_CH];
  bool b_on[B_CH];
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {
    const int r = (tid >> 3) + 32 * i;
    b_on[i] = r < BN;
    int gr = col0 + r;
    gr = gr < g.N ? gr : g.N - 1;
    b_src[i] = g.B + (int64_t)gr * g.ldb;
    b_dst[i] = (b_on[i] ? r : 0) * kRowBytes + ((s_c ^ (r & 7)) << 4);
  }

  // Loads are UNCONDITIONAL (a chunk past K re-reads the start of its row and is zeroed on its way to LDS): hipcc
  // puts every load that sits behind a branch into its own basic block, which breaks up the batch of loads in flight.
  u32x4 ra[A_CH], rb[B_CH];
  bool kin = true;                                             // does the fetched chunk lie inside K?
  auto fetch = [&](int k0) {                                   // global -> registers
    kin = k0 + s_c * 8 < g.K;                                  // K % 8 == 0: a chunk is wholly inside or outside
    const int koff = kin ? k0 + s_c * 8 : 0;
    const int64_t aoff = plane_off(koff, g.a_plog, g.a_pstride);   // (a 16-byte chunk never straddles a plane: 2^plog % 8 == 0)
#pragma unroll
    for (int i = 0; i < A_CH; ++i) ra[i] = *(const u32x4*)(a_src[i] + aoff);
#pragma unroll
    for (int i = 0; i < B_CH; ++i) rb[i] = *(const u32x4*)(b_src[i] + koff);
  };
  auto stash = [&]() {                                         // registers -> LDS, zero past K
    const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < A_CH; ++i) *(u32x4*)(ldsA + a_dst[i]) = kin ? ra[i] : zero;
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      if (b_on[i]) *(u32x4*)(ldsB + b_dst[i]) = kin ? rb[i] : zero;
  };

  // ---- fragment addresses: lane -> (row = lane & 15, chunk = lane >> 4 (+4 for the second half of the K step)) ---
  const int f_r = lane & 15, f_c = lane >> 4;
  int a_off[MT], b_off[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) a_off[i] = ((wm * MT + i) * 16 + f_r) * kRowBytes;
#pragma unroll
  for (int j = 0; j < NT; ++j) b_off[j] = ((wn * NT + j) * 16 + f_r) * kRowBytes;
  const int sw = f_r & 7;                                      // (row & 7): tile bases are multiples of 16

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (g.K + kBK - 1) / kBK;
  fetch(0);
  stash();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) fetch((kt + 1) * kBK);                    // in flight during the MFMAs below
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = (((ks << 2) | f_c) ^ sw) << 4;
      bf16x8 af[MT], bfr[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *(const bf16x8*)(ldsA + a_off[i] + ch);
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[j] = *(const bf16x8*)(ldsB + b_off[j] + ch);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();                                           // every wavefront is done reading this step's tiles
    if (kt + 1 < nk) {
      stash();
      __syncthreads();
    }
  }

  // ---- epilogue: (+ bias) -> bf16 -> LDS image of the C tile [BM][BN] (pitch C_PITCH) ------------------------------
  // accumulator layout of the 16x16 MFMA: column = lane & 15, rows = 4 (lane >> 4) + reg
  uint8_t* const ldsC = lds;
  int rows_valid = g.M - row0;
  rows_valid = rows_valid > BM ? BM : rows_valid;
  // BatchNorm tile moments, taken in REGISTERS from the rounded values: every lane holds 4 MT rows of NT columns;
  // per lane a two-pass (mean, M2) -- nothing cancels --, then Chan's merge over the 4 lane groups that share a
  // column (lanes l, l^16, l^32, l^48) and, through a few LDS words, over the WAVES_M wavefronts stacked on it.
  float cnt = 0.f;                                              // valid rows held by this lane (same for every column)
  float mom_mean[NT], mom_m2[NT];
  if (MOMENTS) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) cnt += ((wm * MT + i) * 16 + f_c * 4 + q) < rows_valid ? 1.f : 0.f;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int cl = (wn * NT + j) * 16 + f_r;
    const int gc = col0 + cl;
    const float bv = (g.bias != nullptr && gc < g.N) ? g.bias[gc] : 0.f;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int rl = (wm * MT + i) * 16 + f_c * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint16_t h = __builtin_bit_cast(uint16_t, (__bf16)(acc[i][j][q] + bv));
        *(uint16_t*)(ldsC + (rl + q) * C_PITCH + cl * 2) = h;
        if (MOMENTS) {
          const float v = __uint_as_float((uint32_t)h << 16);
          acc[i][j][q] = v;                                     // keep the rounded value for the second pass
          sum += (rl + q) < rows_valid ? v : 0.f;
        }
      }
    }
    if (MOMENTS) {
      const float mean = cnt > 0.f ? sum / cnt : 0.f;
      float m2 = 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int rl = (wm * MT + i) * 16 + f_c * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float d = (rl + q) < rows_valid ? acc[i][j][q] - mean : 0.f;
          m2 = fmaf(d, d, m2);
        }
      }
      mom_mean[j] = mean;
      mom_m2[j] = m2;
    }
  }
  if (MOMENTS) {
    float* const red = (float*)(lds + LDS_BYTES - RED_BYTES);   // [WAVES_M][BN][2] + [WAVES_M] counts
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const float n2 = __shfl_xor(cnt, off, 64);
      const float tot = cnt + n2;
      const float w = tot > 0.f ? n2 / tot : 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float mb = __shfl_xor(mom_mean[j], off, 64), qb = __shfl_xor(mom_m2[j], off, 64);
        const float d = mb - mom_mean[j];
        mom_mean[j] = fmaf(d, w, mom_mean[j]);
        mom_m2[j] += qb + d * d * cnt * w;
      }
      cnt = tot;
    }
    if (f_c == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int cl = (wn * NT + j) * 16 + f_r;
        red[(wm * BN + cl) * 2 + 0] = mom_mean[j];
        red[(wm * BN + cl) * 2 + 1] = mom_m2[j];
      }
      if (lane == 0 && wn == 0) red[WAVES_M * BN * 2 + wm] = cnt;
    }
  }
  __syncthreads();

  // 16-B row segments out: BN / 8 chunks per row
  constexpr int CPR = BN / 8;
#pragma unroll
  for (int i = 0; i < BM * CPR / kThreads; ++i) {
    const int q = tid + kThreads * i;
    const int r = q / CPR, c = q % CPR;
    const int gc = col0 + c * 8;
    if (r < rows_valid && gc < g.N)                           // N % 8 == 0: a chunk is wholly inside or outside
      *(u32x4*)(g.C + (int64_t)(row0 + r) * g.ldc + plane_off(gc, g.c_plog, g.c_pstride)) = *(const u32x4*)(ldsC + r * C_PITCH + c * 16);
  }

  if (MOMENTS && tid < BN && col0 + tid < g.N) {               // merge the WAVES_M partials of column tid, write out
    const float* red = (const float*)(lds + LDS_BYTES - RED_BYTES);
    float n = red[WAVES_M * BN * 2], mean = red[tid * 2], m2 = red[tid * 2 + 1];
#pragma unroll
    for (int w = 1; w < WAVES_M; ++w) {
      const float nb = red[WAVES_M * BN * 2 + w], mb = red[(w * BN + tid) * 2], qb = red[(w * BN + tid) * 2 + 1];
      const float tot = n + nb;
      const float wgt = tot > 0.f ? nb / tot : 0.f;
      const float d = mb - mean;
      mean = fmaf(d, wgt, mean);
      m2 += qb + d * d * n * wgt;
      n = tot;
    }
    float* out = g.moments + (int64_t)(tile / g.n_col_tiles) * 2 * g.N;
    out[col0 + tid] = mean;
    out[g.N + col0 + tid] = m2;
  }
}

template <int WAVES_M, int WAVES_N, int MT, int NT>
int launch(GemmArgs g, hipStream_t stream) {
  constexpr int BM = WAVES_M * MT * 16, BN = WAVES_N * NT * 16;
  g.n_col_tiles = (g.N + BN - 1) / BN;
  const int64_t tiles = (int64_t)((g.M + BM - 1) / BM) * g.n_col_tiles;
  SG_REQUIRE(tiles <= INT32_MAX, "sg_gemm_nt: too many tiles");
  g.n_tiles = (int)tiles;
  if (g.moments) gemm_nt_bf16<WAVES_M, WAVES_N, MT, NT, true><<<g.n_tiles, kThreads, 0, stream>>>(g);
  else gemm_nt_bf16<WAVES_M, WAVES_N, MT, NT, false><<<g.n_tiles, kThreads, 0, stream>>>(g);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

inline bool a16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int g_gemm_tile = 0;   // SG_TUNE_GEMM_TILE: 0 = automatic, 1 = 128-row tiles only, 2 = 64 x 256 tiles wherever N > 64,
                       // 3 = the 256 x 256 eight-wavefront kernel (gemm_mfma256.hip) wherever it takes the shape

// automatic choice: the persistent 256 x 256 kernel serves the compute-bound products (K x N above ~100 K elements: the
// 256- and 512-channel layers and [V,384] x [384,256]), the 128-row-tile kernel below the HBM-bound ones
constexpr int64_t kBigMinWeightElems = 90000;
constexpr int64_t kBigMinRows = 16384;

// 64 x 256 tiles (ONE column tile covers N <= 256, so A is fetched by one workgroup only) were measured SLOWER than
// 128 x 128 tiles on every wide product (e.g. [V,384]x[384,256]: 0.43 ms against 0.33 ms; profiles/
// r02_mfma_gemm_bench.json): the column tiles of a row tile already share A through L2 (XCD-contiguous tile order),
// and the narrower row tile re-streams B twice as often.  Kept behind the knob for A/B runs only.
bool wide_tile(int64_t N) { return g_gemm_tile == 2 && N > 64; }
// (measured at V = 1 M, profiles/r04_gemm192_bench.json: N = 192 0.199 -> 0.174 ms at K = 128, 0.143 -> 0.115 at K = 64; N = 384 as
// two such tiles at two workgroups per CU is SLOWER than three 128-column tiles at three: 0.393 -> 0.442 ms; 4: A/B switch, off)
bool tile_192(int64_t N) { return g_gemm_tile != 1 && g_gemm_tile != 4 && (N == 192 || (g_gemm_tile == 5 && N == 384)); }

}  // namespace

int gemm_tile_rows(int64_t N) { return wide_tile(N) ? 64 : 128; }

bool gemm_nt_takes_big_tile(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc) {
  if (g_gemm_tile == 1 || g_gemm_tile == 2 || !gemm_nt_256_supported(M, N, K, lda, ldb, ldc)) return false;
  return g_gemm_tile == 3 || (K * N > kBigMinWeightElems && M >= kBigMinRows);
}

int set_gemm_tuning(int value) {
  g_gemm_tile = value;
  return SG_OK;
}

int launch_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, const float* bias, void* C, int64_t ldc,
                   int64_t M, int64_t N, int64_t K, int dtype, float* moments, hipStream_t stream, Planes pa, Planes pc) {
  if (dtype != SG_BF16) {
    set_error("sg_gemm_nt: bf16 operands only (dtype %d); fp32 products stay with the BLAS library", dtype);
    return SG_ERR_UNSUPPORTED;
  }
  if (K % 8 || N % 8 || lda % 8 || ldb % 8 || ldc % 8 || !a16(A) || !a16(B) || !a16(C)) {
    set_error("sg_gemm_nt: needs K, N and the row strides to be multiples of 8 and 16-byte aligned operands "
              "(K=%lld N=%lld lda=%lld ldb=%lld ldc=%lld)", (long long)K, (long long)N, (long long)lda, (long long)ldb,
              (long long)ldc);
    return SG_ERR_UNSUPPORTED;
  }
  SG_REQUIRE(M <= INT32_MAX && N <= INT32_MAX && K <= INT32_MAX, "sg_gemm_nt: size out of range");
  const bool planes = pa.on() || pc.on();
  if (planes) {
    SG_REQUIRE((!pa.on() || (pa.log2 >= 3 && pa.log2 < 31 && pa.stride % 8 == 0)) &&
               (!pc.on() || (pc.log2 >= 3 && pc.log2 < 31 && pc.stride % 8 == 0)),
               "sg_gemm_nt: planes need >= 8 columns each and a plane stride that is a multiple of 8 elements");
    SG_REQUIRE(!wide_tile(N), "sg_gemm_nt: operands kept as planes run on the 128-row tiles only");
  }
  if (!planes && moments == nullptr && gemm_nt_takes_big_tile(M, N, K, lda, ldb, ldc))
    return launch_gemm_nt_256(A, lda, B, ldb, bias, C, ldc, M, N, K, stream);
  GemmArgs g;
  g.a_plog = pa.log2; g.a_pstride = pa.stride;
  g.c_plog = pc.log2; g.c_pstride = pc.stride;
  g.A = (const uint16_t*)A; g.lda = lda;
  g.B = (const uint16_t*)B; g.ldb = ldb;
  g.bias = bias;
  g.C = (uint16_t*)C; g.ldc = ldc;
  g.moments = moments;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.n_col_tiles = g.n_tiles = 0;
  if (wide_tile(N)) return launch<1, 4, 4, 4>(g, stream);  // 64 x 256, wavefront tile 64 x 64
  // N = 192 (the products around the 64 <-> 128-channel layers: [V,128]x[128,192]): ONE 128 x 192 tile per row tile, wavefront
  // tile 64 x 96 -- no half-empty second column tile, A staged once
  if (tile_192(N) && !moments) return launch<2, 2, 4, 6>(g, stream);
  if (N > 64) return launch<2, 2, 4, 4>(g, stream);     // 128 x 128, wavefront tile 64 x 64
  if (N > 32) return launch<4, 1, 2, 4>(g, stream);     // 128 x 64,  wavefront tile 32 x 64
  if (N > 16) return launch<4, 1, 2, 2>(g, stream);     // 128 x 32
  return launch<4, 1, 2, 1>(g, stream);                 // 128 x 16
}

// =====================================================================================================================
// Weight gradient:  W[N, Kp] = A[M, N]^T * B[M, Kp]   (bf16 operands, fp32 result)
//
// dWcat = dOut^T [Tx0|Tx1|Tx2] of ChebConv's backward (autograd of the `lins[k]` calls, util/networks.py:42,49): the
// reduction runs over ALL M vertices and the result is tiny.  Both operands are row-major with the reduction index m
// as the ROW, so the MFMA fragments (8 consecutive m per lane) are columns of the LDS tiles: they are read with
// ds_read_b64_tr_b16, the gfx950 transposing LDS read (a 16-lane group reads 4 rows x 16 columns and every lane
// receives one column) -- no transposed copy of dOut or T is ever materialised.
//  * one workgroup = one 128 x 128 tile of W for one SLAB of 512 .. 8192 vertices; it walks its slab in steps of 64
//    rows: [64 m][128] tiles of A and B through registers into LDS (256-B rows, 32-B segments XOR-swizzled by the row so
//    that the eight rows a 32-lane half reads hit disjoint banks), 32 MFMAs per wavefront per step;
//  * the slabs' partial tiles go to an fp32 workspace and are summed in slab order by a second kernel: deterministic,
//    no atomics;
//  * workgroups of one slab are neighbours in launch order on one XCD, so the (Kp / 128) x re-read of A's and the
//    (N / 128) x re-read of B's slab rows are L2 hits.
namespace {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
constexpr int kTnStep = 64;
constexpr int kTnMaxSlabRows = 8192;

// rows of one slab: enough slabs that (slabs x output tiles) fills the chip several times over even when W is a single
// tile (N = 32, Kp = 48 at V = 1 M: 1 tile -> ~500 slabs of 2048 rows), never more than kTnMaxSlabRows
__host__ __device__ inline int tn_slab_rows(int64_t M, int64_t n_tiles) {
  int64_t rows = (M * n_tiles / 512 + kTnStep - 1) / kTnStep * kTnStep;
  rows = rows < 512 ? 512 : rows;
  return (int)(rows > kTnMaxSlabRows ? kTnMaxSlabRows : rows);
}

struct TnArgs {
  const uint16_t* A; int64_t lda;
  const uint16_t* B; int64_t ldb;
  float* W;                    // [n_slabs][N][Kp]
  int M, N, Kp;
  int tiles_n, tiles_k, n_tiles, n_blocks, slab_rows;
  int a_plog; int64_t a_pstride;  // A / B kept as planes of 2^plog columns (sg_common.h Planes; 31: plain row-major)
  int b_plog; int64_t b_pstride;
};

__device__ __forceinline__ int tn_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

__global__ __launch_bounds__(kThreads, 3) void gemm_tn_bf16(const TnArgs g) {
  constexpr int TILE_BYTES = kTnStep * 256;                       // [64 m][128 columns] bf16
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * TILE_BYTES];
  uint8_t* const ldsA = lds;
  uint8_t* const ldsB = lds + TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int blk = xcd_run(blockIdx.x, g.n_blocks);
  const int slab = blk / g.n_tiles, tile = blk % g.n_tiles;
  const int n0 = (tile / g.tiles_k) * 128, k0 = (tile % g.tiles_k) * 128;
  const int m_begin = slab * g.slab_rows;
  int m_end = m_begin + g.slab_rows;
  m_end = m_end < g.M ? m_end : g.M;

  // staging: chunk q = tid + 256 i -> tile row q / 16, 16-B chunk q % 16 (16 lanes = one 256-B row segment)
  const int s_c = tid & 15;
  const bool a_col = n0 + s_c * 8 < g.N, b_col = k0 + s_c * 8 < g.Kp;    // N, Kp % 8 == 0
  const uint16_t* a_src = g.A + (a_col ? plane_off(n0 + s_c * 8, g.a_plog, g.a_pstride) : 0);
  const uint16_t* b_src = g.B + (b_col ? plane_off(k0 + s_c * 8, g.b_plog, g.b_pstride) : 0);
  int dst_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (tid >> 4) + 16 * i;
    dst_off[i] = r * 256 + ((((s_c >> 1) ^ tn_swz(r)) << 5) | ((s_c & 1) << 4));
  }
  u32x4 ra[4], rb[4];
  bool rin[4];
  auto fetch = [&](int m0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = m0 + (tid >> 4) + 16 * i;
      rin[i] = m < m_end;
      m = rin[i] ? m : g.M - 1;                                   // clamped: in bounds, zeroed on the way to LDS
      ra[i] = *(const u32x4*)(a_src + (int64_t)m * g.lda);
      rb[i] = *(const u32x4*)(b_src + (int64_t)m * g.ldb);
    }
  };
  auto stash = [&]() {
    const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(u32x4*)(ldsA + dst_off[i]) = (rin[i] && a_col) ? ra[i] : zero;
      *(u32x4*)(ldsB + dst_off[i]) = (rin[i] && b_col) ? rb[i] : zero;
    }
  };

  // transposing fragment reads: lane = 16 g + 4 q + p supplies row (32 ms + 8 g + 4 h + q), columns 4 p .. 4 p + 3 of the
  // tile's 16-column segment; lane 16 g + i receives column i of those 4 rows (h = 0 / 1: the two halves of the 8 m)
  const int fg = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto frag = [&](const uint8_t* base, int seg, int ms) -> bf16x8 {
    bf16x4 h[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int row = 32 * ms + 8 * fg + 4 * hh + fq;
      const uint8_t* p = base + row * 256 + ((seg ^ tn_swz(row)) << 5) + fp * 8;
      h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)p);
    }
    return bf16x8{h[0][0], h[0][1], h[0][2], h[0][3], h[1][0], h[1][1], h[1][2], h[1][3]};
  };

  if (m_begin < m_end) {
    fetch(m_begin);
    stash();
  }
  __syncthreads();
  for (int m0 = m_begin; m0 < m_end; m0 += kTnStep) {
    const bool more = m0 + kTnStep < m_end;
    if (more) fetch(m0 + kTnStep);
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = frag(ldsA, wn * 4 + i, ms);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = frag(ldsB, wk * 4 + j, ms);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (more) {
      stash();
      __syncthreads();
    }
  }

  // partial tile out: D column = lane & 15 (k'), rows = 4 (lane >> 4) + reg (n)
  float* const W = g.W + (int64_t)slab * g.N * g.Kp;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kk = k0 + (wk * 4 + j) * 16 + (lane & 15);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + (wn * 4 + i) * 16 + (lane >> 4) * 4 + q;
        if (n < g.N && kk < g.Kp) W[(int64_t)n * g.Kp + kk] = acc[i][j][q];
      }
    }
}

// out[n][k] = sum over the slabs' partial tiles.  64 x 16 threads: 64 consecutive float4 of the output, 16 slab groups
// (thread y adds slabs y, y + 16, ..), then the 16 group sums in order -- a fixed summation tree, so the result is
// deterministic, and a small output with ~500 slabs is not one thread walking 500 dependent loads.
__global__ __launch_bounds__(1024) void tn_reduce(const float* __restrict__ W, int n_slabs, int64_t elems, int Kp,
                                                  float* __restrict__ out, int64_t ldo, const GradSink sink) {
  __shared__ f32x4 s_part[16][64];
  const int main_blocks = (int)((elems / 4 + 63) / 64);
  if ((int)blockIdx.x >= main_blocks) {                        // the rider's workgroups (GradSink::cs_*)
    colsum_ride(sink, (int)blockIdx.x - main_blocks);
    return;
  }
  const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
  const int64_t e = ((int64_t)blockIdx.x * 64 + x) * 4;      // Kp % 8 == 0: a float4 stays inside one row
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
    if (sink.mode) {                                          // Cin % 4 == 0: the four values lie in one accumulator row
      float* const d = sink_ptr(sink, n, kk);                 // (scalar: a .grad view of a flat buffer is only 4-byte aligned)
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] += acc[q];
    }
  }
}

}  // namespace

// automatic choice for the weight gradient: the 256 x 256 ring (gemm_mfma256.hip) serves the compute-bound products
// (N, Kp multiples of 256, N x Kp above ~100 K elements), the 128 x 128 kernel below the rest
bool gemm_tn_takes_big_tile(int64_t M, int64_t N, int64_t Kp, int64_t lda, int64_t ldb) {
  if (g_gemm_tile == 1 || g_gemm_tile == 2 || !gemm_tn_256_supported(M, N, Kp, lda, ldb)) return false;
  return g_gemm_tile == 3 || (N * Kp > kBigMinWeightElems && M >= kBigMinRows);
}

static int64_t small_tn_slabs(int64_t M, int64_t N, int64_t Kp) {
  if (M <= 0) return 1;
  const int rows = tn_slab_rows(M, ((N + 127) / 128) * ((Kp + 127) / 128));
  return (M + rows - 1) / rows;
}

int64_t gemm_tn_slabs(int64_t M, int64_t N, int64_t Kp) {
  // (sized for either kernel: the 256 x 256 ring's slabs plus one for the rows past the last full 64-row step)
  int64_t big = 0;
  if (gemm_tn_256_supported(M, N, Kp, 8, 8)) big = gemm_tn_256_slabs(M, N, Kp) + 1;
  const int64_t small = small_tn_slabs(M, N, Kp);
  return big > small ? big : small;
}

int launch_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t Kp, int dtype,
                   float* workspace, float* out, int64_t ldo, hipStream_t stream, const GradSink* sink_, Planes pa, Planes pb) {
  const GradSink sink = sink_ ? *sink_ : GradSink{};
  SG_REQUIRE(sink.mode == 0 || sink.Cin % 4 == 0, "sg_gemm_tn: a gradient sink needs Cin to be a multiple of 4");
  if (dtype != SG_BF16) {
    set_error("sg_gemm_tn: bf16 operands only (dtype %d)", dtype);
    return SG_ERR_UNSUPPORTED;
  }
  if (N % 8 || Kp % 8 || lda % 8 || ldb % 8 || ldo % 4 || !a16(A) || !a16(B) || !a16(out) || !a16(workspace)) {
    set_error("sg_gemm_tn: needs N, Kp and the operand row strides to be multiples of 8 and 16-byte aligned buffers");
    return SG_ERR_UNSUPPORTED;
  }
  SG_REQUIRE(M > 0 && M <= INT32_MAX && N <= INT32_MAX && Kp <= INT32_MAX, "sg_gemm_tn: size out of range");
  const bool planes = pa.on() || pb.on();
  if (planes)
    SG_REQUIRE((!pa.on() || (pa.log2 >= 3 && pa.log2 < 31 && pa.stride % 8 == 0)) &&
               (!pb.on() || (pb.log2 >= 3 && pb.log2 < 31 && pb.stride % 8 == 0)),
               "sg_gemm_tn: planes need >= 8 columns each and a plane stride that is a multiple of 8 elements");
  if (!planes && gemm_tn_takes_big_tile(M, N, Kp, lda, ldb)) {
    const int64_t m_full = M / 64 * 64, elems = N * Kp;
    int64_t slabs = gemm_tn_256_slabs(M, N, Kp);
    const int rc = launch_gemm_tn_256(A, lda, B, ldb, M, N, Kp, workspace, stream);
    if (rc != SG_OK) return rc;
    if (m_full < M) {                                   // the last < 64 rows: one slab of the 128 x 128 kernel
      TnArgs t;
      t.A = (const uint16_t*)A + m_full * lda; t.lda = lda;
      t.B = (const uint16_t*)B + m_full * ldb; t.ldb = ldb;
      t.W = workspace + slabs * elems;
      t.M = (int)(M - m_full); t.N = (int)N; t.Kp = (int)Kp;
      t.tiles_n = (int)((N + 127) / 128);
      t.tiles_k = (int)((Kp + 127) / 128);
      t.n_tiles = t.tiles_n * t.tiles_k;
      t.slab_rows = 512;
      t.n_blocks = t.n_tiles;
      t.a_plog = t.b_plog = 31; t.a_pstride = t.b_pstride = 0;
      gemm_tn_bf16<<<t.n_blocks, kThreads, 0, stream>>>(t);
      SG_HIP_TRY(hipGetLastError());
      ++slabs;
    }
    tn_reduce<<<(int)((elems / 4 + 63) / 64) + (sink.cs_partial ? sink.cs_C : 0), 1024, 0, stream>>>(workspace, (int)slabs, elems, (int)Kp, out, ldo, sink);
    SG_HIP_TRY(hipGetLastError());
    return SG_OK;
  }
  TnArgs g;
  g.A = (const uint16_t*)A; g.lda = lda;
  g.B = (const uint16_t*)B; g.ldb = ldb;
  g.W = workspace;
  g.M = (int)M; g.N = (int)N; g.Kp = (int)Kp;
  g.tiles_n = (int)((N + 127) / 128);
  g.tiles_k = (int)((Kp + 127) / 128);
  g.n_tiles = g.tiles_n * g.tiles_k;
  g.slab_rows = tn_slab_rows(M, g.n_tiles);
  g.a_plog = pa.log2; g.a_pstride = pa.stride;
  g.b_plog = pb.log2; g.b_pstride = pb.stride;
  const int64_t slabs = small_tn_slabs(M, N, Kp);
  SG_REQUIRE(slabs * g.n_tiles <= INT32_MAX, "sg_gemm_tn: too many workgroups");
  g.n_blocks = (int)(slabs * g.n_tiles);
  gemm_tn_bf16<<<g.n_blocks, kThreads, 0, stream>>>(g);
  SG_HIP_TRY(hipGetLastError());
  const int64_t elems = N * Kp;
  tn_reduce<<<(int)((elems / 4 + 63) / 64) + (sink.cs_partial ? sink.cs_C : 0), 1024, 0, stream>>>(workspace, (int)slabs, elems, (int)Kp, out, ldo, sink);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

}  // namespace sg
