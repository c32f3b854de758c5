// Internal declarations shared by the HIP translation units of libsemigcn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <string>

#include "semigcn.h"

namespace sg {

void set_error(const char* fmt, ...);

#define SG_HIP_TRY(expr)                                                              \
  do {                                                                                \
    hipError_t e__ = (expr);                                                          \
    if (e__ != hipSuccess) {                                                          \
      sg::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                    __LINE__);                                                        \
      return SG_ERR_HIP;                                                              \
    }                                                                                 \
  } while (0)

#define SG_REQUIRE(cond, ...)      \
  do {                             \
    if (!(cond)) {                 \
      sg::set_error(__VA_ARGS__);  \
      return SG_ERR_INVALID;       \
    }                              \
  } while (0)

// Bipartite CSR: for each of n_rows destination rows, the list of source columns.
struct Csr {
  int64_t n_rows = 0, n_cols = 0, nnz = 0;
  int32_t* rowptr = nullptr;  // [n_rows + 1] device
  int32_t* idx = nullptr;     // [nnz] device, ascending inside a row
  int32_t max_degree = 0;
  // Mini-tiles for the shared-gather aggregation kernel (optional; null when the graph does not qualify):
  // tile t = rows [t*tile_rows, (t+1)*tile_rows); its DISTINCT source rows are
  // tile_uniq[tile_uptr[t] .. tile_uptr[t+1]) (ascending) and edge k reads local slot tile_eloc[k].
  int32_t tile_rows = 0;
  int32_t* tile_uptr = nullptr;   // [n_tiles + 1]
  int32_t* tile_uniq = nullptr;   // [tile_uptr[n_tiles]]
  uint8_t* tile_eloc = nullptr;   // [nnz]
  // Optional copies of idx / tile_uniq with the source-row scale packed beside each id, (j, bits(scale[j])):
  // the kernels then stage one 8-byte stream instead of chasing scale[idx[k]] (a dependent gather that
  // adds a full memory latency to every wavefront's chunk).  Valid for the scale vector `packed_scale`.
  int2* idx_w = nullptr;          // [nnz]
  int2* tile_uniq_w = nullptr;    // [tile_uptr[n_tiles]]
  // LDS tiles (optional, spmm.hip::spmm_lds): kLdsRows rows per tile, its distinct sources lt_uniq[lt_uptr[t] ..) with
  // the scale packed beside each id in lt_uniq_w, and every edge's slot in its tile's list
  int32_t* lt_uptr = nullptr;
  int32_t* lt_uniq = nullptr;
  uint8_t* lt_eloc = nullptr;
  int2* lt_uniq_w = nullptr;
  // tile records of spmm.hip::spmm_ring (see kRec*), built for ONE (source scale, row scale, row id) triple
  uint8_t* lt_rec = nullptr;
  int32_t lt_nrec = 0;
  const float* rec_scale_dst = nullptr;
  const int32_t* rec_row_id = nullptr;
  // the records are built by the FIRST aggregation that can use them (bf16 rows of 128 / 256 channels): a graph that only
  // ever sees float32 features or narrow rows -- fp32 runs, the coarse levels of the MGCN -- never pays for them
  // (896 bytes per 16 rows, a scan and two stream syncs).  rec_pending: what to build them for, set at graph creation.
  bool rec_pending = false;
  const float* pend_scale_src = nullptr;
  const float* pend_scale_dst = nullptr;
  const int32_t* pend_row_id = nullptr;
  void defer_records(const float* ss, const float* sd, const int32_t* rid) {
    rec_pending = true;
    pend_scale_src = ss;
    pend_scale_dst = sd;
    pend_row_id = rid;
  }
  const float* packed_scale = nullptr;
  void release();
};

constexpr int kTileRows = 4;       // rows per mini-tile (one accumulator set per row in registers)
constexpr int kTileSlots = 32;     // max distinct source rows per mini-tile
constexpr int kTileEdges = 128;    // max edges per mini-tile
constexpr int kLdsRows = 16;       // rows per LDS tile
constexpr int kLdsSlots = 48;      // max distinct source rows per LDS tile (16 rows + their ring in a locality order: ~36)
constexpr int kLdsEdges = 256;     // max edges per LDS tile
constexpr int kRingSlots = 56;     // max distinct source rows per tile of spmm_ring (98 % of the 16-row blocks of a flipped Morton-ordered mesh)
// Tile record of the pipelined LDS kernel (spmm.hip::spmm_ring): everything the reduction of one tile of <= kLdsRows
// consecutive (processing-order) rows needs, in ONE contiguous piece that one LDS-DMA instruction brings in:
//   int32 src[kRingSlots]         distinct source rows, ascending, padded with the last one
//   uint8 slot[kLdsRows][16]      slot of the u-th neighbour of row r in src[] (entries past the row's degree: last slot | 64)
//   uint8 deg[kLdsRows]
//   int32 nu                      number of distinct sources; 0: the tile does not fit (more than kRingSlots sources, a row
//                                 with more than 16 neighbours or a repeated one) and is gathered from global memory
//   int32 e0, r0, nrows           first edge, first row, rows of the tile
//   float sd[kLdsRows]            destination-row scale (1 when the operator has none)
//   int32 row[kLdsRows]           the caller's row id of each row (Csr row_id view; r0 + i otherwise)
//   float w[kRingSlots]           source-row scale of every slot
constexpr int kRecSrc = 0;
constexpr int kRecSlot = kRecSrc + 4 * kRingSlots;
constexpr int kRecDeg = kRecSlot + 16 * kLdsRows;
constexpr int kRecNu = kRecDeg + kLdsRows;
constexpr int kRecE0 = kRecNu + 4;
constexpr int kRecR0 = kRecE0 + 4;
constexpr int kRecNrows = kRecR0 + 4;
constexpr int kRecSd = kRecNrows + 4;
constexpr int kRecRow = kRecSd + 4 * kLdsRows;
constexpr int kRecW = kRecRow + 4 * kLdsRows;
constexpr int kRecBytes = 896;    // padded to whole 128-byte lines: every producer wavefront of the kernel fetches an equal share
static_assert(kRecW + 4 * kRingSlots <= kRecBytes && kRecSd % 16 == 0, "tile record layout");
int build_tiles(Csr* c, hipStream_t stream);
// Fills Csr::idx_w (and tile_uniq_w when the CSR carries tiles) for the source scale vector `scale` [n_cols].
int pack_source_scale(Csr* c, const float* scale, hipStream_t stream);
// Where a weight gradient goes besides its own [N, Kp] buffer: the K per-matrix .grad accumulators of a ChebConv
// ([Cout, Cin] each, contiguous), += in the kernel that finishes the reduction over the vertices -- no separate add launch.
// mode 1: the gradient is dWcat [Cout, K*Cin] (column block k -> dst[k]); mode 2: dWstack [K*Cout, Cin] (row block k -> dst[k]).
struct GradSink {
  float* dst[3] = {nullptr, nullptr, nullptr};
  int mode = 0, Cin = 0, Cout = 0;
  // A RIDER (round 6): the column sums that finish the conv bias' gradient (colsum_finalize: one workgroup per channel over the
  // cs_nb workgroup partials of the BatchNorm-backward apply pass) have no consumer inside the backward pass, so they travel in
  // the launch that finishes the layer's weight gradient -- cs_C extra workgroups behind the reduce's own -- instead of a launch
  // of their own: 13 launches fewer per SGCN iteration, the same sums in the same order (colsum_ride).
  const float* cs_partial = nullptr;
  int64_t cs_nb = 0;
  int cs_C = 0;
  float* cs_out = nullptr;
  float* cs_acc = nullptr;
};
#ifdef __HIPCC__
// channel c of the rider, by the whole workgroup (>= 256 threads): the sum order of bn_act.hip's colsum_finalize / block_sum_256
__device__ inline void colsum_ride(const GradSink& s, int c) {
  __shared__ double s_ride[4];
  double v = 0.0;
  if (threadIdx.x < 256)
    for (int64_t b = threadIdx.x; b < s.cs_nb; b += 256) v += s.cs_partial[b * s.cs_C + c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if (threadIdx.x < 256 && (threadIdx.x & 63) == 0) s_ride[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = (s_ride[0] + s_ride[1]) + (s_ride[2] + s_ride[3]);
    s.cs_out[c] = (float)t;
    if (s.cs_acc) s.cs_acc[c] += (float)t;
  }
}
#endif
__device__ __forceinline__ float* sink_ptr(const GradSink& s, int64_t n, int64_t kk) {
  if (s.mode == 1) {
    const int64_t k = kk / s.Cin;
    return s.dst[k] + n * s.Cin + (kk - k * s.Cin);
  }
  const int64_t k = n / s.Cout;
  return s.dst[k] + (n - k * s.Cout) * s.Cin + kk;
}
// A matrix operand kept as PLANES of 2^log2 columns: element (r, c) lives at base + (c >> log2) * stride + r * ld +
// (c & (2^log2 - 1)) (elements; ld = the row pitch inside a plane).  The K column blocks [Tx0 | Tx1 | Tx2] of a narrow
// ChebConv layer (16 .. 64 channels: 32 .. 128-byte rows) are stored this way -- K dense [V, C] tensors -- because an
// aggregation that gathers 32-byte rows from a 96-byte pitch uses a quarter of every line it pulls.
// log2 = 31: one plane, plain row-major.
struct Planes {
  int log2 = 31;
  int64_t stride = 0;
  bool on() const { return log2 != 31; }
};
__host__ __device__ __forceinline__ int64_t plane_off(int c, int log2, int64_t stride) {
  return (int64_t)(c >> log2) * stride + (c & (int)((1u << log2) - 1u));
}
// Products with a tiny weight matrix (thin_gemm.hip)
bool thin_shape(int64_t N, int64_t K);
int64_t thin_tn_blocks(int64_t V);
int launch_thin_nt(const void* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, void* Y, int64_t ldy, int64_t V,
                   int64_t N, int64_t K, int dtype, hipStream_t stream);
int launch_thin_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t V, int64_t N, int64_t K, int dtype,
                   float* workspace, float* out, int64_t ldo, hipStream_t stream, const GradSink* sink = nullptr);
// Tile records for spmm_ring (when enabled): rows scaled by scale_dst (nullable), sources by scale_src, output rows row_id (nullable).
int build_ring_records(Csr* c, const float* scale_src, const float* scale_dst, const int32_t* row_id, hipStream_t stream);
bool ring_enabled();
bool ring_f32_enabled();

// Builds a Csr from (dst, src) int64 device pairs. Pairs with dst == src are dropped when
// drop_self. If keys_out != nullptr the sorted (dst << 32 | src) keys (n entries, dropped
// pairs sort last) are left there for the caller (device buffer of n uint64).
int build_csr(const int64_t* dst, const int64_t* src, int64_t n, int64_t n_rows, int64_t n_cols,
              bool drop_self, hipStream_t stream, Csr* out, uint64_t* keys_out);

// out[i] = (rowptr[i+1]-rowptr[i])^-1/2 (0 where the row is empty); or 1/count when inv_only.
int degree_scale(const Csr& c, bool inv_sqrt, float* out, hipStream_t stream);
// Graph-only locality order of a symmetric square CSR.  *order_out (device, [n_rows], caller frees with hipFree) lists the
// rows in processing order, or stays null when the numbering is already local (mode 0 = decide, 1 = never, 2 = always).
int locality_order(const Csr& c, int mode, hipStream_t stream, int32_t** order_out);
// The rows of `base` in the order `order` (order[p] = base row at position p), column ids unchanged.
int permute_rows(const Csr& base, const int32_t* order, hipStream_t stream, Csr* out);
int gather_floats(const float* src, const int32_t* order, int64_t n, float* out, hipStream_t stream);
int graph_reorder_mode();
int set_graph_reorder_mode(int v);
// *equal = 1 iff the first n keys of a and b are identical.
int keys_equal(const uint64_t* a, const uint64_t* b, int64_t n, hipStream_t stream, int* equal);

struct SpmmArgs {
  const int32_t* rowptr;
  const int32_t* idx;
  const int32_t* tile_uptr = nullptr;   // non-null: the CSR carries row tiles (see Csr)
  const int32_t* tile_uniq = nullptr;
  const uint8_t* tile_eloc = nullptr;
  const int2* idx_w = nullptr;          // non-null: (id, scale bits) pairs replace idx + scale_src lookups
  const int2* tile_uniq_w = nullptr;
  const int32_t* lt_uptr = nullptr;     // LDS tiles (see Csr)
  const uint8_t* lt_eloc = nullptr;
  const int2* lt_uniq_w = nullptr;
  const int2* lt_idx_w = nullptr;       // = idx_w, for the tiles that fall back to global gathers
  const uint8_t* lt_rec = nullptr;      // tile records (see kRec*)
  int32_t lt_nrec = 0;
  const int32_t* row_id = nullptr;      // non-null: the CSR is in PROCESSING order; row p is the caller's row row_id[p]
                                        // (Y / X0 / X1 are addressed by it; scale_dst is indexed by p)
  const float* scale_dst;  // nullable, [n_rows]
  const float* scale_src;  // nullable, [n_cols]
  const void* X;
  const void* X0;  // nullable
  const void* X1;  // nullable
  void* Y;
  int64_t ldx, ldx0, ldx1, ldy;
  int32_t n_rows;
  int64_t n_cols = 0;   // rows of X (bounds the gather offsets)
  int32_t C;
  float alpha, beta, gamma;
};

// Y[r] = alpha * sd[r] * sum_k ss[idx[k]] * X[idx[k]] + beta * X0[r] + gamma * X1[r]
int launch_spmm(const SpmmArgs& a, int dtype, hipStream_t stream);
int set_tuning(int knob, int value);
bool tiles_enabled();
bool lds_tiles_enabled();
int launch_gather_rows(const int32_t* rows, int64_t n, const void* X, int64_t ldx, void* Y,
                       int64_t ldy, int64_t C, int dtype, hipStream_t stream);

// bn_act.hip
int64_t col_blocks(int64_t V);
int launch_col_reduce(int mode, const void* A, int64_t lda, const void* H, int64_t ldh, const float* scale,
                      const float* shift, const float* mean, const float* invstd, float slope, float* out,
                      int64_t nblk, int64_t V, int64_t C, int dtype, hipStream_t stream);
int launch_bn_merge(const float* partial, int64_t nb, int64_t V, int64_t C, float* stats, hipStream_t stream);
int launch_bn_stats_finalize(const float* partial, int64_t nb, int64_t V, int64_t C, const float* gamma,
                             const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                             float* out, int64_t* batches_tracked, hipStream_t stream);
int launch_bn_bwd_coeffs(const float* partial, int64_t nb, int64_t C, double N, const float* gamma,
                         const float* invstd, float* out, float* acc_dweight, float* acc_dbias, const float* count_dev,
                         hipStream_t stream);
int launch_bn_merge_tiles(const float* partial, int64_t nb, int64_t rpb, int64_t V, int64_t C, float* stats,
                          float* count_out, hipStream_t stream);
int launch_bn_finalize_ranks(const float* all, int64_t world, int64_t C, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float momentum, float eps, float* out,
                             float* out_n, int64_t* batches_tracked, hipStream_t stream);
int launch_bn_finalize(const float* stats, double N, int64_t C, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float momentum, float eps, float* out,
                       hipStream_t stream);
int launch_col_apply(int mode, const void* A, int64_t lda, const void* H, int64_t ldh, const float* scale,
                     const float* shift, const float* mean, const float* invstd, const float* kk, const float* c1,
                     const float* c2, float slope, void* Y, int64_t ldy, int64_t V, int64_t C, int dtype,
                     hipStream_t stream, float* colsum = nullptr);
int64_t col_apply_blocks(int64_t V, int64_t C, int dtype);
int launch_colsum_finalize(const float* partial, int64_t nb, int64_t C, float* out, hipStream_t stream, float* acc = nullptr);

int launch_bn_stats_finalize_tiles(const float* partial, int64_t nb, int64_t rpb, int64_t V, int64_t C, const float* gamma,
                                   const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                   float* out, int64_t* batches_tracked, hipStream_t stream);
constexpr int kMultiAddMax = 8;
int launch_multi_add(int n, const float* const* srcs, const int64_t* src_ld, const int64_t* rows, const int64_t* cols,
                     float* const* dsts, hipStream_t stream);

// input_prep.hip
int64_t input_prep_blocks(int64_t V);
int launch_input_prep(const float* z1, const float* dm, const int64_t* order, const float* lo, const float* hi, void* X,
                      int64_t ldx, int64_t V, int dtype, hipStream_t stream);
int launch_input_prep_bwd(const void* gX, int64_t ldg, const float* z1, const float* dm, const int64_t* rank, const float* lo,
                          const float* hi, float* dz1, float* partial, float* d_lo, float* d_hi, int64_t V, int dtype,
                          hipStream_t stream);

int launch_input_bounds(const float* z1, int64_t V, float* pv, int64_t* pi, float* bounds, int64_t* arg, hipStream_t stream);
int launch_bounds_route(const float* d_lo, const float* d_hi, const int64_t* arg, float* dz1, hipStream_t stream);
int launch_mesh_loss_finalize(const float* partial, int64_t nb, float n_v, float n_f, float w_pos, float k1, float* out,
                              hipStream_t stream);

// gemm_mfma.hip
int gemm_tile_rows(int64_t N);      // rows per output tile (= rows per BatchNorm-moments record) for an N-column product
int set_gemm_tuning(int value);
int launch_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, const float* bias, void* C, int64_t ldc,
                   int64_t M, int64_t N, int64_t K, int dtype, float* moments, hipStream_t stream, Planes pa = {}, Planes pc = {});

bool gemm_nt_takes_big_tile(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc);
// gemm_mfma256.hip
bool gemm_nt_256_supported(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc);
int launch_gemm_nt_256(const void* A, int64_t lda, const void* B, int64_t ldb, const float* bias, void* C, int64_t ldc,
                       int64_t M, int64_t N, int64_t K, hipStream_t stream);
bool gemm_tn_256_supported(int64_t M, int64_t N, int64_t Kp, int64_t lda, int64_t ldb);
int gemm_tn_256_slabs(int64_t M, int64_t N, int64_t Kp);
int launch_gemm_tn_256(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t Kp,
                       float* workspace, hipStream_t stream);
bool gemm_tn_takes_big_tile(int64_t M, int64_t N, int64_t Kp, int64_t lda, int64_t ldb);
int64_t gemm_tn_slabs(int64_t M, int64_t N, int64_t Kp);
int launch_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t Kp, int dtype,
                   float* workspace, float* out, int64_t ldo, hipStream_t stream, const GradSink* sink = nullptr, Planes pa = {},
                   Planes pb = {});

// gemm_split.hip -- float32 products on the bf16 matrix cores (exact three-way split, six piece products)
bool gemm_nt_f32s_supported(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc);
int64_t gemm_nt_f32s_workspace(int64_t N, int64_t K);
int launch_gemm_nt_f32s(const float* A, int64_t lda, const float* W, int64_t w_rs, int64_t w_cs, const float* bias, float* C,
                        int64_t ldc, int64_t M, int64_t N, int64_t K, void* ws, int64_t ws_bytes, hipStream_t stream,
                        const void* prepacked = nullptr, int64_t variant_rows = 0);
int gemm_nt_f32s_variant(int64_t M);       // 1: the 128-row tiles (and their image layout) serve a product of M rows, 0: the 256-row ones
int launch_pack_split(const float* W, int64_t w_rs, int64_t w_cs, int64_t variant_rows, int64_t N, int64_t K, void* out,
                      hipStream_t stream);
int set_split_tuning(int value);
int set_bn_rows_tuning(int value);
bool split_engine_enabled(int kind = 0);
bool gemm_tn_f32s_supported(int64_t M, int64_t N, int64_t Kp, int64_t lda, int64_t ldb);
int64_t gemm_tn_f32s_workspace(int64_t M, int64_t N, int64_t Kp);
int launch_split_tn_reduce(const float* ws, int n_slabs, int64_t N, int64_t Kp, float* out, int64_t ldo, const GradSink* sink,
                           hipStream_t stream);
// float32 products with a small weight matrix on the vector ALUs (gemm_mid.hip): N, K in 4 .. 48, multiples of 4
// below this many rows the row-tile stream of gemm_nt_f32s fills too few workgroups and the BLAS library is faster (measured:
// 5 K rows 0.096 ms against 0.033-0.051, 50 K rows 0.21 against 0.41: profiles/r05_configs_n1.jsonl, r04_configs_n1.jsonl);
// the weight gradient (a reduction over the rows, cut into slabs) is level or ahead from 4 K rows on
constexpr int64_t kSplitNtPaysRows = 16384;
bool split_nt_pays(int64_t M, int64_t N, int64_t K);
bool mid_shape(int64_t N, int64_t K);
int64_t mid_tn_workspace(int64_t M, int64_t N, int64_t Kp);      // float32 elements
int launch_mid_nt(const float* X, int64_t ldx, const float* W, int64_t w_rs, int64_t w_cs, const float* bias, float* Y, int64_t ldy,
                  int64_t M, int64_t N, int64_t K, hipStream_t stream);
int launch_mid_tn(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int64_t Kp, float* ws, float* out,
                  int64_t ldo, hipStream_t stream, const GradSink* sink);
int launch_gemm_tn_f32s(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int64_t Kp, void* ws,
                        int64_t ws_bytes, float* out, int64_t ldo, hipStream_t stream, const GradSink* sink = nullptr);

// mesh_loss.hip
int64_t mesh_loss_blocks(int64_t V, int64_t F);
int launch_mesh_loss_fwd(const float* pos, const int64_t* faces, const float* tpos, const float* vkeep, const float* tfn,
                         const float* fkeep, int64_t V, int64_t F, float* partial, hipStream_t stream);
int launch_mesh_loss_bwd(const float* pos, const int64_t* faces, const float* tpos, const float* vkeep, const float* tfn,
                         const float* fkeep, const float* g, int64_t V, int64_t V_ext, int64_t F, float* grad,
                         hipStream_t stream);

// mesh_prep.hip
int mesh_edges(const int64_t* faces, int64_t F, int64_t V, int64_t* edges_out, int64_t* f2f_out,
               int64_t* n_edges_out, int* manifold_out, hipStream_t stream);
int launch_mask_dilate(const Csr& c, const uint64_t* in, uint64_t* out, int64_t W, hipStream_t stream);
int launch_face_mask(const int64_t* faces, int64_t F, int64_t V, const uint64_t* vbits, uint64_t* fbits, int64_t W,
                     hipStream_t stream);

int launch_mesh_loss_bwd_corners(const float* pos, const int64_t* faces, const float* tfn, const float* fkeep, const float* g,
                                 int64_t F, float* corner, hipStream_t stream);
int launch_mesh_loss_bwd_vertex_add(const float* pos, const float* tpos, const float* vkeep, const float* g, int64_t V,
                                    float* grad, hipStream_t stream);

// trace.hip -- optional per-launch event timing (sg_trace_*)
extern std::atomic<bool> g_trace_on;
void trace_open(int kind, int dtype, int engine, int64_t a, int64_t b, int64_t c, hipStream_t stream, int64_t* slot);
void trace_close(int64_t slot, hipStream_t stream);
struct TraceScope {
  int64_t slot = -1;
  hipStream_t stream;
  TraceScope(int kind, int dtype, int engine, int64_t a, int64_t b, int64_t c, hipStream_t s) : stream(s) {
    if (g_trace_on) trace_open(kind, dtype, engine, a, b, c, s, &slot);
  }
  ~TraceScope() {
    if (slot >= 0) trace_close(slot, stream);
  }
};

// block.hip
int set_block_planes(int value);

// dense.hip -- the three product shapes of a ChebConv layer behind one engine choice (thin kernels / own MFMA / BLAS library)
constexpr size_t kBlasWorkspace = 76u << 20;     // scratch handed to the BLAS library per call (PyTorch's default for this GPU family)
bool dense_nt_own(int dtype, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc);
bool dense_tn_own(int dtype, int64_t M, int64_t N, int64_t Kp, int64_t lda, int64_t ldb);
int64_t dense_tn_workspace(int dtype, int64_t M, int64_t N, int64_t Kp);
// C[M, N] = A[M, K] Bp[N, K]^T (+ bias): Bp in the feature dtype, B32 (nullable) the same matrix in float32 for the thin kernels
// pa / pc: A / C kept as planes (served by the 128-row MFMA kernel only: dense_planes_ok says whether a shape gets there)
int dense_nt(const void* A, int64_t lda, const void* Bp, const float* B32, int64_t ldb, const float* bias, void* C, int64_t ldc,
             int64_t M, int64_t N, int64_t K, int dtype, float* moments, bool* moments_done, void* blas_ws, size_t blas_ws_bytes,
             hipStream_t stream, Planes pa = {}, Planes pc = {}, const void* presplit = nullptr, int64_t presplit_rows = 0);
// do the nt product [M, K] x [N, K]^T and the tn product [M, N]^T [M, Kp] of these sizes run on the kernels that take planes?
bool dense_planes_ok_nt(int dtype, int64_t M, int64_t N, int64_t K);
bool dense_planes_ok_tn(int dtype, int64_t M, int64_t N, int64_t Kp);
// C[M, N] = A[M, K] Bp[K, N]; Bt (nullable) = Bp^T [N, K] in the feature dtype, Bt32 (nullable) in float32
int dense_nn(const void* A, int64_t lda, const void* Bp, int64_t ldb, const void* Bt, int64_t ldbt, const float* Bt32, void* C,
             int64_t ldc, int64_t M, int64_t N, int64_t K, int dtype, void* blas_ws, size_t blas_ws_bytes, hipStream_t stream,
             Planes pa = {}, Planes pc = {}, const void* presplit = nullptr, int64_t presplit_rows = 0);
// out[N, Kp] (float32) = A[M, N]^T B[M, Kp]; ws: dense_tn_workspace() floats
// sink (nullable): also += the result into the K weight accumulators; *sunk tells whether the engine that ran could do it
int dense_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t Kp, int dtype, float* ws,
             float* out, int64_t ldo, void* blas_ws, size_t blas_ws_bytes, hipStream_t stream, const GradSink* sink = nullptr,
             bool* sunk = nullptr, Planes pa = {}, Planes pb = {});

}  // namespace sg

struct sg_graph {
  sg::Csr fwd;            // rows = targets, cols = sources
  sg::Csr bwd;            // transposed; empty when symmetric
  float* dis_dst = nullptr;  // [V_dst]
  float* dis_src = nullptr;  // [V_src]; == dis_dst for square graphs
  bool symmetric = true;
  bool square = true;
  // Locality view (symmetric square graphs whose vertex numbering is not local, csr_build.hip::locality_order): the same
  // operator with its ROWS in a cache-friendly processing order -- column ids stay the caller's, so X / Y need no
  // permutation; the aggregation kernels address output rows through row_id.  Empty when the numbering is local.
  sg::Csr loc;
  int32_t* row_id = nullptr;      // [V] caller's row of processing position p
  float* dis_dst_loc = nullptr;   // [V] dis_dst[row_id[p]]
};

struct sg_pool {
  sg::Csr by_coarse;  // rows = coarse, cols = fine
  sg::Csr by_fine;    // rows = fine, cols = coarse
  float* inv_count = nullptr;  // [n_coarse]
};
