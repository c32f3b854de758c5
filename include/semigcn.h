/*
 * semigcn.h -- C ABI of libsemigcn_hip.so: the MI355X (gfx950) implementation of
 * SeMIGCN's graph-convolution message-passing hot path.
 *
 * Boundary.  The reference (100 % Python) reaches this arithmetic through
 * torch-geometric 2.2.0 / torch-scatter 2.1.0; every entry point below names the
 * reference interface (file:line under the reference tree, or the [3P]
 * third-party operator it calls there) that it replaces.  All data pointers are
 * DEVICE pointers owned by the caller (PyTorch-ROCm tensors in the shipped host
 * code); the library only borrows them for the duration of a stream-ordered
 * launch.  Handles own their index buffers.  `stream` is a hipStream_t passed
 * as void* (NULL = the null stream).  No C++ exception crosses this ABI: every
 * call returns SG_OK or a negative code, and sg_last_error() (thread-local)
 * holds the message.  Handles are immutable after creation and may be shared
 * between threads; launches are asynchronous.
 *
 * There is NO CPU implementation behind this ABI.
 */
#ifndef SEMIGCN_H
#define SEMIGCN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define SG_API __attribute__((visibility("default")))
#else
#define SG_API
#endif

#define SG_ABI_VERSION 1

enum sg_status {
  SG_OK = 0,
  SG_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, misaligned, index out of range) */
  SG_ERR_HIP = -2,         /* a HIP runtime call failed; message has hipGetErrorString */
  SG_ERR_UNSUPPORTED = -3, /* dtype / shape not implemented */
  SG_ERR_NO_DEVICE = -4    /* no gfx950 device visible */
};

enum sg_dtype {
  SG_F32 = 0,  /* float32 storage, float32 accumulate (the reference's precision) */
  SG_BF16 = 1  /* bfloat16 storage, float32 accumulate */
};

typedef struct sg_graph sg_graph; /* scaled Laplacian L^ = -D^-1/2 A D^-1/2 of one edge_index, CSR */
typedef struct sg_pool sg_pool;   /* one pool_hash (fine->coarse cluster map), CSR both ways */

SG_API const char* sg_last_error(void);
SG_API int sg_abi_version(void);
/* Number of visible HIP devices, or a negative sg_status. Does not create a context on failure. */
SG_API int sg_device_count(void);

/* ------------------------------------------------------------------------- *
 * Graph preprocessing -- replaces ChebConv.__norm__ [3P torch_geometric 2.2.0]
 * (remove_self_loops, get_laplacian('sym'), 2/lambda_max scaling with
 * lambda_max = 2.0, add_self_loops(+1) / add_self_loops(-1)), which the
 * reference re-runs inside EVERY conv call: util/networks.py:42,49 and
 * util/meshnet.py:40,44,50,54,58,106,112,116,120,124,224,232,240.
 *
 * edge_index: int64 [2, E] row-major COO in the reference layout
 * (util/mesh.py:229-230, util/datamaker.py:76-77); row 0 = source, row 1 =
 * target.  Any order; self-loops are dropped; duplicate edges are kept and
 * counted (as the reference's scatter does).  Done once per edge_index:
 *   rowptr[V+1], colidx[nnz] int32 sorted by (target, source);
 *   dis[v] = deg(v)^-1/2 with deg taken over SOURCE occurrences, 0 where deg = 0.
 * If the edge multiset is not symmetric a transposed CSR is kept for backward.
 * ------------------------------------------------------------------------- */
SG_API int sg_graph_create(const int64_t* edge_index, int64_t E, int64_t V, void* stream, sg_graph** out);

/* Rectangular operator for one vertex partition (no reference
 * counterpart -- the reference is single-device).  Rows = the V_dst owned
 * vertices, columns = V_src >= V_dst "owned | halo" vertices; `dis_src`
 * float32 [V_src] holds the GLOBAL deg^-1/2 of every column vertex (the first
 * V_dst entries are the owned rows').  (dst[i], src[i]) int64 pairs, i < n. */
SG_API int sg_graph_create_rect(const int64_t* dst, const int64_t* src, int64_t n, int64_t V_dst,
                         int64_t V_src, const float* dis_src, void* stream, sg_graph** out);

/* A ROW SUBSET of a partition operator (no reference counterpart; SURVEY 8(e) "overlap with interior-row compute"):
 * n_rows processed rows, row p writes output row row_id[p] (int32, device) and is scaled by dis_rows[p]; pairs
 * (dst_pos[i] in [0, n_rows), src[i] in [0, V_src)).  sg_spmm on such a handle touches only the rows row_id names, so
 * the interior rows of a block can be aggregated while the halo rows are still in flight, and the boundary rows after. */
SG_API int sg_graph_create_rows(const int64_t* dst_pos, const int64_t* src, int64_t n, int64_t n_rows, int64_t V_src,
                                const int32_t* row_id, const float* dis_rows, const float* dis_src, void* stream,
                                sg_graph** out);

SG_API int sg_graph_destroy(sg_graph* g);

typedef struct sg_graph_info {
  int64_t V_dst, V_src; /* rows, columns */
  int64_t nnz;          /* stored entries (self-loops removed) */
  int32_t symmetric;    /* 1: L^ == L^T (backward reuses the forward CSR) */
  int32_t max_degree;   /* longest CSR row */
} sg_graph_info;
SG_API int sg_graph_query(const sg_graph* g, sg_graph_info* info);
/* 1 when sg_graph_create gave the graph a locality view: a symmetric graph whose vertex numbering has no locality (a
 * raw scan; the operator tier `from torch_geometric.nn import ChebConv`, util/networks.py:4, sees no positions to sort
 * by) has its ROWS processed in a graph-derived order (two levels of multi-source-BFS cells); column ids, and with them
 * X, X0, X1 and Y of sg_spmm, stay in the caller's numbering and results are bit-identical (bf16 rows of 128 / 256 channels,
 * which the tiled matrix-core kernel serves: identical up to the accumulation order inside a 16-row tile, see sg_spmm).
 * See SG_TUNE_GRAPH_REORDER. */
SG_API int sg_graph_is_reordered(const sg_graph* g);
/* Copies the forward CSR and dis into caller-owned DEVICE buffers on `stream`:
 * rowptr int32 [V_dst+1], colidx int32 [nnz], dis float32 [V_src]; any may be NULL. */
SG_API int sg_graph_export(const sg_graph* g, int32_t* rowptr, int32_t* colidx, float* dis, void* stream);

/* ------------------------------------------------------------------------- *
 * Edge aggregation -- replaces MessagePassing.propagate -> message ->
 * SumAggregation -> torch_scatter.scatter(reduce='sum') -> ATen scatter_add_
 * [3P], i.e. the two `propagate` calls of every ChebConv.forward and the two
 * index_add_ calls of its autograd backward:
 *
 *     Y = alpha * op(L^) * X  +  beta * X0  +  gamma * X1
 *
 * op = L^ (transpose = 0) or L^T (transpose = 1).  Covers Tx1 = L^ x
 * (alpha 1), Tx2 = 2 L^ Tx1 - x (alpha 2, beta -1, X0 = x) and both backward
 * aggregations.  X [V_src, C], X0/X1/Y [V_dst, C] (transpose swaps the roles),
 * row-major with row strides ldx/ldx0/ldx1/ldy in ELEMENTS (so column blocks of
 * a wider [V, 3C] buffer can be read and written in place); X0/X1 may be NULL
 * (then beta/gamma are ignored).  Y must not alias X.  Edge weights are the
 * products -dis[i]*dis[j] of the per-vertex scales (dis[j] rides beside the
 * neighbour id in the CSR; no per-edge weight array).  Deterministic: one owner
 * per output row, fixed summation order, no atomics.  The order is ascending
 * neighbour id with one fp32 fma per neighbour, EXCEPT for bf16 rows of 128 or 256
 * channels (on any graph handle: sg_graph_create, _rect and _rows all carry the
 * tile records): those are reduced tile by tile
 * (<= 16 rows and their <= 56 distinct sources staged in LDS) on the matrix cores,
 * with the fp32 weights split exactly into three bf16 pieces and the MFMA's own
 * accumulation order -- same error bound, a different last bit in ~1 % of the
 * bf16 outputs (SG_TUNE_FLAGS bit 11 restores the fma chain).  float32 rows of 128
 * channels (and of 256 with at most one epilogue operand) take the same tiled
 * pipeline with float32 operands on v_mfma_f32_16x16x4_f32: the matrix core adds the
 * products of a step in slot order into its float32 accumulator, a tile's slots are its
 * sources in ascending id and an unused slot carries weight 0, so every row's sum is the
 * fma chain over its neighbours in ascending id -- the SAME BITS as the chain kernel
 * (tests/test_gpu_parity.py::test_f32_ring_kernel_equals_the_rows_kernel_bit_for_bit;
 * SG_TUNE_FLAGS bit 13 switches the tiled float32 pipeline off, for A/B timing only).
 * The tile records are built by the first aggregation that can use them: THAT call allocates device memory and synchronises
 * the stream twice (it is not asynchronous; never the case while the stream is being captured -- such a call runs on the chain
 * kernel and leaves the build to the next eager one); a failed build is retried by the next call.  sg_graph_prepare builds
 * them ahead of time (a no-op for other widths / dtypes, or when they exist), after which every sg_spmm is asynchronous.
 * Non-finite inputs: the gathers run in fixed-size batches whose unused slots
 * are switched off by a ZERO WEIGHT on a row that is read anyway (a neighbour of
 * one of the rows the same wavefront works on, or row 0; in the tiled kernel: any
 * source row of the same 16-row tile), so an Inf/NaN in X can turn into NaN in a
 * few output rows that are not its neighbours but lie within 32 rows of one; for
 * finite X the result is exactly the CSR sum.
 * ------------------------------------------------------------------------- */
SG_API int sg_graph_prepare(const sg_graph* g, int64_t C, int dtype, void* stream);
SG_API int sg_spmm(const sg_graph* g, int transpose, const void* X, int64_t ldx, const void* X0,
            int64_t ldx0, const void* X1, int64_t ldx1, void* Y, int64_t ldy, int64_t C, int dtype,
            float alpha, float beta, float gamma, void* stream);

/* ------------------------------------------------------------------------- *
 * Mesh pooling -- replaces MeshPool.forward (util/meshnet.py:14-17: sparse
 * P.x divided by the DENSE row-sum of P, rebuilt every call) and
 * MeshUnpool.forward (util/meshnet.py:25-27: sparse U.x), whose 0/1 matrices
 * come from pool_hash_to_mask / unpool_hash_to_mask (util/meshnet.py:331-341),
 * and their autograd transposes.
 *
 * (fine[i], coarse[i]) int64 pairs, i < n: the rows of Mesh.pool_hash
 * (util/mesh.py:653-676).  Duplicate pairs count twice, like the reference's
 * coalesced sparse matrices.
 *   pool_mean       Y[s] = (sum_{(o,s)} X[o]) / count[s]          X [n_fine,C]  -> Y [n_coarse,C]
 *   pool_mean_bwd   dX[o] = sum_{(o,s)} dY[s] / count[s]          dY [n_coarse,C] -> dX [n_fine,C]
 *   unpool          Y[o] = sum_{(o,s)} X[s]                       X [n_coarse,C] -> Y [n_fine,C]
 *   unpool_bwd      dX[s] = sum_{(o,s)} dY[o]                     dY [n_fine,C] -> dX [n_coarse,C]
 * A coarse row with no member yields 0 (the reference yields 0/0 = NaN there).
 * ------------------------------------------------------------------------- */
SG_API int sg_pool_create(const int64_t* fine, const int64_t* coarse, int64_t n, int64_t n_fine,
                   int64_t n_coarse, void* stream, sg_pool** out);
SG_API int sg_pool_destroy(sg_pool* p);
SG_API int sg_pool_mean(const sg_pool* p, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t C,
                 int dtype, void* stream);
SG_API int sg_pool_mean_bwd(const sg_pool* p, const void* dY, int64_t lddy, void* dX, int64_t lddx,
                     int64_t C, int dtype, void* stream);
SG_API int sg_unpool(const sg_pool* p, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t C,
              int dtype, void* stream);
SG_API int sg_unpool_bwd(const sg_pool* p, const void* dY, int64_t lddy, void* dX, int64_t lddx,
                  int64_t C, int dtype, void* stream);

/* ------------------------------------------------------------------------- *
 * Row gather / scatter for halo exchange (new, no reference
 * counterpart): Y[i] = X[rows[i]] packs boundary rows into a contiguous send
 * buffer; rows int32 [n] device.
 * ------------------------------------------------------------------------- */
SG_API int sg_gather_rows(const int32_t* rows, int64_t n, const void* X, int64_t ldx, void* Y,
                   int64_t ldy, int64_t C, int dtype, void* stream);

/* ------------------------------------------------------------------------- *
 * BatchNorm over the vertex axis fused with LeakyReLU -- replaces the
 * nn.BatchNorm1d + nn.LeakyReLU pair after every ChebConv (util/networks.py:
 * 43-45,50-51; util/meshnet.py:41-62,107-128,226-243), forward and backward.
 * X / H / dA / Y / dH are [V, C] row-major (row strides in elements), float32
 * or bfloat16; all per-channel vectors are float32 [C] on the device.
 * `slope` is the negative slope (0.01 LeakyReLU, 0 ReLU, 1 identity).
 *
 *   sg_col_blocks(V)      number of row blocks nb the two reductions use
 *   sg_col_moments        partial[b][0][c] = mean, partial[b][1][c] = sum (x-mean)^2 of the
 *                         rows of block b = [b*rpb, min(V,(b+1)*rpb)), rpb = ceil(V/nb);
 *                         the caller merges the nb (x world-size) partials (Chan et al.)
 *   sg_scale_shift_act    Y = act(scale*X + shift)           (scale = gamma*invstd, shift = beta - mean*scale)
 *   sg_bn_act_bwd_reduce  partial[b][0][c] = sum dz, partial[b][1][c] = sum dz*xhat,
 *                         dz = dA * act'(scale*H + shift), xhat = (H - mean)*invstd
 *   sg_bn_act_bwd_apply   dH = k * (dz - c1 - xhat*c2)       (training: k = gamma*invstd,
 *                         c1 = sum dz / N, c2 = sum dz*xhat / N; eval: k = scale, c1 = c2 = 0)
 *   sg_bn_merge           stats[0][c] = mean, stats[1][c] = sum (x-mean)^2 over all V rows from
 *                         the nb partials of sg_col_moments (merged in double)
 *   sg_bn_finalize        from stats [2,C] and the vertex count N (all ranks'): out[0..3][c] =
 *                         mean, invstd, scale = gamma*invstd, shift = beta - mean*scale; when
 *                         running_mean / running_var are not NULL they are updated like
 *                         nn.BatchNorm1d does (momentum, unbiased variance)
 *   sg_bn_stats_finalize  sg_bn_merge + sg_bn_finalize in one launch for N = V (one device)
 *   sg_bn_bwd_coeffs      from the nb partials of sg_bn_act_bwd_reduce: out[0][c] = sum dz (= d bias),
 *                         out[1][c] = sum dz*xhat (= d weight), out[2] = out[0]/N, out[3] = out[1]/N
 *                         (the c1, c2 of sg_bn_act_bwd_apply), out[4][c] = gamma*invstd (its k)
 *   sg_bn_finalize_ranks  vertex partition: all [world, 2C+1] = every rank's (mean[C], M2[C], row count) after an
 *                         all-gather -> out [4,C] as sg_bn_finalize for the whole mesh, out_n[0] = total row count
 *                         (kept on the device: no host round trip per BatchNorm)
 * partial is float32 [nb, 2, C].
 * batches_tracked (sg_bn_stats_finalize[_tiles], sg_bn_finalize_ranks; may be NULL): nn.BatchNorm1d's int64
 * num_batches_tracked scalar on the device, incremented by the same launch.
 * acc_dweight / acc_dbias (sg_bn_bwd_coeffs; may be NULL): float32 [C] gradient accumulators of the BatchNorm weight and
 * bias (their .grad): += out[1] / += out[0] in the same launch, instead of one autograd add per parameter.
 * count_dev (sg_bn_bwd_coeffs; may be NULL): the row count N as a float32 scalar ON THE DEVICE (sg_bn_finalize_ranks'
 * out_n), used instead of the host value N -- a vertex partition calls the kernel once on its partials (nb blocks: the
 * local sums, out[0..1]) and, after the all-reduce of those sums, once more with them as a single block (nb = 1) for
 * c1, c2 and k of the whole mesh.
 * ------------------------------------------------------------------------- */
/* Products with a tiny weight matrix (sg_thin_supported(N, K): K <= 8 with N <= 32, or N, K <= 16, or K <= 32 with N <= 8 -- at
 * most 256 entries) -- replace the `lins[k]` calls of the 4 -> 16 input layer ([3P]
 * ChebConv.forward from util/networks.py:42: K = 3 x 4 = 12 columns, no multiple of an MFMA step), `nn.Linear(16, 3)`
 * (util/networks.py:36,55), the `nn.Linear(32, 3)` heads of the MGCN (util/meshnet.py:228,236,244) and their autograd products,
 * the last ones of the iteration on the BLAS library:
 *   sg_thin_nt:  Y[V, N] = X[V, K] * W[N, K]^T (+ bias)   X, Y float32 or bfloat16 (dtype, same for both; row strides in
 *                elements), W [N, K] (row stride ldw) and bias float32; fp32 accumulation in ascending k, bias added last.
 *   sg_thin_tn:  out[N, K] (float32, row stride ldo) = A[V, N]^T * B[V, K], A / B float32 or bfloat16; workspace: float32
 *                [sg_thin_tn_blocks(V)][256] per-block partial sums, added in block order: deterministic. */
SG_API int sg_thin_supported(int64_t N, int64_t K);
SG_API int64_t sg_thin_tn_blocks(int64_t V);
SG_API int sg_thin_nt(const void* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, void* Y, int64_t ldy,
                      int64_t V, int64_t N, int64_t K, int dtype, void* stream);
SG_API int sg_thin_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t V, int64_t N, int64_t K, int dtype,
                      float* workspace, float* out, int64_t ldo, void* stream);
/* sg_bn_act_bwd_apply that also leaves colsum[c] = sum over rows of the dH it wrote (as stored) -- the bias gradient of
 * the ChebConv in front of the BatchNorm (autograd of `out += bias`, [3P] ChebConv.forward): no separate pass over dH.
 * colsum_partial: float32 [sg_col_apply_blocks(V, C, dtype), C] scratch; that count is 0 for shapes the row-owning
 * kernel does not serve (then SG_ERR_UNSUPPORTED: use sg_bn_act_bwd_apply and sum the columns separately). */
SG_API int64_t sg_col_apply_blocks(int64_t V, int64_t C, int dtype);
SG_API int sg_bn_act_bwd_apply_colsum(const void* dA, int64_t ldda, const void* H, int64_t ldh, const float* scale,
                                      const float* shift, const float* mean, const float* invstd, const float* k,
                                      const float* c1, const float* c2, float slope, void* dH, int64_t lddh, int64_t V,
                                      int64_t C, int dtype, float* colsum_partial, float* colsum, void* stream);
SG_API int64_t sg_col_blocks(int64_t V);
SG_API int sg_col_moments(const void* X, int64_t ldx, int64_t V, int64_t C, int dtype, float* partial,
                          int64_t nb, void* stream);
SG_API int sg_bn_merge(const float* partial, int64_t nb, int64_t V, int64_t C, float* stats, void* stream);
SG_API int sg_bn_stats_finalize(const float* partial, int64_t nb, int64_t V, int64_t C, const float* gamma,
                                const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                float* out, int64_t* batches_tracked, void* stream);
SG_API int sg_bn_bwd_coeffs(const float* partial, int64_t nb, int64_t C, double N, const float* gamma,
                            const float* invstd, float* out, float* acc_dweight, float* acc_dbias, const float* count_dev,
                            void* stream);
/* sg_bn_merge for partials cut into uniform tiles of rows_per_tile rows (sg_col_moments' blocks: ceil(V / nb); the per-tile
 * moments of sg_gemm_nt: sg_gemm_tile_rows(N)); count_out (may be NULL): [1] = (float)V -- together with stats [2, C] the
 * (mean, M2, row count) row a rank contributes to the all-gather of a vertex-partitioned BatchNorm. */
SG_API int sg_bn_merge_tiles(const float* partial, int64_t n_tiles, int64_t rows_per_tile, int64_t V, int64_t C, float* stats,
                             float* count_out, void* stream);
SG_API int sg_bn_finalize_ranks(const float* all, int64_t world, int64_t C, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, float* out,
                                float* out_n, int64_t* batches_tracked, void* stream);
/* dst_s[r, c] += src_s[r * src_ld[s] + c] (float32, dst_s contiguous [rows[s], cols[s]]) for n <= 8 small matrices in ONE
 * launch: a layer's parameter gradients -- the K column (or row) blocks of dWcat = dOut^T [Tx0|Tx1|Tx2] and the bias sums
 * (autograd of util/networks.py:42,49) -- added into the parameters' .grad accumulators (sgcn.py:123-146 accumulates
 * five backward passes per optimiser step).  The pointer / size arrays are host arrays. */
SG_API int sg_multi_add(int64_t n, const float* const* srcs, const int64_t* src_ld, const int64_t* rows, const int64_t* cols,
                        float* const* dsts, void* stream);
SG_API int sg_bn_finalize(const float* stats, double N, int64_t C, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, float* out,
                          void* stream);
SG_API int sg_scale_shift_act(const void* X, int64_t ldx, const float* scale, const float* shift, float slope,
                              void* Y, int64_t ldy, int64_t V, int64_t C, int dtype, void* stream);
SG_API int sg_bn_act_bwd_reduce(const void* dA, int64_t ldda, const void* H, int64_t ldh, const float* scale,
                                const float* shift, const float* mean, const float* invstd, float slope,
                                float* partial, int64_t nb, int64_t V, int64_t C, int dtype, void* stream);
SG_API int sg_bn_act_bwd_apply(const void* dA, int64_t ldda, const void* H, int64_t ldh, const float* scale,
                               const float* shift, const float* mean, const float* invstd, const float* k,
                               const float* c1, const float* c2, float slope, void* dH, int64_t lddh, int64_t V,
                               int64_t C, int dtype, void* stream);

/* ------------------------------------------------------------------------- *
 * Mesh connectivity and hole masks -- the producers of the path's inputs, on the
 * device.  Replace Mesh.build_gemm (util/mesh.py:60-100: self.edges, whose order
 * edge_index inherits at util/mesh.py:229-230), the face 1-ring f2f
 * (util/mesh.py:214-227), one ring of dummy-mask dilation
 * Mv1 = (AdjI @ Mv0) > 0 (util/datamaker.py:123-127) and the vertex->face mask
 * (f2v_mat @ (1 - vmask)) == 0 (util/datamaker.py:136,156-159;
 * util/meshnet.py:179,196).  The reference does these with Python loops over
 * faces and dense V x V matrices.
 *
 * sg_mesh_edges: faces int64 [F,3] (device) -> edges_out int64 [n,2] (device,
 *   capacity 3F rows): the distinct (lo, hi) vertex pairs in the order a
 *   face-by-face scan of (f0,f1),(f1,f2),(f2,f0) first meets them;
 *   *n_edges_out (host) = n.  f2f_out (nullable) int64 [F,3]: the faces across
 *   the face's edges, -1 padded at the end of the row (order inside a row is
 *   by the face's own edge number; the reference's is Python-set order).
 *   *manifold_out (host, nullable) = 0 when some edge has more than two faces
 *   (then f2f is incomplete there).  Synchronises the stream.
 * sg_mask_dilate: masks are bit-packed, W 64-bit words per vertex (bit b of word w
 *   = mask 64 w + b); out[v] = in[v] | OR_{j adjacent to v} in[j].  in != out.
 * sg_face_mask: fbits[f] = vbits[f0] & vbits[f1] & vbits[f2] (bits = "kept").
 * ------------------------------------------------------------------------- */
SG_API int sg_mesh_edges(const int64_t* faces, int64_t F, int64_t V, int64_t* edges_out, int64_t* f2f_out,
                         int64_t* n_edges_out, int* manifold_out, void* stream);
SG_API int sg_mask_dilate(const sg_graph* g, const uint64_t* in, uint64_t* out, int64_t W, void* stream);
SG_API int sg_face_mask(const int64_t* faces, int64_t F, int64_t V, const uint64_t* vbits, uint64_t* fbits,
                        int64_t W, void* stream);

/* ------------------------------------------------------------------------- *
 * Loss step of the training loop, fused -- replaces Models.compute_fn
 * (util/models.py:121-126), Loss.mask_pos_rec_loss (util/loss.py:14-34, 'rmse')
 * and Loss.mask_norm_rec_loss (util/loss.py:78-107, 'l1mae') as sgcn.py:130-132
 * calls them on the positions the network just produced.
 *   pos [V_ext,3] float32 (V owned rows first, then halo rows when partitioned),
 *   faces int64 [F,3] indices into pos, target_pos [V,3], v_keep [V] (1 = kept),
 *   target_fn [F,3] unit normals, f_keep [F].
 * forward : partial[b] = (sum keep_v |p - t|^2 , sum keep_f |n - n_t|_1) of block b,
 *           nb = sg_mesh_loss_blocks(V, F) blocks; the caller sums them, then
 *           loss = sqrt(S_p / n_v + 1e-6) + k1 * S_n / n_f.
 * backward: grad_pos [V_ext,3] = g[0] * dS_p/dpos + g[1] * dS_n/dpos (g: 2 floats on
 *           the device); fully overwritten; the face term uses float atomics.
 * sg_mesh_loss_bwd_det: the same gradient without atomics (bit-reproducible): the three
 *           corner gradients of every face are written to corner_scratch [3F,3] and summed
 *           per vertex in ascending corner order through `incidence`, an sg_pool created
 *           with fine = 0..3F-1 (corner ids 3f+i), coarse = faces[f][i], n_fine = 3F,
 *           n_coarse = V_ext.
 * ------------------------------------------------------------------------- */
/* sg_mesh_loss_finalize: out[0] = w_pos * sqrt(S_p / n_v + 1e-6) + k1 * S_n / n_f from the nb block partials of
 * sg_mesh_loss_fwd (the loss of sgcn.py:130-138; w_pos, k1 = 0 / n_f = 0: one resolution's weighted position term of
 * mgcn.py:138-143), out[1] = d loss / d S_p, out[2] = d loss / d S_n: scaled by the incoming gradient they are the `g` of
 * the backward entry points.  One launch instead of a dozen scalar operators and their autograd nodes. */
SG_API int sg_mesh_loss_finalize(const float* partial, int64_t nb, float n_v, float n_f, float w_pos, float k1, float* out,
                                 void* stream);
SG_API int64_t sg_mesh_loss_blocks(int64_t V, int64_t F);
SG_API int sg_mesh_loss_fwd(const float* pos, const int64_t* faces, const float* target_pos, const float* v_keep,
                            const float* target_fn, const float* f_keep, int64_t V, int64_t F, float* partial,
                            void* stream);
SG_API int sg_mesh_loss_bwd_det(const float* pos, const int64_t* faces, const float* target_pos, const float* v_keep,
                                const float* target_fn, const float* f_keep, const float* g, int64_t V, int64_t V_ext,
                                int64_t F, const sg_pool* incidence, float* corner_scratch, float* grad_pos, void* stream);
SG_API int sg_mesh_loss_bwd(const float* pos, const int64_t* faces, const float* target_pos, const float* v_keep,
                            const float* target_fn, const float* f_keep, const float* g, int64_t V, int64_t V_ext,
                            int64_t F, float* grad_pos, void* stream);

/* ------------------------------------------------------------------------- *
 * Launch tuning of the aggregation kernel (process-wide, not thread-safe; for
 * benchmarking -- results never depend on it).
 * ------------------------------------------------------------------------- */
/* ------------------------------------------------------------------------- *
 * Dense per-vertex feature x weight product on the matrix cores (MFMA):
 *     C[M, N] = A[M, K] * B[N, K]^T (+ bias[N])     bf16 operands and result, fp32 accumulate
 * Replaces the K bias-free `lins[k](Tx_k)` + `out += bias` of ChebConv.forward [3P torch_geometric 2.2.0]
 * (call sites util/networks.py:42,49; util/meshnet.py:40-240): A = [Tx0|Tx1|Tx2] ([V, K*Cin], written by sg_spmm),
 * B = [W0|W1|W2] ([Cout, K*Cin]); with B = the transposed weights it is their input gradient dT = dOut * Wcat
 * (autograd of the same call sites).  Row strides lda / ldb / ldc are in elements; A, B, C 16-byte aligned,
 * K, N and the strides multiples of 8 (otherwise SG_ERR_UNSUPPORTED: the caller keeps its BLAS call).
 * moments (nullable): float32 [sg_gemm_row_tiles(M, N), 2, N]; tile t receives the per-column mean and
 * sum (x - mean)^2 of rows [t*R, (t+1)*R) of the ROUNDED result, R = sg_gemm_tile_rows(N) -- the block moments
 * sg_bn_stats_finalize_tiles merges, so BatchNorm needs no separate pass over C (util/networks.py:43).
 * Two kernels serve it: a 128-row tile that streams A and C (the HBM-bound layers) and, for the compute-bound products
 * (K * N > 100 K weight elements, K % 64 == 0, N % 256 == 0, M >= 16 K: the 256- and 512-channel layers), a persistent
 * 256 x 256 tile with eight wavefronts (csrc/gemm_mfma256.hip).  The latter does not emit `moments`: a call that passes
 * moments != NULL is served by the 128-row kernel; sg_gemm_nt_takes_big_tile tells which kernel a moments-free call gets.
 * ------------------------------------------------------------------------- */
SG_API int sg_gemm_nt_takes_big_tile(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc);
SG_API int64_t sg_gemm_tile_rows(int64_t N);             /* R for an N-column product (128, or 64 for wide outputs) */
SG_API int64_t sg_gemm_row_tiles(int64_t M, int64_t N);  /* ceil(M / R) */
SG_API int sg_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, const float* bias, void* C, int64_t ldc,
                      int64_t M, int64_t N, int64_t K, int dtype, float* moments, void* stream);
/* Weight gradient on the matrix cores:  out[N, Kp] (float32, row stride ldo) = A[M, N]^T * B[M, Kp], bf16 operands --
 * dWcat = dOut^T [Tx0|Tx1|Tx2], the autograd of the `lins[k]` calls (util/networks.py:42,49), a reduction over all M
 * vertices.  workspace: float32 [sg_gemm_tn_slabs(M, N, Kp), N, Kp] (per-slab partial sums, added in slab order: the result
 * is deterministic).  N, Kp, lda, ldb multiples of 8, 16-byte aligned buffers, else SG_ERR_UNSUPPORTED. */
SG_API int64_t sg_gemm_tn_slabs(int64_t M, int64_t N, int64_t Kp);
/* 1 when sg_gemm_tn serves this shape with the persistent 256 x 256 ring (N, Kp multiples of 256, N * Kp > 90 K, M >= 16 K:
 * the weight gradients of the 256- and 512-channel layers), 0 when the 128 x 128 kernel does */
SG_API int sg_gemm_tn_takes_big_tile(int64_t M, int64_t N, int64_t Kp, int64_t lda, int64_t ldb);
SG_API int sg_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t Kp, int dtype,
                      float* workspace, float* out, int64_t ldo, void* stream);
/* ------------------------------------------------------------------------- *
 * The same two products at the reference's OWN precision -- float32 features, float32 parameters (util/networks.py:40-53;
 * BASELINE configs c2 / c3; [3P] ChebConv `lins[k]` and their autograd) -- on the bf16 matrix cores: every float32
 * operand is split exactly into three bf16 pieces (x = hi + mid + lo, 3 x 8 significand bits) and a product is
 * accumulated in float32 from the six piece products of weight >= 2^-16 (csrc/gemm_split.hip).  Error against a
 * float64 product: that of a float32 FMA chain (<= 1.5e-7 sum |a b| at K <= 1024); small-integer operands bit for bit.
 *   sg_gemm_nt_f32:  C[M, N] = A[M, K] op(W) (+ bias[N]); W element (n, k) at W[n * w_rs + k * w_cs] -- (ldw, 1) for the
 *                    [N, K] weights of the forward product, (1, ldw) for a [K, N] matrix (the input gradient dOut * Wcat
 *                    without a transposed copy).  workspace: sg_gemm_nt_f32_workspace(N, K) bytes (the split image of W,
 *                    rebuilt by every call: microseconds).  K % 32 == 0, K >= 64, N % 4 == 0, N >= 64, M >= 128, row
 *                    strides multiples of 4, 16-byte aligned buffers: sg_gemm_nt_f32_supported; else SG_ERR_INVALID.
 *   sg_gemm_tn_f32:  out[N, Kp] (row stride ldo) = A[M, N]^T B[M, Kp], a reduction over all M vertices; workspace: float32
 *                    [sg_gemm_tn_f32_slabs(M, N, Kp), N_pad, Kp_pad] slab partials (sg_gemm_tn_f32_workspace bytes), added
 *                    in slab order: deterministic.  N, Kp, lda, ldb multiples of 4, ldo % 4 == 0, M >= 4096.
 * Weight matrices of 4 .. 48 rows and columns (multiples of 4; the 16 -> 32 and 32 -> 16 layers) are HBM-bound and below the
 * matrix-core tile: the same entry points run them as plain float32 FMA chains on the vector ALUs (csrc/gemm_mid.hip; any M,
 * no workspace for nt, sg_gemm_tn_f32_workspace bytes of block partials for tn, same deterministic reduce).
 * ------------------------------------------------------------------------- */
SG_API int sg_gemm_nt_f32_supported(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc);
/* 1 when sg_gemm_nt_f32 is the faster choice for a supported product of this size, 0 when the caller's BLAS library is: the ONE
 * place that rule lives (sg_block_* and the per-module host path both ask here).  Since round 6 a product with fewer than 16 K
 * rows runs on 128 x 128 tiles (four wavefronts, two workgroups per CU); the answer is 0 only where the A/B still loses: fewer
 * than 128 such tiles over K >= 384 (at 5 K rows: [V,768]x[768,256] and [V,384]x[384,256]).  SG_TUNE_F32_ENGINE bit 5 switches
 * the variant off (then: 1 from 16 384 rows on, or always with bit 4). */
SG_API int sg_gemm_nt_f32_pays(int64_t M, int64_t N, int64_t K);
/* Which tile shape -- and therefore which layout of the split weight image -- serves a product of M rows: 1 = 128-row tiles,
 * 0 = 256-row tiles.  An image built for one variant (sg_block::wsplit with that block's V) must not be handed to a block whose
 * V falls into the other: a host that applies one set of weights to meshes of both size classes keeps an image per variant. */
SG_API int sg_gemm_nt_f32_variant(int64_t M);
SG_API int64_t sg_gemm_nt_f32_workspace(int64_t N, int64_t K);
SG_API int sg_gemm_nt_f32(const float* A, int64_t lda, const float* W, int64_t w_rs, int64_t w_cs, const float* bias, float* C,
                          int64_t ldc, int64_t M, int64_t N, int64_t K, void* workspace, int64_t workspace_bytes, void* stream);
SG_API int sg_gemm_tn_f32_supported(int64_t M, int64_t N, int64_t Kp, int64_t lda, int64_t ldb);
SG_API int64_t sg_gemm_tn_f32_workspace(int64_t M, int64_t N, int64_t Kp);
SG_API int sg_gemm_tn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int64_t Kp,
                          void* workspace, int64_t workspace_bytes, float* out, int64_t ldo, void* stream);
/* sg_bn_stats_finalize for partials cut into uniform tiles of rows_per_tile rows (the last one shorter) */
SG_API int sg_bn_stats_finalize_tiles(const float* partial, int64_t n_tiles, int64_t rows_per_tile, int64_t V, int64_t C,
                                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                                      float momentum, float eps, float* out, int64_t* batches_tracked, void* stream);

/* ---------------------------------------------------------------------------
 * The network's input step, fused (SingleScaleGCN.forward, util/networks.py:65-79: bounding-box normalisation with one
 * scale and a per-axis centre, masking, the mask appended as 4th channel) and its autograd.
 *   mid = (lo + hi) / 2, extent = max_k (hi[k] - lo[k]);   v = order ? order[p] : p
 *   X[p] = ( dm[v] (z1[v] - mid) / extent, dm[v] )          X: [V, >= 4] of `dtype`, row stride ldx (elements)
 * z1 float32 [V, 3]; dm float32 [V] or NULL (all ones); order / rank int64 [V] or NULL (inverse permutations of each other:
 * processing row p holds vertex order[p], vertex v sits in processing row rank[v]); lo, hi float32 [3] ON THE DEVICE.
 * sg_input_prep_bwd: gX = dL/dX (rows in processing order) -> dz1 [V, 3] (caller order; may be NULL), d_lo / d_hi [3]
 * (the gradients of the bounds, which the caller's autograd routes to the arg-extreme vertices); partial: float32
 * [sg_input_prep_blocks(V), 4] scratch (block sums in a fixed order: deterministic).
 * ------------------------------------------------------------------------- */
/* sg_input_bounds: lo / hi of z1 over the vertices (util/networks.py:67: torch.min / torch.max over dim 0) in two launches:
 * bounds [6] = lo[3], hi[3]; arg [6] = the vertex each bound was taken from (a tie goes to the lowest id); scratch
 * partial_values float32 [sg_input_prep_blocks(V), 6], partial_index int64 [same, 6].
 * sg_input_prep_bwd_routed: sg_input_prep_bwd with lo = bounds, hi = bounds + 3, and the gradients of the bounds (written to
 * d_bounds [6]) ADDED to dz1 at the arg vertices in the same call -- what autograd does through torch.min / torch.max. */
SG_API int sg_input_bounds(const float* z1, int64_t V, float* partial_values, int64_t* partial_index, float* bounds, int64_t* arg,
                           void* stream);
SG_API int sg_input_prep_bwd_routed(const void* gX, int64_t ldg, const float* z1, const float* dm, const int64_t* rank,
                                    const float* bounds, const int64_t* arg, float* dz1, float* partial, float* d_bounds,
                                    int64_t V, int dtype, void* stream);
SG_API int64_t sg_input_prep_blocks(int64_t V);
SG_API int sg_input_prep(const float* z1, const float* dm, const int64_t* order, const float* lo, const float* hi, void* X,
                         int64_t ldx, int64_t V, int dtype, void* stream);
SG_API int sg_input_prep_bwd(const void* gX, int64_t ldg, const float* z1, const float* dm, const int64_t* rank,
                             const float* lo, const float* hi, float* dz1, float* partial, float* d_lo, float* d_hi, int64_t V,
                             int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * One block of the networks in ONE call -- [ChebConv -> (MeshPool | MeshUnpool)? -> BatchNorm1d -> LeakyReLU], the unit
 * SingleScaleGCN is a chain of (util/networks.py:40-46 builds it, :83-101 runs it) and DownConv / UpConv / the MGCN heads
 * are made of (util/meshnet.py:39-62,105-128,223-245; run at :295-312).  The caller fills an sg_block (plain device
 * pointers and sizes: parameters, the saved activations, scratch) and the library launches the whole chain on `stream`:
 *
 *   forward   [pack the K weight matrices]  ->  Tx1 = L^ Tx0, Tx2 = 2 L^ Tx1 - Tx0 (sg_spmm)  ->  H = [Tx0|Tx1|Tx2] Wcat^T + b
 *             (sg_gemm_nt / sg_thin_nt / the BLAS library for fp32 and odd shapes; per-tile BatchNorm moments from the MFMA
 *             epilogue where it emits them, else one sg_col_moments pass)  ->  [pool]  ->  statistics finalised (running
 *             averages, num_batches_tracked)  ->  Y = act(scale H + shift), written where the next block wants its Tx0.
 *             Layers that narrow (order = 1, Cout < Cin) run the product first and Clenshaw's recurrence after it, Cout wide.
 *   backward  BatchNorm/activation sums -> coefficients (+= into the weight / bias .grad) -> dH (+ its column sums = the
 *             ChebConv bias gradient) -> [pool transpose] -> dWcat = dH^T T (sg_gemm_tn / sg_thin_tn / BLAS), added into the
 *             K weight .grad accumulators -> dT = dH Wcat -> the recurrence unwound with two aggregations -> dX.
 *
 * What one call replaces on the host: ~20 foreign calls, three autograd nodes and a dozen small allocations per block and
 * direction (13 blocks per SGCN iteration, 33 per MGCN iteration) -- the reference's own mesh sizes (5 K - 50 K vertices)
 * are bound by exactly that.  The kernels, their arguments and their order are those of the per-operator entry points
 * above, so results are bit-identical to calling those one by one.
 *
 * All pointers are device pointers owned by the caller; `ws` is scratch of at least sg_block_workspace(blk, backward) bytes,
 * 256-byte aligned, free again when the call's work has run on the stream.  Pointers marked (nullable) may be NULL.
 * --------------------------------------------------------------------------- */
typedef struct sg_block {
  /* operators */
  const sg_graph* graph;   /* L^ of the level the ChebConv runs on (V rows) */
  const sg_pool* pool;     /* (nullable) the pool_hash applied between the conv and its BatchNorm (util/meshnet.py:44-47,106-109) */
  int32_t pool_mode;       /* 0 none, 1 MeshPool (cluster mean: V fine rows -> V_out coarse rows), 2 MeshUnpool (gather: V coarse -> V_out fine) */
  int32_t dtype;           /* sg_dtype of every [rows, C] feature buffer below */
  int32_t K;               /* Chebyshev order, 1..3 (the reference uses 3) */
  int32_t order;           /* 0: aggregate, then ONE product on [Tx0|..|Tx(K-1)]; 1: product first, Clenshaw aggregation after (Cout < Cin, K >= 2) */
  int32_t training;        /* BatchNorm mode: 1 batch statistics (+ running-average update), 0 running statistics */
  int32_t refresh_weights; /* != 0: (re)build the packed weight copies from W[] first (a parameter changed since the last call) */
  int32_t need_dx;         /* backward: 0 skips the input gradient */
  int32_t reserved_;
  int64_t V, V_out;        /* rows of the conv / rows after the pool (V_out = V without one) */
  int64_t Cin, Cout;
  float momentum, eps, slope;   /* BatchNorm momentum and eps, negative slope of the activation (0.01 LeakyReLU, 0 ReLU) */
  float reserved2_;
  /* parameters: float32, as the nn.Modules hold them */
  const float* W[3];       /* lins[k].weight [Cout, Cin], contiguous */
  const float* bias;       /* (nullable) [Cout] */
  const float* gamma;      /* BatchNorm weight [Cout] */
  const float* beta;       /* BatchNorm bias [Cout] */
  float* running_mean;     /* (nullable, both or none) updated in training mode, read in eval mode */
  float* running_var;
  int64_t* batches_tracked; /* (nullable) num_batches_tracked, += 1 in training mode */
  /* packed copies of the weights, caller-owned, (re)written by the library when refresh_weights != 0:
   *   wpack    order 0: Wcat [Cout, K*Cin]; order 1: Wstack [K*Cout, Cin]     in the feature dtype
   *   wpack_t  (nullable) its transpose: bf16 features read it in the input-gradient product on the MFMA kernel
   *   wpack32 / wpack32_t  (nullable) the same two in float32 -- for products with a tiny weight matrix (Cout, K*Cin <= 16: sg_thin_*)
   *   bias_k   (order 1 with a bias) the bias padded with zeros to K*Cout floats: it rides in on Z_0
   *   wsplit / wsplit_t  (nullable; float32 features) the split-bf16 images of the weight matrix for the forward and for the
   *            input-gradient product -- sg_gemm_nt_f32_workspace(N, K) bytes each, 16-byte aligned, N x K = Cout x K*Cin and
   *            K*Cin x Cout (order 0) / K*Cout x Cin and Cin x K*Cout (order 1).  Given, the images are rebuilt ONLY when
   *            refresh_weights != 0 (every fifth iteration of the reference's loop) instead of inside each product; NULL: as
   *            before, every float32 product splits its weights into scratch.  The images are specific to the tile variant of V
   *            (sg_gemm_nt_f32_variant) and to the tuning knobs at the time of the refresh: change a knob, refresh. */
  void* wpack;
  void* wpack_t;
  float* wpack32;
  float* wpack32_t;
  float* bias_k;
  void* wsplit;
  void* wsplit_t;
  /* activations (feature dtype, unit column stride, row strides in elements).
   * order 0: T is the [V, K*Cin] buffer whose first Cin columns hold the input on entry -- X == T when the producer wrote it
   * there, else the library copies X in; order 1: T is not used (NULL) and X [V, Cin] is read in place.
   * H [V_out, Cout] is the BatchNorm input (the conv output, pooled when there is a pool), stats float32 [4, Cout] = mean,
   * invstd, scale, shift; Y [V_out, Cout] the block output (ldy: e.g. a column block of the next block's [V, 3C] buffer).
   * The caller keeps T (order 0) or X (order 1), H and stats alive between forward and backward. */
  const void* X; int64_t ldx;
  void* T; int64_t ldt;
  void* H;
  float* stats;
  void* Y; int64_t ldy;
  /* backward only */
  const void* dY; int64_t lddy;   /* gradient of Y [V_out, Cout] */
  void* dX; int64_t lddx;         /* gradient of the input [V, Cin] (nullable when need_dx == 0) */
  float* dW;        /* float32 [Cout, K*Cin] (order 0) / [K*Cout, Cin] (order 1): this call's weight gradient, always written */
  float* dvec;      /* float32 [6, Cout], always written: rows 0..4 = sg_bn_bwd_coeffs' output (0: sum dz = d beta, 1: sum dz xhat =
                       d gamma, 2..4: c1, c2, k), row 5 = column sums of the conv output's gradient = d bias */
  float* acc_W[3];  /* (nullable, all or none) the K weight .grad accumulators [Cout, Cin]: += the blocks of dW in the same call */
  float* acc_bias;  /* (nullable) += dvec row 5 */
  float* acc_gamma; /* (nullable, both or none) += dvec rows 1 / 0 */
  float* acc_beta;
  /* scratch */
  void* ws; int64_t ws_bytes;
  /* ---- one block of a VERTEX PARTITION (no reference counterpart), run phase by phase between the rank's
   * collectives by sg_block_run; all zero / NULL for a block on one device.  The rank owns V rows of every [V, C] tensor;
   * its buffers T / X / H / Y / G have V_ext rows: [owned | per peer: that peer's rows of the two-ring halo, then 5 pad rows].
   * A block's halo exchange carries the rows of H (the conv output, BEFORE BatchNorm) and, in the pad rows of every peer's
   * segment, this rank's BatchNorm statistics of H: the all-gather of the statistics rides in the exchange, and each rank
   * applies BatchNorm + activation to the halo rows it received itself (SG_PHASE_BN), which clears the pad rows of H once
   * it has read them: no buffer keeps statistics bytes where feature values are expected.  Invariant the layout relies on:
   * no column of `graph` / `graph_wide` names a pad row, and every reduction over rows stops at the V owned rows. */
  int32_t phase;               /* bit mask of sg_block_phase */
  int32_t world;               /* ranks */
  const sg_graph* graph_wide;  /* L^ on the owned AND the ring-1 rows: V_ext rows, V_ext columns (`graph`: V rows, V_ext columns) */
  int64_t V_ext;
  int64_t ldh;                 /* row stride of H (= Cout: H is also the receive buffer of its halo rows and statistics) */
  float* local;                /* float32 [2 Cout + 1]: this rank's (mean, M2, row count) of H's owned rows */
  const int64_t* stats_rows;   /* device int64 [world]: first pad row of peer q's segment in H (its statistics); -1 for this rank */
  float* gathered;             /* float32 [world, 2 Cout + 1]: every rank's statistics (SG_PHASE_BN fills it from H unless ...) */
  int32_t gathered_ready;      /* ... != 0: the caller filled it (an all-gather: the last block has no exchange behind it) */
  int32_t reserved3_;
  float* count;                /* float32 [1] on the device: the mesh's row count (written by SG_PHASE_BN, read by the backward phases) */
  const int32_t* send_index;   /* device int32 [n_send]: the owned row every send row is a copy of; -1 - j: pad row j of a segment */
  int64_t n_send;
  void* send;                  /* [n_send, Cout] after SG_PHASE_CONV; [n_send, 2 Cin] (order 0) / [n_send, Cout] (order 1) after SG_PHASE_BWD_A */
  const void* recv;            /* SG_PHASE_BWD_B: the halo rows of the gradient blocks as received, [V_ext - V, width of `send`] */
  void* G;                     /* [V_ext, K * Cin] (order 0) / [V_ext, K * Cout] (order 1): lives from SG_PHASE_BWD_A to SG_PHASE_BWD_B */
} sg_block;
enum sg_block_phase {
  SG_PHASE_CONV = 1,        /* the conv on [owned | halo] input rows -> H (owned rows), `local`, `send` */
  SG_PHASE_BN = 2,          /* statistics of the whole mesh from `gathered` -> stats, count; Y = act(BN(H)) on V_out rows (V_ext: all) */
  SG_PHASE_BWD_REDUCE = 4,  /* dvec rows 0, 1 = this rank's (sum dz, sum dz xhat) (+= into acc_beta / acc_gamma) -- all-reduce them */
  SG_PHASE_BWD_A = 8,       /* from the all-reduced dvec rows 0, 1: dH, d bias; dW (+= acc_W); the input-gradient blocks -> G, `send` */
  SG_PHASE_BWD_B = 16       /* `recv` -> G's halo rows; the recurrence unwound -> dX (owned rows) */
};
SG_API int64_t sg_block_sizeof(void);   /* sizeof(sg_block) of the library (a binding checks its mirror against it) */
SG_API int64_t sg_block_workspace(const sg_block* blk, int backward);
/* Narrow layers keep the K column blocks of their recurrence buffers as K dense [V, C] PLANES instead of column blocks of one
 * [V, K*C] buffer (an aggregation that gathers 32-byte rows from a 96-byte pitch uses a quarter of every line it pulls):
 * bf16 features, C = 8 .. 64 a power of two, every product of the layer on the library's 128-row matrix-core kernels.
 * For a layer that aggregates after the product the buffers are the library's own scratch and nothing changes for the
 * caller.  For a layer that aggregates first (order 0) the buffer is the caller's T: sg_block_planar(blk) answers 1 when the
 * shape (dtype, V, Cin, Cout, K, order of blk) qualifies, and the caller chooses the layout by passing T as [K][V][Cin]
 * with ldt = Cin (X = T, ldx = Cin: plane 0 is the input; the block in front writes it with ldy = Cin).  ldt >= K*Cin keeps
 * the column-block layout for any shape.  Same arithmetic, same results (tests/test_gpu_blocks.py). */
SG_API int sg_block_planar(const sg_block* blk);
SG_API int sg_block_forward(const sg_block* blk, void* stream);
SG_API int sg_block_backward(const sg_block* blk, void* stream);
/* A run of n consecutive blocks in one call -- the loop of SingleScaleGCN.forward over its 13 blocks (util/networks.py:83-101),
 * the five convolutions of a DownConv / UpConv stage (util/meshnet.py:92-95,157-160): forward runs blks[0] .. blks[n-1],
 * backward blks[n-1] .. blks[0].  The caller wires the descriptors: blks[i+1].X = blks[i].Y (written straight into the
 * next block's T when that one aggregates first), blks[i].dY = blks[i+1].dX; the blocks run one after the other on `stream`,
 * so they may share one scratch area sized for the largest of them. */
/* sg_block_run: the phases of n partition blocks in array order (blks[i].phase), one call between two collectives.
 * sg_block_workspace(blk, 2): scratch a partition block needs for any of its phases. */
SG_API int sg_block_run(const sg_block* blks, int64_t n, void* stream);
SG_API int sg_block_chain_forward(const sg_block* blks, int64_t n, void* stream);
SG_API int sg_block_chain_backward(const sg_block* blks, int64_t n, void* stream);

/* ---- The rank's collectives below the C ABI (SURVEY.md section 8(b) `sg_halo_exchange`, section 8(e)).  No reference
 * counterpart: the reference is one process on one device (util/networks.py:83-101 runs its 13 blocks with no exchange
 * step); a vertex partition adds one exchange of boundary rows per block and direction, and the all-reduce / all-gather of
 * the BatchNorm statistics that couple all vertices (util/networks.py:43).  Replaces, for the phase path, the per-collective
 * torch.distributed calls of semigcn_amd/dist.py (_PartChainFn.forward / .backward: 13 + 1 forward, 26 backward).
 *
 * sg_comm wraps ONE RCCL communicator, bound at run time by dlopen from the copy of librccl.so already in the process (or
 * /opt/rocm/lib): sg_comm_available() == 0 when none can be loaded, and every call below then answers SG_ERR_UNSUPPORTED.
 * Rank 0 draws a 128-byte id (sg_comm_unique_id) and the host passes it to the other ranks (e.g. through torch.distributed's
 * store); every rank then calls sg_comm_create -- collectively, with the HIP device it will use current -- with the rows it
 * sends to / receives from each peer in ONE halo exchange (pad rows included; [world] each; a partition has 0 for itself,
 * equal counts for itself are a copy).
 * All collectives are enqueued on `stream`, the stream of the kernels around them: no event hand-off, no host wait.
 *   sg_halo_exchange       all-to-all with the communicator's per-peer row counts: `send` = [sum send_rows, row_bytes] packed by
 *                          peer, `recv` = [sum recv_rows, row_bytes] (rows [V:] of a partition block's H / its `recv`)
 *   sg_comm_all_reduce_f32 in-place sum of n floats (a block's two BatchNorm backward sums: dvec rows 0, 1)
 *   sg_comm_all_gather     bytes_per_rank from every rank, in rank order (the last block's BatchNorm statistics)
 *   sg_part_run            a schedule: steps[i] is a run of partition-block phases (sg_block_run) or one of the three
 *                          collectives; a rank's whole forward (or backward) pass over its blocks is one call.  `comm` may be
 *                          NULL when no step is a collective. */
typedef struct sg_comm sg_comm;
enum sg_part_step_kind { SG_STEP_BLOCKS = 0, SG_STEP_EXCHANGE = 1, SG_STEP_ALL_REDUCE = 2, SG_STEP_ALL_GATHER = 3 };
typedef struct sg_part_step {
  int32_t kind;            /* sg_part_step_kind */
  int32_t reserved_;
  const sg_block* blocks;  /* SG_STEP_BLOCKS: n descriptors, run in array order */
  int64_t n;               /* BLOCKS: descriptors; EXCHANGE: bytes per row; ALL_REDUCE: floats; ALL_GATHER: bytes per rank */
  const void* send;        /* EXCHANGE, ALL_GATHER: what this rank contributes */
  void* recv;              /* EXCHANGE, ALL_GATHER: where the peers' rows land; ALL_REDUCE: the float32 buffer (in place) */
} sg_part_step;
SG_API int sg_comm_available(void);
SG_API int sg_comm_unique_id(void* id128);
SG_API int sg_comm_create(const void* id128, int rank, int world, const int64_t* send_rows, const int64_t* recv_rows, sg_comm** out);
/* A second handle on the SAME RCCL communicator with other per-peer row counts (one exchange layout per MGCN level): no
 * ncclCommInitRank, no new buffers; the communicator is destroyed with its last handle. */
SG_API int sg_comm_share(const sg_comm* base, const int64_t* send_rows, const int64_t* recv_rows, sg_comm** out);
SG_API int sg_comm_destroy(sg_comm* comm);
SG_API int sg_halo_exchange(sg_comm* comm, const void* send, void* recv, int64_t row_bytes, void* stream);
SG_API int sg_comm_all_reduce_f32(sg_comm* comm, float* buf, int64_t n, void* stream);
SG_API int sg_comm_all_gather(sg_comm* comm, const void* in, void* out, int64_t bytes_per_rank, void* stream);
SG_API int64_t sg_part_step_sizeof(void);
SG_API int sg_part_run(sg_comm* comm, const sg_part_step* steps, int64_t n, void* stream);
/* After a failed sg_part_run on this thread: the index of the step that failed (steps before it are enqueued, so this rank's
 * collective sequence is out of step with its peers' and the job must be aborted, not retried on another communicator);
 * -1 after a successful run. */
SG_API int64_t sg_part_failed_step(void);
/* TEST INFRASTRUCTURE (tests/test_comm_stub.py; never called by the product): sg_comm_test_stub(1) replaces RCCL by an
 * in-process stand-in -- every communicator created afterwards is a "rank" of THIS process, buffers are host memory, sends
 * meet their receives in a mailbox, an all-reduce / all-gather completes when the last rank has called -- so that the peer
 * offsets of sg_halo_exchange and the schedule walk of sg_part_run can be checked for world sizes > 1 on a box with no (or
 * one) GPU; sg_comm_test_stub(0) restores the real library.  sg_comm_test_fail_send(n): the n-th ncclSend from now fails.
 * sg_comm_test_log: the calls made so far as records of five int64 {kind 0 send / 1 recv / 2 all-reduce / 3 all-gather /
 * 4 group start / 5 group end, rank, peer, pointer, bytes}; state[3] = {operations still unmatched, size mismatches, groups
 * left open}. */
SG_API int sg_comm_test_stub(int on);
SG_API int sg_comm_test_fail_send(int nth);
SG_API int64_t sg_comm_test_log(int64_t* out, int64_t n, int64_t* state);

/* Per-launch timing of the kernels the library starts (benchmarking aid; off by default, costs two hipEventRecord per kernel
 * when on).  sg_trace_begin(capacity, kinds) starts recording up to `capacity` launches of the kinds in the bit mask
 * `kinds` (bit k = record kind k) made from this process' sg_spmm / sg_block_* calls: an event pair on the launching stream around each aggregation and each dense product.  After the caller
 * has synchronised the device, sg_trace_read copies up to n records out (returns their number, negative on error) and
 * sg_trace_end releases the events.  Record: kind 0 = aggregation (a = C, b = epilogue operands, c = rows processed),
 * 1 = product C = A B^T / A B (a = M, b = N, c = K), 2 = weight gradient A^T B (a = M, b = N, c = Kp), 3 = MeshPool / MeshUnpool pass and their transposes (sg_pool_mean,
 * sg_unpool, *_bwd: a = C, b = rows written, c = rows read); engine 0 = the
 * library's aggregation kernels, 1 = own MFMA kernels, 2 = thin-product and small-weight kernels (vector ALUs: thin_gemm.hip, gemm_mid.hip), 3 = BLAS library, 4 = split-bf16 MFMA kernels
 * (float32 features). */
typedef struct sg_trace_record {
  int32_t kind, dtype, engine, reserved_;
  int64_t a, b, c;
  float ms;
  float reserved2_;
} sg_trace_record;
SG_API int sg_trace_begin(int64_t capacity, int kinds);
SG_API int64_t sg_trace_read(sg_trace_record* out, int64_t n);
SG_API int sg_trace_end(void);

enum sg_tune_knob {
  SG_TUNE_CHUNK_ROWS = 0, /* rows per wavefront chunk; 0 = automatic */
  SG_TUNE_FLAGS = 1,      /* bit 0: XCD-contiguous tile map, bit 1: never use the shared-gather kernel,
                             bit 2: ignore the (id, scale) packed neighbour lists, bit 3: workgroup barriers
                             between the staging phases instead of wavefront-local ones, bit 4: 64-bit gather addressing
                             even where 32-bit offsets would do, bit 5: fixed-size gather batches also where the
                             row length is wave-uniform, bit 6: nontemporal epilogue loads / stores (no effect measured),
                             bit 7: the experimental unpipelined LDS-tile kernel (set it when the graph is created AND when it is
                             applied; measured slower), bit 8: 4-channel bf16 rows on the one-thread-per-element kernel instead of the
                             one-thread-per-row kernel, bit 11: NO tiled matrix-core kernel (spmm_ring) for bf16 rows of 128 / 256
                             channels (at graph creation: no tile records are built), bit 12: spmm_ring stores straight from the
                             MFMA layout instead of full rows through LDS, bit 13: NO tiled kernel for float32 rows (A/B switches) */
  SG_TUNE_UNROLL = 2,     /* gathers a lane group issues back to back in the aggregation kernel: 8, 6 or 4
                             (fewer = fewer VGPRs = more resident wavefronts); 0 = the shipped choice per shape */
  SG_TUNE_SLAB = 3,       /* channels per column slab (one sweep of all rows per slab); 0 = off */
  SG_TUNE_TILED_MIN_ROW_BYTES = 4, /* shared-gather kernel (each distinct source row of a 4-row mini-tile is
                                     gathered once): default 1024 = fp32 rows of >= 1 KiB where it pays;
                                     negative = force it for every row of at least |value| bytes and both
                                     dtypes; 0 = never, and graphs created from now on carry no mini-tiles */
  SG_TUNE_GRAPH_REORDER = 6, /* sg_graph_create: process the rows in a graph-derived locality order (output rows stay
                                where the caller expects them): 0 = when the vertex numbering has no locality (>= 25 % of
                                the edges span more than 4096 ids, V >= 65536), 1 = never, 2 = always */
  SG_TUNE_GEMM_TILE = 5,  /* sg_gemm_nt kernel: 0 = automatic (shipped: 128 x min(N,128) tiles, the persistent 256 x 256 kernel
                             for the compute-bound products), 1 = 128-row tiles only, 2 = 64 x 256 wherever N > 64 (A/B
                             switch; measured slower), 3 = the 256 x 256 kernel wherever it takes the shape, 4 = no 128 x 192 tiles for
                             N = 192 (A/B switch), 5 = 128 x 192 tiles also for N = 384 (A/B switch; measured slower) */
  SG_TUNE_BLOCK_PLANES = 7, /* sg_block_*: 1 (default) = narrow layers keep their recurrence buffers as planes
                              (sg_block_planar), 0 = column blocks everywhere (A/B switch) */
  SG_TUNE_F32_ENGINE = 8,  /* dense products on float32 features: 0 (default) = the split-bf16 MFMA kernels (csrc/gemm_split.hip)
                              wherever they take the shape, the BLAS library for the rest; bit 0 = the BLAS library for all of
                              them (A/B switch); bits 1 / 2 / 3 = only the forward / input-gradient / weight-gradient products go to the
                              library (bisecting aid); bit 5 = no 128-row tiles: forward / input-gradient products below 16 K rows
                              go to the library as in round 5 (A/B switch) -- unless bit 4 sends them to the 256-row kernel */
  SG_TUNE_BN_ROWS = 9      /* BatchNorm + activation apply passes: 0 (default) = by shape (rows of >= 1 KB, backward passes from 512 B:
                              every workgroup walks ONE contiguous range of rows; else row groups strided over the grid), 1 = contiguous
                              everywhere, 2 = strided everywhere (A/B switch) */
};
SG_API int sg_tuning_set(int knob, int value);

#ifdef __cplusplus
}
#endif
#endif /* SEMIGCN_H */
