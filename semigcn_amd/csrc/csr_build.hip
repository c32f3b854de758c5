// One-off graph preprocessing on the device: COO pairs -> CSR sorted by (dst, src).
//
// Replaces the per-call normalisation of ChebConv.__norm__ [3P torch_geometric 2.2.0]
// (reference call sites: util/networks.py:42,49; util/meshnet.py:40-58,106-124,224-240)
// and the sparse-matrix builders util/meshnet.py:331-341.  Runs once per edge_index /
// pool_hash, so it is written for clarity; the radix sort is hipCUB's.
#include <hipcub/hipcub.hpp>

#include "sg_common.h"

namespace sg {

void Csr::release() {
  if (rowptr) (void)hipFree(rowptr);
  if (idx) (void)hipFree(idx);
  if (tile_uptr) (void)hipFree(tile_uptr);
  if (tile_uniq) (void)hipFree(tile_uniq);
  if (tile_eloc) (void)hipFree(tile_eloc);
  if (idx_w) (void)hipFree(idx_w);
  if (tile_uniq_w) (void)hipFree(tile_uniq_w);
  if (lt_uptr) (void)hipFree(lt_uptr);
  if (lt_uniq) (void)hipFree(lt_uniq);
  if (lt_eloc) (void)hipFree(lt_eloc);
  if (lt_uniq_w) (void)hipFree(lt_uniq_w);
  if (lt_rec) (void)hipFree(lt_rec);
  lt_rec = nullptr;
  lt_nrec = 0;
  lt_uptr = nullptr;
  lt_uniq = nullptr;
  lt_eloc = nullptr;
  lt_uniq_w = nullptr;
  idx_w = nullptr;
  tile_uniq_w = nullptr;
  packed_scale = nullptr;
  rowptr = nullptr;
  idx = nullptr;
  tile_uptr = nullptr;
  tile_uniq = nullptr;
  tile_eloc = nullptr;
  tile_rows = 0;
  nnz = 0;
}

namespace {

constexpr int kThreads = 256;

inline int blocks_for(int64_t n) { return (int)((n + kThreads - 1) / kThreads); }

// flags[0] = out-of-range pair seen, flags[1] = number of dropped (self) pairs
__global__ void make_keys(const int64_t* __restrict__ dst, const int64_t* __restrict__ src,
                          int64_t n, int64_t n_rows, int64_t n_cols, int drop_self,
                          uint64_t* __restrict__ keys, int* __restrict__ flags) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t d = dst[i], s = src[i];
  uint64_t key;
  if (d < 0 || d >= n_rows || s < 0 || s >= n_cols) {
    flags[0] = 1;
    key = (uint64_t)n_rows << 32;
  } else if (drop_self && d == s) {
    atomicAdd(&flags[1], 1);
    key = (uint64_t)n_rows << 32;  // sorts behind every kept pair
  } else {
    key = ((uint64_t)d << 32) | (uint64_t)s;
  }
  keys[i] = key;
}

__global__ void keys_to_idx(const uint64_t* __restrict__ keys, int64_t nnz,
                            int32_t* __restrict__ idx) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nnz) idx[i] = (int32_t)(keys[i] & 0xffffffffull);
}

// rowptr[r] = first position whose key >= (r << 32)
__global__ void keys_to_rowptr(const uint64_t* __restrict__ keys, int64_t n, int64_t n_rows,
                               int32_t* __restrict__ rowptr, int* __restrict__ max_deg) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n_rows) return;
  auto lower = [&](uint64_t target) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if (keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  int64_t p = lower((uint64_t)r << 32);
  rowptr[r] = (int32_t)p;
  if (r < n_rows) {
    int64_t q = lower((uint64_t)(r + 1) << 32);
    atomicMax(max_deg, (int)(q - p));
  }
}

__global__ void degree_scale_kernel(const int32_t* __restrict__ rowptr, int64_t n, int inv_sqrt,
                                    float* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float d = (float)(rowptr[i + 1] - rowptr[i]);
  out[i] = d > 0.f ? (inv_sqrt ? 1.0f / sqrtf(d) : 1.0f / d) : 0.f;
}

__global__ void pack_scale_kernel(const int32_t* __restrict__ ids, int64_t n, const float* __restrict__ scale,
                                  int2* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int j = ids[i];
  out[i] = make_int2(j, __float_as_int(scale[j]));
}

__global__ void keys_equal_kernel(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b,
                                  int64_t n, int* __restrict__ differ) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && a[i] != b[i]) *differ = 1;
}

struct DeviceBuf {
  void* p = nullptr;
  ~DeviceBuf() { if (p) (void)hipFree(p); }
};

// One workgroup per row tile: sort the tile's source ids in LDS (bitonic, padded with INT_MAX),
// keep the distinct ones.  FILL = false: counts[t] = number of distinct ids, or -1 when the tile has
// more than kTileEdges edges.  FILL = true: write them to uniq[uptr[t]..] and each edge's slot.
template <bool FILL, int MAXE>
__global__ __launch_bounds__(64) void tile_pass(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ idx,
                                                 int64_t n_rows, int tile_rows, int32_t* __restrict__ counts,
                                                 const int32_t* __restrict__ uptr, int32_t* __restrict__ uniq,
                                                 uint8_t* __restrict__ eloc) {
  __shared__ int32_t s_key[MAXE];
  __shared__ int32_t s_cnt;
  const int64_t r0 = (int64_t)blockIdx.x * tile_rows;
  int64_t r1 = r0 + tile_rows;
  if (r1 > n_rows) r1 = n_rows;
  const int e0 = rowptr[r0], ne = rowptr[r1] - e0;
  if (ne > MAXE) {
    if (!FILL && threadIdx.x == 0) counts[blockIdx.x] = -1;
    return;
  }
  if (FILL && uptr[blockIdx.x + 1] == uptr[blockIdx.x]) return;   // no list for this tile (empty, or clipped: see build_tile_set)
  if (!FILL) {   // a source listed twice in one row (multigraph) cannot be expressed by one slot + one mask bit
    __shared__ int s_dup;
    if (threadIdx.x == 0) s_dup = 0;
    __syncthreads();
    for (int64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x)
      for (int e = rowptr[r] + 1; e < rowptr[r + 1]; ++e)
        if (idx[e] == idx[e - 1]) s_dup = 1;
    __syncthreads();
    if (s_dup) {
      if (threadIdx.x == 0) counts[blockIdx.x] = -1;
      return;
    }
  }
  int n2 = 1;
  while (n2 < ne) n2 <<= 1;
  for (int i = threadIdx.x; i < n2; i += blockDim.x) s_key[i] = i < ne ? idx[e0 + i] : INT32_MAX;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        const int p = i ^ j;
        if (p > i) {
          const int a = s_key[i], b = s_key[p];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { s_key[i] = b; s_key[p] = a; }
        }
      }
      __syncthreads();
    }
  }
  // position of each distinct value = number of "first occurrences" before it
  if (!FILL) {
    int local = 0;
    for (int i = threadIdx.x; i < ne; i += blockDim.x) local += (i == 0 || s_key[i] != s_key[i - 1]) ? 1 : 0;
    atomicAdd(&s_cnt, local);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_cnt;
  } else {
    // serial compaction by one wavefront-sized scan is plenty for <= 512 keys
    __shared__ int32_t s_rank[MAXE];
    if (threadIdx.x == 0) {
      int r = -1;
      for (int i = 0; i < ne; ++i) {
        if (i == 0 || s_key[i] != s_key[i - 1]) ++r;
        s_rank[i] = r;
      }
    }
    __syncthreads();
    const int u0 = uptr[blockIdx.x];
    for (int i = threadIdx.x; i < ne; i += blockDim.x)
      if (i == 0 || s_key[i] != s_key[i - 1]) uniq[u0 + s_rank[i]] = s_key[i];
    for (int i = threadIdx.x; i < ne; i += blockDim.x) {   // slot of edge i = rank of its id
      const int v = idx[e0 + i];
      int lo = 0, hi = ne - 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (s_key[mid] < v) lo = mid + 1; else hi = mid;
      }
      eloc[e0 + i] = (uint8_t)s_rank[lo];
    }
  }
}


// ---- graph-only locality order ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t vhash(uint32_t v) { return (v * 2654435761u) >> 7; }

// far[0] += number of edges whose ends are more than `span` ids apart
__global__ void count_far_edges(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ idx, int64_t n_rows,
                                int span, unsigned long long* __restrict__ far) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int local = 0;
  if (r < n_rows)
    for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) {
      const int64_t d = (int64_t)idx[e] - r;
      local += (d > span || d < -span) ? 1 : 0;
    }
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(far, (unsigned long long)local);
}

__global__ void seed_labels(int64_t n, uint32_t modulus, uint32_t residue, int32_t* __restrict__ lab) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) lab[v] = (vhash((uint32_t)v) % modulus == residue) ? (int32_t)v : -1;
}

// one round of multi-source BFS (graph Voronoi cells): an unlabelled vertex takes the smallest label among its
// labelled neighbours of the PREVIOUS round (double-buffered, so the result does not depend on scheduling)
__global__ void voronoi_round(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ idx, int64_t n,
                              const int32_t* __restrict__ in, int32_t* __restrict__ out, int* __restrict__ changed) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  int32_t l = in[v];
  if (l < 0) {
    int32_t best = INT32_MAX;
    for (int e = rowptr[v]; e < rowptr[v + 1]; ++e) {
      const int32_t m = in[idx[e]];
      if (m >= 0 && m < best) best = m;
    }
    if (best != INT32_MAX) {
      l = best;
      *changed = 1;
    }
  }
  out[v] = l;
}

__global__ void label_leftovers(int64_t n, int32_t* __restrict__ lab) {   // components without a seed: own cells
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n && lab[v] < 0) lab[v] = (int32_t)v;
}

__global__ void fine_keys(int64_t n, const int32_t* __restrict__ fine, uint64_t* __restrict__ keys) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) keys[v] = ((uint64_t)(uint32_t)fine[v] << 32) | (uint64_t)v;
}

__global__ void coarse_pairs(int64_t n, const uint64_t* __restrict__ sorted, const int32_t* __restrict__ coarse,
                             uint32_t* __restrict__ keys, int32_t* __restrict__ vals) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t v = (int32_t)(sorted[i] & 0xffffffffull);
  keys[i] = (uint32_t)coarse[v];
  vals[i] = v;
}

__global__ void label_keys(int64_t n, const int32_t* __restrict__ seq, const int32_t* __restrict__ lab,
                           uint32_t* __restrict__ keys) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[i] = (uint32_t)lab[seq[i]];
}

__global__ void permuted_degrees(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ order, int64_t n,
                                 int32_t* __restrict__ deg) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) deg[p] = rowptr[order[p] + 1] - rowptr[order[p]];
  else if (p == n) deg[p] = 0;
}

__global__ void copy_rows(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ idx,
                          const int32_t* __restrict__ order, const int32_t* __restrict__ new_rowptr, int64_t n,
                          int32_t* __restrict__ new_idx) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int s = rowptr[order[p]], d = new_rowptr[p], len = new_rowptr[p + 1] - d;
  for (int k = 0; k < len; ++k) new_idx[d + k] = idx[s + k];
}

__global__ void gather_floats_kernel(const float* __restrict__ src, const int32_t* __restrict__ order, int64_t n,
                                     float* __restrict__ out) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) out[p] = src[order[p]];
}

int g_reorder_mode = 0;   // SG_TUNE_GRAPH_REORDER: 0 = decide from the edge spans, 1 = never, 2 = always

}  // namespace

int graph_reorder_mode() { return g_reorder_mode; }
int set_graph_reorder_mode(int v) {
  g_reorder_mode = v;
  return SG_OK;
}

// Graph-only locality order (the operator tier sees no vertex positions: `from torch_geometric.nn import ChebConv`,
// util/networks.py:4).  A raw scan numbers its vertices arbitrarily; the aggregation then re-fetches every source row
// ~deg times from HBM (21-23 % of the roofline).  Cure without touching the caller's X / Y: process
// the ROWS in an order in which consecutive rows are graph neighbours.  Three multi-source BFS partitions (graph Voronoi
// cells around pseudo-random seeds, ~16, ~256 and ~4096 vertices per cell) give every vertex a (large, middle, small
// cell) triple; sorting by it yields compact patches at every scale -- the working set of the rows in flight on one
// XCD then fits its L2, like a Morton order from positions (distinct source rows per 8 K-row window: 1.13 x the rows,
// Morton 1.08 x, random 5.3 x), and the rows of one wavefront's chunk are mostly neighbours.  Deterministic:
// double-buffered rounds, smallest label wins, stable sorts.
int locality_order(const Csr& c, int mode, hipStream_t stream, int32_t** order_out) {
  *order_out = nullptr;
  const int64_t n = c.n_rows;
  if (mode == 1 || n == 0 || c.nnz == 0 || c.n_rows != c.n_cols) return SG_OK;
  DeviceBuf flag;
  SG_HIP_TRY(hipMalloc(&flag.p, sizeof(unsigned long long)));
  if (mode == 0) {
    if (n < 65536) return SG_OK;      // a graph this small lives in L2 whatever its numbering
    SG_HIP_TRY(hipMemsetAsync(flag.p, 0, sizeof(unsigned long long), stream));
    count_far_edges<<<blocks_for(n), kThreads, 0, stream>>>(c.rowptr, c.idx, n, 4096, (unsigned long long*)flag.p);
    unsigned long long far = 0;
    SG_HIP_TRY(hipMemcpyAsync(&far, flag.p, sizeof(far), hipMemcpyDeviceToHost, stream));
    SG_HIP_TRY(hipStreamSynchronize(stream));
    if ((double)far < 0.25 * (double)c.nnz) return SG_OK;   // grid / Morton / already clustered numbering
  }
  DeviceBuf lab, keys, keys2, k32a, k32b, vals, vals2, temp;
  int32_t* order = nullptr;
  constexpr int kLevels = 3;                                   // cells of ~16, ~256 and ~4096 vertices
  SG_HIP_TRY(hipMalloc(&lab.p, 2 * kLevels * n * sizeof(int32_t)));
  int32_t* L = (int32_t*)lab.p;
  int32_t* final_lab[kLevels] = {nullptr, nullptr, nullptr};
  const uint32_t modulus[kLevels] = {16u, 256u, 4096u}, residue[kLevels] = {5u, 0u, 17u};
  int* d_changed = (int*)flag.p;
  for (int level = 0; level < kLevels; ++level) {
    int32_t* a = L + (2 * level) * n;
    int32_t* b = L + (2 * level + 1) * n;
    seed_labels<<<blocks_for(n), kThreads, 0, stream>>>(n, modulus[level], residue[level], a);
    const int per_check = level == 0 ? 4 : 8;                  // rounds between host checks
    for (int round = 0; round < 1 << 16;) {
      SG_HIP_TRY(hipMemsetAsync(d_changed, 0, sizeof(int), stream));
      for (int k = 0; k < per_check; ++k, ++round) {
        voronoi_round<<<blocks_for(n), kThreads, 0, stream>>>(c.rowptr, c.idx, n, a, b, d_changed);
        int32_t* t = a; a = b; b = t;
      }
      int changed = 0;
      SG_HIP_TRY(hipMemcpyAsync(&changed, d_changed, sizeof(int), hipMemcpyDeviceToHost, stream));
      SG_HIP_TRY(hipStreamSynchronize(stream));
      if (!changed) break;
    }
    label_leftovers<<<blocks_for(n), kThreads, 0, stream>>>(n, a);
    final_lab[level] = a;
  }
  SG_HIP_TRY(hipGetLastError());
  SG_HIP_TRY(hipMalloc(&keys.p, n * sizeof(uint64_t)));
  SG_HIP_TRY(hipMalloc(&keys2.p, n * sizeof(uint64_t)));
  SG_HIP_TRY(hipMalloc(&k32a.p, n * sizeof(uint32_t)));
  SG_HIP_TRY(hipMalloc(&k32b.p, n * sizeof(uint32_t)));
  SG_HIP_TRY(hipMalloc(&vals.p, n * sizeof(int32_t)));
  SG_HIP_TRY(hipMalloc(&vals2.p, n * sizeof(int32_t)));
  SG_HIP_TRY(hipMalloc((void**)&order, n * sizeof(int32_t)));
  // sort by (smallest cell, id), then stably by the middle cell, then stably by the largest cell
  fine_keys<<<blocks_for(n), kThreads, 0, stream>>>(n, final_lab[0], (uint64_t*)keys.p);
  size_t tb = 0, tb2 = 0;
  hipError_t e = hipcub::DeviceRadixSort::SortKeys(nullptr, tb, (const uint64_t*)keys.p, (uint64_t*)keys2.p, (int)n, 0, 64, stream);
  if (e == hipSuccess)
    e = hipcub::DeviceRadixSort::SortPairs(nullptr, tb2, (const uint32_t*)k32a.p, (uint32_t*)k32b.p, (const int32_t*)vals.p,
                                           order, (int)n, 0, 32, stream);
  if (tb2 > tb) tb = tb2;
  if (e == hipSuccess) e = hipMalloc(&temp.p, tb ? tb : 16);
  if (e == hipSuccess)
    e = hipcub::DeviceRadixSort::SortKeys(temp.p, tb, (const uint64_t*)keys.p, (uint64_t*)keys2.p, (int)n, 0, 64, stream);
  if (e == hipSuccess) {
    coarse_pairs<<<blocks_for(n), kThreads, 0, stream>>>(n, (const uint64_t*)keys2.p, final_lab[1], (uint32_t*)k32a.p,
                                                         (int32_t*)vals.p);
    e = hipcub::DeviceRadixSort::SortPairs(temp.p, tb, (const uint32_t*)k32a.p, (uint32_t*)k32b.p, (const int32_t*)vals.p,
                                           (int32_t*)vals2.p, (int)n, 0, 32, stream);   // stable: keeps (small cell, id)
  }
  if (e == hipSuccess) {
    label_keys<<<blocks_for(n), kThreads, 0, stream>>>(n, (const int32_t*)vals2.p, final_lab[2], (uint32_t*)k32a.p);
    e = hipcub::DeviceRadixSort::SortPairs(temp.p, tb, (const uint32_t*)k32a.p, (uint32_t*)k32b.p, (const int32_t*)vals2.p,
                                           order, (int)n, 0, 32, stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) {
    (void)hipFree(order);
    set_error("locality_order: %s", hipGetErrorString(e));
    return SG_ERR_HIP;
  }
  *order_out = order;
  return SG_OK;
}

int permute_rows(const Csr& base, const int32_t* order, hipStream_t stream, Csr* out) {
  *out = Csr();
  const int64_t n = base.n_rows;
  out->n_rows = n;
  out->n_cols = base.n_cols;
  out->nnz = base.nnz;
  out->max_degree = base.max_degree;
  DeviceBuf deg, temp;
  SG_HIP_TRY(hipMalloc(&deg.p, (n + 1) * sizeof(int32_t)));
  SG_HIP_TRY(hipMalloc((void**)&out->rowptr, (n + 1) * sizeof(int32_t)));
  SG_HIP_TRY(hipMalloc((void**)&out->idx, (base.nnz > 0 ? base.nnz : 1) * sizeof(int32_t)));
  permuted_degrees<<<blocks_for(n + 1), kThreads, 0, stream>>>(base.rowptr, order, n, (int32_t*)deg.p);
  size_t tb = 0;
  SG_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, (int32_t*)deg.p, out->rowptr, (int)n + 1, stream));
  SG_HIP_TRY(hipMalloc(&temp.p, tb ? tb : 16));
  SG_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(temp.p, tb, (int32_t*)deg.p, out->rowptr, (int)n + 1, stream));
  copy_rows<<<blocks_for(n), kThreads, 0, stream>>>(base.rowptr, base.idx, order, out->rowptr, n, out->idx);
  SG_HIP_TRY(hipGetLastError());
  SG_HIP_TRY(hipStreamSynchronize(stream));
  return SG_OK;
}

int gather_floats(const float* src, const int32_t* order, int64_t n, float* out, hipStream_t stream) {
  if (n == 0) return SG_OK;
  gather_floats_kernel<<<blocks_for(n), kThreads, 0, stream>>>(src, order, n, out);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

namespace {
}  // namespace

int build_csr(const int64_t* dst, const int64_t* src, int64_t n, int64_t n_rows, int64_t n_cols,
              bool drop_self, hipStream_t stream, Csr* out, uint64_t* keys_out) {
  SG_REQUIRE(n >= 0 && n_rows >= 0 && n_cols >= 0, "build_csr: negative size");
  SG_REQUIRE(n < (int64_t)INT32_MAX && n_rows < (int64_t)INT32_MAX - 1 && n_cols < (int64_t)INT32_MAX,
             "build_csr: sizes must fit int32 (n=%lld rows=%lld cols=%lld)", (long long)n,
             (long long)n_rows, (long long)n_cols);
  SG_REQUIRE(n == 0 || (dst && src), "build_csr: null index pointer");
  *out = Csr();
  out->n_rows = n_rows;
  out->n_cols = n_cols;

  DeviceBuf keys_in, keys_sorted, flags, temp;
  SG_HIP_TRY(hipMalloc(&flags.p, 3 * sizeof(int)));
  SG_HIP_TRY(hipMemsetAsync(flags.p, 0, 3 * sizeof(int), stream));
  int* d_flags = (int*)flags.p;
  uint64_t* sorted = nullptr;
  if (n > 0) {
    SG_HIP_TRY(hipMalloc(&keys_in.p, n * sizeof(uint64_t)));
    if (keys_out) {
      sorted = keys_out;
    } else {
      SG_HIP_TRY(hipMalloc(&keys_sorted.p, n * sizeof(uint64_t)));
      sorted = (uint64_t*)keys_sorted.p;
    }
    make_keys<<<blocks_for(n), kThreads, 0, stream>>>(dst, src, n, n_rows, n_cols, drop_self ? 1 : 0,
                                                     (uint64_t*)keys_in.p, d_flags);
    int end_bit = 33;
    while (end_bit < 64 && ((uint64_t)n_rows >> (end_bit - 32)) != 0) ++end_bit;
    size_t temp_bytes = 0;
    SG_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, (const uint64_t*)keys_in.p,
                                                 sorted, (int)n, 0, end_bit, stream));
    SG_HIP_TRY(hipMalloc(&temp.p, temp_bytes ? temp_bytes : 16));
    SG_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(temp.p, temp_bytes, (const uint64_t*)keys_in.p,
                                                 sorted, (int)n, 0, end_bit, stream));
  }
  SG_HIP_TRY(hipMalloc((void**)&out->rowptr, (n_rows + 1) * sizeof(int32_t)));
  if (n > 0) {
    keys_to_rowptr<<<blocks_for(n_rows + 1), kThreads, 0, stream>>>(sorted, n, n_rows, out->rowptr,
                                                                   d_flags + 2);
  } else {
    SG_HIP_TRY(hipMemsetAsync(out->rowptr, 0, (n_rows + 1) * sizeof(int32_t), stream));
  }
  int h_flags[3] = {0, 0, 0};
  SG_HIP_TRY(hipMemcpyAsync(h_flags, d_flags, sizeof(h_flags), hipMemcpyDeviceToHost, stream));
  SG_HIP_TRY(hipStreamSynchronize(stream));
  if (h_flags[0]) {
    out->release();
    set_error("index out of range in (dst, src) pairs (rows=%lld cols=%lld)", (long long)n_rows,
              (long long)n_cols);
    return SG_ERR_INVALID;
  }
  out->nnz = n - h_flags[1];
  out->max_degree = h_flags[2];
  SG_HIP_TRY(hipMalloc((void**)&out->idx, (out->nnz > 0 ? out->nnz : 1) * sizeof(int32_t)));
  if (out->nnz > 0) {
    keys_to_idx<<<blocks_for(out->nnz), kThreads, 0, stream>>>(sorted, out->nnz, out->idx);
    SG_HIP_TRY(hipGetLastError());
    SG_HIP_TRY(hipStreamSynchronize(stream));  // `sorted` may be freed on return
  }
  return SG_OK;
}

// Distinct-source lists of row tiles: tile t = rows [t*tile_rows, (t+1)*tile_rows).  Leaves *uptr / *uniq / *eloc null
// when some tile has more than max_slots distinct sources, more than MAXE edges, or a duplicate edge inside a row.
__global__ void clip_counts(int32_t* __restrict__ counts, int64_t n, int max_slots) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (counts[i] < 0 || counts[i] > max_slots)) counts[i] = 0;
}

// clip = false: the set exists only if EVERY tile fits (max_slots distinct sources, MAXE edges, no duplicate edge);
// clip = true: tiles that do not fit get an EMPTY source list (their rows are then gathered from global memory by the
// kernel's fallback path) -- one awkward tile (a Morton seam, a hub vertex) does not cost the whole graph its tiles.
template <int MAXE>
static int build_tile_set(const Csr* c, int tile_rows, int max_slots, bool clip, hipStream_t stream, int32_t** uptr_out,
                          int32_t** uniq_out, uint8_t** eloc_out) {
  *uptr_out = nullptr;
  *uniq_out = nullptr;
  *eloc_out = nullptr;
  if (c->n_rows == 0 || c->nnz == 0) return SG_OK;
  const int64_t nt = (c->n_rows + tile_rows - 1) / tile_rows;
  DeviceBuf counts, temp;
  SG_HIP_TRY(hipMalloc(&counts.p, (nt + 1) * sizeof(int32_t)));
  SG_HIP_TRY(hipMemsetAsync(counts.p, 0, (nt + 1) * sizeof(int32_t), stream));
  tile_pass<false, MAXE><<<(int)nt, 64, 0, stream>>>(c->rowptr, c->idx, c->n_rows, tile_rows, (int32_t*)counts.p, nullptr,
                                                     nullptr, nullptr);
  SG_HIP_TRY(hipGetLastError());
  if (clip) clip_counts<<<blocks_for(nt), kThreads, 0, stream>>>((int32_t*)counts.p, nt, max_slots);
  // tileable iff every tile has 0 <= distinct <= max_slots: min and max over the counts
  int32_t* d_minmax = nullptr;
  DeviceBuf mm;
  SG_HIP_TRY(hipMalloc(&mm.p, 2 * sizeof(int32_t)));
  d_minmax = (int32_t*)mm.p;
  size_t tb = 0;
  SG_HIP_TRY(hipcub::DeviceReduce::Min(nullptr, tb, (int32_t*)counts.p, d_minmax, (int)nt, stream));
  size_t tb2 = 0;
  SG_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, (int32_t*)counts.p, (int32_t*)counts.p, (int)nt + 1, stream));
  if (tb2 > tb) tb = tb2;
  SG_HIP_TRY(hipMalloc(&temp.p, tb ? tb : 16));
  SG_HIP_TRY(hipcub::DeviceReduce::Min(temp.p, tb, (int32_t*)counts.p, d_minmax, (int)nt, stream));
  SG_HIP_TRY(hipcub::DeviceReduce::Max(temp.p, tb, (int32_t*)counts.p, d_minmax + 1, (int)nt, stream));
  int32_t h_mm[2] = {0, 0};
  SG_HIP_TRY(hipMemcpyAsync(h_mm, d_minmax, sizeof(h_mm), hipMemcpyDeviceToHost, stream));
  SG_HIP_TRY(hipStreamSynchronize(stream));
  if (h_mm[0] < 0 || h_mm[1] > max_slots) return SG_OK;   // not tileable: the generic kernel serves this graph
  int32_t* uptr = nullptr;
  int32_t* uniq = nullptr;
  uint8_t* eloc = nullptr;
  SG_HIP_TRY(hipMalloc((void**)&uptr, (nt + 1) * sizeof(int32_t)));
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(temp.p, tb, (int32_t*)counts.p, uptr, (int)nt + 1, stream);
  int32_t total = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&total, uptr + nt, sizeof(int32_t), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e == hipSuccess) e = hipMalloc((void**)&uniq, (total > 0 ? total : 1) * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&eloc, c->nnz);
  if (e == hipSuccess) {
    tile_pass<true, MAXE><<<(int)nt, 64, 0, stream>>>(c->rowptr, c->idx, c->n_rows, tile_rows, nullptr, uptr, uniq, eloc);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) {
    if (uptr) (void)hipFree(uptr);
    if (uniq) (void)hipFree(uniq);
    if (eloc) (void)hipFree(eloc);
    set_error("build_tile_set: %s", hipGetErrorString(e));
    return SG_ERR_HIP;
  }
  *uptr_out = uptr;
  *uniq_out = uniq;
  *eloc_out = eloc;
  return SG_OK;
}

int build_tiles(Csr* c, hipStream_t stream) {
  int rc = build_tile_set<kTileEdges>(c, kTileRows, kTileSlots, false, stream, &c->tile_uptr, &c->tile_uniq, &c->tile_eloc);
  if (rc != SG_OK) return rc;
  if (c->tile_uptr) c->tile_rows = kTileRows;
  // LDS tiles of the experimental kernel that stages source rows in LDS (spmm.hip::spmm_lds), only on request
  if (lds_tiles_enabled())
    rc = build_tile_set<kLdsEdges>(c, kLdsRows, kLdsSlots, true, stream, &c->lt_uptr, &c->lt_uniq, &c->lt_eloc);
  return rc;
}

int pack_source_scale(Csr* c, const float* scale, hipStream_t stream) {
  if (c->nnz == 0 || scale == nullptr) return SG_OK;
  SG_HIP_TRY(hipMalloc((void**)&c->idx_w, c->nnz * sizeof(int2)));
  pack_scale_kernel<<<blocks_for(c->nnz), kThreads, 0, stream>>>(c->idx, c->nnz, scale, c->idx_w);
  SG_HIP_TRY(hipGetLastError());
  if (c->tile_uptr) {
    const int64_t nt = (c->n_rows + c->tile_rows - 1) / c->tile_rows;
    int32_t total = 0;
    SG_HIP_TRY(hipMemcpyAsync(&total, c->tile_uptr + nt, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    SG_HIP_TRY(hipStreamSynchronize(stream));
    if (total > 0) {
      SG_HIP_TRY(hipMalloc((void**)&c->tile_uniq_w, (size_t)total * sizeof(int2)));
      pack_scale_kernel<<<blocks_for(total), kThreads, 0, stream>>>(c->tile_uniq, total, scale, c->tile_uniq_w);
      SG_HIP_TRY(hipGetLastError());
    }
  }
  if (c->lt_uptr) {
    const int64_t nt = (c->n_rows + kLdsRows - 1) / kLdsRows;
    int32_t total = 0;
    SG_HIP_TRY(hipMemcpyAsync(&total, c->lt_uptr + nt, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    SG_HIP_TRY(hipStreamSynchronize(stream));
    if (total > 0) {
      SG_HIP_TRY(hipMalloc((void**)&c->lt_uniq_w, (size_t)total * sizeof(int2)));
      pack_scale_kernel<<<blocks_for(total), kThreads, 0, stream>>>(c->lt_uniq, total, scale, c->lt_uniq_w);
      SG_HIP_TRY(hipGetLastError());
    }
  }
  c->packed_scale = scale;
  return SG_OK;
}

// ---------------------------------------------------------------------------------------------
// Tile records of spmm.hip::spmm_ring.  One 64-thread block per 16 consecutive rows: if their distinct sources fit the LDS
// budget the block is ONE tile; if not it is cut into two 8-row tiles (a compact patch of a triangle mesh in a locality
// order: ~43 distinct sources for 16 rows, ~28 for 8); what still does not fit gets a record with nu = 0 and is gathered
// from global memory by the kernel.  Records land in a [blocks][2] scratch array and are compacted after a scan.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int kRecMaxEdges = 256;

// all 64 threads of the block; returns true (block-uniform) when the tile fits, and then has written its record
__device__ bool ring_record(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ idx, int r0, int nrows,
                            const float* __restrict__ scale_src, const float* __restrict__ scale_dst,
                            const int32_t* __restrict__ row_id, bool force, uint8_t* __restrict__ rec,
                            int32_t* s_key, int32_t* s_rank, int32_t* s_flag) {
  const int tid = threadIdx.x;
  const int e0 = rowptr[r0], ne = rowptr[r0 + nrows] - e0;
  __syncthreads();                                   // s_flag / s_key of the previous call are free
  if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; }
  __syncthreads();
  if (tid < nrows) {
    const int a = rowptr[r0 + tid], b = rowptr[r0 + tid + 1];
    bool bad = b - a > 16;
    for (int e = a + 1; e < b; ++e) bad |= idx[e] == idx[e - 1];
    if (bad) s_flag[0] = 1;
  }
  __syncthreads();
  bool fits = ne <= kRecMaxEdges && ne > 0 && s_flag[0] == 0;
  int nu = 0;
  if (fits) {
    int n2 = 1;
    while (n2 < ne) n2 <<= 1;
    for (int i = tid; i < n2; i += 64) s_key[i] = i < ne ? idx[e0 + i] : INT32_MAX;
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < n2; i += 64) {
          const int p = i ^ j;
          if (p > i) {
            const int a = s_key[i], b = s_key[p];
            const bool up = (i & k) == 0;
            if ((a > b) == up) { s_key[i] = b; s_key[p] = a; }
          }
        }
        __syncthreads();
      }
    if (tid == 0) {
      int r = -1;
      for (int i = 0; i < ne; ++i) {
        if (i == 0 || s_key[i] != s_key[i - 1]) ++r;
        s_rank[i] = r;
      }
      s_flag[1] = r + 1;
    }
    __syncthreads();
    nu = s_flag[1];
    fits = nu <= kRingSlots;
  }
  if (!fits && !force) return false;
  if (!fits) nu = 0;
  // ---- the record ----
  for (int i = tid; i < kRecBytes / 4; i += 64) ((int32_t*)rec)[i] = 0;
  __syncthreads();
  if (nu > 0) {
    for (int i = tid; i < ne; i += 64)
      if (i == 0 || s_key[i] != s_key[i - 1]) {
        const int u = s_rank[i];
        ((int32_t*)(rec + kRecSrc))[u] = s_key[i];
        ((float*)(rec + kRecW))[u] = scale_src ? scale_src[s_key[i]] : 1.0f;
      }
    __syncthreads();
    for (int i = nu + tid; i < kRingSlots; i += 64) {      // padding: the last source (weight 0: never referenced)
      ((int32_t*)(rec + kRecSrc))[i] = ((const int32_t*)(rec + kRecSrc))[nu - 1];
    }
    for (int t = tid; t < kLdsRows * 16; t += 64) {
      const int r = t >> 4, u = t & 15;
      uint8_t v = 64;
      if (r < nrows) {
        const int a = rowptr[r0 + r], d = rowptr[r0 + r + 1] - a;
        if (d > 0) {
          const int e = a + (u < d ? u : d - 1);
          const int key = idx[e];
          int lo = 0, hi = ne - 1;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_key[mid] < key) lo = mid + 1; else hi = mid;
          }
          v = (uint8_t)(s_rank[lo] | (u < d ? 0 : 64));
        }
      }
      rec[kRecSlot + t] = v;
    }
  }
  if (tid < kLdsRows) {
    const int r = tid < nrows ? tid : nrows - 1;
    rec[kRecDeg + tid] = (tid < nrows && nu > 0) ? (uint8_t)(rowptr[r0 + tid + 1] - rowptr[r0 + tid]) : (uint8_t)0;
    ((float*)(rec + kRecSd))[tid] = tid < nrows ? (scale_dst ? scale_dst[r0 + tid] : 1.0f) : 0.0f;
    ((int32_t*)(rec + kRecRow))[tid] = row_id ? row_id[r0 + r] : r0 + r;
  }
  if (tid == 0) {
    *(int32_t*)(rec + kRecNu) = nu;
    *(int32_t*)(rec + kRecE0) = e0;
    *(int32_t*)(rec + kRecR0) = r0;
    *(int32_t*)(rec + kRecNrows) = nrows;
  }
  return true;
}

__global__ __launch_bounds__(64) void ring_records_pass(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ idx,
                                                         int64_t n_rows, const float* __restrict__ scale_src,
                                                         const float* __restrict__ scale_dst,
                                                         const int32_t* __restrict__ row_id, uint8_t* __restrict__ scratch,
                                                         int32_t* __restrict__ counts) {
  __shared__ int32_t s_key[kRecMaxEdges];
  __shared__ int32_t s_rank[kRecMaxEdges];
  __shared__ int32_t s_flag[2];
  const int64_t blk = blockIdx.x;
  const int r0 = (int)(blk * kLdsRows);
  int nrows = (int)(n_rows - r0);
  nrows = nrows > kLdsRows ? kLdsRows : nrows;
  uint8_t* out = scratch + blk * 2 * kRecBytes;
  int cnt = 1;
  if (!ring_record(rowptr, idx, r0, nrows, scale_src, scale_dst, row_id, nrows <= 8, out, s_key, s_rank, s_flag)) {
    ring_record(rowptr, idx, r0, 8, scale_src, scale_dst, row_id, true, out, s_key, s_rank, s_flag);
    ring_record(rowptr, idx, r0 + 8, nrows - 8, scale_src, scale_dst, row_id, true, out + kRecBytes, s_key, s_rank, s_flag);
    cnt = 2;
  }
  if (threadIdx.x == 0) counts[blk] = cnt;
}

__global__ __launch_bounds__(64) void ring_records_compact(const uint8_t* __restrict__ scratch, const int32_t* __restrict__ offs,
                                                            uint8_t* __restrict__ rec) {
  const int64_t blk = blockIdx.x;
  const int o = offs[blk], n = offs[blk + 1] - o;
  for (int i = threadIdx.x; i < n * (kRecBytes / 16); i += 64)
    ((int4*)(rec + (int64_t)o * kRecBytes))[i] = ((const int4*)(scratch + blk * 2 * kRecBytes))[i];
}
}  // namespace

int build_ring_records(Csr* c, const float* scale_src, const float* scale_dst, const int32_t* row_id, hipStream_t stream) {
  if (c->lt_rec) { (void)hipFree(c->lt_rec); c->lt_rec = nullptr; c->lt_nrec = 0; }
  if (c->n_rows == 0 || c->nnz == 0) return SG_OK;
  const int64_t nb = (c->n_rows + kLdsRows - 1) / kLdsRows;
  DeviceBuf scratch, counts, temp;
  SG_HIP_TRY(hipMalloc(&scratch.p, (size_t)nb * 2 * kRecBytes));
  SG_HIP_TRY(hipMalloc(&counts.p, (nb + 1) * sizeof(int32_t)));
  SG_HIP_TRY(hipMemsetAsync(counts.p, 0, (nb + 1) * sizeof(int32_t), stream));
  ring_records_pass<<<(int)nb, 64, 0, stream>>>(c->rowptr, c->idx, c->n_rows, scale_src, scale_dst, row_id,
                                                (uint8_t*)scratch.p, (int32_t*)counts.p);
  SG_HIP_TRY(hipGetLastError());
  size_t tb = 0;
  SG_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, (int32_t*)counts.p, (int32_t*)counts.p, (int)nb + 1, stream));
  SG_HIP_TRY(hipMalloc(&temp.p, tb ? tb : 16));
  SG_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(temp.p, tb, (int32_t*)counts.p, (int32_t*)counts.p, (int)nb + 1, stream));
  int32_t total = 0;
  SG_HIP_TRY(hipMemcpyAsync(&total, (int32_t*)counts.p + nb, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  SG_HIP_TRY(hipStreamSynchronize(stream));
  SG_HIP_TRY(hipMalloc((void**)&c->lt_rec, (size_t)total * kRecBytes));
  ring_records_compact<<<(int)nb, 64, 0, stream>>>((const uint8_t*)scratch.p, (const int32_t*)counts.p, c->lt_rec);
  SG_HIP_TRY(hipGetLastError());
  SG_HIP_TRY(hipStreamSynchronize(stream));
  c->lt_nrec = total;
  c->rec_scale_dst = scale_dst;
  c->rec_row_id = row_id;
  return SG_OK;
}

int degree_scale(const Csr& c, bool inv_sqrt, float* out, hipStream_t stream) {
  if (c.n_rows == 0) return SG_OK;
  degree_scale_kernel<<<blocks_for(c.n_rows), kThreads, 0, stream>>>(c.rowptr, c.n_rows,
                                                                    inv_sqrt ? 1 : 0, out);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

int keys_equal(const uint64_t* a, const uint64_t* b, int64_t n, hipStream_t stream, int* equal) {
  DeviceBuf flag;
  SG_HIP_TRY(hipMalloc(&flag.p, sizeof(int)));
  SG_HIP_TRY(hipMemsetAsync(flag.p, 0, sizeof(int), stream));
  if (n > 0) keys_equal_kernel<<<blocks_for(n), kThreads, 0, stream>>>(a, b, n, (int*)flag.p);
  int differ = 0;
  SG_HIP_TRY(hipMemcpyAsync(&differ, flag.p, sizeof(int), hipMemcpyDeviceToHost, stream));
  SG_HIP_TRY(hipStreamSynchronize(stream));
  *equal = differ ? 0 : 1;
  return SG_OK;
}

}  // namespace sg
