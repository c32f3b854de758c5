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
template <bool FILL>
__global__ __launch_bounds__(64) void tile_pass(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ idx,
                                                 int64_t n_rows, int tile_rows, int32_t* __restrict__ counts,
                                                 const int32_t* __restrict__ uptr, int32_t* __restrict__ uniq,
                                                 uint8_t* __restrict__ eloc) {
  __shared__ int32_t s_key[kTileEdges];
  __shared__ int32_t s_cnt;
  const int64_t r0 = (int64_t)blockIdx.x * tile_rows;
  int64_t r1 = r0 + tile_rows;
  if (r1 > n_rows) r1 = n_rows;
  const int e0 = rowptr[r0], ne = rowptr[r1] - e0;
  if (ne > kTileEdges) {
    if (!FILL && threadIdx.x == 0) counts[blockIdx.x] = -1;
    return;
  }
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
    __shared__ int32_t s_rank[kTileEdges];
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

int build_tiles(Csr* c, hipStream_t stream) {
  if (c->n_rows == 0 || c->nnz == 0) return SG_OK;
  const int64_t nt = (c->n_rows + kTileRows - 1) / kTileRows;
  DeviceBuf counts, temp;
  SG_HIP_TRY(hipMalloc(&counts.p, (nt + 1) * sizeof(int32_t)));
  SG_HIP_TRY(hipMemsetAsync(counts.p, 0, (nt + 1) * sizeof(int32_t), stream));
  tile_pass<false><<<(int)nt, 64, 0, stream>>>(c->rowptr, c->idx, c->n_rows, kTileRows, (int32_t*)counts.p, nullptr,
                                               nullptr, nullptr);
  SG_HIP_TRY(hipGetLastError());
  // tileable iff every tile has 0 <= distinct <= kTileSlots: min and max over the counts
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
  if (h_mm[0] < 0 || h_mm[1] > kTileSlots) return SG_OK;   // not tileable: the generic kernel serves this graph
  SG_HIP_TRY(hipMalloc((void**)&c->tile_uptr, (nt + 1) * sizeof(int32_t)));
  SG_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(temp.p, tb, (int32_t*)counts.p, c->tile_uptr, (int)nt + 1, stream));
  int32_t total = 0;
  SG_HIP_TRY(hipMemcpyAsync(&total, c->tile_uptr + nt, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  SG_HIP_TRY(hipStreamSynchronize(stream));
  SG_HIP_TRY(hipMalloc((void**)&c->tile_uniq, (total > 0 ? total : 1) * sizeof(int32_t)));
  SG_HIP_TRY(hipMalloc((void**)&c->tile_eloc, c->nnz));
  tile_pass<true><<<(int)nt, 64, 0, stream>>>(c->rowptr, c->idx, c->n_rows, kTileRows, nullptr, c->tile_uptr,
                                              c->tile_uniq, c->tile_eloc);
  SG_HIP_TRY(hipGetLastError());
  SG_HIP_TRY(hipStreamSynchronize(stream));
  c->tile_rows = kTileRows;
  return SG_OK;
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
  c->packed_scale = scale;
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
