// Mesh connectivity from a triangle list, and ring dilation of vertex masks, on the device.
//
// Replaces the Python loops and dense V x V products the reference uses to prepare a mesh for
// the graph convolutions:
//   * util/mesh.py:60-100 (build_gemm -> self.edges): unique undirected edges in the order a
//     face-by-face scan first meets them -- the order edge_index inherits (util/mesh.py:229-230);
//   * util/mesh.py:214-227 (f2f): the (up to three) faces across each face's edges;
//   * util/datamaker.py:123-127: Mv1 = (AdjI @ Mv0) > 0, one ring of dilation for dm_size masks;
//   * util/datamaker.py:136,156-159 and util/meshnet.py:179,196: a face is kept iff none of its
//     three vertices is dropped ((f2v_mat @ (1 - vmask)) == 0).
// All of it is integer work bound by HBM traffic: half-edges are radix-sorted by (lo, hi) key
// (hipCUB), masks travel as one bit per mask packed into 64-bit words.
#include <hipcub/hipcub.hpp>

#include "sg_common.h"

namespace sg {
namespace {

constexpr int kThreads = 256;

inline int blocks_for(int64_t n) { return (int)((n + kThreads - 1) / kThreads); }

struct DeviceBuf {
  void* p = nullptr;
  ~DeviceBuf() { if (p) (void)hipFree(p); }
};

// Half-edge h = 3 f + i joins faces[f][i] and faces[f][(i+1)%3]  (util/mesh.py:72-74).
// flags[0]: vertex id out of range; flags[1]: degenerate face (repeated vertex).
__global__ void half_edge_keys(const int64_t* __restrict__ faces, int64_t n_half, int64_t V,
                               uint64_t* __restrict__ keys, uint32_t* __restrict__ half, int* __restrict__ flags) {
  const int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= n_half) return;
  const int64_t f = h / 3;
  const int i = (int)(h - 3 * f);
  const int64_t a = faces[3 * f + i], b = faces[3 * f + (i == 2 ? 0 : i + 1)];
  uint64_t key = ~0ull;
  if (a < 0 || a >= V || b < 0 || b >= V) {
    flags[0] = 1;
  } else {
    if (a == b) flags[1] = 1;
    const uint64_t lo = (uint64_t)(a < b ? a : b), hi = (uint64_t)(a < b ? b : a);
    key = (lo << 32) | hi;
  }
  keys[h] = key;
  half[h] = (uint32_t)h;
}

// After the stable sort, equal keys are adjacent and their half-edge ids ascend.  The head of a
// run is the edge's first meeting; a run of two names the two faces across the edge.
// flags[2]: an edge with more than two faces.
__global__ void scan_runs(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ half, int64_t n_half,
                          uint8_t* __restrict__ is_head, int64_t* __restrict__ f2f, int* __restrict__ flags) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_half) return;
  const uint64_t k = keys[p];
  const bool head = p == 0 || keys[p - 1] != k;
  is_head[p] = head ? 1 : 0;
  if (!f2f) return;
  const bool prev = !head;
  const bool next = p + 1 < n_half && keys[p + 1] == k;
  if (prev && next) flags[2] = 1;
  if (head && next && p + 2 < n_half && keys[p + 2] == k) flags[2] = 1;
  int64_t other = -1;
  if (head && next) other = half[p + 1] / 3;
  else if (prev && !next && (p < 2 || keys[p - 2] != k)) other = half[p - 1] / 3;
  f2f[half[p]] = other;    // slot = the face's own edge number, compacted below
}

// Rows of f2f list neighbours first and pad with -1 behind them (util/mesh.py:224).
__global__ void compact_f2f(int64_t* __restrict__ f2f, int64_t F) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  int64_t v[3] = {f2f[3 * f], f2f[3 * f + 1], f2f[3 * f + 2]}, o[3] = {-1, -1, -1};
  int n = 0;
  for (int i = 0; i < 3; ++i)
    if (v[i] >= 0) o[n++] = v[i];
  for (int i = 0; i < 3; ++i) f2f[3 * f + i] = o[i];
}

__global__ void edges_from_heads(const uint32_t* __restrict__ first_half, int64_t n_edges,
                                 const int64_t* __restrict__ faces, int64_t* __restrict__ edges) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  const int64_t h = first_half[e], f = h / 3;
  const int i = (int)(h - 3 * f);
  const int64_t a = faces[3 * f + i], b = faces[3 * f + (i == 2 ? 0 : i + 1)];
  edges[2 * e] = a < b ? a : b;       // tuple(sorted(edge)), util/mesh.py:76
  edges[2 * e + 1] = a < b ? b : a;
}

// out[v] = in[v] | OR_{j in N(v)} in[j]; one thread per (vertex, word).
__global__ void dilate_bits(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ idx, int64_t V, int W,
                            const uint64_t* __restrict__ in, uint64_t* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= V * W) return;
  const int64_t v = t / W;
  const int w = (int)(t - v * W);
  uint64_t acc = in[v * W + w];
  const int e1 = rowptr[v + 1];
  for (int e = rowptr[v]; e < e1; ++e) acc |= in[(int64_t)idx[e] * W + w];
  out[t] = acc;
}

__global__ void face_and_bits(const int64_t* __restrict__ faces, int64_t F, int64_t V, int W,
                              const uint64_t* __restrict__ vbits, uint64_t* __restrict__ fbits,
                              int* __restrict__ bad) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= F * W) return;
  const int64_t f = t / W;
  const int w = (int)(t - f * W);
  const int64_t a = faces[3 * f], b = faces[3 * f + 1], c = faces[3 * f + 2];
  if (a < 0 || a >= V || b < 0 || b >= V || c < 0 || c >= V) {
    *bad = 1;
    fbits[t] = 0;
    return;
  }
  fbits[t] = vbits[a * W + w] & vbits[b * W + w] & vbits[c * W + w];
}

}  // namespace

int mesh_edges(const int64_t* faces, int64_t F, int64_t V, int64_t* edges_out, int64_t* f2f_out,
               int64_t* n_edges_out, int* manifold_out, hipStream_t stream) {
  const int64_t n_half = 3 * F;
  SG_REQUIRE(V < ((int64_t)1 << 31) && n_half < ((int64_t)1 << 31), "sg_mesh_edges: sizes must fit int32");
  *n_edges_out = 0;
  if (manifold_out) *manifold_out = 1;
  if (F == 0) return SG_OK;
  DeviceBuf keys_a, keys_b, half_a, half_b, heads, first, first_sorted, count, flags, temp;
  SG_HIP_TRY(hipMalloc(&keys_a.p, n_half * sizeof(uint64_t)));
  SG_HIP_TRY(hipMalloc(&keys_b.p, n_half * sizeof(uint64_t)));
  SG_HIP_TRY(hipMalloc(&half_a.p, n_half * sizeof(uint32_t)));
  SG_HIP_TRY(hipMalloc(&half_b.p, n_half * sizeof(uint32_t)));
  SG_HIP_TRY(hipMalloc(&heads.p, n_half));
  SG_HIP_TRY(hipMalloc(&first.p, n_half * sizeof(uint32_t)));
  SG_HIP_TRY(hipMalloc(&first_sorted.p, n_half * sizeof(uint32_t)));
  SG_HIP_TRY(hipMalloc(&count.p, sizeof(int)));
  SG_HIP_TRY(hipMalloc(&flags.p, 3 * sizeof(int)));
  SG_HIP_TRY(hipMemsetAsync(flags.p, 0, 3 * sizeof(int), stream));
  half_edge_keys<<<blocks_for(n_half), kThreads, 0, stream>>>(faces, n_half, V, (uint64_t*)keys_a.p,
                                                             (uint32_t*)half_a.p, (int*)flags.p);
  SG_HIP_TRY(hipGetLastError());

  int hi_bits = 1;
  while (hi_bits < 32 && ((uint64_t)V >> hi_bits) != 0) ++hi_bits;
  int idx_bits = 1;
  while (idx_bits < 32 && ((uint64_t)n_half >> idx_bits) != 0) ++idx_bits;
  size_t t1 = 0, t2 = 0, t3 = 0;
  SG_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, t1, (const uint64_t*)keys_a.p, (uint64_t*)keys_b.p,
                                                (const uint32_t*)half_a.p, (uint32_t*)half_b.p, (int)n_half, 0,
                                                32 + hi_bits, stream));
  SG_HIP_TRY(hipcub::DeviceSelect::Flagged(nullptr, t2, (const uint32_t*)half_b.p, (const uint8_t*)heads.p,
                                           (uint32_t*)first.p, (int*)count.p, (int)n_half, stream));
  SG_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, t3, (const uint32_t*)first.p, (uint32_t*)first_sorted.p,
                                               (int)n_half, 0, idx_bits, stream));
  size_t tb = t1 > t2 ? t1 : t2;
  if (t3 > tb) tb = t3;
  SG_HIP_TRY(hipMalloc(&temp.p, tb ? tb : 16));
  SG_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(temp.p, t1, (const uint64_t*)keys_a.p, (uint64_t*)keys_b.p,
                                                (const uint32_t*)half_a.p, (uint32_t*)half_b.p, (int)n_half, 0,
                                                32 + hi_bits, stream));
  scan_runs<<<blocks_for(n_half), kThreads, 0, stream>>>((const uint64_t*)keys_b.p, (const uint32_t*)half_b.p, n_half,
                                                        (uint8_t*)heads.p, f2f_out, (int*)flags.p);
  SG_HIP_TRY(hipGetLastError());
  if (f2f_out) compact_f2f<<<blocks_for(F), kThreads, 0, stream>>>(f2f_out, F);
  SG_HIP_TRY(hipcub::DeviceSelect::Flagged(temp.p, t2, (const uint32_t*)half_b.p, (const uint8_t*)heads.p,
                                           (uint32_t*)first.p, (int*)count.p, (int)n_half, stream));
  int h_count = 0, h_flags[3] = {0, 0, 0};
  SG_HIP_TRY(hipMemcpyAsync(&h_count, count.p, sizeof(int), hipMemcpyDeviceToHost, stream));
  SG_HIP_TRY(hipMemcpyAsync(h_flags, flags.p, sizeof(h_flags), hipMemcpyDeviceToHost, stream));
  SG_HIP_TRY(hipStreamSynchronize(stream));
  SG_REQUIRE(!h_flags[0], "sg_mesh_edges: face refers to a vertex outside [0, %lld)", (long long)V);
  SG_REQUIRE(!h_flags[1], "sg_mesh_edges: degenerate face (repeated vertex)");
  if (manifold_out) *manifold_out = h_flags[2] ? 0 : 1;
  if (h_count > 0) {
    SG_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(temp.p, t3, (const uint32_t*)first.p, (uint32_t*)first_sorted.p,
                                                 h_count, 0, idx_bits, stream));
    edges_from_heads<<<blocks_for(h_count), kThreads, 0, stream>>>((const uint32_t*)first_sorted.p, h_count, faces,
                                                                  edges_out);
    SG_HIP_TRY(hipGetLastError());
    SG_HIP_TRY(hipStreamSynchronize(stream));   // the temporaries are freed on return
  }
  *n_edges_out = h_count;
  return SG_OK;
}

int launch_mask_dilate(const Csr& c, const uint64_t* in, uint64_t* out, int64_t W, hipStream_t stream) {
  const int64_t n = c.n_rows * W;
  if (n == 0) return SG_OK;
  dilate_bits<<<blocks_for(n), kThreads, 0, stream>>>(c.rowptr, c.idx, c.n_rows, (int)W, in, out);
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

int launch_face_mask(const int64_t* faces, int64_t F, int64_t V, const uint64_t* vbits, uint64_t* fbits, int64_t W,
                     hipStream_t stream) {
  const int64_t n = F * W;
  if (n == 0) return SG_OK;
  DeviceBuf bad;
  SG_HIP_TRY(hipMalloc(&bad.p, sizeof(int)));
  SG_HIP_TRY(hipMemsetAsync(bad.p, 0, sizeof(int), stream));
  face_and_bits<<<blocks_for(n), kThreads, 0, stream>>>(faces, F, V, (int)W, vbits, fbits, (int*)bad.p);
  SG_HIP_TRY(hipGetLastError());
  int h_bad = 0;
  SG_HIP_TRY(hipMemcpyAsync(&h_bad, bad.p, sizeof(int), hipMemcpyDeviceToHost, stream));
  SG_HIP_TRY(hipStreamSynchronize(stream));
  SG_REQUIRE(!h_bad, "sg_face_mask: face refers to a vertex outside [0, %lld)", (long long)V);
  return SG_OK;
}

}  // namespace sg
