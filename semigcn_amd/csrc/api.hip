// extern "C" entry points of libsemigcn_hip.so (declared in include/semigcn.h).
#include <stdarg.h>

#include <new>

#include <mutex>

#include "sg_common.h"

namespace sg {

static thread_local std::string t_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  t_error = buf;
}

namespace {

struct ScopedFree {
  void* p = nullptr;
  ~ScopedFree() { if (p) (void)hipFree(p); }
};

void destroy_graph(sg_graph* g) {
  if (!g) return;
  g->fwd.release();
  g->bwd.release();
  g->loc.release();
  if (g->row_id) (void)hipFree(g->row_id);
  if (g->dis_dst_loc) (void)hipFree(g->dis_dst_loc);
  if (g->dis_src && g->dis_src != g->dis_dst) (void)hipFree(g->dis_src);
  if (g->dis_dst) (void)hipFree(g->dis_dst);
  delete g;
}

void destroy_pool(sg_pool* p) {
  if (!p) return;
  p->by_coarse.release();
  p->by_fine.release();
  if (p->inv_count) (void)hipFree(p->inv_count);
  delete p;
}

int check_dense(const char* what, const void* X, int64_t ld, int64_t C) {
  SG_REQUIRE(X != nullptr, "%s: null pointer", what);
  SG_REQUIRE(ld >= C, "%s: row stride %lld < C %lld", what, (long long)ld, (long long)C);
  return SG_OK;
}

// The tile records of spmm_ring, at the first aggregation that can use them (see Csr::rec_pending).  Not while the stream is
// being captured into a hipGraph (the build allocates and synchronises): such a call runs on the rows kernel and the records
// wait for the next eager one -- every trainer here runs its warm-up iterations eagerly before it captures.
std::mutex g_rec_mu;
int build_pending_records(const Csr& c_, hipStream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &st) != hipSuccess) (void)hipGetLastError();
  if (st != hipStreamCaptureStatusNone) return SG_OK;
  std::lock_guard<std::mutex> lock(g_rec_mu);          // (forward and backward passes run on different host threads)
  Csr& c = const_cast<Csr&>(c_);
  if (!__atomic_load_n(&c.rec_pending, __ATOMIC_ACQUIRE)) return SG_OK;
  // the flag is cleared only AFTER a successful build (release: a thread that reads it false, outside the mutex, also
  // sees the finished record pointers); a failed build -- out of memory, say -- leaves it set, so the next aggregation tries
  // again instead of the graph staying on the rows kernel for good
  const int rc = build_ring_records(&c, c.pend_scale_src, c.pend_scale_dst, c.pend_row_id, stream);
  if (rc == SG_OK) __atomic_store_n(&c.rec_pending, false, __ATOMIC_RELEASE);
  return rc;
}

int run_csr(const Csr& c, const float* sd, const float* ss, const void* X, int64_t ldx,
            const void* X0, int64_t ldx0, const void* X1, int64_t ldx1, void* Y, int64_t ldy,
            int64_t C, int dtype, float alpha, float beta, float gamma, hipStream_t stream,
            const int32_t* row_id = nullptr) {
  SG_REQUIRE(C >= 0 && C <= INT32_MAX, "C out of range");
  if (c.n_rows == 0 || C == 0) return SG_OK;
  int rc;
  if ((rc = check_dense("X", X, ldx, C)) != SG_OK) return rc;
  if ((rc = check_dense("Y", Y, ldy, C)) != SG_OK) return rc;
  if (X0 && (rc = check_dense("X0", X0, ldx0, C)) != SG_OK) return rc;
  if (X1 && (rc = check_dense("X1", X1, ldx1, C)) != SG_OK) return rc;
  SG_REQUIRE(X != Y, "Y must not alias X");
  if (__atomic_load_n(&c.rec_pending, __ATOMIC_ACQUIRE) && (dtype == SG_BF16 || (dtype == SG_F32 && ring_f32_enabled())) && (C == 128 || C == 256) &&
      (rc = build_pending_records(c, stream)) != SG_OK)
    return rc;
  SpmmArgs a;
  a.rowptr = c.rowptr;
  a.idx = c.idx;
  if (c.tile_rows == kTileRows) {
    a.tile_uptr = c.tile_uptr;
    a.tile_uniq = c.tile_uniq;
    a.tile_eloc = c.tile_eloc;
  }
  if (ss != nullptr && ss == c.packed_scale) {
    a.idx_w = c.idx_w;
    a.tile_uniq_w = c.tile_uniq_w;
    if (c.lt_uptr && c.lt_uniq_w && c.idx_w) {
      a.lt_idx_w = c.idx_w;
      a.lt_uptr = c.lt_uptr;
      a.lt_eloc = c.lt_eloc;
      a.lt_uniq_w = c.lt_uniq_w;
    }
  }
  if (ss != nullptr && ss == c.packed_scale && c.lt_rec && sd == c.rec_scale_dst && row_id == c.rec_row_id) {
    a.lt_rec = c.lt_rec;
    a.lt_nrec = c.lt_nrec;
    a.lt_idx_w = c.idx_w;
  }
  a.row_id = row_id;
  a.scale_dst = sd;
  a.scale_src = ss;
  a.X = X; a.X0 = X0; a.X1 = X1; a.Y = Y;
  a.ldx = ldx; a.ldx0 = X0 ? ldx0 : 0; a.ldx1 = X1 ? ldx1 : 0; a.ldy = ldy;
  a.n_rows = (int32_t)c.n_rows;
  a.n_cols = c.n_cols;
  a.C = (int32_t)C;
  a.alpha = alpha; a.beta = beta; a.gamma = gamma;
  return launch_spmm(a, dtype, stream);
}

}  // namespace
}  // namespace sg

using namespace sg;

extern "C" {

SG_API const char* sg_last_error(void) { return t_error.c_str(); }

SG_API int sg_abi_version(void) { return SG_ABI_VERSION; }

SG_API int sg_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    return SG_ERR_NO_DEVICE;
  }
  return n;
}

SG_API int sg_graph_create(const int64_t* edge_index, int64_t E, int64_t V, void* stream_, sg_graph** out) {
  SG_REQUIRE(out != nullptr, "sg_graph_create: out is null");
  *out = nullptr;
  SG_REQUIRE(E >= 0 && V >= 0, "sg_graph_create: negative size");
  SG_REQUIRE(E == 0 || edge_index != nullptr, "sg_graph_create: edge_index is null");
  hipStream_t stream = (hipStream_t)stream_;
  sg_graph* g = new (std::nothrow) sg_graph();
  SG_REQUIRE(g != nullptr, "out of host memory");
  const int64_t* src = edge_index;      // row 0: source j
  const int64_t* dst = edge_index + E;  // row 1: target i
  ScopedFree kf, kb;
  int rc = SG_OK;
  do {
    if (E > 0) {
      if (hipMalloc(&kf.p, E * sizeof(uint64_t)) != hipSuccess ||
          hipMalloc(&kb.p, E * sizeof(uint64_t)) != hipSuccess) {
        set_error("hipMalloc of sort keys failed");
        rc = SG_ERR_HIP;
        break;
      }
    }
    if ((rc = build_csr(dst, src, E, V, V, true, stream, &g->fwd, (uint64_t*)kf.p)) != SG_OK) break;
    // transposed: rows = sources.  Its row lengths are the PyG degrees (taken over edge_index[0]).
    if ((rc = build_csr(src, dst, E, V, V, true, stream, &g->bwd, (uint64_t*)kb.p)) != SG_OK) break;
    if (hipMalloc((void**)&g->dis_dst, (V > 0 ? V : 1) * sizeof(float)) != hipSuccess) {
      set_error("hipMalloc of dis failed");
      rc = SG_ERR_HIP;
      break;
    }
    g->dis_src = g->dis_dst;
    if ((rc = degree_scale(g->bwd, true, g->dis_dst, stream)) != SG_OK) break;
    int equal = 1;
    if ((rc = keys_equal((const uint64_t*)kf.p, (const uint64_t*)kb.p, g->fwd.nnz, stream, &equal)) != SG_OK) break;
    g->symmetric = equal != 0;
    g->square = true;
    if (g->symmetric) g->bwd.release();
    if (tiles_enabled()) {
      if ((rc = build_tiles(&g->fwd, stream)) != SG_OK) break;
      if (!g->symmetric && (rc = build_tiles(&g->bwd, stream)) != SG_OK) break;
    }
    if ((rc = pack_source_scale(&g->fwd, g->dis_src, stream)) != SG_OK) break;
    if (!g->symmetric && (rc = pack_source_scale(&g->bwd, g->dis_src, stream)) != SG_OK) break;
    if (ring_enabled() && !g->symmetric) g->bwd.defer_records(g->dis_src, g->dis_dst, nullptr);
    if (g->symmetric) {      // numbering without locality (a raw scan): process the rows in a graph-derived order
      if ((rc = locality_order(g->fwd, graph_reorder_mode(), stream, &g->row_id)) != SG_OK) break;
    }
    // tile records of the forward CSR only where sg_spmm will read them: a graph that gets a locality view is always
    // applied through that view (its records are built below), so records of the plain row order would be dead weight
    // (896 bytes per 16 rows for the graph's lifetime, a sort and two stream syncs at creation)
    if (ring_enabled() && !g->row_id) g->fwd.defer_records(g->dis_src, g->dis_dst, nullptr);
    if (g->symmetric) {
      if (g->row_id) {
        if ((rc = permute_rows(g->fwd, g->row_id, stream, &g->loc)) != SG_OK) break;
        if (hipMalloc((void**)&g->dis_dst_loc, V * sizeof(float)) != hipSuccess) {
          set_error("hipMalloc of the permuted dis failed");
          rc = SG_ERR_HIP;
          break;
        }
        if ((rc = gather_floats(g->dis_dst, g->row_id, V, g->dis_dst_loc, stream)) != SG_OK) break;
        if (tiles_enabled() && (rc = build_tiles(&g->loc, stream)) != SG_OK) break;
        if ((rc = pack_source_scale(&g->loc, g->dis_src, stream)) != SG_OK) break;
        if (ring_enabled()) g->loc.defer_records(g->dis_src, g->dis_dst_loc, g->row_id);
      }
    }
    if (hipStreamSynchronize(stream) != hipSuccess) {
      set_error("stream sync failed in sg_graph_create");
      rc = SG_ERR_HIP;
      break;
    }
  } while (0);
  if (rc != SG_OK) {
    destroy_graph(g);
    return rc;
  }
  *out = g;
  return SG_OK;
}

SG_API int sg_graph_create_rect(const int64_t* dst, const int64_t* src, int64_t n, int64_t V_dst,
                         int64_t V_src, const float* dis_src, void* stream_, sg_graph** out) {
  SG_REQUIRE(out != nullptr, "sg_graph_create_rect: out is null");
  *out = nullptr;
  SG_REQUIRE(n >= 0 && V_dst >= 0 && V_src >= V_dst, "sg_graph_create_rect: need V_src >= V_dst >= 0");
  SG_REQUIRE(V_src == 0 || dis_src != nullptr, "sg_graph_create_rect: dis_src is null");
  hipStream_t stream = (hipStream_t)stream_;
  sg_graph* g = new (std::nothrow) sg_graph();
  SG_REQUIRE(g != nullptr, "out of host memory");
  int rc = build_csr(dst, src, n, V_dst, V_src, false, stream, &g->fwd, nullptr);
  if (rc == SG_OK && tiles_enabled()) rc = build_tiles(&g->fwd, stream);
  if (rc == SG_OK) {
    if (hipMalloc((void**)&g->dis_src, (V_src > 0 ? V_src : 1) * sizeof(float)) != hipSuccess) {
      set_error("hipMalloc of dis failed");
      rc = SG_ERR_HIP;
    } else {
      g->dis_dst = g->dis_src;  // owned rows are the first V_dst columns
      if (V_src > 0 &&
          (hipMemcpyAsync(g->dis_src, dis_src, V_src * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess ||
           hipStreamSynchronize(stream) != hipSuccess)) {
        set_error("copy of dis failed");
        rc = SG_ERR_HIP;
      }
      if (rc == SG_OK) rc = pack_source_scale(&g->fwd, g->dis_src, stream);
      if (rc == SG_OK && ring_enabled()) g->fwd.defer_records(g->dis_src, g->dis_dst, nullptr);
      if (rc == SG_OK && hipStreamSynchronize(stream) != hipSuccess) {
        set_error("stream sync failed in sg_graph_create_rect");
        rc = SG_ERR_HIP;
      }
    }
  }
  if (rc != SG_OK) {
    if (g->dis_src) { (void)hipFree(g->dis_src); g->dis_src = g->dis_dst = nullptr; }
    destroy_graph(g);
    return rc;
  }
  g->symmetric = true;  // the partitioned operator is applied owner-computes in both directions
  g->square = false;
  *out = g;
  return SG_OK;
}

SG_API int sg_graph_create_rows(const int64_t* dst_pos, const int64_t* src, int64_t n, int64_t n_rows, int64_t V_src,
                                const int32_t* row_id, const float* dis_rows, const float* dis_src, void* stream_,
                                sg_graph** out) {
  SG_REQUIRE(out != nullptr, "sg_graph_create_rows: out is null");
  *out = nullptr;
  SG_REQUIRE(n >= 0 && n_rows >= 0 && V_src >= 0, "sg_graph_create_rows: negative size");
  SG_REQUIRE((n_rows == 0 || (row_id && dis_rows)) && (V_src == 0 || dis_src), "sg_graph_create_rows: null pointer");
  hipStream_t stream = (hipStream_t)stream_;
  sg_graph* g = new (std::nothrow) sg_graph();
  SG_REQUIRE(g != nullptr, "out of host memory");
  int rc = build_csr(dst_pos, src, n, n_rows, V_src, false, stream, &g->fwd, nullptr);
  if (rc == SG_OK && tiles_enabled()) rc = build_tiles(&g->fwd, stream);
  if (rc == SG_OK) {
    const size_t nr = n_rows > 0 ? n_rows : 1, ns = V_src > 0 ? V_src : 1;
    if (hipMalloc((void**)&g->dis_src, ns * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&g->dis_dst, nr * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&g->row_id, nr * sizeof(int32_t)) != hipSuccess) {
      set_error("hipMalloc failed in sg_graph_create_rows");
      rc = SG_ERR_HIP;
    } else if ((V_src > 0 && hipMemcpyAsync(g->dis_src, dis_src, V_src * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess) ||
               (n_rows > 0 && (hipMemcpyAsync(g->dis_dst, dis_rows, n_rows * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess ||
                               hipMemcpyAsync(g->row_id, row_id, n_rows * sizeof(int32_t), hipMemcpyDeviceToDevice, stream) != hipSuccess)) ||
               hipStreamSynchronize(stream) != hipSuccess) {
      set_error("copy failed in sg_graph_create_rows");
      rc = SG_ERR_HIP;
    }
    if (rc == SG_OK) rc = pack_source_scale(&g->fwd, g->dis_src, stream);
    if (rc == SG_OK && ring_enabled()) g->fwd.defer_records(g->dis_src, g->dis_dst, g->row_id);
    if (rc == SG_OK && hipStreamSynchronize(stream) != hipSuccess) {
      set_error("stream sync failed in sg_graph_create_rows");
      rc = SG_ERR_HIP;
    }
  }
  if (rc != SG_OK) {
    destroy_graph(g);
    return rc;
  }
  g->symmetric = true;   // applied owner-computes in both directions, like the rectangular operator
  g->square = false;
  *out = g;
  return SG_OK;
}

SG_API int sg_graph_destroy(sg_graph* g) {
  destroy_graph(g);
  return SG_OK;
}

SG_API int sg_graph_query(const sg_graph* g, sg_graph_info* info) {
  SG_REQUIRE(g && info, "sg_graph_query: null argument");
  info->V_dst = g->fwd.n_rows;
  info->V_src = g->fwd.n_cols;
  info->nnz = g->fwd.nnz;
  info->symmetric = g->symmetric ? 1 : 0;
  info->max_degree = g->fwd.max_degree;
  return SG_OK;
}

SG_API int sg_graph_is_reordered(const sg_graph* g) {
  if (!g) {
    set_error("sg_graph_is_reordered: null graph");
    return SG_ERR_INVALID;
  }
  return (g->square && g->row_id) ? 1 : 0;
}

SG_API int sg_graph_export(const sg_graph* g, int32_t* rowptr, int32_t* colidx, float* dis, void* stream_) {
  SG_REQUIRE(g != nullptr, "sg_graph_export: null graph");
  hipStream_t stream = (hipStream_t)stream_;
  if (rowptr)
    SG_HIP_TRY(hipMemcpyAsync(rowptr, g->fwd.rowptr, (g->fwd.n_rows + 1) * sizeof(int32_t),
                              hipMemcpyDeviceToDevice, stream));
  if (colidx && g->fwd.nnz > 0)
    SG_HIP_TRY(hipMemcpyAsync(colidx, g->fwd.idx, g->fwd.nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
  if (dis && g->fwd.n_cols > 0)
    SG_HIP_TRY(hipMemcpyAsync(dis, g->dis_src, g->fwd.n_cols * sizeof(float), hipMemcpyDeviceToDevice, stream));
  return SG_OK;
}

SG_API int sg_spmm(const sg_graph* g, int transpose, const void* X, int64_t ldx, const void* X0, int64_t ldx0,
            const void* X1, int64_t ldx1, void* Y, int64_t ldy, int64_t C, int dtype, float alpha,
            float beta, float gamma, void* stream) {
  SG_REQUIRE(g != nullptr, "sg_spmm: null graph");
  const bool t = transpose != 0;
  if (t && !g->square) {
    set_error("sg_spmm: transpose of a rectangular (partition) operator is not available; "
              "apply it owner-computes on exchanged gradient rows");
    return SG_ERR_UNSUPPORTED;
  }
  // (benchmarking aid, off by default: an event pair around the launch, sg_trace_begin)
  TraceScope trace(0, dtype, 0, C, (X0 ? 1 : 0) + (X1 ? 1 : 0), (t && !g->symmetric) ? g->bwd.n_rows : g->fwd.n_rows,
                   (hipStream_t)stream);
  // L^[i,j] = -dis[i] dis[j] (#edges j->i); the transposed CSR carries the same scales
  if (g->square && g->symmetric && g->row_id && g->loc.rowptr)   // locality view: rows in processing order, output rows addressed through row_id
    return run_csr(g->loc, g->dis_dst_loc, g->dis_src, X, ldx, X0, ldx0, X1, ldx1, Y, ldy, C, dtype, -alpha, beta, gamma,
                   (hipStream_t)stream, g->row_id);
  if (!g->square && g->row_id)         // row subset of a partition operator (sg_graph_create_rows)
    return run_csr(g->fwd, g->dis_dst, g->dis_src, X, ldx, X0, ldx0, X1, ldx1, Y, ldy, C, dtype, -alpha, beta, gamma,
                   (hipStream_t)stream, g->row_id);
  const Csr& c = (t && !g->symmetric) ? g->bwd : g->fwd;
  return run_csr(c, g->dis_dst, g->dis_src, X, ldx, X0, ldx0, X1, ldx1, Y, ldy, C, dtype, -alpha,
                 beta, gamma, (hipStream_t)stream);
}

// the lazy build makes the FIRST sg_spmm of 128 / 256 channels on a graph allocate and synchronise.  A host that wants
// every sg_spmm asynchronous (or that captures its first iteration into a hipGraph on another stream) builds the records here.
SG_API int sg_graph_prepare(const sg_graph* g, int64_t C, int dtype, void* stream) {
  SG_REQUIRE(g != nullptr, "sg_graph_prepare: null graph");
  if (!(C == 128 || C == 256) || !(dtype == SG_BF16 || (dtype == SG_F32 && ring_f32_enabled()))) return SG_OK;   // nothing to build
  const Csr* all[3] = {&g->fwd, &g->bwd, &g->loc};
  for (const Csr* c : all) {
    if (!__atomic_load_n(&const_cast<Csr*>(c)->rec_pending, __ATOMIC_ACQUIRE)) continue;
    const int rc = build_pending_records(*c, (hipStream_t)stream);
    if (rc != SG_OK) return rc;
  }
  return SG_OK;
}

SG_API int sg_pool_create(const int64_t* fine, const int64_t* coarse, int64_t n, int64_t n_fine,
                   int64_t n_coarse, void* stream_, sg_pool** out) {
  SG_REQUIRE(out != nullptr, "sg_pool_create: out is null");
  *out = nullptr;
  SG_REQUIRE(n >= 0 && n_fine >= 0 && n_coarse >= 0, "sg_pool_create: negative size");
  hipStream_t stream = (hipStream_t)stream_;
  sg_pool* p = new (std::nothrow) sg_pool();
  SG_REQUIRE(p != nullptr, "out of host memory");
  int rc = build_csr(coarse, fine, n, n_coarse, n_fine, false, stream, &p->by_coarse, nullptr);
  if (rc == SG_OK) rc = build_csr(fine, coarse, n, n_fine, n_coarse, false, stream, &p->by_fine, nullptr);
  if (rc == SG_OK && hipMalloc((void**)&p->inv_count, (n_coarse > 0 ? n_coarse : 1) * sizeof(float)) != hipSuccess) {
    set_error("hipMalloc of inv_count failed");
    rc = SG_ERR_HIP;
  }
  if (rc == SG_OK) rc = degree_scale(p->by_coarse, false, p->inv_count, stream);
  if (rc == SG_OK && hipStreamSynchronize(stream) != hipSuccess) {
    set_error("stream sync failed in sg_pool_create");
    rc = SG_ERR_HIP;
  }
  if (rc != SG_OK) {
    destroy_pool(p);
    return rc;
  }
  *out = p;
  return SG_OK;
}

SG_API int sg_pool_destroy(sg_pool* p) {
  destroy_pool(p);
  return SG_OK;
}

SG_API int sg_pool_mean(const sg_pool* p, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t C, int dtype,
                 void* stream) {
  SG_REQUIRE(p != nullptr, "sg_pool_mean: null pool");
  TraceScope trace(3, dtype, 0, C, p->by_coarse.n_rows, p->by_coarse.n_cols, (hipStream_t)stream);
  return run_csr(p->by_coarse, p->inv_count, nullptr, X, ldx, nullptr, 0, nullptr, 0, Y, ldy, C, dtype,
                 1.f, 0.f, 0.f, (hipStream_t)stream);
}

SG_API int sg_pool_mean_bwd(const sg_pool* p, const void* dY, int64_t lddy, void* dX, int64_t lddx, int64_t C,
                     int dtype, void* stream) {
  SG_REQUIRE(p != nullptr, "sg_pool_mean_bwd: null pool");
  TraceScope trace(3, dtype, 0, C, p->by_fine.n_rows, p->by_fine.n_cols, (hipStream_t)stream);
  return run_csr(p->by_fine, nullptr, p->inv_count, dY, lddy, nullptr, 0, nullptr, 0, dX, lddx, C, dtype,
                 1.f, 0.f, 0.f, (hipStream_t)stream);
}

SG_API int sg_unpool(const sg_pool* p, const void* X, int64_t ldx, void* Y, int64_t ldy, int64_t C, int dtype,
              void* stream) {
  SG_REQUIRE(p != nullptr, "sg_unpool: null pool");
  TraceScope trace(3, dtype, 0, C, p->by_fine.n_rows, p->by_fine.n_cols, (hipStream_t)stream);
  return run_csr(p->by_fine, nullptr, nullptr, X, ldx, nullptr, 0, nullptr, 0, Y, ldy, C, dtype, 1.f, 0.f,
                 0.f, (hipStream_t)stream);
}

SG_API int sg_unpool_bwd(const sg_pool* p, const void* dY, int64_t lddy, void* dX, int64_t lddx, int64_t C,
                  int dtype, void* stream) {
  SG_REQUIRE(p != nullptr, "sg_unpool_bwd: null pool");
  TraceScope trace(3, dtype, 0, C, p->by_coarse.n_rows, p->by_coarse.n_cols, (hipStream_t)stream);
  return run_csr(p->by_coarse, nullptr, nullptr, dY, lddy, nullptr, 0, nullptr, 0, dX, lddx, C, dtype,
                 1.f, 0.f, 0.f, (hipStream_t)stream);
}

SG_API int sg_tuning_set(int knob, int value) {
  if (knob == SG_TUNE_GEMM_TILE) return set_gemm_tuning(value);
  if (knob == SG_TUNE_GRAPH_REORDER) return set_graph_reorder_mode(value);
  if (knob == SG_TUNE_BLOCK_PLANES) return set_block_planes(value);
  if (knob == SG_TUNE_F32_ENGINE) return set_split_tuning(value);
  if (knob == SG_TUNE_BN_ROWS) return set_bn_rows_tuning(value);
  return set_tuning(knob, value);
}

SG_API int64_t sg_mesh_loss_blocks(int64_t V, int64_t F) { return mesh_loss_blocks(V, F); }

SG_API int sg_mesh_loss_fwd(const float* pos, const int64_t* faces, const float* target_pos, const float* v_keep,
                            const float* target_fn, const float* f_keep, int64_t V, int64_t F, float* partial,
                            void* stream) {
  SG_REQUIRE(V >= 0 && F >= 0 && partial, "sg_mesh_loss_fwd: bad size or null partial");
  SG_REQUIRE((V == 0 || (pos && target_pos && v_keep)) && (F == 0 || (pos && faces && target_fn && f_keep)),
             "sg_mesh_loss_fwd: null pointer");
  return launch_mesh_loss_fwd(pos, faces, target_pos, v_keep, target_fn, f_keep, V, F, partial, (hipStream_t)stream);
}

SG_API int sg_mesh_loss_bwd(const float* pos, const int64_t* faces, const float* target_pos, const float* v_keep,
                            const float* target_fn, const float* f_keep, const float* g, int64_t V, int64_t V_ext,
                            int64_t F, float* grad_pos, void* stream) {
  SG_REQUIRE(V >= 0 && F >= 0 && V_ext >= V && g && (V_ext == 0 || grad_pos), "sg_mesh_loss_bwd: bad argument");
  SG_REQUIRE((V == 0 || (pos && target_pos && v_keep)) && (F == 0 || (pos && faces && target_fn && f_keep)),
             "sg_mesh_loss_bwd: null pointer");
  return launch_mesh_loss_bwd(pos, faces, target_pos, v_keep, target_fn, f_keep, g, V, V_ext, F, grad_pos,
                              (hipStream_t)stream);
}

SG_API int sg_mesh_edges(const int64_t* faces, int64_t F, int64_t V, int64_t* edges_out, int64_t* f2f_out,
                         int64_t* n_edges_out, int* manifold_out, void* stream) {
  SG_REQUIRE(F >= 0 && V >= 0 && n_edges_out, "sg_mesh_edges: bad argument");
  SG_REQUIRE(F == 0 || (faces && edges_out), "sg_mesh_edges: null pointer");
  return mesh_edges(faces, F, V, edges_out, f2f_out, n_edges_out, manifold_out, (hipStream_t)stream);
}

SG_API int sg_mask_dilate(const sg_graph* g, const uint64_t* in, uint64_t* out, int64_t W, void* stream) {
  SG_REQUIRE(g != nullptr && W >= 0, "sg_mask_dilate: bad argument");
  SG_REQUIRE(g->square, "sg_mask_dilate: needs a square graph");
  if (W == 0 || g->fwd.n_rows == 0) return SG_OK;
  SG_REQUIRE(in && out && in != out, "sg_mask_dilate: null or aliased buffers");
  return launch_mask_dilate(g->fwd, in, out, W, (hipStream_t)stream);
}

SG_API int sg_face_mask(const int64_t* faces, int64_t F, int64_t V, const uint64_t* vbits, uint64_t* fbits,
                        int64_t W, void* stream) {
  SG_REQUIRE(F >= 0 && V >= 0 && W >= 0, "sg_face_mask: negative size");
  if (F == 0 || W == 0) return SG_OK;
  SG_REQUIRE(faces && vbits && fbits, "sg_face_mask: null pointer");
  return launch_face_mask(faces, F, V, vbits, fbits, W, (hipStream_t)stream);
}

SG_API int sg_mesh_loss_bwd_det(const float* pos, const int64_t* faces, const float* target_pos, const float* v_keep,
                                const float* target_fn, const float* f_keep, const float* g, int64_t V, int64_t V_ext,
                                int64_t F, const sg_pool* incidence, float* corner_scratch, float* grad_pos, void* stream_) {
  SG_REQUIRE(V >= 0 && F >= 0 && V_ext >= V && g && (V_ext == 0 || grad_pos), "sg_mesh_loss_bwd_det: bad argument");
  SG_REQUIRE((V == 0 || (pos && target_pos && v_keep)) && (F == 0 || (pos && faces && target_fn && f_keep)),
             "sg_mesh_loss_bwd_det: null pointer");
  SG_REQUIRE(incidence && incidence->by_coarse.n_rows == V_ext && incidence->by_coarse.n_cols == 3 * F &&
                 (F == 0 || corner_scratch),
             "sg_mesh_loss_bwd_det: incidence must map the 3F face corners onto V_ext vertices");
  hipStream_t stream = (hipStream_t)stream_;
  int rc = launch_mesh_loss_bwd_corners(pos, faces, target_fn, f_keep, g, F, corner_scratch, stream);
  if (rc != SG_OK) return rc;
  if (V_ext == 0) return SG_OK;
  if (F == 0) {
    SG_HIP_TRY(hipMemsetAsync(grad_pos, 0, (size_t)V_ext * 3 * sizeof(float), stream));
  } else {   // grad[v] = sum over the corners incident to v, in CSR (ascending corner id) order
    rc = run_csr(incidence->by_coarse, nullptr, nullptr, corner_scratch, 3, nullptr, 0, nullptr, 0, grad_pos, 3, 3, SG_F32,
                 1.f, 0.f, 0.f, stream);
    if (rc != SG_OK) return rc;
  }
  return launch_mesh_loss_bwd_vertex_add(pos, target_pos, v_keep, g, V, grad_pos, stream);
}

SG_API int64_t sg_col_blocks(int64_t V) { return col_blocks(V); }

SG_API int sg_col_moments(const void* X, int64_t ldx, int64_t V, int64_t C, int dtype, float* partial, int64_t nb,
                          void* stream) {
  SG_REQUIRE(V >= 0 && C >= 0, "sg_col_moments: negative size");
  if (V == 0 || C == 0) return SG_OK;
  SG_REQUIRE(X && partial && ldx >= C, "sg_col_moments: bad argument");
  return launch_col_reduce(0, X, ldx, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0.f, partial, nb, V, C, dtype,
                           (hipStream_t)stream);
}

SG_API int sg_bn_merge(const float* partial, int64_t nb, int64_t V, int64_t C, float* stats, void* stream) {
  SG_REQUIRE(V > 0 && C >= 0 && partial && stats, "sg_bn_merge: bad argument");
  return launch_bn_merge(partial, nb, V, C, stats, (hipStream_t)stream);
}

SG_API int sg_bn_merge_tiles(const float* partial, int64_t n_tiles, int64_t rows_per_tile, int64_t V, int64_t C, float* stats,
                             float* count_out, void* stream) {
  SG_REQUIRE(V > 0 && C >= 0 && partial && stats, "sg_bn_merge_tiles: bad argument");
  return launch_bn_merge_tiles(partial, n_tiles, rows_per_tile, V, C, stats, count_out, (hipStream_t)stream);
}

SG_API int sg_bn_stats_finalize(const float* partial, int64_t nb, int64_t V, int64_t C, const float* gamma,
                                const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                float* out, int64_t* batches_tracked, void* stream) {
  SG_REQUIRE(V > 1 && C >= 0 && partial && gamma && beta && out, "sg_bn_stats_finalize: bad argument");
  SG_REQUIRE((running_mean == nullptr) == (running_var == nullptr),
             "sg_bn_stats_finalize: give both running buffers or none");
  return launch_bn_stats_finalize(partial, nb, V, C, gamma, beta, running_mean, running_var, momentum, eps, out,
                                  batches_tracked, (hipStream_t)stream);
}

SG_API int sg_bn_stats_finalize_tiles(const float* partial, int64_t n_tiles, int64_t rows_per_tile, int64_t V, int64_t C,
                                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                                      float momentum, float eps, float* out, int64_t* batches_tracked, void* stream) {
  SG_REQUIRE(V > 1 && C >= 0 && partial && gamma && beta && out, "sg_bn_stats_finalize_tiles: bad argument");
  SG_REQUIRE((running_mean == nullptr) == (running_var == nullptr),
             "sg_bn_stats_finalize_tiles: give both running buffers or none");
  return launch_bn_stats_finalize_tiles(partial, n_tiles, rows_per_tile, V, C, gamma, beta, running_mean, running_var,
                                        momentum, eps, out, batches_tracked, (hipStream_t)stream);
}

SG_API int sg_gemm_nt_takes_big_tile(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc) {
  return gemm_nt_takes_big_tile(M, N, K, lda, ldb, ldc) ? 1 : 0;
}

SG_API int64_t sg_gemm_tile_rows(int64_t N) { return gemm_tile_rows(N); }

SG_API int64_t sg_gemm_row_tiles(int64_t M, int64_t N) {
  const int64_t r = gemm_tile_rows(N);
  return M <= 0 ? 0 : (M + r - 1) / r;
}

SG_API int sg_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, const float* bias, void* C, int64_t ldc,
                      int64_t M, int64_t N, int64_t K, int dtype, float* moments, void* stream) {
  SG_REQUIRE(M >= 0 && N >= 0 && K >= 0, "sg_gemm_nt: negative size");
  if (M == 0 || N == 0) return SG_OK;
  SG_REQUIRE(A && B && C && K > 0, "sg_gemm_nt: null operand or K = 0");
  SG_REQUIRE(lda >= K && ldb >= K && ldc >= N, "sg_gemm_nt: row stride shorter than the row");
  return launch_gemm_nt(A, lda, B, ldb, bias, C, ldc, M, N, K, dtype, moments, (hipStream_t)stream);
}

SG_API int64_t sg_gemm_tn_slabs(int64_t M, int64_t N, int64_t Kp) { return gemm_tn_slabs(M, N, Kp); }

SG_API int sg_gemm_tn_takes_big_tile(int64_t M, int64_t N, int64_t Kp, int64_t lda, int64_t ldb) {
  return gemm_tn_takes_big_tile(M, N, Kp, lda, ldb) ? 1 : 0;
}

SG_API int sg_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t Kp, int dtype,
                      float* workspace, float* out, int64_t ldo, void* stream) {
  SG_REQUIRE(M >= 0 && N >= 0 && Kp >= 0, "sg_gemm_tn: negative size");
  if (N == 0 || Kp == 0) return SG_OK;
  SG_REQUIRE(out != nullptr && ldo >= Kp, "sg_gemm_tn: bad output");
  if (M == 0) {
    SG_HIP_TRY(hipMemset2DAsync(out, ldo * sizeof(float), 0, Kp * sizeof(float), N, (hipStream_t)stream));
    return SG_OK;
  }
  SG_REQUIRE(A && B && workspace && lda >= N && ldb >= Kp, "sg_gemm_tn: null operand or short row stride");
  return launch_gemm_tn(A, lda, B, ldb, M, N, Kp, dtype, workspace, out, ldo, (hipStream_t)stream);
}

SG_API int sg_gemm_nt_f32_supported(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldc) {
  return (gemm_nt_f32s_supported(M, N, K, lda, ldc) || (M >= 0 && mid_shape(N, K) && lda % 4 == 0 && ldc % 4 == 0)) ? 1 : 0;
}

SG_API int sg_gemm_nt_f32_variant(int64_t M) { return gemm_nt_f32s_variant(M); }

SG_API int sg_gemm_nt_f32_pays(int64_t M, int64_t N, int64_t K) {
  return (mid_shape(N, K) || split_nt_pays(M, N, K)) ? 1 : 0;
}

SG_API int64_t sg_gemm_nt_f32_workspace(int64_t N, int64_t K) {
  return (N > 0 && K > 0 && !mid_shape(N, K)) ? gemm_nt_f32s_workspace(N, K) : 0;
}

SG_API int sg_gemm_nt_f32(const float* A, int64_t lda, const float* W, int64_t w_rs, int64_t w_cs, const float* bias, float* C,
                          int64_t ldc, int64_t M, int64_t N, int64_t K, void* workspace, int64_t workspace_bytes, void* stream) {
  SG_REQUIRE(M >= 0 && N >= 0 && K >= 0, "sg_gemm_nt_f32: negative size");
  if (M == 0 || N == 0) return SG_OK;
  SG_REQUIRE(A && W && C && K > 0, "sg_gemm_nt_f32: null operand or K = 0");
  SG_REQUIRE(lda >= K && ldc >= N, "sg_gemm_nt_f32: row stride shorter than the row");
  if (mid_shape(N, K)) return launch_mid_nt(A, lda, W, w_rs, w_cs, bias, C, ldc, M, N, K, (hipStream_t)stream);
  return launch_gemm_nt_f32s(A, lda, W, w_rs, w_cs, bias, C, ldc, M, N, K, workspace, workspace_bytes, (hipStream_t)stream);
}

SG_API int sg_gemm_tn_f32_supported(int64_t M, int64_t N, int64_t Kp, int64_t lda, int64_t ldb) {
  return (gemm_tn_f32s_supported(M, N, Kp, lda, ldb) || (M > 0 && mid_shape(N, Kp) && lda % 4 == 0 && ldb % 4 == 0)) ? 1 : 0;
}

SG_API int64_t sg_gemm_tn_f32_workspace(int64_t M, int64_t N, int64_t Kp) {
  if (M > 0 && mid_shape(N, Kp)) return mid_tn_workspace(M, N, Kp) * 4;
  return M > 0 && N > 0 && Kp > 0 ? gemm_tn_f32s_workspace(M, N, Kp) : 0;
}

SG_API int sg_gemm_tn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t M, int64_t N, int64_t Kp,
                          void* workspace, int64_t workspace_bytes, float* out, int64_t ldo, void* stream) {
  SG_REQUIRE(M >= 0 && N >= 0 && Kp >= 0, "sg_gemm_tn_f32: negative size");
  if (N == 0 || Kp == 0) return SG_OK;
  SG_REQUIRE(A && B && out && ldo >= Kp && lda >= N && ldb >= Kp, "sg_gemm_tn_f32: null operand or short row stride");
  if (mid_shape(N, Kp)) {
    SG_REQUIRE(workspace && workspace_bytes >= mid_tn_workspace(M, N, Kp) * 4, "sg_gemm_tn_f32: workspace too small");
    return launch_mid_tn(A, lda, B, ldb, M, N, Kp, (float*)workspace, out, ldo, (hipStream_t)stream, nullptr);
  }
  return launch_gemm_tn_f32s(A, lda, B, ldb, M, N, Kp, workspace, workspace_bytes, out, ldo, (hipStream_t)stream);
}

SG_API int64_t sg_thin_tn_blocks(int64_t V) { return thin_tn_blocks(V); }

SG_API int sg_thin_supported(int64_t N, int64_t K) { return thin_shape(N, K) ? 1 : 0; }

SG_API int sg_thin_nt(const void* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, void* Y, int64_t ldy,
                      int64_t V, int64_t N, int64_t K, int dtype, void* stream) {
  SG_REQUIRE(V >= 0 && thin_shape(N, K), "sg_thin_nt: N x K is not a thin shape, see sg_thin_supported (N=%lld K=%lld)", (long long)N, (long long)K);
  SG_REQUIRE(V == 0 || (X && W && Y && ldx >= K && ldw >= K && ldy >= N), "sg_thin_nt: null operand or short row stride");
  SG_REQUIRE(X != Y, "sg_thin_nt: Y must not alias X");
  return launch_thin_nt(X, ldx, W, ldw, bias, Y, ldy, V, N, K, dtype, (hipStream_t)stream);
}

SG_API int sg_thin_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t V, int64_t N, int64_t K, int dtype,
                      float* workspace, float* out, int64_t ldo, void* stream) {
  SG_REQUIRE(V >= 0 && thin_shape(N, K), "sg_thin_tn: N x K is not a thin shape, see sg_thin_supported (N=%lld K=%lld)", (long long)N, (long long)K);
  SG_REQUIRE(out != nullptr && ldo >= K && workspace != nullptr, "sg_thin_tn: bad output or workspace");
  SG_REQUIRE(V == 0 || (A && B && lda >= N && ldb >= K), "sg_thin_tn: null operand or short row stride");
  return launch_thin_tn(A, lda, B, ldb, V, N, K, dtype, workspace, out, ldo, (hipStream_t)stream);
}

SG_API int sg_bn_bwd_coeffs(const float* partial, int64_t nb, int64_t C, double N, const float* gamma,
                            const float* invstd, float* out, float* acc_dweight, float* acc_dbias, const float* count_dev,
                            void* stream) {
  SG_REQUIRE(nb > 0 && C >= 0 && (N > 0 || count_dev) && partial && gamma && invstd && out, "sg_bn_bwd_coeffs: bad argument");
  return launch_bn_bwd_coeffs(partial, nb, C, N, gamma, invstd, out, acc_dweight, acc_dbias, count_dev, (hipStream_t)stream);
}

SG_API int64_t sg_input_prep_blocks(int64_t V) { return input_prep_blocks(V); }

SG_API int sg_input_prep(const float* z1, const float* dm, const int64_t* order, const float* lo, const float* hi, void* X,
                         int64_t ldx, int64_t V, int dtype, void* stream) {
  SG_REQUIRE(V >= 0 && ldx >= 4 && (V == 0 || (z1 && lo && hi && X)), "sg_input_prep: bad argument");
  SG_REQUIRE(dtype == SG_F32 || dtype == SG_BF16, "sg_input_prep: unknown dtype %d", dtype);
  return launch_input_prep(z1, dm, order, lo, hi, X, ldx, V, dtype, (hipStream_t)stream);
}

SG_API int sg_input_prep_bwd(const void* gX, int64_t ldg, const float* z1, const float* dm, const int64_t* rank,
                             const float* lo, const float* hi, float* dz1, float* partial, float* d_lo, float* d_hi, int64_t V,
                             int dtype, void* stream) {
  SG_REQUIRE(V >= 0 && ldg >= 3 && (V == 0 || (gX && z1 && lo && hi && partial && d_lo && d_hi)), "sg_input_prep_bwd: bad argument");
  SG_REQUIRE(dtype == SG_F32 || dtype == SG_BF16, "sg_input_prep_bwd: unknown dtype %d", dtype);
  return launch_input_prep_bwd(gX, ldg, z1, dm, rank, lo, hi, dz1, partial, d_lo, d_hi, V, dtype, (hipStream_t)stream);
}

SG_API int sg_input_bounds(const float* z1, int64_t V, float* partial_values, int64_t* partial_index, float* bounds, int64_t* arg,
                           void* stream) {
  SG_REQUIRE(V > 0 && z1 && partial_values && partial_index && bounds && arg, "sg_input_bounds: bad argument");
  return launch_input_bounds(z1, V, partial_values, partial_index, bounds, arg, (hipStream_t)stream);
}

SG_API int sg_input_prep_bwd_routed(const void* gX, int64_t ldg, const float* z1, const float* dm, const int64_t* rank,
                                    const float* bounds, const int64_t* arg, float* dz1, float* partial, float* d_bounds,
                                    int64_t V, int dtype, void* stream) {
  SG_REQUIRE(V > 0 && ldg >= 3 && gX && z1 && bounds && arg && dz1 && partial && d_bounds, "sg_input_prep_bwd_routed: bad argument");
  SG_REQUIRE(dtype == SG_F32 || dtype == SG_BF16, "sg_input_prep_bwd_routed: unknown dtype %d", dtype);
  int rc = launch_input_prep_bwd(gX, ldg, z1, dm, rank, bounds, bounds + 3, dz1, partial, d_bounds, d_bounds + 3, V, dtype,
                                 (hipStream_t)stream);
  if (rc != SG_OK) return rc;
  return launch_bounds_route(d_bounds, d_bounds + 3, arg, dz1, (hipStream_t)stream);
}

SG_API int sg_mesh_loss_finalize(const float* partial, int64_t nb, float n_v, float n_f, float w_pos, float k1, float* out,
                                 void* stream) {
  SG_REQUIRE(nb > 0 && partial && out && n_v > 0.f && n_f >= 0.f, "sg_mesh_loss_finalize: bad argument");
  return launch_mesh_loss_finalize(partial, nb, n_v, n_f, w_pos, k1, out, (hipStream_t)stream);
}

SG_API int sg_multi_add(int64_t n, const float* const* srcs, const int64_t* src_ld, const int64_t* rows, const int64_t* cols,
                        float* const* dsts, void* stream) {
  SG_REQUIRE(n >= 0 && n <= kMultiAddMax, "sg_multi_add: at most %d matrices per call", kMultiAddMax);
  SG_REQUIRE(n == 0 || (srcs && src_ld && rows && cols && dsts), "sg_multi_add: bad argument");
  for (int64_t s = 0; s < n; ++s) {
    SG_REQUIRE(rows[s] >= 0 && cols[s] >= 0 && cols[s] <= INT32_MAX && (rows[s] * cols[s] == 0 || (srcs[s] && dsts[s])),
               "sg_multi_add: bad matrix %lld", (long long)s);
    SG_REQUIRE(rows[s] <= 1 || src_ld[s] >= cols[s], "sg_multi_add: source row stride below the row length");
  }
  return launch_multi_add((int)n, srcs, src_ld, rows, cols, dsts, (hipStream_t)stream);
}

SG_API int sg_bn_finalize_ranks(const float* all, int64_t world, int64_t C, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, float* out,
                                float* out_n, int64_t* batches_tracked, void* stream) {
  SG_REQUIRE(world > 0 && C >= 0 && all && gamma && beta && out && out_n, "sg_bn_finalize_ranks: bad argument");
  SG_REQUIRE((running_mean == nullptr) == (running_var == nullptr),
             "sg_bn_finalize_ranks: give both running buffers or none");
  return launch_bn_finalize_ranks(all, world, C, gamma, beta, running_mean, running_var, momentum, eps, out, out_n,
                                  batches_tracked, (hipStream_t)stream);
}

SG_API int sg_bn_finalize(const float* stats, double N, int64_t C, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, float* out,
                          void* stream) {
  SG_REQUIRE(N > 0 && C >= 0 && stats && gamma && beta && out, "sg_bn_finalize: bad argument");
  SG_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "sg_bn_finalize: give both running buffers or none");
  return launch_bn_finalize(stats, N, C, gamma, beta, running_mean, running_var, momentum, eps, out,
                            (hipStream_t)stream);
}

SG_API int sg_scale_shift_act(const void* X, int64_t ldx, const float* scale, const float* shift, float slope, void* Y,
                              int64_t ldy, int64_t V, int64_t C, int dtype, void* stream) {
  SG_REQUIRE(V >= 0 && C >= 0, "sg_scale_shift_act: negative size");
  if (V == 0 || C == 0) return SG_OK;
  SG_REQUIRE(X && Y && scale && shift && ldx >= C && ldy >= C, "sg_scale_shift_act: bad argument");
  return launch_col_apply(0, X, ldx, nullptr, 0, scale, shift, nullptr, nullptr, nullptr, nullptr, nullptr, slope, Y,
                          ldy, V, C, dtype, (hipStream_t)stream);
}

SG_API int sg_bn_act_bwd_reduce(const void* dA, int64_t ldda, const void* H, int64_t ldh, const float* scale,
                                const float* shift, const float* mean, const float* invstd, float slope, float* partial,
                                int64_t nb, int64_t V, int64_t C, int dtype, void* stream) {
  SG_REQUIRE(V >= 0 && C >= 0, "sg_bn_act_bwd_reduce: negative size");
  if (V == 0 || C == 0) return SG_OK;
  SG_REQUIRE(dA && H && scale && shift && mean && invstd && partial && ldda >= C && ldh >= C,
             "sg_bn_act_bwd_reduce: bad argument");
  return launch_col_reduce(1, dA, ldda, H, ldh, scale, shift, mean, invstd, slope, partial, nb, V, C, dtype,
                           (hipStream_t)stream);
}

SG_API int sg_bn_act_bwd_apply(const void* dA, int64_t ldda, const void* H, int64_t ldh, const float* scale,
                               const float* shift, const float* mean, const float* invstd, const float* k,
                               const float* c1, const float* c2, float slope, void* dH, int64_t lddh, int64_t V,
                               int64_t C, int dtype, void* stream) {
  SG_REQUIRE(V >= 0 && C >= 0, "sg_bn_act_bwd_apply: negative size");
  if (V == 0 || C == 0) return SG_OK;
  SG_REQUIRE(dA && H && dH && scale && shift && mean && invstd && k && c1 && c2 && ldda >= C && ldh >= C && lddh >= C,
             "sg_bn_act_bwd_apply: bad argument");
  return launch_col_apply(1, dA, ldda, H, ldh, scale, shift, mean, invstd, k, c1, c2, slope, dH, lddh, V, C, dtype,
                          (hipStream_t)stream);
}

SG_API int64_t sg_col_apply_blocks(int64_t V, int64_t C, int dtype) { return col_apply_blocks(V, C, dtype); }

SG_API int sg_bn_act_bwd_apply_colsum(const void* dA, int64_t ldda, const void* H, int64_t ldh, const float* scale,
                                      const float* shift, const float* mean, const float* invstd, const float* k,
                                      const float* c1, const float* c2, float slope, void* dH, int64_t lddh, int64_t V,
                                      int64_t C, int dtype, float* colsum_partial, float* colsum, void* stream) {
  SG_REQUIRE(V > 0 && C > 0, "sg_bn_act_bwd_apply_colsum: empty input");
  SG_REQUIRE(dA && H && dH && scale && shift && mean && invstd && k && c1 && c2 && colsum_partial && colsum &&
                 ldda >= C && ldh >= C && lddh >= C,
             "sg_bn_act_bwd_apply_colsum: bad argument");
  const int64_t nb = col_apply_blocks(V, C, dtype);
  if (nb == 0 || ldda % 4 || ldh % 4 || lddh % 4) {
    set_error("sg_bn_act_bwd_apply_colsum: shape not served (C must fill whole 16-byte vectors, <= 256 of them)");
    return SG_ERR_UNSUPPORTED;
  }
  int rc = launch_col_apply(1, dA, ldda, H, ldh, scale, shift, mean, invstd, k, c1, c2, slope, dH, lddh, V, C, dtype,
                            (hipStream_t)stream, colsum_partial);
  if (rc != SG_OK) return rc;
  return launch_colsum_finalize(colsum_partial, nb, C, colsum, (hipStream_t)stream);
}

SG_API int sg_gather_rows(const int32_t* rows, int64_t n, const void* X, int64_t ldx, void* Y, int64_t ldy,
                   int64_t C, int dtype, void* stream) {
  SG_REQUIRE(n >= 0 && C >= 0, "sg_gather_rows: negative size");
  if (n == 0 || C == 0) return SG_OK;
  SG_REQUIRE(rows && X && Y, "sg_gather_rows: null pointer");
  SG_REQUIRE(ldx >= C && ldy >= C, "sg_gather_rows: stride < C");
  return launch_gather_rows(rows, n, X, ldx, Y, ldy, C, dtype, (hipStream_t)stream);
}

}  // extern "C"
