// The rank's collectives BELOW the C ABI (`sg_halo_exchange`; section 8(e): the reference is
// single-device, util/networks.py:83-101 has no exchange step -- this is what vertex partitioning adds to it).
//
// A partitioned SGCN rank runs its 13 blocks phase by phase (sg_block_run, block.hip) with one collective between two
// phases: forward one all-to-all per block (rows of the conv output + the rank's BatchNorm statistics in the pad rows),
// backward one all-reduce (the two BatchNorm sums) and one all-to-all (gradient rows) per block -- 40 of the iteration's 44
// collectives.  Issued from Python through c10d each of them costs the host ~35 us (a Work object, a stream event pair,
// split lists turned into vectors); here the library enqueues them itself, on the SAME stream as its kernels (no event
// hand-off between two streams), through a communicator of its own:
//   * RCCL is bound at run time from the copy the process already uses (PyTorch ships one; /opt/rocm/lib otherwise), like
//     hipBLASLt in dense.hip: no link-time dependency, and a process that never creates a communicator never touches it;
//   * the communicator is created from a 128-byte unique id that rank 0 draws (sg_comm_unique_id) and the host hands to the
//     other ranks through whatever it has (torch.distributed's store);
//   * sg_halo_exchange = ncclGroupStart, one ncclSend + one ncclRecv per peer with rows to move, ncclGroupEnd: the all-to-all
//     with per-peer row counts (fixed per partition: given once, at sg_comm_create); a receive lands in place -- rows
//     [n_own:] of the [n_ext, C] buffer ARE the receive buffer (dist.FoldedLayout);
//   * sg_part_run walks a schedule of {run these blocks' phases, exchange, all-reduce, all-gather}: a rank's whole forward
//     pass, or its whole backward pass, is ONE foreign call.
// The c10d path (dist.py) stays: it is what the gloo self-tests exercise and what the supervisor falls back to.
#include <dlfcn.h>
#include <string.h>

#include <mutex>
#include <vector>

#include <rccl/rccl.h>

#include "sg_common.h"

struct sg_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  std::vector<int64_t> send_rows, recv_rows;      // per peer, rows of one exchange (pad rows included)
};

namespace sg {
namespace {

struct RcclApi {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  bool ok = false;
  std::string why;
};

std::mutex g_rccl_mu;
RcclApi g_rccl;
bool g_rccl_tried = false;

template <class F>
bool bind(void* so, const char* name, F* out) {
  *out = (F)dlsym(so, name);
  return *out != nullptr;
}

const RcclApi& rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (g_rccl_tried) return g_rccl;
  g_rccl_tried = true;
  void* so = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);           // the copy the process already uses (PyTorch's)
  if (!so) so = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!so) so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!so) so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!so) {
    g_rccl.why = std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "?");
    return g_rccl;
  }
  bool ok = bind(so, "ncclGetUniqueId", &g_rccl.GetUniqueId) && bind(so, "ncclCommInitRank", &g_rccl.CommInitRank) &&
            bind(so, "ncclCommDestroy", &g_rccl.CommDestroy) && bind(so, "ncclGroupStart", &g_rccl.GroupStart) &&
            bind(so, "ncclGroupEnd", &g_rccl.GroupEnd) && bind(so, "ncclSend", &g_rccl.Send) && bind(so, "ncclRecv", &g_rccl.Recv) &&
            bind(so, "ncclAllReduce", &g_rccl.AllReduce) && bind(so, "ncclAllGather", &g_rccl.AllGather) &&
            bind(so, "ncclGetErrorString", &g_rccl.GetErrorString) && bind(so, "ncclGetVersion", &g_rccl.GetVersion);
  if (!ok) g_rccl.why = "librccl.so lacks an expected entry point";
  g_rccl.ok = ok;
  return g_rccl;
}

#define SG_NCCL_TRY(expr)                                                                                 \
  do {                                                                                                    \
    ncclResult_t r__ = (expr);                                                                            \
    if (r__ != ncclSuccess) {                                                                             \
      set_error("%s failed: %s (%s:%d)", #expr, api.GetErrorString(r__), __FILE__, __LINE__);             \
      return SG_ERR_HIP;                                                                                  \
    }                                                                                                     \
  } while (0)

int need_api(const RcclApi** out) {
  const RcclApi& api = rccl();
  if (!api.ok) {
    set_error("the collectives below the C ABI need RCCL, but %s", api.why.c_str());
    return SG_ERR_UNSUPPORTED;
  }
  *out = &api;
  return SG_OK;
}

int exchange(sg_comm* c, const void* send, void* recv, int64_t row_bytes, hipStream_t stream) {
  const RcclApi* ap;
  int rc = need_api(&ap);
  if (rc != SG_OK) return rc;
  const RcclApi& api = *ap;
  SG_NCCL_TRY(api.GroupStart());
  int64_t so = 0, ro = 0;
  for (int q = 0; q < c->world; ++q) {
    const int64_t sb = c->send_rows[q] * row_bytes, rb = c->recv_rows[q] * row_bytes;
    // (rows for the rank itself are legal -- a send and a receive to one's own rank inside a group is a copy -- and let a
    //  ONE-rank communicator drive this whole path on a single GPU: tests/test_gpu_scale.py)
    if (sb) SG_NCCL_TRY(api.Send((const char*)send + so, (size_t)sb, ncclInt8, q, c->comm, stream));
    if (rb) SG_NCCL_TRY(api.Recv((char*)recv + ro, (size_t)rb, ncclInt8, q, c->comm, stream));
    so += sb;
    ro += rb;
  }
  SG_NCCL_TRY(api.GroupEnd());
  return SG_OK;
}

}  // namespace
}  // namespace sg

using namespace sg;

extern "C" {

SG_API int sg_comm_available(void) { return rccl().ok ? 1 : 0; }

SG_API int sg_comm_unique_id(void* id128) {
  SG_REQUIRE(id128 != nullptr, "sg_comm_unique_id: null buffer");
  const RcclApi* ap;
  int rc = need_api(&ap);
  if (rc != SG_OK) return rc;
  const RcclApi& api = *ap;
  static_assert(sizeof(ncclUniqueId) == 128, "the id travels as 128 bytes");
  SG_NCCL_TRY(api.GetUniqueId((ncclUniqueId*)id128));
  return SG_OK;
}

SG_API int sg_comm_create(const void* id128, int rank, int world, const int64_t* send_rows, const int64_t* recv_rows,
                          sg_comm** out) {
  SG_REQUIRE(out != nullptr, "sg_comm_create: null output");
  *out = nullptr;
  SG_REQUIRE(id128 && world >= 1 && rank >= 0 && rank < world, "sg_comm_create: bad rank %d of %d, or no id", rank, world);
  SG_REQUIRE(send_rows && recv_rows, "sg_comm_create: per-peer row counts missing");
  for (int q = 0; q < world; ++q)
    SG_REQUIRE(send_rows[q] >= 0 && recv_rows[q] >= 0 && (q != rank || send_rows[q] == recv_rows[q]),
               "sg_comm_create: bad row counts for peer %d", q);
  const RcclApi* ap;
  int rc = need_api(&ap);
  if (rc != SG_OK) return rc;
  const RcclApi& api = *ap;
  sg_comm* c = new sg_comm;
  c->rank = rank;
  c->world = world;
  c->send_rows.assign(world, 0);
  c->recv_rows.assign(world, 0);
  for (int q = 0; q < world; ++q) {
    c->send_rows[q] = send_rows[q];
    c->recv_rows[q] = recv_rows[q];
  }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclResult_t r = api.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, api.GetErrorString(r));
    delete c;
    return SG_ERR_HIP;
  }
  *out = c;
  return SG_OK;
}

SG_API int sg_comm_destroy(sg_comm* c) {
  if (!c) return SG_OK;
  const RcclApi& api = rccl();
  if (api.ok && c->comm) api.CommDestroy(c->comm);
  delete c;
  return SG_OK;
}

SG_API int sg_halo_exchange(sg_comm* c, const void* send, void* recv, int64_t row_bytes, void* stream) {
  SG_REQUIRE(c != nullptr, "sg_halo_exchange: null communicator");
  SG_REQUIRE(row_bytes > 0, "sg_halo_exchange: row_bytes = %lld", (long long)row_bytes);
  int64_t rows = 0;
  for (int q = 0; q < c->world; ++q) rows += c->send_rows[q] + c->recv_rows[q];
  if (rows == 0) return SG_OK;                            // (a one-rank partition: nothing to move)
  SG_REQUIRE(send && recv, "sg_halo_exchange: null buffer");
  return exchange(c, send, recv, row_bytes, (hipStream_t)stream);
}

SG_API int sg_comm_all_reduce_f32(sg_comm* c, float* buf, int64_t n, void* stream) {
  SG_REQUIRE(c != nullptr && n >= 0 && (n == 0 || buf), "sg_comm_all_reduce_f32: bad argument");
  if (n == 0) return SG_OK;
  const RcclApi* ap;
  int rc = need_api(&ap);
  if (rc != SG_OK) return rc;
  const RcclApi& api = *ap;
  SG_NCCL_TRY(api.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream));
  return SG_OK;
}

SG_API int sg_comm_all_gather(sg_comm* c, const void* in, void* out, int64_t bytes_per_rank, void* stream) {
  SG_REQUIRE(c != nullptr && bytes_per_rank >= 0 && (bytes_per_rank == 0 || (in && out)), "sg_comm_all_gather: bad argument");
  if (bytes_per_rank == 0) return SG_OK;
  const RcclApi* ap;
  int rc = need_api(&ap);
  if (rc != SG_OK) return rc;
  const RcclApi& api = *ap;
  SG_NCCL_TRY(api.AllGather(in, out, (size_t)bytes_per_rank, ncclInt8, c->comm, (hipStream_t)stream));
  return SG_OK;
}

SG_API int64_t sg_part_step_sizeof(void) { return (int64_t)sizeof(sg_part_step); }

SG_API int sg_part_run(sg_comm* c, const sg_part_step* steps, int64_t n, void* stream) {
  SG_REQUIRE(n >= 0 && (n == 0 || steps != nullptr), "sg_part_run: bad argument");
  for (int64_t i = 0; i < n; ++i) {
    const sg_part_step& s = steps[i];
    int rc = SG_OK;
    switch (s.kind) {
      case SG_STEP_BLOCKS:
        rc = sg_block_run(s.blocks, s.n, stream);
        break;
      case SG_STEP_EXCHANGE:
        SG_REQUIRE(c != nullptr, "sg_part_run: step %lld is a collective but there is no communicator", (long long)i);
        rc = sg_halo_exchange(c, s.send, s.recv, s.n, stream);
        break;
      case SG_STEP_ALL_REDUCE:
        SG_REQUIRE(c != nullptr, "sg_part_run: step %lld is a collective but there is no communicator", (long long)i);
        rc = sg_comm_all_reduce_f32(c, (float*)s.recv, s.n, stream);
        break;
      case SG_STEP_ALL_GATHER:
        SG_REQUIRE(c != nullptr, "sg_part_run: step %lld is a collective but there is no communicator", (long long)i);
        rc = sg_comm_all_gather(c, s.send, s.recv, s.n, stream);
        break;
      default:
        set_error("sg_part_run: step %lld has kind %d", (long long)i, s.kind);
        return SG_ERR_INVALID;
    }
    if (rc != SG_OK) return rc;
  }
  return SG_OK;
}

}  // extern "C"
