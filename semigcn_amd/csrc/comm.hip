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

#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <vector>

#include <rccl/rccl.h>

#include "sg_common.h"

// One RCCL communicator per process group, shared by every sg_comm made from it (sg_comm_share): a partitioned MGCN has one
// exchange layout per level and used to pay an ncclCommInitRank -- and its buffers -- for each of them.
struct sg_comm_core {
  ncclComm_t comm = nullptr;
  ~sg_comm_core();
};

struct sg_comm {
  std::shared_ptr<sg_comm_core> core;
  ncclComm_t comm = nullptr;                       // == core->comm
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

// ---- an in-process stand-in for RCCL (sg_comm_test_stub): TEST INFRASTRUCTURE ---------------------------------------------
// No box available to this build has two GPUs, so the one thing a one-rank communicator cannot show -- a wrong peer offset in
// exchange() -- is checked on the CPU: with the stub installed, every "rank" is a communicator of THIS process, its buffers
// are host memory, and the calls of all ranks (made one after the other from one thread) meet in a mailbox: a send from a to
// b is copied into the matching receive of b from a as soon as both have been posted (FIFO per ordered pair, as RCCL matches
// them inside a group), an all-reduce / all-gather completes when the last rank has called.  Every call is logged
// (sg_comm_test_log) so that a test can also compare (pointer, bytes, peer) with the split lists of dist.FoldedLayout.
struct StubComm {
  int rank, world, group;
  int64_t coll_seq = 0;
};
struct StubLogRec {
  int64_t kind, rank, peer, ptr, bytes;             // kind 0 send, 1 recv, 2 all-reduce, 3 all-gather, 4 group start, 5 group end
};
struct StubColl {
  std::vector<std::tuple<int, const void*, void*, size_t>> parts;      // (rank, in, out, count)
};
struct StubState {
  std::mutex mu;
  bool on = false;
  int next_group = 1;
  int fail_send_after = -1;                          // test hook: the n-th ncclSend from now fails (ncclInternalError)
  int64_t mismatched = 0;
  std::map<std::tuple<int, int, int>, std::deque<std::pair<const void*, size_t>>> sends;    // (group, from, to)
  std::map<std::tuple<int, int, int>, std::deque<std::pair<void*, size_t>>> recvs;
  std::map<std::tuple<int, int64_t, int>, StubColl> colls;                                  // (group, seq, kind)
  std::vector<StubLogRec> log;
  int open_groups = 0;
};
StubState g_stub;

void stub_match(int group, int from, int to) {
  auto& sq = g_stub.sends[{group, from, to}];
  auto& rq = g_stub.recvs[{group, from, to}];
  while (!sq.empty() && !rq.empty()) {
    if (sq.front().second != rq.front().second) ++g_stub.mismatched;
    memcpy(rq.front().first, sq.front().first, std::min(sq.front().second, rq.front().second));
    sq.pop_front();
    rq.pop_front();
  }
}
size_t stub_elem(ncclDataType_t t) { return t == ncclFloat32 ? 4 : 1; }
ncclResult_t stub_GetUniqueId(ncclUniqueId* id) {
  std::lock_guard<std::mutex> lock(g_stub.mu);
  memset(id, 0, sizeof(*id));
  const int grp = g_stub.next_group++;
  memcpy(id, &grp, sizeof(grp));
  return ncclSuccess;
}
ncclResult_t stub_CommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
  int grp = 0;
  memcpy(&grp, &id, sizeof(grp));
  *out = (ncclComm_t) new StubComm{rank, world, grp};
  return ncclSuccess;
}
ncclResult_t stub_CommDestroy(ncclComm_t c) {
  delete (StubComm*)c;
  return ncclSuccess;
}
ncclResult_t stub_GroupStart() {
  std::lock_guard<std::mutex> lock(g_stub.mu);
  ++g_stub.open_groups;
  g_stub.log.push_back({4, -1, -1, 0, 0});
  return ncclSuccess;
}
ncclResult_t stub_GroupEnd() {
  std::lock_guard<std::mutex> lock(g_stub.mu);
  --g_stub.open_groups;
  g_stub.log.push_back({5, -1, -1, 0, 0});
  return ncclSuccess;
}
ncclResult_t stub_Send(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t) {
  StubComm* c = (StubComm*)comm;
  std::lock_guard<std::mutex> lock(g_stub.mu);
  if (g_stub.fail_send_after >= 0 && g_stub.fail_send_after-- == 0) return ncclInternalError;
  g_stub.log.push_back({0, c->rank, peer, (int64_t)(uintptr_t)buf, (int64_t)(count * stub_elem(t))});
  g_stub.sends[{c->group, c->rank, peer}].push_back({buf, count * stub_elem(t)});
  stub_match(c->group, c->rank, peer);
  return ncclSuccess;
}
ncclResult_t stub_Recv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t) {
  StubComm* c = (StubComm*)comm;
  std::lock_guard<std::mutex> lock(g_stub.mu);
  g_stub.log.push_back({1, c->rank, peer, (int64_t)(uintptr_t)buf, (int64_t)(count * stub_elem(t))});
  g_stub.recvs[{c->group, peer, c->rank}].push_back({buf, count * stub_elem(t)});
  stub_match(c->group, peer, c->rank);
  return ncclSuccess;
}
ncclResult_t stub_AllReduce(const void* in, void* out, size_t count, ncclDataType_t t, ncclRedOp_t, ncclComm_t comm, hipStream_t) {
  StubComm* c = (StubComm*)comm;
  std::lock_guard<std::mutex> lock(g_stub.mu);
  if (t != ncclFloat32) return ncclInvalidArgument;
  g_stub.log.push_back({2, c->rank, -1, (int64_t)(uintptr_t)out, (int64_t)(count * 4)});
  StubColl& k = g_stub.colls[{c->group, c->coll_seq++, 2}];
  k.parts.emplace_back(c->rank, in, out, count);
  if ((int)k.parts.size() == c->world) {
    std::vector<double> sum(count, 0.0);
    for (auto& p : k.parts) {
      if (std::get<3>(p) != count) ++g_stub.mismatched;
      for (size_t i = 0; i < std::min(count, std::get<3>(p)); ++i) sum[i] += ((const float*)std::get<1>(p))[i];
    }
    for (auto& p : k.parts)
      for (size_t i = 0; i < std::min(count, std::get<3>(p)); ++i) ((float*)std::get<2>(p))[i] = (float)sum[i];
    k.parts.clear();
  }
  return ncclSuccess;
}
ncclResult_t stub_AllGather(const void* in, void* out, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t) {
  StubComm* c = (StubComm*)comm;
  std::lock_guard<std::mutex> lock(g_stub.mu);
  const size_t bytes = count * stub_elem(t);
  g_stub.log.push_back({3, c->rank, -1, (int64_t)(uintptr_t)out, (int64_t)bytes});
  StubColl& k = g_stub.colls[{c->group, c->coll_seq++, 3}];
  k.parts.emplace_back(c->rank, in, out, bytes);
  if ((int)k.parts.size() == c->world) {
    std::vector<std::vector<char>> ins(c->world);
    for (auto& p : k.parts) {
      if (std::get<3>(p) != bytes) ++g_stub.mismatched;
      ins[std::get<0>(p)].assign((const char*)std::get<1>(p), (const char*)std::get<1>(p) + bytes);
    }
    for (auto& p : k.parts)
      for (int q = 0; q < c->world; ++q) memcpy((char*)std::get<2>(p) + q * bytes, ins[q].data(), bytes);
    k.parts.clear();
  }
  return ncclSuccess;
}
const char* stub_GetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "stub: injected / invalid"; }
ncclResult_t stub_GetVersion(int* v) {
  *v = 0;
  return ncclSuccess;
}
RcclApi g_stub_api;

template <class F>
bool bind(void* so, const char* name, F* out) {
  *out = (F)dlsym(so, name);
  return *out != nullptr;
}

const RcclApi& rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (g_stub.on) return g_stub_api;
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
  // An error between GroupStart and GroupEnd must not leave the calling thread's group open: every later RCCL call of the
  // thread (torch.distributed's fallback collectives included) would be queued into it and hang instead of failing.
#define SG_NCCL_TRY_IN_GROUP(expr)                                                                        \
  do {                                                                                                    \
    ncclResult_t r__ = (expr);                                                                            \
    if (r__ != ncclSuccess) {                                                                             \
      (void)api.GroupEnd();                                                                               \
      set_error("%s failed for peer %d of %d: %s (%s:%d); the group was closed", #expr, q, c->world,      \
                api.GetErrorString(r__), __FILE__, __LINE__);                                             \
      return SG_ERR_HIP;                                                                                  \
    }                                                                                                     \
  } while (0)
  for (int q = 0; q < c->world; ++q) {
    const int64_t sb = c->send_rows[q] * row_bytes, rb = c->recv_rows[q] * row_bytes;
    // (rows for the rank itself are legal -- a send and a receive to one's own rank inside a group is a copy -- and let a
    //  ONE-rank communicator drive this whole path on a single GPU: tests/test_gpu_scale.py)
    if (sb) SG_NCCL_TRY_IN_GROUP(api.Send((const char*)send + so, (size_t)sb, ncclInt8, q, c->comm, stream));
    if (rb) SG_NCCL_TRY_IN_GROUP(api.Recv((char*)recv + ro, (size_t)rb, ncclInt8, q, c->comm, stream));
    so += sb;
    ro += rb;
  }
#undef SG_NCCL_TRY_IN_GROUP
  SG_NCCL_TRY(api.GroupEnd());
  return SG_OK;
}

}  // namespace
}  // namespace sg

sg_comm_core::~sg_comm_core() {
  const sg::RcclApi& api = sg::rccl();
  if (api.ok && comm) api.CommDestroy(comm);
}

using namespace sg;

namespace {
thread_local int64_t t_failed_step = -1;
}

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
  c->core = std::make_shared<sg_comm_core>();
  ncclResult_t r = api.CommInitRank(&c->core->comm, world, id, rank);
  if (r != ncclSuccess) {
    set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, api.GetErrorString(r));
    c->core->comm = nullptr;
    delete c;
    return SG_ERR_HIP;
  }
  c->comm = c->core->comm;
  *out = c;
  return SG_OK;
}

SG_API int sg_comm_share(const sg_comm* base, const int64_t* send_rows, const int64_t* recv_rows, sg_comm** out) {
  SG_REQUIRE(out != nullptr, "sg_comm_share: null output");
  *out = nullptr;
  SG_REQUIRE(base != nullptr && base->core && send_rows && recv_rows, "sg_comm_share: null argument");
  for (int q = 0; q < base->world; ++q)
    SG_REQUIRE(send_rows[q] >= 0 && recv_rows[q] >= 0 && (q != base->rank || send_rows[q] == recv_rows[q]),
               "sg_comm_share: bad row counts for peer %d", q);
  sg_comm* c = new sg_comm;
  c->core = base->core;                                  // the RCCL communicator lives as long as its last user
  c->comm = base->comm;
  c->rank = base->rank;
  c->world = base->world;
  c->send_rows.assign(send_rows, send_rows + base->world);
  c->recv_rows.assign(recv_rows, recv_rows + base->world);
  *out = c;
  return SG_OK;
}

SG_API int sg_comm_destroy(sg_comm* c) {
  delete c;                                              // (the shared core destroys the RCCL communicator with its last user)
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

SG_API int64_t sg_part_failed_step(void) { return t_failed_step; }

SG_API int sg_part_run(sg_comm* c, const sg_part_step* steps, int64_t n, void* stream) {
  t_failed_step = -1;
  SG_REQUIRE(n >= 0 && (n == 0 || steps != nullptr), "sg_part_run: bad argument");
  for (int64_t i = 0; i < n; ++i) {
    t_failed_step = i;                                    // (the SG_REQUIREs below return from inside the loop)
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
    if (rc != SG_OK) {
      // steps [0, i) are enqueued and the peers will enqueue theirs: the collective sequence of this rank is now out of step
      // with the group's -- the host must abort the job, not retry on another communicator (sg_part_failed_step() = i)
      const std::string why = sg_last_error();
      set_error("sg_part_run: step %lld of %lld (kind %d) failed after %lld steps were enqueued: %s", (long long)i, (long long)n,
                s.kind, (long long)i, why.c_str());
      return rc;
    }
  }
  t_failed_step = -1;
  return SG_OK;
}

// ---- test infrastructure: the in-process RCCL stand-in (see StubState) ---------------------------------------------------------
SG_API int sg_comm_test_stub(int on) {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  std::lock_guard<std::mutex> lock2(g_stub.mu);
  if (on) {
    g_stub_api.GetUniqueId = stub_GetUniqueId;
    g_stub_api.CommInitRank = stub_CommInitRank;
    g_stub_api.CommDestroy = stub_CommDestroy;
    g_stub_api.GroupStart = stub_GroupStart;
    g_stub_api.GroupEnd = stub_GroupEnd;
    g_stub_api.Send = stub_Send;
    g_stub_api.Recv = stub_Recv;
    g_stub_api.AllReduce = stub_AllReduce;
    g_stub_api.AllGather = stub_AllGather;
    g_stub_api.GetErrorString = stub_GetErrorString;
    g_stub_api.GetVersion = stub_GetVersion;
    g_stub_api.ok = true;
  }
  g_stub.on = on != 0;
  g_stub.sends.clear();
  g_stub.recvs.clear();
  g_stub.colls.clear();
  g_stub.log.clear();
  g_stub.mismatched = 0;
  g_stub.open_groups = 0;
  g_stub.fail_send_after = -1;
  return SG_OK;
}

SG_API int sg_comm_test_fail_send(int nth) {
  std::lock_guard<std::mutex> lock(g_stub.mu);
  g_stub.fail_send_after = nth;
  return SG_OK;
}

// out: n records of five int64 {kind, rank, peer, pointer, bytes}; returns the number of records logged since the stub was
// installed (copies min(n, that)); state[0] = unmatched sends + receives + incomplete collectives, state[1] = size mismatches,
// state[2] = groups left open
SG_API int64_t sg_comm_test_log(int64_t* out, int64_t n, int64_t* state) {
  std::lock_guard<std::mutex> lock(g_stub.mu);
  const int64_t have = (int64_t)g_stub.log.size();
  for (int64_t i = 0; i < have && i < n && out; ++i) {
    const StubLogRec& r = g_stub.log[i];
    out[5 * i + 0] = r.kind; out[5 * i + 1] = r.rank; out[5 * i + 2] = r.peer; out[5 * i + 3] = r.ptr; out[5 * i + 4] = r.bytes;
  }
  if (state) {
    int64_t open = 0;
    for (auto& kv : g_stub.sends) open += (int64_t)kv.second.size();
    for (auto& kv : g_stub.recvs) open += (int64_t)kv.second.size();
    for (auto& kv : g_stub.colls) open += (int64_t)kv.second.parts.size();
    state[0] = open;
    state[1] = g_stub.mismatched;
    state[2] = g_stub.open_groups;
  }
  return have;
}

}  // extern "C"
