// sg_trace_*: optional HIP-event timing of the launches the library makes (semigcn.h).  A benchmarking aid for bench.py's
// roofline figures now that a whole [ChebConv -> BatchNorm -> activation] block is one foreign call: the event pairs are
// recorded on the launching stream around each aggregation / dense product, exactly where the Python-side timer recorded
// them when every launch was a call of its own.
#include <mutex>
#include <vector>

#include <atomic>

#include "sg_common.h"

namespace sg {

std::atomic<bool> g_trace_on{false};      // read on the autograd thread, written on the caller's

namespace {
struct Rec {
  sg_trace_record r;
  hipEvent_t e0 = nullptr, e1 = nullptr;
};
std::mutex g_mu;
std::vector<Rec> g_recs;
int64_t g_capacity = 0;
int g_kinds = 7;
}  // namespace

void trace_open(int kind, int dtype, int engine, int64_t a, int64_t b, int64_t c, hipStream_t stream, int64_t* slot) {
  std::lock_guard<std::mutex> lock(g_mu);
  *slot = -1;
  if (!g_trace_on || !((g_kinds >> kind) & 1) || (int64_t)g_recs.size() >= g_capacity) return;
  Rec rec;
  rec.r = sg_trace_record{kind, dtype, engine, 0, a, b, c, 0.f, 0.f};
  if (hipEventCreate(&rec.e0) != hipSuccess || hipEventCreate(&rec.e1) != hipSuccess) {
    if (rec.e0) (void)hipEventDestroy(rec.e0);
    (void)hipGetLastError();
    return;
  }
  (void)hipEventRecord(rec.e0, stream);
  *slot = (int64_t)g_recs.size();
  g_recs.push_back(rec);
}

void trace_close(int64_t slot, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (slot < 0 || slot >= (int64_t)g_recs.size()) return;
  (void)hipEventRecord(g_recs[slot].e1, stream);
}

}  // namespace sg

using namespace sg;

extern "C" {

SG_API int sg_trace_begin(int64_t capacity, int kinds) {
  SG_REQUIRE(capacity >= 0 && capacity <= (1 << 24), "sg_trace_begin: capacity out of range");
  std::lock_guard<std::mutex> lock(g_mu);
  for (Rec& r : g_recs) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_recs.clear();
  g_recs.reserve((size_t)capacity);
  g_capacity = capacity;
  g_kinds = kinds;
  g_trace_on = capacity > 0;
  return SG_OK;
}

SG_API int64_t sg_trace_read(sg_trace_record* out, int64_t n) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (n < 0 || (n > 0 && !out)) {
    set_error("sg_trace_read: bad argument");
    return SG_ERR_INVALID;
  }
  int64_t k = 0;
  for (Rec& r : g_recs) {
    if (k >= n) break;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) {
      (void)hipGetLastError();
      ms = -1.f;          // not complete yet (the caller did not synchronise), or recorded under a stream capture
    }
    out[k] = r.r;
    out[k].ms = ms;
    ++k;
  }
  return k;
}

SG_API int sg_trace_end(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_trace_on = false;
  for (Rec& r : g_recs) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_recs.clear();
  g_capacity = 0;
  return SG_OK;
}

}  // extern "C"
