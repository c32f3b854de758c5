// Dense per-vertex products behind ONE dispatcher, for the per-block entry points (block.hip).
//
// The K `lins[k]` calls of ChebConv.forward [3P torch_geometric 2.2.0] (call sites util/networks.py:42,49;
// util/meshnet.py:40-240) and their autograd are three product shapes on [V, C] row-major features:
//     nt  C[M, N]  = A[M, K] B[N, K]^T (+ bias)     forward, and the input gradient with the transposed weights
//     nn  C[M, N]  = A[M, K] B[K, N]                the input gradient where no transposed copy is kept
//     tn  C[N, Kp] = A[M, N]^T B[M, Kp]             the weight gradient: a reduction over all M vertices
// Engines, in the order they are tried (the rule functional.py applied call by call before the blocks moved below the C ABI):
//   * the thin-product kernels (thin_gemm.hip) for weight matrices of at most 16 x 16;
//   * the library's MFMA kernels (gemm_mfma.hip / gemm_mfma256.hip) for bf16 operands whose shape they take;
//   * for fp32 features -- the reference's own precision -- the split-bf16 MFMA kernels (gemm_split.hip: every operand as
//     three bf16 pieces, six piece products, fp32-equivalent error) wherever they take the shape;
//   * the BLAS library (hipBLASLt) for everything else: fp32 shapes the split kernels do not take and bf16 shapes that are
//     no multiple of an MFMA step.  hipBLASLt is bound at run time from the copy that
//     is already loaded into the process (PyTorch ships one; /opt/rocm/lib otherwise): no link-time dependency.
#include <dlfcn.h>

#include <map>
#include <mutex>
#include <tuple>

#include <hipblaslt/hipblaslt.h>

#include "sg_common.h"

namespace sg {
namespace {

// ---- hipBLASLt, bound lazily -----------------------------------------------------------------------------------------------
struct LtApi {
  decltype(&hipblasLtCreate) Create = nullptr;
  decltype(&hipblasLtMatmulDescCreate) DescCreate = nullptr;
  decltype(&hipblasLtMatmulDescSetAttribute) DescSet = nullptr;
  decltype(&hipblasLtMatrixLayoutCreate) LayoutCreate = nullptr;
  decltype(&hipblasLtMatrixLayoutSetAttribute) LayoutSet = nullptr;
  decltype(&hipblasLtMatmulPreferenceCreate) PrefCreate = nullptr;
  decltype(&hipblasLtMatmulPreferenceSetAttribute) PrefSet = nullptr;
  decltype(&hipblasLtMatmulPreferenceDestroy) PrefDestroy = nullptr;
  decltype(&hipblasLtMatmulAlgoGetHeuristic) Heuristic = nullptr;
  decltype(&hipblasLtMatmul) Matmul = nullptr;
  decltype(&hipblasLtGetVersion) GetVersion = nullptr;      // (optional: absent from very old builds)
  decltype(&hipblasLtDestroy) Destroy = nullptr;            // (optional: only used to drop a handle of a refused library)
  bool ok = false;
  std::string why;
};

std::mutex g_lt_mu;          // guards the binding, the handles and the plan cache (forward and backward run on different threads)
LtApi g_lt;
bool g_lt_tried = false;
constexpr int kMaxDevices = 16;
hipblasLtHandle_t g_lt_handle[kMaxDevices] = {};

template <class F>
bool bind(void* so, const char* name, F* out) {
  *out = (F)dlsym(so, name);
  return *out != nullptr;
}

const LtApi& lt_api() {     // call with g_lt_mu held
  if (g_lt_tried) return g_lt;
  g_lt_tried = true;
  void* so = dlopen("libhipblaslt.so.1", RTLD_NOW | RTLD_NOLOAD);       // the copy the process already uses (PyTorch's)
  if (!so) so = dlopen("libhipblaslt.so", RTLD_NOW | RTLD_NOLOAD);
  if (!so) so = dlopen("libhipblaslt.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!so) so = dlopen("/opt/rocm/lib/libhipblaslt.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!so) {
    g_lt.why = std::string("cannot load libhipblaslt.so.1: ") + (dlerror() ? dlerror() : "?");
    return g_lt;
  }
  bool ok = bind(so, "hipblasLtCreate", &g_lt.Create) && bind(so, "hipblasLtMatmulDescCreate", &g_lt.DescCreate) &&
            bind(so, "hipblasLtMatmulDescSetAttribute", &g_lt.DescSet) &&
            bind(so, "hipblasLtMatrixLayoutCreate", &g_lt.LayoutCreate) &&
            bind(so, "hipblasLtMatrixLayoutSetAttribute", &g_lt.LayoutSet) &&
            bind(so, "hipblasLtMatmulPreferenceCreate", &g_lt.PrefCreate) &&
            bind(so, "hipblasLtMatmulPreferenceSetAttribute", &g_lt.PrefSet) &&
            bind(so, "hipblasLtMatmulPreferenceDestroy", &g_lt.PrefDestroy) &&
            bind(so, "hipblasLtMatmulAlgoGetHeuristic", &g_lt.Heuristic) && bind(so, "hipblasLtMatmul", &g_lt.Matmul);
  if (!ok) g_lt.why = "libhipblaslt.so.1 lacks an expected entry point";
  bind(so, "hipblasLtGetVersion", &g_lt.GetVersion);
  bind(so, "hipblasLtDestroy", &g_lt.Destroy);
  g_lt.ok = ok;
  return g_lt;
}

// One cached plan per product signature: descriptor, layouts and the heuristic's first algorithm.
struct PlanKey {
  int dev, opA, opB, dt_in, dt_out, bias, batch;
  int64_t M, N, K, lda, ldb, ldc, sa, sb, sc;
  bool operator<(const PlanKey& o) const {
    return std::tie(dev, opA, opB, dt_in, dt_out, bias, batch, M, N, K, lda, ldb, ldc, sa, sb, sc) <
           std::tie(o.dev, o.opA, o.opB, o.dt_in, o.dt_out, o.bias, o.batch, o.M, o.N, o.K, o.lda, o.ldb, o.ldc, o.sa, o.sb, o.sc);
  }
};
struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t ws = 0;
};
std::map<PlanKey, Plan> g_plans;

#define SG_LT_TRY(expr)                                                            \
  do {                                                                             \
    hipblasStatus_t s__ = (expr);                                                  \
    if (s__ != HIPBLAS_STATUS_SUCCESS) {                                           \
      set_error("%s failed with hipblasStatus %d (%s:%d)", #expr, (int)s__, __FILE__, __LINE__); \
      return SG_ERR_HIP;                                                           \
    }                                                                              \
  } while (0)

hipDataType lt_type(int dt) { return dt == SG_BF16 ? HIP_R_16BF : HIP_R_32F; }

// Row-major C[M, N] (ldc) = op(A) op(B) (+ bias[N]); op(A) is M x K: A stored [M, K] (opA = 0) or [K, M] (opA = 1); op(B) is
// K x N: B stored [K, N] (opB = 0) or [N, K] (opB = 1).  Column-major view handed to the library: C^T = op(B)^T op(A)^T, i.e.
// its first operand is OUR B buffer, its second OUR A buffer, m = N, n = M, k = K.  `batch` > 1: strided batches (element strides).
int lt_gemm(int opA, int opB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
            const float* bias, void* C, int64_t ldc, int dt_in, int dt_out, int batch, int64_t sa, int64_t sb, int64_t sc,
            void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!ws) ws_bytes = 0;
  std::lock_guard<std::mutex> lock(g_lt_mu);
  const LtApi& lt = lt_api();
  if (!lt.ok) {
    set_error("the BLAS library is needed for this product (fp32 features, or a bf16 shape the MFMA kernels do not take) but %s",
              lt.why.c_str());
    return SG_ERR_UNSUPPORTED;
  }
  int dev = 0;
  SG_HIP_TRY(hipGetDevice(&dev));
  SG_REQUIRE(dev >= 0 && dev < kMaxDevices, "device index %d out of range", dev);
  static int g_lt_refused = 0;        // (under g_lt_mu) the version the loaded library reported when it was refused: sticky
  if (g_lt_refused) {
    set_error("the loaded libhipblaslt reports version %d; this library was built against major version %d", g_lt_refused, HIPBLASLT_VERSION_MAJOR);
    return SG_ERR_UNSUPPORTED;
  }
  if (!g_lt_handle[dev]) {
    // the structs and enums this file was compiled against (hipblasLtMatmulAlgo_t is passed by value into the plan cache) are
    // those of ONE major version of the library: a process that loaded another one is refused, not miscomputed.  The handle
    // is published only AFTER the check (a handle stored first made every later call skip it), and the refusal sticks.
    hipblasLtHandle_t h = nullptr;
    SG_LT_TRY(lt.Create(&h));
    int ver = 0;
    if (lt.GetVersion && lt.GetVersion(h, &ver) == HIPBLAS_STATUS_SUCCESS) {
      // major * 100000 + minor * 100 + patch in the builds this was checked against, major * 10000 + .. in older ones: a version is
      // refused only when NEITHER decoding gives the major version of the headers (a build with major 0 passes the first test with
      // any version below 100000 -- such a library predates the plan structs used here by years and does not export Heuristic)
      if (ver / 100000 != HIPBLASLT_VERSION_MAJOR && ver / 10000 != HIPBLASLT_VERSION_MAJOR) {
        g_lt_refused = ver ? ver : -1;
        if (lt.Destroy) (void)lt.Destroy(h);
        set_error("the loaded libhipblaslt reports version %d; this library was built against major version %d", ver, HIPBLASLT_VERSION_MAJOR);
        return SG_ERR_UNSUPPORTED;
      }
    }
    g_lt_handle[dev] = h;
  }
  PlanKey key{dev, opA, opB, dt_in, dt_out, bias ? 1 : 0, batch, M, N, K, lda, ldb, ldc, sa, sb, sc};
  auto it = g_plans.find(key);
  if (it == g_plans.end()) {
    Plan p;
    SG_LT_TRY(lt.DescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    const hipblasOperation_t ta = opB ? HIPBLAS_OP_T : HIPBLAS_OP_N;     // library operand 1 = our B
    const hipblasOperation_t tb = opA ? HIPBLAS_OP_T : HIPBLAS_OP_N;     // library operand 2 = our A
    SG_LT_TRY(lt.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
    SG_LT_TRY(lt.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
    if (bias) {
      const hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
      SG_LT_TRY(lt.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)));
      const hipDataType bt = HIP_R_32F;
      SG_LT_TRY(lt.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
    }
    // stored (column-major) shapes: our B is [K, N] row-major = (N x K) column-major when opB = 0, [N, K] = (K x N) when opB = 1
    SG_LT_TRY(lt.LayoutCreate(&p.la, lt_type(dt_in), opB ? K : N, opB ? N : K, ldb));
    SG_LT_TRY(lt.LayoutCreate(&p.lb, lt_type(dt_in), opA ? M : K, opA ? K : M, lda));
    SG_LT_TRY(lt.LayoutCreate(&p.lc, lt_type(dt_out), N, M, ldc));
    if (batch > 1) {
      const int32_t bc = batch;
      for (auto pr : {std::make_pair(p.la, sb), std::make_pair(p.lb, sa), std::make_pair(p.lc, sc)}) {
        SG_LT_TRY(lt.LayoutSet(pr.first, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc)));
        const int64_t st = pr.second;
        SG_LT_TRY(lt.LayoutSet(pr.first, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &st, sizeof(st)));
      }
    }
    hipblasLtMatmulPreference_t pref = nullptr;
    SG_LT_TRY(lt.PrefCreate(&pref));
    const uint64_t max_ws = ws_bytes;
    hipblasStatus_t st = lt.PrefSet(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &max_ws, sizeof(max_ws));
    constexpr int kAsk = 8;
    hipblasLtMatmulHeuristicResult_t res[kAsk];
    int found = 0, pick = -1;
    if (st == HIPBLAS_STATUS_SUCCESS) {
      // (the bias pointer takes part in the heuristic's validity check of some solutions: set a plausible one)
      if (bias) st = lt.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
      if (st == HIPBLAS_STATUS_SUCCESS) st = lt.Heuristic(g_lt_handle[dev], p.desc, p.la, p.lb, p.lc, p.lc, pref, kAsk, res, &found);
    }
    lt.PrefDestroy(pref);
    // the library's ranking, first entry whose scratch fits (it has been seen to rank a solution above the stated limit first)
    for (int i = 0; i < found && pick < 0; ++i)
      if (res[i].state == HIPBLAS_STATUS_SUCCESS && res[i].workspaceSize <= ws_bytes) pick = i;
    if (st != HIPBLAS_STATUS_SUCCESS || pick < 0) {
      set_error("hipBLASLt has no solution for the product M=%lld N=%lld K=%lld (opA=%d opB=%d dtype %d -> %d, batch %d; status %d, "
                "%d candidates, %zu bytes of scratch)", (long long)M, (long long)N, (long long)K, opA, opB, dt_in, dt_out, batch,
                (int)st, found, ws_bytes);
      return SG_ERR_UNSUPPORTED;
    }
    p.algo = res[pick].algo;
    p.ws = res[pick].workspaceSize;
    it = g_plans.emplace(key, p).first;
  }
  Plan& p = it->second;
  SG_REQUIRE(p.ws <= ws_bytes, "hipBLASLt workspace: %zu bytes needed, %zu given", p.ws, ws_bytes);
  if (bias) SG_LT_TRY(lt.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
  const float one = 1.f, zero = 0.f;
  SG_LT_TRY(lt.Matmul(g_lt_handle[dev], p.desc, &one, B, p.la, A, p.lb, &zero, C, p.lc, C, p.lc, &p.algo, ws, ws_bytes, stream));
  return SG_OK;
}

inline bool a16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// the products the MFMA kernels leave to the BLAS library although they take the shape (functional.MFMA_MAX_WEIGHT_ELEMS):
// compute-bound ones that the persistent 256 x 256 kernel does not serve
constexpr int64_t kMfmaMaxWeightElems = 100000;

// sum of n_slabs [N, Kp] slab partials (contiguous, slab stride N * Kp) in slab order -> out (row stride ldo); any shape
__global__ __launch_bounds__(256) void slab_sum(const float* __restrict__ W, int n_slabs, int64_t elems, int Kp,
                                                float* __restrict__ out, int64_t ldo, const GradSink sink) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= elems) return;
  float acc = 0.f;
  for (int t = 0; t < n_slabs; ++t) acc += W[(int64_t)t * elems + e];
  out[(e / Kp) * ldo + e % Kp] = acc;
  if (sink.mode) *sink_ptr(sink, e / Kp, e % Kp) += acc;
}

}  // namespace

bool dense_nt_own(int dtype, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc) {
  if (dtype != SG_BF16 || K % 8 || N % 8 || lda % 8 || ldb % 8 || ldc % 8) return false;
  return N * K <= kMfmaMaxWeightElems || gemm_nt_takes_big_tile(M, N, K, lda, ldb, ldc);
}

bool dense_tn_own(int dtype, int64_t M, int64_t N, int64_t Kp, int64_t lda, int64_t ldb) {
  if (dtype != SG_BF16 || N % 8 || Kp % 8 || lda % 8 || ldb % 8) return false;
  return N * Kp <= kMfmaMaxWeightElems || gemm_tn_takes_big_tile(M, N, Kp, lda, ldb);
}

// slabs of the BLAS weight gradient: the library's own choice for a [N x M] x [M x Kp] product leaves most CUs idle
// (measured at V = 1 M: 5.5 ms fp32 for 256 x 768); M is cut into S slabs, one batched product, the slabs summed in order
static int64_t blas_tn_slabs(int dtype, int64_t M) {
  int64_t S = M / 4096;
  const int64_t cap = dtype == SG_F32 ? 128 : 64;
  return S > cap ? cap : S;
}

int64_t dense_tn_workspace(int dtype, int64_t M, int64_t N, int64_t Kp) {      // float32 elements
  if (thin_shape(N, Kp)) return thin_tn_blocks(M) * 256;
  if (dense_tn_own(dtype, M, N, Kp, 8, 8)) return gemm_tn_slabs(M, N, Kp) * N * Kp;
  const int64_t S = blas_tn_slabs(dtype, M);
  int64_t need = S > 1 ? (S + 1) * N * Kp : 0;
  if (dtype == SG_F32 && gemm_tn_f32s_supported(M, N, Kp, 4, 4)) {        // (sized for either engine: the A/B switch is a run-time knob)
    const int64_t split = gemm_tn_f32s_workspace(M, N, Kp) / 4;
    need = need > split ? need : split;
  }
  if (dtype == SG_F32 && mid_shape(N, Kp)) {
    const int64_t mid = mid_tn_workspace(M, N, Kp);
    need = need > mid ? need : mid;
  }
  return need;
}

// operands kept as planes (sg_common.h Planes) are read and written by the 128-row MFMA kernels only
bool dense_planes_ok_nt(int dtype, int64_t M, int64_t N, int64_t K) {
  return !thin_shape(N, K) && dense_nt_own(dtype, M, N, K, 8, 8, 8) && !gemm_nt_takes_big_tile(M, N, K, K, K, N) &&
         gemm_tile_rows(N) == 128;
}
bool dense_planes_ok_tn(int dtype, int64_t M, int64_t N, int64_t Kp) {
  return !thin_shape(N, Kp) && dense_tn_own(dtype, M, N, Kp, 8, 8) && !gemm_tn_takes_big_tile(M, N, Kp, N, Kp);
}

int dense_nt(const void* A, int64_t lda, const void* Bp, const float* B32, int64_t ldb, const float* bias, void* C, int64_t ldc,
             int64_t M, int64_t N, int64_t K, int dtype, float* moments, bool* moments_done, void* blas_ws, size_t blas_ws_bytes,
             hipStream_t stream, Planes pa, Planes pc, const void* presplit, int64_t presplit_rows) {
  if (moments_done) *moments_done = false;
  if (M == 0 || N == 0) return SG_OK;
  if (pa.on() || pc.on()) {
    SG_REQUIRE(dense_planes_ok_nt(dtype, M, N, K) && a16(A) && a16(Bp) && a16(C),
               "dense_nt: an operand kept as planes for a product the 128-row kernel does not serve (M=%lld N=%lld K=%lld)",
               (long long)M, (long long)N, (long long)K);
    if (moments_done) *moments_done = moments != nullptr;
    TraceScope ts(1, dtype, 1, M, N, K, stream);
    return launch_gemm_nt(A, lda, Bp, ldb, bias, C, ldc, M, N, K, dtype, moments, stream, pa, pc);
  }
  if (thin_shape(N, K) && B32) {
    TraceScope ts(1, dtype, 2, M, N, K, stream);
    return launch_thin_nt(A, lda, B32, ldb, bias, C, ldc, M, N, K, dtype, stream);
  }
  if (dense_nt_own(dtype, M, N, K, lda, ldb, ldc) && a16(A) && a16(Bp) && a16(C)) {
    // (the 256 x 256 kernel leaves no tile moments: a call that wants them for such a shape takes a separate pass)
    const bool big = gemm_nt_takes_big_tile(M, N, K, lda, ldb, ldc);
    float* mom = (moments && !big) ? moments : nullptr;
    if (moments_done) *moments_done = mom != nullptr;
    TraceScope ts(1, dtype, 1, M, N, K, stream);
    return launch_gemm_nt(A, lda, Bp, ldb, bias, C, ldc, M, N, K, dtype, mom, stream);
  }
  if (dtype == SG_F32 && split_engine_enabled(0) && split_nt_pays(M, N, K) && gemm_nt_f32s_supported(M, N, K, lda, ldc) && a16(A) && a16(C) &&
      (presplit || (blas_ws && (int64_t)blas_ws_bytes >= gemm_nt_f32s_workspace(N, K)))) {
    TraceScope ts(1, dtype, 4, M, N, K, stream);
    return launch_gemm_nt_f32s((const float*)A, lda, (const float*)Bp, ldb, 1, bias, (float*)C, ldc, M, N, K, blas_ws,
                               (int64_t)blas_ws_bytes, stream, presplit, presplit_rows);
  }
  if (dtype == SG_F32 && split_engine_enabled(0) && mid_shape(N, K) && lda % 4 == 0 && ldc % 4 == 0 && a16(A) && a16(C)) {
    TraceScope ts(1, dtype, 2, M, N, K, stream);
    return launch_mid_nt((const float*)A, lda, (const float*)Bp, ldb, 1, bias, (float*)C, ldc, M, N, K, stream);
  }
  TraceScope ts(1, dtype, 3, M, N, K, stream);
  return lt_gemm(0, 1, M, N, K, A, lda, Bp, ldb, bias, C, ldc, dtype, dtype, 1, 0, 0, 0, blas_ws, blas_ws_bytes, stream);
}

int dense_nn(const void* A, int64_t lda, const void* Bp, int64_t ldb, const void* Bt, int64_t ldbt, const float* Bt32, void* C,
             int64_t ldc, int64_t M, int64_t N, int64_t K, int dtype, void* blas_ws, size_t blas_ws_bytes, hipStream_t stream,
             Planes pa, Planes pc, const void* presplit, int64_t presplit_rows) {
  // C[M, N] = A[M, K] B[K, N]; Bt = B^T [N, K] where a transposed copy exists (the MFMA / thin kernels read that one)
  if (M == 0 || N == 0) return SG_OK;
  if (pa.on() || pc.on()) {
    SG_REQUIRE(Bt && dense_planes_ok_nt(dtype, M, N, K) && a16(A) && a16(Bt) && a16(C),
               "dense_nn: an operand kept as planes for a product the 128-row kernel does not serve (M=%lld N=%lld K=%lld)",
               (long long)M, (long long)N, (long long)K);
    TraceScope ts(1, dtype, 1, M, N, K, stream);
    return launch_gemm_nt(A, lda, Bt, ldbt, nullptr, C, ldc, M, N, K, dtype, nullptr, stream, pa, pc);
  }
  if (thin_shape(N, K) && Bt32) {
    TraceScope ts(1, dtype, 2, M, N, K, stream);
    return launch_thin_nt(A, lda, Bt32, ldbt, nullptr, C, ldc, M, N, K, dtype, stream);
  }
  if (Bt && dense_nt_own(dtype, M, N, K, lda, ldbt, ldc) && a16(A) && a16(Bt) && a16(C)) {
    TraceScope ts(1, dtype, 1, M, N, K, stream);
    return launch_gemm_nt(A, lda, Bt, ldbt, nullptr, C, ldc, M, N, K, dtype, nullptr, stream);
  }
  if (dtype == SG_F32 && split_engine_enabled(1) && split_nt_pays(M, N, K) && gemm_nt_f32s_supported(M, N, K, lda, ldc) && a16(A) && a16(C) &&
      (presplit || (blas_ws && (int64_t)blas_ws_bytes >= gemm_nt_f32s_workspace(N, K)))) {
    TraceScope ts(1, dtype, 4, M, N, K, stream);      // B is [K, N]: element (n, k) at Bp[k * ldb + n]
    return launch_gemm_nt_f32s((const float*)A, lda, (const float*)Bp, 1, ldb, nullptr, (float*)C, ldc, M, N, K, blas_ws,
                               (int64_t)blas_ws_bytes, stream, presplit, presplit_rows);
  }
  if (dtype == SG_F32 && split_engine_enabled(1) && mid_shape(N, K) && lda % 4 == 0 && ldc % 4 == 0 && a16(A) && a16(C)) {
    TraceScope ts(1, dtype, 2, M, N, K, stream);      // B is [K, N]: element (n, k) at Bp[k * ldb + n]
    return launch_mid_nt((const float*)A, lda, (const float*)Bp, 1, ldb, nullptr, (float*)C, ldc, M, N, K, stream);
  }
  TraceScope ts(1, dtype, 3, M, N, K, stream);
  return lt_gemm(0, 0, M, N, K, A, lda, Bp, ldb, nullptr, C, ldc, dtype, dtype, 1, 0, 0, 0, blas_ws, blas_ws_bytes, stream);
}

int dense_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t Kp, int dtype, float* ws,
             float* out, int64_t ldo, void* blas_ws, size_t blas_ws_bytes, hipStream_t stream, const GradSink* sink, bool* sunk,
             Planes pa, Planes pb) {
  // out[N, Kp] (float32) = A[M, N]^T B[M, Kp]
  if (sunk) *sunk = false;
  if (N == 0 || Kp == 0) return SG_OK;
  SG_REQUIRE(M > 0, "dense_tn: no rows");
  if (pa.on() || pb.on()) {
    SG_REQUIRE(dense_planes_ok_tn(dtype, M, N, Kp) && a16(A) && a16(B) && a16(out) && ldo % 4 == 0,
               "dense_tn: an operand kept as planes for a product the 128 x 128 kernel does not serve (M=%lld N=%lld Kp=%lld)",
               (long long)M, (long long)N, (long long)Kp);
    TraceScope ts(2, dtype, 1, M, N, Kp, stream);
    const bool can = sink != nullptr && sink->Cin % 4 == 0;
    if (sunk) *sunk = can;
    return launch_gemm_tn(A, lda, B, ldb, M, N, Kp, dtype, ws, out, ldo, stream, can ? sink : nullptr, pa, pb);
  }
  if (thin_shape(N, Kp)) {
    TraceScope ts(2, dtype, 2, M, N, Kp, stream);
    if (sunk) *sunk = sink != nullptr;
    return launch_thin_tn(A, lda, B, ldb, M, N, Kp, dtype, ws, out, ldo, stream, sink);
  }
  if (dense_tn_own(dtype, M, N, Kp, lda, ldb) && a16(A) && a16(B) && a16(out) && ldo % 4 == 0) {
    TraceScope ts(2, dtype, 1, M, N, Kp, stream);
    const bool can = sink != nullptr && sink->Cin % 4 == 0;
    if (sunk) *sunk = can;
    return launch_gemm_tn(A, lda, B, ldb, M, N, Kp, dtype, ws, out, ldo, stream, can ? sink : nullptr);
  }
  if (dtype == SG_F32 && split_engine_enabled(2) && gemm_tn_f32s_supported(M, N, Kp, lda, ldb) && a16(A) && a16(B) && a16(out) &&
      a16(ws) && ldo % 4 == 0) {
    TraceScope ts(2, dtype, 4, M, N, Kp, stream);
    const bool can = sink != nullptr && sink->Cin % 4 == 0;
    if (sunk) *sunk = can;
    return launch_gemm_tn_f32s((const float*)A, lda, (const float*)B, ldb, M, N, Kp, ws, gemm_tn_f32s_workspace(M, N, Kp), out, ldo,
                               stream, can ? sink : nullptr);
  }
  if (dtype == SG_F32 && split_engine_enabled(2) && mid_shape(N, Kp) && lda % 4 == 0 && ldb % 4 == 0 && ldo % 4 == 0 && a16(A) &&
      a16(B) && a16(out) && a16(ws)) {
    TraceScope ts(2, dtype, 2, M, N, Kp, stream);
    const bool can = sink != nullptr && sink->Cin % 4 == 0;
    if (sunk) *sunk = can;
    return launch_mid_tn((const float*)A, lda, (const float*)B, ldb, M, N, Kp, ws, out, ldo, stream, can ? sink : nullptr);
  }
  TraceScope ts(2, dtype, 3, M, N, Kp, stream);
  const int64_t S = blas_tn_slabs(dtype, M);
  const int64_t esz = dtype == SG_F32 ? 4 : 2;
  if (S <= 1)
    return lt_gemm(1, 0, N, Kp, M, A, lda, B, ldb, nullptr, out, ldo, dtype, SG_F32, 1, 0, 0, 0, blas_ws, blas_ws_bytes, stream);
  const int64_t Ms = M / S, Mb = Ms * S, elems = N * Kp;
  int rc = lt_gemm(1, 0, N, Kp, Ms, A, lda, B, ldb, nullptr, ws, Kp, dtype, SG_F32, (int)S, Ms * lda, Ms * ldb, elems, blas_ws,
                   blas_ws_bytes, stream);
  if (rc != SG_OK) return rc;
  int n_slabs = (int)S;
  if (Mb < M) {      // the rows past the last full slab: one more slab
    rc = lt_gemm(1, 0, N, Kp, M - Mb, (const char*)A + Mb * lda * esz, lda, (const char*)B + Mb * ldb * esz, ldb, nullptr,
                 ws + S * elems, Kp, dtype, SG_F32, 1, 0, 0, 0, blas_ws, blas_ws_bytes, stream);
    if (rc != SG_OK) return rc;
    ++n_slabs;
  }
  slab_sum<<<(int)((elems + 255) / 256), 256, 0, stream>>>(ws, n_slabs, elems, (int)Kp, out, ldo, sink ? *sink : GradSink{});
  if (sunk) *sunk = sink != nullptr;
  SG_HIP_TRY(hipGetLastError());
  return SG_OK;
}

}  // namespace sg
