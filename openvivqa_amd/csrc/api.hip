// extern "C" entry points of libovqa_hip.so (declared in include/ovqa_hip.h).
// Validates arguments, then routes OVQA_F32 to the exact-fp32 kernels and
// OVQA_BF16 to the MFMA kernels (falling back to the bf16 instantiation of the
// reference-grade kernels only for shapes the MFMA kernels do not tile, or when
// OVQA_FORCE_SIMPLE=1 is set for A/B debugging -- both are HIP kernels, there
// is no CPU path in this library).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include "common.h"
#include "kernels.h"

static thread_local char g_err[512] = "";

void ovqa_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// which kernel family the last entry point called on this thread ran ("mfma" = the gfx950 matrix-core kernels,
// "simple" = the VALU reference-grade kernels, "stream" = elementwise / reduction kernels with a single form)
static thread_local const char* g_dispatch = "";

// OVQA_REQUIRE_MFMA=1: a bf16 call that would fall back to the VALU kernels is an error instead (tests of the
// BASELINE shapes run with it, so a silent fallback cannot pass for the MFMA path)
static bool require_mfma() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("OVQA_REQUIRE_MFMA");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}
#define OVQA_FALLBACK(what)                                                                            \
  do {                                                                                                 \
    g_dispatch = "simple";                                                                             \
    OVQA_REQUIRE(!(dtype == OVQA_BF16 && require_mfma() && !force_simple()), OVQA_ERR_UNSUPPORTED,     \
                 what ": shape/alignment not covered by the MFMA kernels and OVQA_REQUIRE_MFMA=1");    \
  } while (0)

static bool force_simple() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("OVQA_FORCE_SIMPLE");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

// A/B switch: run self-attention as the separate projection GEMM + attention kernels (read per call: tests flip it)
static bool no_fused_qkv() {
  const char* e = getenv("OVQA_NO_FUSED_QKV");
  return e && e[0] == '1';
}

// launch timing state: process-wide (autograd runs backward on its own thread), serialised by a mutex
struct LaunchTimer {
  hipEvent_t* ev = nullptr;  // 2 * cap events: start, stop per launch
  int cap = 0, limit = 0, n = 0;  // events allocated, launches to record this time, launches recorded
  bool on = false;
};
static LaunchTimer g_timer;
static std::mutex g_timer_mu;
static std::atomic<bool> g_timer_on{false};

bool ovqa_timer_next(hipEvent_t* start, hipEvent_t* stop) {
  if (!g_timer_on.load(std::memory_order_acquire)) return false;
  std::lock_guard<std::mutex> lock(g_timer_mu);
  LaunchTimer& t = g_timer;
  if (!t.on || t.n >= t.limit) return false;
  *start = t.ev[2 * t.n];
  *stop = t.ev[2 * t.n + 1];
  t.n++;
  return true;
}

static inline bool dtype_ok(int d) { return d == OVQA_F32 || d == OVQA_BF16; }

extern "C" {

int ovqa_abi_version(void) { return OVQA_ABI_VERSION; }
const char* ovqa_last_error(void) { return g_err; }
const char* ovqa_last_dispatch(void) { return g_dispatch; }

int ovqa_launch_timing_begin(int max_launches) {
  std::lock_guard<std::mutex> lock(g_timer_mu);
  LaunchTimer& t = g_timer;
  OVQA_REQUIRE(max_launches > 0 && max_launches <= 65536, OVQA_ERR_BAD_ARG, "launch_timing_begin: bad capacity");
  OVQA_REQUIRE(!t.on, OVQA_ERR_BAD_ARG, "launch_timing_begin: already armed");
  if (t.cap < max_launches) {
    for (int i = 0; i < 2 * t.cap; i++) (void)hipEventDestroy(t.ev[i]);
    free(t.ev);
    t.ev = (hipEvent_t*)calloc((size_t)2 * max_launches, sizeof(hipEvent_t));
    OVQA_REQUIRE(t.ev, OVQA_ERR_BAD_ARG, "launch_timing_begin: out of memory");
    t.cap = 0;
    for (int i = 0; i < 2 * max_launches; i++) {
      hipError_t e = hipEventCreate(&t.ev[i]);
      OVQA_REQUIRE(e == hipSuccess, OVQA_ERR_LAUNCH, "launch_timing_begin: hipEventCreate: %s", hipGetErrorString(e));
    }
    t.cap = max_launches;
  }
  t.n = 0;
  t.limit = max_launches;
  t.on = true;
  g_timer_on.store(true, std::memory_order_release);
  return OVQA_OK;
}

int ovqa_stream_priority_range(int* least, int* greatest) {
  OVQA_REQUIRE(least && greatest, OVQA_ERR_BAD_ARG, "stream_priority_range: null pointer");
  hipError_t e = hipDeviceGetStreamPriorityRange(least, greatest);
  OVQA_REQUIRE(e == hipSuccess, OVQA_ERR_LAUNCH, "hipDeviceGetStreamPriorityRange: %s", hipGetErrorString(e));
  return OVQA_OK;
}

int ovqa_stream_create(void** stream, int priority, const uint32_t* cu_mask, int32_t n_words) {
  OVQA_REQUIRE(stream != nullptr && n_words >= 0 && (n_words == 0 || cu_mask != nullptr), OVQA_ERR_BAD_ARG,
               "stream_create: bad argument");
  hipStream_t st = nullptr;
  hipError_t e;
  if (n_words > 0) {
    uint32_t any = 0;
    for (int i = 0; i < n_words; i++) any |= cu_mask[i];
    OVQA_REQUIRE(any != 0, OVQA_ERR_BAD_ARG, "stream_create: an empty CU mask would never run anything");
    e = hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, cu_mask);
  } else {
    e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, priority);
  }
  OVQA_REQUIRE(e == hipSuccess, OVQA_ERR_LAUNCH, "stream_create: %s", hipGetErrorString(e));
  *stream = (void*)st;
  return OVQA_OK;
}

int ovqa_stream_destroy(void* stream) {
  if (stream == nullptr) return OVQA_OK;
  hipError_t e = hipStreamDestroy((hipStream_t)stream);
  OVQA_REQUIRE(e == hipSuccess, OVQA_ERR_LAUNCH, "stream_destroy: %s", hipGetErrorString(e));
  return OVQA_OK;
}

int ovqa_launch_timing_count(void) {
  std::lock_guard<std::mutex> lock(g_timer_mu);
  return g_timer.n;
}

int ovqa_launch_timing_end(float* us, int cap) {
  std::lock_guard<std::mutex> lock(g_timer_mu);
  LaunchTimer& t = g_timer;
  OVQA_REQUIRE(t.on, OVQA_ERR_BAD_ARG, "launch_timing_end: not armed");
  t.on = false;
  g_timer_on.store(false, std::memory_order_release);
  OVQA_REQUIRE(us && cap >= t.n, OVQA_ERR_BAD_ARG, "launch_timing_end: output holds %d of %d launches", cap, t.n);
  for (int i = 0; i < t.n; i++) {
    hipError_t e = hipEventSynchronize(t.ev[2 * i + 1]);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, t.ev[2 * i], t.ev[2 * i + 1]);
    OVQA_REQUIRE(e == hipSuccess, OVQA_ERR_LAUNCH, "launch_timing_end: launch %d: %s", i, hipGetErrorString(e));
    us[i] = ms * 1e3f;
  }
  return t.n;
}
int64_t ovqa_workspace_bytes(void) { return ovqa::kWorkspaceBytes; }

int ovqa_linear_fwd(int dtype, int epilogue, const void* x, int64_t ldx, const void* w, const float* bias,
                    const void* residual, int64_t ldres, void* y, int64_t ldy, void* preact, int64_t M, int64_t N,
                    int64_t K, const ovqa_dropout* drop, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "linear_fwd: bad dtype %d", dtype);
  OVQA_REQUIRE(M >= 0 && N > 0 && K > 0, OVQA_ERR_BAD_ARG, "linear_fwd: bad sizes M=%lld N=%lld K=%lld", (long long)M,
               (long long)N, (long long)K);
  if (M == 0) return OVQA_OK;
  OVQA_REQUIRE(x && w && y, OVQA_ERR_BAD_ARG, "linear_fwd: null pointer");
  OVQA_REQUIRE(ldx >= K && ldy >= N, OVQA_ERR_BAD_ARG, "linear_fwd: ldx/ldy smaller than the row length");
  OVQA_REQUIRE(M * (N > K ? N : K) < (1ll << 32), OVQA_ERR_UNSUPPORTED, "linear_fwd: more than 2^32 elements");
  const DropArgs da = make_drop_args(drop);
  hipStream_t st = as_stream(stream);
  if (dtype == OVQA_BF16 && !force_simple() && ovqa::mfma_linear_fwd_supported(epilogue, M, N, K, ldx, ldy, ldres)) {
    g_dispatch = "mfma";
    return ovqa::mfma_linear_fwd(epilogue, x, ldx, w, bias, residual, ldres, y, ldy, preact, M, N, K, da, st);
  }
  OVQA_FALLBACK("linear_fwd");
  return ovqa::simple_linear_fwd(dtype, epilogue, x, ldx, w, bias, residual, ldres, y, ldy, preact, M, N, K, da, st);
}

int ovqa_linear_fwd_split3(int dtype, const void* x, int64_t ldx, const void* w, const float* bias, void* y0, int64_t ld0,
                           void* y1, int64_t ld1, void* y2, int64_t ld2, int64_t M, int64_t F, int64_t K, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "linear_fwd_split3: bad dtype %d", dtype);
  OVQA_REQUIRE(M >= 0 && F > 0 && K > 0, OVQA_ERR_BAD_ARG, "linear_fwd_split3: bad sizes");
  if (M == 0) return OVQA_OK;
  OVQA_REQUIRE(x && w && y0 && y1 && y2, OVQA_ERR_BAD_ARG, "linear_fwd_split3: null pointer");
  OVQA_REQUIRE(ldx >= K && ld0 >= F && ld1 >= F && ld2 >= F, OVQA_ERR_BAD_ARG, "linear_fwd_split3: ld smaller than the row");
  OVQA_REQUIRE(M * (3 * F > K ? 3 * F : K) < (1ll << 32), OVQA_ERR_UNSUPPORTED, "linear_fwd_split3: more than 2^32 elements");
  hipStream_t st = as_stream(stream);
  if (dtype == OVQA_BF16 && !force_simple() &&
      ovqa::mfma_linear_fwd_split3_supported(x, ldx, w, bias, y0, ld0, y1, ld1, y2, ld2, M, F, K)) {
    g_dispatch = "mfma";
    return ovqa::mfma_linear_fwd_split3(x, ldx, w, bias, y0, ld0, y1, ld1, y2, ld2, M, F, K, st);
  }
  OVQA_FALLBACK("linear_fwd_split3");  // the three products one after the other on the VALU kernels
  const size_t es = dtype == OVQA_BF16 ? 2 : 4;
  void* ys[3] = {y0, y1, y2};
  const int64_t lds[3] = {ld0, ld1, ld2};
  const DropArgs none = make_drop_args(nullptr);
  for (int i = 0; i < 3; i++) {
    int rc = ovqa::simple_linear_fwd(dtype, OVQA_EPI_BIAS, x, ldx, (const char*)w + (size_t)i * F * K * es,
                                     bias ? bias + (size_t)i * F : nullptr, nullptr, 0, ys[i], lds[i], nullptr, M, F, K,
                                     none, st);
    if (rc != OVQA_OK) return rc;
  }
  return OVQA_OK;
}

int ovqa_linear_fwd_res32(const void* x, int64_t ldx, const void* w, const float* bias, const float* residual,
                          int64_t ldres, const ovqa_ln_ref* ln, float* pre, int64_t ldpre, int64_t M, int64_t N, int64_t K,
                          const ovqa_dropout* drop, void* stream) {
  OVQA_REQUIRE(M >= 0 && N > 0 && K > 0, OVQA_ERR_BAD_ARG, "linear_fwd_res32: bad sizes M=%lld N=%lld K=%lld", (long long)M,
               (long long)N, (long long)K);
  if (M == 0) return OVQA_OK;
  OVQA_REQUIRE(x && w && residual && pre, OVQA_ERR_BAD_ARG, "linear_fwd_res32: null pointer");
  OVQA_REQUIRE(ldx >= K && ldpre >= N && ldres >= N, OVQA_ERR_BAD_ARG, "linear_fwd_res32: ld smaller than the row length");
  OVQA_REQUIRE(!ln || (ln->mean && ln->rstd && ln->gamma && ln->beta), OVQA_ERR_BAD_ARG,
               "linear_fwd_res32: incomplete LayerNorm reference");
  OVQA_REQUIRE(M * (N > K ? N : K) < (1ll << 32), OVQA_ERR_UNSUPPORTED, "linear_fwd_res32: more than 2^32 elements");
  const DropArgs da = make_drop_args(drop);
  const float *mean = ln ? ln->mean : nullptr, *rstd = ln ? ln->rstd : nullptr;
  const float *gamma = ln ? ln->gamma : nullptr, *beta = ln ? ln->beta : nullptr;
  if (!force_simple() && ovqa::mfma_gemm_supported(N, M, K, K, ldx) && ldpre % 4 == 0 && ldres % 4 == 0) {
    g_dispatch = "mfma";
    return ovqa::mfma_linear_fwd_res32(x, ldx, w, bias, residual, ldres, mean, rstd, gamma, beta, pre, ldpre, M, N, K, da,
                                       as_stream(stream));
  }
  const int dtype = OVQA_BF16;
  OVQA_FALLBACK("linear_fwd_res32");
  return ovqa::simple_linear_fwd_res32(x, ldx, w, bias, residual, ldres, mean, rstd, gamma, beta, pre, ldpre, M, N, K, da,
                                       as_stream(stream));
}

int ovqa_linear_bwd_data(int dtype, const void* dy, int64_t lddy, const void* w, void* dx, int64_t lddx,
                         const void* gelu_preact, const void* addend, int64_t ldadd, int64_t M, int64_t N, int64_t K,
                         const ovqa_dropout* drop, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "linear_bwd_data: bad dtype %d", dtype);
  OVQA_REQUIRE(M >= 0 && N > 0 && K > 0, OVQA_ERR_BAD_ARG, "linear_bwd_data: bad sizes");
  if (M == 0) return OVQA_OK;
  OVQA_REQUIRE(dy && w && dx, OVQA_ERR_BAD_ARG, "linear_bwd_data: null pointer");
  OVQA_REQUIRE(lddy >= N && lddx >= K && (!addend || ldadd >= K), OVQA_ERR_BAD_ARG,
               "linear_bwd_data: ld smaller than the row length");
  OVQA_REQUIRE(M * (N > K ? N : K) < (1ll << 32), OVQA_ERR_UNSUPPORTED, "linear_bwd_data: more than 2^32 elements");
  if (dtype == OVQA_BF16 && !force_simple() && ovqa::mfma_linear_bwd_data_supported(M, N, K, lddy, lddx)) {
    g_dispatch = "mfma";
    return ovqa::mfma_linear_bwd_data(dy, lddy, w, dx, lddx, gelu_preact, addend, ldadd, M, N, K, make_drop_args(drop),
                                      as_stream(stream));
  }
  OVQA_FALLBACK("linear_bwd_data");
  return ovqa::simple_linear_bwd_data(dtype, dy, lddy, w, dx, lddx, gelu_preact, addend, ldadd, M, N, K,
                                      make_drop_args(drop), as_stream(stream));
}

int ovqa_grouped_linear_bwd_weight(int dtype, const ovqa_wgrad_problem* problems_dev, const int32_t* tiles_dev,
                                   int64_t n_tiles, int32_t form, void* stream) {
  OVQA_REQUIRE(dtype == OVQA_BF16, OVQA_ERR_UNSUPPORTED, "grouped_linear_bwd_weight: bf16 only");
  OVQA_REQUIRE(n_tiles >= 0 && (n_tiles == 0 || (problems_dev && tiles_dev)), OVQA_ERR_BAD_ARG,
               "grouped_linear_bwd_weight: bad argument");
  OVQA_REQUIRE(n_tiles < (1ll << 31), OVQA_ERR_UNSUPPORTED, "grouped_linear_bwd_weight: too many tiles");
  OVQA_REQUIRE(form == 0 || form == 1, OVQA_ERR_BAD_ARG, "grouped_linear_bwd_weight: form must be 0 or 1");
  g_dispatch = "mfma";
  return ovqa::mfma_grouped_wgrad(problems_dev, tiles_dev, n_tiles, form != 0 && !force_simple(), as_stream(stream));
}

int ovqa_grouped_linear_bwd_weight_adam(int dtype, const ovqa_wgrad_problem* problems_dev, const int32_t* tiles_dev,
                                        int64_t n_tiles, const ovqa_adam_target* targets_dev,
                                        const ovqa_adam_consts* consts, void* stream) {
  OVQA_REQUIRE(dtype == OVQA_BF16, OVQA_ERR_UNSUPPORTED, "grouped_linear_bwd_weight_adam: bf16 only");
  OVQA_REQUIRE(n_tiles >= 0 && (n_tiles == 0 || (problems_dev && tiles_dev && targets_dev)) && consts, OVQA_ERR_BAD_ARG,
               "grouped_linear_bwd_weight_adam: bad argument");
  OVQA_REQUIRE(n_tiles < (1ll << 31), OVQA_ERR_UNSUPPORTED, "grouped_linear_bwd_weight_adam: too many tiles");
  OVQA_REQUIRE(!force_simple(), OVQA_ERR_UNSUPPORTED,
               "grouped_linear_bwd_weight_adam: the direct-to-LDS form only (no fallback under OVQA_FORCE_SIMPLE)");
  if (n_tiles == 0) return OVQA_OK;
  g_dispatch = "mfma-fused";
  return ovqa::mfma_grouped_linear_bwd_weight_adam(problems_dev, tiles_dev, n_tiles, targets_dev, *consts, as_stream(stream));
}

int ovqa_bias_grad(int dtype, const void* dy, int64_t lddy, float* db, int64_t M, int64_t N, int accumulate,
                   void* stream) {
  OVQA_REQUIRE(dtype == OVQA_BF16, OVQA_ERR_UNSUPPORTED, "bias_grad: bf16 only (fp32 goes through linear_bwd_weight)");
  OVQA_REQUIRE(dy && db && M >= 0 && N > 0 && N % 8 == 0 && lddy % 8 == 0, OVQA_ERR_BAD_ARG, "bias_grad: bad argument");
  return ovqa::colsum_bf16(dy, lddy, db, M, N, accumulate, as_stream(stream));
}

int ovqa_linear_bwd_data_wt(int dtype, const void* dy, int64_t lddy, const void* wt, int64_t ldwt, void* dx, int64_t lddx,
                            const void* gelu_preact, const void* addend, int64_t ldadd, int64_t M, int64_t N, int64_t K,
                            const ovqa_dropout* drop, void* stream) {
  OVQA_REQUIRE(dtype == OVQA_BF16, OVQA_ERR_UNSUPPORTED, "linear_bwd_data_wt: bf16 only");
  OVQA_REQUIRE(M >= 0 && N > 0 && K > 0, OVQA_ERR_BAD_ARG, "linear_bwd_data_wt: bad sizes");
  if (M == 0) return OVQA_OK;
  OVQA_REQUIRE(dy && wt && dx, OVQA_ERR_BAD_ARG, "linear_bwd_data_wt: null pointer");
  OVQA_REQUIRE(lddy >= N && lddx >= K && ldwt >= N && (!addend || ldadd >= K), OVQA_ERR_BAD_ARG,
               "linear_bwd_data_wt: ld smaller than the row length");
  OVQA_REQUIRE(M * (N > K ? N : K) < (1ll << 32), OVQA_ERR_UNSUPPORTED, "linear_bwd_data_wt: more than 2^32 elements");
  OVQA_REQUIRE(ovqa::mfma_gemm_supported(K, M, N, ldwt, lddy) && lddx % 4 == 0, OVQA_ERR_UNSUPPORTED,
               "linear_bwd_data_wt: needs N, K, lddy, ldwt multiples of 8 and lddx a multiple of 4");
  g_dispatch = "mfma";
  return ovqa::mfma_linear_bwd_data_wt(dy, lddy, wt, ldwt, dx, lddx, gelu_preact, addend, ldadd, M, N, K,
                                       make_drop_args(drop), as_stream(stream));
}

int ovqa_grouped_transpose(const ovqa_transpose_problem* problems, int32_t n_problems, int32_t max_tiles, void* stream) {
  OVQA_REQUIRE(n_problems >= 0 && max_tiles >= 0 && n_problems <= 65535, OVQA_ERR_BAD_ARG, "grouped_transpose: bad sizes");
  OVQA_REQUIRE(n_problems == 0 || problems != nullptr, OVQA_ERR_BAD_ARG, "grouped_transpose: null table");
  return ovqa::grouped_transpose_bf16(problems, n_problems, max_tiles, as_stream(stream));
}

int ovqa_linear_bwd_weight(int dtype, const void* dy, int64_t lddy, const void* x, int64_t ldx, float* dw, float* db,
                           int64_t M, int64_t N, int64_t K, int accumulate, void* ws, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "linear_bwd_weight: bad dtype %d", dtype);
  OVQA_REQUIRE(M >= 0 && N > 0 && K > 0, OVQA_ERR_BAD_ARG, "linear_bwd_weight: bad sizes");
  OVQA_REQUIRE(dw != nullptr, OVQA_ERR_BAD_ARG, "linear_bwd_weight: dw is NULL");
  OVQA_REQUIRE(M == 0 || (dy && x), OVQA_ERR_BAD_ARG, "linear_bwd_weight: null pointer");
  (void)ws;
  const int acc_w = accumulate & 1, acc_b = (accumulate >> 1) & 1;
  if (M == 0) {
    hipError_t e = hipSuccess;
    if (!acc_w) e = hipMemsetAsync(dw, 0, (size_t)N * K * sizeof(float), as_stream(stream));
    if (e == hipSuccess && db && !acc_b) e = hipMemsetAsync(db, 0, (size_t)N * sizeof(float), as_stream(stream));
    {
      OVQA_REQUIRE(e == hipSuccess, OVQA_ERR_LAUNCH, "linear_bwd_weight: hipMemsetAsync: %s", hipGetErrorString(e));
    }
    return OVQA_OK;
  }
  if (dtype == OVQA_BF16 && !force_simple() && ovqa::mfma_linear_bwd_weight_supported(M, N, K, lddy, ldx)) {
    g_dispatch = "mfma";
    return ovqa::mfma_linear_bwd_weight(dy, lddy, x, ldx, dw, db, M, N, K, acc_w, acc_b, as_stream(stream));
  }
  OVQA_FALLBACK("linear_bwd_weight");
  return ovqa::simple_linear_bwd_weight(dtype, dy, lddy, x, ldx, dw, db, M, N, K, acc_w, acc_b, as_stream(stream));
}

int ovqa_layernorm_fwd(int dtype, int in_dtype, const void* x, const float* gamma, const float* beta,
                       const float* pos, int64_t pos_rows, void* y, float* y_f32, float* mean, float* rstd, int64_t M,
                       int64_t D, float eps, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype) && dtype_ok(in_dtype), OVQA_ERR_BAD_ARG, "layernorm_fwd: bad dtype");
  OVQA_REQUIRE(M >= 0 && D > 0, OVQA_ERR_BAD_ARG, "layernorm_fwd: bad sizes");
  OVQA_REQUIRE(M == 0 || (x && gamma && beta && y), OVQA_ERR_BAD_ARG, "layernorm_fwd: null pointer");
  OVQA_REQUIRE(pos == nullptr || pos_rows > 0, OVQA_ERR_BAD_ARG, "layernorm_fwd: pos_rows must be > 0");
  g_dispatch = "";
  return ovqa::layernorm_fwd(dtype, in_dtype, x, gamma, beta, pos, pos_rows, y, y_f32, mean, rstd, M, D, eps,
                             as_stream(stream));
}

int ovqa_layernorm_bwd(int dtype, int dx_dtype, const void* dy, const void* x, int x_dtype, const float* gamma,
                       const float* mean, const float* rstd, void* dx, void* dx_dropped, float* dgamma, float* dbeta,
                       int64_t M, int64_t D, int accumulate, const ovqa_dropout* drop, void* ws, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype) && dtype_ok(dx_dtype) && dtype_ok(x_dtype), OVQA_ERR_BAD_ARG, "layernorm_bwd: bad dtype");
  OVQA_REQUIRE(M >= 0 && D > 0, OVQA_ERR_BAD_ARG, "layernorm_bwd: bad sizes");
  OVQA_REQUIRE(M == 0 || (dy && x && gamma && mean && rstd && dx), OVQA_ERR_BAD_ARG, "layernorm_bwd: null pointer");
  OVQA_REQUIRE(M * D < (1ll << 32), OVQA_ERR_UNSUPPORTED, "layernorm_bwd: more than 2^32 elements");
  DropArgs da = make_drop_args(drop);
  if (da.p <= 0.f) dx_dropped = nullptr;
  return ovqa::layernorm_bwd(dtype, dx_dtype, dy, x, x_dtype, gamma, mean, rstd, dx, dx_dropped, dgamma, dbeta, M, D,
                             accumulate, da, ws, as_stream(stream));
}

int ovqa_layernorm_bwd_blocks(int64_t M, int64_t D) { return ovqa::layernorm_bwd_blocks(M, D); }

int ovqa_grouped_partial_reduce(const ovqa_reduce_problem* problems, int32_t n_problems, int32_t max_blocks,
                                int32_t max_D, void* stream) {
  OVQA_REQUIRE(n_problems >= 0 && max_blocks >= 0 && max_D >= 0, OVQA_ERR_BAD_ARG, "grouped_partial_reduce: bad sizes");
  OVQA_REQUIRE(n_problems == 0 || problems != nullptr, OVQA_ERR_BAD_ARG, "grouped_partial_reduce: null table");
  OVQA_REQUIRE(n_problems <= 65535, OVQA_ERR_BAD_ARG, "grouped_partial_reduce: at most 65535 problems per launch");
  return ovqa::grouped_partial_reduce(problems, n_problems, max_blocks, max_D, as_stream(stream));
}

int ovqa_attention_fwd(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       const float* mask, int64_t msb, int64_t msh, int64_t msq, void* o, int64_t ldo, float* lse,
                       void* att, void* o_lo, int64_t B, int64_t H, int64_t nq, int64_t nk, int64_t dk, int64_t dv,
                       float scale, const ovqa_dropout* att_drop, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "attention_fwd: bad dtype %d", dtype);
  OVQA_REQUIRE(B >= 0 && H > 0 && nq >= 0 && nk >= 0 && dk > 0 && dv > 0, OVQA_ERR_BAD_ARG, "attention_fwd: bad sizes");
  if (B == 0 || nq == 0) return OVQA_OK;
  OVQA_REQUIRE(q && k && v && o, OVQA_ERR_BAD_ARG, "attention_fwd: null pointer");
  OVQA_REQUIRE(ldq >= H * dk && ldk >= H * dk && ldv >= H * dv && ldo >= H * dv, OVQA_ERR_BAD_ARG,
               "attention_fwd: row stride smaller than H*d");
  OVQA_REQUIRE(B * H <= 0x7fffffff, OVQA_ERR_UNSUPPORTED, "attention_fwd: B*H too large");
  OVQA_REQUIRE(B * H * nq * nk < (1ll << 32) || !att_drop || att_drop->p <= 0.f, OVQA_ERR_UNSUPPORTED,
               "attention_fwd: dropout on more than 2^32 probabilities");
  ovqa::AttnArgs a{q, k, v, ldq, ldk, ldv, mask, msb, msh, msq, o, ldo, lse, att,
                   (int)B, (int)H, (int)nq, (int)nk, (int)dk, (int)dv, scale, make_drop_args(att_drop)};
  a.o_lo = dtype == OVQA_BF16 ? o_lo : nullptr;
  if (dtype == OVQA_BF16 && !force_simple() && ovqa::mfma_attention_supported(a)) {
    g_dispatch = "mfma";
    return ovqa::mfma_attention_fwd(a, as_stream(stream));
  }
  OVQA_FALLBACK("attention_fwd");
  return ovqa::simple_attention_fwd(dtype, a, as_stream(stream));
}

int ovqa_attention_fwd_prefix_lm(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                                 int64_t ldv, const float* key_mask, int64_t msb, int64_t msh, int64_t causal_tail, void* o,
                                 int64_t ldo, float* lse, int64_t B, int64_t H, int64_t n, int64_t d, float scale,
                                 void* stream) {
  OVQA_REQUIRE(dtype == OVQA_BF16, OVQA_ERR_UNSUPPORTED, "attention_fwd_prefix_lm: bf16 only");
  OVQA_REQUIRE(B >= 0 && H > 0 && n >= 0 && d > 0 && causal_tail >= 0 && causal_tail <= n, OVQA_ERR_BAD_ARG,
               "attention_fwd_prefix_lm: bad sizes");
  if (B == 0 || n == 0) return OVQA_OK;
  OVQA_REQUIRE(q && k && v && o, OVQA_ERR_BAD_ARG, "attention_fwd_prefix_lm: null pointer");
  OVQA_REQUIRE(ldq >= H * d && ldk >= H * d && ldv >= H * d && ldo >= H * d, OVQA_ERR_BAD_ARG,
               "attention_fwd_prefix_lm: row stride smaller than H*d");
  OVQA_REQUIRE(B * H <= 0x7fffffff, OVQA_ERR_UNSUPPORTED, "attention_fwd_prefix_lm: B*H too large");
  ovqa::AttnArgs a{q, k, v, ldq, ldk, ldv, key_mask, msb, msh, 0, o, ldo, lse, nullptr,
                   (int)B, (int)H, (int)n, (int)n, (int)d, (int)d, scale, make_drop_args(nullptr)};
  a.tail = (int)causal_tail;
  OVQA_REQUIRE(!force_simple() && ovqa::mfma_attention_supported(a), OVQA_ERR_UNSUPPORTED,
               "attention_fwd_prefix_lm: shape not covered by the MFMA forward kernel (d in {64, 96, 128}, n <= 256 / 192, "
               "16-byte aligned rows): pass the dense (B, 1, n, n) mask to ovqa_attention_fwd instead");
  g_dispatch = "mfma";
  return ovqa::mfma_attention_fwd(a, as_stream(stream));
}

int ovqa_attention_qkv_fwd(int dtype, const void* x, int64_t ldx, const void* w, const float* bias, void* qkv,
                           int64_t ldqkv, const float* mask, int64_t msb, int64_t msh, void* o, int64_t ldo, float* lse,
                           void* o_lo, int64_t B, int64_t H, int64_t n, int64_t d_model, int64_t d, float scale,
                           void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "attention_qkv_fwd: bad dtype %d", dtype);
  OVQA_REQUIRE(B >= 0 && H > 0 && n >= 0 && d > 0 && d_model > 0, OVQA_ERR_BAD_ARG, "attention_qkv_fwd: bad sizes");
  if (B == 0 || n == 0) return OVQA_OK;
  OVQA_REQUIRE(x && w && qkv && o, OVQA_ERR_BAD_ARG, "attention_qkv_fwd: null pointer");
  OVQA_REQUIRE(ldx >= d_model && ldqkv >= 3 * H * d && ldo >= H * d, OVQA_ERR_BAD_ARG,
               "attention_qkv_fwd: row stride smaller than the row");
  OVQA_REQUIRE(B * H <= 0x7fffffff && B * n <= 0x7fffffff, OVQA_ERR_UNSUPPORTED, "attention_qkv_fwd: batch too large");
  const size_t es = dtype == OVQA_BF16 ? 2 : 4;
  const char* qp = (const char*)qkv;
  ovqa::AttnArgs a{qp, qp + (size_t)(H * d) * es, qp + (size_t)(2 * H * d) * es, ldqkv, ldqkv, ldqkv, mask, msb, msh, 0,
                   o, ldo, lse, nullptr, (int)B, (int)H, (int)n, (int)n, (int)d, (int)d, scale, make_drop_args(nullptr)};
  a.o_lo = dtype == OVQA_BF16 ? o_lo : nullptr;
  if (dtype == OVQA_BF16 && !force_simple() && !no_fused_qkv() &&
      ovqa::mfma_attention_qkv_supported(a, d_model, ldx, ldqkv, x, w, qkv)) {
    g_dispatch = "mfma-fused";
    return ovqa::mfma_attention_qkv_fwd(a, x, ldx, w, bias, qkv, ldqkv, d_model, as_stream(stream));
  }
  int rc = ovqa_linear_fwd(dtype, OVQA_EPI_BIAS, x, ldx, w, bias, nullptr, 0, qkv, ldqkv, nullptr, B * n, 3 * H * d, d_model,
                           nullptr, stream);
  if (rc != OVQA_OK) return rc;
  return ovqa_attention_fwd(dtype, a.q, ldqkv, a.k, ldqkv, a.v, ldqkv, mask, msb, msh, 0, o, ldo, lse, nullptr, o_lo, B, H,
                            n, n, d, d, scale, nullptr, stream);
}

int ovqa_attention_q_fwd(int dtype, const void* x, int64_t ldx, const void* w, const float* bias, void* q, int64_t ldq,
                         const void* k, int64_t ldk, const void* v, int64_t ldv, const float* mask, int64_t msb, int64_t msh,
                         void* o, int64_t ldo, float* lse, void* o_lo, int64_t B, int64_t H, int64_t nq, int64_t nk,
                         int64_t d_model, int64_t d, float scale, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "attention_q_fwd: bad dtype %d", dtype);
  OVQA_REQUIRE(B >= 0 && H > 0 && nq >= 0 && nk >= 1 && d > 0 && d_model > 0, OVQA_ERR_BAD_ARG, "attention_q_fwd: bad sizes");
  if (B == 0 || nq == 0) return OVQA_OK;
  OVQA_REQUIRE(x && w && q && k && v && o, OVQA_ERR_BAD_ARG, "attention_q_fwd: null pointer");
  OVQA_REQUIRE(ldx >= d_model && ldq >= H * d && ldk >= H * d && ldv >= H * d && ldo >= H * d, OVQA_ERR_BAD_ARG,
               "attention_q_fwd: row stride smaller than the row");
  OVQA_REQUIRE(B * H <= 0x7fffffff && B * nq <= 0x7fffffff, OVQA_ERR_UNSUPPORTED, "attention_q_fwd: batch too large");
  ovqa::AttnArgs a{q, k, v, ldq, ldk, ldv, mask, msb, msh, 0, o, ldo, lse, nullptr,
                   (int)B, (int)H, (int)nq, (int)nk, (int)d, (int)d, scale, make_drop_args(nullptr)};
  a.o_lo = dtype == OVQA_BF16 ? o_lo : nullptr;
  if (dtype == OVQA_BF16 && !force_simple() && !no_fused_qkv() &&
      ovqa::mfma_attention_q_supported(a, d_model, ldx, ldq, x, w, q)) {
    g_dispatch = "mfma-fused";
    return ovqa::mfma_attention_q_fwd(a, x, ldx, w, bias, q, ldq, d_model, as_stream(stream));
  }
  int rc = ovqa_linear_fwd(dtype, OVQA_EPI_BIAS, x, ldx, w, bias, nullptr, 0, q, ldq, nullptr, B * nq, H * d, d_model,
                           nullptr, stream);
  if (rc != OVQA_OK) return rc;
  return ovqa_attention_fwd(dtype, q, ldq, k, ldk, v, ldv, mask, msb, msh, 0, o, ldo, lse, nullptr, o_lo, B, H, nq, nk, d,
                            d, scale, nullptr, stream);
}

int ovqa_attention_decode(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                          int64_t kv_batch_stride, int64_t group, const float* mask, int64_t ldmask, void* o, int64_t ldo,
                          int64_t R, int64_t H, int64_t n, int64_t d, float scale, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "attention_decode: bad dtype %d", dtype);
  OVQA_REQUIRE(R >= 0 && H > 0 && n >= 1 && d > 0 && group >= 1, OVQA_ERR_BAD_ARG, "attention_decode: bad sizes");
  if (R == 0) return OVQA_OK;
  OVQA_REQUIRE(q && k && v && o, OVQA_ERR_BAD_ARG, "attention_decode: null pointer");
  OVQA_REQUIRE(ldq >= H * d && ldk >= H * d && ldv >= H * d && ldo >= H * d, OVQA_ERR_BAD_ARG,
               "attention_decode: row stride smaller than H*d");
  OVQA_REQUIRE(R * H <= 0x7fffffff, OVQA_ERR_UNSUPPORTED, "attention_decode: R*H too large");
  ovqa::AttnDecodeArgs a{q, k, v, ldq, ldk, ldv, kv_batch_stride, mask, ldmask, o, ldo,
                         (int)R, (int)H, (int)d, (int)n, (int)group, scale};
  OVQA_REQUIRE(ovqa::attention_decode_supported(a, dtype == OVQA_BF16 ? 2 : 4), OVQA_ERR_UNSUPPORTED,
               "attention_decode: d must be 32 / 64 / 128, n <= 512, 16-byte aligned q / k / v rows");
  g_dispatch = "decode";
  return ovqa::attention_decode(dtype, a, as_stream(stream));
}

int ovqa_decode_embed(int out_dtype, const int64_t* tokens, const float* emb, int64_t ld_emb, int64_t vocab,
                      const float* pos, int64_t ld_pos, int64_t n_pos, int64_t* seq, int64_t pad_idx, float mask_value,
                      float* mask, int64_t ld_mask, int64_t col, float* x32, void* x, int64_t R, int64_t D, void* stream) {
  OVQA_REQUIRE(dtype_ok(out_dtype), OVQA_ERR_BAD_ARG, "decode_embed: bad dtype %d", out_dtype);
  OVQA_REQUIRE(R >= 0 && D > 0 && D % 4 == 0 && vocab > 0 && n_pos > 0, OVQA_ERR_BAD_ARG, "decode_embed: bad sizes");
  if (R == 0) return OVQA_OK;
  OVQA_REQUIRE(tokens && emb && pos && seq && (x32 || x), OVQA_ERR_BAD_ARG, "decode_embed: null pointer");
  OVQA_REQUIRE(ld_emb >= D && ld_pos >= D && ld_emb % 4 == 0 && ld_pos % 4 == 0 && (!mask || (ld_mask > col && col >= 0)),
               OVQA_ERR_BAD_ARG, "decode_embed: row strides");
  OVQA_REQUIRE((((uintptr_t)emb | (uintptr_t)pos | (uintptr_t)x32) & 15) == 0, OVQA_ERR_UNSUPPORTED,
               "decode_embed: tables and x32 must be 16-byte aligned");
  return ovqa::decode_embed(out_dtype, tokens, emb, ld_emb, vocab, pos, ld_pos, n_pos, seq, pad_idx, mask_value, mask,
                            ld_mask, col, x32, x, R, D, as_stream(stream));
}

int ovqa_beam_candidates(int dtype, const void* logits, int64_t ld, int64_t R, int64_t V, int64_t k,
                         const float* seq_logprob, float* seq_mask, const int64_t* prev_words, int64_t eos, float* vals,
                         int64_t* idx, float* wl, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "beam_candidates: bad dtype %d", dtype);
  OVQA_REQUIRE(R >= 0 && V >= 1 && k >= 1 && k <= 8 && k <= V && ld >= V && R <= 0x7fffffff && V <= 0x7ffffffe,
               OVQA_ERR_BAD_ARG, "beam_candidates: bad sizes");
  if (R == 0) return OVQA_OK;
  OVQA_REQUIRE(logits && seq_logprob && seq_mask && vals && idx && wl, OVQA_ERR_BAD_ARG, "beam_candidates: null pointer");
  return ovqa::beam_candidates(dtype, logits, ld, R, V, (int)k, seq_logprob, seq_mask, prev_words, eos, vals, idx, wl,
                               as_stream(stream));
}

int ovqa_beam_commit(const float* vals, const int64_t* idx, const float* wl, const float* seq_mask_in,
                     const int64_t* out_in, const float* lp_in, int64_t* out_out, float* lp_out, float* seq_logprob_out,
                     float* seq_mask_out, int32_t* selected_beam, int64_t* words, int64_t b_s, int64_t cur, int64_t k,
                     int64_t beam, int64_t t, int64_t T, void* stream) {
  OVQA_REQUIRE(b_s >= 0 && cur >= 1 && k >= 1 && beam >= 1 && beam <= 8 && cur * k <= 64 && beam <= cur * k && t >= 0 &&
                   t < T, OVQA_ERR_BAD_ARG, "beam_commit: bad sizes (beam <= 8, cur * k <= 64, t < T)");
  if (b_s == 0) return OVQA_OK;
  OVQA_REQUIRE(vals && idx && wl && seq_mask_in && out_out && lp_out && seq_logprob_out && seq_mask_out && selected_beam &&
                   words && (t == 0 || (out_in && lp_in)), OVQA_ERR_BAD_ARG, "beam_commit: null pointer");
  OVQA_REQUIRE(out_in != out_out && lp_in != lp_out, OVQA_ERR_BAD_ARG, "beam_commit: the histories are double-buffered");
  ovqa::BeamCommitArgs a{vals, idx, wl, seq_mask_in, out_in, lp_in, out_out, lp_out, seq_logprob_out, seq_mask_out,
                         selected_beam, words, (int)cur, (int)k, (int)beam, (int)t, (int)T};
  return ovqa::beam_commit(a, b_s, as_stream(stream));
}

int ovqa_topk_rows(const float* x, int64_t ldx, int64_t R, int64_t V, int64_t k, float* vals, int64_t* idx, void* stream) {
  OVQA_REQUIRE(R >= 0 && V >= 1 && k >= 1 && k <= 8 && k <= V && ldx >= V, OVQA_ERR_BAD_ARG, "topk_rows: bad sizes");
  if (R == 0) return OVQA_OK;
  OVQA_REQUIRE(x && vals && idx, OVQA_ERR_BAD_ARG, "topk_rows: null pointer");
  return ovqa::topk_rows(x, ldx, R, V, (int)k, vals, idx, as_stream(stream));
}

int ovqa_attention_bwd(int dtype, const void* d_o, int64_t lddo, const void* q, int64_t ldq, const void* k, int64_t ldk,
                       const void* v, int64_t ldv, const void* o, int64_t ldo, const void* o_lo, const void* d_att,
                       const float* lse,
                       const float* mask, int64_t msb, int64_t msh, int64_t msq, void* dq, int64_t lddq, void* dk_,
                       int64_t lddk,
                       void* dv_, int64_t lddv, float* delta, const float* d_lse, int64_t B, int64_t H, int64_t nq,
                       int64_t nk, int64_t dk, int64_t dv, float scale, const ovqa_dropout* att_drop, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "attention_bwd: bad dtype %d", dtype);
  OVQA_REQUIRE(B >= 0 && H > 0 && nq >= 0 && nk >= 0 && dk > 0 && dv > 0, OVQA_ERR_BAD_ARG, "attention_bwd: bad sizes");
  if (B == 0 || nq == 0) return OVQA_OK;
  OVQA_REQUIRE(d_o && q && k && v && o && dq && dk_ && dv_, OVQA_ERR_BAD_ARG, "attention_bwd: null pointer");
  OVQA_REQUIRE(B * H <= 0x7fffffff, OVQA_ERR_UNSUPPORTED, "attention_bwd: B*H too large");
  ovqa::AttnBwdArgs a{d_o, q, k, v, o, d_att, lddo, ldq, ldk, ldv, ldo, lse, mask, msb, msh, msq, dq, dk_, dv_,
                      lddq, lddk, lddv, delta, (int)B, (int)H, (int)nq, (int)nk, (int)dk, (int)dv, scale,
                      make_drop_args(att_drop), d_lse};
  a.o_lo = dtype == OVQA_BF16 ? o_lo : nullptr;
  if (dtype == OVQA_BF16 && !force_simple() && ovqa::mfma_attention_bwd_supported(a)) {
    g_dispatch = "mfma";
    return ovqa::mfma_attention_bwd(a, as_stream(stream));
  }
  OVQA_FALLBACK("attention_bwd");
  return ovqa::simple_attention_bwd(dtype, a, as_stream(stream));
}

int ovqa_attention_bwd_do(int dtype, const void* dy, int64_t lddy, const void* wt, int64_t ldwt, void* d_o_scratch,
                          int64_t lddo, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                          const void* o, int64_t ldo, const void* o_lo, const float* lse, const float* mask, int64_t msb,
                          int64_t msh, void* dq, int64_t lddq, void* dk_, int64_t lddk, void* dv_, int64_t lddv, float* delta,
                          int64_t B, int64_t H, int64_t nq, int64_t nk, int64_t d_model, int64_t d, float scale,
                          void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "attention_bwd_do: bad dtype %d", dtype);
  OVQA_REQUIRE(B >= 0 && H > 0 && nq >= 0 && nk >= 0 && d > 0 && d_model > 0, OVQA_ERR_BAD_ARG, "attention_bwd_do: bad sizes");
  if (B == 0 || nq == 0) return OVQA_OK;
  OVQA_REQUIRE(dy && wt && q && k && v && o && dq && dk_ && dv_ && lse, OVQA_ERR_BAD_ARG, "attention_bwd_do: null pointer");
  OVQA_REQUIRE(lddy >= d_model && ldwt >= d_model, OVQA_ERR_BAD_ARG, "attention_bwd_do: row stride smaller than the row");
  OVQA_REQUIRE(B * H <= 0x7fffffff, OVQA_ERR_UNSUPPORTED, "attention_bwd_do: B*H too large");
  ovqa::AttnBwdArgs a{d_o_scratch, q, k, v, o, nullptr, lddo, ldq, ldk, ldv, ldo, lse, mask, msb, msh, 0, dq, dk_, dv_,
                      lddq, lddk, lddv, delta, (int)B, (int)H, (int)nq, (int)nk, (int)d, (int)d, scale,
                      make_drop_args(nullptr), nullptr};
  a.o_lo = dtype == OVQA_BF16 ? o_lo : nullptr;
  if (dtype == OVQA_BF16 && !force_simple() && !no_fused_qkv() &&
      ovqa::mfma_attention_bwd_do_supported(a, d_model, lddy, ldwt, dy, wt)) {
    g_dispatch = "mfma-fused";
    return ovqa::mfma_attention_bwd_do(a, dy, lddy, wt, ldwt, d_model, as_stream(stream));
  }
  OVQA_REQUIRE(d_o_scratch != nullptr, OVQA_ERR_UNSUPPORTED,
               "attention_bwd_do: shape not covered by the fused kernel and no dO buffer for the two-kernel form");
  int rc = ovqa_linear_bwd_data_wt(dtype, dy, lddy, wt, ldwt, d_o_scratch, lddo, nullptr, nullptr, 0, B * nq, d_model, H * d,
                                   nullptr, stream);
  if (rc != OVQA_OK) return rc;
  return ovqa_attention_bwd(dtype, d_o_scratch, lddo, q, ldq, k, ldk, v, ldv, o, ldo, o_lo, nullptr, lse, mask, msb, msh, 0,
                            dq, lddq, dk_, lddk, dv_, lddv, delta, nullptr, B, H, nq, nk, d, d, scale, nullptr, stream);
}

int ovqa_pointer_score(int dtype, const void* q, const void* k, const float* add_mask, const uint8_t* key_fill,
                       const uint8_t* query_fill, float* scores, int64_t B, int64_t T, int64_t Nk, int64_t D,
                       float scale, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "pointer_score: bad dtype %d", dtype);
  OVQA_REQUIRE(B >= 0 && T >= 0 && Nk >= 0 && D > 0, OVQA_ERR_BAD_ARG, "pointer_score: bad sizes");
  if (B == 0 || T == 0 || Nk == 0) return OVQA_OK;
  OVQA_REQUIRE(q && k && scores, OVQA_ERR_BAD_ARG, "pointer_score: null pointer");
  if (dtype == OVQA_BF16 && !force_simple() && ovqa::mfma_batched_nt_supported(q, D, T * D, k, D, Nk * D, B, T, Nk, D)) {
    g_dispatch = "mfma";
    return ovqa::mfma_pointer_score(q, k, add_mask, key_fill, query_fill, scores, B, T, Nk, D, scale, as_stream(stream));
  }
  OVQA_FALLBACK("pointer_score");
  return ovqa::simple_pointer_score(dtype, q, k, add_mask, key_fill, query_fill, scores, B, T, Nk, D, scale,
                                    as_stream(stream));
}

int ovqa_batched_gemm(int dtype, int c_dtype, int trans_a, int trans_b, const void* A, int64_t lda, int64_t stride_a,
                      const void* Bm, int64_t ldb, int64_t stride_b, void* C, int64_t ldc, int64_t stride_c,
                      int64_t batch, int64_t M, int64_t N, int64_t K, float alpha, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype) && dtype_ok(c_dtype), OVQA_ERR_BAD_ARG, "batched_gemm: bad dtype");
  OVQA_REQUIRE(batch >= 0 && M >= 0 && N >= 0 && K >= 0, OVQA_ERR_BAD_ARG, "batched_gemm: bad sizes");
  if (batch == 0 || M == 0 || N == 0) return OVQA_OK;
  OVQA_REQUIRE(A && Bm && C, OVQA_ERR_BAD_ARG, "batched_gemm: null pointer");
  if (dtype == OVQA_BF16 && !trans_a && trans_b && !force_simple() &&
      ovqa::mfma_batched_nt_supported(A, lda, stride_a, Bm, ldb, stride_b, batch, M, N, K)) {
    g_dispatch = "mfma";
    return ovqa::mfma_batched_nt(c_dtype, A, lda, stride_a, Bm, ldb, stride_b, C, ldc, stride_c, batch, M, N, K, alpha,
                                 as_stream(stream));
  }
  g_dispatch = "simple";  // (NN / TN forms and fp32: the gradient products of the pointer scorers, 12 x 50 x 768 each)
  return ovqa::simple_batched_gemm(dtype, c_dtype, trans_a, trans_b, A, lda, stride_a, Bm, ldb, stride_b, C, ldc,
                                   stride_c, batch, M, N, K, alpha, as_stream(stream));
}

int ovqa_adam_step(float* param, const void* grad, int grad_dtype, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                   int64_t n, float lr, const float* lr_scale_ptr, float beta1, float beta2, float eps,
                   float weight_decay, float grad_scale, const uint32_t* step_ptr, void* stream) {
  OVQA_REQUIRE(n >= 0 && dtype_ok(grad_dtype), OVQA_ERR_BAD_ARG, "adam_step: bad n or gradient dtype");
  OVQA_REQUIRE(n == 0 || (param && grad && exp_avg && exp_avg_sq), OVQA_ERR_BAD_ARG, "adam_step: null pointer");
  return ovqa::adam_step(param, grad, grad_dtype, exp_avg, exp_avg_sq, shadow_bf16, n, lr, lr_scale_ptr, beta1, beta2, eps,
                         weight_decay, grad_scale, step_ptr, as_stream(stream));
}

int ovqa_adam_step_tiled(float* param, const void* grad, int grad_dtype, float* exp_avg, float* exp_avg_sq,
                         void* shadow_bf16, void* shadow_t_bf16, const ovqa_adam_tile* tiles, int32_t n_tiles,
                         int64_t flat_lo, int64_t flat_hi, float lr, const float* lr_scale_ptr, float beta1, float beta2,
                         float eps, float weight_decay, float grad_scale, const uint32_t* step_ptr, void* stream) {
  OVQA_REQUIRE(dtype_ok(grad_dtype) && n_tiles >= 0 && flat_lo >= 0 && flat_hi >= flat_lo, OVQA_ERR_BAD_ARG,
               "adam_step_tiled: bad argument");
  OVQA_REQUIRE(param && grad && exp_avg && exp_avg_sq && (n_tiles == 0 || tiles), OVQA_ERR_BAD_ARG,
               "adam_step_tiled: null pointer");
  OVQA_REQUIRE(flat_lo % 4 == 0 && flat_hi % 4 == 0, OVQA_ERR_BAD_ARG, "adam_step_tiled: flat range must be 4-aligned");
  OVQA_REQUIRE(((uintptr_t)param % 16 == 0) && ((uintptr_t)grad % 8 == 0) && ((uintptr_t)exp_avg % 16 == 0) &&
                   ((uintptr_t)exp_avg_sq % 16 == 0) && ((uintptr_t)shadow_bf16 % 8 == 0) &&
                   ((uintptr_t)shadow_t_bf16 % 16 == 0),
               OVQA_ERR_BAD_ARG, "adam_step_tiled: arena pointers must be 16-byte aligned");
  return ovqa::adam_step_tiled(param, grad, grad_dtype, exp_avg, exp_avg_sq, shadow_bf16, shadow_t_bf16, tiles, n_tiles,
                               flat_lo, flat_hi, lr, lr_scale_ptr, beta1, beta2, eps, weight_decay, grad_scale, step_ptr,
                               as_stream(stream));
}

int ovqa_increment_step(uint32_t* step_ptr, void* stream) {
  OVQA_REQUIRE(step_ptr, OVQA_ERR_BAD_ARG, "increment_step: null pointer");
  return ovqa::increment_step(step_ptr, nullptr, as_stream(stream));
}

int ovqa_increment_steps(uint32_t* a, uint32_t* b, void* stream) {
  OVQA_REQUIRE(a && a != b, OVQA_ERR_BAD_ARG, "increment_steps: first counter is NULL or the two are the same");
  return ovqa::increment_step(a, b, as_stream(stream));
}

int ovqa_begin_step(uint32_t* step_ptr, uint32_t* second, const float* lr_table, int32_t n_table, float* lr_out,
                    void* stream) {
  OVQA_REQUIRE(step_ptr && step_ptr != second, OVQA_ERR_BAD_ARG, "begin_step: the step counter is NULL or given twice");
  OVQA_REQUIRE((lr_table == nullptr) == (lr_out == nullptr) && (lr_table == nullptr || n_table > 0), OVQA_ERR_BAD_ARG,
               "begin_step: lr_table and lr_out go together, with n_table > 0");
  return ovqa::begin_step(step_ptr, second, lr_table, (uint32_t)n_table, lr_out, as_stream(stream));
}

int ovqa_cast(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, void* stream) {
  OVQA_REQUIRE(n >= 0, OVQA_ERR_BAD_ARG, "cast: bad n");
  OVQA_REQUIRE(n == 0 || (src && dst), OVQA_ERR_BAD_ARG, "cast: null pointer");
  return ovqa::cast(src_dtype, dst_dtype, src, dst, n, as_stream(stream));
}

int ovqa_gelu_bwd(int dtype, const void* dy, const void* u, void* du, int64_t n, const ovqa_dropout* drop,
                  void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "gelu_bwd: bad dtype");
  OVQA_REQUIRE(n >= 0 && (n == 0 || (dy && u && du)), OVQA_ERR_BAD_ARG, "gelu_bwd: null pointer or bad n");
  return ovqa::gelu_bwd(dtype, dy, u, du, n, make_drop_args(drop), as_stream(stream));
}

int ovqa_row_padding_mask(int dtype, const void* x, float* mask, int64_t M, int64_t D, float pad_value,
                          void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "row_padding_mask: bad dtype");
  OVQA_REQUIRE(M >= 0 && D > 0 && D < (1 << 30) && (M == 0 || (x && mask)), OVQA_ERR_BAD_ARG,
               "row_padding_mask: bad sizes or null pointer");
  return ovqa::row_padding_mask(dtype, x, mask, M, D, pad_value, as_stream(stream));
}

int ovqa_grouped_row_gather(const ovqa_gather_problem* problems, int32_t n_problems, const int32_t* sel, int32_t b_s,
                            int32_t cur_beam, int32_t beam, void* stream) {
  OVQA_REQUIRE(n_problems >= 0 && n_problems <= 65535 && b_s >= 0 && cur_beam >= 1 && beam >= 1, OVQA_ERR_BAD_ARG,
               "grouped_row_gather: bad sizes");
  OVQA_REQUIRE(n_problems == 0 || b_s == 0 || (problems && sel), OVQA_ERR_BAD_ARG, "grouped_row_gather: null pointer");
  OVQA_REQUIRE((int64_t)b_s * beam < (1ll << 31), OVQA_ERR_UNSUPPORTED, "grouped_row_gather: too many rows");
  g_dispatch = "";
  return ovqa::grouped_row_gather(problems, n_problems, sel, b_s, cur_beam, beam, as_stream(stream));
}

int ovqa_dropout_keep_mask(const ovqa_dropout* drop, uint8_t* out, int64_t n, void* stream) {
  OVQA_REQUIRE(drop && out && n >= 0, OVQA_ERR_BAD_ARG, "dropout_keep_mask: bad argument");
  OVQA_REQUIRE(n < (1ll << 32), OVQA_ERR_UNSUPPORTED, "dropout_keep_mask: more than 2^32 elements");
  return ovqa::dropout_keep_mask(make_drop_args(drop), out, n, as_stream(stream));
}

int ovqa_sq_loss_fwd_bwd(int dtype, const void* x, const void* target, void* dx, float* loss, int64_t n,
                         int accumulate_loss, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "sq_loss: bad dtype");
  OVQA_REQUIRE(x && loss, OVQA_ERR_BAD_ARG, "sq_loss: null pointer");
  return ovqa::sq_loss_fwd_bwd(dtype, x, target, dx, loss, n, accumulate_loss, as_stream(stream));
}

int64_t ovqa_lstm_saved_bytes(int64_t B, int64_t T, int64_t H) { return ovqa::lstm_saved_bytes(B, T, H); }
int64_t ovqa_lstm_scratch_bytes(int64_t B, int64_t T, int64_t H) { return ovqa::lstm_scratch_bytes(B, T, H); }
int64_t ovqa_lstm_persistent_max_batch(void) { return force_simple() ? 0 : ovqa::lstm_persistent_max_batch(); }
int ovqa_lstm_status(uint32_t* status, void* stream) {
  OVQA_REQUIRE(status != nullptr, OVQA_ERR_BAD_ARG, "lstm_status: null pointer");
  unsigned v = 0;
  const int rc = ovqa::lstm_status_read_clear(&v, as_stream(stream));
  *status = v;
  return rc;
}

// forward and backward of one sequence must take the same route: the decision uses dtype and sizes only
static bool lstm_route_persistent(int dtype, int64_t B, int64_t T, int64_t I, int64_t H) {
  return !force_simple() && ovqa::lstm_persistent_supported(dtype, B, T, I, H, 8);
}

int ovqa_lstm_fwd(int dtype, const void* x, int64_t ldx, const void* w_ih, const void* w_hh, const float* b_ih,
                  const float* b_hh, float* y, void* y_lp, void* hseq, void* saved, void* scratch, int64_t B, int64_t T,
                  int64_t I, int64_t H, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "lstm_fwd: bad dtype");
  OVQA_REQUIRE(B >= 0 && T >= 0 && I >= 1 && H >= 1 && ldx >= I && B * H < (1ll << 31), OVQA_ERR_BAD_ARG,
               "lstm_fwd: bad size");
  if (B == 0 || T == 0) return OVQA_OK;  // an empty batch / sequence: nothing to write (pointers may be null)
  OVQA_REQUIRE(x && w_ih && w_hh && b_ih && b_hh && y && hseq && saved && scratch, OVQA_ERR_BAD_ARG,
               "lstm_fwd: null pointer");
  const bool persistent = lstm_route_persistent(dtype, B, T, I, H);
  if (persistent)
    OVQA_REQUIRE(ldx % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)w_ih % 16 == 0 && (uintptr_t)w_hh % 16 == 0 &&
                     (uintptr_t)hseq % 16 == 0 && (uintptr_t)y % 16 == 0 && (uintptr_t)scratch % 16 == 0,
                 OVQA_ERR_BAD_ARG, "lstm_fwd(bf16, H = 512): x rows, weights, hseq, y and scratch must be 16-byte aligned");
  else if (dtype == OVQA_BF16 && require_mfma() && !force_simple()) {
    ovqa_set_error("lstm_fwd: shape not covered by the persistent MFMA kernel and OVQA_REQUIRE_MFMA=1");
    return OVQA_ERR_UNSUPPORTED;
  }
  g_dispatch = persistent ? "mfma" : "simple";
  OVQA_REQUIRE(y_lp == nullptr || (uintptr_t)y_lp % 16 == 0, OVQA_ERR_BAD_ARG, "lstm_fwd: y_lp must be 16-byte aligned");
  return ovqa::lstm_fwd(dtype, persistent, x, ldx, w_ih, w_hh, b_ih, b_hh, y, y_lp, hseq, saved, scratch, B, T, I, H,
                        as_stream(stream));
}

int ovqa_lstm_bwd(int dtype, const void* dy, int dy_dtype, const void* w_hh, const void* w_hh_t, int64_t ldwt,
                  const void* saved, void* dgates, void* scratch, int64_t B, int64_t T, int64_t I, int64_t H, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype) && dtype_ok(dy_dtype), OVQA_ERR_BAD_ARG, "lstm_bwd: bad dtype");
  OVQA_REQUIRE(B >= 0 && T >= 0 && H >= 1 && B * H < (1ll << 31), OVQA_ERR_BAD_ARG, "lstm_bwd: bad size");
  if (B == 0 || T == 0) return OVQA_OK;
  OVQA_REQUIRE(dy && w_hh && saved && dgates && scratch, OVQA_ERR_BAD_ARG, "lstm_bwd: null pointer");
  const bool persistent = lstm_route_persistent(dtype, B, T, I, H);
  if (persistent)
    OVQA_REQUIRE(w_hh_t && ldwt >= 4 * H && ldwt % 8 == 0 && (uintptr_t)w_hh_t % 16 == 0 && (uintptr_t)dgates % 16 == 0 &&
                     (uintptr_t)scratch % 16 == 0,
                 OVQA_ERR_BAD_ARG, "lstm_bwd(bf16, H = 512): needs the transposed w_hh copy (16-byte aligned rows)");
  g_dispatch = persistent ? "mfma" : "simple";
  return ovqa::lstm_bwd(dtype, persistent, dy, dy_dtype == OVQA_BF16 ? 1 : 0, w_hh, w_hh_t, ldwt, saved, dgates, scratch, B, T,
                        H, as_stream(stream));
}

int ovqa_embed_gather(int dtype, const int64_t* tokens, const void* table, int64_t ld_table, int64_t vocab, void* out,
                      int64_t ld_out, int64_t B, int64_t T, int64_t width, int time_major, float* mask, int64_t padding_idx,
                      void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "embed_gather: bad dtype");
  OVQA_REQUIRE(B >= 0 && T >= 0 && vocab >= 1 && width >= 1 && ld_table >= width && ld_out >= width && B * T < (1ll << 31),
               OVQA_ERR_BAD_ARG, "embed_gather: bad size");
  if (B == 0 || T == 0) return OVQA_OK;
  OVQA_REQUIRE(tokens && table && out, OVQA_ERR_BAD_ARG, "embed_gather: null pointer");
  g_dispatch = "stream";
  return ovqa::embed_gather(dtype, tokens, table, ld_table, vocab, out, ld_out, B, T, width, time_major, mask, padding_idx,
                            as_stream(stream));
}

int ovqa_decoder_inputs(const int64_t* tokens, const float* emb, const float* pos_table, int64_t pos_rows, float* out,
                        float* self_mask, int64_t B, int64_t T, int64_t D, int64_t padding_idx, void* stream) {
  OVQA_REQUIRE(B >= 0 && T >= 0 && D >= 1 && B * T < (1ll << 31), OVQA_ERR_BAD_ARG, "decoder_inputs: bad size");
  OVQA_REQUIRE(pos_rows >= T + 1, OVQA_ERR_BAD_ARG, "decoder_inputs: the position table has %lld rows, %lld needed",
               (long long)pos_rows, (long long)T + 1);
  if (B == 0 || T == 0) return OVQA_OK;
  OVQA_REQUIRE(tokens && emb && pos_table && out && self_mask, OVQA_ERR_BAD_ARG, "decoder_inputs: null pointer");
  g_dispatch = "stream";
  return ovqa::decoder_inputs(tokens, emb, pos_table, out, self_mask, B, T, D, padding_idx, as_stream(stream));
}

int ovqa_embed_scatter(int dtype, const int64_t* tokens, const void* drows, int64_t ld_rows, float* dtable, int64_t ld_table,
                       int64_t rows_table, int64_t B, int64_t T, int64_t width, int time_major, int64_t padding_idx,
                       int accumulate, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "embed_scatter: bad dtype");
  OVQA_REQUIRE(dtable && (B == 0 || T == 0 || (tokens && drows)), OVQA_ERR_BAD_ARG, "embed_scatter: null pointer");
  OVQA_REQUIRE(B >= 0 && T >= 0 && rows_table >= 1 && width >= 1 && ld_table >= width && ld_rows >= width &&
                   B * T < (1ll << 31) && rows_table < (1ll << 31),
               OVQA_ERR_BAD_ARG, "embed_scatter: bad size");
  if (B == 0 || T == 0) {  // no tokens: the gradient of the table is zero (or stays what it was)
    if (!accumulate) {
      hipError_t e = hipMemset2DAsync(dtable, (size_t)ld_table * 4, 0, (size_t)width * 4, (size_t)rows_table, as_stream(stream));
      if (e != hipSuccess) {
        ovqa_set_error("embed_scatter: hipMemset2DAsync: %s", hipGetErrorString(e));
        return OVQA_ERR_LAUNCH;
      }
    }
    return OVQA_OK;
  }
  g_dispatch = "stream";
  return ovqa::embed_scatter(dtype, tokens, drows, ld_rows, dtable, ld_table, rows_table, B, T, width, time_major,
                             padding_idx, accumulate, as_stream(stream));
}

int ovqa_dropout_apply(int dtype, const void* x, void* y, int64_t n, const ovqa_dropout* drop, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype), OVQA_ERR_BAD_ARG, "dropout_apply: bad dtype");
  OVQA_REQUIRE(n >= 0 && n < (1ll << 32), OVQA_ERR_BAD_ARG, "dropout_apply: bad size");
  if (n == 0) return OVQA_OK;
  OVQA_REQUIRE(x && y, OVQA_ERR_BAD_ARG, "dropout_apply: null pointer");
  g_dispatch = "stream";
  return ovqa::dropout_apply(dtype, x, y, n, make_drop_args(drop), as_stream(stream));
}

int ovqa_pool_fwd(int feat_dtype, int dtype, const void* feat, const void* hpre, const float* w2, const float* b2, float* att,
                  void* pooled, float* pooled32, int64_t B, int64_t N, int64_t D, const ovqa_dropout* drop, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype) && dtype_ok(feat_dtype), OVQA_ERR_BAD_ARG, "pool_fwd: bad dtype");
  OVQA_REQUIRE(B >= 0 && N >= 1 && D >= 1 && B * N * D < (1ll << 32), OVQA_ERR_BAD_ARG, "pool_fwd: bad size");
  if (B == 0) return OVQA_OK;
  OVQA_REQUIRE(feat && hpre && w2 && att && pooled, OVQA_ERR_BAD_ARG, "pool_fwd: null pointer");
  g_dispatch = "stream";
  return ovqa::pool_fwd(feat_dtype, dtype, feat, hpre, w2, b2, att, pooled, pooled32, B, N, D, make_drop_args(drop),
                        as_stream(stream));
}

int ovqa_pool_bwd(int feat_dtype, int dtype, const void* feat, const void* hpre, const float* w2, const float* att,
                  const void* dpooled, void* dh, void* dfeat, float* dw2_part, float* db2_part, int64_t B, int64_t N,
                  int64_t D, const ovqa_dropout* drop, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype) && dtype_ok(feat_dtype), OVQA_ERR_BAD_ARG, "pool_bwd: bad dtype");
  OVQA_REQUIRE(B >= 0 && N >= 1 && D >= 1 && B * N * D < (1ll << 32), OVQA_ERR_BAD_ARG, "pool_bwd: bad size");
  if (B == 0) return OVQA_OK;
  OVQA_REQUIRE(feat && hpre && w2 && att && dpooled && dh && dfeat && dw2_part, OVQA_ERR_BAD_ARG, "pool_bwd: null pointer");
  g_dispatch = "stream";
  return ovqa::pool_bwd(feat_dtype, dtype, feat, hpre, w2, att, dpooled, dh, dfeat, dw2_part, db2_part, B, N, D,
                        make_drop_args(drop), as_stream(stream));
}

int ovqa_log_softmax_fwd(int dtype, const void* x, int64_t ld, float* out, int64_t M, int64_t n, void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype) && M >= 0 && n >= 1 && ld >= n && M < (1ll << 31), OVQA_ERR_BAD_ARG,
               "log_softmax_fwd: bad argument");
  if (M == 0) return OVQA_OK;
  OVQA_REQUIRE(x && out, OVQA_ERR_BAD_ARG, "log_softmax_fwd: null pointer");
  g_dispatch = "stream";
  return ovqa::log_softmax_fwd(dtype, x, ld, out, M, n, as_stream(stream));
}

int ovqa_log_softmax_bwd(int dtype, const float* g, const float* logp, void* dx, int64_t ld, int64_t M, int64_t n,
                         void* stream) {
  OVQA_REQUIRE(dtype_ok(dtype) && M >= 0 && n >= 1 && ld >= n && M < (1ll << 31), OVQA_ERR_BAD_ARG,
               "log_softmax_bwd: bad argument");
  if (M == 0) return OVQA_OK;
  OVQA_REQUIRE(g && logp && dx, OVQA_ERR_BAD_ARG, "log_softmax_bwd: null pointer");
  g_dispatch = "stream";
  return ovqa::log_softmax_bwd(dtype, g, logp, dx, ld, M, n, as_stream(stream));
}

int ovqa_nll_loss(const float* logp, const int64_t* target, float* loss, float* dlogp, const float* gscale, int64_t M,
                  int64_t n, int64_t ignore_index, int accumulate, void* stream) {
  OVQA_REQUIRE(logp && target && (loss || dlogp) && M >= 1 && n >= 1 && M * n < (1ll << 31), OVQA_ERR_BAD_ARG,
               "nll_loss: bad argument");
  g_dispatch = "stream";
  return ovqa::nll_loss(logp, target, loss, dlogp, gscale, M, n, ignore_index, accumulate, as_stream(stream));
}

}  // extern "C"
