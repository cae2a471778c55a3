// Shared device/host helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ovqa_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define WAVE 64

// ---- host-side error plumbing -------------------------------------------
void ovqa_set_error(const char* fmt, ...);
#define OVQA_REQUIRE(cond, code, ...)          \
  do {                                         \
    if (!(cond)) {                             \
      ovqa_set_error(__VA_ARGS__);             \
      return (code);                           \
    }                                          \
  } while (0)

static inline int ovqa_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ovqa_set_error("%s: %s", what, hipGetErrorString(e));
    return OVQA_ERR_LAUNCH;
  }
  return OVQA_OK;
}

// ---- launch timing (diagnostic; ovqa_launch_timing_begin / _end) -------------------------------------------
// While armed, the instrumented launches go through hipExtLaunchKernelGGL with a start / stop event pair: the events
// take the dispatch packet's own begin / end timestamps -- the kernel's execution time as rocprofv3 --kernel-trace
// reports it, free of the cost of separate event-record packets around the launch.
bool ovqa_timer_next(hipEvent_t* start, hipEvent_t* stop);
#define OVQA_LAUNCH_TIMED(kernel, grid, block, lds, st, ...)                              \
  do {                                                                                    \
    hipEvent_t t0_ = nullptr, t1_ = nullptr;                                              \
    if (ovqa_timer_next(&t0_, &t1_))                                                      \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, st, t0_, t1_, 0, __VA_ARGS__);      \
    else                                                                                  \
      hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                      \
  } while (0)

// ---- stores of the FFN pre-activation (26 MB per 6400-row launch, read again only in BACKWARD): the nt (streaming) cache
// policy keeps it from displacing the hidden activation the NEXT kernel reads out of the 4 MB L2s and from sitting dirty in
// them at the kernel boundary.  MEASURED (MCAN step, alternated on one box): 3.322 / 3.319 -> 3.292 / 3.295 ms.  The same
// policy on the saved q / k / v projections and o_lo of the fused attention forwards LOST 0.03 ms (their stores sit at the
// tail of a short kernel, where a slower store is exposed): plain stores there.  -DOVQA_NT_SAVED=0 = plain stores (A/B).
#ifndef OVQA_NT_SAVED
#define OVQA_NT_SAVED 1
#endif
template <typename V>
__device__ __forceinline__ void store_saved(V* p, const V& v) {
#if OVQA_NT_SAVED
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// ---- XCD-consistent ownership of activation rows -------------------------------------------------------------------
// Workgroups are dealt to the 8 XCDs round-robin by block id, each XCD has its own L2, and a kernel's output is still in
// the producing XCD's L2 when the next kernel starts.  The GEMM kernels' block remap gives XCD x the x-th contiguous
// eighth of the activation rows (gemm_mfma.hip); the LayerNorm kernels use the same bijection for their row blocks, so a
// row is normalised by the XCD whose GEMM tiles wrote it and read by that XCD's next GEMM tiles (3.372 -> 3.359 ms per
// step; dealing the attention kernels' (sample, head) problems the same way measured no difference).
__device__ __forceinline__ int xcd_contiguous_block(int bid, int nwg) {
#ifdef OVQA_NO_XCD_REMAP
  return bid;
#else
  const int q = nwg / 8, rem = nwg % 8, xcd = bid % 8, idx = bid / 8;
  return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
#endif
}

// ---- Adam ------------------------------------------------------------------
// One element of torch.optim.Adam's update, written ONCE: the tiled Adam kernel (misc.hip) and the Adam epilogue of the
// grouped weight-gradient kernel (gemm_mfma.hip) must give the same bits for the same gradient (the same expression tree
// is contracted into the same fmas).
struct AdamK {
  float b1, b2, eps, wd, grad_scale, step_size, inv_sqrt_bc2;
};
__device__ __forceinline__ AdamK adam_consts(float lr, const float* lr_scale_ptr, float b1, float b2, float eps, float wd,
                                             float grad_scale, const uint32_t* step_ptr) {
  const float t = (float)(step_ptr ? *step_ptr : 1u);
  const float lr_eff = lr * (lr_scale_ptr ? *lr_scale_ptr : 1.f);
  const float bc1 = 1.f - powf(b1, t);
  const float bc2 = 1.f - powf(b2, t);
  return AdamK{b1, b2, eps, wd, grad_scale, lr_eff / bc1, rsqrtf(bc2)};
}
// (contraction is OFF inside and every fused multiply-add is written out: left to the compiler, the two kernels contracted
// `b1 * m + (1 - b1) * g` differently -- equal results while m = 0, different last bits from the second step on)
__device__ __forceinline__ void adam_update1(const AdamK& k, float g, float& p, float& m, float& v) {
#pragma clang fp contract(off)
  const float gk = __builtin_fmaf(k.wd, p, g * k.grad_scale);
  m = __builtin_fmaf(k.b1, m, (1.f - k.b1) * gk);
  v = __builtin_fmaf(k.b2, v, ((1.f - k.b2) * gk) * gk);
  const float denom = __builtin_fmaf(sqrtf(v), k.inv_sqrt_bc2, k.eps);
  p = p - (k.step_size * m) / denom;
}

// ---- scalar conversion -----------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// ---- wave reductions (64-wide) --------------------------------------------
// All-lane sum / max through DPP row operations (quad swaps, half-row and row mirrors, then the gfx9 row broadcasts
// 15 / 31 with row masks) and one v_readlane of lane 63: six ~8-cycle VALU operations instead of six dependent
// ds_bpermute round trips (__shfl_xor) -- the one-wave-per-row LayerNorm kernels are chains of two such reductions.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float v, float identity) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0xB1, 0xF>(v, 0.f);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E, 0xF>(v, 0.f);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141, 0xF>(v, 0.f);  // row_half_mirror
  v += dpp_move<0x140, 0xF>(v, 0.f);  // row_mirror: every lane holds its 16-lane row's sum
  v += dpp_move<0x142, 0xA>(v, 0.f);  // row_bcast:15 into rows 1 and 3
  v += dpp_move<0x143, 0xC>(v, 0.f);  // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_move<0xB1, 0xF>(v, v));
  v = fmaxf(v, dpp_move<0x4E, 0xF>(v, v));
  v = fmaxf(v, dpp_move<0x141, 0xF>(v, v));
  v = fmaxf(v, dpp_move<0x140, 0xF>(v, v));
  v = fmaxf(v, dpp_move<0x142, 0xA>(v, v));
  v = fmaxf(v, dpp_move<0x143, 0xC>(v, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ---- dropout: stateless counter hash ---------------------------------------
// keep(idx) is a pure function of (seed, site, step, idx); forward and backward
// regenerate it.  One 32-bit hash (fmix32 of murmur3 over pair ^ key) serves the element PAIR
// {2j, 2j+1}: each element compares its own 16 bits with a 16-bit threshold (p is resolved to 2^-16).  The fused
// epilogues own 8 consecutive elements per lane, i.e. 4 hashes instead of 8: the epilogues are VALU-bound and the
// 32-bit integer multiplies of the hash were a third of their instruction stream.
struct DropState {
  uint32_t key;
  uint32_t thresh;  // drop iff (16-bit field of the hash) < thresh, thresh = round(p * 65536)
  float inv_keep;   // 1/(1-p)
  bool on;
};
__host__ __device__ __forceinline__ uint32_t ovqa_fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}
struct DropArgs {  // by-value kernel argument
  float p;
  uint32_t seed;
  uint32_t site;
  const uint32_t* step;
};
static inline DropArgs make_drop_args(const ovqa_dropout* d) {
  DropArgs a{0.f, 0u, 0u, nullptr};
  if (d) { a.p = d->p; a.seed = d->seed; a.site = d->site; a.step = d->step; }
  return a;
}
__device__ __forceinline__ DropState drop_init(const DropArgs& a) {
  DropState s;
  s.on = a.p > 0.f;
  uint32_t step = (a.step != nullptr) ? *a.step : 0u;
  s.key = ovqa_fmix32(a.seed ^ ovqa_fmix32(a.site * 0x9E3779B1u + step * 0x7F4A7C15u + 0x1234567u));
  s.thresh = (a.p >= 1.f) ? 0x10000u : (uint32_t)(a.p * 65536.f + 0.5f);
  s.inv_keep = (a.p < 1.f) ? 1.f / (1.f - a.p) : 0.f;
  return s;
}
__device__ __forceinline__ uint32_t drop_pair_hash(const DropState& s, uint32_t pair) {
  // murmur3's finalizer is a full-avalanche bijection: the counter needs no multiply of its own in front of it (a
  // 32-bit integer multiply is quarter rate: 16 cycles per wave instruction; lag / cross-key / uniformity statistics
  // of the masks are indistinguishable from the pre-multiplied form, scripts/README.md)
  return ovqa_fmix32(pair ^ s.key);
}
__device__ __forceinline__ bool drop_keep(const DropState& s, uint32_t idx) {
  const uint32_t h = drop_pair_hash(s, idx >> 1);
  return ((idx & 1u) ? (h >> 16) : (h & 0xFFFFu)) >= s.thresh;
}
// multiplier applied to a value that went through dropout (1 when off)
__device__ __forceinline__ float drop_mul(const DropState& s, uint32_t idx) {
  if (!s.on) return 1.f;
  return drop_keep(s, idx) ? s.inv_keep : 0.f;
}
// multipliers of 8 consecutive elements idx .. idx+7 (idx EVEN): 4 hashes
__device__ __forceinline__ void drop_mul8(const DropState& s, uint32_t idx, float (&m)[8]) {
  if (!s.on) {
#pragma unroll
    for (int t = 0; t < 8; t++) m[t] = 1.f;
    return;
  }
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const uint32_t h = drop_pair_hash(s, (idx >> 1) + t);
    m[2 * t] = (h & 0xFFFFu) >= s.thresh ? s.inv_keep : 0.f;
    m[2 * t + 1] = (h >> 16) >= s.thresh ? s.inv_keep : 0.f;
  }
}

// ---- exact GELU (erf form, F.gelu default) ----------------------------------
__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float u) {
  const float cdf = 0.5f * (1.f + erff(u * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * u * u);
  return cdf + u * pdf;
}

// GELU of the bf16 MFMA epilogues.  The FORM stays the reference's exact-erf GELU, u * Phi(u) (F.gelu default), not the
// tanh approximation; Phi, the normal CDF 0.5 (1 + erf(u / sqrt 2)), is evaluated as 0.5 + u Q(u^2) with Q a degree-7
// minimax polynomial on |u| <= 4 (|error| <= 2.2e-5 there; u and u^2 are clamped beyond: <= 5.2e-5) -- 11 VALU
// instructions and no transcendental, against ~14 + v_rcp + v_exp (4 cycles x 4 each) for the Abramowitz-Stegun
// 7.1.26 form used before: the GELU epilogues are VALU-bound (a wave64 VALU instruction is 4 cycles, a transcendental
// 16), and the outputs are rounded to bf16 (2^-9) anyway.  The fp32 mode uses erff (gelu_f above).
__device__ __forceinline__ float norm_cdf_fast(float u) {
  const float t = fminf(u * u, 16.f);
  float q = -1.5807682306e-09f;
  q = fmaf(q, t, 1.2171011739e-07f);
  q = fmaf(q, t, -4.1008447793e-06f);
  q = fmaf(q, t, 8.0667156048e-05f);
  q = fmaf(q, t, -1.0482029970e-03f);
  q = fmaf(q, t, 9.6648700100e-03f);
  q = fmaf(q, t, -6.6175373779e-02f);
  q = fmaf(q, t, 3.9884750779e-01f);
  return fmaf(__builtin_amdgcn_fmed3f(u, -4.f, 4.f), q, 0.5f);
}
__device__ __forceinline__ float gelu_fast(float u) { return u * norm_cdf_fast(u); }
// d/du [u Phi(u)] = Phi(u) + u phi(u), phi = exp(-u^2 / 2) / sqrt(2 pi)
__device__ __forceinline__ float gelu_grad_fast(float u) {
  const float e = __builtin_amdgcn_exp2f(u * u * -0.72134752044448170368f);  // exp(-u^2 / 2)
  return fmaf(u * 0.39894228040143267794f, e, norm_cdf_fast(u));
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
