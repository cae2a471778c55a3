// Variant bench of the 256 x 256 GEMM tile (openvivqa_amd/csrc/gemm_tile256.h): every FLAGS variant at the step's shapes and
// at 8192 x 4096 x 4096, checked against a plain fp32-accumulating kernel, next to the library's own ovqa_linear_fwd.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/gemm256_dev.hip -o scripts/_build/gemm256_dev \
//         -Lopenvivqa_amd/csrc -lovqa_hip -Wl,-rpath,'$ORIGIN/../../openvivqa_amd/csrc'
// Development tool: not part of the product, never imported by it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <algorithm>
#include <vector>

#include "../openvivqa_amd/csrc/gemm_tile256.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

struct EpiBias {
  static constexpr bool kWide = true, kTwoPhase = true;
  bf16* y; int64_t ldy; const float* bias;
  struct Ctx { float4 b0, b1; };
  __device__ __forceinline__ void init() {}
  __device__ __forceinline__ Ctx pre(int m, int n) const {
    return Ctx{*reinterpret_cast<const float4*>(bias + n), *reinterpret_cast<const float4*>(bias + n + 4)};
  }
  __device__ __forceinline__ void post(const Ctx& k, int m, int n, const f32x4& lo, const f32x4& hi) const {
    bf16x8 o;
    o[0] = (bf16)(lo[0] + k.b0.x); o[1] = (bf16)(lo[1] + k.b0.y); o[2] = (bf16)(lo[2] + k.b0.z); o[3] = (bf16)(lo[3] + k.b0.w);
    o[4] = (bf16)(hi[0] + k.b1.x); o[5] = (bf16)(hi[1] + k.b1.y); o[6] = (bf16)(hi[2] + k.b1.z); o[7] = (bf16)(hi[3] + k.b1.w);
    *reinterpret_cast<bf16x8*>(y + (int64_t)m * ldy + n) = o;
  }
  __device__ __forceinline__ void wide(int m, int n, const f32x4& lo, const f32x4& hi) const { post(pre(m, n), m, n, lo, hi); }
};
// the one-phase form (every piece its own load -> wait -> store chain), for the A/B
struct EpiBias1 : EpiBias {
  static constexpr bool kTwoPhase = false;
};

__global__ void ref_kernel(const bf16* x, const bf16* w, const float* bias, float* y, int M, int N, int K) {
  const int n = blockIdx.x * 64 + (threadIdx.x & 63), m = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (m >= M || n >= N) return;
  float a = 0.f;
  for (int k = 0; k < K; k++) a += (float)x[(int64_t)m * K + k] * (float)w[(int64_t)n * K + k];
  y[(int64_t)m * N + n] = a + bias[n];
}

__global__ void fill_kernel(bf16* p, int64_t n, uint32_t seed, float scale) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t h = ovqa_fmix32((uint32_t)i ^ seed) ^ ovqa_fmix32((uint32_t)(i >> 32) + seed * 77u);
  // sum of two uniforms: roughly bell shaped, zero mean
  const float u = ((h & 0xFFFF) + (h >> 16)) * (1.f / 65536.f) - 1.f;
  p[i] = (bf16)(u * scale);
}
__global__ void fillf_kernel(float* p, int64_t n, uint32_t seed) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = ((ovqa_fmix32((uint32_t)i ^ seed) & 0xFFFF) * (1.f / 65536.f) - 0.5f);
}
__global__ void cmp_kernel(const bf16* y, const float* ref, int64_t n, float* maxerr, unsigned long long* bad) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float r = ref[i], v = (float)y[i];
  const float err = fabsf(v - r), tol = 0.01f * fabsf(r) + 0.02f;
  if (!(err <= tol)) atomicAdd(bad, 1ull);
  atomicMax(reinterpret_cast<int*>(maxerr), __float_as_int(err));
}

static unsigned long long* g_probe = nullptr;
template <typename E>
static E make_epi(bf16* y, int64_t ldy, const float* bias) {
  E e;
  e.y = y; e.ldy = ldy; e.bias = bias;
  return e;
}
template <int FLAGS, int GROUP = 0, typename E = EpiBias>
static void launch256(const bf16* x, const bf16* w, const float* bias, bf16* y, int M, int N, int K, hipStream_t st) {
  using namespace ovqa_t256;
  Args g{w, K, x, K, N, M, K, (N + T - 1) / T, (M + T - 1) / T, g_probe, GROUP};
  static bool attr = false;
  if (!attr) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel<E, FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                           LDS_BYTES));
    attr = true;
  }
  hipLaunchKernelGGL((kernel<E, FLAGS>), dim3(g.tiles_r * g.tiles_c), dim3(512), LDS_BYTES, st, g, make_epi<E>(y, N, bias));
}

typedef void (*launch_fn)(const bf16*, const bf16*, const float*, bf16*, int, int, int, hipStream_t);
struct Variant { const char* name; launch_fn fn; };

static void lib_fwd(const bf16* x, const bf16* w, const float* bias, bf16* y, int M, int N, int K, hipStream_t st) {
  int rc = ovqa_linear_fwd(OVQA_BF16, OVQA_EPI_BIAS, x, K, w, bias, nullptr, 0, y, N, nullptr, M, N, K, nullptr, st);
  if (rc != OVQA_OK) { fprintf(stderr, "ovqa_linear_fwd: %d %s\n", rc, ovqa_last_error()); exit(3); }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 50;
  const int NSET = 4;  // rotating operand sets: "cold" = every launch reads operands the previous three did not touch
  std::vector<Variant> vars = {
      {"lib(128-tiles)", lib_fwd},
      {"t256 flags=11 (early,rot,prio)", launch256<11>},
      {"t256 flags=11 group_c=4", launch256<11, 4>},
      {"t256 flags=11 one-phase epilogue", launch256<11, 0, EpiBias1>},
      {"ablation: MFMA only", launch256<3 + 16 + 64>},
      {"ablation: reads only", launch256<3 + 16 + 32>},
      {"ablation: MFMA + reads (no DMA)", launch256<3 + 16>},
      {"ablation: MFMA + DMA (no reads)", launch256<3 + 64>},
      {"ablation: DMA + reads (no MFMA)", launch256<3 + 32>},
      {"ablation: MFMA + reads, no ROT", launch256<2 + 16>},
      {"ablation: MFMA only, no ROT", launch256<2 + 16 + 64>},
      {"ablation: DMA only", launch256<3 + 32 + 64>},
      {"ablation: DMA only, 4 B per lane", launch256<3 + 32 + 64 + 512>},
      {"ablation: DMA only, same K block", launch256<3 + 32 + 64 + 1024>},
      {"ablation: all, same K block", launch256<3 + 1024>},
  };
  const int shapes[][3] = {{6400, 2048, 512}, {6400, 1536, 512}, {6400, 512, 2048}, {6400, 3072, 768}, {8192, 4096, 4096},
                           {300, 264, 128}};
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float* d_maxerr; unsigned long long* d_bad;
  CK(hipMalloc(&d_maxerr, 4)); CK(hipMalloc(&d_bad, 8));
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    const int nset = (int64_t)M * N > 20000000 ? 1 : NSET;
    bf16 *x[NSET], *w[NSET], *y[NSET]; float* bias; float* ref;
    for (int i = 0; i < nset; i++) {
      CK(hipMalloc(&x[i], (size_t)M * K * 2)); CK(hipMalloc(&w[i], (size_t)N * K * 2)); CK(hipMalloc(&y[i], (size_t)M * N * 2));
      hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(((int64_t)M * K + 255) / 256)), dim3(256), 0, st, x[i], (int64_t)M * K, 11u + i, 1.f);
      hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(((int64_t)N * K + 255) / 256)), dim3(256), 0, st, w[i], (int64_t)N * K, 23u + i, 0.1f);
    }
    CK(hipMalloc(&bias, (size_t)N * 4)); CK(hipMalloc(&ref, (size_t)M * N * 4));
    hipLaunchKernelGGL(fillf_kernel, dim3((N + 255) / 256), dim3(256), 0, st, bias, (int64_t)N, 5u);
    hipLaunchKernelGGL(ref_kernel, dim3((N + 63) / 64, (M + 3) / 4), dim3(256), 0, st, x[0], w[0], bias, ref, M, N, K);
    CK(hipStreamSynchronize(st));
    printf("shape M=%d N=%d K=%d  (%.2f GFLOP, tiles256=%d)\n", M, N, K, 2.0 * M * N * K * 1e-9, ((M + 255) / 256) * ((N + 255) / 256));
    for (auto& v : vars) {
      CK(hipMemsetAsync(y[0], 0xFF, (size_t)M * N * 2, st));
      CK(hipMemsetAsync(d_maxerr, 0, 4, st)); CK(hipMemsetAsync(d_bad, 0, 8, st));
      v.fn(x[0], w[0], bias, y[0], M, N, K, st);
      hipLaunchKernelGGL(cmp_kernel, dim3((unsigned)(((int64_t)M * N + 255) / 256)), dim3(256), 0, st, y[0], ref, (int64_t)M * N, d_maxerr, d_bad);
      float maxerr; unsigned long long bad;
      CK(hipMemcpyAsync(&maxerr, d_maxerr, 4, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      // warm: same operands back to back
      for (int i = 0; i < 5; i++) v.fn(x[0], w[0], bias, y[0], M, N, K, st);
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; i++) v.fn(x[0], w[0], bias, y[0], M, N, K, st);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms_w; CK(hipEventElapsedTime(&ms_w, e0, e1));
      float ms_c = 0.f;
      if (nset > 1) {
        for (int i = 0; i < 4; i++) v.fn(x[i % nset], w[i % nset], bias, y[i % nset], M, N, K, st);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; i++) v.fn(x[i % nset], w[i % nset], bias, y[i % nset], M, N, K, st);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_c, e0, e1));
      }
      const double us_w = ms_w * 1e3 / reps, us_c = ms_c * 1e3 / reps, fl = 2.0 * M * N * K;
      printf("  %-34s warm %8.2f us %7.1f TF | rotating %8.2f us %7.1f TF | maxerr %.4f bad %llu %s\n", v.name, us_w,
             fl / us_w * 1e-6, us_c, us_c > 0 ? fl / us_c * 1e-6 : 0.0, maxerr, bad, bad ? "(mismatch)" : "ok");
      fflush(stdout);
    }
    {  // in-kernel wall-clock stamps (100 MHz) of three back-to-back launches: where the time outside the K loop goes
      const int nwg = ((M + 255) / 256) * ((N + 255) / 256);
      unsigned long long* pb[3];
      std::vector<unsigned long long> h[3];
      for (int l = 0; l < 3; l++) { CK(hipMalloc(&pb[l], (size_t)nwg * 8 * 6 * 8)); CK(hipMemsetAsync(pb[l], 0, (size_t)nwg * 8 * 6 * 8, st)); }
      launch256<11>(x[0], w[0], bias, y[0], M, N, K, st);
      for (int l = 0; l < 3; l++) { g_probe = pb[l]; launch256<11 + 256>(x[0], w[0], bias, y[0], M, N, K, st); }
      g_probe = nullptr;
      CK(hipStreamSynchronize(st));
      for (int l = 0; l < 3; l++) { h[l].resize((size_t)nwg * 48); CK(hipMemcpy(h[l].data(), pb[l], h[l].size() * 8, hipMemcpyDeviceToHost)); CK(hipFree(pb[l])); }
      auto mn = [&](int l, int f) { unsigned long long v = ~0ull; for (int i = 0; i < nwg * 8; i++) v = std::min(v, h[l][(size_t)i * 6 + f]); return v; };
      auto mx = [&](int l, int f) { unsigned long long v = 0; for (int i = 0; i < nwg * 8; i++) v = std::max(v, h[l][(size_t)i * 6 + f]); return v; };
      for (int l = 1; l < 3; l++) {
        const double t0 = (double)mn(l, 0);
        printf("  probe launch %d: first wave start 0; last wave start %.2f us; K loop begins (median wg0) %.2f; loop done first/last %.2f / %.2f; stores issued last %.2f; stores done last %.2f; gap to previous launch's end %.2f us\n",
               l, (mx(l, 0) - t0) * 0.01, (h[l][1] - (double)h[l][0]) * 0.01, (mn(l, 2) - t0) * 0.01, (mx(l, 2) - t0) * 0.01,
               (mx(l, 3) - t0) * 0.01, (mx(l, 4) - t0) * 0.01, (t0 - (double)mx(l - 1, 4)) * 0.01);
      }
    }
    for (int i = 0; i < nset; i++) { CK(hipFree(x[i])); CK(hipFree(w[i])); CK(hipFree(y[i])); }
    CK(hipFree(bias)); CK(hipFree(ref));
  }
  return 0;
}
