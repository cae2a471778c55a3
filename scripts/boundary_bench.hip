// Kernel-boundary microbenchmark (MI355X): what does one DEPENDENT launch cost on a stream, launched eagerly,
// replayed from a hipGraph captured here, or replayed from a torch CUDAGraph (scripts/boundary_bench.py)?
// Chains: trivial kernels (1 / 256 / 1024 workgroups), streaming copies of S bytes (plain or nontemporal
// stores: how much of the boundary is the predecessor's dirty L2 lines), and copy+trivial alternations.
// Built by scripts/boundary_bench.py with hipcc; plain C ABI so ctypes can drive it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x)                                                                     \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      return -1;                                                                     \
    }                                                                                \
  } while (0)

__global__ __launch_bounds__(256) void k_trivial(uint32_t* p) {
  if (blockIdx.x == 0 && threadIdx.x == 0) p[0] += 1u;
}

template <int MODE>  // 0 plain stores, 1 nontemporal stores
__global__ __launch_bounds__(256) void k_copy(uint4* __restrict__ dst, const uint4* __restrict__ src, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    uint4 v = src[i];
    v.x += 1u;
    if (MODE == 0) {
      dst[i] = v;
    } else {
      __builtin_nontemporal_store(v.x, &dst[i].x);
      __builtin_nontemporal_store(v.y, &dst[i].y);
      __builtin_nontemporal_store(v.z, &dst[i].z);
      __builtin_nontemporal_store(v.w, &dst[i].w);
    }
  }
}

extern "C" {

// kind: 0 trivial, 1 copy (plain), 2 copy (nt), 3 copy(plain)+trivial alternating (2 launches per link)
// Launches `n` links on `stream`.  a/b: ping-pong buffers of `bytes` each; cnt: a uint32 for the trivial kernel.
int bb_chain(int kind, int n, int blocks, void* a, void* b, int64_t bytes, void* cnt, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int64_t n16 = bytes / 16;
  int cb = (int)((n16 + 255) / 256);
  if (cb > 2048) cb = 2048;
  if (cb < 1) cb = 1;
  for (int i = 0; i < n; i++) {
    void* src = (i & 1) ? b : a;
    void* dst = (i & 1) ? a : b;
    switch (kind) {
      case 0: hipLaunchKernelGGL(k_trivial, dim3(blocks), dim3(256), 0, st, (uint32_t*)cnt); break;
      case 1: hipLaunchKernelGGL(k_copy<0>, dim3(cb), dim3(256), 0, st, (uint4*)dst, (const uint4*)src, n16); break;
      case 2: hipLaunchKernelGGL(k_copy<1>, dim3(cb), dim3(256), 0, st, (uint4*)dst, (const uint4*)src, n16); break;
      case 3:
        hipLaunchKernelGGL(k_copy<0>, dim3(cb), dim3(256), 0, st, (uint4*)dst, (const uint4*)src, n16);
        hipLaunchKernelGGL(k_trivial, dim3(blocks), dim3(256), 0, st, (uint32_t*)cnt);
        break;
      default: return -2;
    }
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// mode 0: eager launches from this C loop; mode 1: the chain captured into a hipGraph here, replayed `reps` times.
// Returns microseconds per LINK (a link = one launch, two for kind 3) in *us_per_link.
int bb_time(int kind, int n, int blocks, int64_t bytes, int mode, int reps, float* us_per_link) {
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  void *a = nullptr, *b = nullptr, *cnt = nullptr;
  const int64_t alloc = bytes > 16 ? bytes : 16;
  CHECK(hipMalloc(&a, alloc));
  CHECK(hipMalloc(&b, alloc));
  CHECK(hipMalloc(&cnt, 256));
  CHECK(hipMemsetAsync(a, 1, alloc, st));
  CHECK(hipMemsetAsync(b, 2, alloc, st));
  CHECK(hipMemsetAsync(cnt, 0, 256, st));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float ms = 0.f;
  if (mode == 0) {
    if (bb_chain(kind, n, blocks, a, b, bytes, cnt, st)) return -1;  // warm
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; r++)
      if (bb_chain(kind, n, blocks, a, b, bytes, cnt, st)) return -1;
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
  } else {
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    if (bb_chain(kind, n, blocks, a, b, bytes, cnt, st)) return -1;
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(ge, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; r++) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipGraphExecDestroy(ge));
    CHECK(hipGraphDestroy(g));
  }
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  *us_per_link = ms * 1e3f / ((float)reps * (float)n);
  CHECK(hipFree(a));
  CHECK(hipFree(b));
  CHECK(hipFree(cnt));
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  CHECK(hipStreamDestroy(st));
  return 0;
}

}  // extern "C"
