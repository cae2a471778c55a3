// What limits the operand stream of the GEMM K loops into a CU?  (round 6; the 256 x 256-tile ablation moved 4 instead of 16 bytes
// per lane and DMA instruction in the SAME time: the stream is bound by instructions, not bytes.)  This probe streams a
// [rows][K] bf16 matrix through LDS the way the tiles do -- per K step of 64 columns every wave fetches PIECES pieces of 8 rows x
// 128 bytes -- by three routes, nothing else in the loop (a barrier per step, as the tiles have):
//   dma : global_load_lds_dwordx4            (memory -> LDS, the product's route)
//   reg : global_load_dwordx4 -> VGPRs, ds_write_b128 one step later (two register sets in flight)
//   mix : half the pieces by each route
// Prints microseconds per K step and GB/s into one CU, for `wgs` workgroups of 8 waves (one per CU up to 256).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/fetch_path_probe.hip -o scripts/_build/fetch_path_probe
// Development tool: not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } \
  } while (0)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int PIECES = 8;                      // per wave and K step (8 waves: 64 KB per step, as a 256 x 256 tile)
constexpr int STAGE = 8 * PIECES * 1024;       // bytes per K step and workgroup
constexpr int ROWS = 8 * PIECES * 8;           // matrix rows a workgroup streams (512)

// MODE 0 dma, 1 reg, 2 mix (pieces 0..3 dma, 4..7 reg); BYTES 16 or 4 per lane (dma only)
template <int MODE>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, int64_t ld_bytes, int nkt, unsigned* sink, int distinct) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (int64_t)(blockIdx.x % distinct) * ROWS * ld_bytes;  // distinct < wgs: the row blocks are shared, as the tiles of a GEMM share operand panels
  uint32_t off[PIECES];
#pragma unroll
  for (int p = 0; p < PIECES; p++)
    off[p] = (uint32_t)(((wave * PIECES + p) * 8 + (lane >> 3)) * ld_bytes) + ((((lane & 7) ^ (lane >> 3))) << 4);
  constexpr int NDMA = MODE == 0 ? PIECES : (MODE == 1 ? 0 : PIECES / 2);
  constexpr int NREG = PIECES - NDMA;
  u32x4 regs[2][NREG > 0 ? NREG : 1];
  auto issue = [&](int kt, int set) {
    char* st = smem + (kt & 1) * STAGE + wave * PIECES * 1024;
#pragma unroll
    for (int p = 0; p < PIECES; p++) {
      uint32_t o = off[p];
      asm volatile("" : "+v"(o));
      const char* s = base + (int64_t)kt * 128 + o;
      if (p < NDMA) __builtin_amdgcn_global_load_lds((gbl_void_t*)s, (lds_void_t*)(st + p * 1024), 16, 0, 0);
      else regs[set][p - NDMA] = *reinterpret_cast<const u32x4*>(s);
    }
  };
  auto land = [&](int kt, int set) {  // the register route's second half: VGPRs -> LDS
    char* st = smem + (kt & 1) * STAGE + wave * PIECES * 1024;
#pragma unroll
    for (int p = NDMA; p < PIECES; p++) *reinterpret_cast<u32x4*>(st + p * 1024 + lane * 16) = regs[set][p - NDMA];
  };
  issue(0, 0);
  if (nkt > 1) issue(1, 1);
  unsigned acc = 0;
  // the loop is unrolled by two so that the register sets are compile-time
  for (int kt = 0; kt < nkt; kt += 2) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int k = kt + h;
      if (k >= nkt) break;
      if (k + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (NREG > 0) {
        land(k, h);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      acc += *reinterpret_cast<const unsigned*>(smem + (k & 1) * STAGE + ((threadIdx.x * 68) & (STAGE - 4)));  // (one read keeps the stage live)
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (k + 2 < nkt) issue(k + 2, h);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 2048;  // bf16 columns
  const int distinct_arg = argc > 2 ? atoi(argv[2]) : 0;  // 0: every workgroup its own rows (HBM); n: n row blocks shared (L2 hits)
  const int64_t ld = (int64_t)K * 2;
  const int max_wgs = 256;
  char* src; unsigned* sink;
  CK(hipMalloc(&src, (size_t)max_wgs * ROWS * ld));
  CK(hipMemset(src, 1, (size_t)max_wgs * ROWS * ld));
  CK(hipMalloc(&sink, 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
  const int nkt = K / 64;
  const char* names[3] = {"dma", "reg", "mix"};
  printf("K = %d (%d K steps of %d KB per workgroup), matrix %d MB\n", K, nkt, STAGE / 1024, (int)((size_t)max_wgs * ROWS * ld >> 20));
  for (int wgs : {1, 50, 128, 256}) {
    const int distinct = distinct_arg > 0 ? distinct_arg : wgs;
    for (int mode = 0; mode < 3; mode++) {
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(wgs), dim3(512), 2 * STAGE, st, src, ld, nkt, sink, distinct);
        if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(wgs), dim3(512), 2 * STAGE, st, src, ld, nkt, sink, distinct);
        if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(wgs), dim3(512), 2 * STAGE, st, src, ld, nkt, sink, distinct);
      };
      for (int i = 0; i < 3; i++) launch();
      CK(hipEventRecord(e0, st));
      const int reps = 20;
      for (int i = 0; i < reps; i++) launch();
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps;
      printf("  wgs %3d (%3d row blocks)  %s : %8.2f us per launch, %.3f us per K step, %6.1f GB/s into a CU, %6.2f TB/s in all\n", wgs, distinct, names[mode], us,
             us / nkt, STAGE / (us / nkt) * 1e-3, (double)wgs * STAGE * nkt / us * 1e-6);
    }
  }
  return 0;
}
