// Microbenchmark of the direct-to-LDS streaming rate per CU (global_load_lds_dwordx4) as the GEMM kernels use it:
// which of {piece shape, row stride, residency of the source, waves per workgroup, workgroups per CU, ring depth,
// the per-stage barrier, the fragment reads} sets the ~45 GB/s per CU seen in the step.  Standalone (plain HIP + C ABI
// for scripts/lds_stream_probe.py); not part of the library.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// MODE 0: a piece (1 KiB, one wave instruction) is contiguous.  MODE 1: a piece is 8 rows x 128 B, rows `row_stride`
// bytes apart (a K-contiguous GEMM operand tile, BK = 64).  MODE 2: 16 rows x 64 B (BK = 32).
template <int NW, int NBUF, int PPW, int MODE, bool BARRIER, bool CONSUME, int RD = 1>
__global__ __launch_bounds__(NW * 64) void stream_kernel(const char* __restrict__ src, long long wg_stride, int nsteps,
                                                         int row_stride, int lds_pad, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = NW * PPW * 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const char* base = src + (long long)blockIdx.x * wg_stride;
  const char* lane_src[PPW];
#pragma unroll
  for (int p = 0; p < PPW; p++) {
    const int piece = wave * PPW + p;
    if (MODE == 0) lane_src[p] = base + piece * 1024 + lane * 16;
    else if (MODE == 1) lane_src[p] = base + (long long)(piece * 8 + (lane >> 3)) * row_stride + (lane & 7) * 16;
    else lane_src[p] = base + (long long)(piece * 16 + (lane >> 2)) * row_stride + (lane & 3) * 16;
  }
  const int step_bytes = MODE == 0 ? STAGE : (MODE == 1 ? 128 : 64);  // advance along the row (K) per step
  auto issue = [&](int s) {
    char* buf = smem + (s % NBUF) * STAGE;
#pragma unroll
    for (int p = 0; p < PPW; p++)
      __builtin_amdgcn_global_load_lds((gbl_void*)(lane_src[p] + (long long)s * step_bytes),
                                       (lds_void*)(buf + (wave * PPW + p) * 1024), 16, 0, 0);
  };
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < NBUF - 1; p++)
    if (p < nsteps) issue(p);
  for (int s = 0; s < nsteps; s++) {
    if (s + NBUF - 2 >= nsteps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (NBUF - 2)) : "memory");
    if (BARRIER) __builtin_amdgcn_s_barrier();
    if (s + NBUF - 1 < nsteps) issue(s + NBUF - 1);
    if (CONSUME) {  // every wave reads the whole stage's worth per lane share: STAGE / (NW * 64) bytes per lane
      const char* buf = smem + (s % NBUF) * STAGE;
      // RD x the stage's bytes per workgroup in ds_read_b128 (a 128 x 64 GEMM tile reads ~4x what it stages)
#pragma unroll
      for (int rep = 0; rep < RD; rep++)
#pragma unroll
        for (int r = 0; r < PPW; r++) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(buf + ((((wave + rep) % NW) * PPW + r) * 64 + lane) * 16);
          acc += v;
        }
    }
  }
  if (CONSUME && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[tid] = acc[0];
  (void)lds_pad;
}

template <int NW, int NBUF, int PPW, int MODE, bool BARRIER, bool CONSUME, int RD = 1>
static int run(const char* src, long long wg_stride, int nsteps, int row_stride, int lds_pad, int blocks, int reps,
               float* sink, float* ms_out) {
  const size_t lds = (size_t)NBUF * NW * PPW * 1024 + lds_pad;
  auto k = stream_kernel<NW, NBUF, PPW, MODE, BARRIER, CONSUME, RD>;
  if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(NW * 64), lds, 0, src, wg_stride, nsteps, row_stride, lds_pad, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL(k, dim3(blocks), dim3(NW * 64), lds, 0, src, wg_stride, nsteps, row_stride, lds_pad, sink);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  hipEventElapsedTime(ms_out, e0, e1);
  *ms_out /= reps;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int lsp_run(int variant, const char* src, long long wg_stride, int nsteps, int row_stride, int lds_pad,
                       int blocks, int reps, float* sink, float* ms_out) {
  switch (variant) {
    //                NW NBUF PPW MODE BARRIER CONSUME
    case 0: return run<8, 3, 3, 0, true, true>(src, wg_stride, nsteps, row_stride, lds_pad, blocks, reps, sink, ms_out);
    case 1: return run<8, 3, 3, 1, true, true>(src, wg_stride, nsteps, row_stride, lds_pad, blocks, reps, sink, ms_out);
    case 2: return run<8, 3, 3, 1, false, true>(src, wg_stride, nsteps, row_stride, lds_pad, blocks, reps, sink, ms_out);
    case 3: return run<8, 3, 3, 1, true, false>(src, wg_stride, nsteps, row_stride, lds_pad, blocks, reps, sink, ms_out);
    case 4: return run<8, 3, 3, 1, false, false>(src, wg_stride, nsteps, row_stride, lds_pad, blocks, reps, sink, ms_out);
    case 5: return run<8, 2, 3, 1, true, true>(src, wg_stride, nsteps, row_stride, lds_pad, blocks, reps, sink, ms_out);
    case 6: return run<8, 4, 3, 1, true, true>(src, wg_stride, nsteps, row_stride, lds_pad, blocks, reps, sink, ms_out);
    case 7: return run<4, 3, 6, 1, true, true>(src, wg_stride, nsteps, row_stride, lds_pad, blocks, reps, sink, ms_out);
    case 8: return run<8, 3, 3, 2, true, true>(src, wg_stride, nsteps, row_stride, lds_pad, blocks, reps, sink, ms_out);
    case 9: return run<8, 6, 3, 1, false, false>(src, wg_stride, nsteps, row_stride, lds_pad, blocks, reps, sink, ms_out);
    case 10: return run<8, 3, 6, 1, true, true>(src, wg_stride, nsteps, row_stride, lds_pad, blocks, reps, sink, ms_out);
    case 11: return run<16, 3, 3, 1, true, true>(src, wg_stride, nsteps, row_stride, lds_pad, blocks, reps, sink, ms_out);
    case 12: return run<16, 2, 3, 1, true, true>(src, wg_stride, nsteps, row_stride, lds_pad, blocks, reps, sink, ms_out);
    case 13: return run<16, 3, 2, 1, true, true>(src, wg_stride, nsteps, row_stride, lds_pad, blocks, reps, sink, ms_out);
    case 14: return run<8, 3, 3, 1, true, true, 2>(src, wg_stride, nsteps, row_stride, lds_pad, blocks, reps, sink, ms_out);
    case 15: return run<8, 3, 3, 1, true, true, 4>(src, wg_stride, nsteps, row_stride, lds_pad, blocks, reps, sink, ms_out);
    case 16: return run<8, 3, 3, 1, true, true, 8>(src, wg_stride, nsteps, row_stride, lds_pad, blocks, reps, sink, ms_out);
    default: return -3;
  }
}
