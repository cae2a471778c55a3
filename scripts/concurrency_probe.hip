// Do kernels of two HIP streams share the chip on this stack (MI355X, ROCm 7.2)?  Build: hipcc --offload-arch=gfx950 -O2
// scripts/concurrency_probe.hip -o scripts/_build/concurrency_probe; prints one JSON object.  (the
// question-stack chain next to the grouped dW -- scripts/overlap_probe.py measured "no overlap"; this separates the
// mechanism from the workload.)
//   spin(us): every workgroup waits `us` microseconds on the wall clock -- no memory traffic, a known duration.
//   A = 64 workgroups x 256 threads (a quarter of the CUs), B = the same; C = 2048 workgroups (eight rounds of the chip).
// Cases: A then B on ONE stream (2 T expected); A and B on two non-blocking streams (T if concurrent); A in a hipGraph on
// stream 1 next to eager B on stream 2; a chain of 40 short kernels (64 workgroups, 5 us) next to one long chip-filling
// kernel, plain streams / CU-masked streams (chain 64 CUs, long 192 CUs) / priority streams.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

__global__ void spin(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {
  }
  if (sink && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *sink = 1;
}

static hipStream_t g_null = nullptr;

template <typename F>
static double median_us(F fn, std::vector<hipStream_t> streams, int reps = 15) {
  std::vector<double> ts;
  hipEvent_t a, b, j;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipEventCreateWithFlags(&j, hipEventDisableTiming);
  hipStream_t base;
  hipStreamCreateWithFlags(&base, hipStreamNonBlocking);
  for (int r = 0; r < reps + 2; r++) {
    hipEventRecord(a, base);
    for (auto s : streams) hipStreamWaitEvent(s, a, 0);
    fn();
    for (auto s : streams) {
      hipEventRecord(j, s);
      hipStreamWaitEvent(base, j, 0);
    }
    hipEventRecord(b, base);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    if (r >= 2) ts.push_back(ms * 1e3);
  }
  std::sort(ts.begin(), ts.end());
  hipStreamDestroy(base);
  return ts[ts.size() / 2];
}

int main() {
  int clk_khz = 100000;  // wall_clock64 runs at 100 MHz on gfx9
  CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
  const double tick_per_us = clk_khz / 1e3;
  auto T = [&](double us) { return (long long)(us * tick_per_us); };
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t ph, pl;
  CK(hipStreamCreateWithPriority(&ph, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&pl, hipStreamNonBlocking, lo));
  uint32_t mA[8] = {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0};
  uint32_t mB[8] = {0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  hipStream_t cA, cB;
  CK(hipExtStreamCreateWithCUMask(&cA, 8, mA));
  CK(hipExtStreamCreateWithCUMask(&cB, 8, mB));

  printf("{\"wall_clock_khz\": %d, \"priority_range\": [%d, %d]", clk_khz, lo, hi);
  // ---- two quarter-chip kernels of 500 us
  auto A = [&](hipStream_t s) { hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, T(500), (int*)nullptr); };
  printf(", \"one_stream_A_then_B_us\": %.1f", median_us([&] { A(s1); A(s1); }, {s1}));
  printf(", \"two_streams_A_B_us\": %.1f", median_us([&] { A(s1); A(s2); }, {s1, s2}));
  // ---- A captured in a graph, launched on s1, next to eager B on s2
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
  A(s1);
  CK(hipStreamEndCapture(s1, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  printf(", \"graph_A_next_to_eager_B_us\": %.1f", median_us([&] { hipGraphLaunch(ge, s1); A(s2); }, {s1, s2}));
  // ---- chain of 40 short kernels (graph) next to one long chip-filling kernel
  hipGraph_t gc;
  hipGraphExec_t gce;
  auto chain_eager = [&](hipStream_t s) {
    for (int i = 0; i < 40; i++) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, T(5), (int*)nullptr);
  };
  CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
  chain_eager(s1);
  CK(hipStreamEndCapture(s1, &gc));
  CK(hipGraphInstantiate(&gce, gc, nullptr, nullptr, 0));
  // long: 2048 workgroups x 1024 threads x 50 us: 2 workgroups per CU at a time (16 waves each) -> 4 rounds = 200 us
  auto LONG = [&](hipStream_t s) { hipLaunchKernelGGL(spin, dim3(2048), dim3(1024), 0, s, T(50), (int*)nullptr); };
  printf(", \"chain_alone_us\": %.1f", median_us([&] { hipGraphLaunch(gce, s1); }, {s1}));
  printf(", \"long_alone_us\": %.1f", median_us([&] { LONG(s2); }, {s2}));
  printf(", \"chain_then_long_one_stream_us\": %.1f", median_us([&] { hipGraphLaunch(gce, s1); LONG(s1); }, {s1}));
  printf(", \"long_first_two_plain_streams_us\": %.1f", median_us([&] { LONG(s2); hipGraphLaunch(gce, s1); }, {s1, s2}));
  printf(", \"chain_first_two_plain_streams_us\": %.1f", median_us([&] { hipGraphLaunch(gce, s1); LONG(s2); }, {s1, s2}));
  printf(", \"long_first_priority_streams_us\": %.1f", median_us([&] { LONG(pl); hipGraphLaunch(gce, ph); }, {ph, pl}));
  printf(", \"chain_alone_high_priority_us\": %.1f", median_us([&] { hipGraphLaunch(gce, ph); }, {ph}));
  printf(", \"chain_alone_masked64_us\": %.1f", median_us([&] { hipGraphLaunch(gce, cA); }, {cA}));
  printf(", \"long_alone_masked192_us\": %.1f", median_us([&] { LONG(cB); }, {cB}));
  printf(", \"long_first_masked_streams_us\": %.1f", median_us([&] { LONG(cB); hipGraphLaunch(gce, cA); }, {cA, cB}));
  printf(", \"long_first_masked_streams_eager_chain_us\": %.1f", median_us([&] { LONG(cB); chain_eager(cA); }, {cA, cB}));
  printf(", \"long_first_plain_streams_eager_chain_us\": %.1f", median_us([&] { LONG(s2); chain_eager(s1); }, {s1, s2}));
  printf("}\n");
  return 0;
}
