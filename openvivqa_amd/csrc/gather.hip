// Grouped row gather: the beam-search state reorder (models/modules/beam_search.py:19-34) for ALL state buffers of a
// decoder in one launch.  The reference applies torch.gather to every running K/V cache, mask and position buffer in
// turn (one launch and one expanded index tensor each); here a table of {src, dst, row bytes} shares one
// selected-beam index: dst[(b*beam + j), :] = src[(b*cur + sel[b*beam + j]), :].  Pure byte movement (HBM-bound),
// 16-byte accesses when the row size and the pointers allow, any dtype.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int GATHER_MAX = 24;  // problems per launch: the table travels in the kernel arguments (24 x 40 B)
struct GatherTable {
  ovqa_gather_problem p[GATHER_MAX];
};

__global__ __launch_bounds__(256) void gather_rows_kernel(GatherTable tab, const int32_t* __restrict__ sel, int cur,
                                                          int beam) {
  const ovqa_gather_problem pr = tab.p[blockIdx.y];
  const int orow = blockIdx.x;            // b * beam + j
  const int b = orow / beam;
  const int srow = b * cur + sel[orow];
  const char* src = (const char*)pr.src + (int64_t)srow * (pr.src_stride_bytes ? pr.src_stride_bytes : pr.row_bytes);
  char* dst = (char*)pr.dst + (int64_t)orow * (pr.dst_stride_bytes ? pr.dst_stride_bytes : pr.row_bytes);
  const int64_t n = pr.row_bytes;
  if (((uintptr_t)src | (uintptr_t)dst | (uintptr_t)n) % 16 == 0) {
    const uint4* s4 = (const uint4*)src;
    uint4* d4 = (uint4*)dst;
    for (int64_t i = threadIdx.x; i < n / 16; i += 256) d4[i] = s4[i];
  } else {
    for (int64_t i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
  }
}

}  // namespace

namespace ovqa {

int grouped_row_gather(const ovqa_gather_problem* probs, int n_problems, const int32_t* sel, int b_s, int cur, int beam,
                       hipStream_t st) {
  if (n_problems <= 0 || b_s <= 0 || beam <= 0) return OVQA_OK;
  for (int p0 = 0; p0 < n_problems; p0 += GATHER_MAX) {
    const int np = n_problems - p0 < GATHER_MAX ? n_problems - p0 : GATHER_MAX;
    GatherTable tab;
    for (int i = 0; i < np; i++) tab.p[i] = probs[p0 + i];  // (host memory: see include/ovqa_hip.h)
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(b_s * beam), (unsigned)np), dim3(256), 0, st, tab, sel, cur,
                       beam);
    int rc = ovqa_check_launch("grouped_row_gather");
    if (rc != OVQA_OK) return rc;
  }
  return OVQA_OK;
}

}  // namespace ovqa
