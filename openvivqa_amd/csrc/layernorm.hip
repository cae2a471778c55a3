// LayerNorm forward/backward (one wavefront per row, rows cached in registers),
// fused with the positional-table add of the encoder prologue and with the
// dropout mask of the residual branch in backward.  HBM-bound: vectorised
// 16-byte loads, wave-shuffle row reductions, no LDS in the forward.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int VEC = 8;        // elements per lane per chunk
constexpr int MAX_CHUNKS = 4; // D <= 64*8*4 = 2048

template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&o)[VEC]);
template <>
__device__ __forceinline__ void load8<float>(const float* p, float (&o)[VEC]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
template <>
__device__ __forceinline__ void load8<bf16>(const bf16* p, float (&o)[VEC]) {
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < VEC; i++) o[i] = (float)a[i];
}
// The same 8 elements as RAW registers: issuing the load and converting / using it are separate steps, so that a row can
// be requested one loop iteration ahead (load8 converts on the spot: a use, hence a wait, right behind the load).
template <typename T>
struct Raw8;
template <>
struct Raw8<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) {
    a = *reinterpret_cast<const float4*>(p);
    b = *reinterpret_cast<const float4*>(p + 4);
  }
  __device__ __forceinline__ void get(float (&o)[VEC]) const {
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  }
};
template <>
struct Raw8<bf16> {
  bf16x8 a;
  __device__ __forceinline__ void load(const bf16* p) { a = *reinterpret_cast<const bf16x8*>(p); }
  __device__ __forceinline__ void get(float (&o)[VEC]) const {
#pragma unroll
    for (int i = 0; i < VEC; i++) o[i] = (float)a[i];
  }
};
template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&v)[VEC]);
template <>
__device__ __forceinline__ void store8<float>(float* p, const float (&v)[VEC]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <>
__device__ __forceinline__ void store8<bf16>(bf16* p, const float (&v)[VEC]) {
  bf16x8 a;
#pragma unroll
  for (int i = 0; i < VEC; i++) a[i] = (bf16)v[i];
  *reinterpret_cast<bf16x8*>(p) = a;
}

// ---------------------------------------------------------------- forward
// R rows per wave (all R rows' loads are in flight before the first reduction), NW waves per workgroup.
template <typename TIN, typename TOUT, int CHUNKS, int R = 1, int NW = 4>
__global__ __launch_bounds__(NW * 64) void ln_fwd_kernel(const TIN* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ pos,
                                                     int pos_rows, TOUT* __restrict__ y, float* __restrict__ y32,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out, int M,
                                                     int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row0 = (xcd_contiguous_block(blockIdx.x, gridDim.x) * NW + (threadIdx.x >> 6)) * R;
  if (row0 >= M) return;
  // One round of loads: the rows, gamma, beta (and the positional rows) are all requested here, unconditionally at clamped
  // positions -- gamma / beta used to be asked for after the two reductions, a second dependent round trip per wave.
  Raw8<TIN> xraw[R][CHUNKS];
  Raw8<float> graw[CHUNKS], braw[CHUNKS], praw[R][CHUNKS];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int row = min(row0 + r, M - 1);
    const float* pr = pos ? pos + (int64_t)(row % pos_rows) * D : nullptr;
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
      const int col = min((c * 64 + lane) * VEC, D - VEC);
      xraw[r][c].load(x + (int64_t)row * D + col);
      if (pr) praw[r][c].load(pr + col);
    }
  }
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const int col = min((c * 64 + lane) * VEC, D - VEC);
    graw[c].load(gamma + col);
    braw[c].load(beta + col);
  }
  float v[R][CHUNKS][VEC];
  float s[R], sq[R], mean[R], rstd[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    s[r] = 0.f;
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
      const int col = (c * 64 + lane) * VEC;
      xraw[r][c].get(v[r][c]);
      if (col < D) {
#pragma unroll
        for (int i = 0; i < VEC; i++) s[r] += v[r][c][i];
      } else {
#pragma unroll
        for (int i = 0; i < VEC; i++) v[r][c][i] = 0.f;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++) mean[r] = wave_sum(s[r]) / (float)D;
#pragma unroll
  for (int r = 0; r < R; r++) {
    sq[r] = 0.f;
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
      const int col = (c * 64 + lane) * VEC;
      if (col < D) {
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          const float d = v[r][c][i] - mean[r];
          sq[r] += d * d;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++) rstd[r] = rsqrtf(wave_sum(sq[r]) / (float)D + eps);
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int row = row0 + r;
    if (row >= M) break;
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean[r];
      if (rstd_out) rstd_out[row] = rstd[r];
    }
    TOUT* yr = y + (int64_t)row * D;
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
      const int col = (c * 64 + lane) * VEC;
      if (col < D) {
        float g[VEC], b[VEC], o[VEC];
        graw[c].get(g);
        braw[c].get(b);
#pragma unroll
        for (int i = 0; i < VEC; i++) o[i] = (v[r][c][i] - mean[r]) * rstd[r] * g[i] + b[i];
        if (pos) {
          float p[VEC];
          praw[r][c].get(p);
#pragma unroll
          for (int i = 0; i < VEC; i++) o[i] += p[i];
        }
        store8<TOUT>(yr + col, o);
        if (y32) store8<float>(y32 + (int64_t)row * D + col, o);
      }
    }
  }
}

// ---------------------------------------------------------------- backward
// Each workgroup walks a contiguous block of rows (its waves interleaved); per-lane column
// partials of dgamma/dbeta stay in registers, are combined across the block's
// 4 waves through LDS and written to ws[block][2][D]; ln_bwd_reduce sums them.
// NWAVES = waves per workgroup: 8 for long inputs (half as many [2][D] partial rows for the deferred reduce to read: it
// streams them all, ~90 MB per MCAN step with 4), 4 otherwise.
template <typename TDY, typename TX, typename TDX, int CHUNKS, int NWAVES = 4>
__global__ __launch_bounds__(NWAVES * 64) void ln_bwd_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, TDX* __restrict__ dx,
                                                     TDY* __restrict__ dx_dropped, float* __restrict__ partial,
                                                     int M, int D, DropArgs da) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [NWAVES - 1][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const DropState ds = drop_init(da);
  float dg[CHUNKS][VEC], db[CHUNKS][VEC], g[CHUNKS][VEC];
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const int col = (c * 64 + lane) * VEC;
#pragma unroll
    for (int i = 0; i < VEC; i++) { dg[c][i] = 0.f; db[c][i] = 0.f; g[c][i] = 0.f; }
    if (col < D) load8<float>(gamma + col, g[c]);
  }
  // a workgroup owns a contiguous block of rows (its waves interleave inside it), the blocks dealt XCD-consistently:
  // XCD x ends up with rows [x M / 8, (x + 1) M / 8), the GEMM kernels' split
  const int wgp = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int row_lo = (int)((int64_t)wgp * M / (int)gridDim.x);
  const int row_hi = (int)((int64_t)(wgp + 1) * M / (int)gridDim.x);
  // The NEXT row of the wave is requested before the current one is processed (raw registers, nothing computed from them
  // until their turn): a wave owns 1-2 rows of a 6400-row input, and two dependent round trips were a third of the kernel.
  Raw8<TX> xr[CHUNKS], xn[CHUNKS];
  Raw8<TDY> dr[CHUNKS], dn[CHUNKS];
  float mu = 0.f, rs = 0.f, mun = 0.f, rsn = 0.f;
  // (unconditional loads at clamped positions: behind a per-lane or per-wave `if` the compiler no longer knows how many
  // loads are in flight and waits for all of them, the next row's included, before the current row is touched)
  auto request = [&](int row, Raw8<TX> (&xq)[CHUNKS], Raw8<TDY> (&dq)[CHUNKS], float& m, float& r) {
    m = mean[row];
    r = rstd[row];
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
      const int col = min((c * 64 + lane) * VEC, D - VEC);
      xq[c].load(x + (int64_t)row * D + col);
      dq[c].load(dy + (int64_t)row * D + col);
    }
  };
  if (row_lo + wave < row_hi) request(row_lo + wave, xr, dr, mu, rs);
  for (int row = row_lo + wave; row < row_hi; row += NWAVES) {
    const bool more = row + NWAVES < row_hi;
    request(more ? row + NWAVES : row, xn, dn, mun, rsn);  // (the last round asks for its own row again: an L2 hit)
    float xh[CHUNKS][VEC], d[CHUNKS][VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
      const int col = (c * 64 + lane) * VEC;
      if (col < D) {
        xr[c].get(xh[c]);
        dr[c].get(d[c]);
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          xh[c][i] = (xh[c][i] - mu) * rs;
          dg[c][i] += d[c][i] * xh[c][i];
          db[c][i] += d[c][i];
          d[c][i] *= g[c][i];
          s1 += d[c][i];
          s2 += d[c][i] * xh[c][i];
        }
      }
    }
    const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
      const int col = (c * 64 + lane) * VEC;
      if (col < D) {
        float o[VEC];
#pragma unroll
        for (int i = 0; i < VEC; i++) o[i] = rs * (d[c][i] - c1 - xh[c][i] * c2);
        store8<TDX>(dx + (int64_t)row * D + col, o);
        if (dx_dropped) {
          float dm[VEC];
          drop_mul8(ds, (uint32_t)row * (uint32_t)D + col, dm);  // D and col are multiples of 8: even index
#pragma unroll
          for (int i = 0; i < VEC; i++) o[i] *= dm[i];
          store8<TDY>(dx_dropped + (int64_t)row * D + col, o);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) { xr[c] = xn[c]; dr[c] = dn[c]; }
    mu = mun;
    rs = rsn;
  }
  // combine the waves of the block through LDS.  A lane's 8 columns are kept as two 16-byte halves D / 2 floats apart
  // (half h of column group j at h * D/2 + 4 j): consecutive lanes are 16 bytes apart in each half.  (Stored as one 32-byte
  // run per lane, the ds_read_b128 of the combine hit every bank twice: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50 on
  // every instantiation, profiles/r05_pmc_summary.json.)
  const int hD = D >> 1;
  if (wave > 0) {
    float* s = smem + (int64_t)(wave - 1) * 2 * D;
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
      const int j4 = (c * 64 + lane) * 4;
      if (j4 * 2 < D) {
        *reinterpret_cast<float4*>(s + j4) = make_float4(dg[c][0], dg[c][1], dg[c][2], dg[c][3]);
        *reinterpret_cast<float4*>(s + hD + j4) = make_float4(dg[c][4], dg[c][5], dg[c][6], dg[c][7]);
        *reinterpret_cast<float4*>(s + D + j4) = make_float4(db[c][0], db[c][1], db[c][2], db[c][3]);
        *reinterpret_cast<float4*>(s + D + hD + j4) = make_float4(db[c][4], db[c][5], db[c][6], db[c][7]);
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
    float* out = partial + (int64_t)blockIdx.x * 2 * D;
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
      const int col = (c * 64 + lane) * VEC;
      const int j4 = (c * 64 + lane) * 4;
      if (col < D) {
        for (int w = 0; w < NWAVES - 1; w++) {
          const float* s = smem + (int64_t)w * 2 * D;
          const float4 a0 = *reinterpret_cast<const float4*>(s + j4), a1 = *reinterpret_cast<const float4*>(s + hD + j4);
          const float4 b0 = *reinterpret_cast<const float4*>(s + D + j4), b1 = *reinterpret_cast<const float4*>(s + D + hD + j4);
          const float a[VEC] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          const float b[VEC] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
          for (int i = 0; i < VEC; i++) { dg[c][i] += a[i]; db[c][i] += b[i]; }
        }
        store8<float>(out + col, dg[c]);
        store8<float>(out + D + col, db[c]);
      }
    }
  }
}

// dgamma / dbeta (+)= column sums of the per-workgroup partials [blocks][2*D], for every queued LayerNorm of a backward
// pass in ONE launch (the deferred form) or for one LayerNorm (the immediate form).  DETERMINISTIC (round 4; rounds 1-3
// combined row groups with fp32 atomics, so two runs of the same step differed in the last bits and drifted apart): one
// workgroup owns 256 columns of one problem and sums ALL its partial rows in a fixed order -- row lane r = 0..7 takes
// rows r, r + 8, ... (16-byte loads, 4 in flight per thread), the eight lane sums are added r = 0..7 -- then writes the
// outputs, or adds to them when the problem says `accumulate`, with plain stores: no pre-zeroed outputs, no memset.
// grid (column blocks of 256 over 2*max_D, 1, problems); 512 partial rows x 1 KB per workgroup at D = 512.
__global__ __launch_bounds__(512) void ln_bwd_grouped_reduce_kernel(const ovqa_reduce_problem* __restrict__ probs) {
  __shared__ float4 red[8][64];
  const ovqa_reduce_problem pr = probs[blockIdx.z];
  const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
  const int i = (blockIdx.x * 64 + c) * 4;  // over 2*D (D % 8 == 0: a float4 never straddles dgamma | dbeta)
  const int D = pr.D;
  if (blockIdx.x * 256 >= 2 * D) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < 2 * D) {
#pragma unroll 4
    for (int b = r; b < pr.blocks; b += 8) {
      const float4 v = *reinterpret_cast<const float4*>(pr.partial + (int64_t)b * 2 * D + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[r][c] = s;
  __syncthreads();
  if (r != 0 || i >= 2 * D) return;
  float* base = (i < D) ? pr.out0 : pr.out1;
  if (base == nullptr) return;
  base += (i < D) ? i : i - D;
  float4 t = red[0][c];
#pragma unroll
  for (int w = 1; w < 8; w++) {
    const float4 a = red[w][c];
    t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
  }
  if (pr.accumulate) { t.x += base[0]; t.y += base[1]; t.z += base[2]; t.w += base[3]; }
  base[0] = t.x; base[1] = t.y; base[2] = t.z; base[3] = t.w;
}

// The immediate form: the same kernel on a one-problem table passed BY VALUE (no device table, nothing to upload).
__global__ __launch_bounds__(512) void ln_bwd_reduce_kernel(const ovqa_reduce_problem pr) {
  __shared__ float4 red[8][64];
  const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
  const int i = (blockIdx.x * 64 + c) * 4;
  const int D = pr.D;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < 2 * D) {
#pragma unroll 4
    for (int b = r; b < pr.blocks; b += 8) {
      const float4 v = *reinterpret_cast<const float4*>(pr.partial + (int64_t)b * 2 * D + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[r][c] = s;
  __syncthreads();
  if (r != 0 || i >= 2 * D) return;
  float* base = (i < D) ? pr.out0 : pr.out1;
  if (base == nullptr) return;
  base += (i < D) ? i : i - D;
  float4 t = red[0][c];
#pragma unroll
  for (int w = 1; w < 8; w++) {
    const float4 a = red[w][c];
    t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
  }
  if (pr.accumulate) { t.x += base[0]; t.y += base[1]; t.z += base[2]; t.w += base[3]; }
  base[0] = t.x; base[1] = t.y; base[2] = t.z; base[3] = t.w;
}

// One row per wave, four waves per workgroup.  (Round 5 swept rows per wave x waves per workgroup -- 1 x 2, 1 x 8, 2 x 2, 2 x 4,
// 2 x 8, 4 x 4: all equal or slower, profiles/r05_ln_bench_fwd_forms.jsonl: the launch is one latency chain, and a wave per
// row keeps the most rows in flight.)
template <typename TIN, typename TOUT>
int fwd_dispatch(const void* x, const float* gamma, const float* beta, const float* pos, int64_t pos_rows, void* y,
                 float* y32, float* mean, float* rstd, int64_t M, int64_t D, float eps, hipStream_t st) {
  const int chunks = (int)((D + 64 * VEC - 1) / (64 * VEC));
#define LN_FWD_RW(C, R, NW)                                                                                          \
  hipLaunchKernelGGL((ln_fwd_kernel<TIN, TOUT, C, R, NW>), dim3((unsigned)((M + R * NW - 1) / (R * NW))),            \
                     dim3(NW * 64), 0, st, (const TIN*)x, gamma, beta, pos, (int)(pos ? pos_rows : 1), (TOUT*)y, y32, \
                     mean, rstd, (int)M, (int)D, eps)
  switch (chunks) {
    case 1: LN_FWD_RW(1, 1, 4); break;
    case 2: LN_FWD_RW(2, 1, 4); break;
    default: LN_FWD_RW(4, 1, 4); break;
  }
#undef LN_FWD_RW
  return ovqa_check_launch("layernorm_fwd");
}

template <typename TDY, typename TX, typename TDX>
int bwd_dispatch(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                 void* dx_dropped, float* dgamma, float* dbeta, int64_t M, int64_t D, int accumulate,
                 const DropArgs& da, void* ws, hipStream_t st) {
  const int chunks = (int)((D + 64 * VEC - 1) / (64 * VEC));
  const int nblocks = ovqa::layernorm_bwd_blocks(M, D);
  const int nw = ovqa::layernorm_bwd_waves(M, D);
  OVQA_REQUIRE((int64_t)nblocks * 2 * D * 4 <= ovqa::kWorkspaceBytes, OVQA_ERR_WORKSPACE, "layernorm_bwd: ws too small");
  float* partial = (float*)ws;
  const size_t smem = (size_t)(nw - 1) * 2 * D * sizeof(float);
#define LN_BWD_W(C, W)                                                                                              \
  hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, C, W>), dim3(nblocks), dim3(W * 64), smem, st, (const TDY*)dy,    \
                     (const TX*)x, gamma, mean, rstd, (TDX*)dx, (TDY*)dx_dropped, partial, (int)M, (int)D, da)
#define LN_BWD(C)                                                                                                   \
  if (nw == 8) LN_BWD_W(C, 8);                                                                                      \
  else LN_BWD_W(C, 4);
  switch (chunks) {
    case 1: LN_BWD(1); break;
    case 2: LN_BWD(2); break;
    default: LN_BWD(4); break;
  }
#undef LN_BWD
#undef LN_BWD_W
  int rc = ovqa_check_launch("layernorm_bwd");
  if (rc != OVQA_OK) return rc;
  if (dgamma == nullptr && dbeta == nullptr) return OVQA_OK;  // partials stay in ws (deferred grouped reduce)
  const ovqa_reduce_problem pr{partial, dgamma, dbeta, nblocks, (int32_t)D, accumulate ? 1 : 0, 0};
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((unsigned)((2 * D + 255) / 256)), dim3(512), 0, st, pr);
  return ovqa_check_launch("layernorm_bwd_reduce");
}

}  // namespace

namespace ovqa {

int layernorm_fwd(int dtype, int in_dtype, const void* x, const float* gamma, const float* beta, const float* pos,
                  int64_t pos_rows, void* y, float* y32, float* mean, float* rstd, int64_t M, int64_t D, float eps,
                  hipStream_t st) {
  OVQA_REQUIRE(D % VEC == 0 && D <= 64 * VEC * MAX_CHUNKS, OVQA_ERR_UNSUPPORTED,
               "layernorm: D=%lld must be a multiple of 8 and <= 2048", (long long)D);
  if (M == 0) return OVQA_OK;
  if (dtype == OVQA_F32 && in_dtype == OVQA_F32)
    return fwd_dispatch<float, float>(x, gamma, beta, pos, pos_rows, y, y32, mean, rstd, M, D, eps, st);
  if (dtype == OVQA_BF16 && in_dtype == OVQA_BF16)
    return fwd_dispatch<bf16, bf16>(x, gamma, beta, pos, pos_rows, y, y32, mean, rstd, M, D, eps, st);
  if (dtype == OVQA_BF16 && in_dtype == OVQA_F32)
    return fwd_dispatch<float, bf16>(x, gamma, beta, pos, pos_rows, y, y32, mean, rstd, M, D, eps, st);
  ovqa_set_error("layernorm_fwd: unsupported dtype combination in=%d out=%d", in_dtype, dtype);
  return OVQA_ERR_UNSUPPORTED;
}

// Waves per workgroup of the backward: 8 for the long activations (a wave owns one or two rows of a 6400-row launch), while
// its 7 * 2 * D floats of dynamic LDS stay inside the 64 KiB a launch gets without hipFuncSetAttribute (D <= 1168); wider
// rows and short inputs keep the 4-wave form (3 * 2 * D floats: 48 KiB at 2048).
// (13 waves -- every wave exactly one row -- and 16 measured slower: 12.2 / 11.5 against 9.2-9.4 us cold, round 5: the LDS
// combine of the column partials and the lower occupancy cost more than the second row.)
int layernorm_bwd_waves(int64_t M, int64_t D) {
  if (M < 4096 || 7 * 2 * D * 4 > 64 * 1024) return 4;
  return 8;
}
int layernorm_bwd_blocks(int64_t M, int64_t D) {
  const int nw = layernorm_bwd_waves(M, D);
  // (measured in the MCAN step, 6400 rows: 512 workgroups 3.19 ms; 800 / 1024 -- one row per wave -- 3.23 / 3.24: the
  // partial rows the deferred reduce streams grow with the workgroup count; 384 / 256: 3.20-3.24 / 3.22)
  const int64_t cap = nw >= 8 ? 512 : 1024;
  int64_t nb = (M + nw - 1) / nw;
  return (int)(nb > cap ? cap : (nb < 1 ? 1 : nb));
}

int grouped_partial_reduce(const ovqa_reduce_problem* probs, int n, int max_blocks, int max_D, hipStream_t st) {
  if (n <= 0) return OVQA_OK;
  (void)max_blocks;  // (a workgroup walks all partial rows of its columns: the grid no longer depends on their number)
  dim3 grid((unsigned)((2 * max_D + 255) / 256), 1u, (unsigned)n);
  hipLaunchKernelGGL(ln_bwd_grouped_reduce_kernel, grid, dim3(512), 0, st, probs);
  return ovqa_check_launch("grouped_partial_reduce");
}

int layernorm_bwd(int dtype, int dx_dtype, const void* dy, const void* x, int x_dtype, const float* gamma,
                  const float* mean, const float* rstd, void* dx, void* dx_dropped, float* dgamma, float* dbeta,
                  int64_t M, int64_t D, int accumulate, const DropArgs& da, void* ws, hipStream_t st) {
  OVQA_REQUIRE(D % VEC == 0 && D <= 64 * VEC * MAX_CHUNKS, OVQA_ERR_UNSUPPORTED,
               "layernorm: D=%lld must be a multiple of 8 and <= 2048", (long long)D);
  OVQA_REQUIRE(ws != nullptr, OVQA_ERR_WORKSPACE, "layernorm_bwd: ws is NULL");
  if (M == 0) return OVQA_OK;
  if (dtype == OVQA_F32 && x_dtype == OVQA_F32 && dx_dtype == OVQA_F32)
    return bwd_dispatch<float, float, float>(dy, x, gamma, mean, rstd, dx, dx_dropped, dgamma, dbeta, M, D, accumulate, da, ws, st);
  if (dtype == OVQA_BF16 && x_dtype == OVQA_BF16 && dx_dtype == OVQA_BF16)
    return bwd_dispatch<bf16, bf16, bf16>(dy, x, gamma, mean, rstd, dx, dx_dropped, dgamma, dbeta, M, D, accumulate, da, ws, st);
  if (dtype == OVQA_BF16 && x_dtype == OVQA_F32 && dx_dtype == OVQA_BF16)  // fp32 pre-LN sum of the residual stream
    return bwd_dispatch<bf16, float, bf16>(dy, x, gamma, mean, rstd, dx, dx_dropped, dgamma, dbeta, M, D, accumulate, da, ws, st);
  if (dtype == OVQA_BF16 && x_dtype == OVQA_F32 && dx_dtype == OVQA_F32)
    return bwd_dispatch<bf16, float, float>(dy, x, gamma, mean, rstd, dx, dx_dropped, dgamma, dbeta, M, D, accumulate, da, ws, st);
  ovqa_set_error("layernorm_bwd: unsupported dtype combination dy=%d x=%d dx=%d", dtype, x_dtype, dx_dtype);
  return OVQA_ERR_UNSUPPORTED;
}

}  // namespace ovqa
