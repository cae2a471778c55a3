// Row-complete output tiles: the residual-epilogue GEMM of a block WITH the block's LayerNorm in its epilogue (round 6):
//     pre32[m, :] = res(m, :) + drop(x[m, :] W^T + b)          (MEpiBiasRes32: the fp32 residual stream)
//     y[m, :]     = LN(pre32[m, :]) * gamma + beta  -> bf16     (+ mean[m], rstd[m] for the backward / the next block)
// for N = 512 output features (attentions.py:58,330-331: fc_o + dropout + residual + layer_norm; positionwise_feed_forward.py
// :25-26: fc2 + dropout + residual + layer_norm).  One workgroup owns BC = 64 activation rows and ALL 512 output features, so
// the row statistics never leave the workgroup: the ln_fwd launch behind the GEMM (5.4 us of latency chain per 6400-row
// launch, 18 of them per MCAN step at these shapes) and its 13 MB read of pre32 disappear.
//
// Why this is not slower than the 128 x 128 tiles although a workgroup streams the whole weight matrix: these loops are bound
// by what a CU takes in through its L2 -> LDS path, and that rate depends on how many CUs pull at once (measured with the
// DMA-only ablation of gemm_tile256.h: 78 GB/s per CU with 256 CUs streaming, 113 with 50).  6400 rows are 100 such tiles:
// 100 CUs x 576 KB (K = 512) against 256 CUs x 228 KB for the 16-wave 128 x 128 form.
//
// Schedule: the one of gemm_tile256.h without the rotated half (the matrix pipes are idle most of a K step here): two LDS
// stages of 72 KB (W tile 512 rows | x tile 64 rows, 128-byte rows, XOR swizzle), all 16 fragments of a K step read into
// registers at its start, a second barrier releases the stage, the DMA of step k+2 goes into it while step k+1's is in flight.
// Wave w owns output features [64 w, 64 w + 64) of all 64 rows: acc[4][4].
#pragma once
#include <type_traits>

#include "common.h"

namespace ovqa_rowln {

constexpr int BC = 64, BR = 512, TK = 64;
constexpr int P_BYTES = BR * TK * 2;               // 64 KiB
constexpr int STAGE_BYTES = P_BYTES + BC * TK * 2;  // + 8 KiB
constexpr int RED_OFF = 2 * STAGE_BYTES;           // [64 rows][8 waves] floats behind the ring
constexpr int LDS_BYTES = RED_OFF + BC * 8 * 4;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

struct Args {
  const bf16* W; int64_t ldw;  // [512][K]
  const bf16* X; int64_t ldx;  // [M][K]
  int M, K;
  const float* gamma; const float* beta; float eps;  // THIS block's LayerNorm
  bf16* y; int64_t ldy;
  float* mean; float* rstd;
};

__device__ __forceinline__ int perm32(int s) { return (s & ~31) | ((s & 12) << 1) | (((s >> 4) & 1) << 2) | (s & 3); }

// all-lane-group sum for lanes with equal (lane & 15): the four 16-lane rows of the wave
__device__ __forceinline__ float sum_over_lane_groups(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

template <bool EDGE, typename Epi>
__device__ __forceinline__ void epilogue(const Args& g, Epi& epi, f32x4 (&acc)[4][4], char* smem, int c0, int wave, int lane) {
  float v[4][2][8];
  const int nq = wave * 64 + (lane >> 4) * 8;
  const int r16 = lane & 15;
  auto row = [&](int i) { return EDGE ? min(c0 + i * 16 + r16, g.M - 1) : c0 + i * 16 + r16; };  // loads: clamped on the edge tile
  auto live = [&](int i) { return !EDGE || c0 + i * 16 + r16 < g.M; };
  {
    typename Epi::ColCtx cc[2];
    typename Epi::RowCtx rc[4][2];
#pragma unroll
    for (int jp = 0; jp < 2; jp++) cc[jp] = epi.pre_cols(nq + jp * 32);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int jp = 0; jp < 2; jp++) rc[i][jp] = epi.pre_rows(row(i), nq + jp * 32);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int jp = 0; jp < 2; jp++)
        epi.value(Epi::join(cc[jp], rc[i][jp]), row(i), nq + jp * 32, acc[2 * jp][i], acc[2 * jp + 1][i], v[i][jp]);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int jp = 0; jp < 2; jp++)
        if (live(i)) epi.store(row(i), nq + jp * 32, v[i][jp]);
  }
  // LayerNorm over the 512 features of each row: two passes (mean, then the centred squares), as ln_fwd_kernel
  float* red = reinterpret_cast<float*>(smem + RED_OFF);  // [64 rows][8 waves]
  float gm[2][8], bt[2][8];
#pragma unroll
  for (int jp = 0; jp < 2; jp++) {
    const float4 g0 = *reinterpret_cast<const float4*>(g.gamma + nq + jp * 32), g1 = *reinterpret_cast<const float4*>(g.gamma + nq + jp * 32 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(g.beta + nq + jp * 32), b1 = *reinterpret_cast<const float4*>(g.beta + nq + jp * 32 + 4);
    gm[jp][0] = g0.x; gm[jp][1] = g0.y; gm[jp][2] = g0.z; gm[jp][3] = g0.w; gm[jp][4] = g1.x; gm[jp][5] = g1.y; gm[jp][6] = g1.z; gm[jp][7] = g1.w;
    bt[jp][0] = b0.x; bt[jp][1] = b0.y; bt[jp][2] = b0.z; bt[jp][3] = b0.w; bt[jp][4] = b1.x; bt[jp][5] = b1.y; bt[jp][6] = b1.z; bt[jp][7] = b1.w;
  }
  float mean[4], rstd[4];
  float part[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    float s = 0.f;
#pragma unroll
    for (int jp = 0; jp < 2; jp++)
#pragma unroll
      for (int t = 0; t < 8; t++) s += v[i][jp][t];
    part[i] = sum_over_lane_groups(s);
  }
  if (lane < 16) {
#pragma unroll
    for (int i = 0; i < 4; i++) red[(i * 16 + lane) * 8 + wave] = part[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float4 a = *reinterpret_cast<const float4*>(red + (i * 16 + r16) * 8);
    const float4 b = *reinterpret_cast<const float4*>(red + (i * 16 + r16) * 8 + 4);
    mean[i] = (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) * (1.f / BR);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; i++) {
    float s = 0.f;
#pragma unroll
    for (int jp = 0; jp < 2; jp++)
#pragma unroll
      for (int t = 0; t < 8; t++) {
        const float d = v[i][jp][t] - mean[i];
        s += d * d;
      }
    part[i] = sum_over_lane_groups(s);
  }
  if (lane < 16) {
#pragma unroll
    for (int i = 0; i < 4; i++) red[(i * 16 + lane) * 8 + wave] = part[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float4 a = *reinterpret_cast<const float4*>(red + (i * 16 + r16) * 8);
    const float4 b = *reinterpret_cast<const float4*>(red + (i * 16 + r16) * 8 + 4);
    rstd[i] = rsqrtf((((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) * (1.f / BR) + g.eps);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int jp = 0; jp < 2; jp++) {
      bf16x8 o;
#pragma unroll
      for (int t = 0; t < 8; t++) o[t] = (bf16)((v[i][jp][t] - mean[i]) * rstd[i] * gm[jp][t] + bt[jp][t]);
      if (live(i)) *reinterpret_cast<bf16x8*>(g.y + (int64_t)row(i) * g.ldy + nq + jp * 32) = o;
    }
  }
  if (wave == 0 && lane < 16) {
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (live(i)) {
        g.mean[row(i)] = mean[i];
        g.rstd[row(i)] = rstd[i];
      }
  }
}

// Epi: MEpiBiasRes32 (gemm_mfma.hip): pre_cols(n) / pre_rows(m, n) = the loads of a piece of 8 features, value(ctx, m, n, lo, hi,
// r) = the 8 fp32 values of the pre-LN sum, store(m, n, r) = their store to pre32
template <typename Epi>
__global__ __launch_bounds__(512) void kernel(Args g, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c0 = blockIdx.x * BC;  // consecutive tiles go to different XCDs: every L2 holds W once, x rows are read once

  f32x4 acc[4][4];  // [j: 16 features][i: 16 rows]
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- DMA: 64 pieces of the W tile (8 per wave) + 8 of the x tile (1 per wave); piece = 8 LDS rows
  const int swz = ((lane & 7) ^ (lane >> 3)) << 4;
  uint32_t vw[8], vx;
#pragma unroll
  for (int i = 0; i < 8; i++) vw[i] = (uint32_t)((int64_t)perm32((wave * 8 + i) * 8 + (lane >> 3)) * g.ldw * 2) + swz;
  {
    const int xr = min(c0 + wave * 8 + (lane >> 3), g.M - 1) - c0;
    vx = (uint32_t)((int64_t)xr * g.ldx * 2) + swz;
  }
  const char* wbase = reinterpret_cast<const char*>(g.W);
  const char* xbase = reinterpret_cast<const char*>(g.X + (int64_t)c0 * g.ldx);
  auto dma = [&](int kt) {
    char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
    for (int n = 0; n < 9; n++) {
      uint32_t o = n < 8 ? vw[n & 7] : vx;
      asm volatile("" : "+v"(o));
      const char* src = (n < 8 ? wbase : xbase) + (int64_t)kt * (TK * 2) + o;
      char* dst = st + (n < 8 ? (wave * 8 + n) * 1024 : P_BYTES + wave * 1024);
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)dst, 16, 0, 0);
    }
  };
  // ---- fragments
  const int frow = lane & 15;
  const int fo0 = frow * 128 + (((lane >> 4) ^ (lane & 7)) << 4);
  const int fo1 = frow * 128 + (((4 + (lane >> 4)) ^ (lane & 7)) << 4);
  const int poff = wave * (64 * 128), qoff = P_BYTES;
  bf16x8 pf0[4], qf0[4], pf1[4], qf1[4];

  const int nkt = g.K / TK;
  dma(0);
  if (nkt > 1) dma(1);
  for (int kt = 0; kt < nkt; kt++) {
    if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // S: stage kt has landed
    const char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; i++) qf0[i] = *reinterpret_cast<const bf16x8*>(st + qoff + fo0 + i * 2048);
#pragma unroll
    for (int j = 0; j < 4; j++) pf0[j] = *reinterpret_cast<const bf16x8*>(st + poff + fo0 + j * 2048);
#pragma unroll
    for (int i = 0; i < 4; i++) qf1[i] = *reinterpret_cast<const bf16x8*>(st + qoff + fo1 + i * 2048);
#pragma unroll
    for (int j = 0; j < 4; j++) pf1[j] = *reinterpret_cast<const bf16x8*>(st + poff + fo1 + j * 2048);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // M: stage kt is in registers everywhere
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 2 < nkt) dma(kt + 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int i = 0; i < 4; i++) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf0[j], qf0[i], acc[j][i], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int i = 0; i < 4; i++) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf1[j], qf1[i], acc[j][i], 0, 0, 0);
  }

  // ---- epilogue.  Lane (n = lane & 15, q = lane >> 4) holds, for each of its 4 rows (i), 16 features: pieces jp = 0, 1 of
  // 8 consecutive features n0 = wave * 64 + jp * 32 + q * 8 (perm32 staging of W, as in the other kernels).
  // Tiles inside the matrix run WITHOUT per-lane guards: behind a guard the compiler waits for the previous store's
  // acknowledgement before the next piece (measured on gemm_tile256.h: 5.4 -> 3.9 us of epilogue; here 28 us -> see the header).
  epi.init();
  if (c0 + BC <= g.M) epilogue<false>(g, epi, acc, smem, c0, wave, lane);
  else epilogue<true>(g, epi, acc, smem, c0, wave, lane);
}

}  // namespace ovqa_rowln
