// bf16 MFMA GEMM family for gfx950 (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
// One kernel template computes  Out(c, r) = sum_k Q(c, k) * P(r, k)  for a 128(c) x 128(r) tile per
// 256-thread workgroup (4 waves as 2x2, 64x64 per wave, 64 accumulator VGPRs per lane).  `r` is the
// CONTIGUOUS output dimension: P is the MFMA A operand, so a lane's 4 accumulator registers are 4
// consecutive r of one c and every epilogue access is an 8/16-byte vector.
//
// Each operand is either k-contiguous in memory ([row][k], row stride ld) or k-major ([k][row]):
//
//   forward   Y[m,n]   = sum_k X[m,k] W[n,k]      c=m  r=n   P=W  k-contiguous   Q=X   k-contiguous
//   dX        dX[m,i]  = sum_n dY[m,n] W[n,i]     c=m  r=i   P=W  k-major        Q=dY  k-contiguous
//   dW        dW[n,i]  = sum_m dY[m,n] X[m,i]     c=n  r=i   P=X  k-major        Q=dY  k-major
//
// so nn.Linear's [out,in] weights can serve all three products without a transposed copy.  In practice the
// training path computes dX from a TRANSPOSED weight copy (mfma_linear_bwd_data_wt: P = W^T k-contiguous): with
// direct-to-LDS staging the k-major weight tile was ~2x slower per K step in the MCAN step.
// The direct-to-LDS kernels with a k-contiguous P stage P's rows in a permuted order (perm32) so that a lane's
// accumulators cover 8 consecutive r: 16-byte epilogue accesses.
//   * k-contiguous tiles live in LDS as [128 rows][64 k] (128 B rows), 16-byte chunk index XOR (row&7):
//     ds_write_b128 staging and ds_read_b128 fragment reads are bank-conflict free.
//   * k-major tiles live as [64 k][128 rows] (256 B rows) and fragments are fetched with the gfx950
//     hardware-transposing read ds_read_b64_tr_b16 (4 k-rows x 16 columns per 16-lane group); the
//     32-byte pair index is XOR-ed with (k&3)|((k>>3)&1)<<2, which makes the 8 k-rows a half-wave
//     touches land on 8 distinct 32-byte slots of the 256-byte bank row (conflict free), and keeps the
//     staging ds_write_b128 conflict free too.
//   * global -> registers -> LDS staging, LDS double-buffered: loads of tile t+1 are issued before the
//     MFMAs of tile t and written after them; one barrier per K-tile.  Loads are unconditional
//     (clamped addresses + zero select), fully unrolled: everything stays in VGPRs.
//   * blockIdx is remapped so that the workgroups sharing a Q row-panel run on the same XCD.
#include <stdlib.h>

#include "common.h"
#include "gemm_tile256.h"
#include "kernels.h"

// Phase probe (development only, -DOVQA_PHASE_PROBE; see attention_mfma.hip): wall-clock stamps of thread 0 of the first
// and the middle workgroup at marked points of gemm_tile_glds, read back by ovqa_debug_probe_gemm().
#ifdef OVQA_PHASE_PROBE
__device__ unsigned long long g_probe_gemm[2][16];
// per-workgroup timeline of the last launch: {start, K loop done, end, HW_ID | XCC_ID << 32} (scripts/gemm_wg_timeline.py)
__device__ unsigned long long g_probe_wg[4096][4];
#define OVQA_GPROBE(i)                                                                   \
  do {                                                                                   \
    if (threadIdx.x == 0) {                                                              \
      const unsigned long long t_ = wall_clock64();                                      \
      if (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)                                \
        g_probe_gemm[blockIdx.x == 0 ? 0 : 1][i] = t_;                                   \
      if (blockIdx.x < 4096 && ((i) == 0 || (i) == 3 || (i) == 4)) {                     \
        g_probe_wg[blockIdx.x][(i) == 0 ? 0 : (i) == 3 ? 1 : 2] = t_;                    \
        if ((i) == 0)                                                                    \
          g_probe_wg[blockIdx.x][3] =                                                    \
              (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |            \
              ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);    \
      }                                                                                  \
    }                                                                                    \
  } while (0)
extern "C" int ovqa_debug_probe_gemm(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe_gemm), sizeof(g_probe_gemm));
}
extern "C" int ovqa_debug_probe_gemm_wgs(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe_wg), sizeof(unsigned long long) * 4 * (size_t)n);
}
#else
#define OVQA_GPROBE(i) do {} while (0)
#endif

namespace {

constexpr int BT = 128, BK = 64;
constexpr int TILE_BYTES = BT * BK * 2;  // 16 KiB per operand tile

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

struct GemmArgs {
  const bf16* P; int64_t ldp;  // r operand
  const bf16* Q; int64_t ldq;  // c operand
  int R, C, K;
  int tiles_r, tiles_c;
  int c_step;  // rows of the c operand a tile OWNS (0 = the tile's full height): see the 16-wave form in launch()
};

__device__ __forceinline__ int kc_off(int row, int chunk) { return (row * 8 + (chunk ^ (row & 7))) * 16; }
__device__ __forceinline__ int km_off(int krow, int chunk16) {
  const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
  return krow * 256 + (((((chunk16 >> 1) ^ f) << 1) | (chunk16 & 1)) << 4);
}

// ---- staging: 4 x 16 B per thread per operand ------------------------------------------------
template <bool KMAJOR>
__device__ __forceinline__ void stage_load(uint4 (&r)[4], const bf16* __restrict__ Op, int64_t ld, int row0, int rows,
                                           int k0, int K, int tid) {
  if (!KMAJOR) {
    const int chunk = tid & 7, rb = tid >> 3;
    const int k = k0 + chunk * 8;
    const bool kok = k < K;
    const int kk = kok ? k : 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int row = row0 + rb + 32 * i;
      row = row < rows ? row : rows - 1;
      uint4 v = *reinterpret_cast<const uint4*>(Op + (int64_t)row * ld + kk);
      r[i] = kok ? v : make_uint4(0u, 0u, 0u, 0u);
    }
  } else {
    const int chunk = tid & 15, kb = tid >> 4;
    int col = row0 + chunk * 8;
    col = col <= rows - 8 ? col : rows - 8;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int k = k0 + kb + 16 * i;
      const bool kok = k < K;
      uint4 v = *reinterpret_cast<const uint4*>(Op + (int64_t)(kok ? k : 0) * ld + col);
      r[i] = kok ? v : make_uint4(0u, 0u, 0u, 0u);
    }
  }
}

template <bool KMAJOR>
__device__ __forceinline__ void stage_store(const uint4 (&r)[4], char* tile, int tid) {
  if (!KMAJOR) {
    const int chunk = tid & 7, rb = tid >> 3;
#pragma unroll
    for (int i = 0; i < 4; i++) *reinterpret_cast<uint4*>(tile + kc_off(rb + 32 * i, chunk)) = r[i];
  } else {
    const int chunk = tid & 15, kb = tid >> 4;
#pragma unroll
    for (int i = 0; i < 4; i++) *reinterpret_cast<uint4*>(tile + km_off(kb + 16 * i, chunk)) = r[i];
  }
}

// ---- fragment fetch: 16 rows (base .. base+15) x 32 k (k-step ks) -> 8 bf16 per lane --------------
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 frag(const char* tile, int base, int ks, int lane) {
  if (!KMAJOR) {
    return *reinterpret_cast<const bf16x8*>(tile + kc_off(base + (lane & 15), ks * 4 + (lane >> 4)));
  } else {
    const int kr = ks * 32 + 8 * (lane >> 4) + ((lane >> 2) & 3);
    const int ch = (base >> 3) + ((lane & 3) >> 1);
    const int sub = (lane & 1) * 8;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + km_off(kr, ch) + sub));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + km_off(kr + 4, ch) + sub));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  }
}

// The same k-major fragment as two inline-asm reads (lo: k-rows kr.., hi: kr + 4..; see frag<true>) that the compiler does
// not track: behind a pending LDS-DMA it puts `s_waitcnt vmcnt(0)` in front of the ds_read_b64_tr_b16 BUILTIN (it cannot
// tell the ring slots apart), which drains a direct-to-LDS ring in every K step.  The caller waits (`s_waitcnt
// lgkmcnt(0)`) and ties the pieces to that wait before joining and using them.
__device__ __forceinline__ bf16x4 lds_read_tr(uint32_t addr) {
  bf16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ bf16x4 lds_read_tr_1k(uint32_t addr) {
  bf16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ uint32_t lds_offset(const char* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void frag_tr_nowait(const char* tile, int base, int ks, int lane, bf16x4& lo, bf16x4& hi) {
  const int kr = ks * 32 + 8 * (lane >> 4) + ((lane >> 2) & 3);
  const int ch = (base >> 3) + ((lane & 3) >> 1);
  const uint32_t a = lds_offset(tile) + km_off(kr, ch) + (lane & 1) * 8;
  lo = lds_read_tr(a);
  hi = lds_read_tr_1k(a);
}
__device__ __forceinline__ bf16x8 join8(const bf16x4& lo, const bf16x4& hi) {
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// one 128x128 output tile at (c0, r0): the whole K loop + epilogue.
// COLSUM: the tile at r0 == 0 also reduces the Q operand over k (bias gradient = column sums of dY) with
// one extra MFMA per c sub-tile against an all-ones A fragment -- the sums come out of the matrix pipe,
// no LDS traffic, no cross-lane reduction; epi.colsum(c, value) receives them.
template <bool P_KMAJOR, bool Q_KMAJOR, typename Epi, bool COLSUM = false>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, int c0, int r0, Epi& epi, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave >> 1, wr = wave & 1;

  f32x4 acc[4][4];  // [j: r sub-tile][i: c sub-tile]
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  bool do_colsum = false;
  if constexpr (COLSUM) do_colsum = r0 == 0 && wr == 0 && epi.wants_colsum();
  f32x4 cs[4];
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 4; i++) cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 8; e++) ones[e] = (bf16)1.0f;

  const int nkt = (g.K + BK - 1) / BK;
  uint4 rp[4], rq[4];
  stage_load<P_KMAJOR>(rp, g.P, g.ldp, r0, g.R, 0, g.K, tid);
  stage_load<Q_KMAJOR>(rq, g.Q, g.ldq, c0, g.C, 0, g.K, tid);
  stage_store<P_KMAJOR>(rp, smem, tid);
  stage_store<Q_KMAJOR>(rq, smem + TILE_BYTES, tid);
  __syncthreads();

  for (int kt = 0; kt < nkt; kt++) {
    char* Ps = smem + (kt & 1) * 2 * TILE_BYTES;
    char* Qs = Ps + TILE_BYTES;
    char* Pn = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
    // prefetch of the next tile (the last iteration re-loads a valid tile that is never consumed)
    const int kn = (kt + 1 < nkt ? kt + 1 : kt) * BK;
    stage_load<P_KMAJOR>(rp, g.P, g.ldp, r0, g.R, kn, g.K, tid);
    stage_load<Q_KMAJOR>(rq, g.Q, g.ldq, c0, g.C, kn, g.K, tid);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      bf16x8 pf[4], qf[4];
#pragma unroll
      for (int j = 0; j < 4; j++) pf[j] = frag<P_KMAJOR>(Ps, wr * 64 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; i++) qf[i] = frag<Q_KMAJOR>(Qs, wc * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 4; i++)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[j], qf[i], acc[j][i], 0, 0, 0);
      if constexpr (COLSUM) {
        if (do_colsum) {
#pragma unroll
          for (int i = 0; i < 4; i++) cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, qf[i], cs[i], 0, 0, 0);
        }
      }
    }
    if (kt + 1 < nkt) {
      stage_store<P_KMAJOR>(rp, Pn, tid);
      stage_store<Q_KMAJOR>(rq, Pn + TILE_BYTES, tid);
    }
    __syncthreads();
  }

  if constexpr (COLSUM) {
    if (do_colsum && lane < 16) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int c = c0 + wc * 64 + i * 16 + lane;
        if (c < g.C) epi.colsum(c, cs[i][0]);
      }
    }
  }
  // acc[j][i][reg]: r = r0 + wr*64 + j*16 + (lane>>4)*4 + reg ; c = c0 + wc*64 + i*16 + (lane&15)
  epi.init();
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = c0 + wc * 64 + i * 16 + (lane & 15);
    if (c >= g.C) continue;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int r = r0 + wr * 64 + j * 16 + (lane >> 4) * 4;
      if (r < g.R) epi(c, r, acc[j][i]);
    }
  }
}


// ---- direct-to-LDS staging (global_load_lds_dwordx4): no VGPR round trip, no ds_write -------------------
// One wave-instruction writes 1 KiB linearly (LDS base is wave-uniform, lane l lands at +16 l), so the LDS
// image stays linear and the XOR swizzle is applied to the per-lane SOURCE address (same involution as
// the swizzled fragment reads).  Requires full K tiles (K % 64 == 0); rows are clamped per lane.
// NW = 4 or 8 waves per 128x128 tile: with 8 waves (4 along c x 2 along r, 32x64 each) every wave issues half
// the loads and half the MFMAs, so inside ONE workgroup a wave's load-issue bubble (~100 cycles per
// LDS-DMA instruction) is covered by its SIMD partner's MFMAs -- what matters when the grid is too small
// to put two workgroups on a CU (most launches of this model: M = 1280 or N = 512).
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// PERM (row-major P operand of the wide-epilogue kernels): LDS row slot s of every 32-row group holds GLOBAL row
//   perm32(s) = 8*((s>>2)&3) + 4*(s>>4) + (s&3),
// so the two 16-row MFMA tiles of a slot group cover global rows {8g..8g+3} and {8g+4..8g+7} in accumulator
// row-group g: a lane ends up with 8 CONSECUTIVE output features (one 16-byte access per epilogue tensor instead
// of two 8-byte ones).  Only the source address of the direct-to-LDS load changes; the LDS image, its swizzle
// and the fragment reads are untouched.
__device__ __forceinline__ int perm32(int s) { return (s & ~31) | ((s & 12) << 1) | (((s >> 4) & 1) << 2) | (s & 3); }

template <bool KMAJOR, int NW, int NCHUNK = 16, bool PERM = false>
__device__ __forceinline__ void stage_glds(char* tile, const bf16* __restrict__ Op, int64_t ld, int row0, int rows,
                                           int k0, int lane, int wave) {
  static_assert(!KMAJOR || NCHUNK == 16, "k-major images are always 128 wide");
  static_assert(!(KMAJOR && PERM), "the row permutation is for row-major images");
  constexpr int PER = NCHUNK / NW;  // 1 KiB chunks of the tile per wave
#pragma unroll
  for (int i = 0; i < PER; i++) {
    const int ci = wave * PER + i;
    const bf16* src;
    if (!KMAJOR) {
      const int row = ci * 8 + (lane >> 3), pos = lane & 7;
      int grow = row0 + (PERM ? perm32(row) : row);
      grow = grow < rows ? grow : rows - 1;
      src = Op + (int64_t)grow * ld + k0 + ((pos ^ (row & 7)) << 3);
    } else {
      const int krow = ci * 4 + (lane >> 4), pos = lane & 15;
      const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
      const int c = ((((pos >> 1) ^ f) << 1) | (pos & 1));
      int col = row0 + c * 8;
      col = col <= rows - 8 ? col : rows - 8;
      src = Op + (int64_t)(k0 + krow) * ld + col;
    }
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(tile + ci * 1024), 16, 0, 0);
  }
}

// one 1 KiB piece (8 rows) of a row-major operand tile: piece ci of stage_glds<false, ...>
__device__ __forceinline__ void stage_glds_piece(char* tile, const bf16* __restrict__ Op, int64_t ld, int row0, int rows, int k0,
                                                 int lane, int ci) {
  const int row = ci * 8 + (lane >> 3), pos = lane & 7;
  int grow = row0 + row;
  grow = grow < rows ? grow : rows - 1;
  __builtin_amdgcn_global_load_lds((gbl_void*)(Op + (int64_t)grow * ld + k0 + ((pos ^ (row & 7)) << 3)),
                                   (lds_void*)(tile + ci * 1024), 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// BC = rows of the c (Q) operand per tile: 128, or 64 for problems with few tiles (more workgroups, fewer
// bytes per K-step and CU: the per-CU L2->LDS path, not the matrix pipe, bounds these kernels).
// BR = rows of the r (P) operand per tile: 128, or 64 with 4 waves and BC = 32 (the M = 1280 products: twice the
// workgroups again).
template <bool P_KMAJOR, bool Q_KMAJOR, typename Epi, bool COLSUM, int NBUF, int NW, int BC = 128, int BR = 128>
__device__ __forceinline__ void gemm_tile_glds(const GemmArgs& g, int c0, int r0, Epi& epi, char* smem) {
  static_assert(BC == 128 || ((BC == 64 || BC == 32) && NW == 8 && !Q_KMAJOR) || ((BC == 32 || BC == 64) && NW == 4 && !Q_KMAJOR),
                "unsupported tile");
  static_assert(NW == 4 || NW == 8 || (NW == 16 && BC == 128 && BR == 128), "4, 8 or (128 x 128) 16 waves");
  static_assert(BR == 128 || (BR == 64 && NW == 4 && (BC == 32 || BC == 64) && !P_KMAJOR), "unsupported tile");
  constexpr int NWT = NW;
  constexpr int WC = NWT == 16 ? 4 : (NWT == 8 ? (BC >= 64 ? 4 : BC / 16) : 2);  // wave grid: WC along c x WR along r
  constexpr int WR = NWT / WC;
  constexpr int NI = BC / (16 * WC);   // 16-wide c sub-tiles per wave
  constexpr int NJ = BR / (16 * WR);   // 16-wide r sub-tiles per wave
  constexpr int PCH = BR / 8;          // 1 KiB chunks of the P tile
  constexpr int QCH = BC / 8;          // 1 KiB chunks of the Q tile
  constexpr int QPER = (QCH + NW - 1) / NW;
  constexpr int PT_BYTES = BR * BK * 2;
  constexpr int STAGE = PT_BYTES + BC * BK * 2;  // bytes of one ring stage (P tile + Q tile)
  constexpr int LOADS = PCH / NW + QPER;  // LDS-DMA instructions per (loading) wave per K tile
  constexpr bool WIDE = !P_KMAJOR && Epi::kWide;  // 8 consecutive r per lane (see perm32)
  static_assert(!WIDE || NJ % 2 == 0, "wide epilogue pairs the r sub-tiles");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave / WR, wr = wave % WR;

  f32x4 acc[NJ][NI];
#pragma unroll
  for (int j = 0; j < NJ; j++)
#pragma unroll
    for (int i = 0; i < NI; i++) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bool do_colsum = false;
  if constexpr (COLSUM) do_colsum = r0 == 0 && wr == 0 && epi.wants_colsum();
  f32x4 cs[NI];
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < NI; i++) cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 8; e++) ones[e] = (bf16)1.0f;

  const int nkt = g.K / BK;
  OVQA_GPROBE(0);
  // A tile may OWN fewer c rows than it is high (c_step < BC, the 16-wave form: 6400 rows as 64 tiles of 100 instead of 50
  // of 128, so that 4 x 64 = 256 tiles cover all 256 CUs): the waves whose 8-row piece lies past the owned rows load
  // nothing (those LDS rows stay whatever they were: every output row depends on its own c row only, and the epilogue
  // drops the rows past c_hi).
  // (the 8-wave form with a ring of 2 can do the same -- its waits are vmcnt(0) -- but the host never asks it to)
  constexpr bool CSTEP = !Q_KMAJOR && BC == 128 && (NW == 16 || (NW == 8 && NBUF == 2));
  constexpr int QPW = QCH / NW;  // Q pieces per wave (CSTEP: 1 or 2)
  const int c_rows = (CSTEP && g.c_step) ? g.c_step : BC;
  const int c_hi = min(g.C, c0 + c_rows);
  const int uwave = __builtin_amdgcn_readfirstlane(wave);  // (scalar conditions below)
  const bool q_live = !CSTEP || uwave * QPW * 8 < c_rows;  // the wave's first piece is owned
  auto issue = [&](int kt) {
    char* buf = smem + (kt % NBUF) * STAGE;
    stage_glds<P_KMAJOR, NW, PCH, WIDE>(buf, g.P, g.ldp, r0, g.R, kt * BK, lane, wave);
    if constexpr (CSTEP) {
#pragma unroll
      for (int i = 0; i < QPW; i++) {
        const int ci = uwave * QPW + i;
        if (ci * 8 < c_rows) stage_glds_piece(buf + PT_BYTES, g.Q, g.ldq, c0, g.C, kt * BK, lane, ci);
      }
    } else if constexpr (QCH >= NW) {
      stage_glds<Q_KMAJOR, NW, QCH>(buf + PT_BYTES, g.Q, g.ldq, c0, g.C, kt * BK, lane, wave);
    } else {  // fewer Q chunks than waves (BC = 32): the first QCH waves load one chunk each
      if (wave < QCH) stage_glds<Q_KMAJOR, QCH, QCH>(buf + PT_BYTES, g.Q, g.ldq, c0, g.C, kt * BK, lane, wave);
    }
  };
#pragma unroll
  for (int p = 0; p < NBUF - 1; p++)
    if (p < nkt) issue(p);
  OVQA_GPROBE(1);

  for (int kt = 0; kt < nkt; kt++) {
    if (kt == 1) OVQA_GPROBE(2);
    // tile kt has landed (all but the youngest NBUF-2 tiles' loads are done), and -- after the barrier --
    // every wave has finished reading the buffer that the next issue overwrites
    if (NBUF == 2 || kt + NBUF - 2 >= nkt) {
      wait_vmcnt<0>();
    } else if constexpr (CSTEP) {  // (NW == 16 here: one Q piece per wave, or none)
      if (q_live) wait_vmcnt<LOADS * (NBUF - 2)>();
      else wait_vmcnt<(PCH / NW) * (NBUF - 2)>();
    } else if constexpr (QCH >= NW) {
      wait_vmcnt<LOADS * (NBUF - 2)>();
    } else {  // waves >= QCH issue no Q loads: their count per stage is smaller (wave-uniform branch)
      if (wave < QCH) wait_vmcnt<LOADS * (NBUF - 2)>();
      else wait_vmcnt<(PCH / NW) * (NBUF - 2)>();
    }
    __builtin_amdgcn_s_barrier();
    if (kt + NBUF - 1 < nkt) issue(kt + NBUF - 1);
    const char* Ps = smem + (kt % NBUF) * STAGE;
    const char* Qs = Ps + PT_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      bf16x8 pf[NJ], qf[NI];
      if constexpr (P_KMAJOR && Q_KMAJOR) {  // (the grouped dW: untracked reads, one hand-placed wait -- see frag_tr_nowait)
        bf16x4 plo[NJ], phi[NJ], qlo[NI], qhi[NI];
#pragma unroll
        for (int j = 0; j < NJ; j++) frag_tr_nowait(Ps, wr * (NJ * 16) + j * 16, ks, lane, plo[j], phi[j]);
#pragma unroll
        for (int i = 0; i < NI; i++) frag_tr_nowait(Qs, wc * (NI * 16) + i * 16, ks, lane, qlo[i], qhi[i]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          asm volatile("" : "+v"(plo[j]), "+v"(phi[j])::"memory");
          pf[j] = join8(plo[j], phi[j]);
        }
#pragma unroll
        for (int i = 0; i < NI; i++) {
          asm volatile("" : "+v"(qlo[i]), "+v"(qhi[i])::"memory");
          qf[i] = join8(qlo[i], qhi[i]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < NJ; j++) pf[j] = frag<P_KMAJOR>(Ps, wr * (NJ * 16) + j * 16, ks, lane);
#pragma unroll
        for (int i = 0; i < NI; i++) qf[i] = frag<Q_KMAJOR>(Qs, wc * (NI * 16) + i * 16, ks, lane);
      }
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int i = 0; i < NI; i++)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[j], qf[i], acc[j][i], 0, 0, 0);
      if constexpr (COLSUM) {
        if (do_colsum) {
#pragma unroll
          for (int i = 0; i < NI; i++) cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, qf[i], cs[i], 0, 0, 0);
        }
      }
    }
  }
  OVQA_GPROBE(3);
  if constexpr (COLSUM) {
    if (do_colsum && lane < 16) {
#pragma unroll
      for (int i = 0; i < NI; i++) {
        const int c = c0 + wc * (NI * 16) + i * 16 + lane;
        if (c < g.C) epi.colsum(c, cs[i][0]);
      }
    }
  }
  epi.init();
  // (Round 6: every load of the wave's pieces first -- Epi::pre() with clamped addresses -- and then the guarded stores, as
  // the 256 x 256 tile does it, measured no gain here: 3.025-3.039 against 3.00-3.02 ms per step, three alternations on one
  // box.  With 2-4 pieces per lane and a second workgroup on the CU the store round trips behind the per-lane branches are
  // covered; on the 256 x 256 tile, 16 pieces per lane and nothing beside it, they were 5.4 us of 21.)
#pragma unroll
  for (int i = 0; i < NI; i++) {
    const int c = c0 + wc * (NI * 16) + i * 16 + (lane & 15);
    if (c >= c_hi) continue;
    if constexpr (WIDE) {
#pragma unroll
      for (int jp = 0; jp < NJ / 2; jp++) {
        const int r = r0 + wr * (NJ * 16) + jp * 32 + (lane >> 4) * 8;
        if (r < g.R) epi.wide(c, r, acc[2 * jp][i], acc[2 * jp + 1][i]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const int r = r0 + wr * (NJ * 16) + j * 16 + (lane >> 4) * 4;
        if (r < g.R) epi(c, r, acc[j][i]);
      }
    }
  }
  OVQA_GPROBE(4);
}

template <bool P_KMAJOR, bool Q_KMAJOR, typename Epi, int NBUF, int NW, int BC = 128, int BR = 128>
__global__ __launch_bounds__(NW * 64) void gemm_bf16_glds_kernel(GemmArgs g, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nwg = g.tiles_r * g.tiles_c;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, rem = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  const int tc = bid / g.tiles_r, tr = bid % g.tiles_r;
  gemm_tile_glds<P_KMAJOR, Q_KMAJOR, Epi, false, NBUF, NW, BC, BR>(g, tc * (g.c_step ? g.c_step : BC), tr * BR, epi, smem);
}

// ---- skinny products (a decoding step: M = batch * beam <= 256 activation rows against a whole weight matrix) ----------
// One 16 x 16 output tile per wave, no LDS staging: every lane loads its own MFMA operand pieces (16 B: row lane & 15,
// k-group lane >> 4) of BOTH operands straight from global memory -- all of them up front (<= 16 K steps of 32 per wave,
// 128 VGPRs), so the wave pays ONE memory round trip, then issues its MFMAs back to back.  Longer reductions are split
// over up to 4 waves of the workgroup (K = 2048: 4 x 512) and summed through LDS.  512 x 512 weights against 192 rows are
// 32 x 12 = 384 one-wave workgroups: every CU streams its slice of the weights once (the 12 row tiles of a weight slab
// hit in L2).  The tiled kernels put such a product on 48 workgroups that each walk 8 K steps behind barriers: 5-7 us
// per launch in a decoding step where this form takes ~3.
// acc (valid on wave 0 only; returns false on the other waves) = the 16 x 16 tile  P[r0.., :] Q[c0.., :]^T
template <int S>
__device__ __forceinline__ bool skinny_tile(const bf16* __restrict__ P, int64_t ldp, int R, const bf16* __restrict__ Q,
                                            int64_t ldq, int C, int K, int r0, int c0, f32x4& acc) {
  constexpr int KCH = 16;  // K steps (of 32) per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int pr = r0 + (lane & 15), qc = c0 + (lane & 15);
  pr = pr < R ? pr : R - 1;
  qc = qc < C ? qc : C - 1;
  const int nk = K / 32;
  const int per = (nk + S - 1) / S;      // K steps of this wave: [k_lo, k_hi)
  const int k_lo = wave * per, k_hi = min(nk, k_lo + per);
  const bf16* pp = P + (int64_t)pr * ldp + (lane >> 4) * 8;
  const bf16* qp = Q + (int64_t)qc * ldq + (lane >> 4) * 8;
  bf16x8 pa[KCH], qa[KCH];
#pragma unroll
  for (int i = 0; i < KCH; i++) {  // (past the wave's share: re-read its last step, zeroed below -- no branch around a load)
    const int ks = min(k_lo + i, nk - 1);
    pa[i] = *reinterpret_cast<const bf16x8*>(pp + ks * 32);
    qa[i] = *reinterpret_cast<const bf16x8*>(qp + ks * 32);
  }
  __builtin_amdgcn_sched_barrier(0);  // all 32 loads in flight before the first MFMA waits (the scheduler would
  acc = f32x4{0.f, 0.f, 0.f, 0.f};    // otherwise interleave them ~10 deep: three round trips instead of one)
#pragma unroll
  for (int i = 0; i < KCH; i++) {
    bf16x8 a = pa[i];
    if (k_lo + i >= k_hi) {
#pragma unroll
      for (int e = 0; e < 8; e++) a[e] = (bf16)0.f;
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qa[i], acc, 0, 0, 0);
  }
  if constexpr (S > 1) {
    __shared__ f32x4 part[S - 1][64];
    if (wave > 0) part[wave - 1][lane] = acc;
    __syncthreads();
    if (wave > 0) return false;
#pragma unroll
    for (int w = 0; w < S - 1; w++) {
      const f32x4 o = part[w][lane];
      acc = f32x4{acc[0] + o[0], acc[1] + o[1], acc[2] + o[2], acc[3] + o[3]};
    }
  }
  return true;
}

template <typename Epi, int S>
__global__ __launch_bounds__(64 * S) void gemm_bf16_skinny_kernel(GemmArgs g, Epi epi) {
  const int lane = threadIdx.x & 63;
  const int r0 = blockIdx.x * 16, c0 = blockIdx.y * 16;  // r: the P operand's rows (output features), c: Q's (tokens)
  f32x4 acc;
  if (!skinny_tile<S>(g.P, g.ldp, g.R, g.Q, g.ldq, g.C, g.K, r0, c0, acc)) return;
  epi.init();
  const int c = c0 + (lane & 15), r = r0 + (lane >> 4) * 4;
  if (c < g.C && r < g.R) epi(c, r, acc);
}

// Batched C[b] = alpha * A[b] B[b]^T on the same one-wave tiles (A [M, K], B [N, K], both with K contiguous): the pointer
// scorers of M4C (OcrPtrNet mmf_m4c.py:391-394, DynamicPointerNetwork m4c.py:30-31: 12 x 50 scores per sample over 768
// features) and ovqa_batched_gemm's NT form.  The epilogue gets one element at a time: (batch, m, n, value).
struct BatchedNT {
  const bf16* A; int64_t lda, sa;
  const bf16* B; int64_t ldb, sb;
  int M, N, K;
};
template <typename Epi, int S>
__global__ __launch_bounds__(64 * S) void batched_nt_skinny_kernel(BatchedNT g, Epi epi) {
  const int lane = threadIdx.x & 63, b = blockIdx.z;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  f32x4 acc;  // P = B's rows (n: the MFMA's row operand), Q = A's rows (m)
  if (!skinny_tile<S>(g.B + (int64_t)b * g.sb, g.ldb, g.N, g.A + (int64_t)b * g.sa, g.lda, g.M, g.K, n0, m0, acc)) return;
  const int m = m0 + (lane & 15), n = n0 + (lane >> 4) * 4;
  if (m >= g.M) return;
#pragma unroll
  for (int e = 0; e < 4; e++)
    if (n + e < g.N) epi(b, m, n + e, acc[e]);
}

template <bool P_KMAJOR, bool Q_KMAJOR, typename Epi>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs g, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][P | Q][16 KiB]
  // XCD-aware remap (bijective form): consecutive tile ids on one XCD walk the r-tiles of one c-panel.
  const int nwg = g.tiles_r * g.tiles_c;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, rem = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  }
  const int tc = bid / g.tiles_r, tr = bid % g.tiles_r;
  gemm_tile<P_KMAJOR, Q_KMAJOR, Epi>(g, tc * BT, tr * BT, epi, smem);
}

// ------------------------------------------------------------------ epilogues (c, r..r+3)
__device__ __forceinline__ void store4(bf16* p, float a, float b, float c, float d) {
  bf16x4 v;
  v[0] = (bf16)a; v[1] = (bf16)b; v[2] = (bf16)c; v[3] = (bf16)d;
  *reinterpret_cast<bf16x4*>(p) = v;
}
__device__ __forceinline__ float4 bias4(const float* bias, int n) {
  return bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ void store8(bf16* p, const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int t = 0; t < 8; t++) o[t] = (bf16)v[t];
  *reinterpret_cast<bf16x8*>(p) = o;
}
__device__ __forceinline__ void bias8(const float* bias, int n, const f32x4& lo, const f32x4& hi, float (&u)[8]) {
  const float4 b0 = bias4(bias, n), b1 = bias4(bias ? bias + 4 : nullptr, n);
  u[0] = lo[0] + b0.x; u[1] = lo[1] + b0.y; u[2] = lo[2] + b0.z; u[3] = lo[3] + b0.w;
  u[4] = hi[0] + b1.x; u[5] = hi[1] + b1.y; u[6] = hi[2] + b1.z; u[7] = hi[3] + b1.w;
}

// `wide(m, n, lo, hi)`: 8 consecutive output features n..n+7 of row m (the glds kernels with a row-major P);
// needs 16-byte aligned rows for every tensor it touches (checked by launch()).
// (kTwoPhase epilogues also expose pre() = every load of an 8-feature piece and post() = compute + store: the 256 x 256
// tile, gemm_tile256.h, software-pipelines them so that no store is waited for between pieces)
struct Bias8 { float4 b0, b1; };
__device__ __forceinline__ Bias8 load_bias8(const float* bias, int n) {
  return Bias8{bias4(bias, n), bias4(bias ? bias + 4 : nullptr, n)};
}
__device__ __forceinline__ void add_bias8(const Bias8& b, const f32x4& lo, const f32x4& hi, float (&u)[8]) {
  u[0] = lo[0] + b.b0.x; u[1] = lo[1] + b.b0.y; u[2] = lo[2] + b.b0.z; u[3] = lo[3] + b.b0.w;
  u[4] = hi[0] + b.b1.x; u[5] = hi[1] + b.b1.y; u[6] = hi[2] + b.b1.z; u[7] = hi[3] + b.b1.w;
}
struct MEpiBias {
  static constexpr bool kWide = true, kTwoPhase = true;
  bf16* y; int64_t ldy; const float* bias;
  typedef Bias8 Ctx;
  __device__ __forceinline__ void init() {}
  __device__ __forceinline__ Ctx pre(int m, int n) const { return load_bias8(bias, n); }
  __device__ __forceinline__ void post(const Ctx& k, int m, int n, const f32x4& lo, const f32x4& hi) const {
    float u[8];
    add_bias8(k, lo, hi, u);
    store8(y + (int64_t)m * ldy + n, u);
  }
  __device__ __forceinline__ void wide(int m, int n, const f32x4& lo, const f32x4& hi) const {
    float u[8];
    bias8(bias, n, lo, hi, u);
    store8(y + (int64_t)m * ldy + n, u);
  }
  __device__ __forceinline__ void operator()(int m, int n, const f32x4& a) const {
    const float4 b = bias4(bias, n);
    store4(y + (int64_t)m * ldy + n, a[0] + b.x, a[1] + b.y, a[2] + b.z, a[3] + b.w);
  }
};
// y_i = x W_i^T + b_i for three stacked weight matrices [3 F, K], each output with its own base and row stride (a
// decoding step's q -> a buffer, k and v -> their slots of the in-place caches, in ONE launch)
struct MEpiBiasSplit3 {
  static constexpr bool kWide = true, kTwoPhase = false;
  bf16* y0; int64_t ld0; bf16* y1; int64_t ld1; bf16* y2; int64_t ld2; int F; const float* bias;
  __device__ __forceinline__ void init() {}
  __device__ __forceinline__ bf16* dest(int m, int n) const {
    const int which = n >= 2 * F ? 2 : (n >= F ? 1 : 0);
    bf16* base = which == 2 ? y2 : (which == 1 ? y1 : y0);
    const int64_t ld = which == 2 ? ld2 : (which == 1 ? ld1 : ld0);
    return base + (int64_t)m * ld + (n - which * F);
  }
  __device__ __forceinline__ void wide(int m, int n, const f32x4& lo, const f32x4& hi) const {
    float u[8];
    bias8(bias, n, lo, hi, u);
    store8(dest(m, n), u);
  }
  __device__ __forceinline__ void operator()(int m, int n, const f32x4& a) const {
    const float4 b = bias4(bias, n);
    store4(dest(m, n), a[0] + b.x, a[1] + b.y, a[2] + b.z, a[3] + b.w);
  }
};
struct MEpiBiasGelu {
  static constexpr bool kWide = true, kTwoPhase = true;
  bf16* y; int64_t ldy; const float* bias; bf16* preact; int N; DropArgs da; DropState ds;
  typedef Bias8 Ctx;
  __device__ __forceinline__ void init() { ds = drop_init(da); }
  __device__ __forceinline__ Ctx pre(int m, int n) const { return load_bias8(bias, n); }
  __device__ __forceinline__ void post(const Ctx& k, int m, int n, const f32x4& lo, const f32x4& hi) const {
    float u[8];
    add_bias8(k, lo, hi, u);
    finish(m, n, u);
  }
  __device__ __forceinline__ void wide(int m, int n, const f32x4& lo, const f32x4& hi) const {
    float u[8];
    bias8(bias, n, lo, hi, u);
    finish(m, n, u);
  }
  __device__ __forceinline__ void finish(int m, int n, float (&u)[8]) const {
    if (preact) {
      bf16x8 o_;
#pragma unroll
      for (int t = 0; t < 8; t++) o_[t] = (bf16)u[t];
      store_saved(reinterpret_cast<bf16x8*>(preact + (int64_t)m * N + n), o_);
    }
    const uint32_t idx = (uint32_t)m * (uint32_t)N + (uint32_t)n;
    float dm[8];
    drop_mul8(ds, idx, dm);
#pragma unroll
    for (int t = 0; t < 8; t++) u[t] = gelu_fast(u[t]) * dm[t];
    store8(y + (int64_t)m * ldy + n, u);
  }
  __device__ __forceinline__ void operator()(int m, int n, const f32x4& a) const {
    const float4 b = bias4(bias, n);
    const float u0 = a[0] + b.x, u1 = a[1] + b.y, u2 = a[2] + b.z, u3 = a[3] + b.w;
    if (preact) store4(preact + (int64_t)m * N + n, u0, u1, u2, u3);
    const uint32_t idx = (uint32_t)m * (uint32_t)N + (uint32_t)n;
    store4(y + (int64_t)m * ldy + n, gelu_fast(u0) * drop_mul(ds, idx), gelu_fast(u1) * drop_mul(ds, idx + 1),
           gelu_fast(u2) * drop_mul(ds, idx + 2), gelu_fast(u3) * drop_mul(ds, idx + 3));
  }
};
struct MEpiBiasResidual {
  static constexpr bool kWide = true, kTwoPhase = true;
  bf16* y; int64_t ldy; const float* bias; const bf16* res; int64_t ldres; int N; DropArgs da; DropState ds;
  struct Ctx { Bias8 b; bf16x8 r; };
  __device__ __forceinline__ void init() { ds = drop_init(da); }
  __device__ __forceinline__ Ctx pre(int m, int n) const {
    return Ctx{load_bias8(bias, n), *reinterpret_cast<const bf16x8*>(res + (int64_t)m * ldres + n)};
  }
  __device__ __forceinline__ void post(const Ctx& k, int m, int n, const f32x4& lo, const f32x4& hi) const {
    float u[8];
    add_bias8(k.b, lo, hi, u);
    finish(m, n, u, k.r);
  }
  __device__ __forceinline__ void wide(int m, int n, const f32x4& lo, const f32x4& hi) const {
    float u[8];
    bias8(bias, n, lo, hi, u);
    finish(m, n, u, *reinterpret_cast<const bf16x8*>(res + (int64_t)m * ldres + n));
  }
  __device__ __forceinline__ void finish(int m, int n, float (&u)[8], const bf16x8& r) const {
    const uint32_t idx = (uint32_t)m * (uint32_t)N + (uint32_t)n;
    float dm[8];
    drop_mul8(ds, idx, dm);
#pragma unroll
    for (int t = 0; t < 8; t++) u[t] = (float)r[t] + u[t] * dm[t];
    store8(y + (int64_t)m * ldy + n, u);
  }
  __device__ __forceinline__ void operator()(int m, int n, const f32x4& a) const {
    const float4 b = bias4(bias, n);
    const bf16x4 r = *reinterpret_cast<const bf16x4*>(res + (int64_t)m * ldres + n);
    const uint32_t idx = (uint32_t)m * (uint32_t)N + (uint32_t)n;
    store4(y + (int64_t)m * ldy + n, (float)r[0] + (a[0] + b.x) * drop_mul(ds, idx),
           (float)r[1] + (a[1] + b.y) * drop_mul(ds, idx + 1), (float)r[2] + (a[2] + b.z) * drop_mul(ds, idx + 2),
           (float)r[3] + (a[3] + b.w) * drop_mul(ds, idx + 3));
  }
};
// fp32 residual stream (bf16 mode): pre32 = res + drop(x W^T + b), written in fp32.  The residual
// is either a plain fp32 tensor (mean == nullptr) or the LayerNorm of the PREVIOUS block's fp32 pre-LN sum, recomputed
// here from that sum and its saved row statistics -- (v - mean) * rstd * gamma + beta, the expression ln_fwd_kernel
// evaluates -- so a block's fp32 output never has to be written to HBM: ln_fwd_kernel stores only the bf16 GEMM operand.
struct MEpiBiasRes32 {
  static constexpr bool kWide = true, kTwoPhase = true;
  float* pre32; int64_t ldpre; const float* bias; const float* res; int64_t ldres;
  const float* mean; const float* rstd; const float* gamma; const float* beta;
  int N; DropArgs da; DropState ds;
  struct Ctx { Bias8 b; float4 r0, r1, g0, g1, b0, b1; float mu, rs; };
  __device__ __forceinline__ void init() { ds = drop_init(da); }
  __device__ __forceinline__ Ctx pre(int m, int n) const {  // (mean: a workgroup-uniform condition)
    Ctx k{};
    k.b = load_bias8(bias, n);
    k.r0 = *reinterpret_cast<const float4*>(res + (int64_t)m * ldres + n);
    k.r1 = *reinterpret_cast<const float4*>(res + (int64_t)m * ldres + n + 4);
    if (mean) {
      k.mu = mean[m]; k.rs = rstd[m];
      k.g0 = *reinterpret_cast<const float4*>(gamma + n); k.g1 = *reinterpret_cast<const float4*>(gamma + n + 4);
      k.b0 = *reinterpret_cast<const float4*>(beta + n); k.b1 = *reinterpret_cast<const float4*>(beta + n + 4);
    }
    return k;
  }
  __device__ __forceinline__ void post(const Ctx& k, int m, int n, const f32x4& lo, const f32x4& hi) const {
    float u[8];
    add_bias8(k.b, lo, hi, u);
    finish(m, n, u, k);
  }
  __device__ __forceinline__ void wide(int m, int n, const f32x4& lo, const f32x4& hi) const { post(pre(m, n), m, n, lo, hi); }
  __device__ __forceinline__ void finish(int m, int n, const float (&u)[8], const Ctx& k) const {
    const float4 r0 = k.r0, r1 = k.r1;
    float r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    if (mean) {
      const float mu = k.mu, rs = k.rs;
      const float4 g0 = k.g0, g1 = k.g1, b0 = k.b0, b1 = k.b1;
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      const float nmr = -mu * rs;  // (r - mu) * rs * g + b as two fmas per element
#pragma unroll
      for (int t = 0; t < 8; t++) r[t] = fmaf(fmaf(r[t], rs, nmr), g[t], b[t]);
    }
    const uint32_t idx = (uint32_t)m * (uint32_t)N + (uint32_t)n;
    float dm[8];
    drop_mul8(ds, idx, dm);
#pragma unroll
    for (int t = 0; t < 8; t++) r[t] += u[t] * dm[t];
    float* o = pre32 + (int64_t)m * ldpre + n;
    *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(r[4], r[5], r[6], r[7]);
  }
  __device__ __forceinline__ void operator()(int m, int n, const f32x4& a) const {
    const float4 bb = bias4(bias, n);
    const float4 r0 = *reinterpret_cast<const float4*>(res + (int64_t)m * ldres + n);
    float r[4] = {r0.x, r0.y, r0.z, r0.w};
    if (mean) {
      const float mu = mean[m], rs = rstd[m];
      const float4 g0 = *reinterpret_cast<const float4*>(gamma + n), b0 = *reinterpret_cast<const float4*>(beta + n);
      const float g[4] = {g0.x, g0.y, g0.z, g0.w}, b[4] = {b0.x, b0.y, b0.z, b0.w};
      const float nmr = -mu * rs;
#pragma unroll
      for (int t = 0; t < 4; t++) r[t] = fmaf(fmaf(r[t], rs, nmr), g[t], b[t]);
    }
    const uint32_t idx = (uint32_t)m * (uint32_t)N + (uint32_t)n;
    const float u[4] = {a[0] + bb.x, a[1] + bb.y, a[2] + bb.z, a[3] + bb.w};
#pragma unroll
    for (int t = 0; t < 4; t++) r[t] += u[t] * drop_mul(ds, idx + t);
    *reinterpret_cast<float4*>(pre32 + (int64_t)m * ldpre + n) = make_float4(r[0], r[1], r[2], r[3]);
  }
};
// dX = dY W  [* dropmask * gelu'(u)]  (+ dx)
struct MEpiBwdData {
  static constexpr bool kWide = true, kTwoPhase = true;
  bf16* dx; int64_t lddx; const bf16* preact; int Kcols; const bf16* addend; int64_t ldadd; DropArgs da; DropState ds;
  struct Ctx { bf16x8 u, o; };
  __device__ __forceinline__ void init() { ds = drop_init(da); }
  __device__ __forceinline__ Ctx pre(int m, int n) const {  // (preact / addend: workgroup-uniform conditions)
    Ctx k{};
    if (preact) k.u = *reinterpret_cast<const bf16x8*>(preact + (int64_t)m * Kcols + n);
    if (addend) k.o = *reinterpret_cast<const bf16x8*>(addend + (int64_t)m * ldadd + n);
    return k;
  }
  __device__ __forceinline__ void post(const Ctx& k, int m, int n, const f32x4& lo, const f32x4& hi) const {
    finish(m, n, lo, hi, k.u, k.o);
  }
  __device__ __forceinline__ void wide(int m, int n, const f32x4& lo, const f32x4& hi) const {
    bf16x8 u{}, o{};
    if (preact) u = *reinterpret_cast<const bf16x8*>(preact + (int64_t)m * Kcols + n);
    if (addend) o = *reinterpret_cast<const bf16x8*>(addend + (int64_t)m * ldadd + n);
    finish(m, n, lo, hi, u, o);
  }
  __device__ __forceinline__ void finish(int m, int n, const f32x4& lo, const f32x4& hi, const bf16x8& u, const bf16x8& o) const {
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    if (preact) {
      const uint32_t idx = (uint32_t)m * (uint32_t)Kcols + (uint32_t)n;
      float dm[8];
      drop_mul8(ds, idx, dm);
#pragma unroll
      for (int t = 0; t < 8; t++) v[t] *= dm[t] * gelu_grad_fast((float)u[t]);
    }
    if (addend) {
#pragma unroll
      for (int t = 0; t < 8; t++) v[t] += (float)o[t];
    }
    store8(dx + (int64_t)m * lddx + n, v);
  }
  __device__ __forceinline__ void operator()(int m, int n, const f32x4& a) const {
    float v[4] = {a[0], a[1], a[2], a[3]};
    if (preact) {
      const bf16x4 u = *reinterpret_cast<const bf16x4*>(preact + (int64_t)m * Kcols + n);
      const uint32_t idx = (uint32_t)m * (uint32_t)Kcols + (uint32_t)n;
#pragma unroll
      for (int t = 0; t < 4; t++) v[t] *= drop_mul(ds, idx + t) * gelu_grad_fast((float)u[t]);
    }
    if (addend) {
      const bf16x4 o = *reinterpret_cast<const bf16x4*>(addend + (int64_t)m * ldadd + n);
#pragma unroll
      for (int t = 0; t < 4; t++) v[t] += (float)o[t];
    }
    store4(dx + (int64_t)m * lddx + n, v[0], v[1], v[2], v[3]);
  }
};
// dW (fp32) (+)= acc
struct MEpiWgrad {
  static constexpr bool kWide = false, kTwoPhase = false;
  float* dw; int64_t ld; int accumulate; float* db; int accumulate_db;
  __device__ __forceinline__ void init() {}
  __device__ __forceinline__ bool wants_colsum() const { return db != nullptr; }
  __device__ __forceinline__ void colsum(int n, float v) const { db[n] = accumulate_db ? db[n] + v : v; }
  __device__ __forceinline__ void operator()(int n, int i, const f32x4& a) const {
    float4* p = reinterpret_cast<float4*>(dw + (int64_t)n * ld + i);
    float4 v = make_float4(a[0], a[1], a[2], a[3]);
    if (accumulate) {
      const float4 o = *p;
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *p = v;
  }
};

// Grouped weight gradients: tile table entry {problem, tile_c (n), tile_r (i), -} -> one 128x128 tile of
// dW_problem = dY^T X.  One launch covers every linear layer of a backward pass (no split-K, no slabs).
__global__ __launch_bounds__(256) void gemm_bf16_grouped_wgrad_kernel(const ovqa_wgrad_problem* __restrict__ probs,
                                                                      const int4* __restrict__ tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int4 t = tiles[blockIdx.x];
  if (t.x < 0) return;  // padding entry (keeps the host's blockIdx % 8 -> XCD grouping aligned)
  const ovqa_wgrad_problem pr = probs[t.x];
  MEpiWgrad epi{pr.dw, pr.K, pr.accumulate & 1, pr.db, (pr.accumulate >> 1) & 1};
  // (reading both operands row-major from transposed activation copies was timed: no gain for this
  // register-staged kernel, unlike the direct-to-LDS dX kernels)
  GemmArgs g{(const bf16*)pr.x, pr.ldx, (const bf16*)pr.dy, pr.lddy, pr.K, pr.N, pr.M, 0, 0};
  gemm_tile<true, true, MEpiWgrad, true>(g, t.y * BT, t.z * BT, epi, smem);
}

// The same grouped dW on the direct-to-LDS 8-wave tile (both operands k-major); needs M % 64 == 0 for every problem
template <int NBUF>
__global__ __launch_bounds__(512, 4) void gemm_bf16_grouped_wgrad_glds_kernel(const ovqa_wgrad_problem* __restrict__ probs,
                                                                              const int4* __restrict__ tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int4 t = tiles[blockIdx.x];
  if (t.x < 0) return;
  const ovqa_wgrad_problem pr = probs[t.x];
  MEpiWgrad epi{pr.dw, pr.K, pr.accumulate & 1, pr.db, (pr.accumulate >> 1) & 1};
  GemmArgs g{(const bf16*)pr.x, pr.ldx, (const bf16*)pr.dy, pr.lddy, pr.K, pr.N, pr.M, 0, 0};
  gemm_tile_glds<true, true, MEpiWgrad, true, NBUF, 8, 128, 128>(g, t.y * BT, t.z * BT, epi, smem);
}

// ---- the optimiser step inside the grouped dW (round 5) -------------------------------------------------------------------
// For a weight whose gradient is exactly one product of the step (and no exchange between ranks follows), the workgroup
// that finished a 128 x 128 tile of dW applies Adam to that tile right away: the fp32 tile goes to LDS (the ring is idle
// after the K loop; 16-byte chunks XOR-swizzled by the row so that the fragment stores -- 16 rows of one chunk column per
// wave instruction -- and the row-wise reads are both conflict free), every thread then owns 16-byte pieces of whole rows:
// master, both moments in and out (512-byte row segments per 32 lanes, streaming cache policy like the tiled Adam kernel),
// the bf16 shadow out, its bf16 values back into LDS, and after a barrier the tile of the TRANSPOSED shadow is gathered
// from there (8 output features per 16-byte store).  The gradient never reaches HBM (8 B per weight of traffic less) and
// the optimiser state is streamed while other workgroups' K loops keep the fetch path busy with L2 hits.
// Same update function as adam_tiled_kernel (common.h): same bits.
struct MEpiWgradAdam {
  static constexpr bool kWide = false, kTwoPhase = false;
  float* dw; int64_t ld; int accumulate; float* db; int accumulate_db;
  float* tile;  // fused: the fp32 tile in LDS; nullptr: the plain gradient store
  int c0, r0;
  __device__ __forceinline__ void init() {
    if (tile) __syncthreads();  // (workgroup-uniform) every wave is done with the last K tile's fragments
  }
  __device__ __forceinline__ bool wants_colsum() const { return db != nullptr; }
  __device__ __forceinline__ void colsum(int n, float v) const { db[n] = accumulate_db ? db[n] + v : v; }
  __device__ __forceinline__ void operator()(int n, int i, const f32x4& a) const {
    if (tile) {
      const int ln = n - c0, li = i - r0;
      *reinterpret_cast<f32x4*>(tile + ln * 128 + ((((li >> 2) ^ ln) & 31) << 2)) = a;
      return;
    }
    float4* p = reinterpret_cast<float4*>(dw + (int64_t)n * ld + i);
    float4 v = make_float4(a[0], a[1], a[2], a[3]);
    if (accumulate) {
      const float4 o = *p;
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *p = v;
  }
};

__global__ __launch_bounds__(512, 4) void gemm_bf16_grouped_wgrad_adam_kernel(const ovqa_wgrad_problem* __restrict__ probs,
                                                                              const int4* __restrict__ tiles,
                                                                              const ovqa_adam_target* __restrict__ targets,
                                                                              const ovqa_adam_consts kc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int4 t = tiles[blockIdx.x];
  if (t.x < 0) return;
  const ovqa_wgrad_problem pr = probs[t.x];
  const ovqa_adam_target tg = targets[t.x];
  const bool fused = tg.param != nullptr;  // (workgroup-uniform)
  const int c0 = t.y * BT, r0 = t.z * BT;
  float* T = reinterpret_cast<float*>(smem);
  MEpiWgradAdam epi{pr.dw, pr.K, pr.accumulate & 1, pr.db, (pr.accumulate >> 1) & 1, fused ? T : nullptr, c0, r0};
  GemmArgs g{(const bf16*)pr.x, pr.ldx, (const bf16*)pr.dy, pr.lddy, pr.K, pr.N, pr.M, 0, 0};
  gemm_tile_glds<true, true, MEpiWgradAdam, true, 2, 8, 128, 128>(g, c0, r0, epi, smem);
  if (!fused) return;
  __syncthreads();  // the tile is complete
  const AdamK ak = adam_consts(kc.lr, kc.lr_scale_ptr, kc.beta1, kc.beta2, kc.eps, kc.weight_decay, kc.grad_scale, kc.step_ptr);
  const int tid = threadIdx.x, q = tid & 31;
#pragma unroll 2
  for (int pass = 0; pass < 8; pass++) {
    const int ln = pass * 16 + (tid >> 5);
    float* slot = T + ln * 128 + (((q ^ ln) & 31) << 2);
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(slot);
    const int64_t e4 = ((int64_t)(c0 + ln) * pr.K + r0 + q * 4) >> 2;  // (K % 128 == 0: 16-byte aligned)
    f32x4 pp = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(tg.param) + e4);
    f32x4 mm = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(tg.exp_avg) + e4);
    f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(tg.exp_avg_sq) + e4);
    bf16x4 s4;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      float pk = pp[k], mk = mm[k], vk = vv[k];
      adam_update1(ak, g4[k], pk, mk, vk);
      pp[k] = pk; mm[k] = mk; vv[k] = vk;
      s4[k] = (bf16)pk;
    }
    __builtin_nontemporal_store(pp, reinterpret_cast<f32x4*>(tg.param) + e4);
    __builtin_nontemporal_store(mm, reinterpret_cast<f32x4*>(tg.exp_avg) + e4);
    __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(tg.exp_avg_sq) + e4);
    reinterpret_cast<bf16x4*>(tg.shadow)[e4] = s4;
    *reinterpret_cast<bf16x4*>(slot) = s4;  // (this thread's own 16-byte slot: its first 8 bytes now hold the bf16 values)
  }
  __syncthreads();
  bf16* tt = reinterpret_cast<bf16*>(tg.transposed);
#pragma unroll
  for (int pass = 0; pass < 4; pass++) {
    const int li = pass * 32 + (tid >> 4);  // input feature = row of the transposed tile
    const int cn = (tid & 15) * 8;          // 8 consecutive output features
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int ln = cn + e;
      o[e] = reinterpret_cast<const bf16*>(T + ln * 128 + ((((li >> 2) ^ ln) & 31) << 2))[li & 3];
    }
    *reinterpret_cast<bf16x8*>(tt + (int64_t)(r0 + li) * tg.ld_transposed + c0 + cn) = o;
  }
}

// (Rounds 4-5 carried a 256 x 256-tile form of the grouped dW here -- one 16-wave workgroup per CU, ring of four 32-deep half
// steps -- and a 256 x 128 one: 458-493 and 424 us against 480 / 393 for the 128 x 128 form below once that one read its
// fragments without the compiler's ring-draining wait; the numbers and the per-half-step decomposition are in
// profiles/README.md.  Removed in round 6 with the other forms that lost.)
__global__ __launch_bounds__(256) void gemm_bf16_wgrad_kernel(GemmArgs g, MEpiWgrad epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tc = blockIdx.x / g.tiles_r, tr = blockIdx.x % g.tiles_r;
  gemm_tile<true, true, MEpiWgrad, true>(g, tc * BT, tr * BT, epi, smem);
}

// Rows of the c operand a 128-row tile should OWN so that the grid fills whole rounds of `per_cu` workgroups per CU (0: keep
// full tiles): 6400 x 512 on the 16-wave form (one per CU): 4 x 50 tiles on 200 of 256 CUs -> 4 x 64 tiles of 100 rows:
// step 3.151 / 3.165 -> 3.093 / 3.105 / 3.126 ms on one box, 3.170 -> 3.153 on another.  NOT for the 8-wave form with two
// workgroups per CU (6400 x 2048: 16 x 50 = 800 tiles = 1.56 rounds -> 16 x 64 = two rounds exactly): measured +0.12 ms
// per step -- its second, partial round runs with the CUs half empty and therefore fast, which two full rounds of
// smaller tiles (more weight-tile traffic) do not beat.
inline int device_cus() {
  static int cus = -1;
  if (cus < 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cus = 256;
  }
  return cus;
}

inline int owned_rows(int tiles_r, int tiles_c, int64_t C, int per_cu) {
  const int slots = device_cus() * per_cu, tiles = tiles_r * tiles_c;
  if (tiles % slots == 0) return 0;
  const int rounds = (tiles + slots - 1) / slots;
  const int n_c = rounds * slots / tiles_r;  // c tiles that fill `rounds` rounds
  if (n_c <= tiles_c) return 0;
  const int step = (int)((C + n_c - 1) / n_c);
  return (step >= 64 && step < BT) ? step : 0;
}

// Tile tiers of the k-contiguous x k-contiguous products (measured in the MCAN step, rounds 2-5; the forms that lost -- rings
// of other depths, 4-wave 128 x 128 tiles, 64 x 64 tiles, k-split wave grids -- are gone from the source, their numbers
// are in profiles/README.md):
constexpr int kSmallTiles = 320;    // at most this many 128 x 128 tiles: 128 x 64 tiles (twice the workgroups) ...
constexpr int kTinyTiles = 330;     // ... and at most this many of those: 64 x 32 tiles with 4 waves, ring of 4
constexpr int kTinyMaxRows = 1024;  // products with more weight rows leave the tiny tier for 16-wave 128 x 128 tiles
constexpr int kSkinnyMaxRows = 128; // activation rows up to which a product takes the one-wave-per-tile form (decoding)

template <typename K>
inline int set_max_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
      ovqa_set_error("hipFuncSetAttribute(%zu): %s", bytes, hipGetErrorString(e));
      return OVQA_ERR_LAUNCH;
    }
  }
  return OVQA_OK;
}

// K % 64 == 0 and (for the 8-feature epilogues of a row-major P) 16-byte aligned rows of every tensor the epilogue touches:
// the direct-to-LDS forms below; anything else: the register-staged kernel (any K % 8 == 0, 8-byte epilogue accesses).
template <bool PK, bool QK, typename Epi>
int launch(const void* P, int64_t ldp, const void* Q, int64_t ldq, int64_t R, int64_t C, int64_t K, Epi epi,
           hipStream_t st, const char* what, bool wide_ok = true) {
  GemmArgs g{(const bf16*)P, ldp, (const bf16*)Q, ldq, (int)R, (int)C, (int)K,
             (int)((R + BT - 1) / BT), (int)((C + BT - 1) / BT)};
  const bool glds = K % BK == 0 && (PK || !Epi::kWide || wide_ok);
  // few 128x128 tiles (the M = 1280 question stack): halve the c tile -> twice the workgroups
  const bool small_c = !QK && glds && g.tiles_r * g.tiles_c <= kSmallTiles;
  // products with more than kTinyMaxRows weight rows leave the tiny tier -- the question stack's 1280 x 2048 outputs (fc1
  // forward, fc2 dX) run as 160 tiles of 128 x 128 on 16 waves instead of 1280 tiles of 64 x 32: 262 KB instead of
  // 5 x 98 KB through a CU's fetch path (GELU forward 11.84 -> 9.89 us, dX 13.6 -> 10.3 us per launch) -- but only where
  // the 128 x 128 tiling still has >= 128 tiles (a beam-3 decoding step's 192 x 4000 logits are 64 such tiles: 9 % slower)
  const bool wide_enough = (int64_t)g.tiles_r * ((C + 127) / 128) >= 128;
  const bool tiny_c = small_c && g.tiles_r * (int)((C + 63) / 64) <= kTinyTiles && (R <= kTinyMaxRows || !wide_enough);
  if (small_c) g.tiles_c = (int)((C + (tiny_c ? 31 : 63)) / (tiny_c ? 32 : 64));
  const dim3 grid(g.tiles_r * g.tiles_c);
#define OVQA_GLDS(NBUF, NW, BCV, BRV, GRID)                                                                         \
  {                                                                                                                \
    const size_t lds = (size_t)NBUF * (BRV * BK * 2 + BCV * BK * 2);                                               \
    int rc = set_max_lds(gemm_bf16_glds_kernel<PK, QK, Epi, NBUF, NW, BCV, BRV>, lds);                             \
    if (rc != OVQA_OK) return rc;                                                                                  \
    OVQA_LAUNCH_TIMED((gemm_bf16_glds_kernel<PK, QK, Epi, NBUF, NW, BCV, BRV>), GRID, dim3(NW * 64), lds, st, g,   \
                      epi);                                                                                        \
    return ovqa_check_launch(what);                                                                                \
  }
  if constexpr (!QK && !PK && Epi::kWide) {
    // 256 x 256 tiles on one 8-wave workgroup per CU (gemm_tile256.h, round 6) for long reductions on grids that fill whole
    // rounds of the chip: 8192 x 4096 x 4096 1290-1357 TFLOP/s against 900-995 on the 128 x 128 tiles (hipBLASLt 1424-1540).
    // NOT for the step's own K = 512 products: alone, 6400 x 2048 <- 512 takes 19.8-22.2 us against 23.2-26.2 (hipBLASLt
    // 21.2-22.6), but a tile's epilogue (4-6 us: 128 KB of stores per CU, and for GELU / GELU' ~20 VALU instructions per
    // element) has nothing to hide behind with one workgroup per CU, where two co-resident 128 x 128 workgroups hide each
    // other's: in the MCAN step 3.115 ms with fc1 forward + fc2 dX on this form, 3.06-3.08 with fc1 forward only, 3.05-3.06
    // without (same box, alternated).  OVQA_GEMM_T256=0 switches the form off.
    static int t256 = -1;
    if (t256 < 0) {
      const char* e = getenv("OVQA_GEMM_T256");
      t256 = e ? atoi(e) : 1;
    }
    if (t256 && glds && K >= 1024 && R % 8 == 0 && ldp < (1 << 22) && ldq < (1 << 22)) {
      const int tr = (int)((R + 255) / 256), tc = (int)((C + 255) / 256), cus = device_cus();
      const int64_t tiles = (int64_t)tr * tc, rounds = (tiles + cus - 1) / cus;
      if (tiles * 4 >= rounds * cus * 3 && tiles < (1 << 30)) {
        ovqa_t256::Args a{(const bf16*)P, ldp, (const bf16*)Q, ldq, (int)R, (int)C, (int)K, tr, tc, nullptr,
                          (tiles >= 2 * cus && tr >= 8 && tc >= 8) ? 4 : 0};
        int rc = set_max_lds(ovqa_t256::kernel<Epi, 3>, ovqa_t256::LDS_BYTES);
        if (rc != OVQA_OK) return rc;
        OVQA_LAUNCH_TIMED((ovqa_t256::kernel<Epi, 3>), dim3((unsigned)tiles), dim3(512), ovqa_t256::LDS_BYTES, st, a, epi);
        return ovqa_check_launch(what);
      }
    }
  }
  if constexpr (!QK && !PK) {
    // a decoding step's products: few activation rows against a whole weight matrix (MEASURED, bench.py --workload decode,
    // B = 64: 64 rows (greedy) 250 -> 197 us per decoding step; 192 rows (beam 3) 259 -> 258: there the 64 x 32 tiles are
    // as fast, so the form stops at 128 rows)
    if (glds && C <= kSkinnyMaxRows && K % 32 == 0 && K <= 2048 && R % 4 == 0) {
      const dim3 gs((unsigned)((R + 15) / 16), (unsigned)((C + 15) / 16));
      const int nk = (int)(K / 32);
      if (nk <= 16) {
        OVQA_LAUNCH_TIMED((gemm_bf16_skinny_kernel<Epi, 1>), gs, dim3(64), 0, st, g, epi);
      } else if (nk <= 32) {
        OVQA_LAUNCH_TIMED((gemm_bf16_skinny_kernel<Epi, 2>), gs, dim3(128), 0, st, g, epi);
      } else {
        OVQA_LAUNCH_TIMED((gemm_bf16_skinny_kernel<Epi, 4>), gs, dim3(256), 0, st, g, epi);
      }
      return ovqa_check_launch(what);
    }
    // fewest tiles (the M = 1280 question stack): 64 x 32 tiles with 4 waves -- twice the workgroups of a 128 x 32 tier
    // (320 instead of 160 for 1280 x 512: every CU gets one), 12 KiB per ring stage, ring of 4.  In the step 3.53 -> 3.49 ms;
    // a ring of 3 / 6 and 64 x 64 tiles (160 workgroups: 14.0-14.5 against 13.3 us) measured slower.
    if (tiny_c) {
      g.tiles_r = (int)((R + 63) / 64);
      OVQA_GLDS(4, 4, 32, 64, dim3(g.tiles_r * g.tiles_c))
    }
    // ONE 16-wave workgroup (4 x 4 wave grid) on a 128 x 128 tile for the products that would otherwise run two co-resident
    // 8-wave workgroups on 128 x 64 tiles per CU (6400 x 512 outputs: 200 tiles): the 128-row weight tile is staged once per
    // CU instead of twice -- 32 instead of 48 KB per K step through the CU's L2 fetch path, which is what these loops wait
    // for.  MEASURED (scripts/gemm_wg_timeline.py): span 8.2-8.5 -> 7.5-7.7 us (ring of 3; 6.9-7.3 with a ring of 2, which
    // loses in the step, where operands are cold); step 3.407 / 3.389 -> 3.371 / 3.372 ms in two alternations (ring of 4:
    // 3.412 / 3.365).
    if (small_c) {
      g.tiles_c = (int)((C + BT - 1) / BT);
      if (const int step = owned_rows(g.tiles_r, g.tiles_c, C, 1)) {
        g.c_step = step;
        g.tiles_c = (int)((C + step - 1) / step);
      }
      OVQA_GLDS(3, 16, 128, 128, dim3(g.tiles_r * g.tiles_c))
    }
  }
  if constexpr (!QK) {  // (a k-major P: the [N, K]-weight form of dX, used where no transposed weight copy exists)
    if (tiny_c) OVQA_GLDS(4, 8, 32, 128, grid)
    if (small_c) OVQA_GLDS(3, 8, 64, 128, grid)
  }
  // (round 6: this tier on 4 waves of 64 x 64 outputs each -- 512 instead of 768 LDS bytes per MFMA, two workgroups per CU as
  // here -- measured 3.066 / 3.070 against 3.028 / 3.016 ms per step, alternated on one box; 6400 x 2048 <- 512 alone 26.3
  // against 26.1 us: fewer LDS bytes do not pay for half the waves)
  if (glds) OVQA_GLDS(2, 8, 128, 128, grid)
#undef OVQA_GLDS
  OVQA_LAUNCH_TIMED((gemm_bf16_kernel<PK, QK, Epi>), grid, dim3(256), 4 * TILE_BYTES, st, g, epi);
  return ovqa_check_launch(what);
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// column sums of a bf16 [M, N] matrix into fp32 db (bias gradient).  HBM-bound streaming reduction:
// one workgroup = 64 rows x 512 columns (16-byte loads, 16 rows in flight per wave), partial sums
// combined across the 4 waves through LDS and across row slabs with fp32 atomics (db pre-zeroed).
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16* __restrict__ dy, int64_t lddy,
                                                          float* __restrict__ db, int M, int N, int rows_per_slab) {
  __shared__ float red[4][64][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + lane) * 8;
  const int m0 = blockIdx.y * rows_per_slab;
  const int m1 = min(M, m0 + rows_per_slab);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < N) {
    const bf16* p = dy + col;
    int m = m0 + wave;
    for (; m + 12 < m1; m += 16) {  // 4 independent 16-byte loads per iteration
      const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(p + (int64_t)m * lddy);
      const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(p + (int64_t)(m + 4) * lddy);
      const bf16x8 v2 = *reinterpret_cast<const bf16x8*>(p + (int64_t)(m + 8) * lddy);
      const bf16x8 v3 = *reinterpret_cast<const bf16x8*>(p + (int64_t)(m + 12) * lddy);
#pragma unroll
      for (int t = 0; t < 8; t++) s[t] += ((float)v0[t] + (float)v1[t]) + ((float)v2[t] + (float)v3[t]);
    }
    for (; m < m1; m += 4) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(p + (int64_t)m * lddy);
#pragma unroll
      for (int t = 0; t < 8; t++) s[t] += (float)v[t];
    }
  }
#pragma unroll
  for (int t = 0; t < 8; t++) red[wave][lane][t] = s[t];
  __syncthreads();
  if (wave == 0 && col < N) {
#pragma unroll
    for (int t = 0; t < 8; t++)
      atomicAdd(db + col + t, red[0][lane][t] + red[1][lane][t] + red[2][lane][t] + red[3][lane][t]);
  }
}

struct MEpiPointerScore {  // OcrPtrNet / DynamicPointerNetwork scores: scale, additive key mask, -inf fills
  float* s; int T_; int Nk; float scale; const float* add_mask; const uint8_t* key_fill; const uint8_t* query_fill;
  __device__ __forceinline__ void operator()(int b, int m, int n, float acc) const {
    float v = acc * scale;
    if (add_mask) v += add_mask[(int64_t)b * Nk + n];
    if (key_fill && key_fill[(int64_t)b * Nk + n]) v = -INFINITY;
    if (query_fill && query_fill[(int64_t)b * T_ + m]) v = -INFINITY;
    s[((int64_t)b * T_ + m) * Nk + n] = v;
  }
};
template <typename TC>
struct MEpiBatchedOut {
  TC* c; int64_t ldc; int64_t stride_c; float alpha;
  __device__ __forceinline__ void operator()(int b, int m, int n, float acc) const {
    c[(int64_t)b * stride_c + (int64_t)m * ldc + n] = from_f32<TC>(alpha * acc);
  }
};

template <typename Epi>
int launch_batched_nt(const BatchedNT& g, int64_t batch, Epi epi, hipStream_t st, const char* what) {
  const dim3 grid((unsigned)((g.N + 15) / 16), (unsigned)((g.M + 15) / 16), (unsigned)batch);
  const int nk = g.K / 32;
  if (nk <= 16) hipLaunchKernelGGL((batched_nt_skinny_kernel<Epi, 1>), grid, dim3(64), 0, st, g, epi);
  else if (nk <= 32) hipLaunchKernelGGL((batched_nt_skinny_kernel<Epi, 2>), grid, dim3(128), 0, st, g, epi);
  else hipLaunchKernelGGL((batched_nt_skinny_kernel<Epi, 4>), grid, dim3(256), 0, st, g, epi);
  return ovqa_check_launch(what);
}

}  // namespace

namespace ovqa {

bool mfma_batched_nt_supported(const void* A, int64_t lda, int64_t sa, const void* B, int64_t ldb, int64_t sb,
                               int64_t batch, int64_t M, int64_t N, int64_t K) {
  return M >= 1 && N >= 1 && K >= 32 && K % 32 == 0 && K <= 2048 && lda % 8 == 0 && ldb % 8 == 0 && sa % 8 == 0 &&
         sb % 8 == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0 && batch <= 65535 && (M + 15) / 16 <= 65535 &&
         M < (1 << 30) && N < (1 << 30);
}

int mfma_pointer_score(const void* q, const void* k, const float* add_mask, const uint8_t* key_fill,
                       const uint8_t* query_fill, float* scores, int64_t B, int64_t T, int64_t Nk, int64_t D, float scale,
                       hipStream_t st) {
  BatchedNT g{(const bf16*)q, D, T * D, (const bf16*)k, D, Nk * D, (int)T, (int)Nk, (int)D};
  return launch_batched_nt(g, B, MEpiPointerScore{scores, (int)T, (int)Nk, scale, add_mask, key_fill, query_fill}, st,
                           "pointer_score");
}

int mfma_batched_nt(int c_dtype, const void* A, int64_t lda, int64_t sa, const void* B, int64_t ldb, int64_t sb, void* C,
                    int64_t ldc, int64_t sc, int64_t batch, int64_t M, int64_t N, int64_t K, float alpha, hipStream_t st) {
  BatchedNT g{(const bf16*)A, lda, sa, (const bf16*)B, ldb, sb, (int)M, (int)N, (int)K};
  if (c_dtype == OVQA_F32)
    return launch_batched_nt(g, batch, MEpiBatchedOut<float>{(float*)C, ldc, sc, alpha}, st, "batched_gemm NT");
  return launch_batched_nt(g, batch, MEpiBatchedOut<bf16>{(bf16*)C, ldc, sc, alpha}, st, "batched_gemm NT");
}

bool mfma_gemm_supported(int64_t R, int64_t C, int64_t K, int64_t ld_p, int64_t ld_q) {
  return R >= 8 && C >= 1 && K >= 8 && R % 8 == 0 && K % 8 == 0 && ld_p % 8 == 0 && ld_q % 8 == 0 &&
         R < (1 << 30) && C < (1 << 30) && K < (1 << 30);
}

bool mfma_linear_fwd_supported(int epilogue, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy,
                               int64_t ldres) {
  (void)epilogue;
  return mfma_gemm_supported(N, M, K, K, ldx) && ldy % 4 == 0 && ldres % 4 == 0;
}

int mfma_linear_fwd(int epilogue, const void* x, int64_t ldx, const void* w, const float* bias, const void* residual,
                    int64_t ldres, void* y, int64_t ldy, void* preact, int64_t M, int64_t N, int64_t K,
                    const DropArgs& da, hipStream_t st) {
  OVQA_REQUIRE(aligned16(x) && aligned16(w) && ((uintptr_t)y % 8 == 0) && (!bias || aligned16(bias)) &&
                   (!residual || (uintptr_t)residual % 8 == 0) && (!preact || (uintptr_t)preact % 8 == 0),
               OVQA_ERR_BAD_ARG, "linear_fwd(bf16): pointer alignment (x/w/bias 16 B, y/residual/preact 8 B)");
  const bool wide = aligned16(y) && ldy % 8 == 0 && (!residual || (aligned16(residual) && ldres % 8 == 0)) &&
                    (!preact || aligned16(preact));
  switch (epilogue) {
    case OVQA_EPI_BIAS:
      return launch<false, false>(w, K, x, ldx, N, M, K, MEpiBias{(bf16*)y, ldy, bias}, st, "linear_fwd(mfma,bias)", wide);
    case OVQA_EPI_BIAS_GELU:
      return launch<false, false>(w, K, x, ldx, N, M, K,
                                  MEpiBiasGelu{(bf16*)y, ldy, bias, (bf16*)preact, (int)N, da, DropState{}}, st,
                                  "linear_fwd(mfma,gelu)", wide);
    case OVQA_EPI_BIAS_RESIDUAL:
      OVQA_REQUIRE(residual != nullptr, OVQA_ERR_BAD_ARG, "linear_fwd: residual epilogue needs a residual");
      return launch<false, false>(w, K, x, ldx, N, M, K,
                                  MEpiBiasResidual{(bf16*)y, ldy, bias, (const bf16*)residual, ldres, (int)N, da, DropState{}}, st,
                                  "linear_fwd(mfma,residual)", wide);
  }
  ovqa_set_error("linear_fwd: unknown epilogue %d", epilogue);
  return OVQA_ERR_BAD_ARG;
}

bool mfma_linear_fwd_split3_supported(const void* x, int64_t ldx, const void* w, const float* bias, const void* y0,
                                      int64_t ld0, const void* y1, int64_t ld1, const void* y2, int64_t ld2, int64_t M,
                                      int64_t F, int64_t K) {
  return mfma_gemm_supported(3 * F, M, K, K, ldx) && F % 8 == 0 && ld0 % 8 == 0 && ld1 % 8 == 0 && ld2 % 8 == 0 &&
         aligned16(x) && aligned16(w) && aligned16(y0) && aligned16(y1) && aligned16(y2) && (!bias || aligned16(bias));
}

int mfma_linear_fwd_split3(const void* x, int64_t ldx, const void* w, const float* bias, void* y0, int64_t ld0, void* y1,
                           int64_t ld1, void* y2, int64_t ld2, int64_t M, int64_t F, int64_t K, hipStream_t st) {
  return launch<false, false>(w, K, x, ldx, 3 * F, M, K,
                              MEpiBiasSplit3{(bf16*)y0, ld0, (bf16*)y1, ld1, (bf16*)y2, ld2, (int)F, bias}, st,
                              "linear_fwd_split3", true);
}

int mfma_linear_fwd_res32(const void* x, int64_t ldx, const void* w, const float* bias, const float* residual,
                          int64_t ldres, const float* mean, const float* rstd, const float* gamma, const float* beta,
                          float* pre, int64_t ldpre, int64_t M, int64_t N, int64_t K, const DropArgs& da, hipStream_t st) {
  OVQA_REQUIRE(aligned16(x) && aligned16(w) && aligned16(pre) && aligned16(residual) && (!bias || aligned16(bias)) &&
                   (!gamma || (aligned16(gamma) && aligned16(beta))) && ldpre % 4 == 0 && ldres % 4 == 0,
               OVQA_ERR_BAD_ARG, "linear_fwd_res32: pointers must be 16-byte aligned, ldpre / ldres multiples of 4");
  return launch<false, false>(w, K, x, ldx, N, M, K,
                              MEpiBiasRes32{pre, ldpre, bias, residual, ldres, mean, rstd, gamma, beta, (int)N, da, DropState{}},
                              st, "linear_fwd_res32(mfma)", true);
}

// dX from a TRANSPOSED weight copy wt[K, N] (row stride ldwt): the forward-type kernel (row-major P tile).
// MEASURED in the MCAN step: the k-major P staging of the [N, K] form costs +0.44 ms per step (4.55 -> 4.11 ms).
int mfma_linear_bwd_data_wt(const void* dy, int64_t lddy, const void* wt, int64_t ldwt, void* dx, int64_t lddx,
                            const void* preact, const void* addend, int64_t ldadd, int64_t M, int64_t N, int64_t K,
                            const DropArgs& da, hipStream_t st) {
  OVQA_REQUIRE(aligned16(dy) && aligned16(wt) && ((uintptr_t)dx % 8 == 0) && (!preact || (uintptr_t)preact % 8 == 0) &&
                   (!addend || ((uintptr_t)addend % 8 == 0 && ldadd % 4 == 0)),
               OVQA_ERR_BAD_ARG, "linear_bwd_data_wt(bf16): pointer alignment");
  const bool wide = aligned16(dx) && lddx % 8 == 0 && (!preact || aligned16(preact)) &&
                    (!addend || (aligned16(addend) && ldadd % 8 == 0));
  return launch<false, false>(wt, ldwt, dy, lddy, K, M, N,
                              MEpiBwdData{(bf16*)dx, lddx, (const bf16*)preact, (int)K, (const bf16*)addend, ldadd, da, DropState{}}, st,
                              "linear_bwd_data_wt(mfma)", wide);
}

// Grouped transpose of bf16 matrices through LDS: dst[c, r] = src[r, c], 64x64 tiles, 16-byte global accesses.
// grid (max tiles of any problem, problems).
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const ovqa_transpose_problem* __restrict__ probs) {
  __shared__ bf16 tile[64][64 + 8];
  const ovqa_transpose_problem pr = probs[blockIdx.y];
  const int tiles_c = (pr.cols + 63) / 64, tiles_r = (pr.rows + 63) / 64;
  if ((int)blockIdx.x >= tiles_r * tiles_c) return;
  const int tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
  const bf16* src = (const bf16*)pr.src;
  bf16* dst = (bf16*)pr.dst;
  const int t = threadIdx.x;
  // load: 64 rows x 8 chunks of 8 elements; thread -> (row = t / 8 + 32 * pass, chunk = t % 8)
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const int r = t / 8 + 32 * pass, ch = t % 8;
    const int gr = tr * 64 + r, gc = tc * 64 + ch * 8;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (bf16)0.f;
    if (gr < pr.rows && gc < pr.cols) v = *reinterpret_cast<const bf16x8*>(src + (int64_t)gr * pr.ld_src + gc);
#pragma unroll
    for (int e = 0; e < 8; e++) tile[r][ch * 8 + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const int c = t / 8 + 32 * pass, ch = t % 8;  // output row = source column
    const int gc = tc * 64 + c, gr = tr * 64 + ch * 8;
    if (gc < pr.cols && gr < pr.rows) {
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = tile[ch * 8 + e][c];
      *reinterpret_cast<bf16x8*>(dst + (int64_t)gc * pr.ld_dst + gr) = v;
    }
  }
}

int grouped_transpose_bf16(const ovqa_transpose_problem* probs, int n, int max_tiles, hipStream_t st) {
  if (n <= 0 || max_tiles <= 0) return OVQA_OK;
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((unsigned)max_tiles, (unsigned)n), dim3(256), 0, st, probs);
  return ovqa_check_launch("grouped_transpose");
}

bool mfma_linear_bwd_data_supported(int64_t M, int64_t N, int64_t K, int64_t lddy, int64_t lddx) {
  // dx[m,i] = sum_n dy[m,n] w[n,i]: r = i (R = K), c = m, reduction over N
  return mfma_gemm_supported(K, M, N, K, lddy) && lddx % 4 == 0;
}

int mfma_linear_bwd_data(const void* dy, int64_t lddy, const void* w, void* dx, int64_t lddx, const void* preact,
                         const void* addend, int64_t ldadd, int64_t M, int64_t N, int64_t K, const DropArgs& da,
                         hipStream_t st) {
  OVQA_REQUIRE(aligned16(dy) && aligned16(w) && ((uintptr_t)dx % 8 == 0) && (!preact || (uintptr_t)preact % 8 == 0) &&
                   (!addend || ((uintptr_t)addend % 8 == 0 && ldadd % 4 == 0)),
               OVQA_ERR_BAD_ARG, "linear_bwd_data(bf16): pointer alignment");
  return launch<true, false>(w, K, dy, lddy, K, M, N,
                             MEpiBwdData{(bf16*)dx, lddx, (const bf16*)preact, (int)K, (const bf16*)addend, ldadd, da, DropState{}}, st,
                             "linear_bwd_data(mfma)");
}

bool mfma_linear_bwd_weight_supported(int64_t M, int64_t N, int64_t K, int64_t lddy, int64_t ldx) {
  // dw[n,i] = sum_m dy[m,n] x[m,i]: r = i (R = K), c = n (C = N), reduction over M (any length, zero filled)
  return K >= 8 && K % 8 == 0 && N >= 8 && N % 8 == 0 && lddy % 8 == 0 && ldx % 8 == 0 && M >= 1;
}

int mfma_linear_bwd_weight(const void* dy, int64_t lddy, const void* x, int64_t ldx, float* dw, float* db, int64_t M,
                           int64_t N, int64_t K, int accumulate, int accumulate_db, hipStream_t st) {
  OVQA_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(dw), OVQA_ERR_BAD_ARG,
               "linear_bwd_weight(bf16): pointer alignment");
  // the reduction length is passed as "K" of the kernel; K % 8 is not required for k-major operands
  GemmArgs g{(const bf16*)x, ldx, (const bf16*)dy, lddy, (int)K, (int)N, (int)M,
             (int)((K + BT - 1) / BT), (int)((N + BT - 1) / BT)};
  hipLaunchKernelGGL(gemm_bf16_wgrad_kernel, dim3(g.tiles_r * g.tiles_c), dim3(256), 4 * TILE_BYTES, st, g,
                     MEpiWgrad{dw, K, accumulate, db, accumulate_db});
  return ovqa_check_launch("linear_bwd_weight(mfma)");
}

int mfma_grouped_wgrad(const ovqa_wgrad_problem* probs_dev, const int32_t* tiles_dev, int64_t n_tiles, bool direct_to_lds,
                       hipStream_t st) {
  if (n_tiles == 0) return OVQA_OK;
  if (direct_to_lds) {  // (every reduction length a multiple of 64)
    hipLaunchKernelGGL((gemm_bf16_grouped_wgrad_glds_kernel<2>), dim3((unsigned)n_tiles), dim3(512), 4 * TILE_BYTES, st,
                       probs_dev, reinterpret_cast<const int4*>(tiles_dev));
    return ovqa_check_launch("grouped_linear_bwd_weight(mfma,glds)");
  }
  hipLaunchKernelGGL(gemm_bf16_grouped_wgrad_kernel, dim3((unsigned)n_tiles), dim3(256), 4 * TILE_BYTES, st, probs_dev,
                     reinterpret_cast<const int4*>(tiles_dev));
  return ovqa_check_launch("grouped_linear_bwd_weight(mfma)");
}

int mfma_grouped_linear_bwd_weight_adam(const ovqa_wgrad_problem* probs_dev, const int32_t* tiles_dev, int64_t n_tiles,
                                        const ovqa_adam_target* targets_dev, const ovqa_adam_consts& consts, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_grouped_wgrad_adam_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    if (e != hipSuccess) {
      ovqa_set_error("grouped_linear_bwd_weight_adam: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return OVQA_ERR_LAUNCH;
    }
    attr = true;
  }
  hipLaunchKernelGGL(gemm_bf16_grouped_wgrad_adam_kernel, dim3((unsigned)n_tiles), dim3(512), 4 * TILE_BYTES, st, probs_dev,
                     reinterpret_cast<const int4*>(tiles_dev), targets_dev, consts);
  return ovqa_check_launch("grouped_linear_bwd_weight_adam(mfma,glds)");
}

int colsum_bf16(const void* dy, int64_t lddy, float* db, int64_t M, int64_t N, int accumulate, hipStream_t st) {
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(db, 0, (size_t)N * sizeof(float), st);
    if (e != hipSuccess) {
      ovqa_set_error("colsum: hipMemsetAsync: %s", hipGetErrorString(e));
      return OVQA_ERR_LAUNCH;
    }
  }
  const int col_blocks = (int)((N / 8 + 63) / 64);
  int slabs = (int)((M + 63) / 64);
  if (slabs > 512) slabs = 512;
  const int rows_per_slab = (int)((M + slabs - 1) / slabs);
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3(col_blocks, slabs), dim3(256), 0, st, (const bf16*)dy, lddy, db, (int)M,
                     (int)N, rows_per_slab);
  return ovqa_check_launch("colsum_bf16");
}

}  // namespace ovqa
