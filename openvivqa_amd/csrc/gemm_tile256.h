// 256 x 256 output tiles on ONE 8-wave workgroup per CU (round 6): Out(c, r) = sum_k Q(c, k) P(r, k), both operands
// k-contiguous ([row][k]), K % 64 == 0.  Shared by gemm_mfma.hip (the product's launch()) and scripts/gemm256_dev.hip (the
// variant bench this file was tuned with; numbers in profiles/README.md, round 6).
//
// Why this tile: a CU takes in 60-80 GB/s through its L2 -> LDS path whatever the workgroup does with the bytes (DESIGN.md
// section 5; measured here with every MFMA and fragment read removed: 78 GB/s per CU with all 256 CUs streaming, 113 with
// 50).  A 128 x 128 tile moves 32 KB per 64-deep K step for 2.1 MFLOP (64 flop/B: <= 4.5-5 TFLOP/s per CU -- where the
// 128 x 128 kernels of rounds 1-5 sit: 0.9-1.05 PFLOP/s at 8192 x 4096 x 4096); a 256 x 256 tile moves 64 KB for 8.4 MFLOP
// (128 flop/B): at the matrix pipes' own rate (1.08 us per K step measured with DMA and reads removed) the fetch path is
// asked for 60 GB/s.
//
// Schedule:
//   * 2 (r) x 4 (c) wave grid, 128 (r) x 64 (c) outputs per wave: 128 accumulator registers, 64 MFMAs (16x16x32) per wave
//     and K step; a wave reads 24 KB of fragments per K step (8 waves: 192 KB against 256 KB for 16 waves of 64 x 64).
//   * two 64-KB LDS stages (P tile | Q tile, the 128-byte-row XOR-swizzled image of the other kernels) filled by
//     global_load_lds_dwordx4: 8 one-KB pieces per wave and K step, lean addressing (a 32-bit per-lane offset per piece
//     computed once + a scalar base that advances by 128 B per K step: the saddr form, one s_add to M0 per piece).
//   * the fragment REGISTERS are the third pipeline stage (tile_e below): all 24 fragments of a K step are read at its start,
//     a second barrier releases the LDS stage, and the DMA of step k+2 is requested into it while step k+1's is still in
//     flight -- a ring of two alone pays the DMA's issue -> landed latency in every K step.
//   * ROT: waves 4-7 (the second wave of every SIMD) run half a K step out of phase: their loop is rotated so that they
//     issue the 32 MFMAs of their second half step out of registers while waves 0-3 sit in their read burst, and read while
//     waves 0-3 issue MFMAs.  PRIO: one s_setprio 1 for waves 4-7 in front of the loop (MI355X_MICROARCH.md, "Static
//     priority for the younger half").  Every instruction group is pinned by sched_barriers: the compiler only allocates
//     registers and places the waits.
// Measured (one box, eager launches back to back, scripts/gemm256_dev.hip): 8192 x 4096 x 4096: 128 x 128 kernels 894-909
// TFLOP/s, this tile 1250-1270 (ring of two with one stage in flight: 1165-1200; without ROT: 1090; without PRIO: 1180;
// ks = 1 fragments read between the MFMAs of ks = 0: 1210), hipBLASLt on the same box 1424.  Per K step of a CU: 1.69 us;
// MFMAs alone 1.08, DMA alone 0.84, fragment reads alone 0.59, reads + MFMA 1.44, DMA + MFMA 1.26, DMA + reads 1.05: about
// half of the non-MFMA work is still exposed.  6400 x 2048 x 512 (200 tiles, 8 K steps): 21.2 us against 23.4 (hipBLASLt
// 21.2): the K loop is 13.6 of them, launch + first stages + the 128-KB-per-CU epilogue the rest.
#pragma once
#include <type_traits>

#include "common.h"

namespace ovqa_t256 {

constexpr int T = 256, TK = 64;
constexpr int HALF_BYTES = T * TK * 2;   // one operand tile: 32 KiB
constexpr int STAGE_BYTES = 2 * HALF_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;  // ring of two

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

struct Args {
  const bf16* P; int64_t ldp;  // r operand (rows = output features)
  const bf16* Q; int64_t ldq;  // c operand (rows = tokens)
  int R, C, K;
  int tiles_r, tiles_c;
  unsigned long long* probe;  // variant bench only (FLAGS bit 8): 4 wall-clock stamps (100 MHz) per workgroup
  int group_c;  // > 1: tile order in groups of `group_c` c panels (c fastest inside a group): the tiles resident together on an XCD
                // cover group_c x (32 / group_c) panels instead of 2 x 16 -- fewer distinct bytes per K step through its L2
};

__device__ __forceinline__ int perm32(int s) { return (s & ~31) | ((s & 12) << 1) | (((s >> 4) & 1) << 2) | (s & 3); }

// ---- two LDS stages + the fragment registers = a pipeline THREE K steps deep ------------------------------------------------
// FLAGS: bit 0 = ROT, bit 1 = PRIO (bits 4-6: ablation switches of the variant bench).
// A ring of two with one stage in flight measured 1.6-1.8 us per K step whether 50 or 256 CUs were busy: it pays the DMA's whole issue -> landed latency in every step (stage k+2 can only be requested once stage k+1 has
// landed and stage k has been consumed).  Here ALL 24 fragments of a K step are read into registers first (96 registers),
// a second barrier (M) says that every wave has done so, and stage k+2 is requested right behind it into the buffer just read
// -- while stage k+1 is still in flight and the 64 MFMAs of step k have not started: two stages in flight at any time.
// MEASURED (8192 x 4096 x 4096, profiles/r06_gemm256_variants.txt sweeps D, E): MFMAs alone 1.05 us per K step, fragment reads alone
// 0.57, both 1.47 (1.68 without ROT, 1.55 with every wave requesting a half step's reads one 32-MFMA block ahead of their use):
// on this chip ds_read_b128 traffic and MFMAs add up rather than overlap, whatever the phase or the interleaving -- the 1.43 us
// per equivalent step that hipBLASLt measures here is what the same sum gives for 128 x 128 outputs per wave (256 instead of
// 384 LDS bytes per MFMA: 1.05 + 0.38); its kernel was not inspected.  That shape built here from this header's pieces -- FOUR waves of
// 128 x 128 outputs, accumulators in AGPRs, no spills -- ran 2.03 us per K step (1059 TFLOP/s): MFMAs 1.12 + reads 0.46 add up there
// too (1.62), and with one wave per SIMD the DMA's 0.94 has nothing to hide behind (sweep F).  Removed again.
//   straight waves: [vmcnt, S_k] RD(k)  MM0(k)  16 of MM1(k)  [M_k]  the other 16 of MM1(k) || DMA(k+2)
//   rotated waves : [vmcnt, S_k] MM1(k-1)  RD(k)  [M_k]  MM0(k) || DMA(k+2)              (ROT; MM1 of the last step after the loop)
// Without ROT every wave runs: [vmcnt, S_k] RD(k) [M_k] DMA(k+2) MM0(k) MM1(k).
template <typename Epi, int FLAGS>
__device__ __forceinline__ void tile_e(const Args& g, int c0, int r0, Epi& epi, char* smem) {
  constexpr bool ROTF = FLAGS & 1, PRIO = FLAGS & 2;
  // ablation switches of the variant bench (wrong results, timing only): no DMA in the loop / no MFMAs / no fragment reads;
  // bit 9: 4 instead of 16 bytes per lane and DMA instruction; bit 10: every K step fetches the same two 64-byte column blocks
  constexpr bool NO_DMA = FLAGS & 16, NO_MFMA = FLAGS & 32, NO_READ = FLAGS & 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave & 1, wc = (wave >> 1) & 3;
  const bool rot = ROTF && wave >= 4;

  f32x4 acc[8][4];
#pragma unroll
  for (int j = 0; j < 8; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int swz = ((lane & 7) ^ (lane >> 3)) << 4;
  uint32_t vp[4], vq[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    int pr = r0 + perm32(row), qr = c0 + row;
    pr = (pr < g.R ? pr : g.R - 1) - r0;
    qr = (qr < g.C ? qr : g.C - 1) - c0;
    vp[i] = (uint32_t)((int64_t)pr * g.ldp * 2) + swz;
    vq[i] = (uint32_t)((int64_t)qr * g.ldq * 2) + swz;
  }
  const char* pbase = reinterpret_cast<const char*>(g.P + (int64_t)r0 * g.ldp);
  const char* qbase = reinterpret_cast<const char*>(g.Q + (int64_t)c0 * g.ldq);
  char* const dma_dst = smem + wave * 4096;
  auto dma_piece = [&](int kt, int n) {
    if constexpr (NO_DMA) {
      if (kt > 1) return;
    }
    char* dst = dma_dst + (kt & 1) * STAGE_BYTES + (n >> 2) * HALF_BYTES + (n & 3) * 1024;
    uint32_t o = n < 4 ? vp[n & 3] : vq[n & 3];
    asm volatile("" : "+v"(o));
    const char* src = (n < 4 ? pbase : qbase) + (int64_t)((FLAGS & 1024) ? (kt & 1) : kt) * (TK * 2) + o;
    if constexpr (FLAGS & 512) __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)dst, 4, 0, 0);
    else __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)dst, 16, 0, 0);
  };

  const int frow = lane & 15;
  const int fo0 = frow * 128 + (((lane >> 4) ^ (lane & 7)) << 4);
  const int fo1 = frow * 128 + (((4 + (lane >> 4)) ^ (lane & 7)) << 4);
  const int poff = wr * (128 * 128), qoff = HALF_BYTES + wc * (64 * 128);
  bf16x8 pf0[8], qf0[4], pf1[8], qf1[4];
#define T256_PIN() __builtin_amdgcn_sched_barrier(0)
#define T256_MFMA(pf, qf, j, i)                                                                          \
  do {                                                                                                   \
    if constexpr (!NO_MFMA) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[j], qf[i], acc[j][i], 0, 0, 0); \
  } while (0)
  if constexpr (NO_READ) {
#pragma unroll
    for (int j = 0; j < 8; j++) pf0[j] = pf1[j] = bf16x8{};
#pragma unroll
    for (int i = 0; i < 4; i++) qf0[i] = qf1[i] = bf16x8{};
  }
  auto read_all = [&](int kt) {
    if constexpr (NO_READ) return;
    const char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; i++) qf0[i] = *reinterpret_cast<const bf16x8*>(st + qoff + fo0 + i * 2048);
#pragma unroll
    for (int j = 0; j < 8; j++) pf0[j] = *reinterpret_cast<const bf16x8*>(st + poff + fo0 + j * 2048);
#pragma unroll
    for (int i = 0; i < 4; i++) qf1[i] = *reinterpret_cast<const bf16x8*>(st + qoff + fo1 + i * 2048);
#pragma unroll
    for (int j = 0; j < 8; j++) pf1[j] = *reinterpret_cast<const bf16x8*>(st + poff + fo1 + j * 2048);
    if constexpr (NO_MFMA) {  // keep the reads alive
#pragma unroll
      for (int j = 0; j < 8; j++) asm volatile("" ::"v"(pf0[j]), "v"(pf1[j]));
#pragma unroll
      for (int i = 0; i < 4; i++) asm volatile("" ::"v"(qf0[i]), "v"(qf1[i]));
    }
  };

  const int nkt = g.K / TK;
  unsigned long long t_start = 0, t_loop = 0, t_epi = 0;
  if constexpr (FLAGS & 256) t_start = wall_clock64();
#pragma unroll
  for (int n = 0; n < 8; n++) dma_piece(0, n);
  if (nkt > 1) {
#pragma unroll
    for (int n = 0; n < 8; n++) dma_piece(1, n);
  }
  if constexpr (PRIO) {
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  }
  if constexpr (FLAGS & 256) t_loop = wall_clock64();

  // MORE: stage kt+1 has been requested (it stays in flight across barrier S); NEXT2: stage kt+2 exists
  auto step = [&](int kt, auto rot_c, auto first_c, auto more_c, auto next2_c) {
    constexpr bool ROTW = decltype(rot_c)::value;  // this wave runs the rotated program
    constexpr bool FIRST = decltype(first_c)::value, MORE = decltype(more_c)::value, NEXT2 = decltype(next2_c)::value;
    if constexpr (MORE) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // S: stage kt has landed, for every wave
    T256_PIN();
    if constexpr (!ROTW) {
      read_all(kt);
      T256_PIN();
      if constexpr (!ROTF) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // M: stage kt is in registers everywhere
        T256_PIN();
        if constexpr (NEXT2) {
#pragma unroll
          for (int n = 0; n < 8; n++) dma_piece(kt + 2, n);
          T256_PIN();
        }
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
          for (int i = 0; i < 4; i++) T256_MFMA(pf0, qf0, j, i);
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
          for (int i = 0; i < 4; i++) T256_MFMA(pf1, qf1, j, i);
        T256_PIN();
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
          for (int i = 0; i < 4; i++) T256_MFMA(pf0, qf0, j, i);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int i = 0; i < 4; i++) T256_MFMA(pf1, qf1, j, i);
        T256_PIN();
        __builtin_amdgcn_s_barrier();  // M (the rotated waves have read stage kt by now)
        T256_PIN();
#pragma unroll
        for (int j = 4; j < 8; j++) {
#pragma unroll
          for (int i = 0; i < 4; i++) {
            T256_MFMA(pf1, qf1, j, i);
            if constexpr (NEXT2) {
              if (i & 1) {
                T256_PIN();
                dma_piece(kt + 2, (j - 4) * 2 + (i >> 1));
                T256_PIN();
              }
            }
          }
        }
        T256_PIN();
      }
    } else {
      if constexpr (!FIRST) {
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
          for (int i = 0; i < 4; i++) T256_MFMA(pf1, qf1, j, i);
        T256_PIN();
      }
      read_all(kt);
      T256_PIN();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // M
      T256_PIN();
#pragma unroll
      for (int j = 0; j < 8; j++) {
#pragma unroll
        for (int i = 0; i < 4; i++) T256_MFMA(pf0, qf0, j, i);
        if constexpr (NEXT2) {
          T256_PIN();
          dma_piece(kt + 2, j);
          T256_PIN();
        }
      }
      T256_PIN();
    }
  };
  using TT = std::integral_constant<bool, true>;
  using FF = std::integral_constant<bool, false>;
  auto run = [&](auto rot_c) {
    constexpr bool ROTW = decltype(rot_c)::value;
    if (nkt == 1) {
      step(0, rot_c, TT{}, FF{}, FF{});
    } else if (nkt == 2) {
      step(0, rot_c, TT{}, TT{}, FF{});
      step(1, rot_c, FF{}, FF{}, FF{});
    } else {
      step(0, rot_c, TT{}, TT{}, TT{});
      for (int kt = 1; kt < nkt - 2; kt++) step(kt, rot_c, FF{}, TT{}, TT{});
      step(nkt - 2, rot_c, FF{}, TT{}, FF{});
      step(nkt - 1, rot_c, FF{}, FF{}, FF{});
    }
    if constexpr (ROTW) {
#pragma unroll
      for (int j = 0; j < 8; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) T256_MFMA(pf1, qf1, j, i);
    }
  };
  if constexpr (ROTF) {
    if (rot) run(TT{});
    else run(FF{});
  } else {
    run(FF{});
  }
  if constexpr (PRIO) {
    if (wave >= 4) __builtin_amdgcn_s_setprio(0);
  }
  if constexpr (FLAGS & 256) t_epi = wall_clock64();
#undef T256_MFMA
#undef T256_PIN

  // Epilogue.  A lane owns 16 pieces of 8 consecutive output features: (jp, i) -> row c0 + wc*64 + i*16 + (lane & 15),
  // features r0 + wr*128 + jp*32 + (lane >> 4)*8.  Measured with in-kernel stamps (scripts/gemm256_dev.hip): written as 16
  // guarded `if (in range) wide(...)` calls the epilogue took 5.4 us of a 21-us launch (4.3 us even for a tile that stores
  // almost nothing): behind every per-lane branch the compiler waits with vmcnt(0), i.e. for the PREVIOUS piece's store to
  // be acknowledged -- 16 store round trips in a row.  So: a tile that lies inside the matrix runs straight-line code, and
  // epilogues with kTwoPhase expose pre() = every load of a piece / post() = compute + store, software-pipelined by jp: the
  // loads of the next four pieces are in flight while the current four are stored, and no store is ever waited for.
  epi.init();
  const bool interior = c0 + T <= g.C && r0 + T <= g.R;  // (workgroup-uniform)
  if constexpr (Epi::kTwoPhase) {
    if (interior) {
      const int cb = c0 + wc * 64 + (lane & 15), rb = r0 + wr * 128 + (lane >> 4) * 8;
      typename Epi::Ctx ctx[2][4];
#pragma unroll
      for (int i = 0; i < 4; i++) ctx[0][i] = epi.pre(cb + i * 16, rb);
#pragma unroll
      for (int jp = 0; jp < 4; jp++) {
        if (jp < 3) {
#pragma unroll
          for (int i = 0; i < 4; i++) ctx[(jp + 1) & 1][i] = epi.pre(cb + i * 16, rb + (jp + 1) * 32);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) epi.post(ctx[jp & 1][i], cb + i * 16, rb + jp * 32, acc[2 * jp][i], acc[2 * jp + 1][i]);
      }
      goto done;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = c0 + wc * 64 + i * 16 + (lane & 15);
    if (c >= g.C) continue;
#pragma unroll
    for (int jp = 0; jp < 4; jp++) {
      const int r = r0 + wr * 128 + jp * 32 + (lane >> 4) * 8;
      if (r < g.R) epi.wide(c, r, acc[2 * jp][i], acc[2 * jp + 1][i]);
    }
  }
done:
  if constexpr (FLAGS & 256) {
    const unsigned long long t_issued = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_end = wall_clock64();
    if (g.probe && lane == 0) {
      unsigned long long* o = g.probe + ((size_t)blockIdx.x * 8 + wave) * 6;
      o[0] = t_start; o[1] = t_loop; o[2] = t_epi; o[3] = t_issued; o[4] = t_end; o[5] = nkt;
    }
  }
}

template <typename Epi, int FLAGS>
__global__ __launch_bounds__(512) void kernel(Args g, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nwg = g.tiles_r * g.tiles_c;
  const int bid = xcd_contiguous_block(blockIdx.x, nwg);
  int tc = bid / g.tiles_r, tr = bid % g.tiles_r;
  if (g.group_c > 1) {
    const int per = g.group_c * g.tiles_r, grp = bid / per, first = grp * g.group_c;
    const int gsz = min(g.group_c, g.tiles_c - first), in = bid - grp * per;
    tc = first + in % gsz;
    tr = in / gsz;
  }
  tile_e<Epi, FLAGS>(g, tc * T, tr * T, epi, smem);
}

}  // namespace ovqa_t256
