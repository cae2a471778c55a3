// Single-query attention for autoregressive decoding (models/modules/attentions.py:314-327 with one new
// query per sequence, decoders.py:46-63, beam_search.py:41-62): o[r] = softmax(q[r] K[r / group]^T * scale + mask[r]) V[r / group].
//
//   * one WAVE per (row r, head h): no 32-query MFMA tile padded around a single query; K and V of the (sample, head)
//     are streamed exactly once per wave, 16 bytes per lane;
//   * the K / V caches are [rows_kv, Lmax, H*d] buffers written in place by the projection GEMMs (row stride Lmax * H*d):
//     nothing is concatenated; `n` keys are live;
//   * `group` consecutive query rows share one cache row (the beams of a sample attend to the SAME encoder positions:
//     the projected encoder K / V are kept once per sample, beam_search.py:19-34 would gather them into every beam);
//   * scores: 4 lanes per key (16 features each, two 16-byte loads), 16 keys per pass, fp32; softmax over LDS-resident
//     scores with DPP wave reductions; P.V: a lane owns a feature pair, the two half-waves take alternate keys.
// HBM-bound byte streaming (d = 64: 256 B per key and head), VALU arithmetic: the roofline is the cache bytes.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int LMAX_KEYS = 512;  // scores of one wave and query live in LDS: 4 waves x G <= 4 queries x 2 KiB

template <typename T> struct Vec16;  // 16 bytes of T
template <> struct Vec16<bf16> { typedef bf16x8 type; static constexpr int N = 8; };
template <> struct Vec16<float> { typedef f32x4 type; static constexpr int N = 4; };

// G = query rows per wave: the `group` beams of a sample (G = group <= 4: K and V of the sample's head are read ONCE for
// all of them) or 1.  Scores: 4 lanes per key, 16 keys per pass; P.V: 16-byte chunks of a V row per lane, 64 / chunks
// keys per pass, partial sums of the key groups reduced with xor shuffles at the end.
// SPLIT = false: a wave owns a (row group, head) -- 4 independent problems per workgroup (short key lists: the
// self-attention prefix).  SPLIT = true: the 4 waves of a workgroup share ONE (row group, head) and take a quarter of the
// keys each (the 237 encoder positions: a lone wave streams its 60 KB at the latency of its own dependent round trips,
// 28 us measured; four waves and twice the workgroups cut the chain to a quarter), partial softmaxes merged through LDS
// (running maximum / sum, flash-decoding style).
template <typename T, int D, int G, bool SPLIT>
__global__ __launch_bounds__(256) void attn_decode_kernel(ovqa::AttnDecodeArgs a) {
  constexpr int SLOT = SPLIT ? LMAX_KEYS / 4 : LMAX_KEYS;  // scores a wave keeps per query
  __shared__ float sc[4][G][SLOT];
  __shared__ float red_m[4][G], red_s[4][G];
  __shared__ float red_o[SPLIT ? 4 : 1][G][D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wid = SPLIT ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
  const int rows_g = a.R / G;
  const bool live = wid < rows_g * a.H;
  if (!SPLIT && !live) return;  // (whole waves; the non-split form has no workgroup barrier)
  const int rg = live ? wid / a.H : 0, h = live ? wid - rg * a.H : 0;
  const int r0 = rg * G;  // first query row
  // this wave's keys [j_lo, j_hi): everything, or a quarter rounded up to whole 16-key passes
  const int quarter = ((a.n + 3) / 4 + 15) & ~15;
  const int j_lo = SPLIT ? min(wave * quarter, a.n) : 0;
  const int n = SPLIT ? min(j_lo + quarter, a.n) : a.n;  // j_hi
  const int64_t kvrow = (int64_t)(r0 / a.group) * a.kv_batch_stride;
  const T* kb = (const T*)a.k + kvrow + h * D;
  const T* vb = (const T*)a.v + kvrow + h * D;
  constexpr int PER = D / 4;                      // features per lane of a key's 4-lane group
  constexpr int NV = Vec16<T>::N, NL = PER / NV;  // 16-byte loads per lane and key
  static_assert(PER % NV == 0, "a lane's share of a key is a whole number of 16-byte loads");
  typedef typename Vec16<T>::type vec_t;
  const int part = lane & 3, kk = lane >> 2;
  float qf[G][PER];
#pragma unroll
  for (int g = 0; g < G; g++) {
    const T* q = (const T*)a.q + (int64_t)(r0 + g) * a.ldq + h * D;
#pragma unroll
    for (int l = 0; l < NL; l++) {
      const vec_t v = *reinterpret_cast<const vec_t*>(q + part * PER + l * NV);
#pragma unroll
      for (int e = 0; e < NV; e++) qf[g][l * NV + e] = to_f32<T>(v[e]) * a.scale;
    }
  }
  // ---- the first V rows this lane will need are requested NOW, next to the K rows: they do not depend on the scores, and
  // the wave's time is its chain of dependent memory round trips (K -> scores -> softmax -> V -> o was two of them + one
  // more per further 32 keys; a quarter of 237 keys is now ONE round trip for K and V together)
  constexpr int CH = D * (int)sizeof(T) / 16;  // 16-byte chunks per V row: 8 (bf16, d = 64)
  constexpr int KG = 64 / CH;                  // keys in flight per pass
  static_assert(CH >= 1 && CH <= 64 && 64 % CH == 0, "chunks per row divide the wave");
  const int c = lane % CH, kg = lane / CH;
  constexpr int UV = 8;  // V rows in flight per lane
  vec_t vpre[UV];
#pragma unroll
  for (int u = 0; u < UV; u++)
    vpre[u] = *reinterpret_cast<const vec_t*>(vb + (int64_t)min(j_lo + kg + KG * u, max(n - 1, 0)) * a.ldv + c * NV);
  // ---- scores: 16 keys per pass, every key row loaded once for the G queries
  float mx[G];
#pragma unroll
  for (int g = 0; g < G; g++) mx[g] = -INFINITY;
  constexpr int UK = 4;  // passes in flight: the loads of 64 keys are issued before the first dot product
  for (int j0 = j_lo; j0 < n; j0 += 16 * UK) {
    vec_t kv[UK][NL];
    float mk[UK][G];  // the additive mask of these keys: requested with the K rows, not one dependent load per use
#pragma unroll
    for (int u = 0; u < UK; u++) {
      const int j = min(j0 + 16 * u + kk, n - 1);  // (clamped: always a valid row; results beyond n are dropped)
      const T* kr = kb + (int64_t)j * a.ldk + part * PER;
#pragma unroll
      for (int l = 0; l < NL; l++) kv[u][l] = *reinterpret_cast<const vec_t*>(kr + l * NV);
#pragma unroll
      for (int g = 0; g < G; g++) mk[u][g] = a.mask ? a.mask[(int64_t)(r0 + g) * a.ldmask + j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UK; u++) {
      const int j = j0 + 16 * u + kk;
      float dot[G];
#pragma unroll
      for (int g = 0; g < G; g++) dot[g] = 0.f;
#pragma unroll
      for (int l = 0; l < NL; l++)
#pragma unroll
        for (int e = 0; e < NV; e++) {
          const float kf = to_f32<T>(kv[u][l][e]);
#pragma unroll
          for (int g = 0; g < G; g++) dot[g] = fmaf(qf[g][l * NV + e], kf, dot[g]);
        }
#pragma unroll
      for (int g = 0; g < G; g++) {
        dot[g] += __shfl_xor(dot[g], 1, 64);
        dot[g] += __shfl_xor(dot[g], 2, 64);
        if (j < n) {
          const float sv = dot[g] + mk[u][g];
          if (part == 0) sc[wave][g][j - j_lo] = sv;
          mx[g] = fmaxf(mx[g], sv);
        }
      }
    }
  }
  // ---- softmax numerators (the wave's own LDS rows: written and read by this wave only)
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the score stores of every lane have landed
  __builtin_amdgcn_wave_barrier();
  float inv[G], wmax[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    const float m = wave_max(mx[g]);
    float sum = 0.f;
    for (int j = j_lo + lane; j < n; j += 64) {
      const float p = __expf(sc[wave][g][j - j_lo] - m);
      sc[wave][g][j - j_lo] = p;
      sum += p;
    }
    sum = wave_sum(sum);
    wmax[g] = m;
    inv[g] = SPLIT ? sum : 1.f / sum;  // (split: the row sum itself, merged below)
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  // ---- o = P V: lane -> 16-byte chunk c of key group kg
  float acc[G][NV];
#pragma unroll
  for (int g = 0; g < G; g++)
#pragma unroll
    for (int e = 0; e < NV; e++) acc[g][e] = 0.f;
  bool first = true;
  for (int j0 = j_lo + kg; j0 < n; j0 += KG * UV, first = false) {
    vec_t vv[UV];
    if (first) {  // (the first trip's rows are already here)
#pragma unroll
      for (int u = 0; u < UV; u++) vv[u] = vpre[u];
    } else {
#pragma unroll
      for (int u = 0; u < UV; u++)
        vv[u] = *reinterpret_cast<const vec_t*>(vb + (int64_t)min(j0 + KG * u, n - 1) * a.ldv + c * NV);
    }
#pragma unroll
    for (int u = 0; u < UV; u++) {
      const int j = j0 + KG * u;
      if (j < n) {
#pragma unroll
        for (int g = 0; g < G; g++) {
          const float p = sc[wave][g][j - j_lo];
#pragma unroll
          for (int e = 0; e < NV; e++) acc[g][e] = fmaf(p, to_f32<T>(vv[u][e]), acc[g][e]);
        }
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; g++) {
#pragma unroll
    for (int e = 0; e < NV; e++) {
      float v = acc[g][e];
#pragma unroll
      for (int off = CH; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
      acc[g][e] = v;
    }
    if constexpr (!SPLIT) {
      if (kg == 0) {
        T* orow = (T*)a.o + (int64_t)(r0 + g) * a.ldo + h * D + c * NV;
        vec_t ov;
#pragma unroll
        for (int e = 0; e < NV; e++) ov[e] = from_f32<T>(acc[g][e] * inv[g]);
        *reinterpret_cast<vec_t*>(orow) = ov;
      }
    } else {
      if (kg == 0) {
#pragma unroll
        for (int e = 0; e < NV; e++) red_o[wave][g][c * NV + e] = acc[g][e];
      }
      if (lane == 0) {
        red_m[wave][g] = wmax[g];  // -inf for a wave without keys: its terms vanish in the merge
        red_s[wave][g] = inv[g];
      }
    }
  }
  if constexpr (SPLIT) {
    __syncthreads();
    for (int t = threadIdx.x; t < G * D; t += 256) {
      const int g = t / D, f = t - g * D;
      float M = red_m[0][g];
#pragma unroll
      for (int w = 1; w < 4; w++) M = fmaxf(M, red_m[w][g]);
      float S = 0.f, o = 0.f;
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const float e = __expf(red_m[w][g] - M);  // exp(-inf) = 0: empty quarters drop out
        S = fmaf(red_s[w][g], e, S);
        o = fmaf(red_o[w][g][f], e, o);
      }
      if (live) ((T*)a.o)[(int64_t)(r0 + g) * a.ldo + h * D + f] = from_f32<T>(o / S);
    }
  }
}

// ---- the k best of every row (k <= 8): candidate selection of a beam-search step (beam_search.py:36-39 sorts all
// cur_beam * |V| candidates; any of the overall best `beam` is among the best `beam` of its own beam's |V| words).
// One wave per row: every lane keeps the best k of its strided share in registers (sorted, insertion), then k rounds
// of a wave-wide arg-max pop the winners.  Ties: the smaller index first.
template <int K>
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ x, int64_t ldx, int R, int V,
                                                        float* __restrict__ vals, int64_t* __restrict__ idx) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float* xr = x + (int64_t)row * ldx;
  float bv[K];
  int bi[K];
#pragma unroll
  for (int t = 0; t < K; t++) { bv[t] = -INFINITY; bi[t] = 0x7fffffff; }
  auto offer = [&](float v, int i) {
    // (a lane visits its indices in increasing order: '>' keeps the smaller index on ties; an empty slot takes anything)
    if (v > bv[K - 1] || bi[K - 1] == 0x7fffffff) {
#pragma unroll
      for (int t = 0; t < K; t++) {
        const bool take = v > bv[t] || bi[t] == 0x7fffffff;
        const float ov = bv[t];
        const int oi = bi[t];
        bv[t] = take ? v : ov;
        bi[t] = take ? i : oi;
        v = take ? ov : v;
        i = take ? oi : i;
      }
    }
  };
  // 16-byte loads, four of them in flight per lane (one load per step left the wave at the latency of ~60 round trips)
  const bool vec = (((uintptr_t)xr | (uintptr_t)(ldx * 4)) & 15) == 0;
  const int V4 = vec ? V / 4 : 0;
  for (int c0 = lane; c0 < V4; c0 += 64 * 4) {
    float4 q[4];
#pragma unroll
    for (int u = 0; u < 4; u++) q[u] = *reinterpret_cast<const float4*>(xr + 4 * min(c0 + 64 * u, V4 - 1));
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int c = c0 + 64 * u;
      if (c < V4) {
        offer(q[u].x, 4 * c); offer(q[u].y, 4 * c + 1); offer(q[u].z, 4 * c + 2); offer(q[u].w, 4 * c + 3);
      }
    }
  }
  for (int j = 4 * V4 + lane; j < V; j += 64) offer(xr[j], j);
#pragma unroll
  for (int t = 0; t < K; t++) {
    const float best = wave_max(bv[0]);
    // among the lanes that hold `best` at their head, the smallest index wins
    int cand = bv[0] == best ? bi[0] : 0x7fffffff;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) cand = min(cand, __shfl_xor(cand, off, 64));
    if (lane == 0) {
      vals[(int64_t)row * K + t] = best;
      idx[(int64_t)row * K + t] = cand;
    }
    if (bv[0] == best && bi[0] == cand) {  // pop the winner's head
#pragma unroll
      for (int u = 0; u < K - 1; u++) { bv[u] = bv[u + 1]; bi[u] = bi[u + 1]; }
      bv[K - 1] = -INFINITY;
      bi[K - 1] = 0x7fffffff;
    }
  }
}

}  // namespace

namespace ovqa {

int topk_rows(const float* x, int64_t ldx, int64_t R, int64_t V, int k, float* vals, int64_t* idx, hipStream_t st) {
  if (R == 0) return OVQA_OK;
  const dim3 grid((unsigned)((R + 3) / 4)), block(256);
#define OVQA_TK(KV) hipLaunchKernelGGL((topk_rows_kernel<KV>), grid, block, 0, st, x, ldx, (int)R, (int)V, vals, idx)
  switch (k) {
    case 1: OVQA_TK(1); break;
    case 2: OVQA_TK(2); break;
    case 3: OVQA_TK(3); break;
    case 4: OVQA_TK(4); break;
    case 5: OVQA_TK(5); break;
    case 6: OVQA_TK(6); break;
    case 7: OVQA_TK(7); break;
    default: OVQA_TK(8);
  }
#undef OVQA_TK
  return ovqa_check_launch("topk_rows");
}

bool attention_decode_supported(const AttnDecodeArgs& a, int esize) {
  const auto al = [&](const void* p) { return ((uintptr_t)p & 15) == 0; };
  return (a.d == 64 || a.d == 32 || a.d == 128) && a.n >= 1 && a.n <= LMAX_KEYS && a.group >= 1 && al(a.q) && al(a.k) &&
         al(a.v) && (a.ldq * esize) % 16 == 0 && (a.ldk * esize) % 16 == 0 && (a.ldv * esize) % 4 == 0 &&
         (a.ldv * esize) % 16 == 0 && (a.kv_batch_stride * esize) % 16 == 0 && (a.ldo * esize) % 16 == 0 && al(a.o);
}

template <typename T, int D>
static int launch_decode(const AttnDecodeArgs& a, hipStream_t st) {
  // G queries per wave: the beams of a sample share its K / V rows (group in {2, 3, 4} and R a multiple of it)
  const int G = (a.group >= 2 && a.group <= 4 && a.R % a.group == 0) ? a.group : 1;
  const int64_t problems = (int64_t)(a.R / G) * a.H;
  static int split_min = -1;
  if (split_min < 0) {
    const char* e = getenv("OVQA_DECODE_SPLIT_MIN");
    split_min = e ? atoi(e) : 64;
  }
  const bool split = a.n >= split_min;  // long key lists: the four waves of a workgroup share one problem
  const dim3 grid((unsigned)(split ? problems : (problems + 3) / 4)), block(256);
#define OVQA_DEC(GV)                                                                              \
  if (split) hipLaunchKernelGGL((attn_decode_kernel<T, D, GV, true>), grid, block, 0, st, a);     \
  else hipLaunchKernelGGL((attn_decode_kernel<T, D, GV, false>), grid, block, 0, st, a)
  switch (G) {
    case 2: OVQA_DEC(2); break;
    case 3: OVQA_DEC(3); break;
    case 4: OVQA_DEC(4); break;
    default: OVQA_DEC(1);
  }
#undef OVQA_DEC
  return ovqa_check_launch("attention_decode");
}

int attention_decode(int dtype, const AttnDecodeArgs& a, hipStream_t st) {
  if (a.R == 0 || a.H == 0) return OVQA_OK;
  if (dtype == OVQA_BF16) {
    switch (a.d) {
      case 32: return launch_decode<bf16, 32>(a, st);
      case 64: return launch_decode<bf16, 64>(a, st);
      default: return launch_decode<bf16, 128>(a, st);
    }
  }
  switch (a.d) {
    case 32: return launch_decode<float, 32>(a, st);
    case 64: return launch_decode<float, 64>(a, st);
    default: return launch_decode<float, 128>(a, st);
  }
}

}  // namespace ovqa
