// The index / elementwise work AROUND the kernels of one autoregressive decoding step, in three launches instead of ~55
// stock elementwise launches (4 us each in a replayed graph, half of a 64 x 20-token beam-3 decode):
//   decode_embed      word + position embedding of the previous tokens, the running position counter, the new column of the
//                     running self-attention mask (models/modules/decoders.py:46-66 in its stateful branch)
//   beam_candidates   log-softmax of the logits + the candidate scores of models/modules/beam_search.py:41-57 (finished
//                     sequences keep their score on word 0 and get -999 elsewhere) + the k best of every row
//   beam_commit       the `beam` best of a sample's cur_beam * k survivors (beam_search.py:36-39), the bookkeeping of
//                     beam_search.py:58-83 (scores, sequence masks, word / log-prob histories that follow their beams) and
//                     the gather index of the state reorder
// All of it is byte / index movement and a row reduction: launch-bound at these sizes, nothing to tile.
#include <math.h>

#include "common.h"
#include "kernels.h"

namespace {

template <typename T>
__global__ __launch_bounds__(128) void decode_embed_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ emb,
                                                           int64_t ld_emb, int64_t vocab, const float* __restrict__ pos,
                                                           int64_t ld_pos, int64_t n_pos, int64_t* __restrict__ seq,
                                                           int64_t pad_idx, float mask_value, float* __restrict__ mask,
                                                           int64_t ld_mask, int64_t col, float* __restrict__ x32,
                                                           T* __restrict__ x, int D) {
  const int r = blockIdx.x;
  int64_t tok = tokens[r];
  const int64_t s = seq[r] + 1;  // decoders.py:61-63: running_seq.add_(1); seq = running_seq
  __syncthreads();               // every thread has read seq[r] before thread 0 overwrites it
  if (threadIdx.x == 0) {
    seq[r] = s;
    if (mask) mask[(int64_t)r * ld_mask + col] = tok == pad_idx ? mask_value : 0.f;
  }
  tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
  const int64_t sp = s < 0 ? 0 : (s >= n_pos ? n_pos - 1 : s);
  const float* e = emb + tok * ld_emb;
  const float* p = pos + sp * ld_pos;
  for (int f = threadIdx.x * 4; f < D; f += 128 * 4) {
    const float4 a = *reinterpret_cast<const float4*>(e + f), b = *reinterpret_cast<const float4*>(p + f);
    const float4 v = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    if (x32) *reinterpret_cast<float4*>(x32 + (int64_t)r * D + f) = v;
    if (x) {
      T* o = x + (int64_t)r * D + f;
      o[0] = from_f32<T>(v.x); o[1] = from_f32<T>(v.y); o[2] = from_f32<T>(v.z); o[3] = from_f32<T>(v.w);
    }
  }
}

// One wave per row of logits.  The row is read ONCE with 16-byte loads, NCH of them in flight per lane, and kept in
// registers (rows of up to 64 * NCH * (16 / sizeof(T)) words: 4096 bf16 logits; longer rows stream through twice):
// (max, sum of exponentials) -> log-sum-exp, then the candidate score of every word and the per-lane best K (sorted
// insertion), then K wave-wide arg-max pops (ties: the smaller word index).
template <typename T, int K>
__global__ __launch_bounds__(256) void beam_candidates_kernel(const T* __restrict__ logits, int64_t ld, int R, int V,
                                                              const float* __restrict__ seq_logprob,
                                                              float* __restrict__ seq_mask,
                                                              const int64_t* __restrict__ prev_words, int64_t eos,
                                                              float* __restrict__ vals, int64_t* __restrict__ idx,
                                                              float* __restrict__ wl) {
  constexpr int VEC = 16 / (int)sizeof(T), NCH = 8;
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const T* xr = logits + (int64_t)row * ld;
  const bool vec = (((uintptr_t)xr | (uintptr_t)(ld * sizeof(T))) & 15) == 0;
  const int NV = vec ? V / VEC : 0;               // whole 16-byte chunks of the row
  const bool cached = NV <= 64 * NCH;             // ... all of them fit in this wave's registers
  uint4 q[NCH];
#pragma unroll
  for (int u = 0; u < NCH; u++) q[u] = make_uint4(0u, 0u, 0u, 0u);
  if (cached && NV > 0) {
#pragma unroll
    for (int u = 0; u < NCH; u++) q[u] = *reinterpret_cast<const uint4*>(xr + (int64_t)VEC * min(lane + 64 * u, NV - 1));
  }
  auto unpack = [&](const uint4& w, float (&f)[VEC]) {
    if constexpr (sizeof(T) == 4) {
      f[0] = __uint_as_float(w.x); f[1] = __uint_as_float(w.y); f[2] = __uint_as_float(w.z); f[3] = __uint_as_float(w.w);
    } else {  // bf16 -> fp32: the 16 bits are the high half
      const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int e = 0; e < 4; e++) {
        f[2 * e] = __uint_as_float(ws[e] << 16);
        f[2 * e + 1] = __uint_as_float(ws[e] & 0xffff0000u);
      }
    }
  };
  float m = -INFINITY, s = 0.f;
  auto fold = [&](float v) {  // online log-sum-exp; a -inf logit (a masked word) contributes 0, also as the first
    const float nm = fmaxf(m, v);  // value of a lane: exp(-inf - -inf) would be NaN (F.log_softmax handles -inf, so do we)
    s = (m == -INFINITY ? 0.f : s * __expf(m - nm)) + (v == -INFINITY ? 0.f : __expf(v - nm));
    m = nm;
  };
  if (cached) {
#pragma unroll
    for (int u = 0; u < NCH; u++) {
      if (lane + 64 * u < NV) {
        float f[VEC];
        unpack(q[u], f);
#pragma unroll
        for (int e = 0; e < VEC; e++) fold(f[e]);
      }
    }
  } else {
    for (int c = lane; c < NV; c += 64) {
      float f[VEC];
      unpack(*reinterpret_cast<const uint4*>(xr + (int64_t)VEC * c), f);
#pragma unroll
      for (int e = 0; e < VEC; e++) fold(f[e]);
    }
  }
  for (int j = NV * VEC + lane; j < V; j += 64) fold(to_f32<T>(xr[j]));
  const float M = wave_max(m);
  const float S = wave_sum(m == -INFINITY ? 0.f : s * __expf(m - M));
  const float logS = __logf(S);
  const float sl = seq_logprob[row];
  float alive = seq_mask[row];
  if (prev_words) {  // beam_search.py:49-51: a sequence that emitted <eos> stops collecting log-probabilities
    alive *= prev_words[row] != eos ? 1.f : 0.f;
    if (lane == 0) seq_mask[row] = alive;
  }
  const bool live = alive != 0.f;
  float bv[K];
  int bi[K];
#pragma unroll
  for (int t = 0; t < K; t++) { bv[t] = -INFINITY; bi[t] = 0x7fffffff; }
  auto offer = [&](float x, int j) {
    const float lp = (x - M) - logS;
    // beam_search.py:46,52-55: seq_logprob + word_logprob for a live sequence; a finished one keeps its score on word 0
    float v = live ? sl + lp : (j == 0 ? sl : -999.f);
    int i = j;
    if (v > bv[K - 1] || bi[K - 1] == 0x7fffffff) {  // (a lane meets its words in increasing order: ties keep the first)
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
  if (cached) {
#pragma unroll
    for (int u = 0; u < NCH; u++) {
      const int c = lane + 64 * u;
      if (c < NV) {
        float f[VEC];
        unpack(q[u], f);
#pragma unroll
        for (int e = 0; e < VEC; e++) offer(f[e], c * VEC + e);
      }
    }
  } else {
    for (int c = lane; c < NV; c += 64) {
      float f[VEC];
      unpack(*reinterpret_cast<const uint4*>(xr + (int64_t)VEC * c), f);
#pragma unroll
      for (int e = 0; e < VEC; e++) offer(f[e], c * VEC + e);
    }
  }
  for (int j = NV * VEC + lane; j < V; j += 64) offer(to_f32<T>(xr[j]), j);
#pragma unroll
  for (int t = 0; t < K; t++) {
    const float best = wave_max(bv[0]);
    int cand = bv[0] == best ? bi[0] : 0x7fffffff;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) cand = min(cand, __shfl_xor(cand, off, 64));
    if (lane == 0) {
      vals[(int64_t)row * K + t] = best;
      idx[(int64_t)row * K + t] = cand;
      // the (masked) log-probability of that word: what beam_search.py:66 gathers into the history
      wl[(int64_t)row * K + t] = cand < V ? ((to_f32<T>(xr[cand]) - M) - logS) * alive : 0.f;
    }
    if (bv[0] == best && bi[0] == cand) {
#pragma unroll
      for (int u = 0; u < K - 1; u++) { bv[u] = bv[u + 1]; bi[u] = bi[u + 1]; }
      bv[K - 1] = -INFINITY;
      bi[K - 1] = 0x7fffffff;
    }
  }
}

// One wave per sample: lane i holds survivor i of the cur * k (<= 64); `beam` arg-max pops in flat-index order on ties.
__global__ __launch_bounds__(64) void beam_commit_kernel(ovqa::BeamCommitArgs a) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int nc = a.cur * a.k;
  float v = -INFINITY;
  if (lane < nc) v = a.vals[(int64_t)b * nc + lane];
  bool taken = lane >= nc;
  __shared__ int sel_lane[8];
  for (int j = 0; j < a.beam; j++) {
    const float mine = taken ? -INFINITY : v;
    const float best = wave_max(mine);
    int cand = (!taken && mine == best) ? lane : 0x7fffffff;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) cand = min(cand, __shfl_xor(cand, off, 64));
    if (cand == 0x7fffffff) cand = 0;  // fewer finite survivors than beams (cannot happen for k >= 1; stay in range)
    if (lane == cand) taken = true;
    if (lane == 0) sel_lane[j] = cand;
  }
  __syncthreads();
  if (lane < a.beam) {
    const int c = sel_lane[lane], from = c / a.k;
    const int64_t src = (int64_t)b * nc + c, dst = (int64_t)b * a.beam + lane;
    const int64_t word = a.idx[src];
    a.seq_logprob_out[dst] = a.vals[src];
    a.seq_mask_out[dst] = a.seq_mask_in[(int64_t)b * a.cur + from];
    a.selected_beam[dst] = from;
    a.words[dst] = word;
    a.out_out[dst * a.T + a.t] = word;
    a.lp_out[dst * a.T + a.t] = a.wl[src];
  }
  // histories follow their beams (beam_search.py:67-72): columns < t of the chosen source rows
  for (int e = lane; e < a.beam * a.t; e += 64) {
    const int j = e / a.t, c = e - j * a.t;
    const int from = sel_lane[j] / a.k;
    const int64_t s = ((int64_t)b * a.cur + from) * a.T + c, d = ((int64_t)b * a.beam + j) * a.T + c;
    a.out_out[d] = a.out_in[s];
    a.lp_out[d] = a.lp_in[s];
  }
}

}  // namespace

namespace ovqa {

int decode_embed(int out_dtype, const int64_t* tokens, const float* emb, int64_t ld_emb, int64_t vocab, const float* pos,
                 int64_t ld_pos, int64_t n_pos, int64_t* seq, int64_t pad_idx, float mask_value, float* mask,
                 int64_t ld_mask, int64_t col, float* x32, void* x, int64_t R, int64_t D, hipStream_t st) {
  if (R == 0) return OVQA_OK;
  if (out_dtype == OVQA_BF16)
    hipLaunchKernelGGL((decode_embed_kernel<bf16>), dim3((unsigned)R), dim3(128), 0, st, tokens, emb, ld_emb, vocab, pos,
                       ld_pos, n_pos, seq, pad_idx, mask_value, mask, ld_mask, col, x32, (bf16*)x, (int)D);
  else
    hipLaunchKernelGGL((decode_embed_kernel<float>), dim3((unsigned)R), dim3(128), 0, st, tokens, emb, ld_emb, vocab, pos,
                       ld_pos, n_pos, seq, pad_idx, mask_value, mask, ld_mask, col, x32, (float*)x, (int)D);
  return ovqa_check_launch("decode_embed");
}

template <typename T>
static int launch_candidates(const void* logits, int64_t ld, int64_t R, int64_t V, int k, const float* seq_logprob,
                             float* seq_mask, const int64_t* prev_words, int64_t eos, float* vals, int64_t* idx, float* wl,
                             hipStream_t st) {
  const dim3 grid((unsigned)((R + 3) / 4)), block(256);
#define OVQA_BC(KV)                                                                                              \
  hipLaunchKernelGGL((beam_candidates_kernel<T, KV>), grid, block, 0, st, (const T*)logits, ld, (int)R, (int)V, \
                     seq_logprob, seq_mask, prev_words, eos, vals, idx, wl)
  switch (k) {
    case 1: OVQA_BC(1); break;
    case 2: OVQA_BC(2); break;
    case 3: OVQA_BC(3); break;
    case 4: OVQA_BC(4); break;
    case 5: OVQA_BC(5); break;
    case 6: OVQA_BC(6); break;
    case 7: OVQA_BC(7); break;
    default: OVQA_BC(8);
  }
#undef OVQA_BC
  return ovqa_check_launch("beam_candidates");
}

int beam_candidates(int dtype, const void* logits, int64_t ld, int64_t R, int64_t V, int k, const float* seq_logprob,
                    float* seq_mask, const int64_t* prev_words, int64_t eos, float* vals, int64_t* idx, float* wl,
                    hipStream_t st) {
  if (R == 0) return OVQA_OK;
  if (dtype == OVQA_BF16)
    return launch_candidates<bf16>(logits, ld, R, V, k, seq_logprob, seq_mask, prev_words, eos, vals, idx, wl, st);
  return launch_candidates<float>(logits, ld, R, V, k, seq_logprob, seq_mask, prev_words, eos, vals, idx, wl, st);
}

int beam_commit(const BeamCommitArgs& a, int64_t b_s, hipStream_t st) {
  if (b_s == 0) return OVQA_OK;
  hipLaunchKernelGGL(beam_commit_kernel, dim3((unsigned)b_s), dim3(64), 0, st, a);
  return ovqa_check_launch("beam_commit");
}

}  // namespace ovqa
