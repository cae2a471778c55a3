// The two ends of the model around the encoder stacks: token-embedding gather / scatter,
// elementwise dropout, the attention-pooling head of MCAN / CrossModalityTransformer (models/mcan.py:12-25,70-76) and
// log_softmax + NLLLoss (mcan.py:81, tasks/classification_task.py:125-127).  Small, HBM / latency-bound kernels: each
// replaces a chain of 3-8 stock elementwise / reduction launches (and their autograd twins) by one launch.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace ovqa {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- embedding rows ----------------------------------------------------------------------------------------------------
// out[r][0 .. width) = table[tokens(r)][0 .. width), r = t * B + b (time-major) or b * T + t; `width` includes the zero
// padding of a ragged table (runtime._footprint), so the rows come out 16-byte aligned and zero-padded.  One wave per row.
template <typename T>
__global__ __launch_bounds__(256) void embed_gather_kernel(const int64_t* __restrict__ tokens, const T* __restrict__ table,
                                                           int64_t ld_table, int64_t vocab, T* __restrict__ out,
                                                           int64_t ld_out, int B, int Tn, int width, int time_major,
                                                           float* __restrict__ mask, int64_t padding_idx) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
  if (r >= B * Tn) return;
  const int b = time_major ? r % B : r / Tn, t = time_major ? r / B : r % Tn;
  int64_t tok = tokens[(int64_t)b * Tn + t];
  // generate_padding_mask on token ids (models/utils.py:44-58): (tok == pad) * -10e4, i.e. -1e5 or -0.0, fp32 [B,1,1,T]
  if (mask != nullptr && l == 0) mask[(int64_t)b * Tn + t] = tok == padding_idx ? -100000.f : -0.f;
  tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);  // (torch raises on an out-of-range index; never read outside)
  constexpr int V = 16 / (int)sizeof(T);
  const T* src = table + tok * ld_table;
  T* dst = out + (int64_t)r * ld_out;
  for (int c = l * V; c < width; c += 64 * V) *reinterpret_cast<u32x4*>(dst + c) = *reinterpret_cast<const u32x4*>(src + c);
}

// The inputs of a teacher-forced decoder pass from the answer tokens (decoders.py:50-60,66), one wave per position (b, t):
//   out[b][t][:]       = emb[b][t][:] + pos_table[seq][:],   seq = tokens[b][t] == padding_idx ? 0 : t + 1
//   self_mask[b][t][j] = (tokens[b][j] == padding_idx || j > t) * -10e4      (generate_self_attention_masks of the padding and
//                        the causal mask, models/utils.py:59-73; "no mask" is -0.0 as there: long 0 * -10e4)
// -- fourteen stock elementwise / reduce / gather launches of ~5 us in the decoder_train step before (profiles/README.md).
__global__ __launch_bounds__(256) void decoder_inputs_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ emb,
                                                             const float* __restrict__ pos_table, float* __restrict__ out,
                                                             float* __restrict__ self_mask, int R, int Tn, int D,
                                                             int64_t padding_idx) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
  if (r >= R) return;
  const int b = r / Tn, t = r % Tn;
  const int64_t* trow = tokens + (int64_t)b * Tn;
  const int seq = trow[t] == padding_idx ? 0 : t + 1;
  const float* e = emb + (int64_t)r * D;
  const float* p = pos_table + (int64_t)seq * D;
  float* o = out + (int64_t)r * D;
  for (int c = l * 4; c < D; c += 256) {
    const float4 a = *reinterpret_cast<const float4*>(e + c), q = *reinterpret_cast<const float4*>(p + c);
    *reinterpret_cast<float4*>(o + c) = make_float4(a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w);
  }
  float* m = self_mask + (int64_t)r * Tn;
  for (int j = l; j < Tn; j += 64) m[j] = (trow[j] == padding_idx || j > t) ? -100000.f : -0.f;
}

// dtable[v][0 .. width) (=|+=) sum over the positions with token v of their rows of drows, in a fixed order (the order the
// tokens lie in memory: ANY fixed order makes the sum deterministic, and this one needs an integer division per MATCH only),
// no atomics (torch's embedding_dense_backward adds atomically).  One wave per TABLE row, including the rows no token names:
// they are stored as zeros, so the gradient buffer needs no memset; row `padding_idx` gets zeros (nn.Embedding).  The four
// waves of a workgroup share the token list through LDS, 2048 positions per round of loads.
// MEASURED (1280 positions, 4000 x 320 table, in the model step): walking the list from global memory, one dependent load per
// 64 positions, 13.8 us; four table rows per wave 18.4; a wave per POSITION (owner = first occurrence) + a memset node for
// the untouched rows 10.6-12.8 + 4.9; this form: profiles/README.md.
// A token many positions share (<bos> of every sample: 64 hits in decoder_train) is ONE wave's chain of row reads: rows are read
// 16 bytes per lane (VEC: rows 16-byte aligned, width % (16 / sizeof(T)) == 0) and four hits' loads are issued before the
// first is added -- in position order, so the sums are those of the one-at-a-time form bit for bit.
// MEASURED (decoder_train, 1280 positions, 4000 x 512 table, <bos> x 64): 137 us -> see profiles/README.md.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void embed_scatter_kernel(const int64_t* __restrict__ tokens, const T* __restrict__ drows,
                                                            int64_t ld_rows, float* __restrict__ dtable, int64_t ld_table,
                                                            int64_t rows_table, int B, int Tn, int width, int time_major,
                                                            int64_t padding_idx, int accumulate) {
  constexpr int MAXC = 16;    // columns per lane: width <= 1024
  constexpr int CHUNK = 2048;  // positions staged in LDS per round
  constexpr int V = VEC ? 16 / (int)sizeof(T) : 1;  // consecutive columns per lane and load
  constexpr int NL = MAXC / V;                      // loads per row and lane
  constexpr int HB = 4;                             // hits whose loads are in flight together
  struct alignas(V * sizeof(T)) Vec { T e[V]; };
  __shared__ int64_t s_tok[CHUNK];
  const int R = B * Tn;
  const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int l = threadIdx.x & 63;
  const bool live = v < rows_table && v != padding_idx;
  float acc[MAXC];
#pragma unroll
  for (int j = 0; j < MAXC; j++) acc[j] = 0.f;
  for (int base = 0; base < R; base += CHUNK) {
    __syncthreads();
    for (int i = threadIdx.x; i < CHUNK; i += 256) s_tok[i] = base + i < R ? tokens[base + i] : -1;
    __syncthreads();
    if (!live) continue;  // (wave-uniform; every wave reaches the barriers)
    const int n = min(CHUNK, R - base);
    for (int i0 = 0; i0 < n; i0 += 64) {
      uint64_t hits = __ballot(s_tok[i0 + l] == v);
      while (hits) {
        Vec val[HB][NL];
        int nh = 0;
#pragma unroll
        for (int u = 0; u < HB; u++) {
          if (!hits) break;  // (wave-uniform: a ballot)
          const int k = __ffsll((long long)hits) - 1;
          hits &= hits - 1;
          const int mm = base + i0 + k;
          const int r = time_major ? (mm % Tn) * B + mm / Tn : mm;  // the row of drows that position (b, t) owns
          const T* src = drows + (int64_t)r * ld_rows;
#pragma unroll
          for (int j = 0; j < NL; j++) {
            const int c = (j * 64 + l) * V;
            if (c < width) val[u][j] = *reinterpret_cast<const Vec*>(src + c);
          }
          nh = u + 1;
        }
#pragma unroll
        for (int u = 0; u < HB; u++) {
          if (u >= nh) break;
#pragma unroll
          for (int j = 0; j < NL; j++)
            if ((j * 64 + l) * V < width) {
#pragma unroll
              for (int t = 0; t < V; t++) acc[j * V + t] += to_f32<T>(val[u][j].e[t]);
            }
        }
      }
    }
  }
  if (v >= rows_table) return;
  float* dst = dtable + v * ld_table;
#pragma unroll
  for (int j = 0; j < NL; j++) {
    const int c = (j * 64 + l) * V;
    if (c >= width) continue;
#pragma unroll
    for (int t = 0; t < V; t++) dst[c + t] = accumulate ? dst[c + t] + acc[j * V + t] : acc[j * V + t];
  }
}

// y[i] = x[i] * keep(i) / (1 - p), flat element index i (forward and backward of an nn.Dropout call site)
template <typename T>
__global__ __launch_bounds__(256) void dropout_apply_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n,
                                                            DropArgs da) {
  const DropState ds = drop_init(da);
  constexpr int V = 16 / (int)sizeof(T);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i * V < n; i += stride) {
    if (i * V + V <= n) {
      alignas(16) T xv[V], yv[V];
      *reinterpret_cast<u32x4*>(xv) = *reinterpret_cast<const u32x4*>(x + i * V);
#pragma unroll
      for (int e = 0; e < V; e++) yv[e] = from_f32<T>(to_f32<T>(xv[e]) * drop_mul(ds, (uint32_t)(i * V + e)));
      *reinterpret_cast<u32x4*>(y + i * V) = *reinterpret_cast<const u32x4*>(yv);
    } else {
      for (int64_t e = i * V; e < n; e++) y[e] = from_f32<T>(to_f32<T>(x[e]) * drop_mul(ds, (uint32_t)e));
    }
  }
}

// ---- attention pooling (mcan.py:12-25, 70-76) --------------------------------------------------------------------------
//   logit[b,n] = fc2 . dropout(relu(hpre[b,n,:])) + b2     (hpre = fc1(feat) from the GEMM, bias inside)
//   att[b,:]   = softmax over the N positions of sample b (padded positions included, as the reference)
//   pooled[b,:] = sum_n att[b,n] feat[b,n,:]
// One workgroup (8 waves) per sample; a wave per row for the two row passes, partial pooled sums meet in LDS.
constexpr int POOL_WAVES = 8;
constexpr int POOL_MAXN = 1024;  // positions per sample the LDS arrays cover
constexpr int POOL_MAXD = 1024;

template <typename F>
__device__ __forceinline__ void load8(const F* p, float (&v)[8]);
template <>
__device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <>
__device__ __forceinline__ void load8<bf16>(const bf16* p, float (&v)[8]) {
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int t = 0; t < 8; t++) v[t] = (float)a[t];
}
template <typename F>
__device__ __forceinline__ void store8t(F* p, const float (&v)[8]);
template <>
__device__ __forceinline__ void store8t<float>(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <>
__device__ __forceinline__ void store8t<bf16>(bf16* p, const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int t = 0; t < 8; t++) o[t] = (bf16)v[t];
  *reinterpret_cast<bf16x8*>(p) = o;
}

template <typename F, typename T>
__global__ __launch_bounds__(POOL_WAVES * 64) void pool_fwd_kernel(const F* __restrict__ feat, const T* __restrict__ hpre,
                                                                  const float* __restrict__ w2, const float* __restrict__ b2,
                                                                  float* __restrict__ att, T* __restrict__ pooled,
                                                                  float* __restrict__ pooled32, int N, int D, DropArgs da) {
  __shared__ float s_logit[POOL_MAXN];
  __shared__ float s_part[POOL_WAVES][POOL_MAXD];
  __shared__ float s_red[POOL_WAVES];
  const int b = blockIdx.x, w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const DropState ds = drop_init(da);
  const float bias2 = b2 ? b2[0] : 0.f;
  // pass 1: logits
  for (int n = w; n < N; n += POOL_WAVES) {
    const int64_t row = (int64_t)b * N + n;
    float s = 0.f;
    for (int c = l * 8; c < D; c += 512) {
      float h[8], wv[8], dm[8];
      load8<T>(hpre + row * D + c, h);
      load8<float>(w2 + c, wv);
      drop_mul8(ds, (uint32_t)(row * D + c), dm);
#pragma unroll
      for (int t = 0; t < 8; t++) s = fmaf(fmaxf(h[t], 0.f) * dm[t], wv[t], s);
    }
    s = wave_sum(s);
    if (l == 0) s_logit[n] = s + bias2;
  }
  __syncthreads();
  // softmax over the N logits (fixed order: every thread walks its stripe, waves meet in LDS)
  float mx = -INFINITY;
  for (int n = threadIdx.x; n < N; n += POOL_WAVES * 64) mx = fmaxf(mx, s_logit[n]);
  mx = wave_max(mx);
  if (l == 0) s_red[w] = mx;
  __syncthreads();
  mx = s_red[0];
#pragma unroll
  for (int i = 1; i < POOL_WAVES; i++) mx = fmaxf(mx, s_red[i]);
  __syncthreads();
  float sm = 0.f;
  for (int n = threadIdx.x; n < N; n += POOL_WAVES * 64) {
    const float e = __expf(s_logit[n] - mx);
    s_logit[n] = e;
    sm += e;
  }
  sm = wave_sum(sm);
  if (l == 0) s_red[w] = sm;
  __syncthreads();
  sm = 0.f;
#pragma unroll
  for (int i = 0; i < POOL_WAVES; i++) sm += s_red[i];
  const float inv = 1.f / sm;
  for (int n = threadIdx.x; n < N; n += POOL_WAVES * 64) {
    const float a = s_logit[n] * inv;
    s_logit[n] = a;
    att[(int64_t)b * N + n] = a;
  }
  __syncthreads();
  // pass 2: pooled = sum_n att[n] feat[n]: wave w takes rows w, w + 8, ...; lane l owns columns l*8 + 512*j
  for (int c0 = 0; c0 < D; c0 += 512) {
    const int c = c0 + l * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < D) {
      for (int n = w; n < N; n += POOL_WAVES) {
        float f[8];
        load8<F>(feat + ((int64_t)b * N + n) * D + c, f);
        const float a = s_logit[n];
#pragma unroll
        for (int t = 0; t < 8; t++) acc[t] = fmaf(a, f[t], acc[t]);
      }
#pragma unroll
      for (int t = 0; t < 8; t++) s_part[w][c + t] = acc[t];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += POOL_WAVES * 64) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < POOL_WAVES; i++) s += s_part[i][c];
    pooled[(int64_t)b * D + c] = from_f32<T>(s);
    if (pooled32) pooled32[(int64_t)b * D + c] = s;
  }
}

// Backward of the pooling of one sample, given dpooled[b,:] (fp32):
//   dfeat_direct[b,n,:] = att[b,n] dpooled[b,:]                    (the addend of fc1's dX product)
//   datt[n] = feat[b,n,:] . dpooled[b,:];  dlogit[n] = att[n] (datt[n] - sum_m att[m] datt[m])
//   dh[b,n,:] = dlogit[n] w2[:] relu'(hpre) keep/(1-p)              (gradient w.r.t. fc1's output)
//   dw2 partial[b][:] = sum_n dlogit[n] dropout(relu(hpre[b,n,:]))   ([B][2*D] rows: the deferred grouped reduce sums them)
//   db2 partial[b][0] = sum_n dlogit[n]  (analytically 0: softmax shift invariance), one [2 * 8]-float row per sample for the
//   same deferred reduce (fc2.bias occupies an 8-element footprint in the arena).
template <typename F, typename T>
__global__ __launch_bounds__(POOL_WAVES * 64) void pool_bwd_kernel(const F* __restrict__ feat, const T* __restrict__ hpre,
                                                                  const float* __restrict__ w2, const float* __restrict__ att,
                                                                  const T* __restrict__ dpooled, T* __restrict__ dh,
                                                                  T* __restrict__ dfeat, float* __restrict__ dw2_part,
                                                                  float* __restrict__ db2_part, int N, int D, DropArgs da) {
  __shared__ float s_a[POOL_MAXN];   // att, then dlogit
  __shared__ float s_d[POOL_MAXN];   // datt
  __shared__ float s_part[POOL_WAVES][POOL_MAXD];
  __shared__ float s_red[POOL_WAVES];
  const int b = blockIdx.x, w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const DropState ds = drop_init(da);
  const T* dp = dpooled + (int64_t)b * D;
  for (int n = threadIdx.x; n < N; n += POOL_WAVES * 64) s_a[n] = att[(int64_t)b * N + n];
  __syncthreads();
  // pass 1: datt[n] and the direct gradient of the features
  for (int n = w; n < N; n += POOL_WAVES) {
    const int64_t row = (int64_t)b * N + n;
    const float a = s_a[n];
    float s = 0.f;
    for (int c = l * 8; c < D; c += 512) {
      float f[8], g[8], o[8];
      load8<F>(feat + row * D + c, f);
      load8<T>(dp + c, g);
#pragma unroll
      for (int t = 0; t < 8; t++) {
        s = fmaf(f[t], g[t], s);
        o[t] = a * g[t];
      }
      store8t<T>(dfeat + row * D + c, o);
    }
    s = wave_sum(s);
    if (l == 0) s_d[n] = s;
  }
  __syncthreads();
  float dot = 0.f;
  for (int n = threadIdx.x; n < N; n += POOL_WAVES * 64) dot += s_a[n] * s_d[n];
  dot = wave_sum(dot);
  if (l == 0) s_red[w] = dot;
  __syncthreads();
  dot = 0.f;
#pragma unroll
  for (int i = 0; i < POOL_WAVES; i++) dot += s_red[i];
  __syncthreads();
  float sdl = 0.f;
  for (int n = threadIdx.x; n < N; n += POOL_WAVES * 64) {
    const float dl = s_a[n] * (s_d[n] - dot);
    s_a[n] = dl;
    sdl += dl;
  }
  sdl = wave_sum(sdl);
  if (l == 0) s_red[w] = sdl;
  __syncthreads();
  // pass 2: dh rows and this sample's dw2 partial
  for (int c0 = 0; c0 < D; c0 += 512) {
    const int c = c0 + l * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < D) {
      float wv[8];
      load8<float>(w2 + c, wv);
      for (int n = w; n < N; n += POOL_WAVES) {
        const int64_t row = (int64_t)b * N + n;
        float h[8], dm[8], o[8];
        load8<T>(hpre + row * D + c, h);
        drop_mul8(ds, (uint32_t)(row * D + c), dm);
        const float dl = s_a[n];
#pragma unroll
        for (int t = 0; t < 8; t++) {
          const float act = fmaxf(h[t], 0.f) * dm[t];
          acc[t] = fmaf(dl, act, acc[t]);
          o[t] = h[t] > 0.f ? dl * wv[t] * dm[t] : 0.f;
        }
        store8t<T>(dh + row * D + c, o);
      }
#pragma unroll
      for (int t = 0; t < 8; t++) s_part[w][c + t] = acc[t];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += POOL_WAVES * 64) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < POOL_WAVES; i++) s += s_part[i][c];
    dw2_part[(int64_t)b * 2 * D + c] = s;
    dw2_part[(int64_t)b * 2 * D + D + c] = 0.f;  // (second half of the reduce row: unused)
  }
  // db2: this sample's sum of dlogit as row b of a [B][2 * 8] partial block (column 0; the rest zeros): the deferred grouped
  // reduce adds the rows in a fixed order into fc2.bias's 8-element footprint in the gradient arena -- no ticket, no memset
  if (db2_part != nullptr && threadIdx.x < 16) {
    float mine = 0.f;
#pragma unroll
    for (int i = 0; i < POOL_WAVES; i++) mine += s_red[i];
    db2_part[(int64_t)b * 16 + threadIdx.x] = threadIdx.x == 0 ? mine : 0.f;
  }
}

// ---- log_softmax over the first n columns of every row ([M, ld] input) -> fp32 [M, n]; one wave per row -------------------
template <typename T>
__global__ __launch_bounds__(256) void log_softmax_fwd_kernel(const T* __restrict__ x, int64_t ld, float* __restrict__ out,
                                                              int M, int n) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
  if (r >= M) return;
  const T* p = x + (int64_t)r * ld;
  float mx = -INFINITY;
  for (int c = l; c < n; c += 64) mx = fmaxf(mx, to_f32<T>(p[c]));
  mx = wave_max(mx);
  float s = 0.f;
  for (int c = l; c < n; c += 64) s += __expf(to_f32<T>(p[c]) - mx);
  s = wave_sum(s);
  const float lse = mx + __logf(s);
  for (int c = l; c < n; c += 64) out[(int64_t)r * n + c] = to_f32<T>(p[c]) - lse;
}

// dlogits[r][c] = g[r][c] - exp(logp[r][c]) sum_c g[r][c] for c < n, 0 for n <= c < ld (the padded columns of a ragged
// classifier); g, logp fp32 [M, n]
template <typename T>
__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float* __restrict__ g, const float* __restrict__ logp,
                                                              T* __restrict__ dx, int64_t ld, int M, int n) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
  if (r >= M) return;
  const float* gp = g + (int64_t)r * n;
  const float* lp = logp + (int64_t)r * n;
  float s = 0.f;
  for (int c = l; c < n; c += 64) s += gp[c];
  s = wave_sum(s);
  T* d = dx + (int64_t)r * ld;
  for (int c = l; c < (int)ld; c += 64) d[c] = from_f32<T>(c < n ? gp[c] - __expf(lp[c]) * s : 0.f);
}

// ---- the same two for WIDE rows (a generation vocabulary: 1280 positions x 4000 words in decoder_train): one 256-thread
// workgroup per row, the row read ONCE in 16-byte pieces and kept in registers over the passes (n <= 8192), 16-byte stores.
// MEASURED (decoder_train): wave-per-row forms 31.7 (fwd) / 44.1 us (bwd) -> profiles/README.md.
constexpr int LSM_THREADS = 256, LSM_MAXN = 8192;
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sm) {  // 4 waves; every thread gets the result
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();  // (sm may still be read from the previous reduction)
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return is_max ? fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])) : (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

template <typename T>
__global__ __launch_bounds__(LSM_THREADS) void log_softmax_fwd_row_kernel(const T* __restrict__ x, int64_t ld,
                                                                          float* __restrict__ out, int n) {
  constexpr int V = 16 / (int)sizeof(T), NV = LSM_MAXN / (LSM_THREADS * V);
  struct alignas(16) Vec { T e[V]; };
  __shared__ float sm[4];
  const T* p = x + (int64_t)blockIdx.x * ld;
  float val[NV][V];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < NV; j++) {
    const int c = (j * LSM_THREADS + (int)threadIdx.x) * V;
    if (c < n) {
      const Vec u = *reinterpret_cast<const Vec*>(p + c);
#pragma unroll
      for (int t = 0; t < V; t++) { val[j][t] = to_f32<T>(u.e[t]); mx = fmaxf(mx, val[j][t]); }
    }
  }
  mx = block_reduce(mx, true, sm);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; j++)
    if ((j * LSM_THREADS + (int)threadIdx.x) * V < n) {
#pragma unroll
      for (int t = 0; t < V; t++) s += __expf(val[j][t] - mx);
    }
  s = block_reduce(s, false, sm);
  const float lse = mx + __logf(s);
  float* o = out + (int64_t)blockIdx.x * n;
#pragma unroll
  for (int j = 0; j < NV; j++) {
    const int c = (j * LSM_THREADS + (int)threadIdx.x) * V;
    if (c < n) {
#pragma unroll
      for (int t = 0; t < V; t += 4)
        *reinterpret_cast<float4*>(o + c + t) = make_float4(val[j][t] - lse, val[j][t + 1] - lse, val[j][t + 2] - lse, val[j][t + 3] - lse);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(LSM_THREADS) void log_softmax_bwd_row_kernel(const float* __restrict__ g, const float* __restrict__ logp,
                                                                          T* __restrict__ dx, int64_t ld, int n) {
  constexpr int V = 16 / (int)sizeof(T), NV = LSM_MAXN / (LSM_THREADS * V);
  struct alignas(16) Vec { T e[V]; };
  __shared__ float sm[4];
  const float* gp = g + (int64_t)blockIdx.x * n;
  const float* lp = logp + (int64_t)blockIdx.x * n;
  float gv[NV][V], lv[NV][V];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; j++) {
    const int c = (j * LSM_THREADS + (int)threadIdx.x) * V;
    if (c < n) {
#pragma unroll
      for (int t = 0; t < V; t += 4) {
        const float4 a = *reinterpret_cast<const float4*>(gp + c + t), b = *reinterpret_cast<const float4*>(lp + c + t);
        gv[j][t] = a.x; gv[j][t + 1] = a.y; gv[j][t + 2] = a.z; gv[j][t + 3] = a.w;
        lv[j][t] = b.x; lv[j][t + 1] = b.y; lv[j][t + 2] = b.z; lv[j][t + 3] = b.w;
        s += (a.x + a.y) + (a.z + a.w);
      }
    }
  }
  s = block_reduce(s, false, sm);
  T* d = dx + (int64_t)blockIdx.x * ld;
#pragma unroll
  for (int j = 0; j < NV; j++) {
    const int c = (j * LSM_THREADS + (int)threadIdx.x) * V;
    if (c < (int)ld) {  // (n <= c < ld: the padded columns of a ragged classifier get zeros)
      Vec u;
#pragma unroll
      for (int t = 0; t < V; t++) u.e[t] = from_f32<T>(c < n ? gv[j][t] - __expf(lv[j][t]) * s : 0.f);
      *reinterpret_cast<Vec*>(d + c) = u;
    }
  }
}

// NLLLoss(reduction = mean, ignore_index) on log-probabilities fp32 [M, n]: loss = -sum_{t_r != ignore} logp[r][t_r] / cnt
// and (optionally) its gradient dlogp (dense, -scale / cnt at the targets).  Every workgroup sums the M targets itself, in the
// same fixed order (M int64 reads: nothing beside a dense gradient of M x n floats), workgroup 0 stores the loss, and each
// workgroup (4 waves) writes the gradient rows [blockIdx.x * rows_per_wg, ...): a wave per row, 16-byte stores.  (Round 5 wrote
// the whole gradient from ONE workgroup with a division per element: 1.2 ms of a 2.6-ms teacher-forced decoder step at
// 1280 positions x 4000 words, profiles/r06a_decoder_train_kernel_stats_before.csv; 16 rows per 1024-thread workgroup left 80
// workgroups for those 20 MB: 21 us.)
__global__ __launch_bounds__(256) void nll_loss_kernel(const float* __restrict__ logp, const int64_t* __restrict__ target,
                                                       float* __restrict__ loss, float* __restrict__ dlogp,
                                                       const float* __restrict__ gscale, int M, int n, int64_t ignore_index,
                                                       int accumulate, int rows_per_wg) {
  __shared__ float s_sum[4];
  __shared__ float s_cnt[4];
  float s = 0.f, cnt = 0.f;
  // eight targets per thread and round: their loads, then the eight dependent loads of logp, are in flight together (one
  // at a time this chain was 10 of the kernel's 17 us at M = 1280); only workgroup 0 stores the loss and needs the sum
  const bool need_sum = blockIdx.x == 0 && loss != nullptr;
  for (int r0 = threadIdx.x; r0 < M; r0 += 256 * 8) {
    int64_t t[8];
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int r = r0 + u * 256;
      t[u] = r < M ? target[r] : ignore_index;
      if (t[u] < 0 || t[u] >= n) t[u] = ignore_index;
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
      v[u] = (need_sum && t[u] != ignore_index) ? logp[(int64_t)(r0 + u * 256) * n + t[u]] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (t[u] != ignore_index) {
        s -= v[u];
        cnt += 1.f;
      }
  }
  s = wave_sum(s);
  cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = s; s_cnt[threadIdx.x >> 6] = cnt; }
  __syncthreads();
  const float ts = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]), tc = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
  const float inv = tc > 0.f ? 1.f / tc : 0.f;  // (torch returns nan for an all-ignored batch; the gradient is 0 either way)
  if (blockIdx.x == 0 && threadIdx.x == 0 && loss) {
    const float v = tc > 0.f ? ts * inv : __int_as_float(0x7fc00000);
    *loss = accumulate ? *loss + v : v;
  }
  if (dlogp == nullptr) return;
  const float sc = (gscale ? gscale[0] : 1.f) * inv;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r_end = min(M, ((int)blockIdx.x + 1) * rows_per_wg);
  const bool vec = (n & 3) == 0 && ((uintptr_t)dlogp & 15) == 0;
  for (int r = (int)blockIdx.x * rows_per_wg + wave; r < r_end; r += 4) {
    const int64_t t = target[r];
    const int hot = (t != ignore_index && t >= 0 && t < n) ? (int)t : -1;
    float* row = dlogp + (int64_t)r * n;
    if (vec) {
      for (int c = lane * 4; c < n; c += 256) {
        const int d = hot - c;  // (hot = -1: never 0..3 for c >= 0)
        *reinterpret_cast<float4*>(row + c) =
            make_float4(d == 0 ? -sc : 0.f, d == 1 ? -sc : 0.f, d == 2 ? -sc : 0.f, d == 3 ? -sc : 0.f);
      }
    } else {
      for (int c = lane; c < n; c += 64) row[c] = c == hot ? -sc : 0.f;
    }
  }
}

}  // namespace

int embed_gather(int dtype, const int64_t* tokens, const void* table, int64_t ld_table, int64_t vocab, void* out,
                 int64_t ld_out, int64_t B, int64_t T, int64_t width, int time_major, float* mask, int64_t padding_idx,
                 hipStream_t st) {
  const int es = dtype == OVQA_BF16 ? 2 : 4;
  OVQA_REQUIRE(width * es % 16 == 0 && ld_table * es % 16 == 0 && ld_out * es % 16 == 0 && (uintptr_t)table % 16 == 0 &&
                   (uintptr_t)out % 16 == 0,
               OVQA_ERR_BAD_ARG, "embed_gather: rows must be 16-byte aligned and a multiple of 16 bytes wide");
  const unsigned grid = (unsigned)((B * T + 3) / 4);
  if (dtype == OVQA_BF16)
    hipLaunchKernelGGL(embed_gather_kernel<bf16>, dim3(grid), dim3(256), 0, st, tokens, (const bf16*)table, ld_table, vocab,
                       (bf16*)out, ld_out, (int)B, (int)T, (int)width, time_major, mask, padding_idx);
  else
    hipLaunchKernelGGL(embed_gather_kernel<float>, dim3(grid), dim3(256), 0, st, tokens, (const float*)table, ld_table, vocab,
                       (float*)out, ld_out, (int)B, (int)T, (int)width, time_major, mask, padding_idx);
  return ovqa_check_launch("embed_gather");
}

int decoder_inputs(const int64_t* tokens, const float* emb, const float* pos_table, float* out, float* self_mask, int64_t B,
                   int64_t T, int64_t D, int64_t padding_idx, hipStream_t st) {
  OVQA_REQUIRE(D % 4 == 0 && (uintptr_t)emb % 16 == 0 && (uintptr_t)pos_table % 16 == 0 && (uintptr_t)out % 16 == 0,
               OVQA_ERR_BAD_ARG, "decoder_inputs: rows must be 16-byte aligned and a multiple of 16 bytes wide");
  hipLaunchKernelGGL(decoder_inputs_kernel, dim3((unsigned)((B * T + 3) / 4)), dim3(256), 0, st, tokens, emb, pos_table, out,
                     self_mask, (int)(B * T), (int)T, (int)D, padding_idx);
  return ovqa_check_launch("decoder_inputs");
}

int embed_scatter(int dtype, const int64_t* tokens, const void* drows, int64_t ld_rows, float* dtable, int64_t ld_table,
                  int64_t rows_table, int64_t B, int64_t T, int64_t width, int time_major, int64_t padding_idx,
                  int accumulate, hipStream_t st) {
  OVQA_REQUIRE(width <= 1024, OVQA_ERR_UNSUPPORTED, "embed_scatter: rows wider than 1024 elements");
  const unsigned grid = (unsigned)((rows_table + 3) / 4);  // one wave per table row
  const int64_t vw = dtype == OVQA_BF16 ? 8 : 4;
  const bool vec = (uintptr_t)drows % 16 == 0 && ld_rows % vw == 0 && width % vw == 0;
#define OVQA_SCATTER(T, VEC)                                                                                              \
  hipLaunchKernelGGL((embed_scatter_kernel<T, VEC>), dim3(grid), dim3(256), 0, st, tokens, (const T*)drows, ld_rows, dtable, \
                     ld_table, rows_table, (int)B, (int)T_, (int)width, time_major, padding_idx, accumulate)
  const int64_t T_ = T;
  if (dtype == OVQA_BF16) { if (vec) OVQA_SCATTER(bf16, true); else OVQA_SCATTER(bf16, false); }
  else { if (vec) OVQA_SCATTER(float, true); else OVQA_SCATTER(float, false); }
#undef OVQA_SCATTER
  return ovqa_check_launch("embed_scatter");
}

int dropout_apply(int dtype, const void* x, void* y, int64_t n, const DropArgs& da, hipStream_t st) {
  OVQA_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0, OVQA_ERR_BAD_ARG, "dropout_apply: 16-byte alignment");
  const int V = dtype == OVQA_BF16 ? 8 : 4;
  int64_t blocks = ((n + V - 1) / V + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  if (dtype == OVQA_BF16)
    hipLaunchKernelGGL(dropout_apply_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16*)x, (bf16*)y, n, da);
  else
    hipLaunchKernelGGL(dropout_apply_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, (float*)y, n,
                       da);
  return ovqa_check_launch("dropout_apply");
}

static bool pool_shape_ok(int64_t N, int64_t D) { return N >= 1 && N <= POOL_MAXN && D >= 8 && D <= POOL_MAXD && D % 8 == 0; }

int pool_fwd(int feat_dtype, int dtype, const void* feat, const void* hpre, const float* w2, const float* b2, float* att,
             void* pooled, float* pooled32, int64_t B, int64_t N, int64_t D, const DropArgs& da, hipStream_t st) {
  OVQA_REQUIRE(pool_shape_ok(N, D), OVQA_ERR_UNSUPPORTED, "pool_fwd: N <= 1024, D <= 1024 and D %% 8 == 0 required");
  OVQA_REQUIRE((uintptr_t)feat % 16 == 0 && (uintptr_t)hpre % 16 == 0 && (uintptr_t)w2 % 16 == 0, OVQA_ERR_BAD_ARG,
               "pool_fwd: 16-byte alignment");
  const dim3 grid((unsigned)B), block(POOL_WAVES * 64);
#define OVQA_POOL_FWD(F, T)                                                                                             \
  hipLaunchKernelGGL((pool_fwd_kernel<F, T>), grid, block, 0, st, (const F*)feat, (const T*)hpre, w2, b2, att, (T*)pooled, \
                     pooled32, (int)N, (int)D, da)
  if (dtype == OVQA_BF16 && feat_dtype == OVQA_BF16) OVQA_POOL_FWD(bf16, bf16);
  else if (dtype == OVQA_BF16) OVQA_POOL_FWD(float, bf16);
  else if (feat_dtype == OVQA_F32) OVQA_POOL_FWD(float, float);
  else { ovqa_set_error("pool_fwd: bf16 features with fp32 hidden activations"); return OVQA_ERR_UNSUPPORTED; }
#undef OVQA_POOL_FWD
  return ovqa_check_launch("pool_fwd");
}

int pool_bwd(int feat_dtype, int dtype, const void* feat, const void* hpre, const float* w2, const float* att,
             const void* dpooled, void* dh, void* dfeat, float* dw2_part, float* db2_part, int64_t B, int64_t N, int64_t D,
             const DropArgs& da, hipStream_t st) {
  OVQA_REQUIRE(pool_shape_ok(N, D), OVQA_ERR_UNSUPPORTED, "pool_bwd: N <= 1024, D <= 1024, D %% 8 == 0");
  const dim3 grid((unsigned)B), block(POOL_WAVES * 64);
#define OVQA_POOL_BWD(F, T)                                                                                              \
  hipLaunchKernelGGL((pool_bwd_kernel<F, T>), grid, block, 0, st, (const F*)feat, (const T*)hpre, w2, att, (const T*)dpooled, (T*)dh, \
                     (T*)dfeat, dw2_part, db2_part, (int)N, (int)D, da)
  if (dtype == OVQA_BF16 && feat_dtype == OVQA_BF16) OVQA_POOL_BWD(bf16, bf16);
  else if (dtype == OVQA_BF16) OVQA_POOL_BWD(float, bf16);
  else if (feat_dtype == OVQA_F32) OVQA_POOL_BWD(float, float);
  else { ovqa_set_error("pool_bwd: bf16 features with fp32 hidden activations"); return OVQA_ERR_UNSUPPORTED; }
#undef OVQA_POOL_BWD
  return ovqa_check_launch("pool_bwd");
}

// wide rows whose pieces are 16-byte aligned: a workgroup per row
static bool lsm_rows(int dtype, const void* x, int64_t ld, const void* f32a, const void* f32b, int64_t n) {
  const int64_t v = dtype == OVQA_BF16 ? 8 : 4;
  return n >= 1024 && n <= LSM_MAXN && ld <= LSM_MAXN && n % v == 0 && ld % v == 0 && (uintptr_t)x % 16 == 0 &&
         (uintptr_t)f32a % 16 == 0 && (uintptr_t)f32b % 16 == 0;
}

int log_softmax_fwd(int dtype, const void* x, int64_t ld, float* out, int64_t M, int64_t n, hipStream_t st) {
  if (lsm_rows(dtype, x, ld, out, nullptr, n)) {
    if (dtype == OVQA_BF16)
      hipLaunchKernelGGL(log_softmax_fwd_row_kernel<bf16>, dim3((unsigned)M), dim3(LSM_THREADS), 0, st, (const bf16*)x, ld, out, (int)n);
    else
      hipLaunchKernelGGL(log_softmax_fwd_row_kernel<float>, dim3((unsigned)M), dim3(LSM_THREADS), 0, st, (const float*)x, ld, out, (int)n);
    return ovqa_check_launch("log_softmax_fwd");
  }
  const unsigned grid = (unsigned)((M + 3) / 4);
  if (dtype == OVQA_BF16)
    hipLaunchKernelGGL(log_softmax_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)x, ld, out, (int)M, (int)n);
  else
    hipLaunchKernelGGL(log_softmax_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, ld, out, (int)M, (int)n);
  return ovqa_check_launch("log_softmax_fwd");
}

int log_softmax_bwd(int dtype, const float* g, const float* logp, void* dx, int64_t ld, int64_t M, int64_t n, hipStream_t st) {
  if (lsm_rows(dtype, dx, ld, g, logp, n)) {
    if (dtype == OVQA_BF16)
      hipLaunchKernelGGL(log_softmax_bwd_row_kernel<bf16>, dim3((unsigned)M), dim3(LSM_THREADS), 0, st, g, logp, (bf16*)dx, ld, (int)n);
    else
      hipLaunchKernelGGL(log_softmax_bwd_row_kernel<float>, dim3((unsigned)M), dim3(LSM_THREADS), 0, st, g, logp, (float*)dx, ld, (int)n);
    return ovqa_check_launch("log_softmax_bwd");
  }
  const unsigned grid = (unsigned)((M + 3) / 4);
  if (dtype == OVQA_BF16)
    hipLaunchKernelGGL(log_softmax_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, st, g, logp, (bf16*)dx, ld, (int)M, (int)n);
  else
    hipLaunchKernelGGL(log_softmax_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, g, logp, (float*)dx, ld, (int)M, (int)n);
  return ovqa_check_launch("log_softmax_bwd");
}

int nll_loss(const float* logp, const int64_t* target, float* loss, float* dlogp, const float* gscale, int64_t M, int64_t n,
             int64_t ignore_index, int accumulate, hipStream_t st) {
  // the gradient rows are dealt to the workgroups 4 at a time (a wave per row), at most 4096 workgroups
  int rows_per_wg = 4;
  while ((M + rows_per_wg - 1) / rows_per_wg > 4096) rows_per_wg *= 2;
  const unsigned grid = dlogp ? (unsigned)std::max<int64_t>(1, (M + rows_per_wg - 1) / rows_per_wg) : 1u;
  hipLaunchKernelGGL(nll_loss_kernel, dim3(grid), dim3(256), 0, st, logp, target, loss, dlogp, gscale, (int)M, (int)n,
                     ignore_index, accumulate, rows_per_wg);
  return ovqa_check_launch("nll_loss");
}

}  // namespace ovqa
