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
#include "common.h"
#include "kernels.h"

namespace {

constexpr int LMAX_KEYS = 1024;  // scores of one wave live in LDS: 4 waves x 4 KiB

template <typename T> struct Vec16;  // 16 bytes of T
template <> struct Vec16<bf16> { typedef bf16x8 type; static constexpr int N = 8; };
template <> struct Vec16<float> { typedef f32x4 type; static constexpr int N = 4; };

template <typename T, int D>
__global__ __launch_bounds__(256) void attn_decode_kernel(ovqa::AttnDecodeArgs a) {
  __shared__ float sc[4][LMAX_KEYS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wid = blockIdx.x * 4 + wave;
  if (wid >= a.R * a.H) return;  // (whole waves: no barrier below)
  const int r = wid / a.H, h = wid - r * a.H;
  const int n = a.n;
  const T* q = (const T*)a.q + (int64_t)r * a.ldq + h * D;
  const int64_t kvrow = (int64_t)(r / a.group) * a.kv_batch_stride;
  const T* kb = (const T*)a.k + kvrow + h * D;
  const T* vb = (const T*)a.v + kvrow + h * D;
  const float* mrow = a.mask ? a.mask + (int64_t)r * a.ldmask : nullptr;
  float* s = sc[wave];
  constexpr int PER = D / 4;                    // features per lane of a key's 4-lane group
  constexpr int NV = Vec16<T>::N, NL = PER / NV;  // 16-byte loads per lane and key
  static_assert(PER % NV == 0, "a lane's share of a key is a whole number of 16-byte loads");
  typedef typename Vec16<T>::type vec_t;
  const int part = lane & 3, kk = lane >> 2;
  float qf[PER];
#pragma unroll
  for (int l = 0; l < NL; l++) {
    const vec_t v = *reinterpret_cast<const vec_t*>(q + part * PER + l * NV);
#pragma unroll
    for (int e = 0; e < NV; e++) qf[l * NV + e] = to_f32<T>(v[e]) * a.scale;
  }
  // ---- scores: 16 keys per pass
  float mx = -INFINITY;
  for (int j0 = 0; j0 < n; j0 += 16) {
    const int j = j0 + kk;
    float dot = 0.f;
    if (j < n) {
      const T* kr = kb + (int64_t)j * a.ldk + part * PER;
#pragma unroll
      for (int l = 0; l < NL; l++) {
        const vec_t v = *reinterpret_cast<const vec_t*>(kr + l * NV);
#pragma unroll
        for (int e = 0; e < NV; e++) dot = fmaf(qf[l * NV + e], to_f32<T>(v[e]), dot);
      }
    }
    dot += __shfl_xor(dot, 1, 64);
    dot += __shfl_xor(dot, 2, 64);
    if (j < n) {
      const float sv = dot + (mrow ? mrow[j] : 0.f);
      if (part == 0) s[j] = sv;
      mx = fmaxf(mx, sv);
    }
  }
  mx = wave_max(mx);
  // ---- softmax numerators (the wave's own LDS row: written and read by this wave only)
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the score stores of every lane have landed
  __builtin_amdgcn_wave_barrier();
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) {
    const float p = __expf(s[j] - mx);
    s[j] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  // ---- o = P V: lane -> feature pair (lane & 31) of a 64-feature slice, key parity (lane >> 5)
  constexpr int FP = D / 2;  // feature pairs
  const int kp = lane >> 5;
#pragma unroll
  for (int f0 = 0; f0 < FP; f0 += 32) {
    const int fp = f0 + (lane & 31);
    float o0 = 0.f, o1 = 0.f;
    if (fp < FP) {
      const T* vr = vb + 2 * fp;
      int j = kp;
      for (; j + 6 < n; j += 8) {  // four keys of this half-wave in flight
        float p[4], x0[4], x1[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const T* pv = vr + (int64_t)(j + 2 * u) * a.ldv;
          x0[u] = to_f32<T>(pv[0]);
          x1[u] = to_f32<T>(pv[1]);
          p[u] = s[j + 2 * u];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          o0 = fmaf(p[u], x0[u], o0);
          o1 = fmaf(p[u], x1[u], o1);
        }
      }
      for (; j < n; j += 2) {
        const T* pv = vr + (int64_t)j * a.ldv;
        const float p = s[j];
        o0 = fmaf(p, to_f32<T>(pv[0]), o0);
        o1 = fmaf(p, to_f32<T>(pv[1]), o1);
      }
    }
    o0 += __shfl_xor(o0, 32, 64);
    o1 += __shfl_xor(o1, 32, 64);
    if (kp == 0 && fp < FP) {
      T* orow = (T*)a.o + (int64_t)r * a.ldo + h * D + 2 * fp;
      orow[0] = from_f32<T>(o0 * inv);
      orow[1] = from_f32<T>(o1 * inv);
    }
  }
}

}  // namespace

namespace ovqa {

bool attention_decode_supported(const AttnDecodeArgs& a, int esize) {
  const auto al = [&](const void* p) { return ((uintptr_t)p & 15) == 0; };
  return (a.d == 64 || a.d == 32 || a.d == 128) && a.n >= 1 && a.n <= LMAX_KEYS && a.group >= 1 && al(a.q) && al(a.k) &&
         al(a.v) && (a.ldq * esize) % 16 == 0 && (a.ldk * esize) % 16 == 0 && (a.ldv * esize) % 4 == 0 &&
         (a.kv_batch_stride * esize) % 16 == 0 && (a.ldo * esize) % 4 == 0 && ((uintptr_t)a.o & 3) == 0;
}

int attention_decode(int dtype, const AttnDecodeArgs& a, hipStream_t st) {
  if (a.R == 0 || a.H == 0) return OVQA_OK;
  const dim3 grid((unsigned)(((int64_t)a.R * a.H + 3) / 4)), block(256);
#define OVQA_DEC(TT, DD) hipLaunchKernelGGL((attn_decode_kernel<TT, DD>), grid, block, 0, st, a)
  if (dtype == OVQA_BF16) {
    switch (a.d) {
      case 32: OVQA_DEC(bf16, 32); break;
      case 64: OVQA_DEC(bf16, 64); break;
      default: OVQA_DEC(bf16, 128);
    }
  } else {
    switch (a.d) {
      case 32: OVQA_DEC(float, 32); break;
      case 64: OVQA_DEC(float, 64); break;
      default: OVQA_DEC(float, 128);
    }
  }
#undef OVQA_DEC
  return ovqa_check_launch("attention_decode");
}

}  // namespace ovqa
