// Reference-grade attention core for gfx950 (exact fp32 math, any dtype storage).
//
// softmax(q k^T * scale + mask) v per (batch, head) with the head split/merge
// folded into addressing (element (b, n, h, c) lives at [(b*N + n)*ld + h*d + c]).
// K and V (forward, dQ pass) or Q and dO (dK/dV pass) of one (b, h) stay
// resident in LDS as fp32; one wavefront owns one query (or key) row at a time,
// lanes stride over the opposite axis, row reductions are wave shuffles.
// This is the OVQA_F32 path and the on-device cross-check of the MFMA kernel.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int ROWS_PER_BLOCK = 32;  // query (or key) rows owned by one workgroup

__device__ __forceinline__ float mask_at(const float* mask, int64_t msb, int64_t msh, int64_t msq, int b, int h,
                                         int i, int j) {
  return mask ? mask[(int64_t)b * msb + (int64_t)h * msh + (int64_t)i * msq + j] : 0.f;
}

// ---------------------------------------------------------------- forward
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_simple_kernel(ovqa::AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nk = a.nk, dk = a.dk, dv = a.dv, nq = a.nq;
  float* Ks = smem;                        // [nk][dk+1]
  float* Vs = Ks + (size_t)nk * (dk + 1);  // [nk][dv+1]
  float* sbuf = Vs + (size_t)nk * (dv + 1);  // [4][nk]
  float* qbuf = sbuf + 4 * (size_t)nk;       // [4][dk]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const T* q = (const T*)a.q;
  const T* k = (const T*)a.k;
  const T* v = (const T*)a.v;
  T* o = (T*)a.o;
  T* att = (T*)a.att;
  const DropState ds = drop_init(a.drop);

  for (int e = tid; e < nk * dk; e += 256) {
    const int j = e / dk, c = e % dk;
    Ks[j * (dk + 1) + c] = to_f32<T>(k[((int64_t)b * nk + j) * a.ldk + h * dk + c]);
  }
  for (int e = tid; e < nk * dv; e += 256) {
    const int j = e / dv, c = e % dv;
    Vs[j * (dv + 1) + c] = to_f32<T>(v[((int64_t)b * nk + j) * a.ldv + h * dv + c]);
  }
  __syncthreads();

  float* sw = sbuf + (size_t)wave * nk;
  float* qw = qbuf + (size_t)wave * dk;
  const int row0 = blockIdx.y * ROWS_PER_BLOCK;
  for (int r = wave; r < ROWS_PER_BLOCK; r += 4) {
    const int i = row0 + r;
    const bool live = i < nq;  // wave-uniform
    if (live)
      for (int c = lane; c < dk; c += 64) qw[c] = to_f32<T>(q[((int64_t)b * nq + i) * a.ldq + h * dk + c]);
    __syncthreads();
    float mx = -INFINITY;
    if (live) {
      for (int j = lane; j < nk; j += 64) {
        const float* kr = Ks + (size_t)j * (dk + 1);
        float s = 0.f;
        for (int c = 0; c < dk; c++) s = fmaf(qw[c], kr[c], s);
        s = s * a.scale + mask_at(a.mask, a.msb, a.msh, a.msq, b, h, i, j);
        sw[j] = s;
        mx = fmaxf(mx, s);
      }
    }
    mx = wave_max(mx);
    float sum = 0.f;
    if (live) {
      for (int j = lane; j < nk; j += 64) {
        const float p = __expf(sw[j] - mx);
        sw[j] = p;
        sum += p;
      }
    }
    sum = wave_sum(sum);
    float inv = 1.f / sum;
    if (live && ds.on) {  // dropout on the probabilities (HF BertSelfAttention): p~ = p * keep / (1 - p_drop)
      const uint32_t base = (uint32_t)((((int64_t)b * a.H + h) * nq + i) * nk);
      for (int j = lane; j < nk; j += 64) sw[j] = sw[j] * inv * drop_mul(ds, base + j);
      inv = 1.f;
    }
    __syncthreads();
    if (live) {
      if (a.lse && lane == 0) a.lse[((int64_t)b * a.H + h) * nq + i] = mx + __logf(sum);
      if (att)
        for (int j = lane; j < nk; j += 64)
          att[(((int64_t)b * a.H + h) * nq + i) * nk + j] = from_f32<T>(sw[j] * inv);
      for (int c = lane; c < dv; c += 64) {
        float acc = 0.f;
        for (int j = 0; j < nk; j++) acc = fmaf(sw[j], Vs[(size_t)j * (dv + 1) + c], acc);
        const T ov = from_f32<T>(acc * inv);
        o[((int64_t)b * nq + i) * a.ldo + h * dv + c] = ov;
        if (a.o_lo)  // rounding residual for the backward's delta (bf16 mode; see kernels.h)
          ((T*)a.o_lo)[((int64_t)b * nq + i) * a.ldo + h * dv + c] = from_f32<T>(acc * inv - to_f32<T>(ov));
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- backward: dQ (+ delta)
// dS = P * (dP - delta), dP = dO V^T (+ d_att when the caller differentiates the returned
// attention weights), delta_i = sum_j P_ij dP_ij  (== dO_i . O_i when d_att is absent).
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_simple_kernel(ovqa::AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nk = a.nk, dk = a.dk, dv = a.dv, nq = a.nq;
  float* Ks = smem;
  float* Vs = Ks + (size_t)nk * (dk + 1);
  float* sbuf = Vs + (size_t)nk * (dv + 1);  // [4][nk]  p, then dS
  float* dpbuf = sbuf + 4 * (size_t)nk;      // [4][nk]  dP
  float* qbuf = dpbuf + 4 * (size_t)nk;      // [4][dk]
  float* dobuf = qbuf + 4 * (size_t)dk;      // [4][dv]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const T* q = (const T*)a.q;
  const T* k = (const T*)a.k;
  const T* v = (const T*)a.v;
  const T* d_o = (const T*)a.d_o;
  const T* datt = (const T*)a.d_att;
  T* dq = (T*)a.dq;
  const DropState ds = drop_init(a.drop);

  for (int e = tid; e < nk * dk; e += 256) {
    const int j = e / dk, c = e % dk;
    Ks[j * (dk + 1) + c] = to_f32<T>(k[((int64_t)b * nk + j) * a.ldk + h * dk + c]);
  }
  for (int e = tid; e < nk * dv; e += 256) {
    const int j = e / dv, c = e % dv;
    Vs[j * (dv + 1) + c] = to_f32<T>(v[((int64_t)b * nk + j) * a.ldv + h * dv + c]);
  }
  __syncthreads();

  float* sw = sbuf + (size_t)wave * nk;
  float* pw = dpbuf + (size_t)wave * nk;
  float* qw = qbuf + (size_t)wave * dk;
  float* dw = dobuf + (size_t)wave * dv;
  const int row0 = blockIdx.y * ROWS_PER_BLOCK;
  for (int r = wave; r < ROWS_PER_BLOCK; r += 4) {
    const int i = row0 + r;
    const bool live = i < nq;
    if (live) {
      for (int c = lane; c < dk; c += 64) qw[c] = to_f32<T>(q[((int64_t)b * nq + i) * a.ldq + h * dk + c]);
      for (int c = lane; c < dv; c += 64) dw[c] = to_f32<T>(d_o[((int64_t)b * nq + i) * a.lddo + h * dv + c]);
    }
    __syncthreads();
    float dsum = 0.f;
    if (live) {
      const float lse = a.lse[((int64_t)b * a.H + h) * nq + i];
      for (int j = lane; j < nk; j += 64) {
        const float* kr = Ks + (size_t)j * (dk + 1);
        const float* vr = Vs + (size_t)j * (dv + 1);
        float s = 0.f, dp = 0.f;
        for (int c = 0; c < dk; c++) s = fmaf(qw[c], kr[c], s);
        for (int c = 0; c < dv; c++) dp = fmaf(dw[c], vr[c], dp);
        if (datt) dp += to_f32<T>(datt[(((int64_t)b * a.H + h) * nq + i) * nk + j]);
        dp *= drop_mul(ds, (uint32_t)((((int64_t)b * a.H + h) * nq + i) * nk + j));  // d p = d p~ * keep/(1-p_drop)
        s = s * a.scale + mask_at(a.mask, a.msb, a.msh, a.msq, b, h, i, j);
        const float p = __expf(s - lse);
        sw[j] = p;
        pw[j] = dp;
        dsum += p * dp;
      }
    }
    // d lse / d S_ij = P_ij: the gradient of the returned log-sum-exp folds into delta (dS = P (dP - delta))
    float delta = wave_sum(dsum);
    if (live && a.d_lse) delta -= a.d_lse[((int64_t)b * a.H + h) * nq + i];
    if (live) {
      if (lane == 0) a.delta[((int64_t)b * a.H + h) * nq + i] = delta;
      for (int j = lane; j < nk; j += 64) sw[j] = sw[j] * (pw[j] - delta);
    }
    __syncthreads();
    if (live) {
      for (int c = lane; c < dk; c += 64) {
        float acc = 0.f;
        for (int j = 0; j < nk; j++) acc = fmaf(sw[j], Ks[(size_t)j * (dk + 1) + c], acc);
        dq[((int64_t)b * nq + i) * a.lddq + h * dk + c] = from_f32<T>(acc * a.scale);
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- backward: dK, dV
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_simple_kernel(ovqa::AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nk = a.nk, dk = a.dk, dv = a.dv, nq = a.nq;
  float* Qs = smem;                          // [nq][dk+1]
  float* Gs = Qs + (size_t)nq * (dk + 1);    // dO [nq][dv+1]
  float* lse_s = Gs + (size_t)nq * (dv + 1); // [nq]
  float* del_s = lse_s + nq;                 // [nq]
  float* pbuf = del_s + nq;                  // [4][nq]
  float* dsbuf = pbuf + 4 * (size_t)nq;      // [4][nq]
  float* kbuf = dsbuf + 4 * (size_t)nq;      // [4][dk]
  float* vbuf = kbuf + 4 * (size_t)dk;       // [4][dv]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const T* q = (const T*)a.q;
  const T* k = (const T*)a.k;
  const T* v = (const T*)a.v;
  const T* d_o = (const T*)a.d_o;
  const T* datt = (const T*)a.d_att;
  T* dkp = (T*)a.dk_;
  T* dvp = (T*)a.dv_;
  const DropState ds = drop_init(a.drop);

  for (int e = tid; e < nq * dk; e += 256) {
    const int i = e / dk, c = e % dk;
    Qs[i * (dk + 1) + c] = to_f32<T>(q[((int64_t)b * nq + i) * a.ldq + h * dk + c]);
  }
  for (int e = tid; e < nq * dv; e += 256) {
    const int i = e / dv, c = e % dv;
    Gs[i * (dv + 1) + c] = to_f32<T>(d_o[((int64_t)b * nq + i) * a.lddo + h * dv + c]);
  }
  for (int i = tid; i < nq; i += 256) {
    lse_s[i] = a.lse[((int64_t)b * a.H + h) * nq + i];
    del_s[i] = a.delta[((int64_t)b * a.H + h) * nq + i];
  }
  __syncthreads();

  float* pw = pbuf + (size_t)wave * nq;
  float* dsw = dsbuf + (size_t)wave * nq;
  float* kw = kbuf + (size_t)wave * dk;
  float* vw = vbuf + (size_t)wave * dv;
  const int row0 = blockIdx.y * ROWS_PER_BLOCK;
  for (int r = wave; r < ROWS_PER_BLOCK; r += 4) {
    const int j = row0 + r;
    const bool live = j < nk;
    if (live) {
      for (int c = lane; c < dk; c += 64) kw[c] = to_f32<T>(k[((int64_t)b * nk + j) * a.ldk + h * dk + c]);
      for (int c = lane; c < dv; c += 64) vw[c] = to_f32<T>(v[((int64_t)b * nk + j) * a.ldv + h * dv + c]);
    }
    __syncthreads();
    if (live) {
      for (int i = lane; i < nq; i += 64) {
        const float* qr = Qs + (size_t)i * (dk + 1);
        const float* gr = Gs + (size_t)i * (dv + 1);
        float s = 0.f, dp = 0.f;
        for (int c = 0; c < dk; c++) s = fmaf(qr[c], kw[c], s);
        for (int c = 0; c < dv; c++) dp = fmaf(gr[c], vw[c], dp);
        if (datt) dp += to_f32<T>(datt[(((int64_t)b * a.H + h) * nq + i) * nk + j]);
        const float dm = drop_mul(ds, (uint32_t)((((int64_t)b * a.H + h) * nq + i) * nk + j));
        dp *= dm;
        s = s * a.scale + mask_at(a.mask, a.msb, a.msh, a.msq, b, h, i, j);
        const float p = __expf(s - lse_s[i]);
        pw[i] = p * dm;  // dV = P~^T dO
        dsw[i] = p * (dp - del_s[i]);
      }
    }
    __syncthreads();
    if (live) {
      for (int c = lane; c < dv; c += 64) {
        float acc = 0.f;
        for (int i = 0; i < nq; i++) acc = fmaf(pw[i], Gs[(size_t)i * (dv + 1) + c], acc);
        dvp[((int64_t)b * nk + j) * a.lddv + h * dv + c] = from_f32<T>(acc);
      }
      for (int c = lane; c < dk; c += 64) {
        float acc = 0.f;
        for (int i = 0; i < nq; i++) acc = fmaf(dsw[i], Qs[(size_t)i * (dk + 1) + c], acc);
        dkp[((int64_t)b * nk + j) * a.lddk + h * dk + c] = from_f32<T>(acc * a.scale);
      }
    }
    __syncthreads();
  }
}

constexpr size_t kMaxLds = 160 * 1024;

template <typename K>
int set_lds_limit(K kernel, size_t bytes, const char* what) {
  if (bytes > kMaxLds) {
    ovqa_set_error("%s: needs %zu B of LDS (> 160 KiB): sequence too long for the LDS-resident kernel", what, bytes);
    return OVQA_ERR_UNSUPPORTED;
  }
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
      ovqa_set_error("%s: hipFuncSetAttribute(%zu): %s", what, bytes, hipGetErrorString(e));
      return OVQA_ERR_LAUNCH;
    }
  }
  return OVQA_OK;
}

template <typename T>
int fwd_t(const ovqa::AttnArgs& a, hipStream_t st) {
  const size_t lds = ((size_t)a.nk * (a.dk + 1) + (size_t)a.nk * (a.dv + 1) + 4 * (size_t)a.nk + 4 * (size_t)a.dk) * 4;
  int rc = set_lds_limit(attn_fwd_simple_kernel<T>, lds, "attention_fwd");
  if (rc != OVQA_OK) return rc;
  dim3 grid((unsigned)(a.B * a.H), (unsigned)((a.nq + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
  hipLaunchKernelGGL(attn_fwd_simple_kernel<T>, grid, dim3(256), lds, st, a);
  return ovqa_check_launch("attention_fwd");
}

template <typename T>
int bwd_t(const ovqa::AttnBwdArgs& a, hipStream_t st) {
  const size_t lds1 = ((size_t)a.nk * (a.dk + 1) + (size_t)a.nk * (a.dv + 1) + 8 * (size_t)a.nk + 4 * (size_t)a.dk +
                       4 * (size_t)a.dv) * 4;
  int rc = set_lds_limit(attn_bwd_dq_simple_kernel<T>, lds1, "attention_bwd(dq)");
  if (rc != OVQA_OK) return rc;
  const size_t lds2 = ((size_t)a.nq * (a.dk + 1) + (size_t)a.nq * (a.dv + 1) + 2 * (size_t)a.nq + 8 * (size_t)a.nq +
                       4 * (size_t)a.dk + 4 * (size_t)a.dv) * 4;
  rc = set_lds_limit(attn_bwd_dkv_simple_kernel<T>, lds2, "attention_bwd(dkv)");
  if (rc != OVQA_OK) return rc;
  dim3 g1((unsigned)(a.B * a.H), (unsigned)((a.nq + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
  hipLaunchKernelGGL(attn_bwd_dq_simple_kernel<T>, g1, dim3(256), lds1, st, a);
  rc = ovqa_check_launch("attention_bwd(dq)");
  if (rc != OVQA_OK) return rc;
  dim3 g2((unsigned)(a.B * a.H), (unsigned)((a.nk + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK));
  hipLaunchKernelGGL(attn_bwd_dkv_simple_kernel<T>, g2, dim3(256), lds2, st, a);
  return ovqa_check_launch("attention_bwd(dkv)");
}

}  // namespace

namespace ovqa {

int simple_attention_fwd(int dtype, const AttnArgs& a, hipStream_t st) {
  if (a.B * a.H == 0 || a.nq == 0) return OVQA_OK;
  OVQA_REQUIRE(a.nk > 0, OVQA_ERR_BAD_ARG, "attention_fwd: nk must be > 0");
  return dtype == OVQA_F32 ? fwd_t<float>(a, st) : fwd_t<bf16>(a, st);
}

int simple_attention_bwd(int dtype, const AttnBwdArgs& a, hipStream_t st) {
  if (a.B * a.H == 0 || a.nq == 0) return OVQA_OK;
  OVQA_REQUIRE(a.nk > 0, OVQA_ERR_BAD_ARG, "attention_bwd: nk must be > 0");
  OVQA_REQUIRE(a.lse && a.delta, OVQA_ERR_BAD_ARG, "attention_bwd: lse and delta are required");
  return dtype == OVQA_F32 ? bwd_t<float>(a, st) : bwd_t<bf16>(a, st);
}

}  // namespace ovqa
