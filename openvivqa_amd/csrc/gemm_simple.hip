// Reference-grade tiled GEMM family (VALU FMA, fp32 accumulate) for gfx950.
//
// Role: the exact-fp32 path of the library (OVQA_F32) and the on-device
// cross-check for the MFMA bf16 kernels.  One 64x64x16 tile per 256-thread
// workgroup, LDS-staged in fp32, 4x4 outputs per thread, fused epilogues.
// Handles every (transA, transB) form by indexing at tile-load time, ragged
// edges by predication.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16;

template <typename T, bool TRANS_A, bool TRANS_B, typename Epi>
__global__ __launch_bounds__(256) void gemm_simple_kernel(
    const T* __restrict__ A, int64_t lda, int64_t stride_a,
    const T* __restrict__ B, int64_t ldb, int64_t stride_b,
    int M, int N, int K, Epi epi) {
  __shared__ float As[TK][TM + 4];
  __shared__ float Bs[TK][TN + 4];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int batch = blockIdx.z;
  A += (int64_t)batch * stride_a;
  B += (int64_t)batch * stride_b;

  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = 0.f;

  for (int k0 = 0; k0 < K; k0 += TK) {
    // ---- stage A tile: As[k][m] = A(m0+m, k0+k)
    if (!TRANS_A) {
      const int m = tid >> 2, kb = (tid & 3) * 4;
      const int gm = m0 + m;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int gk = k0 + kb + i;
        float v = 0.f;
        if (gm < M && gk < K) v = to_f32<T>(A[(int64_t)gm * lda + gk]);
        As[kb + i][m] = v;
      }
    } else {
      const int k = tid >> 4, mb = (tid & 15) * 4;
      const int gk = k0 + k;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int gm = m0 + mb + i;
        float v = 0.f;
        if (gm < M && gk < K) v = to_f32<T>(A[(int64_t)gk * lda + gm]);
        As[k][mb + i] = v;
      }
    }
    // ---- stage B tile: Bs[k][n] = B(k0+k, n0+n)
    if (!TRANS_B) {
      const int k = tid >> 4, nb = (tid & 15) * 4;
      const int gk = k0 + k;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int gn = n0 + nb + i;
        float v = 0.f;
        if (gn < N && gk < K) v = to_f32<T>(B[(int64_t)gk * ldb + gn]);
        Bs[k][nb + i] = v;
      }
    } else {
      const int n = tid >> 2, kb = (tid & 3) * 4;
      const int gn = n0 + n;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int gk = k0 + kb + i;
        float v = 0.f;
        if (gn < N && gk < K) v = to_f32<T>(B[(int64_t)gn * ldb + gk]);
        Bs[kb + i][n] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; kk++) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; i++) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; j++) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int gm = m0 + ty * 4 + i;
    if (gm >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int gn = n0 + tx * 4 + j;
      if (gn < N) epi(batch, gm, gn, acc[i][j]);
    }
  }
}

// ------------------------------------------------------------ epilogues
template <typename T>
struct EpiBias {
  T* y; int64_t ldy; const float* bias;
  __device__ void operator()(int, int m, int n, float acc) const {
    y[(int64_t)m * ldy + n] = from_f32<T>(acc + (bias ? bias[n] : 0.f));
  }
};
template <typename T>
struct EpiBiasGelu {
  T* y; int64_t ldy; const float* bias; T* preact; int N; DropArgs da;
  __device__ void operator()(int, int m, int n, float acc) const {
    const DropState ds = drop_init(da);
    const float u = acc + (bias ? bias[n] : 0.f);
    if (preact) preact[(int64_t)m * N + n] = from_f32<T>(u);
    y[(int64_t)m * ldy + n] = from_f32<T>(gelu_f(u) * drop_mul(ds, (uint32_t)m * (uint32_t)N + n));
  }
};
template <typename T>
struct EpiBiasResidual {
  T* y; int64_t ldy; const float* bias; const T* res; int64_t ldres; int N; DropArgs da;
  __device__ void operator()(int, int m, int n, float acc) const {
    const DropState ds = drop_init(da);
    const float v = (acc + (bias ? bias[n] : 0.f)) * drop_mul(ds, (uint32_t)m * (uint32_t)N + n);
    y[(int64_t)m * ldy + n] = from_f32<T>(to_f32<T>(res[(int64_t)m * ldres + n]) + v);
  }
};
// fp32 residual stream: pre32 = res + drop(acc + bias), res = plain fp32 or LayerNorm(res; mean, rstd, gamma, beta)
struct EpiBiasRes32 {
  float* pre; int64_t ldpre; const float* bias; const float* res; int64_t ldres;
  const float* mean; const float* rstd; const float* gamma; const float* beta; int N; DropArgs da;
  __device__ void operator()(int, int m, int n, float acc) const {
    const DropState ds = drop_init(da);
    float r = res[(int64_t)m * ldres + n];
    if (mean) r = (r - mean[m]) * rstd[m] * gamma[n] + beta[n];
    pre[(int64_t)m * ldpre + n] = r + (acc + (bias ? bias[n] : 0.f)) * drop_mul(ds, (uint32_t)m * (uint32_t)N + n);
  }
};
// dX = dY W [* dropmask * gelu'(u)]
template <typename T>
struct EpiBwdData {
  T* dx; int64_t lddx; const T* preact; int Kcols; const T* addend; int64_t ldadd; DropArgs da;
  __device__ void operator()(int, int m, int n, float acc) const {
    float v = acc;
    if (preact) {
      const DropState ds = drop_init(da);
      const uint32_t idx = (uint32_t)m * (uint32_t)Kcols + n;
      v *= drop_mul(ds, idx) * gelu_grad_f(to_f32<T>(preact[(int64_t)m * Kcols + n]));
    }
    if (addend) v += to_f32<T>(addend[(int64_t)m * ldadd + n]);
    dx[(int64_t)m * lddx + n] = from_f32<T>(v);
  }
};
struct EpiF32Out {
  float* c; int64_t ldc; int accumulate;
  __device__ void operator()(int, int m, int n, float acc) const {
    float* p = c + (int64_t)m * ldc + n;
    *p = accumulate ? (*p + acc) : acc;
  }
};
template <typename TC>
struct EpiBatched {
  TC* c; int64_t ldc; int64_t stride_c; float alpha;
  __device__ void operator()(int b, int m, int n, float acc) const {
    c[(int64_t)b * stride_c + (int64_t)m * ldc + n] = from_f32<TC>(alpha * acc);
  }
};
struct EpiPointer {
  float* s; int T_; int Nk; float scale; const float* add_mask; const uint8_t* key_fill; const uint8_t* query_fill;
  __device__ void operator()(int b, int m, int n, float acc) const {
    float v = acc * scale;
    if (add_mask) v += add_mask[(int64_t)b * Nk + n];
    if (key_fill && key_fill[(int64_t)b * Nk + n]) v = -INFINITY;
    if (query_fill && query_fill[(int64_t)b * T_ + m]) v = -INFINITY;
    s[((int64_t)b * T_ + m) * Nk + n] = v;
  }
};

template <typename T, bool TA, bool TB, typename Epi>
int launch(const void* A, int64_t lda, int64_t sa, const void* B, int64_t ldb, int64_t sb,
           int64_t batch, int64_t M, int64_t N, int64_t K, Epi epi, hipStream_t st, const char* what) {
  if (M <= 0 || N <= 0 || batch <= 0) return OVQA_OK;
  dim3 grid((unsigned)((N + TN - 1) / TN), (unsigned)((M + TM - 1) / TM), (unsigned)batch);
  OVQA_REQUIRE(grid.y <= 65535 && grid.z <= 65535, OVQA_ERR_UNSUPPORTED, "%s: grid too large", what);
  hipLaunchKernelGGL((gemm_simple_kernel<T, TA, TB, Epi>), grid, dim3(256), 0, st,
                     (const T*)A, lda, sa, (const T*)B, ldb, sb, (int)M, (int)N, (int)K, epi);
  return ovqa_check_launch(what);
}

// column sums of dY: db[n] (+)= sum_m dy[m, n]
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, int64_t lddy, float* __restrict__ db,
                                                     int M, int N, int accumulate) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  float s = 0.f;
  if (n < N)
    for (int m = r; m < M; m += 4) s += to_f32<T>(dy[(int64_t)m * lddy + n]);
  red[r][c] = s;
  __syncthreads();
  if (r == 0 && n < N) {
    const float t = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    db[n] = accumulate ? db[n] + t : t;
  }
}

}  // namespace

namespace ovqa {

template <typename T>
static int linear_fwd_t(int epilogue, const void* x, int64_t ldx, const void* w, const float* bias,
                        const void* residual, int64_t ldres, void* y, int64_t ldy, void* preact,
                        int64_t M, int64_t N, int64_t K, const DropArgs& da, hipStream_t st) {
  switch (epilogue) {
    case OVQA_EPI_BIAS:
      return launch<T, false, true>(x, ldx, 0, w, K, 0, 1, M, N, K, EpiBias<T>{(T*)y, ldy, bias}, st, "linear_fwd(bias)");
    case OVQA_EPI_BIAS_GELU:
      return launch<T, false, true>(x, ldx, 0, w, K, 0, 1, M, N, K,
                                    EpiBiasGelu<T>{(T*)y, ldy, bias, (T*)preact, (int)N, da}, st, "linear_fwd(gelu)");
    case OVQA_EPI_BIAS_RESIDUAL:
      OVQA_REQUIRE(residual != nullptr, OVQA_ERR_BAD_ARG, "linear_fwd: residual epilogue needs a residual");
      return launch<T, false, true>(x, ldx, 0, w, K, 0, 1, M, N, K,
                                    EpiBiasResidual<T>{(T*)y, ldy, bias, (const T*)residual, ldres, (int)N, da}, st,
                                    "linear_fwd(residual)");
  }
  ovqa_set_error("linear_fwd: unknown epilogue %d", epilogue);
  return OVQA_ERR_BAD_ARG;
}

int simple_linear_fwd(int dtype, int epilogue, const void* x, int64_t ldx, const void* w, const float* bias,
                      const void* residual, int64_t ldres, void* y, int64_t ldy, void* preact,
                      int64_t M, int64_t N, int64_t K, const DropArgs& da, hipStream_t st) {
  if (dtype == OVQA_F32)
    return linear_fwd_t<float>(epilogue, x, ldx, w, bias, residual, ldres, y, ldy, preact, M, N, K, da, st);
  return linear_fwd_t<bf16>(epilogue, x, ldx, w, bias, residual, ldres, y, ldy, preact, M, N, K, da, st);
}

int simple_linear_fwd_res32(const void* x, int64_t ldx, const void* w, const float* bias, const float* residual,
                            int64_t ldres, const float* mean, const float* rstd, const float* gamma, const float* beta,
                            float* pre, int64_t ldpre, int64_t M, int64_t N, int64_t K, const DropArgs& da, hipStream_t st) {
  return launch<bf16, false, true>(x, ldx, 0, w, K, 0, 1, M, N, K,
                                   EpiBiasRes32{pre, ldpre, bias, residual, ldres, mean, rstd, gamma, beta, (int)N, da}, st,
                                   "linear_fwd_res32");
}

int simple_linear_bwd_data(int dtype, const void* dy, int64_t lddy, const void* w, void* dx, int64_t lddx,
                           const void* preact, const void* addend, int64_t ldadd, int64_t M, int64_t N, int64_t K,
                           const DropArgs& da, hipStream_t st) {
  // dx[M,K] = dy[M,N] . w[N,K]   (A = dy, B = w as [red=N, cols=K], not transposed)
  if (dtype == OVQA_F32)
    return launch<float, false, false>(dy, lddy, 0, w, K, 0, 1, M, K, N,
                                       EpiBwdData<float>{(float*)dx, lddx, (const float*)preact, (int)K, (const float*)addend, ldadd, da},
                                       st, "linear_bwd_data");
  return launch<bf16, false, false>(dy, lddy, 0, w, K, 0, 1, M, K, N,
                                    EpiBwdData<bf16>{(bf16*)dx, lddx, (const bf16*)preact, (int)K, (const bf16*)addend, ldadd, da}, st,
                                    "linear_bwd_data");
}

int simple_linear_bwd_weight(int dtype, const void* dy, int64_t lddy, const void* x, int64_t ldx, float* dw,
                             float* db, int64_t M, int64_t N, int64_t K, int accumulate, int accumulate_db,
                             hipStream_t st) {
  // dw[N,K] = dy^T[N,M] . x[M,K]  (A = dy transposed, B = x not transposed, reduction over M)
  int rc;
  if (dtype == OVQA_F32) {
    rc = launch<float, true, false>(dy, lddy, 0, x, ldx, 0, 1, N, K, M, EpiF32Out{dw, K, accumulate}, st,
                                    "linear_bwd_weight");
    if (rc == OVQA_OK && db) {
      hipLaunchKernelGGL(colsum_kernel<float>, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, st,
                         (const float*)dy, lddy, db, (int)M, (int)N, accumulate_db);
      rc = ovqa_check_launch("colsum");
    }
  } else {
    rc = launch<bf16, true, false>(dy, lddy, 0, x, ldx, 0, 1, N, K, M, EpiF32Out{dw, K, accumulate}, st,
                                   "linear_bwd_weight");
    if (rc == OVQA_OK && db) {
      hipLaunchKernelGGL(colsum_kernel<bf16>, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, st,
                         (const bf16*)dy, lddy, db, (int)M, (int)N, accumulate_db);
      rc = ovqa_check_launch("colsum");
    }
  }
  return rc;
}

template <typename T, typename TC>
static int batched_t(int ta, int tb, const void* A, int64_t lda, int64_t sa, const void* B, int64_t ldb, int64_t sb,
                     void* C, int64_t ldc, int64_t sc, int64_t batch, int64_t M, int64_t N, int64_t K, float alpha,
                     hipStream_t st) {
  EpiBatched<TC> e{(TC*)C, ldc, sc, alpha};
  if (!ta && !tb) return launch<T, false, false>(A, lda, sa, B, ldb, sb, batch, M, N, K, e, st, "batched_gemm NN");
  if (!ta && tb) return launch<T, false, true>(A, lda, sa, B, ldb, sb, batch, M, N, K, e, st, "batched_gemm NT");
  if (ta && !tb) return launch<T, true, false>(A, lda, sa, B, ldb, sb, batch, M, N, K, e, st, "batched_gemm TN");
  return launch<T, true, true>(A, lda, sa, B, ldb, sb, batch, M, N, K, e, st, "batched_gemm TT");
}

int simple_batched_gemm(int dtype, int c_dtype, int ta, int tb, const void* A, int64_t lda, int64_t sa,
                        const void* B, int64_t ldb, int64_t sb, void* C, int64_t ldc, int64_t sc, int64_t batch,
                        int64_t M, int64_t N, int64_t K, float alpha, hipStream_t st) {
  if (dtype == OVQA_F32 && c_dtype == OVQA_F32)
    return batched_t<float, float>(ta, tb, A, lda, sa, B, ldb, sb, C, ldc, sc, batch, M, N, K, alpha, st);
  if (dtype == OVQA_BF16 && c_dtype == OVQA_BF16)
    return batched_t<bf16, bf16>(ta, tb, A, lda, sa, B, ldb, sb, C, ldc, sc, batch, M, N, K, alpha, st);
  if (dtype == OVQA_BF16 && c_dtype == OVQA_F32)
    return batched_t<bf16, float>(ta, tb, A, lda, sa, B, ldb, sb, C, ldc, sc, batch, M, N, K, alpha, st);
  ovqa_set_error("batched_gemm: unsupported dtype combination %d -> %d", dtype, c_dtype);
  return OVQA_ERR_UNSUPPORTED;
}

int simple_pointer_score(int dtype, const void* q, const void* k, const float* add_mask, const uint8_t* key_fill,
                         const uint8_t* query_fill, float* scores, int64_t B, int64_t T, int64_t Nk, int64_t D,
                         float scale, hipStream_t st) {
  EpiPointer e{scores, (int)T, (int)Nk, scale, add_mask, key_fill, query_fill};
  if (dtype == OVQA_F32)
    return launch<float, false, true>(q, D, T * D, k, D, Nk * D, B, T, Nk, D, e, st, "pointer_score");
  return launch<bf16, false, true>(q, D, T * D, k, D, Nk * D, B, T, Nk, D, e, st, "pointer_score");
}

}  // namespace ovqa
