// bf16 MFMA GEMM for gfx950: Y[M,N] = X[M,K] . W[N,K]^T (+ fused epilogue).
//
// * 128x128x64 tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave),
//   v_mfma_f32_16x16x32_bf16, fp32 accumulators (64 VGPRs per lane).
// * Orientation is swapped (D = W_tile . X_tile^T) so that a lane's 4
//   accumulator registers are 4 CONSECUTIVE output columns of one row:
//   the epilogue reads bias/residual and writes y/preact as 8-byte vectors.
// * LDS tiles are [128 rows][64 k] bf16 (128 B per row) with the 16-byte chunk
//   index XOR-ed by (row & 7): both the ds_write_b128 of the staging pass and
//   the ds_read_b128 fragment reads are bank-conflict free.
// * global -> registers -> LDS staging, double-buffered in LDS: the loads of
//   tile t+1 are issued before the MFMAs of tile t and written after them,
//   one barrier per K-tile.
// * blockIdx is remapped so that the workgroups sharing an X row-panel run on
//   the same XCD (its L2 then serves the panel to all N-tiles).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

__device__ __forceinline__ int swz(int row, int chunk) { return (row * 8 + (chunk ^ (row & 7))) * 16; }

struct StageRegs {
  uint4 a[4];
  uint4 b[4];
};

// thread t loads chunk (t & 7) of rows (t >> 3) + 32*i, i = 0..3, of both tiles
__device__ __forceinline__ void stage_load(StageRegs& r, const bf16* __restrict__ X, int64_t ldx,
                                           const bf16* __restrict__ W, int64_t ldw, int m0, int n0, int k0, int M,
                                           int tid) {
  const int chunk = tid & 7, rbase = tid >> 3;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = rbase + 32 * i;
    int gm = m0 + row;
    gm = gm < M ? gm : M - 1;  // clamp: rows past M are never stored
    r.a[i] = *reinterpret_cast<const uint4*>(X + (int64_t)gm * ldx + k0 + chunk * 8);
    r.b[i] = *reinterpret_cast<const uint4*>(W + (int64_t)(n0 + row) * ldw + k0 + chunk * 8);
  }
}

__device__ __forceinline__ void stage_store(const StageRegs& r, char* As, char* Bs, int tid) {
  const int chunk = tid & 7, rbase = tid >> 3;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = rbase + 32 * i;
    *reinterpret_cast<uint4*>(As + swz(row, chunk)) = r.a[i];
    *reinterpret_cast<uint4*>(Bs + swz(row, chunk)) = r.b[i];
  }
}

template <typename Epi>
__global__ __launch_bounds__(256) void gemm_nt_bf16_kernel(const bf16* __restrict__ X, int64_t ldx,
                                                           const bf16* __restrict__ W, int64_t ldw, int M, int N,
                                                           int K, int tiles_m, int tiles_n, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][A|B][16 KiB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware remap: consecutive ids on one XCD walk the N-tiles of one M-panel.
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / tiles_n, tn = bid % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  f32x4 acc[4][4];  // [j: n sub-tile][i: m sub-tile]
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  StageRegs regs;
  const int nkt = K / BK;
  stage_load(regs, X, ldx, W, ldw, m0, n0, 0, M, tid);
  stage_store(regs, smem, smem + TILE_BYTES, tid);
  __syncthreads();

  const int frow = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nkt; kt++) {
    const int cur = kt & 1;
    char* As = smem + cur * 2 * TILE_BYTES;
    char* Bs = As + TILE_BYTES;
    if (kt + 1 < nkt) stage_load(regs, X, ldx, W, ldw, m0, n0, (kt + 1) * BK, M, tid);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      bf16x8 xa[4], wb[4];
#pragma unroll
      for (int i = 0; i < 4; i++)
        xa[i] = *reinterpret_cast<const bf16x8*>(As + swz(wm * 64 + i * 16 + frow, ks * 4 + fq));
#pragma unroll
      for (int j = 0; j < 4; j++)
        wb[j] = *reinterpret_cast<const bf16x8*>(Bs + swz(wn * 64 + j * 16 + frow, ks * 4 + fq));
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 4; i++)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[j][i], 0, 0, 0);
    }
    if (kt + 1 < nkt) {
      char* An = smem + (cur ^ 1) * 2 * TILE_BYTES;
      stage_store(regs, An, An + TILE_BYTES, tid);
    }
    __syncthreads();
  }

  // D[j][i][reg]: n = n0 + wn*64 + j*16 + fq*4 + reg ; m = m0 + wm*64 + i*16 + frow
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int m = m0 + wm * 64 + i * 16 + frow;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int n = n0 + wn * 64 + j * 16 + fq * 4;
      epi(m, n, acc[j][i]);
    }
  }
}

// ------------------------------------------------------------------ epilogues (4 consecutive n)
__device__ __forceinline__ void store4(bf16* p, float a, float b, float c, float d) {
  bf16x4 v;
  v[0] = (bf16)a; v[1] = (bf16)b; v[2] = (bf16)c; v[3] = (bf16)d;
  *reinterpret_cast<bf16x4*>(p) = v;
}
__device__ __forceinline__ float4 bias4(const float* bias, int n) {
  return bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
}

struct MEpiBias {
  bf16* y; int64_t ldy; const float* bias;
  __device__ __forceinline__ void operator()(int m, int n, const f32x4& a) const {
    const float4 b = bias4(bias, n);
    store4(y + (int64_t)m * ldy + n, a[0] + b.x, a[1] + b.y, a[2] + b.z, a[3] + b.w);
  }
};
struct MEpiBiasGelu {
  bf16* y; int64_t ldy; const float* bias; bf16* preact; int N; DropArgs da;
  __device__ __forceinline__ void operator()(int m, int n, const f32x4& a) const {
    const DropState ds = drop_init(da);
    const float4 b = bias4(bias, n);
    const float u0 = a[0] + b.x, u1 = a[1] + b.y, u2 = a[2] + b.z, u3 = a[3] + b.w;
    if (preact) store4(preact + (int64_t)m * N + n, u0, u1, u2, u3);
    const uint32_t idx = (uint32_t)m * (uint32_t)N + (uint32_t)n;
    store4(y + (int64_t)m * ldy + n, gelu_f(u0) * drop_mul(ds, idx), gelu_f(u1) * drop_mul(ds, idx + 1),
           gelu_f(u2) * drop_mul(ds, idx + 2), gelu_f(u3) * drop_mul(ds, idx + 3));
  }
};
struct MEpiBiasResidual {
  bf16* y; int64_t ldy; const float* bias; const bf16* res; int64_t ldres; int N; DropArgs da;
  __device__ __forceinline__ void operator()(int m, int n, const f32x4& a) const {
    const DropState ds = drop_init(da);
    const float4 b = bias4(bias, n);
    const bf16x4 r = *reinterpret_cast<const bf16x4*>(res + (int64_t)m * ldres + n);
    const uint32_t idx = (uint32_t)m * (uint32_t)N + (uint32_t)n;
    store4(y + (int64_t)m * ldy + n, (float)r[0] + (a[0] + b.x) * drop_mul(ds, idx),
           (float)r[1] + (a[1] + b.y) * drop_mul(ds, idx + 1), (float)r[2] + (a[2] + b.z) * drop_mul(ds, idx + 2),
           (float)r[3] + (a[3] + b.w) * drop_mul(ds, idx + 3));
  }
};

template <typename Epi>
int launch_nt(const void* x, int64_t ldx, const void* w, int64_t ldw, int64_t M, int64_t N, int64_t K, Epi epi,
              hipStream_t st, const char* what) {
  const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (int)(N / BN);
  const size_t lds = 4 * TILE_BYTES;  // 64 KiB
  hipLaunchKernelGGL((gemm_nt_bf16_kernel<Epi>), dim3(tiles_m * tiles_n), dim3(256), lds, st, (const bf16*)x, ldx,
                     (const bf16*)w, ldw, (int)M, (int)N, (int)K, tiles_m, tiles_n, epi);
  return ovqa_check_launch(what);
}

}  // namespace

namespace ovqa {

bool mfma_linear_fwd_supported(int epilogue, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy,
                               int64_t ldres) {
  (void)epilogue;
  return M >= 1 && N % BN == 0 && K % BK == 0 && ldx % 8 == 0 && ldy % 4 == 0 && ldres % 4 == 0;
}

int mfma_linear_fwd(int epilogue, const void* x, int64_t ldx, const void* w, const float* bias, const void* residual,
                    int64_t ldres, void* y, int64_t ldy, void* preact, int64_t M, int64_t N, int64_t K,
                    const DropArgs& da, hipStream_t st) {
  OVQA_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)y % 8 == 0), OVQA_ERR_BAD_ARG,
               "linear_fwd(bf16): x/w must be 16-byte and y 8-byte aligned");
  switch (epilogue) {
    case OVQA_EPI_BIAS:
      return launch_nt(x, ldx, w, K, M, N, K, MEpiBias{(bf16*)y, ldy, bias}, st, "linear_fwd(mfma,bias)");
    case OVQA_EPI_BIAS_GELU:
      return launch_nt(x, ldx, w, K, M, N, K, MEpiBiasGelu{(bf16*)y, ldy, bias, (bf16*)preact, (int)N, da}, st,
                       "linear_fwd(mfma,gelu)");
    case OVQA_EPI_BIAS_RESIDUAL:
      OVQA_REQUIRE(residual != nullptr, OVQA_ERR_BAD_ARG, "linear_fwd: residual epilogue needs a residual");
      return launch_nt(x, ldx, w, K, M, N, K,
                       MEpiBiasResidual{(bf16*)y, ldy, bias, (const bf16*)residual, ldres, (int)N, da}, st,
                       "linear_fwd(mfma,residual)");
  }
  ovqa_set_error("linear_fwd: unknown epilogue %d", epilogue);
  return OVQA_ERR_BAD_ARG;
}

}  // namespace ovqa
