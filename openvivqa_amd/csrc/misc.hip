// Flat-arena optimiser step, casts, dropout-mask materialisation and the
// stack-level bench loss.  All HBM-bound streaming kernels: 16-byte accesses,
// grid-stride, <= 2048 workgroups.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int kMaxBlocks = 2048;

// TG = float (the arena's own gradient buffer) or bf16 (the data-parallel staging buffer after the all-reduce:
// reading it directly saves the cast back to fp32)
template <typename TG>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const TG* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   bf16* __restrict__ shadow, int64_t n, float lr,
                                                   const float* __restrict__ lr_scale_ptr, float b1, float b2,
                                                   float eps, float wd, float grad_scale,
                                                   const uint32_t* __restrict__ step_ptr) {
  const AdamK ak = adam_consts(lr, lr_scale_ptr, b1, b2, eps, wd, grad_scale, step_ptr);
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    typedef __attribute__((ext_vector_type(4))) TG g4_t;
    const g4_t g4 = reinterpret_cast<const g4_t*>(g)[i];
    const float4 gg = make_float4(to_f32<TG>(g4[0]), to_f32<TG>(g4[1]), to_f32<TG>(g4[2]), to_f32<TG>(g4[3]));
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* pa = &pp.x;
    const float* ga = &gg.x;
    float* ma = &mm.x;
    float* va = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; k++) adam_update1(ak, ga[k], pa[k], ma[k], va[k]);
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (shadow) {
      bf16x4 s;
#pragma unroll
      for (int k = 0; k < 4; k++) s[k] = (bf16)pa[k];
      reinterpret_cast<bf16x4*>(shadow)[i] = s;
    }
  }
  // tail (n not a multiple of 4)
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float pk = p[i], mk = m[i], vk = v[i];
    adam_update1(ak, to_f32<TG>(g[i]), pk, mk, vk);
    m[i] = mk;
    v[i] = vk;
    p[i] = pk;
    if (shadow) shadow[i] = (bf16)pk;
  }
}

// Adam over 64 x 64 tiles of the weight matrices, writing the bf16 shadow and its transpose (through LDS) in the same
// pass; trailing workgroups (blockIdx >= n_tiles) update the flat range of 1-D parameters.
// master weights, gradients and moments are touched ONCE per step (30 B per parameter, 1.3 GB): with the nt (streaming)
// cache policy they do not displace the bf16 shadows the next forward reads.  MEASURED (MCAN step, alternated on one box):
// 3.337 / 3.314 -> 3.288 / 3.292 ms.  -DOVQA_NT_ADAM=0 = the default policy (A/B).
#ifndef OVQA_NT_ADAM
#define OVQA_NT_ADAM 1
#endif
template <typename TG>
__global__ __launch_bounds__(256) void adam_tiled_kernel(float* __restrict__ p, const TG* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         bf16* __restrict__ shadow, bf16* __restrict__ shadow_t,
                                                         const ovqa_adam_tile* __restrict__ tiles, int n_tiles,
                                                         int64_t flat_lo, int64_t flat_hi, float lr,
                                                         const float* __restrict__ lr_scale_ptr, float b1, float b2,
                                                         float eps, float wd, float grad_scale,
                                                         const uint32_t* __restrict__ step_ptr) {
  __shared__ bf16 tile[64][64 + 8];
  const AdamK ak = adam_consts(lr, lr_scale_ptr, b1, b2, eps, wd, grad_scale, step_ptr);
  typedef __attribute__((ext_vector_type(4))) TG g4_t;
  auto update4 = [&](int64_t i4, bf16x4& s) {  // elements 4 * i4 .. 4 * i4 + 3 of the arena
#if OVQA_NT_ADAM
    const f32x4 pp_ = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + i4);
    const g4_t g4 = __builtin_nontemporal_load(reinterpret_cast<const g4_t*>(g) + i4);
    const f32x4 mm_ = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m) + i4);
    const f32x4 vv_ = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v) + i4);
    float4 pp = make_float4(pp_[0], pp_[1], pp_[2], pp_[3]), mm = make_float4(mm_[0], mm_[1], mm_[2], mm_[3]),
           vv = make_float4(vv_[0], vv_[1], vv_[2], vv_[3]);
#else
    float4 pp = reinterpret_cast<float4*>(p)[i4];
    const g4_t g4 = reinterpret_cast<const g4_t*>(g)[i4];
    float4 mm = reinterpret_cast<float4*>(m)[i4];
    float4 vv = reinterpret_cast<float4*>(v)[i4];
#endif
    float* pa = &pp.x;
    float* ma = &mm.x;
    float* va = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      adam_update1(ak, to_f32<TG>(g4[k]), pa[k], ma[k], va[k]);
      s[k] = (bf16)pa[k];
    }
#if OVQA_NT_ADAM
    __builtin_nontemporal_store(f32x4{pp.x, pp.y, pp.z, pp.w}, reinterpret_cast<f32x4*>(p) + i4);
    __builtin_nontemporal_store(f32x4{mm.x, mm.y, mm.z, mm.w}, reinterpret_cast<f32x4*>(m) + i4);
    __builtin_nontemporal_store(f32x4{vv.x, vv.y, vv.z, vv.w}, reinterpret_cast<f32x4*>(v) + i4);
#else
    reinterpret_cast<float4*>(p)[i4] = pp;
    reinterpret_cast<float4*>(m)[i4] = mm;
    reinterpret_cast<float4*>(v)[i4] = vv;
#endif
    if (shadow) reinterpret_cast<bf16x4*>(shadow)[i4] = s;
  };
  if ((int)blockIdx.x >= n_tiles) {  // flat range
    const int64_t nb = (int64_t)gridDim.x - n_tiles;
    const int64_t stride = nb * blockDim.x;
    for (int64_t i4 = (flat_lo >> 2) + ((int64_t)blockIdx.x - n_tiles) * blockDim.x + threadIdx.x; i4 < (flat_hi >> 2);
         i4 += stride) {
      bf16x4 s;
      update4(i4, s);
    }
    return;
  }
  const ovqa_adam_tile tl = tiles[blockIdx.x];
  const int tid = threadIdx.x;
  // thread -> rows tid / 16 + 16 * i (i = 0..3), columns (tid % 16) * 4 .. + 3: 16 lanes cover a 256-byte row segment
  const int c = (tid & 15) * 4;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int r = (tid >> 4) + 16 * i;
    bf16x4 s;
#pragma unroll
    for (int k = 0; k < 4; k++) s[k] = (bf16)0.f;
    if (tl.r0 + r < tl.rows && tl.c0 + c < tl.cols)
      update4((tl.off + (int64_t)(tl.r0 + r) * tl.cols + tl.c0 + c) >> 2, s);
    *reinterpret_cast<bf16x4*>(&tile[r][c]) = s;
  }
  __syncthreads();
  if (shadow_t == nullptr) return;
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const int cc = tid / 8 + 32 * pass, ch = tid % 8;  // output row = source column
    const int gc = tl.c0 + cc, gr = tl.r0 + ch * 8;
    if (gc < tl.cols && gr < tl.rows) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = tile[ch * 8 + e][cc];
      *reinterpret_cast<bf16x8*>(shadow_t + tl.off + (int64_t)gc * tl.rows + gr) = o;
    }
  }
}

__global__ void increment_kernel(uint32_t* s, uint32_t* s2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    *s = *s + 1u;
    if (s2) *s2 = *s2 + 1u;
  }
}

// The device side of "optimiser step begins": the learning rate of THIS step out of a schedule table (entry s % n for
// the s-th step), then the counters.  With it a whole training step -- Adam and its LambdaLR schedule included -- is ONE
// graph: nothing about a step comes from the host any more.
__global__ void begin_step_kernel(uint32_t* s, uint32_t* s2, const float* lr_table, uint32_t n_table, float* lr_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const uint32_t k = *s;
    if (lr_table && lr_out) *lr_out = lr_table[k % n_table];
    *s = k + 1u;
    if (s2) *s2 = *s2 + 1u;
  }
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = from_f32<TD>(to_f32<TS>(src[i]));
}

// 8 elements per lane and iteration (16-byte accesses on the bf16 side, 2 x 16 bytes on the fp32 side); used when
// both pointers are 16-byte aligned (arena segments are).  The last n % 8 elements are done by the first lanes.
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_vec8_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int64_t n) {
  typedef __attribute__((ext_vector_type(8))) TS vs_t;
  typedef __attribute__((ext_vector_type(8))) TD vd_t;
  const int64_t n8 = n >> 3;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = tid; i < n8; i += stride) {
    const vs_t v = reinterpret_cast<const vs_t*>(src)[i];
    vd_t o;
#pragma unroll
    for (int k = 0; k < 8; k++) o[k] = from_f32<TD>(to_f32<TS>(v[k]));
    reinterpret_cast<vd_t*>(dst)[i] = o;
  }
  for (int64_t i = (n8 << 3) + tid; i < n; i += stride) dst[i] = from_f32<TD>(to_f32<TS>(src[i]));
}

// du = dy * keep/(1-p) * gelu'(u): gradient through dropout(gelu(u)) of a bias+GELU linear whose output is NOT
// followed by a second fused GEMM (FeatureEmbedding).  8 elements per lane; dropout index = flat element index,
// as in the forward epilogue.
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ u,
                                                       T* __restrict__ du, int64_t n, DropArgs da) {
  typedef __attribute__((ext_vector_type(8))) T v8;
  const DropState ds = drop_init(da);
  const int64_t n8 = n >> 3;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = tid; i < n8; i += stride) {
    const v8 g = reinterpret_cast<const v8*>(dy)[i];
    const v8 x = reinterpret_cast<const v8*>(u)[i];
    v8 o;
#pragma unroll
    for (int k = 0; k < 8; k++)
      o[k] = from_f32<T>(to_f32<T>(g[k]) * drop_mul(ds, (uint32_t)(i * 8 + k)) * gelu_grad_f(to_f32<T>(x[k])));
    reinterpret_cast<v8*>(du)[i] = o;
  }
  for (int64_t i = (n8 << 3) + tid; i < n; i += stride)
    du[i] = from_f32<T>(to_f32<T>(dy[i]) * drop_mul(ds, (uint32_t)i) * gelu_grad_f(to_f32<T>(u[i])));
}

// mask[m] = (sum_d x[m, d] == pad_value * D) ? -1e5 : 0   (models/utils.py:44-58): one wave per row.
template <typename T>
__global__ __launch_bounds__(256) void row_padding_mask_kernel(const T* __restrict__ x, float* __restrict__ mask,
                                                               int64_t M, int D, float pad_total) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const T* p = x + row * D;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += to_f32<T>(p[d]);
  s = wave_sum(s);
  if (lane == 0) mask[row] = (s == pad_total) ? -100000.0f : 0.0f;
}

__global__ __launch_bounds__(256) void keep_mask_kernel(DropArgs da, uint8_t* __restrict__ out, int64_t n) {
  const DropState ds = drop_init(da);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = (!ds.on || drop_keep(ds, (uint32_t)i)) ? 1 : 0;
}

// Loss scalar without a memset and without order-dependent arithmetic (round 4: deterministic).  Every workgroup
// publishes its partial sum; the LAST one to arrive (a ticket counter) adds all partials in index order -- the same order
// whatever the arrival order -- stores (or, with `accumulate`, adds to) the loss, and resets the counter for the next
// launch.  The partials travel through device-scope atomics (exchange in, fetch-add of 0 out): those are performed at the
// point where the 8 XCDs' L2s agree, so no device-wide fence is needed -- a __threadfence() per workgroup writes back the
// XCD's whole L2 (the 6.5 MB of dx just stored included) and made the first version of this kernel 23 us instead of 8.
// The scratch is per device and process; launches that share it are ordered by the stream they run on.
constexpr int kLossBlocks = 256;
//  one scratch slot per STREAM the launches come from (hashed to kLossSlots slots): two launches on different
// streams no longer interleave tickets unless their stream handles collide in the hash -- launches of ONE stream are
// ordered anyway.  The last arriver resets its slot's ticket; an aborted launch would leave it non-zero (include/ovqa_hip.h).
constexpr int kLossSlots = 16;
__device__ float g_loss_partial[kLossSlots][kLossBlocks];
__device__ unsigned int g_loss_ticket[kLossSlots];

template <typename T>
__global__ __launch_bounds__(256) void sq_loss_kernel(const T* __restrict__ x, const T* __restrict__ tgt,
                                                      T* __restrict__ dx, float* __restrict__ loss, int64_t n, int vec,
                                                      int accumulate, int slot) {
  __shared__ float red[4];
  __shared__ bool last;
  const float inv_n = 1.f / (float)n;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float s = 0.f;
  constexpr int V = 16 / (int)sizeof(T);  // elements per 16-byte access
  const int64_t nv = vec ? n / V : 0;     // (vec: x, tgt, dx are 16-byte aligned)
  for (int64_t i = t0; i < nv; i += stride) {
    alignas(16) T xv[V], tv[V], gv[V];
    *reinterpret_cast<uint4*>(xv) = *reinterpret_cast<const uint4*>(x + i * V);
    if (tgt) *reinterpret_cast<uint4*>(tv) = *reinterpret_cast<const uint4*>(tgt + i * V);
#pragma unroll
    for (int e = 0; e < V; e++) {
      const float v = to_f32<T>(xv[e]) - (tgt ? to_f32<T>(tv[e]) : 0.f);
      s += v * v;
      gv[e] = from_f32<T>(2.f * v * inv_n);
    }
    if (dx) *reinterpret_cast<uint4*>(dx + i * V) = *reinterpret_cast<const uint4*>(gv);
  }
  for (int64_t i = nv * V + t0; i < n; i += stride) {  // tail (or everything, unaligned)
    const float v = to_f32<T>(x[i]) - (tgt ? to_f32<T>(tgt[i]) : 0.f);
    s += v * v;
    if (dx) dx[i] = from_f32<T>(2.f * v * inv_n);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float mine = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;
    // the exchange RETURNS (so it has been performed) before the ticket is taken: the ticket's increment depends on it
    const float before = atomicExch(&g_loss_partial[slot][blockIdx.x], mine);
    unsigned int one = 1u;
    asm volatile("; the ticket waits for the returned value of the exchange" : "+v"(one) : "v"(before));
    last = atomicAdd(&g_loss_ticket[slot], one) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  float t = 0.f;
  for (int b = threadIdx.x; b < (int)gridDim.x; b += 256)  // thread k: partial k (at most one: <= 256 workgroups)
    t += atomicAdd(&g_loss_partial[slot][b], 0.f);               // (a device-scope read of what the other XCDs published)
  t = wave_sum(t);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float total = (red[0] + red[1]) + (red[2] + red[3]);
    *loss = accumulate ? *loss + total : total;
    atomicExch(&g_loss_ticket[slot], 0u);
  }
}

inline int blocks_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > kMaxBlocks) b = kMaxBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

namespace ovqa {

int adam_step(float* param, const void* grad, int grad_dtype, float* m, float* v, void* shadow, int64_t n, float lr,
              const float* lr_scale_ptr, float b1, float b2, float eps, float wd, float grad_scale,
              const uint32_t* step_ptr, hipStream_t st) {
  if (n == 0) return OVQA_OK;
  OVQA_REQUIRE(((uintptr_t)param % 16 == 0) && ((uintptr_t)grad % 8 == 0) && ((uintptr_t)m % 16 == 0) &&
                   ((uintptr_t)v % 16 == 0) && (shadow == nullptr || (uintptr_t)shadow % 8 == 0),
               OVQA_ERR_BAD_ARG, "adam_step: arena pointers must be 16-byte aligned");
  if (grad_dtype == OVQA_BF16)
    hipLaunchKernelGGL(adam_kernel<bf16>, dim3(blocks_for((n + 3) / 4)), dim3(256), 0, st, param, (const bf16*)grad, m, v,
                       (bf16*)shadow, n, lr, lr_scale_ptr, b1, b2, eps, wd, grad_scale, step_ptr);
  else
    hipLaunchKernelGGL(adam_kernel<float>, dim3(blocks_for((n + 3) / 4)), dim3(256), 0, st, param, (const float*)grad, m,
                       v, (bf16*)shadow, n, lr, lr_scale_ptr, b1, b2, eps, wd, grad_scale, step_ptr);
  return ovqa_check_launch("adam_step");
}

int adam_step_tiled(float* param, const void* grad, int grad_dtype, float* m, float* v, void* shadow, void* shadow_t,
                    const ovqa_adam_tile* tiles, int n_tiles, int64_t flat_lo, int64_t flat_hi, float lr,
                    const float* lr_scale_ptr, float b1, float b2, float eps, float wd, float grad_scale,
                    const uint32_t* step_ptr, hipStream_t st) {
  const int64_t flat4 = (flat_hi - flat_lo) >> 2;
  int flat_blocks = flat4 > 0 ? (int)((flat4 + 255) / 256) : 0;
  if (flat_blocks > 64) flat_blocks = 64;
  if (n_tiles + flat_blocks == 0) return OVQA_OK;
  const dim3 grid((unsigned)(n_tiles + flat_blocks));
  if (grad_dtype == OVQA_BF16)
    hipLaunchKernelGGL(adam_tiled_kernel<bf16>, grid, dim3(256), 0, st, param, (const bf16*)grad, m, v, (bf16*)shadow,
                       (bf16*)shadow_t, tiles, n_tiles, flat_lo, flat_hi, lr, lr_scale_ptr, b1, b2, eps, wd, grad_scale,
                       step_ptr);
  else
    hipLaunchKernelGGL(adam_tiled_kernel<float>, grid, dim3(256), 0, st, param, (const float*)grad, m, v, (bf16*)shadow,
                       (bf16*)shadow_t, tiles, n_tiles, flat_lo, flat_hi, lr, lr_scale_ptr, b1, b2, eps, wd, grad_scale,
                       step_ptr);
  return ovqa_check_launch("adam_step_tiled");
}

int increment_step(uint32_t* step_ptr, uint32_t* second, hipStream_t st) {
  hipLaunchKernelGGL(increment_kernel, dim3(1), dim3(64), 0, st, step_ptr, second);
  return ovqa_check_launch("increment_step");
}

int begin_step(uint32_t* step_ptr, uint32_t* second, const float* lr_table, uint32_t n_table, float* lr_out,
               hipStream_t st) {
  hipLaunchKernelGGL(begin_step_kernel, dim3(1), dim3(64), 0, st, step_ptr, second, lr_table, n_table, lr_out);
  return ovqa_check_launch("begin_step");
}

int cast(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, hipStream_t st) {
  if (n == 0) return OVQA_OK;
  dim3 grid(blocks_for(n)), block(256);
  if (((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0) && n >= 8 && src_dtype != dst_dtype) {
    dim3 g8(blocks_for((n + 7) / 8));
    if (src_dtype == OVQA_F32 && dst_dtype == OVQA_BF16) {
      hipLaunchKernelGGL((cast_vec8_kernel<float, bf16>), g8, block, 0, st, (const float*)src, (bf16*)dst, n);
      return ovqa_check_launch("cast");
    }
    if (src_dtype == OVQA_BF16 && dst_dtype == OVQA_F32) {
      hipLaunchKernelGGL((cast_vec8_kernel<bf16, float>), g8, block, 0, st, (const bf16*)src, (float*)dst, n);
      return ovqa_check_launch("cast");
    }
  }
  if (src_dtype == OVQA_F32 && dst_dtype == OVQA_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16>), grid, block, 0, st, (const float*)src, (bf16*)dst, n);
  else if (src_dtype == OVQA_BF16 && dst_dtype == OVQA_F32)
    hipLaunchKernelGGL((cast_kernel<bf16, float>), grid, block, 0, st, (const bf16*)src, (float*)dst, n);
  else if (src_dtype == OVQA_F32 && dst_dtype == OVQA_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), grid, block, 0, st, (const float*)src, (float*)dst, n);
  else if (src_dtype == OVQA_BF16 && dst_dtype == OVQA_BF16)
    hipLaunchKernelGGL((cast_kernel<bf16, bf16>), grid, block, 0, st, (const bf16*)src, (bf16*)dst, n);
  else {
    ovqa_set_error("cast: bad dtypes %d -> %d", src_dtype, dst_dtype);
    return OVQA_ERR_BAD_ARG;
  }
  return ovqa_check_launch("cast");
}

int gelu_bwd(int dtype, const void* dy, const void* u, void* du, int64_t n, const DropArgs& da, hipStream_t st) {
  if (n == 0) return OVQA_OK;
  OVQA_REQUIRE(((uintptr_t)dy % 16 == 0) && ((uintptr_t)u % 16 == 0) && ((uintptr_t)du % 16 == 0), OVQA_ERR_BAD_ARG,
               "gelu_bwd: pointers must be 16-byte aligned");
  OVQA_REQUIRE(n < (1ll << 32), OVQA_ERR_UNSUPPORTED, "gelu_bwd: more than 2^32 elements");
  dim3 grid(blocks_for((n + 7) / 8)), block(256);
  if (dtype == OVQA_BF16)
    hipLaunchKernelGGL(gelu_bwd_kernel<bf16>, grid, block, 0, st, (const bf16*)dy, (const bf16*)u, (bf16*)du, n, da);
  else
    hipLaunchKernelGGL(gelu_bwd_kernel<float>, grid, block, 0, st, (const float*)dy, (const float*)u, (float*)du, n, da);
  return ovqa_check_launch("gelu_bwd");
}

int row_padding_mask(int dtype, const void* x, float* mask, int64_t M, int64_t D, float pad_value, hipStream_t st) {
  if (M == 0) return OVQA_OK;
  dim3 grid((unsigned)((M + 3) / 4)), block(256);
  const float total = pad_value * (float)D;
  if (dtype == OVQA_BF16)
    hipLaunchKernelGGL(row_padding_mask_kernel<bf16>, grid, block, 0, st, (const bf16*)x, mask, M, (int)D, total);
  else
    hipLaunchKernelGGL(row_padding_mask_kernel<float>, grid, block, 0, st, (const float*)x, mask, M, (int)D, total);
  return ovqa_check_launch("row_padding_mask");
}

int dropout_keep_mask(const DropArgs& da, uint8_t* out, int64_t n, hipStream_t st) {
  if (n == 0) return OVQA_OK;
  hipLaunchKernelGGL(keep_mask_kernel, dim3(blocks_for(n)), dim3(256), 0, st, da, out, n);
  return ovqa_check_launch("dropout_keep_mask");
}

int sq_loss_fwd_bwd(int dtype, const void* x, const void* target, void* dx, float* loss, int64_t n,
                    int accumulate_loss, hipStream_t st) {
  OVQA_REQUIRE(n > 0, OVQA_ERR_BAD_ARG, "sq_loss: n must be > 0");
  const int vec = (((uintptr_t)x | (uintptr_t)target | (uintptr_t)dx) & 15) == 0;
  const int64_t per = dtype == OVQA_F32 ? 4 : 8;  // elements per thread and pass
  int blocks = blocks_for((n + per - 1) / per);
  if (blocks > kLossBlocks) blocks = kLossBlocks;  // one partial per workgroup, summed in index order by the last one
  dim3 grid(blocks), block(256);
  const int slot = (int)((((uintptr_t)st) >> 6) % kLossSlots);
  if (dtype == OVQA_F32)
    hipLaunchKernelGGL(sq_loss_kernel<float>, grid, block, 0, st, (const float*)x, (const float*)target, (float*)dx,
                       loss, n, vec, accumulate_loss, slot);
  else
    hipLaunchKernelGGL(sq_loss_kernel<bf16>, grid, block, 0, st, (const bf16*)x, (const bf16*)target, (bf16*)dx, loss,
                       n, vec, accumulate_loss, slot);
  return ovqa_check_launch("sq_loss");
}

}  // namespace ovqa
