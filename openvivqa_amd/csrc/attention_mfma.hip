// MFMA attention core for gfx950 (bf16 storage, fp32 softmax), d_k = d_v in {64, 96, 128}, n_k <= 256 (192 for the
// larger heads).
//
// Whole K and V of one (batch, head) stay resident in LDS (sequences are <= 237, so a
// single-tile kernel without online-softmax rescaling is the right shape).  Scores are computed
// TRANSPOSED, S^T = K Q^T, with v_mfma_f32_32x32x16_bf16: a lane then owns ONE query (its column) and
// the key axis lives in its accumulator registers, so
//   * the softmax row reduction is in-register plus one cross-half shuffle (lane ^ 32),
//   * log-sum-exp / delta are per-lane scalars,
//   * P^T is already laid out as the B operand of the second product  O^T = V^T P^T  (the key index is
//     the accumulator row index), so P never leaves registers: no LDS round trip, no (B,H,nq,nk) tensor.
//     V^T fragments come from the row-major V image through the transposing read ds_read_b64_tr_b16.
// One 16-byte-chunk XOR swizzle, x(row) = ((row>>2)&3) | (((row>>1)&1)<<2), makes the 32-row
// ds_read_b128 operand reads AND the transposing reads of the same [rows][64] image bank-conflict free.
// Small problems are packed: a workgroup's 4 waves cover ceil(nq/32) query tiles of 4/W different
// (batch, head) pairs, so the 20-token question stack still fills its waves.
#include "common.h"
#include "kernels.h"

// Phase probe (development only, -DOVQA_PHASE_PROBE): wall-clock timestamps (100 MHz) of one thread of two workgroups
// at marked points of a kernel, read back by ovqa_debug_probe().
#ifdef OVQA_PHASE_PROBE
__device__ unsigned long long g_probe[2][16];
#define OVQA_PROBE_T(i, t)                                                               \
  do {                                                                                   \
    if (threadIdx.x == (t) && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2))          \
      g_probe[blockIdx.x == 0 ? 0 : 1][i] = wall_clock64();                              \
  } while (0)
#define OVQA_PROBE(i) OVQA_PROBE_T(i, 0)
extern "C" int ovqa_debug_probe(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe), sizeof(g_probe));
}
#else
#define OVQA_PROBE(i) do {} while (0)
#define OVQA_PROBE_T(i, t) do {} while (0)
#endif

namespace {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
constexpr float LOG2E = 1.4426950408889634f;
// raw v_exp_f32: the arguments are <= ~0 (score minus row maximum / log-sum-exp), results in [0, 1]; the libm
// exp2f wraps the instruction in denormal-range fix-ups (~4 extra VALU ops per element of a VALU-bound softmax)
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ int xs(int row) { return ((row >> 2) & 3) | (((row >> 1) & 1) << 2); }
// Head size D (64, 96 or 128 features): an image row holds D bf16 in a pitch of 128 B (D = 64) or 256 B (D = 96, 128: one
// full 64-bank row per image row; the 96-feature image leaves its last 4 chunk slots unused).  With the 256-B pitch the
// XOR pattern takes the row's parity as a fourth bit: the 16 rows of a ds_read_b128 lane group then land on 16 distinct
// 16-byte slots, and the 4 rows x 4 chunks x 2 halves of a transposing read cover the bank row exactly once -- both
// read kinds stay conflict free, as with the 128-B pitch (where parity selects the half of the bank row instead).
template <int D> struct Img {
  static constexpr int PITCH = D <= 64 ? 128 : 256;
  static constexpr int CH = D / 8;   // 16-byte chunks of a row
  static constexpr int KS = D / 16;  // k steps of a 32x32x16 product over the features
  static constexpr int DT = D / 32;  // 32-feature output tiles
  static __device__ __forceinline__ int sw(int row) { return D <= 64 ? xs(row) : (xs(row) | ((row & 1) << 3)); }
  // byte offset of 16-byte chunk `ch` of row `row`
  static __device__ __forceinline__ int off(int row, int ch) { return row * PITCH + ((ch ^ sw(row)) << 4); }
};
__device__ __forceinline__ int img_off(int row, int ch) { return Img<64>::off(row, ch); }

// One [rows_pad][64] bf16 LDS image to fill from `rows` rows of global memory (row stride ld), zero padded.
struct ImgDesc {
  char* img;
  const bf16* src;
  int64_t ld;
  int rows, rows_pad;
};

// Cooperative staging of N images.  All 16-byte loads of a pass (up to 4 per image per thread) are issued
// before the first LDS store: the staging phase is pure latency (a (b,h) problem is only 10-60 KB), so the
// number of loads in flight is what matters.
template <int N, int D = 64>
__device__ __forceinline__ void load_images(const ImgDesc (&d)[N], int tid) {
  constexpr int CH = Img<D>::CH;
  int maxtot = 0;
#pragma unroll
  for (int n = 0; n < N; n++) maxtot = max(maxtot, d[n].rows_pad * CH);
  for (int e0 = tid; e0 < maxtot; e0 += 4 * 256) {
    uint4 v[N][4];
#pragma unroll
    for (int n = 0; n < N; n++)
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int e = e0 + u * 256;
        const int row = e / CH, ch = e % CH;
        v[n][u] = make_uint4(0u, 0u, 0u, 0u);
        if (e < d[n].rows_pad * CH && row < d[n].rows)
          v[n][u] = *reinterpret_cast<const uint4*>(d[n].src + (int64_t)row * d[n].ld + ch * 8);
      }
#pragma unroll
    for (int n = 0; n < N; n++)
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int e = e0 + u * 256;
        if (e < d[n].rows_pad * CH) *reinterpret_cast<uint4*>(d[n].img + Img<D>::off(e / CH, e % CH)) = v[n][u];
      }
  }
}

// key-padding masks (msq == 0: one fp32 row per (b,h)) are staged in LDS once per problem; keys beyond nk
// get -inf so the bound check folds into the same add.  Other mask shapes are read from global memory.
__device__ __forceinline__ void load_mask_row(float* dst, const float* mask, int nk, int nk_pad, int tid) {
  for (int j = tid; j < nk_pad; j += 256) dst[j] = j < nk ? (mask ? mask[j] : 0.f) : -INFINITY;
}

// k-contiguous 32-row operand fragment (A or B of 32x32x16): lane -> row base+(lane&31), k = 16*ks + 8*(lane>>5) + j
template <int D = 64>
__device__ __forceinline__ bf16x8 frag_rows(const char* img, int base, int ks, int lane) {
  return *reinterpret_cast<const bf16x8*>(img + Img<D>::off(base + (lane & 31), 2 * ks + (lane >> 5)));
}
// transposed operand fragment from a [k rows][64] image: lane -> column cbase+(lane&31),
// element j <-> k row  kbase + 8*(j>>2) + 4*(lane>>5) + (j&3)   (the order an accumulator tile has)
template <int D = 64>
__device__ __forceinline__ bf16x8 frag_tr(const char* img, int kbase, int cbase, int lane) {
  const int row = kbase + 4 * (lane >> 5) + ((lane >> 2) & 3);
  const int col = cbase + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int ch = col >> 3, sub = (col & 7) * 2;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + Img<D>::off(row, ch) + sub));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + Img<D>::off(row + 8, ch) + sub));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// A probability as two bf16 in one 32-bit register: low half = bf16(p), high half = bf16(p - bf16(p)).
__device__ __forceinline__ float pack_hi_lo(float p) {
  const bf16 hb = (bf16)p;
  const bf16 lb = (bf16)(p - (float)hb);
  return __uint_as_float((uint32_t)__builtin_bit_cast(unsigned short, hb) |
                         ((uint32_t)__builtin_bit_cast(unsigned short, lb) << 16));
}
__device__ __forceinline__ bf16 packed_hi(float v) {
  return __builtin_bit_cast(bf16, (unsigned short)(__float_as_uint(v) & 0xffffu));
}
__device__ __forceinline__ bf16 packed_lo(float v) {
  return __builtin_bit_cast(bf16, (unsigned short)(__float_as_uint(v) >> 16));
}

// ------------------------------------------------------------------------------------------ forward
// grid.x = ceil(B*H / G), grid.y = ceil(nq / (32*W)) ; W = query tiles per problem in this workgroup
// ROWMASK: the mask is one fp32 row per (b,h) (key padding; staged in LDS, -inf beyond n_k) -- the common case,
// compiled without the per-element bound checks / global mask reads; WANT_ATT: also write the probabilities.
// One wave: 32 queries (q0 .. q0+31 of problem (b, h), staged at rows qrow0.. of the Q image) against all keys of the
// problem's K / V images; writes o, lse (and the probabilities).  Shared by the plain forward kernel and by the fused
// projection + attention kernel.
template <int NKT, bool ROWMASK, bool WANT_ATT, int D>
__device__ __forceinline__ void attn_fwd_core(const ovqa::AttnArgs& a, int b, int h, int q0, int qrow0, const char* Qs,
                                              const char* Ks, const char* Vs, const float* mlds, int lane) {
  constexpr int KS = Img<D>::KS, DT = Img<D>::DT;
  const int nk = a.nk, nq = a.nq;
  // ---- S^T = K Q^T  (rows = keys, columns = this wave's 32 queries)
  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ks++) qf[ks] = frag_rows<D>(Qs, qrow0, ks, lane);
  f32x16 st[NKT];
#pragma unroll
  for (int t = 0; t < NKT; t++) {
#pragma unroll
    for (int r = 0; r < 16; r++) st[t][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
      st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<D>(Ks, t * 32, ks, lane), qf[ks], st[t], 0, 0, 0);
  }

  // ---- softmax over keys (in-lane registers + the partner half-wave)
  const int q = q0 + (lane & 31);
  const bool qok = q < nq;
  const float* mrow = a.mask ? a.mask + (int64_t)b * a.msb + (int64_t)h * a.msh + (int64_t)(qok ? q : 0) * a.msq : nullptr;
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NKT; t++)
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) {
      const int key0 = t * 32 + 8 * g4 + 4 * (lane >> 5);  // acc_row(4*g4 + e, lane) = key0 + e
      if constexpr (ROWMASK) {
        const float4 m4 = *reinterpret_cast<const float4*>(mlds + key0);
        const float mm[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          float sv = st[t][4 * g4 + e] * a.scale + mm[e];
          // prefix-LM corner (M4C's multimodal transformer, mmf_m4c.py:333-340: the decoding positions are the last `tail`
          // of the sequence and see each other causally; everything else is the key mask row): computed, not read
          if (a.tail && key0 + e > q && q >= nq - a.tail) sv = -INFINITY;
          st[t][4 * g4 + e] = sv;
          mx = fmaxf(mx, sv);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int key = key0 + e;
          float sv = -INFINITY;
          if (key < nk) sv = st[t][4 * g4 + e] * a.scale + (mrow ? mrow[key] : 0.f);
          st[t][4 * g4 + e] = sv;
          mx = fmaxf(mx, sv);
        }
      }
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NKT; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const float p = exp2_fast((st[t][r] - mx) * LOG2E);
      sum += p;
      // from here on the register holds p as TWO bf16: its bf16 value (low half: the operand of P.V) and the bf16 of
      // what that rounding dropped (high half: the operand of the o_lo pass) -- 16 bits of p in the same register
      st[t][r] = pack_hi_lo(p);
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
  if (qok && a.lse && lane < 32) a.lse[((int64_t)b * a.H + h) * nq + q] = mx + __logf(sum);
  if (WANT_ATT && qok) {
    bf16* arow = (bf16*)a.att + (((int64_t)b * a.H + h) * nq + q) * nk;
#pragma unroll
    for (int t = 0; t < NKT; t++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int key = t * 32 + acc_row(r, lane);
        if (key < nk) arow[key] = (bf16)(((float)packed_hi(st[t][r]) + (float)packed_lo(st[t][r])) * inv);
      }
  }

  // ---- O^T = V^T P^T : P^T accumulator registers are the B operand (k = key), V^T via transposing reads.
  // One 32-feature slice of the output at a time (16 accumulator registers live instead of 16 * DT: the packed
  // probabilities must stay live across both passes).
  // o_lo (training, bf16): what the backward needs for delta = dO . O is O = P V with the UNROUNDED probabilities it
  // recomputes (dS = P (dP - delta) only cancels when delta = sum_j P_ij dP_ij with that same P): the low halves of the
  // probabilities go through the matrix cores once more, on top of the rounding residual of o in the same
  // accumulators; o_lo = bf16 of the total, so that o + o_lo = P V to ~16 bits.
  bf16* orow = (bf16*)a.o + ((int64_t)b * nq + (qok ? q : 0)) * a.ldo + h * D;
  bf16* lrow = (bf16*)a.o_lo + ((int64_t)b * nq + (qok ? q : 0)) * a.ldo + h * D;
  const bool want_lo = a.o_lo != nullptr;  // workgroup-uniform
#pragma unroll 1  // (a real loop: unrolled, hipcc interleaves the slices and keeps ~40 more registers live)
  for (int d = 0; d < DT; d++) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; t++)
#pragma unroll
      for (int s = 0; s < 2; s++) {
        bf16x8 pb;
#pragma unroll
        for (int j = 0; j < 8; j++) pb[j] = packed_hi(st[t][8 * s + j]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(Vs, t * 32 + 16 * s, d * 32, lane), pb, acc, 0, 0, 0);
      }
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) {
      bf16x4 o4;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float of = acc[4 * g4 + e] * inv;
        o4[e] = (bf16)of;
        acc[4 * g4 + e] = (of - (float)o4[e]) * sum;  // what the rounding took, in the accumulator's units
      }
      if (qok) *reinterpret_cast<bf16x4*>(orow + d * 32 + 8 * g4 + 4 * (lane >> 5)) = o4;
    }
    if (want_lo) {
#pragma unroll
      for (int t = 0; t < NKT; t++)
#pragma unroll
        for (int s = 0; s < 2; s++) {
          bf16x8 pl;
#pragma unroll
          for (int j = 0; j < 8; j++) pl[j] = packed_lo(st[t][8 * s + j]);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(Vs, t * 32 + 16 * s, d * 32, lane), pl, acc, 0, 0, 0);
        }
      if (qok) {
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
          bf16x4 l4;
#pragma unroll
          for (int e = 0; e < 4; e++) l4[e] = (bf16)(acc[4 * g4 + e] * inv);
          *reinterpret_cast<bf16x4*>(lrow + d * 32 + 8 * g4 + 4 * (lane >> 5)) = l4;
        }
      }
    }
  }
}

template <int NKT, bool ROWMASK, bool WANT_ATT, int D = 64>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(ovqa::AttnArgs a, int W, int G) {
  constexpr int PITCH = Img<D>::PITCH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nk = a.nk, nq = a.nq;
  const int q_rows = 32 * W;                         // query rows staged per problem
  const int prob_bytes = (q_rows + 2 * NKT * 32) * PITCH + NKT * 32 * 4;  // Q | K | V | mask row
  const int slot = wave / W, tq = wave % W;
  const int q_blk0 = blockIdx.y * q_rows;

  // ---- stage Q / K / V of the G problems of this workgroup (all threads)
  for (int g = 0; g < G; g++) {
    const int64_t pid = (int64_t)blockIdx.x * G + g;
    if (pid >= (int64_t)a.B * a.H) break;
    const int b = (int)(pid / a.H), h = (int)(pid % a.H);
    char* base = smem + g * prob_bytes;
    const int qr = min(q_rows, nq - q_blk0);
    const ImgDesc d[3] = {
        {base, (const bf16*)a.q + ((int64_t)b * nq + q_blk0) * a.ldq + h * D, a.ldq, qr, q_rows},
        {base + q_rows * PITCH, (const bf16*)a.k + (int64_t)b * nk * a.ldk + h * D, a.ldk, nk, NKT * 32},
        {base + (q_rows + NKT * 32) * PITCH, (const bf16*)a.v + (int64_t)b * nk * a.ldv + h * D, a.ldv, nk, NKT * 32}};
    load_images<3, D>(d, tid);
    if (ROWMASK)
      load_mask_row(reinterpret_cast<float*>(base + (q_rows + 2 * NKT * 32) * PITCH),
                    a.mask ? a.mask + (int64_t)b * a.msb + (int64_t)h * a.msh : nullptr, nk, NKT * 32, tid);
  }
  __syncthreads();

  const int64_t pid = (int64_t)blockIdx.x * G + slot;
  if (slot >= G || pid >= (int64_t)a.B * a.H) return;
  const int b = (int)(pid / a.H), h = (int)(pid % a.H);
  const int q0 = q_blk0 + tq * 32;
  if (q0 >= nq) return;
  const char* Qs = smem + slot * prob_bytes;
  const char* Ks = Qs + q_rows * PITCH;
  const char* Vs = Ks + NKT * 32 * PITCH;
  const float* mlds = reinterpret_cast<const float*>(Vs + NKT * 32 * PITCH);
  attn_fwd_core<NKT, ROWMASK, WANT_ATT, D>(a, b, h, q0, tq * 32, Qs, Ks, Vs, mlds, lane);
}

// ------------------------------------------------------------------------- fused Q/K/V projection + attention
// Self-attention forward with the projections inside (hard part 1): a workgroup owns S samples of
// ONE head -- x rows [S * RP][512] against the head's 192 weight rows (64 of fc_q, fc_k, fc_v each) -- computes
// Q | K | V = x W_h^T + b with a GEMM main loop like gemm_mfma.hip's (v_mfma_f32_16x16x32_bf16, direct-to-LDS ring of
// four 32-deep K tiles, XOR-swizzled [rows][32] images, weight rows staged in the permuted order that gives a lane 8
// consecutive output features), stores the bf16 projections BOTH to HBM (backward needs them) and into the LDS images the
// attention core reads, and runs that core on them: the projected Q, K, V are never re-read from HBM, the separate
// QKV GEMM launch is gone, and per x row panel the head's weights are streamed once for S samples.
//   RP = rows per sample, padded (32, 64 or 128; rows beyond n re-read row n-1: finite, masked / not stored),
//   M = S * RP = 128 or 256 GEMM rows per workgroup, 8 waves as 4 (rows) x 2 (features): NI = M / 64 row units x 6
//   feature units of 16 per wave.  Attention: wave w takes query tile w % (RP/32) of sample w / (RP/32).
__device__ __forceinline__ int perm32(int t) { return (t & ~31) | ((t & 12) << 1) | (((t >> 4) & 1) << 2) | (t & 3); }
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

struct QkvAttnArgs {
  const bf16* x; int64_t ldx;       // [B * n, Dm]
  const bf16* w;                    // packed [3 * H * 64, Dm]: fc_q | fc_k | fc_v rows
  const float* bias;                // [3 * H * 64] or nullptr
  bf16* qkv; int64_t ldqkv;         // [B * n, 3 * H * 64] out
  ovqa::AttnArgs att;               // o, ldo, lse, mask (row mask or none), B, H, nq = nk = n, scale
  int Dm;
};

// BKF = K step (32 or 64), NBUF = ring depth.  The direct-to-LDS stream is faster with 128-byte row segments (BKF = 64)
// and with two workgroups per CU (scripts/lds_stream_probe.py: 105 GB/s per CU against 64-71), which is what the
// 128-row single-sample form <128, 1, 64, 2> buys (80 KiB of LDS: two workgroups per CU, the attention of one under the
// projection of the other) at the price of streaming the head's weights once per sample instead of once per pair.
template <int RP, int S, bool ROWMASK, int BKF = 32, int NBUF = 4>
__global__ __launch_bounds__(512) void attn_qkv_fwd_mfma_kernel(QkvAttnArgs g) {
  constexpr int M = RP * S, NI = M / 64, NJ = 6, NKT = RP / 32, TQ = RP / 32;
  constexpr int PROWS = 1024 / (BKF * 2);            // rows of a 1 KiB staging piece: 16 (64-B rows) or 8 (128-B rows)
  constexpr int CHR = BKF / 8;                       // 16-byte chunks per row: 4 or 8
  constexpr int WCH = 192 / PROWS, XCH = M / PROWS;  // pieces of the W / x tiles
  constexpr int PCS = WCH + XCH;                     // pieces per stage
  constexpr int PER = (PCS + 7) / 8;                 // per wave: PER pieces, the last one only for waves < PCS - 8 * (PER - 1)
  constexpr int FULL = PCS - 8 * (PER - 1);          // waves with PER pieces
  constexpr int STAGE = PCS * 1024;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ovqa::AttnArgs& a = g.att;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave >> 1, wr = wave & 1;
  const int h = blockIdx.y, b0 = blockIdx.x * S, n = a.nq, HD = a.H * 64;

  f32x4 acc[NJ][NI];
#pragma unroll
  for (int j = 0; j < NJ; j++)
#pragma unroll
    for (int i = 0; i < NI; i++) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // [rows][32] bf16 image, 64 B per row: chunk ch of row r sits in 16-byte slot ch ^ g((r >> 2) & 3), g = {0, 2, 3, 1}.
  // ds_read_b128 serves the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): with
  // lane -> (row base + (lane & 15), chunk lane >> 4) a group holds rows {0-3, 12-15} of one chunk and rows {4-11} of
  // the next, and this g puts their 16 slots on 16 distinct 16-byte bank columns (the plain (r >> 2) & 3 is 2-way
  // conflicting: 1.5 M conflict cycles per launch).  A staging piece is 16 rows = 1 KiB; lane l of the loading wave
  // fills slot l & 3 of row l >> 2, i.e. fetches global chunk (l & 3) ^ g.
  // (128-byte rows, BKF = 64: the GEMM kernels' slot = chunk ^ (row & 7))
  auto swz = [](int row) { return BKF == 32 ? (0x78 >> (2 * ((row >> 2) & 3))) & 3 : (row & 7); };
  auto off32 = [&](int row, int ch) { return row * (BKF * 2) + ((ch ^ swz(row)) << 4); };
  const bf16* src[PER];  // this lane's source row (+ chunk) of each of its pieces, K offset 0
#pragma unroll
  for (int i = 0; i < PER; i++) {
    const int ci = wave + 8 * i;
    const int prow = lane / CHR, pch = lane % CHR;
    if (ci < WCH) {  // weight rows of this head: tile row t -> kind t/64 (q, k, v), feature t%64, permuted within 32
      const int row = ci * PROWS + prow;
      const int t = perm32(row);
      src[i] = g.w + (int64_t)((t >> 6) * HD + h * 64 + (t & 63)) * g.Dm + ((pch ^ swz(row)) << 3);
    } else {         // x rows: sample (row / RP), position clamped to the sample's last row
      const int row = (ci - WCH) * PROWS + prow;
      int bs = b0 + row / RP;
      bs = bs < a.B ? bs : a.B - 1;
      int r = row % RP;
      r = r < n ? r : n - 1;
      src[i] = g.x + ((int64_t)bs * n + r) * g.ldx + ((pch ^ swz(row)) << 3);
    }
  }
  // the epilogue's bias values (8 consecutive features per (jp) of this lane), requested before the K loop
  float4 bias_r[NJ / 2][2];
#pragma unroll
  for (int jp = 0; jp < NJ / 2; jp++) {
    const int f = wr * 96 + jp * 32 + (lane >> 4) * 8;
    const int gcol = (f >> 6) * HD + h * 64 + (f & 63);
    bias_r[jp][0] = g.bias ? *reinterpret_cast<const float4*>(g.bias + gcol) : make_float4(0.f, 0.f, 0.f, 0.f);
    bias_r[jp][1] = g.bias ? *reinterpret_cast<const float4*>(g.bias + gcol + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const bool full = wave < FULL;  // wave-uniform
  auto issue = [&](int kt) {
    char* buf = smem + (kt % NBUF) * STAGE;
#pragma unroll
    for (int i = 0; i < PER; i++) {
      if (i == PER - 1 && !full) break;
      __builtin_amdgcn_global_load_lds((gbl_void*)(src[i] + kt * BKF), (lds_void*)(buf + (wave + 8 * i) * 1024), 16, 0, 0);
    }
  };
  const int nkt = g.Dm / BKF;
  OVQA_PROBE(0);
#pragma unroll
  for (int p = 0; p < NBUF - 1; p++)
    if (p < nkt) issue(p);
  for (int kt = 0; kt < nkt; kt++) {
    // stage kt has landed once at most the loads of the two younger stages are outstanding (in-order return)
    if (kt + NBUF - 2 >= nkt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (full) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * (NBUF - 2)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PER - 1) * (NBUF - 2)) : "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (kt + NBUF - 1 < nkt) issue(kt + NBUF - 1);
    const char* Ws = smem + (kt % NBUF) * STAGE;
    const char* Xs = Ws + WCH * 1024;
#pragma unroll
    for (int ks = 0; ks < BKF / 32; ks++) {
      bf16x8 pf[NJ], qf[NI];
#pragma unroll
      for (int j = 0; j < NJ; j++)
        pf[j] = *reinterpret_cast<const bf16x8*>(Ws + off32(wr * 96 + j * 16 + (lane & 15), ks * 4 + (lane >> 4)));
#pragma unroll
      for (int i = 0; i < NI; i++)
        qf[i] = *reinterpret_cast<const bf16x8*>(Xs + off32(wc * (NI * 16) + i * 16 + (lane & 15), ks * 4 + (lane >> 4)));
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int i = 0; i < NI; i++) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[j], qf[i], acc[j][i], 0, 0, 0);
    }
  }
  OVQA_PROBE(1);
  __syncthreads();  // every wave is done with the staging buffers: they become the Q | K | V images
  OVQA_PROBE(2);

  // ---- epilogue: + bias -> bf16 -> HBM (rows < n) and the LDS images [sample][kind][RP rows][64]
  float* mrow_s = reinterpret_cast<float*>(smem + S * 3 * RP * 128);  // [S][RP] mask rows behind the images
#pragma unroll
  for (int i = 0; i < NI; i++) {
    const int row = wc * (NI * 16) + i * 16 + (lane & 15);
    const int sm = row / RP, r = row % RP;
    const int bs = b0 + sm;
    bf16* grow = g.qkv + ((int64_t)(bs < a.B ? bs : 0) * n + (r < n ? r : 0)) * g.ldqkv + h * 64;
    const bool live = bs < a.B && r < n;
#pragma unroll
    for (int jp = 0; jp < NJ / 2; jp++) {
      const int f = wr * 96 + jp * 32 + (lane >> 4) * 8;  // 8 consecutive features of one of q / k / v
      const int kind = f >> 6, col = f & 63;
      const float4 c0 = bias_r[jp][0], c1 = bias_r[jp][1];
      bf16x8 o8;
      o8[0] = (bf16)(acc[2 * jp][i][0] + c0.x); o8[1] = (bf16)(acc[2 * jp][i][1] + c0.y);
      o8[2] = (bf16)(acc[2 * jp][i][2] + c0.z); o8[3] = (bf16)(acc[2 * jp][i][3] + c0.w);
      o8[4] = (bf16)(acc[2 * jp + 1][i][0] + c1.x); o8[5] = (bf16)(acc[2 * jp + 1][i][1] + c1.y);
      o8[6] = (bf16)(acc[2 * jp + 1][i][2] + c1.z); o8[7] = (bf16)(acc[2 * jp + 1][i][3] + c1.w);
      *reinterpret_cast<bf16x8*>(smem + ((sm * 3 + kind) * RP) * 128 + Img<64>::off(r, col >> 3)) = o8;
      if (live) *reinterpret_cast<bf16x8*>(grow + kind * HD + col) = o8;
    }
  }
  if (ROWMASK) {
    for (int e = tid; e < S * RP; e += 512) {
      const int sm = e / RP, j = e % RP;
      int bs = b0 + sm;
      bs = bs < a.B ? bs : a.B - 1;
      const float* mr = a.mask ? a.mask + (int64_t)bs * a.msb + (int64_t)h * a.msh : nullptr;
      mrow_s[e] = j < n ? (mr ? mr[j] : 0.f) : -INFINITY;
    }
  }
  OVQA_PROBE(3);
  __syncthreads();
  OVQA_PROBE(4);

  // ---- attention on the images: one wave per (sample, 32-query tile)
  const int sm = wave / TQ, tq = wave % TQ;
  if (sm >= S || b0 + sm >= a.B || tq * 32 >= n) return;
  const char* Qs = smem + (sm * 3 + 0) * RP * 128;
  const char* Ks = smem + (sm * 3 + 1) * RP * 128;
  const char* Vs = smem + (sm * 3 + 2) * RP * 128;
  attn_fwd_core<NKT, ROWMASK, false, 64>(a, b0 + sm, h, tq * 32, tq * 32, Qs, Ks, Vs, mrow_s + sm * RP, lane);
  OVQA_PROBE(5);
}

// ------------------------------------------------------------------------- fused Q projection + attention (round 3)
// Cross / guided attention forward with the QUERY projection inside: the keys and values are already projected (the
// hoisted K / V GEMM of the guided stack, or a packed K | V projection), only q = x W_q^T + b_q is computed here -- the
// main loop of attn_qkv_fwd_mfma_kernel on 64 weight rows instead of 192 -- stored to HBM (backward reads it) and into
// the LDS image the attention core reads; K / V of the head (NKT * 32 padded keys) are staged next to it.  One launch
// instead of a 6400 x 512 <- 512 GEMM + the attention kernel, and q is not re-read.
// RP = 128 query rows of ONE sample per workgroup (8 waves as 4 (rows) x 2 (features): 32 x 32 per wave), K steps of 64,
// ring of NBUF stages.
struct QAttnArgs {
  const bf16* x; int64_t ldx;       // [B * nq, Dm]
  const bf16* w;                    // fc_q weight [H * 64, Dm]
  const float* bias;                // [H * 64] or nullptr
  bf16* q; int64_t ldq;             // [B * nq, H * 64] out
  ovqa::AttnArgs att;               // k, v (+ strides), o, lse, o_lo, key mask row, B, H, nq, nk, scale
  int Dm;
};

// NH = heads per workgroup (1: 8 waves; 2: 16 waves, the x rows are staged ONCE for both heads -- 32 instead of 48 KB per
// K step through the CU's fetch path -- and 64 samples x 4 head pairs are exactly one workgroup per CU).
template <int NKT, int NBUF, int NH>
__global__ __launch_bounds__(512 * NH) void attn_q_fwd_mfma_kernel(QAttnArgs g) {
  constexpr int RP = 128, BKF = 64, NI = 2, NJ = 2, TQ = RP / 32, KP = NKT * 32, NW = 8 * NH;
  constexpr int PROWS = 8, CHR = 8;                       // a 1 KiB staging piece = 8 rows of 128 B
  constexpr int WCH = 64 * NH / PROWS, XCH = RP / PROWS;  // weight / x pieces per stage
  constexpr int PER = (WCH + XCH) / NW;
  constexpr int STAGE = (WCH + XCH) * 1024;
  constexpr int IMG = (RP + 2 * KP) * 128;                // Q | K | V images of one head
  static_assert((WCH + XCH) % NW == 0, "every wave stages the same number of pieces");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ovqa::AttnArgs& a = g.att;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave / (2 * NH), wr = wave % (2 * NH);  // 4 row quarters x 2 NH feature slabs of 32
  const int h0 = blockIdx.y * NH, b = blockIdx.x, nq = a.nq, nk = a.nk;

  f32x4 acc[NJ][NI];
#pragma unroll
  for (int j = 0; j < NJ; j++)
#pragma unroll
    for (int i = 0; i < NI; i++) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto off32 = [&](int row, int ch) { return row * (BKF * 2) + ((ch ^ (row & 7)) << 4); };
  const bf16* src[PER];
#pragma unroll
  for (int i = 0; i < PER; i++) {
    const int ci = wave + NW * i;
    const int prow = lane / CHR, pch = lane % CHR;
    if (ci < WCH) {  // the heads' fc_q rows, permuted within 32 (a lane then owns 8 consecutive output features)
      const int row = ci * PROWS + prow;
      src[i] = g.w + (int64_t)(h0 * 64 + perm32(row)) * g.Dm + ((pch ^ (row & 7)) << 3);
    } else {         // x rows of the sample, clamped to its last row
      const int row = (ci - WCH) * PROWS + prow;
      const int r = row < nq ? row : nq - 1;
      src[i] = g.x + ((int64_t)b * nq + r) * g.ldx + ((pch ^ (row & 7)) << 3);
    }
  }
  float4 bias_r[2];
  {
    const int f = h0 * 64 + wr * 32 + (lane >> 4) * 8;
    bias_r[0] = g.bias ? *reinterpret_cast<const float4*>(g.bias + f) : make_float4(0.f, 0.f, 0.f, 0.f);
    bias_r[1] = g.bias ? *reinterpret_cast<const float4*>(g.bias + f + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  auto issue = [&](int kt) {
    char* buf = smem + (kt % NBUF) * STAGE;
#pragma unroll
    for (int i = 0; i < PER; i++)
      __builtin_amdgcn_global_load_lds((gbl_void*)(src[i] + kt * BKF), (lds_void*)(buf + (wave + NW * i) * 1024), 16, 0, 0);
  };
  const int nkt = g.Dm / BKF;
#pragma unroll
  for (int p = 0; p < NBUF - 1; p++)
    if (p < nkt) issue(p);
  for (int kt = 0; kt < nkt; kt++) {
    if (kt + NBUF - 2 >= nkt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * (NBUF - 2)) : "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (kt + NBUF - 1 < nkt) issue(kt + NBUF - 1);
    const char* Ws = smem + (kt % NBUF) * STAGE;
    const char* Xs = Ws + WCH * 1024;
#pragma unroll
    for (int ks = 0; ks < BKF / 32; ks++) {
      bf16x8 pf[NJ], qf[NI];
#pragma unroll
      for (int j = 0; j < NJ; j++)
        pf[j] = *reinterpret_cast<const bf16x8*>(Ws + off32(wr * 32 + j * 16 + (lane & 15), ks * 4 + (lane >> 4)));
#pragma unroll
      for (int i = 0; i < NI; i++)
        qf[i] = *reinterpret_cast<const bf16x8*>(Xs + off32(wc * 32 + i * 16 + (lane & 15), ks * 4 + (lane >> 4)));
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int i = 0; i < NI; i++) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[j], qf[i], acc[j][i], 0, 0, 0);
    }
  }
  __syncthreads();  // every wave is done with the staging buffers: they become the Q | K | V images

  // ---- images per head: Q [RP][64] from the accumulators (+ bias; also to HBM), K | V [KP][64] from HBM; mask rows behind
  float* mrow_all = reinterpret_cast<float*>(smem + NH * IMG);
  if (tid < 256 * NH) {
    const int hh = tid >> 8, t = tid & 255;
    char* base = smem + hh * IMG;
    const ImgDesc kv[2] = {{base + RP * 128, (const bf16*)a.k + (int64_t)b * nk * a.ldk + (h0 + hh) * 64, a.ldk, nk, KP},
                           {base + (RP + KP) * 128, (const bf16*)a.v + (int64_t)b * nk * a.ldv + (h0 + hh) * 64, a.ldv, nk, KP}};
    load_images<2, 64>(kv, t);
    load_mask_row(mrow_all + hh * KP, a.mask ? a.mask + (int64_t)b * a.msb + (int64_t)(h0 + hh) * a.msh : nullptr, nk, KP, t);
  }
  {
    const int hh = wr >> 1;
    char* Qs = smem + hh * IMG;
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int r = wc * 32 + i * 16 + (lane & 15);
      const int col = (wr & 1) * 32 + (lane >> 4) * 8;
      bf16x8 o8;
      o8[0] = (bf16)(acc[0][i][0] + bias_r[0].x); o8[1] = (bf16)(acc[0][i][1] + bias_r[0].y);
      o8[2] = (bf16)(acc[0][i][2] + bias_r[0].z); o8[3] = (bf16)(acc[0][i][3] + bias_r[0].w);
      o8[4] = (bf16)(acc[1][i][0] + bias_r[1].x); o8[5] = (bf16)(acc[1][i][1] + bias_r[1].y);
      o8[6] = (bf16)(acc[1][i][2] + bias_r[1].z); o8[7] = (bf16)(acc[1][i][3] + bias_r[1].w);
      *reinterpret_cast<bf16x8*>(Qs + Img<64>::off(r, col >> 3)) = o8;
      if (r < nq) *reinterpret_cast<bf16x8*>(g.q + ((int64_t)b * nq + r) * g.ldq + (h0 + hh) * 64 + col) = o8;
    }
  }
  __syncthreads();

  // ---- attention on the images: one wave per (head, 32-query tile); the other waves are done
  const int hh = wave / TQ, tq = wave % TQ;
  if (hh >= NH || tq * 32 >= nq) return;
  const char* Qs = smem + hh * IMG;
  attn_fwd_core<NKT, true, false, 64>(a, b, h0 + hh, tq * 32, tq * 32, Qs, Qs + RP * 128, Qs + (RP + KP) * 128,
                                      mrow_all + hh * KP, lane);
}

// ------------------------------------------------------------------------------------------ backward
// Kernel A: one wave = 32 queries (columns).  delta_q = dO_q . O_q ;  for every key tile:
//   S^T = K Q^T, dP^T = V dO^T, P^T = exp(S^T*scale + mask - lse_q), dS^T = P^T (dP^T - delta_q),
//   dQ^T += K^T dS^T   (K^T through transposing reads, dS^T straight from the accumulator registers).
template <bool ROWMASK, int D = 64>
__global__ __launch_bounds__(256) void attn_bwd_dq_mfma_kernel(ovqa::AttnBwdArgs a, int W, int G, int nkt) {
  constexpr int PITCH = Img<D>::PITCH, KS = Img<D>::KS, DT = Img<D>::DT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nk = a.nk, nq = a.nq;
  const int q_rows = 32 * W, k_rows = nkt * 32;
  const int prob_bytes = (2 * q_rows + 2 * k_rows) * PITCH + k_rows * 4;  // Q | dO | K | V | mask row
  const int slot = wave / W, tq = wave % W;
  const int q_blk0 = blockIdx.y * q_rows;

  for (int g = 0; g < G; g++) {
    const int64_t pid = (int64_t)blockIdx.x * G + g;
    if (pid >= (int64_t)a.B * a.H) break;
    const int b = (int)(pid / a.H), h = (int)(pid % a.H);
    char* base = smem + g * prob_bytes;
    const int qr = min(q_rows, nq - q_blk0);
    const ImgDesc d[4] = {
        {base, (const bf16*)a.q + ((int64_t)b * nq + q_blk0) * a.ldq + h * D, a.ldq, qr, q_rows},
        {base + q_rows * PITCH, (const bf16*)a.d_o + ((int64_t)b * nq + q_blk0) * a.lddo + h * D, a.lddo, qr, q_rows},
        {base + 2 * q_rows * PITCH, (const bf16*)a.k + (int64_t)b * nk * a.ldk + h * D, a.ldk, nk, k_rows},
        {base + (2 * q_rows + k_rows) * PITCH, (const bf16*)a.v + (int64_t)b * nk * a.ldv + h * D, a.ldv, nk, k_rows}};
    load_images<4, D>(d, tid);
    if (a.msq == 0)
      load_mask_row(reinterpret_cast<float*>(base + (2 * q_rows + 2 * k_rows) * PITCH),
                    a.mask ? a.mask + (int64_t)b * a.msb + (int64_t)h * a.msh : nullptr, nk, k_rows, tid);
  }
  __syncthreads();

  const int64_t pid = (int64_t)blockIdx.x * G + slot;
  if (slot >= G || pid >= (int64_t)a.B * a.H) return;
  const int b = (int)(pid / a.H), h = (int)(pid % a.H);
  const int q0 = q_blk0 + tq * 32;
  if (q0 >= nq) return;
  const char* Qs = smem + slot * prob_bytes;
  const char* Gs = Qs + q_rows * PITCH;
  const char* Ks = Gs + q_rows * PITCH;
  const char* Vs = Ks + k_rows * PITCH;
  const float* mlds = reinterpret_cast<const float*>(Vs + k_rows * PITCH);
  constexpr bool row_mask = ROWMASK;  // compile-time: the common key-padding form carries no per-element checks

  const int q = q0 + (lane & 31);
  const bool qok = q < nq;
  const int qc = qok ? q : nq - 1;
  // delta = dO . O over this lane's half of the D features, combined with the partner half-wave
  float delta = 0.f;
  {
    const bf16* gr = (const bf16*)a.d_o + ((int64_t)b * nq + qc) * a.lddo + h * D + (D / 2) * (lane >> 5);
    const bf16* orow = (const bf16*)a.o + ((int64_t)b * nq + qc) * a.ldo + h * D + (D / 2) * (lane >> 5);
    const bf16* lrow = a.o_lo ? (const bf16*)a.o_lo + ((int64_t)b * nq + qc) * a.ldo + h * D + (D / 2) * (lane >> 5) : nullptr;
#pragma unroll
    for (int c = 0; c < D / 16; c++) {
      const bf16x8 g8 = *reinterpret_cast<const bf16x8*>(gr + 8 * c);
      const bf16x4 oa = *reinterpret_cast<const bf16x4*>(orow + 8 * c);
      const bf16x4 ob = *reinterpret_cast<const bf16x4*>(orow + 8 * c + 4);
#pragma unroll
      for (int e = 0; e < 4; e++) delta += (float)g8[e] * (float)oa[e] + (float)g8[4 + e] * (float)ob[e];
      if (lrow) {
        const bf16x4 la = *reinterpret_cast<const bf16x4*>(lrow + 8 * c);
        const bf16x4 lb = *reinterpret_cast<const bf16x4*>(lrow + 8 * c + 4);
#pragma unroll
        for (int e = 0; e < 4; e++) delta += (float)g8[e] * (float)la[e] + (float)g8[4 + e] * (float)lb[e];
      }
    }
    delta += __shfl_xor(delta, 32, 64);
  }
  const float lse = a.lse[((int64_t)b * a.H + h) * nq + qc];
  if (qok && lane < 32) a.delta[((int64_t)b * a.H + h) * nq + q] = delta;
  const float* mrow = a.mask ? a.mask + (int64_t)b * a.msb + (int64_t)h * a.msh + (int64_t)qc * a.msq : nullptr;

  bf16x8 qf[KS], gf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ks++) {
    qf[ks] = frag_rows<D>(Qs, tq * 32, ks, lane);
    gf[ks] = frag_rows<D>(Gs, tq * 32, ks, lane);
  }
  f32x16 dqt[DT];
#pragma unroll
  for (int d = 0; d < DT; d++)
#pragma unroll
    for (int r = 0; r < 16; r++) dqt[d][r] = 0.f;

  for (int t = 0; t < nkt; t++) {
    f32x16 st, dp;
#pragma unroll
    for (int r = 0; r < 16; r++) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<D>(Ks, t * 32, ks, lane), qf[ks], st, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<D>(Vs, t * 32, ks, lane), gf[ks], dp, 0, 0, 0);
    }
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int key = t * 32 + acc_row(r, lane);
      float p;
      if (row_mask) {
        p = exp2_fast((st[r] * a.scale + mlds[key] - lse) * LOG2E);  // -inf beyond nk -> 0
      } else {
        p = 0.f;
        if (key < nk) p = exp2_fast((st[r] * a.scale + (mrow ? mrow[key] : 0.f) - lse) * LOG2E);
      }
      ds[r] = p * (dp[r] - delta);
    }
#pragma unroll
    for (int s = 0; s < 2; s++) {
      bf16x8 db;
#pragma unroll
      for (int j = 0; j < 8; j++) db[j] = (bf16)ds[8 * s + j];
#pragma unroll
      for (int d = 0; d < DT; d++)
        dqt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(Ks, t * 32 + 16 * s, d * 32, lane), db, dqt[d], 0, 0, 0);
    }
  }
  if (qok) {
    bf16* drow = (bf16*)a.dq + ((int64_t)b * nq + q) * a.lddq + h * D;
#pragma unroll
    for (int d = 0; d < DT; d++)
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        bf16x4 o4;
#pragma unroll
        for (int e = 0; e < 4; e++) o4[e] = (bf16)(dqt[d][4 * g4 + e] * a.scale);
        *reinterpret_cast<bf16x4*>(drow + d * 32 + 8 * g4 + 4 * (lane >> 5)) = o4;
      }
  }
}

// Kernel B: one wave = 32 keys (columns).  For every query tile (rows):
//   S = Q K^T, dP = dO V^T, P = exp(S*scale + mask - lse_row), dS = P (dP - delta_row),
//   dV^T += dO^T P ,  dK^T += Q^T dS    (dO^T / Q^T through transposing reads).
template <bool ROWMASK, int D = 64>
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma_kernel(ovqa::AttnBwdArgs a, int W, int G, int nqt) {
  constexpr int PITCH = Img<D>::PITCH, KS = Img<D>::KS, DT = Img<D>::DT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nk = a.nk, nq = a.nq;
  const int k_rows = 32 * W, q_rows = nqt * 32;
  const int prob_bytes = (2 * q_rows + 2 * k_rows) * PITCH + 2 * q_rows * 4;  // Q | dO | K | V | lse | delta
  const int slot = wave / W, tk = wave % W;
  const int k_blk0 = blockIdx.y * k_rows;

  for (int g = 0; g < G; g++) {
    const int64_t pid = (int64_t)blockIdx.x * G + g;
    if (pid >= (int64_t)a.B * a.H) break;
    const int b = (int)(pid / a.H), h = (int)(pid % a.H);
    char* base = smem + g * prob_bytes;
    const int kr = min(k_rows, nk - k_blk0);
    const ImgDesc d[4] = {
        {base, (const bf16*)a.q + (int64_t)b * nq * a.ldq + h * D, a.ldq, nq, q_rows},
        {base + q_rows * PITCH, (const bf16*)a.d_o + (int64_t)b * nq * a.lddo + h * D, a.lddo, nq, q_rows},
        {base + 2 * q_rows * PITCH, (const bf16*)a.k + ((int64_t)b * nk + k_blk0) * a.ldk + h * D, a.ldk, kr, k_rows},
        {base + (2 * q_rows + k_rows) * PITCH, (const bf16*)a.v + ((int64_t)b * nk + k_blk0) * a.ldv + h * D, a.ldv, kr,
         k_rows}};
    load_images<4, D>(d, tid);
    float* ls = reinterpret_cast<float*>(base + (2 * q_rows + 2 * k_rows) * PITCH);
    for (int i = tid; i < q_rows; i += 256) {
      ls[i] = i < nq ? a.lse[((int64_t)b * a.H + h) * nq + i] : 0.f;
      ls[q_rows + i] = i < nq ? a.delta[((int64_t)b * a.H + h) * nq + i] : 0.f;
    }
  }
  __syncthreads();

  const int64_t pid = (int64_t)blockIdx.x * G + slot;
  if (slot >= G || pid >= (int64_t)a.B * a.H) return;
  const int b = (int)(pid / a.H), h = (int)(pid % a.H);
  const int k0 = k_blk0 + tk * 32;
  if (k0 >= nk) return;
  const char* Qs = smem + slot * prob_bytes;
  const char* Gs = Qs + q_rows * PITCH;
  const char* Ks = Gs + q_rows * PITCH;
  const char* Vs = Ks + k_rows * PITCH;
  const float* ls = reinterpret_cast<const float*>(Vs + k_rows * PITCH);

  const int key = k0 + (lane & 31);
  const bool kok = key < nk;
  const float* mcol = a.mask ? a.mask + (int64_t)b * a.msb + (int64_t)h * a.msh + (kok ? key : 0) : nullptr;
  constexpr bool row_mask = ROWMASK;  // key-padding mask: one value per key column, i.e. per lane
  const float mconst = (row_mask && mcol) ? mcol[0] : 0.f;

  bf16x8 kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ks++) {
    kf[ks] = frag_rows<D>(Ks, tk * 32, ks, lane);
    vf[ks] = frag_rows<D>(Vs, tk * 32, ks, lane);
  }
  f32x16 dvt[DT], dkt[DT];
#pragma unroll
  for (int d = 0; d < DT; d++)
#pragma unroll
    for (int r = 0; r < 16; r++) { dvt[d][r] = 0.f; dkt[d][r] = 0.f; }

  for (int t = 0; t < nqt; t++) {
    f32x16 s_, dp;
#pragma unroll
    for (int r = 0; r < 16; r++) { s_[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      s_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<D>(Qs, t * 32, ks, lane), kf[ks], s_, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<D>(Gs, t * 32, ks, lane), vf[ks], dp, 0, 0, 0);
    }
    float p[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int qi = t * 32 + acc_row(r, lane);
      float pv = 0.f;
      if (qi < nq && kok) {
        const float mv = row_mask ? mconst : (mcol ? mcol[(int64_t)qi * a.msq] : 0.f);
        pv = exp2_fast((s_[r] * a.scale + mv - ls[qi]) * LOG2E);
      }
      p[r] = pv;
      ds[r] = pv * (dp[r] - ls[q_rows + qi]);
    }
#pragma unroll
    for (int s = 0; s < 2; s++) {
      bf16x8 pb, db;
#pragma unroll
      for (int j = 0; j < 8; j++) { pb[j] = (bf16)p[8 * s + j]; db[j] = (bf16)ds[8 * s + j]; }
#pragma unroll
      for (int d = 0; d < DT; d++) {
        dvt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(Gs, t * 32 + 16 * s, d * 32, lane), pb, dvt[d], 0, 0, 0);
        dkt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(Qs, t * 32 + 16 * s, d * 32, lane), db, dkt[d], 0, 0, 0);
      }
    }
  }
  if (kok) {
    bf16* dkrow = (bf16*)a.dk_ + ((int64_t)b * nk + key) * a.lddk + h * D;
    bf16* dvrow = (bf16*)a.dv_ + ((int64_t)b * nk + key) * a.lddv + h * D;
#pragma unroll
    for (int d = 0; d < DT; d++)
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        bf16x4 k4, v4;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          k4[e] = (bf16)(dkt[d][4 * g4 + e] * a.scale);
          v4[e] = (bf16)dvt[d][4 * g4 + e];
        }
        *reinterpret_cast<bf16x4*>(dkrow + d * 32 + 8 * g4 + 4 * (lane >> 5)) = k4;
        *reinterpret_cast<bf16x4*>(dvrow + d * 32 + 8 * g4 + 4 * (lane >> 5)) = v4;
      }
  }
}


// ------------------------------------------------------------------------------ merged backward, n_k <= 32
// Question self-attention (20 x 20) and guided attention (100 queries x 20 question tokens): one key tile, and
// a workgroup holds ALL queries of its problems, so dQ, dK and dV come out of ONE launch (the two-kernel form
// costs a second ~5 us dependent launch and stages Q / dO / K / V twice).  One wave = one 32-query tile (with a
// single tile per problem the two orientations are split over two waves, see `both` below):
//   transposed orientation (lane = query):  S^T, dP^T -> dS^T -> dQ^T += K^T dS^T            (as kernel A)
//   direct orientation     (lane = key):    S, dP -> P, dS -> dV^T += dO^T P, dK^T += Q^T dS  (as kernel B)
// W (query tiles per problem: 1, 2 or 4) is a template parameter: G = 4 / W problems are packed per workgroup, and all
// staging index arithmetic folds to shifts.  These kernels are latency chains (launch -> loads -> ~20 MFMAs -> stores),
// so instruction count matters like nowhere else: a wave64 VALU instruction is 4 cycles, 600 instructions are 1 us.
// The compute half of the merged backward (everything after the staging barrier): reads the images Q | dO | K | V, the
// mask row, lse and delta of its problem from LDS.  `pid_base` = first problem of the workgroup; waves with `extra`
// (a launch wider than this kernel's own NT threads: the fused dO-projection form) only keep the barriers company.
template <bool ROWMASK, int W, int G>
__device__ __forceinline__ void smallk_bwd_compute(const ovqa::AttnBwdArgs& a, char* smem, int pid_base, int wave, int lane,
                                                   bool extra) {
  const int nk = a.nk, nq = a.nq;
  constexpr int q_rows = 32 * W, k_rows = 32;
  constexpr int img_bytes = (2 * q_rows + 2 * k_rows) * 128;
  constexpr int prob_bytes = img_bytes + k_rows * 4 + 2 * q_rows * 4 + 4096 * 4;
  constexpr bool both = W > 1;
  const int role = W == 1 ? wave / G : 0, w4 = W == 1 ? wave % G : wave & 3;
  const int slot = w4 / W, tq = w4 % W;
  const int pidw = pid_base + slot;
  const bool prob_ok = !extra && slot < G && pidw < a.B * a.H;
  const bool active = prob_ok && tq * 32 < nq;  // this wave's query tile exists
  const int b = prob_ok ? pidw / a.H : 0, h = prob_ok ? pidw - (pidw / a.H) * a.H : 0;
  const char* Qs = smem + (prob_ok ? slot : 0) * prob_bytes;
  const char* Gs = Qs + q_rows * 128;
  const char* Ks = Gs + q_rows * 128;
  const char* Vs = Ks + k_rows * 128;
  float* mlds = reinterpret_cast<float*>(const_cast<char*>(Vs) + k_rows * 128);
  float* lse_s = mlds + k_rows;
  float* del_s = lse_s + q_rows;
  char* xch = reinterpret_cast<char*>(del_s + q_rows);  // [tile][P | dS][lane][16 bf16]: 4 KiB per query tile
  constexpr bool row_mask = ROWMASK;  // compile-time: the common key-padding form carries no per-element checks

  if (active) {
    const int q = tq * 32 + (lane & 31);
    const bool qok = q < nq;
    const int qc = qok ? q : nq - 1;
    const float lse2 = lse_s[qc], delta = del_s[qc];  // (mask row and lse are staged in log2 units)
    const float scale2 = a.scale * LOG2E;
    const float* mrow = a.mask ? a.mask + (int64_t)b * a.msb + (int64_t)h * a.msh + (int64_t)qc * a.msq : nullptr;

    bf16x8 qf[4], gf[4], kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      qf[ks] = frag_rows(Qs, tq * 32, ks, lane);
      gf[ks] = frag_rows(Gs, tq * 32, ks, lane);
      kf[ks] = frag_rows(Ks, 0, ks, lane);
      vf[ks] = frag_rows(Vs, 0, ks, lane);
    }
    // ---- transposed orientation: dQ (role 0)
    if (both || role == 0) {
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; r++) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ks], gf[ks], dp, 0, 0, 0);
      }
      float ds[16];
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const int key0 = 8 * g4 + 4 * (lane >> 5);  // acc_row(4 * g4 + e, lane) = key0 + e
        float mm[4];
        if (row_mask) {
          const float4 m4 = *reinterpret_cast<const float4*>(mlds + key0);
          mm[0] = m4.x - lse2; mm[1] = m4.y - lse2; mm[2] = m4.z - lse2; mm[3] = m4.w - lse2;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int r = 4 * g4 + e;
          float p;
          if (row_mask) {
            p = exp2_fast(st[r] * scale2 + mm[e]);  // -inf beyond nk -> 0
          } else {
            p = 0.f;
            if (key0 + e < nk) p = exp2_fast(st[r] * scale2 + ((mrow ? mrow[key0 + e] : 0.f) * LOG2E - lse2));
          }
          ds[r] = p * (dp[r] - delta);
        }
      }
      f32x16 dqt[2];
#pragma unroll
      for (int d = 0; d < 2; d++)
#pragma unroll
        for (int r = 0; r < 16; r++) dqt[d][r] = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        bf16x8 db;
#pragma unroll
        for (int j = 0; j < 8; j++) db[j] = (bf16)ds[8 * s2 + j];
#pragma unroll
        for (int d = 0; d < 2; d++)
          dqt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Ks, 16 * s2, d * 32, lane), db, dqt[d], 0, 0, 0);
      }
      if (qok) {
        bf16* drow = (bf16*)a.dq + ((int64_t)b * nq + q) * a.lddq + h * 64;
#pragma unroll
        for (int d = 0; d < 2; d++)
#pragma unroll
          for (int g4 = 0; g4 < 4; g4++) {
            bf16x4 o4;
#pragma unroll
            for (int e = 0; e < 4; e++) o4[e] = (bf16)(dqt[d][4 * g4 + e] * a.scale);
            *reinterpret_cast<bf16x4*>(drow + d * 32 + 8 * g4 + 4 * (lane >> 5)) = o4;
          }
      }
    }
    OVQA_PROBE(4);
    // ---- direct orientation (role 1): P and dS of this query tile, rows = queries, lane = key
    if (both || role == 1) {
      const int key = lane & 31;
      const bool kok = key < nk;
      const float* mcol = a.mask ? a.mask + (int64_t)b * a.msb + (int64_t)h * a.msh + (kok ? key : 0) : nullptr;
      const float mconst = row_mask ? mlds[key] : 0.f;  // log2 units; -inf beyond nk
      f32x16 s_, dp;
#pragma unroll
      for (int r = 0; r < 16; r++) { s_[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        s_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[ks], kf[ks], s_, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf[ks], vf[ks], dp, 0, 0, 0);
      }
      bf16x8 pb[2], db[2];  // the B operands (k = query) of dV^T += dO^T P and dK^T += Q^T dS
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const int q0 = tq * 32 + 8 * g4 + 4 * (lane >> 5);  // acc_row(4 * g4 + e, lane) = q0 - tq * 32 + e
        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + q0);  // +inf beyond nq: p = 0
        const float4 d4 = *reinterpret_cast<const float4*>(del_s + q0);
        const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int r = 4 * g4 + e;
          float pv;
          if (row_mask) {
            pv = exp2_fast(s_[r] * scale2 + (mconst - ll[e]));
          } else {
            pv = 0.f;
            if (q0 + e < nq && kok)
              pv = exp2_fast(s_[r] * scale2 + ((mcol ? mcol[(int64_t)(q0 + e) * a.msq] : 0.f) * LOG2E - ll[e]));
          }
          pb[r >> 3][r & 7] = (bf16)pv;
          db[r >> 3][r & 7] = (bf16)(pv * (dp[r] - dd[e]));
        }
      }
      if constexpr (both) {
        // several query tiles per problem: hand P / dS to the waves that own the output accumulators (below)
        bf16x8* px = reinterpret_cast<bf16x8*>(xch + ((tq * 2 + 0) * 64 + lane) * 32);
        bf16x8* dx = reinterpret_cast<bf16x8*>(xch + ((tq * 2 + 1) * 64 + lane) * 32);
        px[0] = pb[0]; px[1] = pb[1];
        dx[0] = db[0]; dx[1] = db[1];
      } else {
        // one query tile: this wave has the complete dK^T / dV^T
        f32x16 dvt[2], dkt[2];
#pragma unroll
        for (int d = 0; d < 2; d++)
#pragma unroll
          for (int r = 0; r < 16; r++) { dvt[d][r] = 0.f; dkt[d][r] = 0.f; }
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
          for (int d = 0; d < 2; d++) {
            dvt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Gs, 16 * s2, d * 32, lane), pb[s2], dvt[d], 0, 0, 0);
            dkt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Qs, 16 * s2, d * 32, lane), db[s2], dkt[d], 0, 0, 0);
          }
        OVQA_PROBE(5);
        if (kok) {
          bf16* dkrow = (bf16*)a.dk_ + ((int64_t)b * nk + key) * a.lddk + h * 64;
          bf16* dvrow = (bf16*)a.dv_ + ((int64_t)b * nk + key) * a.lddv + h * 64;
#pragma unroll
          for (int d = 0; d < 2; d++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
              bf16x4 k4, v4;
#pragma unroll
              for (int e = 0; e < 4; e++) {
                v4[e] = (bf16)dvt[d][4 * g4 + e];
                k4[e] = (bf16)(dkt[d][4 * g4 + e] * a.scale);
              }
              *reinterpret_cast<bf16x4*>(dkrow + d * 32 + 8 * g4 + 4 * (lane >> 5)) = k4;
              *reinterpret_cast<bf16x4*>(dvrow + d * 32 + 8 * g4 + 4 * (lane >> 5)) = v4;
            }
        }
      }
    }
  }
  if constexpr (both) {
    // dK^T / dV^T: four [32 keys x 32 features] accumulators per problem (dV features 0-31, 32-63, dK likewise), dealt
    // over the problem's W waves; each sums over ALL query tiles with the P / dS operands the tiles' waves left in LDS
    // -- one barrier and 4 KiB per tile instead of a W-phase fp32 reduction of the accumulators.
    OVQA_PROBE(5);
    __syncthreads();
    OVQA_PROBE(6);
    if (prob_ok) {
      const int key = lane & 31;
      for (int acc = tq; acc < 4; acc += W) {
        const bool is_k = acc >= 2;
        const int d = acc & 1;
        const char* Xs = is_k ? Qs : Gs;
        f32x16 out;
#pragma unroll
        for (int r = 0; r < 16; r++) out[r] = 0.f;
        for (int t = 0; t < W && t * 32 < nq; t++) {
          const bf16x8* src = reinterpret_cast<const bf16x8*>(xch + ((t * 2 + (is_k ? 1 : 0)) * 64 + lane) * 32);
#pragma unroll
          for (int s2 = 0; s2 < 2; s2++)
            out = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Xs, t * 32 + 16 * s2, d * 32, lane), src[s2], out, 0, 0, 0);
        }
        if (key < nk) {
          bf16* row = (is_k ? (bf16*)a.dk_ + ((int64_t)b * nk + key) * a.lddk : (bf16*)a.dv_ + ((int64_t)b * nk + key) * a.lddv) + h * 64;
          const float sc = is_k ? a.scale : 1.f;
#pragma unroll
          for (int g4 = 0; g4 < 4; g4++) {
            bf16x4 o4;
#pragma unroll
            for (int e = 0; e < 4; e++) o4[e] = (bf16)(out[4 * g4 + e] * sc);
            *reinterpret_cast<bf16x4*>(row + d * 32 + 8 * g4 + 4 * (lane >> 5)) = o4;
          }
        }
      }
    }
  }
  OVQA_PROBE(7);
}

template <bool ROWMASK, int W, int G = 4 / W>
__global__ __launch_bounds__(W == 1 ? 128 * G : 256) void attn_bwd_smallk_mfma_kernel(ovqa::AttnBwdArgs a) {
  static_assert(W * G == 4 || (W == 1 && G == 2), "problems per workgroup: 4 / W, or 2 single-tile problems");
  constexpr int NT = W == 1 ? 128 * G : 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nk = a.nk, nq = a.nq;
  constexpr int q_rows = 32 * W, k_rows = 32;
  constexpr int img_bytes = (2 * q_rows + 2 * k_rows) * 128;                       // Q | dO | K | V
  constexpr int prob_bytes = img_bytes + k_rows * 4 + 2 * q_rows * 4 + 4096 * 4;   // + mask row | lse | delta | P,dS
  // 512 threads (W == 1: the 20 x 20 question attention): waves 0-3 take the dQ role, waves 4-7 the dK/dV role of
  // the same 4 packed problems -- half the dependent chain per wave.  256 threads (W >= 2): every wave does both
  // (with 4 query tiles per problem the doubled wave count only adds VALU contention: 14.8 vs 18.3 us).
  // (W == 1, G == 2: the same role split with 2 + 2 waves -- 256 instead of 128 workgroups for 64 samples x 8 heads)
  constexpr bool stage_all = NT == 256;  // every thread stages a chunk of every image
  OVQA_PROBE(0);

  // ---- staging: ONE round of global loads for everything the workgroup needs (all loads of all G problems are
  // issued before the first LDS store).  Q / dO: W 16-byte chunks per thread and problem, K / V: one.  The thread that
  // owns a dO chunk also fetches the matching chunk of O: delta = dO . O falls out of the staging (8 consecutive lanes
  // hold a row) instead of a second pass over dO and O behind the first one; mask row and log-sum-exp ride along.
  // W == 1: threads 0-255 stage Q and K, threads 256-511 dO (+ O) and V.
  {
    const int pid0 = (int)blockIdx.x * G, nprob = a.B * a.H;
    const bool ld_q = stage_all || tid < 256, ld_d = stage_all || tid >= 256;  // wave-uniform
    const int t = tid & 255;
    const int ch = t & 7;
    float mval = 0.f;  // mask row entries: thread -> (problem tid / 32, key tid % 32)
    const bool mtask = a.msq == 0 && tid < G * k_rows && pid0 + tid / k_rows < nprob;
    if (mtask) {
      const int pid = pid0 + tid / k_rows;
      const int key = tid % k_rows;
      const int mb = pid / a.H, mh = pid - mb * a.H;
      mval = key < nk ? (a.mask ? a.mask[(int64_t)mb * a.msb + (int64_t)mh * a.msh + key] * LOG2E : 0.f) : -INFINITY;
    }
    uint4 vq[G][W], vd[G][W], vo[G][W], vl[G][W], vk[G], vv[G];
    float lse_r[G][W];
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int g = 0; g < G; g++) {
      const int pid = pid0 + g;  // uniform
      const bool ok = pid < nprob;
      const int b = ok ? pid / a.H : 0, h = ok ? pid - (pid / a.H) * a.H : 0;
      const bf16* qb = (const bf16*)a.q + (int64_t)b * nq * a.ldq + h * 64 + ch * 8;
      const bf16* gb = (const bf16*)a.d_o + (int64_t)b * nq * a.lddo + h * 64 + ch * 8;
      const bf16* ob = (const bf16*)a.o + (int64_t)b * nq * a.ldo + h * 64 + ch * 8;
      const bf16* lob = a.o_lo ? (const bf16*)a.o_lo + (int64_t)b * nq * a.ldo + h * 64 + ch * 8 : nullptr;
      const float* lb = a.lse + ((int64_t)b * a.H + h) * nq;
#pragma unroll
      for (int i = 0; i < W; i++) {
        const int row = (t >> 3) + 32 * i;
        const bool v = ok && row < nq;
        vq[g][i] = zero4; vd[g][i] = zero4; vo[g][i] = zero4; vl[g][i] = zero4; lse_r[g][i] = INFINITY;  // p = 0 beyond nq
        if (ld_q && v) vq[g][i] = *reinterpret_cast<const uint4*>(qb + (int64_t)row * a.ldq);
        if (ld_d && v) {
          vd[g][i] = *reinterpret_cast<const uint4*>(gb + (int64_t)row * a.lddo);
          vo[g][i] = *reinterpret_cast<const uint4*>(ob + (int64_t)row * a.ldo);
          if (lob) vl[g][i] = *reinterpret_cast<const uint4*>(lob + (int64_t)row * a.ldo);
          if (ch == 0) lse_r[g][i] = lb[row] * LOG2E;
        }
      }
      const int krow = t >> 3;
      const bool kv = ok && krow < nk;
      vk[g] = zero4; vv[g] = zero4;
      if (ld_q && kv) vk[g] = *reinterpret_cast<const uint4*>((const bf16*)a.k + ((int64_t)b * nk + krow) * a.ldk + h * 64 + ch * 8);
      if (ld_d && kv) vv[g] = *reinterpret_cast<const uint4*>((const bf16*)a.v + ((int64_t)b * nk + krow) * a.ldv + h * 64 + ch * 8);
    }
    OVQA_PROBE(1);
#pragma unroll
    for (int g = 0; g < G; g++) {
      char* base = smem + g * prob_bytes;
      float* lse_g = reinterpret_cast<float*>(base + img_bytes) + k_rows;
      const int pid = pid0 + g;
#pragma unroll
      for (int i = 0; i < W; i++) {
        const int row = (t >> 3) + 32 * i;
        if (ld_q) *reinterpret_cast<uint4*>(base + img_off(row, ch)) = vq[g][i];
        if (ld_d) {
          *reinterpret_cast<uint4*>(base + q_rows * 128 + img_off(row, ch)) = vd[g][i];
          const bf16x8 g8 = *reinterpret_cast<const bf16x8*>(&vd[g][i]);
          const bf16x8 o8 = *reinterpret_cast<const bf16x8*>(&vo[g][i]);
          const bf16x8 l8 = *reinterpret_cast<const bf16x8*>(&vl[g][i]);
          float dl = 0.f;
#pragma unroll
          for (int e = 0; e < 8; e++) dl += (float)g8[e] * ((float)o8[e] + (float)l8[e]);
          dl += __shfl_xor(dl, 1, 64);
          dl += __shfl_xor(dl, 2, 64);
          dl += __shfl_xor(dl, 4, 64);
          if (ch == 0) {
            lse_g[row] = lse_r[g][i];
            lse_g[q_rows + row] = dl;
            if (row < nq && pid < nprob) a.delta[(int64_t)pid * nq + row] = dl;
          }
        }
      }
      const int krow = t >> 3;
      if (ld_q) *reinterpret_cast<uint4*>(base + 2 * q_rows * 128 + img_off(krow, ch)) = vk[g];
      if (ld_d) *reinterpret_cast<uint4*>(base + (2 * q_rows + k_rows) * 128 + img_off(krow, ch)) = vv[g];
    }
    if (mtask) reinterpret_cast<float*>(smem + (tid / k_rows) * prob_bytes + img_bytes)[tid % k_rows] = mval;
  }
  OVQA_PROBE(2);
  __syncthreads();
  OVQA_PROBE(3);

  smallk_bwd_compute<ROWMASK, W, G>(a, smem, (int)blockIdx.x * G, wave, lane, false);
}


// ------------------------------------------------ merged backward with the dO projection inside (round 3)
// Guided attention backward (65-128 queries x <= 32 keys): dO of the head is NOT read from HBM -- it is computed here
// from the gradient of the block's pre-LayerNorm sum, dO_h = dY W_o[:, head]  (the fc_o dX product restricted to the
// head's 64 features: dY [128 rows, d_model] against 64 rows of the transposed weight shadow), with the projection loop
// of attn_q_fwd_mfma_kernel; rounded to bf16 into the dO image, as the separate GEMM would have stored it.  Then Q, K, V,
// the mask row and log-sum-exp are staged, delta = dO . (O + o_lo) is taken from the image and the O rows, and
// smallk_bwd_compute runs on waves 0-3.  One launch instead of a 6400 x 512 <- 512 GEMM + the backward kernel; dO never
// travels through HBM.
struct DoBwdArgs {
  const bf16* dy; int64_t lddy;     // [B * nq, Dm]: gradient w.r.t. the fc_o output
  const bf16* wt; int64_t ldwt;     // transposed fc_o weights [H * 64, Dm]: row = head-major input feature of fc_o
  ovqa::AttnBwdArgs att;            // q, k, v, o, o_lo, lse, mask, dq, dk, dv, delta, sizes (d_o unused)
  int Dm;
};

// NH = heads per workgroup (1: 8 waves, two workgroups per CU; 2: 16 waves, dY staged once for both heads, one workgroup
// per CU and 64 samples x 4 head pairs = one workgroup on every CU).
template <int NBUF, int NH>
__global__ __launch_bounds__(512 * NH, NH == 1 ? 4 : 4) void attn_bwd_do_smallk_mfma_kernel(DoBwdArgs g) {
  constexpr int RP = 128, BKF = 64, NI = 2, NJ = 2, W = 4, NW = 8 * NH;
  constexpr int PROWS = 8, CHR = 8, WCH = 64 * NH / PROWS, XCH = RP / PROWS, PER = (WCH + XCH) / NW;
  constexpr int STAGE = (WCH + XCH) * 1024;
  constexpr int q_rows = 32 * W, k_rows = 32;
  constexpr int img_bytes = (2 * q_rows + 2 * k_rows) * 128;
  constexpr int prob_bytes = img_bytes + k_rows * 4 + 2 * q_rows * 4 + 4096 * 4;  // (smallk_bwd_compute's layout)
  static_assert((WCH + XCH) % NW == 0, "every wave stages the same number of pieces");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ovqa::AttnBwdArgs& a = g.att;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave / (2 * NH), wr = wave % (2 * NH);
  const int h0 = blockIdx.y * NH, b = blockIdx.x, nq = a.nq, nk = a.nk;

  f32x4 acc[NJ][NI];
#pragma unroll
  for (int j = 0; j < NJ; j++)
#pragma unroll
    for (int i = 0; i < NI; i++) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto off32 = [&](int row, int ch) { return row * (BKF * 2) + ((ch ^ (row & 7)) << 4); };
  const bf16* src[PER];
#pragma unroll
  for (int i = 0; i < PER; i++) {
    const int ci = wave + NW * i;
    const int prow = lane / CHR, pch = lane % CHR;
    if (ci < WCH) {
      const int row = ci * PROWS + prow;
      src[i] = g.wt + (int64_t)(h0 * 64 + perm32(row)) * g.ldwt + ((pch ^ (row & 7)) << 3);
    } else {
      const int row = (ci - WCH) * PROWS + prow;
      const int r = row < nq ? row : nq - 1;
      src[i] = g.dy + ((int64_t)b * nq + r) * g.lddy + ((pch ^ (row & 7)) << 3);
    }
  }
  auto issue = [&](int kt) {
    char* buf = smem + (kt % NBUF) * STAGE;
#pragma unroll
    for (int i = 0; i < PER; i++)
      __builtin_amdgcn_global_load_lds((gbl_void*)(src[i] + kt * BKF), (lds_void*)(buf + (wave + NW * i) * 1024), 16, 0, 0);
  };
  const int nkt = g.Dm / BKF;
#pragma unroll
  for (int p = 0; p < NBUF - 1; p++)
    if (p < nkt) issue(p);
  for (int kt = 0; kt < nkt; kt++) {
    if (kt + NBUF - 2 >= nkt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * (NBUF - 2)) : "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (kt + NBUF - 1 < nkt) issue(kt + NBUF - 1);
    const char* Ws = smem + (kt % NBUF) * STAGE;
    const char* Xs = Ws + WCH * 1024;
#pragma unroll
    for (int ks = 0; ks < BKF / 32; ks++) {
      bf16x8 pf[NJ], qf[NI];
#pragma unroll
      for (int j = 0; j < NJ; j++)
        pf[j] = *reinterpret_cast<const bf16x8*>(Ws + off32(wr * 32 + j * 16 + (lane & 15), ks * 4 + (lane >> 4)));
#pragma unroll
      for (int i = 0; i < NI; i++)
        qf[i] = *reinterpret_cast<const bf16x8*>(Xs + off32(wc * 32 + i * 16 + (lane & 15), ks * 4 + (lane >> 4)));
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int i = 0; i < NI; i++) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[j], qf[i], acc[j][i], 0, 0, 0);
    }
  }
  // (round 5 measured the requests IN FRONT of the projection loop, as attn_bwd_do_smallk1_mfma_kernel has them: 18.4 -> 19.4 us
  // per launch in the step -- nine register loads per thread of two co-resident workgroups queue ahead of the ring's first
  // tiles on the same fetch path; the role-split form showed the same)
  // ---- everything else the backward needs is requested now (one round of loads, as in the plain kernel); per head 512
  // threads: the first 256 take Q (4 chunks) and K, the other 256 O / o_lo (4 chunks each), V, lse; the mask row by 32
  const int hs = tid >> 9, t5 = tid & 511;   // staging head and thread within its 512
  const int hst = h0 + hs;
  const int t = t5 & 255, ch = t & 7;
  const bool ld_q = __builtin_amdgcn_readfirstlane((int)(t5 < 256)) != 0;  // (whole waves)
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  // Every load is UNCONDITIONAL per lane (clamped row; a missing mask row / o_lo is read from another valid address and
  // dropped below) and nothing is computed from a loaded value before all are issued -- see attn_bwd_roles_mfma_kernel:
  // with per-lane `if`s this staging was seven dependent round trips.  va: the Q rows (Q waves) or the O rows; vb: o_lo.
  const bool has_mask = a.mask != nullptr, has_lo = a.o_lo != nullptr;
  uint4 va[W], vb[W], vkv;
  float lse_r[W];
  const float* lb = a.lse + ((int64_t)b * a.H + hst) * nq;
  const float* mp = has_mask ? a.mask + (int64_t)b * a.msb + (int64_t)hst * a.msh : lb;
  float mval = mp[min(t5 & (k_rows - 1), (has_mask ? nk : nq) - 1)];
  {
    const bf16* pa = ld_q ? (const bf16*)a.q + (int64_t)b * nq * a.ldq + hst * 64 + ch * 8
                          : (const bf16*)a.o + (int64_t)b * nq * a.ldo + hst * 64 + ch * 8;
    const int64_t lda = ld_q ? a.ldq : a.ldo;
    const bool lo = !ld_q && has_lo;
    const bf16* pb = lo ? (const bf16*)a.o_lo + (int64_t)b * nq * a.ldo + hst * 64 + ch * 8 : pa;
#pragma unroll
    for (int i = 0; i < W; i++) {
      const int rc = min((t >> 3) + 32 * i, nq - 1);
      va[i] = *reinterpret_cast<const uint4*>(pa + (int64_t)rc * lda);
      vb[i] = *reinterpret_cast<const uint4*>(pb + (int64_t)rc * lda);
      lse_r[i] = lb[rc];
    }
    const int kc = min(t >> 3, nk - 1);
    const bf16* pk = ld_q ? (const bf16*)a.k + ((int64_t)b * nk + kc) * a.ldk : (const bf16*)a.v + ((int64_t)b * nk + kc) * a.ldv;
    vkv = *reinterpret_cast<const uint4*>(pk + hst * 64 + ch * 8);
  }
  __syncthreads();  // every wave is done with the staging ring: it becomes the images

  // ---- dO image [128][64] (bf16) of the wave's head from the accumulators; rows beyond nq are zero (p = 0 there anyway)
  {
    char* Gs = smem + (wr >> 1) * prob_bytes + q_rows * 128;
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int r = wc * 32 + i * 16 + (lane & 15);
      const int col = (wr & 1) * 32 + (lane >> 4) * 8;
      bf16x8 o8;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        o8[e] = r < nq ? (bf16)acc[0][i][e] : (bf16)0.f;
        o8[4 + e] = r < nq ? (bf16)acc[1][i][e] : (bf16)0.f;
      }
      *reinterpret_cast<bf16x8*>(Gs + img_off(r, col >> 3)) = o8;
    }
  }
  __syncthreads();  // the dO images are complete: delta reads them
  {
    char* base = smem + hs * prob_bytes;
    const char* Gs = base + q_rows * 128;
    float* lse_g = reinterpret_cast<float*>(base + img_bytes) + k_rows;
    const int pid = b * a.H + hst;
#pragma unroll
    for (int i = 0; i < W; i++) {
      const int row = (t >> 3) + 32 * i;
      const bool v = row < nq;
      if (ld_q) {
        if (!v) va[i] = zero4;
        *reinterpret_cast<uint4*>(base + img_off(row, ch)) = va[i];
      } else {
        if (!has_lo) vb[i] = zero4;
        const bf16x8 g8 = *reinterpret_cast<const bf16x8*>(Gs + img_off(row, ch));  // (zero beyond nq: dl = 0 there)
        const bf16x8 o8 = *reinterpret_cast<const bf16x8*>(&va[i]);
        const bf16x8 l8 = *reinterpret_cast<const bf16x8*>(&vb[i]);
        float dl = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) dl += (float)g8[e] * ((float)o8[e] + (float)l8[e]);
        dl += __shfl_xor(dl, 1, 64);
        dl += __shfl_xor(dl, 2, 64);
        dl += __shfl_xor(dl, 4, 64);
        if (ch == 0) {
          lse_g[row] = v ? lse_r[i] * LOG2E : INFINITY;  // p = 0 beyond nq
          lse_g[q_rows + row] = dl;
          if (v && a.delta) a.delta[(int64_t)pid * nq + row] = dl;
        }
      }
    }
    const int krow = t >> 3;
    if (krow >= nk) vkv = zero4;
    if (ld_q) *reinterpret_cast<uint4*>(base + 2 * q_rows * 128 + img_off(krow, ch)) = vkv;
    else *reinterpret_cast<uint4*>(base + (2 * q_rows + k_rows) * 128 + img_off(krow, ch)) = vkv;
    if (t5 < k_rows) reinterpret_cast<float*>(base + img_bytes)[t5] = t5 < nk ? (has_mask ? mval * LOG2E : 0.f) : -INFINITY;
  }
  __syncthreads();
  // waves 0 .. 4 NH - 1: head wave / 4, query tile wave % 4; the others keep the barriers company
  const int hc = wave >> 2;
  smallk_bwd_compute<true, W, 1>(a, smem + (hc < NH ? hc : 0) * prob_bytes, b * a.H + h0 + (hc < NH ? hc : 0), wave & 3, lane,
                                 hc >= NH);
}

// The same for <= 32 queries x <= 32 keys (the question self-attention, 20 x 20): a 4-wave workgroup takes TWO heads of a
// sample -- dO of both from dY [32 rows (nq, padded), d_model] against their 128 rows of the transposed fc_o weights --
// and runs the single-tile merged backward (waves 0-1: dQ of the two heads, waves 2-3: dK / dV).
template <int NBUF>
__global__ __launch_bounds__(256) void attn_bwd_do_smallk1_mfma_kernel(DoBwdArgs g) {
  constexpr int BKF = 64, NI = 2, NJ = 2, G = 2, NW = 4;
  constexpr int PROWS = 8, CHR = 8, WCH = 128 / PROWS, XCH = 32 / PROWS, PER = (WCH + XCH) / NW;
  constexpr int STAGE = (WCH + XCH) * 1024;
  constexpr int q_rows = 32, k_rows = 32;
  constexpr int img_bytes = (2 * q_rows + 2 * k_rows) * 128;
  constexpr int prob_bytes = img_bytes + k_rows * 4 + 2 * q_rows * 4 + 4096 * 4;  // (smallk_bwd_compute's layout)
  static_assert((WCH + XCH) % NW == 0, "every wave stages the same number of pieces");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ovqa::AttnBwdArgs& a = g.att;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave;  // feature slab of 32: head wr >> 1
  const int h0 = blockIdx.y * G, b = blockIdx.x, nq = a.nq, nk = a.nk;

  OVQA_PROBE(0);
  f32x4 acc[NJ][NI];
#pragma unroll
  for (int j = 0; j < NJ; j++)
#pragma unroll
    for (int i = 0; i < NI; i++) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto off32 = [&](int row, int ch) { return row * (BKF * 2) + ((ch ^ (row & 7)) << 4); };
  const bf16* src[PER];
#pragma unroll
  for (int i = 0; i < PER; i++) {
    const int ci = wave + NW * i;
    const int prow = lane / CHR, pch = lane % CHR;
    if (ci < WCH) {
      const int row = ci * PROWS + prow;
      src[i] = g.wt + (int64_t)(h0 * 64 + perm32(row)) * g.ldwt + ((pch ^ (row & 7)) << 3);
    } else {
      const int row = (ci - WCH) * PROWS + prow;
      const int r = row < nq ? row : nq - 1;
      src[i] = g.dy + ((int64_t)b * nq + r) * g.lddy + ((pch ^ (row & 7)) << 3);
    }
  }
  auto issue = [&](int kt) {
    char* buf = smem + (kt % NBUF) * STAGE;
#pragma unroll
    for (int i = 0; i < PER; i++)
      __builtin_amdgcn_global_load_lds((gbl_void*)(src[i] + kt * BKF), (lds_void*)(buf + (wave + NW * i) * 1024), 16, 0, 0);
  };
  // ---- one round of loads for both heads: every thread owns chunk (row tid >> 3, 16 bytes tid & 7) of every image
  const int ch = tid & 7, row = tid >> 3;
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  // (unconditional loads at clamped rows, nothing computed from them here: see attn_bwd_roles_mfma_kernel)
  const bool has_mask = a.mask != nullptr, has_lo = a.o_lo != nullptr;
  uint4 vq[G], vo[G], vl[G], vk[G], vv[G];
  float lse_r[G];
  const int rq = min(row, nq - 1), rk = min(row, nk - 1);
  const int mkey = min(tid & (k_rows - 1), nk - 1), mhead = h0 + ((tid / k_rows) & (G - 1));
  const float* mp = has_mask ? a.mask + (int64_t)b * a.msb + (int64_t)mhead * a.msh + mkey : a.lse + ((int64_t)b * a.H + h0) * nq;
  float mval = *mp;
#pragma unroll
  for (int gi = 0; gi < G; gi++) {
    const int hh = h0 + gi;
    const bf16* ob = (const bf16*)a.o + ((int64_t)b * nq + rq) * a.ldo + hh * 64 + ch * 8;
    vq[gi] = *reinterpret_cast<const uint4*>((const bf16*)a.q + ((int64_t)b * nq + rq) * a.ldq + hh * 64 + ch * 8);
    vo[gi] = *reinterpret_cast<const uint4*>(ob);
    vl[gi] = *reinterpret_cast<const uint4*>(has_lo ? (const bf16*)a.o_lo + ((int64_t)b * nq + rq) * a.ldo + hh * 64 + ch * 8 : ob);
    lse_r[gi] = a.lse[((int64_t)b * a.H + hh) * nq + rq];
    vk[gi] = *reinterpret_cast<const uint4*>((const bf16*)a.k + ((int64_t)b * nk + rk) * a.ldk + hh * 64 + ch * 8);
    vv[gi] = *reinterpret_cast<const uint4*>((const bf16*)a.v + ((int64_t)b * nk + rk) * a.ldv + hh * 64 + ch * 8);
  }
  // (requested BEFORE the projection loop, round 5: Q, K, V, O arrive under it instead of in a round trip of their own
  // behind it -- 1.1 us per workgroup by the phase probe, 3.1 from cold caches)
  const int nkt = g.Dm / BKF;
#pragma unroll
  for (int p = 0; p < NBUF - 1; p++)
    if (p < nkt) issue(p);
  for (int kt = 0; kt < nkt; kt++) {
    if (kt + NBUF - 2 >= nkt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * (NBUF - 2)) : "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (kt + NBUF - 1 < nkt) issue(kt + NBUF - 1);
    const char* Ws = smem + (kt % NBUF) * STAGE;
    const char* Xs = Ws + WCH * 1024;
#pragma unroll
    for (int ks = 0; ks < BKF / 32; ks++) {
      bf16x8 pf[NJ], qf[NI];
#pragma unroll
      for (int j = 0; j < NJ; j++)
        pf[j] = *reinterpret_cast<const bf16x8*>(Ws + off32(wr * 32 + j * 16 + (lane & 15), ks * 4 + (lane >> 4)));
#pragma unroll
      for (int i = 0; i < NI; i++)
        qf[i] = *reinterpret_cast<const bf16x8*>(Xs + off32(i * 16 + (lane & 15), ks * 4 + (lane >> 4)));
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int i = 0; i < NI; i++) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[j], qf[i], acc[j][i], 0, 0, 0);
    }
  }
  OVQA_PROBE(1);
  OVQA_PROBE(2);
  __syncthreads();  // every wave is done with the staging ring: it becomes the images

  // ---- dO image [32][64] (bf16) of the wave's head from the accumulators; rows beyond nq are zero
  {
    char* Gs = smem + (wr >> 1) * prob_bytes + q_rows * 128;
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const int r = i * 16 + (lane & 15);
      const int col = (wr & 1) * 32 + (lane >> 4) * 8;
      bf16x8 o8;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        o8[e] = r < nq ? (bf16)acc[0][i][e] : (bf16)0.f;
        o8[4 + e] = r < nq ? (bf16)acc[1][i][e] : (bf16)0.f;
      }
      *reinterpret_cast<bf16x8*>(Gs + img_off(r, col >> 3)) = o8;
    }
  }
  __syncthreads();  // the dO images are complete: delta reads them
#pragma unroll
  for (int gi = 0; gi < G; gi++) {
    char* base = smem + gi * prob_bytes;
    float* lse_g = reinterpret_cast<float*>(base + img_bytes) + k_rows;
    if (row >= nq) vq[gi] = zero4;
    if (row >= nk) { vk[gi] = zero4; vv[gi] = zero4; }
    if (!has_lo) vl[gi] = zero4;
    *reinterpret_cast<uint4*>(base + img_off(row, ch)) = vq[gi];
    const bf16x8 g8 = *reinterpret_cast<const bf16x8*>(base + q_rows * 128 + img_off(row, ch));  // (zero beyond nq: dl = 0)
    const bf16x8 o8 = *reinterpret_cast<const bf16x8*>(&vo[gi]);
    const bf16x8 l8 = *reinterpret_cast<const bf16x8*>(&vl[gi]);
    float dl = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) dl += (float)g8[e] * ((float)o8[e] + (float)l8[e]);
    dl += __shfl_xor(dl, 1, 64);
    dl += __shfl_xor(dl, 2, 64);
    dl += __shfl_xor(dl, 4, 64);
    if (ch == 0) {
      lse_g[row] = row < nq ? lse_r[gi] * LOG2E : INFINITY;  // p = 0 beyond nq
      lse_g[q_rows + row] = dl;
      if (row < nq && a.delta) a.delta[((int64_t)b * a.H + h0 + gi) * nq + row] = dl;
    }
    *reinterpret_cast<uint4*>(base + 2 * q_rows * 128 + img_off(row, ch)) = vk[gi];
    *reinterpret_cast<uint4*>(base + (2 * q_rows + k_rows) * 128 + img_off(row, ch)) = vv[gi];
  }
  if (tid < G * k_rows)
    reinterpret_cast<float*>(smem + (tid / k_rows) * prob_bytes + img_bytes)[tid % k_rows] =
        tid % k_rows < nk ? (has_mask ? mval * LOG2E : 0.f) : -INFINITY;
  __syncthreads();
  OVQA_PROBE(3);
  smallk_bwd_compute<true, 1, G>(a, smem, b * a.H + h0, wave, lane, false);
}

// ------------------------------------------------------- role-split backward, 32 < n_k <= 128 and n_q <= 128
// Image self-attention (100 x 100): ONE launch, 8 waves.  Q, dO, K, V of a (batch, head) are staged once; waves
// 0-3 play kernel A (one 32-query tile each, all key tiles: dQ), waves 4-7 play kernel B (one 32-key tile each, all
// query tiles: dK, dV).  Nothing is reduced across waves; delta = dO . O is computed once per query row during
// staging.  Saves the second launch (~5 us floor) and the second staging of the same 50 KB.
// NQT_T / NKT_T: compile-time tile counts (0 = use the runtime arguments): with constants the two tile loops are
// fully unrolled and the MFMA chains of different tiles interleave.
// OVQA_ROLES_WAVES_PER_EU = 4: at most 128 VGPRs, so that TWO workgroups share a CU (their 65.5 KB of LDS each allow it):
// with the 152 registers the compiler takes when left alone, the 512 workgroups of a 64-sample, 8-head launch ran as two
// rounds of one workgroup per CU (10 us each by the phase probe, 25 us per launch).
#ifndef OVQA_ROLES_WAVES_PER_EU
#define OVQA_ROLES_WAVES_PER_EU 4
#endif
// FUSE_NBUF > 0 (round 5; needs NQT_T = NKT_T = 4): the fc_o dX product inside, the form attn_bwd_do_smallk_mfma_kernel
// has for the guided attention -- dO of the head = dY [128 rows, d_model] x 64 rows of the transposed fc_o weights, with
// that kernel's projection loop (ring of FUSE_NBUF K steps of 64: 24 KB each, aliased with the images), rounded to bf16
// straight into the dO image; grid (B, H), so that the 8 heads of a sample -- which all stream the sample's dY rows --
// share an XCD.  dO never travels through HBM and the 6400 x 512 <- 512 dX launch in front of this kernel is gone.
template <int NQT_T, int NKT_T, bool ROWMASK, int FUSE_NBUF = 0, bool DMA = false>
__global__ __launch_bounds__(512, (NQT_T && NKT_T) ? OVQA_ROLES_WAVES_PER_EU : 1) void attn_bwd_roles_mfma_kernel(DoBwdArgs g, int nqt_rt, int nkt_rt) {
  constexpr bool FUSE = FUSE_NBUF > 0, EARLY = FUSE_NBUF >= 10;
  static_assert(!DMA || ROWMASK, "clamped rows rely on the key mask row / lse to vanish");
  constexpr int RING = FUSE ? FUSE_NBUF % 10 : 1;  // (1: never used, keeps the un-fused instantiations free of a % 0)
  static_assert(!FUSE || (NQT_T == 4 && NKT_T == 4), "the fused form is built for 4 x 4 tiles");
  const ovqa::AttnBwdArgs& a = g.att;
  const int nqt = NQT_T ? NQT_T : nqt_rt, nkt = NKT_T ? NKT_T : nkt_rt;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nk = a.nk, nq = a.nq;
  const int q_rows = nqt * 32, k_rows = nkt * 32;
  const int b = FUSE ? (int)blockIdx.x : (int)blockIdx.x / a.H;
  const int h = FUSE ? (int)blockIdx.y : (int)blockIdx.x - b * a.H;
  const int pid = b * a.H + h;
  char* Qs = smem;
  char* Gs = Qs + q_rows * 128;
  char* Ks = Gs + q_rows * 128;
  char* Vs = Ks + k_rows * 128;
  // fp32 rows behind the images, all in log2 units so that a probability is one fma + one exp2:
  //   mlds[key] = mask * log2(e) (-inf beyond nk),  lse_s[q] = lse * log2(e) (+inf beyond nq: p = 0 without a branch)
  float* mlds = reinterpret_cast<float*>(Vs + k_rows * 128);
  float* lse_s = mlds + k_rows;
  float* del_s = lse_s + q_rows;
  OVQA_PROBE(0);
  // ---- (fused form) dO of the head into accumulators: acc[j][i] = 16 rows (wc * 32 + i * 16 ..) x 16 features
  // (wr * 32 + j * 16 ..) per wave, the staging ring over the (not yet written) images
  constexpr int PJ_NI = 2, PJ_NJ = 2;
  f32x4 acc[PJ_NJ][PJ_NI];
  const int wc = wave >> 1, wr = wave & 1;
  // the projection loop of the fused form; EARLY (FUSE_NBUF >= 10): it runs BEHIND the staging requests below, so that Q, K, V,
  // O arrive under it.  MEASURED (phase probe, round 5): slower -- the 14 register loads per thread of two co-resident
  // workgroups take 3.4 us just to issue (8-9 from cold caches: the CU's fetch path is the bottleneck either way) and the
  // ring's first tiles queue behind them: 19.9 against 17.6-18.8 us per workgroup warm, 27.4 against 26.4 cold.
  auto project = [&]() {
    constexpr int BKF = 64, NW = 8, PROWS = 8, CHR = 8, WCH = 64 / PROWS, XCH = 128 / PROWS, PER = (WCH + XCH) / NW;
    constexpr int STAGE = (WCH + XCH) * 1024;
#pragma unroll
    for (int j = 0; j < PJ_NJ; j++)
#pragma unroll
      for (int i = 0; i < PJ_NI; i++) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto off32 = [&](int row, int ch) { return row * (BKF * 2) + ((ch ^ (row & 7)) << 4); };
    const bf16* src[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
      const int ci = wave + NW * i;
      const int prow = lane / CHR, pch = lane % CHR;
      if (ci < WCH) {
        const int row = ci * PROWS + prow;
        src[i] = g.wt + (int64_t)(h * 64 + perm32(row)) * g.ldwt + ((pch ^ (row & 7)) << 3);
      } else {
        const int row = (ci - WCH) * PROWS + prow;
        const int r = row < nq ? row : nq - 1;
        src[i] = g.dy + ((int64_t)b * nq + r) * g.lddy + ((pch ^ (row & 7)) << 3);
      }
    }
    auto issue = [&](int kt) {
      char* buf = smem + (kt % RING) * STAGE;
#pragma unroll
      for (int i = 0; i < PER; i++)
        __builtin_amdgcn_global_load_lds((gbl_void*)(src[i] + kt * BKF), (lds_void*)(buf + (wave + NW * i) * 1024), 16, 0, 0);
    };
    const int nkt_p = g.Dm / BKF;
#pragma unroll
    for (int p = 0; p < RING - 1; p++)
      if (p < nkt_p) issue(p);
    for (int kt = 0; kt < nkt_p; kt++) {
      if (kt + RING - 2 >= nkt_p) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * (RING - 2)) : "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (kt + RING - 1 < nkt_p) issue(kt + RING - 1);
      const char* Ws = smem + (kt % RING) * STAGE;
      const char* Xs = Ws + WCH * 1024;
#pragma unroll
      for (int ks = 0; ks < BKF / 32; ks++) {
        bf16x8 pf[PJ_NJ], qf[PJ_NI];
#pragma unroll
        for (int j = 0; j < PJ_NJ; j++)
          pf[j] = *reinterpret_cast<const bf16x8*>(Ws + off32(wr * 32 + j * 16 + (lane & 15), ks * 4 + (lane >> 4)));
#pragma unroll
        for (int i = 0; i < PJ_NI; i++)
          qf[i] = *reinterpret_cast<const bf16x8*>(Xs + off32(wc * 32 + i * 16 + (lane & 15), ks * 4 + (lane >> 4)));
#pragma unroll
        for (int j = 0; j < PJ_NJ; j++)
#pragma unroll
          for (int i = 0; i < PJ_NI; i++) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[j], qf[i], acc[j][i], 0, 0, 0);
      }
    }
    OVQA_PROBE(6);
  };
  if constexpr (FUSE && !EARLY) project();
  // ---- staging: one round of global loads (all issued before the first LDS store): a thread owns chunk `ch` of rows
  // r0 and r0 + 64 of each of Q, dO, K, V -- and of O, so that delta = dO . O falls out of the staging (8 consecutive
  // lanes hold a row) instead of a second, dependent pass over dO and O.
  {
    const int ch = tid & 7, r0 = tid >> 3;
    const bf16* qb = (const bf16*)a.q + (int64_t)b * nq * a.ldq + h * 64 + ch * 8;
    const bf16* gb = (const bf16*)a.d_o + (int64_t)b * nq * a.lddo + h * 64 + ch * 8;
    const bf16* ob = (const bf16*)a.o + (int64_t)b * nq * a.ldo + h * 64 + ch * 8;
    const bf16* lob = a.o_lo ? (const bf16*)a.o_lo + (int64_t)b * nq * a.ldo + h * 64 + ch * 8 : nullptr;
    const bf16* kb = (const bf16*)a.k + (int64_t)b * nk * a.ldk + h * 64 + ch * 8;
    const bf16* vb = (const bf16*)a.v + (int64_t)b * nk * a.ldv + h * 64 + ch * 8;
    const float* lb = a.lse + (int64_t)pid * nq;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    // Every load is UNCONDITIONAL per lane (clamped row; branches only on kernel-uniform conditions) and nothing is
    // computed from a loaded value before all of them are issued: behind a per-lane `if` the compiler ends the branch with
    // `s_waitcnt vmcnt(0)` (the * LOG2E, or a register copy, lands inside it) -- four dependent round trips at the head of
    // this kernel instead of one.  Rows beyond nq / nk become zero / +-inf where the images are written.
    uint4 vq[2], vd[2], vo[2], vl[2], vk[2], vv[2];
    float lse_r[2];
    // (no mask row / no o_lo: the load goes to another valid address and its result is dropped below)
    const bool has_mrow = a.msq == 0 && a.mask != nullptr, has_lo = a.o_lo != nullptr;
    const float* mp = has_mrow ? a.mask + (int64_t)b * a.msb + (int64_t)h * a.msh : lb;
    // DMA form (round 5): the Q, dO, K, V images are filled by direct-to-LDS loads -- a wave instruction fills 8 image rows
    // (1 KiB), lane l the 16-byte slot l & 7 of row l >> 3 with the chunk the XOR layout puts there -- instead of travelling
    // through 12 of this thread's 14 uint4 registers.  Rows beyond nq / nk re-read the last valid row: finite values that
    // lse = +inf / mask = -inf turn into p = 0 (what the zero fill achieved), their outputs are not stored.
    auto dma_image = [&](char* img, const bf16* base, int64_t ld, int n, int rows_pad) {
      const int prow = lane >> 3, slot = lane & 7;
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int piece = wave + 8 * i;
        if (piece * 8 < rows_pad) {  // (wave-uniform)
          const int row = piece * 8 + prow;
          const bf16* src = base + (int64_t)min(row, n - 1) * ld + ((slot ^ xs(row)) << 3);
          __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(img + piece * 1024), 16, 0, 0);
        }
      }
    };
    auto dma_qkv = [&]() {
      dma_image(Qs, qb - ch * 8, a.ldq, nq, q_rows);
      if constexpr (!FUSE) dma_image(Gs, gb - ch * 8, a.lddo, nq, q_rows);
      dma_image(Ks, kb - ch * 8, a.ldk, nk, k_rows);
      dma_image(Vs, vb - ch * 8, a.ldv, nk, k_rows);
    };
    if constexpr (DMA && !FUSE) dma_qkv();
    float mval = mp[min(tid, (has_mrow ? nk : nq) - 1)];
    const bf16* lob2 = has_lo ? lob : ob;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int row = r0 + 64 * i;
      const int rq = min(row, nq - 1), rk = min(row, nk - 1);
      if constexpr (!DMA) vq[i] = *reinterpret_cast<const uint4*>(qb + (int64_t)rq * a.ldq);
      if constexpr (!DMA && !FUSE) vd[i] = *reinterpret_cast<const uint4*>(gb + (int64_t)rq * a.lddo);
      vo[i] = *reinterpret_cast<const uint4*>(ob + (int64_t)rq * a.ldo);
      vl[i] = *reinterpret_cast<const uint4*>(lob2 + (int64_t)rq * a.ldo);
      lse_r[i] = lb[rq];
      if constexpr (!DMA) {
        vk[i] = *reinterpret_cast<const uint4*>(kb + (int64_t)rk * a.ldk);
        vv[i] = *reinterpret_cast<const uint4*>(vb + (int64_t)rk * a.ldv);
      }
    }
    OVQA_PROBE(1);
    if constexpr (FUSE && EARLY) project();
    if constexpr (FUSE) {
      __syncthreads();  // every wave is done with the staging ring: it becomes the images
      if constexpr (DMA) dma_qkv();
      // dO image [128][64] (bf16) from the accumulators; rows beyond nq are zero
#pragma unroll
      for (int i = 0; i < PJ_NI; i++) {
        const int r = wc * 32 + i * 16 + (lane & 15);
        const int col = wr * 32 + (lane >> 4) * 8;
        bf16x8 o8;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          o8[e] = r < nq ? (bf16)acc[0][i][e] : (bf16)0.f;
          o8[4 + e] = r < nq ? (bf16)acc[1][i][e] : (bf16)0.f;
        }
        *reinterpret_cast<bf16x8*>(Gs + img_off(r, col >> 3)) = o8;
      }
      if constexpr (!DMA) __syncthreads();  // the dO image is complete: delta reads it
    }
    if constexpr (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // every wave's pieces have landed (and the dO image of the fused form is written)
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int row = r0 + 64 * i;
      const bool qv = row < nq, kv = row < nk;
      if (!has_lo) vl[i] = zero4;
      if constexpr (FUSE || DMA) vd[i] = *reinterpret_cast<const uint4*>(Gs + img_off(min(row, q_rows - 1), ch));
      const bf16x8 g8 = *reinterpret_cast<const bf16x8*>(&vd[i]);
      const bf16x8 o8 = *reinterpret_cast<const bf16x8*>(&vo[i]);
      const bf16x8 l8 = *reinterpret_cast<const bf16x8*>(&vl[i]);
      float dl = 0.f;
#pragma unroll
      for (int e = 0; e < 8; e++) dl += (float)g8[e] * ((float)o8[e] + (float)l8[e]);
      dl += __shfl_xor(dl, 1, 64);
      dl += __shfl_xor(dl, 2, 64);
      dl += __shfl_xor(dl, 4, 64);
      if constexpr (!DMA) {
        if (!qv) { vq[i] = zero4; vd[i] = zero4; }
        if (!kv) { vk[i] = zero4; vv[i] = zero4; }
      }
      if (row < q_rows) {
        if constexpr (!DMA) *reinterpret_cast<uint4*>(Qs + img_off(row, ch)) = vq[i];
        if constexpr (!DMA && !FUSE) *reinterpret_cast<uint4*>(Gs + img_off(row, ch)) = vd[i];
        if (ch == 0) {
          lse_s[row] = qv ? lse_r[i] * LOG2E : INFINITY;
          del_s[row] = qv ? dl : 0.f;
          if (qv) a.delta[(int64_t)pid * nq + row] = dl;
        }
      }
      if constexpr (!DMA) {
        if (row < k_rows) {
          *reinterpret_cast<uint4*>(Ks + img_off(row, ch)) = vk[i];
          *reinterpret_cast<uint4*>(Vs + img_off(row, ch)) = vv[i];
        }
      }
    }
    if (a.msq == 0 && tid < k_rows) mlds[tid] = tid < nk ? (has_mrow ? mval * LOG2E : 0.f) : -INFINITY;
  }
  OVQA_PROBE(2);
  __syncthreads();
  OVQA_PROBE(3);
  constexpr bool row_mask = ROWMASK;  // compile-time: the common key-padding form carries no per-element checks
  const float scale2 = a.scale * LOG2E;

  if (wave < 4) {
    // ------------------------------------------------ role A: dQ of query tile tq (lane = query)
    const int tq = wave;
    if (tq * 32 >= nq) return;
    const int q = tq * 32 + (lane & 31);
    const bool qok = q < nq;
    const int qc = qok ? q : nq - 1;
    const float lse2 = lse_s[qc], delta = del_s[qc];
    const float* mrow = a.mask ? a.mask + (int64_t)b * a.msb + (int64_t)h * a.msh + (int64_t)qc * a.msq : nullptr;
    bf16x8 qf[4], gf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      qf[ks] = frag_rows(Qs, tq * 32, ks, lane);
      gf[ks] = frag_rows(Gs, tq * 32, ks, lane);
    }
    f32x16 dqt[2];
#pragma unroll
    for (int d = 0; d < 2; d++)
#pragma unroll
      for (int r = 0; r < 16; r++) dqt[d][r] = 0.f;
#pragma unroll
    for (int t = 0; t < nkt; t++) {
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; r++) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Ks, t * 32, ks, lane), qf[ks], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Vs, t * 32, ks, lane), gf[ks], dp, 0, 0, 0);
      }
      float ds[16];
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const int key0 = t * 32 + 8 * g4 + 4 * (lane >> 5);  // acc_row(4 * g4 + e, lane) = key0 + e
        float mm[4];
        if (row_mask) {
          const float4 m4 = *reinterpret_cast<const float4*>(mlds + key0);
          mm[0] = m4.x - lse2; mm[1] = m4.y - lse2; mm[2] = m4.z - lse2; mm[3] = m4.w - lse2;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int r = 4 * g4 + e;
          float p;
          if (row_mask) {
            p = exp2_fast(st[r] * scale2 + mm[e]);  // -inf beyond nk -> 0
          } else {
            p = 0.f;
            if (key0 + e < nk) p = exp2_fast(st[r] * scale2 + ((mrow ? mrow[key0 + e] : 0.f) * LOG2E - lse2));
          }
          ds[r] = p * (dp[r] - delta);
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        bf16x8 db;
#pragma unroll
        for (int j = 0; j < 8; j++) db[j] = (bf16)ds[8 * s2 + j];
#pragma unroll
        for (int d = 0; d < 2; d++)
          dqt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Ks, t * 32 + 16 * s2, d * 32, lane), db, dqt[d], 0, 0, 0);
      }
    }
    if (qok) {
      bf16* drow = (bf16*)a.dq + ((int64_t)b * nq + q) * a.lddq + h * 64;
#pragma unroll
      for (int d = 0; d < 2; d++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
          bf16x4 o4;
#pragma unroll
          for (int e = 0; e < 4; e++) o4[e] = (bf16)(dqt[d][4 * g4 + e] * a.scale);
          *reinterpret_cast<bf16x4*>(drow + d * 32 + 8 * g4 + 4 * (lane >> 5)) = o4;
        }
    }
    OVQA_PROBE(4);
  } else {
    // ------------------------------------------------ role B: dK / dV of key tile tk (lane = key)
    const int tk = wave - 4;
    if (tk * 32 >= nk) return;
    const int key = tk * 32 + (lane & 31);
    const bool kok = key < nk;
    const float* mcol = a.mask ? a.mask + (int64_t)b * a.msb + (int64_t)h * a.msh + (kok ? key : 0) : nullptr;
    const float mconst = row_mask ? mlds[key] : 0.f;  // log2 units; -inf beyond nk
    // (the wave's own K / V fragments are read from LDS again for every query tile, behind a compiler barrier that keeps
    // the reads from being merged and the tiles from overlapping: 32 registers + one tile's accumulators fewer -- what
    // brings the kernel under 128 VGPRs, i.e. two workgroups per CU, without spilling)
    f32x16 dvt[2], dkt[2];
#pragma unroll
    for (int d = 0; d < 2; d++)
#pragma unroll
      for (int r = 0; r < 16; r++) { dvt[d][r] = 0.f; dkt[d][r] = 0.f; }
#pragma unroll
    for (int t = 0; t < nqt; t++) {
      f32x16 s_, dp;
#pragma unroll
      for (int r = 0; r < 16; r++) { s_[r] = 0.f; dp[r] = 0.f; }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        s_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Qs, t * 32, ks, lane), frag_rows(Ks, tk * 32, ks, lane), s_, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Gs, t * 32, ks, lane), frag_rows(Vs, tk * 32, ks, lane), dp, 0, 0, 0);
      }
      bf16x8 pb[2], db[2];
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const int q0 = t * 32 + 8 * g4 + 4 * (lane >> 5);  // acc_row(4 * g4 + e, lane) = q0 + e
        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + q0);  // +inf beyond nq: p = 0
        const float4 d4 = *reinterpret_cast<const float4*>(del_s + q0);
        const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int r = 4 * g4 + e;
          float pv;
          if (row_mask) {
            pv = exp2_fast(s_[r] * scale2 + (mconst - ll[e]));
          } else {
            pv = 0.f;
            if (q0 + e < nq && kok)
              pv = exp2_fast(s_[r] * scale2 + ((mcol ? mcol[(int64_t)(q0 + e) * a.msq] : 0.f) * LOG2E - ll[e]));
          }
          pb[r >> 3][r & 7] = (bf16)pv;
          db[r >> 3][r & 7] = (bf16)(pv * (dp[r] - dd[e]));
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int d = 0; d < 2; d++) {
          dvt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Gs, t * 32 + 16 * s2, d * 32, lane), pb[s2], dvt[d], 0, 0, 0);
          dkt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Qs, t * 32 + 16 * s2, d * 32, lane), db[s2], dkt[d], 0, 0, 0);
        }
    }
    if (kok) {
      bf16* dkrow = (bf16*)a.dk_ + ((int64_t)b * nk + key) * a.lddk + h * 64;
      bf16* dvrow = (bf16*)a.dv_ + ((int64_t)b * nk + key) * a.lddv + h * 64;
#pragma unroll
      for (int d = 0; d < 2; d++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
          bf16x4 k4, v4;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            k4[e] = (bf16)(dkt[d][4 * g4 + e] * a.scale);
            v4[e] = (bf16)dvt[d][4 * g4 + e];
          }
          *reinterpret_cast<bf16x4*>(dkrow + d * 32 + 8 * g4 + 4 * (lane >> 5)) = k4;
          *reinterpret_cast<bf16x4*>(dvrow + d * 32 + 8 * g4 + 4 * (lane >> 5)) = v4;
        }
    }
    OVQA_PROBE_T(5, 256);
  }
}

template <typename K>
int ensure_lds(K kernel, size_t bytes, const char* what) {
  if (bytes > 160 * 1024) {
    ovqa_set_error("%s: needs %zu B of LDS", what, bytes);
    return OVQA_ERR_UNSUPPORTED;
  }
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
      ovqa_set_error("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
      return OVQA_ERR_LAUNCH;
    }
  }
  return OVQA_OK;
}

// problems packed per workgroup: at most 4/W (one wave per 32-row tile) and what fits in ~150 KiB of LDS
inline int pack_factor(int W, size_t prob_bytes) {
  int G = 4 / W;
  const int fit = (int)((150 * 1024) / prob_bytes);
  if (G > fit) G = fit;
  return G < 1 ? 1 : G;
}

// query tiles per problem and workgroup: as many as the queries need (1, 2 or 4), fewer if the images do not fit
inline int fit_tiles(int n, size_t fixed_bytes, size_t bytes_per_tile) {
  int W = (n + 31) / 32;
  if (W > 4) W = 4;
  if (W == 3) W = 4;
  while (W > 1 && fixed_bytes + (size_t)W * bytes_per_tile > 150 * 1024) W >>= 1;
  return W;
}

template <int NKT, bool ROWMASK, bool WANT_ATT, int D>
int launch_fwd_t(const ovqa::AttnArgs& a, hipStream_t st) {
  constexpr int PITCH = Img<D>::PITCH;
  const int W = fit_tiles(a.nq, (size_t)2 * NKT * 32 * PITCH + NKT * 32 * 4, (size_t)32 * PITCH);
  const size_t prob = (size_t)(32 * W + 2 * NKT * 32) * PITCH + NKT * 32 * 4;
  const int G = pack_factor(W, prob);
  const size_t lds = (size_t)G * prob;
  int rc = ensure_lds(attn_fwd_mfma_kernel<NKT, ROWMASK, WANT_ATT, D>, lds, "attention_fwd(mfma)");
  if (rc != OVQA_OK) return rc;
  const int64_t nprob = (int64_t)a.B * a.H;
  dim3 grid((unsigned)((nprob + G - 1) / G), (unsigned)((a.nq + 32 * W - 1) / (32 * W)));
  OVQA_LAUNCH_TIMED((attn_fwd_mfma_kernel<NKT, ROWMASK, WANT_ATT, D>), grid, dim3(256), lds, st, a, W, G);
  return ovqa_check_launch("attention_fwd(mfma)");
}

template <int NKT, int D>
int launch_fwd(const ovqa::AttnArgs& a, hipStream_t st) {
  const bool rowmask = a.msq == 0, att = a.att != nullptr;
  if (rowmask) return att ? launch_fwd_t<NKT, true, true, D>(a, st) : launch_fwd_t<NKT, true, false, D>(a, st);
  return att ? launch_fwd_t<NKT, false, true, D>(a, st) : launch_fwd_t<NKT, false, false, D>(a, st);
}

template <int D>
int launch_fwd_d(const ovqa::AttnArgs& a, hipStream_t st) {
  if (a.nk <= 32) return launch_fwd<1, D>(a, st);
  if (a.nk <= 64) return launch_fwd<2, D>(a, st);
  if (a.nk <= 128) return launch_fwd<4, D>(a, st);
  if (a.nk <= 192) return launch_fwd<6, D>(a, st);
  if constexpr (D == 64) return launch_fwd<8, D>(a, st);
  ovqa_set_error("attention_fwd(mfma): n_k = %d > 192 with heads of %d features", a.nk, D);
  return OVQA_ERR_UNSUPPORTED;
}

// two-kernel backward (dQ, then dK/dV) for any supported head size
template <int D>
int launch_bwd_two(const ovqa::AttnBwdArgs& a, hipStream_t st) {
  constexpr int PITCH = Img<D>::PITCH;
  const int64_t nprob = (int64_t)a.B * a.H;
  const bool rowmask = a.msq == 0;
  {  // dQ: waves over query tiles, all keys resident
    const int nkt = (a.nk + 31) / 32;
    const int W = fit_tiles(a.nq, (size_t)2 * nkt * 32 * PITCH + nkt * 32 * 4, (size_t)2 * 32 * PITCH);
    const size_t prob = (size_t)(2 * 32 * W + 2 * nkt * 32) * PITCH + nkt * 32 * 4;
    const int G = pack_factor(W, prob);
    const size_t lds = (size_t)G * prob;
    int rc = rowmask ? ensure_lds(attn_bwd_dq_mfma_kernel<true, D>, lds, "attention_bwd(mfma,dq)")
                     : ensure_lds(attn_bwd_dq_mfma_kernel<false, D>, lds, "attention_bwd(mfma,dq)");
    if (rc != OVQA_OK) return rc;
    dim3 grid((unsigned)((nprob + G - 1) / G), (unsigned)((a.nq + 32 * W - 1) / (32 * W)));
    if (rowmask) OVQA_LAUNCH_TIMED((attn_bwd_dq_mfma_kernel<true, D>), grid, dim3(256), lds, st, a, W, G, nkt);
    else OVQA_LAUNCH_TIMED((attn_bwd_dq_mfma_kernel<false, D>), grid, dim3(256), lds, st, a, W, G, nkt);
    rc = ovqa_check_launch("attention_bwd(mfma,dq)");
    if (rc != OVQA_OK) return rc;
  }
  {  // dK/dV: waves over key tiles, all queries resident
    const int nqt = (a.nq + 31) / 32;
    const int W = fit_tiles(a.nk, (size_t)2 * nqt * 32 * PITCH + 2 * nqt * 32 * 4, (size_t)2 * 32 * PITCH);
    const size_t prob = (size_t)(2 * nqt * 32 + 2 * 32 * W) * PITCH + 2 * nqt * 32 * 4;
    const int G = pack_factor(W, prob);
    const size_t lds = (size_t)G * prob;
    int rc = rowmask ? ensure_lds(attn_bwd_dkv_mfma_kernel<true, D>, lds, "attention_bwd(mfma,dkv)")
                     : ensure_lds(attn_bwd_dkv_mfma_kernel<false, D>, lds, "attention_bwd(mfma,dkv)");
    if (rc != OVQA_OK) return rc;
    dim3 grid((unsigned)((nprob + G - 1) / G), (unsigned)((a.nk + 32 * W - 1) / (32 * W)));
    if (rowmask) OVQA_LAUNCH_TIMED((attn_bwd_dkv_mfma_kernel<true, D>), grid, dim3(256), lds, st, a, W, G, nqt);
    else OVQA_LAUNCH_TIMED((attn_bwd_dkv_mfma_kernel<false, D>), grid, dim3(256), lds, st, a, W, G, nqt);
    return ovqa_check_launch("attention_bwd(mfma,dkv)");
  }
}

// (The role-split backward fills its Q / dO / K / V images by direct-to-LDS loads where the mask is a key row; through
// registers -- OVQA_ROLES_DMA=0 of round 5 -- it measured 26.07 against 25.55 us per launch with the fc_o projection inside,
// 21.13 against 21.15 without: the staging is bound by the CU's fetch path either way.)
int launch_bwd(const ovqa::AttnBwdArgs& a, hipStream_t st) {
  if (a.dk == 96) return launch_bwd_two<96>(a, st);
  if (a.dk == 128) return launch_bwd_two<128>(a, st);
  const int64_t nprob = (int64_t)a.B * a.H;
  const bool rowmask = a.msq == 0;
  if (a.nk > 32 && a.nk <= 128 && a.nq <= 128) {  // one launch, role-split waves
    const int nqt = (a.nq + 31) / 32, nkt = (a.nk + 31) / 32;
    const size_t lds = (size_t)(2 * nqt * 32 + 2 * nkt * 32) * 128 + (size_t)(nkt * 32 + 2 * nqt * 32) * 4;
    const DoBwdArgs g0{nullptr, 0, nullptr, 0, a, 0};
#define OVQA_ROLES(NQ, NK, RM)                                                                                        \
  {                                                                                                                   \
    int rc = ensure_lds(attn_bwd_roles_mfma_kernel<NQ, NK, RM>, lds, "attention_bwd(mfma,roles)");                    \
    if (rc != OVQA_OK) return rc;                                                                                     \
    OVQA_LAUNCH_TIMED((attn_bwd_roles_mfma_kernel<NQ, NK, RM>), dim3((unsigned)nprob), dim3(512), lds, st, g0, nqt, nkt); \
  }
    if (nqt == 4 && nkt == 4) {  // the 100 x 100 image self-attention: fully unrolled tile loops
      if (rowmask) {
        int rc = ensure_lds(attn_bwd_roles_mfma_kernel<4, 4, true, 0, true>, lds, "attention_bwd(mfma,roles)");
        if (rc != OVQA_OK) return rc;
        OVQA_LAUNCH_TIMED((attn_bwd_roles_mfma_kernel<4, 4, true, 0, true>), dim3((unsigned)nprob), dim3(512), lds, st, g0, nqt, nkt);
      } else OVQA_ROLES(4, 4, false)
    } else {
      if (rowmask) OVQA_ROLES(0, 0, true) else OVQA_ROLES(0, 0, false)
    }
#undef OVQA_ROLES
    return ovqa_check_launch("attention_bwd(mfma,roles)");
  }
  if (a.nk <= 32 && a.nq <= 128) {  // one launch for dQ, dK and dV
    int W = (a.nq + 31) / 32;
    if (W == 3) W = 4;
    // problems per workgroup (compile-time in the kernel): 4 / W; single-tile problems 2 at a time when 4 at a time
    // would leave CUs without a workgroup
    const int G = (W == 1 && (nprob + 3) / 4 < 256) ? 2 : 4 / W;
    const size_t prob = (size_t)(2 * 32 * W + 2 * 32) * 128 + 32 * 4 + 2 * 32 * W * 4 + 4096 * 4;
    const size_t lds = (size_t)G * prob;
    const dim3 grid((unsigned)((nprob + G - 1) / G));
#define OVQA_SMALLK(RM, WV, GV)                                                                                       \
  {                                                                                                                   \
    int rc = ensure_lds(attn_bwd_smallk_mfma_kernel<RM, WV, GV>, lds, "attention_bwd(mfma,merged)");                  \
    if (rc != OVQA_OK) return rc;                                                                                     \
    OVQA_LAUNCH_TIMED((attn_bwd_smallk_mfma_kernel<RM, WV, GV>), grid, dim3(WV == 1 ? 128 * GV : 256), lds, st, a);  \
  }
    if (W == 1 && G == 2) {
      if (rowmask) OVQA_SMALLK(true, 1, 2) else OVQA_SMALLK(false, 1, 2)
    } else if (W == 1) {
      if (rowmask) OVQA_SMALLK(true, 1, 4) else OVQA_SMALLK(false, 1, 4)
    } else if (W == 2) {
      if (rowmask) OVQA_SMALLK(true, 2, 2) else OVQA_SMALLK(false, 2, 2)
    } else {
      if (rowmask) OVQA_SMALLK(true, 4, 1) else OVQA_SMALLK(false, 4, 1)
    }
#undef OVQA_SMALLK
    return ovqa_check_launch("attention_bwd(mfma,merged)");
  }
  return launch_bwd_two<64>(a, st);
}

}  // namespace

namespace ovqa {

bool mfma_attention_bwd_supported(const AttnBwdArgs& a) {
  auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  auto al8 = [](const void* p) { return ((uintptr_t)p & 7) == 0; };
  // gradients of the returned attention weights / log-sum-exp and dropout on the probabilities: VALU kernels
  const bool d_ok = a.dk == a.dv && (a.dk == 64 || a.dk == 96 || a.dk == 128);
  const int n_max = a.dk == 64 ? 256 : 192;  // what stays LDS-resident next to two 32-row tiles of the other side
  return a.d_att == nullptr && a.d_lse == nullptr && a.drop.p <= 0.f && d_ok && a.nk >= 1 &&
         a.nk <= n_max && a.nq >= 1 && a.nq <= n_max &&
         a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.lddo % 8 == 0 && a.ldo % 4 == 0 && a.lddq % 4 == 0 &&
         a.lddk % 4 == 0 && a.lddv % 4 == 0 && al(a.q) && al(a.k) && al(a.v) && al(a.d_o) && al8(a.o) && al8(a.dq) &&
         al8(a.dk_) && al8(a.dv_) && a.lse && a.delta;
}

int mfma_attention_bwd(const AttnBwdArgs& a, hipStream_t st) { return launch_bwd(a, st); }

bool mfma_attention_supported(const AttnArgs& a) {
  auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  const bool d_ok = a.dk == a.dv && (a.dk == 64 || a.dk == 96 || a.dk == 128);
  return a.drop.p <= 0.f && d_ok && a.nk >= 1 && a.nk <= (a.dk == 64 ? 256 : 192) && a.nq >= 1 && a.ldq % 8 == 0 &&
         a.ldk % 8 == 0 &&
         a.ldv % 8 == 0 && a.ldo % 4 == 0 && al(a.q) && al(a.k) && al(a.v) && (((uintptr_t)a.o & 7) == 0);
}

bool mfma_attention_qkv_supported(const AttnArgs& a, int64_t Dm, int64_t ldx, int64_t ldqkv, const void* x, const void* w,
                                  const void* qkv) {
  auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  return a.dk == 64 && a.dv == 64 && a.nq == a.nk && a.nq >= 1 && a.nq <= 128 && Dm % 32 == 0 && Dm >= 32 && ldx % 8 == 0 &&
         ldqkv % 8 == 0 && a.ldo % 4 == 0 && a.msq == 0 && a.att == nullptr && a.drop.p <= 0.f && al(x) && al(w) &&
         al(qkv) && (((uintptr_t)a.o & 7) == 0);
}

int mfma_attention_qkv_fwd(const AttnArgs& a, const void* x, int64_t ldx, const void* w, const float* bias, void* qkv,
                           int64_t ldqkv, int64_t Dm, hipStream_t st) {
  QkvAttnArgs g{(const bf16*)x, ldx, (const bf16*)w, bias, (bf16*)qkv, ldqkv, a, (int)Dm};
  // the key mask row (zeros when there is no mask, -inf beyond n) always lives in LDS: ROWMASK = true
#define OVQA_QKV(RPV, SV, BKV, NBV)                                                                                \
  {                                                                                                                \
    constexpr int Mv = RPV * SV;                                                                                   \
    const size_t stage = (size_t)(192 + Mv) * BKV * 2, images = (size_t)SV * 3 * RPV * 128 + (size_t)SV * RPV * 4; \
    const size_t lds = NBV * stage > images ? NBV * stage : images;                                                \
    const dim3 grid((unsigned)((a.B + SV - 1) / SV), (unsigned)a.H);                                               \
    int rc = ensure_lds(attn_qkv_fwd_mfma_kernel<RPV, SV, true, BKV, NBV>, lds, "attention_qkv_fwd");              \
    if (rc != OVQA_OK) return rc;                                                                                  \
    OVQA_LAUNCH_TIMED((attn_qkv_fwd_mfma_kernel<RPV, SV, true, BKV, NBV>), grid, dim3(512), lds, st, g);          \
  }
  const bool k64 = Dm % 64 == 0;
  // short sequences: 4 samples per workgroup, or 2 when 4 would leave CUs without a workgroup (64 samples x 8 heads:
  // 128 -> 256 workgroups, 3.455 -> 3.442 ms per MCAN step).  The two-sample form keeps 96 KB in flight per CU: K steps of
  // 64 in a ring of 4 (round 5, same box: 10.09 us per launch against 10.7-11.0 with K steps of 32 in a ring of 4, 10.25
  // with 64 / ring of 3, 10.8 with 32 / ring of 6; the K loop is 5.7 of the kernel's 7.9 us per workgroup at 45 GB/s per CU)
  if (a.nq <= 32 && (int64_t)((a.B + 3) / 4) * a.H < 256) {
    if (k64) OVQA_QKV(32, 2, 64, 4)
    else OVQA_QKV(32, 2, 32, 4)
  }
  else if (a.nq <= 32) OVQA_QKV(32, 4, 32, 4)
  else if (a.nq <= 64) OVQA_QKV(64, 4, 32, 4)
  // 65-128 positions (100 regions): one sample per workgroup, K steps of 64, two workgroups per CU: 3.408 -> 3.382 ms
  // per MCAN step (two samples per workgroup with K steps of 64: 3.390; one sample with K steps of 32 in a ring of 4 / 3:
  // 28.7 / 28.8 against 26.7 us per launch)
  else if (k64) OVQA_QKV(128, 1, 64, 2)
  else OVQA_QKV(128, 2, 32, 4)
#undef OVQA_QKV
  return ovqa_check_launch("attention_qkv_fwd(mfma)");
}

bool mfma_attention_q_supported(const AttnArgs& a, int64_t Dm, int64_t ldx, int64_t ldq, const void* x, const void* w,
                                const void* q) {
  auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  return a.dk == 64 && a.dv == 64 && a.nq >= 1 && a.nq <= 128 && a.nk >= 1 && a.nk <= 128 && Dm % 64 == 0 && Dm >= 64 &&
         ldx % 8 == 0 && ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 4 == 0 && a.msq == 0 &&
         a.att == nullptr && a.drop.p <= 0.f && al(x) && al(w) && al(q) && al(a.k) && al(a.v) &&
         (((uintptr_t)a.o & 7) == 0);
}

int mfma_attention_q_fwd(const AttnArgs& a, const void* x, int64_t ldx, const void* w, const float* bias, void* q,
                         int64_t ldq, int64_t Dm, hipStream_t st) {
  QAttnArgs g{(const bf16*)x, ldx, (const bf16*)w, bias, (bf16*)q, ldq, a, (int)Dm};
  // two heads per 16-wave workgroup where the head count is even, ring of three K steps in the projection loop (a ring of
  // two and one head per 8-wave workgroup both measured slower in the step, rounds 3-4)
  const int NHv = a.H % 2 == 0 ? 2 : 1;
  const dim3 grid((unsigned)a.B, (unsigned)(a.H / NHv));
#define OVQA_QATT_L(NKTV, NBV, NHV)                                                                            \
  {                                                                                                            \
    const size_t stage = (size_t)(64 * NHV + 128) * 64 * 2;                                                    \
    const size_t images = (size_t)NHV * ((128 + 2 * NKTV * 32) * 128 + NKTV * 32 * 4);                         \
    const size_t lds = NBV * stage > images ? NBV * stage : images;                                            \
    int rc = ensure_lds(attn_q_fwd_mfma_kernel<NKTV, NBV, NHV>, lds, "attention_q_fwd");                       \
    if (rc != OVQA_OK) return rc;                                                                              \
    OVQA_LAUNCH_TIMED((attn_q_fwd_mfma_kernel<NKTV, NBV, NHV>), grid, dim3(512 * NHV), lds, st, g);           \
  }
#define OVQA_QATT(NKTV)                     \
  {                                         \
    if (NHv == 2) OVQA_QATT_L(NKTV, 3, 2)   \
    else OVQA_QATT_L(NKTV, 3, 1)            \
  }
  if (a.nk <= 32) OVQA_QATT(1)
  else if (a.nk <= 64) OVQA_QATT(2)
  else OVQA_QATT(4)
#undef OVQA_QATT
#undef OVQA_QATT_L
  return ovqa_check_launch("attention_q_fwd(mfma)");
}

bool mfma_attention_bwd_do_supported(const AttnBwdArgs& a, int64_t Dm, int64_t lddy, int64_t ldwt, const void* dy,
                                     const void* wt) {
  auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  const bool guided = a.nq > 64 && a.nq <= 128, single = a.nq >= 1 && a.nq <= 32 && a.H % 2 == 0;
  // (round 5) 97-128 queries x 97-128 keys, the image self-attention: the role-split backward with the projection inside
  const bool roles = a.nq > 96 && a.nq <= 128 && a.nk > 96 && a.nk <= 128;
  return a.dk == 64 && a.dv == 64 && (roles || ((guided || single) && a.nk >= 1 && a.nk <= 32)) && a.msq == 0 &&
         a.d_att == nullptr && a.d_lse == nullptr && a.drop.p <= 0.f && Dm % 64 == 0 && Dm >= 64 && lddy % 8 == 0 &&
         ldwt % 8 == 0 && a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 && a.lddq % 4 == 0 &&
         a.lddk % 4 == 0 && a.lddv % 4 == 0 && al(dy) && al(wt) && al(a.q) && al(a.k) && al(a.v) && al(a.o) &&
         (a.o_lo == nullptr || al(a.o_lo)) && a.lse != nullptr;
}

int mfma_attention_bwd_do(const AttnBwdArgs& a, const void* dy, int64_t lddy, const void* wt, int64_t ldwt, int64_t Dm,
                          hipStream_t st) {
  DoBwdArgs g{(const bf16*)dy, lddy, (const bf16*)wt, ldwt, a, (int)Dm};
  if (a.nk > 32) {  // the role-split form (image self-attention)
    const size_t img = (size_t)(2 * 128 + 2 * 128) * 128 + (size_t)(128 + 2 * 128) * 4;
    const size_t stage = (size_t)(64 + 128) * 64 * 2;
    // ring of three K steps in the projection loop, images filled by direct-to-LDS loads (round 5, same box: 25.6 us per
    // launch; ring of 2: 28.7; staging requests in front of the projection loop: 27.8; images through registers: 26.1)
    const size_t lds = 3 * stage > img ? 3 * stage : img;
    int rc = ensure_lds(attn_bwd_roles_mfma_kernel<4, 4, true, 3, true>, lds, "attention_bwd_do(roles)");
    if (rc != OVQA_OK) return rc;
    OVQA_LAUNCH_TIMED((attn_bwd_roles_mfma_kernel<4, 4, true, 3, true>), dim3((unsigned)a.B, (unsigned)a.H), dim3(512), lds, st, g, 4, 4);
    return ovqa_check_launch("attention_bwd_do(mfma,roles)");
  }
  if (a.nq <= 32) {  // single query tile (20 x 20): two heads per 4-wave workgroup
    const size_t stage1 = (size_t)(128 + 32) * 64 * 2;
    const size_t prob1 = (size_t)(2 * 32 + 2 * 32) * 128 + 32 * 4 + 2 * 32 * 4 + 4096 * 4;
    // ring of four K steps of 20 KB in the projection loop (in the MCAN step 9.9 us per launch against 10.35-10.54 with
    // three, 10.6 with five: round 5)
    const size_t ldsn = 4 * stage1 > 2 * prob1 ? 4 * stage1 : 2 * prob1;
    int rc = ensure_lds(attn_bwd_do_smallk1_mfma_kernel<4>, ldsn, "attention_bwd_do");
    if (rc != OVQA_OK) return rc;
    OVQA_LAUNCH_TIMED((attn_bwd_do_smallk1_mfma_kernel<4>), dim3((unsigned)a.B, (unsigned)(a.H / 2)), dim3(256), ldsn, st, g);
    return ovqa_check_launch("attention_bwd_do(mfma)");
  }
  const size_t prob = (size_t)(2 * 128 + 2 * 32) * 128 + 32 * 4 + 2 * 128 * 4 + 4096 * 4;
  // one head per 8-wave workgroup, two workgroups per CU (two heads per 16-wave workgroup: 3.265 / 3.256 against
  // 3.241 / 3.239 ms per MCAN step since the staging became one round of loads)
  {
    const size_t stage = (size_t)(64 + 128) * 64 * 2;
    const size_t lds = 3 * stage > prob ? 3 * stage : prob;
    int rc = ensure_lds(attn_bwd_do_smallk_mfma_kernel<3, 1>, lds, "attention_bwd_do");
    if (rc != OVQA_OK) return rc;
    OVQA_LAUNCH_TIMED((attn_bwd_do_smallk_mfma_kernel<3, 1>), dim3((unsigned)a.B, (unsigned)a.H), dim3(512), lds, st, g);
  }
  return ovqa_check_launch("attention_bwd_do(mfma)");
}

int mfma_attention_fwd(const AttnArgs& a, hipStream_t st) {
  if (a.dk == 96) return launch_fwd_d<96>(a, st);
  if (a.dk == 128) return launch_fwd_d<128>(a, st);
  return launch_fwd_d<64>(a, st);
}

}  // namespace ovqa
