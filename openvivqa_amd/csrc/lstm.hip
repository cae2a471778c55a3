// LSTM recurrence of LSTMTextEmbedding (models/modules/text_embeddings.py:236,243: torch.nn.LSTM, one layer,
// batch_first, zero initial state; gate order i, f, g, o) -- forward and backward, each as ONE persistent launch.
//
// Why one launch: the recurrence is T = 20 dependent steps of a [B, 512] x [512, 2048] product (134 MFLOP) and a few
// thousand transcendentals -- MIOpen runs it as ~45 launches forward and ~65 backward (two small GEMMs + an update
// kernel per step, 0.95 ms of the 5.1 ms model-level step, profiles/r05a_model_kernel_stats.csv).  Here the step is
// split over workgroups that never leave the chip:
//
//   * workgroup (sg, ub) owns 16 samples x 16 hidden units, i.e. 64 gate columns; B / 16 x 32 workgroups of four waves
//     (128 for B = 64), one per CU (96 KiB of LDS are declared to keep two from sharing a CU: the hand-off below is
//     measured for that placement, MI355X_MICROARCH.md "Valid forms");
//   * a wave's slice of BOTH weight matrices (16 gate rows x 512 inputs each: 64 + 64 VGPRs) is loaded ONCE into
//     registers in MFMA A-operand layout and stays there for all T steps: neither LDS nor L2 sees the weights again;
//   * the gate rows of a wave are ordered (unit, gate) so that the 16x16x32 accumulator of lane (sample n, quad q)
//     holds i, f, g, o of ONE (sample, unit): the cell update is register-local, c_t never leaves its lane;
//   * the input half x_t W_ih^T does not depend on the recurrence: it is computed one step ahead (operands of step
//     t + 2 in flight), so a step's critical chain is wait -> 16 KB of h_{t-1} -> 16 MFMAs -> gates -> publish;
//   * h_t travels between the 32 workgroups of a sample group through `hseq` (the [T+1, B, 512] bf16 sequence that is
//     also the operand of the W_hh weight gradient): write-through (sc1) 16-byte stores, and "the data is the flag" (below):
//     no counter, no drain -- the consumer re-reads the payload with sc1 loads until no unit shows the sentinel the buffer
//     was filled with.  (Rounds 4-5 also carried the counter form of cdna_hip_programming.md Guideline 16 -- drained stores,
//     one arrival counter per sample group and step, a polling lane, a workgroup barrier -- with and without the agent-scope
//     acquire: 104 / 127 and 120 / 146 us forward / backward against 77 / 81 for this form, identical bits; removed in
//     round 6, when the give-up path got a host-visible status word and a test of its own.)
//
// Backward is the same structure in reverse: workgroup (sg, ub) owns dh for 16 samples x 16 units; dh_{t} needs
// dgates_{t+1} W_hh over ALL 2048 gate columns of its samples, exchanged through `dgates` itself ([T, B, 2048] bf16,
// which the dX / dW GEMMs read afterwards); wave w holds the rows of W_hh^T for gate w (from the arena's transposed
// shadow) and reduces over that gate's 512 columns; the four partial tiles meet in LDS.
//
// Any other hidden size and the fp32 mode run one small VALU launch per step (`lstm_step_*_simple`): same results, no
// cross-workgroup protocol.  A batch that is not a multiple of 16 samples, or larger than the device's CUs can keep
// co-resident (16 samples per 32 CUs), is padded / split by the host side (ops.lstm_fwd).
#include "common.h"
#include "kernels.h"

namespace ovqa {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int LH = 512;         // hidden (and input) size the persistent kernels are built for
constexpr int KS = LH / 32;     // K steps of the 16x16x32 MFMA over one 512-long reduction
constexpr int NUB = LH / 16;    // workgroups per sample group
constexpr int SPIN_LIMIT = 1 << 21;

// Process-lifetime status word (round 6): every give-up of a hand-off wait ORs its code in here as well as into the
// call's own scratch word, so that the host finds it without knowing which call's scratch to look at
// (ovqa_lstm_status reads and clears it).  A give-up also shows in the data: the wave goes on with the 0xFFFF sentinels
// it read -- bf16 NaNs -- so h, y and every later step of the sample group are NaN.
__device__ unsigned g_lstm_status;
__device__ __forceinline__ void give_up(unsigned* status, unsigned code) {
  __hip_atomic_store(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_or(&g_lstm_status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ float sigmoid_f(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) { return 2.f * __frcp_rn(1.f + __expf(-2.f * x)) - 1.f; }

// workgroup -> (sample group, unit block): the 32 workgroups of a sample group share as few XCDs as the group count
// allows (blocks b and b + 8 share an XCD); placement is a speed matter only, never correctness
__device__ __forceinline__ void wg_role(int b, int nsg, int& sg, int& ub) {
  if (nsg <= 8 && 8 % nsg == 0) {
    const int per = 8 / nsg, xcd = b % 8, idx = b / 8;
    sg = xcd / per;
    ub = idx * per + xcd % per;
  } else {
    sg = b / NUB;
    ub = b % NUB;
  }
}

// ---- the hand-off: the data is the flag ------------------------------------------------------------------
// The exchange buffer is filled with 0xFF bytes by a memset node in front of the launch: 0xFFFF is a bf16 NaN that neither
// h = o tanh(c) nor a finite gradient can be.  A consumer wave simply re-reads its 16 KB with sc1 loads until no 8-byte
// unit shows the sentinel (cdna_hip_programming.md Guideline 16, recipe R2: 8-byte granules written by ONE sc1 store are
// observed untorn; here the tag is "not 0xFFFF" in the unit's first element, so the payload is not doubled).  Against
// the counter form this drops, per step: the producer's store drain, the atomic, the poll round trip and a workgroup
// barrier -- two of the four dependent memory round trips (MEASURED, scripts/lstm_bench.py, B = 64, T = 20, memset nodes
// included: forward 104 -> 77 us, backward 127 -> 81 us; with the acquire fence 120 / 146).
constexpr int SWEEP_LIMIT = 1 << 17;  // default of the kernels' `sweep_limit` argument (OVQA_LSTM_SWEEP_LIMIT: tests)
// aux of the hand-off loads: sc1 (bit 4) + LLVM's volatile marker (bit 31, stripped at lowering): two sweeps of the same
// addresses are two loads, and none is hoisted out of a polling loop
constexpr int AUX_POLL = (int)(0x80000000u | 16u);
__device__ __forceinline__ bool unit_ready(const u32x4& v) {  // both 8-byte halves of a 16-byte piece
  return ((v[0] & 0xFFFFu) != 0xFFFFu) & ((v[2] & 0xFFFFu) != 0xFFFFu);
}
template <typename RS>
__device__ __forceinline__ void sweep_until_ready(const RS& rs, unsigned off, u32x4 (&v)[KS], unsigned* status, int limit) {
  for (int spins = 0;;) {
    bool ok = true;
#pragma unroll
    for (int kk = 0; kk < KS; kk++) v[kk] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + kk * 64, 0, AUX_POLL);
#pragma unroll
    for (int kk = 0; kk < KS; kk++) ok &= unit_ready(v[kk]);
    if (__all(ok)) return;
    if (++spins > limit) {  // a producer workgroup never ran (or produced the NaN pattern itself): give up loudly
      give_up(status, 2u);
      return;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

struct LstmFwdArgs {
  const bf16* x;        // [T*B][512] time-major (row t*B + b), row stride ldx
  int64_t ldx;
  const bf16* w_ih;     // [2048][512]
  const bf16* w_hh;     // [2048][512]
  const float* b_ih;    // [2048]
  const float* b_hh;    // [2048]
  float* y;             // [B][T][512] fp32
  bf16* y16;            // the same in bf16 (the GEMM / LayerNorm operand of the stack behind it), or NULL
  bf16* hseq;           // [(T+1)*B][512] time-major: block 0 = zeros, block t+1 = h_t
  float* saved;         // [T][nwg][5][256]: i, f, g, o (post-activation), c_t of the workgroup's lanes
  unsigned* status;
  int B, T;
  int sweep_limit;  // sweeps of the sentinel form before a wait gives up
  int drop_wg;      // tests only (OVQA_LSTM_DEBUG_DROP_WG): workgroup drop_wg - 1 exits at once, as if it never became resident
  unsigned* probe;  // diagnostic (OVQA_LSTM_PROBE=1): 100 MHz stamps of one wave's phases, 8 words per step; else NULL
};
#define LSTM_STAMP(slot)                                                                                   \
  do {                                                                                                     \
    if (a.probe != nullptr && blockIdx.x == 0 && tid == 128)                                               \
      a.probe[t * 8 + (slot)] = (unsigned)__builtin_amdgcn_s_memrealtime();                                \
  } while (0)

__global__ __launch_bounds__(256) void lstm_fwd_persistent_kernel(LstmFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  // h_t tiles, double-buffered by step parity: with the sentinel hand-off ONE barrier per step is left, which orders the
  // writes of step t + 2 behind the reads of step t, not those of step t + 1
  bf16* tile16_0 = reinterpret_cast<bf16*>(lds_raw);          // [2][16 samples][16 units] bf16: h_t for the exchange
  float* tile32_0 = reinterpret_cast<float*>(lds_raw + 1024); // [2][16 samples][16 units] fp32: h_t for y
  unsigned char* hbuf = lds_raw + 4096;                       // [16 samples][HB_STRIDE bytes]: h_{t-1}, shared by the waves
  constexpr int HB_STRIDE = 1024 + 16;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, q = l >> 4;
  const int nsg = a.B / 16, nwg = gridDim.x;
  int sg, ub;
  wg_role(blockIdx.x, nsg, sg, ub);
  const int B = a.B, T = a.T;
  if (blockIdx.x == 0 && tid == 0)  // (a give-up is reported after >= 1e5 sweeps: long after this store)
    __hip_atomic_store(a.status, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a.drop_wg != 0 && (int)blockIdx.x == a.drop_wg - 1) return;  // (workgroup-uniform; tests of the give-up path)
  // ---- this wave's 16 gate rows of W_ih and W_hh, A-operand layout, resident for the whole launch
  // A row r = l & 15 -> (unit 4 w + (r >> 2), gate r & 3): C row 4 q + reg = (unit 4 w + q, gate reg)
  const int arow = (n & 3) * LH + ub * 16 + w * 4 + (n >> 2);
  bf16x8 wih[KS], whh[KS];
#pragma unroll
  for (int kk = 0; kk < KS; kk++) {
    wih[kk] = *reinterpret_cast<const bf16x8*>(a.w_ih + (int64_t)arow * LH + kk * 32 + q * 8);
    whh[kk] = *reinterpret_cast<const bf16x8*>(a.w_hh + (int64_t)arow * LH + kk * 32 + q * 8);
  }
  const int unit = ub * 16 + w * 4 + q;  // this lane's hidden unit; its sample is sg * 16 + n
  f32x4 bias;
#pragma unroll
  for (int g = 0; g < 4; g++) bias[g] = a.b_ih[g * LH + unit] + a.b_hh[g * LH + unit];
  // block 0 of hseq (h_{-1} = 0): the operand of the W_hh gradient's first time step
  if (w == 0 && l < 32)
    *reinterpret_cast<u32x4*>(a.hseq + (int64_t)(sg * 16 + (l >> 1)) * LH + ub * 16 + (l & 1) * 8) = u32x4{0u, 0u, 0u, 0u};
  // exchange descriptors (wave-uniform): all of hseq
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.hseq, 0, (int)((int64_t)(T + 1) * B * LH * 2), 0x00020000);
  // B operand of the input half: x[t*B + sg*16 + n][kk*32 + q*8 ..]
  const bf16* xrow = a.x + (int64_t)(sg * 16 + n) * a.ldx + q * 8;
  bf16x8 xf[KS];
#pragma unroll
  for (int kk = 0; kk < KS; kk++) xf[kk] = *reinterpret_cast<const bf16x8*>(xrow + kk * 32);
  // (two accumulators per product: a 16-deep chain on ONE accumulator is 16 dependent 8-pass MFMAs)
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 accx = bias, accx1 = zero4;
#pragma unroll
  for (int kk = 0; kk < KS; kk += 2) {
    accx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wih[kk], xf[kk], accx, 0, 0, 0);
    accx1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wih[kk + 1], xf[kk + 1], accx1, 0, 0, 0);
  }
  accx += accx1;
  if (T > 1) {
#pragma unroll
    for (int kk = 0; kk < KS; kk++) xf[kk] = *reinterpret_cast<const bf16x8*>(xrow + (int64_t)B * a.ldx + kk * 32);
  }
  float c = 0.f;
  for (int t = 0; t < T; t++) {
    f32x4 acc = accx;
    LSTM_STAMP(0);
    if (t > 0) {
      // h_{t-1} of the 16 samples: block t of hseq, sc1 loads straight into the B-operand layout
      const unsigned hoff = (unsigned)((((int64_t)t * B + sg * 16 + n) * LH + q * 8) * 2);
      u32x4 hf[KS];
      {
        // the four waves need the SAME 16 KB: each sweeps a quarter (K steps 4 w .. 4 w + 3) until it is ready and
        // shares it through LDS -- four full sweeps per CU were bound by the CU's fetch path (64 KB at ~60 GB/s: 1.1 us
        // per sweep, scripts/lstm_probe.py), a quarter each is a plain round trip
        // One sweep at a time, issued here.  MEASURED alternatives (scripts/lstm_probe.py, lstm_bench.py; B = 64, T = 20):
        // two sweeps in flight half a round trip apart, and a first sweep issued right behind this workgroup's own
        // publish -- both slower (forward 101-111 us against 68): whatever the compiler waits for behind a polling loop
        // (the input half's operands, requested in front of it) it waits for with vmcnt(0), i.e. also for the sweep
        // that is deliberately in flight.
        u32x4 pq[4];
        for (int spins = 0;;) {
          bool ok = true;
#pragma unroll
          for (int j = 0; j < 4; j++) pq[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, hoff + (4 * w + j) * 64, 0, AUX_POLL);
#pragma unroll
          for (int j = 0; j < 4; j++) ok &= unit_ready(pq[j]);
          if (__all(ok)) break;
          if (++spins > a.sweep_limit) {
            give_up(a.status, 2u);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) *reinterpret_cast<u32x4*>(hbuf + n * HB_STRIDE + (4 * w + j) * 64 + q * 16) = pq[j];
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KS; kk++) hf[kk] = *reinterpret_cast<const u32x4*>(hbuf + n * HB_STRIDE + kk * 64 + q * 16);
      }
      f32x4 acc1 = zero4;
#pragma unroll
      for (int kk = 0; kk < KS; kk += 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whh[kk], __builtin_bit_cast(bf16x8, hf[kk]), acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whh[kk + 1], __builtin_bit_cast(bf16x8, hf[kk + 1]), acc1, 0, 0, 0);
      }
      LSTM_STAMP(1);
      acc += acc1;
    }
    const float ig = sigmoid_f(acc[0]), fg = sigmoid_f(acc[1]), gg = tanh_f(acc[2]), og = sigmoid_f(acc[3]);
    c = fg * c + ig * gg;
    const float h = og * tanh_f(c);
    bf16* tile16 = tile16_0 + (t & 1) * 256;
    float* tile32 = tile32_0 + (t & 1) * 256;
    tile16[n * 16 + w * 4 + q] = (bf16)h;
    tile32[n * 16 + w * 4 + q] = h;
    LSTM_STAMP(2);
    __syncthreads();
    LSTM_STAMP(3);
    if (w == 0) {
      if (l < 32) {  // lane -> (sample l >> 1, half l & 1): 16 bytes of the row's 32
        const u32x4 v = *reinterpret_cast<const u32x4*>(tile16 + (l >> 1) * 16 + (l & 1) * 8);
        const unsigned off = (unsigned)((((int64_t)(t + 1) * B + sg * 16 + (l >> 1)) * LH + ub * 16 + (l & 1) * 8) * 2);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
      }
    } else if (w == 1) {  // y[b][t][ub*16 ..]: lane -> (sample l >> 2, 16-byte quarter l & 3)
      const f32x4 v = *reinterpret_cast<const f32x4*>(tile32 + (l >> 2) * 16 + (l & 3) * 4);
      *reinterpret_cast<f32x4*>(a.y + ((int64_t)(sg * 16 + (l >> 2)) * T + t) * LH + ub * 16 + (l & 3) * 4) = v;
    } else if (w == 2 && a.y16 != nullptr && l < 32) {  // the bf16 twin of y, batch-major like y
      const u32x4 v = *reinterpret_cast<const u32x4*>(tile16 + (l >> 1) * 16 + (l & 1) * 8);
      *reinterpret_cast<u32x4*>(a.y16 + ((int64_t)(sg * 16 + (l >> 1)) * T + t) * LH + ub * 16 + (l & 1) * 8) = v;
    }
    LSTM_STAMP(4);
    if (t + 1 < T) {  // the input half of step t + 1 (its operands were requested a step ago), then request t + 2
      accx = bias;
      accx1 = zero4;
#pragma unroll
      for (int kk = 0; kk < KS; kk += 2) {
        accx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wih[kk], xf[kk], accx, 0, 0, 0);
        accx1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wih[kk + 1], xf[kk + 1], accx1, 0, 0, 0);
      }
      accx += accx1;
      if (t + 2 < T) {
#pragma unroll
        for (int kk = 0; kk < KS; kk++)
          xf[kk] = *reinterpret_cast<const bf16x8*>(xrow + (int64_t)(t + 2) * B * a.ldx + kk * 32);
      }
    }
    {  // what backward needs, LAST: behind a loop the compiler waits for every outstanding memory operation before the
      // first use of a loaded register, so these stores drain beside the next step's sweep, not in front of the MFMAs above
      float* sv = a.saved + ((int64_t)t * nwg + blockIdx.x) * 5 * 256 + tid;
      store_saved(sv, ig);
      store_saved(sv + 256, fg);
      store_saved(sv + 512, gg);
      store_saved(sv + 768, og);
      store_saved(sv + 1024, c);
    }
    LSTM_STAMP(5);
  }
}

struct LstmBwdArgs {
  const void* dy;       // [B][T][512] fp32, or bf16 when dy_bf16 (the gradient arrives from a bf16 LayerNorm backward)
  int dy_bf16;
  const bf16* whh_t;    // transposed W_hh: row u' (input unit), 2048 gate columns, row stride ldwt
  int64_t ldwt;
  const float* saved;   // as written by the forward kernel
  bf16* dgates;         // [T*B][2048] time-major, columns gate*512 + unit: output AND exchange buffer
  unsigned* status;
  int B, T;
  int sweep_limit, drop_wg;  // as in LstmFwdArgs
};

__global__ __launch_bounds__(256) void lstm_bwd_persistent_kernel(LstmBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* part = reinterpret_cast<float*>(lds_raw);              // [4 waves][4 regs][64 lanes] fp32 partial dh tiles
  bf16* tile = reinterpret_cast<bf16*>(lds_raw + 4096);         // [16 samples][4 gates][16 units] bf16
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, q = l >> 4;
  const int nsg = a.B / 16, nwg = gridDim.x;
  int sg, ub;
  wg_role(blockIdx.x, nsg, sg, ub);
  const int B = a.B, T = a.T;
  if (blockIdx.x == 0 && tid == 0)
    __hip_atomic_store(a.status, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a.drop_wg != 0 && (int)blockIdx.x == a.drop_wg - 1) return;
  // A operand of the recurrent product dh[u'][n] = sum_col W_hh[col][u'] dgates[n][col]: wave w reduces over gate w's
  // 512 columns; A row = l & 15 -> input unit ub*16 + (l & 15), k = w*512 + kk*32 + q*8 ..
  bf16x8 wt[KS];
#pragma unroll
  for (int kk = 0; kk < KS; kk++)
    wt[kk] = *reinterpret_cast<const bf16x8*>(a.whh_t + (int64_t)(ub * 16 + n) * a.ldwt + w * LH + kk * 32 + q * 8);
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.dgates, 0, (int)((int64_t)T * B * 4 * LH * 2), 0x00020000);
  // elementwise role of this lane: sample n, unit_in = 4 q + w (the rows 4 q + reg of the accumulator tile, reg = w);
  // the forward kernel saved that (sample, unit) from its thread q * 64 + w * 16 + n
  const int unit_in = 4 * q + w;
  const int ftid = q * 64 + w * 16 + n;
  const float* sv0 = a.saved + (int64_t)blockIdx.x * 5 * 256 + ftid;
  const int64_t sv_step = (int64_t)nwg * 5 * 256;
  const int64_t dy0 = (int64_t)(sg * 16 + n) * T * LH + ub * 16 + unit_in;
  auto dy_at = [&](int t) {
    return a.dy_bf16 ? (float)((const bf16*)a.dy)[dy0 + (int64_t)t * LH] : ((const float*)a.dy)[dy0 + (int64_t)t * LH];
  };
  float dc_carry = 0.f;  // dc_{t+1} * f_{t+1}
  // operands of step t that do not depend on the recurrence are requested one step ahead
  float ig, fg, gg, og, ct, cprev, dyv;
  {
    const float* sv = sv0 + (int64_t)(T - 1) * sv_step;
    ig = sv[0]; fg = sv[256]; gg = sv[512]; og = sv[768]; ct = sv[1024];
    cprev = T > 1 ? (sv - sv_step)[1024] : 0.f;
    dyv = dy_at(T - 1);
  }
  for (int t = T - 1; t >= 0; t--) {
    float dh = dyv;
    if (t < T - 1) {
      const unsigned goff = (unsigned)((((int64_t)(t + 1) * B + sg * 16 + n) * (4 * LH) + w * LH + q * 8) * 2);
      u32x4 gf[KS];
      {
        // a full sweep is 64 KB per CU (every wave its own gate's columns): 1.1 us on the CU's fetch path, so a sweep that
        // comes too early is expensive.  Probe first: one 16-byte piece per lane, chosen so that the wave sees a piece of
        // every one of the 32 producers (lane (n, q) reads K step n: producer 2 n + q / 2), re-read until none shows the
        // sentinel; the full sweep behind it still validates every unit
        for (int spins = 0;;) {
          const u32x4 pv = __builtin_amdgcn_raw_buffer_load_b128(rs, goff + n * 64, 0, AUX_POLL);
          if (__all(unit_ready(pv))) break;
          if (++spins > a.sweep_limit) break;  // (the sweep below reports it)
          __builtin_amdgcn_s_sleep(1);
        }
        sweep_until_ready(rs, goff, gf, a.status, a.sweep_limit);
      }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KS; kk += 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[kk], __builtin_bit_cast(bf16x8, gf[kk]), acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[kk + 1], __builtin_bit_cast(bf16x8, gf[kk + 1]), acc1, 0, 0, 0);
      }
      acc += acc1;
#pragma unroll
      for (int r = 0; r < 4; r++) part[(w * 4 + r) * 64 + l] = acc[r];
      __syncthreads();
      dh += (part[(0 * 4 + w) * 64 + l] + part[(1 * 4 + w) * 64 + l]) + (part[(2 * 4 + w) * 64 + l] + part[(3 * 4 + w) * 64 + l]);
    }
    const float tc = tanh_f(ct);
    const float d_o = dh * tc * og * (1.f - og);
    const float dc = dh * og * (1.f - tc * tc) + dc_carry;
    const float d_i = dc * gg * ig * (1.f - ig);
    const float d_g = dc * ig * (1.f - gg * gg);
    const float d_f = dc * cprev * fg * (1.f - fg);
    dc_carry = dc * fg;
    bf16* tp = tile + n * 64 + unit_in;
    tp[0] = (bf16)d_i;
    tp[16] = (bf16)d_f;
    tp[32] = (bf16)d_g;
    tp[48] = (bf16)d_o;
    // next (earlier) step's saved operands: in flight while this step publishes and the others catch up
    if (t > 0) {
      const float* sv = sv0 + (int64_t)(t - 1) * sv_step;
      ig = sv[0]; fg = sv[256]; gg = sv[512]; og = sv[768]; ct = sv[1024];
      cprev = t > 1 ? (sv - sv_step)[1024] : 0.f;
      dyv = dy_at(t - 1);
    }
    __syncthreads();
    if (w == 0) {  // 16 samples x 4 gates x 32 bytes: two 16-byte stores per lane
#pragma unroll
      for (int rep = 0; rep < 2; rep++) {
        const int e = rep * 64 + l, s = e >> 3, g = (e >> 1) & 3, hlf = e & 1;
        const u32x4 v = *reinterpret_cast<const u32x4*>(tile + s * 64 + g * 16 + hlf * 8);
        const unsigned off = (unsigned)((((int64_t)t * B + sg * 16 + s) * (4 * LH) + g * LH + ub * 16 + hlf * 8) * 2);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
      }
    }
  }
}

// ---- one launch per time step: the fp32 mode and every shape the persistent kernels do not cover -----------------------
// forward: thread (b, u): gates = b_ih + b_hh + x_t[b] . W_ih[g*H+u] + h_{t-1}[b] . W_hh[g*H+u], fp32 accumulation (the
// input half is NOT rounded to the storage type in between, as in the persistent kernel); saved = post-activation
// gates [T*B][4H] fp32 followed by c [T*B][H] fp32
template <typename T>
__global__ __launch_bounds__(256) void lstm_step_fwd_simple(const T* __restrict__ x, int64_t ldx, const T* __restrict__ w_ih,
                                                            const float* __restrict__ b_ih, const T* __restrict__ w_hh,
                                                            const float* __restrict__ b_hh, T* __restrict__ hseq,
                                                            float* __restrict__ y, T* __restrict__ y16, float* __restrict__ gates,
                                                            float* __restrict__ cs, int B, int Tn, int I, int H, int t) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, u = idx % H;
  float acc[4];
#pragma unroll
  for (int g = 0; g < 4; g++) acc[g] = b_ih[g * H + u] + b_hh[g * H + u];
  const T* xp = x + ((int64_t)t * B + b) * ldx;
  for (int k = 0; k < I; k++) {
    const float xv = to_f32<T>(xp[k]);
#pragma unroll
    for (int g = 0; g < 4; g++) acc[g] = fmaf(xv, to_f32<T>(w_ih[((int64_t)g * H + u) * I + k]), acc[g]);
  }
  if (t > 0) {
    const T* hp = hseq + ((int64_t)t * B + b) * H;
    for (int k = 0; k < H; k++) {
      const float hv = to_f32<T>(hp[k]);
#pragma unroll
      for (int g = 0; g < 4; g++) acc[g] = fmaf(hv, to_f32<T>(w_hh[((int64_t)g * H + u) * H + k]), acc[g]);
    }
  } else {
    hseq[(int64_t)b * H + u] = from_f32<T>(0.f);
  }
  const float ig = 1.f / (1.f + expf(-acc[0])), fg = 1.f / (1.f + expf(-acc[1])), gg = tanhf(acc[2]),
              og = 1.f / (1.f + expf(-acc[3]));
  const float cp = t > 0 ? cs[((int64_t)(t - 1) * B + b) * H + u] : 0.f;
  const float c = fg * cp + ig * gg;
  const float h = og * tanhf(c);
  float* gp = gates + ((int64_t)t * B + b) * 4 * H + u;
  gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
  cs[((int64_t)t * B + b) * H + u] = c;
  hseq[((int64_t)(t + 1) * B + b) * H + u] = from_f32<T>(h);
  y[((int64_t)b * Tn + t) * H + u] = h;
  if (y16) y16[((int64_t)b * Tn + t) * H + u] = from_f32<T>(h);
}

template <typename T>
__global__ __launch_bounds__(256) void lstm_step_bwd_simple(const void* __restrict__ dy_, int dy_bf16,
                                                            const T* __restrict__ w_hh,
                                                            const float* __restrict__ gates, const float* __restrict__ cs,
                                                            T* __restrict__ dgates, float* __restrict__ dc_carry, int B,
                                                            int Tn, int H, int t) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, u = idx % H;
  const int64_t di = ((int64_t)b * Tn + t) * H + u;
  float dh = dy_bf16 ? (float)((const bf16*)dy_)[di] : ((const float*)dy_)[di];
  if (t < Tn - 1) {
    const T* gp = dgates + ((int64_t)(t + 1) * B + b) * 4 * H;
    float s = 0.f;
    for (int col = 0; col < 4 * H; col++) s = fmaf(to_f32<T>(gp[col]), to_f32<T>(w_hh[(int64_t)col * H + u]), s);
    dh += s;
  }
  const float* g = gates + ((int64_t)t * B + b) * 4 * H + u;
  const float ig = g[0], fg = g[H], gg = g[2 * H], og = g[3 * H];
  const float ct = cs[((int64_t)t * B + b) * H + u];
  const float cp = t > 0 ? cs[((int64_t)(t - 1) * B + b) * H + u] : 0.f;
  const float tc = tanhf(ct);
  const float carry = t < Tn - 1 ? dc_carry[idx] : 0.f;
  const float dc = dh * og * (1.f - tc * tc) + carry;
  dc_carry[idx] = dc * fg;
  T* dp = dgates + ((int64_t)t * B + b) * 4 * H + u;
  dp[0] = from_f32<T>(dc * gg * ig * (1.f - ig));
  dp[H] = from_f32<T>(dc * cp * fg * (1.f - fg));
  dp[2 * H] = from_f32<T>(dc * ig * (1.f - gg * gg));
  dp[3 * H] = from_f32<T>(dh * tc * og * (1.f - og));
}

int debug_env(const char* name, int dflt) {  // read per call: tests flip these
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

int device_cus() {
  static int cus = -1;
  if (cus < 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cus = 0;  // (unknown: the persistent route is not taken)
  }
  return cus;
}

constexpr int kSyncBytes = 4096;                 // counters (one 128-byte line per sample group) + status word
constexpr int kPersistentLds = 96 * 1024;        // more than half a CU's LDS: one workgroup per CU

}  // namespace

// The persistent kernels need the 32 workgroups of a sample group co-resident (96 KiB of LDS each: one per CU) -- the grid
// must fit the device's CUs (a partition or a part with fewer than 256 of them takes fewer samples per launch: the host
// side, ops.lstm_fwd, splits the batch into sample-group chunks that fit and pads it to whole groups of 16).
int64_t lstm_persistent_max_batch() { return (int64_t)(device_cus() / NUB) * 16; }

bool lstm_persistent_supported(int dtype, int64_t B, int64_t T, int64_t I, int64_t H, int64_t ldx) {
  return dtype == OVQA_BF16 && H == LH && I == LH && B >= 16 && B % 16 == 0 && B <= lstm_persistent_max_batch() && T >= 1 &&
         ldx % 8 == 0 && (int64_t)(T + 1) * B * LH * 8 < (1ll << 31);
}

int lstm_status_read_clear(unsigned* out, hipStream_t st) {
  unsigned v = 0;
  hipError_t e = hipMemcpyFromSymbolAsync(&v, HIP_SYMBOL(g_lstm_status), sizeof(v), 0, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e == hipSuccess && v != 0) {
    const unsigned zero = 0;
    e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_lstm_status), &zero, sizeof(zero), 0, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
  }
  OVQA_REQUIRE(e == hipSuccess, OVQA_ERR_LAUNCH, "lstm_status: %s", hipGetErrorString(e));
  *out = v;
  return OVQA_OK;
}

int64_t lstm_saved_bytes(int64_t B, int64_t T, int64_t H) { return 5 * T * B * H * 4; }

int64_t lstm_scratch_bytes(int64_t B, int64_t T, int64_t H) {
  (void)T;  // sync block + (per-step path) the carried dc [B][H]
  return kSyncBytes + B * H * 4;
}

int lstm_fwd(int dtype, bool persistent, const void* x, int64_t ldx, const void* w_ih, const void* w_hh,
             const float* b_ih, const float* b_hh, float* y, void* y16, void* hseq, void* saved, void* scratch, int64_t B, int64_t T,
             int64_t I, int64_t H, hipStream_t st) {
  if (persistent) {
    // ONE memset node, the sentinel fill (the status word is zeroed by the launch's first workgroup)
    hipError_t e = hipMemsetAsync(hseq, 0xFF, (size_t)((T + 1) * B * LH * 2), st);
    OVQA_REQUIRE(e == hipSuccess, OVQA_ERR_LAUNCH, "lstm_fwd: hipMemsetAsync: %s", hipGetErrorString(e));
    LstmFwdArgs a{(const bf16*)x, ldx, (const bf16*)w_ih, (const bf16*)w_hh, b_ih, b_hh, y, (bf16*)y16, (bf16*)hseq, (float*)saved,
                  (unsigned*)scratch + 1000, (int)B, (int)T,
                  debug_env("OVQA_LSTM_SWEEP_LIMIT", SWEEP_LIMIT), debug_env("OVQA_LSTM_DEBUG_DROP_WG", 0),
                  (getenv("OVQA_LSTM_PROBE") && T * 8 <= 480) ? (unsigned*)scratch + 512 : nullptr};
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)lstm_fwd_persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                kPersistentLds);
      attr_set = true;
    }
    hipLaunchKernelGGL(lstm_fwd_persistent_kernel, dim3((unsigned)((B / 16) * NUB)), dim3(256), kPersistentLds, st, a);
    return ovqa_check_launch("lstm_fwd(persistent)");
  }
  float* gates = (float*)saved;
  float* cs = gates + T * B * 4 * H;
  const int blocks = (int)((B * H + 255) / 256);
  for (int t = 0; t < T; t++) {
    if (dtype == OVQA_BF16)
      hipLaunchKernelGGL(lstm_step_fwd_simple<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)x, ldx, (const bf16*)w_ih,
                         b_ih, (const bf16*)w_hh, b_hh, (bf16*)hseq, y, (bf16*)y16, gates, cs, (int)B, (int)T, (int)I, (int)H, t);
    else
      hipLaunchKernelGGL(lstm_step_fwd_simple<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, ldx,
                         (const float*)w_ih, b_ih, (const float*)w_hh, b_hh, (float*)hseq, y, (float*)y16, gates, cs, (int)B,
                         (int)T, (int)I, (int)H, t);
  }
  return ovqa_check_launch("lstm_fwd(simple)");
}

int lstm_bwd(int dtype, bool persistent, const void* dy, int dy_bf16, const void* w_hh, const void* w_hh_t, int64_t ldwt,
             const void* saved, void* dgates, void* scratch, int64_t B, int64_t T, int64_t H, hipStream_t st) {
  if (persistent) {
    hipError_t e = hipMemsetAsync(dgates, 0xFF, (size_t)(T * B * 4 * LH * 2), st);
    OVQA_REQUIRE(e == hipSuccess, OVQA_ERR_LAUNCH, "lstm_bwd: hipMemsetAsync: %s", hipGetErrorString(e));
    LstmBwdArgs a{dy, dy_bf16, (const bf16*)w_hh_t, ldwt, (const float*)saved, (bf16*)dgates,
                  (unsigned*)scratch + 1000, (int)B, (int)T, debug_env("OVQA_LSTM_SWEEP_LIMIT", SWEEP_LIMIT),
                  debug_env("OVQA_LSTM_DEBUG_DROP_WG", 0)};
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)lstm_bwd_persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                kPersistentLds);
      attr_set = true;
    }
    hipLaunchKernelGGL(lstm_bwd_persistent_kernel, dim3((unsigned)((B / 16) * NUB)), dim3(256), kPersistentLds, st, a);
    return ovqa_check_launch("lstm_bwd(persistent)");
  }
  const float* gates = (const float*)saved;
  const float* cs = gates + T * B * 4 * H;
  float* carry = (float*)((unsigned char*)scratch + kSyncBytes);
  const int blocks = (int)((B * H + 255) / 256);
  for (int t = (int)T - 1; t >= 0; t--) {
    if (dtype == OVQA_BF16)
      hipLaunchKernelGGL(lstm_step_bwd_simple<bf16>, dim3(blocks), dim3(256), 0, st, dy, dy_bf16, (const bf16*)w_hh, gates, cs,
                         (bf16*)dgates, carry, (int)B, (int)T, (int)H, t);
    else
      hipLaunchKernelGGL(lstm_step_bwd_simple<float>, dim3(blocks), dim3(256), 0, st, dy, dy_bf16, (const float*)w_hh, gates, cs,
                         (float*)dgates, carry, (int)B, (int)T, (int)H, t);
  }
  return ovqa_check_launch("lstm_bwd(simple)");
}

}  // namespace ovqa
