/*
 * ovqa_hip.h -- C ABI of the MI355X (gfx950) cross-modal attention hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI: its
 * hot path is a chain of stock PyTorch ops inside Python modules.  Each entry
 * point below replaces one such chain and cites it (paths are relative to the
 * reference tree).  Signatures carry only plain pointers (device memory),
 * sizes, strides and a HIP stream handle -- no torch types -- so any host
 * (ctypes, pybind, cgo, JNI) can bind them; INTEGRATION.md shows the ctypes
 * stub the Python host side uses.
 *
 * Conventions
 *   - all tensors are row-major; `ld*` are row strides in ELEMENTS.
 *   - dtype: OVQA_F32 (exact fp32 kernels) or OVQA_BF16 (bf16 storage, fp32
 *     accumulation, MFMA contractions).  Weight/bias GRADIENTS, LayerNorm
 *     statistics, log-sum-exp and masks are always fp32.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *     Nothing here allocates, frees or synchronises: every entry point is
 *     capturable into a hipGraph.  Scratch comes from the caller (`ws`).
 *   - return value: OVQA_OK or a negative ovqa_status; ovqa_last_error() gives
 *     a human-readable message for the calling thread.
 *   - dropout: keep(element) is a pure function of (seed, site, *step_ptr,
 *     element index) -- see ovqa_dropout_keep_mask -- so backward regenerates
 *     the mask instead of storing it.  p == 0 disables it.
 */
#ifndef OVQA_HIP_H
#define OVQA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OVQA_ABI_VERSION 10

typedef enum {
  OVQA_OK = 0,
  OVQA_ERR_BAD_ARG = -1,      /* null pointer, negative size, bad enum            */
  OVQA_ERR_UNSUPPORTED = -2,  /* shape outside what the kernels cover (documented) */
  OVQA_ERR_LAUNCH = -3,       /* hipGetLastError() != hipSuccess after a launch    */
  OVQA_ERR_WORKSPACE = -4     /* caller-provided scratch too small                 */
} ovqa_status;

typedef enum { OVQA_F32 = 0, OVQA_BF16 = 1 } ovqa_dtype;

/* Epilogues of ovqa_linear_fwd. */
typedef enum {
  OVQA_EPI_BIAS = 0,           /* y = xW^T + b                                   */
  OVQA_EPI_BIAS_GELU = 1,      /* u = xW^T + b ; preact = u ; y = drop(gelu(u))  */
  OVQA_EPI_BIAS_RESIDUAL = 2   /* y = residual + drop(xW^T + b)                  */
} ovqa_epilogue;

typedef struct {
  float p;                 /* drop probability, 0 = off                        */
  uint32_t seed;           /* per-process seed                                 */
  uint32_t site;           /* unique id of the dropout call site               */
  const uint32_t* step;    /* device pointer to the step counter (may be NULL) */
} ovqa_dropout;

int ovqa_abi_version(void);
const char* ovqa_last_error(void);
/* Which kernel family the last entry point called on this thread ran: "mfma" (gfx950 matrix-core kernels), "simple"
 * (VALU reference-grade kernels: the fp32 mode, and bf16 shapes the MFMA kernels do not tile) or "" (kernels with a
 * single form).  With OVQA_REQUIRE_MFMA=1 in the environment a bf16 call that would fall back returns
 * OVQA_ERR_UNSUPPORTED instead, so a test can assert which kernel it validated. */
const char* ovqa_last_dispatch(void);
/* Launch timing (diagnostic, used by bench.py's roofline figure).  Between ovqa_launch_timing_begin(max) and
 * ovqa_launch_timing_end(), every launch of the bf16 GEMM kernels behind ovqa_linear_fwd / ovqa_linear_fwd_res32 /
 * ovqa_linear_bwd_data / ovqa_linear_bwd_data_wt and (ABI 6) of the MFMA attention kernels behind ovqa_attention_fwd / _bwd /
 * _qkv_fwd / _q_fwd / _bwd_do made by this PROCESS (any thread: autograd runs backward on its
 * own) carries a start / stop event pair
 * (hipExtLaunchKernel): the dispatch packet's own begin / end timestamps, i.e. the kernel's execution time as
 * rocprofv3 --kernel-trace reports it.  ovqa_launch_timing_count() = launches recorded so far (a caller maps its
 * calls to records with it); ovqa_launch_timing_end(us, cap) waits for the recorded kernels, writes their durations
 * in microseconds in launch order, returns their number (or a negative status) and disarms.  Not for use under
 * stream capture; launches beyond `max` are not timed. */
int ovqa_launch_timing_begin(int max_launches);
int ovqa_launch_timing_count(void);
int ovqa_launch_timing_end(float* us, int cap);
/* Scratch bytes the caller must provide to the entry points that take `ws`. */
int64_t ovqa_workspace_bytes(void);
/* Streams for running independent parts of a step side by side (ABI 6): the latency-bound chain of the question stack
 * next to the throughput-bound grouped weight gradients (encoders.py:101-117 vs 137-164: the two stacks only meet in the
 * guided attention).  `cu_mask` / `n_words` (NULL / 0 = every CU): the compute units the stream's kernels may run on, bit
 * i of word i / 32 = CU i (hipExtStreamCreateWithCUMask); otherwise `priority` within ovqa_stream_priority_range
 * (numerically lower = served first, hipStreamCreateWithPriority).  A masked stream has the default priority. */
int ovqa_stream_priority_range(int* least, int* greatest);
int ovqa_stream_create(void** stream, int priority, const uint32_t* cu_mask, int32_t n_words);
int ovqa_stream_destroy(void* stream);

/* ---------------------------------------------------------------------------
 * nn.Linear forward (+ fused epilogue).
 *   replaces: fc_q/fc_k/fc_v/fc_o  models/modules/attentions.py:49-51,58
 *             fc1 + F.gelu + dropout_1, fc2 + dropout_2 + residual add
 *                                   models/modules/positionwise_feed_forward.py:24-26
 *             dropout + residual add of MultiHeadAttention  attentions.py:330-331
 *             OcrPtrNet.query/key   models/mmf_m4c.py:376-389
 *   x [M,K] (ldx), w [N,K] row-major ([out,in], the nn.Linear layout) of
 *   `dtype`, bias fp32 [N] (may be NULL), y [M,N] (ldy).  `preact` [M,N]
 *   (ld = N) receives u for BIAS_GELU (may be NULL when no backward is
 *   needed).  `residual` [M,N] (ldres) for BIAS_RESIDUAL.
 * ------------------------------------------------------------------------- */
int ovqa_linear_fwd(int dtype, int epilogue,
                    const void* x, int64_t ldx, const void* w, const float* bias,
                    const void* residual, int64_t ldres,
                    void* y, int64_t ldy, void* preact,
                    int64_t M, int64_t N, int64_t K,
                    const ovqa_dropout* drop, void* stream);

/* fp32 residual stream of the bf16 mode (the post-LN residual chain of attentions.py:330-331 and
 * positionwise_feed_forward.py:25-26 kept in fp32 between blocks, so that a 6-layer stack stays within 1e-2 of the
 * fp32 reference):
 *     pre[m,n] = res(m,n) + drop(x W^T + b)[m,n]          x, w bf16; bias, residual, pre fp32
 * `ln` == NULL: res = residual[m,n].  Otherwise `residual` is the PREVIOUS block's fp32 pre-LayerNorm sum and
 *     res(m,n) = (residual[m,n] - ln->mean[m]) * ln->rstd[m] * ln->gamma[n] + ln->beta[n]
 * i.e. the previous block's LayerNorm output is recomputed in fp32 on the fly instead of being stored. */
typedef struct ovqa_ln_ref {
  const float* mean;   /* [M] */
  const float* rstd;   /* [M] */
  const float* gamma;  /* [N] */
  const float* beta;   /* [N] */
} ovqa_ln_ref;
int ovqa_linear_fwd_res32(const void* x, int64_t ldx, const void* w, const float* bias,
                          const float* residual, int64_t ldres, const ovqa_ln_ref* ln,
                          float* pre, int64_t ldpre, int64_t M, int64_t N, int64_t K,
                          const ovqa_dropout* drop, void* stream);

/* dX = dY W  (autograd of nn.Linear w.r.t. its input).
 *   dy [M,N] (lddy), w [N,K], dx [M,K] (lddx).
 *   If `gelu_preact` != NULL the FFN backward is fused:
 *       dx = (dY W) * dropmask/(1-p) * gelu'(gelu_preact)      (fc2 -> fc1 seam,
 *       positionwise_feed_forward.py:24-25), with `drop` describing dropout_1.
 *   If `addend` [M,K] (ldadd) != NULL it is added (the residual-branch gradient;
 *   may alias dx). */
int ovqa_linear_bwd_data(int dtype, const void* dy, int64_t lddy, const void* w,
                         void* dx, int64_t lddx, const void* gelu_preact,
                         const void* addend, int64_t ldadd,
                         int64_t M, int64_t N, int64_t K,
                         const ovqa_dropout* drop, void* stream);

/* dX from a TRANSPOSED bf16 copy of the weight: wt[K, N] with row stride ldwt (dx[m,i] = sum_n dy[m,n] wt[i,n]).
 * Same epilogue arguments as ovqa_linear_bwd_data.  The [N, K] form has to stage its weight tile k-major; in the
 * MCAN step that costs +0.44 ms, so the training harness keeps a transposed shadow of every matrix
 * (ovqa_grouped_transpose after each optimiser step: one launch, 2 x 88 MB of traffic).
 * bf16 only; requires N % 8 == 0, K % 8 == 0, lddy % 8 == 0, ldwt % 8 == 0, lddx % 4 == 0. */
int ovqa_linear_bwd_data_wt(int dtype, const void* dy, int64_t lddy, const void* wt, int64_t ldwt,
                            void* dx, int64_t lddx, const void* gelu_preact,
                            const void* addend, int64_t ldadd,
                            int64_t M, int64_t N, int64_t K, const ovqa_dropout* drop, void* stream);

/* dst[c, r] = src[r, c] for many bf16 matrices in one launch (`problems` is a DEVICE array; rows/cols need not
 * be multiples of the 64x64 tile, ld_src / ld_dst / pointers must allow 16-byte accesses: multiples of 8). */
typedef struct ovqa_transpose_problem {
  const void* src;
  void* dst;
  int64_t ld_src;
  int64_t ld_dst;
  int32_t rows;
  int32_t cols;
} ovqa_transpose_problem;
int ovqa_grouped_transpose(const ovqa_transpose_problem* problems, int32_t n_problems, int32_t max_tiles,
                           void* stream);

/* dW = dY^T X (fp32 [N,K]), db = column sums of dY (fp32 [N], may be NULL).
 *   accumulate: bit 0 -> dw += (else overwrite), bit 1 -> db += (else overwrite).
 *   ws: scratch of ovqa_workspace_bytes(). */
int ovqa_linear_bwd_weight(int dtype, const void* dy, int64_t lddy,
                           const void* x, int64_t ldx, float* dw, float* db,
                           int64_t M, int64_t N, int64_t K, int accumulate,
                           void* ws, void* stream);

/* Grouped form: every weight gradient of a backward pass in ONE launch (the
 * individual products have 16..64 output tiles each -- far fewer than 256 CUs --
 * so they are deferred and tiled together instead of being split along M).
 * `problems`/`tiles` are DEVICE arrays written by the host side: tiles[i] =
 * {problem index, tile over N, tile over K, 0} in units of 128; an entry with problem index -1 is
 * padding (the workgroup returns). bf16 only.
 * form (the parameter named all_m_mult64 up to ABI 6):
 *   0  register-staged 128 x 128 tiles, any shapes the single-product entry point accepts;
 *   1  direct-to-LDS 128 x 128 tiles (8 waves): the caller promises that every problem's M is a multiple of 64, its
 *      pointers 16-byte aligned, lddy / ldx multiples of 8;
 *   (ABI 7-9 had a form 2, 256 x 256 tiles on one 16-wave workgroup per CU: measured slower than form 1 in the MCAN step,
 *   3.231 against 3.195 ms, and removed in ABI 10.) */
typedef struct {
  const void* dy;   /* [M,N], row stride lddy */
  const void* x;    /* [M,K], row stride ldx  */
  float* dw;        /* [N,K] fp32             */
  float* db;        /* [N] fp32 bias gradient (column sums of dy) or NULL */
  int64_t lddy, ldx;
  int32_t M, N, K;
  int32_t accumulate; /* bit 0: dw +=, bit 1: db += */
} ovqa_wgrad_problem;
int ovqa_grouped_linear_bwd_weight(int dtype, const ovqa_wgrad_problem* problems_dev,
                                   const int32_t* tiles_dev, int64_t n_tiles, int32_t form,
                                   void* stream);
/* The same launch with the OPTIMISER STEP of a weight inside the epilogue of its last gradient tile (round 5; form 1 only):
 * for problem i with targets[i].param != NULL the fp32 tile of dW is NOT stored -- the workgroup that finished it applies
 * Adam to the 128 x 128 tile of the master weights and both moments (same arithmetic as ovqa_adam_step_tiled: the two
 * kernels share one update function and give the same bits) and writes the tile of the bf16 shadow and of its transposed
 * copy.  What a training step saves: the gradient's round trip through HBM (8 B per weight) and a second pass over the
 * optimiser state; valid when this product is the ONLY contribution to the weight's gradient in the step and no exchange
 * between ranks stands between gradient and update (world size 1).  Such a problem needs accumulate bit 0 clear and
 * N % 128 == 0, K % 128 == 0; the bias gradient (db) is written as before.  Problems with a NULL target behave like
 * ovqa_grouped_linear_bwd_weight.
 *   replaces: loss.backward() of the nn.Linear weights + optim.step() for them (tasks/base_task.py:46,
 *             classification_task.py:131-133), fused.
 *   targets_dev: DEVICE array, one entry per problem; transposed[(k) * ld_transposed + n] receives element (n, k). */
typedef struct {
  float* param;        /* fp32 master [N,K] (row stride K), or NULL: plain gradient store */
  float* exp_avg;      /* fp32 [N,K] */
  float* exp_avg_sq;   /* fp32 [N,K] */
  void* shadow;        /* bf16 [N,K] */
  void* transposed;    /* bf16 [K, ld_transposed]: column n of it is row n of the weight */
  int64_t ld_transposed;
} ovqa_adam_target;
typedef struct {
  float lr, beta1, beta2, eps, weight_decay, grad_scale;
  const float* lr_scale_ptr;   /* device scalar multiplied into lr (ovqa_begin_step's lr_out), or NULL */
  const uint32_t* step_ptr;    /* device step counter t (bias correction 1 - beta^t), already incremented for this step */
} ovqa_adam_consts;
int ovqa_grouped_linear_bwd_weight_adam(int dtype, const ovqa_wgrad_problem* problems_dev, const int32_t* tiles_dev,
                                        int64_t n_tiles, const ovqa_adam_target* targets_dev,
                                        const ovqa_adam_consts* consts, void* stream);
/* db (fp32 [N]) (+)= column sums of dy [M,N] (bias gradient on its own). */
int ovqa_bias_grad(int dtype, const void* dy, int64_t lddy, float* db, int64_t M, int64_t N,
                   int accumulate, void* stream);

/* ---------------------------------------------------------------------------
 * LayerNorm over the last dim (+ optional positional table add).
 *   replaces: nn.LayerNorm in MultiHeadAttention / PositionWiseFeedForward
 *             (attentions.py:312,331; positionwise_feed_forward.py:21,26) and
 *             the encoder prologue LN(x) + SinusoidPositionalEmbedding(x)
 *             (encoders.py:113,154,192-193,243-244; pos_embeddings.py:58-72).
 *   x [M,D] of `in_dtype`, y [M,D] of `dtype`; gamma/beta fp32 [D];
 *   y_f32 [M,D] fp32 or NULL: the same result unrounded (the fp32 residual stream of the bf16 mode: the
 *   encoder prologue hands the first block both the bf16 GEMM operand and the fp32 residual);
 *   mean/rstd fp32 [M] (saved for backward, may be NULL);
 *   pos fp32 [pos_rows, D] or NULL: y[m] += pos[m % pos_rows].
 *   Supported (in_dtype -> dtype): f32->f32, bf16->bf16, f32->bf16.
 * ------------------------------------------------------------------------- */
int ovqa_layernorm_fwd(int dtype, int in_dtype, const void* x, const float* gamma,
                       const float* beta, const float* pos, int64_t pos_rows,
                       void* y, float* y_f32, float* mean, float* rstd,
                       int64_t M, int64_t D, float eps, void* stream);

/* dx = LN backward; dgamma/dbeta fp32 [D] (accumulate flag as above).
 *   If drop != NULL and drop->p > 0, `dx_dropped` [M,D] additionally receives
 *   dx * keep/(1-p): the gradient flowing into the branch that went through
 *   dropout before the residual add (attentions.py:330, pwff.py:25).
 *   dx_dtype is the dtype of dx (fp32 for the prologue whose input was fp32).
 *   Supported (dy, x, dx): (f32,f32,f32), (bf16,bf16,bf16), (bf16,f32,bf16) [fp32 pre-LN sum], (bf16,f32,f32). */
int ovqa_layernorm_bwd(int dtype, int dx_dtype, const void* dy, const void* x, int x_dtype,
                       const float* gamma, const float* mean, const float* rstd,
                       void* dx, void* dx_dropped, float* dgamma, float* dbeta,
                       int64_t M, int64_t D, int accumulate,
                       const ovqa_dropout* drop, void* ws, void* stream);

/* Deferred dgamma/dbeta: ovqa_layernorm_bwd called with dgamma == dbeta == NULL computes dx only and leaves
 * its row-slab partials in `ws` as fp32 [blocks][2][D] (dgamma partials, then dbeta partials), where
 * blocks = ovqa_layernorm_bwd_blocks(M, D).  ovqa_grouped_partial_reduce then sums the partials of MANY LayerNorms
 * in one launch (each ~5 us dependent launch saved matters: the MCAN step has 32 of them):
 *   out0[D] (+)= sum_b partial[b][0][:],  out1[D] (+)= sum_b partial[b][1][:].  ABI 6: DETERMINISTIC -- a fixed summation
 *   order and plain stores (ABI <= 5 used fp32 atomics across row groups and needed pre-zeroed outputs); `accumulate`
 *   selects = or +=.  No two problems of one call may share an output.  `problems` is a DEVICE array. */
typedef struct ovqa_reduce_problem {
  const float* partial;
  float* out0;
  float* out1;
  int32_t blocks;
  int32_t D;
  int32_t accumulate; /* 0: out = sum, 1: out += sum */
  int32_t reserved_;
} ovqa_reduce_problem;
int ovqa_layernorm_bwd_blocks(int64_t M, int64_t D);
int ovqa_grouped_partial_reduce(const ovqa_reduce_problem* problems, int32_t n_problems, int32_t max_blocks,
                                int32_t max_D, void* stream);

/* ---------------------------------------------------------------------------
 * Attention core: softmax(q k^T * scale + mask) v, all heads of all samples.
 *   replaces: ScaledDotProductAttention.forward  attentions.py:49-57 (the
 *             permutes are folded into addressing: q/k/v/o are indexed as
 *             [b, n, h*d + c] with row strides ld*, so the kernels read the
 *             projection outputs in place -- also from a packed QKV buffer).
 *   mask: fp32 additive, element (b,h,i,j) at mask[b*msb + h*msh + i*msq + j]
 *         (strides 0 broadcast; NULL = no mask).  Mask value semantics follow
 *         models/utils.py:44-73 (-1e5, not -inf: fully masked rows give a
 *         uniform distribution).
 *   lse fp32 [B,H,nq] (may be NULL); att [B,H,nq,nk] probabilities of `dtype`
 *   (NULL unless the caller wants the second return value of the reference).
 *   att_drop (may be NULL / p == 0): dropout on the attention PROBABILITIES, p~ = p * keep/(1-p) with the
 *   element index ((b*H+h)*nq+i)*nk+j -- what transformers' BertSelfAttention does inside M4C's multimodal
 *   transformer (models/mmf_m4c.py:349-351); `att` then holds p~.  The reference's own attention modules have
 *   none (attentions.py:56-57).  Runs on the VALU kernels (the MFMA kernels cover p == 0).
 * ------------------------------------------------------------------------- */
int ovqa_attention_fwd(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk,
                       const void* v, int64_t ldv, const float* mask,
                       int64_t msb, int64_t msh, int64_t msq,
                       void* o, int64_t ldo, float* lse, void* att, void* o_lo,
                       int64_t B, int64_t H, int64_t nq, int64_t nk, int64_t dk, int64_t dv,
                       float scale, const ovqa_dropout* att_drop, void* stream);
/* The same forward for a PREFIX-LM mask given by its structure instead of a dense (B, 1, n, n) tensor (round 5): every query
 * sees the keys that the key-mask row admits, and among the LAST `causal_tail` positions of the sequence query i does not
 * see keys j > i.
 *   replaces: the mask MMT.forward builds for M4C's multimodal transformer -- the padding masks of [txt; obj; ocr] and a
 *             zero row for the decoding steps, repeated over the queries, with the causal corner of the decoding steps
 *             written in (mmf_m4c.py:310-340) -- as the attention core of its BertSelfAttention sees it at inference;
 *             the kernel reads one mask row per (b, h) out of LDS instead of n x n mask values per (b, h) from HBM
 *             (configs[3]: 126 -> see DESIGN.md us per launch).
 *   key_mask: fp32 additive row, element (b, h, j) at key_mask[b*msb + h*msh + j], or NULL; self-attention (n queries =
 *   n keys), bf16 only, inference only (no log-sum-exp consumers beyond `lse`, no dropout, no probabilities).
 *   Shapes outside the MFMA forward kernel: OVQA_ERR_UNSUPPORTED (callers fall back to the dense mask). */
int ovqa_attention_fwd_prefix_lm(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                                 int64_t ldv, const float* key_mask, int64_t msb, int64_t msh, int64_t causal_tail,
                                 void* o, int64_t ldo, float* lse, int64_t B, int64_t H, int64_t n, int64_t d,
                                 float scale, void* stream);

/* Self-attention forward with the Q/K/V projections inside: MultiHeadAttention.forward with queries is keys is
 * values (attentions.py:316-326 -> :49-57; the MCAN SA blocks, encoders.py:46-49, and the M4C MMT layers):
 *     qkv[B*n, 3*H*d] = x[B*n, d_model] @ w^T + bias     (w = fc_q | fc_k | fc_v weight rows, packed [3*H*d, d_model];
 *                                                         qkv is kept: the backward pass reads it)
 *     o, lse          = attention(q = qkv[:, 0:H*d], k = qkv[:, H*d:2*H*d], v = qkv[:, 2*H*d:], mask)
 * with nq = nk = n, dk = dv = d and a key mask (msq = 0) or none.  For bf16, d = 64, n <= 128 and d_model a multiple
 * of 32 this is ONE kernel -- a workgroup projects the rows of a few samples for one head with MFMA, keeps the projected
 * tiles in LDS and runs the attention on them; the projections are written to HBM once and never re-read in forward.
 * Any other shape / dtype runs the two separate entry points (ovqa_linear_fwd + ovqa_attention_fwd): same results
 * either way up to bf16 rounding of identical fp32 sums in a different order.  ovqa_last_dispatch() = "mfma-fused".
 * ------------------------------------------------------------------------- */
int ovqa_attention_qkv_fwd(int dtype, const void* x, int64_t ldx, const void* w, const float* bias,
                           void* qkv, int64_t ldqkv, const float* mask, int64_t msb, int64_t msh,
                           void* o, int64_t ldo, float* lse, void* o_lo,
                           int64_t B, int64_t H, int64_t n, int64_t d_model, int64_t d, float scale, void* stream);

/* Single-query attention of an autoregressive decoding step (replaces attentions.py:314-327 called with ONE new query
 * per sequence: decoders.py:46-63 under beam_search.py:41-62).  q [R, H*d] (row stride ldq), one query per row;
 * k / v: in-place caches, query row r reads cache row r / group (the `group` beams of a sample share the projected
 * encoder K / V; group = 1 for the self-attention caches), key j of that row at `kv_batch_stride * (r / group) +
 * j * ldk` (elements), `n` live keys (1 <= n <= 512); mask: additive fp32 [R, ldmask] or NULL; o [R, H*d].
 * d in {32, 64, 128}; pointers 16-byte aligned, ldq / ldk multiples of 16 bytes.  One wave per (row, head), K and V
 * streamed once: HBM-bound.  ovqa_last_dispatch() = "decode". */
int ovqa_attention_decode(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                          int64_t kv_batch_stride, int64_t group, const float* mask, int64_t ldmask,
                          void* o, int64_t ldo, int64_t R, int64_t H, int64_t n, int64_t d, float scale, void* stream);

/* The k best entries of every row of x [R, V] (fp32, row stride ldx), best first, ties by the smaller index:
 * vals fp32 [R, k], idx int64 [R, k]; 1 <= k <= 8 and k <= V.  Candidate selection of a beam-search step
 * (beam_search.py:36-39 takes them from a full sort of the cur_beam * |V| candidates). */
int ovqa_topk_rows(const float* x, int64_t ldx, int64_t R, int64_t V, int64_t k, float* vals, int64_t* idx, void* stream);

/* Cross / guided attention forward with the QUERY projection inside (ABI 5): q = x W_q^T + b_q (written: backward reads
 * it), o, lse = attention(q, k, v, key mask) for keys and values that are ALREADY projected -- the hoisted K / V
 * projection of GuidedAttentionEncoder (encoders.py:83-96 applies fc_k / fc_v of every layer to the same language
 * features) or a packed K | V projection; MultiHeadAttention.forward with queries != keys (attentions.py:316-326 ->
 * :49-57).  x [B * nq, d_model], w = fc_q weight [H * d, d_model]; k / v [B * nk, >= H * d] with their row strides; mask:
 * fp32 additive key mask, element (b, h, j) at mask[b * msb + h * msh + j], or NULL.  bf16, d = 64, nq <= 128, nk <= 128,
 * d_model a multiple of 64: one kernel; otherwise ovqa_linear_fwd + ovqa_attention_fwd inside. */
int ovqa_attention_q_fwd(int dtype, const void* x, int64_t ldx, const void* w, const float* bias, void* q, int64_t ldq,
                         const void* k, int64_t ldk, const void* v, int64_t ldv, const float* mask, int64_t msb, int64_t msh,
                         void* o, int64_t ldo, float* lse, void* o_lo, int64_t B, int64_t H, int64_t nq, int64_t nk,
                         int64_t d_model, int64_t d, float scale, void* stream);

/* Attention backward with the dO projection inside (ABI 5): the gradient reaches the attention core through fc_o,
 * dO = dY W_o (autograd of attentions.py:58), and the fc_o dX product is folded into the backward kernel -- per head,
 * dO_h = dY wt[h*d:(h+1)*d, :]^T from the TRANSPOSED weight copy wt [H*d, d_model] (what ovqa_adam_step_tiled maintains) --
 * so dO never travels through HBM.  dy [B*nq, d_model]; everything else as ovqa_attention_bwd with a key mask (msq = 0),
 * no d_att / d_lse / dropout.  bf16, d = 64, nk <= 32 and either 64 < nq <= 128 (guided attention: 100 queries x 20 keys) or
 * nq <= 32 with an even head count (the 20 x 20 question self-attention: q, k, v slices of the packed projection): one kernel; otherwise, if `d_o_scratch` [B*nq, lddo] is given, ovqa_linear_bwd_data_wt into it + ovqa_attention_bwd;
 * otherwise OVQA_ERR_UNSUPPORTED. */
int ovqa_attention_bwd_do(int dtype, const void* dy, int64_t lddy, const void* wt, int64_t ldwt, void* d_o_scratch,
                          int64_t lddo, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                          const void* o, int64_t ldo, const void* o_lo, const float* lse, const float* mask, int64_t msb,
                          int64_t msh, void* dq, int64_t lddq, void* dk_, int64_t lddk, void* dv_, int64_t lddv, float* delta,
                          int64_t B, int64_t H, int64_t nq, int64_t nk, int64_t d_model, int64_t d, float scale,
                          void* stream);

/* Three products of one input in one launch (ABI 5): y_i[m, :] = x[m, :] W_i^T + b_i, i = 0..2, for three weight matrices
 * stacked as w [3 F, K] (bias [3 F] or NULL), every output with its own base pointer and row stride.  A decoding step's
 * fc_q / fc_k / fc_v (attentions.py:49-51 on the one new position): q into a buffer, k and v straight into their slots of
 * the in-place caches.  bf16: MFMA kernels (F % 8 == 0, 16-byte aligned pointers and rows); otherwise three VALU products. */
int ovqa_linear_fwd_split3(int dtype, const void* x, int64_t ldx, const void* w, const float* bias, void* y0, int64_t ld0,
                           void* y1, int64_t ld1, void* y2, int64_t ld2, int64_t M, int64_t F, int64_t K, void* stream);

/* ---- the index / elementwise work around one autoregressive decoding step (ABI 5) -------------------------------------
 * Three launches instead of the ~55 stock elementwise launches a decoding step with beam search spends on it.
 *
 * ovqa_decode_embed: the stateful branch of Decoder.forward before the layers (models/modules/decoders.py:46-66) for ONE
 * new position per row: seq[r] += 1 (running_seq.add_(1)); x[r] = emb[tokens[r]] + pos[seq[r]] (word_emb + pos_emb), written
 * as fp32 (x32, may be NULL) and / or in out_dtype (x, may be NULL); mask[r * ld_mask + col] = tokens[r] == pad_idx ?
 * mask_value : 0 -- the new column of running_mask_self_attention (additive, models/utils.py:44-73), mask may be NULL.
 * emb fp32 [vocab, ld_emb], pos fp32 [n_pos, ld_pos] (indices clamped), D % 4 == 0, tables 16-byte aligned. */
int ovqa_decode_embed(int out_dtype, const int64_t* tokens, const float* emb, int64_t ld_emb, int64_t vocab,
                      const float* pos, int64_t ld_pos, int64_t n_pos, int64_t* seq, int64_t pad_idx, float mask_value,
                      float* mask, int64_t ld_mask, int64_t col, float* x32, void* x, int64_t R, int64_t D, void* stream);

/* ovqa_beam_candidates: for every row r (= sample * cur_beam + beam) of the step's logits [R, V] (dtype; row stride ld):
 * word_logprob = log_softmax(logits[r]) in fp32 (base_transformer.py:31-44 -> decoders.py:76), the candidate scores of
 * models/modules/beam_search.py:41-57 -- seq_logprob[r] + word_logprob for a live sequence; a sequence whose previous word
 * was <eos> (prev_words[r] == eos: seq_mask[r] is set to 0 IN PLACE, beam_search.py:49-51) keeps seq_logprob[r] on word 0
 * and gets -999 elsewhere -- and the k best of them, best first, ties by the smaller word: vals fp32 [R, k], idx int64
 * [R, k], wl fp32 [R, k] = word_logprob[idx] * seq_mask[r] (what beam_search.py:66 gathers into the log-prob history).
 * prev_words == NULL at the first step (nothing is finished).  1 <= k <= 8. */
int ovqa_beam_candidates(int dtype, const void* logits, int64_t ld, int64_t R, int64_t V, int64_t k,
                         const float* seq_logprob, float* seq_mask, const int64_t* prev_words, int64_t eos, float* vals,
                         int64_t* idx, float* wl, void* stream);

/* ovqa_beam_commit: per sample, the `beam` best of its cur * k survivors (beam_search.py:36-39: the first `beam` of the
 * sorted candidates; ties by the smaller flat index) and the bookkeeping of beam_search.py:58-83: seq_logprob_out /
 * seq_mask_out [b_s, beam], selected_beam int32 [b_s, beam] (the source beam: the index ovqa_grouped_row_gather takes),
 * words int64 [b_s, beam] (the next step's tokens), and the histories: out_out / lp_out [b_s, beam, T] = columns < t of
 * the source beams' rows of out_in / lp_in [b_s, cur, T], column t = the chosen word / its wl.  The histories are
 * double-buffered (in != out).  beam <= 8, cur * k <= 64, t < T. */
int ovqa_beam_commit(const float* vals, const int64_t* idx, const float* wl, const float* seq_mask_in,
                     const int64_t* out_in, const float* lp_in, int64_t* out_out, float* lp_out, float* seq_logprob_out,
                     float* seq_mask_out, int32_t* selected_beam, int64_t* words, int64_t b_s, int64_t cur, int64_t k,
                     int64_t beam, int64_t t, int64_t T, void* stream);

/* o_lo (OVQA_BF16 only; may be NULL in all three calls): the rounding residual of the attention output, bf16, same
 * layout as o: o_lo = bf16(o_exact - float(o)).  The forward calls write it, ovqa_attention_bwd reads it for
 * delta_i = dO_i . (o_i + o_lo_i): dS = P (dP - delta) is a cancellation, and with near-uniform attention (a freshly
 * initialised stack) the bf16 rounding of o alone put errors of 5-8x their own size on the fc_q / fc_k gradients of the
 * last layers (which are ~3000x smaller than the FFN gradients there); with the residual delta is exact to 2^-17.
 *
 * Gradients of the attention core.  `delta` fp32 [B,H,nq] is scratch owned by
 * the caller (rowsum(P*dP)); dq/dk/dv use the same [b,n,h*d+c] addressing.
 * d_att [B,H,nq,nk] (dtype, may be NULL) is the gradient w.r.t. the returned
 * attention weights -- the reference's second return value is differentiable
 * (attentions.py:56,60).  d_lse fp32 [B,H,nq] (may be NULL) is the gradient w.r.t. the returned log-sum-exp
 * (d lse_i / d S_ij = P_ij): callers that extend the softmax by extra columns of their own -- the per-query
 * language-signal key of AdaptiveScaledDotProductAttention, attentions.py:262-277 -- merge through lse.
 * att_drop: the forward call's dropout on the probabilities. */
int ovqa_attention_bwd(int dtype, const void* d_o, int64_t lddo,
                       const void* q, int64_t ldq, const void* k, int64_t ldk,
                       const void* v, int64_t ldv, const void* o, int64_t ldo, const void* o_lo,
                       const void* d_att, const float* lse, const float* mask,
                       int64_t msb, int64_t msh, int64_t msq,
                       void* dq, int64_t lddq, void* dk_, int64_t lddk, void* dv_, int64_t lddv,
                       float* delta, const float* d_lse,
                       int64_t B, int64_t H, int64_t nq, int64_t nk, int64_t dk, int64_t dv,
                       float scale, const ovqa_dropout* att_drop, void* stream);

/* ---------------------------------------------------------------------------
 * Pointer scorer: scores[b,t,n] = q[b,t,:].k[b,n,:] * scale (+ mask | -inf fill).
 *   replaces: OcrPtrNet.forward  models/mmf_m4c.py:391-394 (additive mask) and
 *             DynamicPointerNetwork.forward  models/m4c.py:30-31 (key-axis
 *             -inf fill), models/iterative_m4c.py:29-30 (query-axis fill).
 *   add_mask fp32 [B,n] or NULL; key_fill / query_fill uint8 [B,n] / [B,t] or
 *   NULL (non-zero => -inf).  scores fp32 [B,t,n].  Backward = two calls of
 *   ovqa_batched_gemm.
 * ------------------------------------------------------------------------- */
int ovqa_pointer_score(int dtype, const void* q, const void* k, const float* add_mask,
                       const uint8_t* key_fill, const uint8_t* query_fill, float* scores,
                       int64_t B, int64_t T, int64_t Nk, int64_t D, float scale, void* stream);

/* C[b] = alpha * op(A[b]) op(B[b])  (batched, strided; C of c_dtype). */
int ovqa_batched_gemm(int dtype, int c_dtype, int trans_a, int trans_b,
                      const void* A, int64_t lda, int64_t stride_a,
                      const void* Bm, int64_t ldb, int64_t stride_b,
                      void* C, int64_t ldc, int64_t stride_c,
                      int64_t batch, int64_t M, int64_t N, int64_t K, float alpha, void* stream);

/* ---------------------------------------------------------------------------
 * Optimiser step on a flat parameter arena (row T of SURVEY 8a):
 *   Adam(betas) exactly as torch.optim.Adam (tasks/base_task.py:46) on fp32
 *   master weights; also refreshes the bf16 shadow copy used by the kernels.
 *   lr_scale_ptr / step_ptr are device scalars so a captured graph can replay
 *   with a schedule (tasks/base_task.py:73-76).  grad_scale multiplies g first
 *   (1/world_size after a sum all-reduce).  grad_dtype: OVQA_F32 (the arena's gradient buffer) or OVQA_BF16
 *   (the data-parallel staging buffer the all-reduce ran on -- saves the cast back).
 * ------------------------------------------------------------------------- */
int ovqa_adam_step(float* param, const void* grad, int grad_dtype, float* exp_avg, float* exp_avg_sq,
                   void* shadow_bf16, int64_t n, float lr, const float* lr_scale_ptr,
                   float beta1, float beta2, float eps, float weight_decay,
                   float grad_scale, const uint32_t* step_ptr, void* stream);

/* The same Adam step with the transposed bf16 weight copy written in the same pass.  `tiles` (DEVICE array, n_tiles
 * entries) lists the 64 x 64 tiles of every weight matrix (or adjacency group of matrices) of the arena; one workgroup
 * updates param / exp_avg / exp_avg_sq of a tile, writes the bf16 shadow (row-major, same offsets as param) AND the
 * tile's transpose into `shadow_t` ([cols, rows] at the matrix' offset: what ovqa_linear_bwd_data_wt reads) -- the
 * separate ovqa_grouped_transpose pass over the shadow (2 x 2 B per parameter of HBM traffic and a launch) goes away.
 * Elements of [flat_lo, flat_hi) (the 1-D parameters behind the matrices) are updated like ovqa_adam_step does.
 * rows, cols multiples of 8; offsets, flat_lo, flat_hi multiples of 4. */
typedef struct ovqa_adam_tile {
  int64_t off;            /* element offset of the matrix in the arena */
  int32_t rows, cols;     /* of the matrix (row-major [rows, cols]) */
  int32_t r0, c0;         /* origin of this tile */
  int32_t reserved[2];
} ovqa_adam_tile;
int ovqa_adam_step_tiled(float* param, const void* grad, int grad_dtype, float* exp_avg, float* exp_avg_sq,
                         void* shadow_bf16, void* shadow_t_bf16, const ovqa_adam_tile* tiles, int32_t n_tiles,
                         int64_t flat_lo, int64_t flat_hi, float lr, const float* lr_scale_ptr,
                         float beta1, float beta2, float eps, float weight_decay,
                         float grad_scale, const uint32_t* step_ptr, void* stream);

/* step counter++ (device side, inside the graph) */
int ovqa_increment_step(uint32_t* step_ptr, void* stream);
/* two counters in one launch (the optimiser's step and the dropout step of a training loop); `b` may be NULL */
int ovqa_increment_steps(uint32_t* a, uint32_t* b, void* stream);

/* An optimiser step begins (ABI 6): *lr_out = lr_table[*step_ptr % n_table] (the LambdaLR schedule of tasks/base_task.py:
 * 73-76 as a DEVICE table indexed by the number of steps taken; both may be NULL), then *step_ptr += 1 and, if given,
 * *second += 1.  With ovqa_adam_step*'s lr_scale_ptr = lr_out the whole step replays from one graph: the order of
 * classification_task.py:129-139 (backward -> optim.step -> scheduler.step) with no host value in it. */
int ovqa_begin_step(uint32_t* step_ptr, uint32_t* second, const float* lr_table, int32_t n_table, float* lr_out,
                    void* stream);

/* fp32 -> bf16 / bf16 -> fp32 flat casts (shadow refresh after load_state_dict). */
int ovqa_cast(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, void* stream);

/* ---------------------------------------------------------------------------
 * Embedding step in front of the path ("next" row 2):
 *   FeatureEmbedding = dropout(gelu(proj(x))) + zero-row padding mask   vision_embeddings.py:10-25
 *   forward is ovqa_linear_fwd(OVQA_EPI_BIAS_GELU, preact = u); its backward needs
 *   du[i] = dy[i] * keep(i)/(1-p) * gelu'(u[i])  (flat index i, same dropout site as the forward), after
 *   which dW/db/dx are ovqa_linear_bwd_weight / ovqa_linear_bwd_data on du.
 *   ovqa_row_padding_mask: mask[m] = (sum_d x[m,d] == pad_value*D) ? -1e5 : 0, fp32 [M]  (models/utils.py:44-58,
 *   generate_padding_mask on feature tensors) in one pass over x.
 * ------------------------------------------------------------------------- */
int ovqa_gelu_bwd(int dtype, const void* dy, const void* u, void* du, int64_t n,
                  const ovqa_dropout* drop, void* stream);
int ovqa_row_padding_mask(int dtype, const void* x, float* mask, int64_t M, int64_t D, float pad_value,
                          void* stream);

/* ---------------------------------------------------------------------------
 * Beam-search state reorder for ALL state buffers of a decoder in one launch ("next" row 1):
 *   replaces: BeamSearch._expand_state applied through Module.apply_to_states
 *             models/modules/beam_search.py:19-34, models/modules/containers.py:26-31
 *             (one torch.gather with an expanded index tensor per running K/V cache / mask / position buffer).
 *   Every problem is a row-major buffer src [b_s*cur_beam, row_bytes] -> dst [b_s*beam, row_bytes] (any dtype,
 *   out of place):  dst[b*beam + j] = src[b*cur_beam + sel[b*beam + j]],  sel int32 [b_s*beam] in [0, cur_beam).
 *   `sel` is a DEVICE array; `problems` is a HOST array (copied into the launch's kernel arguments, 24 problems per
 *   launch): nothing is uploaded, so the call can be captured into a hipGraph as it is.
 * ------------------------------------------------------------------------- */
typedef struct ovqa_gather_problem {
  const void* src;
  void* dst;
  int64_t row_bytes;        /* bytes moved per row */
  int64_t src_stride_bytes; /* distance between consecutive source rows (0 = row_bytes): a live prefix of an in-place */
  int64_t dst_stride_bytes; /* K / V cache [rows, Lmax, D] is row_bytes = n*D*es at stride Lmax*D*es                  */
} ovqa_gather_problem;
int ovqa_grouped_row_gather(const ovqa_gather_problem* problems, int32_t n_problems, const int32_t* sel,
                            int32_t b_s, int32_t cur_beam, int32_t beam, void* stream);

/* Materialise the dropout keep-mask the fused kernels use (tests, debugging):
 * out[i] = 1 if element i of a [rows, cols] site is kept. */
int ovqa_dropout_keep_mask(const ovqa_dropout* drop, uint8_t* out, int64_t n, void* stream);

/* Mean-squared-error loss of the stack-level bench harness (fused forward+backward):
 * loss (+)= sum((x-target)^2)/n (fp32 device scalar), dx = 2*(x-target)/n.
 * target (dtype, may be NULL = 0).  NB a NULL target on LayerNorm outputs is a constant.
 * The cross-workgroup sum goes through a process-global scratch slot chosen by the STREAM handle (16 slots): launches of
 * one stream are ordered; launches on different streams are independent unless their handles hash to the same slot --
 * do not run two of these concurrently on more streams than that.  A launch that is aborted mid-way leaves its slot's
 * arrival ticket non-zero (later losses of that slot would be wrong): reload the library. */
int ovqa_sq_loss_fwd_bwd(int dtype, const void* x, const void* target, void* dx, float* loss, int64_t n,
                         int accumulate_loss, void* stream);

/* ---------------------------------------------------------------------------
 * LSTM recurrence of the text embedding (ABI 8; "next" row 2):
 *   replaces: self.lstm = nn.LSTM(D_MODEL, D_MODEL, batch_first=True); features, _ = self.lstm(features)
 *             models/modules/text_embeddings.py:236,243   (one layer, zero initial state, gate order i, f, g, o;
 *             padded positions are ordinary time steps: the reference does not pack the sequence)
 *   Forward: x [T*B, I] TIME-MAJOR rows (row t*B + b; ldx), w_ih [4H, I], w_hh [4H, H] of `dtype`, b_ih / b_hh fp32 [4H]
 *     -> y fp32 [B, T, H] (batch-major, what the module returns), y_lp (may be NULL): the same values in `dtype`, for a
 *        consumer that takes the compute dtype (the operand of the stack behind it) without a cast launch,
 *        hseq `dtype` [(T+1)*B, H] time-major: block 0 = zeros, block t+1 = h_t  (the operand of the w_hh gradient:
 *        dW_hh = dgates^T hseq[0 : T*B]), and `saved` (opaque, ovqa_lstm_saved_bytes; handed to ovqa_lstm_bwd).
 *   Backward: dy [B, T, H] of dy_dtype (fp32, or bf16 from a bf16 LayerNorm backward) -> dgates `dtype` [T*B, 4H] time-major, columns gate*H + unit (gradient w.r.t. the
 *     pre-activations).  The caller finishes with the library's GEMMs: dx = dgates w_ih, dW_ih = dgates^T x,
 *     dW_hh = dgates^T hseq[0 : T*B], db_ih = db_hh = column sums of dgates.  `w_hh_t` [H, 4H] (row stride ldwt) is the
 *     transposed copy of w_hh the bf16 persistent kernel reads (NULL in fp32 mode).
 *   `scratch` (ovqa_lstm_scratch_bytes, uninitialised, private to the call while it runs): the hand-off counters of the
 *     persistent kernels -- zeroed by the call itself with a memset node -- and the per-step kernels' temporaries.
 *   bf16 with H == I == 512, B a multiple of 16 and B <= ovqa_lstm_persistent_max_batch() (16 samples per 32 of the
 *     device's CUs: the 32 workgroups of a sample group must be co-resident): ONE persistent launch each way (weights
 *     resident in registers, h_t / dgates_t handed between workgroups in-launch: csrc/lstm.hip).  Anything else, fp32 and
 *     OVQA_FORCE_SIMPLE=1: one VALU launch per time step.  Forward and backward of one sequence take the same route
 *     (decided from dtype, B, I, H alone).  A caller with another batch size pads it to whole groups of 16 with zero
 *     rows and splits it into chunks of at most max_batch samples (the Python host side does: ops.lstm_fwd / lstm_bwd).
 *   Give-up of a hand-off wait (a workgroup of the launch never ran within ~1e5 polling sweeps): word 1000 of `scratch`
 *     (uint32) is non-zero afterwards, the same code is OR-ed into a process-lifetime device word that
 *     ovqa_lstm_status reads -- and clears -- after synchronising `stream` (ABI 10; 0 = every wait so far was served), and
 *     the outputs of that sample group are NaN from the step of the give-up on (the wave continues with the bf16-NaN
 *     sentinels it read), so a loss computed from them shows it as well.
 * ------------------------------------------------------------------------- */
int64_t ovqa_lstm_saved_bytes(int64_t B, int64_t T, int64_t H);
int64_t ovqa_lstm_scratch_bytes(int64_t B, int64_t T, int64_t H);
int64_t ovqa_lstm_persistent_max_batch(void);
int ovqa_lstm_status(uint32_t* status, void* stream);
int ovqa_lstm_fwd(int dtype, const void* x, int64_t ldx, const void* w_ih, const void* w_hh, const float* b_ih,
                  const float* b_hh, float* y, void* y_lp, void* hseq, void* saved, void* scratch,
                  int64_t B, int64_t T, int64_t I, int64_t H, void* stream);
int ovqa_lstm_bwd(int dtype, const void* dy, int dy_dtype, const void* w_hh, const void* w_hh_t, int64_t ldwt,
                  const void* saved, void* dgates, void* scratch, int64_t B, int64_t T, int64_t I, int64_t H, void* stream);

/* ---------------------------------------------------------------------------
 * The two ends of the model around the encoder stacks (ABI 8; "next" rows 2 and 4): csrc/model_ends.hip
 *
 * Token embedding rows.   replaces: self.embedding(tokens) / self.components(tokens) (nn.Embedding lookups)
 *                          models/modules/text_embeddings.py:71-80,240 and their autograd (embedding_dense_backward)
 *   ovqa_embed_gather: out[r][0..width) = table[tokens[b][t]][0..width) for r = t*B + b (time_major) or b*T + t; tokens int64
 *     [B, T]; table [vocab, >= width] of `dtype` (row stride ld_table).  `width` may include the zero padding of a table
 *     whose rows are padded to 16 bytes.  Rows 16-byte aligned.  `mask` (may be NULL): fp32 [B, T] = the additive padding
 *     mask of the token ids, (tokens == padding_idx) * -10e4 (generate_padding_mask, models/utils.py:44-58), from the
 *     same pass.
 *   ovqa_embed_scatter: dtable[v][0..width) (=|+=) sum of drows[r][0..width) over the rows r with token v, in a fixed order (the
 *     order the tokens lie in memory; no atomics), for EVERY v < rows_table: rows no token names are stored as zeros (no
 *     memset needed), row padding_idx gets zeros (nn.Embedding's padding_idx).  dtable fp32.
 * Decoder inputs.   replaces: the padding / causal mask and position arithmetic of a teacher-forced Decoder.forward,
 *                   models/modules/decoders.py:50-60 and the position add :66 (generate_padding_mask,
 *                   generate_sequential_mask, generate_self_attention_masks: models/utils.py:44-73)
 *   ovqa_decoder_inputs: tokens int64 [B, T], emb fp32 [B, T, D] (the word embeddings), pos_table fp32 [pos_rows >= T + 1, D]
 *     -> out fp32 [B, T, D] = emb + pos_table[seq], seq = 0 at a padding token, t + 1 otherwise (out may alias emb), and
 *     self_mask fp32 [B, T, T] = (tokens[b][j] == padding_idx || j > t) * -10e4 (-0.0 where unmasked, as the reference's
 *     long * float product gives).  One launch.
 * ovqa_dropout_apply: y[i] = x[i] * keep(i) / (1 - p), flat index i -- forward and backward of an nn.Dropout call site whose
 *   producer has no fused epilogue (text_embeddings.py:241).
 *
 * Attention pooling of MCAN / CrossModalityTransformer.   replaces: models/mcan.py:12-25 (MLP: fc2(dropout(relu(fc1 x)))),
 *     :70-76 (softmax over dim=1, weighted sum), cross_modality_transformer.py:51-73
 *   ovqa_pool_fwd: hpre [B*N, D] = fc1(feat) (the GEMM's output, bias inside) of `dtype`; feat [B, N, D] of `feat_dtype`;
 *     w2 fp32 [D], b2 fp32 [1] (may be NULL) = fc2; dropout call site `drop` on relu(hpre), flat index into [B*N, D]
 *     -> att fp32 [B, N] (the softmax weights, padded positions included as upstream), pooled [B, D] of `dtype`
 *        (and pooled32 fp32 [B, D] if not NULL).
 *   ovqa_pool_bwd: dpooled [B, D] of `dtype` -> dh [B*N, D] (gradient w.r.t. fc1's output), dfeat [B*N, D] (= att * dpooled, the
 *     direct gradient of the features: the addend of fc1's dX product), dw2_part fp32 [B, 2*D] (per-sample partial of fc2's
 *     weight gradient in columns 0..D, zeros after) and db2_part fp32 [B, 16] (per-sample partial of fc2's bias gradient in
 *     column 0, zeros after; may be NULL): rows for ovqa_grouped_partial_reduce (D and 8), which adds them in a fixed order.
 *
 * log_softmax + NLLLoss.   replaces: F.log_softmax(output, dim=-1) mcan.py:81; nn.NLLLoss(ignore_index)
 *                           tasks/classification_task.py:125-127 (and open_ended_task.py:155-157)
 *   ovqa_log_softmax_fwd: x [M, ld] of `dtype`, the first n columns of every row -> out fp32 [M, n].
 *   ovqa_log_softmax_bwd: g, logp fp32 [M, n] -> dx [M, ld] of `dtype`: g - exp(logp) * rowsum(g); columns n..ld-1 are
 *     written as zeros (the padded columns of a ragged classifier).
 *   ovqa_nll_loss: logp fp32 [M, n], target int64 [M] -> *loss (=|+=) -sum_{target != ignore_index} logp[r][target[r]] / count
 *     (mean reduction; may be NULL) and, if dlogp != NULL, the dense gradient fp32 [M, n] (-(*gscale or 1) / count at the
 *     targets).  Fixed summation order (every workgroup sums the M targets itself and writes its share of the gradient rows).
 * ------------------------------------------------------------------------- */
int ovqa_embed_gather(int dtype, const int64_t* tokens, const void* table, int64_t ld_table, int64_t vocab, void* out,
                      int64_t ld_out, int64_t B, int64_t T, int64_t width, int time_major, float* mask, int64_t padding_idx,
                      void* stream);
int ovqa_embed_scatter(int dtype, const int64_t* tokens, const void* drows, int64_t ld_rows, float* dtable, int64_t ld_table,
                       int64_t rows_table, int64_t B, int64_t T, int64_t width, int time_major, int64_t padding_idx,
                       int accumulate, void* stream);
int ovqa_decoder_inputs(const int64_t* tokens, const float* emb, const float* pos_table, int64_t pos_rows, float* out,
                        float* self_mask, int64_t B, int64_t T, int64_t D, int64_t padding_idx, void* stream);
int ovqa_dropout_apply(int dtype, const void* x, void* y, int64_t n, const ovqa_dropout* drop, void* stream);
int ovqa_pool_fwd(int feat_dtype, int dtype, const void* feat, const void* hpre, const float* w2, const float* b2, float* att,
                  void* pooled, float* pooled32, int64_t B, int64_t N, int64_t D, const ovqa_dropout* drop, void* stream);
int ovqa_pool_bwd(int feat_dtype, int dtype, const void* feat, const void* hpre, const float* w2, const float* att,
                  const void* dpooled, void* dh, void* dfeat, float* dw2_part, float* db2_part, int64_t B, int64_t N,
                  int64_t D, const ovqa_dropout* drop, void* stream);
int ovqa_log_softmax_fwd(int dtype, const void* x, int64_t ld, float* out, int64_t M, int64_t n, void* stream);
int ovqa_log_softmax_bwd(int dtype, const float* g, const float* logp, void* dx, int64_t ld, int64_t M, int64_t n,
                         void* stream);
int ovqa_nll_loss(const float* logp, const int64_t* target, float* loss, float* dlogp, const float* gscale, int64_t M,
                  int64_t n, int64_t ignore_index, int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OVQA_HIP_H */
