// Internal (C++) interface between api.hip and the kernel translation units.
#pragma once
#include "common.h"

namespace ovqa {

constexpr int64_t kWorkspaceBytes = 64ll << 20;  // scratch the caller provides (`ws`)

// ---- gemm_simple.hip (exact fp32 path, also instantiated for bf16) ----------
int simple_linear_fwd(int dtype, int epilogue, const void* x, int64_t ldx, const void* w, const float* bias,
                      const void* residual, int64_t ldres, void* y, int64_t ldy, void* preact,
                      int64_t M, int64_t N, int64_t K, const DropArgs& da, hipStream_t st);
int simple_linear_fwd_res32(const void* x, int64_t ldx, const void* w, const float* bias, const float* residual,
                            int64_t ldres, const float* mean, const float* rstd, const float* gamma, const float* beta,
                            float* pre, int64_t ldpre, int64_t M, int64_t N, int64_t K, const DropArgs& da, hipStream_t st);
int simple_linear_bwd_data(int dtype, const void* dy, int64_t lddy, const void* w, void* dx, int64_t lddx,
                           const void* preact, const void* addend, int64_t ldadd, int64_t M, int64_t N, int64_t K,
                           const DropArgs& da, hipStream_t st);
int simple_linear_bwd_weight(int dtype, const void* dy, int64_t lddy, const void* x, int64_t ldx, float* dw,
                             float* db, int64_t M, int64_t N, int64_t K, int accumulate, int accumulate_db,
                             hipStream_t st);
int simple_batched_gemm(int dtype, int c_dtype, int ta, int tb, const void* A, int64_t lda, int64_t sa,
                        const void* B, int64_t ldb, int64_t sb, void* C, int64_t ldc, int64_t sc, int64_t batch,
                        int64_t M, int64_t N, int64_t K, float alpha, hipStream_t st);
int simple_pointer_score(int dtype, const void* q, const void* k, const float* add_mask, const uint8_t* key_fill,
                         const uint8_t* query_fill, float* scores, int64_t B, int64_t T, int64_t Nk, int64_t D,
                         float scale, hipStream_t st);

// ---- attention_simple.hip ----------------------------------------------------
struct AttnArgs {
  const void *q, *k, *v;
  int64_t ldq, ldk, ldv;
  const float* mask;
  int64_t msb, msh, msq;
  void* o;
  int64_t ldo;
  float* lse;
  void* att;
  int B, H, nq, nk, dk, dv;
  float scale;
  DropArgs drop;  // dropout on the attention probabilities (p == 0: off); element index ((b*H+h)*nq+i)*nk+j
  void* o_lo = nullptr;  // bf16 mode, optional: o_lo = bf16(o_fp32 - float(bf16(o_fp32))), same layout as o (see AttnBwdArgs)
  int tail = 0;  // prefix-LM form (key-mask rows only): among the LAST `tail` positions, query i does not see keys j > i
};
struct AttnBwdArgs {
  const void *d_o, *q, *k, *v, *o, *d_att;
  int64_t lddo, ldq, ldk, ldv, ldo;
  const float *lse, *mask;
  int64_t msb, msh, msq;
  void *dq, *dk_, *dv_;
  int64_t lddq, lddk, lddv;
  float* delta;
  int B, H, nq, nk, dk, dv;
  float scale;
  DropArgs drop;       // as in AttnArgs
  const float* d_lse;  // gradient w.r.t. the returned log-sum-exp [B,H,nq] or nullptr
  // bf16 mode, optional: the rounding residual of o written by the forward call.  delta_i = dO_i . O_i is the one
  // place where the bf16 rounding of O is amplified: dS = P (dP - delta) is a cancellation, and for near-uniform
  // attention (a freshly initialised stack) dQ / dK are 100-3000x smaller than the terms that cancel; with
  // O = o + o_lo (16 significant bits) the error of delta drops by 2^8 (tests/test_kernels_gpu.py).
  const void* o_lo = nullptr;
};
// ---- attention_decode.hip: one query per row against an in-place K/V cache --------------
struct AttnDecodeArgs {
  const void *q, *k, *v;   // q [R, H*d] (row stride ldq); k, v: row (r / group) at kv_batch_stride, key j at ldk / ldv
  int64_t ldq, ldk, ldv, kv_batch_stride;
  const float* mask;       // additive fp32 [R, ldmask] or nullptr
  int64_t ldmask;
  void* o;                 // [R, H*d] (row stride ldo)
  int64_t ldo;
  int R, H, d, n, group;
  float scale;
};
int topk_rows(const float* x, int64_t ldx, int64_t R, int64_t V, int k, float* vals, int64_t* idx, hipStream_t st);
bool attention_decode_supported(const AttnDecodeArgs& a, int esize);
int attention_decode(int dtype, const AttnDecodeArgs& a, hipStream_t st);

// ---- decode_glue.hip: the index / elementwise work around a decoding step ---------------
struct BeamCommitArgs {
  const float* vals; const int64_t* idx; const float* wl;  // [b_s, cur, k] survivors of beam_candidates
  const float* seq_mask_in;                                // [b_s, cur]
  const int64_t* out_in; const float* lp_in;               // [b_s, cur, T] histories (columns < t live)
  int64_t* out_out; float* lp_out;                         // [b_s, beam, T]
  float* seq_logprob_out; float* seq_mask_out;             // [b_s, beam]
  int32_t* selected_beam; int64_t* words;                  // [b_s, beam]
  int cur, k, beam, t, T;
};
int decode_embed(int out_dtype, const int64_t* tokens, const float* emb, int64_t ld_emb, int64_t vocab, const float* pos,
                 int64_t ld_pos, int64_t n_pos, int64_t* seq, int64_t pad_idx, float mask_value, float* mask,
                 int64_t ld_mask, int64_t col, float* x32, void* x, int64_t R, int64_t D, hipStream_t st);
int beam_candidates(int dtype, const void* logits, int64_t ld, int64_t R, int64_t V, int k, const float* seq_logprob,
                    float* seq_mask, const int64_t* prev_words, int64_t eos, float* vals, int64_t* idx, float* wl,
                    hipStream_t st);
int beam_commit(const BeamCommitArgs& a, int64_t b_s, hipStream_t st);

int simple_attention_fwd(int dtype, const AttnArgs& a, hipStream_t st);
int simple_attention_bwd(int dtype, const AttnBwdArgs& a, hipStream_t st);

// ---- attention_mfma.hip (bf16, d=64, nk<=256) --------------------------------------
bool mfma_attention_supported(const AttnArgs& a);
int mfma_attention_fwd(const AttnArgs& a, hipStream_t st);
bool mfma_attention_qkv_supported(const AttnArgs& a, int64_t Dm, int64_t ldx, int64_t ldqkv, const void* x, const void* w,
                                  const void* qkv);
bool mfma_attention_q_supported(const AttnArgs& a, int64_t Dm, int64_t ldx, int64_t ldq, const void* x, const void* w,
                                const void* q);
int mfma_attention_q_fwd(const AttnArgs& a, const void* x, int64_t ldx, const void* w, const float* bias, void* q,
                         int64_t ldq, int64_t Dm, hipStream_t st);
int mfma_attention_qkv_fwd(const AttnArgs& a, const void* x, int64_t ldx, const void* w, const float* bias, void* qkv,
                           int64_t ldqkv, int64_t Dm, hipStream_t st);
bool mfma_attention_bwd_supported(const AttnBwdArgs& a);
bool mfma_attention_bwd_do_supported(const AttnBwdArgs& a, int64_t Dm, int64_t lddy, int64_t ldwt, const void* dy,
                                     const void* wt);
int mfma_attention_bwd_do(const AttnBwdArgs& a, const void* dy, int64_t lddy, const void* wt, int64_t ldwt, int64_t Dm,
                          hipStream_t st);
int mfma_attention_bwd(const AttnBwdArgs& a, hipStream_t st);

// ---- layernorm.hip -----------------------------------------------------------
int layernorm_fwd(int dtype, int in_dtype, const void* x, const float* gamma, const float* beta, const float* pos,
                  int64_t pos_rows, void* y, float* y32, float* mean, float* rstd, int64_t M, int64_t D, float eps,
                  hipStream_t st);
int layernorm_bwd(int dtype, int dx_dtype, const void* dy, const void* x, int x_dtype, const float* gamma,
                  const float* mean, const float* rstd, void* dx, void* dx_dropped, float* dgamma, float* dbeta,
                  int64_t M, int64_t D, int accumulate, const DropArgs& da, void* ws, hipStream_t st);
int layernorm_bwd_blocks(int64_t M, int64_t D);
int layernorm_bwd_waves(int64_t M, int64_t D);
int grouped_partial_reduce(const ovqa_reduce_problem* probs, int n, int max_blocks, int max_D, hipStream_t st);

// ---- misc.hip ------------------------------------------------------------------
int adam_step(float* param, const void* grad, int grad_dtype, float* m, float* v, void* shadow, int64_t n, float lr,
              const float* lr_scale_ptr, float b1, float b2, float eps, float wd, float grad_scale,
              const uint32_t* step_ptr, hipStream_t st);
int adam_step_tiled(float* param, const void* grad, int grad_dtype, float* m, float* v, void* shadow, void* shadow_t,
                    const ovqa_adam_tile* tiles, int n_tiles, int64_t flat_lo, int64_t flat_hi, float lr,
                    const float* lr_scale_ptr, float b1, float b2, float eps, float wd, float grad_scale,
                    const uint32_t* step_ptr, hipStream_t st);
int increment_step(uint32_t* step_ptr, uint32_t* second, hipStream_t st);
int begin_step(uint32_t* step_ptr, uint32_t* second, const float* lr_table, uint32_t n_table, float* lr_out,
               hipStream_t st);
int cast(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, hipStream_t st);
int dropout_keep_mask(const DropArgs& da, uint8_t* out, int64_t n, hipStream_t st);
int gelu_bwd(int dtype, const void* dy, const void* u, void* du, int64_t n, const DropArgs& da, hipStream_t st);
int row_padding_mask(int dtype, const void* x, float* mask, int64_t M, int64_t D, float pad_value, hipStream_t st);
int sq_loss_fwd_bwd(int dtype, const void* x, const void* target, void* dx, float* loss, int64_t n,
                    int accumulate_loss, hipStream_t st);

// ---- lstm.hip ------------------------------------------------------------------
bool lstm_persistent_supported(int dtype, int64_t B, int64_t T, int64_t I, int64_t H, int64_t ldx);
int64_t lstm_persistent_max_batch();
int lstm_status_read_clear(unsigned* out, hipStream_t st);
int64_t lstm_saved_bytes(int64_t B, int64_t T, int64_t H);
int64_t lstm_scratch_bytes(int64_t B, int64_t T, int64_t H);
int lstm_fwd(int dtype, bool persistent, const void* x, int64_t ldx, const void* w_ih, const void* w_hh,
             const float* b_ih, const float* b_hh, float* y, void* y16, void* hseq, void* saved, void* scratch, int64_t B,
             int64_t T, int64_t I, int64_t H, hipStream_t st);
int lstm_bwd(int dtype, bool persistent, const void* dy, int dy_bf16, const void* w_hh, const void* w_hh_t, int64_t ldwt,
             const void* saved, void* dgates, void* scratch, int64_t B, int64_t T, int64_t H, hipStream_t st);

// ---- model_ends.hip: embedding rows, dropout, attention pooling, log_softmax + NLL ---------------------------------------
int embed_gather(int dtype, const int64_t* tokens, const void* table, int64_t ld_table, int64_t vocab, void* out,
                 int64_t ld_out, int64_t B, int64_t T, int64_t width, int time_major, float* mask, int64_t padding_idx,
                 hipStream_t st);
int decoder_inputs(const int64_t* tokens, const float* emb, const float* pos_table, float* out, float* self_mask, int64_t B,
                   int64_t T, int64_t D, int64_t padding_idx, hipStream_t st);
int embed_scatter(int dtype, const int64_t* tokens, const void* drows, int64_t ld_rows, float* dtable, int64_t ld_table,
                  int64_t rows_table, int64_t B, int64_t T, int64_t width, int time_major, int64_t padding_idx,
                  int accumulate, hipStream_t st);
int dropout_apply(int dtype, const void* x, void* y, int64_t n, const DropArgs& da, hipStream_t st);
int pool_fwd(int feat_dtype, int dtype, const void* feat, const void* hpre, const float* w2, const float* b2, float* att,
             void* pooled, float* pooled32, int64_t B, int64_t N, int64_t D, const DropArgs& da, hipStream_t st);
int pool_bwd(int feat_dtype, int dtype, const void* feat, const void* hpre, const float* w2, const float* att,
             const void* dpooled, void* dh, void* dfeat, float* dw2_part, float* db2_part, int64_t B, int64_t N, int64_t D,
             const DropArgs& da, hipStream_t st);
int log_softmax_fwd(int dtype, const void* x, int64_t ld, float* out, int64_t M, int64_t n, hipStream_t st);
int log_softmax_bwd(int dtype, const float* g, const float* logp, void* dx, int64_t ld, int64_t M, int64_t n, hipStream_t st);
int nll_loss(const float* logp, const int64_t* target, float* loss, float* dlogp, const float* gscale, int64_t M, int64_t n,
             int64_t ignore_index, int accumulate, hipStream_t st);

// ---- gather.hip -----------------------------------------------------------------
int grouped_row_gather(const ovqa_gather_problem* probs, int n_problems, const int32_t* sel, int b_s, int cur, int beam,
                       hipStream_t st);

// ---- gemm_mfma.hip (bf16, MFMA) ----------------------------------------------
bool mfma_gemm_supported(int64_t R, int64_t C, int64_t K, int64_t ld_p, int64_t ld_q);
bool mfma_linear_fwd_split3_supported(const void* x, int64_t ldx, const void* w, const float* bias, const void* y0,
                                      int64_t ld0, const void* y1, int64_t ld1, const void* y2, int64_t ld2, int64_t M,
                                      int64_t F, int64_t K);
int mfma_linear_fwd_split3(const void* x, int64_t ldx, const void* w, const float* bias, void* y0, int64_t ld0, void* y1,
                           int64_t ld1, void* y2, int64_t ld2, int64_t M, int64_t F, int64_t K, hipStream_t st);
bool mfma_batched_nt_supported(const void* A, int64_t lda, int64_t sa, const void* B, int64_t ldb, int64_t sb,
                               int64_t batch, int64_t M, int64_t N, int64_t K);
int mfma_pointer_score(const void* q, const void* k, const float* add_mask, const uint8_t* key_fill,
                       const uint8_t* query_fill, float* scores, int64_t B, int64_t T, int64_t Nk, int64_t D, float scale,
                       hipStream_t st);
int mfma_batched_nt(int c_dtype, const void* A, int64_t lda, int64_t sa, const void* B, int64_t ldb, int64_t sb, void* C,
                    int64_t ldc, int64_t sc, int64_t batch, int64_t M, int64_t N, int64_t K, float alpha, hipStream_t st);
bool mfma_linear_bwd_data_supported(int64_t M, int64_t N, int64_t K, int64_t lddy, int64_t lddx);
int mfma_linear_bwd_data(const void* dy, int64_t lddy, const void* w, void* dx, int64_t lddx, const void* preact,
                         const void* addend, int64_t ldadd, int64_t M, int64_t N, int64_t K, const DropArgs& da,
                         hipStream_t st);
int mfma_linear_bwd_data_wt(const void* dy, int64_t lddy, const void* wt, int64_t ldwt, void* dx, int64_t lddx,
                            const void* preact, const void* addend, int64_t ldadd, int64_t M, int64_t N, int64_t K,
                            const DropArgs& da, hipStream_t st);
int grouped_transpose_bf16(const ovqa_transpose_problem* probs, int n, int max_tiles, hipStream_t st);
int mfma_grouped_wgrad(const ovqa_wgrad_problem* probs_dev, const int32_t* tiles_dev, int64_t n_tiles, bool direct_to_lds,
                       hipStream_t st);
int mfma_grouped_linear_bwd_weight_adam(const ovqa_wgrad_problem* probs_dev, const int32_t* tiles_dev, int64_t n_tiles,
                                        const ovqa_adam_target* targets_dev, const ovqa_adam_consts& consts, hipStream_t st);
bool mfma_linear_bwd_weight_supported(int64_t M, int64_t N, int64_t K, int64_t lddy, int64_t ldx);
int mfma_linear_bwd_weight(const void* dy, int64_t lddy, const void* x, int64_t ldx, float* dw, float* db, int64_t M,
                           int64_t N, int64_t K, int accumulate, int accumulate_db, hipStream_t st);
int colsum_bf16(const void* dy, int64_t lddy, float* db, int64_t M, int64_t N, int accumulate, hipStream_t st);
bool mfma_linear_fwd_supported(int epilogue, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldy, int64_t ldres);
int mfma_linear_fwd(int epilogue, const void* x, int64_t ldx, const void* w, const float* bias, const void* residual,
                    int64_t ldres, void* y, int64_t ldy, void* preact, int64_t M, int64_t N, int64_t K,
                    const DropArgs& da, hipStream_t st);

int mfma_linear_fwd_res32(const void* x, int64_t ldx, const void* w, const float* bias, const float* residual,
                          int64_t ldres, const float* mean, const float* rstd, const float* gamma, const float* beta,
                          float* pre, int64_t ldpre, int64_t M, int64_t N, int64_t K, const DropArgs& da, hipStream_t st);

}  // namespace ovqa
