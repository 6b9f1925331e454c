/* unast_hip.h — C ABI of libunast_hip.so: hand-written HIP kernels (gfx950 / MI355X) for the UNAST
 * adversarial speech/text train-step hot path.
 *
 * The reference (Lucaskabela/UNAST) has no FFI: its arithmetic is reached through torch.nn modules and
 * torch.nn.functional calls.  Each entry point below therefore cites the reference call site (file:line in
 * /root/reference) whose arithmetic it replaces.  Conventions:
 *   - every function returns 0 on success or a negative status; unast_last_error() gives the message;
 *   - the library never allocates, frees or synchronises: all pointers are caller-owned DEVICE pointers,
 *     borrowed for the duration of the work enqueued on `stream`;
 *   - activations are fp32, batch-first, row-major [B*T, C]; `lens` are int32 device arrays [B];
 *   - nsplit selects the MFMA operand precision: 1 = bf16 operands, 3 = split-bf16 (hi/lo) operands,
 *     both with fp32 accumulation;
 *   - dropout/noise masks are a pure function of (seed, stream_id, row, col) so forward and backward agree.
 */
#ifndef UNAST_HIP_H
#define UNAST_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

const char* unast_last_error(void);
int unast_version(void);
const char* unast_arch(void);

/* Dense contraction C[M,N] = epi(alpha * A*B^T) on the MFMA pipe.  Replaces torch.nn.Linear / nn.Conv1d /
 * in-/out-projection GEMMs and their autograd backward: src/module.py:31,63 (Linear/Conv wrappers),
 * src/module.py:273-274,286-287 (torch Transformer layers), src/module.py:152-153,240 (heads),
 * src/module.py:306-310 (LSTM input projections, reduce_h_W), src/network.py:182 (fc2).
 * a_mode/b_mode: 0 K-contiguous, 1 K-contiguous conv gather (A only), 2 row-contiguous,
 *                3 conv-dgrad weights (B only), 4 conv-wgrad gather (B only).
 * K must be a multiple of 4 for K-contiguous operands (zero-pad); kb_valid (<= K, 0 = K) is the number of rows of a
 * row-contiguous B that really exist, so a zero-padded A can be multiplied by an un-padded weight.
 * Epilogue: x = alpha*acc (+bias[n]) -> relu if act==1 -> dropout(drop_p) -> gate (G>0 ? x*gate_scale : 0)
 *           -> + R[m,n] -> (+C if beta).  split-K (splitk>1, plain alpha/beta epilogue only): partial sums go to
 *           splitk_ws ([splitk][M][ceil4(N)] floats, caller-owned) and a second launch reduces them into C; with
 *           splitk_ws == NULL the partials are added to C with fp32 atomics (needs beta=1).
 * rowsum_a (a_mode 2 only, may be NULL): rowsum_a[m] += sum_k A[m][k] — the bias gradient sum_tokens dY fused into
 *           the weight-gradient pass that already streams dY.
 * tile_wn: 2 = 128x128 tile / 4 waves, 8 = 128x128 tile / 8 waves, 4 = 128x256 tile / 8 waves; 0 lets the library
 *           choose (8 waves for forward / dgrad forms, 4 waves for weight gradients).
 * b_presplit: B points into a copy of the weights kept in the pre-split operand format written by unast_adamw /
 *           unast_split_f32 (same byte offsets as the fp32 weights), so the kernel does not re-split them per row panel.
 * out_split: C is stored in that same format (per 4 consecutive columns one 16-byte chunk [hi x4 | lo x4] of bf16 at the fp32
 *           byte offset): the in-projection outputs Q / K / V and the out-projection's input gradient dO, which only the
 *           attention kernels read (qkv_split there); needs N % 4 == 0, beta = 0, no split-K.
 * colstats: (conv forward form only: a_mode = OP_KC_CONV, b_mode = OP_KC, bias-only epilogue) 2*N doubles, zeroed by the caller, that
 *           receive sum_m C[m][n] and sum_m C[m][n]^2: the batch statistics of the BatchNorm1d that follows every Conv1d of the
 *           prenet / postnet stacks (src/module.py:162-165, 223-230), taken in the GEMM epilogue instead of by a pass over C
 *           (unast_bn_fwd have_sums = 1).  NULL: not computed. */
int unast_gemm(int a_mode, int b_mode, int nsplit,
               const float* A, int lda, const float* B, int ldb, float* C, int ldc,
               int M, int N, int K, int kb_valid,
               int conv_T, int conv_ca, int conv_cb, int conv_shift,
               const float* bias, const float* R, int ldr, const float* G, int ldg, float gate_scale,
               float alpha, int beta, int act,
               float drop_p, unsigned int seed, unsigned int stream_id,
               int splitk, float* splitk_ws, int64_t splitk_ws_floats, float* rowsum_a, int tile_wn, int b_presplit,
               int out_split, double* colstats, hipStream_t stream);

/* Fused multi-head attention core (head_dim 64), flash-style.  Replaces the softmax(QK^T/sqrt(d)+mask) -> dropout -> V
 * core of torch.nn.MultiheadAttention inside torch.nn.TransformerEncoderLayer/DecoderLayer
 * (src/module.py:273-274, 286-287; masks built at src/network.py:404-415, src/utils.py:77-83).
 * Q [B,Tq,ldq], K/V [B,Tk,ld*]: head h occupies columns [64h, 64h+64).  Keys >= lens_k[b] are masked (lens_k may be
 * NULL); causal masks key > query.  O [B,Tq,ldo] (heads concatenated), LSE [B,H,Tq] saved for the backward.
 * qkv_split = 1: Q, K, V (and dO in the backward) are in the pre-split operand format a unast_gemm with out_split = 1 wrote
 * (same shapes and strides; per 4 consecutive columns one 16-byte chunk [hi x4 | lo x4] of bf16). */
int unast_attn_fwd(int nsplit, const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                   float* LSE, const int* lens_k, int B, int H, int Tq, int Tk, int head_dim, int causal, float scale,
                   float drop_p, unsigned int seed, unsigned int stream_id, int qkv_split, hipStream_t stream);
/* Forward kernel choice: 1 = 32x32x16 MFMA tiles (default), 0 = 16x16x32 tiles; m32 < 0 only queries.  Returns the previous value.
 * Both compute the call above (same masks, same dropout decisions, same three-term products); tests compare them. */
int unast_attn_fwd_variant(int m32);
/* Backward of the above (autograd of the same torch call sites): dQ, dK, dV from dO; delta_ws is [B,H,Tq] scratch.
 * fused = 1: one pass over the (query, key) tiles produces all three (dS crosses LDS for dQ, key blocks are summed into dQ with
 * fp32 atomics, dQ is zeroed first); fused = 0: a dQ kernel and a dK/dV kernel that each recompute the probabilities;
 * fused = 2: the one-pass form with the kernel-made operands P and dS as ONE bf16 part in the dV, dK and dQ products (two MFMAs per
 * product instead of three; ~2e-3 of the largest element on the three gradients instead of ~2e-5: a fast mode, not the default).
 * lens_q (may be NULL; fused = 1 only): the caller's guarantee that dO[b, t, :] == 0 for t >= lens_q[b] -- those query tiles are not
 * visited (their dQ stays zero, they add nothing to dK / dV).  An encoder's self-attention in the train step: nothing downstream of an
 * encoder reads a padded position, so their gradients are exactly zero (opt-in: unast_amd.config.ENC_SKIP_PAD_GRADS). */
int unast_attn_bwd(int nsplit, const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* O, int ldo,
                   const float* dO, int lddo, const float* LSE, float* delta_ws, float* dQ, int lddq, float* dK, int lddk,
                   float* dV, int lddv, const int* lens_k, int B, int H, int Tq, int Tk, int head_dim, int causal, float scale,
                   float drop_p, unsigned int seed, unsigned int stream_id, int fused, int qkv_split, const int* lens_q, hipStream_t stream);

/* LayerNorm(eps) of the post-LN transformer blocks (norm1/2/3 inside torch layers, src/module.py:273-274,286-287).
 * bwd: dz (and optionally dz_drop = dz * dropout mask/(1-p), the gradient of the dropped sub-layer output);
 * dgamma/dbeta are ACCUMULATED (+=) and may be NULL for frozen parameters (then ws may be NULL too). */
int unast_layernorm_fwd(const float* z, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                        int rows, int C, float eps, hipStream_t stream);
int unast_layernorm_bwd(const float* dy, const float* z, const float* gamma, const float* mean, const float* rstd,
                        float* dz, float* dz_drop, float* dgamma, float* dbeta, float* ws, int64_t ws_floats, int rows, int C,
                        float drop_p, unsigned int seed, unsigned int stream_id, int finalize, hipStream_t stream);
/* finalize = 0 leaves the per-workgroup partial sums of dgamma / dbeta in ws; this entry reduces them into dgamma / dbeta
 * (accumulating).  The parameter gradients are read by the optimizer only, so the caller may run it on another stream. */
int unast_layernorm_bwd_finalize(const float* ws, int64_t ws_floats, int rows, int C, float* dgamma, float* dbeta, hipStream_t stream);
/* floats of `ws` unast_layernorm_bwd needs for (rows, C): per-workgroup column partials, reduced by a second launch
 * (no same-address atomics). */
int64_t unast_layernorm_bwd_ws_floats(int rows, int C);

/* Column sums sum[c] += sum_r x[r,c] (bias gradients of every nn.Linear/nn.Conv1d on the path). */
int unast_colsum_f32(const float* x, int ldx, int rows, int C, float* sum, hipStream_t stream);
/* The same sums in a FIXED order (row chunks in order, then the chunks in order; no atomics): what unast_amd.utils.set_deterministic(True)
 * -- the parity mode of SURVEY.md Appendix C -- selects, together with unast_embed_bwd_det, the two-kernel attention backward and bias
 * gradients taken by this function instead of the weight-gradient GEMM's atomics, so that two runs of a step, eager or replayed, agree
 * to the bit in every fp32 sum.  ws: unast_colsum_det_ws_floats(rows, C) floats. */
int unast_colsum_det(const float* x, int ldx, int rows, int C, float* sum, float* ws, int64_t ws_floats, hipStream_t stream);
int64_t unast_colsum_det_ws_floats(int rows, int C);

/* Train-mode BatchNorm1d + activation (1 relu, 2 tanh) + dropout over [rows=B*T, C]: TextPrenet.forward_fcn
 * (src/module.py:223-230) and SpeechPostnet.forward (src/module.py:162-165).  Updates running stats (momentum,
 * unbiased variance) when running_mean != NULL.  ws: 2*C doubles of scratch; have_sums = 1: ws already holds the column
 * sums and sums of squares of x (written by unast_gemm colstats), the statistics pass over x is skipped.  The statistics are
 * finalized inside the normalising pass itself (no separate launch); num_batches_tracked (may be NULL): the layer's int64 batch
 * counter, incremented by that same launch (nn.BatchNorm1d's buffer of the same name).
 * bwd: dy_inout is overwritten with d(pre-activation); dgamma/dbeta accumulated (may be NULL); ws_zeroed = 1: the caller hands
 * over 2*C doubles that are already zero (no memset launch). */
int unast_bn_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                 float* running_mean, float* running_var, double* ws, int rows, int C,
                 float eps, float momentum, int act, float drop_p, unsigned int seed, unsigned int stream_id,
                 int have_sums, int64_t* num_batches_tracked, hipStream_t stream);
/* Eval-mode BatchNorm1d + activation (model.eval(): running statistics, no dropout), as evaluate() runs it
 * (src/train.py:484; src/module.py:162-165, 223-230).  mean/rstd: C floats of scratch. */
int unast_bn_eval_fwd(const float* x, const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float* y, float* mean, float* rstd, int rows, int C, float eps, int act,
                      hipStream_t stream);
int unast_bn_bwd(float* dy_inout, const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                 float* dx, float* dgamma, float* dbeta, double* ws, int rows, int C, int act,
                 float drop_p, unsigned int seed, unsigned int stream_id, int ws_zeroed, hipStream_t stream);

/* nn.Embedding(padding_idx=0) + emb_dropout + noise_fn (src/module.py:189,226; src/network.py:429-438, 483-487;
 * src/utils.py:40-49).  shift_sos >= 0 builds the decoder input [SOS, ids[:-1]] on the fly. dE is accumulated. */
int unast_embed_fwd(const int64_t* ids, const float* E, float* out, int rows, int T, int D, int shift_sos,
                    float drop_p, unsigned int seed, unsigned int stream_id, float noise_p, unsigned int noise_stream,
                    hipStream_t stream);
int unast_embed_bwd(const int64_t* ids, const float* dout, float* dE, int rows, int T, int D, int vocab, int shift_sos,
                    int padding_idx, float drop_p, unsigned int seed, unsigned int stream_id, float noise_p,
                    unsigned int noise_stream, hipStream_t stream);
/* unast_embed_bwd with a fixed summation order (one workgroup per vocabulary id walks the rows in order; parity mode, see unast_colsum_det). */
int unast_embed_bwd_det(const int64_t* ids, const float* dout, float* dE, int rows, int T, int D, int vocab, int shift_sos,
                        int padding_idx, float drop_p, unsigned int seed, unsigned int stream_id, float noise_p,
                        unsigned int noise_stream, hipStream_t stream);

/* PositionalEncoding.forward (src/module.py:265-267): y = dropout(x*scale + pe[t]); bwd optionally gated by gate>0. */
int unast_posenc_fwd(const float* x, const float* pe, float* y, int rows, int T, int D, float scale, float drop_p,
                     unsigned int seed, unsigned int stream_id, hipStream_t stream);
int unast_posenc_bwd(const float* dy, const float* gate, float* dx, int rows, int D, float scale, float drop_p,
                     unsigned int seed, unsigned int stream_id, hipStream_t stream);

/* noise_fn (src/utils.py:40-49): zero whole timesteps with probability p, no rescale. */
int unast_rowmask(const float* x, float* y, int rows, int D, float p, unsigned int seed, unsigned int stream_id,
                  hipStream_t stream);
/* a += b : gradient accumulation for activations consumed by several ops (autograd's implicit add). */
int unast_add_inplace(float* a, const float* b, int64_t n, hipStream_t stream);
/* dst = a + b over n floats (b may be NULL: a copy).  The gradients that reach the two halves of a paired encoder call (its decoder's
 * and the discriminator's, autograd of TextTransformer / SpeechTransformer.encode, src/network.py:203-208, 427-444) are summed straight
 * into their row block of the stack's gradient buffer; a kernel, not a memcpy, so that it can live in a captured step. */
int unast_sum2(float* dst, const float* a, const float* b, int64_t n, hipStream_t stream);
/* Autoregressive inference helpers (TextTransformer/SpeechTransformer.infer_sequence, src/network.py:219-252, 455-481):
 * first-maximum argmax per row of the logits, and zeroing of generated frames/tokens at t >= lens[b] (int64 lengths). */
int unast_argmax_rows(const float* x, int ld, int rows, int cols, int64_t* out, hipStream_t stream);
int unast_mask_by_len(float* x, const int64_t* lens, int B, int T, int D, hipStream_t stream);
/* ---- one autoregressive decoding position (csrc/decode.hip; reference: infer_sequence, src/network.py:219-252, 455-481) -------
 * The position is kept in DEVICE memory (*pos), so a position's launches can be replayed from a captured graph.
 *
 * unast_decode_linear: latency-built contraction for the few rows of one position: Y[M,N] = epilogue(X'[M,K] . W[N,K]^T), epilogue
 * as unast_gemm (bias, act 1 = ReLU, dropout, + R).  X rows start at X + *pos * x_pos_stride (x_pos_stride 0: plain X).
 * prologue selects X' (for 1..4: K <= 256; xn_out, if not NULL, receives X'):
 *   0  X
 *   1  LayerNorm(X)                          (the decoder layers' norm1/2/3, nn.TransformerDecoderLayer as built by src/module.py:282-291)
 *   2  dropout(LayerNorm(X), pro_drop2)      (text postnet: dropout before fc1, src/module.py TextPostnet)
 *   3  dropout(dropout(emb[tokens[m, *pos]], pro_drop1) * pro_scale + pe[*pos], pro_drop2)    (TextPrenet + PositionalEncoding, src/module.py:265-267)
 *   4  dropout(X * pro_scale + pe[*pos], pro_drop2)                                           (PositionalEncoding on the speech prenet output)
 * With `cache`, output columns >= split_col go to cache[(m*cache_rows + *pos)*ld_cache + n - split_col] (the self-attention K|V of
 * this position) and Y receives the columns below split_col only. */
int unast_decode_linear(const float* X, int ldx, int64_t x_pos_stride, const float* W, int ldw, const float* bias, float* Y, int ldy, int M, int N, int K,
                        int act, float drop_p, unsigned int seed, unsigned int stream_id, const float* R, int ldr,
                        int prologue, const float* ln_gamma, const float* ln_beta, float ln_eps,
                        const int64_t* tokens, int ld_tok, const float* emb, const float* pe, float pro_scale,
                        float pro_drop1, unsigned int pro_stream1, float pro_drop2, unsigned int pro_stream2,
                        float* xn_out, int ld_xn,
                        float* cache, int ld_cache, int cache_rows, int split_col, const int64_t* pos, hipStream_t stream);
/* One query per (sequence, head) against cached keys/values (head dim 64): O[b, 64h..] = dropout(softmax(scale * q.K^T over the
 * first n_b of the rows_per_seq cached rows of sequence b)) . V with n_b = lens[b], or -- lens NULL -- min(stop_lens[b]+1, *pos+1),
 * the reference's dec_mask (src/network.py:226-231, 461-465): the attention of torch's multi_head_attention_forward as the
 * reference's decoder layers call it at the last position of the prefix (src/network.py:232, 466). */
int unast_decode_attn(const float* Q, int ldq, const float* K, const float* V, int ldkv, int rows_per_seq, const int* lens,
                      const int64_t* stop_lens, const int64_t* pos, float* O, int ldo,
                      int B, int H, float scale, float drop_p, unsigned int seed, unsigned int stream_id, hipStream_t stream);
/* End of a position: prediction -> position *pos+1 (tokens[b,*pos+1] = argmax logits; outputs[b,*pos+1,:] / stops[b,*pos+1] =
 * head[b,:M] / head[b,M]), stop rule (argmax == EOS, src/network.py:470-472; sigmoid(stop) >= .5, src/network.py:240-243), then
 * *pos += 1 and *epoch += 1 (epoch may be NULL; see unast_set_rng_epoch). */
int unast_decode_end_text(const float* logits, int ld, int V, int B, int64_t* tokens, int ld_tok, int64_t* stop_lens, int64_t max_len, int eos,
                          int64_t* pos, int* epoch, hipStream_t stream);
int unast_decode_end_speech(const float* head, int ld, int M, int B, float* outputs, int ld_out, float* stops, int ld_stop, int64_t* stop_lens,
                            int64_t max_len, int64_t* pos, int* epoch, hipStream_t stream);
/* a *= alpha: averaging of all-reduced gradients across data-parallel ranks (new vs. the single-device reference). */
int unast_scale_inplace(float* a, float alpha, int64_t n, hipStream_t stream);
/* Test infrastructure: keeps `stream` busy for `us` microseconds (one spinning wave).  unast_amd.config.STREAM_JITTER puts one of random
 * length at the head of every side-stream call to flush out missing cross-stream dependencies (tests/test_gpu_streams.py). */
int unast_spin(int us, hipStream_t stream);
/* Teacher-forced decoder input (src/network.py:254-262): dst[b,0,:] = 0, dst[b,t,:] = src[b,t-1,:] for [B,T,M] fp32 (M % 4 == 0). */
int unast_shift_frames(const float* src, float* dst, int B, int T, int M, hipStream_t stream);
/* dst[:, :cols] += src[:, :cols] with independent row strides (autograd's add for padded gradient buffers). */
int unast_add_strided(float* dst, int ldd, const float* src, int lds, int rows, int cols, hipStream_t stream);
/* specaugment (src/utils.py:51-75), including its quirk of masking two TIME spans and no frequency columns. */
int unast_specaugment(const float* mel, const int* lens, float* out, float* ws /* B floats of scratch */, int B, int T, int M,
                      int freq_mask, int time_mask, unsigned int seed, unsigned int stream_id, hipStream_t stream);
/* discriminator_shuffle_batch (src/train.py:296-329): pad to common T, concat on batch, permute; and its backward. */
int unast_disc_gather(const float* t_hid, const float* s_hid, const int* t_len, const int* s_len, const int64_t* perm,
                      float* out, int* out_len, int B, int Tt, int Ts, int D, hipStream_t stream);
int unast_disc_scatter(const float* dout, const int64_t* perm, float* dt_hid, float* ds_hid, int B, int Tt, int Ts, int D,
                       hipStream_t stream);

/* The sum of a sub-step's losses over accum_steps (src/train.py:376-378, 403-405, 455: loss = (a + b + c) / accum_steps) and its
 * backward (grad / accum_steps) as one single-thread launch: out[0] = ((a[0] + b[0]) + c[0]) / div; b and c may be NULL. */
int unast_scalar_combine(const float* a, const float* b, const float* c, float div, float* out, hipStream_t stream);
/* masked_mse (src/train.py:100-103) on its own: out[0] = sum((gold - pred)^2 * mask) / sum(mask) over n elements.  ws3: three
 * doubles of workspace, zero on entry, left zero on exit.  (The train step's two masked MSEs come out of unast_speech_loss_*.) */
int unast_masked_mse(const float* gold, const float* pred, const float* mask, int64_t n, double* ws3, float* out, hipStream_t stream);
/* speech_loss (src/train.py:100-103, 113-122): head = [pre-net mel (M cols) | stop logit | pad] with row stride ldh.
 * fwd writes loss[0] in ONE launch (the last workgroup to arrive forms the scalar); bwd writes d_head (same layout) and d_post scaled by
 * the device scalar *gscale.  ws: 4 doubles, ZERO on entry, left zero on exit (no memset in front, one workspace serves call after call). */
int unast_speech_loss_fwd(const float* gold, const float* head, int ldh, const float* post, const int* lens, int B, int T, int M,
                          float eos_weight, double* ws, float* loss, hipStream_t stream);
int unast_speech_loss_bwd(const float* gold, const float* head, int ldh, const float* post, const int* lens, int B, int T, int M,
                          float eos_weight, const float* gscale, float* d_head, float* d_post, hipStream_t stream);
/* text_loss (src/train.py:105-111): weighted CE, ignore_index 0, EOS(2) weight; one launch forward.  ws: 5 doubles -- [0..3] zero on entry and
 * left zero on exit, [4] receives the weight sum the backward of THIS call reads (keep ws untouched until then). */
int unast_text_loss_fwd(const float* logits, int ldl, const int64_t* gold, int rows, int V, float eos_weight,
                        double* ws, float* loss, hipStream_t stream);
int unast_text_loss_bwd(const float* logits, int ldl, const int64_t* gold, int rows, int V, float eos_weight, const double* ws,
                        const float* gscale, float* dlogits, hipStream_t stream);
/* The text head AND its loss in one launch (north_star: "heads plus losses fused"): logits[rows, ldl] = X[rows, K = 256] W[V, K]^T + bias
 * (TextPostnet.fc1, src/module.py:243-246), text_loss of them (src/train.py:105-111: weighted cross-entropy, ignore_index 0, EOS = 2 weighted)
 * and dlogits = gscale * d(loss)/d(logits) -- the softmax runs on the head GEMM's accumulators (a workgroup holds whole rows: V <= 48),
 * the logits are stored once for the caller, the normaliser sum(w) is recomputed from `gold` by every workgroup.  gscale: the upstream
 * gradient of the loss as the HOST knows it (1 / accum_steps in the train step); ws: as unast_text_loss_fwd. */
int unast_text_head_loss(const float* X, int ldx, const float* W, const float* bias, const int64_t* gold, int rows, int K, int V,
                         float eos_weight, float gscale, float* logits, float* dlogits, int ldl, double* ws, float* loss, hipStream_t stream);
/* The speech heads AND their loss terms in one launch: head[B T, ldh] = X [linear_project | stop_linear]^T + bias (M mel columns, the stop
 * logit, zero padding; src/module.py:170-171), the pre-net masked MSE and the stop-token BCE of speech_loss (src/train.py:113-122) and
 * d_head = gscale * d(loss)/d(head), from the head GEMM's accumulators.  ws (doubles, zero on entry) receives the two partial sums; the
 * post-net term and the scalar follow with unast_speech_post_loss once the post-net has run (same stream): it adds
 * sum mask (gold - post)^2, writes d_post = gscale * d(loss)/d(post), forms the loss and leaves ws zero. */
int unast_speech_head_loss(const float* X, int ldx, const float* W, const float* bias, const float* gold, const int* lens, int B, int T, int K, int M,
                           float eos_weight, float gscale, float* head, float* d_head, int ldh, double* ws, hipStream_t stream);
int unast_speech_post_loss(const float* gold, const float* post, const int* lens, int B, int T, int M, float gscale, float* d_post, double* ws,
                           float* loss, hipStream_t stream);
/* discriminator_target (src/train.py:150-164, 319-320): smoothed labels 0.9 (text rows: perm[i] < B) / 0.1 (speech),
 * flipped (1-y) when flip=1 (generator phase). */
int unast_disc_targets(const int64_t* perm, int n, int B, int flip, float smoothing, float* out, hipStream_t stream);
/* the row permutation of discriminator_shuffle_batch (torch.randperm, src/train.py:323) from the counter RNG of the dropout
 * kernels: uniform over permutations, a function of (seed, stream_id, RNG epoch), n <= 4096; capturable in a HIP graph. */
int unast_randperm(int64_t* out, int n, unsigned int seed, unsigned int stream_id, hipStream_t stream);
/* discriminator_loss (src/train.py:147-148): mean BCE-with-logits; logits/dlogits may be strided (ldx/ldd elements).
 * loss/dlogits may be NULL; dlogits is scaled by the device scalar *gscale. */
int unast_bce_logits(const float* logits, int ldx, const float* targets, int n, const float* gscale, float* loss, float* dlogits,
                     int ldd, hipStream_t stream);

/* Recurrent part of nn.LSTM over packed sequences (src/module.py:306, 315-316), hidden 64, gate order i,f,g,o.
 * xproj [Bd,T,ndir*256] = X W_ih^T (no bias), y [Bd,T,ndir*64] and hprev pre-zeroed by the caller; gates/cs saved. */
int unast_lstm_fwd(const float* xproj, const float* whh, const float* b_ih, const float* b_hh, const int* lens, float* y, float* gates,
                   float* cs, float* hprev, float* hfinal, int Bd, int T, int ndir, int hidden, int64_t whh_dir_stride,
                   int64_t bias_dir_stride, hipStream_t stream);
int unast_lstm_bwd(const float* dy, const float* dhfinal, const float* whh, const float* gates, const float* cs, const int* lens,
                   float* dgates, int Bd, int T, int ndir, int hidden, int64_t whh_dir_stride, hipStream_t stream);
/* LeakyReLU(slope)+Dropout of LSTMDiscriminator / Discriminator (src/network.py:160-170, 179-186); dy != NULL => backward. */
int unast_leaky_dropout(const float* x, const float* dy, float* out, int rows, int D, float slope, float drop_p, unsigned int seed,
                        unsigned int stream_id, hipStream_t stream);

/* Registers a 4-byte counter in device memory that every dropout / noise kernel adds to its stream id (NULL = none).  Launches
 * replayed from a captured HIP graph carry fixed (seed, stream_id) arguments; the generation loop of infer_sequence
 * (src/network.py:219-252, 455-481: fresh dropout masks at every decoded position when the model is in training mode)
 * advances the counter once per position instead.  The counter must read 0 whenever forward/backward pairs run.
 * Blocking (hipMemcpyToSymbol); call it while no kernel of this library is in flight. */
int unast_set_rng_epoch(const unsigned int* counter);

/* Several weight gradients of one backward closure in ONE launch (+ one reduction): for each problem
 * C[M,N] += A[tokens,M]^T B[tokens,N] (A = dY, B = X of an nn.Linear, src/module.py:18-39 / torch's TransformerEncoder/DecoderLayer
 * linears) and, if rowsum_a != NULL, rowsum_a[M] += column sums of A (the bias gradient).  Stands where torch's autograd
 * issues one addmm per weight in loss.backward() (src/train.py:380, 401).  count <= 8.  ws: caller-owned slab workspace of
 * unast_wgrad_group_ws_floats(...) floats; target_blocks (0 = 512) = workgroups to aim for when cutting the token reduction
 * into K-slices. */
typedef struct {
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    float* rowsum_a;
    int M, N, K;
} unast_wgrad_item;
int unast_wgrad_group(int nsplit, int count, const unast_wgrad_item* items, float* ws, int64_t ws_floats, int target_blocks,
                      hipStream_t stream);
int64_t unast_wgrad_group_ws_floats(int count, const unast_wgrad_item* items, int target_blocks);

/* optimizer_step (src/train.py:358-363): clip_grad_norm_ + torch.optim.AdamW (decoupled = 1) or torch.optim.Adam with L2
 * weight decay (decoupled = 0; optim_type 'adam', src/train.py:929-930) over flat fp32 buffers.
 * dev_hyper (may be NULL): three floats in device memory {lr, 1 - beta1^t, sqrt(1 - beta2^t)} that replace lr / step, so a
 * captured HIP graph of the step replays with the current schedule values.
 * split_out (may be NULL; needs n % 4 == 0): the updated parameters once more in the GEMM's pre-split operand format --
 * per 4 consecutive elements one 16-byte chunk [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3] of bf16, hi = RNE(x), lo = RNE(x - hi).
 * unast_split_f32 produces the same format from any fp32 buffer (model load, load_state_dict).
 * zero_grad != 0: g is set to zero behind the update (optimizer.zero_grad() of src/train.py:363 in the same pass). */
int unast_sumsq(const float* g, int64_t n, double* out, hipStream_t stream);
int unast_adamw(float* p, float* g, float* m, float* v, int64_t n, const double* sumsq, float max_norm, float lr,
                float beta1, float beta2, float eps, float weight_decay, int step, float* split_out, int decoupled,
                const float* dev_hyper, int zero_grad, hipStream_t stream);
int unast_split_f32(const float* src, float* dst, int64_t n, hipStream_t stream);
/* W^T of every 2-D weight of a parameter region once more in the pre-split operand format (dst[k][n], 4 consecutive n per 16-byte
 * chunk; destination rows padded to a multiple of 4 with zeros), refreshed after each optimizer step: the input-gradient GEMMs
 * (autograd of nn.Linear, src/module.py:18-39) then read it as a K-contiguous B operand.  tiles_dev: ntiles x 6 int32 in device
 * memory {src offset, dst offset, rows, cols, row0, col0} per 64 x 64 tile (float offsets from the bases; cols % 4 == 0, source
 * offsets multiples of 4). */
int unast_transpose_split(const float* src_base, float* dst_base, const int* tiles_dev, int ntiles, hipStream_t stream);

/* Row-panel GEMM for the K <= 256 contractions of the transformer layers, C[M,N] = epi(A[M,K] W[N,K]^T): in-projections,
 * out-projections, linear1, cross-attention projections and (with W^T planes) their input gradients -- the nn.Linear /
 * MultiheadAttention projections inside torch.nn.TransformerEncoderLayer / DecoderLayer (src/module.py:273-274, 286-287) and the
 * autograd of the same call sites.  A workgroup keeps a 128- or 64-row panel of A in registers (split into hi / lo bf16 once) and
 * streams the weights through LDS by LDS-DMA from TILED BF16 PLANES written by unast_retile_weights.
 * w_planes: hi plane of W (lo plane at + plane_bytes); N and K may be smaller than the planes' padded extents.
 * Epilogue: x = acc + bias[n] -> relu if act == 1 -> dropout(drop_p; mask = f(seed, stream_id, m, n) as unast_gemm) ->
 *           gate (G > 0 ? x * gate_scale : 0) -> + R[m,n]; out_split as unast_gemm.
 * ln_gamma != NULL (needs N = 256): the post-LN sub-layer is finished here -- C receives z = R + dropout(acc + bias),
 *           Y = LayerNorm(z) * gamma + beta (eps), mean / rstd [M] the row statistics the LayerNorm backward reads
 *           (replaces unast_layernorm_fwd behind the out-projection / linear2).
 * gate_bits (may be NULL; needs N % 64 == 0): one keep bit per element of an [M,N] activation, ceil128(M) * N / 8 bytes (whole row panels) --
 *           word ((m / 16) * (N / 16) + n / 16) * 4 + r of 64 bits holds at bit 4 * ... 16 * ((n % 16) / 4) + m % 16 whether element
 *           (m, 16 (n / 16) + 4 ((n % 16) / 4) + r) is > 0.  With act == 1 and dropout (linear1: relu -> dropout, src/module.py:273-274
 *           -> torch TransformerEncoderLayer) the launch WRITES them next to C; with act == 0 and no bias (the input gradient through
 *           linear2) it READS them as the gate x = bit ? x * gate_scale : 0 -- relu' and the dropout mask of the hidden activation
 *           without re-reading it.
 * rows_per_wg: 128 (8 waves x 32 rows), 1128 (128 rows as 16 waves x 16 rows), 64 or 0 (library picks).
 * K > 256 (needs K % 64 == 0, N % 256 == 0): the K-STREAMED form -- a workgroup owns 128 rows x 256 columns of C, K comes in groups
 *           of 64 (weights by LDS-DMA from the same planes, activations one group ahead into registers).  Serves linear2 (K = 1024) with
 *           its residual + LayerNorm epilogue and the input gradients of linear1 / the in-projections (K = 1024 / 768 / 512) with their
 *           residual-gradient operand R.  Epilogues: bias, R, and with ln_gamma the LayerNorm one (dropout only there); no act / G /
 *           out_split / gate_bits; rows_per_wg is ignored. */
int unast_panel_gemm(const float* A, int lda, const void* w_planes, int64_t plane_bytes, float* C, int ldc, int M, int N, int K,
                     const float* bias, const float* R, int ldr, const float* G, int ldg, float gate_scale, int act,
                     float drop_p, unsigned int seed, unsigned int stream_id, int out_split,
                     const float* ln_gamma, const float* ln_beta, float* Y, int ldy, float* mean, float* rstd, float eps,
                     void* gate_bits, int rows_per_wg, hipStream_t stream);
/* The weights of a parameter region once more as tiled bf16 planes for unast_panel_gemm, refreshed after each optimizer step.
 * descs_dev: ndesc records of 48 bytes in device memory, one per 64 x 64 block of a destination matrix Wd[n][k]:
 *   int32 {src offset (floats from src_base), src row stride, transposed, N, K, n0, k0, ksteps = ceil(K / 32)},
 *   int64 {dst offset of the hi plane (bytes from dst_base), plane bytes = ceil64(N) * ksteps * 64}.
 * transposed = 0: Wd[n][k] = src[n * ld + k] (forward operand); 1: Wd[n][k] = src[k * ld + n] (input-gradient operand W^T).
 * Plane layout: 1-KB sub-tiles of 16 n x 32 k at ((n / 16) * ksteps + k / 32) * 1024, inside a sub-tile 16-byte units
 * [(k % 32) / 8][n % 16] of 8 consecutive k; hi = RNE_bf16(x), lo = RNE_bf16(x - hi); padding is zero. */
int unast_retile_weights(const float* src_base, void* dst_base, const void* descs_dev, int ndesc, hipStream_t stream);
/* The input-gradient GEMM that ends a sub-layer's backward with the PREVIOUS sub-layer's LayerNorm backward in its epilogue
 * (torch post-LN layers, /root/reference/src/module.py:273-274, 286-287): dy = R + A W^T (A [M,K], K > 256, W^T as tiled planes, N = 256),
 * dz = LayerNorm-backward(dy; z, mean, rstd, gamma) -> dz, dropout(dz) -> dz_drop (NULL: none), column sums of dy xhat / dy per 128-row
 * panel -> part (unast_panel_gemm_lnbwd_ws_floats(M) floats; NULL: none), reduced into dgamma / dbeta by
 * unast_layernorm_partials_finalize(part, ceil(M / 128), 256, ...).  Replaces unast_panel_gemm + unast_layernorm_bwd where the row-panel
 * kernel serves the shape. */
int unast_panel_gemm_lnbwd(const float* A, int lda, const void* w_planes, int64_t plane_bytes, int M, int K, const float* R, int ldr,
                           const float* z, int ldz, const float* mean, const float* rstd, const float* gamma,
                           float* dz, int lddz, float* dz_drop, int lddrop, float* part, int64_t part_floats,
                           float drop_p, unsigned int seed, unsigned int stream_id, hipStream_t stream);
int64_t unast_panel_gemm_lnbwd_ws_floats(int M);
int unast_layernorm_partials_finalize(const float* part, int nblk, int C, float* dgamma, float* dbeta, hipStream_t stream);
/* Writes n <= 16 32-bit words (read from HOST memory at call time, passed by value in the kernel arguments) to device memory:
 * refreshes the block a captured step reads (RNG epoch of unast_set_rng_epoch, dev_hyper triples of unast_adamw) once per
 * replay; stands where the reference's Python passes lr / step to torch.optim (src/train.py:361, 654-655). */
int unast_set_words(unsigned int* dst, const unsigned int* host_words, int n, hipStream_t stream);

/* Stream replay of a captured HIP graph (csrc/graph_exec.cpp): the nodes of `graph` (a hipGraph_t, e.g. torch.cuda.CUDAGraph's
 * raw handle of the captured train step -- the reference has no counterpart: it launches src/train.py:602-655 op by op) are laid out
 * on `nstreams` ordinary streams with events on the cross-stream edges and re-issued with hipLaunchKernel / hipMemsetAsync /
 * hipMemcpyAsync per replay.  create returns 0 (and sets unast_last_error) for graphs with node kinds it does not handle; the
 * graph must outlive the plan.  info: {kernel nodes, memset nodes, memcpy nodes, cross-stream edges}. */
int64_t unast_graph_plan_create(void* graph, int nstreams);
int unast_graph_plan_info(int64_t plan, int* out4);
int unast_graph_plan_replay(int64_t plan, hipStream_t origin);
int unast_graph_plan_destroy(int64_t plan);
/* Gradient exchanges inside a replayed step: nodes captured by unast_allreduce_marker are issued as unast_allreduce on the communicator
 * given here (allreduces: how many such nodes the plan holds). */
int unast_graph_plan_allreduces(int64_t plan);
int unast_graph_plan_set_comm(int64_t plan, int64_t comm);
/* The element counts of the plan's gradient exchanges in the order the replay issues them (up to n written to out; returns how many the
 * plan holds).  Collectives of one communicator pair up across ranks by issue order: tests compare this list with the eager step's. */
int unast_graph_plan_allreduce_counts(int64_t plan, long long* out, int n);
/* Keeping the program's own stream structure in the replay.  A hipGraph_t does not record which stream a node was captured on; the
 * capture can: unast_capture_note(stream), called right after a launch on `stream` while a capture is open, remembers the node that
 * launch created (hipStreamGetCaptureInfo_v2: the stream's dependency set is then exactly that node) as belonging to `stream`;
 * unast_capture_reset() forgets everything (call when a capture begins).  unast_graph_plan_create then puts every noted node on a plan
 * stream of its own capture stream when UNAST_REPLAY_LABELS=1 (as many plan streams as capture streams were seen, each with its capture
 * stream's priority): the replay then overlaps exactly what the eager step overlaps (the side streams of unast_amd/engine.py over
 * /root/reference/src/train.py:602-655's sub-steps).  Off by default -- it measured slower than the DAG layout.
 * plan_streams: how many streams the plan uses. */
int unast_capture_reset(void);
int unast_capture_note(hipStream_t stream);
/* Pruning what a side stream inherits from the capture's origin stream (unast_amd/engine.py relays every hand-off between two side streams
 * through the origin: ROCm 7.2 cannot end a capture in which two side streams waited on each other; the origin's dependency set therefore
 * accumulates the producers of all hand-offs).  get_deps: the graph nodes the stream's next captured node would depend on (their number;
 * -1 not capturing, -2 more than cap).  prune: drops from the stream's dependency set every node noted (unast_capture_note) on ANOTHER side
 * stream unless it is listed in `keep` -- nodes of the stream itself, of `origin`, and un-noted nodes always stay; returns how many were
 * dropped (hipStreamUpdateCaptureDependencies). */
int unast_capture_get_deps(hipStream_t stream, void** out, int cap);
int unast_capture_prune(hipStream_t stream, hipStream_t origin, void** keep, int nkeep);
int unast_graph_plan_streams(int64_t plan);

/* Data-parallel gradient exchange over RCCL / xGMI (csrc/comm.cpp).  New with respect to the reference, which is single-device
 * (src/utils.py:101-106); the order it has to keep -- generator update before the discriminator phase -- is src/train.py:628-637.
 * unique_id: 128 bytes made by ONE rank and handed to the others by the launcher (torch.distributed's store, a file, the
 * environment); init blocks until all `world` ranks have called it.  allreduce: in-place fp32 sum of buf[0, count) over the
 * ranks, enqueued on `stream`.  marker: what a CAPTURED step records at the place of an all-reduce -- an empty kernel node carrying
 * (buf, count) that the stream-replay executor turns into unast_allreduce on the node's stream. */
int unast_comm_unique_id(void* out128);
int64_t unast_comm_init(const void* unique_id128, int rank, int world);
int unast_allreduce(int64_t comm, float* buf, int64_t count, hipStream_t stream);
int unast_allreduce_marker(float* buf, int64_t count, hipStream_t stream);
int unast_comm_destroy(int64_t comm);

#ifdef __cplusplus
}
#endif
#endif
