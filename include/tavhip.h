/* libtavhip — C ABI of the MI355X-native TAV (text+audio+video) fusion forward/backward hot path.
 *
 * The reference (g8a9/multi-modal-emotion) has no native/FFI layer: the hot path is the ATen op call-sites of
 *   models/tav.py:344-504 (PreFormer.forward, TAVForMAE.forward), utils/TAVFormer.py:10-439 (fusion encoders) and the
 *   Hugging Face modules they call (roberta/bert, wav2vec2, videomae), driven by train_model/tav_train.py:15-65.
 * Each entry point below is the operator a binding for that path would need, and cites the call-site it replaces.
 *
 * Rules of the ABI
 *   - extern "C", plain pointers and sizes only; every pointer is a DEVICE pointer unless stated otherwise.
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work on it (no allocation, no synchronisation,
 *     safe under hipGraph capture).  Workspaces are supplied by the caller.
 *   - return 0 on success, a negative TAV_ERR_* for a rejected argument (nothing was launched), or a positive
 *     hipError_t from the launch.  The Python host layer turns any non-zero value into RuntimeError.
 *   - dtype codes: TAV_F32 (exact-f32 parity path, f32-input MFMA) and TAV_BF16 (bf16 operands, f32 accumulate).
 *     Residual streams, LayerNorm statistics, softmax statistics, losses and all parameter gradients are f32.
 *   - matrices are row-major "token-major": [tokens, features]; ld* are row strides in ELEMENTS.
 */
#ifndef TAVHIP_H
#define TAVHIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { TAV_F32 = 0, TAV_BF16 = 1, TAV_FP8 = 2 /* OCP e4m3, GEMM operands of tav_gemm_nt only */ };
enum {
    TAV_ERR_NULL = -1,  /* required pointer missing           */
    TAV_ERR_SHAPE = -2, /* size not supported by the kernel   */
    TAV_ERR_DTYPE = -3, /* dtype code / combination rejected  */
    TAV_ERR_ALIGN = -4, /* stride or offset breaks 16-B access */
    TAV_ERR_NO_RCCL = -5 /* v5: tav_comm_* / tav_allreduce_bucket called, but no librccl could be loaded (it is resolved lazily) */
};

int tav_version(void);                 /* ABI version, bumped on any signature change */
const char* tav_error_string(int code);

/* ---------------------------------------------------------------------------------------------------------------
 * GEMM, "NT": C[z][m][n] = epi( alpha * sum_k A[z][m][k] * B[z][n][k] )
 *   epi(v): v += bias[n]; if C_pre: C_pre = (act & 2) ? gelu'(v) : v; if act & 1: v = gelu_erf(v);
 *           if gelu_in: v *= (act & 4) ? gelu_in[m][n] : gelu'(gelu_in[m][n]);
 *           if resid: v += resid[m][n] (f32); if accumulate: v += C[m][n]; C = v.
 *   (act 3 in the forward FFN1 + act 4 in its dgrad: the derivative of the GELU is computed once, next to the GELU itself -- they share
 *    the exponential -- and the backward epilogue is a plain multiply.)
 * Replaces nn.functional.linear / nn.Linear on the path (utils/TAVFormer.py:348-350,401-403,419,432; HF linears),
 * their input-gradients (with B = W^T), nn.Conv1d of the wav2vec2 feature encoder (A rows overlap: lda = stride*C_in,
 * K = k*C_in, z = batch) and the grouped positional conv (z = batch x group).
 * z = zb * nzg + zg; A/C offsets use both zb and zg strides, B/bias are offset per zg (and zb if b_zb != 0).
 * Constraints: K*sizeof(in) % 128 == 0, N % 4 == 0, strides multiples of 16 bytes.  in f32 => out f32.
 * in_dtype TAV_FP8 (BASELINE config 5): A and B are e4m3 bytes quantised with per-tensor scales (tav_fp8_amax / tav_fp8_quantize); the
 * products run on the block-scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales: twice the bf16 FLOP rate), accumulate in
 * f32, and the epilogue multiplies by *a_dequant * *b_dequant before the bias; gelu_in is bf16, outputs bf16 or f32.  The weight
 * gradient of a linear layer is the same call on the TRANSPOSED quantised copies (K = tokens, split over z = nzb with f32 slabs summed
 * by tav_splitk_reduce). */
typedef struct tav_gemm_nt_args {
    const void* A; const void* B; void* C;
    void* C_pre;            /* optional, dtype/layout of C                               */
    const float* bias;      /* optional [N] f32                                          */
    const void* gelu_in;    /* optional, dtype of A, layout of C (ld_gelu_in)            */
    const float* resid;     /* optional f32, layout of C (ld_resid)                      */
    int64_t M, N, K;
    int64_t lda, ldb, ldc, ld_pre, ld_gelu_in, ld_resid;
    int32_t nzb, nzg;       /* 0 is read as 1 */
    int64_t a_zb, a_zg, b_zb, b_zg, c_zb, c_zg, bias_zg;
    int32_t in_dtype, out_dtype;
    int32_t act;            /* bit 0: exact-erf GELU; bit 1: C_pre receives gelu'(pre-activation) instead of the pre-activation;
                               bit 2: gelu_in already holds that derivative (multiply, do not differentiate) */
    int32_t accumulate;
    float alpha;
    const float* a_dequant; /* in_dtype == TAV_FP8: device scalars (amax/448 of each operand, tav_fp8_amax) multiplied into alpha; NULL = 1 */
    const float* b_dequant;
    int64_t gelu_zb;        /* v5: zb batch stride of gelu_in in elements when it differs from c_zb (0 = c_zb): the side input of a strided / padded output */
    int32_t tile_m_hint;    /* 0 = let the library choose (it may cover the rows with two launches: whole rounds of 256 x 256 tiles, then 128-wide tiles over the rest); bits 0-4: 2/3/4 = force 64/96/128 x 128 workgroup tiles (4 waves), 8 = 256 x 128, 16 = 256 x 256 (8 waves, bf16 operands), 17 = library's choice of ONE tile for all rows; bits 5-7: LDS ring depth 2-4 (tuning / tests) */
} tav_gemm_nt_args;
int tav_gemm_nt(const tav_gemm_nt_args* args, void* stream);
/* Host-only: the tile plan tav_gemm_nt would use for these arguments (pointers are not read).  *tile as in tile_m_hint; the first
 * *rows_first rows take it; *tile_rest != 0 means a second launch with that tile covers rows [*rows_first, M). */
int tav_gemm_nt_schedule(const tav_gemm_nt_args* args, int32_t* tile, int32_t* rows_first, int32_t* tile_rest);

/* GEMM, "TN" (weight gradients): out[n1][perm(n2)] (+)= scale * sum_{z,t} A[z][t][n1] * B[z][t][n2].
 * The token axis is split into `nsplit` chunks that write f32 slabs [nsplit][N1][N2]; a second kernel sums the slabs
 * in a fixed order (bitwise reproducible; no atomics).  perm(n2) = (n2 % perm_inner)*perm_outer + n2/perm_inner maps
 * the [k][c_in] column order of the conv-as-GEMM back to nn.Conv1d's [c_in][k]; perm_outer <= 1 means identity.
 * Replaces the autograd weight-gradient of every nn.Linear / nn.Conv1d on the path (train_model/tav_train.py:60).
 * tav_gemm_tn_splits() (host-only) returns the chunking the library wants for a shape. */
typedef struct tav_gemm_tn_args {
    const void* A; const void* B;
    float* slabs;           /* workspace, nsplit*N1*N2 f32 */
    float* out;             /* [N1][N2] f32 */
    float* dbias;           /* optional [N1] f32: (+)= scale * sum_{z,t} A[z][t][n1] -- the bias gradient, fused (the dY tiles
                               are already in LDS); needs bias_partials = workspace nsplit*N1 f32 */
    float* bias_partials;
    int64_t N1, N2;
    int64_t lda, ldb;
    int64_t rows_per_batch, nbatch;
    int64_t a_zb, b_zb;
    int32_t chunk_rows, nsplit;
    int32_t perm_inner, perm_outer;
    int32_t dtype;
    int32_t accumulate;
    float scale;            /* 0 is read as 1 */
} tav_gemm_tn_args;
int tav_gemm_tn_splits(int64_t n1, int64_t n2, int64_t rows_per_batch, int64_t nbatch, int32_t* chunk_rows, int32_t* nsplit);
int tav_gemm_tn(const tav_gemm_tn_args* args, void* stream);

/* Up to 4 weight gradients over the SAME token axis (rows, contiguous, one batch) in one launch, e.g. the dWqkv / dWo / dW1 / dW2 of
 * one transformer layer (utils/TAVFormer.py:243-271 under autograd): out_k[n1][n2] = sum_t A_k[t][n1] * B_k[t][n2], and, when dbias
 * is given, dbias_k[n1] = sum_t A_k[t][n1].  Every tile sums over all rows (no split over tokens, no workspace); results overwrite. */
typedef struct tav_gemm_tn_problem {
    const void* A; const void* B;   /* [rows, N1] (row stride lda), [rows, N2] (row stride ldb) */
    float* out;                     /* [N1][N2] f32 */
    float* dbias;                   /* optional [N1] f32 */
    int64_t N1, N2, lda, ldb;
} tav_gemm_tn_problem;
int tav_gemm_tn_grouped(const tav_gemm_tn_problem* problems, int32_t nproblems, int64_t rows, int32_t dtype, void* stream);
/* The same with a caller-provided workspace, which lets the library (a) use its 256 x 256 tile with the token axis split over workgroups
 * (bf16, large problems: the four gradients of a 768-wide layer are only 108 such tiles) -- f32 slabs in the workspace, summed in a fixed
 * order by a second kernel -- and (b) share the bias-gradient column sums among the tiles of a tile row instead of loading them all onto
 * the first tile (partials in the workspace, summed by the same / a tiny second kernel).  Bitwise reproducible, no atomics.
 * tav_gemm_tn_grouped_ws_bytes() returns the size the plan for these problems needs; with a smaller (or NULL) workspace the call falls back
 * to tav_gemm_tn_grouped.  flags: 0 = library's choice; bit 0 forces the 256-wide form, bit 1 the 128-wide one, bits 8-11 an explicit
 * split count (tests / tuning). */
int64_t tav_gemm_tn_grouped_ws_bytes(const tav_gemm_tn_problem* problems, int32_t nproblems, int64_t rows, int32_t dtype, int32_t flags);
int tav_gemm_tn_grouped_ws(const tav_gemm_tn_problem* problems, int32_t nproblems, int64_t rows, int32_t dtype, void* workspace,
                           int64_t workspace_bytes, int32_t flags, void* stream);

/* FP8 operand preparation (sync free: the scale never leaves the device).  scales[0] = 448/amax (quantisation), scales[1] = amax/448
 * (the dequantisation factor for tav_gemm_nt), scales[2] = amax.  partials: tav_fp8_amax_partials(rows, cols) floats. */
int tav_fp8_amax_partials(int64_t rows, int64_t cols);
int tav_fp8_amax(const void* x, int32_t dtype, int64_t rows, int64_t cols, int64_t ld, float* partials, float* scales, void* stream);
/* q[r][c] = e4m3(x[r][c] * scales[0]) (row stride ld_q bytes) and/or the transposed copy qt[c][r] (row stride ld_qt >= rows_pad), whose
 * token axis is padded with zeros up to rows_pad -- the K axis of the weight-gradient GEMM.  x is f32 or bf16, cols % 4 == 0. */
int tav_fp8_quantize(const void* x, int32_t dtype, int64_t rows, int64_t cols, int64_t ld, const float* scales, void* q, int64_t ld_q, void* qt,
                     int64_t ld_qt, int64_t rows_pad, void* stream);
/* v5, DELAYED scaling: `state` = 4 device floats {448/amax, amax/448, amax, running amax}.  tav_fp8_quantize_delayed quantises with state[0]
 * -- the scale the previous step (or a calibration pass: tav_fp8_amax writes the first three) left there -- and gathers the tensor's own
 * absolute maximum into state[3] on the way: one pass and one launch per tensor instead of three.  tav_fp8_roll_states, once per step after
 * the last quantisation, turns every state[3] > 0 of the `n` consecutive states into the next step's scale and clears it.  state + 1 is the
 * dequantisation factor tav_gemm_nt takes (a_dequant / b_dequant); it does not change within a step. */
int tav_fp8_quantize_delayed(const void* x, int32_t dtype, int64_t rows, int64_t cols, int64_t ld, float* state, void* q, int64_t ld_q, void* qt,
                             int64_t ld_qt, int64_t rows_pad, void* stream);
int tav_fp8_roll_states(float* states, int64_t n, void* stream);
/* out[i] (+)= sum_s slabs[s][i], s < nsplit, i < n_elems (n_elems % 4 == 0): fixed order, bitwise reproducible */
int tav_splitk_reduce(const float* slabs, float* out, int32_t nsplit, int64_t n_elems, int32_t accumulate, void* stream);

/* Column sums (bias gradients): out[n] (+)= sum_m x[m][n]; partials = workspace nparts*N f32. */
int tav_colsum(const void* x, int32_t dtype, int64_t M, int64_t N, int64_t ld, float* partials, int32_t nparts, float* out,
               int32_t accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Fused multi-head self-attention, head_dim 64, flash-style (no S x S buffer).
 * q/k/v/o/do/dq/dk/dv are token-major with arbitrary row stride (e.g. slices of a fused [tokens, 3H] buffer):
 *   element (b, s, head, d) at ptr[(b*S + s)*ld + head*64 + d].
 * mask_mode 0: none                                  (HF videomae/wav2vec2 attention, no mask on the path)
 *           1: scores += key_mask[b][key] BEFORE softmax (HF roberta/bert additive mask; utils/TAVFormer.py:68-72)
 *           2: probs  += key_mask[b][key] AFTER  softmax (the fusion encoder's quirk, utils/TAVFormer.py:362-383):
 *              o = softmax(s) v + (sum_key mask[key] v[key]) for every query row; `corr` [B][nheads][64] f32 receives
 *              that rank-1 term (needed again by the backward).
 * scale is applied to q.k^T (1/8 on the path).  lse [B][nheads][S] f32 = log-sum-exp of the scaled (masked) scores.
 * Replaces utils/TAVFormer.py:357-383, :57-81 and HF eager_attention_forward. */
typedef struct tav_attn_args {
    const void* q; const void* k; const void* v;
    void* o;
    const float* key_mask;  /* [B][S] f32 or NULL */
    float* lse;             /* [B][nheads][S]     */
    float* corr;            /* [B][nheads][64], mask_mode 2 only */
    void* o_soft;           /* mask_mode 2 only: softmax(s) v WITHOUT the rank-1 term, layout/dtype of o; the backward
                               forms delta = dO . o_soft from it (o - corr would cancel catastrophically, |mask| ~ 6.5e4) */
    /* backward only */
    const void* dout; void* dq; void* dk; void* dv;
    float* delta;           /* workspace [B][nheads][S] f32 */
    int64_t B, S, nheads;
    int64_t ld_q, ld_k, ld_v, ld_o, ld_do, ld_dq, ld_dk, ld_dv;
    int32_t dtype, mask_mode;
    float scale;
    int32_t q_prescaled;    /* 1: q already holds q * scale * log2(e) (the caller folded the factor into the q rows of its QKV weight copy, so no
                               value is rounded twice): the kernels skip one multiply per score and start the running maximum inside the
                               MFMA accumulator.  Outputs are unchanged: dq / dk / dv are gradients w.r.t. the UNSCALED q, k, v. */
} tav_attn_args;
int tav_attn_fwd(const tav_attn_args* args, void* stream);
int tav_attn_bwd(const tav_attn_args* args, void* stream);
/* Slow-path helpers for VideoMAEEncoder.forward(head_mask=, output_attentions=True) (reference utils/TAVFormer.py:190, :368-370, :389; never
 * used by the reference's training loop).  tav_attn_probs materialises, from q, k and the lse tav_attn_fwd wrote,
 *   probs[b][h][i][j] = head_scale[b*hs_bstride + h] * softmax_j(scale q_i.k_j (+ key_mask[b][j], mask_mode 1)) (+ key_mask[b][j], mask_mode 2)
 * as f32 [B][nheads][S][S] (head_scale NULL = 1; hs_bstride 0: one factor per head, nheads: per batch entry and head).
 * tav_head_scale: out[r][h*64+d] = (a ? a[r][h*64+d] : 0) + (c0 + (head_scale ? head_scale[(r/S)*hs_bstride + h] : 0)) * b[r][h*64+d], r < B*S:
 * the head-masked context  o + (m_h - 1) o_soft  (c0 = -1, a = o, b = o_soft), its gradient factors, and plain sums (head_scale NULL, c0 = 1). */
int tav_attn_probs(const tav_attn_args* args, float* probs, const float* head_scale, int64_t hs_bstride, void* stream);
int tav_head_scale(const void* a, const void* b, void* out, int32_t dtype, const float* head_scale, int64_t hs_bstride, float c0, int64_t B, int64_t S,
                   int64_t nheads, int64_t lda, int64_t ldb, int64_t ldo, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * LayerNorm over the last axis (width W <= 1024, W % 4 == 0), one wave per row, f32 statistics.
 * fwd: y = (x - mean) * rstd * gamma + beta; writes y_f32 and/or y_lp (low-precision copy in `lp_dtype`, the next
 *      GEMM's operand) and mean/rstd [rows] f32.
 * bwd: dx = rstd*(dy*gamma - mean(dy*gamma) - xhat*mean(dy*gamma*xhat)) (+ dx_add); writes dx_f32 and/or dx_lp;
 *      dgamma/dbeta (+)= column sums (two deterministic stages through `partials`, >= tav_ln_bwd_partials(rows)*2*W f32).
 * Replaces nn.LayerNorm on the path (utils/TAVFormer.py:237,239,108,118; models/tav.py:439-447; HF LN sites). */
typedef struct tav_ln_args {
    const void* x; int32_t x_dtype;
    const float* gamma; const float* beta;
    float* y_f32; void* y_lp; int32_t lp_dtype;
    float* mean; float* rstd;
    /* backward */
    const void* dy; int32_t dy_dtype;
    const float* dx_add;    /* optional f32 [rows][W] added to dx */
    float* dx_f32; void* dx_lp;
    float* dgamma; float* dbeta; float* partials; int32_t accumulate_params;
    int64_t rows, W;
    int64_t ld_x, ld_y, ld_dy, ld_dx;
    float eps;
    int32_t act;            /* 1: y = gelu_erf(layernorm(x)) (wav2vec2 "layer" conv blocks, HF wav2vec2:275-301); bwd needs beta */
    int32_t defer_param_reduce; /* v5, bwd: 1 = write the per-workgroup column sums to `partials` (required; dgamma / dbeta ignored) and do NOT
                                   launch the second stage -- the caller reduces them later with tav_ln_param_reduce_multi */
} tav_ln_args;
int tav_ln_fwd(const tav_ln_args* args, void* stream);
int tav_ln_bwd(const tav_ln_args* args, void* stream);
int tav_ln_bwd_partials(int64_t rows);
/* v5.  Second stage of up to TAV_LN_REDUCE_MAX deferred LayerNorm backwards in ONE launch (a training step has ~210 of them; nobody reads
 * dgamma / dbeta before the optimizer): item i sums the `nblocks` = tav_ln_bwd_partials(rows) partial rows its tav_ln_bwd wrote, in the same
 * fixed order as the immediate form (bitwise equal results).  `items` is a HOST array (copied into the kernel arguments). */
#define TAV_LN_REDUCE_MAX 64
typedef struct tav_ln_reduce_item {
    const float* partials; float* dgamma; float* dbeta;
    int32_t nblocks, W, accumulate, pad_;
} tav_ln_reduce_item;
int tav_ln_param_reduce_multi(const tav_ln_reduce_item* items, int32_t n, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Element-wise / data-movement kernels (all HBM-bound). */

/* dst[r][c] = cast(src[r][c]) (row stride ld_dst, 0 = C) and/or dst_t[c][r] = cast(src[r][c]) (row stride ld_dst_t,
 * 0 = R): the per-step operand copies of an f32 parameter [R][C]; the transposed copy is the dgrad operand.  Strides let
 * q/k/v weights land in one fused [3H][K] / [K][3H] operand. */
int tav_cast_weight(const float* src, int64_t R, int64_t C, void* dst, int64_t ld_dst, void* dst_t, int64_t ld_dst_t, int32_t dst_dtype,
                    void* stream);
/* Multi-tensor form of tav_cast_weight: `descs` is a DEVICE array of n descriptors
 *   { const float* src; void* dst; void* dst_t; int64 ld_dst, ld_dst_t; int32 R, C, dst_dtype; float scale_n }   (48 bytes each)
 * (scale_n multiplies what is written to dst, NOT dst_t: the attention pre-scale of the q rows, tav_attn_args.q_prescaled)
 * handled by one launch (grid = blocks_per_tensor x n) -- all operand copies of a transformer layer at once. */
int tav_cast_weights_multi(const void* descs, int32_t n, int32_t blocks_per_tensor, void* stream);
/* Conv1d weight [co][ci][k] f32 -> GEMM operand [co][k][ci] (dst), its column-buffer dgrad form [k*ci][co] (dst_t) and (v5) the
 * per-output-phase dgrad operands (dst_phase, conv_stride): for r = 0 .. conv_stride-1 the block ph_r[ci][q'*co + c] =
 * W[c][ci][r + (Q_r-1-q')*conv_stride], Q_r = ceil((k-r)/conv_stride), blocks back to back (co*ci*k elements in all).  With them
 * dx[s*m + r] is ONE NT GEMM per phase over Q_r consecutive dy rows (overlapping rows, lda = co) -- see engine.ConvGemmFn. */
int tav_cast_conv_weight(const float* src, int64_t co, int64_t ci, int64_t k, void* dst, void* dst_t, void* dst_phase, int64_t conv_stride,
                         int32_t dst_dtype, void* stream);
/* v5: zero rows [0, pad) and [pad + T, T + 2*pad) of each of the B batch entries of a [B][T + 2*pad][C] buffer */
int tav_zero_pad_rows(void* buf, int32_t dtype, int64_t B, int64_t T, int64_t C, int64_t pad, void* stream);
/* generic strided cast/copy: dst[r][c] = src[r][c] for r<R, c<C (dtypes may differ) */
int tav_cast2d(const void* src, int32_t src_dtype, int64_t ld_src, void* dst, int32_t dst_dtype, int64_t ld_dst, int64_t R, int64_t C,
               void* stream);
/* dst[z][c][r] = src[z][r][c] for z < nbatch.  The reference's TransformerEncoder "concat" (utils/TAVFormer.py:84) does
 * scores[B*h,S,d].transpose(1,2).contiguous().view(B,S,h*d): per batch that is exactly the [S,H] -> [H,S] transpose of the
 * token-major attention output, re-read as [S,H]. */
int tav_transpose2d(const void* src, void* dst, int32_t dtype, int64_t R, int64_t C, int64_t nbatch, void* stream);
/* y_f32 = a + b (f32) and optional low-precision copy */
int tav_add_f32(const float* a, const float* b, float* y, void* y_lp, int32_t lp_dtype, int64_t n, void* stream);
/* fill f32 */
int tav_fill_f32(float* p, float value, int64_t n, void* stream);

/* out[b][s][:] = x[b][s][:] + table[ids[b][s]][:]   (models/tav.py:474, nn.Embedding(3,768) add); x,out f32 */
int tav_embed_add_fwd(const float* x, const int64_t* ids, const float* table, float* out, int64_t rows, int64_t W, int64_t ntable,
                      void* stream);
/* dtable[t][:] (+)= sum_{rows with ids==t} dy[row][:]  (ntable <= 8; deterministic) */
int tav_embed_add_bwd(const float* dy, const int64_t* ids, float* dtable, float* partials, int64_t rows, int64_t W, int64_t ntable,
                      int32_t accumulate, void* stream);
int tav_embed_add_bwd_parts(int64_t rows);   /* partials = parts * ntable * W f32 */

/* BERT/RoBERTa embeddings (HF roberta/modeling_roberta.py:75-121): e = word[ids] + pos[pos_ids] + type[0], then LayerNorm.
 * pos_ids computed on device: roberta: cumsum(ids != pad)*(ids != pad) + pad ; bert (pad_id < 0): arange(S).
 * Outputs LN result as f32 and/or lp copy; saves mean/rstd and the pre-LN sum (pre, f32) for the backward. */
typedef struct tav_text_embed_args {
    const int64_t* ids;     /* [B][S] */
    const float* word; const float* pos; const float* type;   /* tables f32 */
    const float* gamma; const float* beta;
    float* pre;             /* [B*S][W] f32 pre-LayerNorm sum   */
    int64_t* pos_ids;       /* [B][S] out                       */
    float* y_f32; void* y_lp; int32_t lp_dtype;
    float* mean; float* rstd;
    int64_t B, S, W, vocab, max_pos;
    int32_t pad_id;         /* >= 0: RoBERTa position ids; < 0: BERT */
    float eps;
} tav_text_embed_args;
int tav_text_embed_fwd(const tav_text_embed_args* args, void* stream);
/* scatter-add of row gradients into a table: dtable[idx[r]][:] += d[r][:] (table zeroed by caller).  No atomics: the first row
 * of each distinct index sums all rows carrying it in ascending row order, so the result is bitwise reproducible.
 * W % 4 == 0, W <= 1024, rows*4 + 64*W bytes of LDS <= 150 KB (rows <= ~22 k at W = 768); indices outside [0, ntable) are skipped. */
int tav_scatter_add_rows(const float* d, const int64_t* idx, float* dtable, int64_t rows, int64_t W, int64_t ntable, void* stream);

/* VideoMAE tubelet patch gather (HF videomae/modeling_videomae.py:157-177 Conv3d k=s=(2,16,16) as a GEMM A operand):
 * video f32 [B][F][3][H][W] (frames-first, models/tav.py:243) -> patches [B*nkeep][3*2*16*16] in `dtype`,
 * only for the token indices keep_idx[B][nkeep] (int32, ascending = the rows `embeddings[~mask]` keeps). */
int tav_patchify(const float* video, const int32_t* keep_idx, void* patches, int32_t dtype, int64_t B, int64_t F, int64_t H, int64_t W,
                 int64_t nkeep, void* stream);
/* per row of mask [B][n] (uint8/bool; keep where mask == keep_value): write ascending indices [B][nkeep];
 * counts[B] receives the number found (host checks == nkeep when it wants to).  A row with fewer than nkeep kept tokens gets its
 * last kept index (0 if none) in the remaining slots and a row with more is truncated: keep_idx is always fully written with
 * values in [0, n), so tav_patchify / tav_gather_rows (which also clamp) never read outside their operands. */
int tav_mask_to_index(const uint8_t* mask, int32_t keep_value, int32_t* keep_idx, int32_t* counts, int64_t B, int64_t n, int64_t nkeep,
                      void* stream);
/* out[r][:] = table[idx[r]][:] f32 -- rows of the fixed sin-cos position table (HF videomae:80-124) for the kept tokens;
 * the result is handed to the patch-embedding GEMM as its `resid` so the add is fused.  idx is clamped into [0, ntable). */
int tav_gather_rows(const float* table, const int32_t* idx, float* out, int64_t rows, int64_t W, int64_t ntable, void* stream);

/* mean over tokens: y[b][:] = mean_s x[b][s][:] (models/tav.py:478,481,488) and its backward dx[b][s][:] = dy[b][:]/S */
int tav_mean_pool_fwd(const float* x, float* y, int64_t B, int64_t S, int64_t W, void* stream);
int tav_mean_pool_bwd(const float* dy, float* dx, void* dx_lp, int32_t lp_dtype, int64_t B, int64_t S, int64_t W, void* stream);

/* small dense head (N <= 16, e.g. Linear(3072,7), models/tav.py:499): y = x W^T + b ; all f32.
 * bwd: dx = dy W ; dW (+)= dy^T x ; db (+)= colsum(dy) */
int tav_head_fwd(const float* x, const float* W, const float* b, float* y, int64_t B, int64_t K, int64_t N, void* stream);
int tav_head_bwd(const float* x, const float* W, const float* dy, float* dx, float* dW, float* db, int64_t B, int64_t K, int64_t N,
                 int32_t accumulate, void* stream);
/* tanh pooler epilogue (HF roberta:530-536): y = tanh(x) and dx = dy*(1-y^2) */
int tav_tanh_fwd(const float* x, float* y, int64_t n, void* stream);
int tav_tanh_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream);

/* (weighted) cross entropy, mean reduction (utils/global_functions.py:63-64,76,83; torch.nn.CrossEntropyLoss):
 * loss = sum_i w[t_i] * (lse_i - z_i[t_i]) / sum_i w[t_i];  dlogits written in the same pass (scaled by grad_scale). */
int tav_cross_entropy(const float* logits, const int64_t* target, const float* class_weight /*opt*/, float* loss, float* dlogits,
                      int64_t B, int64_t C, float grad_scale, void* stream);

/* dropout with a counter-based RNG (models/tav.py:497-498): y = x * keep / (1-p); mask bytes saved for backward */
int tav_dropout_fwd(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream);
int tav_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, int64_t n, float p, void* stream);

/* wav2vec2 feature encoder pieces (HF wav2vec2/modeling_wav2vec2.py:275-323,382-419) -- activations are kept
 * channels-last [B][T][C] so every later conv is a GEMM over overlapping rows.
 * conv0: Conv1d(1, C, k=10, s=5) direct: y[b][t][c] = sum_j x[b][5t+j] w[c][j] (+ bias[c]) */
int tav_conv0_fwd(const float* wave, const float* w, const float* bias, void* y, int32_t y_dtype, int64_t B, int64_t T_in, int64_t T_out,
                  int64_t C, int64_t K, int64_t stride, void* stream);
int tav_conv0_bwd_w(const float* wave, const void* dy, int32_t dy_dtype, float* dw, float* dbias, float* partials, int64_t B, int64_t T_in,
                    int64_t T_out, int64_t C, int64_t K, int64_t stride, int32_t accumulate, void* stream);
int tav_conv0_bwd_partials(int64_t B, int64_t T_out, int64_t C, int64_t K);   /* floats */
/* GroupNorm(num_groups == channels) over time, then GELU: y = gelu((x - mean_bc) * rstd_bc * gamma_c + beta_c).
 * stats [B][C][2] f32 (mean, rstd).  bwd returns dx and dgamma/dbeta. */
int tav_gn_workspace_floats(int64_t B, int64_t C);
int tav_gn_gelu_fwd(const void* x, void* y, int32_t dtype, const float* gamma, const float* beta, float* stats, float* workspace, int64_t B,
                    int64_t T, int64_t C, float eps, void* stream);
int tav_gn_gelu_bwd(const void* x, const void* dy, void* dx, int32_t dtype, const float* gamma, const float* beta, const float* stats,
                    float* workspace, float* dgamma, float* dbeta, int64_t B, int64_t T, int64_t C, int32_t accumulate, void* stream);
/* y = gelu(x) and dx = dy * gelu'(x) for tensors in `dtype` */
int tav_gelu_fwd(const void* x, void* y, int32_t dtype, int64_t n, void* stream);
int tav_gelu_bwd(const void* x, const void* dy, void* dx, int32_t dtype, int64_t n, void* stream);
/* col2im for the strided convs' input gradient: dcol [B][T_out][k*C] (the dgrad GEMM's output) -> dx [B][T_in][C],
 * dx[b][tau][:] = sum_{j : (tau-j) % s == 0, 0 <= (tau-j)/s < T_out} dcol[b][(tau-j)/s][j*C + :] ; optional fused
 * multiply by gelu'(pre[b][tau][:]) of the producing layer. */
int tav_col2im_1d(const void* dcol, void* dx, const void* pre_act, int32_t dtype, int64_t B, int64_t T_in, int64_t T_out, int64_t C,
                  int64_t K, int64_t stride, void* stream);
/* grouped positional conv support (HF wav2vec2:326-379): zero-padded, group-major copy
 * xg[b][g][pad_l + t][cg] = x[b][t][g*Cg + cg]; rows outside [0,T) are zero.  Inverse (add) for the gradient. */
int tav_group_pad(const void* x, int32_t x_dtype, void* xg, int32_t xg_dtype, int64_t B, int64_t T, int64_t H, int64_t G, int64_t pad_l,
                  int64_t pad_r, void* stream);
/* weight-norm reparametrisation w = g * v / ||v||  with the norm over (co, ci) per tap k (weight_norm dim=2):
 * v [H][Cg][K] f32, g [K] f32 -> w (GEMM layout [G][Cg_out][K][Cg_in] in `dtype`, and its flipped dgrad form);
 * norms [K] f32 saved.  bwd: dv, dg from dw (dw in the GEMM layout, f32). */
int tav_weight_norm_partials(int64_t H, int64_t Cg);   /* fwd partials = this * K floats */
int tav_weight_norm_fwd(const float* v, const float* g, float* norms, float* partials, void* w, void* w_flip, int32_t dtype, int64_t H,
                        int64_t Cg, int64_t K, void* stream);
/* workspace floats: H*Cg*K + tav_weight_norm_partials(H,Cg)*K + K */
int tav_weight_norm_bwd(const float* v, const float* g, const float* norms, const float* dw_gemm, float* workspace, float* dv, float* dg,
                        int64_t H, int64_t Cg, int64_t K, int32_t accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Optimiser-side kernels (train_model/tav_train.py:61-62,148): multi-tensor via device pointer tables.
 * ptrs/sizes are device arrays of `ntensors` entries. */
int tav_sumsq_partials(int32_t ntensors);   /* floats of workspace */
int tav_sumsq_multi(const float* const* ptrs, const int64_t* sizes, int32_t ntensors, float* partials, float* out_sumsq, void* stream);
/* p -= lr*(m_hat/(sqrt(v_hat)+eps)) after p *= (1-lr*wd); grads scaled by *clip_coef_ptr (device scalar, computed by
 * tav_clip_coef from the global norm) so no host sync is needed between norm and step. */
int tav_clip_coef(const float* sumsq, float max_norm, float* coef_out, float* norm_out, void* stream);
/* lr (f32 scalar), step (int32 counter, incremented by the call) and bias_corr (2 f32 of scratch) live on the DEVICE so that a
 * captured hipGraph replays with the right learning rate and bias correction (kernel arguments are frozen under replay). */
int tav_adamw_multi(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, const int64_t* sizes,
                    int32_t ntensors, const float* clip_coef /*opt*/, const float* lr, float beta1, float beta2, float eps, float weight_decay,
                    int32_t* step, float* bias_corr, void* stream);
/* Chunked forms of the two calls above for long parameter lists with very uneven sizes: chunk_prefix[t] (device, int32, ntensors entries)
 * = index of tensor t's first chunk of tav_optim_chunk_elems() elements, nchunks = total; block c works on chunk c only.
 * partials = workspace nchunks f32.  Same arithmetic as tav_sumsq_multi / tav_adamw_multi. */
int tav_optim_chunk_elems(void);
int tav_sumsq_chunked(const float* const* ptrs, const int64_t* sizes, const int32_t* chunk_prefix, int32_t ntensors, int32_t nchunks,
                      float* partials, float* out_sumsq, void* stream);
/* v6: the second stage of tav_sumsq_chunked alone (out = sum of n partials, fixed order): for an optimizer SHARDED over data-parallel ranks, each
 * of which computes the partials of its own chunks; the exchanged array summed here gives every rank the replicated optimizer's norm, bit for bit. */
int tav_sum_partials(const float* partials, int64_t n, float* out, void* stream);
int tav_adamw_chunked(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, const int64_t* sizes,
                      const int32_t* chunk_prefix, int32_t ntensors, int32_t nchunks, const float* clip_coef, const float* lr, float beta1,
                      float beta2, float eps, float weight_decay, int32_t* step, float* bias_corr, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Data-parallel exchange: the ONE collective of the path, the mean of a gradient bucket over the ranks of a node
 * (RCCL over xGMI), for hosts that do not come with a collective library of their own.  One process per GPU; every rank calls
 * tav_allreduce_bucket for the same buckets in the same order.  `comm` is an RCCL communicator created with tav_comm_init_rank (rank 0
 * makes the 128-byte id with tav_comm_unique_id and ships it to the others by its own means).  The all-reduce is enqueued on `stream`
 * (in place, mean, f32 or bf16 elements); returns 1000 + ncclResult_t when RCCL refuses.  The reference has no distributed code; this
 * replaces what torch DistributedDataParallel would do around train_model/tav_train.py:59-62.  The Python host layer (ddp.py) issues the
 * same collective through torch.distributed's "nccl" backend, which IS RCCL on ROCm. */
int tav_comm_rccl_version(int32_t* version);   /* v5: ncclGetVersion() of the RCCL that was loaded (resolved lazily with dlopen: libtavhip has no link-time dependency on it) */
int tav_comm_unique_id(void* out128);
int tav_comm_init_rank(void** comm, int32_t nranks, const void* unique_id128, int32_t rank);
int tav_comm_destroy(void* comm);
int tav_allreduce_bucket(void* buf, int64_t nbytes, int32_t dtype, void* comm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TAVHIP_H */
