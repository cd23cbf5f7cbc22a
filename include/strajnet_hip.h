/* strajnet_hip.h -- C ABI of libstrajnet_hip.so (MI355X / gfx950 kernels of the STrajNet hot path).
 *
 * The reference (georgeliu233/STrajNet) has no FFI / plugin interface: its hot path is TensorFlow ops issued from
 * Python (modules.py, FG_MSA.py, trajNet.py, loss.py).  This header is therefore the boundary a maintainer would bind
 * (ctypes stub in INTEGRATION.md); each entry point names the reference op sequence it replaces (file:line relative
 * to the reference repo).
 *
 * Conventions
 *   - all pointers are DEVICE pointers owned by the caller (no allocation, no synchronisation inside; every call is
 *     one or a few stream-ordered launches on `stream` => hipGraph-capturable); 16-byte alignment where noted.
 *   - dtype: STJ_F32 = 0 (exact-f32 MFMA, parity mode), STJ_BF16 = 1 (bf16 storage, f32 accumulate: the training throughput
 *     mode) or STJ_F16 = 2 (fp16 storage, f32 accumulate: the inference mode of BASELINE config 4; every entry point accepts it,
 *     the backward ones without any loss scaling and through the generic conv kernels) selects the activation type `T`; parameters, biases, LN gamma/beta, tables and all gradients of parameters are f32.
 *   - return 0 on success, negative stj_status otherwise; message via stj_last_error() (thread local).
 *   - tensors are NHWC / row-major exactly as in the reference.
 *   - "+=" outputs are ACCUMULATED with f32 atomics (they point into the flat gradient buffer).
 */
#ifndef STRAJNET_HIP_H
#define STRAJNET_HIP_H
#include <hip/hip_runtime_api.h>
#ifdef __cplusplus
extern "C" {
#endif

enum stj_status { STJ_OK = 0, STJ_EINVAL = -1, STJ_ELAUNCH = -2, STJ_EUNSUPPORTED = -3 };
enum stj_dtype { STJ_F32 = 0, STJ_BF16 = 1, STJ_F16 = 2 };
enum stj_act { STJ_ACT_NONE = 0, STJ_ACT_GELU = 1, STJ_ACT_ELU = 2 };
enum stj_unary { STJ_U_GELU = 1, STJ_U_ELU = 2, STJ_U_TANH_SCALE = 3 };

const char* stj_last_error(void);
int stj_abi_version(void);

/* Batched strided GEMM with fused epilogue:  C[z] = act(alpha * A[z] B[z] + bias[z]) + res[z],  z = z1*nb2 + z2.
 * Replaces Keras Dense (modules.py:36-37,76-79,270-271), 1x1 Conv2D (FG_MSA.py:54-64), tfa MultiHeadAttention
 * projections and q k^T / attn v einsums (trajNet.py:33,42,71,80,195,225; FG_MSA.py:147,176), the time-collapsed
 * Conv3D(8,1,1) skips (modules.py:693-717,750-765) and all their dgrad/wgrad (tape.gradient, train.py:223).
 * Element strides: A(m,k) at sAm*m + sAk*k (+ batch), B(k,n) at sBk*k + sBn*n; C row-major with ldc.
 * c_f32: C is f32; accumulate: C += (f32 atomics; splitk may be >1, 0 = auto); colsum (optional): colsum[z][n] +=
 * sum_k B[z](k,n) (the bias gradient belonging to dW = x^T dY), addressed with the bias batch strides.
 * nkb >= 1 K segments: the contraction also runs over nkb segments of A / B that lie sAkb / sBkb elements apart (the input
 * gradient of a layer applied with 8 per-waypoint weight sets to ONE shared input is a single GEMM with K' = 8 K). */
int stj_gemm(const void* A, const void* B, void* C, const float* bias, const void* res, float* colsum,
             int M, int N, int K, int nb1, int nb2,
             long long sAb1, long long sAb2, long long sAm, long long sAk,
             long long sBb1, long long sBb2, long long sBk, long long sBn,
             long long sCb1, long long sCb2, long long ldc,
             long long sBias1, long long sBias2, long long sRes1, long long sRes2, long long ldres,
             int act, float alpha, int dtype, int c_f32, int accumulate, int splitk,
             int nkb, long long sAkb, long long sBkb, void* group, hipStream_t stream);
/* Grouped launch.  `group` (last pointer argument of stj_gemm; NULL = launch at once) is CALLER-OWNED HOST memory of
 * stj_gemm_group_workspace_bytes() bytes, initialised by stj_gemm_group_begin: stj_gemm calls handed the group are RECORDED in it
 * (arguments validated) and launched by stj_gemm_group_end(group, stream) as ONE kernel per 4 problems of equal dtype (64x64 or 32x32
 * tiles): the input and weight gradient of a Dense layer (tape.gradient of modules.py:36-37 etc.), the q / k / v projections of a
 * tfa MultiHeadAttention (trajNet.py:33,71,195), dP / dV and dQ / dK of an attention.  The problems of a group must not depend on
 * each other.  The library keeps no state between calls: two host threads use two groups. */
long long stj_gemm_group_workspace_bytes(void);
int stj_gemm_group_begin(void* group);
int stj_gemm_group_end(void* group, hipStream_t stream);
/* Grouped stream-K weight gradients (csrc/wgrad_sk.hip): for every job  dw[z] += x[z]^T dy[z]  and, if db != NULL,
 * db[z] += column sums of dy[z]  (z = z1*nb2 + z2 < nb1*nb2) -- tape.gradient (train.py:223) w.r.t. the kernel / bias of the Keras
 * Dense layers, 1x1 convs and tfa-MHA projections (modules.py:36-37,76-83,270-272; trajNet.py:71-77,195-211; FG_MSA.py:54-64) --
 * ALL jobs of the list in ONE launch (per 28 jobs) on `wg_budget` workgroups (<= 0: one per CU).  x [rows, cin] with row stride ldx,
 * dy [rows, cout] with row stride lddy (activation dtype, 16-byte aligned), dw f32 [cin, cout] with row stride lddw, db f32 [cout];
 * s*1 / s*2 are the element strides of the two batch levels.  A workgroup owns a full-width 96 x 384 tile of a job, streams the rows
 * once through an LDS-DMA ring, and the launch's workgroups share the 32-row slabs of all jobs evenly (stream-K): f32 atomics only
 * where a workgroup leaves a tile.  Jobs must satisfy stj_wgrad_job_supported (16-bit dtype, rows % 32 == 0, cin, cout and all
 * strides of x / dy multiples of 8); anything else is stj_gemm's (accumulate = 1). */
typedef struct stj_wgrad_job {
  const void* x; const void* dy; float* dw; float* db;
  int rows, cin, cout, nb1, nb2;
  long long ldx, lddy, lddw;
  long long sx1, sx2, sdy1, sdy2, sdw1, sdw2, sdb1, sdb2;
} stj_wgrad_job;
int stj_wgrad_job_supported(const stj_wgrad_job* job, int dtype);
int stj_wgrad_group(const stj_wgrad_job* jobs, int njobs, int dtype, int wg_budget, hipStream_t stream);
/* out[n] += sum_m X[m,n]  (bias gradients of the conv heads). */
int stj_colsum(const void* X, float* out, int M, int N, long long ld, int dtype, hipStream_t stream);
/* f32 <-> bf16 / fp16 copy (16-bit compute copy of the flat parameter buffer). */
int stj_cast(const void* src, int sdtype, void* dst, int ddtype, long long n, hipStream_t stream);

/* Gelu (tanh form, modules.py:18-29 / FG_MSA.py:7-18), ELU (Keras activation='elu'), tanh*scale (FG_MSA.py:116-117).
 * bwd: saved = x for GELU, y for ELU / tanh. */
int stj_unary_fwd(const void* x, void* y, long long n, int op, float p0, int dtype, hipStream_t stream);
int stj_unary_bwd(const void* dy, const void* saved, void* dx, long long n, int op, float p0, int dtype, hipStream_t stream);
/* GlobalMaxPooling1D over the 11 time steps (trajNet.py:34,44); backward splits ties evenly like tf.reduce_max. */
int stj_maxpool_fwd(const void* x, void* y, int* idx, long long outer, int Tn, int C, int dtype, hipStream_t stream);
int stj_maxpool_bwd(const void* dy, const void* x, const void* y, void* dx, long long outer, int Tn, int C, int dtype, hipStream_t stream);

/* Keras LayerNormalization (modules.py:179,184,272,433,517,557 eps 1e-5; FG_MSA.py:52, trajNet.py:72-73,110-111,
 * 206-207 eps 1e-3).  gather_res != 0 fuses the PatchMerging 2x2 gather-concat (modules.py:282-287): x is
 * [B,res,res,C0], C = 4*C0.  group_rows/ngroups/gstride: row runs of group_rows rows use parameter set
 * (run % ngroups) at gamma + g*gstride (the 8 per-waypoint LayerNorms, trajNet.py:257). */
int stj_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      long long rows, int C, float eps, int gather_res, int C0, long long group_rows, int ngroups,
                      long long gstride, int dtype, hipStream_t stream);
/* y = LayerNorm(x) + res (res [rows,C], type T; no gather): the norm followed by the sum with another branch in one pass
 * (vec + maps ahead of all_patch_norm, modules.py:589; Cross_AttentionT output + query, trajNet.py:305-317).  Backward:
 * stj_layernorm_bwd for x; the gradient of res is dy itself. */
int stj_layernorm_res_fwd(const void* x, const float* gamma, const float* beta, const void* res, void* y, float* mean,
                          float* rstd, long long rows, int C, float eps, long long group_rows, int ngroups,
                          long long gstride, int dtype, hipStream_t stream);
int stj_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                      void* dx, float* dgamma, float* dbeta, long long rows, int C, int gather_res, int C0,
                      long long group_rows, int ngroups, long long gstride, const void* dres, int nparts, long long part_stride,
                      int dtype, hipStream_t stream);
/* Two LayerNorm backward passes in one launch, for y = LN2(LN1(x1) [+ add]) (the stem: modules.py:437-446 with :578 / :590): d2 = dLN2(dy)
 * at x2 (written to d2 when non-NULL: the gradient of `add`), dx1 = dLN1(d2) at x1; d2 is rounded to T in between, as two stj_layernorm_bwd
 * calls hand it over.  Plain [rows, C] tensors, one parameter set per norm, nparts / part_stride as in stj_layernorm_bwd (per norm). */
int stj_layernorm_bwd_chain_supported(int C, int dtype);
int stj_layernorm_bwd_chain(const void* dy, const void* x2, const float* gamma2, const float* mean2, const float* rstd2, const void* x1,
                            const float* gamma1, const float* mean1, const float* rstd1, void* d2, void* dx1, float* dgamma2, float* dbeta2,
                            float* dgamma1, float* dbeta1, long long rows, int C, int nparts2, long long part_stride2, int nparts1,
                            long long part_stride1, int dtype, hipStream_t stream);
/* dres (optional, x's layout, no gather): gradient arriving over the residual connection that bypasses the norm; dx += dres.
 * nparts / part_stride: dgamma and dbeta are "+=" into nparts copies that lie part_stride floats apart (workgroups rotate over
 * them: 256 same-address atomics per channel otherwise); the caller sums the copies.  nparts = 1: plain [C] buffers. */

/* Fused (shifted-)window attention, window 8x8, head_dim 32: roll + window_partition + softmax(q k^T*scale +
 * relative_position_bias[+shift mask]) v + window_reverse + roll (modules.py:49-63,103-134,189-216,229-255).
 * qkv [B,res*res,3*heads*32] in original token order, table f32 [225,heads], out [B,res*res,heads*32]. */
int stj_win_attn_fwd(const void* qkv, const float* table, void* out, int B, int res, int heads, int shift,
                     int dtype, hipStream_t stream);
int stj_win_attn_bwd(const void* qkv, const float* table, const void* dout, void* dqkv, float* dtable, int nparts,
                     int B, int res, int heads, int shift, int dtype, hipStream_t stream);
/* bwd: dtable f32 [nparts][225,heads], "+=": workgroup i adds the bias-table gradient of its window into copy i % nparts and the
 * caller sums the copies (one copy = hundreds of same-address atomics per table entry on the 64x64 stage; nparts = 1 is valid). */

/* Fused MLP half of SwinTransformerBlock (modules.py:260 = x + drop_path(mlp(norm2(x))); Mlp.call :40-46; Gelu :18-29; drop_path
 * :137-151), ONE kernel per direction (csrc/swin_fused.hip).  x, y [M,C] (C in {96,192,384}); gamma/beta f32 [C] (LN eps);
 * w1 [C,4C], w2 [4C,C] activation dtype (Keras [in,out]); b1 [4C], b2 [C] f32.
 * DropPath: rng_state = device int64[2] {seed, step} (NULL or p_drop = 0: none), one draw per sample = run of rows_per_sample rows,
 * drawn exactly as stj_dropout(inner = rows_per_sample * C) draws it.
 *   fwd: y = x + dp * (gelu(LN(x) w1 + b1) w2 + b2)
 *   bwd: dx = dy + LN'(...), dgamma/dbeta "+=" (nparts copies part_stride floats apart, as stj_layernorm_bwd), and the operands
 *        of the two weight gradients written once: h = gelu(pre) [M,4C], dpre [M,4C], ln = LN(x) [M,C], dys = dp*dy [M,C]
 *        (dys may be NULL when there is no DropPath: use dy).  dW1 = ln^T dpre, db1 = colsum(dpre), dW2 = h^T dys,
 *        db2 = colsum(dys) are stj_gemm split-K launches.
 * ws: workspace of stj_swin_split_workspace_bytes(M, C) bytes (0 = this (M, C) takes none: pass NULL), ZERO-INITIALISED ONCE by the caller
 *   and then left to the kernels: where a stage has fewer row blocks / windows than the chip has CUs, a unit is cut into slices of the hidden
 *   dimension (MLP half) or of the heads (attention half) that run as separate workgroups:
 *     C = 384, many slices (the 2048-row stage: (row block, 1/8 of the hidden dimension) workgroups, 256 at B = 8 instead of 32, each
 *       streaming 1/8 of the weights): partial sums [slices][M][C] f32 that a second launch adds up and finishes (bias + DropPath +
 *       shortcut; LayerNorm backward + dgamma / dbeta);
 *     two slices (C = 192 below 32768 rows; C = 384 at 8192 rows): the slices meet INSIDE the launch -- each writes its accumulators as a
 *       slab (write-through stores) and draws a ticket from the unit's arrival counter; the workgroup that draws the last one adds the
 *       other slab and runs the epilogue.  The counters live behind the partial-sum region and re-arm themselves (hence "zeroed once").
 *   REQUIRED at C = 384 (STJ_EINVAL without it); at C = 192 NULL selects one workgroup per unit.  The four stj_swin_* entry points share
 *   the workspace layout (launches that share a workspace must be ordered on one stream); stj_swin_attn_* at C = 384 also need a 16-bit
 *   dtype: (window, 2 of the 12 heads) workgroups. */
long long stj_swin_split_workspace_bytes(long long M, int C);
int stj_swin_mlp_fwd(const void* x, const float* gamma, const float* beta, const void* w1, const float* b1, const void* w2,
                     const float* b2, void* y, long long M, int C, float eps, const long long* rng_state, int site,
                     float p_drop, long long rows_per_sample, int dtype, void* ws, hipStream_t stream);
int stj_swin_mlp_bwd(const void* x, const void* dy, const float* gamma, const float* beta, const void* w1, const float* b1,
                     const void* w2, void* dx, void* h, void* dpre, void* ln, void* dys, float* dgamma, float* dbeta,
                     int nparts, long long part_stride, long long M, int C, float eps, const long long* rng_state, int site,
                     float p_drop, long long rows_per_sample, int dtype, void* ws, hipStream_t stream);

/* Fused attention half of SwinTransformerBlock, forward (modules.py:225-258: norm1, roll, window_partition, WindowAttention :103-134
 * with the shift mask :189-216, window_reverse, roll, drop_path + shortcut), one workgroup per 8x8 window (csrc/swin_fused.hip):
 *   y = x + dp * (proj(window_attention(LN(x) wqkv + bqkv)) + bproj);  x, y [B,res*res,C], C in {96,192,384}, heads = C/32;
 *   wqkv [C,3C], wproj [C,C] activation dtype (Keras [in,out]); bqkv [3C], bproj [C], gamma/beta [C], table [225,heads] f32.
 * Training hand-offs (all or none; NULL for inference): qkv [B,N,3C], a = attention output before proj [B,N,C], ln = LN(x) [B,N,C],
 * mean / rstd f32 [B*N] -- the operands stj_win_attn_bwd, stj_layernorm_bwd and the dgrad / wgrad stj_gemm launches of backward read.
 * DropPath: one draw per sample b at (rng_state, site), as stj_dropout(inner = N*C) draws it. */
int stj_swin_attn_fwd(const void* x, const float* gamma, const float* beta, const void* wqkv, const float* bqkv,
                      const float* table, const void* wproj, const float* bproj, void* y, void* qkv, void* a, void* ln,
                      float* mean, float* rstd, int B, int res, int C, int shift, float eps, const long long* rng_state,
                      int site, float p_drop, int dtype, void* ws, hipStream_t stream);

/* Backward of stj_swin_attn_fwd in one launch (one workgroup per window): dys = dp*dy, da = dys wproj^T, window-attention backward,
 * dLN = dqkv wqkv^T, dx = dy + LayerNorm'(dLN); "+=" outputs: dtable [tparts][225,heads] (workgroup i adds into copy i % tparts),
 * dgamma / dbeta (nparts copies part_stride floats apart).  Reads x, dy, the saved qkv / mean / rstd; writes dx, dqkv [B,N,3C] and
 * (when rng_state != NULL and p_drop > 0: else may be NULL) dys [B,N,C] -- the operands of the two weight gradients, which stay
 * stj_gemm launches: dWproj = a^T dys (+ colsum), dWqkv = ln^T dqkv (+ colsum), a / ln saved by the forward kernel.  C in {96,192,384}. */
int stj_swin_attn_bwd(const void* x, const void* dy, const void* qkv, const float* mean, const float* rstd, const float* gamma,
                      const void* wqkv, const void* wproj, const float* table, void* dx, void* dqkv, void* dys,
                      float* dtable, int tparts, float* dgamma, float* dbeta, int nparts, long long part_stride,
                      int B, int res, int C, int shift, const long long* rng_state, int site, float p_drop, int dtype, void* ws,
                      hipStream_t stream);

/* Fused Cross_AttentionT block: the 8 time-separated cross-attentions of TrajNetCrossAttention (trajNet.py:189-234 Cross_AttentionT --
 * tfa MultiHeadAttention(head 42 x 3, out 128, dropout .1), LayerNorm(1e-3), Dense(512, elu), Dropout, Dense(384), Dropout,
 * LayerNorm(1e-3) -- and :305-317, the loop over the waypoints with `+ query`) as ONE kernel per direction (csrc/xattn_fused.hip):
 * one workgroup per 64 tokens of a (set z, scene b), every intermediate chained in registers, weights streamed through LDS.
 *   stj_xattn_pack: the set's weights (f32 masters wq [3,384,42], wo [3,42,128], w1 [128,512], w2 [512,384] of set 0; set z lies
 *     zstride elements further) -> `pack`, Z streams of stj_xattn_pack_workspace_bytes(dtype) bytes in the activation dtype, laid out
 *     as the LDS images the kernels stage (once per step: the weights change).
 *   stj_xattn_fwd: y [Z,B,HW,384] = LN2(dropout(dropout(elu(LN1(MHA(query,k,v)) W1 + b1)) W2 + b2)) + query.  query [Z,B,HW,384];
 *     k, v [Z,B,64,126] = key Wk[z], key Wv[z] (stj_gemm); kvalid int32 [B,64] or NULL; bo/g1/be1 [128], b1 [512], b2/g2/be2 [384]
 *     f32 vectors of set 0.  rng_state != NULL: training, the three dropout sites draw with the layouts [Z,B,3,HW,64],
 *     [Z,B*HW,512], [Z,B*HW,384] (as stj_dropout would).  sq, so [Z,B,HW,144], sv1 [.,128], su2 [.,384]: saves for backward
 *     (all or none; NULL for inference).  HW % 64 == 0.
 *   stj_xattn_bwd: reads dy, query, k, v, sq, sv1, su2 and the pack; writes dquery, dk, dv (via the f32 per-tile partials dkp / dvp,
 *     stj_xattn_bwd_workspace_bytes(Z,B,HW) bytes each, reduced by a second launch) and the operands of the weight-gradient
 *     stj_gemm launches the caller makes: hd, dpre [Z,B,HW,512], du2 [.,384], n1, dv1 [.,128], dq [.,144]
 *     (dW2 = hd^T du2 + colsum, dW1 = n1^T dpre + colsum, dWo[h] = so_h^T dv1, dWq[h] = query^T dq_h);
 *     "+=": dg1, dbe1, dbo [128], dg2, dbe2 [384] of set z at + z * zstride. */
long long stj_xattn_pack_workspace_bytes(int dtype);
long long stj_xattn_pack_tail_workspace_bytes(int dtype);    /* once behind the Z streams: the kernels copy fixed-size pieces */
int stj_xattn_pack(const float* wq, const float* wo, const float* w1, const float* w2, long long zstride, int Z, void* pack, int dtype,
                   hipStream_t stream);
int stj_xattn_fwd(const void* query, const void* k, const void* v, const int* kvalid, const void* pack, const float* bo,
                  const float* g1, const float* be1, const float* b1, const float* b2, const float* g2, const float* be2,
                  long long zstride, void* y, void* sq, void* so, void* sv1, void* su2, int Z, int B, int HW,
                  const long long* rng_state, int site_a, int site_1, int site_2, float p_drop, int dtype, hipStream_t stream);
long long stj_xattn_bwd_workspace_bytes(int Z, int B, int HW);
int stj_xattn_bwd(const void* dy, const void* query, const void* k, const void* v, const int* kvalid, const void* pack,
                  const float* g1, const float* be1, const float* b1, const float* g2, long long zstride, const void* sq,
                  const void* sv1, const void* su2, void* dquery, void* dk, void* dv, float* dkp, float* dvp, void* hd,
                  void* dpre, void* du2, void* n1, void* dv1, void* dq, float* dg1, float* dbe1, float* dbo, float* dg2,
                  float* dbe2, int Z, int B, int HW, const long long* rng_state, int site_a, int site_1, int site_2,
                  float p_drop, int dtype, hipStream_t stream);

/* Row softmax of the global attentions: P = softmax(S + bias + (-10e9 where !(qvalid&kvalid))) (tfa MHA mask
 * semantics, f32 add); S f32 [batch,H,Nq,Nk] (Nk <= 256).  bwd: dS = P*(dP - sum(P dP)). */
int stj_softmax_fwd(const float* S, void* P, const int* qvalid, const int* kvalid, const float* bias,
                    long long batch, int H, int Nq, int Nk, int dtype, hipStream_t stream);
int stj_softmax_bwd(const void* P, const float* dP, void* dS, long long rows, int Nk, int dtype, hipStream_t stream);
/* Tiny attention (at most 16 queries / keys per batch element and head: the TrajEncoder's self-attention over 11 time steps, trajNet.py:33,42)
 * in one launch per direction: o [Bt,N,H*d] = dropout(softmax(scale q k^T, masked logits += -10e9)) v; q, k, v [Bt,N,H*d]; qvalid / kvalid
 * int32 [Bt,N] or NULL; dropout on the coefficients [Bt,H,N,N] at (rng_state, site), drawn as stj_dropout draws that tensor (NULL / p = 0:
 * none).  Backward recomputes the probabilities: dq, dk, dv written.  stj_small_attn_supported(N, H, d, dtype) = 1 for the geometries it
 * takes (otherwise: stj_gemm + stj_softmax_* + stj_dropout). */
int stj_small_attn_supported(int N, int H, int d, int dtype);
int stj_small_attn_fwd(const void* q, const void* k, const void* v, const int* qvalid, const int* kvalid, void* o, long long Bt, int N, int H,
                       int d, float scale, const long long* rng_state, int site, float p_drop, int dtype, hipStream_t stream);
int stj_small_attn_bwd(const void* q, const void* k, const void* v, const int* qvalid, const int* kvalid, const void* dO, void* dq, void* dk,
                       void* dv, long long Bt, int N, int H, int d, float scale, const long long* rng_state, int site, float p_drop, int dtype,
                       hipStream_t stream);
/* Fused agent branch (csrc/agent_fused.hip).
 * stj_agent_pack: transposed copies ([N][K], K contiguous, activation dtype) of the branch's eleven Dense / tfa kernels -- the layout
 * the forward kernels stream as MFMA A fragments -- into `out` (stj_agent_pack_workspace_bytes(dtype) bytes); once per step.
 * stj_agent_enc_fwd / _bwd: TrajEncoder.call (trajNet.py:38-48) for all B (n_obs + n_occ) agents in ONE launch per direction: Conv1D(5 -> 64)
 * + ELU, the 4-head tfa self-attention over the 11 steps with the (x != 0) mask and dropout on the coefficients, GlobalMaxPooling1D,
 * Dense(3 -> 64) on the step-0 type one-hot, concat, Dense(384 -> 384) + ELU.  Forward writes enc and the agent mask cmi, and, when the
 * five s_* pointers are given, what backward reads.  Backward writes the three dY tensors whose weight gradients are the CALLER's
 * (dW += X^T dY through stj_wgrad_group / stj_gemm with X = s_cat / s_att / s_nodes) and accumulates the 5 x 64, 64 and 3 x 64 ones itself. */
typedef struct stj_agent_weights {
  const float* e_wq; const float* e_wk; const float* e_wv; const float* e_wo; const float* e_ws;
  const float* i_wq; const float* i_wk; const float* i_wv; const float* i_wo; const float* i_w1; const float* i_w2;
} stj_agent_weights;
typedef struct stj_agent_enc_args {
  const float* obs; const float* occ;
  int n_obs, n_occ, B, dtype;
  const void* pack;
  const float* wn; const float* bn; const float* wv3; const float* bo; const float* bs;
  void* enc; int* cmi;
  void* s_nodes; void* s_qkv; void* s_att; void* s_pmask; void* s_cat;
  const long long* rng_state; int site; float p_drop;
  const void* d_enc; int d_enc_f32; const void* wq; const void* wk; const void* wv; const void* wo; const void* ws;
  void* dpre_s; void* dout; void* dqkv; float* dwn; float* dbn; float* dwv3;
} stj_agent_enc_args;
typedef struct stj_agent_int_args {
  const void* enc; const int* cmi;
  int n_obs, n_occ, B, dtype;
  const void* pack;
  const void* seg;
  const float* bo; const float* g1; const float* be1; const float* b1; const float* b2; const float* g2; const float* be2;
  const float* g_obs; const float* b_obs; const float* g_occ; const float* b_occ;
  void* key;
  float* ws_v1; float* ws_u2;
  void* s_concat; void* s_qin; void* s_q; void* s_k; void* s_v; void* s_att; void* s_v1; void* s_n1; void* s_h; void* s_u2; void* s_out;
  const long long* rng_state; int site_a, site_1, site_2; float p_drop;
  const void* dkey;
  const void* wq; const void* wk; const void* wv; const void* wo; const void* w1; const void* w2;
  float* d_enc;
  float* ws_dn1;
  void* dq; void* dk; void* dv; void* dv1; void* dpre1; void* dz2;
  float* dseg; float* dg1; float* dbe1; float* dg2; float* dbe2; float* dg_obs; float* db_obs; float* dg_occ; float* db_occ;
} stj_agent_int_args;
long long stj_agent_pack_workspace_bytes(int dtype);
int stj_agent_pack(const stj_agent_weights* w, void* out, int dtype, hipStream_t stream);
int stj_agent_enc_supported(int n_obs, int n_occ, int Tn, int dtype);
int stj_agent_enc_fwd(const stj_agent_enc_args* a, hipStream_t stream);
int stj_agent_enc_bwd(const stj_agent_enc_args* a, hipStream_t stream);
/* stj_agent_int_fwd / _bwd: the 64-agent interaction block of TrajNet.call (trajNet.py:135-187 with Cross_Attention.call :79-87) per scene in
 * ONE launch per direction: masked concat, segment embedding, the 6-head tfa attention (mask cm (x) cm, dropout on the coefficients), output
 * projection, LayerNorm(1e-3), Dense(1536, elu), Dropout, Dense(384), Dropout, LayerNorm(1e-3), enc + value + embed, obs_norm | occ_norm.
 * Three launches per direction: (scene, head) workgroups for the attention, (scene, hidden chunk) workgroups for the FFN -- their partial
 * sums are written as f32 slabs (ws_v1 [6][B 64][384], ws_u2 [4][..] forward; ws_dn1 [4][..], d_enc [7][..] backward) that the next launch adds
 * in a fixed order (no atomics on activations: bitwise reproducible) -- and a row-wise tail.  stj_agent_enc_bwd takes d_enc with d_enc_f32 = 7.
 * 16-bit dtypes, 64 agents per scene (stj_agent_int_supported); the f32 parity mode keeps the layer-by-layer chain.  Forward writes key
 * [B 64][384] and, when the eleven s_* pointers are given, what backward reads.  Backward writes d_enc and the six dY tensors whose weight
 * gradients are the caller's (X = s_qin / s_concat / s_concat / s_att / s_n1 / s_h), and accumulates seg_embed and the LayerNorm parameters. */
int stj_agent_int_supported(int n_obs, int n_occ, int dtype);
int stj_agent_int_fwd(const stj_agent_int_args* a, hipStream_t stream);
int stj_agent_int_bwd(const stj_agent_int_args* a, hipStream_t stream);
/* Fused FG-MSA offset head (csrc/fgoff_fused.hip): offset = tanh(conv_offset(q)) * (H / 2) with conv_offset = grouped 3x3 conv
 * (8 groups of 48 channels, SAME) -> LayerNorm(eps) -> gelu -> per-group 1x1 conv 48 -> 2 without bias (FG_MSA.py:84-92,109-123), ONE launch
 * per direction instead of stj_im2col3 + stj_gemm + stj_layernorm_fwd + stj_unary_fwd + stj_fg_offset_fwd (and their backward).  One workgroup
 * per 16 or 32 pixels of whole image rows; the conv is an implicit GEMM over an LDS halo tile with the weights streamed as MFMA A fragments from the pack.
 * stj_fgoff_pack: w = conv_offset_0/kernel [3][3][48][384] f32 -> both directions' fragments (stj_fgoff_pack_workspace_bytes(dtype)
 * bytes), once per step.  Training (cols / c / mean / rstd given) also writes the im2col matrix and the saved tensors; the
 * backward writes dq and dc and ADDS the small parameter gradients; the conv kernel's gradient cols^T dc is the caller's GEMM.
 * C = 384, 8 groups; W = 16 (any dtype), W = 8 with even H (any dtype), W = 32 (16-bit dtypes): stj_fgoff_supported. */
typedef struct stj_fgoff_args {
  int B, H, W, dtype;
  float scale, eps;                          /* offset range H / 2 (FG_MSA.py:139); LayerNorm epsilon */
  const void* q;                             /* [B][H][W][384], activation dtype */
  const void* pack;                          /* stj_fgoff_pack output */
  const float* bias; const float* gamma; const float* beta;   /* conv_offset_0/bias, conv_norm gamma / beta: f32 masters [384] */
  const void* w1;                            /* conv_offset_proj/kernel [48][2], ACTIVATION dtype */
  void* off;                                 /* [B][8][H W][2] (forward: out; backward: in) */
  void* cols; void* c; float* mean; float* rstd;     /* training: [B H W][8][432], [B H W][384], [B H W], [B H W]; all NULL: inference */
  const void* doff;                          /* backward: gradient of off */
  void* dc; void* dq;                        /* written: gradient of the conv output [B H W][384], of q [B][H][W][384] */
  float* d_w1; float* d_gamma; float* d_beta; float* d_bias;     /* += (atomics) */
} stj_fgoff_args;
int stj_fgoff_supported(int H, int W, int C, int G, int dtype);
long long stj_fgoff_pack_workspace_bytes(int dtype);
int stj_fgoff_pack(const float* w, void* out, int dtype, hipStream_t stream);
int stj_fgoff_fwd(const stj_fgoff_args* a, hipStream_t stream);
int stj_fgoff_bwd(const stj_fgoff_args* a, hipStream_t stream);
/* FG-MSA relative-position bias: bilinear `sample` of rpe_table at (query - key - offset) displacements
 * (FG_MSA.py:150-172 via occu_metric.py:345-409 + tfa_image.py:87-173).  off [B,G,H*W,2], table f32 [2H-1,2W-1,G],
 * bias f32 [B,G,HW,HW]; bwd: dtable +=, doff f32 [B,G,HW,2] += (zeroed by the caller: query slices accumulate). */
int stj_fg_bias_fwd(const void* off, const float* table, float* bias, int B, int G, int Hh, int Ww, int dtype, hipStream_t stream);
int stj_fg_bias_bwd(const void* off, const float* table, const void* dbias, float* dtable, float* doff,
                    int B, int G, int Hh, int Ww, int dtype, hipStream_t stream);
/* Fused FG-MSA attention core (FG_MSA.py:138-178; csrc/fgattn.hip): a [B,HW,G*48] = softmax(scale q k^T + sampled bias) v per sample and
 * group, the bias sampled from `table` at the key's offsets as stj_fg_bias_fwd does -- the [B,G,HW,HW] logits / bias / probabilities
 * never reach HBM.  q, k, v [B,HW,G*48], off [B,G,HW,2] (activation dtype), table f32 [2Hh-1,2Ww-1,G]; lse f32 [B,G,HW] (log-sum-exp of
 * each row, the backward's input) or NULL.  Hh = Ww in {8, 16}; dtype STJ_BF16 / STJ_F16 (STJ_F32: STJ_EUNSUPPORTED -- the f32 parity
 * mode runs the layer-by-layer kernels).
 * Backward: dq, dk, dv written (dk / dv through the f32 per-tile partials dkp / dvp, stj_fg_attn_bwd_workspace_bytes() bytes each, and a
 * second launch); dtable f32 "+="; doff f32 [B,G,HW,2] written when Hh = 8 and "+=" (zeroed by the caller) when Hh = 16. */
int stj_fg_attn_fwd(const void* q, const void* k, const void* v, const void* off, const float* table, void* a, float* lse, int B, int G,
                    int Hh, int Ww, float scale, int dtype, hipStream_t stream);
long long stj_fg_attn_bwd_workspace_bytes(int B, int G, int Hh, int Ww);
int stj_fg_attn_bwd(const void* q, const void* k, const void* v, const void* off, const float* table, const void* a, const float* lse,
                    const void* da, void* dq, void* dk, void* dv, float* dkp, float* dvp, float* dtable, float* doff, int B, int G,
                    int Hh, int Ww, float scale, int dtype, hipStream_t stream);
/* FG-MSA offset head: off[b,g,hw,:] = tanh(o[b,hw,g,:] . W1) * scale (1x1 conv gc -> 2, no bias) and, when fh != NULL,
 * fh = off . W2 + b2 (1x1 conv 2 -> C2) in one launch (FG_MSA.py:136-146).  o [B,HW,G,gc] (the offset conv's own layout, no
 * regrouped copy), W1 [gc,2], W2 [2,C2] (T), b2 f32 [C2] or NULL, off [B,G,HW,2]; fh [B,G,HW,C2], or with zmajor [G,B,HW,C2];
 * qres [B,HW,C2] or NULL is added to every group's fh (zmajor + qres = the decoder query of modules.py:827-831).
 * o == NULL: off is an input and only fh is produced.  HW % 16 == 0, G <= 8.
 * bwd: doff / dfh (either may be NULL) -> dO [B,HW,G,gc] (written), dq [B,HW,C2] (sum of dfh over the groups, written; NULL to
 * skip), dW1 / dW2 / db2 f32 += (db2 may be NULL).  dO == NULL: only the fh half is differentiated and the offset gradient
 * (doff + dfh . W2^T) is written to doff_out [B,G,HW,2]. */
int stj_fg_offset_fwd(const void* o, const void* W1, const void* W2, const float* b2, const void* qres, void* off, void* fh,
                      int B, int HW, int G, int gc, int C2, float scale, int zmajor, int dtype, hipStream_t stream);
int stj_fg_offset_bwd(const void* o, const void* off, const void* W1, const void* W2, const void* doff, const void* dfh,
                      void* dO, void* dq, void* doff_out, float* dW1, float* dW2, float* db2, int B, int HW, int G, int gc,
                      int C2, float scale, int zmajor, int dtype, hipStream_t stream);

/* Decoder: UpSampling3D(1,2,2) nearest + Conv2D 3x3 SAME + bias + ELU (modules.py:746-748,732-735) with the upsample
 * folded into 16 effective 2x2-tap matrices.  prep: W f32 [3,3,Cin,Cout] -> Wf [16,Cout,Cin], Wd [16,Cin,Cout] (T).
 * fwd: X [F,Hi,Wi,Cin] -> Y [F,2Hi,2Wi,Cout].  dgrad: dP (= dY*ELU') -> dX (times ELU'(Xelu) when Xelu != NULL:
 * the layer input is itself an ELU output and its producer skips its own ELU' pass).  wgrad: dWeff f32 [16,Cout,Cin] += (zeroed
 * by the caller), dbias f32 [Cout] += ; fold: dW [3,3,Cin,Cout] += fold(dWeff). */
int stj_upconv_prep(const float* W, void* Wf, void* Wd, int Cin, int Cout, int dtype, hipStream_t stream);
int stj_upconv_fold(const float* dWeff, float* dW, int Cin, int Cout, hipStream_t stream);
int stj_upconv_fwd(const void* X, const void* Wf, const float* bias, void* Y, int F, int Hi, int Wi, int Cin,
                   int Cout, int act, int dtype, hipStream_t stream);
/* Up-conv with the decoder skip sums in the epilogue (modules.py:750-765: x = upconv(x) + skip): Y = ELU(conv + bias) + R1 and, when
 * Y2 / R2 are not NULL, Y2 = Y + R2; every sum rounded to the activation dtype like a separate add.  16-bit dtypes, Cin in {192, 384, ...}
 * (multiple of 32 above 128), Cout multiple of 32; STJ_EUNSUPPORTED otherwise.  Backward of the sums + ELU: stj_elu_res_bwd:
 * g = dy (+ dy2, then g is also written to gsum: the gradient of R1), dpre = g * ELU'(Y - R1) (r == NULL: y is the ELU output itself). */
int stj_upconv_fwd_res(const void* X, const void* Wf, const float* bias, void* Y, const void* R1, void* Y2, const void* R2, int F, int Hi,
                       int Wi, int Cin, int Cout, int dtype, hipStream_t stream);
int stj_elu_res_bwd(const void* dy, const void* dy2, const void* y, const void* r, void* dpre, void* gsum, long long n, int dtype,
                    hipStream_t stream);
/* Backward junction of a decoder level whose skips R1 / R2 are ELU outputs themselves (the time-collapsed Conv3D + ELU of an encoder stage,
 * modules.py:750-765), one pass: g = dy1 (+ dy2, rounded), dpre = g ELU'(y) (y = the up-conv's ELU output), dr1 = g ELU'(r1) and
 * dr2 = dy2 ELU'(r2) -- the gradients of the skips' PRE-activations -- in place of stj_elu_res_bwd + one stj_unary_bwd per skip.
 * dy2 / r2 / dr2 all NULL: a level with one skip. */
int stj_skip_junction_bwd(const void* dy1, const void* dy2, const void* y, const void* r1, const void* r2, void* dpre, void* dr1,
                          void* dr2, long long n, int dtype, hipStream_t stream);
int stj_upconv_dgrad(const void* dP, const void* Wd, void* dX, const void* Xelu, int F, int Hi, int Wi, int Cin, int Cout,
                     int dtype, hipStream_t stream);
/* wg_budget: workgroups the two large weight-gradient launches (>= 64x64 inputs) may occupy.  0 = 128: half the CUs, because in a
 * step whose branches run on concurrent streams these launches are deferred next to chains of short kernels, which then find the
 * other half free (alone the kernel is 1.5x faster on 256).  A host that runs every kernel alone (serial / per-kernel timing) passes
 * 256.  An argument, not library state: the ABI is stateless and re-entrant. */
int stj_upconv_wgrad(const void* X, const void* dP, float* dWeff, float* dbias, int db_parts, int F, int Hi, int Wi, int Cin,
                     int Cout, int wg_budget, int dtype, hipStream_t stream);
/* wgrad: dbias (optional) is f32 [db_parts][Cout], "+=": workgroup i adds its share of the bias gradient into copy i % db_parts
 * and the caller sums the copies (db_parts = 1: plain [Cout]). */
/* Output heads: Conv2D 3x3 SAME C->2, no activation (modules.py:767-770), written with strides straight into the
 * [B,H,W,32] f32 model output (concat + transpose of modules.py:770,838).  Y element (b,t,y,x,o) at
 * Y + b*y_bstride + t*y_tstride + (y*W+x)*y_pstride + o.  bwd with elu_in != 0: X is an ELU output, dX is multiplied by
 * ELU'(x) (gradient w.r.t. the producing conv's pre-activation; the producer then skips its own ELU' pass). */
int stj_outconv_fwd(const void* X, const float* W, const float* bias, float* Y, int F, int Hh, int Ww, int C, int Tn,
                    long long y_bstride, long long y_tstride, long long y_pstride, int dtype, hipStream_t stream);
/* Both heads in ONE launch (the pair of Conv2D 48->2 of modules.py:767-770 + concat + transpose :838): Y [B,H,W,4*Tn] f32, channel
 * 4 t + 2 head + o; X0 / X1 the two decoder branches [F,H,W,48], frames f = b*Tn + t (t_major = 0) or t*B + b (t_major = 1).  A
 * workgroup runs all 16 (waypoint, head) frames of a spatial tile and writes whole 128-byte output lines.  16-bit dtypes, C = 48,
 * Tn = 8 only (STJ_EUNSUPPORTED otherwise: call stj_outconv_fwd per head). */
int stj_outconv_pair_fwd(const void* X0, const void* X1, const float* W0, const float* W1, const float* bias0, const float* bias1,
                         float* Y, int B, int Tn, int H, int W, int C, int t_major, int dtype, hipStream_t stream);
/* Inference form of the last decoder level + both heads (modules.py:746-748 at 96 -> 48, then :767-770,838) without the [F,H,W,48]
 * tensor between them: out[p][o] = sum_taps z[p + tap][tap, o] with z[q][tap, o] = sum_c Whead[tap][c][o] ELU(upconv)[q][c].
 * stj_upconv_fwd_head: the up-conv of stj_upconv_fwd (same X, Wf, bias) whose epilogue projects every output pixel onto the head kernel
 * Whead f32 [3,3,48,2] and writes Z [F,2Hi,2Wi,20] (18 + 2 zero channels; 24 until round 6) in the activation dtype; stj_outconv_pair_gather: Y [B,H,W,32]
 * f32, channel 4 t + 2 head + o = bias + the 9-neighbour sum of Z0 / Z1 (the two decoder branches; frames as in stj_outconv_pair_fwd).
 * 16-bit dtypes, Cin = 96, Cout = 48, whole 8 x 16 tiles, Tn = 8; STJ_EUNSUPPORTED otherwise. */
int stj_upconv_fwd_head(const void* X, const void* Wf, const float* bias, const float* Whead, void* Z, int F, int Hi, int Wi, int Cin,
                        int Cout, int dtype, hipStream_t stream);
int stj_outconv_pair_gather(const void* Z0, const void* Z1, const float* bias0, const float* bias1, float* Y, int B, int Tn, int H, int W,
                            int t_major, int dtype, hipStream_t stream);
int stj_outconv_bwd(const void* X, const float* W, const float* dY, void* dX, float* dW, float* db, int F, int Hh, int Ww,
                    int C, int Tn, long long y_bstride, long long y_tstride, long long y_pstride, int elu_in, void* ws,
                    long long ws_bytes, int dtype, hipStream_t stream);
long long stj_outconv_bwd_workspace_bytes(void);   /* size of the caller-owned scratch `ws` (not zeroed; may be NULL: slower path) */
/* PatchEmbed Conv2D k=4 s=4 VALID as im2col (+ f32->T cast, + stride-2 pick of ogm[...,0]; modules.py:430-431,572). */
int stj_im2col_patch(const float* src, void* dst, int B, int H, int W, int Cin, long long pix_stride, int ch_stride,
                     int dtype, hipStream_t stream);
/* Fused PatchEmbed + the stem's sums / norms (modules.py:430-446 PatchEmbed.call: Conv2D k=4 s=4 VALID -> reshape -> LayerNorm(1e-5);
 * modules.py:572-590: patch_embed_vecicle(ogm[...,0]) + patch_embed_map(map) -> all_patch_norm; :576-578 patch_embed_flow -> flow_norm):
 *   pre = cols(src) @ w + bias ; x2 = LN(pre; gamma, beta) [+ add] ; y = gamma2 ? LN(x2; gamma2, beta2) : x2
 * src f32 rasters read as src[((b*H+y)*W+x)*pix_stride + c*ch_stride] (as stj_im2col_patch); w [16*Cin, Cout] type T (Keras kernel
 * [4,4,Cin,Cout] flattened); bias / gamma / beta f32; add, pre, x2, y [B*(H/4)*(W/4), Cout] type T; cols [.., 16*Cin] type T.
 * Optional outputs (NULL = not written): cols (the im2col rows, for dW = cols^T dpre), pre + mean + rstd, x2 + mean2 + rstd2 (what
 * stj_layernorm_bwd needs).  pre and x2 are rounded to T before they are normalised.  Built for Cin in {11, 3, 2}, Cout = 96. */
int stj_patch_embed_supported(int Cin, int Cout, int dtype);
int stj_patch_embed_fwd(const float* src, const void* w, const float* bias, const float* gamma, const float* beta, const void* add,
                        const float* gamma2, const float* beta2, void* cols, void* pre, void* x2, void* y, float* mean, float* rstd,
                        float* mean2, float* rstd2, int B, int H, int W, int Cin, long long pix_stride, int ch_stride, int Cout,
                        float eps, int dtype, hipStream_t stream);
/* grouped 3x3 SAME conv of FG-MSA (FG_MSA.py:51) as im2col / col2im around the batched GEMM. */
int stj_im2col3(const void* x, void* cols, int N, int H, int W, int G, int Cg, int dtype, hipStream_t stream);
int stj_col2im3(const void* dcols, void* dx, int N, int H, int W, int G, int Cg, int dtype, hipStream_t stream);

/* OGMFlow_loss (loss.py:50-170 with train.py:195-196 flags).  All tensors f32: logits [B,H,W,32] (channel 4k+{0,1,2,3},
 * train.py:105-123), gt_obs/gt_occ/origin [B,8,H,W,1], gt_flow [B,8,H,W,2].
 * auc_gate: res_k = [Keras PR-AUC(true_all, warp(origin, id+gt_flow)*true_all) > 0] (loss.py:127-137); hist int[8*202] scratch, zero on entry.
 * fwd: sums f32[32*40] scratch (32 copies of the 40 accumulators the workgroups spread their atomics over; zero on entry), loss f32[5] = observed_xe, occluded_xe, flow, flow_warp_xe, their sum (train.py:221); coef f32[32] for bwd.
 * bwd: dlogits = sum_j upstream[j] * dloss_j/dlogits; with flag bit 3 (bwd only) upstream is ONE value for all four terms.
 * flags: bit 0 = flow-warp term on (not no_use_warp), bit 1 = use_focal_loss (tfa SigmoidFocalCrossEntropy added to the three
 * occupancy terms, loss.py:183-190,212-219,244-245), bit 2 = use_pred (loss.py:151-154,253-268); the same value goes to fwd and bwd. */
int stj_loss_auc_gate(const float* gt_obs, const float* gt_occ, const float* gt_flow, const float* origin,
                      int* hist, float* gate, float* auc_out, int B, int H, int W, hipStream_t stream);
int stj_loss_fwd(const float* logits, const float* gt_obs, const float* gt_occ, const float* gt_flow, const float* origin,
                 const float* gate, float* sums, float* loss, float* coef, int B, int H, int W, float ogm_w, float occ_w,
                 float flow_origin_w, float replica, int flags, hipStream_t stream);
int stj_loss_bwd(const float* logits, const float* gt_obs, const float* gt_occ, const float* gt_flow, const float* origin,
                 const float* coef, const float* upstream, float* dlogits, int B, int H, int W, int flags, hipStream_t stream);
/* The same two passes as ONE (round 6): every backward coefficient of loss.py:161-170 under train.py:221-223 (unit gradient on the sum of
 * the four terms) depends on the ground truth alone -- weights, pixel count, the AUC gate, the per-waypoint count of pixels with a
 * non-zero true flow (loss.py:279-291) -- so
 * coef:    coef f32[32] = what stj_loss_fwd would write, from gt_flow and the gate (cnt int[8] scratch, zero on entry);
 * fwd_bwd: loss f32[5], coef_out f32[32] as stj_loss_fwd, AND dlogits as stj_loss_bwd with upstream == 1 (flag bit 3), on one read of
 *          the logits and the ground truth; coef_in = coef's output; sums f32[128*40] scratch, zero on entry.  loss == NULL: the pass
 *          alone; stj_loss_finalize (any stream behind it) then writes loss and coef_out from sums. */
int stj_loss_coef(const float* gt_flow, const float* gate, int* cnt, float* coef, int B, int H, int W, float ogm_w, float occ_w,
                  float flow_origin_w, float replica, int flags, hipStream_t stream);
int stj_loss_finalize(const float* sums, const float* gate, float* loss, float* coef_out, int B, int H, int W, float ogm_w,
                      float occ_w, float flow_origin_w, float replica, int flags, hipStream_t stream);
/* auc_gate + coef on one pass over the ground truth: hist int[8*202 + 8] scratch (zero on entry; the last 8 collect the flow counts) */
int stj_loss_gate_coef(const float* gt_obs, const float* gt_occ, const float* gt_flow, const float* origin, int* hist, float* gate,
                       float* auc_out, float* coef, int B, int H, int W, float ogm_w, float occ_w, float flow_origin_w,
                       float replica, int flags, hipStream_t stream);
int stj_loss_fwd_bwd(const float* logits, const float* gt_obs, const float* gt_occ, const float* gt_flow, const float* origin,
                     const float* gate, const float* coef_in, float* sums, float* loss, float* coef_out, float* dlogits,
                     int B, int H, int W, float ogm_w, float occ_w, float flow_origin_w, float replica, int flags, hipStream_t stream);

/* TFRecord feature decode (train.py:87-103, inference.py:84-96 _parse_image_function: tf.io.decode_raw + reshape + centre crop
 * + cast).  src: the raw feature bytes of a batch, [n_outer][H][W][C] elements of kind 0 bool/uint8 (v != 0), 1 int8,
 * 2 float32, 3 float64; dst f32 [n_outer][Ho][Wo][C] = scale * src[:, y0:y0+Ho, x0:x0+Wo, :]. */
int stj_decode_raw(const void* src, int kind, float* dst, long long n_outer, int H, int W, int C, int y0, int x0, int Ho,
                   int Wo, float scale, hipStream_t stream);

/* trajNet input plumbing (trajNet.py:125-140), one launch: obs [B,n_obs,Tn,8] and occ [B,n_occ,Tn,8] (f32, 16-byte aligned) ->
 * x5 [B*A*Tn,5] node features, v3 [B*A,3] vector features of step 0, vt [B*A,Tn] int32 step-valid (feature 0 != 0), cmi [B*A] int32 /
 * cmf [B*A] (type T) agent-valid (any step valid); A = n_obs + n_occ, obs rows first. */
int stj_agent_prep(const float* obs, const float* occ, int n_obs, int n_occ, int B, int Tn, void* x5, void* v3, int* vt, int* cmi,
                   void* cmf, int dtype, hipStream_t stream);
/* trajNet branch sums (trajNet.py:166-171), enc / value / concat / qin / out [B,A,C], embed [A,C], cm [B,A], all type T:
 * mix: concat = enc * cm, qin = concat + embed;  bwd: denc = (dconcat + dqin) * cm, dembed = sum_b dqin (either gradient may be NULL).
 * sum: out = enc + value + embed;                bwd: dembed = sum_b dout (the gradients of enc and value are dout itself). */
int stj_agent_mix_fwd(const void* enc, const void* embed, const void* cm, void* concat, void* qin, int B, int A, int C, int dtype,
                      hipStream_t stream);
int stj_agent_mix_bwd(const void* dconcat, const void* dqin, const void* cm, void* denc, void* dembed, int B, int A, int C, int dtype,
                      hipStream_t stream);
int stj_agent_sum_fwd(const void* enc, const void* value, const void* embed, void* out, int B, int A, int C, int dtype, hipStream_t stream);
int stj_agent_sum_bwd(const void* dout, void* dembed, int B, int A, int C, int dtype, hipStream_t stream);
/* The tail of TrajNet.call in one launch per direction (trajNet.py:171-187): out = enc + value + embed (rounded like stj_agent_sum_fwd), then
 * LayerNorm(eps) with parameters (g0, b0) on the first n0 agents of every scene (obs_norm) and (g1, b1) on the others (occ_norm): y [B,A,C].
 * Saves out [B,A,C], mean / rstd f32 [B*A].  Backward: dout [B,A,C] (= d_enc = d_value), dembed [A,C] = sum over scenes of dout (written);
 * dg0 / db0 / dg1 / db1 f32 "+=".  C = 384. */
int stj_agent_out_fwd(const void* enc, const void* value, const void* embed, const float* g0, const float* b0, const float* g1, const float* b1,
                      void* out, void* y, float* mean, float* rstd, int B, int A, int n0, int C, float eps, int dtype, hipStream_t stream);
int stj_agent_out_bwd(const void* dy, const void* out, const float* mean, const float* rstd, const float* g0, const float* g1, void* dout,
                      void* dembed, float* dg0, float* db0, float* dg1, float* db1, int B, int A, int n0, int C, int dtype, hipStream_t stream);
/* Time-kernel collapse of the decoder's Conv3D(8,1,1) SAME skips (modules.py:693-698,709-716,750-765): the
 * input is the same frame at all 8 steps, so step t needs W_t = sum_{j=max(0,3-t)}^{min(7,10-t)} W[j].  W f32 [8][n] (n = Cin*Cout),
 * Wz T [8][n].  fold (backward): dW[j] += sum over the t whose window contains j of dWz[t]. */
int stj_time_collapse(const float* W, void* Wz, long long n, int dtype, hipStream_t stream);
int stj_time_fold(const float* dWz, float* dW, long long n, hipStream_t stream);
/* g[idx[i]] += parts[i]; parts[i] = 0 for i < n: the partial-gradient copies some backward kernels rotate their atomics over
 * (LayerNorm gamma / beta, bias tables, conv biases) folded into the flat gradient buffer and re-zeroed, one launch. */
int stj_fold_parts(float* g, const long long* idx, float* parts, long long n, hipStream_t stream);

/* Host-side CRC-32C (Castagnoli) for the two TensorFlow file formats on either side of the hot path: TFRecord framing
 * (train.py:75-78) and the checkpoint bundle written / read by save_weights / load_weights (train.py:358,366,372;
 * inference.py:283).  *crc is the running, unmasked CRC: 0 before the first chunk, the checksum after the last.  Runs on the
 * calling thread; touches no device state. */
int stj_crc32c(const void* data, long long n, unsigned int* crc);

/* Evaluation metrics (occu_metric.py:26-140, evaluated every train / validation step: train.py:243-249,280-282): observed and
 * occluded PR-AUC + soft IoU, flow EPE, flow-warped occupancy AUC + IoU, means over the 8 waypoints.  pred [B,H,W,32]
 * (channel 4k+{obs,occ,flow_x,flow_y}); pred_is_logits: sigmoid is applied to the occupancy channels (train.py:142-154).
 * hist int[8*3*202] and sums f32[8*11] MUST BE ZERO on entry; auc f32[24] scratch; out f32[7] in the order of the proto
 * fields at occu_metric.py:130-139. */
int stj_metrics(const float* pred, const float* gt_obs, const float* gt_occ, const float* gt_flow, const float* origin,
                int* hist, float* sums, float* auc, float* out, int B, int H, int W, int pred_is_logits, int use_warp,
                hipStream_t stream);

/* Training-time randomness.  Keras Dropout / tfa-MHA attention dropout (trajNet.py:33,71,75,77,195,209,211) and DropPath
 * (modules.py:137-151) share one rule: y = [res +] keep(draw) * x / (1 - p), keep = U[0,1) >= p, draw(i) = i / inner
 * (inner = 1: per element; inner = elements per sample: DropPath with p = drop_prob).  U comes from Philox-4x32-10 keyed by
 * state = device int64[2] {seed, step} and `site`; backward calls stj_dropout on dY with the same (state, site).
 * stj_dropout_mask writes the keep bytes of the first ndraw draws (test hook: the oracle is fed the same masks). */
int stj_rng_advance(long long* state, hipStream_t stream);
/* stj_rng_advance that also writes the advanced {seed, step} to snap[2] (one launch instead of the advance + a device copy). */
int stj_rng_advance_snap(long long* state, long long* snap, hipStream_t stream);
int stj_dropout(const void* x, const void* res, void* y, long long n, long long inner, float p, const long long* state,
                int site, int dtype, hipStream_t stream);
int stj_dropout_mask(unsigned char* mask, long long ndraw, float p, const long long* state, int site, hipStream_t stream);
/* Keras Nadam (train.py:197,224) over flat f32 buffers: one fused pass.  Host-side scalars:
 * cg = (1-mu_t)/(1-prod_t), cm = mu_{t+1}/(1-prod_{t+1}), vhat_scale = 1/(1-b2^t); g is multiplied by gscale first. */
int stj_nadam_step(float* w, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
                   float cg, float cm, float vhat_scale, float gscale, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
