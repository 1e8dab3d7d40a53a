/* C ABI of libwm_hip.so — the MI355X (gfx950) kernels behind the watermark train step.
 *
 * The reference has no native/FFI surface: its hot path is PyTorch ATen calls inside
 * uformerWM/model.py.  Each entry point below replaces the ATen op group named in its comment
 * (file:line in /root/reference).  Conventions:
 *   - every pointer is DEVICE memory (fp32 unless stated), owned by the caller, never retained; `void*` tensor
 *     arguments carry fp32 or bf16 according to the `*_bf16` flag next to them (bf16 = the wide activations of
 *     mode 1; the residual stream, parameters, gradients and statistics are always fp32);
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it, no host sync,
 *     no allocation (hipGraph-capturable);
 *   - return 0 on success, WM_EINVAL (22) for a shape/alignment the kernel does not support,
 *     1000+hipError_t for a launch failure.  Nothing is written on a non-zero return;
 *   - `mode`: 0 = fp32 MFMA (exact fmaf-chain accumulation, "parity mode"),
 *             1 = bf16 MFMA operands with fp32 accumulation ("fast mode");
 *   - parameter-gradient outputs are ACCUMULATED with fp32 atomics into buffers the caller
 *     zeroes once per step (the flat gradient arena).
 */
#ifndef WM_KERNELS_H
#define WM_KERNELS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* 7 (round 4): wm_attn_half_*, wm_leff_*, wm_sum_bf16_rows, wm_conv4x4s2_*; `terms` argument of the n_fft = 255 transforms;
 * WmGemm grew by the grid-addressing fields at its END (a binding built against version 6 passes a shorter struct: rebuild it) */
int wm_abi_version(void);

/* ---- GEMM with fused prologue/epilogue -----------------------------------------------------
 * C[m,n] = epi( sum_k opA(A(m,k)) * kscale[k/k_per_scale] * opB(B(n,k)) )
 * epi(v) = ((v + bias[n]) * rowscale[m/rows_per_scale]) * gelu'(gaux[m,n]) + res[m,n]
 * (each factor optional).  Replaces nn.Linear fwd/bwd + GELU + residual/DropPath adds of
 * LinearProjection (model.py:460-471), WindowAttention.proj (:549), LeFF.linear1/2 (:700,711),
 * Downsample/Upsample as im2col GEMMs (:763,789) and their autograd backward.
 */
/* ---- deferred partial-sum reductions ------------------------------------------------------------------------
 * Kernels that accumulate small parameter gradients from many workgroups store per-workgroup partial sums to a scratch
 * buffer and normally launch a small reduce kernel right away.  Given a `defer` pointer they instead describe that
 * reduction in WmReduceJob entries (their scratch slice must then stay untouched) and ONE wm_reduce_jobs() call performs
 * up to 16 of them: dst(r, c) += sum_p src[p * pstride + r * src_ld + c], r < rows, c < cols, p < count, with
 * dst(r, c) at dst[r * dst_rs + c * dst_cs].  count == 0 marks an entry whose producer had nothing to defer.           */
typedef struct WmReduceJob {
  const float* src; int64_t pstride; int64_t src_ld;
  int32_t count, rows, cols, reserved;
  float* dst; int64_t dst_rs, dst_cs;
} WmReduceJob;
int wm_reduce_jobs(const WmReduceJob* jobs, int32_t njobs, void* stream);

typedef struct WmGemm {
  const void* A; const void* B; void* C;    /* fp32, or bf16 where the *_bf16 flag below is set (mode 1 only)      */
  int32_t M, N, K;
  int64_t lda, ldb, ldc;     /* lda: row stride if a_kc else k stride (elements); same for ldb */
  int32_t a_kc, b_kc;        /* 1: operand is K-contiguous, 0: row-contiguous                   */
  int32_t a_op, b_op;        /* 0 none, 1 GELU applied while staging the operand                */
  const float* bias;         /* [N] or NULL                                                     */
  const float* res; int64_t ldr;              /* residual [M,N] or NULL                        */
  const float* rowscale; int32_t rows_per_scale; /* per-sample DropPath scale or NULL           */
  const float* kscale; int32_t k_per_scale;   /* scale on A along K (weight-grad GEMMs) or NULL */
  const void* gaux; int64_t ldg;              /* multiply by gelu'(gaux[m,n]) or NULL           */
  int32_t b_ones;            /* B has a virtual row n==N of ones; that column is added to C2[m] */
  float* C2;
  int32_t atomic_out;        /* 1: atomicAdd into C (gradient accumulation, split-K)            */
  int32_t ksplit;            /* >=1 number of K splits (needs atomic_out when >1); 0 = auto     */
  int32_t mode;
  int32_t a_bf16, b_bf16, c_bf16, gaux_bf16; /* storage type of A / B / C / gaux: 0 fp32, 1 bf16 (bf16 C: no atomics)   */
  int32_t gaux_raw;          /* 1: gaux already holds the factor (gelu'(h2) saved by wm_dwconv_fwd): plain multiply  */
  float* scratch;            /* optional device workspace for split-K partial tiles (weight gradients, mode 1); NULL =  */
  int64_t scratch_floats;    /* fp32 atomics into C.  wm_gemm_scratch_floats() is always enough; contents irrelevant    */
  WmReduceJob* defer;        /* optional HOST pointer to 1 entry: the split-K reduction is described there, not launched */
  /* Optional fused LayerNorm of the OUTPUT rows (mode 1, fp32 C with residual, N in {32, 64, 128, 256}: the tile holds whole
   * rows): ln_out[M,N] (bf16) = LayerNorm(C row) * ln_gamma + ln_beta (+ ln_mod[window position of the row][n], the decoder's
   * modulator, reference model.py:996-997), ln_mean / ln_rstd [M] for the backward.  This is norm2 behind the attention
   * projection, or the NEXT block's norm1 behind fc2 (model.py:1016-1017 -> :982).  wm_gemm_ln_fusable() says whether a launch
   * qualifies; otherwise wm_gemm returns WM_EINVAL when ln_out is set.                                                    */
  const float* ln_gamma; const float* ln_beta; const float* ln_mod;
  void* ln_out; float* ln_mean; float* ln_rstd;
  int32_t ln_hw, ln_shift;
  /* Operand addressing of the strided convolution without an im2col image (Downsample, model.py:759-775; mode 1, bf16 operands,
   * the 128 x 128-family kernels only; all 0 = plain GEMM).  The layer input lives as 2 x 2 pixel blocks on a grid of
   * (n + 2)^2 rows per image (wm_conv4x4s2_* below); then every operand row of the three products is a grid row at a constant
   * row offset from its partner's:
   *   a_grid_n   > 0: A row m = (image b, Y, X) of the compact n x n enumeration is grid row b (n+2)^2 + (Y+1)(n+2) + X+1 (a_kc);
   *   a_jump_k   > 0: after a_jump_k elements of K the A pointer skips a_jump further elements (a_kc; a multiple of 128 / 8);
   *   b_seg_cols > 0: (weight gradient) column block s = n / b_seg_cols of B is read from rows b_seg_rows[s] further down,
   *                   its columns are n - s b_seg_cols of the stored row (b_seg_cols a multiple of 128).                        */
  int32_t a_grid_n, a_jump_k;
  int64_t a_jump;
  int32_t b_seg_cols, b_seg_rows[4];
} WmGemm;
int32_t wm_gemm_ln_fusable(int32_t M, int32_t N, int32_t K, int32_t mode, int32_t a_bf16);
int64_t wm_gemm_scratch_floats(void);
int wm_gemm(const WmGemm* g, void* stream);
/* profiling aid: the kernel family the last wm_gemm call of this host thread launched: 0 gemm_kernel (fp32 MFMA / generic), 1 gemm_nt_kernel,
 * 2 gemm_tn_kernel, 3 gemm_nt_pp_kernel, 4 gemm_tn_pp_kernel, 5 gemm_nt_big_kernel, 6 gemm_tn_big_kernel */
int32_t wm_last_gemm_family(void);

/* ---- LayerNorm (+ window modulator) — model.py:982,1017 (norm1/norm2), :996-997 (modulator) ---- */
int wm_layernorm_fwd(const float* x, const float* gamma, const float* beta, const float* modulator,
                     void* y, int32_t y_bf16, float* mean, float* rstd, int64_t rows, int32_t C, int32_t hw,
                     int32_t shift, void* stream);
/* dx = dres + LN'(dy); dgamma/dbeta accumulated; dmod[64,C] accumulated if not NULL.
 * scratch (optional, device, >= wm_layernorm_bwd_scratch_floats(C) floats, contents irrelevant): per-workgroup partial
 * sums of dgamma / dbeta, reduced by a second kernel; NULL = every workgroup adds with global atomics (slower).   */
int64_t wm_layernorm_bwd_scratch_floats(int32_t C);
/* dx_bf16 (optional): a bf16 copy of dx for the GEMMs that consume it next (they round it to bf16 anyway);
 * copy_scale (optional, [rows / rows_per_scale]): the copy holds dx * copy_scale[sample] — the DropPath factor of the branch
 * whose dX and dW products read it (reference model.py:1016-1017: x + drop_path(f(x))), so those launches need no scale */
int wm_layernorm_bwd(const void* dy, int32_t dy_bf16, const float* x, const float* gamma, const float* mean,
                     const float* rstd, const float* dres, float* dx, void* dx_bf16, float* dgamma, float* dbeta,
                     float* dmod, float* scratch, WmReduceJob* defer /* 2 entries or NULL */, int64_t rows, int32_t C,
                     int32_t hw, int32_t shift, const float* copy_scale, int32_t rows_per_scale, void* stream);

/* ---- 8x8 window attention — model.py:523-551 with roll/partition/reverse (:987-1013) and the
 * shift mask (:954-972) folded into the indexing.  qkv: [B*hw*hw, 3C] (q|k|v), heads of 32.    */
/* lse (optional, fp32 [tokens, heads]): log-sum-exp of every softmax row, saved for the fast backward.
 * mode 2 (fp32 storage only): split-precision products on the matrix cores (attn_split.hip), 1e-6 of the fp32 result */
int wm_winattn_fwd(const void* qkv, const float* bias_table, void* out, float* lse, int32_t io_bf16, int32_t B,
                   int32_t hw, int32_t C, int32_t shift, int32_t mode, void* stream);
/* out / lse: the forward's results; when both are given in mode 1 with bf16 storage the register-resident backward
 * is used (mode 2 with fp32 storage: its split-precision form), otherwise (NULL, or mode 0) scores are re-normalised through LDS.
 * scratch (optional, device, >= wm_winattn_bwd_scratch_floats() floats, contents irrelevant): per-workgroup partial
 * sums of the relative-position-bias gradient, reduced by a second kernel; NULL = global atomics (slower).        */
int64_t wm_winattn_bwd_scratch_floats(void);
int wm_winattn_bwd(const void* qkv, const float* bias_table, const void* out, const float* lse, const void* dout,
                   void* dqkv, float* dbias_table, float* scratch, WmReduceJob* defer /* 1 entry or NULL */, int32_t io_bf16,
                   int32_t B, int32_t hw, int32_t C, int32_t shift, int32_t mode, void* stream);

/* ---- LeFF depthwise 3x3 on token-major [B,hw,hw,Ch] — model.py:703-709.  Input is the
 * PRE-GELU linear1 output; GELU is applied while staging.  Outputs: h2 (pre-GELU, kept for gelu' in
 * backward) and optionally a2 = GELU(h2), the A operand of linear2.                            */
int wm_dwconv_fwd(const void* h1, const float* w, const float* b, void* h2 /* or NULL */, void* a2 /* gelu(h2) or NULL */,
                  void* g2 /* gelu'(h2) or NULL */, int32_t io_bf16, int32_t B, int32_t hw, int32_t Ch, void* stream);
/* dh1 = gelu'(h1) * conv^T(dh2);  dw[Ch,9], db[Ch] accumulated.
 * scratch (optional, device, >= wm_dwconv_bwd_scratch_floats(B, hw, Ch) floats, contents irrelevant): per-workgroup
 * partial sums of dw / db, reduced by a second kernel; NULL = every workgroup adds with global atomics (slower). */
int64_t wm_dwconv_bwd_scratch_floats(int32_t B, int32_t hw, int32_t Ch);
int wm_dwconv_bwd(const void* h1, const void* dh2, const float* w, void* dh1, float* dw, float* db, float* scratch,
                  WmReduceJob* defer /* 2 entries or NULL */, int32_t io_bf16, int32_t B, int32_t hw, int32_t Ch, void* stream);

/* ---- The attention half of a LeWin block as one window-local kernel (blocks with C in {32, 64, 128}, bf16 operands) —
 * LeWinTransformerBlock.forward, model.py:982-1016 (norm1, roll / window partition, modulator :996-997, WindowAttention :523-551,
 * window reverse / roll back, x + drop_path(...)) and the norm2 of :1017.  Replaces, for those blocks, wm_layernorm_fwd +
 * the q | k | v product + wm_winattn_fwd + the projection product (+ its fused norm2).
 *   x1 = x + rowscale[sample] * (attention(norm1(x) (+ modulator)) Wproj^T + bproj),  xn2 = norm2(x1) (bf16),
 *   mean1 / rstd1 / mean2 / rstd2 [tokens]: the row statistics of the two LayerNorms, lse [tokens, heads]: softmax log-sum-exp.
 *   wqkv [3C, C] (q rows, then k, then v) / wproj [C, C]: bf16 copies of the parameters; bias_table [225, heads].
 *   xn1 [tokens, C], qkv [tokens, 3C], o [tokens, C] (bf16; all three or none): what the un-fused backward launches read.  */
int32_t wm_attn_half_fusable(int32_t hw, int32_t C);
int wm_attn_half_fwd(const float* x, const float* gamma1, const float* beta1, const float* modulator /* [64, C] or NULL */,
                     const void* wqkv_bf16, const float* bqkv, const float* bias_table, const void* wproj_bf16, const float* bproj,
                     const float* rowscale /* [B] or NULL */, const float* gamma2, const float* beta2, float* x1, void* xn2,
                     float* mean1, float* rstd1, float* mean2, float* rstd2, float* lse, void* xn1, void* qkv, void* o,
                     int32_t B, int32_t hw, int32_t C, int32_t shift, void* stream);

/* The same half backward, between the two weight-gradient products, in one window-local launch:
 *   dx1 = dy + norm2'(dxn2);  dO = (dx1 * rowscale) Wproj;  dq|dk|dv = window-attention backward (qkv, o, lse from the forward);
 *   dxn1 = dq|dk|dv Wqkv;  dx = dx1 + norm1'(dxn1)
 * replacing wm_layernorm_bwd (norm2), the projection's input-gradient product, wm_winattn_bwd, the q | k | v input-gradient
 * product and wm_layernorm_bwd (norm1).  dxn2 [tokens, C] bf16 (gradient of the norm2 output), dy / x1 / x [tokens, C] fp32,
 * wproj [C, C] bf16, wqkvT [C, 3C] bf16 = the TRANSPOSED q | k | v weight.  Written: dx1h = bf16(dx1 * rowscale) and dqkv (the
 * operands of the two weight-gradient products), dx, and dxh = bf16(dx * copy_scale) if not NULL.  Accumulated through `scratch`
 * (>= wm_attn_half_bwd_scratch_floats floats; per-workgroup partial sums reduced by wm_reduce_jobs, deferred into 6 entries
 * when `defer` is given): dgamma2, dbeta2, dgamma1, dbeta1 [C], dmod [64, C] (or NULL), dbias_table [225, heads].          */
int64_t wm_attn_half_bwd_scratch_floats(int32_t B, int32_t hw, int32_t C, int32_t has_mod);
int wm_attn_half_bwd(const void* dxn2, const float* dy, const float* x1, const float* mean2, const float* rstd2,
                     const float* gamma2, const float* rowscale /* [B] or NULL */, const float* x, const float* mean1,
                     const float* rstd1, const float* gamma1, const void* qkv, const void* o, const float* lse,
                     const float* bias_table, const void* wproj_bf16, const void* wqkvT_bf16, const float* copy_scale /* [B] or NULL */,
                     void* dx1h, void* dqkv, float* dx, void* dxh /* or NULL */, float* dgamma2, float* dbeta2, float* dgamma1,
                     float* dbeta1, float* dmod /* or NULL */, float* dbias_table, float* scratch, WmReduceJob* defer /* 6 entries or NULL */,
                     int32_t B, int32_t hw, int32_t C, int32_t shift, void* stream);

/* ---- LeFF as two fused kernels (blocks with C in {32, 64, 128}, bf16 storage) — model.py:695-714 + the residual / DropPath add of
 * :1016-1017.  Replaces, for those blocks, linear1 + wm_dwconv_fwd + linear2 (forward) and linear2's input gradient +
 * wm_dwconv_bwd + linear1's input gradient (backward); the two weight-gradient products stay on wm_gemm (fc2's reads the saved
 * h2 with GELU applied while staging: b_op = 1).  An 8 x 8-pixel tile stays in LDS / registers across the three operators.
 *   forward : x2 = (gelu(dwconv3x3(gelu(xn2 W1^T + b1)) + bd) W2^T + b2) * rowscale[sample] + x1
 *             xn2 [tokens, C] bf16 (LayerNorm output), x1 / x2 [tokens, C] fp32, w1 [4C, C] / w2 [C, 4C] bf16 copies of the
 *             parameters, wd [4C, 9] / bd [4C] fp32; h1, h2 [tokens, 4C] bf16 (both or neither): the two pre-activations,
 *             saved for the backward.
 *   backward: dy [tokens, C] (bf16 copy or fp32) -> dh1 [tokens, 4C] bf16 (operand of fc1's weight gradient) and
 *             dxn2 [tokens, C] bf16; dwd [4C, 9] / dbd [4C] accumulated (scratch: >= wm_leff_bwd_scratch_floats floats of
 *             per-strip partial sums, reduced by wm_reduce_jobs — deferred into 2 entries when `defer` is given; NULL = atomics). */
int32_t wm_leff_fusable(int32_t hw, int32_t C);
int wm_leff_fwd(const void* xn2, const float* x1, const void* w1_bf16, const float* b1, const float* wd, const float* bd,
                const void* w2_bf16, const float* b2, const float* rowscale /* [B] or NULL */, float* x2, void* h1, void* h2,
                int32_t B, int32_t hw, int32_t C, void* stream);
int64_t wm_leff_bwd_scratch_floats(int32_t B, int32_t hw, int32_t C);
int wm_leff_bwd(const void* dy, int32_t dy_bf16, const float* rowscale /* [B] or NULL */, const void* h1, const void* h2,
                const void* w1_bf16, const float* wd, const void* w2_bf16, void* dh1, void* dxn2, float* dwd, float* dbd,
                float* scratch, WmReduceJob* defer /* 2 entries or NULL */, int32_t B, int32_t hw, int32_t C, void* stream);

/* ---- data movement for Downsample (conv4x4 s2 p1, model.py:763) / Upsample (convT2x2 s2, :789) */
int wm_im2col_4x4s2(const float* x, void* cols, int32_t cols_bf16, int32_t B, int32_t hw, int32_t C, void* stream);
int wm_col2im_4x4s2(const void* dcols, int32_t dcols_bf16, float* dx, int32_t B, int32_t hw, int32_t C, void* stream);
/* The same layer WITHOUT the im2col image (csrc/conv_grid.hip; bf16 mode): the input as 2 x 2 pixel blocks on a grid of (hw/2 + 2)^2 rows
 * per image, so that the forward, dX and dW products are ordinary wm_gemm launches with grid addressing (WmGemm::a_grid_n, a_jump_k,
 * b_seg_cols) and these layout passes replace im2col / col2im:  x [B,hw,hw,C] fp32 -> xg [wm_conv4x4s2_xg_rows][4C] bf16;
 * dy [B (hw/2)^2, 2C] (fp32 or bf16) -> dyg [wm_conv4x4s2_dyg_rows][2C] bf16 (zero border, hw/2 + 3 zero rows in front);
 * dxg [wm_conv4x4s2_grid_rows][4C] bf16 -> dx [B,hw,hw,C] fp32;  weight [2C,C,4,4] fp32 -> w_fwd [2C,16C] and w_dx [4C,8C] bf16 in
 * block order; a weight gradient in w_fwd order added into grad [2C,C,4,4].                                                         */
int64_t wm_conv4x4s2_xg_rows(int32_t B, int32_t hw);
int64_t wm_conv4x4s2_dyg_rows(int32_t B, int32_t hw);
int64_t wm_conv4x4s2_grid_rows(int32_t B, int32_t hw);
int wm_conv4x4s2_x_to_grid(const float* x, void* xg, int32_t B, int32_t hw, int32_t C, void* stream);
int wm_conv4x4s2_dy_to_grid(const void* dy, int32_t dy_bf16, void* dyg, int32_t B, int32_t hw, int32_t C, void* stream);
int wm_conv4x4s2_grid_to_dx(const void* dxg, float* dx, int32_t B, int32_t hw, int32_t C, void* stream);
int wm_conv4x4s2_weights(const float* w, void* w_fwd, void* w_dx, int32_t C, void* stream);
int wm_conv4x4s2_dweight(const float* dw_fwd_order, float* grad, int32_t C, void* stream);
/* y4: [B*hw*hw, 4*Co] columns ordered (di,dj,o)  <->  dst[B,(2hw)^2, ldd] channels [c0, c0+Co) */
int wm_pixel_shuffle2(const void* y4, int32_t y4_bf16, const float* bias /* [Co] or NULL */, float* dst, int32_t B,
                      int32_t hw, int32_t Co, int64_t ldd, int32_t c0, void* stream);
int wm_pixel_unshuffle2(const float* dsrc, void* dy4, int32_t dy4_bf16, int32_t B, int32_t hw, int32_t Co, int64_t lds,
                        int32_t c0, void* stream);
/* dst[r, c0:c0+C] = src[r, s0:s0+C] (skip concat, model.py:1225; and its backward slice)      */
int wm_copy_cols(const float* src, int64_t lds, int32_t s0, float* dst, int64_t ldd, int32_t c0,
                 int64_t rows, int32_t C, int32_t accumulate, void* stream);
/* out[i0*so0+i1*so1+i2*so2] (=|atomic +=) in[i0*si0+i1*si1+i2*si2]: weight re-layouts for the conv GEMMs
 * ([o][c][tap] <-> [o][tap][c], ConvTranspose [c][o][tap] <-> [tap][o][c]) and their gradient fold-back */
int wm_strided3(const float* in, void* out, int32_t out_bf16, int32_t n0, int32_t n1, int32_t n2, int64_t si0,
                int64_t si1, int64_t si2, int64_t so0, int64_t so1, int64_t so2, int32_t atomic, void* stream);
/* bf16 copy of (a slice of) the fp32 parameter arena: the B operands of the forward / dX GEMMs in bf16 mode */
int wm_cast_bf16(const float* src, void* dst, int64_t n, void* stream);
/* dst[i] = sum_r float(src[r * n + i]), r < rows (n a multiple of 8): fp32 accumulation of the ranks' bf16-rounded gradient buckets
 * (data-parallel exchange in bf16: the reference is single-device; train.GradSync(exchange="bf16")) */
int wm_sum_bf16_rows(const void* src_bf16, float* dst, int32_t rows, int64_t n, void* stream);
/* Range-style audio_scale ("lo-hi") of the reference's data preparation and evaluation loop: min / max over the whole tile
 * tensor (normalize_batch, audio_test.py:33-55) left on the device, and the affine map of a tensor into [lo, hi] with those
 * statistics (audio_test.py:331-338, :698-702) or back (inverse != 0; audio_test.py:567-571).
 * scratch: >= wm_minmax_scratch_floats() floats, contents irrelevant; out_min_max / min_max: 2 floats (min, max). */
int64_t wm_minmax_scratch_floats(void);
int wm_minmax(const float* x, int64_t n, float* out_min_max, float* scratch, void* stream);
int wm_range_map(const float* x, float* out, const float* min_max, float lo, float hi, int32_t inverse, int64_t n, void* stream);
/* transposed bf16 copies of many [rows, cols] fp32 matrices of the parameter arena in one launch: the dX GEMMs of the
 * 256 x 256 tile kernel (csrc/gemm_big.hip) take their weight K-contiguous, i.e. W^T of nn.Linear's [out, in] weight
 * (model.py:460-471,549,700,711), and wm_attn_half_bwd takes Wqkv^T.  jobs (device, int32 x 4 per job, 16-byte aligned):
 * {element offset of the matrix in BOTH arenas, rows, cols, index of the job's first 64 x 64 tile}; a job has
 * ceil(rows / 64) * ceil(cols / 64) tiles; rows and cols are multiples of 4, offsets of 8. */
int wm_transpose_cast_jobs(const float* arena, void* arena_t_bf16, const int32_t* jobs, int32_t njobs, int32_t total_tiles,
                           void* stream);

/* ---- small direct convolutions (InputProj :814, OutputProj :846, stft_layer :2305,
 * ConvAutoencoder :1700-1709, EncoderTransformerWM.conv2 :1566).  Layout flags: 0 = NCHW,
 * 1 = token-major NHWC.  act: 0 none, 1 ReLU, 2 LeakyReLU(0.01), 3 sigmoid.                    */
typedef struct WmConv {
  int32_t B, Cin, H, W, Cout, KH, KW, SH, SW, PH, PW, OH, OW;
  int32_t in_nhwc, out_nhwc, act, transposed;   /* transposed: ConvTranspose2d k == stride, no pad */
} WmConv;
int wm_conv_fwd(const WmConv* c, const float* x, const float* w, const float* b, float* y, void* stream);
/* dy is the gradient w.r.t. the ACTIVATED output y; y is needed to undo the activation.
 * dx (optional) is written or accumulated; dw/db (optional) are atomically accumulated.         */
int wm_conv_bwd(const WmConv* c, const float* x, const float* w, const float* y, const float* dy,
                float* dx, int32_t dx_accumulate, float* dw, float* db, void* stream);
int wm_maxpool_fwd(const float* x, float* y, int32_t planes, int32_t H, int32_t W, int32_t kh, int32_t kw,
                   void* stream);
int wm_maxpool_bwd(const float* x, const float* dy, float* dx, int32_t planes, int32_t H, int32_t W,
                   int32_t kh, int32_t kw, int32_t accumulate, void* stream);

/* ---- data-prep STFT (audio_test.py:465-488,507-519): torch.stft(n_fft=256, hop 128, rectangular window, centre /
 * reflect) of U utterances of L samples each, bin 128 dropped, frames zero-padded to wm_stft256_tiles_per_utterance(L)
 * tiles of 128 frames (the reference always appends at least one padded frame column), times `scale`.
 * wave: [U, L], tiles: [U * tiles_per_utterance, 2, 128, 128] (re|im planes, [bin][frame]).                        */
int32_t wm_stft256_tiles_per_utterance(int64_t L);
int wm_stft256_tiles(const float* wave, int64_t L, int32_t U, float* tiles, float scale, void* stream);

/* ---- whole-utterance STFT pair of the evaluation path.
 * wm_stft255_tiles: SpeechDataTest.prepare_data (audio_test.py:313-343) and the re-analysis after the attack
 *   (:676-688): torch.stft(n_fft=255) (hop 63, rectangular, centre / reflect) of U utterances of L samples,
 *   wm_stft255_frames(L) = 1 + (L - 1) / 63 frames zero-padded to wm_stft255_tiles_per_utterance(L) tiles, times `scale`.
 *   wave: [U, L], tiles: [U * tiles_per_utterance, 2, 128, 128].  len_last_clip of the reference = frames % 128.
 * wm_istft255_stitch: reconstruct_audio (audio_test.py:594-600): the n_tiles tiles of an utterance concatenated along
 *   the frame axis, the last one trimmed to last_len frames, torch.istft(n_fft=255, length=L) (zero beyond the
 *   overlap-add length), times `out_scale` (the 1 / audio_scale factor of :561-563).  wave: [U, L].
 * wm_wave_attack: the state-free waveform attacks of the evaluation loop on whole utterances (rows N2 / N3), one per
 *   call as audio_test.py:628-660 dispatches them, or combined in this order: bit1 AWGN at snr_db
 *   (audio_attack.py:99-125; Philox noise), bit3 amplitude scaling (:55-58), bit4 echo addition out[i] += echo_gain *
 *   in[i - echo_delay] (:33-52), bit6 8-bit requantisation rint(clip(x, -1, 1) * 127) / 128 (the PCM_U8 write / read of
 *   :85-96), bit5 jittering_2: jitter_count uniformly drawn samples zeroed (:176-193).  power: [U] floats scratch. */
typedef struct WmWaveAttack {
  int32_t flags; float snr_db; float amp; float echo_gain; int64_t echo_delay; int32_t jitter_count; int32_t reserved;
  uint64_t seed;
} WmWaveAttack;
int32_t wm_stft255_frames(int64_t L);
int32_t wm_stft255_tiles_per_utterance(int64_t L);
int wm_stft255_tiles(const float* wave, int64_t L, int32_t U, float* tiles, float scale, void* stream);
int wm_istft255_stitch(const float* tiles, int32_t n_tiles, int32_t last_len, int32_t U, float* wave, int64_t L,
                       float out_scale, void* stream);
int wm_wave_attack(const WmWaveAttack* a, const float* wave, float* out, int64_t L, int32_t U, float* power, void* stream);
/* Whole-utterance 'low_pass' (audio_test.py:637 -> audio_attack.py:21-30: butter(8, 0.5) + filtfilt as two 128-tap FIR passes,
 * lp_h = taps | steady-state tails as in WmAttack; tmp: [U, L + 54] floats) and 'resampling' (:644 -> :71-83, 16k -> 8k -> 16k
 * with the 41-tap half-band filter rs_h, scipy resample_poly conventions: the librosa kernel is parity-unpinned; tmp: [U,
 * ceil(L / 2)] floats) on [U, L] waveforms of any length; out: [U, L].                                                    */
int wm_wave_lowpass(const float* wave, float* out, float* tmp, int64_t L, int32_t U, const float* lp_h, int32_t lp_n, void* stream);
int wm_wave_resample(const float* wave, float* out, float* tmp, int64_t L, int32_t U, const float* rs_h, int32_t rs_n, void* stream);
/* Phase-vocoder attacks (audio_attack.py:195-231 time_scaling = librosa.effects.time_stretch, :233-239 pitch_scaling =
 * librosa.effects.pitch_shift; librosa's published algorithm restated, PARITY UNPINNED): wm_time_stretch = STFT(2048 / 512,
 * Hann, reflect) -> phase vocoder at `rate` -> inverse STFT of length Lout (= round(L / rate) for time_scaling).  Caller-owned
 * work buffers: spec [U, 1025, wm_pv_frames(L) + 2] complex64, stretched [U, 1025, wm_pv_steps(L, rate)] complex64, frames
 * [U, wm_pv_steps(L, rate), 2048] floats.  wm_sinc_resample: band-limited interpolation by `ratio` (= target / source rate)
 * from a half filter table `win` and its forward differences `delta` of nwin entries, `precision` table samples per zero
 * crossing (resampy's 'kaiser_best' layout: 64 crossings x 512), out[t] for t < Lout.                                        */
int32_t wm_pv_frames(int64_t L);
int32_t wm_pv_steps(int64_t L, double rate);
int wm_time_stretch(const float* wave, int64_t L, int32_t U, double rate, void* spec, void* stretched, float* frames, float* out,
                    int64_t Lout, void* stream);
int wm_sinc_resample(const float* in, int64_t Lin, int32_t U, double ratio, const float* win, const float* delta, int32_t nwin,
                     int32_t precision, float* out, int64_t Lout, void* stream);

/* ---- in-graph STFT round trip — torch.istft / torch.stft with n_fft=255, hop 63, rectangular
 * window, center/reflect (model.py:2458,2463).  spec: [B,2,128,128] (re|im planes), wave: [B,8002] */
/* `tables`: wm_stft255_table_bytes(terms) bytes of caller-owned device memory (16-byte aligned) filled ONCE by
 * wm_stft255_tables(terms): the folded cos / sin DFT matrices as `terms` bf16 planes each, in MFMA operand order.  The
 * transforms run as split-precision bf16 MFMA products with fp32 accumulation:
 *   terms = 2  hi + lo operands, hi*hi + hi*lo + lo*hi: ~2e-6 of full scale (the bf16 compute mode);
 *   terms = 3  hi + mid + lo operands, six products: fp32-grade (~1e-7), the fp32 / split compute modes, whose outputs behind
 *              the extractor are held to 1e-4 of the CPU oracle.
 * `terms` of a transform call must be the value its tables were filled with; anything else than 2 or 3: WM_EINVAL.        */
int64_t wm_stft255_table_bytes(int32_t terms);
int wm_stft255_tables(void* tables, int32_t terms, void* stream);
int wm_istft255_fwd(const float* spec, float* wave, const void* tables, int32_t terms, int32_t B, void* stream);
int wm_stft255_fwd(const float* wave, float* spec, const void* tables, int32_t terms, int32_t B, void* stream);
int wm_istft255_bwd(const float* dwave, float* dspec, const void* tables, int32_t terms, int32_t B, void* stream);
int wm_stft255_bwd(const float* dspec, float* dwave, const void* tables, int32_t terms, int32_t B, void* stream);

/* ---- waveform attacks (audio_attack.py:21-30,55-58,71-83,99-125; torch awgn model.py:15-41).
 * One workgroup per waveform keeps the row in LDS through the whole chain.
 * flags bit0 resample 16k->8k->16k, bit1 AWGN(snr_db), bit2 low-pass (FIR-expanded Butterworth-8
 * filtfilt, taps `lp_h[lp_n]`), bit3 amplitude scale; then the SURVEY 8f row-N3 stages on the row: bit4 echo addition
 * (audio_attack.py:33-52: + echo_gain * row delayed by echo_delay samples), bit5 jittering_2 (:176-193: jitter_count
 * uniformly drawn samples zeroed), bit6 8-bit requantisation (:85-96: rint(clip(x,-1,1) * 127) / 128, straight-through
 * gradient).  bwd applies the adjoint chain.                                                                        */
typedef struct WmAttack {
  int32_t flags; float snr_db; float amp; const float* lp_h; int32_t lp_n;
  const float* rs_h; int32_t rs_n;          /* polyphase resampling prototype filter              */
  uint64_t seed; const int64_t* step;       /* Philox stream = (seed, *step, row)                 */
  int32_t echo_delay; float echo_gain; int32_t jitter_count; int32_t reserved;
  const void* lp_mfma;                      /* optional: wm_attack_fir_tables() image of lp_h (lp_n == 128): the two FIR   */
} WmAttack;                                 /* passes then run as split-precision bf16 MFMA products; NULL = scalar loops   */
/* Banded-Toeplitz operand tables of the 128-tap low-pass response for the MFMA form of the FIR passes:
 * wm_attack_fir_table_bytes() bytes of caller-owned device memory (16-byte aligned), filled once per filter.             */
int64_t wm_attack_fir_table_bytes(void);
int wm_attack_fir_tables(const float* lp_h, int32_t lp_n, void* tables, void* stream);
int wm_attack_fwd(const WmAttack* a, const float* wave, float* out, float* sigma_out, int32_t B, int32_t L,
                  void* stream);
int wm_attack_bwd(const WmAttack* a, const float* wave, const float* dout, const float* sigma, float* dwave,
                  int32_t B, int32_t L, void* stream);
/* z[B, L] = the standard-normal stream the AWGN stage (bit1) adds as sigma * z for (seed, *step): lets a caller replay
 * the noise of a step elsewhere (np.random.normal of audio_attack.py:120 is replaced by Philox4x32-10 + Box-Muller)   */
int wm_awgn_noise(uint64_t seed, const int64_t* step, float* z, int32_t B, int32_t L, void* stream);

/* ---- loss of audio_uformer_stft.py:463-482 and its gradients ------------------------------- */
/* losses[0..3] = l1..l4, [4] = total, [5] = ||noise||.  grads may alias nothing.               */
int wm_loss_fwd_bwd(const float* stft_new, const float* x, const float* noise, const float* wm_pred,
                    const float* wm, const float* msg, float* losses, float* d_stft_new, float* d_noise,
                    float* d_wm_pred, float* d_wm, int32_t B, float* scratch, void* stream);

/* ---- uformerWM/losses.py as device ops (imported by audio_uformer_stft.py:26): CharbonnierLoss (:41-52), TVLoss (:20-37),
 * tv_loss (:8-18) on NCHW fp32.  *loss += value (caller zeroes it); dx / dy (optional) = gscale * d loss / d input.      */
int wm_charbonnier_fwd_bwd(const float* x, const float* y, int64_t n, float eps, float* loss, float* dx, float* dy,
                           float gscale, void* stream);
int wm_tvloss_fwd_bwd(const float* x, int32_t N, int32_t C, int32_t H, int32_t W, float weight, float* loss, float* dx,
                      float gscale, void* stream);
int wm_tv_pow_fwd_bwd(const float* x, int32_t N, int32_t C, int32_t H, int32_t W, float beta, float reg_coeff, float* loss,
                      float* dx, float gscale, void* stream);

/* ---- optimizer / utilities ----------------------------------------------------------------- */
/* lr, step and grad_scale are read from DEVICE memory so a captured graph follows the scheduler */
int wm_adamw(float* p, const float* g, float* m, float* v, int64_t n, const float* lr, float beta1, float beta2,
             float eps, float wd, const int64_t* step, const float* grad_scale, void* stream);
int wm_step_inc(int64_t* step, void* stream);
/* *dst = *src on the stream: a forward pass keeps the step value its random draws (AWGN, jitter positions) were keyed with, so
 * that its backward replays the same draws even if another training forward has advanced the counter in between */
int wm_step_copy(const int64_t* src, int64_t* dst, void* stream);
int wm_droppath_scales(float* scales, const float* rates, int32_t nslots, int32_t B, uint64_t seed,
                       const int64_t* step, void* stream);
/* p[0..n) = 0 (16-byte aligned p): per-step clear of the gradient arena and of accumulation targets               */
int wm_zero(float* p, int64_t n, void* stream);
int wm_axpbyz(float* out, float a, const float* x, float b, const float* y /* or NULL */, int64_t n,
              void* stream);                                     /* out = a*x + b*y */
/* payload feature [B,4,64] tiled (1,16,8) into channels [0,512) of the bottleneck concat [B,64,1024]
 * (model.py:2389,2411); bwd ACCUMULATES the folded gradient into dfeat [B,256]                   */
int wm_msg_expand_fwd(const float* feat, float* dst, int32_t B, void* stream);
int wm_msg_expand_bwd(const float* dcat, float* dfeat, int32_t B, void* stream);

/* ---- layers of the small CNN models: ModelA (model.py:3000-3066, train_modelA.py:423-445) and the HiDDeN decoder
 * (hidden/model/decoder.py:6-40, conv_bn_relu.py:3-18); their convolutions / pools use wm_conv_* / wm_maxpool_*.
 * BatchNorm2d on NCHW fp32 with a fused activation (act: 0 none, 1 ReLU, 2 LeakyReLU(slope), 3 sigmoid).
 * training != 0: batch statistics (biased variance for the normalisation; running_mean/var, if given, updated with
 * `momentum` and the unbiased variance, as nn.BatchNorm2d does); training == 0: the running statistics.
 * save_mean / save_invstd [C] are outputs kept for the backward; bwd ACCUMULATES dgamma / dbeta, `sums` is [2C] scratch. */
int wm_batchnorm2d_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                       float* y, float* save_mean, float* save_invstd, int32_t B, int32_t C, int32_t HW, float eps,
                       float momentum, int32_t training, int32_t act, float slope, void* stream);
int wm_batchnorm2d_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* save_mean,
                       const float* save_invstd, float* dx, float* dgamma, float* dbeta, float* sums, int32_t B, int32_t C,
                       int32_t HW, int32_t training, int32_t act, float slope, void* stream);
/* y = x * mask / (1 - p), mask = Philox(seed, offset, index) < 1 - p (nn.Dropout); the backward is the same call on dy  */
int wm_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream);
/* *loss += mean((a-b)^2);  da = gscale * 2 (a-b) / n, db = -da (each NULL or output) — nn.MSELoss terms of
 * train_modelA.py:435-445 and hidden's decoder loss                                                               */
int wm_mse_fwd_bwd(const float* a, const float* b, int64_t n, float* loss, float* da, float* db, float gscale,
                   void* stream);

/* ---- HiDDeN encoder / discriminator / adversarial loss.  hidden/options.py:19-50 configures them (encoder_blocks,
 * encoder_channels, discriminator_blocks, discriminator_channels, adversarial_loss) and hidden/utils.py:85-108 saves and
 * restores `encoder_decoder` / `discriminator`, but the modules themselves are absent from the reference tree; these
 * entry points serve the architecture of the HiDDeN paper (Zhu et al. 2018, section 3) built from ConvBNRelu blocks:
 *   wm_plane_fill   dst[b][c0 + l][:] = scale * v[b][l]: the message replicated over H x W as L planes of the concat
 *                   buffer [B, Ctot, H, W] (encoder), and the backward of AdaptiveAvgPool2d(1) (scale = 1 / HW)
 *   wm_plane_sum    v[b][l] = scale * sum of plane (b, c0 + l): AdaptiveAvgPool2d(1) (discriminator) and the message
 *                   gradient
 *   wm_bce_logits_fwd_bwd  nn.BCEWithLogitsLoss against a constant target (cover = 1, encoded = 0): *loss += mean,
 *                   dlogits = gscale * (sigmoid(x) - target) / n (each NULL or output)
 *   wm_linear_small_*  the Linear(channels, 1) head on the pooled features [M, K] (N <= K outputs; wm_gemm does not take an
 *                   inner dimension of 1): y = x w^T + b; bwd: dx (NULL or output), dw / db ACCUMULATE (NULL to skip)       */
int wm_plane_fill(const float* v, float* dst, int32_t B, int32_t L, int32_t Ctot, int32_t c0, int32_t HW, float scale,
                  void* stream);
int wm_plane_sum(const float* src, float* v, int32_t B, int32_t L, int32_t Ctot, int32_t c0, int32_t HW, float scale,
                 void* stream);
int wm_bce_logits_fwd_bwd(const float* logits, int32_t n, float target, float* loss, float* dlogits, float gscale,
                          void* stream);
int wm_linear_small_fwd(const float* x, const float* w, const float* b, float* y, int32_t M, int32_t N, int32_t K,
                        void* stream);
int wm_linear_small_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, int32_t M, int32_t N,
                        int32_t K, void* stream);

/* ---- HiDDeN image-domain distortions on [B, C, H, W] fp32 tiles (hidden/noise_layers/).  The random
 * rectangles / ratios / masks are drawn by the caller as the reference draws them (numpy) and passed in.  backward != 0
 * selects the adjoint / gradient form of the same entry point.
 *   wm_crop2d          crop.py:48-74: dst[p, i, j] = src[p, h0 + i, w0 + j] (oh x ow); backward: src = dy [p, oh, ow] -> dst = dx
 *   wm_blend2d         cropout.py:6-27 (rectangle [h0, h1) x [w0, w1)) / dropout.py:5-29 (mask[H * W] != 0): out0 = sel ? a : b;
 *                      backward: a = dy, out0 = d/da (or NULL), out1 = d/db (or NULL)
 *   wm_bernoulli_mask  device-side keep mask for Dropout (Philox; the numpy stream of dropout.py:22 is the compatible default)
 *   wm_resize_nearest  resize.py:6-26 F.interpolate(mode='nearest'): src index = min(floor(dst * scale), in - 1), scale =
 *                      float(1 / ratio); backward ACCUMULATES dy into a zeroed dx
 *   wm_quantization_*  quantization.py:17-45: rescale to [0, 255], Fourier soft rounding, rescale to the input's range; f: [n]
 *                      scratch kept for the backward, state: wm_quantization_state_bytes() bytes kept for the backward
 *   wm_jpeg_mask       jpeg_compression.py:65-160 on [B, 3, H, W]: RGB -> YUV, 8x8 DCT, zig-zag keep masks, inverse, YUV -> RGB;
 *                      adjoint != 0: the adjoint map (gradient with respect to the input)                                  */
int wm_crop2d(const float* src, float* dst, int32_t planes, int32_t H, int32_t W, int32_t h0, int32_t w0, int32_t oh, int32_t ow,
              int32_t backward, void* stream);
int wm_blend2d(const float* a, const float* b, const float* mask, float* out0, float* out1, int32_t planes, int32_t H, int32_t W,
               int32_t h0, int32_t h1, int32_t w0, int32_t w1, int32_t backward, void* stream);
int wm_bernoulli_mask(float* mask, int32_t n, float keep, uint64_t seed, uint64_t offset, void* stream);
int wm_resize_nearest(const float* src, float* dst, int32_t planes, int32_t H, int32_t W, int32_t OH, int32_t OW, float scale_h,
                      float scale_w, int32_t backward, void* stream);
int64_t wm_quantization_state_bytes(void);
int wm_quantization_fwd(const float* x, float* f, float* out, int64_t n, void* state, void* stream);
int wm_quantization_bwd(const float* g, const float* x, const float* f, float* dx, int64_t n, void* state, void* stream);
int wm_jpeg_mask(const float* x, float* y, int32_t B, int32_t H, int32_t W, int32_t keep_y, int32_t keep_u, int32_t keep_v,
                 int32_t adjoint, void* stream);

#ifdef __cplusplus
}
#endif
#endif
