/* mumpy_hip.h — C ABI of libmumpy_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * hot path of Mumpy (Multilateral Temporal-view Pyramid Transformer): the forward (inference) entry points first, the
 * training entry points (loss, optimizer, backward kernels) at the end.
 *
 * The reference is pure Python/PyTorch and has no FFI of its own; the "interface" each entry point
 * replaces is therefore the reference Python operator it computes, cited as file:line below
 * (paths relative to the reference repo).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions (all entry points)
 *   - plain C, no C++/torch types; every pointer is a DEVICE pointer to fp32 unless stated otherwise.
 *   - the CALLER allocates and owns every buffer including scratch; nothing here allocates, frees or
 *     synchronises.  Work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream),
 *     so calls are capturable into a hipGraph.
 *   - return 0 on success; a negative MUMPY_E* code for a rejected argument (nothing is launched);
 *     a positive value is the hipError_t of a failed launch.  mumpy_last_error() returns a
 *     thread-local message for the last non-zero return.  No global mutable state: reentrant.
 *   - layouts are row-major with the last index contiguous; "token-major" means (tokens, channels).
 *   - window size is 7 (49 tokens) and attention head width is 32 throughout (factory:26; hidden/heads).
 */
#ifndef MUMPY_HIP_H
#define MUMPY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MUMPY_ABI_VERSION 2

#define MUMPY_EINVAL   (-1) /* bad shape / size                              */
#define MUMPY_EALIGN   (-2) /* pointer not 16-byte aligned where required    */
#define MUMPY_ENULL    (-3) /* required pointer is NULL                      */
#define MUMPY_ERANGE   (-4) /* size exceeds what the kernel was built for    */

/* activation selector for mumpy_linear_fwd */
#define MUMPY_ACT_NONE 0
#define MUMPY_ACT_GELU 1    /* exact erf GELU, as nn.GELU() */
/* OR-ed into `act` of the linear / conv entry points: round x and W to bf16 while staging and multiply on the bf16
 * MFMA (fp32 accumulate, fp32 bias/activation/residual, fp32 tensors in memory).  Config 3's matrix arithmetic; the
 * default (flag absent) is exact fp32 on v_mfma_f32_32x32x2_f32. */
#define MUMPY_MATH_BF16 0x100
/* OR-ed into `act` likewise (exclusive with MUMPY_MATH_BF16): fp32 products on the bf16 matrix pipe.  Each fp32 operand is
 * split while staged into three bf16 pieces (24+ mantissa bits kept) and the six piece products of weight >= 2^-16 are
 * accumulated in fp32; the dropped terms are <= 2^-24 |a||b|, one fp32 rounding.  fp32-level accuracy at 6/16 of the
 * fp32 MFMA time.  Operands outside bf16's exponent handling (inf, NaN, |v| < 2^-110) are not split faithfully. */
#define MUMPY_MATH_BF16X3 0x200
/* Likewise with TWO pieces per operand (16 mantissa bits kept) and the three piece products a0b0 + a0b1 + a1b0: operand
 * precision 2^-17, i.e. TF32-class and better (TF32 keeps 11 bits; gfx950 has no xf32 MFMA) at 3/16 of the fp32 MFMA time.
 * A reduced-precision mode like MUMPY_MATH_BF16 (error ~1e-5 relative instead of ~1e-2), never the default. */
#define MUMPY_MATH_BF16X2 0x400

int         mumpy_abi_version(void);
const char* mumpy_last_error(void);
/* 1: diagnostics build (`make TUNING=1`): planner / kernel A/B hooks read MUMPY_* environment variables.  0: the shipped library,
 * which reads no environment variable and keeps no mutable global state besides the thread-local error string and the cached
 * compute-unit count of the first device used. */
int         mumpy_tuning_build(void);

/* Sticky status word of a workspace kept for mumpy_linear_wsz_fwd / mumpy_linear_lnx_fwd (the one call of this library that
 * synchronises: it copies the word back).  *status = 0: fine; b + 1: a split ("stream-K") launch gave up waiting for workgroup b's
 * part -- its output is incomplete; discard it and re-zero the workspace's first 4096 bytes before the next launch;
 * -1: a LayerNorm-folding consumer (mumpy_linear_lnx_fwd) met a row whose |mean| exceeds 256 standard deviations: results are
 * complete but that row carries ~1e-4 relative error instead of ~1e-6 (the product is taken on the un-centred x) -- take the
 * two-launch route (mumpy_layernorm_fwd + mumpy_linear_wsz_fwd) for such data; the word is sticky until re-zeroed. */
int mumpy_workspace_status(const void* workspace, int* status);

/* ---- "background" kernels: NO LDS, so that they can be resident beside the persistent GEMM (round 3) ----
 * The persistent GEMM behind mumpy_linear_wsz_fwd holds every CU's whole LDS; a kernel that allocates any LDS runs strictly after
 * it, an LDS-free one overlaps (tools/coresidency_probe.py).  For work forked beside a chain of large GEMMs (views 1 / 2 beside
 * view 3 inside a pyramid stage, mTVE:445-450) the caller may ask for these forms; results equal the regular entry points'.
 * mumpy_linear_rd_fwd: y = act(x W^T + bias) + residual, fp32 MFMA, operands straight from global memory into registers; no
 * workspace, no planner: slower than mumpy_linear_fwd when it runs alone.  K % 32 == 0, N % 32 == 0. */
int mumpy_linear_rd_fwd(const float* x, const float* W, const float* bias, const float* residual, float* y, int64_t M, int N, int K,
                        int act, void* stream);
/* mumpy_window_attention_fwd without LDS (token tables in registers, bias rows from L1). */
int mumpy_window_attention_bg_fwd(const float* qkv, float* out, const float* bias, const float* mask_tab, const int32_t* mask_id,
                                  int n_mask, int B, int Hs, int W, int C, int shift, float scale, void* stream);

/* ---- LayerNorm folded into the GEMMs either side of it (round 3; swin:266,305: x + f(norm(x)) chains) ----
 * A pre-norm block computes  y = act(LayerNorm(x) W^T + b)  right after a residual GEMM produced x.  Instead of a LayerNorm
 * launch (one read + one write of x between two GEMMs):
 *   producer  (stats_out != NULL): the residual GEMM's epilogue also writes, per output row and 128-column tile, {mean_t, M2_t} of
 *             the values it stores -- stats_out (M, ceil(N/128), 2), two-pass inside the tile (no cancellation);
 *   consumer  (ln_stats != NULL): x is the RAW x, W is W diag(gamma) (caller-prepared), bias is W beta + b, ln_colsum[n] =
 *             sum_k W[n][k] gamma[k]; the epilogue finishes  rstd (acc - mean colsum) + bias  with mean / rstd of the row combined
 *             from the producer's ln_gn = ceil(K/128) partials (Chan's parallel update: stable), eps = ln_eps; no residual.
 * Both run only on the persistent 128x128 kernel: mumpy_linear_ln_tiles(M,N,K) > 0 says a shape takes it (and is the tile count);
 * otherwise MUMPY_EINVAL.  workspace: the kept, zero-initialised workspace of mumpy_linear_wsz_fwd (required).  fp32 math only. */
int mumpy_linear_ln_tiles(int64_t M, int N, int K);
int mumpy_linear_lnx_fwd(const float* x, const float* W, const float* bias, const float* residual, float* y, int64_t M, int N, int K,
                         int act, void* workspace, int64_t workspace_bytes, float* stats_out, const float* ln_stats, int ln_gn,
                         const float* ln_colsum, float ln_eps, void* stream);

/* ---- SwinDAttention with the index work folded into its GEMMs (round 3; csrc/cva_fused.hip) ----
 * kv = [proj_k | proj_v](grid_sample(x2, pos)) (deform:353-362) in ONE launch: the bilinear sampling (align_corners=True, zeros
 * padding, per-group positions, kv window i pairs with q window i mod nq) is the A-operand loader of the projection; the sampled
 * map is never written.  x2 (B, Hs2*W, C) raster (already through `pre`), pos (nq,3,49,2) from mumpy_deform_offsets_fwd,
 * Wkv (2C, C) = [W_k; W_v], bkv (2C), kv (B * Hs2/7 * W/7, 49, 2C).  C in {96,192,384,768}. */
int mumpy_deform_sample_kv_fwd(const float* x2, const float* pos, const float* Wkv, const float* bkv, float* kv, int B, int Hs2,
                               int W, int C, int nq, void* stream);

/* out = x1 + x1[window order] + reshape_(C,49)->(49,C)(proj_out(o)) (deform:402-403; mTVE:138,285-286) in ONE launch: the
 * projection is computed transposed so that the un-permuted reshape lands on coalesced stores and the caller's two residual
 * terms ride in the epilogue.  o (B * H/7 * W/7, 49, C) from mumpy_deform_attention_fwd, Wout (C, C), bout (C), x1 / out
 * (B, H*W, C) raster; out must not alias x1.  Replaces a mumpy_linear_fwd + mumpy_deform_combine_fwd pair. */
int mumpy_deform_out_combine_fwd(const float* o, const float* Wout, const float* bout, const float* x1, float* out, int B, int H,
                                 int W, int C, void* stream);

/* ---- data-movement / wiring kernels of the decoder and the encoder tail (round 3: formerly ATen launches) ----
 * 2x2 average pooling, stride 2 (the nn.AvgPool2d(2) of decoder.py:149-178's frequency blocks): x (B,H,W,C) NHWC -- or, nchw_in
 * = 1, (B,C,H,W) contiguous as FAF emits it (dct:79) -- -> out (B,H/2,W/2,Cpad) NHWC with channels [C,Cpad) zero (the
 * implicit-GEMM convolution wants Cin % 32 == 0).  H, W even; Cpad % 4 == 0; NHWC input: C % 4 == 0. */
int mumpy_avgpool2_pad_nhwc_fwd(const float* x, float* out, int B, int H, int W, int C, int Cpad, int nchw_in, void* stream);

/* rows x cols floats between row pitches (floats): a channel slice of a concatenated NHWC map (decoder.py:197,210,213) or the
 * first three temporal slices of the global tokens (mTVE:745).  cols, pitches % 4 == 0. */
int mumpy_copy_rows_fwd(const float* src, int64_t src_stride, float* dst, int64_t dst_stride, int64_t rows, int cols, void* stream);

/* merge_views_along_channel_axis + the (b t n c) -> (b n) t c regrouping in front of the global embedding (mTVE:710-718,739):
 * view k is (B, t_k * n, C_k), frames stacked on rows, t_k in {1, T}; out ((B n T), C1+C2+C3) row (b, site, t) =
 * [v1[b, t or 0, site] | v2[...] | v3[...]]. */
int mumpy_merge_views_fwd(const float* v1, const float* v2, const float* v3, float* out, int B, int T, int n, int C1, int C2,
                          int C3, int t1, int t2, int t3, void* stream);

/* decoder.py:198-205: z = gcn * freq + PixelShuffle(2)(g * f); g, f (B,h,w,4C), gcn, freq, z (B,2h,2w,C), NHWC. */
int mumpy_trunk_head_fwd(const float* g, const float* f, const float* gcn, const float* freq, float* z, int B, int h, int w, int C,
                         void* stream);

/* ---- LayerNorm over the last dim (nn.LayerNorm, eps 1e-5, affine)  — swin:266,305; blocks:86-88 ----
 * x, y: (rows, C) token-major; y may alias x.  C % 4 == 0, C <= 4096. */
int mumpy_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                        int64_t rows, int C, float eps, void* stream);

/* Same with y written as bf16 (config 3's activation storage; statistics and arithmetic stay fp32). */
int mumpy_layernorm_bf16_fwd(const float* x, const float* gamma, const float* beta, void* y,
                             int64_t rows, int C, float eps, void* stream);

/* ---- Linear: y = act(x @ W^T + bias) + residual   — nn.Linear at swin:142,164,46-49; blocks:57-71,27-33;
 *      1x1 convs of deform:333,361,362,402; mTVE:283 (pre), mTVE:740 (globalembedding); swin:365 (reduction).
 * x (M,K), W (N,K) [nn.Linear weight layout], bias (N) or NULL, residual (M,N) or NULL, y (M,N).
 * fp32 in, fp32 accumulate on v_mfma_f32_32x32x2_f32.  K % 32 == 0, N % 32 == 0.  y may alias residual. */
int mumpy_linear_fwd(const float* x, const float* W, const float* bias, const float* residual, float* y,
                     int64_t M, int N, int K, int act, void* stream);

/* Same, with a caller-owned scratch buffer that lets small-M / deep-K shapes split K across workgroups (partial sums
 * to the scratch slab, combined in a fixed order by a second kernel: bitwise reproducible).  Size it with
 * mumpy_linear_workspace_bytes(M,N,K) (0 = this shape does not split); a NULL / too-small workspace just disables
 * the split.  The workspace must not be shared by launches that may overlap. */
int64_t mumpy_linear_workspace_bytes(int64_t M, int N, int K);
int mumpy_linear_ws_fwd(const float* x, const float* W, const float* bias, const float* residual, float* y,
                        int64_t M, int N, int K, int act, void* workspace, int64_t workspace_bytes, void* stream);

/* Same for a workspace the caller KEEPS across calls: its first 4096 bytes must be zero on entry (zero them once after
 * allocating it) and are zero again on exit -- the persistent kernel's arrival flags clean up after themselves -- which
 * saves the reset node the _ws_ form enqueues in front of a split launch.  One such workspace per stream: it must not be
 * shared by launches that may overlap. */
int mumpy_linear_wsz_fwd(const float* x, const float* W, const float* bias, const float* residual, float* y,
                         int64_t M, int N, int K, int act, void* workspace, int64_t workspace_bytes, void* stream);

/* bf16 STORAGE (BASELINE config 3 as written: bf16 weights and activations in HBM, fp32 accumulate): x (M,K) and W (N,K) are
 * bf16 (the caller keeps bf16 copies of the nn.Linear weights), bias and residual fp32, y bf16 (out_bf16 != 0) or fp32 (the
 * residual stream stays fp32).  Products on v_mfma_f32_32x32x16_bf16; no split-K.  The reference has no bf16 path: the
 * tolerance against its fp32 forward is build-defined (2e-2 relative on the logits). */
int mumpy_linear_bf16s_fwd(const void* x, const void* W, const float* bias, const float* residual, void* y,
                           int64_t M, int N, int K, int act, int out_bf16, void* stream);

/* Same, with the rows of x grouped in blocks: row m lives at x + (m / rows_per_block) * block_stride +
 * (m % rows_per_block) * K (floats).  Feeds one time slice of the (B, T*n, C) view-3 tokens to the decoder's
 * Conv3d(k=(T,1,1)) heads (decoder.py:98-120) as a (B*n, C) operand without the merge/permute copy of decoder.py:43-53;
 * the T slices are chained through `residual`. */
int mumpy_linear_rows_fwd(const float* x, int64_t rows_per_block, int64_t block_stride, const float* W,
                          const float* bias, const float* residual, float* y, int64_t M, int N, int K, int act,
                          void* workspace, int64_t workspace_bytes, void* stream);

/* Rows mode with a SEGMENTED contraction index: row (blk, r) of the A operand is K / kseg segments of kseg floats, segment j at
 * x + blk * block_stride + j * kstride + r * kseg.  E.g. the (B, T, n, C) token tensor as the (B n) x (T C) operand of a
 * Conv3d(k = s = (T,1,1)) head (decoder.py:62-66): rows_per_block = n, block_stride = T n C, kseg = C, kstride = n C, W =
 * the kernel re-laid (Cout, T, C).  kseg % 32 == 0, K % kseg == 0.  One launch instead of T chained mumpy_linear_rows_fwd. */
int mumpy_linear_rows_kseg_fwd(const float* x, int64_t rows_per_block, int64_t block_stride, int kseg, int kstride, const float* W,
                               const float* bias, const float* residual, float* y, int64_t M, int N, int K, int act,
                               void* workspace, int64_t workspace_bytes, void* stream);

/* ---- Convolution, NHWC, stride 1, zero "same" padding, odd kernel: y = act(conv(x, w) + bias) + residual
 *      — the decoder's 3x3 / 7x1 / 1x7 nn.Conv2d (decoder.py:9,24-31,68-95,149-178) as an implicit GEMM on the
 *      mumpy_linear_fwd tile machinery (K index = (tap, channel); borders predicated, nothing is unfolded).
 * x (B,H,W,Cin), w_krsc (Cout,kh,kw,Cin) = the channels_last image of the nn.Conv2d weight, y/residual (B,H,W,Cout).
 * Cin % 32 == 0, Cout % 32 == 0.  Workspace as for mumpy_linear_ws_fwd (mumpy_conv2d_workspace_bytes). */
int64_t mumpy_conv2d_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw);
int mumpy_conv2d_nhwc_fwd(const float* x, const float* w_krsc, const float* bias, const float* residual, float* y,
                          int B, int H, int W, int Cin, int Cout, int kh, int kw, int act, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* ---- Swin window attention core  — swin:54-83 (partition/reverse), 273,295 (roll), 145-163 (softmax(QK^T)V)
 * qkv:  (B, Hs*W, 3*C) raster token order over the stacked grid Hs = t*H rows by W columns; channel
 *       layout [q|k|v][head][32] exactly as nn.Linear(C,3C) emits it (swin:142).
 * out:  (B, Hs*W, C) raster order: attention output BEFORE the output projection, already un-shifted /
 *       window-reversed (the gather, cyclic shift, scatter are folded into the addressing; nothing is
 *       materialised).
 * bias: (nH,64,64) relative-position bias expanded by the caller from relative_position_bias_table
 *       [relative_position_index] (swin:148-151), laid out [head][query i][key j], zero-padded rows
 *       i>=49 and filled with -1e30 for key columns j>=49 (that is what masks the 49->64 padding).
 * mask_tab: (nU,64,64) distinct attn_mask patterns padded the same way (0 / -100, swin:252), or NULL;
 * mask_id:  (n_mask) int32 pattern index, -1 = no mask; or NULL.  Window bw (batch-major partition order) uses
 *           mask_id[bw % n_mask] — the broadcast of swin:155 (n_mask = nW = Hs/7 * W/7 windows per image).
 * shift: cyclic shift (0 or 3).  scale is applied to q before QK^T (swin:145). */
int mumpy_window_attention_fwd(const float* qkv, float* out, const float* bias, const float* mask_tab,
                               const int32_t* mask_id, int n_mask, int B, int Hs, int W, int C, int shift,
                               float scale, void* stream);

/* Same with qkv and out stored as bf16 (bias / mask tables fp32, fp32 arithmetic). */
int mumpy_window_attention_bf16_fwd(const void* qkv, void* out, const float* bias, const float* mask_tab,
                                    const int32_t* mask_id, int n_mask, int B, int Hs, int W, int C, int shift,
                                    float scale, void* stream);

/* ---- Deformable cross-view attention (SwinDAttention, deform:324-405) — four kernels -------------- */

/* offsets: q (B, H*W, C) raster, one frame per batch entry (t=1), gathered per 7x7 window.
 * Computes, per q-window and group g<3: depthwise 5x5 conv (pad 2) -> LayerNorm(Cg) -> GELU -> 1x1 conv
 * to 2 -> tanh * (2/7) + reference points (deform:334-349, 311-322).
 * pos: (B*nWf, 3, 49, 2) fp32 as (y,x) in [-1,1] grid units.  dw_w (Cg,25), dw_b (Cg), ln_g/ln_b (Cg),
 * pw_w (2,Cg).  Cg = C/3 <= 256. */
int mumpy_deform_offsets_fwd(const float* q, const float* dw_w, const float* dw_b, const float* ln_g,
                             const float* ln_b, const float* pw_w, float* pos, int B, int H, int W, int C,
                             void* stream);

/* sampling: x2 (B, Hs2*W, C) raster kv tokens (already through `pre`, mTVE:283), windows n2 = 0..B*nW2-1 in
 * partition order; kv window n2 uses the offsets of q-window (n2 mod nq) (x1.repeat, deform:330).
 * Bilinear, align_corners=True, zero padding (deform:353-356).  out: (B*nW2, 49, C) window-major. */
int mumpy_deform_sample_fwd(const float* x2, const float* pos, float* out, int B, int Hs2, int W, int C,
                            int nq, void* stream);

/* attention + aggregation: q (B, H*W, C) raster (proj_q output); kv (B2w, 49, 2*C) window-major
 * [k|v] (proj_k, proj_v of the sampled map); r = B2w / B1w.  For each output window b1 and head:
 *   o[b1] = sum_{t<r} softmax(q[(b1*r+t) mod B1w] k[b1*r+t]^T * scale) v[b1*r+t]     (deform:360-395)
 * out: (B1w, 49, C) window-major.  padmask: (1,64,64) with 0 for j<49 and -1e30 for j>=49. */
int mumpy_deform_attention_fwd(const float* q, const float* kv, const float* padmask, float* out, int B, int H,
                               int W, int C, int r, float scale, void* stream);

/* combine (deform:403 un-permuted reshape + mTVE:138 + mTVE:285-286):
 *   out[b, n*49+p, c] = x1[b, n*49+p, c] + x1[b, raster(n,p), c] + Yt[b*nWf+n][(p*C+c) % 49][(p*C+c) / 49]
 * x1, out: (B, H*W, C) (out must not alias x1); Yt: (B*nWf, 49, C) = proj_out output, window-major. */
int mumpy_deform_combine_fwd(const float* x1, const float* Yt, float* out, int B, int H, int W, int C,
                             void* stream);

/* ---- FAF: DCT band-pass features of ONE frame per clip  — dct:71-79, mTVE:734 ------------------------
 * x: (B,T,3,224,224); D, Dt: (224,224) DCT-II matrix and its transpose (built in fp64 then cast, dct:42-45,60);
 * out: (B,9,224,224), channel = band*3 + rgb, bands low/mid/high = i+j in [0,lo_hi], [mid_lo,mid_hi],
 * [224,448] (dct:66-68).  scratch: B*3*224*224 floats (the spectrum). */
int mumpy_faf_fwd(const float* x, const float* D, const float* Dt, float* scratch, float* out, int B, int T,
                  int frame, int lo_hi, int mid_lo, int mid_hi, void* stream);

/* ---- Tokenizer: Conv3d(3->C, k=s=(t,4,4)) + LayerNorm as implicit GEMM  — mTVE:605-618 ---------------
 * x: (B,T,3,H,W); Wt: (K,C) = conv weight (C,3,t,4,4) flattened to (C,K) and transposed, K = 48*t;
 * out: (B, t_out*(H/4)*(W/4), C) with t_out = (T - t)/t + 1, frames stacked on the token axis. */
int mumpy_patch_embed_fwd(const float* x, const float* Wt, const float* bias, const float* gamma,
                          const float* beta, float* out, int B, int T, int H, int W, int t, int C,
                          float eps, void* stream);

/* ---- Patch merging front half: 2x2 gather + LayerNorm(4C)  — swin:355-364 ---------------------------
 * x: (B, Hs*W, C) on the stacked grid; out: (B, Hs/2*W/2, 4C) with channel blocks
 * [(0,0),(1,0),(0,1),(1,1)] (row offset, col offset).  The Linear(4C->2C) is mumpy_linear_fwd. */
int mumpy_patch_merge_ln_fwd(const float* x, const float* gamma, const float* beta, float* out, int B, int Hs,
                             int W, int C, float eps, void* stream);

/* ---- Temporal attention of the global ViT blocks  — blocks:57-71 under vmap(in_dims=2), mTVE:741 -----
 * qkv: (S, T, 3*C) with S = B*49 spatial sites, heads of width 64; out: (S, T, C).  T <= 16. */
int mumpy_temporal_attention_fwd(const float* qkv, float* out, int64_t S, int T, int C, int heads,
                                 float scale, void* stream);
/* The same with queries for the first Tq <= T temporal tokens only: out (S, Tq, C); keys / values are all T tokens.  The encoder tail
 * keeps temporal slices 0..2 (mTVE:745), so the last global block computes nothing for t >= 3. */
int mumpy_temporal_attention_q_fwd(const float* qkv, float* out, int64_t S, int T, int Tq, int C, int heads, float scale, void* stream);

/* Attention MAPS for the `return_attention=True` variants (blocks:85-87; deform:364-396; mTVE:134-137) -- a visualisation path:
 * out (outer, heads, nq, nk) = softmax(scale * q k^T) per (outer, head); q row i of unit (o, h) at q + (o % q_mod) * q_outer_stride +
 * i * q_row_stride + h * d, k row j at k + o * k_outer_stride + j * k_row_stride + h * d (strides in floats); nq, nk, d <= 64. */
int mumpy_attention_probs_fwd(const float* q, const float* k, float* out, int64_t outer, int heads, int nq, int nk, int d,
                              int64_t q_outer_stride, int64_t q_row_stride, int64_t k_outer_stride, int64_t k_row_stride,
                              int64_t q_mod, float scale, void* stream);

/* ---- Decoder glue in NHWC (decoder.py:67-225): everything between two convolutions ---------------------------
 * gn_stats: x (B,HW,C) NHWC -> partial (B, nsplit, G, 2) = per-slice {sum, sum of squares} of each GroupNorm group
 * (nn.GroupNorm statistics, decoder.py:70,77,84,91,101-119,151-180), combined in fixed order by gn_apply. */
int mumpy_gn_stats_nhwc_fwd(const float* x, float* partial, int B, int64_t HW, int C, int G, int nsplit,
                            void* stream);

/* gn_apply_resample: out = epilogue( resample( mean4( act( GroupNorm(x) ) ) ) ), one pass, NHWC.
 *   partial/gamma/beta/G/eps: GroupNorm (partial == NULL: no normalisation, pure resample);
 *   act: 0 none, 1 ReLU, 2 Sigmoid;  mean4: 1 = PixelShuffle(2)+AvgPool(2) == mean over channel quadruples
 *   (DAP, decoder.py:140-143; commutes with the bilinear upsample);  scale 1|2|4 bilinear, align_corners as in
 *   nn.Upsample (True: decoder.py:72-93; False: decoder.py:10,136-137);
 *   ep_mode 0: none, 1: + ep_a*ep_b (decoder.py:219-220), 2: * ep_a (decoder.py:221, 14), operands dense
 *   (B,Ho,Wo,Cout);  out: (B,Ho,Wo,out_ctot) written at channel offset out_coff (lets a torch.cat operand be
 *   produced in place, decoder.py:210,213).  C <= 256. */
int mumpy_gn_apply_resample_nhwc_fwd(const float* x, const float* partial, int nsplit, const float* gamma,
                                     const float* beta, int G, float eps, int act, int mean4, int scale,
                                     int align_corners, int ep_mode, const float* ep_a, const float* ep_b,
                                     float* out, int out_ctot, int out_coff, int B, int H, int W, int C,
                                     void* stream);

/* final_out (decoder.py:95,223; BaselineDecoder decoder.py:275): Conv2d(C -> 1, 3x3, pad 1) on x (B,H,W,C) NHWC with
 * C a multiple of 32 (32 for Decoder, 256 for BaselineDecoder), w_krsc (1,3,3,C), bias (1); logits (B,1,H,W) fp32;
 * mask (B,1,H,W) uint8 = sigmoid(logit) > thr, or NULL (the eval tail of test.py:100-108 fused). */
int mumpy_final_conv_fwd(const float* x, const float* w_krsc, const float* bias, float* logits, uint8_t* mask, int B,
                         int H, int W, int C, float thr, void* stream);

/* Backward of mumpy_final_conv_fwd for C = 32 (the three-view Decoder's final_out under loss.backward(), decoder.py:95,225):
 * x (B,H,W,32) NHWC, w_krsc (1,3,3,32), dy (B,H,W) -> dx (B,H,W,32), dw (3,3,32) and db (1), all written (not accumulated).
 * One streaming pass + a fixed-tree reduce of one partial row per workgroup: deterministic.
 * workspace: mumpy_final_conv_bwd_workspace_bytes(B,H,W) bytes. */
int64_t mumpy_final_conv_bwd_workspace_bytes(int B, int H, int W);
int mumpy_final_conv_bwd(const float* x, const float* w_krsc, const float* dy, float* dx, float* dw, float* db, void* workspace,
                         int64_t workspace_bytes, int B, int H, int W, int C, void* stream);

/* ---- eval tail: sigmoid -> >0.5 -> uint8 mask  — test.py:100-108 ----------------------- */
int mumpy_sigmoid_threshold_fwd(const float* logits, uint8_t* mask, int64_t n, float thr, void* stream);

/* ---- input staging: ToTensor + Normalize + HWC->CHW of the eval pipeline (test.py:22-25)
 * frames (nframes,H,W,3) uint8 on the DEVICE -> out (nframes,3,H,W) fp32 = (v/255 - mean[c]) / std[c].
 * mean3 / std3 are HOST pointers to 3 floats (read at launch time). */
int mumpy_normalize_u8_fwd(const uint8_t* frames, float* out, int64_t nframes, int H, int W, const float* mean3,
                           const float* std3, void* stream);

/* ---- the same with the loader's resize in front (universaldataset.py:75-79: PIL `img.resize(inputRes)` with the default
 * filter of the pinned pillow==4.0.0, NEAREST): frames (nframes,Hs,Ws,3) uint8 -> out (nframes,3,H,W) fp32, out pixel (y,x) =
 * source pixel (ytab[y], xtab[x]).  How non-224x224 footage (432x240, config 4) enters the model in the reference.
 * ytab (H) / xtab (W) are DEVICE int32 tables holding Pillow's NEAREST source indices; build them on the host with
 * mumpy_resize_nearest_table (Pillow's own double-accumulator walk, which decides exact ties) and upload.  W % 4 == 0. */
int mumpy_resize_nearest_table(int src, int dst, int32_t* table_host);
int mumpy_resize_normalize_u8_fwd(const uint8_t* frames, float* out, const int32_t* ytab, const int32_t* xtab,
                                  int64_t nframes, int Hs, int Ws, int H, int W, const float* mean3, const float* std3,
                                  void* stream);

/* ---- out = a + b (n floats, n % 4 == 0): the residual add of CrossSwinBlock whose un-added operand is also
 *      consumed by the next view (mTVE:275-276).  out may alias a or b. */
int mumpy_add_fwd(const float* a, const float* b, float* out, int64_t n, void* stream);

/* ---- training tail -------------------------------------------------------------------
 * mask loss = softIoULoss + WeightedFocalLoss exactly as train.py:107-113 calls utils/loss.py:6-55 on logits (B,P) and
 * 0/1 targets (B,P): per-sample soft IoU with guard `eps` (the reference call site passes `recall`=False into the `e`
 * slot, loss.py:49, so eps = 0 there), focal with alpha=[1,1], gamma=2 averaged over all B*P elements.
 * loss3[0..2] = {(iou + focal) * loss_scale, iou, focal}; dlogits (B,P) = d(loss3[0])/dlogits, or NULL for loss only.
 * loss_scale = 1 / accumulation_steps (train.py:115).  Deterministic (fixed-order reductions, no atomics).
 * workspace: device scratch of mumpy_mask_loss_workspace_bytes(B,P) bytes, caller-owned. */
int64_t mumpy_mask_loss_workspace_bytes(int B, int64_t P);
int mumpy_mask_loss_fwd_bwd(const float* logits, const float* target, float* dlogits, float* loss3, void* workspace,
                            int64_t workspace_bytes, int B, int64_t P, float eps, float loss_scale, void* stream);

/* fused AdamW step over a FLAT buffer of n parameters (torch.optim.AdamW single-tensor semantics, utils/utils.py:258):
 *   p *= 1 - lr*wd;  m += (1-b1)(g - m);  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * with g = grad * grad_scale (e.g. 1/world_size after a sum all-reduce).  step t >= 1 is the count INCLUDING this
 * update.  In place on param / exp_avg / exp_avg_sq.  Hyper-parameters are doubles (the Python floats torch receives):
 * each derived factor is rounded to fp32 once, as torch does, so the update matches torch.optim.AdamW to round-off. */
int mumpy_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                     double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale,
                     void* stream);

/* the same update with its 8 step-dependent constants read from DEVICE memory, for hipGraph replay of a training step (the
 * launch is frozen at capture; lr, bias corrections and gradient scale are not).  mumpy_adamw_hyper fills a HOST array of 8
 * floats with exactly the constants mumpy_adamw_step would use; the caller copies it to `hyper_dev` before each replay. */
int mumpy_adamw_hyper(float* out8_host, double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                      double grad_scale);
int mumpy_adamw_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                         const float* hyper_dev, void* stream);

/* ---- backward kernels of the Swin block (rows 5-7 of 8a in training) ---------------------------------
 * LayerNorm backward (swin:266,305): x, dy, dx (rows,C); gamma, dgamma, dbeta (C); C % 4 == 0, C <= 2048.
 * dx_add (rows,C) or null: added to dx -- the gradient arriving over the residual branch that bypasses the LayerNorm
 * (x feeds both `norm(x)` and `x + ...`, swin:302-305), so the sum needs no separate add kernel.
 * accumulate = 1: dgamma += / dbeta += (the caller's flat gradient buffer); 0: overwritten.
 * workspace: mumpy_layernorm_bwd_workspace_bytes(rows, C) bytes of device scratch.  Deterministic. */
int64_t mumpy_layernorm_bwd_workspace_bytes(int64_t rows, int C);
int mumpy_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* dx_add, float* dx, float* dgamma,
                        float* dbeta, void* workspace, int64_t workspace_bytes, int64_t rows, int C, float eps, int accumulate,
                        void* stream);

/* exact-erf GELU (nn.GELU(), swin:42) as its own kernel for training, where the pre-activation must be kept:
 * y = gelu(x);  dx = dy * gelu'(x).  n % 4 == 0. */
int mumpy_gelu_fwd(const float* x, float* y, int64_t n, void* stream);
int mumpy_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);

/* out (C,R) = in (R,C)^T: operand layout for the weight-gradient GEMMs (dW = dY^T X = linear(dY^T, X^T)). */
int mumpy_transpose_fwd(const float* in, float* out, int64_t R, int64_t C, void* stream);

/* out[c] = sum_r x[r][c] (bias gradients), fixed-order two-stage reduction; workspace from the _bytes query. */
int64_t mumpy_col_sum_workspace_bytes(int64_t R, int C);
int mumpy_col_sum_fwd(const float* x, float* out, void* workspace, int64_t workspace_bytes, int64_t R, int C, void* stream);

/* Backward of y = x W^T + b (nn.Linear under train.py:117-120's loss.backward()) from the row-major tensors as they are, no
 * transposed copies: dx (M,K) = dy (M,N) W (N,K); dW (N,K) = dy^T x; db (N) = column sums of dy.  Any of dx / dW / db may be
 * null (skipped).  accumulate bit 0: dW += (else overwritten); bit 1: db += -- the caller's flat gradient buffer, so no
 * separate add kernels.  dx is always overwritten.  N % 32 == 0, K % 32 == 0, M free.  fp32 MFMA, deterministic
 * (deep contractions are split over workgroups into workspace slabs and reduced in split order).
 * db costs no launch of its own when dW is computed too (row sums of the dY tiles the dW product stages anyway).
 * workspace: mumpy_linear_bwd_workspace_bytes(M,N,K) bytes of device scratch (required for db without dW; without it the
 * products are not split).
 * accumulate | MUMPY_MATH_BF16: the same products with both operands rounded to bf16 (RNE) while staged and multiplied on
 * v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 tensors in memory (config 5's arithmetic; db sums the rounded dY as a
 * bf16 autocast backward does).  Still one call, no transposed copies. */
int64_t mumpy_linear_bwd_workspace_bytes(int64_t M, int N, int K);
int mumpy_linear_bwd(const float* x, const float* W, const float* dy, float* dx, float* dW, float* db, int64_t M, int N, int K,
                     int accumulate, void* workspace, int64_t workspace_bytes, void* stream);

/* PatchMerging's 2x2 gather (reference swin_transformer PatchMerging.forward, swin:357-361) and its inverse as one permutation:
 * merged (B,H/2,W/2,4C), channel block q from pixel (2i + (q&1), 2j + (q>>1)) of x (B,H,W,C).  inverse != 0: `in` is the merged
 * layout and `out` the (B,H,W,C) one (the gather's backward).  H, W even, C % 4 == 0. */
int mumpy_patch_gather_fwd(const float* in, float* out, int64_t B, int H, int W, int C, int inverse, void* stream);

/* Weight of the data-gradient convolution: out (Cin,kh,kw,Cout), out[ci][r][s][co] = w_krsc[co][kh-1-r][kw-1-s][ci] -- dX of
 * mumpy_conv2d_nhwc_fwd is mumpy_conv2d_nhwc_fwd(dY, this) (one launch instead of torch's permute + flip + contiguous). */
int mumpy_conv_weight_dgrad_fwd(const float* w_krsc, float* out, int Cout, int Cin, int kh, int kw, void* stream);

/* Weight gradient of mumpy_conv2d_nhwc_fwd (the decoder's nn.Conv2d under loss.backward(), decoder.py:9,24-31,68-95):
 * dW (Cout,kh,kw,Cin) (+)= sum over output pixels p of dy[p][:] (x) x[p + tap displacement][:], taps outside the image
 * contributing zero -- ONE launch over all taps on the NHWC tensors as they are (no padded or shifted copies, no transposes):
 * x (B,H,W,Cin), dy (B,H,W,Cout) NHWC; stride 1, odd taps, zero "same" padding; Cin % 32 == 0, Cout % 32 == 0.
 * accumulate = 1: dW += (the caller's flat gradient buffer); | MUMPY_MATH_BF16: bf16 operands as in mumpy_linear_bwd.
 * Deterministic (pixel ranges split over workgroups into workspace slabs, reduced in split order).  workspace: mumpy_conv2d_wgrad_workspace_bytes(...) bytes (optional: without it no split). */
int64_t mumpy_conv2d_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw);
int mumpy_conv2d_wgrad_nhwc(const float* x, const float* dy, float* dW, int B, int H, int W, int Cin, int Cout, int kh, int kw,
                            int accumulate, void* workspace, int64_t workspace_bytes, void* stream);

/* window attention backward (row 5 of 8a in training; swin:145-163 differentiated): qkv (B,Hs*W,3C) and dout (B,Hs*W,C)
 * in raster order as in the forward, bias / mask_tab / mask_id as in the forward, rel_index = the (49*49) int32 image of
 * `relative_position_index`.  Writes dqkv (B,Hs*W,3C) (every element) and dtable (169, C/32) = gradient of
 * `relative_position_bias_table` (accumulate = 1: dtable +=).  P is recomputed from q,k (nothing else is kept from the
 * forward).  Deterministic.
 * workspace: mumpy_window_attention_bwd_workspace_bytes(B,Hs,W,C) bytes of device scratch. */
int64_t mumpy_window_attention_bwd_workspace_bytes(int B, int Hs, int W, int C);
int mumpy_window_attention_bwd(const float* qkv, const float* dout, const float* bias, const float* mask_tab,
                               const int32_t* mask_id, int n_mask, const int32_t* rel_index, float* dqkv, float* dtable,
                               void* workspace, int64_t workspace_bytes, int B, int Hs, int W, int C, int shift, float scale,
                               int accumulate, void* stream);
/* The same with the INVERSE of relative_position_index supplied by the caller (built once per module): rel_csr = int32
 * [ptr (170) | pairs (49*49)], the pairs p = 49 i + j with rel_index[p] == t are pairs[ptr[t] .. ptr[t+1]) in increasing p.  The table
 * gradient then reads its <= 49 values per entry instead of scanning the index (20 -> 5 us per Swin block of a training step). */
int mumpy_window_attention_bwd_csr(const float* qkv, const float* dout, const float* bias, const float* mask_tab,
                                   const int32_t* mask_id, int n_mask, const int32_t* rel_index, const int32_t* rel_csr, float* dqkv,
                                   float* dtable, void* workspace, int64_t workspace_bytes, int B, int Hs, int W, int C, int shift,
                                   float scale, int accumulate, void* stream);

/* relative_position_bias_table (169,nH) gathered through relative_position_index (49*49, int32) into the padded bias the
 * attention kernels read: out (nH,64,64) [head][query][key], rows >= 49 zero, key columns >= 49 = -1e30 (swin:148-151).
 * One launch (training rebuilds it every step: the table is a parameter). */
int mumpy_relpos_bias_expand_fwd(const float* table, const int32_t* rel_index, float* out, int nH, void* stream);

/* GroupNorm (+ReLU) backward, NHWC (BaselineDecoder blocks, decoder.py:233-271): z, dy, dz (B,HW,C); stats_partial /
 * nsplit_stats = the partial sums mumpy_gn_stats_nhwc_fwd produced for z; `relu` is the activation that followed the
 * norm: 0 none, 1 ReLU (dy masked where GN(z) <= 0), 2 sigmoid (dy scaled by s(1-s)).
 * dgamma, dbeta (C): written, or -- `relu | MUMPY_GN_ACCUMULATE` -- added to what the buffers hold (the caller's flat gradient
 * buffer: no separate add launches).  Same C / G limits as mumpy_gn_stats_nhwc_fwd.  Deterministic. */
#define MUMPY_GN_ACCUMULATE 0x100
int64_t mumpy_gn_bwd_workspace_bytes(int B, int64_t HW, int C);
int mumpy_gn_bwd_nhwc(const float* z, const float* stats_partial, int nsplit_stats, const float* gamma, const float* beta,
                      const float* dy, float* dz, float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes,
                      int B, int64_t HW, int C, int G, float eps, int relu, void* stream);

/* backward of mumpy_temporal_attention_fwd (blocks:57-71 differentiated): qkv (S,T,3C), dout (S,T,C) -> dqkv (S,T,3C). */
int mumpy_temporal_attention_bwd(const float* qkv, const float* dout, float* dqkv, int64_t S, int T, int C, int heads,
                                 float scale, void* stream);

/* stochastic depth (timm DropPath as used at swin:302,305): out[b][:] = x[b][:] * scale[b], scale = Bernoulli(keep)/keep
 * drawn by the caller; the backward is the same call on the gradient.  per_sample % 4 == 0. */
int mumpy_scale_samples_fwd(const float* x, const float* scale, float* out, int B, int64_t per_sample, void* stream);

/* backward of nn.Upsample(scale_factor=s, mode="bilinear", align_corners=...) on NHWC, s in {2, 4}: dy (B,sH,sW,C) -> dx (B,H,W,C). */
int mumpy_upsample_bwd_nhwc(const float* dy, float* dx, int B, int H, int W, int C, int scale, int align_corners, void* stream);

/* ---- training kernels of the deformable cross-view attention (row 10; deform:324-405) -------------------------------
 * depthwise 5x5 convolution (padding 2) inside 7x7 windows, the first layer of `conv_offset` (deform:228-233):
 * x, u, du, dx (N, 49, C) token-major windows; w (C, 25) = the module's (C,1,5,5) weight; b (C).  The backward writes dx,
 * dw as a (25, C) image (tap-major: transpose to get (C,25)) and db (C).  C <= 384. */
int mumpy_dwconv5_window_fwd(const float* x, const float* w, const float* b, float* u, int64_t N, int C, void* stream);
int64_t mumpy_dwconv5_window_bwd_workspace_bytes(int64_t N, int C);
int mumpy_dwconv5_window_bwd(const float* x, const float* w, const float* du, float* dx, float* dw, float* db,
                             void* workspace, int64_t workspace_bytes, int64_t N, int C, void* stream);

/* backward of mumpy_deform_attention_fwd in window form (deform:360-395): q (B1w,49,C), kv (B1w*r,49,2C), dout (B1w,49,C).
 * dq_part (B1w*r,49,C): the contribution of each kv window to its q window (b2 % B1w) -- the caller sums the r members;
 * dkv (B1w*r,49,2C).  workspace from the _bytes query. */
int64_t mumpy_deform_attention_bwd_workspace_bytes(int64_t B2w, int C);
int mumpy_deform_attention_bwd(const float* q, const float* kv, const float* dout, float* dq_part, float* dkv, void* workspace,
                               int64_t workspace_bytes, int64_t B1w, int r, int C, float scale, void* stream);

/* backward of mumpy_deform_sample_fwd in window form (every kv window its own 7x7 image): x2, dsampled, dx2 (B2,49,C),
 * pos (nq,3,49,2); kv window b2 uses pos[b2 % nq].  dpos_part (B2,3,49,2) holds each kv window's contribution; the
 * caller sums the B2/nq windows that share a q window. */
int mumpy_deform_sample_bwd(const float* x2, const float* pos, const float* dsampled, float* dx2, float* dpos_part,
                            int64_t B2, int C, int nq, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MUMPY_HIP_H */
