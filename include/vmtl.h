/* vmtl.h — C ABI of libvmtl.so: the MI355X (gfx950) kernels behind the
 * vision_mtl multi-task dense-prediction training step.
 *
 * The reference (kirilllzaitsev/vision_mtl) has no FFI of its own: its "operator
 * API" for this path is torch.nn.Module.forward + autograd reaching ATen.  Each
 * entry point below therefore cites the reference call site (file:line, relative
 * to the reference repo root) whose ATen dispatch it replaces.  INTEGRATION.md
 * shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory (HBM) unless noted.
 *  - activations are NHWC fp32, [B][H][W][Cs], Cs = round_up(C,4); channels [C,Cs) are 0.
 *  - `stream` is a hipStream_t (void*); nothing here allocates, frees or synchronises,
 *    so every call can be captured into a hipGraph.
 *  - return 0 on success, <0 on error (never throws): -1 bad argument, -2 launch
 *    failure, -3 unsupported configuration.
 */
#ifndef VMTL_H
#define VMTL_H

#ifdef __cplusplus
extern "C" {
#endif

#define VMTL_ACT_NONE 0
#define VMTL_ACT_RELU 1
#define VMTL_ACT_HSWISH 2
#define VMTL_ACT_HSIGMOID 3
#define VMTL_ACT_SIGMOID 4

const char* vmtl_version(void);
/* HIP's text for the last launch failure (status -2) seen on the calling thread */
const char* vmtl_last_error_string(void);
/* The VMTL_* tuning overrides (VMTL_FORCE_TILE, VMTL_BF16X3, ...) are read from the environment once and cached;
 * this makes the next use re-read them.  Returns the new epoch. */
int vmtl_reload_env(void);
/* tuning aid: *out = 100 MHz device wall clock at the moment `stream` gets there */
int vmtl_timestamp(long long* out, void* stream);

/* ---- convolution (implicit GEMM on exact-fp32 MFMA) -------------------------------------
 * replaces nn.Conv2d / nn.ConvTranspose2d at utils/model_utils.py:71,74;
 * models/mtan_model.py:31-46,105-129,214-216,369; models/basic_model.py:30-41 (SegmentationHead);
 * utils/model_utils.py:25-34 (smp.Unet decoder + timm pointwise convs). */

/* y[m][n] = act(bias[n] + sum_kk gather(x)[m][kk] * wp[n][kk]).  wp is [Nw][KH*KW*Cs] (see
 * vmtl_pack_weights).  The same call with the tap-flipped packing and pad' = K-1-pad is the
 * data gradient of a stride-1 conv.  shuffle=1 scatters rows n=(u*2+v)*Cout+co to output pixel
 * (2h+u, 2w+v) channel co: ConvTranspose2d(k=2,s=2).  stats (optional) receives per-row-block
 * column (mean, M2 = sum (v-mean)^2), [vmtl_conv2d_stats_rows()][2][ldy], for BatchNorm. */
int vmtl_conv2d_fwd(const float* x, const float* wp, const float* bias, float* y, float* stats,
                    int B, int H, int W, int Cs, int Ho, int Wo, int ldy, int Nw, int Cout,
                    int KH, int KW, int stride, int pad, int act, int shuffle, void* stream);
/* split-K form for contractions without activation / statistics / shuffle (data gradients, and forward convs of
 * tile-starved layers - decoder blocks 0-1 at small batch - whose BatchNorm then takes its statistics from a sweep):
 * ws = vmtl_conv2d_ksplit(...)*B*Ho*Wo*ldy floats; bias (nullable) is added by the slab sum */
int vmtl_conv2d_ksplit(int B, int Ho, int Wo, int ldy, int Ktot);
int vmtl_conv2d_fwd_ws(const float* x, const float* wp, const float* bias, float* y, float* ws, int B, int H, int W, int Cs,
                       int Ho, int Wo, int ldy, int Nw, int Cout, int KH, int KW, int stride, int pad, void* stream);
int vmtl_conv2d_stats_rows(int B, int Ho, int Wo, int ldy);
int vmtl_conv2d_stats_block(int B, int Ho, int Wo, int ldy); /* output rows per stats row block */

/* Pointwise (1x1 / stride 1) conv as a plain GEMM y[M][ldy] = x[M][Ks] * wp[Nw][Ks]^T (+ bias) for small problems on a
 * dependent chain (timm pointwise convs via utils/model_utils.py:25-34; models/mtan_model.py:31,39,105,113,369):
 * fragments straight from global memory, K split over the waves of a workgroup when the tile grid is small
 * (csrc/conv_pw.hip).  stats (optional): [vmtl_conv1x1_stats_rows][2][ldy] per-row-block (mean, M2) of y,
 * vmtl_conv1x1_stats_block rows each. */
/* variant 0: vmtl_conv1x1_fwd / _cat_fwd / _bn_fwd / _cat_dgrad / _bnbwd (large problems run on the persistent large-M
 * kernel, whose row block differs); 1: launches with a residual operand or an addend (vmtl_conv1x1_bn_res_fwd,
 * vmtl_conv1x1_bnbwd_add: always the fragment-from-global kernel) */
int vmtl_conv1x1_stats_block(int M, int ldy, int Ks, int variant);
int vmtl_conv1x1_stats_rows(int M, int ldy, int Ks, int variant);
int vmtl_conv1x1_fwd(const float* x, const float* wp, const float* bias, float* y, float* stats, int M, int Ks, int ldy,
                     int Nw, int Cout, void* stream);
/* conv1x1(cat[x, x2]) without the concat (mtan_model.py:57-59,139-141): x [M][K1] (K1 % 4 == 0), x2 [M][K2s], packed
 * weight rows [Nw][K1 + K2s]; and its data gradient writing [dx | dx2] (dx [M][N1], dx2 [M][N2s]) without a split pass */
int vmtl_conv1x1_cat_fwd(const float* x, int K1, const float* x2, int K2s, const float* wp, const float* bias, float* y,
                         float* stats, int M, int ldy, int Nw, int Cout, void* stream);
int vmtl_conv1x1_cat_dgrad(const float* dy, const float* wp, float* dx, int N1, float* dx2, int N2s, int N2, int M,
                           int Ks, void* stream);
/* and its weight gradient in one launch: slabs [splits][Nw][K1 + K2s], splits = vmtl_conv2d_wgrad_splits(M, Nw, K1 + K2s) */
int vmtl_conv1x1_cat_wgrad(const float* x, int K1, const float* x2, int K2s, const float* dy, float* slabs, int splits,
                           int M, int ldy, int Nw, void* stream);
/* pointwise pre-activation node (the 1x1 counterpart of vmtl_conv3x3_small's prologue / vmtl_conv2d_bnbwd):
 * y = conv1x1(act(coef_a[k]*x + coef_c[k])) - BatchNorm + activation (none / relu / hardswish) of the layer that produced
 * x applied to the operand fragments, a_out (nullable [M][Ks]) = the activated matrix for the weight gradient;
 * and the data gradient ending with that activation's + BatchNorm's backward: dz = (dy*W) * act'(gamma*xhat + beta),
 * stats [vmtl_conv1x1_stats_rows(M, ldy, Ks)][2][ldy] = per-row-block (sum dz, sum dz*xhat). */
int vmtl_conv1x1_bn_fwd(const float* x, const float* coef_a, const float* coef_c, int act_in, float* a_out,
                        const float* wp, const float* bias, float* y, float* stats, int M, int Ks, int ldy, int Nw,
                        int Cout, void* stream);
int vmtl_conv1x1_bnbwd(const float* dy, const float* wp, float* dz, float* stats, const float* ez_x,
                       const float* ez_mean, const float* ez_invstd, const float* ez_gamma, const float* ez_beta,
                       int ez_act, int M, int Ks, int ldy, int Nw, int Cout, void* stream);
/* with a residual operand: the GEMM input is a = act(coef_a*x + coef_c) + res (an inverted-residual block's bn3 output +
 * skip connection consumed by the next block's expand conv; a_out receives a), and with a second gradient of the
 * differentiated tensor added before the backward: dz = (dy*W + addend) * act'(...) */
int vmtl_conv1x1_bn_res_fwd(const float* x, const float* coef_a, const float* coef_c, int act_in, const float* res,
                            float* a_out, const float* wp, const float* bias, float* y, float* stats, int M, int Ks,
                            int ldy, int Nw, int Cout, void* stream);
int vmtl_conv1x1_bnbwd_add(const float* dy, const float* wp, const float* addend, float* dz, float* stats,
                           const float* ez_x, const float* ez_mean, const float* ez_invstd, const float* ez_gamma,
                           const float* ez_beta, int ez_act, int M, int Ks, int ldy, int Nw, int Cout, void* stream);

/* vmtl_conv2d_fwd used as a DATA GRADIENT with the BatchNorm + activation backward of the layer that produced the
 * differentiated tensor fused into the epilogue (reference utils/model_utils.py:72-76 run backwards): y = conv *
 * act'(ez_gamma*xhat + ez_beta), xhat = (ez_x - ez_mean)*ez_invstd; stats[vmtl_conv2d_stats_rows(B,Ho,Wo,ldy)][2][ldy] =
 * per-row-block (sum y, sum y*xhat) -> vmtl_bn_bwd_finalize -> vmtl_bn_bwd_apply. */
int vmtl_conv2d_bnbwd(const float* x, const float* wp, float* y, float* stats, const float* ez_x, const float* ez_mean,
                      const float* ez_invstd, const float* ez_gamma, const float* ez_beta, int ez_act, int B, int H,
                      int W, int Cs, int Ho, int Wo, int ldy, int Nw, int Cout, int KH, int KW, int stride, int pad,
                      void* stream);

/* nearest-x2 upsample of xl + concat with skip + 3x3/pad-1 conv (smp DecoderBlock entry, reference
 * utils/model_utils.py:25-34) as four 2x2 phase convolutions on the low-res map: wp_eff from
 * vmtl_pack_up2_fwd ([4][Cout][4*C0s + 9*C1s]); y is [B][2*H2][2*W2][ldy].  Backward uses
 * vmtl_conv2d_fwd(k4,s2,p1) over dY with vmtl_pack_up2_dgrad, vmtl_conv2d_wgrad(k4,s2,p1) + vmtl_unpack_up2. */
int vmtl_conv2d_up2_fwd(const float* xl, const float* skip, const float* wp_eff, float* y, float* stats,
                        int B, int H2, int W2, int C0s, int C1s, int ldy, int Cout, void* stream);
int vmtl_conv2d_up2_stats_block(int B, int H2, int W2, int ldy);
/* split-K form of vmtl_conv2d_up2_fwd (no statistics): ws = vmtl_conv2d_up2_ksplit(...)*B*2H2*2W2*ldy floats */
int vmtl_conv2d_up2_ksplit(int B, int H2, int W2, int ldy, int Ktot);
int vmtl_conv2d_up2_fwd_ws(const float* xl, const float* skip, const float* wp_eff, float* y, float* ws, int B, int H2,
                           int W2, int C0s, int C1s, int ldy, int Cout, void* stream);
int vmtl_pack_up2_fwd(const float* w, float* dst, int Cout, int C0, int C0s, int C1, int C1s, void* stream);
int vmtl_pack_up2_dgrad(const float* w, float* dst, int Cout, int Cos, int C0, int Cin, void* stream);
int vmtl_unpack_up2(const float* slabs, float* grad, int Cout, int Cos, int C0, int Cin, int nslabs, void* stream);

/* slabs[z][n][kk] = sum_{m in pixel slice z} dy[m][n] * gather(x)[m][kk]  (packed layout, plain
 * stores, no atomics).  splits = vmtl_conv2d_wgrad_splits(B*Ho*Wo, Nw, KH*KW*Cs); the caller
 * provides splits*Nw*KH*KW*Cs floats and vmtl_unpack_weights(..., nslabs=splits) adds the slabs
 * in index order while converting to the torch layout (deterministic weight gradient). */
int vmtl_conv2d_wgrad_splits(int M, int Nw, int Ktot);
int vmtl_conv2d_wgrad(const float* x, const float* dy, float* slabs, int splits, int B, int H, int W, int Cs,
                      int Ho, int Wo, int ldy, int Nw, int KH, int KW, int stride, int pad, void* stream);

/* Weight gradient of a NARROW 3x3 / stride 1 / pad 1 conv on a strip-walking halo kernel (csrc/conv_wgrad_small.hip): x is read
 * from memory once instead of once per tap.  Replaces the same conv2d backward-weight call sites of the reference as
 * vmtl_conv2d_wgrad (reference models: smp DecoderBlock / SegmentationHead convs, utils/model_utils.py:61-80 DoubleConv) where
 * supported(Cs, ldy, W) says so: Cs, ldy <= 36 (multiples of 4), W a multiple of 32.  slabs: [vmtl_conv3x3_wgrad_small_slabs(B, H, W)]
 * [Nw][9*Cs], summed by vmtl_unpack_weights(..., nslabs) exactly like the slabs of vmtl_conv2d_wgrad. */
int vmtl_conv3x3_wgrad_small_supported(int Cs, int ldy, int W);
int vmtl_conv3x3_wgrad_small_slabs(int B, int H, int W);
int vmtl_conv3x3_wgrad_small(const float* x, const float* dy, float* slabs, int nslabs, int B, int H, int W, int Cs,
                             int ldy, int Nw, void* stream);


/* 3x3 / stride 1 / pad 1 conv for the narrow full-resolution layers (Cs in {16,20,32,36} storage channels in,
 * Nw <= 36 rows of the packed [Nw][9*Cs] weight out): the last decoder block and the heads of `basic`
 * (models/basic_model.py:30-51; smp DecoderBlock conv2 via utils/model_utils.py:25-34) and their data gradients.
 * Halo tile in LDS, weights resident in LDS, persistent workgroups (csrc/conv_small.hip).
 *   prologue: v = act_in(pa[c]*x + pb[c]*x2 + pc[c]) once per input element (pa null: identity; x2/pb null:
 *             one operand); a_out (optional, needs pa) receives the transformed input.
 *   ep_mode 0: y = conv + bias.  yb != null: NCHW split store, channels [0,Ca) -> y [B][Ca][H][W], the rest -> yb.
 *   ep_mode 1: y = conv, stats[row][2][ldy] = (mean, M2) of y over the row's pixels; rows / pixels per row:
 *              vmtl_conv3x3_small_stat_rows / _stat_block (a row is one 128-pixel tile, or all tiles of a workgroup).
 *   ep_mode 2: y = conv * act'(z), z = ez_gamma*xhat + ez_beta, xhat = (ez_x - ez_mean)*ez_invstd (BatchNorm + activation
 *              backward of the layer that produced the tensor whose gradient this conv computes);
 *              stats[row][2][ldy] = (sum y, sum y*xhat) over the row's pixels.
 * ep_mode 1/2 need H % 4 == 0 and W % 32 == 0. */
int vmtl_conv3x3_small_supported(int Cs, int Nw);
int vmtl_conv3x3_small_tiles(int B, int H, int W);
int vmtl_conv3x3_small_stat_rows(int B, int H, int W);  /* rows of stats ... */
int vmtl_conv3x3_small_stat_block(int B, int H, int W); /* ... and the pixels each row covers */
int vmtl_conv3x3_small(const float* x, const float* x2, const float* pa, const float* pb, const float* pc,
                       int act_in, float* a_out, const float* wp, const float* bias, float* y, float* yb, int Ca,
                       float* stats, int ep_mode, const float* ez_x, const float* ez_mean, const float* ez_invstd,
                       const float* ez_gamma, const float* ez_beta, int ez_act, int B, int H, int W, int Cs, int ldy,
                       int Nw, int Cout, void* stream);

/* depthwise KxK (K in {3,5}, stride in {1,2}); wp is packed [K*K][Cs]. */
int vmtl_dwconv_fwd(const float* x, const float* wp, float* y, int B, int H, int W, int Cs, int Ho, int Wo,
                    int K, int stride, int pad, void* stream);
/* depthwise conv as a pre-activation node: v = act(coef_a[c]*x + coef_c[c]) applied to the taps while loading (the
 * BatchNorm + activation of the pointwise conv that produced x; coefficients from vmtl_bn_stats_coef), y = dwconv(v),
 * a_out (optional) = v, partial (optional) = [vmtl_dwconv_bn_stats_rows(M, Cs)][2][Cs] (mean, M2) rows of y over
 * vmtl_dwconv_bn_stats_block(M, Cs) output pixels each, M = B*Ho*Wo.  pad must be (K-1)/2. */
int vmtl_dwconv_bn_stats_rows(int M, int Cs);
int vmtl_dwconv_bn_stats_block(int M, int Cs);
int vmtl_dwconv_bn_fwd(const float* x, const float* coef_a, const float* coef_c, int act, const float* wp, float* y,
                       float* a_out, float* partial, int B, int H, int W, int Cs, int Ho, int Wo, int K, int stride,
                       int pad, void* stream);
int vmtl_dwconv_bwd_data(const float* dy, const float* wp, float* dx, int B, int H, int W, int Cs, int Ho,
                         int Wo, int K, int stride, int pad, void* stream);
/* dx = dwconv^T(dy) + addend (a second gradient of the same tensor, e.g. the residual branch of the block) */
int vmtl_dwconv_bwd_data_add(const float* dy, const float* wp, const float* addend, float* dx, int B, int H, int W,
                             int Cs, int Ho, int Wo, int K, int stride, int pad, void* stream);
/* partial: vmtl_dwconv_bwd_weight_rows(B*Ho*Wo, Cs) * K*K * Cs floats of scratch; dw in the torch (C,1,K,K) layout */
int vmtl_dwconv_bwd_weight_rows(int M, int Cs);
int vmtl_dwconv_bwd_weight(const float* x, const float* dy, float* partial, float* dw, int B, int H, int W,
                           int C, int Cs, int Ho, int Wo, int K, int stride, int pad, void* stream);

/* torch parameter layout <-> packed GEMM operand (formula in csrc/pack.hip). */
int vmtl_pack_weights(const float* src, float* dst, int R1, int R0, int T, int C, int Cs, long long sr1,
                      long long sr0, long long st, long long sc, int flip, void* stream);
/* the same with a per-input-channel factor folded into the operand: the cross-stitch scale of the tensor the conv reads
 * (models/cross_stitch_model.py:32-37 followed by the next conv of the CSNet walk :108-142) - conv(W, s*x) = conv(W*s, x).
 * smode 1: packed column channel = input channel (forward packing); 2: packed row (data-gradient packing);
 * factor = scale[channel * sstride] (sstride 0: layer-wise stitching, one scalar) */
int vmtl_pack_weights_scaled(const float* src, float* dst, int R1, int R0, int T, int C, int Cs, long long sr1,
                             long long sr0, long long st, long long sc, int flip, const float* scale, int sstride,
                             int smode, void* stream);
/* weight gradient behind a folded stitch scale: slabs hold dL/d(W*s); grad <- dL/dW (torch layout (R0, C, T)),
 * ds <- dL/ds (C entries, or one scalar when reduce_all) - the CrossStitchLayer weight gradient without a pass over
 * activations; work: R0*C*T + C floats */
int vmtl_unpack_weights_stitch(const float* packed, float* grad, const float* w, const float* scale, int sstride,
                               float* ds, float* work, int R0, int T, int C, int Cs, int nslabs, long long slab_stride,
                               int reduce_all, void* stream);
/* batched form: descs = device array of n records {src*, dst*, i64 sr1, sr0, st, sc, start; scale*; i32 R1, R0, T, C,
 * Cs, flip, sstride, smode} (vmtl_pack_desc_bytes() bytes each), start = first flat work index; total = sum R1*R0*T*Cs. */
int vmtl_pack_desc_bytes(void);
int vmtl_pack_weights_batch(const void* descs, int n, long long total, void* stream);
int vmtl_pack_weights_slice(const float* src, float* dst, int R0, int T, int C, int group, long long sr0,
                            long long st, long long sc, int flip, void* stream);
int vmtl_unpack_weights(const float* packed, float* grad, int R1, int R0, int T, int C, int Cs,
                        long long sr1, long long sr0, long long st, long long sc, int flip, int nslabs,
                        long long slab_stride, void* stream); /* slab_stride 0: R1*R0*T*Cs */

/* ---- BatchNorm2d (+ activation, gate multiply, residual add) -----------------------------
 * replaces nn.BatchNorm2d/ReLU/Sigmoid/mul at utils/model_utils.py:72-76;
 * models/mtan_model.py:67-81,139-167. */
int vmtl_reduce_rows(int M); /* partial rows used by the two-stage reductions */
int vmtl_bn_stats(const float* x, int M, int C, int Cs, float* partial, int nblk_from_conv,
                  int rows_per_blk_from_conv, float eps, float momentum, float* running_mean, float* running_var, long long* num_batches_tracked,
                  float* save_mean, float* save_invstd, void* stream);
int vmtl_bn_eval_stats(const float* running_mean, const float* running_var, int C, int Cs, float eps,
                       float* save_mean, float* save_invstd, void* stream);
/* the same two, additionally emitting the normalisation as per-channel prologue coefficients of the consumer
 * conv (vmtl_conv3x3_small): act(BN(x)) = act(coef_a[c] * x + coef_c[c]), zeros on the pad channels */
int vmtl_bn_stats_coef(const float* x, int M, int C, int Cs, float* partial, int nblk_from_conv,
                       int rows_per_blk_from_conv, float eps, float momentum, float* running_mean, float* running_var,
                       long long* num_batches_tracked, float* save_mean, float* save_invstd, const float* gamma,
                       const float* beta, float* coef_a, float* coef_c, void* stream);
int vmtl_bn_eval_stats_coef(const float* running_mean, const float* running_var, int C, int Cs, float eps,
                            float* save_mean, float* save_invstd, const float* gamma, const float* beta,
                            float* coef_a, float* coef_c, void* stream);
int vmtl_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma,
                  const float* beta, const float* mul, const float* res, float* y, long long M, int C,
                  int Cs, int act, void* stream);
/* finalize + apply in one launch for training-mode BatchNorm whose statistics come as few per-block rows
 * (nblk <= vmtl_bn_fuse_max_rows(); same arguments as vmtl_bn_stats with conv partials + vmtl_bn_apply) */
int vmtl_bn_fuse_max_rows(void);
int vmtl_bn_apply_fused(const float* x, const float* partial, int nblk, int rows_per_blk, float eps, float momentum,
                        float* running_mean, float* running_var, long long* num_batches_tracked, float* save_mean,
                        float* save_invstd, const float* gamma, const float* beta, const float* mul, const float* res,
                        float* y, long long M, int C, int Cs, int act, void* stream);
int vmtl_bn_bwd(const float* x, const float* dy, const float* mean, const float* invstd, const float* gamma,
                const float* beta, const float* mul, float* dmul, float* partial, float* sum_dz,
                float* sum_dzx, float* dx, int M, int C, int Cs, int act, int training, void* stream);
/* vmtl_bn_bwd in two halves, for producers whose epilogue already emitted dz = dy * act'(z) and the per-block
 * column sums rows [nblk][2][Cs] = (sum dz, sum dz*xhat): finalize -> BatchNorm parameter gradients (+ optional
 * coefficients of dx = coef_a*dz + coef_b*x + coef_c for a consumer that applies it while loading); apply -> dx. */
int vmtl_bn_bwd_finalize(const float* partial, int nblk, int M, int C, int Cs, float* sum_dz, float* sum_dzx,
                         const float* mean, const float* invstd, const float* gamma, int training, float* coef_a,
                         float* coef_b, float* coef_c, void* stream);
int vmtl_bn_bwd_apply(const float* x, const float* dz, const float* mean, const float* invstd, const float* gamma,
                      const float* sum_dz, const float* sum_dzx, float* dx, int M, int C, int Cs, int training,
                      void* stream);

/* BatchNorm + activation + MaxPool2d(2) as ONE node (csrc/bn.hip): reference models/mtan_model.py:67-83 (AttentionModuleEncoder:
 * conv3 -> bn3 -> ReLU -> max_pool2d, nobody else reads the full-resolution activation).  fwd: y [B][H/2][W/2][Cs] from x
 * [B][H][W][Cs] (H, W even) with mean / invstd from vmtl_bn_stats; bwd: the pooled gradient dyp is scattered to each window's
 * arg-max inside the BatchNorm reduce / apply sweeps (sum_dz, sum_dzx [C] = dbeta, dgamma; partial:
 * vmtl_reduce_rows(B*(H/2)*(W/2))*2*Cs floats). */
int vmtl_bn_act_pool2_fwd(const float* x, const float* mean, const float* invstd, const float* gamma,
                          const float* beta, float* y, int B, int H, int W, int C, int Cs, int act, void* stream);
int vmtl_bn_act_pool2_bwd(const float* x, const float* dyp, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, float* partial, float* sum_dz, float* sum_dzx,
                          float* dx, int B, int H, int W, int C, int Cs, int act, int training, void* stream);

/* out[c] = sum_m a[m][c] (mode 0) or a*b (mode 1); reduce_all sums over channels too.
 * partial: (vmtl_reduce_rows(M) + 1) * Cs floats of scratch */
int vmtl_colsum(const float* a, const float* b, int M, int C, int Cs, int mode, int reduce_all,
                float* partial, float* out, void* stream);

/* ---- gathers / elementwise ---------------------------------------------------------------
 * concat2: utils/model_utils.py:46-58; models/mtan_model.py:65,152,229;
 *          models/cross_stitch_model.py:126-134; smp DecoderBlock (nearest x2 + cat). */
int vmtl_concat2(const float* a, int Ha, int Wa, int Ca, int Csa, int upa, int oha, int owa,
                 const float* b, int Hb, int Wb, int Cb, int Csb, int upb, int ohb, int owb, float* y,
                 int B, int H, int W, int Cd, void* stream);
int vmtl_concat2_bwd(const float* dy, float* dx, int B, int H, int W, int Cd, int c_off, int Hs, int Ws,
                     int C, int Cs, int up, int oh, int ow, void* stream);
/* models/mtan_model.py:49,81,364,388 */
int vmtl_maxpool2_fwd(const float* x, float* y, int B, int H, int W, int Cs, void* stream);
int vmtl_maxpool2_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int Cs, void* stream);
/* models/mtan_model.py:125,143-144 (bilinear x2, align_corners=True) */
int vmtl_bilinear_up2_fwd(const float* x, float* y, int B, int H, int W, int Cs, void* stream);
int vmtl_bilinear_up2_bwd(const float* dy, float* dx, int B, int H, int W, int Cs, void* stream);
/* timm SqueezeExcite pieces (utils/model_utils.py:25-34) */
int vmtl_spatial_mean(const float* x, float* y, int B, int HW, int Cs, void* stream);
int vmtl_channel_bcast(const float* x, const float* s, float* y, int B, int HW, int Cs, int mode, void* stream);
int vmtl_channel_scale_bwd_s(const float* x, const float* dy, float* ds, int B, int HW, int Cs, void* stream);
/* The squeeze-excite gate as batch-sized GEMMs (M = B <= vmtl_fc_max_rows() rows): the 1x1 convs of timm
 * SqueezeExcite on a (B,1,1,C) map (encoder of models/basic_model.py:17-28).
 * vmtl_fc_fwd: z = bias + A W^T, y = act(z); A = a_scale * sum_{s<a_parts} a[s] (partial sums of a spatial
 *   reduction), optionally multiplied by act'(a_z) with activation a_act (data gradient with the activation
 *   backward fused: pass the transposed weight as w).  w is [N][ldw] with K contiguous; z may be NULL;
 *   a_out (may be NULL) receives the finished A operand [M][lda] (summed, scaled, unmasked).
 * vmtl_fc_wgrad: dw[n][k] = sum_m dz[m][n] x[m][k], db[n] = sum_m dz[m][n], dz = (sum of dy parts) * act'(zo);
 *   dw in the torch (N, K, 1, 1) layout.
 * vmtl_hw_reduce: part[s][b][c] = sum over HW slice s of x (* y if given); S = vmtl_hw_reduce_parts(B,HW,Cs).
 * vmtl_channel_scale_add: y = x * s[b][c] + t[b][c] * t_scale (t may be NULL). */
int vmtl_fc_max_rows(void);
int vmtl_fc_fwd(const float* a, int a_parts, long long a_part_stride, float a_scale, const float* a_z, int a_act,
                float* a_out, const float* w, const float* bias, float* z, float* y, int M, int K, int N, int lda,
                int ldw, int ldy, int act, void* stream);
int vmtl_fc_wgrad(const float* x, int x_parts, long long x_part_stride, float x_scale, const float* dyo, int dy_parts,
                  long long dy_part_stride, const float* zo, float* dw, float* db, int M, int K, int N, int lda, int ldn,
                  int act, void* stream);
int vmtl_hw_reduce_parts(int B, int HW, int Cs);
int vmtl_hw_reduce(const float* x, const float* y, float* part, int B, int HW, int Cs, void* stream);
int vmtl_channel_scale_add(const float* x, const float* s, const float* t, float t_scale, float* y, int B, int HW,
                           int Cs, void* stream);
/* models/cross_stitch_model.py:32-37 (diagonal of the 2x2 stitch matrix) */
int vmtl_stitch(const float* x, const float* w, float* y, long long M, int C, int Cs, int wstride, void* stream);
/* its backward in one sweep: dx (nullable) = w * dy, dw[c] = sum_m x*dy (reduce_all: one scalar);
 * partial: (vmtl_reduce_rows(M) + 1) * Cs floats of scratch */
int vmtl_stitch_bwd(const float* x, const float* dy, const float* w, float* dx, float* partial, float* dw, int M,
                    int C, int Cs, int wstride, int reduce_all, void* stream);
/* mode 0 add, 1 sigmoid, 2 sigmoid-backward-from-output, 3 scale by *b */
int vmtl_eltwise(const float* a, const float* b, float* y, int mode, long long total, void* stream);
/* y = a + b (+ c) (+ d), c / d nullable, total % 4 == 0: the gradient sum of an activation with 2..4 consumers - what
 * autograd's AccumulateGrad / add nodes do for x, d, merged in models/mtan_model.py:364-399 and every residual branch */
int vmtl_add_n(const float* a, const float* b, const float* c, const float* d, float* y, long long total, void* stream);
/* y = wa*a + wb*b: the weighted sum of the task losses (lit_module.py:127-129) */
int vmtl_axpby(const float* a, const float* b, float* y, float wa, float wb, long long total, void* stream);
int vmtl_fill_zero(float* p, long long n, void* stream); /* n floats <- 0 (a memset node) */
/* lit_module.py:137-138 (argmax of softmax == argmax of logits) */
int vmtl_argmax_channels(const float* z, long long* out, int B, int HW, int C, long long sb, long long sc,
                         long long sp, void* stream);
int vmtl_nchw_to_nhwc(const float* x, float* y, int B, int C, int HW, int Cs, int Cw, void* stream);
int vmtl_nhwc_to_nchw(const float* x, float* y, int B, int C, int HW, int Cs, void* stream);
/* input side (lit_module.py:211-219 + the dataset sample contract of data_modules/cityscapes.py:39-83,
 * nyuv2.py:100-141): HWC pixel rows [P][C] -> internal NHWC storage [P][Cs] (zero pad channels), y = x * scale */
int vmtl_hwc_to_nhwc_pad(const float* x, float* y, long long P, int C, int Cs, float scale, void* stream);

/* ---- losses ------------------------------------------------------------------------------
 * lit_module.py:31,123 (CrossEntropyLoss); losses.py:14-36 (SILogLoss); lit_module.py:68,112 (MAE). */
long long vmtl_ce_workspace_bytes(long long P);
/* element (b,c,hw) of logits / dlogits sits at [b*sb + c*sc + hw*sp] (NCHW: C*HW, HW, 1). */
int vmtl_ce_fwd(const float* logits, const long long* target, float* loss, void* workspace, int B, int HW,
                int C, long long sb, long long sc, long long sp, void* stream);
/* the same, also writing argmax_c logits per pixel (the prediction of lit_module.py:137-138) */
int vmtl_ce_fwd_argmax(const float* logits, const long long* target, float* loss, void* workspace,
                       long long* argmax, int B, int HW, int C, long long sb, long long sc, long long sp,
                       void* stream);
int vmtl_ce_bwd(const float* logits, const long long* target, const float* grad_out, float* dlogits, int B,
                int HW, int C, long long sb, long long sc, long long sp, void* stream);
/* as vmtl_ce_bwd with separate strides (dsb, dsc, dsp) for dlogits, e.g. NHWC (HW*ld, 1, ld) */
int vmtl_ce_bwd_strided(const float* logits, const long long* target, const float* grad_out, float* dlogits, int B,
                        int HW, int C, long long sb, long long sc, long long sp, long long dsb, long long dsc,
                        long long dsp, void* stream);
long long vmtl_silog_workspace_bytes(long long P);
int vmtl_silog_fwd(const float* pred, const float* target, float min_depth, float* loss, float* stats,
                   void* workspace, long long P, void* stream);
int vmtl_silog_bwd(const float* pred, const float* target, const float* stats, const float* grad_out,
                   float min_depth, float* dpred, long long P, void* stream);
int vmtl_l1_fwd(const float* pred, const float* target, float* loss, void* workspace, long long P, void* stream);
int vmtl_l1_bwd(const float* pred, const float* target, const float* grad_out, float* dpred, long long P,
                void* stream);

/* ---- per-step metrics (lit_module.py:48-69,106-118: torchmetrics Accuracy / Jaccard / FBeta) ---- */
int vmtl_confusion_matrix(const long long* pred, const long long* target, int* cm, long long P, int C,
                          void* stream);
int vmtl_segm_metrics(const int* cm, int C, float beta, float* out, void* stream);

/* ---- optimizer (training_lit.py:51,87: torch.optim.Adam) ---------------------------------- */
int vmtl_adam_step(float* p, const float* g, float* m, float* v, const float* step_ptr, float lr, float b1,
                   float b2, float eps, float weight_decay, float grad_scale, long long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VMTL_H */
