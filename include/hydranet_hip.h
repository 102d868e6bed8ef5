/* hydranet_hip.h -- C ABI of libhydranet_hip.so, the MI355X (gfx950) kernels behind the HydraNet forward/backward hot path.
 *
 * Drop-in boundary.  The reference (FlowEternal/multitask-hydranet) has no FFI of its own: its hot path is the ATen op layer
 * underneath model/model.py:159-264 and the one torch.autograd.Function it authors (SwishImplementation, model/net/common.py:11-22).
 * Each entry point below replaces the ATen/cuDNN call(s) named in its comment (paths relative to /root/reference/model).
 *
 * Conventions
 *   - Activations are NHWC ("channels last") bf16: a tensor is [rows = N*H*W][C] with an explicit row stride `ld*` in ELEMENTS so
 *     producers can write channel slices of a concatenation buffer.  C and every ld must be a multiple of 8 (16-byte pieces).
 *   - Parameters and statistics are fp32.  Packed weight operands are bf16 (see hn_pack_weight / hn_gconv_pack / hn_dw_pack).
 *   - Every buffer is owned by the caller (PyTorch's caching allocator) and only borrowed for the call; the library keeps no state.
 *   - Every function only enqueues kernels on `stream` (no allocation, no synchronisation => hipGraph-capturable) and returns
 *     0 on success, 1 = bad argument, 2 = launch failure, 3 = unsupported shape.  No C++ exceptions cross the boundary.
 *   - Activation codes: 0 none, 1 ReLU, 2 Swish, 3 ELU, 4 sigmoid.
 */
#ifndef HYDRANET_HIP_H
#define HYDRANET_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

/* ---- dense contractions on MFMA (hn_gemm.hip) ------------------------------------------------------------------------------- */

/* fp32 conv weight [Cout][Cin][taps] -> bf16 forward operand wp[Cout][taps][KP(Cin)] and (optional) dgrad operand
 * wt[Cin][taps][KP(Cout)], KP(x) = x rounded up to 32, zero filled.  Replaces the implicit weight cast of autocast convs. */
int hn_gconv_pack_diag(const float* w, void* wk, void* wd, int C, hipStream_t stream);
int hn_pack_weight(const float* w, void* wp, void* wt, int Cout, int Cin, int taps, hipStream_t stream);
/* hn_pack_weight for a channel slice [ci0, ci0+Cin) of w [Cout][Cin_total][taps] (phase = 0), or (phase = 1, taps = 9) the PHASE-FORM
 * effective weights of that slice: a 3x3 conv over a nearest-x2 up-sampled map (head_seg/segmentation.py:92-104: Upsample -> ReflectionPad2d
 * -> Conv2d) evaluated on the low-resolution grid has 4*Cout outputs (one Cout-vector per output phase), each tap of which is the sum of
 * the original taps that land on it: wp [4*Cout][9][KP(Cin)], wt [Cin][9][KP(4*Cout)], b_eff [4*Cout] (optional) = bias per phase. */
int hn_pack_weight_ex(const float* w, void* wp, void* wt, int Cout, int Cin_total, int ci0, int Cin, int taps, int phase, const float* bias,
                      float* b_eff, hipStream_t stream);
/* the transpose: dw [K][C0+C1][3][3] (db [K], optional) from the effective-weight gradient dw_eff [4K][C0][3][3] (db_eff [4K]); channels
 * [C0, C0+C1) are copied from the skip operand's own gradient dw1 [K][C1][3][3] */
int hn_phase_fold(const float* dw_eff, const float* dw1, const float* db_eff, float* dw, float* db, int K, int C0, int C1, hipStream_t stream);
/* every conv weight of a model in one launch: jobs = DEVICE table njobs x 8 int64 {w, wp, wt, Cout, Cin, taps, first_block, ci_tiles}; job j
 * owns one block per 32 x 32 (cout, cin) tile -- (KP(Cout)/32) * ci_tiles blocks with ci_tiles = KP(Cin)/32 -- starting at first_block;
 * the tile passes through LDS so that both operand layouts are written in contiguous runs.  taps = 1 or 9.  block_job (optional, DEVICE
 * int32 [total_blocks]) = job index of every block (otherwise each block searches the table). */
int hn_pack_weights_batched(const long* jobs, int njobs, long total_blocks, const int* block_job, hipStream_t stream);
/* the remaining per-step packs of a model (depthwise taps = hn_dw_pack, grouped-conv stencil operands = hn_gconv_pack, block-diagonal MFMA
 * operands = hn_gconv_pack_diag, channel-slice / phase-form packs = hn_pack_weight_ex) in one launch: jobs = DEVICE table njobs x 16 int64
 * {w, out0, out1, out2, bias, kind 1..4, first_block, p0..p5, 0, 0, 0} with the parameters of the single-weight entry point of that kind,
 * one thread per output element; block_job = DEVICE int32 [total_blocks].  Kind 3 writes only the 8 x 8 diagonal blocks of the block-diagonal
 * operands: the caller's (persistent) buffers must have been zero-filled once. */
int hn_pack_small_batched(const long* jobs, const int* block_job, long total_blocks, hipStream_t stream);

/* out[pixel][cout] = act(bias[cout] + sum_{tap,c} X(pixel, tap)[c] * w[cout][tap][c]); X is gathered on the fly:
 *   mode 0: X = x0 rows (1x1 conv; also every dgrad of a 1x1 conv)            nn.Conv2d k=1: net/anynet.py:29-33,52-60;
 *           net/bifpn.py:58-102; net/common.py:95; head_lane/lanedetect.py:45-64
 *   mode 1: 1x1 conv stride 2 (XBlock shortcut)                                 net/anynet.py:57-60
 *   mode 2: 3x3 conv over reflect-pad-1 of cat[nearest_up2(x0) if up else x0, x1]   head_seg/segmentation.py:16-48,84-105
 *   mode 3: 3x3 "full" correlation of zero-extended x0 (dgrad of mode 2 on the padded (H+2)x(W+2) grid; H, W passed here are the
 *           PADDED sizes); fold back with hn_seg_fold.
 *   mode 4: mode 2 with replicate (clamp) padding, no up-sampling / concat: the low-resolution form of a 3x3 reflect-pad conv over a
 *           nearest-x2 up-sampled map (reflection of the up-sampled index == clamping of the source index); used with 4-phase
 *           effective weights for the final seg conv (head_seg/segmentation.py:101-104).  With out_f32 and
 *           img_stride = -k (k = Nout / 4) the epilogue stores depth-to-space itself: out is fp32 [N][2H][2W][k].
 *   mode 5: grouped 3x3 conv, group width 8, stride 1, zero padding 1 (XBlock conv_block_2, net/anynet.py:34-38) on MFMA: cout tile t
 *           (64 couts = 8 groups) contracts only over input channels [64t, 64t+64) with block-diagonal weights from
 *           hn_gconv_pack_diag (w = wk for the forward, wd for the data gradient); Nout == C0, KP == 64, no statistics.
 * (n_img, H, W) describe the OUTPUT pixel grid, M = n_img*H*W rows.  psum/psq (optional) receive per-wave partial sums / sums of
 * squares of the bf16-rounded outputs, [hn_nt_stat_rows(M, Nout)][Nout], for training-mode BatchNorm (F.batch_norm statistics).
 * rpi/img_stride (optional, 0 = off): out offset(pixel) = (pixel / rpi) * img_stride + (pixel % rpi) * ldc, which writes a pyramid
 * level straight into the per-image concatenation of head_detect/detection.py:37-44,74-83. */
int hn_conv_gemm_nt(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1, int up, long M,
                    const void* w, int Nout, int KP, int taps, const float* bias, int act, void* out, int out_f32, int ldc, long rpi,
                    long img_stride, float* psum, float* psq, hipStream_t stream);
int hn_nt_stat_rows(long M, int Nout);
/* pixel rows summed into one partial row (64 or 128; the last tile may be ragged): a caller that reads the rows per IMAGE (the SE
 * gate-gradient partials, emode 1) needs H*W to be a multiple of it -- M / hn_nt_stat_rows is NOT the tile size when M is ragged */
int hn_nt_stat_tile(long M, int Nout);
/* mode 5 with psum/psq: one partial row per 16x16 output patch */
int hn_direct_stat_rows(int n_img, int H, int W);
/* Plain-rows 1x1 GEMM (hn_conv_gemm_nt mode 0, one tap, bf16 out) with ONE PACKED WEIGHT MATRIX PER IMAGE: rows [n * rows_per_image,
 * (n + 1) * rows_per_image) of x0 use w + n * w_img_stride ([Nout][KP] bf16 each; rows_per_image % 128 == 0).  addend (optional): added
 * BEFORE the activation.  hn_scale_weight_gate makes such operands: out[n][co][k] = bf16(wp[co][k] * gate[n][k]).  Inference: an XBlock's
 * conv_block_3 with the SE gate folded into its weights per image (net/anynet.py:68-75) instead of a b * gate pass over the activation. */
int hn_conv_gemm_nt_imgw(const void* x0, int ld0, long M, int C0, const void* w, long w_img_stride, long rows_per_image, int Nout, int KP,
                         const float* bias, int act, void* out, int ldc, const void* addend, int ld_add, hipStream_t stream);
int hn_scale_weight_gate(const void* wp, const float* gate, void* out, int N, int Cout, int C, int KP, hipStream_t stream);
/* The same plain-rows GEMM on LEVEL-PACKED rows with the per-level eval-mode BatchNorm + activation of the det towers in the epilogue
 * (head_detect/detection.py:60-75): out = act(coef[l][0][c] * (x W^T + bias) + coef[l][1][c]), l = the level of the row; rows [nlev] =
 * rows of every level (multiples of 128: hn_bn_act_levels' argument), coef [nlev][4][Nout].  Inference: one launch per tower layer's
 * pointwise conv + BatchNorm + Swish instead of two. */
int hn_conv_gemm_nt_lvl(const void* x0, int ld0, long M, int C0, const void* w, int Nout, int KP, const float* bias, int act, void* out,
                        int ldc, const float* coef, int nlev, const long* rows, hipStream_t stream);
/* hn_conv_gemm_nt for level-packed plain rows whose fp32 output is the per-image concatenation of the pyramid levels (Regressor / Classifier,
 * head_detect/detection.py:36-60: the same convs on every level, torch.cat along the anchor axis): row image * H_l W_l + pixel of level l ->
 * out + image * img_stride + (sum_{k<l} H_k W_k + pixel) * ldc.  Levels start on row_align-aligned rows (a multiple of 128); alignment rows
 * are not stored.  M = the packed tensor's rows.  One launch instead of one hn_conv_gemm_nt per level. */
int hn_conv_gemm_nt_lvlout(const void* x0, int ld0, long M, int C0, const void* w, int Nout, int KP, const float* bias, int act, float* out,
                           int ldc, long img_stride, int n_img, int nlev, const int* H, const int* W, int row_align, hipStream_t stream);

/* hn_conv_gemm_nt with (a) an operand transform for modes 0/1 (bf16 output): the pixel operand is act(xscale[c]*x + xshift[c]) rounded to
 * bf16 and optionally multiplied by xgate[row / xhw][c] -- BatchNorm apply (+ReLU, + SE gate) of the producer folded into this conv's
 * operand path (net/anynet.py:67-70: conv_block_3 consumes SE(relu(bn2(conv_block_2)))), nothing materialised; and (b) an addend for the
 * staged bf16 epilogue: out = bf16(bf16(acc) + addend[pixel][cout]) (the residual-gradient add of an XBlock's conv_block_1 dgrad); with
 * ld_add < 0 the addend (row stride -ld_add) goes in BEFORE the activation: out = act(acc + bias + addend) -- inference with folded
 * BatchNorm: conv_block_3 + identity branch + ReLU of an XBlock in one launch (net/anynet.py:70-76).  add_mode 1 (mode 0, ld_add > 0,
 * even H, W): the addend lives on the stride-2 sub-grid [N][H/2][W/2] and is added at even (y, x) only -- the data gradient of a
 * stride-2 XBlock's shortcut conv joins the data gradient of its conv_block_1 (both read the same input, net/anynet.py:65-76) without a
 * zero-filled full-resolution tensor and a separate addition. */
int hn_conv_gemm_nt_ex(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1, int up, long M,
                       const void* w, int Nout, int KP, int taps, const float* bias, int act, void* out, int out_f32, int ldc, long rpi,
                       long img_stride, float* psum, float* psq, const float* xscale, const float* xshift, const float* xgate, long xhw,
                       int xact, const void* addend, int ld_add, int add_mode, hipStream_t stream);
/* hn_conv_gemm_nt_ex (no operand transform) whose statistics rows carry a backward reduction over (output, ez) instead of the output's
 * BatchNorm statistics, so that the reduce pass that would follow the launch is not needed (net/anynet.py:50-78 XBlock backward):
 *   emode 1: psum[tile][c] = sum_rows q * bf16(relu(ecoef[0][c] * ez + ecoef[1][c]))   the SE gate-gradient partials of dbg = dz3 W3
 *            (what hn_se_bwd_reduce_fused computes; psq may be NULL);
 *   emode 2: g = q * [ecoef[0][c] * ez + ecoef[1][c] > 0];  psum = sum g,  psq = sum g * (ez - ecoef[2][c]) * ecoef[3][c]
 *            (the partial sums of hn_bn_bwd_reduce_fused with ReLU; modes 0/1: one row per pixel tile, hn_nt_stat_rows; mode 5: one row
 *            per 16x16 patch, hn_direct_stat_rows), consumed by hn_bn_bwd_apply_fused.
 * q = the bf16-rounded output; ez = bf16 [M][ld_ez] on the output's pixel rows; ecoef = fp32 [4][Nout]; bf16 output, no activation. */
int hn_conv_gemm_nt_stat(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1, int up, long M,
                         const void* w, int Nout, int KP, int taps, const float* bias, int act, void* out, int out_f32, int ldc, long rpi,
                         long img_stride, float* psum, float* psq, const void* addend, int ld_add, int add_mode, int emode, const void* ez,
                         int ld_ez, const float* ecoef, hipStream_t stream);
/* Data-gradient GEMM of an identity XBlock's conv_block_1 (dx = dz1 W1 + g, the addend joining after the rounding), whose statistics rows
 * are the reduce pass of the PREVIOUS block's masked BatchNorm-3 backward over the dx it produces: psum / psq [ceil(M / 64)][Nout] =
 * sum g', sum g' (ez - mean) rstd with g' = dx [ey > 0] (ez = that block's conv_block_3 output, ey = its output, ecoef fp32 [4][Nout] its
 * BatchNorm-3 coefficients).  Consumed by hn_bn_bwd_apply_fused.  Only where hn_nt_stat_rows(M, Nout) == ceil(M / 64) (the 64 x 64 tiling);
 * HN_ERR_ARG otherwise.  Replaces the first pass of aten::native_batch_norm_backward of net/anynet.py:65-76's last BatchNorm. */
int hn_conv_gemm_nt_stat3(const void* x0, int ld0, long M, int C0, const void* w, int Nout, int KP, void* out, int ldc, float* psum, float* psq,
                          const void* addend, int ld_add, const void* ez, int ld_ez, const void* ey, int ld_ey, const float* ecoef,
                          hipStream_t stream);
/* wgrad: dw[Cout][Cin][taps] (PyTorch layout, fp32) = sum_pixel dz[pixel][cout] * X(pixel, tap)[c]; X modes 0..2 as above.
 * dz rows must be zero padded up to ldz >= Nout rounded up to 8.  workspace: fp32, size from hn_wgrad_plan.
 * Replaces aten::convolution_backward's weight gradient. */
int hn_wgrad_plan(int mode, int n_img, int H, int W, long M, int Nout, int KP, int taps, int* splits, long* rows_per_split, long* ws_bytes);
int hn_conv_gemm_tn(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1, int up, long M,
                    const void* dz, int ldz, int Nout, int KP, int taps, float* workspace, float* dw, hipStream_t stream);

/* ---- NHWC stencils (hn_stencil.hip) ----------------------------------------------------------------------------------------- */

/* Stem: x NCHW fp32 [N,3,H,W], w fp32 [32][3][3][3] -> z NHWC bf16 [N,H/2,W/2,32] (conv 3x3 s2 p1, net/anynet.py:12,17).
 * patches (optional): bf16 im2col rows [N*H/2*W/2][32] (27 taps in weight order + 5 zeros) so that the weight gradient is one
 * hn_conv_gemm_tn(mode 0) call on MFMA. */
int hn_stem_fwd(const float* x, const float* w, void* z, void* patches, int N, int H, int W, hipStream_t stream);

/* Grouped 3x3 conv, group width 8, pad 1, stride 1|2 (XBlock conv_block_2, net/anynet.py:34-35).
 * hn_gconv_pack: fp32 [C][8][3][3] -> wk[tap][i][G][o] and wd (o/i swapped; flip=1 also flips the taps for stride-1 dgrad).
 * The stride-2 kernels contract with packed bf16 dot products and want the CONTRACTION index contiguous: hn_gconv_fwd(stride 2) takes the
 * pack whose last index is the input channel = `wd` of hn_gconv_pack(flip = 0); hn_gconv_dgrad_s2 takes the pack whose last index is the
 * output channel = `wk`.  (stride 1, the VALU form: hn_gconv_fwd takes `wk`, its data gradient = hn_gconv_fwd on dz with `wd` of flip = 1.) */
int hn_gconv_pack(const float* w, void* wk, void* wd, int C, int flip, hipStream_t stream);
int hn_gconv_fwd(const void* in, int ldi, const void* wk, void* out, int ldo, int N, int Hi, int Wi, int C, int stride, hipStream_t stream);
int hn_gconv_dgrad_s2(const void* dz, int ldz, const void* wd, void* dx, int ldx, int N, int Hi, int Wi, int C, hipStream_t stream);
long hn_wgrad_chunks(long pixels, long items);
int hn_gconv_wgrad(const void* x, int ldx, const void* dz, int ldz, float* part, int N, int Hi, int Wi, int C, int stride,
                   hipStream_t stream);

/* Depthwise 3x3, stride 1, zero pad 1 (SeparableConvBlock.depthwise_conv, net/common.py:91-92,104). */
int hn_dw_pack(const float* w, void* wk, void* wkf, int C, hipStream_t stream);
int hn_dwconv_fwd(const void* in, int ldi, const void* wk, void* out, int ldo, int N, int H, int W, int C, hipStream_t stream);
/* level-packed form: rows of nlev (<= 5) pyramid levels [N,H[l],W[l],C] stacked in one tensor, one launch (the det-head towers apply the
 * same SeparableConvBlock to every level, head_detect/detection.py:30-35,67-72) */
/* row_align (>= 1): every level starts on a multiple of row_align rows ("ragged" packing: levels whose row count is not a multiple of the
 * 128-row GEMM / BatchNorm blocks are padded; this kernel writes ZEROS to the alignment rows so downstream sums can be corrected exactly).
 * accumulate = 1: out += (used as the data gradient of a packed map that feeds both det towers: the second tower adds in place). */
int hn_dwconv_fwd_levels(const void* in, int ldi, const void* wk, void* out, int ldo, int N, int C, int nlev, const int* H, const int* W,
                         int row_align, int accumulate, hipStream_t stream);
/* partial rows of hn_dwconv_wgrad*: strips = sum over levels of N * H * ceil(W / 4) (the kernel walks 4-pixel strips) */
long hn_dwconv_wgrad_blocks(long strips, int C);
int hn_dwconv_wgrad(const void* x, int ldx, const void* dz, int ldz, float* part, int N, int H, int W, int C, hipStream_t stream);
/* level-packed: part is fp32 [hn_dwconv_wgrad_blocks(total pixels, C)][C*9] */
int hn_dwconv_wgrad_levels(const void* x, int ldx, const void* dz, int ldz, float* part, int N, int C, int nlev, const int* H, const int* W,
                           int row_align, hipStream_t stream);
/* depthwise 3x3 backward in one pass over (dz, x): dx (optional; `accumulate`: added to an existing tensor) = conv(dz, flipped weights wf
 * [9][C] bf16) and the weight-gradient partial rows part [hn_dwconv_bwd_blocks(strips, C)][C*9] (strips = sum over levels of
 * N * H * ceil(W / 4)); reduce with hn_rows_reduce.  Reference: the backward of SeparableConvBlock.depthwise_conv (net/common.py:91-92,104). */
long hn_dwconv_bwd_blocks(long strips, int C);
/* ... for these maps, whichever form of the kernel hn_dwconv_bwd_levels takes (strips walked by lanes, or -- C <= 120, >= 512 tiles -- 8 x 16-pixel
 * tiles staged in LDS: two workgroups per CU, one partial row each); -1: bad argument.  Callers size `part` with THIS. */
long hn_dwconv_bwd_blocks_levels(int N, int C, int nlev, const int* H, const int* W);
int hn_dwconv_bwd_levels(const void* dz, int ldz, const void* x, int ldx, const void* wf, void* dx, int lddx, float* part, int N, int C,
                         int nlev, const int* H, const int* W, int row_align, int accumulate, hipStream_t stream);

/* 3x3/s2 max pools: mode 0 = zero pad right/bottom, zeros take part in the max (MaxPool2dStaticSamePadding, net/common.py:138-151);
 * mode 1 = nn.MaxPool2d(3,2,1) (head_lane/lanedetect.py:40).  Backward recomputes the arg-max (first maximum wins). */
int hn_maxpool_fwd(const void* in, int ldi, void* out, int ldo, int N, int H, int W, int C, int mode, hipStream_t stream);
int hn_maxpool_bwd(const void* in, int ldi, const void* dout, int ldd, void* dx, int ldx, const float* wscale, int N, int H, int W, int C,
                   int mode, hipStream_t stream);
/* two-pass form of the same backward (arg-max bytes of every window in arg_ws = N*(H/2)*(W/2)*C bytes, then a gather per input pixel) */
int hn_maxpool_bwd2(const void* in, int ldi, const void* dout, int ldd, void* dx, int ldx, const float* wscale, void* arg_ws, int N, int H,
                    int W, int C, int mode, int accumulate, hipStream_t stream);

/* Nearest x2 up-sampling and its backward (F.interpolate / nn.Upsample, net/bifpn.py:43-46, head_lane/lanedetect.py:10-13). */
int hn_up2_fwd(const void* in, int ldi, void* out, int ldo, int N, int H, int W, int C, hipStream_t stream);
int hn_sum2x2(const void* g, int ldg, void* out, int ldo, const float* wscale, int N, int H, int W, int C, int accumulate, hipStream_t stream);

/* BiFPN fusion node out = swish(sum_i w[i]*T_i(in_i)) (net/bifpn.py:177-231); mode[i]: 0 absent, 1 same res, 2 nearest x2 of a
 * half-res map, 3 zero-pad-same max-pool of a double-res map.  w: 3 fp32 in device memory. */
/* backward of w = relu(p)/(sum relu(p)+eps) (net/bifpn.py:179-180; the forward normalisation runs inside hn_fuse_fwd) from the per-block
 * partials of hn_fuse_bwd */
int hn_fuse_dweights(const float* pw, int blocks, const float* praw, int nw, float eps, float* dp, hipStream_t stream);
int hn_fuse_fwd(const void* const* in, const int* ld, const int* mode, const float* w, void* out, int ldo, int N, int H, int W, int C,
                hipStream_t stream);
/* the same with the normalisation of the raw fusion parameters inside the kernel (wn [3] is written for the backward pass) */
int hn_fuse_fwd_raw(const void* const* in, const int* ld, const int* mode, const float* praw, int nw, float eps, float* wn, void* out, int ldo,
                    int N, int H, int W, int C, hipStream_t stream);
/* `accumulate` / acc[i] = 1 in hn_fuse_bwd, hn_sum2x2 and hn_maxpool_bwd2: the destination already holds the gradient another consumer of
 * the same tensor wrote (a BiFPN map feeds 2-3 nodes, net/bifpn.py:186-231) and this consumer's contribution is added in place (fp32 add,
 * one bf16 rounding) -- the autograd engine's separate gradient-accumulation kernels disappear (ops.Share / ops.GradSlot). */
/* hn_fuse_bwd, din[i]: destination of input i's gradient for mode 1 (same grid) AND mode 2 inputs (nearest x2 of a half-resolution map:
 * din[i] is [N][H/2][W/2][.], the kernel walks the output in 2x2 quads and writes w_i * the quad sum itself; NULL: the caller runs
 * hn_sum2x2 over g); mode 3 (max-pooled) inputs: NULL, routed by hn_maxpool_bwd2. */
int hn_fuse_bwd_blocks(int N, int H, int W, int C);
int hn_fuse_bwd(const void* const* in, const int* ld, const int* mode, const float* w, const void* dout, int ldd, void* g, int ldg,
                void* const* din, const int* ldin, const int* acc, float* pw, int N, int H, int W, int C, hipStream_t stream);
/* hn_fuse_bwd that also writes the arg-max bytes of the pooling windows of every mode-3 input i with arg_out[i] != NULL ([N][H][W][C] uint8:
 * the kernel recomputes those windows anyway), and the second pass of hn_maxpool_bwd2 on its own, fed with such bytes (H, W = the pool's
 * INPUT resolution): one launch per pooled fusion input in the backward pass instead of two (net/bifpn.py:206-231) */
int hn_fuse_bwd_arg(const void* const* in, const int* ld, const int* mode, const float* w, const void* dout, int ldd, void* g, int ldg,
                    void* const* din, const int* ldin, const int* acc, float* pw, void* const* arg_out, int N, int H, int W, int C,
                    hipStream_t stream);
int hn_maxpool_bwd_from_arg(const void* arg, const void* dout, int ldd, void* dx, int ldx, const float* wscale, int N, int H, int W, int C,
                            int mode, int accumulate, hipStream_t stream);

/* Data gradient of a 3x3 conv over a reflection-padded (clamp = 0: ConvBlock / Conv3x3, head_seg/segmentation.py:40-58) or, in phase form,
 * replicate-padded (clamp = 1) input, written straight to the unpadded gradient: dx [N][H][W][Nout] (row stride ldo) = fold(full
 * correlation of dz with the transposed weights) [* ELU'(yprev)].  Replaces hn_conv_gemm_nt(mode 3) on the padded grid + hn_seg_fold:
 * the conv epilogue writes the interior of the padded grid in place and the one-pixel ring to `ring`
 * [N][hn_fold_ring_rows(H, W)][Nout] bf16; a border fix-up adds the ring to the 2 (H + W) pixels per image it mirrors onto.
 * phase_k = 0: dz [N][H][W][Cz], wt [Nout][9][KP];  phase_k > 0: dz = space-to-depth gradient [N][H][W][4 k], wt = transposed effective
 * weights.  HN_ERR_UNSUPPORTED (3) unless Nout % 8 == 0, Nout > 32, H, W >= 4 -- callers keep the two-pass path for those shapes. */
long hn_fold_ring_rows(int H, int W);
int hn_conv3x3_dgrad_fold(const void* dz, int ldz, int Cz, int n_img, int H, int W, const void* wt, int Nout, int KP, int phase_k, int clamp,
                          void* out, int ldo, const void* yprev, int ldy, void* ring, hipStream_t stream);
/* The same gradient (also) in SPACE-TO-DEPTH order: out_s2d [N][H/2][W/2][4 Nout] (row stride ld_s2d >= 4 Nout; H, W even), pixel (y, x)
 * channel c at row (y/2, x/2), channel ((y&1) 2 + (x&1)) Nout + c -- the form in which the phase-form block that produced this conv's input
 * consumes its gradient (that block's hn_space_to_depth_bf16 pass is not needed; head_seg/segmentation.py:92-104).  out = NULL: only that
 * form (a block without a skip operand reads nothing else); otherwise both.  yprev stays [N][H][W] (row stride ldy). */
int hn_conv3x3_dgrad_fold_s2d(const void* dz, int ldz, int Cz, int n_img, int H, int W, const void* wt, int Nout, int KP, int phase_k, int clamp,
                              void* out, int ldo, void* out_s2d, int ld_s2d, const void* yprev, int ldy, void* ring, hipStream_t stream);
/* Backward of ReflectionPad2d(1) (+ nearest x2, + channel split of the concat) for the seg decoder (head_seg/segmentation.py:40,92-99). */
int hn_seg_fold(const void* dvp, int ldv, int c0, void* out, int ldo, const void* yprev, int ldy, int N, int H, int W, int C, int up,
                hipStream_t stream);
/* up = 2 selects the replicate-padding fold (backward of mode 4). */

/* Pixel shuffle of the phase-decomposed final seg conv, gradient side: fp32 [N][2h][2w][k] -> zero-padded bf16 [N][h][w][ldo] with channel
 * (py*2+px)*k + o (the forward shuffle is the output conv's own epilogue, mode 4 above). */
int hn_space_to_depth(const float* dy, void* out, int ldo, int N, int h, int w, int k, hipStream_t stream);

/* bf16 form: in [N][2h][2w][k] (row stride ldi) -> out [N][h][w][4k], phase-major channels (the operand layout of the phase-form convs);
 * psum (optional, fp32 [hn_space_to_depth_blocks][k], needs 256 % (k/2) == 0): per-block channel sums of the tensor on its way through =
 * partial rows of the conv's bias gradient (head_seg/segmentation.py:60-67 ConvBlock bias), reduce with hn_rows_reduce */
int hn_space_to_depth_blocks(int N, int h, int w, int k);
int hn_space_to_depth_bf16(const void* in, int ldi, void* out, int N, int h, int w, int k, float* psum, hipStream_t stream);

/* Phase form of Conv3x3(ReflectionPad2d(1)(nearest_up2(x0))) (head_seg/segmentation.py:92-104, decoder blocks 1/3/5/7) on the LOW-resolution
 * grid: effective weights W_eff[(py,px,o)][c][dy][dx] = sum of the original taps that land on low-res offset (dy, dx) for output phase
 * (py, px); phase (py, px) is non-zero only on ky in {py, py+1}, kx in {px, px+1}, and only those 4 taps are visited (16 instead of 36
 * tap products per low-res pixel = 2.25x fewer MACs than convolving the up-sampled map).
 *   mode 4 (forward): x0 [N][H][W][C0] (low-res), w = packed W_eff [4k][9][KP(C0)], bias [4k], out bf16 [N][2H][2W][k] (row stride ldc),
 *           addend (optional, row stride ld_add, the output's layout) = pre-activation partial result of the full-resolution skip operand;
 *   mode 3 (data gradient): x0 = space-to-depth output gradient [N][H-2][W-2][4k] (H, W are the PADDED sizes), w = packed transposed
 *           weights [C][9][KP(4k)], out = padded-grid gradient [N][H][W][Nout] (fold with hn_seg_fold, up = 2).
 * hn_conv_gemm_tn_phase: gradient of W_eff (fp32 [4k][C0][3][3], zeros at unused taps) from x0 and the space-to-depth output gradient. */
int hn_conv3x3_phase(const void* x0, int mode, int n_img, int H, int W, int C0, int ld0, const void* w, int Nout, int KP, const float* bias,
                     int act, void* out, int ldc, int k, const void* addend, int ld_add, hipStream_t stream);
/* deploy forward of the seg head's output layer: Conv3x3(ReflectionPad2d(1)(nearest_up2(x0))) (head_seg/segmentation.py:101-104, phase
 * form: w = effective weights [4k][9][KP] from hn_pack_weight_ex, bias [4k]) fused with torch.argmax over the k classes
 * (model/model.py:197): mask int64 [N][2H][2W], first maximum wins; the fp32 logits are never written.  C0 <= 64, 4k <= 32. */
int hn_conv3x3_out_argmax(const void* x0, int n_img, int H, int W, int C0, int ld0, const void* w, int k, int KP, const float* bias, long* mask,
                          hipStream_t stream);
int hn_wgrad_plan_phase(int n_img, int H, int W, int Nout, int KP, int phase_span, int* splits, long* rows_per_split, long* ws_bytes);
int hn_conv_gemm_tn_phase(const void* x0, int n_img, int H, int W, int C0, int ld0, const void* dz, int ldz, int Nout, int KP, int phase_span,
                          float* workspace, float* dw, float* dbias_eff, hipStream_t stream);
/* hn_conv_gemm_tn without its slab reduce: job [8] (HOST array) receives {part, dw, splits, Nout, Cin, KP, taps, kind}; hn_wgrad_reduce_jobs
 * reduces up to four such jobs (njobs x 8 longs, host) in ONE launch -- the weight gradients of an XBlock's conv_block_1 / 2 / 3 / shortcut
 * (net/anynet.py:25-76) each had their own 5-7 us reduce launch.  dw is complete only after hn_wgrad_reduce_jobs. */
int hn_conv_gemm_tn_deferred(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1, int up, long M,
                             const void* dz, int ldz, int Nout, int KP, int taps, float* workspace, float* dw, long* job, hipStream_t stream);
int hn_wgrad_reduce_jobs(const long* jobs, int njobs, hipStream_t stream);
/* Deferred, grouped weight gradients of grouped 3x3 convs (group width 8, stride 1): up to 32 dw_j [C][8][3][3] (fp32) in one launch of the
 * patch kernel + one extract launch -- the conv_block_2 weight gradients of the identity XBlocks of one backbone stage (net/anynet.py:34-38).
 * jobs: HOST table, 9 int64 per job {x, dz, dw, n_img, H, W, C, ldx, ldz}; workspace: hn_gconv_wgrad_group_ws_bytes() bytes. */
long hn_gconv_wgrad_group_ws_bytes(const long* jobs, int njobs);
int hn_gconv_wgrad_group(const long* jobs, int njobs, float* workspace, hipStream_t stream);

/* Deferred parameter-gradient tails in ONE launch (ops.GradQueue): up to 64 small reductions that finish parameter gradients off the
 * backward pass's critical path.  jobs: HOST table, 8 int64 per job {kind, a, b, out, out2, n0, n1, n2}:
 *   kind 0: out[c] = sum_r a[r][c], a fp32 [n0 rows][n1 cols] -- partial-row folds of depthwise / stride-2 grouped conv weight gradients
 *           (reference ops: the weight gradient of net/common.py:85-95 depthwise convs, net/anynet.py:34-38 grouped convs);
 *   kind 1: BiFPN fusion-weight Jacobian (net/bifpn.py:179-180): a = per-block sums [n0][3], b = raw parameter [n1 <= 3], out = gradient
 *           [n1], n2 = float bits of eps;
 *   kind 2: SE MLP outer product (net/anynet.py:43-47): out[i][j] = sum_n a[n][i] * b[n][j], out2[i] = sum_n a[n][i]; a [n2][n0], b [n2][n1]. */
int hn_grad_tail(const long* jobs, int njobs, hipStream_t stream);

/* Deferred, grouped 1x1 weight gradients: up to 32 independent dw_j [Nout][Cin] (fp32) = dz_j^T . x_j in ONE GEMM launch (+ one slab-reduce
 * launch only when the jobs cannot fill the chip without a pixel split).  Replaces, for a whole backbone stage, the per-conv weight
 * gradients autograd's convolution_backward produces one by one (reference: conv_block_1 / conv_block_3 / shortcut of every XBlock,
 * net/anynet.py:29-33,52-60).  jobs: HOST table, 12 int64 per job {x0, dz, dw, mode, n_img, H, W, Cin, ld0, ldz, Nout, M}: x0 bf16 rows
 * (row stride ld0), dz bf16 [M][ldz], dw fp32 [Nout][Cin] out; mode 0: x rows = dz rows; mode 1: x is the [n_img][2H][2W] input of a
 * stride-2 1x1 conv with output grid [n_img][H][W] (M = n_img*H*W).  workspace: hn_wgrad_group_ws_bytes() bytes (-1: bad job table). */
long hn_wgrad_group_ws_bytes(const long* jobs, int njobs);
int hn_wgrad_group(const long* jobs, int njobs, float* workspace, hipStream_t stream);

/* hn_conv_gemm_tn (any mode but the grouped mode 5) that also returns the conv's bias gradient dbias [Nout] = column sums of dz (ConvBlock /
 * Conv3x3 bias, head_seg/segmentation.py:40-58; head output convs): one extra MFMA per k-step against an all-ones operand while the dz
 * fragments are in registers; the launch that reduces the weight-gradient slabs reduces the bias partials.
 * (dbias_eff of hn_conv_gemm_tn_phase: the same for the 4 k effective outputs, optional.) */
int hn_conv_gemm_tn_bias(const void* x0, const void* x1, int mode, int n_img, int H, int W, int C0, int C1, int ld0, int ld1, int up, long M,
                         const void* dz, int ldz, int Nout, int KP, int taps, float* workspace, float* dw, float* dbias, hipStream_t stream);

/* fp32 head-output gradient [N][rows][Nout] -> zero padded bf16 dz [N*rpi][ldz] (optionally times sigmoid'). */
int hn_head_grad(const float* dy, const float* y, long rpi, long img_stride, int lds, int Nout, void* dz, int ldz, long M, int sigmoid,
                 hipStream_t stream);

/* Stacks the pyramid levels a shared-weight head runs on (head_detect/detection.py:36-60 loops over them) into the level-packed operand of
 * the *_levels entry points: src[l] bf16 [N][H_l][W_l][C] (row stride ld[l]); level l's rows start at the row_align-aligned offset of dst
 * (row stride ldd); alignment rows are not written. */
int hn_pack_levels(const void* const* src, const int* ld, void* dst, int ldd, int N, int C, int nlev, const int* H, const int* W, int row_align,
                   hipStream_t stream);
/* hn_head_grad for every pyramid level of a level-packed head output in one launch: dy / y fp32 [N][sum_l H_l W_l][lds] (the per-image
 * concatenation of head_detect/detection.py:36-60), dz bf16 with level l's N*H_l*W_l rows at the row_align-aligned offsets of the packing
 * (as hn_dwconv_fwd_levels); alignment rows are not written. */
int hn_head_grad_levels(const float* dy, const float* y, long img_stride, int lds, int Nout, void* dz, int ldz, int N, int nlev, const int* H,
                        const int* W, int row_align, int sigmoid, hipStream_t stream);

/* ---- reductions, BatchNorm, SE, elementwise (hn_norm.hip) ------------------------------------------------------------------- */

long hn_colred_rows(long M, long align);
/* per-channel partial sums / sums of squares over row blocks of R rows: psum, psq are [ceil(M/R)][C] */
int hn_col_stats(const void* x, int ldx, long M, int C, long R, float* psum, float* psq, hipStream_t stream);
int hn_col_dot(const void* a, int lda, const void* b, int ldb, long M, int C, long R, float* pdot, float* psum, hipStream_t stream);
int hn_rows_reduce(const float* in, float* out, int G, int S, int C, float alpha, hipStream_t stream);
/* two arrays at once with ragged groups: outK[g][c] = sum of inK rows [g*S, min((g+1)*S, rows)), S = ceil(rows/G); folds the per-wave
 * partial statistic rows of a GEMM epilogue (or of hn_bn_bwd_reduce) before hn_bn_finalize / hn_bn_bwd_finalize */
int hn_rows_reduce2(const float* in1, const float* in2, float* out1, float* out2, int rows, int G, int C, hipStream_t stream);

/* Training-mode nn.BatchNorm2d (F.batch_norm): statistics -> (scale, shift, mean, rstd) + running-stat update. */
int hn_bn_finalize(const float* psum, const float* psq, int prows, int C, long count, const float* gamma, const float* beta, float eps,
                   float momentum, float* running_mean, float* running_var, float* scale, float* shift, float* mean, float* rstd,
                   hipStream_t stream);
int hn_bn_eval_coeff(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C, float* scale, float* shift,
                     hipStream_t stream);
/* out = act(scale*z + shift [+ (rscale*res + rshift | res)])   (BN apply + ReLU/Swish + residual add, net/anynet.py:65-76) */
int hn_bn_act(const void* z, int ldz, const float* scale, const float* shift, const void* res, int ldr, const float* rscale,
              const float* rshift, int act, void* out, int ldo, long M, int C, hipStream_t stream);
/* BN backward: g = dout*act'(pre) (or dout*[y>0] when the saved post-ReLU output y is given); partial sums of g and g*xhat */
int hn_bn_bwd_reduce(const void* dout, int ldd, const void* z, int ldz, const void* y, int ldy, const float* scale, const float* shift,
                     const float* mean, const float* rstd, int act, long M, int C, long R, float* pg, float* pgx, hipStream_t stream);
int hn_bn_bwd_finalize(const float* pg, const float* pgx, int prows, int C, long count, float* dgamma, float* dbeta, float* mg, float* mgx,
                       hipStream_t stream);
int hn_bn_bwd_apply(const void* dout, int ldd, const void* z, int ldz, const void* y, int ldy, const float* scale, const float* shift,
                    const float* mean, const float* rstd, const float* mg, const float* mgx, int act, void* dz, int lddz, void* gout, int ldg,
                    long M, int C, hipStream_t stream);

/* Level-packed BatchNorm with per-level parameters (bn_list[level][i] of the det towers, head_detect/detection.py:23,31-35,60,68-72).
 * rows[l] = tensor rows of level l INCLUDING its alignment rows (each a multiple of 128), count[l] = real rows; the alignment rows of the
 * conv output hold bf16(conv_bias) exactly and are subtracted from the statistics; coef = [nlev][4][C] (scale, shift, mean, rstd), red = [nlev][2][C].
 * gamma/beta/running_*/dgamma/dbeta are HOST arrays of nlev device pointers.  Partial-row inputs hold one row per `div` tensor rows. */
int hn_bn_finalize_levels(const float* psum, const float* psq, int div, int C, int nlev, const long* rows, const long* count,
                          const void* const* gamma, const void* const* beta, void* const* running_mean, void* const* running_var, float eps,
                          float momentum, const float* conv_bias, float* coef, hipStream_t stream);
int hn_bn_act_levels(const void* z, int ldz, const float* coef, int act, void* out, int ldo, int C, int nlev, const long* rows,
                     hipStream_t stream);
int hn_bn_bwd_reduce_levels(const void* dout, int ldd, const void* z, int ldz, const void* y, int ldy, const float* coef, int act, int C,
                            long R, int nlev, const long* rows, float* pg, float* pgx, hipStream_t stream);
int hn_bn_bwd_finalize_levels(const float* pg, const float* pgx, int div, int C, int nlev, const long* rows, const long* count,
                              void* const* dgamma, void* const* dbeta, float* red, float* zero_c, hipStream_t stream);
int hn_bn_bwd_apply_levels(const void* dout, int ldd, const void* z, int ldz, const void* y, int ldy, const float* coef, const float* red,
                           int act, void* dz, int lddz, int C, int nlev, const long* rows, hipStream_t stream);

/* ---- fused BatchNorm passes (hn_fused.hip) ----------------------------------------------------------------------------------
 * The finalize step of a training-mode nn.BatchNorm2d (partial statistics -> scale/shift/mean/rstd + running statistics; partial gradient
 * sums -> dgamma/dbeta + the two means) runs in the PROLOGUE of the elementwise kernel that consumes it: grid = (64-channel chunks,
 * row blocks of RB rows), every workgroup reduces the P partial rows of its own 64 channels.  Replaces hn_bn_finalize + hn_bn_act and
 * hn_bn_bwd_finalize + hn_bn_bwd_apply (and the hn_rows_reduce2 folds in front of them) for net/anynet.py:31,36,54,59,65-76,
 * net/common.py:98, net/bifpn.py:58-102, head_lane/lanedetect.py:45-64. */
/* rows per row block for M rows x C channels.  kind 0: apply passes (P = partial rows their prologue reduces); kind 1: reduce passes (their
 * row-block count becomes the P of the apply that follows).  align > 0 (rows per image): RB divides align, <= 8 row blocks per image. */
long hn_fused_row_block(long M, int C, long align, int P, int kind);
/* out = act(bn(z) [+ res]) [* gate[row / HW][c]].  P > 0: training mode, statistics from psum/psq [P][C] (count = rows normalised over),
 * row block 0 writes coef [4][C] (scale, shift, mean, rstd) and updates rm/rv; P == 0: use coef as is (null = identity); P < 0: eval mode
 * (running statistics).  out may be null when only pool is wanted; pool (optional) [ceil(M/RB)][C] = per-row-block channel sums of the
 * bf16-rounded output (SE squeeze, net/anynet.py:42,68); gate (optional, RB divides HW): SE excite applied to the rounded output. */
int hn_bn_apply_fused(const void* z, int ldz, long M, int C, const float* psum, const float* psq, int P, long count, const float* gamma,
                      const float* beta, float eps, float momentum, float* rm, float* rv, float* coef, const void* res, int ldr, int act,
                      void* out, int ldo, float* pool, const float* gate, long HW, long RB, hipStream_t stream);
/* SE excite + gated apply in one launch (net/anynet.py:44-48,68-69: the second nn.Conv2d of the SE block, its Sigmoid and the
 * `x * se(x)` product together with the BatchNorm + ReLU in front of them): gate[n][c] = sigmoid(b2[c] + sum_j w2[c][j] hid[n][j]) (stored
 * to `gate` [N][C] when non-null), out[row][c] = bf16(act(coef[0][c] z + coef[1][c])) * gate[row / HW][c]; coef null = identity */
int hn_se_gate_apply(const void* z, int ldz, const float* coef, int act, const float* hid, const float* w2, const float* b2, float* gate,
                     void* out, int ldo, int N, long HW, int C, int Cs, hipStream_t stream);
/* BatchNorm backward.  g = dout * act'(scale*z+shift), or dout * [y > 0] when the saved block output y is given (ReLU after the residual
 * add), or, with gate/dpool (SE, RB divides HW): g = (dout*gate[n][c] + dpool[n][c]/HW) * [scale*z+shift > 0] where dout is the gradient of
 * the gated tensor.  reduce: pg/pgx [ceil(M/RB)][C] partial sums of g and g*xhat.  apply: dz = scale*(g - mean g - xhat*mean(g*xhat)),
 * dgamma/dbeta written by row block 0, gout (optional) = g in bf16 (the residual branch's gradient), zero_c (optional) = C zeros (the
 * gradient of a conv bias that feeds this BatchNorm: written here instead of by a fill launch). */
int hn_bn_bwd_reduce_fused(const void* dout, int ldd, const void* z, int ldz, const void* y, int ldy, const float* coef, int act,
                           const float* gate, const float* dpool, long HW, long M, int C, long RB, float* pg, float* pgx, hipStream_t stream);
int hn_bn_bwd_apply_fused(const void* dout, int ldd, const void* z, int ldz, const void* y, int ldy, const float* coef, int act,
                          const float* gate, const float* dpool, long HW, const float* pg, const float* pgx, int P, long count, float* dgamma,
                          float* dbeta, void* dz, int lddz, void* gout, int ldg, long M, int C, long RB, float* zero_c, hipStream_t stream);
/* per-row-block channel sums / sums of squares [ceil(M/RB)][C] of a bf16 tensor (statistics of convs without a statistics epilogue) */
int hn_col_stats_fused(const void* x, int ldx, long M, int C, long RB, float* psum, float* psq, hipStream_t stream);
/* SE backward, first pass over (dbg = gradient of the gated tensor, z = pre-BN conv_block_2 output): b = relu(scale*z+shift),
 * pdot [ceil(M/RB)][C] = per-row-block sums of dbg*b (gate gradient), bg (optional) = b*gate[n][c] in bf16 (operand of conv_block_3's
 * weight gradient). */
int hn_se_bwd_reduce_fused(const void* dbg, int ldd, const void* z, int ldz, const float* coef, const float* gate, long HW, void* bg, int ldb,
                           float* pdot, long M, int C, long RB, hipStream_t stream);

/* SE gating x * gate[n][c] and its data-path backward (net/anynet.py:40-48,68-69). */
int hn_scale_rows(const void* x, int ldx, const float* gate, long HW, void* out, int ldo, long M, int C, hipStream_t stream);
int hn_se_bwd_apply(const void* dout, int ldd, const float* gate, const float* dpool, long HW, void* db, int ldb, long M, int C,
                    hipStream_t stream);

/* SE excitation MLP on the pooled [N,C] vectors: hid = relu(W1 p + b1) [N,Cs], gate = sigmoid(W2 hid + b2) [N,C]; backward gives dpool and
 * the four parameter gradients (dpre2 [N,C], dpre1 [N,Cs] are scratch).  Replaces the two 1x1 nn.Conv2d of net/anynet.py:44-47. */
int hn_se_mlp_fwd(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2, float* hid, float* gate, int N,
                  int C, int Cs, hipStream_t stream);
/* forms fed by partial rows (S per image): pooled = alpha * sum of the squeeze partials of hn_bn_apply_fused (stored to `pooled`), and
 * dgate = sum of the partials of hn_se_bwd_reduce_fused -- the reductions ride on the first MLP kernel instead of their own launches.
 * hn_se_mlp_fwd_parts with gate == NULL runs the first layer only (w2 / b2 unused): the second one is hn_se_gate_apply's prologue */
int hn_se_mlp_fwd_parts(const float* pool_part, int S, float alpha, const float* w1, const float* b1, const float* w2, const float* b2,
                        float* pooled, float* hid, float* gate, int N, int C, int Cs, hipStream_t stream);
int hn_se_mlp_bwd_parts(const float* dgate_part, int S, const float* gate, const float* hid, const float* pooled, const float* w1,
                        const float* w2, float* dpre2, float* dpre1, float* dpool, float* dw1, float* db1, float* dw2, float* db2, int N,
                        int C, int Cs, hipStream_t stream);
int hn_se_mlp_bwd(const float* dgate, const float* gate, const float* hid, const float* pooled, const float* w1, const float* w2,
                  float* dpre2, float* dpre1, float* dpool, float* dw1, float* db1, float* dw2, float* db2, int N, int C, int Cs,
                  hipStream_t stream);

/* ---- persistent stage kernel (hn_xstage.hip) ----------------------------------------------------------------------------------- */

/* The stride-1 identity XBlocks 1..d-1 of a backbone stage (net/anynet.py:65-76,84-86; training mode) as ONE launch: 256 workgroups of 512
 * threads, workgroup = (image, channel slice), the workgroups of an image on one XCD (read from HW_REG_XCC_ID), in-image exchanges through
 * that XCD's L2, BatchNorm statistics exchanged device-wide as tagged 8-byte granules.  Replaces nb x the 8 launches of ops/backbone.py
 * XBlockFn.forward and leaves every tensor that node saves: z1 / a / z2 / bg / z3 / out [nb][N*H*W][C] bf16 (dense rows), coef [nb][3][4][C]
 * (scale, shift, mean, rstd of BatchNorm 1 / 2 / 3), pooled / gate [nb][N][C], hid [nb][N][Cs]; running statistics updated in place.
 * tab = HOST table nb x 19 int64: {w1 packed [C][KP], w2 block-diagonal pack (hn_gconv_pack_diag wk), w3 packed, se.1.weight [Cs][C],
 * se.1.bias, se.3.weight [C][Cs], se.3.bias, then (gamma, beta, running_mean, running_var) of BatchNorm 1, 2, 3}.  x0 = input of the first
 * block [N*H*W][C].  alpha = 1 / (H*W).  ws = hn_xstage_ws_bytes() bytes, ZEROED ONCE when allocated and then owned by these launches
 * (counters / tags continue across launches; one workspace per device, launches on it serialised).  ws word 64 (uint32) is the status word:
 * 0 = every launch so far completed; non-zero = a bounded wait expired (the launch could not become co-resident): outputs invalid.
 * stamps (optional) = [nb][16] uint64 real-time-counter stamps of one workgroup.  mode 0: XCD-local counters; 1: agent-scope counters +
 * release / acquire fences (placement independent).  hn_xstage_supported: 0 = shape not covered (caller keeps the launch chain). */
long hn_xstage_ws_bytes(void);
int hn_xstage_supported(int N, int H, int W, int C, int Cs);
int hn_xstage_fwd(const long* tab, int nb, const void* x0, void* z1, void* a, void* z2, void* bg, void* z3, void* out, float* coef,
                  float* pooled, float* hid, float* gate, int N, int H, int W, int C, int Cs, float eps, float momentum, float alpha,
                  void* ws, long* stamps, int mode, hipStream_t stream);
/* The backward of the same blocks (ops/backbone.py XBlockFn.backward, 21 launches per block), last block first, in ONE launch of the same
 * shape: BatchNorm-backward sums cross XCDs as granules, dz3 / dz1 (the operands of the two data-gradient GEMMs) and the SE partials stay in
 * the image's XCD, a block's dx is the next block's dout of the same workgroup.  tab = HOST table nb x 5 int64 per block, forward block
 * order: {wt1 (transposed pack of conv_block_1, hn_pack_weight), wd2 (hn_gconv_pack_diag's data-gradient operand), wt3, se.1.weight,
 * se.3.weight}.  Inputs: dout = gradient of the last block's output [N*H*W][C] bf16 (dense rows); the forward's stacked tensors z1 / z2 /
 * z3 / out, coef, hid, gate.  Outputs: dz1 / dz2 / dz3 [nb][N*H*W][C] bf16 (operands of the weight gradients, which stay separate launches),
 * dx [N*H*W][C] (gradient of the first block's input), dgb [nb][3][2][C] = (dgamma, dbeta) of BatchNorm 1 / 2 / 3, dpre2 [nb][N][C] and
 * dpre1 [nb][N][Cs] = pre-activation gradients of the two SE layers (their outer products with hid / pooled are the SE weight gradients).
 * ws, stamps, mode: as hn_xstage_fwd (the same workspace). */
int hn_xstage_bwd(const long* tab, int nb, const void* dout, const void* z1, const void* z2, const void* z3, const void* out,
                  const float* coef, const float* hid, const float* gate, void* dz1, void* dz2, void* dz3, void* dx, float* dgb,
                  float* dpre2, float* dpre1, int N, int H, int W, int C, int Cs, void* ws, long* stamps, int mode, hipStream_t stream);

/* out += b0 [+ b1] [+ b2] (bf16 [M][C] tensors, fp32 sum, one rounding): the gradient sum of a multi-consumer map whose consumers return
 * separate gradients (ops.Share.backward; net/bifpn.py outputs feed three heads) in one launch instead of one per extra consumer */
int hn_add_n(void* out, int ldo, const void* b0, int ld0, const void* b1, int ld1, const void* b2, int ld2, long M, int C, hipStream_t stream);
/* op 0: a+b, 1: a*act'(b = post-activation; ELU/ReLU), 2: alpha*a, 3: act(a), 4: a*act'(b = pre-activation) */
int hn_eltwise(int op, const void* a, int lda, const void* b, int ldb, void* out, int ldo, long M, int C, int act, float alpha,
               hipStream_t stream);
int hn_add_strided2(void* dx, int ldx, const void* dxs, int lds, int N, int Ho, int Wo, int C, hipStream_t stream);
int hn_cast_f32_to_bf16_pad(const float* src, int lds, void* dst, int ldo, long M, int C, hipStream_t stream);

/* ---- losses (hn_loss.hip) --------------------------------------------------------------------------------------------------- */

/* Weighted cross entropy with ignore_index and optional top-k hardest pixels per image (CrossEntropyLoss.forward, use_focal=False,
 * head_seg/segmentation_loss.py:48-65).  logits: fp32 NHWC [N*HW][C] (row stride ldl); target: int64 or float32 class ids [N*HW];
 * k = int(top_k_ratio * HW).  The per-image k-th largest loss is found with an exact 3-level radix select instead of torch.sort.
 * ws: hn_seg_loss_ws_bytes(N, HW) bytes, written by fwd and read by bwd.  out[0] = mean loss; bwd writes dlogits (fp32, row stride ldd). */
long hn_seg_loss_ws_bytes(int N, long HW);
int hn_seg_loss_fwd(const float* logits, int ldl, int C, const void* target, int target_is_float, const float* cw, int ignore_index, int N,
                    long HW, int use_topk, long k, void* ws, float* out, hipStream_t stream);
int hn_seg_loss_bwd(const float* logits, int ldl, int C, const void* target, int target_is_float, const float* cw, int ignore_index, int N,
                    long HW, int use_topk, long k, const void* ws, const float* gout, float* dlogits, int ldd, hipStream_t stream);
/* the same gradient in the layout the phase-form 5-class output conv's backward consumes (hn_space_to_depth of dlogits, rounded to bf16):
 * dz bf16 [N][H/2][W/2][ldz], channel (py*2+px)*C + c of low-res pixel (y, x) = dlogits(2y+py, 2x+px, c), zeros in [4C, ldz) -- the fp32
 * dlogits tensor and the pass over it are skipped (head_seg/segmentation.py:101-104 -> segmentation_loss.py:48-65 chain) */
int hn_seg_loss_bwd_s2d(const float* logits, int ldl, int C, const void* target, int target_is_float, const float* class_weights,
                        int ignore_index, int N, int H, int W, int use_topk, long k, const void* ws, const float* gout, void* dz, int ldz,
                        hipStream_t stream);
/* Focal variant of the seg loss (CrossEntropyLoss.forward with use_focal, head_seg/segmentation_loss.py:31-46; cfgs/hydranet_joint_small_backbone.yml):
 * p = softmax + 1e-8, t = one_hot + 1e-8, loss = mean over all N*HW pixels of sum_c t_c * (-alpha (1-p_c)^gamma log(p_c) w_c).  logits fp32
 * [N*HW][ldl], target float32 / int64 class ids (no ignore_index on this path, as in the reference); ws: fp32 [hn_seg_loss_blocks(N, HW)]. */
int hn_seg_loss_blocks(int N, long HW);
int hn_seg_focal_fwd(const float* logits, int ldl, int C, const void* target, int target_is_float, const float* class_weights, float gamma,
                     float alpha, int N, long HW, void* ws, float* out, hipStream_t stream);
int hn_seg_focal_bwd(const float* logits, int ldl, int C, const void* target, int target_is_float, const float* class_weights, float gamma,
                     float alpha, int N, long HW, const float* gout, float* dlogits, int ldd, hipStream_t stream);
/* Detection loss (FocalLoss.forward, head_detect/detection_loss.py:132-267): cls fp32 [N][A][K] (post-sigmoid), reg [N][A][4], anchors
 * [A][4] (y1,x1,y2,x2), ann [N][Mx][5] (x1,y1,x2,y2,class; rows with class -1 are padding).  out[0] / out[1] = batch-mean classification /
 * regression loss.  assign: int16 [N][A]; part: fp32 [N][hn_det_loss_blocks(A)][3]; npos: fp32 [N] (all written by fwd, read by bwd). */
int hn_det_loss_blocks(int A);
int hn_det_loss_fwd(const float* cls, const float* reg, const float* anchors, const float* ann, int N, int A, int K, int Mx, void* assign,
                    float* part, float* npos, float* out, hipStream_t stream);
int hn_det_loss_bwd(const float* cls, const float* reg, const float* anchors, const float* ann, int N, int A, int K, int Mx,
                    const void* assign, const float* npos, const float* gout, float* dcls, float* dreg, hipStream_t stream);
/* torch.argmax(seg, dim=1) of deploy mode (model/model.py:197): fp32 NHWC logits -> int64 class ids, first maximum wins. */
/* Lane losses (head_lane/lanedetect_loss.py:18-78).  cls: logits / one-hot target fp32 [M][2]; out[0] = positive term, out[1] = OHEM
 * negative term (NEGATIVE_RATIO 15, ALPHA 10: the k-th smallest background log-prob is found by a radix select, not a sort);
 * lsm [M][2], pmask [M] bytes and aux[4] = {threshold, max(#pos,1), #pos, #neg} are kept for the backward pass and the location loss.
 * loc: pred / target fp32 [M][L]; the x`alpha` weights sit at columns wcol, wcol+1 (the reference hard-codes wcol = 160). */
int hn_lane_cls_loss_fwd(const float* logits, const float* target, long M, float neg_ratio, float alpha, float* lsm, void* pmask,
                         float* out, float* aux, hipStream_t stream);
int hn_lane_cls_loss_bwd(const float* lsm, const void* pmask, const float* aux, const float* gpos, const float* gneg, float alpha, long M,
                         float* dlogits, hipStream_t stream);
int hn_lane_loc_loss_fwd(const float* pred, const float* target, const void* pmask, const float* aux, long M, int L, int wcol, float alpha,
                         float* rowloss, float* rownorm, float* out, hipStream_t stream);
int hn_lane_loc_loss_bwd(const float* pred, const float* target, const void* pmask, const float* rownorm, const float* aux,
                         const float* gout, long M, int L, int wcol, float alpha, float* dpred, hipStream_t stream);
/* Greedy NMS on the device for the detection post-process (head_detect/detection_loss.py:70-108; torchvision batched_nms semantics):
 * boxes fp32 [K][4] (x1,y1,x2,y2) sorted by descending score with their class offsets added; suppress j > i when IoU > threshold, IoU in
 * separately rounded fp32 ops (bit-identical decisions to the host path).  mask: hn_nms_mask_words(K) uint64 scratch; keep: K bytes. */
long hn_nms_mask_words(int K);
int hn_nms_sorted(const float* boxes, int K, float iou_threshold, void* mask, void* keep, hipStream_t stream);
/* ---- stages either side of the hot path (hn_post.hip; SURVEY.md section 8(f)) ---------------------------------------------------- */

/* Detection post-process for a whole batch on the device (head_detect/detection_loss.py:7-108 BBoxTransform + ClipBoxes + postprocess;
 * torchvision.ops.batched_nms semantics restated: stable descending-score order, suppress when IoU > threshold, classes separated by an
 * offset of class_id * (max kept coordinate + 1)).  anchors fp32 [A][4] (y1,x1,y2,x2), regression [N][A][4], classification [N][A][K]
 * (post-sigmoid).  cap <= 32768 = per-image capacity.  Per image n: kept[n] boxes in descending-score order in rois [N][cap][4] (x1,y1,x2,y2),
 * class_ids int64 [N][cap], scores [N][cap]; total[n] = anchors over the threshold (total[n] > cap: overflow, results invalid).
 * ws: hn_det_post_ws_bytes(N, cap) bytes of scratch. */
long hn_det_post_ws_bytes(int N, int cap);
int hn_det_postprocess(const float* anchors, const float* regression, const float* classification, int N, int A, int K, int img_h, int img_w,
                       float threshold, float iou_threshold, int cap, void* ws, float* rois, long* class_ids, float* scores, int* kept,
                       int* total, hipStream_t stream);

/* Lane decode + lane NMS (LaneHeader.decode, head_lane/lanedetect.py:103-116 = softmax + LaneCodec.decode_lane, lane_codec.py:116-219 +
 * nms_with_pos, lane_codec_utils.py:487-543) for a batch: one workgroup per image.  predict_cls fp32 [N][hw][2] (logits), predict_loc fp32
 * [N][hw][2*ppl+2], hw = (W/stride)*(H/stride) <= 7168 (anchors are walked with a workgroup stride; 1152x1920 has 2160).  Outputs: X [N][hw][ppl] = x of anchor a at line position p (start[a] <= p < end[a]),
 * prob / start / end [N][hw] per anchor, order [N][hw] = candidates in descending-prob order (counts[n] of them), keep [N][hw] = 1 for the
 * candidates that survive. */
int hn_lane_decode_nms(const float* predict_cls, const float* predict_loc, int N, int W, int H, int stride, int ppl, float exist_threshold,
                       float nms_threshold, int use_mean, float margin, float* X, float* prob, int* start, int* end, int* order, int* keep,
                       int* counts, hipStream_t stream);

/* Input pre-processing (demo.py:26-50,186-196; dataset/utility.py:213-227): uint8 BGR frames [N][Hs][Ws][3] -> bilinear resize (cv2.resize
 * INTER_LINEAR fixed-point form for 8-bit images) -> RGB -> (v/255 - mean)/std -> fp32 [N][3][Hd][Wd]. */
int hn_preprocess_bgr(const void* src, int N, int Hs, int Ws, float* dst, int Hd, int Wd, hipStream_t stream);

/* Segmentation overlay = SegmentHeader.decode (head_seg/segmentation.py:107-125; deploy/src/model/hydranet_model.cpp:758) after the
 * arg-max: mask int64 [N][H][W] -> colour LUT uint8 [ncls][3] (ids without a colour stay black) -> cv2.resize to the frame size (the
 * reference passes cv2.INTER_NEAREST in the `dst` position, so the resize is the default 8-bit INTER_LINEAR) -> cv2.addWeighted(frame,
 * 0.8, colours, 0.5, 0) with float32 arithmetic, round-half-even, saturation.  frames / out: uint8 [N][Ho][Wo][3]. */
int hn_seg_overlay(const long* mask, int N, int H, int W, const void* lut, int ncls, const void* frames, void* out, int Ho, int Wo,
                   hipStream_t stream);

/* Lane F1 (head_lane/lane_metric.py:166-266, used by train.py:188,397,433): the bitwise IoU of lanes drawn as thick polylines.
 * hn_lane_raster paints every segment (p_i, p_{i+1}) of every lane's spline-interpolated polyline with width lane_width into the lane's
 * uint8 mask (cv2.line restated as "pixel centre within lane_width / 2 of the segment": parity with OpenCV's fill rules unpinned);
 * hn_lane_iou counts |mask_g & mask_p| and |mask| for ground-truth lanes g < G and predictions p < P (G, P <= 32) with integer atomics.
 * pts int32 [npts][2]; seg_lane / seg_first int32 [nseg]; masks uint8 [lanes][H][W] zeroed by the caller; inter uint64 [G][P] and area
 * uint64 [G + P] zeroed by the caller. */
int hn_lane_raster(const int* pts, const int* seg_lane, const int* seg_first, int nseg, int lane_width, int H, int W, void* masks,
                   hipStream_t stream);
int hn_lane_iou(const void* masks, int G, int P, long HW, void* inter, void* area, hipStream_t stream);

/* Streaming confusion counts for the segmentation mIoU (head_seg/seg_metrics.py:12-47): conf uint64 [(C+1)*(C+1)] += counts of
 * (pred, target) pairs, both clamped to C (the ignore bucket).  pred int64 [M]; target int64 or float32 [M]. */
int hn_seg_confusion(const long* pred, const void* target, int target_is_float, long M, int C, void* conf, hipStream_t stream);

int hn_argmax_channels(const float* logits, int ldl, int C, long M, long* out, hipStream_t stream);

/* HydraTrainer.cal_total_loss (model/train.py:192-203) in one launch: total = sum_g (sum_{i in g} x_i*w_i) * gw_g, left to right in fp32
 * without fused multiply-add (the reference rounds after every mul / add); xs = HOST array of n <= 8 device pointers to fp32 scalars,
 * w [n], gw [number of groups], grp [n] (non-decreasing group id per term) = HOST arrays.  out (optional) = total; grads (optional, needs
 * gout = d loss / d total on the device) [n] = (gout * gw_g) * w_i. */
int hn_weighted_sum(const void* const* xs, const float* w, const float* gw, const int* grp, int n, const float* gout, float* out, float* grads,
                    hipStream_t stream);

/* Adam step of all parameters in one launch (torch.optim.Adam as model/train.py:147 constructs it: L2 weight decay on the gradient, bias
 * correction, eps outside the square root; not amsgrad).  jobs (DEVICE) = n x 6 int64 {p, g, m, v (fp32 pointers), numel, first_block};
 * a block = 256 threads x 4 consecutive elements of one tensor; block_job (DEVICE int32 [total_blocks]) = job index of every block;
 * step = 1-based iteration count (bias corrections are computed on the host in double). */
int hn_adam_step(const long* jobs, const int* block_job, long total_blocks, double lr, double beta1, double beta2, double eps,
                 double weight_decay, long step, hipStream_t stream);

/* Many contiguous tensors copied in one launch (the gather of a gradient bucket before its all-reduce, train.py:130-137's DDP buckets):
 * jobs (DEVICE) = n x 4 int64 {src, dst, numel, first_block}, block = 256 threads x 4 elements, block_job (DEVICE int32) = job of every
 * block; kind 0: fp32 -> fp32, 1: fp32 -> bf16 (reduced-precision payload), 2: bf16 -> fp32. */
int hn_copy_many(const long* jobs, const int* block_job, long total_blocks, int kind, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
