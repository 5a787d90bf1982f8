/*
 * dasac_hip.h -- C-ABI of libdasac_hip.so: the MI355X (gfx950) kernels under the da-sac
 * per-step hot path.
 *
 * The reference (visinf/da-sac) has no FFI of its own: every kernel it runs is an implicit
 * ATen/cuDNN call made from models/{sac,deeplabv2,fcn,basenet}.py.  This header is the new
 * seam directly under those modules; each entry point names the reference
 * lines whose arithmetic it replaces.  INTEGRATION.md shows the ctypes binding a maintainer
 * of the reference would add.
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes; NCHW contiguous fp32 activations / weights, int64 labels
 *     (255 = ignore), bool masks as uint8;  no torch types anywhere in a signature;
 *   - `stream` is a hipStream_t (NULL = the legacy default stream);
 *   - return 0 on success, a negative DASAC_E* code otherwise; never throws, never allocates
 *     device memory, never synchronises the stream.  Scratch comes from the caller:
 *     `dasac_*_workspace(...)` returns the bytes a call needs (16-byte aligned pointer);
 *   - re-entrant per stream; dasac_last_error() is thread-local.
 */
#ifndef DASAC_HIP_H
#define DASAC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DASAC_OK 0
#define DASAC_EINVAL (-1)   /* bad argument (null pointer, shape, unsupported geometry) */
#define DASAC_EWORKSPACE (-2) /* workspace too small */
#define DASAC_ELAUNCH (-3)  /* HIP launch / runtime error, see dasac_last_error() */

typedef void* dasac_stream_t; /* hipStream_t */

int dasac_version(void);                 /* ABI version, currently 1 */
const char* dasac_last_error(void);      /* thread-local message of the last failure */
int dasac_device_info(int* cu_count, int* wave_size, char* arch, size_t arch_len);
/* Compute units this process leaves to kernels that run beside its own -- RCCL's all-reduce kernels when the gradient
 * reduction is overlapped with the backward pass (train.py:104 DistributedDataParallel; here dasac_hip.parallel).  The
 * persistent stream-K grid (3 workers per CU, equal matrix work per worker) and the grid cap of the streaming kernels are
 * sized to 256 - n CUs, so that a collective's workgroups do not land on CUs whose workers then finish last.  n is rounded
 * up to a multiple of 8 and capped at 128; 0 (the default, or DASAC_SK_RESERVE_CUS in the environment) = the whole chip.
 * dasac_set_reserved_cus returns the previous value.  Results do not depend on n beyond the summation order of stream-K tiles. */
int dasac_reserved_cus(void);
int dasac_set_reserved_cus(int n);

/* ------------------------------------------------------------------------------------------
 * Pseudo-label extraction -- models/sac.py:154-187 (`SAC._pseudo_labels_probs`).
 *   (m,k) = max/argmax_c probs (ties -> lowest c);  peak[b,c] = max{m : k == c};
 *   thr[b,c] = max(peak*upper*disc[c], lower)  (fp32, that op order; disc may be NULL);
 *   labels = k if m > thr[b,k] else 255;  labels = 255 where ignore != 0.
 * probs [B,C,HW] f32, ignore [B,HW] u8 (may be NULL), labels [B,HW] i64, max_conf [B,HW] f32,
 * max_idx [B,HW] i64 (may be NULL: the reference never reads it, sac.py:357).  Requires
 * lower > 0 and C <= 64.  Integer outputs are bit-exact w.r.t. the CPU reference.
 */
size_t dasac_pseudo_labels_workspace(int B, int C, int64_t HW);
int dasac_pseudo_labels(const float* probs, const uint8_t* ignore, const float* disc,
                        float upper, float lower, int B, int C, int64_t HW,
                        int64_t* labels, float* max_conf, int64_t* max_idx,
                        void* workspace, size_t ws_bytes, dasac_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Convolutions as implicit GEMM on fp32 MFMA -- every nn.Conv2d of models/deeplabv2.py
 * (:59,65-66,70,107,122,147-148,262-263) and models/fcn.py (:49,53,57,78,88), with the
 * BN(eval)/bias/residual/ReLU chain of Bottleneck.forward (deeplabv2.py:79-99) in the epilogue.
 *
 * A convolution is described by "tap branches" (kh,kw,dilation,padding): one branch for a plain
 * conv, four for the ASPP sum of deeplabv2.py:112-116 evaluated as a single contraction.
 *   K = (sum of kh*kw) * C;  `order` 0: k = tap*C + c (tap-major, required by the weight gradient);
 *   `order` 1: k = ((c/16)*taps + tap)*16 + c%16 (chunk-major, C % 16 == 0: the taps of a 16-channel
 *   chunk are contracted back to back so their shifted re-reads stay in L2) -- table and packed
 *   weights of one GEMM must use the same order.
 * dasac_conv_table   builds the gather table [Kpad][4] int32 for planes of plane_h x plane_w;
 *                    transposed=1 gives the data-gradient geometry (C = Cout, dh = pad - kh*dil).
 * dasac_conv_pack    lays W [Cout,Cin,kh,kw] out as the k-interleaved [Kpad/4][Mpad][4] GEMM operand
 *                    (transposed=1: rows (tap,co), columns ci), optionally scaled per co by
 *                    `scale` = folded BN gamma*invstd -- forward and data-gradient both contract
 *                    against scale*W, so the epilogue only adds the shift.
 *                    Call once per branch (tap0 = first tap of the branch, total_taps = all).
 * dasac_conv_gemm    out[n,m,oh*os,ow*os] = epi( sum_k packed[k][m] * x[n, c, oh*stride+dh, ow*stride+dw] )
 *                    epi: v + shift[m] (+res) (ReLU) (zeroed where mask <= 0 / where its mask bit is clear).
 *                    ReLU patterns may travel as ONE BIT per element instead of the fp32 activation:
 *                    `relu_bits_out` (with relu = 1) receives bit (pix & 31) of word [m][pix >> 5] = (out > 0), pix =
 *                    flattened (n, oh, ow) index, dasac_relu_bits_words(M, Nb*OH*OW) uint32 words; `mask_bits`
 *                    consumes such a pattern in the data-gradient GEMM whose output has the producer's shape
 *                    (1/32 of the mask bytes; exclusive with `mask`; both need ostride = 1 and
 *                    dasac_conv_gemm_bits_ok(M, Cx): the 128-row fp32 tile, Cx % 16 == 0).  The launch covers the
 *                    output pixels [pix_begin, pix_begin + pix_count) of the flattened (n, oh, ow) grid
 *                    (whole 128-pixel tiles; pix_count 0 = to the end); schedule 0 = pick, 1 = one block
 *                    per tile, 2 = persistent stream-K.  With a workspace of
 *                    dasac_conv_gemm_workspace() bytes (129 MB) schedule 0 may pick, for a long contraction whose
 *                    tile count leaves a ragged last round of resident workgroups: over the whole pixel range, ONE
 *                    launch of the leading whole rounds one block per tile + the remaining tiles cut into K-ranges
 *                    (split-K tail, dasac_conv_gemm_tail_split); with fewer tiles than resident workgroups, the
 *                    persistent stream-K schedule (equal matrix work per CU).  workspace NULL = one block per tile.
 *                    Every choice is a function of the shape and dasac_reserved_cus() alone: run-to-run identical bits.
 *                    The workspace must be ZERO-FILLED by the caller when it is allocated and belong to one
 *                    stream at a time: its hand-off flags are self-cleaning (every launch leaves them
 *                    zero), which saves a memset per launch.
 * dasac_conv_wgrad   partial weight gradients (split over pixels) into `workspace`;
 * dasac_conv_wgrad_finish  sums the splits, writes dW[co,ci,kh,kw] = scale[co]*G and, when `dot`
 *                    is given (single-branch convolutions only), the frozen-BN gamma gradient term sum_k W*G as
 *                    dasac_conv_wgrad_dot_rows(Cin, taps) PARTIAL rows dot[row][co] (row = a block of 64 input
 *                    channels; every element is written, nothing to zero): dasac_bn_param_grads adds the rows in a
 *                    fixed order, no atomics, so two runs give the same bits; `sum_dz`
 *                    (optional) receives sum over batch and pixels of dz per channel (d beta / d bias),
 *                    accumulated for free by the wgrad kernel while it streams dz.
 */
int dasac_conv_mpad(int M);
int dasac_conv_kpad(int K);
int dasac_conv_table(const int32_t* kh, const int32_t* kw, const int32_t* dil, const int32_t* pad,
                     int n_branches, int C, int plane_h, int plane_w, int transposed, int order,
                     int32_t* table, dasac_stream_t stream);
int dasac_conv_pack(const float* w, const float* scale, int Cout, int Cin, int taps, int tap0,
                    int total_taps, int transposed, int order, float* packed, dasac_stream_t stream);
size_t dasac_relu_bits_words(int M, int64_t Npix);
int dasac_conv_gemm_bits_ok(int M, int Cx);   /* 1: dasac_conv_gemm takes mask_bits / relu_bits_out for this (M, gathered channels) */
int dasac_conv_gemm(const float* x, const float* packed, const int32_t* table, float* out,
                    int Nb, int Cx, int H, int W, int OH, int OW, int stride, int M, int K,
                    int OutH, int OutW, int ostride,
                    const float* shift, const float* res, const float* mask,
                    const uint32_t* mask_bits, uint32_t* relu_bits_out,
                    int relu, int pix_begin, int pix_count, int schedule,
                    void* workspace, size_t ws_bytes, dasac_stream_t stream);
/* Split-bf16 ("bf16x3") variant of the same contraction: every fp32 operand x is split into
 * head = bf16(x) and tail = bf16(x - head) and x*y is evaluated as xh*yh + xh*yl + xl*yh on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation (relative error of a product <= ~2^-15, typically
 * 2^-17; the dropped tail*tail term is below 2^-16).  The activations are split inside the kernel
 * (NCHW fp32 in HBM, nothing else changes), the weights once by dasac_conv_pack_x3 from the output of
 * dasac_conv_pack (same byte size).  Same table, epilogue, workspace and schedule as dasac_conv_gemm;
 * M must exceed 32.  Opt-in: the host picks it per call (DASAC_PRECISION=bf16x3 in the Python layer). */
int dasac_conv_pack_x3(const float* packed, int M, int K, void* packed_x3, dasac_stream_t stream);
int dasac_conv_gemm_x3(const float* x, const void* packed_x3, const int32_t* table, float* out,
                       int Nb, int Cx, int H, int W, int OH, int OW, int stride, int M, int K,
                       int OutH, int OutW, int ostride,
                       const float* shift, const float* res, const float* mask,
                       const uint32_t* mask_bits, uint32_t* relu_bits_out,
                       int relu, int pix_begin, int pix_count, int schedule,
                       void* workspace, size_t ws_bytes, dasac_stream_t stream);
/* dasac_conv_gemm (fp32, no residual / mask / bit masks) whose epilogue ALSO leaves the per-channel statistics a
 * batch-statistics BatchNorm behind the convolution needs (nn.SyncBatchNorm in train mode: deeplabv2.py:15 with
 * models/__init__.py:29 BASELINE, train.py:281-289): stats [dasac_conv_gemm_stats_tiles(Nb, OH, OW)][2][dasac_conv_mpad(M)]
 * floats = per 128-pixel tile the sum and the sum of squares of every output row as stored (shift / bias included).  Every
 * slot is written by exactly one workgroup (also under the stream-K schedule); dasac_bn_train_finalize_tiles /
 * dasac_bn_tile_stats_reduce add the tiles in a fixed order -- no atomics, run-to-run identical.  Removes the stand-alone
 * statistics pass over the activation (one full read per BN layer).  Needs dasac_conv_gemm_stats_ok(M, Cx) (the 128-row
 * tile, Cx % 16 == 0) and ostride = 1. */
int dasac_conv_gemm_stats_ok(int M, int Cx);
int dasac_conv_gemm_stats_tiles(int Nb, int OH, int OW);
int dasac_conv_gemm_stats(const float* x, const float* packed, const int32_t* table, float* out,
                          int Nb, int Cx, int H, int W, int OH, int OW, int stride, int M, int K,
                          int OutH, int OutW, int ostride, const float* shift, const float* res, int relu,
                          int pix_begin, int pix_count, int schedule, void* workspace, size_t ws_bytes,
                          float* stats, dasac_stream_t stream);
size_t dasac_conv_gemm_workspace(void);
int dasac_conv_gemm_schedule(int Nb, int OH, int OW, int M, int K);   /* 1 = stream-K, 0 = one block per tile */
/* > 0: dasac_conv_gemm(schedule 0, the whole pixel range, workspace given) runs this shape as ONE launch -- the leading whole
 * rounds of resident workgroups one block per tile, the remaining tiles cut into that many K-ranges of one block each (split-K
 * tail: the later ranges deposit their accumulators in the workspace, the piece with the first K-steps adds them and runs
 * the epilogue; deterministic: a function of the shape and the reserved CUs alone).  0: it does not. */
int dasac_conv_gemm_tail_split(int Nb, int OH, int OW, int M, int K);
size_t dasac_conv_wgrad_workspace(int Nb, int OH, int OW, int M, int K);
int dasac_conv_wgrad(const float* dz, const float* x, const int32_t* table,
                     int Nb, int Cx, int H, int W, int OH, int OW, int stride, int M, int K,
                     void* workspace, size_t ws_bytes, dasac_stream_t stream);
/* split-bf16 variant of dasac_conv_wgrad (same arguments, workspace layout and dasac_conv_wgrad_finish):
 * both operands are split on their way into LDS, three bf16 MFMAs per product, fp32 accumulate. */
int dasac_conv_wgrad_x3(const float* dz, const float* x, const int32_t* table,
                        int Nb, int Cx, int H, int W, int OH, int OW, int stride, int M, int K,
                        void* workspace, size_t ws_bytes, dasac_stream_t stream);
int dasac_conv_wgrad_dot_rows(int Cin, int taps);
int dasac_conv_wgrad_finish(const void* workspace, int Nb, int OH, int OW, int M, int K,
                            const float* w, const float* scale, float* dw, float* dot,
                            float* sum_dz, int Cin, int taps, int tap0, dasac_stream_t stream);

/* Tap-expanded evaluation of few-output-channel, many-tap convolutions -- the ASPP classifiers
 * (deeplabv2.py:101-116): Y[(tap,co)] = 1x1 GEMM over taps*Cp channels (dasac_conv_gemm with weights
 * from dasac_conv_pack_expanded), out = bias + sum_tap shift(Y) (dasac_tap_gather); backward:
 * D = dasac_tap_scatter(dout), then plain 1x1 wgrad / dgrad on D.  Cp >= Cout pads the channels so
 * that taps*Cp % 16 == 0.  Branches are (kh,kw,dilation,padding) as for dasac_conv_table. */
int dasac_conv_pack_expanded(const float* w, int Cout, int Cin, int taps, int tap0, int total_taps,
                             int Cp, int transposed, float* packed, dasac_stream_t stream);
int dasac_tap_gather(const float* y, const int32_t* kh, const int32_t* kw, const int32_t* dil,
                     const int32_t* pad, int n_branches, int Cp, int Cout, const float* bias, int B,
                     int H, int W, float* out, dasac_stream_t stream);
int dasac_tap_scatter(const float* dout, const int32_t* kh, const int32_t* kw, const int32_t* dil,
                      const int32_t* pad, int n_branches, int Cp, int Cout, int B, int H, int W,
                      float* d, dasac_stream_t stream);
int dasac_conv_wgrad_finish_expanded(const void* workspace, int Nb, int OH, int OW, int E, int Cin,
                                     float* dw, int Cout, int taps, int tap0, int Cp,
                                     dasac_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * SAC head (models/sac.py) -- HBM-bound streaming kernels over [B,C,H,W] fp32, C <= 32.
 *
 * dasac_upsample_softmax  F.interpolate(bilinear, align_corners=True) of logits [B,C,h,w]
 *     (deeplabv2.py:217, sac.py:275); optional outputs: `up` (upsampled logits), `probs` =
 *     softmax zeroed where ignore != 0 (sac.py:276,282), `class_sums[C]` (double) = sum over
 *     batch and pixels of the unmasked softmax (sac.py:108), accumulated order-independently (Q32 fixed point,
 *     B*H*W < 2^31) so that the class prior is bit-identical from run to run.
 * dasac_upsample_bwd      transpose of the upsampling: grad_low = gscale[0] * U^T grad_up
 *     (gscale: device scalar or NULL), planes = B*C.
 * dasac_ce_loss           mode 0: mean over ALL pixels of CE(ignore 255) (deeplabv2.py:223-224,
 *     sac.py:119-132); mode 1: focal_ce_conf with its [B,B,H,W] broadcast (sac.py:134-149):
 *     loss = sum_hw (sum_i conf_i)(sum_j ce_j)/(B*B*HW).  class_weight [C] or NULL.  dlogits
 *     (optional) receives gscale[0] * d loss / d logits (gscale: device scalar = upstream
 *     gradient of the loss, NULL = 1); per_class (optional) [C] as sac.py:138-145 (order-independent Q28 fixed-point
 *     accumulation: per-pixel values are clamped to |ce| <= 4096 there; the loss itself is not clamped).
 * dasac_warp_affine       grid_sample(x, affine_grid(theta), bilinear, zeros, align_corners=False)
 * dasac_warp_pool         sac.py:289-305 with _avg_pool (mode 0, :238-269) or _minentropy_pool
 *     (mode 1, :218-236): probs [N*T,C,H,W] -> pooled [N,C,H,W], mask [N,H,W]; `aligned`
 *     (optional) = teacher_aligned diagnostic [N*T,C,H,W].  theta == theta_inv == NULL: `probs` are
 *     views that are already aligned and coverage-weighted (the pooling functions on their own).
 * dasac_warp_back         sac.py:309-311: refined[b] = warp(pooled[b/views_per_group]) * warp(mask); any H, W >= 1 (W >= 2
 *     fetches the two taps of a source row as one 8-byte pair, one-column maps take four scalar taps: the same arithmetic)
 * dasac_class_state       sac.py:104-117 running prior update (if update) and the derived
 *     vectors disc = 1-exp(-chi/beta) (:152), focal = (1-max(chi,0))^p (:120); any may be NULL.
 */
int dasac_upsample_softmax(const float* logits, int B, int C, int h, int w, int H, int W,
                           const uint8_t* ignore, float* up, float* probs, double* class_sums,
                           dasac_stream_t stream);
size_t dasac_upsample_bwd_workspace(int planes, int H, int w);
int dasac_upsample_bwd(const float* grad_up, int planes, int h, int w, int H, int W,
                       const float* gscale, float* grad_low, void* workspace, size_t ws_bytes,
                       dasac_stream_t stream);
size_t dasac_ce_loss_workspace(int B, int C, int64_t HW);
int dasac_ce_loss(const float* logits, const int64_t* labels, const float* class_weight,
                  const float* conf, int B, int C, int64_t HW, int mode, const float* gscale,
                  float* loss, float* dlogits, float* per_class, void* workspace, size_t ws_bytes,
                  dasac_stream_t stream);
/* Gradient of dasac_ce_loss w.r.t. the LOW-resolution logits the upsampled ones came from (deeplabv2.py:217 then
 * :223-224 / sac.py:119-149): grad_low [B,C,h,w] = gscale[0] * U^T (d loss / d logits_up) without materialising the
 * full-resolution gradient (same arithmetic and summation order as dasac_ce_loss(dlogits) + dasac_upsample_bwd).
 * The workspace holds the row buffer [B*C][H][w] and, behind it, H*W floats for the per-pixel confidence sums of mode 1
 * (added up once per launch instead of once per image; the size function reserves H*w*16 floats = any up-factor to 16; a
 * workspace with only the row buffer still works, every block then adds the B confidences itself). */
size_t dasac_ce_loss_bwd_low_workspace(int B, int C, int H, int w);
int dasac_ce_loss_bwd_low(const float* logits_up, const int64_t* labels, const float* class_weight, const float* conf,
                          int B, int C, int H, int W, int h, int w, int mode, const float* gscale, float* grad_low,
                          void* workspace, size_t ws_bytes, dasac_stream_t stream);
/* Inference (infer_val.py:160-163 and the result writer's argmax + trainId->labelId mapping, :60-65):
 * labels[b,y,x] = lut[argmax_c softmax(bilinear_ac(logits))[b,c,y,x]] (lut null: the class index), optional
 * conf = the winning probability.  1 (+4) bytes written per output pixel. */
int dasac_infer_labels(const float* logits, int B, int C, int h, int w, int H, int W, const uint8_t* lut,
                       uint8_t* labels, float* conf, dasac_stream_t stream);
int dasac_warp_affine(const float* x, const float* theta, int B, int C, int H, int W, float* out,
                      dasac_stream_t stream);
int dasac_warp_pool(const float* probs, const float* theta, const float* theta_inv, int N, int T,
                    int C, int H, int W, int mode, float tolerance, float* aligned, float* pooled,
                    float* mask, dasac_stream_t stream);
int dasac_warp_back(const float* pooled, const float* mask, const float* theta_inv, int B,
                    int views_per_group, int C, int H, int W, float* refined,
                    dasac_stream_t stream);
int dasac_class_state(float* running_conf, const double* class_sums, int B, int64_t HW, int C,
                      float beta, float stat_momentum, int update, float focal_p, float* disc,
                      float* focal, dasac_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Around the convolutions.
 * dasac_bn_fold          frozen BatchNorm (models/__init__.py:29, basenet.py:97-100) as per-channel
 *     scale/shift for the conv epilogue: scale = gamma/sqrt(var+eps), shift = beta-(mean-b)*scale.
 * dasac_bn_param_grads   gamma/beta (and conv-bias) gradients of the folded form from
 *     sum_rows dot[row][c] = sum dz*conv(x) (the `dot_rows` partial rows of dasac_conv_wgrad_finish) and sum_dz[c]
 *     (dasac_channel_sums or the wgrad kernel).
 * dasac_maxpool_fwd/bwd  nn.MaxPool2d (deeplabv2.py:126 ceil_mode 3x3/2; VGG 2x2/2); caller passes
 *     the resolved OH/OW.  bwd with relu_mask folds the ReLU backward of the producer.  `argmax` is an opaque byte per
 *     output handed from fwd to bwd: bits 0-6 the winning position kh*k + kw, bit 7 (windows up to 11x11) = pooled value > 0,
 *     so that the ReLU-folding backward does not read the pooled tensor (`y` is then unused; larger windows do read it).
 * dasac_ema_update       momentum teacher (sac.py:83-102) over all tensors in one launch:
 *     out[0] = sum_t ||slow_t - fast_t||_2 (before the update); slow = slow*m + fast*(1-m).
 *     `pairs`: device array of {const float* fast; float* slow; int64 n}; `chunks`: device (tensor, chunk) int32 pairs of
 *     dasac_ema_chunk_elems() elements, SORTED by tensor; `sq`: scratch of n_tensors + n_chunks doubles (one partial per
 *     chunk, added per tensor in a fixed order: the result is bit-identical from run to run).
 * dasac_scale_planes     Dropout2d with an explicit per-(n,c) keep mask (fcn.py:52,56).
 */
int dasac_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                  const float* conv_bias, float eps, int C, float* scale, float* shift,
                  float* invstd, dasac_stream_t stream);
int dasac_bn_param_grads(const float* dot, int dot_rows, const float* sum_dz, const float* mean,
                         const float* invstd, const float* scale, const float* conv_bias, int C,
                         float* dgamma, float* dbeta, float* dbias, dasac_stream_t stream);
int dasac_channel_sums(const float* x, int N, int C, int64_t HW, float* out, dasac_stream_t stream);
int dasac_maxpool_fwd(const float* x, int planes, int H, int W, int OH, int OW, int k, int s,
                      int pad, float* y, uint8_t* argmax, dasac_stream_t stream);
int dasac_maxpool_bwd(const float* dy, const float* y, const uint8_t* argmax, int planes, int H,
                      int W, int OH, int OW, int k, int s, int pad, int relu_mask, float* dx,
                      dasac_stream_t stream);
int dasac_ema_chunk_elems(void);
/* torch.optim.SGD(momentum, no nesterov, dampening 0) over all parameters of up to 8 groups in ONE launch
 * (base_trainer.py:63-66 over basenet.py:73-95):  d = g + wd*p; buf = first ? d : momentum*buf + d; p -= lr*buf.
 * tensors: device array of {float* p; const float* g; const float* g2 (NULL or a second gradient, summed as g2 + g before
 * anything else -- what two backward passes would have accumulated into .grad); float* buf; int64 n; int64 group}; chunks as for
 * dasac_ema_update ((tensor, chunk) pairs of dasac_ema_chunk_elems() elements); group_lr/group_wd: HOST arrays. */
int dasac_sgd_step(const void* tensors, int n_tensors, const int32_t* chunks, int n_chunks,
                   const float* group_lr, const float* group_wd, int n_groups, float momentum, int first,
                   dasac_stream_t stream);
int dasac_ema_update(const void* pairs, int n_tensors, const int32_t* chunks, int n_chunks,
                     float momentum, int update, double* sq, float* out, dasac_stream_t stream);
int dasac_scale_planes(const float* x, const float* plane_scale, int64_t planes, int64_t HW,
                       float* y, dasac_stream_t stream);
int dasac_add(const float* a, const float* b, float* out, int64_t n, dasac_stream_t stream);
/* out = y > 0 ? dy : 0  (ReLU backward, F.relu / nn.ReLU(inplace) of deeplabv2.py:84,88,97) */
int dasac_relu_mask(const float* dy, const float* y, float* out, int64_t n, dasac_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Train-mode BatchNorm (baseline / AdaBN mode, models/__init__.py:29; nn.SyncBatchNorm of
 * deeplabv2.py:15, fcn.py:8; running-stat re-estimation of train.py:281-289).
 * Both reductions are two-stage and atomic-free (one partial pair per plane and 4096-element chunk in `workspace` of
 * dasac_bn_stats_workspace(N, C, HW) bytes, then a fixed-order sum per channel): bit-identical from run to run.
 * forward:  dasac_bn_stats (sums[0:C] = sum z, sums[C:2C] = sum z^2, doubles) -> [caller all-reduces
 *           `sums` and the element count across ranks = SyncBN] -> dasac_bn_train_finalize (batch
 *           mean / biased var -> scale, shift, mean, invstd; running stats updated with momentum and
 *           the unbiased var; running_* may be NULL) -> dasac_bn_apply (y = relu?(z*scale+shift(+res))).
 * backward: dasac_bn_bwd_reduce (sums = sum dy, sum dy*xhat) -> [all-reduce] -> dasac_bn_bwd_apply
 *           (dz = gamma*invstd*(dy - sum_dy/n - xhat*sum_dy_xhat/n); dgamma, dbeta optional).
 * `count` = elements per channel over all ranks; `count_dev` (device double, may be NULL) overrides it so
 * that an all-reduced count never has to visit the host.
 */
size_t dasac_bn_stats_workspace(int N, int C, int64_t HW);
int dasac_bn_stats(const float* z, int N, int C, int64_t HW, double* sums, void* workspace, size_t ws_bytes,
                   dasac_stream_t stream);
int dasac_bn_train_finalize(const double* sums, double count, const double* count_dev, const float* gamma,
                            const float* beta, float* running_mean, float* running_var,
                            int64_t* num_batches_tracked /* += 1 when given */, float momentum, float eps, int C,
                            float* scale, float* shift, float* mean, float* invstd,
                            dasac_stream_t stream);
/* The same with the statistics left by dasac_conv_gemm_stats: dasac_bn_train_finalize_tiles sums the tiles itself (one launch
 * per BN layer on one rank); dasac_bn_tile_stats_reduce only produces `sums` (then all-reduce + dasac_bn_train_finalize = SyncBN). */
int dasac_bn_train_finalize_tiles(const float* tile_stats, int n_tiles, int mpad, double count, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var,
                                  int64_t* num_batches_tracked, float momentum, float eps, int C,
                                  float* scale, float* shift, float* mean, float* invstd, dasac_stream_t stream);
int dasac_bn_tile_stats_reduce(const float* tile_stats, int n_tiles, int C, int mpad, double* sums, dasac_stream_t stream);
/* One rank, nothing to all-reduce between statistics and use: finalize + normalise in ONE launch per BN layer
 * (dasac_bn_train_apply_tiles = dasac_bn_train_finalize_tiles + dasac_bn_apply; every block re-adds its channel's tile statistics
 * in the same fixed order), and the whole BN backward in two (dasac_bn_bwd_fused = the reduction's first stage + a dz kernel whose
 * blocks add the (plane, chunk) partials themselves and also write d gamma / d beta; workspace of dasac_bn_stats_workspace bytes). */
int dasac_bn_train_apply_tiles(const float* z, const float* tile_stats, int n_tiles, int mpad, double count,
                               const float* gamma, const float* beta, float* running_mean, float* running_var,
                               int64_t* num_batches_tracked, float momentum, float eps, const float* res, int relu,
                               int N, int C, int64_t HW, float* y, float* mean, float* invstd, dasac_stream_t stream);
int dasac_bn_bwd_fused(const float* dy, const float* z, const float* mean, const float* invstd, const float* gamma,
                       double count, int N, int C, int64_t HW, float* dz, float* dgamma, float* dbeta,
                       void* workspace, size_t ws_bytes, dasac_stream_t stream);
int dasac_bn_apply(const float* z, const float* scale, const float* shift, const float* res, int relu,
                   int N, int C, int64_t HW, float* y, dasac_stream_t stream);
int dasac_bn_bwd_reduce(const float* dy, const float* z, const float* mean, const float* invstd,
                        int N, int C, int64_t HW, double* sums, float* dgamma /* optional: this rank's sum dy*xhat */,
                        float* dbeta /* optional: this rank's sum dy */, void* workspace, size_t ws_bytes,
                        dasac_stream_t stream);
int dasac_bn_bwd_apply(const float* dy, const float* z, const float* mean, const float* invstd,
                       const float* gamma, const double* sums, double count, const double* count_dev, int N, int C,
                       int64_t HW, float* dz, float* dgamma, float* dbeta, dasac_stream_t stream);

/* Whole-network refresh of the derived operands after an optimiser / EMA step -- what ~100 dasac_bn_fold and ~200
 * dasac_conv_pack calls per step did, in two launches.  `jobs` are DEVICE arrays of the structs below, `chunks` device
 * (job, chunk) int32 pairs: one chunk = 256 channels of a fold job / dasac_pack_chunk_elems() floats of a pack job's output
 * (ceil(Kpad*Mpad / chunk) chunks per job; K and M padding are written as zeros).  Single-branch convolutions only
 * (taps = kh*kw of the one branch); mode / order as in dasac_conv_pack, Mpad = dasac_conv_mpad(M), Kpad = dasac_conv_kpad(K). */
typedef struct dasac_fold_job {
  const float *gamma, *beta, *mean, *var, *conv_bias; /* conv_bias may be NULL */
  float *scale, *shift, *invstd;
  float eps;
  int32_t C;
} dasac_fold_job;
typedef struct dasac_pack_job {
  const float* w;     /* [Cout][Cin][taps] */
  const float* scale; /* [Cout] folded into the operand, or NULL */
  float* out;         /* [Kpad/4][Mpad][4] */
  int32_t Cout, Cin, taps, Mpad, Kpad, mode, order, reserved;
} dasac_pack_job;
int dasac_pack_chunk_elems(void);
int dasac_bn_fold_multi(const dasac_fold_job* jobs, const int32_t* chunks, int n_chunks, dasac_stream_t stream);
int dasac_conv_pack_multi(const dasac_pack_job* jobs, const int32_t* chunks, int n_chunks, dasac_stream_t stream);

/* models/sac.py:337-338  `ignore_mask = (y == -1); y[ignore_mask] = 255` in one pass: mask[i] = labels[i] == pad_label,
 * labels[i] = ignore_label there (in place, like the reference).  labels i64 [n], mask u8 [n]. */
int dasac_label_pad_mask(int64_t* labels, uint8_t* mask, int64_t n, int pad_label, int ignore_label,
                         dasac_stream_t stream);
/* nn.Dropout2d(p) in train mode (models/fcn.py:52,56): keep_scale[plane] = Bernoulli(1-p) / (1-p) for the (n, c) planes,
 * consumed by dasac_scale_planes.  Counter-based Philox4x32-10: the draw is a pure function of (seed, offset, plane) --
 * the caller advances `offset` per call; it is NOT ATen's stream (parity tests inject the mask instead). */
int dasac_dropout_planes(uint64_t seed, uint64_t offset, float p, int64_t planes, float* keep_scale,
                         dasac_stream_t stream);

/* Validation counts (utils/metrics.py:9-53, train.py:339-469): counts[0:C] += tp, counts[C:2C] += fp,
 * counts[2C:3C] += fn of argmax_c logits vs gt (pixels with gt == ignore_index skipped); the caller
 * zeroes `counts` (int64 [3*C]) once per evaluation and all-reduces it across ranks. */
int dasac_iou_counts(const float* logits, const int64_t* gt, int B, int C, int64_t HW, int ignore_index,
                     int64_t* counts, dasac_stream_t stream);

/* K augmented views of one target crop: the pixel work of DataTarget.__getitem__'s tail
 * (datasets/dataloader_target.py:281-306) -- GuidedRandHFlip (datasets/tf_target.py:141-157), MaskRandScaleCrop
 * (:159-239; Pillow resize BILINEAR for the image, NEAREST for label / padding mask) and ToTensorMask / Normalize /
 * ApplyMask (:33-98) -- for all L views in one launch, byte-exact with Pillow's fixed-point resampling.
 * image u8 [3,H,W] planar, label u8 [H,W], mask u8 [H,W] or NULL (0 = valid); `tables`: L rows of
 * dasac_make_views_table_ints(H, W) int32 built on the host (views.py: view_tables): header {flip, ii, jj, win_h,
 * win_w, identity, 0, 0}, bounds_h[W][2], coeff_h[W][8], bounds_v[H][2], coeff_v[H][8], nearest_x[W], nearest_y[H].
 * mean3 / std3: HOST arrays of 3 floats.  Outputs: frames f32 [L,3,H,W] (normalised, 0 under the mask), gt i64
 * [L,H,W] (ignore_label under the mask), views_u8 (optional) u8 [L,3,H,W] = the resampled bytes. */
int dasac_make_views_table_ints(int H, int W);
int dasac_make_views(const uint8_t* image, const uint8_t* label, const uint8_t* mask, int H, int W, int L,
                     const int32_t* tables, const float* mean3, const float* std3, int ignore_label, float* frames,
                     int64_t* gt, uint8_t* views_u8, dasac_stream_t stream);

/* Photometric augmentations of the student's views (second half): `tf_augm` of DataTarget
 * (datasets/dataloader_target.py:116-123,292-296) = RandGaussianBlur (datasets/tf_target.py:331-349), MaskRandJitter
 * (:365-390, torchvision ColorJitter) and MaskRandGreyscale (:351-363) on the u8 views dasac_make_views emits, then
 * ToTensorMask / Normalize / ApplyMask (:33-98) -> frames1.  Byte-exact with Pillow's BoxBlur.c / Blend.c / Convert.c.
 * views_u8 u8 [L,3,H,W] (L <= 16); gt i64 [L,H,W] from dasac_make_views or NULL (pixels with gt == ignore_label are
 * the padding: frame value 0).  `params`: HOST array of L rows of DASAC_PHOTO_PARAMS doubles:
 *   [0] Gaussian radius (<= 0: no blur)        [1] colour jitter on/off
 *   [2..5] order of the four adjustments (0 brightness, 1 contrast, 2 saturation, 3 hue)
 *   [6..9] brightness, contrast, saturation, hue factors        [10] greyscale on/off        [11] reserved
 * Outputs: frames f32 [L,3,H,W]; out_u8 (optional) the augmented bytes.  `workspace`: device scratch of
 * dasac_view_photometric_workspace(H, W, L) bytes. */
#define DASAC_PHOTO_PARAMS 12
size_t dasac_view_photometric_workspace(int H, int W, int L);
int dasac_view_photometric(const uint8_t* views_u8, const int64_t* gt, int H, int W, int L, const double* params,
                           const float* mean3, const float* std3, int ignore_label, float* frames, uint8_t* out_u8,
                           void* workspace, size_t ws_bytes, dasac_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DASAC_HIP_H */
