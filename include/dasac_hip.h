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
 *   K = (sum of kh*kw) * C,  k = tap*C + c (tap-major).
 * dasac_conv_table   builds the gather table [Kpad][4] int32 for planes of plane_h x plane_w;
 *                    transposed=1 gives the data-gradient geometry (C = Cout, dh = pad - kh*dil).
 * dasac_conv_pack    lays W [Cout,Cin,kh,kw] out as [Kpad][Mpad] (transposed=1: rows (tap,co),
 *                    columns ci, optionally scaled per co by `scale` = folded BN gamma*invstd).
 *                    Call once per branch (tap0 = first tap of the branch, total_taps = all).
 * dasac_conv_gemm    out[n,m,oh*os,ow*os] = epi( sum_k packed[k][m] * x[n, c, oh*stride+dh, ow*stride+dw] )
 *                    epi: v*scale[m] + shift[m] (+res) (ReLU) (zeroed where mask <= 0).
 * dasac_conv_wgrad   partial weight gradients (split over pixels) into `workspace`;
 * dasac_conv_wgrad_finish  sums the splits, writes dW[co,ci,kh,kw] = scale[co]*G and, when `dot`
 *                    is given, dot[co] += sum_k W*G (the frozen-BN gamma gradient term).
 */
int dasac_conv_mpad(int M);
int dasac_conv_kpad(int K);
int dasac_conv_table(const int32_t* kh, const int32_t* kw, const int32_t* dil, const int32_t* pad,
                     int n_branches, int C, int plane_h, int plane_w, int transposed,
                     int32_t* table, dasac_stream_t stream);
int dasac_conv_pack(const float* w, const float* scale, int Cout, int Cin, int taps, int tap0,
                    int total_taps, int transposed, float* packed, dasac_stream_t stream);
int dasac_conv_gemm(const float* x, const float* packed, const int32_t* table, float* out,
                    int Nb, int Cx, int H, int W, int OH, int OW, int stride, int M, int K,
                    int OutH, int OutW, int ostride,
                    const float* scale, const float* shift, const float* res, const float* mask,
                    int relu, dasac_stream_t stream);
size_t dasac_conv_wgrad_workspace(int Nb, int OH, int OW, int M, int K);
int dasac_conv_wgrad(const float* dz, const float* x, const int32_t* table,
                     int Nb, int Cx, int H, int W, int OH, int OW, int stride, int M, int K,
                     void* workspace, size_t ws_bytes, dasac_stream_t stream);
int dasac_conv_wgrad_finish(const void* workspace, int Nb, int OH, int OW, int M, int K,
                            const float* w, const float* scale, float* dw, float* dot,
                            int Cin, int taps, int tap0, dasac_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DASAC_HIP_H */
