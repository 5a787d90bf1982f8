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

#ifdef __cplusplus
}
#endif
#endif /* DASAC_HIP_H */
