// K14: per-class running-threshold + argmax pseudo-label extraction.
// Replaces models/sac.py:154-187 (reference, eager ATen: max, scatter_, view.max, gt_, sum, clone,
// masked stores = 4 reads + 2 writes of a [B,C,H,W] tensor) by two streaming passes:
//   pass A  reads probs once (C*4 B/px), writes max_conf (4 B/px) + argmax (1 B/px) and folds the
//           per-(image,class) peak into a [B,C] table (LDS atomics -> one global atomic per block/class)
//   pass B  reads conf+argmax+ignore (6 B/px), writes int64 labels (8 B/px) [+ int64 argmax if asked]
// HBM-bound; algorithmic bytes = C*4 + 4 + 8 (+8) per pixel.
#include "common.hpp"

namespace dasac {

constexpr int kPlBlock = 256;
constexpr int kPlMaxC = 64;

// All probabilities are >= +0, so the IEEE bit pattern is monotone as a signed int.
// FOUR consecutive pixels per thread: every class plane is one (4-byte aligned) dwordx4 load, and with the compile-time class
// count CT (19: the reference's NUM_CLASSES; 0 = runtime count, loads four classes at a time) all of a thread's plane loads
// are in flight together; max_conf leaves as one dwordx4 store, the four argmax bytes as one dword.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

template <int CT>
__global__ __launch_bounds__(kPlBlock) void pl_argmax_peaks(const float* __restrict__ probs, int Crt, int64_t HW,
                                                            int blocks_per_image, float* __restrict__ max_conf,
                                                            uint8_t* __restrict__ arg8, int* __restrict__ peaks) {
  __shared__ int s_peak[kPlMaxC];
  const int C = CT ? CT : Crt;
  const int b = blockIdx.x / blocks_per_image;
  const int chunk = blockIdx.x % blocks_per_image;
  if (threadIdx.x < kPlMaxC) s_peak[threadIdx.x] = 0;
  __syncthreads();
  const float* img = probs + (int64_t)b * C * HW;
  const int64_t stride = (int64_t)blocks_per_image * kPlBlock * 4;
  for (int64_t p = ((int64_t)chunk * kPlBlock + threadIdx.x) * 4; p < HW; p += stride) {
    const int nx = (int)(HW - p < 4 ? HW - p : 4);
    float m[4];
    int k[4] = {0, 0, 0, 0};
    if (nx == 4) {
      if (CT) {
        f32x4u v[CT ? CT : 1];
#pragma unroll
        for (int c = 0; c < CT; ++c) v[c] = *reinterpret_cast<const f32x4u*>(img + (int64_t)c * HW + p);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = v[0][e];
#pragma unroll
        for (int c = 1; c < CT; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (v[c][e] > m[e]) {  // strict: ties keep the lowest class (ATen max over dim)
              m[e] = v[c][e];
              k[e] = c;
            }
      } else {
        const f32x4u v0 = *reinterpret_cast<const f32x4u*>(img + p);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = v0[e];
#pragma unroll 4
        for (int c = 1; c < C; ++c) {
          const f32x4u v = *reinterpret_cast<const f32x4u*>(img + (int64_t)c * HW + p);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (v[e] > m[e]) {
              m[e] = v[e];
              k[e] = c;
            }
        }
      }
      *reinterpret_cast<f32x4u*>(max_conf + (int64_t)b * HW + p) = f32x4u{m[0], m[1], m[2], m[3]};
    } else {                                   // the last, partial quad of an image
#pragma unroll
      for (int e = 0; e < 3; ++e)
        if (e < nx) {
          m[e] = img[p + e];
          for (int c = 1; c < C; ++c) {
            const float v = img[(int64_t)c * HW + p + e];
            if (v > m[e]) {
              m[e] = v;
              k[e] = c;
            }
          }
          max_conf[(int64_t)b * HW + p + e] = m[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < nx) {
        arg8[(int64_t)b * HW + p + e] = (uint8_t)k[e];
        const int mi = __float_as_int(m[e]);
        if (mi > s_peak[k[e]]) atomicMax(&s_peak[k[e]], mi);
      }
  }
  __syncthreads();
  if (threadIdx.x < C) {
    const int v = s_peak[threadIdx.x];
    if (v > 0) atomicMax(&peaks[b * C + threadIdx.x], v);
  }
}

__global__ __launch_bounds__(kPlBlock) void pl_threshold(const float* __restrict__ max_conf,
                                                         const uint8_t* __restrict__ arg8,
                                                         const uint8_t* __restrict__ ignore,
                                                         const int* __restrict__ peaks, const float* __restrict__ disc,
                                                         float upper, float lower, int C, int64_t HW,
                                                         int blocks_per_image, int64_t* __restrict__ labels,
                                                         int64_t* __restrict__ max_idx) {
  __shared__ float s_thr[kPlMaxC];
  const int b = blockIdx.x / blocks_per_image;
  const int chunk = blockIdx.x % blocks_per_image;
  if (threadIdx.x < C) {
    // sac.py:168-174: top_peaks *= UPPER; top_peaks *= disc; clamp_(LOWER) -- same fp32 op order
    float t = __int_as_float(peaks[b * C + threadIdx.x]) * upper;
    if (disc) t = t * disc[threadIdx.x];
    s_thr[threadIdx.x] = fmaxf(t, lower);
  }
  __syncthreads();
  const int64_t stride = (int64_t)blocks_per_image * kPlBlock;
  const int64_t base = (int64_t)b * HW;
  for (int64_t p = (int64_t)chunk * kPlBlock + threadIdx.x; p < HW; p += stride) {
    const int k = arg8[base + p];
    const float m = max_conf[base + p];
    int64_t lab = (m > s_thr[k]) ? (int64_t)k : (int64_t)255;
    if (ignore && ignore[base + p]) lab = 255;
    labels[base + p] = lab;
    if (max_idx) max_idx[base + p] = k;
  }
}

}  // namespace dasac

using namespace dasac;

extern "C" size_t dasac_pseudo_labels_workspace(int B, int C, int64_t HW) {
  return align_up((size_t)B * (size_t)C * sizeof(int), 256) + align_up((size_t)B * (size_t)HW, 256);
}

extern "C" int dasac_pseudo_labels(const float* probs, const uint8_t* ignore, const float* disc, float upper,
                                   float lower, int B, int C, int64_t HW, int64_t* labels, float* max_conf,
                                   int64_t* max_idx, void* workspace, size_t ws_bytes, dasac_stream_t stream) {
  DASAC_REQUIRE(probs && labels && max_conf && workspace, "pseudo_labels: null pointer");
  DASAC_REQUIRE(B > 0 && C > 0 && C <= kPlMaxC && HW > 0, "pseudo_labels: bad shape B=%d C=%d HW=%lld", B, C, (long long)HW);
  DASAC_REQUIRE(lower > 0.f, "pseudo_labels: RUN_CONF_LOWER must be > 0 (got %g)", (double)lower);
  if (ws_bytes < dasac_pseudo_labels_workspace(B, C, HW)) return fail(DASAC_EWORKSPACE, "pseudo_labels: workspace too small");
  int* peaks = reinterpret_cast<int*>(workspace);
  uint8_t* arg8 = reinterpret_cast<uint8_t*>(workspace) + align_up((size_t)B * C * sizeof(int), 256);
  hipStream_t s = as_stream(stream);
  DASAC_HIP(hipMemsetAsync(peaks, 0, (size_t)B * C * sizeof(int), s));
  int per_image = stream_grid(HW, kPlBlock, (kNumCu * 16 + B - 1) / B);
  const int per_image4 = stream_grid((HW + 3) / 4, kPlBlock, (kNumCu * 16 + B - 1) / B);      // four pixels per thread
  if (C == 19)
    hipLaunchKernelGGL(pl_argmax_peaks<19>, dim3(per_image4 * B), dim3(kPlBlock), 0, s, probs, C, HW, per_image4, max_conf, arg8, peaks);
  else
    hipLaunchKernelGGL(pl_argmax_peaks<0>, dim3(per_image4 * B), dim3(kPlBlock), 0, s, probs, C, HW, per_image4, max_conf, arg8, peaks);
  DASAC_CHECK_LAUNCH("pl_argmax_peaks");
  hipLaunchKernelGGL(pl_threshold, dim3(per_image * B), dim3(kPlBlock), 0, s, max_conf, arg8, ignore, peaks, disc, upper,
                     lower, C, HW, per_image, labels, max_idx);
  DASAC_CHECK_LAUNCH("pl_threshold");
  return DASAC_OK;
}
