// Photometric augmentations of the student's K views on the device -- `tf_augm` of the reference's target loader
// (/root/reference/datasets/dataloader_target.py:116-123,292-296), applied to `images1` only:
//   RandGaussianBlur   datasets/tf_target.py:331-349   PIL ImageFilter.GaussianBlur(radius)
//   MaskRandJitter     datasets/tf_target.py:365-390   torchvision ColorJitter: brightness / contrast / saturation / hue in a
//                                                      random order, each one a PIL ImageEnhance blend or an HSV hue shift
//   MaskRandGreyscale  datasets/tf_target.py:351-363   F.to_grayscale(img, 3)
// followed by ToTensorMask / Normalize / ApplyMask (:33-98).  The reference does this per view with Pillow in the loader
// workers; byte-exact parity means Pillow's arithmetic (restated and pinned in oracle/photometric_ref.py):
//   blur   BoxBlur.c: 3 passes per axis of an extended box filter, weights in 1/2^24, every pass rounded to u8
//   blend  Blend.c:   a + alpha (b - a) in fp32, truncated (clipped when alpha is outside [0, 1])
//   grey   Convert.c: (19595 R + 38470 G + 7471 B + 2^15) >> 16;  RGB <-> HSV in its float / double mix
// Every stage is a pure byte stream over [L,3,H,W] u8 (<= 5 taps per output byte for radius <= 2): HBM/L2-bound, ~15 small
// launches per group of views, all views of a group in each launch (per-view parameters ride in the kernel arguments).
#include <cmath>

#include "common.hpp"

namespace dasac {

constexpr int kMaxViews = 16;

struct BlurArgs {
  int radius[kMaxViews];          // integer box radius, < 0: this view is not blurred (copy)
  unsigned ww[kMaxViews], fw[kMaxViews];
};

// grid (pixels / 256, 3, L); axis 0 = along x, 1 = along y
__global__ __launch_bounds__(256) void box_blur_pass(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W, int axis,
                                                     BlurArgs a) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= H * W) return;
  const int v = blockIdx.z;
  const size_t plane = ((size_t)v * 3 + blockIdx.y) * H * W;
  const uint8_t* src = in + plane;
  const int r = a.radius[v];
  if (r < 0) {
    out[plane + p] = src[p];
    return;
  }
  const int y = p / W, x = p - y * W;
  const int n = axis ? H : W, c = axis ? y : x, stride = axis ? W : 1, base = axis ? x : y * W;
  unsigned acc = 0;                                   // window [c-r, c+r] with edge replication (BoxBlur.c's running sum)
  for (int d = -r; d <= r; ++d) acc += src[base + min(max(c + d, 0), n - 1) * stride];
  const unsigned far = (unsigned)src[base + min(max(c - r - 1, 0), n - 1) * stride] + src[base + min(max(c + r + 1, 0), n - 1) * stride];
  out[plane + p] = (uint8_t)((acc * a.ww[v] + far * a.fw[v] + (1u << 23)) >> 24);     // UINT32 arithmetic, as in C
}

struct JitterArgs {
  int op[kMaxViews];              // -1 nothing, 0 brightness, 1 contrast, 2 saturation, 3 hue, 4 greyscale
  float factor[kMaxViews];        // blend alpha (ImagingBlend takes a C float)
  int hue[kMaxViews];             // byte added to H (mod 256)
};

__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

__device__ __forceinline__ int blend(int a, int b, float alpha) {
  if (alpha == 0.f) return a;
  if (alpha == 1.f) return b;
  const float t = __fadd_rn((float)a, __fmul_rn(alpha, (float)(b - a)));
  if (alpha >= 0.f && alpha <= 1.f) return (int)t;
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}

__device__ __forceinline__ void rgb2hsv(int r, int g, int b, int& uh, int& us, int& uv) {
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  uv = maxc;
  if (minc == maxc) {
    uh = us = 0;
    return;
  }
  const float cr = (float)(maxc - minc);
  const float s = __fdiv_rn(cr, (float)maxc);
  const float rc = __fdiv_rn((float)(maxc - r), cr), gc = __fdiv_rn((float)(maxc - g), cr), bc = __fdiv_rn((float)(maxc - b), cr);
  float h;
  if (r == maxc) h = __fsub_rn(bc, gc);
  else if (g == maxc) h = (float)__dsub_rn(__dadd_rn(2.0, (double)rc), (double)bc);
  else h = (float)__dsub_rn(__dadd_rn(4.0, (double)gc), (double)rc);
  h = (float)fmod(__dadd_rn(__ddiv_rn((double)h, 6.0), 1.0), 1.0);
  uh = min(max((int)__dmul_rn((double)h, 255.0), 0), 255);
  us = min(max((int)__dmul_rn((double)s, 255.0), 0), 255);
}

__device__ __forceinline__ void hsv2rgb(int h, int s, int v, int& r, int& g, int& b) {
  if (s == 0) {
    r = g = b = v;
    return;
  }
  const double h6 = __ddiv_rn(__dmul_rn((double)(float)h, 6.0), 255.0);
  const int i = (int)floor(h6);
  const double f = (double)(float)__dsub_rn(h6, (double)i);
  const double fs = (double)(float)__ddiv_rn((double)(float)s, 255.0);
  const double vf = (double)(float)v;
  const int p = min(max((int)rint(__dmul_rn(vf, __dsub_rn(1.0, fs))), 0), 255);
  const int q = min(max((int)rint(__dmul_rn(vf, __dsub_rn(1.0, __dmul_rn(fs, f)))), 0), 255);
  const int t = min(max((int)rint(__dmul_rn(vf, __dsub_rn(1.0, __dmul_rn(fs, __dsub_rn(1.0, f))))), 0), 255);
  switch (i % 6) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

// sum of the luma bytes of every view whose op is `contrast` (ImageStat.Stat(img.convert("L")).sum); grid (blocks, L)
__global__ __launch_bounds__(256) void luma_sums(const uint8_t* __restrict__ img, int HW, JitterArgs a, unsigned long long* __restrict__ sums) {
  const int v = blockIdx.y;
  if (a.op[v] != 1) return;
  const uint8_t* base = img + (size_t)v * 3 * HW;
  unsigned long long acc = 0;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) acc += (unsigned)luma(base[p], base[HW + p], base[2 * HW + p]);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(&sums[v], acc);
}

// one adjustment per view, in place; grid (pixels / 256, L)
__global__ __launch_bounds__(256) void jitter_step(uint8_t* __restrict__ img, int HW, JitterArgs a, const unsigned long long* __restrict__ sums) {
  const int v = blockIdx.y, op = a.op[v];
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (op < 0 || p >= HW) return;
  uint8_t* base = img + (size_t)v * 3 * HW;
  int r = base[p], g = base[HW + p], b = base[2 * HW + p];
  const float alpha = a.factor[v];
  if (op == 0) {
    r = blend(0, r, alpha); g = blend(0, g, alpha); b = blend(0, b, alpha);
  } else if (op == 1) {
    const int m = (int)__dadd_rn(__ddiv_rn((double)sums[v], (double)HW), 0.5);          // int(sum / count + 0.5)
    r = blend(m, r, alpha); g = blend(m, g, alpha); b = blend(m, b, alpha);
  } else if (op == 2) {
    const int l = luma(r, g, b);
    r = blend(l, r, alpha); g = blend(l, g, alpha); b = blend(l, b, alpha);
  } else if (op == 3) {
    int h, s, val;
    rgb2hsv(r, g, b, h, s, val);
    hsv2rgb((h + a.hue[v]) & 255, s, val, r, g, b);
  } else {
    r = g = b = luma(r, g, b);
  }
  base[p] = (uint8_t)r; base[HW + p] = (uint8_t)g; base[2 * HW + p] = (uint8_t)b;
}

// ToTensorMask (/255), Normalize, ApplyMask: same fp32 operation order as make_views
__global__ __launch_bounds__(256) void photo_finish(const uint8_t* __restrict__ img, const int64_t* __restrict__ gt, int HW, int L, float m0,
                                                    float m1, float m2, float s0, float s1, float s2, int ignore_label,
                                                    float* __restrict__ frames) {
  const int64_t total = (int64_t)L * HW;
  const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int v = (int)(idx / HW), p = (int)(idx - (int64_t)v * HW);
    const bool masked = gt && gt[idx] == (int64_t)ignore_label;
    for (int c = 0; c < 3; ++c) {
      const size_t o = ((size_t)v * 3 + c) * HW + p;
      const float val = __fdiv_rn(__fsub_rn(__fdiv_rn((float)img[o], 255.f), mean[c]), stdv[c]);
      frames[o] = masked ? 0.f : val;
    }
  }
}

// BoxBlur.c _gaussian_blur_radius + ImagingHorizontalBoxBlur's weights, in C float arithmetic like the original
static void box_weights(float radius, int& r, unsigned& ww, unsigned& fw) {
  const int passes = 3;
  float sigma2, L, l, a;
  sigma2 = radius * radius / passes;
  L = sqrt(12.0 * sigma2 + 1.0);
  l = floor((L - 1.0) / 2.0);
  a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2);
  a /= 6 * (sigma2 - (l + 1) * (l + 1));
  const float fr = l + a;
  r = (int)fr;
  ww = (unsigned)((1 << 24) / (fr * 2 + 1));
  fw = ((1 << 24) - (r * 2 + 1) * ww) / 2;
}

}  // namespace dasac

using namespace dasac;

extern "C" size_t dasac_view_photometric_workspace(int H, int W, int L) {
  return 2 * (size_t)L * 3 * H * W + 256 + (size_t)kMaxViews * sizeof(unsigned long long) * 4;
}

extern "C" int dasac_view_photometric(const uint8_t* views_u8, const int64_t* gt, int H, int W, int L, const double* params,
                                      const float* mean3, const float* std3, int ignore_label, float* frames, uint8_t* out_u8,
                                      void* workspace, size_t ws_bytes, dasac_stream_t stream) {
  DASAC_REQUIRE(views_u8 && params && mean3 && std3 && frames && workspace, "view_photometric: null pointer");
  DASAC_REQUIRE(H > 0 && W > 0 && L > 0 && L <= kMaxViews && (int64_t)H * W < (1ll << 30), "view_photometric: bad shape (at most 16 views)");
  DASAC_REQUIRE(ws_bytes >= dasac_view_photometric_workspace(H, W, L), "view_photometric: workspace too small");
  hipStream_t s = as_stream(stream);
  const int HW = H * W;
  const size_t bytes = (size_t)L * 3 * HW;
  uint8_t* bufA = reinterpret_cast<uint8_t*>(workspace);
  uint8_t* bufB = bufA + bytes;
  unsigned long long* sums = reinterpret_cast<unsigned long long*>(bufA + align_up(2 * bytes, 256));      // 4 steps x kMaxViews
  const dim3 pix_grid((HW + 255) / 256, 3, L), view_grid((HW + 255) / 256, L);

  // ---- blur: 3 passes along x, 3 along y (views without blur are copied through) -----------------
  BlurArgs ba;
  bool any_blur = false;
  for (int v = 0; v < L; ++v) {
    const double radius = params[v * DASAC_PHOTO_PARAMS + 0];
    ba.radius[v] = -1;
    ba.ww[v] = ba.fw[v] = 0;
    if (radius > 0.0) {
      box_weights((float)radius, ba.radius[v], ba.ww[v], ba.fw[v]);
      any_blur = true;
    }
  }
  const uint8_t* cur = views_u8;
  if (any_blur) {
    for (int pass = 0; pass < 6; ++pass) {
      uint8_t* dst = (pass & 1) ? bufB : bufA;
      hipLaunchKernelGGL(box_blur_pass, pix_grid, dim3(256), 0, s, cur, dst, H, W, pass / 3, ba);
      cur = dst;
    }
  } else {
    DASAC_HIP(hipMemcpyAsync(bufB, views_u8, bytes, hipMemcpyDeviceToDevice, s));
    cur = bufB;
  }
  uint8_t* img = const_cast<uint8_t*>(cur);            // bufB either way: the adjustments below work in place

  // ---- colour jitter: step k applies every view's k-th adjustment; then greyscale -----------------
  bool any_contrast = false;
  for (int v = 0; v < L; ++v)
    if (params[v * DASAC_PHOTO_PARAMS + 1] != 0.0) any_contrast = true;
  if (any_contrast) DASAC_HIP(hipMemsetAsync(sums, 0, 4 * kMaxViews * sizeof(unsigned long long), s));
  for (int step = 0; step < 5; ++step) {
    JitterArgs ja;
    bool any = false, contrast = false;
    for (int v = 0; v < L; ++v) {
      const double* pr = params + v * DASAC_PHOTO_PARAMS;
      ja.op[v] = -1;
      ja.factor[v] = 1.f;
      ja.hue[v] = 0;
      if (step < 4 && pr[1] != 0.0) {
        const int op = (int)pr[2 + step];
        DASAC_REQUIRE(op >= 0 && op < 4, "view_photometric: adjustment index outside 0..3");
        ja.op[v] = op;
        ja.factor[v] = (float)pr[6 + op];
        ja.hue[v] = (int)(pr[9] * 255) & 0xFF;           // np.uint8(hue_factor * 255): truncation, modulo 256
        any = true;
        contrast |= op == 1;
      } else if (step == 4 && pr[10] != 0.0) {
        ja.op[v] = 4;
        any = true;
      }
    }
    if (!any) continue;
    unsigned long long* step_sums = sums + (step & 3) * kMaxViews;
    if (contrast) hipLaunchKernelGGL(luma_sums, dim3(min((HW + 255) / 256, 1024), L), dim3(256), 0, s, img, HW, ja, step_sums);
    hipLaunchKernelGGL(jitter_step, view_grid, dim3(256), 0, s, img, HW, ja, step_sums);
  }

  hipLaunchKernelGGL(photo_finish, dim3(stream_grid((int64_t)L * HW, 256)), dim3(256), 0, s, img, gt, HW, L, mean3[0], mean3[1], mean3[2],
                     std3[0], std3[1], std3[2], ignore_label, frames);
  if (out_u8) DASAC_HIP(hipMemcpyAsync(out_u8, img, bytes, hipMemcpyDeviceToDevice, s));
  DASAC_CHECK_LAUNCH("view_photometric");
  return DASAC_OK;
}
