// K augmented views of one target crop on the device -- the pixel work of DataTarget.__getitem__'s tail
// (/root/reference/datasets/dataloader_target.py:281-306): GuidedRandHFlip (tf_target.py:141-157), MaskRandScaleCrop
// (:159-239: crop / pad window + `resize(BILINEAR)` for the image, `resize(NEAREST)` for label and padding mask) and the
// post transforms ToTensorMask / Normalize / ApplyMask (:33-98), for all L views in ONE launch.
//
// The reference does this with Pillow on the host, per view; byte-exact parity means Pillow's arithmetic:
// separable triangle filter whose normalised coefficients are quantised to 22-bit fixed point (Resample.c), a u8-rounded
// horizontal pass feeding the vertical pass, and ImagingScaleAffine's index tables for NEAREST.  The tables depend on
// the window size only and are built on the host in double precision (views.py); the kernel is pure integer work:
// HBM-bound, 3 + 1 + 1 bytes read per output pixel (window taps hit L2), 12 + 8 bytes written.
#include "common.hpp"

namespace dasac {

constexpr int kViewKs = 8;        // taps per output position the tables reserve (zoom windows up to ~3x the crop)
constexpr int kViewHdr = 8;

struct ViewSrc {
  const uint8_t* img;    // [3,H,W] planar
  const uint8_t* lab;    // [H,W]
  const uint8_t* msk;    // [H,W] or null (no padding anywhere)
  int H, W;
};

// one view's table: header {flip, ii, jj, win_h, win_w, identity, -, -}, bh[W][2], kh[W][KS], bv[H][2], kv[H][KS], tx[W], ty[H]
__device__ __forceinline__ int win_pixel(const uint8_t* __restrict__ plane, int H, int W, int flip, int ii, int jj, int r, int c,
                                         int fill) {
  const int br = ii + r, fc = jj + c;                  // window coordinates -> (flipped) crop coordinates
  if ((unsigned)br >= (unsigned)H || (unsigned)fc >= (unsigned)W) return fill;     // F.pad region of a zoom-out window
  return plane[br * W + (flip ? W - 1 - fc : fc)];
}

__global__ __launch_bounds__(256) void make_views(ViewSrc s, const int* __restrict__ tables, int table_stride, int L, float m0,
                                                  float m1, float m2, float s0, float s1, float s2, int ignore_label,
                                                  float* __restrict__ frames, int64_t* __restrict__ gt,
                                                  uint8_t* __restrict__ out_u8) {
  const int H = s.H, W = s.W, HW = H * W;
  const int64_t total = (int64_t)L * HW;
  constexpr int kPrec = 22;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int v = (int)(idx / HW), p = (int)(idx - (int64_t)v * HW);
    const int y = p / W, x = p - y * W;
    const int* tb = tables + (size_t)v * table_stride;
    const int flip = tb[0], ii = tb[1], jj = tb[2], ident = tb[5];
    const int* bh = tb + kViewHdr;
    const int* kh = bh + 2 * W;
    const int* bv = kh + kViewKs * W;
    const int* kv = bv + 2 * H;
    const int* tx = kv + kViewKs * H;
    const int* ty = tx + W;
    int px[3], lb, mk;
    if (ident) {
      for (int c = 0; c < 3; ++c) px[c] = win_pixel(s.img + (size_t)c * HW, H, W, flip, 0, 0, y, x, 0);
      lb = win_pixel(s.lab, H, W, flip, 0, 0, y, x, 1);
      mk = s.msk ? win_pixel(s.msk, H, W, flip, 0, 0, y, x, 1) : 0;
    } else {
      const int hmin = bh[2 * x], hn = bh[2 * x + 1], vmin = bv[2 * y], vn = bv[2 * y + 1];
      for (int c = 0; c < 3; ++c) {
        const uint8_t* plane = s.img + (size_t)c * HW;
        int acc_v = 1 << (kPrec - 1);
        for (int j = 0; j < vn; ++j) {
          int acc_h = 1 << (kPrec - 1);
          for (int i = 0; i < hn; ++i) acc_h += kh[kViewKs * x + i] * win_pixel(plane, H, W, flip, ii, jj, vmin + j, hmin + i, 0);
          const int t = min(max(acc_h >> kPrec, 0), 255);            // the horizontal pass is stored as u8 (Resample.c)
          acc_v += kv[kViewKs * y + j] * t;
        }
        px[c] = min(max(acc_v >> kPrec, 0), 255);
      }
      const int sy = ty[y], sx = tx[x];
      const bool in = sy >= 0 && sx >= 0;                            // ImagingScaleAffine leaves positions outside at 0
      lb = in ? win_pixel(s.lab, H, W, flip, ii, jj, sy, sx, 1) : 0;
      mk = in ? (s.msk ? win_pixel(s.msk, H, W, flip, ii, jj, sy, sx, 1) : (((unsigned)(ii + sy) >= (unsigned)H || (unsigned)(jj + sx) >= (unsigned)W) ? 1 : 0))
              : 0;
    }
    const bool masked = mk > 0;
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    for (int c = 0; c < 3; ++c) {
      const size_t o = ((size_t)v * 3 + c) * HW + p;
      // to_tensor (/255), Normalize (sub, div), ApplyMask (x * 0): plain fp32 ops in the reference's order
      const float val = __fdiv_rn(__fsub_rn(__fdiv_rn((float)px[c], 255.f), mean[c]), stdv[c]);
      frames[o] = masked ? 0.f : val;
      if (out_u8) out_u8[o] = (uint8_t)px[c];
    }
    gt[idx] = masked ? (int64_t)ignore_label : (int64_t)lb;
  }
}

}  // namespace dasac

using namespace dasac;

extern "C" int dasac_make_views_table_ints(int H, int W) { return kViewHdr + (2 + kViewKs) * (H + W) + H + W; }

extern "C" int dasac_make_views(const uint8_t* image, const uint8_t* label, const uint8_t* mask, int H, int W, int L,
                                const int32_t* tables, const float* mean3, const float* std3, int ignore_label, float* frames,
                                int64_t* gt, uint8_t* views_u8, dasac_stream_t stream) {
  DASAC_REQUIRE(image && label && tables && mean3 && std3 && frames && gt, "make_views: null pointer");
  DASAC_REQUIRE(H > 0 && W > 0 && L > 0 && (int64_t)H * W < (1ll << 30), "make_views: bad shape");
  ViewSrc s{image, label, mask, H, W};
  const int64_t total = (int64_t)L * H * W;
  hipLaunchKernelGGL(make_views, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), s, tables,
                     dasac_make_views_table_ints(H, W), L, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], ignore_label, frames,
                     gt, views_u8);
  DASAC_CHECK_LAUNCH("make_views");
  return DASAC_OK;
}
