// The SAC head on MI355X: everything between the stride-8 logits and the losses.  All kernels are
// HBM-bound streaming passes over [B,C,H,W] fp32 planes (C = 19): lanes run along pixels
// (coalesced NCHW plane reads), the class dimension is a register loop.
//
//   upsample_softmax   models/deeplabv2.py:217, models/sac.py:275-282   bilinear(ac=True) [+softmax, prior sums, pad mask]
//   upsample_bwd_x/y   transpose of the above (separable gather, deterministic)
//   ce_loss            models/deeplabv2.py:223-224, models/sac.py:119-149  CE / focal CE(+conf) value and dlogits
//   warp_pool          models/sac.py:289-305 + 238-269 / 218-236          affine warp of T views -> fused pooling
//   warp_back          models/sac.py:309-311
//   warp_affine        models/sac.py:295-296 (diagnostic frame warp)
//   class_state        models/sac.py:104-117,120,151-152                  running class prior, discount, focal weights
#include "common.hpp"

#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include <atomic>

namespace dasac {

constexpr int kMaxC = 32;   // classes held in registers
constexpr int kHB = 256;
constexpr float kQ32 = 4294967296.f;     // fixed-point scales of the order-independent (integer) reductions
constexpr float kQ28 = 268435456.f;

// ---- bilinear taps, align_corners=True (ATen upsample_bilinear2d: src = scale*dst) ----------
struct Tap {
  int i0, i1;
  float w0, w1;
};
__host__ __device__ __forceinline__ Tap tap_ac(int dst, float scale, int n_in) {
  const float src = scale * (float)dst;
  int i0 = (int)src;
  if (i0 > n_in - 1) i0 = n_in - 1;
  Tap t;
  t.i0 = i0;
  t.i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  t.w1 = src - (float)i0;
  t.w0 = 1.f - t.w1;
  return t;
}

// scalar base + 32-bit per-lane byte offset: the form the global_load / global_store saddr encoding takes without any VALU
__device__ __forceinline__ float ld_off(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float* at_off(float* base, unsigned byte_off) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off);
}

// Four consecutive high-res pixels of one row per thread, loop over classes.  Optional outputs:
//   up    [B,C,H,W]  upsampled logits
//   probs [B,C,H,W]  softmax(up) * (ignore ? 0 : 1)
//   csum  [C] class sums of the UNMASKED softmax (running class prior, sac.py:108), accumulated in Q32 fixed point
//         (order-independent => run-to-run bit-identical); the launcher converts the slots to double afterwards
// CT = compile-time class count (19 for Cityscapes) so the per-pixel class vectors stay in registers.
// HBM-bound: 76 B written per pixel and output.  Four pixels per thread make every store a (4-byte aligned) dwordx4 and
// let the pixels share their low-resolution taps: at the 8x factor of the backbone the four x positions touch at most
// three low-res columns, so a class costs 6 L1/L2 loads per 4 pixels instead of 16.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

template <int CT, bool SOFTMAX, bool NARROW>
__global__ __launch_bounds__(kHB, (SOFTMAX && NARROW) ? 4 : 1) void upsample_softmax(const float* __restrict__ x, int Crt, int h, int w, int H, int W,
                                                        float sh, float sw, const uint8_t* __restrict__ ignore,
                                                        float* __restrict__ up, float* __restrict__ probs,
                                                        unsigned long long* __restrict__ csum, int items, FastDiv div_wq) {
  const int C = CT < kMaxC ? CT : Crt;          // CT == kMaxC is the generic (runtime-C) instantiation
  const int HW = H * W, hw = h * w, Wq = (W + 3) >> 2;
  const int b = blockIdx.y;                     // one image per block row: every plane base below is a scalar
  __shared__ unsigned long long s_sum[kMaxC];
  if (csum) {
    if (threadIdx.x < kMaxC) s_sum[threadIdx.x] = 0ull;
    __syncthreads();
  }
  float acc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) acc[c] = 0.f;
  // Round 5 (the kernel is issue-bound: ~4000 VALU instructions per 4 pixels in the probs path, tools/isa_count.py): the image
  // is the block row, so the class planes are scalar bases + ONE 32-bit per-lane byte offset shared by all classes (no 64-bit
  // per-lane address arithmetic per load / store); row / quad index by FastDiv; one select per tap instead of two.
  for (int it = blockIdx.x * kHB + threadIdx.x; it < items; it += gridDim.x * kHB) {
    const int oy = fdiv(it, div_wq), q = it - oy * Wq;
    const int ox = q * 4, nx = min(4, W - ox);
    const float* xb = x + (size_t)b * C * hw;
    const Tap ty = tap_ac(oy, sh, h);
    Tap tx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) tx[e] = tap_ac(min(ox + e, W - 1), sw, w);
    const int c_lo = tx[0].i0;
    // the usual case: the 4 pixels start in low-res column c_lo or c_lo + 1, so their taps are (t0,t1) or (t1,t2) -- t1, t2 read
    // at clamped columns, which is exactly where tap_ac puts i1 on the last column
    // (NARROW: the launcher has checked every quad of a row -- the up-factor-8 case; the kernel then carries no second path:
    // 131 instead of 207 VGPRs in the probs instantiation, three waves per SIMD instead of two)
    const bool narrow = NARROW || tx[3].i0 - c_lo <= 1;
    const int r0 = ty.i0 * w, r1 = ty.i1 * w;
    const int k1 = min(c_lo + 1, w - 1), k2 = min(c_lo + 2, w - 1);
    const unsigned a00 = (unsigned)(r0 + c_lo) * 4u, a01 = (unsigned)(r0 + k1) * 4u, a02 = (unsigned)(r0 + k2) * 4u;
    const unsigned a10 = (unsigned)(r1 + c_lo) * 4u, a11 = (unsigned)(r1 + k1) * 4u, a12 = (unsigned)(r1 + k2) * 4u;
    bool hi[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) hi[e] = tx[e].i0 != c_lo;
    float v[4][SOFTMAX ? CT : 1];                 // SOFTMAX: all classes of the 4 pixels stay in registers
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    const size_t obase = (size_t)b * C * HW;      // scalar
    const unsigned ooff = (unsigned)(oy * W + ox) * 4u;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      if (c < C) {
        const float* pl = xb + (size_t)c * hw;
        float val[4];
        if (narrow) {
          const float t0 = ld_off(pl, a00), t1 = ld_off(pl, a01), t2 = ld_off(pl, a02);
          const float b0 = ld_off(pl, a10), b1 = ld_off(pl, a11), b2 = ld_off(pl, a12);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float ta = hi[e] ? t1 : t0, tb = hi[e] ? t2 : t1;
            const float ba = hi[e] ? b1 : b0, bb = hi[e] ? b2 : b1;
            const float top = tx[e].w0 * ta + tx[e].w1 * tb;
            const float bot = tx[e].w0 * ba + tx[e].w1 * bb;
            val[e] = ty.w0 * top + ty.w1 * bot;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float top = tx[e].w0 * pl[r0 + tx[e].i0] + tx[e].w1 * pl[r0 + tx[e].i1];
            const float bot = tx[e].w0 * pl[r1 + tx[e].i0] + tx[e].w1 * pl[r1 + tx[e].i1];
            val[e] = ty.w0 * top + ty.w1 * bot;
          }
        }
        if (SOFTMAX) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e][c] = val[e];
            mx[e] = fmaxf(mx[e], val[e]);
          }
        } else {                                     // logits only: stream the class plane out, nothing kept
          float* o = at_off(up + obase + (size_t)c * HW, ooff);
          if (nx == 4) {
            *reinterpret_cast<f32x4u*>(o) = f32x4u{val[0], val[1], val[2], val[3]};
          } else {
#pragma unroll
            for (int e = 0; e < 3; ++e)
              if (e < nx) o[e] = val[e];
          }
        }
      }
    }
    if (SOFTMAX && up) {
#pragma unroll
      for (int c = 0; c < CT; ++c)
        if (c < C) {
          float* o = at_off(up + obase + (size_t)c * HW, ooff);
          if (nx == 4) {
            *reinterpret_cast<f32x4u*>(o) = f32x4u{v[0][c], v[1][c], v[2][c], v[3][c]};
          } else {
#pragma unroll
            for (int e = 0; e < 3; ++e)
              if (e < nx) o[e] = v[e][c];
          }
        }
    }
    if (SOFTMAX && (probs || csum)) {
      float inv[4];
      bool ign[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float den = 0.f;
#pragma unroll
        for (int c = 0; c < CT; ++c)
          if (c < C) {
            v[e][c] = expf(v[e][c] - mx[e]);
            den += v[e][c];
          }
        inv[e] = 1.f / den;
        ign[e] = ignore && e < nx && ignore[(size_t)b * HW + (size_t)(oy * W + ox + e)];
      }
#pragma unroll
      for (int c = 0; c < CT; ++c)
        if (c < C) {
          float pr[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pr[e] = v[e][c] * inv[e];
            if (e < nx) acc[c] += pr[e];
            if (ign[e]) pr[e] = 0.f;
          }
          if (probs) {
            float* o = at_off(probs + obase + (size_t)c * HW, ooff);
            if (nx == 4) {
              *reinterpret_cast<f32x4u*>(o) = f32x4u{pr[0], pr[1], pr[2], pr[3]};
            } else {
#pragma unroll
              for (int e = 0; e < 3; ++e)
                if (e < nx) o[e] = pr[e];
            }
          }
        }
    }
  }
  if (csum) {
    // per-thread partials -> wave sums (shuffles, fixed order) -> Q32 fixed point: integer adds commute, so the LDS atomic per
    // wave and class and the global one per block and class give the SAME bits whatever order the waves / blocks arrive in
    // (a wave's sum of probabilities is < 2^13, the total < B*HW < 2^31: no overflow; 2^-33 rounding per wave partial).
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < C) {
        const float ws = wave_sum(acc[c]);
        if ((threadIdx.x & 63) == 0) atomicAdd(&s_sum[c], __float2ull_rn(ws * kQ32));
      }
    __syncthreads();
    if (threadIdx.x < C) atomicAdd(&csum[threadIdx.x], s_sum[threadIdx.x]);
  }
}

// (Round 4, measured and rejected: a two-pixels-per-thread variant of the probs path -- 152 instead of 225 VGPRs, three instead of two
// waves per SIMD, 8-byte stores -- runs 238 us against this kernel's 214 at 8 x 19 x 769^2, 297 us when forced to four waves per SIMD
// with 92 bytes of scratch: sharing the low-resolution taps among four pixels and the dwordx4 stores outweigh the occupancy.)
// class sums: Q32 fixed point -> double, in place (the caller's buffer holds 8-byte slots either way)
__global__ void q32_to_double(unsigned long long* __restrict__ q, int C) {
  const int c = threadIdx.x;
  if (c >= C) return;
  const unsigned long long v = q[c];
  reinterpret_cast<double*>(q)[c] = (double)v * (1.0 / 4294967296.0);
}

// ---- inference (infer_val.py:160-163 + the writer's argmax / trainId->labelId LUT, :60-65): bilinear(ac=True) +
// softmax + argmax + LUT in one pass over the low-resolution logits; writes 1 byte (+ optional confidence) per
// high-resolution pixel instead of two [C,H,W] fp32 tensors.  Same per-pixel arithmetic as upsample_softmax.
template <int CT>
__global__ __launch_bounds__(kHB) void infer_labels(const float* __restrict__ x, int Crt, int h, int w, int H, int W, float sh,
                                                    float sw, const uint8_t* __restrict__ lut, uint8_t* __restrict__ labels,
                                                    float* __restrict__ conf, int blocks_per_image) {
  const int C = CT < kMaxC ? CT : Crt;
  const int b = blockIdx.x / blocks_per_image, chunk = blockIdx.x % blocks_per_image;
  const int HW = H * W, hw = h * w;
  const float* xb = x + (size_t)b * C * hw;
  for (int p = chunk * kHB + threadIdx.x; p < HW; p += blocks_per_image * kHB) {
    const int oy = p / W, ox = p - oy * W;
    const Tap ty = tap_ac(oy, sh, h), tx = tap_ac(ox, sw, w);
    const int o00 = ty.i0 * w + tx.i0, o01 = ty.i0 * w + tx.i1, o10 = ty.i1 * w + tx.i0, o11 = ty.i1 * w + tx.i1;
    float v[CT];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      if (c < C) {
        const float* pl = xb + (size_t)c * hw;
        const float top = tx.w0 * pl[o00] + tx.w1 * pl[o01];
        const float bot = tx.w0 * pl[o10] + tx.w1 * pl[o11];
        v[c] = ty.w0 * top + ty.w1 * bot;
        mx = fmaxf(mx, v[c]);
      }
    }
    float den = 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < C) {
        v[c] = expf(v[c] - mx);
        den += v[c];
      }
    const float inv = 1.f / den;
    int best = 0;
    float bp = -1.f;
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < C) {
        const float pr = v[c] * inv;
        if (pr > bp) {             // strict: the first maximum wins, as torch.argmax / numpy.argmax
          bp = pr;
          best = c;
        }
      }
    labels[(size_t)b * HW + p] = lut ? lut[best] : (uint8_t)best;
    if (conf) conf[(size_t)b * HW + p] = bp;
  }
}

// ---- transpose of the bilinear upsampling, separable and gather-based (deterministic) ---------
// pass X: tmp[plane][y][j] = sum_x wx(x->j) * g[plane][y][x]          (reads the big gradient once)
// pass Y: d[plane][i][j]   = gscale * sum_y wy(y->i) * tmp[plane][y][j]
__host__ __device__ __forceinline__ float weight_to(int dst, float scale, int n_in, int target) {
  const Tap t = tap_ac(dst, scale, n_in);
  float wgt = 0.f;
  if (t.i0 == target) wgt += t.w0;
  if (t.i1 == target) wgt += t.w1;   // i0 == i1 at the border: both weights land on the same source
  return wgt;
}
__host__ __device__ __forceinline__ void src_range(int target, float scale, int n_out, int& lo, int& hi) {
  // high-res positions whose taps can touch `target`:  target-1 < scale*dst < target+1
  if (scale <= 0.f) {
    lo = 0;
    hi = n_out - 1;
    return;
  }
  lo = (int)floorf((float)(target - 1) / scale) - 1;
  hi = (int)ceilf((float)(target + 1) / scale) + 1;
  if (lo < 0) lo = 0;
  if (hi > n_out - 1) hi = n_out - 1;
}

__global__ __launch_bounds__(kHB) void upsample_bwd_x(const float* __restrict__ g, int H, int W, int w, float sw,
                                                      float* __restrict__ tmp, int64_t total) {
  // total = planes*H*w; thread -> (plane, y, j), j fastest
  for (int64_t i = (int64_t)blockIdx.x * kHB + threadIdx.x; i < total; i += (int64_t)gridDim.x * kHB) {
    const int j = (int)(i % w);
    const int64_t row = i / w;   // plane*H + y
    int lo, hi;
    src_range(j, sw, W, lo, hi);
    const float* gr = g + row * W;
    float s = 0.f;
    for (int x = lo; x <= hi; ++x) s += weight_to(x, sw, w, j) * gr[x];
    tmp[i] = s;
  }
}

__global__ __launch_bounds__(kHB) void upsample_bwd_y(const float* __restrict__ tmp, int H, int h, int w, float sh,
                                                      const float* __restrict__ gscale, float* __restrict__ d,
                                                      int64_t total) {
  const float gs = gscale ? gscale[0] : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * kHB + threadIdx.x; i < total; i += (int64_t)gridDim.x * kHB) {
    const int j = (int)(i % w);
    const int64_t r = i / w;
    const int ii = (int)(r % h);
    const int64_t plane = r / h;
    int lo, hi;
    src_range(ii, sh, H, lo, hi);
    const float* tp = tmp + plane * H * w + j;
    float s = 0.f;
    for (int y = lo; y <= hi; ++y) s += weight_to(y, sh, h, ii) * tp[(size_t)y * w];
    d[i] = s * gs;
  }
}

// ---- cross entropy -------------------------------------------------------------------------
// Each thread loops over the B images of its pixels (needed by the cross-batch product
// of sac.py:148):   loss = sum_hw pw(hw) * sum_b ce_b(hw),
//   mode 0  plain mean (deeplabv2.py:224, sac.py:132):  pw = 1/(B*HW)
//   mode 1  focal_ce_conf (sac.py:148):                 pw = sum_i conf_i(hw) / (B*B*HW)
// ce_b = -cw[y]*log_softmax(x)[y], 0 where y == 255.  dlogits (optional) = pw * cw[y] * (softmax - onehot).
// per_class (optional, [C] Q28 fixed-point accumulators): sum over pixels of ce scattered by label (ignored pixels add nothing).
// Four consecutive pixels per thread (the [B,C,HW] layout is contiguous in the flattened pixel index): every class plane
// is one 4-byte-aligned dwordx4 load and all CT of an image are in flight together.
template <int CT>
__global__ __launch_bounds__(kHB) void ce_loss(const float* __restrict__ x, const int64_t* __restrict__ y,
                                               const float* __restrict__ cw, const float* __restrict__ conf, int B, int Crt,
                                               int HW, int mode, const float* __restrict__ gscale, float* __restrict__ dx,
                                               double* __restrict__ partial, unsigned long long* __restrict__ per_class) {
  const int C = CT < kMaxC ? CT : Crt;
  double lsum = 0.0;
  __shared__ unsigned long long s_pc[kMaxC];
  if (per_class) {
    if (threadIdx.x < kMaxC) s_pc[threadIdx.x] = 0ull;
    __syncthreads();
  }
  const float gs = gscale ? gscale[0] : 1.f;   // upstream gradient of the scalar loss (device side)
  const float norm = mode == 1 ? 1.f / ((float)B * (float)B * (float)HW) : 1.f / ((float)B * (float)HW);
  const int items = (HW + 3) >> 2;
  for (int it = blockIdx.x * kHB + threadIdx.x; it < items; it += gridDim.x * kHB) {
    const int p = it * 4, nx = min(4, HW - p);
    float cs[4] = {1.f, 1.f, 1.f, 1.f};
    if (mode == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) cs[e] = 0.f;
      for (int b = 0; b < B; ++b) {
        if (nx == 4) {
          const f32x4u cv = *reinterpret_cast<const f32x4u*>(conf + (size_t)b * HW + p);
#pragma unroll
          for (int e = 0; e < 4; ++e) cs[e] += cv[e];
        } else {
#pragma unroll
          for (int e = 0; e < 3; ++e)
            if (e < nx) cs[e] += conf[(size_t)b * HW + p + e];
        }
      }
    }
    float pw[4], cesum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) pw[e] = cs[e] * norm;
    for (int b = 0; b < B; ++b) {
      const size_t base = (size_t)b * C * HW + p;
      f32x4u v[CT];
      if (nx == 4) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
          if (c < C) v[c] = *reinterpret_cast<const f32x4u*>(x + base + (size_t)c * HW);
      } else {
#pragma unroll
        for (int c = 0; c < CT; ++c)
          if (c < C) {
            v[c] = f32x4u{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 3; ++e)
              if (e < nx) v[c][e] = x[base + (size_t)c * HW + e];
          }
      }
      int64_t lab[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) lab[e] = e < nx ? y[(size_t)b * HW + p + e] : (int64_t)-1;
      float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, den[4] = {0.f, 0.f, 0.f, 0.f}, xl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < CT; ++c)
        if (c < C) {
#pragma unroll
          for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], v[c][e]);
        }
#pragma unroll
      for (int c = 0; c < CT; ++c)
        if (c < C) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (c == lab[e]) xl[e] = v[c][e];
            const float ex = expf(v[c][e] - mx[e]);
            den[e] += ex;
            v[c][e] = ex;
          }
        }
      float k[4], gw[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool valid = lab[e] >= 0 && lab[e] < C;   // 255 (ignore) and anything out of range carry no loss
        const float wgt = valid ? (cw ? cw[lab[e]] : 1.f) : 0.f;
        const float ce = valid ? wgt * (logf(den[e]) - (xl[e] - mx[e])) : 0.f;
        cesum[e] += ce;
        // Q28 two's-complement fixed point: integer adds commute, so the scatter is bit-identical from run to run
        // (exact for ce >= 2^-4, 2^-29 rounding below; |ce| clamped to 4096: N*4096*2^28 < 2^63 for N < 2^23 pixels)
        if (per_class && valid && ce != 0.f && e < nx)
          atomicAdd(&s_pc[lab[e]], (unsigned long long)__float2ll_rn(fminf(fmaxf(ce, -4096.f), 4096.f) * kQ28));
        gw[e] = pw[e] * gs * wgt;
        k[e] = gw[e] / den[e];
      }
      if (dx) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
          if (c < C) {
            f32x4u d;
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = k[e] * v[c][e] - ((c == lab[e]) ? gw[e] : 0.f);
            float* o = dx + base + (size_t)c * HW;
            if (nx == 4) {
              *reinterpret_cast<f32x4u*>(o) = d;
            } else {
#pragma unroll
              for (int e = 0; e < 3; ++e)
                if (e < nx) o[e] = d[e];
            }
          }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < nx) lsum += (double)pw[e] * (double)cesum[e];
  }
  __shared__ double red[kHB / 64];
  lsum = wave_sum(lsum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  if (per_class && threadIdx.x < C && s_pc[threadIdx.x] != 0ull) atomicAdd(&per_class[threadIdx.x], s_pc[threadIdx.x]);
}

__global__ void ce_finish(const double* __restrict__ partial, int n, float* __restrict__ loss,
                          const unsigned long long* __restrict__ per_class, float* __restrict__ per_class_out, int C, double pc_norm) {
  double s = 0;
  for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) loss[0] = (float)s;
  if (per_class_out && threadIdx.x < C)
    per_class_out[threadIdx.x] = (float)((double)(long long)per_class[threadIdx.x] * (1.0 / 268435456.0) * pc_norm);
}

// ---- cross-entropy backward straight into the low-resolution gradient (K15 -> K9^T) ------------------------------
// The loss is a function of logits_up = U(logits) with U the bilinear (ac=True) upsampling; its gradient w.r.t. the
// stride-8 logits is U^T applied to the per-pixel d loss / d logits_up.  The two-kernel path wrote that full-resolution
// tensor (359.5 MB for 8 crops) and read it back; here one block per high-res row (b, y) keeps it in LDS:
//   phase 1  lanes along x (coalesced class-plane reads): softmax + the ce_loss weights -> d[c][x] in LDS,
//   phase 2  x-reduction with the horizontal tap weights -> tmp[(b,c)][y][j]   (same weights, same summation order as
//            upsample_bwd_x, so the result is the two-kernel path's bit for bit),
// and upsample_bwd_y finishes with the vertical taps.  LDS index x + x/8 spreads the stride-8 phase-2 reads over all banks.
__host__ __device__ __forceinline__ int ce_lds_index(int x) { return x + (x >> 3); }

// Round 4: a block handles a SEGMENT of `seg_cols` low-resolution columns of one high-res row (the x range their taps touch,
// a few hundred pixels) instead of the whole row: 19 x ~300 floats of LDS instead of 19 x 867 (66 KB, two blocks per CU whose
// two serial phases rarely overlapped) -- six 128-thread blocks per CU, phases of different blocks overlap.  The pixels between
// two segments' column ranges are evaluated by both (softmax is per pixel: same bits); every column is summed by ONE block over
// ascending x exactly as before, so the result stays bit-identical to dasac_ce_loss(dlogits) + dasac_upsample_bwd.
// cs[p] = sum_i conf_i[p] in image order (0 + c_0 + c_1 + ...: the order every row block used to repeat for itself, B times over)
__global__ __launch_bounds__(256) void conf_pixel_sums(const float* __restrict__ conf, int B, int HW, float* __restrict__ cs) {
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    float a = 0.f;
    for (int bb = 0; bb < B; ++bb) a += conf[(size_t)bb * HW + p];
    cs[p] = a;
  }
}

constexpr int kCB = 128;
template <int CT>
__global__ __launch_bounds__(kCB) void ce_bwd_rows(const float* __restrict__ xup, const int64_t* __restrict__ y,
                                                  const float* __restrict__ cw, const float* __restrict__ conf, int B, int Crt,
                                                  int H, int W, int w, float sw, int mode, const float* __restrict__ gscale,
                                                  float* __restrict__ tmp, int seg_cols, int n_seg, int pitch,
                                                  const float* __restrict__ cs_pix) {
  extern __shared__ float s_d[];                      // [C][pitch]
  const int C = CT < kMaxC ? CT : Crt;
  const int seg = blockIdx.x % n_seg, rowid = blockIdx.x / n_seg;
  const int b = rowid / H, oy = rowid - b * H;
  const int j0 = seg * seg_cols, j1 = min(w, j0 + seg_cols);
  int xs, xe, dummy;
  src_range(j0, sw, W, xs, dummy);
  src_range(j1 - 1, sw, W, dummy, xe);
  xs &= ~3;                                           // quads start on the same grid for every segment
  const int HW = H * W;
  const float gs = gscale ? gscale[0] : 1.f;
  const float norm = mode == 1 ? 1.f / ((float)B * (float)B * (float)HW) : 1.f / ((float)B * (float)HW);
  // phase 1: four consecutive pixels per thread -- every class plane is ONE (4-byte aligned) dwordx4 load and all CT of
  // them are in flight together (compile-time class count: no per-class predicate between the loads)
  for (int ox = xs + threadIdx.x * 4; ox <= xe; ox += kCB * 4) {
    const int nx = min(4, W - ox);
    const int p = oy * W + ox;
    const size_t base = (size_t)b * C * HW + p;
    f32x4u v[CT];
    if (nx == 4) {
#pragma unroll
      for (int c = 0; c < CT; ++c)
        if (c < C) v[c] = *reinterpret_cast<const f32x4u*>(xup + base + (size_t)c * HW);
    } else {
#pragma unroll
      for (int c = 0; c < CT; ++c)
        if (c < C) {
          v[c] = f32x4u{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 3; ++e)
            if (e < nx) v[c][e] = xup[base + (size_t)c * HW + e];
        }
    }
    float cs[4] = {1.f, 1.f, 1.f, 1.f};
    if (mode == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) cs[e] = 0.f;
      if (cs_pix) {                                       // sum_i conf_i per pixel, added up ONCE per launch (conf_pixel_sums)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < nx) cs[e] = cs_pix[p + e];
      } else if (nx == 4) {
#pragma unroll 8
        for (int bb = 0; bb < B; ++bb) {
          const f32x4u cv = *reinterpret_cast<const f32x4u*>(conf + (size_t)bb * HW + p);
#pragma unroll
          for (int e = 0; e < 4; ++e) cs[e] += cv[e];
        }
      } else {
        for (int bb = 0; bb < B; ++bb) {
          const float* cp = conf + (size_t)bb * HW + p;
#pragma unroll
          for (int e = 0; e < 3; ++e)
            if (e < nx) cs[e] += cp[e];
        }
      }
    }
    int64_t lab[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) lab[e] = e < nx ? y[(size_t)b * HW + p + e] : (int64_t)-1;
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, den[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < C) {
#pragma unroll
        for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], v[c][e]);
      }
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < C) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[c][e] = expf(v[c][e] - mx[e]);
          den[e] += v[c][e];
        }
      }
    float k[4], gw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool valid = lab[e] >= 0 && lab[e] < C;
      const float wgt = valid ? (cw ? cw[lab[e]] : 1.f) : 0.f;
      const float gpw = cs[e] * norm * gs;
      gw[e] = gpw * wgt;
      k[e] = gw[e] / den[e];
    }
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < C) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < nx) s_d[c * pitch + ce_lds_index(ox + e - xs)] = k[e] * v[c][e] - ((c == lab[e]) ? gw[e] : 0.f);
      }
  }
  __syncthreads();
  // phase 2: thread -> (column j, class group).  The tap weights of column j are computed once (registers) and reused
  // for every class of the group; the sum runs over ascending x like upsample_bwd_x does.
  constexpr int kMaxSpan = 24;                           // 2/scale + 3 taps: covers up-factors to 10
  const int nj = j1 - j0;
  const int groups = max(1, min(C, kCB / max(nj, 1)));
  for (int o = threadIdx.x; o < nj * groups; o += kCB) {
    const int j = j0 + o % nj, g0 = o / nj;
    int lo, hi;
    src_range(j, sw, W, lo, hi);
    const int n = hi - lo + 1;
    if (n <= kMaxSpan) {
      float wt[kMaxSpan];
      int li[kMaxSpan];
#pragma unroll
      for (int i = 0; i < kMaxSpan; ++i) {
        wt[i] = i < n ? weight_to(lo + i, sw, w, j) : 0.f;
        li[i] = ce_lds_index(min(lo + i, W - 1) - xs);
      }
      for (int c = g0; c < C; c += groups) {
        const float* row = s_d + c * pitch;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < kMaxSpan; ++i)
          if (i < n) acc += wt[i] * row[li[i]];
        tmp[((size_t)(b * C + c) * H + oy) * w + j] = acc;
      }
    } else {
      for (int c = g0; c < C; c += groups) {
        float acc = 0.f;
        for (int xx = lo; xx <= hi; ++xx) acc += weight_to(xx, sw, w, j) * s_d[c * pitch + ce_lds_index(xx - xs)];
        tmp[((size_t)(b * C + c) * H + oy) * w + j] = acc;
      }
    }
  }
}

// Round 5: the same two phases with ONE WAVE per block walking `rows` consecutive high-res rows of its segment.
// What the kernel above spends its time on (tools/isa_count.py, the PMC table): every block loads, waits, computes, synchronises
// and reduces exactly once -- the 19 plane loads of a row are never in flight while anything else happens, each row rebuilds the
// (row-independent) tap weights of its columns (as many instructions as the reduction itself), and 75 of its 128 threads have a
// quad in phase 1.  Here
//   * the x range of a segment is at most 256 pixels = one quad per lane, and the NEXT row's planes, labels and confidence sums
//     are requested right after this row's softmax has left the registers: they land during phase 2;
//   * a thread keeps its column's SPAN tap weights and LDS indices in registers for all rows;
//   * a one-wave block needs no cross-wave barrier, and seven of them fit a CU (22 KB of LDS each).
// Per pixel and per column the arithmetic and its order are those of ce_bwd_rows: identical bits.
__host__ __device__ __forceinline__ void ce_taps(int j, float sw, int W, int w, int& lo, int& hi) {
  src_range(j, sw, W, lo, hi);
  while (lo < hi && weight_to(lo, sw, w, j) == 0.f) ++lo;
  while (hi > lo && weight_to(hi, sw, w, j) == 0.f) --hi;
}
constexpr int kCW = 64;
typedef long long i64x2u __attribute__((ext_vector_type(2), aligned(8)));
template <int CT, int SPAN>
__global__ __launch_bounds__(kCW, 2) void ce_bwd_rows_wave(const float* __restrict__ xup, const int64_t* __restrict__ y,
                                                       const float* __restrict__ cw, int B, int H, int W, int w, float sw, int mode,
                                                       const float* __restrict__ gscale, float* __restrict__ tmp, int seg_cols,
                                                       int n_seg, int pitch, const float* __restrict__ cs_pix, int rows, int n_chunks) {
  extern __shared__ float s_d[];                      // [CT][pitch]
  const int seg = blockIdx.x % n_seg, rest = blockIdx.x / n_seg;
  const int rc = rest % n_chunks, b = rest / n_chunks;
  const int y0 = rc * rows, y1 = min(H, y0 + rows);
  const int j0 = seg * seg_cols, j1 = min(w, j0 + seg_cols);
  int xs, xe, dummy;
  src_range(j0, sw, W, xs, dummy);
  src_range(j1 - 1, sw, W, dummy, xe);
  xs &= ~3;
  const int HW = H * W;
  const float gs = gscale ? gscale[0] : 1.f;
  const float norm = mode == 1 ? 1.f / ((float)B * (float)B * (float)HW) : 1.f / ((float)B * (float)HW);
  // phase-1 role: the quad at xs + 4 * lane (the launcher guarantees xe - xs < 256 and W >= 4).  A quad that hangs over the end
  // of the row is LOADED four pixels back from the row's end and rotated into place (one lane of one segment per row): every
  // load is a full dwordx4, and a quad's four LDS slots are consecutive (xs and the quads are multiples of 4, the index skew
  // steps every 8), so the 76 LDS writes take 19 base registers + immediate offsets.
  const int ox = xs + (int)threadIdx.x * 4;
  const bool act = ox <= xe;
  const int rot = max(0, ox + 4 - W), ox_ld = ox - rot;
  // phase-2 role: column j, classes g0, g0 + groups, ...
  const int nj = j1 - j0;
  const int groups = max(1, min(CT, kCW / nj));
  const bool red = (int)threadIdx.x < nj * groups;
  const int j = j0 + (int)threadIdx.x % nj, g0 = (int)threadIdx.x / nj;
  // its taps: src_range is generous by two positions on either side; the zero weights at both ends are dropped, and the slots
  // behind the last tap (up to SPAN) carry weight 0 on that tap's LDS index.  acc starts at +0 and can never become -0, so
  // adding 0 * (a finite gradient) anywhere leaves every bit of the sum as it is: no predicate on the accumulation.
  int lo, hi;
  ce_taps(j, sw, W, w, lo, hi);
  const int n = hi - lo + 1;                              // <= SPAN (launcher)
  float wt[SPAN];
  unsigned la[SPAN];                                      // LDS byte offsets inside a class row
#pragma unroll
  for (int i = 0; i < SPAN; ++i) {
    wt[i] = i < n ? weight_to(lo + i, sw, w, j) : 0.f;
    la[i] = (unsigned)ce_lds_index(min(lo + i, hi) - xs) * 4u;
  }
  const unsigned row_bytes = (unsigned)pitch * 4u;
  float* const tmp_col = tmp + ((size_t)b * CT * H) * w + j;  // + (c * H + oy) * w
  __shared__ float s_cw[CT];                              // class weights: an LDS lookup by label, not a dependent global load
  if (threadIdx.x < CT) s_cw[threadIdx.x] = cw ? cw[threadIdx.x] : 1.f;
  __syncthreads();
  f32x4u v[CT];
  f32x4u cs;
  int64_t lab[4];
  auto load_row = [&](int oy) {
    if (!act) return;
    const int p = oy * W + ox_ld;
    const float* base = xup + (size_t)b * CT * HW + p;
    const int64_t* yp = y + (size_t)b * HW + p;
#pragma unroll
    for (int c = 0; c < CT; ++c) v[c] = *reinterpret_cast<const f32x4u*>(base + (size_t)c * HW);
    const i64x2u l01 = *reinterpret_cast<const i64x2u*>(yp), l23 = *reinterpret_cast<const i64x2u*>(yp + 2);
    lab[0] = l01[0]; lab[1] = l01[1]; lab[2] = l23[0]; lab[3] = l23[1];
    cs = mode == 1 ? *reinterpret_cast<const f32x4u*>(cs_pix + p) : f32x4u{1.f, 1.f, 1.f, 1.f};
  };
  load_row(y0);
  for (int oy = y0; oy < y1; ++oy) {
    if (act) {
      if (rot) {                                          // element e <- element e + rot (what lands beyond the row is never read)
        auto turn = [&](auto& q) {
          q[0] = rot == 1 ? q[1] : (rot == 2 ? q[2] : q[3]);
          q[1] = rot == 1 ? q[2] : q[3];
          q[2] = q[3];
        };
#pragma unroll
        for (int c = 0; c < CT; ++c) turn(v[c]);
        turn(cs);
      }
      int lb[4];                                          // label, or -1 (ignored / out of range: matches no class)
#pragma unroll
      for (int e = 0; e < 4; ++e) lb[e] = (lab[e] >= 0 && lab[e] < CT) ? (int)lab[e] : -1;
      if (rot) {
        lb[0] = rot == 1 ? lb[1] : (rot == 2 ? lb[2] : lb[3]);
        lb[1] = rot == 1 ? lb[2] : lb[3];
        lb[2] = lb[3];
      }
      float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, den[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], v[c][e]);
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[c][e] = expf(v[c][e] - mx[e]);
          den[e] += v[c][e];
        }
      float k[4], gw[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float wgt = lb[e] >= 0 ? s_cw[max(lb[e], 0)] : 0.f;
        const float gpw = cs[e] * norm * gs;
        gw[e] = gpw * wgt;
        k[e] = gw[e] / den[e];
      }
      float* wr = s_d + ce_lds_index(ox - xs);
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) wr[c * pitch + e] = k[e] * v[c][e] - ((c == lb[e]) ? gw[e] : 0.f);
    }
    __builtin_amdgcn_sched_barrier(0);                    // not earlier: the next row reuses this row's 76 registers
    if (oy + 1 < y1) load_row(oy + 1);                    // in flight while this row is reduced
    __syncthreads();
    if (red) {
#pragma unroll 1
      for (int c = g0; c < CT; c += groups) {
        const char* row = reinterpret_cast<const char*>(s_d) + (unsigned)c * row_bytes;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < SPAN; ++i) acc += wt[i] * *reinterpret_cast<const float*>(row + la[i]);
        tmp_col[(unsigned)(c * H + oy) * (unsigned)w] = acc;
      }
    }
    __syncthreads();
  }
}

// ---- affine warps: affine_grid + grid_sample(bilinear, zeros, align_corners=False) -------------
struct Sample {
  int o00, o01, o10, o11;     // flat offsets (clamped)
  float w00, w01, w10, w11;   // weights, 0 when the corner is out of bounds
};
__device__ __forceinline__ Sample make_sample_ex(const float* __restrict__ th, int oy, int ox, int H, int W, int& cx0o, int& cx1o,
                                                 int& cy0o, int& cy1o) {
  const float xb = (2.f * (float)ox + 1.f) / (float)W - 1.f;
  const float yb = (2.f * (float)oy + 1.f) / (float)H - 1.f;
  const float gx = th[0] * xb + th[1] * yb + th[2];
  const float gy = th[3] * xb + th[4] * yb + th[5];
  const float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f;
  const float iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float wx1 = ix - x0f, wx0 = (x0f + 1.f) - ix;
  const float wy1 = iy - y0f, wy0 = (y0f + 1.f) - iy;
  // keep the integer conversion safe for wild thetas
  const float xc = fminf(fmaxf(x0f, -2.f), (float)W + 1.f), yc = fminf(fmaxf(y0f, -2.f), (float)H + 1.f);
  const int x0 = (int)xc, y0 = (int)yc, x1 = x0 + 1, y1 = y0 + 1;
  const bool far = (xc != x0f) || (yc != y0f);
  const bool vx0 = !far && x0 >= 0 && x0 < W, vx1 = !far && x1 >= 0 && x1 < W;
  const bool vy0 = !far && y0 >= 0 && y0 < H, vy1 = !far && y1 >= 0 && y1 < H;
  const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
  const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
  Sample s;
  s.o00 = cy0 * W + cx0; s.o01 = cy0 * W + cx1; s.o10 = cy1 * W + cx0; s.o11 = cy1 * W + cx1;
  s.w00 = (vx0 && vy0) ? wx0 * wy0 : 0.f;
  s.w01 = (vx1 && vy0) ? wx1 * wy0 : 0.f;
  s.w10 = (vx0 && vy1) ? wx0 * wy1 : 0.f;
  s.w11 = (vx1 && vy1) ? wx1 * wy1 : 0.f;
  cx0o = cx0; cx1o = cx1; cy0o = cy0; cy1o = cy1;
  return s;
}
__device__ __forceinline__ Sample make_sample(const float* __restrict__ th, int oy, int ox, int H, int W) {
  int a, b, c, d;
  return make_sample_ex(th, oy, ox, H, W, a, b, c, d);
}
// The same sample with its four taps as TWO 8-byte gathers (round 5: the warp kernels are bound by the number of gather / store
// instructions their CU's address unit processes -- 130 us of gathers + 91 us of stores = warp_back's 220 -- not by bytes).  The
// two taps of a source row sit in columns {pb, pb + 1}, pb = min(cx0, W - 2): one dwordx2 at (row, pb) holds both, and a select
// per tap puts each into place (clamped columns included: cx0, cx1 are always pb or pb + 1).  W >= 2.
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
struct SampleP {
  unsigned a_top, a_bot;      // byte offsets of the two pairs
  bool hi0, hi1;              // tap column == pb + 1
  float w00, w01, w10, w11;
};
__device__ __forceinline__ SampleP make_sample_pair(const float* __restrict__ th, int oy, int ox, int H, int W) {
  int cx0, cx1, cy0, cy1;
  const Sample s = make_sample_ex(th, oy, ox, H, W, cx0, cx1, cy0, cy1);
  const int pb = min(cx0, W - 2);
  SampleP p;
  p.a_top = (unsigned)(cy0 * W + pb) * 4u;
  p.a_bot = (unsigned)(cy1 * W + pb) * 4u;
  p.hi0 = cx0 != pb;
  p.hi1 = cx1 != pb;
  p.w00 = s.w00; p.w01 = s.w01; p.w10 = s.w10; p.w11 = s.w11;
  return p;
}
__device__ __forceinline__ f32x2u ld_pair(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const f32x2u*>(reinterpret_cast<const char*>(base) + byte_off);
}
// pl[o00]*w00 + pl[o01]*w01 + pl[o10]*w10 + pl[o11]*w11 from the two pairs, same order as take()
__device__ __forceinline__ float blend_pairs(const f32x2u t, const f32x2u b, const SampleP& p) {
  const float v00 = p.hi0 ? t[1] : t[0], v01 = p.hi1 ? t[1] : t[0];
  const float v10 = p.hi0 ? b[1] : b[0], v11 = p.hi1 ? b[1] : b[0];
  return v00 * p.w00 + v01 * p.w01 + v10 * p.w10 + v11 * p.w11;
}
__device__ __forceinline__ float take(const float* __restrict__ pl, const Sample& s) {
  return pl[s.o00] * s.w00 + pl[s.o01] * s.w01 + pl[s.o10] * s.w10 + pl[s.o11] * s.w11;
}

// (Round 4, measured and rejected: walking the output in 64 x 4 pixel tiles -- a wave per row segment, four adjacent rows per block,
// so that vertically adjacent outputs share their source row inside a block -- to cut the over-fetch the counters show for these
// kernels (`warp_back` moves 2.4x, `warp_pool` 1.4x its algorithmic bytes: profiles/r4_head_kernel_pmc.md).  `warp_pool` went
// 294 -> 755 us, `warp_back` 220 -> 245 us: the linear pixel order keeps every wave's 19 x T class-plane streams sequential in
// DRAM, the tile order breaks each of them into 256-byte pieces three rows apart.  The linear order stays.)
// generic warp of a [B,C,H,W] tensor (frames_aligned diagnostic)
__global__ __launch_bounds__(kHB) void warp_affine(const float* __restrict__ x, const float* __restrict__ theta, int C,
                                                   int H, int W, float* __restrict__ out, int blocks_per_image) {
  const int b = blockIdx.x / blocks_per_image, chunk = blockIdx.x % blocks_per_image;
  const int HW = H * W;
  const float* th = theta + b * 6;
  for (int p = chunk * kHB + threadIdx.x; p < HW; p += blocks_per_image * kHB) {
    const Sample s = make_sample(th, p / W, p % W, H, W);
    for (int c = 0; c < C; ++c) {
      const size_t pb = ((size_t)b * C + c) * HW;
      out[pb + p] = take(x + pb, s);
    }
  }
}

// views -> reference frame -> pooled.  One thread per (group, pixel):
//   a_t   = sample(probs[n*T+t], theta[n*T+t])                     (teacher_aligned, optional output)
//   cov_t = coverage(theta_inv[n*T+t]) at this pixel               (sac.py:299-301, quirk 4)
//   avg (mode 0):  S = sum_t a_t*cov_t ; Z = sum_c S ; mask = Z > tol ; S /= max(Z, 1e-3)
//   minentropy (mode 1): S = a_t*cov_t of the view with the lowest entropy (first minimum)
__global__ __launch_bounds__(kHB) void warp_pool(const float* __restrict__ probs, const float* __restrict__ theta,
                                                 const float* __restrict__ theta_inv, int T, int C, int H, int W,
                                                 int mode, float tol, float* __restrict__ aligned,
                                                 float* __restrict__ pooled, float* __restrict__ mask,
                                                 int blocks_per_group) {
  const int n = blockIdx.x / blocks_per_group, chunk = blockIdx.x % blocks_per_group;
  const int HW = H * W;
  for (int p = chunk * kHB + threadIdx.x; p < HW; p += blocks_per_group * kHB) {
    const int oy = p / W, ox = p - oy * W;
    float S[kMaxC];
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) S[c] = 0.f;
    float best_ent = INFINITY, zsum = 0.f;
    for (int t = 0; t < T; ++t) {
      const int b = n * T + t;
      float v[kMaxC];
      float vs = 0.f;
      if (theta) {
        const Sample s = make_sample(theta + b * 6, oy, ox, H, W);
        const Sample si = make_sample(theta_inv + b * 6, oy, ox, H, W);
        const float cov = si.w00 + si.w01 + si.w10 + si.w11;
#pragma unroll
        for (int c = 0; c < kMaxC; ++c)
          if (c < C) {
            const size_t pb = ((size_t)b * C + c) * HW;
            const float a = take(probs + pb, s);
            if (aligned) aligned[pb + p] = a;
            v[c] = a * cov;
            vs += v[c];
          }
      } else {   // views already aligned and coverage-weighted by the caller (SAC._avg_pool / _minentropy_pool on their own)
#pragma unroll
        for (int c = 0; c < kMaxC; ++c)
          if (c < C) {
            v[c] = probs[((size_t)b * C + c) * HW + p];
            vs += v[c];
          }
      }
      if (mode == 0) {
#pragma unroll
        for (int c = 0; c < kMaxC; ++c)
          if (c < C) S[c] += v[c];
      } else {
        // sac.py:189-196 entropy with eps = 1e-5; empty pixels get 1/eps
        float ent = 0.f;
#pragma unroll
        for (int c = 0; c < kMaxC; ++c)
          if (c < C) ent -= v[c] * logf((v[c] + 1e-5f) / (1.f + 1e-5f));
        if (vs < 0.1f) ent = 1.f / 1e-5f;
        zsum += vs;
        if (ent < best_ent) {
          best_ent = ent;
#pragma unroll
          for (int c = 0; c < kMaxC; ++c)
            if (c < C) S[c] = v[c];
        }
      }
    }
    float m;
    if (mode == 0) {
      float Z = 0.f;
#pragma unroll
      for (int c = 0; c < kMaxC; ++c)
        if (c < C) Z += S[c];
      m = Z > tol ? 1.f : 0.f;
      const float den = fmaxf(Z, 1e-3f);
#pragma unroll
      for (int c = 0; c < kMaxC; ++c)
        if (c < C) S[c] = S[c] / den;
    } else {
      m = zsum > tol ? 1.f : 0.f;
    }
    mask[(size_t)n * HW + p] = m;
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c < C) pooled[((size_t)n * C + c) * HW + p] = S[c];
  }
}

// The default configuration (avg pooling of TT <= 4 warped views, CT classes known at compile time) with the memory-level
// parallelism the generic kernel above lacks: that one walks the views one after the other and, the class count being a
// runtime value, gets four dependent-looking gathers in flight at a time -- 1.5 TB/s of algorithmic bytes, latency-bound.
// Here the T sample descriptors of a pixel are built first, then every class issues its 4*T taps as ONE batch (16 independent
// loads for T = 4; the unrolled class loop lets the compiler run several classes ahead).  Same arithmetic in the same order
// (per class S = ((a0*cov0 + a1*cov1) + a2*cov2) + a3*cov3 accumulated from 0 in view order, then Z, mask, S / max(Z, 1e-3)):
// bit-identical to the generic kernel, which stays for min-entropy pooling, pre-aligned views, T > 4 and other class counts.
// Round 5: every plane base is a scalar (group pointer + compile-time (view, class) offset) and the pixel a 32-bit lane offset --
// no 64-bit per-lane address arithmetic in front of the 16 gathers and 4 stores of a class: 2054 instead of 2672 VALU instructions
// per pixel, 325 -> 235 us at 2 x 4 x 19 x 769^2 (same box, tools/head_exp.py).  DASAC_WP_ROWS vertically adjacent pixels per
// thread (the row below shares a source row: fewer L2 fills, as in warp_back): 2 rows need 256 VGPRs and run 566 us -- one row.
constexpr int kWpRows = 1;
__device__ __forceinline__ float take_off(const float* __restrict__ pl, const Sample& s) {
  return ld_off(pl, (unsigned)s.o00 * 4u) * s.w00 + ld_off(pl, (unsigned)s.o01 * 4u) * s.w01 + ld_off(pl, (unsigned)s.o10 * 4u) * s.w10 +
         ld_off(pl, (unsigned)s.o11 * 4u) * s.w11;
}
template <int CT, int TT, int R>
__device__ __forceinline__ void warp_pool_rows(const float* __restrict__ probs_n, const float* __restrict__ theta_n,
                                               const float* __restrict__ theta_inv_n, int H, int W, int HW, float tol,
                                               float* __restrict__ aligned_n, float* __restrict__ pooled_n,
                                               float* __restrict__ mask_n, int oy, int ox) {
  Sample s[R][TT];                              // (8-byte pair gathers as in warp_back: 280 against 225 us here -- four dword gathers stay)
  float cov[R][TT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      s[r][t] = make_sample(theta_n + t * 6, oy + r, ox, H, W);
      const Sample si = make_sample(theta_inv_n + t * 6, oy + r, ox, H, W);
      cov[r][t] = si.w00 + si.w01 + si.w10 + si.w11;
    }
  const unsigned o0 = (unsigned)(oy * W + ox) * 4u, pitch = (unsigned)W * 4u;
  float S[R][CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    float a[R][TT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int t = 0; t < TT; ++t) a[r][t] = take_off(probs_n + (size_t)(t * CT + c) * HW, s[r][t]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        if (aligned_n) *at_off(aligned_n + (size_t)(t * CT + c) * HW, o0 + (unsigned)r * pitch) = a[r][t];
        acc += a[r][t] * cov[r][t];
      }
      S[r][c] = acc;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float Z = 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c) Z += S[r][c];
    const float den = fmaxf(Z, 1e-3f);
    *at_off(mask_n, o0 + (unsigned)r * pitch) = Z > tol ? 1.f : 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c) *at_off(pooled_n + (size_t)c * HW, o0 + (unsigned)r * pitch) = S[r][c] / den;
  }
}
template <int CT, int TT>
__global__ __launch_bounds__(kHB) void warp_pool_avg(const float* __restrict__ probs, const float* __restrict__ theta,
                                                     const float* __restrict__ theta_inv, int H, int W, float tol,
                                                     float* __restrict__ aligned, float* __restrict__ pooled,
                                                     float* __restrict__ mask, int blocks_per_group, int items, FastDiv div_w) {
  const int n = blockIdx.x / blocks_per_group, chunk = blockIdx.x % blocks_per_group;
  const int HW = H * W;
  const float* probs_n = probs + (size_t)n * TT * CT * HW;
  float* aligned_n = aligned ? aligned + (size_t)n * TT * CT * HW : nullptr;
  float* pooled_n = pooled + (size_t)n * CT * HW;
  float* mask_n = mask + (size_t)n * HW;
  const float* th = theta + n * TT * 6;
  const float* thi = theta_inv + n * TT * 6;
  for (int it = chunk * kHB + threadIdx.x; it < items; it += blocks_per_group * kHB) {
    const int rr = fdiv(it, div_w), ox = it - rr * W;
    const int oy = kWpRows * rr;
    if (oy + kWpRows <= H) {
      warp_pool_rows<CT, TT, kWpRows>(probs_n, th, thi, H, W, HW, tol, aligned_n, pooled_n, mask_n, oy, ox);
    } else {
      for (int y = oy; y < H; ++y) warp_pool_rows<CT, TT, 1>(probs_n, th, thi, H, W, HW, tol, aligned_n, pooled_n, mask_n, y, ox);
    }
  }
}

// refined[b] = sample(pooled[g(b)], theta_inv[b]) * sample(mask[g(b)], theta_inv[b]),  g(b) = group_of[b]
// (round 4, measured: a compile-time class count with all 4 x 19 taps of a pixel in flight makes this kernel SLOWER -- 327 vs
// 213 us at 8 x 19 x 769^2: the registers of 76 gathers in flight cost more occupancy than the batching wins; the class loop stays)
// Round 5: a thread owns kWbRows vertically adjacent pixels.  The counters showed 2.4x the algorithmic bytes entering L2: the
// row below re-reads the lower source row of the row above, and in the linear pixel order that neighbour is another block on
// another XCD (its own L2).  With the rows in one thread a shared source row is fetched once per group of rows, and sixteen
// taps are in flight instead of four.  The block still walks the plane of row groups linearly, so the store streams stay
// sequential in DRAM; per pixel the same make_sample / take arithmetic: identical bits.  Class planes are scalar
// bases + 32-bit per-lane byte offsets.  Same-box A/B of two rows against the one-row kernel (tools/head_exp.py, three pairs):
// 204 / 215 / 205 us against 214 / 232 / 215 at 8 x 19 x 769^2.  L2 fills per launch (FETCH_SIZE x 2): 687 MB with one row, 568
// with two, 503 with four (kWbRows) against 360 MB for one pass per view; kernel time 217 / 208 / 202 us.  Knock-outs: without
// its stores 130 us, without its gathers 91 us, without both 32 us -- 130 + 91 = the kernel: loads and stores do not overlap,
// what adds up is the number of gather / store instructions the CU's address unit works through.  Hence (late in the round) the
// four taps as two 8-byte pair gathers (SampleP) and the next class's gathers issued in front of this class's stores: 219 -> 192
// us, same box.  (The same pairs in warp_pool_avg: 225 -> 281 us -- kept there as four dword gathers.)
// (Measured and rejected on the way: four horizontally adjacent pixels per thread with dwordx4 stores 364 us -- gathers whose
// lanes sit 16 bytes apart; an XCD-aware chunk order (XCD x walks the x-th eighth of every pass, so that vertical neighbours
// share an L2) 224 against 208 us here and 445 against 311 us for warp_pool: what these kernels need is the linear pixel
// order's sequential DRAM streams, not fewer L2 fills.)
constexpr int kWbRows = 4;       // vertically adjacent output pixels per thread
// R rows of one output column: R sample descriptors, then the classes NC at a time with all NC * R * 4 taps issued before the
// first is used (R, NC compile-time: as run-time predicates the compiler sinks the lower rows' taps under their test and waits
// for four loads at a time).
template <int R>
__device__ __forceinline__ void warp_back_rows(const float* __restrict__ pooled_n, const float* __restrict__ mask_n,
                                               const float* __restrict__ th, float* __restrict__ refined_b, int C, int H, int W,
                                               int HW, int oy, int ox) {
  SampleP s[R];
  float mv[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    s[r] = make_sample_pair(th, oy + r, ox, H, W);
    mv[r] = blend_pairs(ld_pair(mask_n, s[r].a_top), ld_pair(mask_n, s[r].a_bot), s[r]);
  }
  const unsigned o0 = (unsigned)(oy * W + ox) * 4u, pitch = (unsigned)W * 4u;
  // One class = R * 2 pair gathers and R stores.  vmcnt retires a wave's loads and stores in issue order, so a class whose
  // gathers are issued AFTER the previous class's stores waits for those stores to drain: two register sets, the next class's
  // gathers go out before this class's stores (208 against 221 us, same box).
  auto load = [&](f32x2u (&t)[R][2], int c) {
    const float* pl = pooled_n + (size_t)c * HW;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      t[r][0] = ld_pair(pl, s[r].a_top);
      t[r][1] = ld_pair(pl, s[r].a_bot);
    }
  };
  auto store = [&](const f32x2u (&t)[R][2], int c) {
    float* out = refined_b + (size_t)c * HW;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float a = blend_pairs(t[r][0], t[r][1], s[r]);
      *at_off(out, o0 + (unsigned)r * pitch) = a * mv[r];
    }
  };
  f32x2u ta[R][2], tb[R][2];
  load(ta, 0);
  int c = 0;
  for (; c + 2 <= C; c += 2) {
    load(tb, c + 1);
    __builtin_amdgcn_sched_barrier(0);          // the gathers of class c + 1 stay in front of the stores of class c
    store(ta, c);
    if (c + 2 < C) load(ta, c + 2);
    __builtin_amdgcn_sched_barrier(0);
    store(tb, c + 1);
  }
  if (c < C) store(ta, c);
}
__global__ __launch_bounds__(kHB) void warp_back(const float* __restrict__ pooled, const float* __restrict__ mask,
                                                 const float* __restrict__ theta_inv, int group_div, int C, int H, int W,
                                                 float* __restrict__ refined, int blocks_per_image, int items, FastDiv div_w) {
  const int b = blockIdx.x / blocks_per_image, chunk = blockIdx.x % blocks_per_image;
  const int n = b / group_div;
  const int HW = H * W;
  const float* pooled_n = pooled + (size_t)n * C * HW;
  const float* mask_n = mask + (size_t)n * HW;
  float* refined_b = refined + (size_t)b * C * HW;
  for (int it = chunk * kHB + threadIdx.x; it < items; it += blocks_per_image * kHB) {
    const int r = fdiv(it, div_w), ox = it - r * W;
    const int oy = kWbRows * r;
    if (oy + kWbRows <= H) {
      warp_back_rows<kWbRows>(pooled_n, mask_n, theta_inv + b * 6, refined_b, C, H, W, HW, oy, ox);
    } else {                                    // the last rows of a height that is no multiple of kWbRows
      for (int y = oy; y < H; ++y) warp_back_rows<1>(pooled_n, mask_n, theta_inv + b * 6, refined_b, C, H, W, HW, y, ox);
    }
  }
}

// W == 1: no pair of columns exists for the 8-byte gathers above -- four scalar taps per sample (make_sample / take, the same
// products in the same order), one thread per output pixel.  Any size grid_sample accepts is accepted here too.
__global__ __launch_bounds__(kHB) void warp_back_scalar(const float* __restrict__ pooled, const float* __restrict__ mask,
                                                        const float* __restrict__ theta_inv, int group_div, int C, int H, int W,
                                                        float* __restrict__ refined, int blocks_per_image) {
  const int b = blockIdx.x / blocks_per_image, chunk = blockIdx.x % blocks_per_image;
  const int n = b / group_div;
  const int HW = H * W;
  const float* pooled_n = pooled + (size_t)n * C * HW;
  float* refined_b = refined + (size_t)b * C * HW;
  for (int p = chunk * kHB + threadIdx.x; p < HW; p += blocks_per_image * kHB) {
    const Sample s = make_sample(theta_inv + b * 6, p / W, p % W, H, W);
    const float mv = take(mask + (size_t)n * HW, s);
    for (int c = 0; c < C; ++c) refined_b[(size_t)c * HW + p] = take(pooled_n + (size_t)c * HW, s) * mv;
  }
}

// ---- class prior state (19 floats; one wave) ------------------------------------------------------
// chi update (sac.py:104-117) from the class sums of this batch, then the two derived vectors:
//   disc = 1 - exp(-chi/beta) (sac.py:152), focal = (1 - max(chi,0))^p (sac.py:120,135)
__global__ void class_state(float* __restrict__ chi, const double* __restrict__ csum, double count_hw, int B, int C,
                            float beta, float momentum, float tolerance, int update, float focal_p,
                            float* __restrict__ disc, float* __restrict__ focal) {
  const int c = threadIdx.x;
  if (c >= C) return;
  float v = chi[c];
  if (update) {
    // probs.mean(0).view(C,-1).mean(-1): mean over the batch, then over pixels
    const float avg = (float)((csum[c] / (double)B) / count_hw);
    if (avg > tolerance && v == beta) v = avg;
    v = v * momentum;
    v = v + (1.f - momentum) * avg;
    chi[c] = v;
  }
  // exp correctly rounded to fp32 through the fp64 library (19 lanes: free).  NOT guaranteed bit-equal to the reference's CPU ATen
  // (Sleef's 1-ULP expf for full vectors, libm's for vector tails -- which of the 19 classes take which depends on the host's
  // vector width): the module's default evaluates these vectors on the host for that reason (ops.HostClassVectors); this output
  // serves `SAC.device_thresholds = True` (no host round trip in the step; thresholds within 1 ULP of the host's).
  if (disc) disc[c] = 1.f - (float)exp((double)(-(v / beta)));
  if (focal) {
    const float base = 1.f - fmaxf(v, 0.f);
    float r;
    if (focal_p == 3.f) r = base * base * base;
    else if (focal_p == 2.f) r = base * base;
    else if (focal_p == 1.f) r = base;
    else r = powf(base, focal_p);
    focal[c] = r;
  }
}

}  // namespace dasac

using namespace dasac;

static float ac_scale(int n_in, int n_out) { return n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f; }

extern "C" int dasac_upsample_softmax(const float* logits, int B, int C, int h, int w, int H, int W,
                                      const uint8_t* ignore, float* up, float* probs, double* class_sums,
                                      dasac_stream_t stream) {
  DASAC_REQUIRE(logits && (up || probs || class_sums), "upsample_softmax: null pointer");
  DASAC_REQUIRE(B > 0 && C > 0 && C <= kMaxC && h > 0 && w > 0 && H > 0 && W > 0, "upsample_softmax: bad shape");
  hipStream_t s = as_stream(stream);
  if (class_sums) DASAC_HIP(hipMemsetAsync(class_sums, 0, C * sizeof(double), s));
  DASAC_REQUIRE(B < 65536 && (int64_t)H * W < (1ll << 30) && (int64_t)h * w < (1ll << 30), "upsample_softmax: plane too large");
  const int Wq = (W + 3) / 4, items = H * Wq;    // per image: grid.y is the image
  const bool softmax = probs || class_sums;
  // softmax path: a few items per thread so that the class-sum reduction at the end is amortised
  const int total = stream_grid((int64_t)B * items, kHB, softmax ? kNumCu * 4 : kNumCu * 16);
  const int grid = std::max(1, std::min((items + kHB - 1) / kHB, (total + B - 1) / B));
  const float sw = ac_scale(w, W);
  bool narrow = true;                            // tap_ac's column arithmetic, on the host (one fp32 multiply: same bits)
  for (int ox = 0; ox < W && narrow; ox += 4) {
    auto col = [&](int dst) { return std::min((int)(sw * (float)dst), w - 1); };
    narrow = col(std::min(ox + 3, W - 1)) - col(ox) <= 1;
  }
#define DASAC_UPS(CT, SM, NW)                                                                                                  \
  hipLaunchKernelGGL((upsample_softmax<CT, SM, NW>), dim3(grid, B), dim3(kHB), 0, s, logits, C, h, w, H, W, ac_scale(h, H), \
                     sw, ignore, up, probs, reinterpret_cast<unsigned long long*>(class_sums), items, fast_div(Wq))
  if (C == 19 && narrow) {
    if (softmax) DASAC_UPS(19, true, true); else DASAC_UPS(19, false, true);
  } else if (C == 19) {
    if (softmax) DASAC_UPS(19, true, false); else DASAC_UPS(19, false, false);
  } else {
    if (softmax) DASAC_UPS(kMaxC, true, false); else DASAC_UPS(kMaxC, false, false);
  }
#undef DASAC_UPS
  DASAC_CHECK_LAUNCH("upsample_softmax");
  if (class_sums) {
    hipLaunchKernelGGL(q32_to_double, dim3(1), dim3(64), 0, s, reinterpret_cast<unsigned long long*>(class_sums), C);
    DASAC_CHECK_LAUNCH("q32_to_double");
  }
  return DASAC_OK;
}

extern "C" int dasac_infer_labels(const float* logits, int B, int C, int h, int w, int H, int W, const uint8_t* lut,
                                  uint8_t* labels, float* conf, dasac_stream_t stream) {
  DASAC_REQUIRE(logits && labels, "infer_labels: null pointer");
  DASAC_REQUIRE(B > 0 && C > 0 && C <= kMaxC && h > 0 && w > 0 && H > 0 && W > 0, "infer_labels: bad shape");
  const int per = stream_grid((int64_t)H * W, kHB, (kNumCu * 16 + B - 1) / B);
  hipStream_t s = as_stream(stream);
  if (C == 19)
    hipLaunchKernelGGL(infer_labels<19>, dim3(per * B), dim3(kHB), 0, s, logits, C, h, w, H, W, ac_scale(h, H), ac_scale(w, W), lut,
                       labels, conf, per);
  else
    hipLaunchKernelGGL(infer_labels<kMaxC>, dim3(per * B), dim3(kHB), 0, s, logits, C, h, w, H, W, ac_scale(h, H), ac_scale(w, W), lut,
                       labels, conf, per);
  DASAC_CHECK_LAUNCH("infer_labels");
  return DASAC_OK;
}

extern "C" size_t dasac_upsample_bwd_workspace(int planes, int H, int w) { return (size_t)planes * H * w * sizeof(float); }

extern "C" int dasac_upsample_bwd(const float* grad_up, int planes, int h, int w, int H, int W, const float* gscale,
                                  float* grad_low, void* workspace, size_t ws_bytes, dasac_stream_t stream) {
  DASAC_REQUIRE(grad_up && grad_low && workspace, "upsample_bwd: null pointer");
  if (ws_bytes < dasac_upsample_bwd_workspace(planes, H, w)) return fail(DASAC_EWORKSPACE, "upsample_bwd: workspace too small");
  hipStream_t s = as_stream(stream);
  float* tmp = reinterpret_cast<float*>(workspace);
  const int64_t t1 = (int64_t)planes * H * w, t2 = (int64_t)planes * h * w;
  hipLaunchKernelGGL(upsample_bwd_x, dim3(stream_grid(t1, kHB)), dim3(kHB), 0, s, grad_up, H, W, w, ac_scale(w, W), tmp, t1);
  DASAC_CHECK_LAUNCH("upsample_bwd_x");
  hipLaunchKernelGGL(upsample_bwd_y, dim3(stream_grid(t2, kHB)), dim3(kHB), 0, s, tmp, H, h, w, ac_scale(h, H), gscale, grad_low, t2);
  DASAC_CHECK_LAUNCH("upsample_bwd_y");
  return DASAC_OK;
}

static int ce_blocks(int64_t HW) { return stream_grid((HW + 3) / 4, kHB, kNumCu * 8); }

extern "C" size_t dasac_ce_loss_workspace(int B, int C, int64_t HW) {
  return align_up((size_t)ce_blocks(HW) * sizeof(double), 256) + align_up((size_t)kMaxC * sizeof(double), 256);
}

extern "C" int dasac_ce_loss(const float* logits, const int64_t* labels, const float* class_weight, const float* conf,
                             int B, int C, int64_t HW, int mode, const float* gscale, float* loss, float* dlogits,
                             float* per_class, void* workspace, size_t ws_bytes, dasac_stream_t stream) {
  DASAC_REQUIRE(logits && labels && loss && workspace, "ce_loss: null pointer");
  DASAC_REQUIRE(B > 0 && C > 0 && C <= kMaxC && HW > 0 && HW < (1ll << 31) && (mode == 0 || (mode == 1 && conf)), "ce_loss: bad arguments");
  if (ws_bytes < dasac_ce_loss_workspace(B, C, HW)) return fail(DASAC_EWORKSPACE, "ce_loss: workspace too small");
  hipStream_t s = as_stream(stream);
  const int blocks = ce_blocks(HW);
  double* partial = reinterpret_cast<double*>(workspace);
  unsigned long long* pc =
      reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(workspace) + align_up((size_t)blocks * sizeof(double), 256));
  if (per_class) DASAC_HIP(hipMemsetAsync(pc, 0, kMaxC * sizeof(double), s));
  if (C == 19)
    hipLaunchKernelGGL(ce_loss<19>, dim3(blocks), dim3(kHB), 0, s, logits, labels, class_weight, conf, B, C, (int)HW, mode, gscale,
                       dlogits, partial, per_class ? pc : nullptr);
  else
    hipLaunchKernelGGL(ce_loss<kMaxC>, dim3(blocks), dim3(kHB), 0, s, logits, labels, class_weight, conf, B, C, (int)HW, mode, gscale,
                       dlogits, partial, per_class ? pc : nullptr);
  DASAC_CHECK_LAUNCH("ce_loss");
  hipLaunchKernelGGL(ce_finish, dim3(1), dim3(64), 0, s, partial, blocks, loss, pc, per_class, C, 1.0 / ((double)HW * B));
  DASAC_CHECK_LAUNCH("ce_finish");
  return DASAC_OK;
}

// tmp [B*C][H][w] floats, then (rounded up to 256 bytes) room for the per-pixel confidence sums of mode 1: H * kCeMaxW floats would
// over-allocate, so the sums live in the tail only when H * W is known to fit -- the launcher checks ws_bytes and falls back
static size_t ce_bwd_tmp_bytes(int B, int C, int H, int w) { return align_up((size_t)B * C * H * w * sizeof(float), 256); }
extern "C" size_t dasac_ce_loss_bwd_low_workspace(int B, int C, int H, int w) {
  // the confidence sums need H*W floats; W is not an argument here: w * 16 covers every up-factor to 16 (the backbone's is 8)
  return ce_bwd_tmp_bytes(B, C, H, w) + (size_t)H * w * 16 * sizeof(float);
}

extern "C" int dasac_ce_loss_bwd_low(const float* logits_up, const int64_t* labels, const float* class_weight, const float* conf,
                                     int B, int C, int H, int W, int h, int w, int mode, const float* gscale, float* grad_low,
                                     void* workspace, size_t ws_bytes, dasac_stream_t stream) {
  DASAC_REQUIRE(logits_up && labels && grad_low && workspace, "ce_loss_bwd_low: null pointer");
  DASAC_REQUIRE(B > 0 && C > 0 && C <= kMaxC && H > 0 && W > 0 && h > 0 && w > 0 && (int64_t)H * W < (1ll << 31) &&
                    (mode == 0 || (mode == 1 && conf)),
                "ce_loss_bwd_low: bad arguments");
  if (ws_bytes < ce_bwd_tmp_bytes(B, C, H, w)) return fail(DASAC_EWORKSPACE, "ce_loss_bwd_low: workspace too small");
  // segments of low-resolution columns per block: the x range of a segment (+ the taps' reach on both sides) bounds the LDS
  // row; ~300 pixels (about 24 KB for 19 classes) lets six blocks share a CU
  const float sw = ac_scale(w, W);
  int seg_cols = w;
  if (sw > 0.f) seg_cols = (int)(300.f * sw) - 2;
  seg_cols = seg_cols < 4 ? 4 : (seg_cols > w ? w : seg_cols);
  const int n_seg = (w + seg_cols - 1) / seg_cols;
  int span = 0;
  for (int sg = 0; sg < n_seg; ++sg) {
    int xs, xe, dummy;
    src_range(sg * seg_cols, sw, W, xs, dummy);
    src_range(std::min(w, (sg + 1) * seg_cols) - 1, sw, W, dummy, xe);
    xs &= ~3;
    span = std::max(span, (xe | 3) + 1 - xs);              // whole quads
  }
  const int pitch = ce_lds_index(span - 1) + 2;
  const size_t lds = (size_t)C * pitch * sizeof(float);
  DASAC_REQUIRE(lds <= 160 * 1024, "ce_loss_bwd_low: a segment of C x W gradients does not fit LDS");
  hipStream_t s = as_stream(stream);
  float* tmp = reinterpret_cast<float*>(workspace);
  if (lds > 64 * 1024) {                                 // raise the kernels' dynamic-LDS limit once per device, not per launch
    static std::atomic<unsigned long long> raised{0};
    int dev = 0;
    DASAC_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(raised.load(std::memory_order_relaxed) & bit)) {
      DASAC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ce_bwd_rows<19>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      DASAC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ce_bwd_rows<kMaxC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      raised.fetch_or(bit, std::memory_order_relaxed);
    }
  }
  DASAC_REQUIRE((int64_t)B * H * n_seg < (1ll << 31), "ce_loss_bwd_low: grid too large");
  // mode 1: every image's rows need sum_i conf_i of their pixels -- B x B plane reads if each block adds them up itself; one pass
  // into the workspace tail instead (when the caller's workspace has the room: H * W floats behind tmp)
  const float* cs_pix = nullptr;
  if (mode == 1 && B > 1 && ws_bytes >= ce_bwd_tmp_bytes(B, C, H, w) + (size_t)H * W * sizeof(float)) {
    float* cs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ce_bwd_tmp_bytes(B, C, H, w));
    hipLaunchKernelGGL(conf_pixel_sums, dim3(stream_grid((int64_t)H * W, 256)), dim3(256), 0, s, conf, B, H * W, cs);
    DASAC_CHECK_LAUNCH("conf_pixel_sums");
    cs_pix = cs;
  }
  // the one-wave kernel: 19 classes, segments of at most 256 pixels, at most 24 taps per column, confidence sums precomputed
  bool wave_done = false;
  if (C == 19 && sw > 0.f && W >= 4 && (mode == 0 || cs_pix)) {
    int cols_max = std::min((int)(250.f * sw) - 1, kCW);   // a segment: <= 256 pixels for phase 1, <= 64 columns for phase 2
    cols_max = cols_max < 1 ? 1 : (cols_max > w ? w : cols_max);
    const int wcols = (w + (w + cols_max - 1) / cols_max - 1) / ((w + cols_max - 1) / cols_max);   // balanced segments
    const int wseg = (w + wcols - 1) / wcols;
    int wspan = 0, taps = 0;
    for (int sg = 0; sg * wcols < w; ++sg) {
      int xs, xe, dummy;
      src_range(sg * wcols, sw, W, xs, dummy);
      src_range(std::min(w, (sg + 1) * wcols) - 1, sw, W, dummy, xe);
      xs &= ~3;
      wspan = std::max(wspan, (xe | 3) + 1 - xs);
    }
    for (int jj = 0; jj < w; ++jj) {
      int lo, hi;
      ce_taps(jj, sw, W, w, lo, hi);
      taps = std::max(taps, hi - lo + 1);
    }
    // rows per block: ONE round of blocks that nearly fills the chip (two 224-register waves per SIMD = 8 blocks per CU); all
    // blocks do the same work, so a second, partly filled round would cost a whole round's time
    const int slots = (kNumCu - reserved_cus()) * 8;
    const int chunks_max = std::max(1, slots / std::max(1, B * wseg));
    const int rows = std::max(4, (H + chunks_max - 1) / chunks_max);
    const int n_chunks = (H + rows - 1) / rows;
    const int wpitch = ce_lds_index(wspan - 1) + 2;
    if (wspan <= 256 && taps <= 24 && (int64_t)B * n_chunks * wseg < (1ll << 31) && (size_t)C * wpitch * sizeof(float) <= 60 * 1024) {
      const dim3 grid((unsigned)(B * n_chunks * wseg));
      const size_t wlds = (size_t)C * wpitch * sizeof(float);
      if (taps <= 16)
        hipLaunchKernelGGL((ce_bwd_rows_wave<19, 16>), grid, dim3(kCW), wlds, s, logits_up, labels, class_weight, B, H, W, w, sw, mode,
                           gscale, tmp, wcols, wseg, wpitch, cs_pix, rows, n_chunks);
      else
        hipLaunchKernelGGL((ce_bwd_rows_wave<19, 24>), grid, dim3(kCW), wlds, s, logits_up, labels, class_weight, B, H, W, w, sw, mode,
                           gscale, tmp, wcols, wseg, wpitch, cs_pix, rows, n_chunks);
      wave_done = true;
    }
  }
  if (wave_done) {
  } else if (C == 19)
    hipLaunchKernelGGL(ce_bwd_rows<19>, dim3(B * H * n_seg), dim3(kCB), lds, s, logits_up, labels, class_weight, conf, B, C, H, W, w,
                       sw, mode, gscale, tmp, seg_cols, n_seg, pitch, cs_pix);
  else
    hipLaunchKernelGGL(ce_bwd_rows<kMaxC>, dim3(B * H * n_seg), dim3(kCB), lds, s, logits_up, labels, class_weight, conf, B, C, H, W, w,
                       sw, mode, gscale, tmp, seg_cols, n_seg, pitch, cs_pix);
  DASAC_CHECK_LAUNCH("ce_bwd_rows");
  const int64_t t2 = (int64_t)B * C * h * w;
  hipLaunchKernelGGL(upsample_bwd_y, dim3(stream_grid(t2, kHB)), dim3(kHB), 0, s, tmp, H, h, w, ac_scale(h, H), nullptr, grad_low, t2);
  DASAC_CHECK_LAUNCH("upsample_bwd_y");
  return DASAC_OK;
}

extern "C" int dasac_warp_affine(const float* x, const float* theta, int B, int C, int H, int W, float* out,
                                 dasac_stream_t stream) {
  DASAC_REQUIRE(x && theta && out && B > 0 && C > 0 && H > 0 && W > 0, "warp_affine: bad arguments");
  const int per = stream_grid((int64_t)H * W, kHB, (kNumCu * 16 + B - 1) / B);
  hipLaunchKernelGGL(warp_affine, dim3(per * B), dim3(kHB), 0, as_stream(stream), x, theta, C, H, W, out, per);
  DASAC_CHECK_LAUNCH("warp_affine");
  return DASAC_OK;
}

extern "C" int dasac_warp_pool(const float* probs, const float* theta, const float* theta_inv, int N, int T, int C, int H,
                               int W, int mode, float tolerance, float* aligned, float* pooled, float* mask,
                               dasac_stream_t stream) {
  DASAC_REQUIRE(probs && pooled && mask && ((theta && theta_inv) || (!theta && !theta_inv && !aligned)), "warp_pool: null pointer");
  DASAC_REQUIRE(N > 0 && T > 0 && C > 0 && C <= kMaxC && (mode == 0 || mode == 1), "warp_pool: bad arguments");
  DASAC_REQUIRE(H > 0 && W > 0 && (int64_t)H * W < (1ll << 30), "warp_pool: bad shape");
  const int per = stream_grid((int64_t)H * W, kHB, (kNumCu * 16 + N - 1) / N);
  hipStream_t s = as_stream(stream);
  const int items_r = (H + kWpRows - 1) / kWpRows * W;
  const int per_r = stream_grid(items_r, kHB, (kNumCu * 16 + N - 1) / N);
#define DASAC_WPA(TT) hipLaunchKernelGGL((warp_pool_avg<19, TT>), dim3(per_r * N), dim3(kHB), 0, s, probs, theta, theta_inv, H, W, tolerance, aligned, pooled, mask, per_r, items_r, fast_div(W))
  if (mode == 0 && theta && C == 19 && T == 4) DASAC_WPA(4);
  else if (mode == 0 && theta && C == 19 && T == 2) DASAC_WPA(2);
  else if (mode == 0 && theta && C == 19 && T == 1) DASAC_WPA(1);
  else
    hipLaunchKernelGGL(warp_pool, dim3(per * N), dim3(kHB), 0, s, probs, theta, theta_inv, T, C, H, W, mode, tolerance, aligned, pooled,
                       mask, per);
#undef DASAC_WPA
  DASAC_CHECK_LAUNCH("warp_pool");
  return DASAC_OK;
}

extern "C" int dasac_warp_back(const float* pooled, const float* mask, const float* theta_inv, int B, int views_per_group,
                               int C, int H, int W, float* refined, dasac_stream_t stream) {
  DASAC_REQUIRE(pooled && mask && theta_inv && refined && B > 0 && views_per_group > 0, "warp_back: bad arguments");
  DASAC_REQUIRE(C > 0 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 30), "warp_back: bad shape");
  if (W < 2) {          // the pair gathers need two columns: scalar taps for one-column maps
    const int per1 = stream_grid((int64_t)H * W, kHB, (kNumCu * 16 + B - 1) / B);
    hipLaunchKernelGGL(warp_back_scalar, dim3(per1 * B), dim3(kHB), 0, as_stream(stream), pooled, mask, theta_inv, views_per_group, C,
                       H, W, refined, per1);
    DASAC_CHECK_LAUNCH("warp_back");
    return DASAC_OK;
  }
  const int items = (H + kWbRows - 1) / kWbRows * W;      // groups of kWbRows rows
  const int per = stream_grid(items, kHB, (kNumCu * 16 + B - 1) / B);
  hipLaunchKernelGGL(warp_back, dim3(per * B), dim3(kHB), 0, as_stream(stream), pooled, mask, theta_inv, views_per_group, C, H,
                     W, refined, per, items, fast_div(W));
  DASAC_CHECK_LAUNCH("warp_back");
  return DASAC_OK;
}

extern "C" int dasac_class_state(float* running_conf, const double* class_sums, int B, int64_t HW, int C, float beta,
                                 float stat_momentum, int update, float focal_p, float* disc, float* focal,
                                 dasac_stream_t stream) {
  DASAC_REQUIRE(running_conf && C > 0 && C <= 64 && (!update || class_sums), "class_state: bad arguments");
  hipLaunchKernelGGL(class_state, dim3(1), dim3(64), 0, as_stream(stream), running_conf, class_sums, (double)HW, B, C, beta,
                     stat_momentum, 1e-8f, update, focal_p, disc, focal);
  DASAC_CHECK_LAUNCH("class_state");
  return DASAC_OK;
}

