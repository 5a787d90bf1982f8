// Small HBM-bound helpers around the conv GEMMs: frozen-BN folding and parameter gradients,
// max-pooling, per-channel reductions, the momentum-teacher multi-tensor update, Dropout2d scaling.
#include "common.hpp"

namespace dasac {

// ---- frozen BatchNorm (eval mode; models/__init__.py:29 + models/basenet.py:97-100) -------------
// ATen eval batch_norm: invstd = 1/sqrt(var+eps); alpha = gamma*invstd; beta' = beta - mean*alpha
__global__ void bn_fold(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                        const float* __restrict__ var, const float* __restrict__ conv_bias, float eps, int C,
                        float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.f / sqrtf(var[c] + eps);
  const float a = gamma[c] * is;
  const float m = conv_bias ? mean[c] - conv_bias[c] : mean[c];   // BN(conv + b) = a*conv + (beta - (mean-b)*a)
  scale[c] = a;
  shift[c] = beta[c] - m * a;
  if (invstd) invstd[c] = is;
}

// every frozen BN of a network in one launch: chunk (j, c) = 256 channels of job j
__global__ __launch_bounds__(256) void bn_fold_multi(const dasac_fold_job* __restrict__ jobs, const int2* __restrict__ chunks) {
  const int2 ch = chunks[blockIdx.x];
  const dasac_fold_job jb = jobs[ch.x];
  const int c = ch.y * 256 + threadIdx.x;
  if (c >= jb.C) return;
  const float is = 1.f / sqrtf(jb.var[c] + jb.eps);
  const float a = jb.gamma[c] * is;
  const float m = jb.conv_bias ? jb.mean[c] - jb.conv_bias[c] : jb.mean[c];
  jb.scale[c] = a;
  jb.shift[c] = jb.beta[c] - m * a;
  jb.invstd[c] = is;
}

// dgamma = invstd*(dot + (b - mean)*sum_dz), dbeta = sum_dz, dbias_conv = scale*sum_dz
// dot: [dot_rows][C] partial rows of dasac_conv_wgrad_finish, added here in row order (deterministic)
__global__ void bn_param_grads(const float* __restrict__ dot, int dot_rows, const float* __restrict__ sum_dz,
                               const float* __restrict__ mean,
                               const float* __restrict__ invstd, const float* __restrict__ scale,
                               const float* __restrict__ conv_bias, int C, float* __restrict__ dgamma,
                               float* __restrict__ dbeta, float* __restrict__ dbias) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float s = sum_dz[c];
  const float b = conv_bias ? conv_bias[c] : 0.f;
  if (dgamma) {
    float d = 0.f;
    for (int r = 0; r < dot_rows; ++r) d += dot[(size_t)r * C + c];
    dgamma[c] = invstd[c] * (d + (b - mean[c]) * s);
  }
  if (dbeta) dbeta[c] = s;
  if (dbias) dbias[c] = (scale ? scale[c] : 1.f) * s;
}

// out[c] = sum_{n,p} x[n,c,p]; one block per channel
__global__ __launch_bounds__(256) void channel_sums(const float* __restrict__ x, int N, int C, int HW, float* __restrict__ out) {
  const int c = blockIdx.x;
  double s = 0;
  for (int n = 0; n < N; ++n) {
    const float* p = x + ((size_t)n * C + c) * HW;
    float part = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) part += p[i];
    s += part;
  }
  __shared__ double red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[c] = (float)(red[0] + red[1] + red[2] + red[3]);
}

// ---- max pooling (deeplabv2.py:126 3x3/2 pad 1 ceil_mode; torchvision VGG 2x2/2) -------------------
// forward also records the argmax position inside the window (kh*k + kw), first maximum wins.
// KT / ST: compile-time window and stride (0 = take the runtime values): the 3x3/2 stem pool and the 2x2/2, 3x3/1 VGG pools
// get unrolled windows and no integer division by a runtime value; all index math is 32-bit (total < 2^31, checked by the host).
// argmax byte: bits 0-6 = position inside the window (kh*k + kw), bit 7 = (pooled value > 0) for windows up to 11 x 11 -- the
// backward pass with the producer's ReLU folded in (`relu_mask`) then needs no read of the pooled tensor (a third of its bytes)
constexpr int kPoolSignMaxK = 11;
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

template <int KT, int ST>
__global__ __launch_bounds__(256) void maxpool_fwd(const float* __restrict__ x, int H, int W, int OH, int OW, int k_rt, int s_rt,
                                                   int pad, float* __restrict__ y, uint8_t* __restrict__ arg, int total) {
  const int k = KT ? KT : k_rt, s = ST ? ST : s_rt;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int ow = i % OW, r = i / OW;
    const int oh = r % OH, plane = r / OH;
    const float* xp = x + (size_t)plane * H * W;
    const int ih0 = oh * s - pad, iw0 = ow * s - pad;
    float best = -INFINITY;
    int code = 0;
#pragma unroll
    for (int a = 0; a < (KT ? KT : 15); ++a) {
      if (!KT && a >= k) break;
      const int ih = ih0 + a;
      const bool rowok = (unsigned)ih < (unsigned)H;
#pragma unroll
      for (int b = 0; b < (KT ? KT : 15); ++b) {
        if (!KT && b >= k) break;
        const int iw = iw0 + b;
        if (rowok && (unsigned)iw < (unsigned)W) {
          const float v = xp[ih * W + iw];
          if (v > best || v != v) {
            best = v;
            code = a * k + b;
          }
        }
      }
    }
    y[i] = best;
    arg[i] = (uint8_t)(code | ((k <= kPoolSignMaxK && best > 0.f) ? 0x80 : 0));
  }
}

// The 3x3 / stride-2 stem pool (deeplabv2.py:126) and the 2x2 / 2 VGG pools with FOUR horizontally adjacent outputs per thread:
// the 3*ST + KT input columns under them are read once per window row as two dwordx4 + (KT + 3*ST - 8) scalars (9 load
// instructions per 4 outputs for 3x3/2 where the one-output kernel issues 36 dword loads), y leaves as one dwordx4.  Same scan
// order per output (rows outer, columns inner, first maximum wins, NaN propagates): identical y and argmax codes.
template <int KT, int ST>
__global__ __launch_bounds__(256) void maxpool_fwd_quad(const float* __restrict__ x, int H, int W, int OH, int OW, int pad,
                                                        float* __restrict__ y, uint8_t* __restrict__ arg, int items) {
  constexpr int NCOL = 3 * ST + KT;
  static_assert(NCOL >= 8 && NCOL <= 12, "maxpool_fwd_quad: two dwordx4 loads + up to four scalars per window row");
  const int OWq = (OW + 3) >> 2;
  for (int it = blockIdx.x * 256 + threadIdx.x; it < items; it += gridDim.x * 256) {
    const int q = it % OWq, r = it / OWq;
    const int oh = r % OH, plane = r / OH;
    const int ow0 = q * 4, nx = min(4, OW - ow0);
    const float* xp = x + (size_t)plane * H * W;
    const int ih0 = oh * ST - pad, iw0 = ow0 * ST - pad;
    const bool inner = iw0 >= 0 && iw0 + NCOL <= W;
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int code[4] = {0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < KT; ++a) {
      const int ih = ih0 + a;
      if ((unsigned)ih >= (unsigned)H) continue;
      const float* row = xp + ih * W + iw0;
      float v[NCOL];
      if (inner) {
        const f32x4u v0 = *reinterpret_cast<const f32x4u*>(row), v1 = *reinterpret_cast<const f32x4u*>(row + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          v[c] = v0[c];
          v[4 + c] = v1[c];
        }
#pragma unroll
        for (int c = 8; c < NCOL; ++c) v[c] = row[c];
      } else {
#pragma unroll
        for (int c = 0; c < NCOL; ++c) v[c] = (unsigned)(iw0 + c) < (unsigned)W ? row[c] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int b = 0; b < KT; ++b) {
          const int c = e * ST + b;
          const float val = v[c];
          if ((inner || (unsigned)(iw0 + c) < (unsigned)W) && (val > best[e] || val != val)) {
            best[e] = val;
            code[e] = a * KT + b;
          }
        }
    }
    const size_t o = ((size_t)plane * OH + oh) * OW + ow0;
    if (nx == 4) {
      *reinterpret_cast<f32x4u*>(y + o) = f32x4u{best[0], best[1], best[2], best[3]};
    } else {
#pragma unroll
      for (int e = 0; e < 3; ++e)
        if (e < nx) y[o + e] = best[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < nx) arg[o + e] = (uint8_t)(code[e] | (best[e] > 0.f ? 0x80 : 0));
  }
}

// dx[ih,iw] = sum over windows o containing (ih,iw) with argmax(o) == this position of dy(o)
// relu_mask: additionally require y(o) > 0 -- the pooled tensor is ReLU output, so this is exactly
// the ReLU backward of the producer folded in (no need to keep the un-pooled activation).
// Four consecutive input pixels of one row per thread: the (at most 2 x 3 for the 3x3/2 stem pool) windows that can
// have picked any of them are read ONCE (argmax code, pooled value, gradient) and each routes its gradient to the
// pixel its code names -- a scatter inside the thread's registers, no atomics; one dwordx4 store.  32-bit index math.
template <int KT, int ST>
__global__ __launch_bounds__(256) void maxpool_bwd(const float* __restrict__ dy, const float* __restrict__ y,
                                                   const uint8_t* __restrict__ arg, int H, int W, int OH, int OW, int k_rt,
                                                   int s_rt, int pad, int relu_mask, float* __restrict__ dx, int items) {
  const int k = KT ? KT : k_rt, s = ST ? ST : s_rt;
  const int Wq = (W + 3) >> 2;
  for (int it = blockIdx.x * 256 + threadIdx.x; it < items; it += gridDim.x * 256) {
    const int q = it % Wq, row = it / Wq;            // row = plane*H + ih
    const int ih = row % H, plane = row / H;
    const int iw0 = q * 4, nx = min(4, W - iw0);
    // windows: oh*s - pad <= ih <= oh*s - pad + k - 1, same for the columns iw0 .. iw0+3
    int oh_lo = (ih + pad - k + 1 + s - 1) / s, oh_hi = min((ih + pad) / s, OH - 1);
    int ow_lo = (iw0 + pad - k + 1 + s - 1) / s, ow_hi = min((iw0 + nx - 1 + pad) / s, OW - 1);
    if (ih + pad - k + 1 < 0) oh_lo = 0;
    if (iw0 + pad - k + 1 < 0) ow_lo = 0;
    const size_t ob = (size_t)plane * OH * OW;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (KT > 0 && ST > 0) {
      // compile-time window / stride: at most NR x NC windows can have picked one of the four pixels.  ALL their argmax
      // codes, pooled values and gradients are loaded first (clamped addresses, one batch in flight), then routed -- the
      // runtime-bounded loop below serialises three dependent loads per window.
      constexpr int NR = (KT + ST - 1) / ST, NC = (3 + KT + ST - 1) / ST;
      int code[NR][NC];
      float yv[NR][NC], dv[NR][NC];
#pragma unroll
      for (int a = 0; a < NR; ++a)
#pragma unroll
        for (int b = 0; b < NC; ++b) {
          const size_t o = ob + (size_t)min(oh_lo + a, OH - 1) * OW + min(ow_lo + b, OW - 1);
          const int ab = arg[o];
          code[a][b] = ab & 0x7f;
          dv[a][b] = dy[o];
          yv[a][b] = !relu_mask ? 1.f : (KT <= kPoolSignMaxK ? (float)(ab >> 7) : y[o]);    // bit 7: pooled value > 0
        }
#pragma unroll
      for (int a = 0; a < NR; ++a)
#pragma unroll
        for (int b = 0; b < NC; ++b) {
          const int oh = oh_lo + a, ow = ow_lo + b;
          const int r = oh * s - pad + code[a][b] / k, c = ow * s - pad + code[a][b] % k - iw0;
          if (oh <= oh_hi && ow <= ow_hi && r == ih && (unsigned)c < 4u && yv[a][b] > 0.f) {
            const float d = dv[a][b];
            g[0] += c == 0 ? d : 0.f;
            g[1] += c == 1 ? d : 0.f;
            g[2] += c == 2 ? d : 0.f;
            g[3] += c == 3 ? d : 0.f;
          }
        }
    } else {
    for (int oh = oh_lo; oh <= oh_hi; ++oh)
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const size_t o = ob + (size_t)oh * OW + ow;
        const int ab = arg[o];
        const int code = k <= kPoolSignMaxK ? (ab & 0x7f) : ab;
        const int r = oh * s - pad + code / k, c = ow * s - pad + code % k - iw0;
        if (r == ih && (unsigned)c < 4u && (!relu_mask || (k <= kPoolSignMaxK ? (ab >> 7) != 0 : y[o] > 0.f))) {
          const float d = dy[o];
          g[0] += c == 0 ? d : 0.f;
          g[1] += c == 1 ? d : 0.f;
          g[2] += c == 2 ? d : 0.f;
          g[3] += c == 3 ? d : 0.f;
        }
      }
    }
    float* out = dx + (size_t)row * W + iw0;
    if (nx == 4) {
      *reinterpret_cast<f32x4u*>(out) = f32x4u{g[0], g[1], g[2], g[3]};
    } else {
#pragma unroll
      for (int e = 0; e < 3; ++e)
        if (e < nx) out[e] = g[e];
    }
  }
}

// The stem pool (3x3, stride 2, padding 1: deeplabv2.py:126) with a 2 x 4 input block per thread -- rows 2p, 2p+1, columns
// 4q .. 4q+3.  Round 5: the kernel above is ISSUE-bound, not byte-bound (tools/isa_count.py: 306 VALU + 122 SALU instructions per
// 16 bytes stored = 330 of its 400 us at 16 x 64 x 385^2); this one spends ~1/4 of that per pixel:
//   * window oh covers input rows 2oh-1 .. 2oh+1, so the block's rows meet window rows p and p+1 only and its columns meet window
//     columns 2q .. 2q+2: SIX windows per eight pixels (the one-row kernel reads six per four), each read once;
//   * which (window, argmax code) pairs can name which pixel is known at compile time: row 2p <- (p, a=1); row 2p+1 <- (p, a=2),
//     (p+1, a=0); column 4q <- (2q, b=1); 4q+1 <- (2q, b=2), (2q+1, b=0); 4q+2 <- (2q+1, b=1); 4q+3 <- (2q+1, b=2), (2q+2, b=0)
//     -- 18 compare/select/add triples instead of a computed scatter over every pixel of the thread;
//   * the two index divisions are multiplications (FastDiv).
// Every pixel adds its candidate windows in the order the kernel above does (window rows ascending, columns ascending, from
// +0), a window that does not exist or whose pooled value is not positive (ReLU bit) contributes +0: identical bits.
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
__global__ __launch_bounds__(256) void maxpool_bwd_3s2p1(const float* __restrict__ dy, const uint8_t* __restrict__ arg, int H, int W,
                                                         int OH, int OW, int relu_mask, float* __restrict__ dx, int items, int Hp,
                                                         int Wq, FastDiv div_wq, FastDiv div_hp) {
  for (int it = blockIdx.x * 256 + threadIdx.x; it < items; it += gridDim.x * 256) {
    const int rowp = fdiv(it, div_wq), q = it - rowp * Wq;       // rowp = plane * Hp + p
    const int plane = fdiv(rowp, div_hp), p = rowp - plane * Hp;
    const int ih = 2 * p, iw0 = 4 * q, ow0 = 2 * q;
    const size_t ob = (size_t)plane * OH * OW;
    float d[2][3];
    int code[2][3];
    const bool wide = ow0 + 2 < OW;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const bool rok = p + a < OH;
      const size_t o = ob + (size_t)min(p + a, OH - 1) * OW + ow0;
      float v[3];
      int ab[3];
      if (wide) {
        const f32x2u v01 = *reinterpret_cast<const f32x2u*>(dy + o);
        v[0] = v01[0];
        v[1] = v01[1];
        v[2] = dy[o + 2];
#pragma unroll
        for (int b = 0; b < 3; ++b) ab[b] = arg[o + b];
      } else {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const size_t ob_ = o + min(b, OW - 1 - ow0);
          v[b] = dy[ob_];
          ab[b] = arg[ob_];
        }
      }
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const bool ok = rok && ow0 + b < OW && (!relu_mask || (ab[b] & 0x80));
        d[a][b] = ok ? v[b] : 0.f;
        code[a][b] = ab[b] & 0x7f;
      }
    }
    auto pick = [&](int a, int ka, int b, int kb) { return code[a][b] == ka * 3 + kb ? d[a][b] : 0.f; };
    float g0[4], g1[4];
    // row 2p: window row p with a = 1
    g0[0] = 0.f + pick(0, 1, 0, 1);
    g0[1] = (0.f + pick(0, 1, 0, 2)) + pick(0, 1, 1, 0);
    g0[2] = 0.f + pick(0, 1, 1, 1);
    g0[3] = (0.f + pick(0, 1, 1, 2)) + pick(0, 1, 2, 0);
    // row 2p+1: window row p with a = 2, then window row p+1 with a = 0
    g1[0] = (0.f + pick(0, 2, 0, 1)) + pick(1, 0, 0, 1);
    g1[1] = (((0.f + pick(0, 2, 0, 2)) + pick(0, 2, 1, 0)) + pick(1, 0, 0, 2)) + pick(1, 0, 1, 0);
    g1[2] = (0.f + pick(0, 2, 1, 1)) + pick(1, 0, 1, 1);
    g1[3] = (((0.f + pick(0, 2, 1, 2)) + pick(0, 2, 2, 0)) + pick(1, 0, 1, 2)) + pick(1, 0, 2, 0);
    float* out = dx + ((size_t)plane * H + ih) * W + iw0;
    const bool two = ih + 1 < H;
    if (iw0 + 4 <= W) {
      *reinterpret_cast<f32x4u*>(out) = f32x4u{g0[0], g0[1], g0[2], g0[3]};
      if (two) *reinterpret_cast<f32x4u*>(out + W) = f32x4u{g1[0], g1[1], g1[2], g1[3]};
    } else {
      const int nx = W - iw0;
#pragma unroll
      for (int e = 0; e < 3; ++e)
        if (e < nx) {
          out[e] = g0[e];
          if (two) out[W + e] = g1[e];
        }
    }
  }
}

// ---- momentum teacher (models/sac.py:83-102) as one multi-tensor launch --------------------------
// chunk table: for chunk i, tensor id + chunk index.  sq_chunk[i] = sum (slow-fast)^2 over the chunk (pre-update);
// if update: slow = slow*m + fast*(1-m).
struct TensorPair {
  const float* fast;
  float* slow;
  int64_t n;
};
constexpr int kEmaChunk = 256 * 16;

__global__ __launch_bounds__(256) void ema_chunks(const TensorPair* __restrict__ pairs, const int2* __restrict__ chunks,
                                                  float momentum, float one_minus, int update, double* __restrict__ sq_chunk) {
  const int2 ch = chunks[blockIdx.x];
  const TensorPair tp = pairs[ch.x];
  const int64_t base = (int64_t)ch.y * kEmaChunk;
  double acc = 0;
#pragma unroll 4
  for (int j = 0; j < 16; ++j) {
    const int64_t i = base + j * 256 + threadIdx.x;
    if (i < tp.n) {
      const float f = tp.fast[i], sl = tp.slow[i];
      const float d = sl - f;
      acc += (double)d * (double)d;
      if (update) {
        float v = sl * momentum;
        v = v + f * one_minus;
        tp.slow[i] = v;
      }
    }
  }
  __shared__ double red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) sq_chunk[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);      // one partial per chunk: no atomics
}

// sq[t] = sum of tensor t's chunk partials, in a fixed order (one wave per tensor; `chunks` is sorted by tensor id)
__global__ __launch_bounds__(64) void ema_tensor_sums(const int2* __restrict__ chunks, int n_chunks, const double* __restrict__ sq_chunk,
                                                      double* __restrict__ sq) {
  const int t = blockIdx.x;
  int lo = 0, hi = n_chunks;                       // first chunk whose tensor id is >= t
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (chunks[mid].x < t) lo = mid + 1; else hi = mid;
  }
  double s = 0;
  for (int i = lo + threadIdx.x; i < n_chunks && chunks[i].x == t; i += 64) s += sq_chunk[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) sq[t] = s;
}

__global__ void ema_finish(const double* __restrict__ sq, int n, float* __restrict__ out) {
  double s = 0;
  for (int i = threadIdx.x; i < n; i += 64) s += sqrt(sq[i]);
  s = wave_sum(s);
  if (threadIdx.x == 0) out[0] = (float)s;
}

// ---- torch.optim.SGD(momentum, dampening 0, no nesterov) over every parameter in one launch ---------
// (base_trainer.py:63-66 builds it over the four groups of basenet.py:73-95; per-group lr / weight decay)
//   d = g (+ g2) + wd*p ;  buf = first ? d : momentum*buf + d ;  p = p - lr*buf        (same op order as ATen)
// g2 (optional): a second gradient of the same parameter -- the source-pass gradient the driver set aside while the target
// pass ran (train.py accumulates both into .grad: one `add_` launch per parameter, 320 per step; here the sum happens in
// the update itself, g2 + g in AccumulateGrad's operand order, so the result is bit-identical).
struct SgdTensor {
  float* p;
  const float* g;
  const float* g2;
  float* buf;
  int64_t n;
  int64_t group;
};
struct SgdGroups {
  float lr[8], wd[8];
};

__global__ __launch_bounds__(256) void sgd_chunks(const SgdTensor* __restrict__ tensors, const int2* __restrict__ chunks,
                                                  SgdGroups hp, float momentum, int first) {
  const int2 ch = chunks[blockIdx.x];
  const SgdTensor t = tensors[ch.x];
  const float lr = hp.lr[t.group], wd = hp.wd[t.group];
  const int64_t base = (int64_t)ch.y * kEmaChunk;
#pragma unroll 4
  for (int j = 0; j < 16; ++j) {
    const int64_t i = base + j * 256 + threadIdx.x;
    if (i < t.n) {
      const float p = t.p[i];
      float d = t.g[i];
      if (t.g2) d = t.g2[i] + d;
      if (wd != 0.f) d = d + wd * p;
      float b = d;
      if (!first) {
        b = t.buf[i] * momentum;
        b = b + d;
      }
      t.buf[i] = b;
      t.p[i] = p - lr * b;
    }
  }
}

// y[n,c,:] = x[n,c,:] * m[n,c]   (Dropout2d with an explicit keep/(1-p) mask, fcn.py:52,56)
__global__ __launch_bounds__(256) void scale_planes(const float* __restrict__ x, const float* __restrict__ m, int HW,
                                                    float* __restrict__ y, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    y[i] = x[i] * m[i / HW];
}

// out = a + b (gradient joins); c = a*alpha elementwise helpers
__global__ __launch_bounds__(256) void add2(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                            int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = a[i] + b[i];
}

__global__ __launch_bounds__(256) void relu_mask(const float* __restrict__ dy, const float* __restrict__ y,
                                                 float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = y[i] > 0.f ? dy[i] : 0.f;
}

}  // namespace dasac

using namespace dasac;

extern "C" int dasac_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                             const float* conv_bias, float eps, int C, float* scale, float* shift, float* invstd,
                             dasac_stream_t stream) {
  DASAC_REQUIRE(gamma && beta && mean && var && scale && shift && C > 0, "bn_fold: bad arguments");
  hipLaunchKernelGGL(bn_fold, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), gamma, beta, mean, var, conv_bias, eps, C,
                     scale, shift, invstd);
  DASAC_CHECK_LAUNCH("bn_fold");
  return DASAC_OK;
}

extern "C" int dasac_bn_fold_multi(const dasac_fold_job* jobs, const int32_t* chunks, int n_chunks, dasac_stream_t stream) {
  DASAC_REQUIRE(jobs && chunks && n_chunks > 0, "bn_fold_multi: bad arguments");
  hipLaunchKernelGGL(bn_fold_multi, dim3(n_chunks), dim3(256), 0, as_stream(stream), jobs, reinterpret_cast<const int2*>(chunks));
  DASAC_CHECK_LAUNCH("bn_fold_multi");
  return DASAC_OK;
}

extern "C" int dasac_bn_param_grads(const float* dot, int dot_rows, const float* sum_dz, const float* mean, const float* invstd,
                                    const float* scale, const float* conv_bias, int C, float* dgamma, float* dbeta,
                                    float* dbias, dasac_stream_t stream) {
  DASAC_REQUIRE(sum_dz && C > 0 && (!dgamma || (dot && dot_rows > 0 && mean && invstd)), "bn_param_grads: bad arguments");
  hipLaunchKernelGGL(bn_param_grads, dim3((C + 63) / 64), dim3(64), 0, as_stream(stream), dot, dot_rows, sum_dz, mean, invstd, scale,
                     conv_bias, C, dgamma, dbeta, dbias);
  DASAC_CHECK_LAUNCH("bn_param_grads");
  return DASAC_OK;
}

extern "C" int dasac_channel_sums(const float* x, int N, int C, int64_t HW, float* out, dasac_stream_t stream) {
  DASAC_REQUIRE(x && out && N > 0 && C > 0 && HW > 0 && HW < (1ll << 31), "channel_sums: bad arguments");
  hipLaunchKernelGGL(channel_sums, dim3(C), dim3(256), 0, as_stream(stream), x, N, C, (int)HW, out);
  DASAC_CHECK_LAUNCH("channel_sums");
  return DASAC_OK;
}

extern "C" int dasac_maxpool_fwd(const float* x, int planes, int H, int W, int OH, int OW, int k, int s, int pad, float* y,
                                 uint8_t* argmax, dasac_stream_t stream) {
  DASAC_REQUIRE(x && y && argmax && planes > 0 && k > 0 && k <= 15 && s > 0, "maxpool_fwd: bad arguments");
  const int64_t total = (int64_t)planes * OH * OW;
  DASAC_REQUIRE(total < (1ll << 31), "maxpool_fwd: tensor too large");
  const dim3 grid(stream_grid(total, 256));
  hipStream_t st = as_stream(stream);
  const int items4 = planes * OH * ((OW + 3) / 4);
  const dim3 grid4(stream_grid(items4, 256));
  if (k == 3 && s == 2) hipLaunchKernelGGL((maxpool_fwd_quad<3, 2>), grid4, dim3(256), 0, st, x, H, W, OH, OW, pad, y, argmax, items4);
  else if (k == 2 && s == 2) hipLaunchKernelGGL((maxpool_fwd_quad<2, 2>), grid4, dim3(256), 0, st, x, H, W, OH, OW, pad, y, argmax, items4);
  else if (k == 3 && s == 1) hipLaunchKernelGGL((maxpool_fwd<3, 1>), grid, dim3(256), 0, st, x, H, W, OH, OW, k, s, pad, y, argmax, (int)total);
  else hipLaunchKernelGGL((maxpool_fwd<0, 0>), grid, dim3(256), 0, st, x, H, W, OH, OW, k, s, pad, y, argmax, (int)total);
  DASAC_CHECK_LAUNCH("maxpool_fwd");
  return DASAC_OK;
}

extern "C" int dasac_maxpool_bwd(const float* dy, const float* y, const uint8_t* argmax, int planes, int H, int W, int OH,
                                 int OW, int k, int s, int pad, int relu_mask, float* dx, dasac_stream_t stream) {
  DASAC_REQUIRE(dy && y && argmax && dx && planes > 0, "maxpool_bwd: bad arguments");
  const int64_t items = (int64_t)planes * H * ((W + 3) / 4);
  DASAC_REQUIRE(items < (1ll << 31) && k >= 1 && s >= 1, "maxpool_bwd: tensor too large");
  const dim3 grid(stream_grid(items, 256));
  hipStream_t st = as_stream(stream);
  const int Hp = (H + 1) / 2, Wq = (W + 3) / 4;
  if (k == 3 && s == 2 && pad == 1 && OH >= 1 && OW >= 1) {
    const int items2 = planes * Hp * Wq;
    hipLaunchKernelGGL(maxpool_bwd_3s2p1, dim3(stream_grid(items2, 256)), dim3(256), 0, st, dy, argmax, H, W, OH, OW, relu_mask, dx,
                       items2, Hp, Wq, fast_div(Wq), fast_div(Hp));
  } else if (k == 3 && s == 2) hipLaunchKernelGGL((maxpool_bwd<3, 2>), grid, dim3(256), 0, st, dy, y, argmax, H, W, OH, OW, k, s, pad, relu_mask, dx, (int)items);
  else if (k == 2 && s == 2) hipLaunchKernelGGL((maxpool_bwd<2, 2>), grid, dim3(256), 0, st, dy, y, argmax, H, W, OH, OW, k, s, pad, relu_mask, dx, (int)items);
  else if (k == 3 && s == 1) hipLaunchKernelGGL((maxpool_bwd<3, 1>), grid, dim3(256), 0, st, dy, y, argmax, H, W, OH, OW, k, s, pad, relu_mask, dx, (int)items);
  else hipLaunchKernelGGL((maxpool_bwd<0, 0>), grid, dim3(256), 0, st, dy, y, argmax, H, W, OH, OW, k, s, pad, relu_mask, dx, (int)items);
  DASAC_CHECK_LAUNCH("maxpool_bwd");
  return DASAC_OK;
}

extern "C" int dasac_ema_chunk_elems(void) { return kEmaChunk; }

// pairs: device array of {fast*, slow*, int64 n} (24 bytes each); chunks: device array of int32 pairs
// (tensor id, chunk index within the tensor), SORTED by tensor id; sq: device [n_tensors + n_chunks] doubles (scratch: per-tensor
// sums, then one partial per chunk -- summed in a fixed order, so teacher_diff is bit-identical from run to run); out: device [1].
extern "C" int dasac_ema_update(const void* pairs, int n_tensors, const int32_t* chunks, int n_chunks, float momentum,
                                int update, double* sq, float* out, dasac_stream_t stream) {
  DASAC_REQUIRE(pairs && chunks && sq && out && n_tensors > 0 && n_chunks > 0, "ema_update: bad arguments");
  hipStream_t s = as_stream(stream);
  double* sq_chunk = sq + n_tensors;
  // sac.py:95 multiplies by the python double (1. - momentum) cast to fp32
  const float one_minus = (float)(1.0 - (double)momentum);
  hipLaunchKernelGGL(ema_chunks, dim3(n_chunks), dim3(256), 0, s, reinterpret_cast<const TensorPair*>(pairs),
                     reinterpret_cast<const int2*>(chunks), momentum, one_minus, update, sq_chunk);
  DASAC_CHECK_LAUNCH("ema_chunks");
  hipLaunchKernelGGL(ema_tensor_sums, dim3(n_tensors), dim3(64), 0, s, reinterpret_cast<const int2*>(chunks), n_chunks, sq_chunk, sq);
  DASAC_CHECK_LAUNCH("ema_tensor_sums");
  hipLaunchKernelGGL(ema_finish, dim3(1), dim3(64), 0, s, sq, n_tensors, out);
  DASAC_CHECK_LAUNCH("ema_finish");
  return DASAC_OK;
}

extern "C" int dasac_sgd_step(const void* tensors, int n_tensors, const int32_t* chunks, int n_chunks, const float* group_lr,
                              const float* group_wd, int n_groups, float momentum, int first, dasac_stream_t stream) {
  DASAC_REQUIRE(tensors && chunks && group_lr && group_wd && n_tensors > 0 && n_chunks > 0, "sgd_step: bad arguments");
  DASAC_REQUIRE(n_groups >= 1 && n_groups <= 8, "sgd_step: 1..8 parameter groups");
  SgdGroups hp;
  for (int i = 0; i < 8; ++i) {
    hp.lr[i] = i < n_groups ? group_lr[i] : 0.f;
    hp.wd[i] = i < n_groups ? group_wd[i] : 0.f;
  }
  hipLaunchKernelGGL(sgd_chunks, dim3(n_chunks), dim3(256), 0, as_stream(stream), reinterpret_cast<const SgdTensor*>(tensors),
                     reinterpret_cast<const int2*>(chunks), hp, momentum, first);
  DASAC_CHECK_LAUNCH("sgd_chunks");
  return DASAC_OK;
}

extern "C" int dasac_scale_planes(const float* x, const float* plane_scale, int64_t planes, int64_t HW, float* y,
                                  dasac_stream_t stream) {
  DASAC_REQUIRE(x && plane_scale && y && planes > 0 && HW > 0, "scale_planes: bad arguments");
  const int64_t total = planes * HW;
  hipLaunchKernelGGL(scale_planes, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), x, plane_scale, (int)HW, y, total);
  DASAC_CHECK_LAUNCH("scale_planes");
  return DASAC_OK;
}

extern "C" int dasac_add(const float* a, const float* b, float* out, int64_t n, dasac_stream_t stream) {
  DASAC_REQUIRE(a && b && out && n > 0, "add: bad arguments");
  hipLaunchKernelGGL(add2, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), a, b, out, n);
  DASAC_CHECK_LAUNCH("add");
  return DASAC_OK;
}

extern "C" int dasac_relu_mask(const float* dy, const float* y, float* out, int64_t n, dasac_stream_t stream) {
  DASAC_REQUIRE(dy && y && out && n > 0, "relu_mask: bad arguments");
  hipLaunchKernelGGL(relu_mask, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), dy, y, out, n);
  DASAC_CHECK_LAUNCH("relu_mask");
  return DASAC_OK;
}

// ------------------------------------------------------------------------------------------------
// Train-mode BatchNorm (baseline / AdaBN mode: models/__init__.py:29 leaves BN un-frozen, every
// nn.SyncBatchNorm of deeplabv2.py:15 / fcn.py:8 normalises with batch statistics; train.py:281-289
// re-estimates the running statistics on target crops).  Cross-rank statistics (SyncBN) are summed by
// the caller with one RCCL all-reduce of the raw (sum, sum of squares) / (sum dy, sum dy*xhat) vectors.
// ------------------------------------------------------------------------------------------------
namespace dasac {

constexpr int kBnChunk = 256 * 16;

// Two-stage, atomic-free (deterministic) per-channel reductions: stage 1 writes one (s, q) pair of doubles per (plane, chunk),
// stage 2 (bn_sums_finish, one wave per channel) adds the N*chunks pairs of a channel in a fixed order.
// partial[(plane*chunks + chunk)*2 + {0,1}] = sum z, sum z^2 of the chunk        (grid = (chunks, N*C planes))
__global__ __launch_bounds__(256) void bn_stats(const float* __restrict__ z, int C, int HW, double* __restrict__ partial) {
  const int plane = blockIdx.y;
  const float* p = z + (size_t)plane * HW;
  double s = 0, q = 0;
  const int hi = min(HW, (int)(blockIdx.x + 1) * kBnChunk);
  for (int i = blockIdx.x * kBnChunk + threadIdx.x * 4; i < hi; i += 256 * 4) {      // 16-byte loads (4-byte aligned planes)
    if (i + 4 <= hi) {
      const f32x4u v4 = *reinterpret_cast<const f32x4u*>(p + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double v = v4[e];
        s += v;
        q += v * v;
      }
    } else {
      for (int e = i; e < hi; ++e) {
        const double v = p[e];
        s += v;
        q += v * v;
      }
    }
  }
  __shared__ double rs[4], rq[4];
  s = wave_sum(s);
  q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) {
    rs[threadIdx.x >> 6] = s;
    rq[threadIdx.x >> 6] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = partial + ((size_t)plane * gridDim.x + blockIdx.x) * 2;
    o[0] = (rs[0] + rs[1]) + (rs[2] + rs[3]);
    o[1] = (rq[0] + rq[1]) + (rq[2] + rq[3]);
  }
}

// sums[c] = sum over (n, chunk) of the s-partials, sums[C + c] = of the q-partials; grid C, one wave each
__global__ __launch_bounds__(64) void bn_sums_finish(const double* __restrict__ partial, int N, int C, int chunks,
                                                     double* __restrict__ sums, float* __restrict__ out0,
                                                     float* __restrict__ out1) {
  const int c = blockIdx.x;
  double s = 0, q = 0;
  for (int j = threadIdx.x; j < N * chunks; j += 64) {
    const int n = j / chunks, k = j - n * chunks;
    const double* o = partial + ((size_t)(n * C + c) * chunks + k) * 2;
    s += o[0];
    q += o[1];
  }
  s = wave_sum(s);
  q = wave_sum(q);
  if (threadIdx.x == 0) {
    sums[c] = s;
    sums[C + c] = q;
    if (out0) out0[c] = (float)s;       // backward: d beta = sum dy, d gamma = sum dy*xhat of THIS rank (no extra launch)
    if (out1) out1[c] = (float)q;
  }
}

// per-tile channel statistics left by dasac_conv_gemm_stats -> sums[c], sums[C + c] (doubles): a thread per channel adds the
// tiles in ascending order (coalesced across the channels of a block; fixed order = deterministic)
__global__ __launch_bounds__(64) void bn_tile_stats_reduce(const float* __restrict__ ts, int n_tiles, int C, int Mpad,
                                                           double* __restrict__ sums) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  double s = 0, q = 0;
#pragma unroll 4
  for (int i = 0; i < n_tiles; ++i) {
    s += (double)ts[(size_t)(2 * i) * Mpad + c];
    q += (double)ts[(size_t)(2 * i + 1) * Mpad + c];
  }
  sums[c] = s;
  sums[C + c] = q;
}

// batch mean / biased var -> scale, shift, mean, invstd; running stats: momentum update with the unbiased var
// `sums` null: the sums come from `tile_stats` (dasac_conv_gemm_stats), added here in tile order -- one launch per BN layer
__global__ void bn_train_finalize(const double* __restrict__ sums, const float* __restrict__ tile_stats, int n_tiles, int Mpad,
                                  double count_host, const double* __restrict__ count_dev,
                                  const float* __restrict__ gamma,
                                  const float* __restrict__ beta, float* __restrict__ running_mean,
                                  float* __restrict__ running_var, int64_t* __restrict__ num_batches_tracked, float momentum,
                                  float eps, int C,
                                  float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
                                  float* __restrict__ invstd_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && num_batches_tracked) num_batches_tracked[0] += 1;      // nn.BatchNorm bookkeeping, no extra launch
  if (c >= C) return;
  const double count = count_dev ? count_dev[0] : count_host;   // SyncBN: the all-reduced count stays on the device
  double s1, s2;
  if (sums) {
    s1 = sums[c];
    s2 = sums[C + c];
  } else {
    s1 = s2 = 0;
#pragma unroll 4
    for (int i = 0; i < n_tiles; ++i) {
      s1 += (double)tile_stats[(size_t)(2 * i) * Mpad + c];
      s2 += (double)tile_stats[(size_t)(2 * i + 1) * Mpad + c];
    }
  }
  const double m = s1 / count;
  double var = s2 / count - m * m;
  if (var < 0) var = 0;
  const float meanf = (float)m, varf = (float)var;
  const float is = 1.f / sqrtf(varf + eps);
  const float a = gamma[c] * is;
  scale[c] = a;
  shift[c] = beta[c] - meanf * a;
  mean_out[c] = meanf;
  invstd_out[c] = is;
  if (running_mean) {
    const float unbiased = count > 1 ? (float)(var * count / (count - 1)) : varf;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// y = relu?(z*scale[c] + shift[c] (+res))
__global__ __launch_bounds__(256) void bn_apply(const float* __restrict__ z, const float* __restrict__ scale,
                                                const float* __restrict__ shift, const float* __restrict__ res, int relu,
                                                int C, int HW, float* __restrict__ y, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)((i / HW) % C);
    float v = z[i] * scale[c] + shift[c];
    if (res) v += res[i];
    if (relu) v = fmaxf(v, 0.f);
    y[i] = v;
  }
}

// One rank (no all-reduce between statistics and normalisation): finalize + apply in ONE launch.  A block owns a 16K-element chunk
// of one (n, c) plane; it first adds its channel's tile statistics (dasac_conv_gemm_stats) -- 256 threads stride over the tiles,
// wave butterflies, the four wave sums added in a fixed order: every block of a channel gets the same bits -- derives scale /
// shift exactly as bn_train_finalize does, normalises its chunk with 16-byte accesses; the block of (n = 0, chunk 0) also writes
// mean / invstd (for the backward pass) and moves the running statistics.  Saves a 5 us launch per BN layer and pass.
constexpr int kBnBig = 256 * 64;
__global__ __launch_bounds__(256) void bn_apply_tiles(const float* __restrict__ z, const float* __restrict__ ts, int n_tiles, int Mpad,
                                                      double count, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ running_mean, float* __restrict__ running_var,
                                                      int64_t* __restrict__ num_batches_tracked, float momentum, float eps,
                                                      const float* __restrict__ res, int relu, int C, int HW,
                                                      float* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ invstd_out) {
  const int plane = blockIdx.y, c = plane % C, n = plane / C;
  double s1 = 0, s2 = 0;
  for (int i = threadIdx.x; i < n_tiles; i += 256) {
    s1 += (double)ts[(size_t)(2 * i) * Mpad + c];
    s2 += (double)ts[(size_t)(2 * i + 1) * Mpad + c];
  }
  __shared__ double r1[4], r2[4];
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) {
    r1[threadIdx.x >> 6] = s1;
    r2[threadIdx.x >> 6] = s2;
  }
  __syncthreads();
  s1 = (r1[0] + r1[1]) + (r1[2] + r1[3]);
  s2 = (r2[0] + r2[1]) + (r2[2] + r2[3]);
  const double m = s1 / count;
  double var = s2 / count - m * m;
  if (var < 0) var = 0;
  const float meanf = (float)m, varf = (float)var;
  const float is = 1.f / sqrtf(varf + eps);
  const float a = gamma[c] * is;
  const float sh = beta[c] - meanf * a;
  if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    mean_out[c] = meanf;
    invstd_out[c] = is;
    if (running_mean) {
      const float unbiased = count > 1 ? (float)(var * count / (count - 1)) : varf;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
    if (c == 0 && num_batches_tracked) num_batches_tracked[0] += 1;
  }
  const size_t pb = (size_t)plane * HW;
  const int lo = blockIdx.x * kBnBig, hi = min(HW, lo + kBnBig);
  for (int i = lo + threadIdx.x * 4; i < hi; i += 256 * 4) {
    if (i + 4 <= hi) {
      f32x4u v = *reinterpret_cast<const f32x4u*>(z + pb + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] * a + sh;
      if (res) {
        const f32x4u rv = *reinterpret_cast<const f32x4u*>(res + pb + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += rv[e];
      }
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      *reinterpret_cast<f32x4u*>(y + pb + i) = v;
    } else {
      for (int e = i; e < hi; ++e) {
        float v = z[pb + e] * a + sh;
        if (res) v += res[pb + e];
        if (relu) v = fmaxf(v, 0.f);
        y[pb + e] = v;
      }
    }
  }
}

// The backward twin: a block adds its channel's (plane, chunk) partial pairs of bn_bwd_reduce itself (N * chunks of them, fixed
// order, the same in every block), forms dz for a 16K-element chunk of one plane, and the block of (n = 0, chunk 0) writes
// d gamma / d beta -- no separate finish / parameter launches on one rank.
__global__ __launch_bounds__(256) void bn_bwd_apply_partials(const float* __restrict__ dy, const float* __restrict__ z,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const double* __restrict__ partial,
                                                             int N, int chunks_r, double count, int C, int HW,
                                                             float* __restrict__ dz, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta) {
  const int plane = blockIdx.y, c = plane % C, n = plane / C;
  // the channel's N * chunks partial pairs: strided over the block (independent loads) and folded by a fixed tree -- the same
  // order in every block and every run; round 4 had every thread walk all of them serially (75-300 dependent loads per block)
  __shared__ double red_s[256], red_q[256];
  double s = 0, q = 0;
  for (int j = threadIdx.x; j < N * chunks_r; j += 256) {
    const int nn = j / chunks_r, k = j - nn * chunks_r;
    const double* o = partial + ((size_t)(nn * C + c) * chunks_r + k) * 2;
    s += o[0];
    q += o[1];
  }
  red_s[threadIdx.x] = s;
  red_q[threadIdx.x] = q;
  __syncthreads();
#pragma unroll
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red_s[threadIdx.x] += red_s[threadIdx.x + o];
      red_q[threadIdx.x] += red_q[threadIdx.x + o];
    }
    __syncthreads();
  }
  s = red_s[0];
  q = red_q[0];
  if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)q;
  }
  const float is = invstd[c], mu = mean[c];
  const float a = (float)(s / count), b = (float)(q / count), gi = gamma[c] * is;
  const size_t pb = (size_t)plane * HW;
  const int lo = blockIdx.x * kBnBig, hi = min(HW, lo + kBnBig);
  for (int i = lo + threadIdx.x * 4; i < hi; i += 256 * 4) {
    if (i + 4 <= hi) {
      const f32x4u d = *reinterpret_cast<const f32x4u*>(dy + pb + i), zz = *reinterpret_cast<const f32x4u*>(z + pb + i);
      f32x4u o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (zz[e] - mu) * is;
        o[e] = gi * (d[e] - a - xh * b);
      }
      *reinterpret_cast<f32x4u*>(dz + pb + i) = o;
    } else {
      for (int e = i; e < hi; ++e) {
        const float xh = (z[pb + e] - mu) * is;
        dz[pb + e] = gi * (dy[pb + e] - a - xh * b);
      }
    }
  }
}

// partial pairs (as bn_stats) of sum dy, sum dy * xhat,  xhat = (z - mean)*invstd
__global__ __launch_bounds__(256) void bn_bwd_reduce(const float* __restrict__ dy, const float* __restrict__ z,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd, int C,
                                                     int HW, double* __restrict__ partial) {
  const int plane = blockIdx.y, c = plane % C;
  const float* pd = dy + (size_t)plane * HW;
  const float* pz = z + (size_t)plane * HW;
  const float m = mean[c], is = invstd[c];
  double s = 0, q = 0;
  const int hi = min(HW, (int)(blockIdx.x + 1) * kBnChunk);
  for (int i = blockIdx.x * kBnChunk + threadIdx.x * 4; i < hi; i += 256 * 4) {      // 16-byte loads (4-byte aligned planes)
    if (i + 4 <= hi) {
      const f32x4u d4 = *reinterpret_cast<const f32x4u*>(pd + i), z4 = *reinterpret_cast<const f32x4u*>(pz + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s += d4[e];
        q += (double)d4[e] * (double)((z4[e] - m) * is);
      }
    } else {
      for (int e = i; e < hi; ++e) {
        const float d = pd[e];
        s += d;
        q += (double)d * (double)((pz[e] - m) * is);
      }
    }
  }
  __shared__ double rs[4], rq[4];
  s = wave_sum(s);
  q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) {
    rs[threadIdx.x >> 6] = s;
    rq[threadIdx.x >> 6] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = partial + ((size_t)plane * gridDim.x + blockIdx.x) * 2;
    o[0] = (rs[0] + rs[1]) + (rs[2] + rs[3]);
    o[1] = (rq[0] + rq[1]) + (rq[2] + rq[3]);
  }
}

// dz = gamma*invstd * (dy - sum_dy/n - xhat * sum_dy_xhat/n);  dgamma = sum_dy_xhat, dbeta = sum_dy
__global__ __launch_bounds__(256) void bn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ z,
                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                    const float* __restrict__ gamma, const double* __restrict__ sums,
                                                    double count_host, const double* __restrict__ count_dev, int C, int HW,
                                                    float* __restrict__ dz, int64_t total) {
  const double count = count_dev ? count_dev[0] : count_host;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)((i / HW) % C);
    const float is = invstd[c];
    const float xh = (z[i] - mean[c]) * is;
    const float a = (float)(sums[c] / count), b = (float)(sums[C + c] / count);
    dz[i] = gamma[c] * is * (dy[i] - a - xh * b);
  }
}

__global__ void bn_bwd_params(const double* __restrict__ sums, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (dbeta) dbeta[c] = (float)sums[c];
  if (dgamma) dgamma[c] = (float)sums[C + c];
}

}  // namespace dasac

extern "C" size_t dasac_bn_stats_workspace(int N, int C, int64_t HW) {
  return (size_t)N * C * (size_t)((HW + kBnChunk - 1) / kBnChunk) * 2 * sizeof(double);
}

extern "C" int dasac_bn_stats(const float* z, int N, int C, int64_t HW, double* sums, void* workspace, size_t ws_bytes,
                              dasac_stream_t stream) {
  DASAC_REQUIRE(z && sums && workspace && N > 0 && C > 0 && HW > 0 && HW < (1ll << 31), "bn_stats: bad arguments");
  if (ws_bytes < dasac_bn_stats_workspace(N, C, HW)) return fail(DASAC_EWORKSPACE, "bn_stats: workspace too small");
  hipStream_t s = as_stream(stream);
  const int chunks = (int)((HW + kBnChunk - 1) / kBnChunk);
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(bn_stats, dim3((unsigned)chunks, N * C), dim3(256), 0, s, z, C, (int)HW, partial);
  DASAC_CHECK_LAUNCH("bn_stats");
  hipLaunchKernelGGL(bn_sums_finish, dim3(C), dim3(64), 0, s, partial, N, C, chunks, sums, (float*)nullptr, (float*)nullptr);
  DASAC_CHECK_LAUNCH("bn_sums_finish");
  return DASAC_OK;
}

static int bn_finalize_launch(const double* sums, const float* tile_stats, int n_tiles, int mpad, double count,
                              const double* count_dev, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, int64_t* num_batches_tracked, float momentum, float eps, int C, float* scale,
                              float* shift, float* mean, float* invstd, dasac_stream_t stream) {
  DASAC_REQUIRE((sums || (tile_stats && n_tiles > 0 && mpad >= C)) && gamma && beta && scale && shift && mean && invstd && C > 0 &&
                    (count > 0 || count_dev),
                "bn_train_finalize: bad arguments");
  hipLaunchKernelGGL(bn_train_finalize, dim3((C + 63) / 64), dim3(64), 0, as_stream(stream), sums, tile_stats, n_tiles, mpad, count,
                     count_dev, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, C, scale, shift, mean, invstd);
  DASAC_CHECK_LAUNCH("bn_train_finalize");
  return DASAC_OK;
}

extern "C" int dasac_bn_train_finalize(const double* sums, double count, const double* count_dev, const float* gamma,
                                       const float* beta, float* running_mean, float* running_var,
                                       int64_t* num_batches_tracked, float momentum, float eps, int C, float* scale,
                                       float* shift, float* mean, float* invstd, dasac_stream_t stream) {
  DASAC_REQUIRE(sums, "bn_train_finalize: null sums");
  return bn_finalize_launch(sums, nullptr, 0, 0, count, count_dev, gamma, beta, running_mean, running_var, num_batches_tracked,
                            momentum, eps, C, scale, shift, mean, invstd, stream);
}

extern "C" int dasac_bn_train_finalize_tiles(const float* tile_stats, int n_tiles, int mpad, double count, const float* gamma,
                                             const float* beta, float* running_mean, float* running_var,
                                             int64_t* num_batches_tracked, float momentum, float eps, int C, float* scale,
                                             float* shift, float* mean, float* invstd, dasac_stream_t stream) {
  DASAC_REQUIRE(tile_stats, "bn_train_finalize_tiles: null statistics");
  return bn_finalize_launch(nullptr, tile_stats, n_tiles, mpad, count, nullptr, gamma, beta, running_mean, running_var,
                            num_batches_tracked, momentum, eps, C, scale, shift, mean, invstd, stream);
}

extern "C" int dasac_bn_tile_stats_reduce(const float* tile_stats, int n_tiles, int C, int mpad, double* sums,
                                          dasac_stream_t stream) {
  DASAC_REQUIRE(tile_stats && sums && n_tiles > 0 && C > 0 && mpad >= C, "bn_tile_stats_reduce: bad arguments");
  hipLaunchKernelGGL(bn_tile_stats_reduce, dim3((C + 63) / 64), dim3(64), 0, as_stream(stream), tile_stats, n_tiles, C, mpad, sums);
  DASAC_CHECK_LAUNCH("bn_tile_stats_reduce");
  return DASAC_OK;
}

extern "C" int dasac_bn_apply(const float* z, const float* scale, const float* shift, const float* res, int relu, int N,
                              int C, int64_t HW, float* y, dasac_stream_t stream) {
  DASAC_REQUIRE(z && scale && shift && y && N > 0 && C > 0 && HW > 0, "bn_apply: bad arguments");
  const int64_t total = (int64_t)N * C * HW;
  hipLaunchKernelGGL(bn_apply, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), z, scale, shift, res, relu, C,
                     (int)HW, y, total);
  DASAC_CHECK_LAUNCH("bn_apply");
  return DASAC_OK;
}

extern "C" int dasac_bn_bwd_reduce(const float* dy, const float* z, const float* mean, const float* invstd, int N, int C,
                                   int64_t HW, double* sums, float* dgamma, float* dbeta, void* workspace, size_t ws_bytes,
                                   dasac_stream_t stream) {
  DASAC_REQUIRE(dy && z && mean && invstd && sums && workspace && N > 0 && C > 0 && HW > 0 && HW < (1ll << 31),
                "bn_bwd_reduce: bad arguments");
  if (ws_bytes < dasac_bn_stats_workspace(N, C, HW)) return fail(DASAC_EWORKSPACE, "bn_bwd_reduce: workspace too small");
  hipStream_t s = as_stream(stream);
  const int chunks = (int)((HW + kBnChunk - 1) / kBnChunk);
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(bn_bwd_reduce, dim3((unsigned)chunks, N * C), dim3(256), 0, s, dy, z, mean, invstd, C, (int)HW, partial);
  DASAC_CHECK_LAUNCH("bn_bwd_reduce");
  hipLaunchKernelGGL(bn_sums_finish, dim3(C), dim3(64), 0, s, partial, N, C, chunks, sums, dbeta, dgamma);
  DASAC_CHECK_LAUNCH("bn_sums_finish");
  return DASAC_OK;
}

extern "C" int dasac_bn_train_apply_tiles(const float* z, const float* tile_stats, int n_tiles, int mpad, double count,
                                          const float* gamma, const float* beta, float* running_mean, float* running_var,
                                          int64_t* num_batches_tracked, float momentum, float eps, const float* res, int relu,
                                          int N, int C, int64_t HW, float* y, float* mean, float* invstd, dasac_stream_t stream) {
  DASAC_REQUIRE(z && tile_stats && gamma && beta && y && mean && invstd && n_tiles > 0 && mpad >= C && count > 0 && N > 0 && C > 0 &&
                    HW > 0 && HW < (1ll << 31) && (int64_t)N * C < 65536,
                "bn_train_apply_tiles: bad arguments");
  hipLaunchKernelGGL(bn_apply_tiles, dim3((unsigned)((HW + kBnBig - 1) / kBnBig), N * C), dim3(256), 0, as_stream(stream), z, tile_stats,
                     n_tiles, mpad, count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, res, relu, C,
                     (int)HW, y, mean, invstd);
  DASAC_CHECK_LAUNCH("bn_apply_tiles");
  return DASAC_OK;
}

extern "C" int dasac_bn_bwd_fused(const float* dy, const float* z, const float* mean, const float* invstd, const float* gamma,
                                  double count, int N, int C, int64_t HW, float* dz, float* dgamma, float* dbeta, void* workspace,
                                  size_t ws_bytes, dasac_stream_t stream) {
  DASAC_REQUIRE(dy && z && mean && invstd && gamma && dz && workspace && count > 0 && N > 0 && C > 0 && HW > 0 && HW < (1ll << 31) &&
                    (int64_t)N * C < 65536,
                "bn_bwd_fused: bad arguments");
  if (ws_bytes < dasac_bn_stats_workspace(N, C, HW)) return fail(DASAC_EWORKSPACE, "bn_bwd_fused: workspace too small");
  hipStream_t s = as_stream(stream);
  const int chunks = (int)((HW + kBnChunk - 1) / kBnChunk);
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(bn_bwd_reduce, dim3((unsigned)chunks, N * C), dim3(256), 0, s, dy, z, mean, invstd, C, (int)HW, partial);
  DASAC_CHECK_LAUNCH("bn_bwd_reduce");
  hipLaunchKernelGGL(bn_bwd_apply_partials, dim3((unsigned)((HW + kBnBig - 1) / kBnBig), N * C), dim3(256), 0, s, dy, z, mean, invstd,
                     gamma, partial, N, chunks, count, C, (int)HW, dz, dgamma, dbeta);
  DASAC_CHECK_LAUNCH("bn_bwd_apply_partials");
  return DASAC_OK;
}

extern "C" int dasac_bn_bwd_apply(const float* dy, const float* z, const float* mean, const float* invstd,
                                  const float* gamma, const double* sums, double count, const double* count_dev, int N, int C,
                                  int64_t HW, float* dz, float* dgamma, float* dbeta, dasac_stream_t stream) {
  DASAC_REQUIRE(dy && z && mean && invstd && gamma && sums && dz && (count > 0 || count_dev), "bn_bwd_apply: bad arguments");
  const int64_t total = (int64_t)N * C * HW;
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(bn_bwd_apply, dim3(stream_grid(total, 256)), dim3(256), 0, s, dy, z, mean, invstd, gamma, sums, count,
                     count_dev, C, (int)HW, dz, total);
  DASAC_CHECK_LAUNCH("bn_bwd_apply");
  if (dgamma || dbeta) {
    hipLaunchKernelGGL(bn_bwd_params, dim3((C + 255) / 256), dim3(256), 0, s, sums, C, dgamma, dbeta);
    DASAC_CHECK_LAUNCH("bn_bwd_params");
  }
  return DASAC_OK;
}

// ------------------------------------------------------------------------------------------------
// Validation counts: utils/metrics.py:9-53 `Jaccard.add_sample` + the argmax of
// train.py:339-469 -- per-class true-positive / false-positive / false-negative pixel counts from the
// upsampled logits in ONE pass (the reference loops over the 19 classes with .item() syncs per batch).
// ------------------------------------------------------------------------------------------------
namespace dasac {

__global__ __launch_bounds__(256) void iou_counts(const float* __restrict__ logits, const int64_t* __restrict__ gt, int C,
                                                  int64_t HW, int64_t total, int ignore_index,
                                                  unsigned long long* __restrict__ counts) {
  __shared__ unsigned int s[3 * 64];
  for (int i = threadIdx.x; i < 3 * 64; i += 256) s[i] = 0;
  __syncthreads();
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (int64_t)gridDim.x * 256) {
    const int64_t b = p / HW, r = p - b * HW;
    const int64_t y = gt[p];
    if (y == ignore_index) continue;        // utils/metrics.py:28-30: only ignore_index pixels are dropped
    const float* lp = logits + b * C * HW + r;
    float best = lp[0];
    int k = 0;
    for (int c = 1; c < C; ++c) {
      const float v = lp[(int64_t)c * HW];
      if (v > best) {
        best = v;
        k = c;
      }
    }
    if (y == k) {
      atomicAdd(&s[k], 1u);                 // tp
    } else {
      atomicAdd(&s[64 + k], 1u);            // fp of the predicted class (also when gt is no class at all, e.g. -1)
      if (y >= 0 && y < C) atomicAdd(&s[128 + (int)y], 1u);      // fn of the true class
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * 64; i += 256)
    if (s[i] && (i & 63) < C) atomicAdd(&counts[(i >> 6) * C + (i & 63)], (unsigned long long)s[i]);
}

}  // namespace dasac

extern "C" int dasac_iou_counts(const float* logits, const int64_t* gt, int B, int C, int64_t HW, int ignore_index,
                                int64_t* counts, dasac_stream_t stream) {
  DASAC_REQUIRE(logits && gt && counts && B > 0 && C > 0 && C <= 64 && HW > 0, "iou_counts: bad arguments");
  const int64_t total = (int64_t)B * HW;
  hipLaunchKernelGGL(iou_counts, dim3(stream_grid(total, 256, kNumCu * 8)), dim3(256), 0, as_stream(stream), logits, gt, C, HW, total,
                     ignore_index, reinterpret_cast<unsigned long long*>(counts));
  DASAC_CHECK_LAUNCH("iou_counts");
  return DASAC_OK;
}

// ------------------------------------------------------------------------------------------------
// Label preparation and Dropout2d masks (what was left on ATen inside the step)
// ------------------------------------------------------------------------------------------------
namespace dasac {

// models/sac.py:337-338: ignore_mask = (y == -1); y[ignore_mask] = 255  -- one pass, in place
__global__ __launch_bounds__(256) void label_pad_mask(int64_t* __restrict__ y, uint8_t* __restrict__ mask, int64_t n,
                                                      int64_t pad_label, int64_t ignore_label) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const bool pad = y[i] == pad_label;
    mask[i] = pad ? 1 : 0;
    if (pad) y[i] = ignore_label;
  }
}

// Philox4x32-10 (Salmon et al., SC'11): counter (i, stream offset), key = seed
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * ctr.x, p1 = (uint64_t)0xCD9E8D57u * ctr.z;
    ctr = make_uint4((uint32_t)(p1 >> 32) ^ ctr.y ^ key.x, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ ctr.w ^ key.y, (uint32_t)p0);
    key.x += 0x9E3779B9u;
    key.y += 0xBB67AE85u;
  }
  return ctr;
}

// Dropout2d (fcn.py:52,56): per (n, c) plane  keep/(1-p) with keep ~ Bernoulli(1-p)
__global__ __launch_bounds__(256) void dropout_planes(uint64_t seed, uint64_t offset, float p, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint4 r = philox4x32_10(make_uint4((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)),
                                make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
  const float u = (float)(r.x >> 8) * (1.0f / 16777216.0f);      // uniform on [0, 1), 24 bits
  out[i] = u >= p ? 1.0f / (1.0f - p) : 0.0f;
}

}  // namespace dasac

extern "C" int dasac_label_pad_mask(int64_t* labels, uint8_t* mask, int64_t n, int pad_label, int ignore_label,
                                    dasac_stream_t stream) {
  DASAC_REQUIRE(labels && mask && n > 0, "label_pad_mask: bad arguments");
  hipLaunchKernelGGL(label_pad_mask, dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream), labels, mask, n,
                     (int64_t)pad_label, (int64_t)ignore_label);
  DASAC_CHECK_LAUNCH("label_pad_mask");
  return DASAC_OK;
}

extern "C" int dasac_dropout_planes(uint64_t seed, uint64_t offset, float p, int64_t planes, float* keep_scale,
                                    dasac_stream_t stream) {
  DASAC_REQUIRE(keep_scale && planes > 0 && p >= 0.f && p < 1.f, "dropout_planes: bad arguments");
  hipLaunchKernelGGL(dropout_planes, dim3((unsigned)((planes + 255) / 256)), dim3(256), 0, as_stream(stream), seed, offset, p,
                     planes, keep_scale);
  DASAC_CHECK_LAUNCH("dropout_planes");
  return DASAC_OK;
}
