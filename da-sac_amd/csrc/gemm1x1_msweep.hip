// "M-sweep" GEMM for the 1x1 stride-1 convolutions with a short contraction (K = 64 / 128 / 256 input channels): the bottleneck
// expansions conv3 + bn3 + residual + ReLU (models/deeplabv2.py:70-71, 91-97) and the data gradient of conv1 (:59), whose
// epilogue accumulates into the producer's gradient and applies its ReLU mask.  exact fp32, v_mfma_f32_32x32x2_f32.
//
// Why a second kernel (profiles/r5_epilogue_anatomy.txt): in the tile-per-block conv_gemm these layers spend a quarter of every
// workgroup's life in an epilogue whose few hundred instructions issue at a crawl next to three K-loop waves, and the K loop
// itself -- 16 K-steps, each a global load -> LDS -> barrier -> MFMA chain, on an activation tile that the 8 M tiles of a pixel
// tile each gather again -- reaches 107 TFLOP/s even with the epilogue removed.  Here
//   * ONE persistent 512-thread workgroup per CU owns a run of 64-pixel tiles.  The activation tile [K x 64 px] is staged in LDS
//     ONCE (k-interleaved 16-byte words, double buffered: the next tile is fetched while this one is swept) and all M rows are
//     swept over it: 8 waves x (32*TM rows x 64 px) per pass, M / (256*TM) passes per tile;
//   * the packed weights (dasac_conv_pack's [(k/4)][Mpad][4] layout = MFMA operand order; <= 1 MB per layer, resident in every
//     XCD's L2) stream from L2 STRAIGHT INTO REGISTERS, two K-steps ahead: the K loop has no LDS write, no barrier and no address
//     arithmetic -- 4 buffer_load_dwordx4 + 4 ds_read_b128 (immediate offsets) per 32 MFMAs;
//   * the two waves that share a SIMD take turns on the matrix pipe (a lock per SIMD in LDS): while one runs its K loop alone at
//     the pipe's rate, the other runs its epilogue (residual loads, add / ReLU / mask, 64 stores per lane) and prefetches --
//     the epilogue overlaps matrix work by construction instead of by the luck of four independent workgroups;
//   * the activation tile is read from HBM once per pixel tile instead of once per M tile.
// Per accumulator the MFMA sequence is exactly conv_gemm's (k pairs {8g+e, 8g+4+e}, K-steps ascending) and the epilogue arithmetic
// is the same expression: outputs are BIT-IDENTICAL to the tile-per-block kernel (tests/test_gpu_conv.py holds them to torch.equal).
//
// Roofline: MFMA-bound, 2*M*Npix*K flop per launch against 157.3 TFLOP/s; algorithmic bytes = x + out (+ residual) + 1/32 mask.
#include "common.hpp"

#include <cstdlib>
#include <type_traits>

namespace dasac {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace msweep {

constexpr int kThreads = 512;            // 8 waves: two per SIMD
constexpr int kWaves = 8;
constexpr int kBN = 64;                  // pixels per tile
constexpr unsigned kPoison = 0x80000000u;
constexpr int kRsrcFlags = 0x00020000;   // raw buffer, 32-bit data format (gfx9 family word 3)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, kRsrcFlags);
}
__device__ __forceinline__ float buf_f32(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 buf_f32x4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
}
extern "C" __device__ int dasac_llvm_writelane_ms(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

// mask words of the two rows a ballot over accumulator register rg covers (see conv_gemm's put_mask_rows)
__device__ __forceinline__ int put_mask_rows(int rg, unsigned long long ballot, int acc) {
  const int lo = (int)(unsigned)ballot, hi = (int)(unsigned)(ballot >> 32);
  const int r = (rg & 3) + 8 * (rg >> 2);
  acc = dasac_llvm_writelane_ms(lo, r, acc);
  acc = dasac_llvm_writelane_ms(hi, r + 4, acc);
  return acc;
}

#ifdef DASAC_TRACE_TILES
// diagnostic build only (tools/ms_timeline.py): per wave the time spent waiting for a tile / waiting for the matrix pipe / in the K
// loop / in everything else, in s_memtime ticks
__device__ unsigned long long* g_ms_trace = nullptr;
#define MS_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define MS_ACC(slot, a, b) tr[slot] += (b) - (a)
#else
#define MS_T(var)
#define MS_ACC(slot, a, b)
#endif

struct Geom {
  int HW, CxHW, Npix;            // plane size, image stride of x, Nb*HW
  int M, Mpad;                   // output channels, padded row count of the packed weights
  int w32;                       // mask words per row
  int x_bytes, w_bytes, out_bytes;
  int n_tiles, passes;           // 64-pixel tiles, M / (256*TM)
};
struct Epi {
  const float* shift;
  const float* res;
  const unsigned* mbits;
  unsigned* obits;
  int relu;
};

// sync words in LDS: [0..1] tiles published per buffer (8 per tile), [2..5] matrix-pipe lock per SIMD
template <int K, int TM, int BITS>
__global__ __launch_bounds__(kThreads, 2) void gemm1x1_msweep(const float* __restrict__ X, const float* __restrict__ Wp,
                                                              float* __restrict__ Out, Geom g, Epi ep) {
  constexpr int KQ = K / 4;                 // k quads = 16-byte words per pixel
  constexpr int KT = K / 16;                // K-steps
  constexpr int UNITS = (16 * KQ + kThreads - 1) / kThreads;    // loader units (4 px x 4 channels) per thread and tile
#ifndef DASAC_MS_D
#define DASAC_MS_D 2
#endif
  constexpr int D = DASAC_MS_D < KT ? DASAC_MS_D : KT - 1;    // weight prefetch distance in K-steps
  static_assert(K % 16 == 0 && K <= 256, "activation tile: 2 x K x 64 x 4 bytes of LDS");
  static_assert(TM == 1 || TM == 2, "rows per wave and pass: 32 or 64");

  __shared__ f32x4 sB[2][KQ * kBN];         // [buffer][(k/4)][pixel] x {k%4}
  __shared__ int s_sync[8];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 31, lh = lane >> 5;
  // the SIMD this wave really runs on (HW_ID bits 5:4): the waves that share it -- normally w and w + 4 -- share a lock
  const int simd = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_s_getreg(((2 - 1) << 11) | (4 << 6) | 4));

  const __amdgpu_buffer_rsrc_t rx = make_rsrc(X, g.x_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(Wp, g.w_bytes);
  const __amdgpu_buffer_rsrc_t ro = make_rsrc(Out, g.out_bytes);
  const __amdgpu_buffer_rsrc_t rres = make_rsrc(ep.res ? ep.res : Out, g.out_bytes);
  const __amdgpu_buffer_rsrc_t rsh = make_rsrc(ep.shift ? ep.shift : Out, ep.shift ? g.M * 4 : 0);

  // ---- my run of (tile, pass) units: equal shares, contiguous (a workgroup streams a contiguous pixel range) ----
  const long long units = (long long)g.n_tiles * g.passes;
  const int u_lo = (int)(units * blockIdx.x / gridDim.x), u_hi = (int)(units * (blockIdx.x + 1) / gridDim.x);
  if (u_lo >= u_hi) return;
  const int tile_lo = u_lo / g.passes, tile_hi = (u_hi - 1) / g.passes;

  if (t < 8) s_sync[t] = 0;
  __syncthreads();
#ifdef DASAC_TRACE_TILES
  unsigned long long tr[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long tr_start = __builtin_amdgcn_s_memtime();
#endif

  // ---- activation tile loader: thread -> pixel quad pq (4 consecutive pixels) x channel quads cq0 + 32*u ----
  const int pq = t & 15, cq0 = t >> 4;
  int ld_p = tile_lo * kBN + 4 * pq;                        // first pixel of my quad (advanced by 64 per tile)
  int ld_n = ld_p / g.HW, ld_r = ld_p - ld_n * g.HW;        // image, position inside it
  f32x4 ld[UNITS][4];                                       // [unit][channel of the quad] x 4 pixels

  auto load_tile = [&]() __attribute__((always_inline)) {
    const bool whole = (ld_p + 3 < g.Npix) & (ld_r + 3 < g.HW);
    if (__builtin_amdgcn_ballot_w64(!whole) == 0) {
#pragma unroll
      for (int u = 0; u < UNITS; ++u) {
        const int cq = cq0 + 32 * u;
        const unsigned voff = (16 * KQ % kThreads == 0 || cq < KQ) ? (unsigned)(ld_n * g.CxHW + 4 * cq * g.HW + ld_r) * 4u : kPoison;
#pragma unroll
        for (int c = 0; c < 4; ++c) ld[u][c] = buf_f32x4(rx, voff, c * g.HW * 4);
      }
    } else {      // a quad of this wave leaves its image or the tensor: per element for the whole wave (once per image)
      unsigned voe[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int re = ld_r + e, ne = ld_n;
        while (re >= g.HW) { re -= g.HW; ++ne; }
        voe[e] = ld_p + e < g.Npix ? (unsigned)(ne * g.CxHW + re) * 4u : kPoison;
      }
#pragma unroll
      for (int u = 0; u < UNITS; ++u) {
        const int cq = cq0 + 32 * u;
        const bool cok = 16 * KQ % kThreads == 0 || cq < KQ;
        float tmp[4][4];                                   // all 16 loads of the unit in flight, then the four vectors
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            tmp[c][e] = buf_f32(rx, (cok && voe[e] != kPoison) ? voe[e] + (unsigned)(4 * cq * g.HW) * 4u : kPoison, c * g.HW * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) ld[u][c] = f32x4{tmp[c][0], tmp[c][1], tmp[c][2], tmp[c][3]};
      }
    }
  };
  auto store_tile = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
      const int cq = cq0 + 32 * u;
      if (16 * KQ % kThreads == 0 || cq < KQ) {
#pragma unroll
        for (int e = 0; e < 4; ++e) sB[buf][cq * kBN + 4 * pq + e] = f32x4{ld[u][0][e], ld[u][1][e], ld[u][2][e], ld[u][3][e]};
      }
    }
  };
  auto advance_loader = [&]() __attribute__((always_inline)) {
    ld_p += kBN;
    ld_r += kBN;
    while (ld_r >= g.HW) { ld_r -= g.HW; ++ld_n; }
  };
  auto publish = [&](int buf) __attribute__((always_inline)) {     // my share of the tile is in LDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(&s_sync[buf], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  load_tile();
  store_tile(0);
  publish(0);

  // ---- epilogue pixel columns of this lane (j = 0, 1): pixel tile*64 + j*32 + li; (image, position) advanced per tile ----
  int ep_n[2], ep_r[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = tile_lo * kBN + j * 32 + li;
    ep_n[j] = p / g.HW;
    ep_r[j] = p - ep_n[j] * g.HW;
  }

  // weights: per-lane part of the operand address is fixed for the whole kernel (lane half -> k quad, lane -> row)
  const unsigned voff_a = (unsigned)(lh * g.Mpad + li) * 16u;
  f32x4 a[D + 1][TM][2];                                    // [stage][row group][gq]
#define MS_LOAD_A(kt, mb)                                                                                         \
  {                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                \
      _Pragma("unroll") for (int gq = 0; gq < 2; ++gq)                                                            \
        a[(kt) % (D + 1)][i][gq] = buf_f32x4(rw, voff_a, ((4 * (kt) + 2 * gq) * g.Mpad + (mb) + i * 32) * 16);  \
  }

  int published[2] = {0, 0};                                // tiles this workgroup has put into each buffer so far
  int mb_next = ((u_lo - tile_lo * g.passes) * kWaves + wave) * 32 * TM;   // row block of my first pass
#pragma unroll
  for (int kt = 0; kt < D && kt < KT; ++kt) MS_LOAD_A(kt, mb_next);

  for (int tile = tile_lo; tile <= tile_hi; ++tile) {
    const int buf = (tile - tile_lo) & 1;
    published[buf] += kWaves;
    MS_T(t_r0);
    if (lane == 0)
      while (__hip_atomic_load(&s_sync[buf], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < published[buf]) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    MS_T(t_r1);
    MS_ACC(0, t_r0, t_r1);
    const bool has_next = tile < tile_hi;
    if (has_next) {
      advance_loader();
      load_tile();                                          // in flight during this tile's first K loop
    }
    const int p_lo = tile == tile_lo ? u_lo - tile * g.passes : 0;
    const int p_hi = tile == tile_hi ? u_hi - tile * g.passes : g.passes;
    // store offsets of my two pixel columns
    unsigned vo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = tile * kBN + j * 32 + li;
      vo[j] = p < g.Npix ? (unsigned)(ep_n[j] * g.M * g.HW + ep_r[j] + 4 * lh * g.HW) * 4u : kPoison;
    }
    const f32x4* bbase = &sB[buf][lh * kBN + li];           // + (4*kt + 2*gq) * 64 + j*32 : immediates

    for (int p = p_lo; p < p_hi; ++p) {
      const int mb = (p * kWaves + wave) * 32 * TM;         // first row of my block (wave-uniform)
      f32x16 acc[TM][2];
      // ---- K loop: alone on this SIMD's matrix pipe ----
      MS_T(t_l0);
#ifndef DASAC_MS_NOLOCK
      if (lane == 0)
        while (__hip_atomic_exchange(&s_sync[2 + simd], 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
#endif
      MS_T(t_l1);
      MS_ACC(1, t_l0, t_l1);
      f32x4 b[2][2][2];                                     // [stage][gq][j]
#define MS_READ_B(kt)                                                                                  \
  {                                                                                                    \
    _Pragma("unroll") for (int gq = 0; gq < 2; ++gq)                                                   \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) b[(kt) & 1][gq][j] = bbase[(4 * (kt) + 2 * gq) * kBN + j * 32]; \
  }
      MS_READ_B(0);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        if (kt + D < KT) MS_LOAD_A(kt + D, mb);
        if (kt + 1 < KT) MS_READ_B(kt + 1);
#pragma unroll
        for (int gq = 0; gq < 2; ++gq)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const f32x16 c = (kt == 0 && gq == 0 && e == 0) ? f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f} : acc[i][j];
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kt % (D + 1)][i][gq][e], b[kt & 1][gq][j][e], c, 0, 0, 0);
              }
        __builtin_amdgcn_sched_barrier(0);
      }
#if defined(DASAC_TRACE_TILES) && defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("s_nop 0" ::"v"(acc[i][j]));     // the stamp below waits for the last MFMA
#endif
      MS_T(t_l2);
      MS_ACC(2, t_l1, t_l2);
#ifndef DASAC_MS_NOLOCK
      if (lane == 0) __hip_atomic_store(&s_sync[2 + simd], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif

      // ---- beside my SIMD partner's K loop: publish the next tile, prefetch the next pass's weights, epilogue ----
      const bool last_pass = p + 1 == p_hi;
      if (last_pass && has_next) {       // every wave has finished sweeping the OTHER buffer (it published this tile after that)
        store_tile(buf ^ 1);
        publish(buf ^ 1);
      }
      if (!last_pass || has_next) {
        mb_next = ((last_pass ? 0 : p + 1) * kWaves + wave) * 32 * TM;
#pragma unroll
        for (int kt = 0; kt < D && kt < KT; ++kt) MS_LOAD_A(kt, mb_next);
      }

      MS_T(t_e1);
      MS_ACC(3, t_l2, t_e1);
#ifdef DASAC_TRACE_TILES
      unsigned long long tr_mid = 0;
#endif
      // per-row shift: the four rows of accumulator registers 4q .. 4q+3 of row group i are mb + i*32 + 8q + 4*lh + {0..3}
      f32x4 sh[TM][4];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) sh[i][q] = buf_f32x4(rsh, (unsigned)(mb + i * 32 + 8 * q + 4 * lh) * 4u, 0);   // (null shift: 0 records -> zeros)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int wcol = (tile * kBN + j * 32) >> 5;
        const int wcol_ld = wcol < g.w32 ? wcol : g.w32 - 1;   // (a column group entirely past the last pixel has no word: any value will do)
        float rs[TM][16];
        if (ep.res) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) rs[i][rg] = buf_f32(rres, vo[j], (mb + i * 32 + (rg & 3) + 8 * (rg >> 2)) * g.HW * 4);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int mrow = mb + i * 32;
          int bitrows = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 sh4 = sh[i][q];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int rg = 4 * q + e;
              const int row = mrow + (rg & 3) + 8 * (rg >> 2);
              float v = acc[i][j][rg] + sh4[e];
              if (ep.res) v = v + rs[i][rg];
              if (ep.relu) v = fmaxf(v, 0.f);
              if constexpr (BITS == 2) {
                // lanes 0-31 are the 32 pixels of row `row`, lanes 32-63 those of the row 4 below: the two mask words ARE the lane mask
                const unsigned lo = ep.mbits[(size_t)row * g.w32 + wcol_ld], hi = ep.mbits[(size_t)(row + 4) * g.w32 + wcol_ld];
                v = __builtin_amdgcn_inverse_ballot_w64(((unsigned long long)hi << 32) | lo) ? v : 0.f;
              }
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro, vo[j], row * g.HW * 4, 0);
              if constexpr (BITS == 1) bitrows = put_mask_rows(rg, __builtin_amdgcn_ballot_w64(v > 0.f), bitrows);
            }
          }
          if constexpr (BITS == 1) {
            if (lane < 32 && wcol < g.w32) ep.obits[(size_t)(mrow + lane) * g.w32 + wcol] = (unsigned)bitrows;
          }
        }
        asm volatile("" ::: "memory");
#ifdef DASAC_TRACE_TILES
        if (j == 0) {
          MS_T(t_e2);
          MS_ACC(4, t_e1, t_e2);
          tr_mid = t_e2;
        }
#endif
      }
      MS_T(t_l3);
      MS_ACC(5, tr_mid, t_l3);
    }
    // my pixel columns of the next tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      ep_r[j] += kBN;
      while (ep_r[j] >= g.HW) { ep_r[j] -= g.HW; ++ep_n[j]; }
    }
  }
#undef MS_LOAD_A
#undef MS_READ_B
#ifdef DASAC_TRACE_TILES
  if (g_ms_trace && lane == 0) {
    unsigned long long* o = g_ms_trace + ((size_t)blockIdx.x * kWaves + wave) * 12;
    o[0] = tr[0]; o[1] = tr[1]; o[2] = tr[2]; o[3] = tr[3];
    o[4] = __builtin_amdgcn_s_memtime() - tr_start;
    o[5] = simd;
    o[6] = u_hi - u_lo;
    o[7] = tr[4]; o[8] = tr[5];
  }
#endif
}

template <int K, int TM, int BITS>
static void launch(const float* x, const float* packed, float* out, const Geom& g, const Epi& ep, hipStream_t s) {
  const long long units = (long long)g.n_tiles * g.passes;
  const int grid = units < kNumCu ? (int)units : kNumCu;
  hipLaunchKernelGGL((gemm1x1_msweep<K, TM, BITS>), dim3(grid), dim3(kThreads), 0, s, x, packed, out, g, ep);
}

}  // namespace msweep
}  // namespace dasac

using namespace dasac;

#ifdef DASAC_TRACE_TILES
extern "C" int dasac_debug_set_ms_trace(void* buffer) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(buffer);
  return hipMemcpyToSymbol(HIP_SYMBOL(dasac::msweep::g_ms_trace), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

// 1 when dasac_conv_gemm routes this 1x1 stride-1 layer (K input channels -> M output channels) to the M-sweep kernel
// (DASAC_MSWEEP=0 keeps everything on the tile-per-block / stream-K kernels; outputs are bit-identical either way)
extern "C" int dasac_gemm1x1_msweep_ok(int M, int K) {
  // Off by default: measured on the cfg-3 shapes (profiles/r5_msweep_experiments.txt) the kernel wins only on the plain forward
  // (769 vs 805 us) -- a shape the networks do not launch; with residual / bit masks conv_gemm's tile-per-block kernel is ahead.
  static const int mode = getenv("DASAC_MSWEEP") ? atoi(getenv("DASAC_MSWEEP")) : 0;
  if (!mode) return 0;
  if (K != 64 && K != 128 && K != 256) return 0;
  return (M >= 256 && M % 256 == 0) ? 1 : 0;
}

// out[n][m][p] = epilogue(sum_k Wp[k][m] * x[n][k][p]): x [Nb,K,H,W], out / res [Nb,M,H,W], mask bits as in dasac_conv_gemm.
// No workspace, never allocates or synchronises.  Called by dasac_conv_gemm when dasac_gemm1x1_msweep_ok and the call is a
// plain 1x1 stride-1 convolution over the whole pixel range; exported for tools / tests that want to force it.
extern "C" int dasac_gemm1x1_msweep(const float* x, const float* packed, float* out, int Nb, int K, int HW, int M,
                                    const float* shift, const float* res, const uint32_t* mask_bits, uint32_t* relu_bits_out,
                                    int relu, dasac_stream_t stream) {
  DASAC_REQUIRE(x && packed && out, "gemm1x1_msweep: null pointer");
  DASAC_REQUIRE(Nb > 0 && HW > 0 && (K == 64 || K == 128 || K == 256) && M >= 256 && M % 256 == 0,
                "gemm1x1_msweep: needs K in {64,128,256} and M %% 256 == 0 (got K=%d M=%d)", K, M);
  DASAC_REQUIRE(!(mask_bits && relu_bits_out), "gemm1x1_msweep: one bit-mask direction per call");
  DASAC_REQUIRE(!relu_bits_out || relu, "gemm1x1_msweep: relu_bits_out records the pattern of a ReLU epilogue");
  DASAC_REQUIRE((int64_t)Nb * K * HW * 4 < (1ll << 31) && (int64_t)Nb * M * HW * 4 < (1ll << 31),
                "gemm1x1_msweep: tensor exceeds the 2 GiB buffer-descriptor window");
  msweep::Geom g;
  g.HW = HW; g.CxHW = K * HW; g.Npix = Nb * HW; g.M = M; g.Mpad = dasac_conv_mpad(M);
  g.w32 = (g.Npix + 31) / 32;
  g.x_bytes = Nb * K * HW * 4; g.w_bytes = dasac_conv_kpad(K) * g.Mpad * 4; g.out_bytes = Nb * M * HW * 4;
  g.n_tiles = (g.Npix + msweep::kBN - 1) / msweep::kBN;
  const int tm = M % 512 == 0 ? 2 : 1;
  g.passes = M / (256 * tm);
  msweep::Epi ep{shift, res, mask_bits, relu_bits_out, relu};
  hipStream_t s = as_stream(stream);
  const int bits = relu_bits_out ? 1 : (mask_bits ? 2 : 0);
#define MS_CASE(KK, TT, BB) \
  if (K == KK && tm == TT && bits == BB) msweep::launch<KK, TT, BB>(x, packed, out, g, ep, s);
#define MS_CASES(KK) MS_CASE(KK, 1, 0) MS_CASE(KK, 1, 1) MS_CASE(KK, 1, 2) MS_CASE(KK, 2, 0) MS_CASE(KK, 2, 1) MS_CASE(KK, 2, 2)
  MS_CASES(64) MS_CASES(128) MS_CASES(256)
#undef MS_CASES
#undef MS_CASE
  DASAC_CHECK_LAUNCH("gemm1x1_msweep");
  return DASAC_OK;
}
