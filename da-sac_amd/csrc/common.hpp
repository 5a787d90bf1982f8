// Shared plumbing for libdasac_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "dasac_hip.h"

namespace dasac {

constexpr int kWave = 64;          // CDNA4 wavefront
constexpr int kNumXcd = 8;         // MI355X: 8 XCDs x 32 CUs
constexpr int kNumCu = 256;

int fail(int code, const char* fmt, ...);   // records dasac_last_error(), returns code

// CUs this process leaves to kernels that run BESIDE its own (RCCL's all-reduce kernels under the overlapped data-parallel
// wrapper): 0 by default, a multiple of 8 (whole CUs per XCD); dasac_set_reserved_cus / DASAC_SK_RESERVE_CUS.  Sizes the
// persistent stream-K grid and the grid cap of the streaming kernels.
int reserved_cus();

inline hipStream_t as_stream(dasac_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define DASAC_REQUIRE(cond, ...)                                        \
  do {                                                                  \
    if (!(cond)) return ::dasac::fail(DASAC_EINVAL, __VA_ARGS__);       \
  } while (0)

#define DASAC_CHECK_LAUNCH(what)                                                          \
  do {                                                                                    \
    hipError_t e_ = hipGetLastError();                                                    \
    if (e_ != hipSuccess) return ::dasac::fail(DASAC_ELAUNCH, "%s: %s", what, hipGetErrorString(e_)); \
  } while (0)

#define DASAC_HIP(call)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) return ::dasac::fail(DASAC_ELAUNCH, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// grid for a grid-stride streaming kernel: enough blocks to fill 256 CUs several times over
inline int stream_grid(int64_t work_items, int block, int max_blocks = 0) {
  if (max_blocks <= 0) max_blocks = (kNumCu - reserved_cus()) * 16;
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return static_cast<int>(g);
}

// ---- n / d for 0 <= n < 2^31 and a launch-constant d >= 1: one v_mul_hi_u32 + one shift instead of the ~30-instruction
// runtime division (the streaming kernels are issue-bound on exactly such index math: tools/isa_count.py).
// l = ceil(log2 d), mul = ceil(2^(31+l) / d) < 2^32, n / d = (n * mul) >> (31 + l)  (exact: 2^(31+l) <= mul*d <= 2^(31+l) + 2^l);
// d = 1 is flagged by mul = 0.
struct FastDiv {
  unsigned mul, shift;
};
inline FastDiv fast_div(int d) {
  if (d <= 1) return FastDiv{0u, 0u};
  int l = 1;
  while ((1ll << l) < d) ++l;
  const unsigned long long p = 1ull << (31 + l);
  return FastDiv{(unsigned)((p + (unsigned)d - 1) / (unsigned)d), (unsigned)(l - 1)};
}
__device__ __forceinline__ int fdiv(int n, FastDiv f) { return f.mul ? (int)(__umulhi((unsigned)n, f.mul) >> f.shift) : n; }

// ---- wave / block reductions (64-wide) ------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace dasac
