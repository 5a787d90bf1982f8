// Which non-MFMA ingredient of the conv loop costs matrix-pipe time?  (3 waves/SIMD, 256-thread blocks)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const int4* tab, int iters) {
  __shared__ f32x4 lds[2][1024];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 0.001f, b = blockIdx.x * 0.002f + 1.f;
  int v0 = threadIdx.x, v1 = blockIdx.x, v2 = 3, v3 = 5;
  f32x4 st = {a, b, a, b};
  for (int it = 0; it < iters; ++it) {
    if (MODE & 1) {   // ~128 VALU integer ops (address-math stand-in)
#pragma unroll
      for (int u = 0; u < 32; ++u) { v0 = v0 * 3 + v1; v1 = v1 ^ (v0 >> 3); v2 = v2 + v0; v3 = max(v3, v2 & 1023); }
    }
    if (MODE & 2) {   // 8 scalar table loads + use
      int s = 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int4 e = tab[(it * 8 + u) & 1023]; s += e.x + e.y + e.z; }
      v2 += s;
    }
    f32x4 fa[4], fb[4];
    if (MODE & 4) {   // 8 ds_read_b128
#pragma unroll
      for (int u = 0; u < 4; ++u) { fa[u] = lds[it & 1][(threadIdx.x + u * 64) & 1023]; fb[u] = lds[it & 1][(threadIdx.x + u * 64 + 512) & 1023]; }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) { fa[u] = st; fb[u] = st; }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2 * u][c], fb[2 * u][c], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2 * u][c], fb[2 * u + 1][c], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2 * u + 1][c], fb[2 * u][c], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2 * u + 1][c], fb[2 * u + 1][c], acc[3], 0, 0, 0);
      }
    if (MODE & 8) {   // 4 ds_write_b128
#pragma unroll
      for (int u = 0; u < 4; ++u) lds[(it + 1) & 1][(threadIdx.x + u * 256) & 1023] = st + (float)v3;
    }
    if (MODE & 16) __syncthreads();
  }
  float s = v0 + v1 + v2 + v3;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(float* out, int4* tab, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 1000, grid = 256 * 3;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, tab, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, tab, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %.3f ms  %.1f TF\n", name, ms, (double)grid * 4 * iters * 32 * 4096.0 / ms / 1e9);
}
int main() {
  float* out; int4* tab; hipMalloc(&out, 256 * 4096 * 4); hipMalloc(&tab, 1024 * 16); hipMemset(tab, 0, 1024 * 16);
  run<0>(out, tab, "mfma only");
  run<1>(out, tab, "+128 VALU");
  run<2>(out, tab, "+8 s_load");
  run<3>(out, tab, "+VALU +s_load");
  run<4>(out, tab, "+8 ds_read_b128");
  run<12>(out, tab, "+ds_read +4 ds_write_b128");
  run<28>(out, tab, "+ds_read +ds_write +barrier");
  run<31>(out, tab, "everything");
  return 0;
}
