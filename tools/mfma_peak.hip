// fp32 MFMA peak probe: waves-per-SIMD sweep.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 0.001f, b = blockIdx.x * 0.002f + 1.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks_per_cu = 1; blocks_per_cu <= 4; ++blocks_per_cu) {
    for (int rep = 0; rep < 2; ++rep) {
      int iters = 4000, grid = 256 * blocks_per_cu;
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)grid * 4 * iters * 8 * 4 * 4096.0;
      printf("waves/SIMD=%d  %.3f ms  %.1f TF\n", blocks_per_cu, ms, flops / ms / 1e9);
    }
  }
  // long run to see sustained clocks
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k<4>, dim3(512), dim3(256), 0, 0, out, 4000);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("sustained 20 launches 2 waves/SIMD: %.1f TF\n", 20.0 * 512 * 4 * 4000 * 8 * 4 * 4096.0 / ms / 1e9);
  return 0;
}
