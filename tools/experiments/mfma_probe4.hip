// Per-wave tile 128x64 (8 accumulator tiles, 64 MFMAs per K-step) at 2 waves/SIMD vs 64x64 (4 tiles, 32 MFMAs) at 4:
// same auxiliary work per K-step except the operand reads (12 vs 8 ds_read_b128) and staging (6 vs 4 ds_write_b128,
// 12 vs 10 buffer loads).  hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/mfma_probe4.hip -o /tmp/probe4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int TM, int WPS, bool AUX>
__global__ __launch_bounds__(256, WPS) void k(float* out, const float* src, int iters, int bytes) {
  __shared__ f32x4 lds[2][1536];
  f32x16 acc[TM][2];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int t = threadIdx.x;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
  f32x4 st = {t * 0.001f, 1.f, 2.f, 3.f};
  for (int u = 0; u < 6; ++u) lds[0][(t + u * 256) % 1536] = st, lds[1][(t + u * 256) % 1536] = st;
  __syncthreads();
  int soff = (blockIdx.x & 255) * 4096, v0 = t, v1 = blockIdx.x;
  f32x4 ra[TM], rb[2];
  for (int i = 0; i < TM; ++i) ra[i] = st;
  rb[0] = rb[1] = st;
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    if (AUX) {
#pragma unroll
      for (int u = 0; u < 12; ++u) { v0 = v0 * 3 + v1; }
#pragma unroll
      for (int u = 0; u < 24; ++u) { soff = (soff * 5 + it) & 0xffff0; }
      const unsigned vo = (unsigned)((t * 16 + (v0 & 0)) & 0xffff0);
#pragma unroll
      for (int u = 0; u < TM; ++u) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, soff + u * 8192, 0);
        ra[u] = f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) rb[u][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)(t * 4), soff + (u * 4 + j) * 1024, 0));
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f32x4 fa[TM], fb[2];
      if (AUX) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = lds[buf][(t + g * 512 + i * 64) % 1536];
        fb[0] = lds[buf][(t + g * 128 + 1024) % 1536];
        fb[1] = lds[buf][(t + g * 128 + 1088) % 1536];
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = st;
        fb[0] = fb[1] = st;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][c], fb[0][c], acc[i][0], 0, 0, 0);
          acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][c], fb[1][c], acc[i][1], 0, 0, 0);
        }
    }
    if (AUX) {
#pragma unroll
      for (int i = 0; i < TM; ++i) lds[buf ^ 1][t + i * 256] = ra[i];
      lds[buf ^ 1][(t + 1024) % 1536] = rb[0];
      lds[buf ^ 1][(t + 1280) % 1536] = rb[1];
      __syncthreads();
    }
  }
  float s = v0 + soff;
  for (int i = 0; i < TM; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + t] = s + ra[0].x + rb[1].w;
}
template <int TM, int WPS, bool AUX> void run(float* out, float* src, int bytes, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, grid = 256 * WPS;
  hipLaunchKernelGGL((k<TM, WPS, AUX>), dim3(grid), dim3(256), 0, 0, out, src, iters, bytes);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<TM, WPS, AUX>), dim3(grid), dim3(256), 0, 0, out, src, iters, bytes);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double tf = (double)grid * 4 * iters * (TM * 2 * 8) * 4096.0 / ms / 1e9;
  printf("per-wave tile %3dx64, %d waves/SIMD, %-22s %8.3f ms %7.1f TF (%.1f%% of 157.3)\n", TM * 32, WPS, name, ms, tf, tf / 1.573);
}
int main() {
  const int bytes = 8 << 20;
  float *out, *src; hipMalloc(&out, 256 * 4 * 256 * 4 * 4); hipMalloc(&src, bytes + (1 << 20)); hipMemset(src, 0, bytes + (1 << 20));
  run<2, 4, false>(out, src, bytes, "MFMA only");
  run<2, 4, true>(out, src, bytes, "full K-step");
  run<2, 3, true>(out, src, bytes, "full K-step");
  run<2, 2, true>(out, src, bytes, "full K-step");
  run<4, 2, false>(out, src, bytes, "MFMA only");
  run<4, 2, true>(out, src, bytes, "full K-step");
  run<4, 1, true>(out, src, bytes, "full K-step");
  return 0;
}
