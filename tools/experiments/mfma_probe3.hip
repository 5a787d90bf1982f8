// What does a K-step of the conv GEMM loop cost beyond its 32 MFMAs (2048 matrix-pipe cycles per wave)?
// 256-thread blocks, WPS waves per SIMD (blocks per CU), the kernel's accumulator dependency pattern.
//   bit 0: 8 ds_read_b128, each issued right before its use (the compiler's schedule at 128 registers)
//   bit 1: all 8 ds_read_b128 hoisted to the top of the step
//   bit 2: 4 ds_write_b128 after the MFMAs
//   bit 3: s_barrier at the end of the step
//   bit 4: 10 buffer loads at the top (2 dwordx4 + 8 dword), consumed by the LDS writes
//   bit 5: 24 SALU + 12 VALU of address arithmetic
// hipcc --offload-arch=gfx950 -O3 tools/mfma_probe3.hip -o /tmp/probe3 && /tmp/probe3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float* out, const float* src, int iters, int bytes) {
  __shared__ f32x4 lds[2][1024];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int t = threadIdx.x;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, bytes, 0x00020000);
  f32x4 st = {t * 0.001f, 1.f, 2.f, 3.f};
  for (int u = 0; u < 4; ++u) lds[0][(t + u * 256) & 1023] = st, lds[1][(t + u * 256) & 1023] = st;
  __syncthreads();
  int soff = (blockIdx.x & 255) * 4096, v0 = t, v1 = blockIdx.x;
  f32x4 ra[2] = {st, st}, rb[2] = {st, st};
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    if (MODE & 32) {
#pragma unroll
      for (int u = 0; u < 12; ++u) { v0 = v0 * 3 + v1; }
#pragma unroll
      for (int u = 0; u < 24; ++u) { soff = (soff * 5 + it) & 0xffff0; }
    }
    if (MODE & 16) {
      const unsigned vo = (unsigned)((t * 16 + (v0 & 0)) & 0xffff0);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, soff + u * 8192, 0);
        ra[u] = f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) rb[u][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)(t * 4), soff + (u * 4 + j) * 1024, 0));
    }
    f32x4 fa[4], fb[4];
    if (MODE & 2) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { fa[u] = lds[buf][(t + u * 64) & 1023]; fb[u] = lds[buf][(t + u * 64 + 512) & 1023]; }
    } else if (!(MODE & 1)) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { fa[u] = st; fb[u] = st; }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (MODE & 1) {   // read right before use: a0 b0 b1 | a1 (as the compiler schedules the real loop)
        fa[2 * g] = lds[buf][(t + g * 128) & 1023];
        fb[2 * g] = lds[buf][(t + g * 128 + 512) & 1023];
        fb[2 * g + 1] = lds[buf][(t + g * 128 + 576) & 1023];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2 * g][c], fb[2 * g][c], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2 * g][c], fb[2 * g + 1][c], acc[1], 0, 0, 0);
      }
      if (MODE & 1) fa[2 * g + 1] = lds[buf][(t + g * 128 + 64) & 1023];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2 * g + 1][c], fb[2 * g][c], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2 * g + 1][c], fb[2 * g + 1][c], acc[3], 0, 0, 0);
      }
    }
    if (MODE & 4) {
      lds[buf ^ 1][t] = ra[0]; lds[buf ^ 1][t + 256] = ra[1]; lds[buf ^ 1][t + 512] = rb[0]; lds[buf ^ 1][t + 768] = rb[1];
    }
    if (MODE & 8) __syncthreads();
  }
  float s = v0 + soff;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + t] = s + ra[0].x + rb[1].w;
}
template <int MODE, int WPS> void run(float* out, float* src, int bytes, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, grid = 256 * WPS;
  hipLaunchKernelGGL((k<MODE, WPS>), dim3(grid), dim3(256), 0, 0, out, src, iters, bytes);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, WPS>), dim3(grid), dim3(256), 0, 0, out, src, iters, bytes);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double tf = (double)grid * 4 * iters * 32 * 4096.0 / ms / 1e9;
  printf("%d waves/SIMD  %-58s %8.3f ms %7.1f TF  %6.0f pipe cycles per K-step\n", WPS, name, ms, tf, 2048.0 * 157.3 / tf * 2.4 / 2.4);
}
template <int WPS> void all(float* out, float* src, int bytes) {
  run<0, WPS>(out, src, bytes, "32 MFMA only");
  run<1, WPS>(out, src, bytes, "+8 ds_read_b128 just in time");
  run<2, WPS>(out, src, bytes, "+8 ds_read_b128 hoisted");
  run<1 | 4, WPS>(out, src, bytes, "+reads(jit) +4 ds_write_b128");
  run<1 | 4 | 8, WPS>(out, src, bytes, "+reads(jit) +writes +barrier");
  run<2 | 4 | 8, WPS>(out, src, bytes, "+reads(hoisted) +writes +barrier");
  run<8, WPS>(out, src, bytes, "+barrier only");
  run<16 | 4, WPS>(out, src, bytes, "+10 buffer loads +writes");
  run<32, WPS>(out, src, bytes, "+24 SALU +12 VALU");
  run<1 | 4 | 8 | 16, WPS>(out, src, bytes, "+reads(jit) +writes +barrier +loads");
  run<1 | 4 | 8 | 16 | 32, WPS>(out, src, bytes, "everything (jit reads)");
  run<2 | 4 | 8 | 16 | 32, WPS>(out, src, bytes, "everything (hoisted reads)");
}
int main() {
  const int bytes = 8 << 20;
  float *out, *src; hipMalloc(&out, 256 * 4 * 256 * 4 * 4); hipMalloc(&src, bytes + (1 << 20)); hipMemset(src, 0, bytes + (1 << 20));
  all<4>(out, src, bytes);
  all<3>(out, src, bytes);
  all<2>(out, src, bytes);
  return 0;
}
