// Write bandwidth of the conv epilogue's store pattern vs tile shape (diagnostic, gfx950): every workgroup of 256 threads writes
// one [rows x px] tile of a [planes][plane_px] fp32 tensor with 4-byte stores, lanes along pixels -- 128 x 128 is what
// conv_gemm's epilogue does (512-byte runs at a 37.6 KB pitch).  Build: hipcc --offload-arch=gfx950 -O3 tools/store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ROWS, int PX, int VEC>
__global__ __launch_bounds__(256) void tile_store(float* out, int planes, int plane_px, int tiles_px) {
  const int tile = blockIdx.x;
  const int pt = tile % tiles_px, rt = tile / tiles_px;
  const int p0 = pt * PX, r0 = rt * ROWS;
  constexpr int LANES_PX = PX / VEC;                 // vector stores along a tile row
  const float v = (float)tile;
#pragma unroll 4
  for (int idx = threadIdx.x; idx < ROWS * LANES_PX; idx += 256) {      // lanes run along pixels first, like the epilogue's
    const int r = idx / LANES_PX, lx = idx - r * LANES_PX;
    const int row = r0 + r, px = p0 + lx * VEC;
    if (row < planes && px + VEC <= plane_px) {
      float* p = out + (size_t)row * plane_px + px;
      if (VEC == 1) p[0] = v;
      else {
        typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
        *reinterpret_cast<f4*>(p) = f4{v, v, v, v};
      }
    }
  }
}

template <int ROWS, int PX, int VEC>
static void run(const char* name, float* buf, int planes, int plane_px) {
  const int tiles_px = (plane_px + PX - 1) / PX, tiles_r = (planes + ROWS - 1) / ROWS;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((tile_store<ROWS, PX, VEC>), dim3(tiles_px * tiles_r), dim3(256), 0, 0, buf, planes, plane_px, tiles_px);
  hipEventRecord(a);
  const int iters = 10;
  for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((tile_store<ROWS, PX, VEC>), dim3(tiles_px * tiles_r), dim3(256), 0, 0, buf, planes, plane_px, tiles_px);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)planes * plane_px * 4.0;
  printf("%-34s %4d x %4d px tiles, %2d B/lane: %7.1f us  %5.2f TB/s\n", name, ROWS, PX, VEC * 4, ms / iters * 1e3, bytes / (ms / iters * 1e-3) / 1e12);
}

int main() {
  // the M = 1024 output of the fused student pass: 1024 channel planes... as the GEMM sees it: rows = n*M + m planes of 9409 px.
  // Flattened-pixel tiling crosses images; this probe tiles [16*1024 planes][9409 px] -- same run lengths and pitches.
  const int planes = 16 * 1024, plane_px = 97 * 97;
  float* buf; hipMalloc(&buf, (size_t)planes * plane_px * 4);
  run<128, 128, 1>("epilogue pattern (dword)", buf, planes, plane_px);
  run<128, 128, 4>("same tile, dwordx4 stores", buf, planes, plane_px);
  run<64, 256, 1>("64 rows x 1 KB runs", buf, planes, plane_px);
  run<32, 512, 1>("32 rows x 2 KB runs", buf, planes, plane_px);
  run<16, 1024, 1>("16 rows x 4 KB runs", buf, planes, plane_px);
  run<16, 1024, 4>("16 rows x 4 KB runs, dwordx4", buf, planes, plane_px);
  run<256, 64, 1>("256 rows x 256 B runs", buf, planes, plane_px);
  run<1, 1024, 4>("1 row x 4 KB: contiguous stream", buf, planes, plane_px);
  return 0;
}
