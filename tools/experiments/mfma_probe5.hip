// Round 5: what does a SECOND wave on the same SIMD cost a wave that issues fp32 MFMAs back to back -- and how fast does that
// second wave's instruction stream run?  512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run ITERS x 32 (or 64)
// independent-accumulator MFMAs from registers, waves 4-7 run a partner stream until the compute waves are done:
//   partner 0: none (exit)   1: v_add_f32 on 32 independent registers   2: buffer_store_dword, the conv epilogue's pattern
//   3: s_add / s_mul chain   4: ds_read_b128
// MFMA type 0: v_mfma_f32_32x32x2_f32 (4 accumulators of 16 registers, 64 cycles each)
//           1: v_mfma_f32_16x16x4_f32 (16 accumulators of 4 registers, 32 cycles each) -- same flops per iteration.
// Prints per configuration: cycles per 32x32x2-equivalent MFMA of the compute waves, partner operations per microsecond.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_probe5.hip -o /tmp/probe5 && /tmp/probe5
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MT, int PM, int SWAP = 0, int PRIO = 0, int YIELD = 0>
__global__ __launch_bounds__(512, 2) void k(float* out, float* scratch, unsigned long long* stats, int iters, int bytes) {
  __shared__ int s_done;
  __shared__ f32x4 lds[1024];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) s_done = 0;
  lds[t] = f32x4{1.f, 2.f, 3.f, 4.f};
  lds[t + 512] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if ((wave < 4) != (SWAP != 0)) {
    float a = 1.0f + lane * 1e-3f, b = 0.5f;
    if (MT == 0) {
      f32x16 acc[4];
      for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            if (YIELD == 1) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 15"); asm volatile("s_nop 15"); asm volatile("s_nop 15"); __builtin_amdgcn_sched_barrier(0); }
            if (YIELD == 2) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(1); __builtin_amdgcn_sched_barrier(0); }
            if (YIELD == 3) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 15"); __builtin_amdgcn_sched_barrier(0); }
            if (YIELD == 4 && (i & 1)) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(1); __builtin_amdgcn_sched_barrier(0); }
          }
      }
      float s = 0.f;
      for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
      out[blockIdx.x * 512 + t] = s;
    } else {
      f32x4 acc[16];
      for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      }
      float s = 0.f;
      for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
      out[blockIdx.x * 512 + t] = s;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __hip_atomic_fetch_add(&s_done, lane == 0 ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (lane == 0) stats[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
  } else {
    unsigned long long ops = 0;
    if (PM == 0) return;
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(scratch, 0, bytes, 0x00020000);
    float v[32];
    for (int i = 0; i < 32; ++i) v[i] = lane * 0.5f + i;
    int sa = blockIdx.x + 1, sb = 3;
    const unsigned vo = (unsigned)((blockIdx.x * 8 + wave) * 16384 + (lane & 31) * 4 + (lane >> 5) * 4 * 9409 * 4) & 0x3ffffffc;
    while (__hip_atomic_load(&s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
      if (PM == 1) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 32; ++i) v[i] = v[i] + 1.0f;
        ops += 64;
      } else if (PM == 2) {
#pragma unroll
        for (int i = 0; i < 32; ++i) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[i]), rs, vo, ((i & 3) + 8 * (i >> 2)) * 9409 * 4, 0);
        ops += 32;
      } else if (PM == 3) {
#pragma unroll
        for (int i = 0; i < 64; ++i) { sa = sa * 5 + sb; asm volatile("" : "+s"(sa)); }
        ops += 64;
      } else if (PM == 4) {
        f32x4 x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = lds[(t + i * 64) & 1023];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += x[i].x;
        ops += 8;
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += v[i];
    out[blockIdx.x * 512 + t] = s + sa;
    if (lane == 0) { stats[(blockIdx.x * 8 + wave) * 2] = t1 - t0; stats[(blockIdx.x * 8 + wave) * 2 + 1] = ops; }
  }
}

template <int MT, int PM, int SWAP = 0, int PRIO = 0, int YIELD = 0>
void run(const char* name, float* out, float* scratch, unsigned long long* stats, int bytes) {
  const int iters = SWAP ? 100 : 2000;
  hipMemset(stats, 0, 256 * 8 * 2 * 8);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<MT, PM, SWAP, PRIO, YIELD>), dim3(256), dim3(512), 0, 0, out, scratch, stats, 10, bytes);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MT, PM, SWAP, PRIO, YIELD>), dim3(256), dim3(512), 0, 0, out, scratch, stats, iters, bytes);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  static unsigned long long h[256 * 8 * 2];
  hipMemcpy(h, stats, sizeof(h), hipMemcpyDeviceToHost);
  double cticks = 0, pticks = 0, pops = 0;
  for (int bq = 0; bq < 256; ++bq)
    for (int w = 0; w < 8; ++w) {
      if ((w < 4) != (SWAP != 0)) cticks += h[(bq * 8 + w) * 2];
      else { pticks += h[(bq * 8 + w) * 2]; pops += h[(bq * 8 + w) * 2 + 1]; }
    }
  cticks /= 1024; pticks /= 1024; pops /= 1024;
  // s_memtime ticks are calibrated on the launch: the compute waves live (almost) the whole launch
  const double tick_us = ms * 1e3 / cticks;
  const double mfma_equiv = iters * 32.0;
  printf("%-56s launch %8.1f us = %6.1f ns per 32x32x2-equivalent MFMA (%6.1f TFLOP/s) | partner: %10.0f ops in %8.1f us = %8.1f ops/us per wave\n", name,
         ms * 1e3, ms * 1e6 / mfma_equiv, mfma_equiv * 4096.0 * 1024 / (ms * 1e-3) / 1e12, pops, pticks * tick_us, pops / (pticks * tick_us + 1e-9));
  fflush(stdout);
}

int main() {
  float *out, *scratch;
  unsigned long long* stats;
  const int bytes = 256 * 8 * 16384 + 64 * 9409 * 4 * 2;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&scratch, (size_t)bytes + (1 << 22));
  hipMalloc(&stats, 256 * 8 * 2 * 8);
  run<0, 0>("32x32x2  alone", out, scratch, stats, bytes);
  run<0, 1>("32x32x2  + partner v_add", out, scratch, stats, bytes);
  run<0, 2>("32x32x2  + partner dword stores", out, scratch, stats, bytes);
  run<0, 3>("32x32x2  + partner SALU", out, scratch, stats, bytes);
  run<0, 4>("32x32x2  + partner ds_read_b128", out, scratch, stats, bytes);
  run<0, 1, 0, 0, 1>("32x32x2 + 3 x s_nop 15 after each MFMA; partner v_add", out, scratch, stats, bytes);
  run<0, 2, 0, 0, 1>("32x32x2 + 3 x s_nop 15 after each MFMA; partner stores", out, scratch, stats, bytes);
  run<0, 1, 0, 0, 3>("32x32x2 + 1 x s_nop 15 after each MFMA; partner v_add", out, scratch, stats, bytes);
  run<0, 1, 0, 0, 2>("32x32x2 + s_sleep 1 after each MFMA; partner v_add", out, scratch, stats, bytes);
  run<0, 2, 0, 0, 2>("32x32x2 + s_sleep 1 after each MFMA; partner stores", out, scratch, stats, bytes);
  run<0, 1, 0, 0, 4>("32x32x2 + s_sleep 1 after every 2nd MFMA; partner v_add", out, scratch, stats, bytes);
  run<0, 0, 0, 0, 1>("32x32x2 + 3 x s_nop 15, alone", out, scratch, stats, bytes);
  run<0, 0, 0, 0, 2>("32x32x2 + s_sleep 1, alone", out, scratch, stats, bytes);
  return 0;
}
