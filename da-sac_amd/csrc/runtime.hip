// Error channel + device info for libdasac_hip.so.
#include "common.hpp"

#include <atomic>
#include <cstdlib>
#include <cstring>

namespace dasac {
static thread_local char g_err[512] = "";

static int clamp_reserved(int n) {
  if (n < 0) n = 0;
  if (n > kNumCu / 2) n = kNumCu / 2;
  return (n + kNumXcd - 1) / kNumXcd * kNumXcd;          // whole CUs per XCD
}
static std::atomic<int>& reserved_slot() {
  static std::atomic<int> v{clamp_reserved(getenv("DASAC_SK_RESERVE_CUS") ? atoi(getenv("DASAC_SK_RESERVE_CUS")) : 0)};
  return v;
}
int reserved_cus() { return reserved_slot().load(std::memory_order_relaxed); }

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace dasac

extern "C" int dasac_version(void) { return 1; }

extern "C" int dasac_reserved_cus(void) { return dasac::reserved_cus(); }
extern "C" int dasac_set_reserved_cus(int n) {
  return dasac::reserved_slot().exchange(dasac::clamp_reserved(n), std::memory_order_relaxed);
}

extern "C" const char* dasac_last_error(void) { return dasac::g_err; }

extern "C" int dasac_device_info(int* cu_count, int* wave_size, char* arch, size_t arch_len) {
  int dev = 0;
  DASAC_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  DASAC_HIP(hipGetDeviceProperties(&p, dev));
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (wave_size) *wave_size = p.warpSize;
  if (arch && arch_len) {
    strncpy(arch, p.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return DASAC_OK;
}
