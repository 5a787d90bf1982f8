// Error channel + device info for libdasac_hip.so.
#include "common.hpp"

#include <cstring>

namespace dasac {
static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace dasac

extern "C" int dasac_version(void) { return 1; }

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
