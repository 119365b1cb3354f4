// Version / error-string entry points of libxai_hip.so.
#include "xai_common.h"

#include <atomic>

int xai_cu_count() {
  constexpr int kMaxDev = 64;
  static std::atomic<int> cached[kMaxDev];                 // zero-initialised; 0 = not queried yet
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
  int n = cached[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

XAI_EXPORT int xai_version(void) { return XAI_ABI_VERSION; }
XAI_EXPORT int xai_version_minor(void) { return XAI_ABI_MINOR; }

XAI_EXPORT const char* xai_strerror(int code) {
  switch (code) {
    case XAI_OK: return "success";
    case XAI_E_NULL: return "required pointer is NULL";
    case XAI_E_SHAPE: return "bad extent, inconsistent shape or misaligned buffer";
    case XAI_E_UNSUPPORTED: return "extent not supported by this build of the kernel";
    default: break;
  }
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  return "unknown xai error code";
}
