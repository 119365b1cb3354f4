// Version / error-string entry points of libxai_hip.so.
#include "xai_common.h"

XAI_EXPORT int xai_version(void) { return XAI_ABI_VERSION; }

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
