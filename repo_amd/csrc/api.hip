// ABI bookkeeping entry points.
#include "common.h"

extern "C" int repo_abi_version(void) { return REPO_ABI_VERSION; }

extern "C" const char* repo_strerror(int code) {
  switch (code) {
    case REPO_OK: return "ok";
    case REPO_E_BADARG: return "bad argument (null pointer or unsupported option)";
    case REPO_E_SHAPE: return "unsupported shape or index range";
    case REPO_E_ALIGN: return "misaligned pointer";
    case REPO_E_WS_TOO_SMALL: return "workspace too small";
    case REPO_E_ARCH: return "device is not gfx950";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown error";
}
