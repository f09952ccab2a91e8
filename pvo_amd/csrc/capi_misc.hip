// capi_misc.hip — error strings / version of libpvo_hip.
#include "common.h"

extern "C" const char* pvo_strerror(int code) {
  switch (code) {
    case PVO_OK: return "ok";
    case PVO_EINVAL: return "invalid argument";
    case PVO_ELAUNCH: return "HIP launch error";
    case PVO_EWORKSPACE: return "workspace too small";
    case PVO_EUNSUPPORTED: return "unsupported size or configuration";
    default: return "unknown error";
  }
}

extern "C" int pvo_version(void) { return 100; }
