// capi_misc.hip — error strings / version of libpvo_hip.  (The shader-clock and memory-request probes bench.py reports moved to
// probe_tools.hip -> libpvo_probe.so in round 4: measurement kernels are not part of the product library.)
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

// 101: pvo_graph_update_args grew by context_ahead / context_ready (round 3) - a caller built against a 100 header passes a
// shorter struct; pvo_graph_update_args_size() lets any caller compare its sizeof with the library's before the first call
// 102: pvo_ba_pack / pvo_ba_finish_packed / pvo_ba_last_partition added, the BA workspace grew (round 4)
// 103: pvo_debug_config / pvo_knob replace the library's environment switches (round 5)
extern "C" int pvo_version(void) { return PVO_ABI_VERSION; }

static int g_knobs[PVO_KNOB_COUNT] = {};
extern "C" int pvo_debug_config(int knob, int value) {
  if (knob < 0 || knob >= PVO_KNOB_COUNT) return PVO_EINVAL;
  g_knobs[knob] = value;
  return PVO_OK;
}
extern "C" int pvo_knob(int knob) { return (knob < 0 || knob >= PVO_KNOB_COUNT) ? 0 : g_knobs[knob]; }
extern "C" size_t pvo_graph_update_args_size(void) { return sizeof(pvo_graph_update_args); }

static thread_local int g_last_hip_error = 0;
extern "C" void pvo_note_hip_error(int code) { g_last_hip_error = code; }
extern "C" const char* pvo_last_hip_error(void) {
  return g_last_hip_error ? hipGetErrorString(static_cast<hipError_t>(g_last_hip_error)) : "no HIP error recorded";
}
