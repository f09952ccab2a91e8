// capi_misc.hip — error strings / version of libpvo_hip, and the shader-clock probe.
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

static thread_local int g_last_hip_error = 0;
extern "C" void pvo_note_hip_error(int code) { g_last_hip_error = code; }
extern "C" const char* pvo_last_hip_error(void) {
  return g_last_hip_error ? hipGetErrorString(static_cast<hipError_t>(g_last_hip_error)) : "no HIP error recorded";
}

// One wave runs a dependent chain of `iters` x 64 v_fma_f32 and reports how many shader cycles (s_memtime) and how many
// 10 ns ticks of the constant 100 MHz counter (s_memrealtime) it took: launched on a second stream beside a kernel, the
// ratio is the clock the chip sustains under that kernel's load (MI355X lowers the clock to hold its power budget).
namespace {
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, int iters) {
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  float a = static_cast<float>(threadIdx.x), b = 1.0001f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 64; ++k) a = __builtin_fmaf(a, b, 0.5f);
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = static_cast<unsigned long long>(a); }
}
}  // namespace

extern "C" int pvo_clock_probe(void* out3_u64, int iters, void* stream) {
  if (!out3_u64 || iters <= 0) return PVO_EINVAL;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, pvo_stream(stream), static_cast<unsigned long long*>(out3_u64), iters);
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}
