// common.h — shared device/host helpers for libpvo_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pvo_hip.h"

// the HIP error behind the most recent PVO_ELAUNCH (pvo_last_hip_error reports it)
extern "C" void pvo_note_hip_error(int code);
#define PVO_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t _e = hipGetLastError();                       \
    if (_e != hipSuccess) { pvo_note_hip_error(static_cast<int>(_e)); return PVO_ELAUNCH; } \
  } while (0)

static inline hipStream_t pvo_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---------------------------------------------------------------------------
// 16-bit storage types. Arithmetic on them follows the reference's c10::Half
// model (correlation_kernels.cu:56-65): every * and + is done in fp32 and the
// result rounded back to the storage type (round-to-nearest-even).
// ---------------------------------------------------------------------------
struct pvo_half { _Float16 v; };
struct pvo_bf16 { uint16_t v; };

__device__ __forceinline__ float pvo_bf16_to_f32(uint16_t h) {
  return __uint_as_float(static_cast<uint32_t>(h) << 16);
}
// round to nearest even, NaN stays a (quiet) NaN: one v_cvt_pk_bf16_f32 on gfx950.  Same values as the integer
// formulation in oracle/oracle_corr.c (f2b) for everything but the NaN payload.
__device__ __forceinline__ uint16_t pvo_f32_to_bf16(float f) {
  return __builtin_bit_cast(uint16_t, static_cast<__bf16>(f));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  using store_t = float;
  static __device__ __forceinline__ float to_f32(float x) { return x; }
  static __device__ __forceinline__ float from_f32(float x) { return x; }
};
template <> struct Elem<pvo_half> {
  using store_t = _Float16;
  static __device__ __forceinline__ float to_f32(_Float16 x) { return static_cast<float>(x); }
  static __device__ __forceinline__ _Float16 from_f32(float x) { return static_cast<_Float16>(x); }
};
template <> struct Elem<pvo_bf16> {
  using store_t = uint16_t;
  static __device__ __forceinline__ float to_f32(uint16_t x) { return pvo_bf16_to_f32(x); }
  static __device__ __forceinline__ uint16_t from_f32(float x) { return pvo_f32_to_bf16(x); }
};

// floor(x) -> int with saturation (the reference's static_cast<int>(floor(x)),
// correlation_kernels.cu:49-50; v_cvt_i32_f32 saturates and maps NaN to 0, which
// is also what the CUDA conversion does). Clamped to +-2^30 so that the +-radius
// offsets cannot wrap.
__device__ __forceinline__ int pvo_floor_to_int(float x) {
  float f = floorf(x);
  f = fminf(fmaxf(f, -1073741824.0f), 1073741824.0f);
  return (f != f) ? 0 : static_cast<int>(f);
}

// wave64 sum without LDS traffic: four DPP butterfly steps inside each 16-lane row
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then the four row sums are
// combined through v_readlane.  The result is uniform across the wave.
// (`__shfl_down` lowers to ds_bpermute_b32: measured 270 ns per 6-step reduction in the
// BA kernels, 3x the cost of everything else in them.)
template <int CTRL>
__device__ __forceinline__ float pvo_dpp_step(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
  return v + __int_as_float(moved);
}
__device__ __forceinline__ float pvo_wave_sum(float v) {
  v = pvo_dpp_step<0xB1>(v);    // quad_perm [1,0,3,2]
  v = pvo_dpp_step<0x4E>(v);    // quad_perm [2,3,0,1]
  v = pvo_dpp_step<0x141>(v);   // row_half_mirror
  v = pvo_dpp_step<0x140>(v);   // row_mirror
  const int b = __float_as_int(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}
