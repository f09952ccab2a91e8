// conv1x1_tile.h — one workgroup's share of y = x W^T + b for a 128-channel channels-last tensor (pvo_conv1x1_c128): 64 rows x
// 192 output channels.  A device function so that two kernels can carry it: conv1x1_c128_kernel (operator_small.hip) and, as a
// rider beside the one-workgroup pose solve, ba_solve_kernel (ba.hip).
#pragma once
#include "common.h"

namespace c1t {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef __bf16 v8b __attribute__((ext_vector_type(8)));

template <typename T> __device__ __forceinline__ v4f mfma(u32x4 a, u32x4 b, v4f c);
template <> __device__ __forceinline__ v4f mfma<pvo_half>(u32x4 a, u32x4 b, v4f c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ v4f mfma<pvo_bf16>(u32x4 a, u32x4 b, v4f c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8b, a), __builtin_bit_cast(v8b, b), c, 0, 0, 0);
}
template <typename T> __device__ __forceinline__ uint32_t bits(float x);
template <> __device__ __forceinline__ uint32_t bits<pvo_half>(float x) {
  union { _Float16 h; uint16_t u; } c; c.h = static_cast<_Float16>(x); return c.u;
}
template <> __device__ __forceinline__ uint32_t bits<pvo_bf16>(float x) { return pvo_f32_to_bf16(x); }

constexpr int kStride = 272;                 // input tile row stride (bytes)
constexpr int kTileBytes = 64 * 400;         // LDS the function needs: the input tile (64 x 272 B), later the output slab (64 x 400 B)

// rows [64 rb, 64 rb + 64) x output channels [192 cb, 192 cb + 192); 256 threads; `tile` = kTileBytes of LDS, 16-byte aligned
template <typename T>
__device__ __forceinline__ void conv1x1_c128_tile(unsigned char* tile, const uint16_t* __restrict__ x, const uint16_t* __restrict__ wt,
                                                  const float* __restrict__ bias, uint16_t* __restrict__ y,
                                                  long long rows, int Cout, int relu, long long rb, int cb) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const long long r0 = rb * 64;
  const int c0 = cb * 192 + wave * 48;
  u32x4 bf[4][3];
#pragma unroll
  for (int kc = 0; kc < 4; ++kc)
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
      bf[kc][nt] = *reinterpret_cast<const u32x4*>(wt + static_cast<size_t>(c0 + nt * 16 + li) * 128 + kc * 32 + lk * 8);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int id = tid + 256 * it, px = id >> 4, c = id & 15;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r0 + px < rows) v = *reinterpret_cast<const u32x4*>(x + static_cast<size_t>(r0 + px) * 128 + c * 8);
    *reinterpret_cast<u32x4*>(tile + px * kStride + c * 16) = v;
  }
  __syncthreads();
  v4f d[4][3];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) d[g][nt] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(tile + (g * 16 + li) * kStride + kc * 64 + lk * 16);
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) d[g][nt] = mfma<T>(a, bf[kc][nt], d[g][nt]);
    }
  __syncthreads();
  float bb[3];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) bb[nt] = bias ? bias[c0 + nt * 16 + li] : 0.0f;
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {                        // D rows lk*4 + r = pixels, column li = channel
        float v = d[g][nt][r] + bb[nt];
        if (relu) v = fmaxf(v, 0.0f);
        *reinterpret_cast<uint16_t*>(tile + (g * 16 + lk * 4 + r) * 400 + (wave * 48 + nt * 16 + li) * 2) = static_cast<uint16_t>(bits<T>(v));
      }
  __syncthreads();
  for (int id = tid; id < 64 * 24; id += 256) {             // 24 chunks of 16 B per row
    const int px = id / 24, c = id - px * 24;
    if (r0 + px < rows)
      *reinterpret_cast<u32x4*>(y + static_cast<size_t>(r0 + px) * Cout + cb * 192 + c * 8) =
          *reinterpret_cast<const u32x4*>(tile + px * 400 + c * 16);
  }
}

}  // namespace c1t
