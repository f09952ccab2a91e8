// glo_tile.h — the ConvGRU's global context (VO_Module/droid_slam/modules/gru.py:13-15,22-30) as two device functions, so that the
// kernels of pvo_gru_glo_fused / pvo_gate_context (conv_small.hip, operator_small.hip) and the riders of the pose solve
// (ba.hip: the NEXT update's context computed inside this update's two solve dispatches) share one body each.
#pragma once
#include "common.h"
#include "conv1x1_tile.h"

namespace glt {

using c1t::u32x4;
using c1t::v4f;

constexpr int kTile = 64, kStride = 272;
constexpr int kTileBytes = kTile * kStride;          // LDS of glo_partial_means

// glo_part[e][chunk][c] = (1 / HW) * sum over the chunk's pixels of sigmoid(w(net) + b)[c] * net[c]: 64-pixel tiles through
// LDS, wave w owns output channels [32w, 32w+32) with its 8 weight fragments in registers (v_mfma_f32_16x16x32), the sigmoid
// gate and the pixel sum stay in registers in the accumulator layout (lane = channel).  256 threads; tile = kTileBytes of LDS.
template <typename T>
__device__ __forceinline__ void glo_partial_means(unsigned char* tile, const uint16_t* __restrict__ net, const uint16_t* __restrict__ ww,
                                                  const float* __restrict__ bias, float* __restrict__ glo,
                                                  int HW, int chunk, int e, int cidx, int chunks) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  u32x4 bf[4][2];
#pragma unroll
  for (int kc = 0; kc < 4; ++kc)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      bf[kc][nt] = *reinterpret_cast<const u32x4*>(ww + static_cast<size_t>(wave * 32 + nt * 16 + li) * 128 + kc * 32 + lk * 8);
  const float b0 = bias ? bias[wave * 32 + li] : 0.0f, b1 = bias ? bias[wave * 32 + 16 + li] : 0.0f;
  float s0 = 0.0f, s1 = 0.0f;
  const int p_begin = cidx * chunk, p_end = min(p_begin + chunk, HW);
  const uint16_t* ne = net + static_cast<size_t>(e) * HW * 128;
  // software pipeline: the next tile's 16 KB are in flight (registers) while this tile is multiplied
  u32x4 pre[4];
  auto fetch = [&](int p0) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {                      // 64 px x 16 chunks of 16 B
      const int id = tid + 256 * it, px = id >> 4, c = id & 15;
      pre[it] = u32x4{0u, 0u, 0u, 0u};
      if (p0 + px < p_end) pre[it] = *reinterpret_cast<const u32x4*>(ne + static_cast<size_t>(p0 + px) * 128 + c * 8);
    }
  };
  fetch(p_begin);
  for (int p0 = p_begin; p0 < p_end; p0 += kTile) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int id = tid + 256 * it, px = id >> 4, c = id & 15;
      *reinterpret_cast<u32x4*>(tile + px * kStride + c * 16) = pre[it];
    }
    __syncthreads();
    if (p0 + kTile < p_end) fetch(p0 + kTile);
    u32x4 afr[4][4];                                       // the tile's 16 A fragments, requested up front
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
        afr[g][kc] = *reinterpret_cast<const u32x4*>(tile + (g * 16 + li) * kStride + kc * 64 + lk * 16);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      v4f d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        d0 = c1t::mfma<T>(afr[g][kc], bf[kc][0], d0);
        d1 = c1t::mfma<T>(afr[g][kc], bf[kc][1], d1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {                       // D rows lk*4 + r = pixels, column li = channel
        const unsigned char* row = tile + (g * 16 + lk * 4 + r) * kStride + (wave * 32 + li) * 2;
        const float n0 = Elem<T>::to_f32(*reinterpret_cast<const typename Elem<T>::store_t*>(row));
        const float n1 = Elem<T>::to_f32(*reinterpret_cast<const typename Elem<T>::store_t*>(row + 32));
        // (v_exp + v_rcp: an IEEE division here was a third of this kernel's 128 sigmoids x 72 cycles per lane; padded pixels carry net = 0)
        s0 = fmaf(n0, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * (d0[r] + b0))), s0);
        s1 = fmaf(n1, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * (d1[r] + b1))), s1);
      }
    }
  }
  s0 += __shfl_xor(s0, 16, 64); s0 += __shfl_xor(s0, 32, 64);
  s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
  if (lk == 0) {                                          // this workgroup's partial mean: no zero fill, no atomics
    const float inv = 1.0f / static_cast<float>(HW);
    float* o = glo + (static_cast<size_t>(e) * chunks + cidx) * 128 + wave * 32;
    o[li] = s0 * inv;
    o[16 + li] = s1 * inv;
  }
}

// g[e, 0:384] = Wg glo[e] + bg, glo[e] = sum of the chunks' partial means; the 128-term dot products run as four independent
// partial sums (L2 latency, not bandwidth, is the cost).  glo = 128 floats of LDS.
__device__ __forceinline__ void glo_sum_chunks(float* glo, const float* __restrict__ part, int chunks, int e) {
  const int tid = threadIdx.x;
  if (tid < 128) {
    float s = 0.0f;
    for (int k0 = 0; k0 < chunks; k0 += 16) {
      float pv[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) pv[k] = k0 + k < chunks ? part[(static_cast<size_t>(e) * chunks + k0 + k) * 128 + tid] : 0.0f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += pv[k];               // (same order as a serial sum: + 0.0f for the padding)
    }
    glo[tid] = s;
  }
}
// ... 384 threads, one output each: all 128 weights of the output are requested before anything waits
__device__ __forceinline__ void gate_context_384(float* glo, const float* __restrict__ part, const float* __restrict__ wg_t,
                                                 const float* __restrict__ gb, float* __restrict__ g, int chunks, int e) {
  const int t = threadIdx.x;
  float wv[128];
#pragma unroll
  for (int q = 0; q < 128; ++q) wv[q] = wg_t[static_cast<size_t>(q) * 384 + t];
  glo_sum_chunks(glo, part, chunks, e);
  __syncthreads();
  float a[4] = {gb[t], 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int q = 0; q < 128; ++q) a[q & 3] = fmaf(glo[q], wv[q], a[q & 3]);
  g[static_cast<size_t>(e) * 384 + t] = (a[0] + a[1]) + (a[2] + a[3]);
}
// ... 256 threads (the rider: in no hurry), outputs t and t + 256, the weights in rounds of 32; the same sums in the same order
__device__ __forceinline__ void gate_context_256(float* glo, const float* __restrict__ part, const float* __restrict__ wg_t,
                                                 const float* __restrict__ gb, float* __restrict__ g, int chunks, int e) {
  glo_sum_chunks(glo, part, chunks, e);
  __syncthreads();
#pragma unroll 1
  for (int t = threadIdx.x; t < 384; t += 256) {
    float a[4] = {gb[t], 0.0f, 0.0f, 0.0f};
#pragma unroll 1
    for (int c0 = 0; c0 < 128; c0 += 32) {
      float wv[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) wv[q] = wg_t[static_cast<size_t>(c0 + q) * 384 + t];
#pragma unroll
      for (int q = 0; q < 32; ++q) a[q & 3] = fmaf(glo[c0 + q], wv[q], a[q & 3]);
    }
    g[static_cast<size_t>(e) * 384 + t] = (a[0] + a[1]) + (a[2] + a[3]);
  }
}

}  // namespace glt
