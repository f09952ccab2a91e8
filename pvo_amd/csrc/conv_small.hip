// conv_small.hip — the update operator's small-K convolutions on the matrix cores.
//
// pvo_conv7x7_c8: y = relu(conv7x7(x, w) + bias), x [E,H,W,8] -> y [E,H,W,128], zero padding 3.
//   reference: flow_encoder[0:2] = Conv2d(4+2+2, 128, 7, padding=3) + ReLU (VO_Module/droid_slam/droid_net.py:176-180),
//   applied to the motion features of factor_graph.py:233-237.
//   MIOpen needs 27 us for this convolution (K = 392, 8 input channels: a poor implicit-GEMM shape) plus a 9 us
//   bias+ReLU pass; here it is one kernel bound by its 28 MB output write.
//   Mapping: 8 input channels in 16-bit = 16 bytes = exactly one lane's k-group of v_mfma_f32_16x16x32, so one MFMA
//   consumes 4 taps (lane group lk = lane >> 4 selects the tap) of 16 pixels against 16 output channels.  49 taps = 13
//   MFMAs (the last one padded with zero weights).  A workgroup owns a 16x16 pixel tile whose 22x22 halo (7.7 KB) sits
//   in LDS; wave w owns output channels [32w, 32w+32) and keeps its 26 weight fragments in registers for all 16 tile rows.
//   Weights arrive pre-arranged as [52 taps (49 + 3 zero)][128 outputs][8 channels] 16-bit.
#include "common.h"

namespace {

typedef uint32_t cs_u32x4 __attribute__((ext_vector_type(4)));
typedef float cs_v4f __attribute__((ext_vector_type(4)));
typedef _Float16 cs_v8h __attribute__((ext_vector_type(8)));
typedef __bf16 cs_v8b __attribute__((ext_vector_type(8)));

template <typename T> __device__ __forceinline__ cs_v4f cs_mfma(cs_u32x4 a, cs_u32x4 b, cs_v4f c);
template <> __device__ __forceinline__ cs_v4f cs_mfma<pvo_half>(cs_u32x4 a, cs_u32x4 b, cs_v4f c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cs_v8h, a), __builtin_bit_cast(cs_v8h, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ cs_v4f cs_mfma<pvo_bf16>(cs_u32x4 a, cs_u32x4 b, cs_v4f c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cs_v8b, a), __builtin_bit_cast(cs_v8b, b), c, 0, 0, 0);
}

template <typename T> __device__ __forceinline__ uint32_t cs_bits(float x);
template <> __device__ __forceinline__ uint32_t cs_bits<pvo_half>(float x) {
  union { _Float16 h; uint16_t u; } c; c.h = static_cast<_Float16>(x); return c.u;
}
template <> __device__ __forceinline__ uint32_t cs_bits<pvo_bf16>(float x) { return pvo_f32_to_bf16(x); }

constexpr int kTH = 8, kTW = 16, kR = 3;
constexpr int kTH7 = 16;                                      // the 7x7 kernel's tile is 16 rows high: weight fragments are
                                                              // fetched once per 256 pixels (they were 3x the output in L2 reads)
constexpr int kHW_ = kTW + 2 * kR, kHH7 = kTH7 + 2 * kR;      // 22 x 22 halo
constexpr int kTaps = 49, kSteps = 13;                        // 13 MFMAs x 4 taps

template <typename T>
__global__ __launch_bounds__(256) void conv7x7_c8_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ wt,
                                                         const float* __restrict__ bias, uint16_t* __restrict__ y,
                                                         int H, int W) {
  __shared__ __attribute__((aligned(16))) unsigned char halo[kHH7 * kHW_ * 16];
  __shared__ __attribute__((aligned(16))) unsigned char slab[2][16 * 272];     // one tile row: 16 pixels x 128 channels (+16 B pad), double buffered
  const int e = blockIdx.z, y0 = blockIdx.y * kTH7, x0 = blockIdx.x * kTW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;

  // weight fragments of this wave's two 16-channel column tiles: B[k = (tap, ch)][n]; lane (li, lk) holds column li,
  // k-group lk = tap 4*s + lk of step s, all 8 channels
  cs_u32x4 bf[kSteps][2];
#pragma unroll
  for (int s = 0; s < kSteps; ++s)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      bf[s][nt] = *reinterpret_cast<const cs_u32x4*>(wt + (static_cast<size_t>(4 * s + lk) * 128 + wave * 32 + nt * 16 + li) * 8);

  const uint16_t* xe = x + static_cast<size_t>(e) * H * W * 8;
  for (int pos = tid; pos < kHH7 * kHW_; pos += 256) {
    const int hy = y0 - kR + pos / kHW_, hx = x0 - kR + pos % kHW_;
    cs_u32x4 v = {0u, 0u, 0u, 0u};
    if (hy >= 0 && hy < H && hx >= 0 && hx < W) v = *reinterpret_cast<const cs_u32x4*>(xe + (static_cast<size_t>(hy) * W + hx) * 8);
    *reinterpret_cast<cs_u32x4*>(halo + pos * 16) = v;
  }
  // LDS offset of this lane's tap in step s, relative to the tile row: taps >= 49 have zero weights, any address does
  int toff[kSteps];
#pragma unroll
  for (int s = 0; s < kSteps; ++s) {
    const int tap = min(4 * s + lk, kTaps - 1);
    toff[s] = ((tap / 7) * kHW_ + (tap % 7) + li) * 16;
  }
  float bb[2];
  bb[0] = bias[wave * 32 + li];
  bb[1] = bias[wave * 32 + 16 + li];
  __syncthreads();

  for (int py = 0; py < kTH7; ++py) {
    if (y0 + py >= H) break;                                  // (uniform) rows below the image
    cs_v4f d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    const unsigned char* rowp = halo + py * kHW_ * 16;
    // (clock64 stamps: 5k cycles of set-up + 1364 cycles per row per workgroup, i.e. the kernel's ~19 us is one workgroup's
    // latency - only 1.7 workgroups per CU exist at S-B - not a throughput limit; forcing the reads ahead changed nothing)
    cs_u32x4 afr[kSteps];
#pragma unroll
    for (int s = 0; s < kSteps; ++s) afr[s] = *reinterpret_cast<const cs_u32x4*>(rowp + toff[s]);
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
      d0 = cs_mfma<T>(afr[s], bf[s][0], d0);
      d1 = cs_mfma<T>(afr[s], bf[s][1], d1);
    }
    // D: column li = channel, rows lk*4 + r = pixels.  bias + ReLU, then the four waves' 32-channel slices meet in a
    // workgroup slab so that every pixel leaves as one contiguous 256-byte row (a wave storing its own 64-byte slice
    // writes half cache lines: 1.3 TB/s measured)
    unsigned char* sl = slab[py & 1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = lk * 4 + r;
      *reinterpret_cast<uint16_t*>(sl + p * 272 + (wave * 32 + li) * 2) = static_cast<uint16_t>(cs_bits<T>(fmaxf(d0[r] + bb[0], 0.0f)));
      *reinterpret_cast<uint16_t*>(sl + p * 272 + (wave * 32 + 16 + li) * 2) = static_cast<uint16_t>(cs_bits<T>(fmaxf(d1[r] + bb[1], 0.0f)));
    }
    __syncthreads();                                          // (double buffered: one barrier per tile row)
    const int p = tid >> 4, c = tid & 15;
    const cs_u32x4 v = *reinterpret_cast<const cs_u32x4*>(sl + p * 272 + c * 16);
    const int gy = y0 + py, gx = x0 + p;
    if (gy < H && gx < W)
      *reinterpret_cast<cs_u32x4*>(y + ((static_cast<size_t>(e) * H + gy) * W + gx) * 128 + c * 8) = v;
  }
}

// ---------------------------------------------------------------------------
// pvo_gru_glo_fused: glo[e,c] = mean over pixels of sigmoid(w(net) + b)[c] * net[c]   (ConvGRU's global context,
// VO_Module/droid_slam/modules/gru.py:22-24) with the 1x1 convolution `w` (128 -> 128) done in the kernel.
//   Before: a 13 us MIOpen 1x1 convolution writing wn (28 MB) + a 20 us reduction kernel reading wn and net.
//   Here: net is read once; 64-pixel tiles go through LDS, wave w owns output channels [32w, 32w+32) with its 8 weight
//   fragments in registers (v_mfma_f32_16x16x32), the sigmoid gate and the pixel sum stay in registers in the
//   accumulator layout (lane = channel), and each workgroup writes the partial means of its 256-pixel chunk: glo_part [E][chunks][128]; the
//   consumer (a [E, chunks*128] x [chunks*128, 384] GEMM against row-tiled gate weights) sums the chunks for free.
// ---------------------------------------------------------------------------
constexpr int kGloTile = 64, kGloStride = 272;

template <typename T>
__global__ __launch_bounds__(256) void gru_glo_mfma_kernel(const uint16_t* __restrict__ net, const uint16_t* __restrict__ ww,
                                                           const float* __restrict__ bias, float* __restrict__ glo,
                                                           int HW, int chunk) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[kGloTile * kGloStride];
  const int e = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  cs_u32x4 bf[4][2];
#pragma unroll
  for (int kc = 0; kc < 4; ++kc)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      bf[kc][nt] = *reinterpret_cast<const cs_u32x4*>(ww + static_cast<size_t>(wave * 32 + nt * 16 + li) * 128 + kc * 32 + lk * 8);
  const float b0 = bias ? bias[wave * 32 + li] : 0.0f, b1 = bias ? bias[wave * 32 + 16 + li] : 0.0f;
  float s0 = 0.0f, s1 = 0.0f;
  const int p_begin = blockIdx.x * chunk, p_end = min(p_begin + chunk, HW);
  const uint16_t* ne = net + static_cast<size_t>(e) * HW * 128;
  // software pipeline: the next tile's 16 KB are in flight (registers) while this tile is multiplied
  cs_u32x4 pre[4];
  auto fetch = [&](int p0) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {                      // 64 px x 16 chunks of 16 B
      const int id = tid + 256 * it, px = id >> 4, c = id & 15;
      pre[it] = cs_u32x4{0u, 0u, 0u, 0u};
      if (p0 + px < p_end) pre[it] = *reinterpret_cast<const cs_u32x4*>(ne + static_cast<size_t>(p0 + px) * 128 + c * 8);
    }
  };
  fetch(p_begin);
  for (int p0 = p_begin; p0 < p_end; p0 += kGloTile) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int id = tid + 256 * it, px = id >> 4, c = id & 15;
      *reinterpret_cast<cs_u32x4*>(tile + px * kGloStride + c * 16) = pre[it];
    }
    __syncthreads();
    if (p0 + kGloTile < p_end) fetch(p0 + kGloTile);
    cs_u32x4 afr[4][4];                                    // the tile's 16 A fragments, requested up front
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
        afr[g][kc] = *reinterpret_cast<const cs_u32x4*>(tile + (g * 16 + li) * kGloStride + kc * 64 + lk * 16);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      cs_v4f d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        d0 = cs_mfma<T>(afr[g][kc], bf[kc][0], d0);
        d1 = cs_mfma<T>(afr[g][kc], bf[kc][1], d1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {                       // D rows lk*4 + r = pixels, column li = channel
        const unsigned char* row = tile + (g * 16 + lk * 4 + r) * kGloStride + (wave * 32 + li) * 2;
        const float n0 = Elem<T>::to_f32(*reinterpret_cast<const typename Elem<T>::store_t*>(row));
        const float n1 = Elem<T>::to_f32(*reinterpret_cast<const typename Elem<T>::store_t*>(row + 32));
        s0 += n0 / (1.0f + __expf(-(d0[r] + b0)));        // padded pixels carry net = 0
        s1 += n1 / (1.0f + __expf(-(d1[r] + b1)));
      }
    }
  }
  s0 += __shfl_xor(s0, 16, 64); s0 += __shfl_xor(s0, 32, 64);
  s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
  if (lk == 0) {                                          // this workgroup's partial mean: no zero fill, no atomics
    const float inv = 1.0f / static_cast<float>(HW);
    float* o = glo + (static_cast<size_t>(e) * gridDim.x + blockIdx.x) * 128 + wave * 32;
    o[li] = s0 * inv;
    o[16 + li] = s1 * inv;
  }
}

// ---------------------------------------------------------------------------
// pvo_conv3x3_c128: y = act(conv3x3(x, w) + bias), x [E,H,W,128] -> y [E,H,W,Cout], zero padding 1, Cout in {64,128,256,512}.
//   The update operator's 128-input 3x3 convolutions (corr_encoder[2], flow_encoder[2], GraphAgg.conv1; droid_net.py:
//   79-95,172-180) run at 0.3-0.5 PFLOP/s in MIOpen/CK at these shapes (67 us for 128 -> 128 over 36 x 48 x 64 pixels).
//   Workgroup = 8x16 pixel tile x 128 output channels (64 when Cout = 64); the 10x18 halo (46 KB) is staged once in LDS
//   with a 272-byte row stride; wave w owns NT 16-channel column tiles and, tap by tap, keeps that tap's 4*NT weight
//   fragments in registers (the next tap's are in flight) while it sweeps the 8 tile rows: 8*4*NT v_mfma_f32_16x16x32 per
//   tap on 8*NT independent accumulators, one 16-byte LDS read per NT MFMAs.  Results leave through an LDS slab as whole
//   pixel rows.  Weights arrive as [9 taps][Cout][128 input channels] 16-bit.
// ---------------------------------------------------------------------------
constexpr int kC3Halo = (kTH + 2) * (kTW + 2), kC3Stride = 272;

template <typename T, int NT>
__global__ __launch_bounds__(256) void conv3x3_c128_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ wt,
                                                           const float* __restrict__ bias, uint16_t* __restrict__ y,
                                                           int H, int W, int Cout, int relu) {
  extern __shared__ __attribute__((aligned(16))) unsigned char c3s[];          // [180][272 B]; later the output slab
  constexpr int kCoutWG = 64 * NT;                                            // output channels per workgroup
  const int ntx = (W + kTW - 1) / kTW;
  const int cg = blockIdx.x / ntx, tx_ = blockIdx.x - cg * ntx;                // output-channel group, tile column
  const int e = blockIdx.z, y0 = blockIdx.y * kTH, x0 = tx_ * kTW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int co0 = cg * kCoutWG + wave * (16 * NT);                            // first output channel of this wave

  // stage the halo tile
  const uint16_t* xe = x + static_cast<size_t>(e) * H * W * 128;
  {
    const int ch = tid & 15;
#pragma unroll
    for (int it = 0; it < (kC3Halo * 16 + 255) / 256; ++it) {
      const int pos = (tid >> 4) + 16 * it;
      if (pos < kC3Halo) {
        const int hy = y0 - 1 + pos / (kTW + 2), hx = x0 - 1 + pos % (kTW + 2);
        cs_u32x4 v = {0u, 0u, 0u, 0u};
        if (hy >= 0 && hy < H && hx >= 0 && hx < W) v = *reinterpret_cast<const cs_u32x4*>(xe + (static_cast<size_t>(hy) * W + hx) * 128 + ch * 8);
        *reinterpret_cast<cs_u32x4*>(c3s + pos * kC3Stride + ch * 16) = v;
      }
    }
  }
  // weight fragments: B[k = cin][n = cout] of tap t; lane (li, lk) holds column li of tile nt, input channels kc*32 + lk*8 ..+8
  const uint16_t* wl = wt + (static_cast<size_t>(co0 + li)) * 128 + lk * 8;
  const size_t tap_stride = static_cast<size_t>(Cout) * 128;
  cs_u32x4 bcur[4][NT], bnxt[4][NT];
#pragma unroll
  for (int kc = 0; kc < 4; ++kc)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bcur[kc][nt] = *reinterpret_cast<const cs_u32x4*>(wl + static_cast<size_t>(nt) * 16 * 128 + kc * 32);
  cs_v4f acc[kTH][NT];
#pragma unroll
  for (int py = 0; py < kTH; ++py)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[py][nt] = cs_v4f{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

#pragma unroll 1
  for (int t = 0; t < 9; ++t) {                              // (not unrolled: all 9 taps' fragments would be hoisted: 256 VGPRs)
    if (t < 8) {
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          bnxt[kc][nt] = *reinterpret_cast<const cs_u32x4*>(wl + (t + 1) * tap_stride + static_cast<size_t>(nt) * 16 * 128 + kc * 32);
    }
    const unsigned char* tp = c3s + ((t / 3) * (kTW + 2) + (t % 3) + li) * kC3Stride + lk * 16;
#pragma unroll
    for (int py = 0; py < kTH; ++py) {
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        const cs_u32x4 a = *reinterpret_cast<const cs_u32x4*>(tp + py * (kTW + 2) * kC3Stride + kc * 64);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[py][nt] = cs_mfma<T>(a, bcur[kc][nt], acc[py][nt]);
      }
    }
    if (t < 8) {
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bcur[kc][nt] = bnxt[kc][nt];
    }
  }
  __syncthreads();                                           // the halo tile is consumed: its LDS becomes the output slab
  // slab [8 rows][16 px][kCoutWG channels] with a (kCoutWG*2 + 16)-byte pixel stride
  constexpr int kOutStride = kCoutWG * 2 + 16;
  float bb[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bb[nt] = bias ? bias[co0 + nt * 16 + li] : 0.0f;
#pragma unroll
  for (int py = 0; py < kTH; ++py)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {                           // D rows lk*4 + r = pixels, column li = channel
        float v = acc[py][nt][r] + bb[nt];
        if (relu) v = fmaxf(v, 0.0f);
        *reinterpret_cast<uint16_t*>(c3s + (py * 16 + lk * 4 + r) * kOutStride + (wave * 16 * NT + nt * 16 + li) * 2) =
            static_cast<uint16_t>(cs_bits<T>(v));
      }
  __syncthreads();
  constexpr int kChunks = kCoutWG / 8;                        // 16-byte chunks per pixel
  for (int id = tid; id < kTH * 16 * kChunks; id += 256) {
    const int p = id / kChunks, c = id - p * kChunks;
    const int gy = y0 + (p >> 4), gx = x0 + (p & 15);
    if (gy < H && gx < W)
      *reinterpret_cast<cs_u32x4*>(y + ((static_cast<size_t>(e) * H + gy) * W + gx) * Cout + cg * kCoutWG + c * 8) =
          *reinterpret_cast<const cs_u32x4*>(c3s + p * kOutStride + c * 16);
  }
}

}  // namespace

extern "C" int pvo_conv7x7_c8(const void* x, const void* w_taps, const float* bias, void* y,
                              int E, int H, int W, int dtype, void* stream) {
  if (E < 0 || H < 0 || W < 0) return PVO_EINVAL;
  if (E == 0 || H == 0 || W == 0) return PVO_OK;
  if (!x || !w_taps || !bias || !y || E > 65535) return PVO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_taps) | reinterpret_cast<uintptr_t>(y)) & 15) return PVO_EINVAL;
  hipStream_t st = pvo_stream(stream);
  dim3 grid((W + kTW - 1) / kTW, (H + kTH7 - 1) / kTH7, E);
  if (dtype == PVO_F16)
    hipLaunchKernelGGL(conv7x7_c8_kernel<pvo_half>, grid, dim3(256), 0, st, static_cast<const uint16_t*>(x),
                       static_cast<const uint16_t*>(w_taps), bias, static_cast<uint16_t*>(y), H, W);
  else if (dtype == PVO_BF16)
    hipLaunchKernelGGL(conv7x7_c8_kernel<pvo_bf16>, grid, dim3(256), 0, st, static_cast<const uint16_t*>(x),
                       static_cast<const uint16_t*>(w_taps), bias, static_cast<uint16_t*>(y), H, W);
  else
    return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_gru_glo_chunks(int HW) { return HW <= 0 ? 0 : (HW + 255) / 256; }

extern "C" int pvo_gru_glo_fused(const void* net, const void* w_weight, const float* w_bias, float* glo_part,
                                 int E, int HW, int dtype, void* stream) {
  if (E < 0 || HW < 0) return PVO_EINVAL;
  if (E == 0 || HW == 0) return PVO_OK;
  if (!net || !w_weight || !glo_part || E > 65535) return PVO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(net) | reinterpret_cast<uintptr_t>(w_weight)) & 15) return PVO_EINVAL;
  hipStream_t st = pvo_stream(stream);
  const int chunk = 256;
  dim3 grid(pvo_gru_glo_chunks(HW), E);
  if (dtype == PVO_F16)
    hipLaunchKernelGGL(gru_glo_mfma_kernel<pvo_half>, grid, dim3(256), 0, st, static_cast<const uint16_t*>(net),
                       static_cast<const uint16_t*>(w_weight), w_bias, glo_part, HW, chunk);
  else if (dtype == PVO_BF16)
    hipLaunchKernelGGL(gru_glo_mfma_kernel<pvo_bf16>, grid, dim3(256), 0, st, static_cast<const uint16_t*>(net),
                       static_cast<const uint16_t*>(w_weight), w_bias, glo_part, HW, chunk);
  else
    return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_conv3x3_c128(const void* x, const void* w_taps, const float* bias, void* y,
                                int E, int H, int W, int Cout, int relu, int dtype, void* stream) {
  if (E < 0 || H < 0 || W < 0) return PVO_EINVAL;
  if (Cout != 64 && (Cout <= 0 || (Cout & 127))) return PVO_EUNSUPPORTED;
  if (E == 0 || H == 0 || W == 0) return PVO_OK;
  if (!x || !w_taps || !y || E > 65535) return PVO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_taps) | reinterpret_cast<uintptr_t>(y)) & 15) return PVO_EINVAL;
  hipStream_t st = pvo_stream(stream);
  const int ntx = (W + kTW - 1) / kTW;
  const size_t lds = static_cast<size_t>(kC3Halo) * kC3Stride;      // 48960 B >= the output slab (128 px x 272 B)
  const uint16_t* xp = static_cast<const uint16_t*>(x);
  const uint16_t* wp = static_cast<const uint16_t*>(w_taps);
  uint16_t* yp = static_cast<uint16_t*>(y);
  if (Cout == 64) {
    dim3 grid(ntx, (H + kTH - 1) / kTH, E);
    if (dtype == PVO_F16) hipLaunchKernelGGL((conv3x3_c128_kernel<pvo_half, 1>), grid, dim3(256), lds, st, xp, wp, bias, yp, H, W, Cout, relu);
    else if (dtype == PVO_BF16) hipLaunchKernelGGL((conv3x3_c128_kernel<pvo_bf16, 1>), grid, dim3(256), lds, st, xp, wp, bias, yp, H, W, Cout, relu);
    else return PVO_EUNSUPPORTED;
  } else {
    dim3 grid(ntx * (Cout / 128), (H + kTH - 1) / kTH, E);
    if (dtype == PVO_F16) hipLaunchKernelGGL((conv3x3_c128_kernel<pvo_half, 2>), grid, dim3(256), lds, st, xp, wp, bias, yp, H, W, Cout, relu);
    else if (dtype == PVO_BF16) hipLaunchKernelGGL((conv3x3_c128_kernel<pvo_bf16, 2>), grid, dim3(256), lds, st, xp, wp, bias, yp, H, W, Cout, relu);
    else return PVO_EUNSUPPORTED;
  }
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}
