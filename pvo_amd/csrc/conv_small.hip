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
#include <type_traits>
#include "glo_tile.h"
#include <stdlib.h>

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

template <typename T> __device__ __forceinline__ float cs_val(uint32_t b);
template <> __device__ __forceinline__ float cs_val<pvo_half>(uint32_t b) {
  union { uint16_t u; _Float16 h; } c; c.u = static_cast<uint16_t>(b); return static_cast<float>(c.h);
}
template <> __device__ __forceinline__ float cs_val<pvo_bf16>(uint32_t b) { return pvo_bf16_to_f32(static_cast<uint16_t>(b)); }
template <typename T> __device__ __forceinline__ void cs_unpack8(cs_u32x4 v, float f[8]) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) { f[2 * k] = cs_val<T>(w[k] & 0xffffu); f[2 * k + 1] = cs_val<T>(w[k] >> 16); }
}
template <typename T> __device__ __forceinline__ cs_u32x4 cs_pack8(const float f[8]) {
  cs_u32x4 v;
  v.x = cs_bits<T>(f[0]) | (cs_bits<T>(f[1]) << 16); v.y = cs_bits<T>(f[2]) | (cs_bits<T>(f[3]) << 16);
  v.z = cs_bits<T>(f[4]) | (cs_bits<T>(f[5]) << 16); v.w = cs_bits<T>(f[6]) | (cs_bits<T>(f[7]) << 16);
  return v;
}

constexpr int kTH = 8, kTW = 16, kR = 3;
#ifndef PVO_CONV7_ROWS                                      // (experiment hook, tools/variant.py: 8 / 4 rows measured, no change in the update)
#define PVO_CONV7_ROWS 16
#endif
constexpr int kTH7 = PVO_CONV7_ROWS;                          // the 7x7 kernel's tile is 16 rows high: weight fragments are
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
template <typename T>
__global__ __launch_bounds__(256) void gru_glo_mfma_kernel(const uint16_t* __restrict__ net, const uint16_t* __restrict__ ww,
                                                           const float* __restrict__ bias, float* __restrict__ glo,
                                                           int HW, int chunk) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[glt::kTileBytes];
  glt::glo_partial_means<T>(tile, net, ww, bias, glo, HW, chunk, blockIdx.y, blockIdx.x, gridDim.x);      // (glo_tile.h)
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
                                                           int H, int W, int Cout, int relu, int ystride, int yoff) {
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
      *reinterpret_cast<cs_u32x4*>(y + ((static_cast<size_t>(e) * H + gy) * W + gx) * ystride + yoff + cg * kCoutWG + c * 8) =
          *reinterpret_cast<const cs_u32x4*>(c3s + p * kOutStride + c * 16);
  }
}

// ---------------------------------------------------------------------------
// pvo_conv3x3: y = act(conv3x3(x, w) + bias) for wide layers, x [E,H,W,Cin] -> y [E,H,W,Cout], Cin % 32 == 0,
// Cout % 128 == 0 - the GRU gate / candidate convolutions (320 -> 256, 320 -> 128), the heads' first stage (128 -> 512),
// corr_encoder[2], GraphAgg.conv1 and the static-input terms (MIOpen/CK ran these at 0.69-0.74 PFLOP/s).
//   Implicit GEMM on v_mfma_f32_32x32x16: workgroup = 16x16 pixel tile (M = 256) x 128 output channels, 4 waves as 2 x 2,
//   each wave 128 pixels (8 rows x 16 columns) x 64 channels = 8 accumulator tiles.  K runs over (32-channel chunk, tap).
//   A: the chunk's 18x18 halo sits in LDS (80-byte position stride, 20-position row pitch), double buffered, staged once
//      per chunk and shared by its 9 taps; one barrier per CHUNK.  An M-tile is 8 rows x 4 columns of pixels: with the
//      80-byte stride and the 20-position pitch every ds_read_b128 lane group hits 64 distinct banks (the 2-row x 16-column
//      M-tile used before had SQ_LDS_BANK_CONFLICT = 47 % of SQ_LDS_IDX_ACTIVE, profiles/r02_kernel_counters.json).
//   B: the filter arrives in MFMA-FRAGMENT order, [Cout/128][chunk][tap][wave column wn][nt][ks][lane][8] (the host
//      arranges it once), and is streamed global -> registers: one coalesced 1 KB load per fragment, requested two
//      steps ahead into one of three rotating register sets.  No LDS, ds_write, ds_read or barrier for the filter, and
//      with the 9 taps unrolled every fragment address is base + immediate.  (The first version staged tap-major filter
//      slabs through LDS with a barrier per (chunk, tap): ~160 non-MFMA instructions per 16-MFMA step, 10-14 % slower.)
// ---------------------------------------------------------------------------
typedef float cs_v16f __attribute__((ext_vector_type(16)));
template <typename T> __device__ __forceinline__ cs_v16f cs_mfma32(cs_u32x4 a, cs_u32x4 b, cs_v16f c);
template <> __device__ __forceinline__ cs_v16f cs_mfma32<pvo_half>(cs_u32x4 a, cs_u32x4 b, cs_v16f c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cs_v8h, a), __builtin_bit_cast(cs_v8h, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ cs_v16f cs_mfma32<pvo_bf16>(cs_u32x4 a, cs_u32x4 b, cs_v16f c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cs_v8b, a), __builtin_bit_cast(cs_v8b, b), c, 0, 0, 0);
}

// Optional fused ConvGRU epilogues (VO_Module/droid_slam/modules/gru.py:26-31) and a segmented input:
//   mode 1 (gates, Cout = 256): channel group 0 -> y  = Z  = sigmoid(acc + g[e, c] + P[row, c])                [rows,128]
//                               channel group 1 -> y2 = RN = sigmoid(acc + g[e,128+c] + P[row,128+c]) * net    [rows,128]
//   mode 2 (candidate, Cout = 128):               y  = (1 - Z) * net + Z * tanh(acc + g[e,256+c] + P[row, c])  [rows,128]
//   mode 3 (output heads, droid_net.py:184-210: four heads of Conv3x3(128 -> 128) + ReLU + Conv3x3(128 -> 2); Cout = 512,
//       channel group cg = head): the hidden activations relu(acc + bias) never leave the workgroup.  The second
//       convolution is linear, so its result at pixel p is the sum over the nine taps t of W2[t] . hidden(p + t): here
//       every pixel q multiplies its 128 hidden channels by all nine tap filters at once - a [pixels x 128] x [128 x 18]
//       product on the matrix cores, straight from the fp16 slab of the epilogue - and stores z[q][head][2 t + o]; a
//       small gather kernel (pvo_heads_gather) adds the nine neighbours.  Instead of writing and re-reading the
//       [rows, 512] hidden tensor (2 x 113 MB at S-B) the heads move 32 MB, and the 3x3 halo exchange is nine 8-byte reads.
//   segmented input (nseg > 0 replaces x): the input channels are the concatenation of up to three tensors, each with its
//       own pixel stride - the ConvGRU reads [net | encoder features] (gates) and [r * net | encoder features]
//       (candidate) straight from the tensors their producers wrote; no concatenated copy is assembled.
struct BigEpi {
  int mode;
  const float* g;            // [E,384] f32: context of z | r | q (biases folded in)
  const uint16_t* P;         // precomputed static-input term, [rows,256] (mode 1) or [rows,128] (mode 2)
  const int* p_slots;        // non-null: P is a slot pool [slots,H,W,C] and edge e reads slot p_slots[e] (the volume pool's slot)
  const uint16_t* net;       // [rows,128]
  const uint16_t* Z;         // [rows,128] (mode 2)
  uint16_t* y2;              // [rows,128] (mode 1)
  int nseg;
  const uint16_t* seg_p[3]; int seg_stride[3]; int seg_chunks[3];
  const uint16_t* w2f;       // mode 3: second-stage filter of the heads as MFMA B fragments, [Cout/128][8 k-steps][64 lanes][8]
  float* z;                  // mode 3: [rows][Cout/128][18] f32 tap contributions
};

constexpr int kBT = 16;                                   // 16 x 16 pixel tile
constexpr int kBHalo = (kBT + 2) * (kBT + 2);             // 324 halo positions
constexpr int kBPitch = 20;                               // LDS positions per halo row (18 used)
constexpr int kBStride = 80;                              // bytes per position of a 32-channel chunk
constexpr int kBA = 420 * kBStride;                       // 18 x 20 positions + 60 dummy ones (every thread parks 6 pieces, no exec branches)

// pixel (row, column) inside a wave's 8 x 16 block of accumulator row `m` (0..31) of M-tile `mt`
__device__ __forceinline__ int big_pix(int mt, int m) { return (m >> 2) * 16 + 4 * mt + (m & 3); }

// tools/conv_timeline.py builds this file with -DPVO_CONV_PROBE: every workgroup then records its shader-clock stamps
// (start, main loop entered, main loop left, end) and the compute unit it ran on; the shipped library has none of it.
#ifdef PVO_CONV_PROBE
__device__ unsigned long long* g_conv_probe = nullptr;
#define CONV_WG ((static_cast<size_t>(blockIdx.z) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x)
#define CONV_PROBE(slot)                                                                                          \
  do {                                                                                                            \
    if (g_conv_probe && threadIdx.x == 0) g_conv_probe[CONV_WG * 8 + (slot)] = __builtin_readcyclecounter();      \
  } while (0)
#else
#define CONV_PROBE(slot)
#endif

template <typename T, bool HEADS>
__global__ __launch_bounds__(256, 2) void conv3x3_big_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ wt,
                                                          const float* __restrict__ bias, uint16_t* __restrict__ y,
                                                          int H, int W, int Cin, int Cout, int relu, int ystride, int yoff, BigEpi ep) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bs[];      // halo[2]; later the output slab
  unsigned char* As = bs;
  CONV_PROBE(0);
#ifdef PVO_CONV_PROBE
  if (g_conv_probe && threadIdx.x == 0) {
    g_conv_probe[CONV_WG * 8 + 4] = (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg(20 | (31 << 11))) << 32) | __builtin_amdgcn_s_getreg(4 | (31 << 11));
    g_conv_probe[CONV_WG * 8 + 5] = wall_clock64();
  }
#endif
  const int ntx = (W + kBT - 1) / kBT;
  const int cg = blockIdx.x / ntx, tx_ = blockIdx.x - cg * ntx;
  const int e = blockIdx.z, y0 = blockIdx.y * kBT, x0 = tx_ * kBT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kg = lane >> 5;
  const int nC = Cin >> 5;                                // 32-channel chunks
  const uint16_t* xe = x + static_cast<size_t>(e) * H * W * Cin;

  // global -> register -> LDS staging of the halo chunk needed next (6 x 16 B per thread)
  cs_u32x4 ra[3];                                         // (three pieces at a time: registers)
  int apix[6];                                            // pixel index of this thread's halo pieces inside the image, -1 = zero
  int lpos[6];                                            // their LDS byte offsets
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int id = tid + 256 * it;                        // piece = (position, 16-byte quarter of the 64-byte chunk row)
    const int pos = id >> 2;
    apix[it] = -1;
    lpos[it] = (360 + (pos - kBHalo)) * kBStride + (id & 3) * 16;      // dummy slot
    if (pos < kBHalo) {
      const int hr = pos / (kBT + 2), hc = pos % (kBT + 2);
      const int hy = y0 - 1 + hr, hx = x0 - 1 + hc;
      if (hy >= 0 && hy < H && hx >= 0 && hx < W) apix[it] = hy * W + hx;
      lpos[it] = (hr * kBPitch + hc) * kBStride + (id & 3) * 16;
    }
  }
  const size_t img = static_cast<size_t>(e) * H * W;
  // (loads and stores are unconditional - out-of-image pieces read a clamped address and are zeroed by a select when they
  // are parked: exec-masked branches around VMEM operations make the compiler wait vmcnt(0) at every join)
  const uint16_t* a_src = xe; int a_stride = Cin, a_coff = 0;
  auto select_a = [&](int cc) {                                    // which tensor / channel offset chunk cc comes from
    a_src = xe; a_stride = Cin; a_coff = cc * 32;
    if (ep.nseg > 0) {                                              // uniform walk over at most three segments
      int sgi = 0, c0 = cc;
      while (sgi + 1 < ep.nseg && c0 >= ep.seg_chunks[sgi]) { c0 -= ep.seg_chunks[sgi]; ++sgi; }
      a_stride = ep.seg_stride[sgi];
      a_src = ep.seg_p[sgi] + img * a_stride;
      a_coff = c0 * 32;
    }
  };
  auto fetch_a = [&](int half) {                                   // pieces 3 half .. 3 half + 2
#pragma unroll
    for (int k = 0; k < 3; ++k)
      ra[k] = *reinterpret_cast<const cs_u32x4*>(a_src + static_cast<size_t>(max(apix[3 * half + k], 0)) * a_stride + a_coff + (tid & 3) * 8);
  };
  auto store_a = [&](int buf, int half) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      cs_u32x4 v = ra[k];
      if (apix[3 * half + k] < 0) v = cs_u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<cs_u32x4*>(As + buf * kBA + lpos[3 * half + k]) = v;
    }
  };

  // accumulators start from the per-(edge, channel) gate context of the fused GRU epilogues (column li of a tile = one
  // channel): one addition per value less in the epilogue, which is instruction-bound
  float ginit[2] = {0.0f, 0.0f};
  if (!HEADS && ep.mode != 0) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      ginit[nt] = ep.g[static_cast<size_t>(e) * 384 + (ep.mode == 1 ? cg * 128 : 256) + wn * 64 + nt * 32 + li];
  }
  cs_v16f acc[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = ginit[nt];

  const int S = nC * 9;                                   // steps; S >= 9
  {
    // this wave's fragments of step s: wf + ((cg * S + s) * 8 + wn * 4 + nt * 2 + ks) * 512 + lane * 8   (16-bit elements)
    const uint16_t* wf = wt + (static_cast<size_t>(cg) * S * 8 + wn * 4) * 512 + lane * 8;
    cs_u32x4 bset[3][4];                                    // three rotating sets of {nt0 ks0, nt0 ks1, nt1 ks0, nt1 ks1}
    auto fetch_bf = [&](cs_u32x4 (&r)[4], int s) {          // (clamped, unconditional: exact s_waitcnt vmcnt counts)
      const uint16_t* p = wf + static_cast<size_t>(min(s, S - 1)) * 4096;
#pragma unroll
      for (int f = 0; f < 4; ++f) r[f] = *reinterpret_cast<const cs_u32x4*>(p + f * 512);
    };
    select_a(0);
    {
      // prologue: all six halo pieces of chunk 0 and the first two filter sets are requested together (the accumulators do
      // not exist yet, so the registers are there): ONE exposed memory latency in front of the loop instead of two
      cs_u32x4 r6[6];
#pragma unroll
      for (int k = 0; k < 6; ++k)
        r6[k] = *reinterpret_cast<const cs_u32x4*>(a_src + static_cast<size_t>(max(apix[k], 0)) * a_stride + a_coff + (tid & 3) * 8);
      fetch_bf(bset[0], 0); fetch_bf(bset[1], 1);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        cs_u32x4 v = r6[k];
        if (apix[k] < 0) v = cs_u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<cs_u32x4*>(As + lpos[k]) = v;
      }
    }
    __syncthreads();
    CONV_PROBE(1);
    // this lane's A rows: M-tile mt of wave-row wm = tile rows 8 wm + (li >> 2), columns 4 mt + (li & 3)
    const unsigned char* Abase = As + ((8 * wm + (li >> 2)) * kBPitch + (li & 3)) * kBStride + kg * 16;
    // Software pipeline at half-step (k = 16) granularity, pinned with sched_barrier: the four A fragments of the NEXT
    // half-step are requested before the eight MFMAs of the current one, so a fragment has 8 MFMAs (256 matrix-pipe
    // cycles) to arrive.  Left to itself the compiler sinks each ds_read next to its first use ("read, wait lgkmcnt(0),
    // two MFMAs" - the LDS latency exposed eight times per step; SQ_WAIT_ANY was 40 % of the wave cycles).
    // The main loop exists twice: NMT = 4 for a full tile, NMT = 2 for a tile of which only the first two 4-pixel column groups
    // (M-tiles) lie inside the image - the right-most tile column of a map whose width is not a multiple of 16 (101 = 6 x 16 + 5: a
    // seventh of all tiles at the reference driver's 30 x 101).  Such a workgroup issues half the MFMAs; its waves wait at the
    // same barriers, and the matrix cores they leave idle go to the second workgroup of the CU.
    auto main_loop = [&](auto nmt_c) {
    constexpr int NMT = decltype(nmt_c)::value;
    auto read_half = [&](cs_u32x4 (&a)[4], const unsigned char* Ac, int toff, int ks) {
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) a[mt] = *reinterpret_cast<const cs_u32x4*>(Ac + toff + mt * 4 * kBStride + ks * 32);
    };
#pragma unroll 1
    for (int cc = 0; cc < nC; ++cc) {
      const unsigned char* Ac = Abase + (cc & 1) * kBA;
      select_a(min(cc + 1, nC - 1));                        // next chunk's halo: two halves, each in flight for three taps
      fetch_a(0);
      cs_u32x4 a0[4], a1[4];
      read_half(a0, Ac, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int toff = ((t / 3) * kBPitch + (t % 3)) * kBStride;         // compile-time: ds_read immediates
        fetch_bf(bset[(t + 2) % 3], cc * 9 + t + 2);
        read_half(a1, Ac, toff, 1);
        const cs_u32x4 (&bf)[4] = bset[t % 3];
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = cs_mfma32<T>(a0[mt], bf[nt * 2], acc[mt][nt]);
        // issue order inside the half-step: one LDS read / one global load behind each MFMA (an in-order wave hides
        // about five single-issue instructions in the 32 cycles an MFMA holds the pipe; clumped between the groups of
        // eight they are exposed)
#pragma unroll
        for (int i = 0; i < NMT; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
        for (int i = 0; i < NMT; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x20, 4 / NMT, 0); }
        __builtin_amdgcn_sched_barrier(0);
        if (t < 8) read_half(a0, Ac, ((((t + 1) / 3) * kBPitch) + ((t + 1) % 3)) * kBStride, 0);
        if (t == 3) { store_a((cc + 1) & 1, 0); fetch_a(1); }   // (the other halo buffer was last read before the previous barrier)
        if (t == 7) store_a((cc + 1) & 1, 1);
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = cs_mfma32<T>(a1[mt], bf[nt * 2 + 1], acc[mt][nt]);
        if (NMT == 4) {
          if (t < 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
          }
          if (t == 3 || t == 7) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
          }
          if (t == 3) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x20, 3, 0); }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
    };
    if (x0 + 8 >= W) main_loop(std::integral_constant<int, 2>{});
    else main_loop(std::integral_constant<int, 4>{});
  }
  CONV_PROBE(2);

  if (!HEADS && ep.mode != 0) {
    // fused ConvGRU epilogue: pre-activations cross the workgroup through an fp32 slab [128 px][128 ch] (528-byte pixel
    // stride, 67.6 KB of the 72 KB), then every thread finishes 8 channels of a pixel with coalesced 16-byte accesses
    float* slab = reinterpret_cast<float*>(bs);
    // every thread finishes channels 8c .. 8c+7 of pixels m = (tid >> 4) + 16 k, k = 0..7, of each half; the per-pixel
    // operands (P, net, Z) are requested four pixels at a time, the first four BEFORE the accumulators cross the slab:
    // the probe build showed 32.7k cycles (19 % of a workgroup's life) in this epilogue
    // when each pixel's loads were waited for one after the other
    const int c = tid & 15, m0 = tid >> 4;
    const int pstride = ep.mode == 1 ? 256 : 128;
    const uint16_t* gate_p = ep.P + (ep.mode == 1 ? cg * 128 : 0) + c * 8;
    uint16_t* const dst = (ep.mode == 1 && cg != 0) ? ep.y2 : y;
    const int ep_ = ep.p_slots ? ep.p_slots[e] : e;         // image of the static term (workgroup-uniform)
    cs_u32x4 pv[4], nv[4], zv[4];
    auto request = [&](int half, int b) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int m = m0 + 16 * (4 * b + k);
        const int gy = y0 + 8 * half + (m >> 4), gx = x0 + (m & 15);
        const bool in = gy < H && gx < W;
        const size_t rr = in ? (static_cast<size_t>(e) * H + gy) * W + gx : 0;      // a valid address either way; the value is dropped
        const size_t rp = in ? (static_cast<size_t>(ep_) * H + gy) * W + gx : 0;
        pv[k] = *reinterpret_cast<const cs_u32x4*>(gate_p + rp * pstride);
        nv[k] = *reinterpret_cast<const cs_u32x4*>(ep.net + rr * 128 + c * 8);
        if (ep.mode == 2) zv[k] = *reinterpret_cast<const cs_u32x4*>(ep.Z + rr * 128 + c * 8);
      }
    };
    auto finish = [&](int half, int b) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int m = m0 + 16 * (4 * b + k);
        const int gy = y0 + 8 * half + (m >> 4), gx = x0 + (m & 15);
        float a[8], pp[8], nn[8], o[8];
        const float4 a0 = *reinterpret_cast<const float4*>(slab + m * 132 + c * 8), a1 = *reinterpret_cast<const float4*>(slab + m * 132 + c * 8 + 4);
        a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
        cs_unpack8<T>(pv[k], pp);
        cs_unpack8<T>(nv[k], nn);
        if (ep.mode == 1) {
          // sigmoid = 1 / (1 + 2^(-x log2 e)) on the raw v_exp_f32 / v_rcp_f32 (1 ulp each; exp -> inf gives 0, exp -> 0 gives 1:
          // the library exp's denormal-range rescaling - two selects, an add and a multiply per value - buys nothing here)
#pragma unroll
          for (int q = 0; q < 8; ++q) o[q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * (a[q] + pp[q])));
          if (cg != 0) {                                    // (workgroup-uniform: the r half of the gates leaves as r * net)
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] *= nn[q];
          }
        } else {
          float zz[8];
          cs_unpack8<T>(zv[k], zz);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            // tanh(x) = 1 - 2 / (1 + exp(2x)): exact limits at both ends (exp -> inf gives 1, exp -> 0 gives -1), ~1e-6
            // absolute error in between - the result is rounded to 16 bits
            const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * (a[q] + pp[q])));
            o[q] = (1.0f - zz[q]) * nn[q] + zz[q] * th;
          }
        }
        if (gy < H && gx < W) *reinterpret_cast<cs_u32x4*>(dst + ((static_cast<size_t>(e) * H + gy) * W + gx) * 128 + c * 8) = cs_pack8<T>(o);
      }
    };
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      request(half, 0);
      if (wm == half) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int m = big_pix(mt, (r & 3) + 8 * (r >> 2) + 4 * kg);
              slab[m * 132 + wn * 64 + nt * 32 + li] = acc[mt][nt][r];
            }
      }
      __syncthreads();
      if (half == 0) CONV_PROBE(6);
      finish(half, 0);
      request(half, 1);
      finish(half, 1);
      __syncthreads();
      if (half == 0) CONV_PROBE(7);
    }
    CONV_PROBE(3);
    return;
  }
  // epilogue: two halves of 128 pixels through an LDS slab [128 px][128 ch] (272-byte pixel stride) -> whole-row stores
  // D layout of a 32x32 tile: column li = channel, rows (r & 3) + 8 * (r >> 2) + 4 * kg = pixel inside the M-tile
  float bb[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) bb[nt] = bias ? bias[cg * 128 + wn * 64 + nt * 32 + li] : 0.0f;
  unsigned char* w2s = bs + 36 * 1024;                      // mode 3: this head's second-stage fragments (8 KB) above the slab
  if (HEADS) {
    const cs_u32x4* src = reinterpret_cast<const cs_u32x4*>(ep.w2f + static_cast<size_t>(cg) * 4096) + tid;
    const cs_u32x4 f0 = src[0], f1 = src[256];
    *reinterpret_cast<cs_u32x4*>(w2s + tid * 16) = f0;        // (the main loop's last barrier is behind every wave: the halo is dead)
    *reinterpret_cast<cs_u32x4*>(w2s + 4096 + tid * 16) = f1;
  }
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    if (wm == half) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = big_pix(mt, (r & 3) + 8 * (r >> 2) + 4 * kg);         // pixel inside this wave-row's 8 x 16 block
            float v = acc[mt][nt][r] + bb[nt];
            if (relu) v = fmaxf(v, 0.0f);
            *reinterpret_cast<uint16_t*>(bs + m * 272 + (wn * 64 + nt * 32 + li) * 2) = static_cast<uint16_t>(cs_bits<T>(v));
          }
    }
    __syncthreads();
    if (HEADS) {
      // wave w takes rows 32 w .. 32 w + 31 of the slab: Z[32 px][32 (18 used)] = hidden[32 px][128] . W2'[128][32]
      cs_v16f zz;
#pragma unroll
      for (int r = 0; r < 16; ++r) zz[r] = 0.0f;
      const unsigned char* arow = bs + (32 * wave + li) * 272 + kg * 16;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        zz = cs_mfma32<T>(*reinterpret_cast<const cs_u32x4*>(arow + ks * 32), *reinterpret_cast<const cs_u32x4*>(w2s + (ks * 64 + lane) * 16), zz);
      if (li < 18) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kg;
          const int gy = y0 + 8 * half + (m >> 4), gx = x0 + (m & 15);
          if (gy < H && gx < W) ep.z[(((static_cast<size_t>(e) * H + gy) * W + gx) * (Cout >> 7) + cg) * 18 + li] = zz[r];
        }
      }
    } else {
      for (int id = tid; id < 128 * 16; id += 256) {
        const int m = id >> 4, c = id & 15;
        const int gy = y0 + 8 * half + (m >> 4), gx = x0 + (m & 15);
        if (gy < H && gx < W)
          *reinterpret_cast<cs_u32x4*>(y + ((static_cast<size_t>(e) * H + gy) * W + gx) * ystride + yoff + cg * 128 + c * 8) =
              *reinterpret_cast<const cs_u32x4*>(bs + m * 272 + c * 16);
      }
    }
    __syncthreads();
  }
  CONV_PROBE(3);
}

// ---------------------------------------------------------------------------
// pvo_flow_encoder: the update operator's flow encoder, Conv2d(8,128,7,padding=3) + ReLU -> Conv2d(128,64,3,padding=1) + ReLU
// (droid_net.py:176-180), in ONE kernel: the 128-channel intermediate never leaves LDS.  As two launches
// (conv7x7_c8_kernel, conv3x3_c128_kernel<.., 1>) the pair is the chain the gate convolution waits for once the ConvGRU's
// context is computed ahead - 29 + 40 us alone, 52 + 53 us beside the correlation lookup - and the intermediate costs a
// 28 MB write and a 40 MB halo read.  Workgroup = 8 x 14 output pixels: their 3x3 halo is 10 rows x 16 columns, exactly ten
// 16-pixel MFMA rows of the 7x7 stage (22 x 16 input halo, 5.6 KB), whose bias + ReLU results go to the LDS tile the 3x3
// stage reads (positions outside the image are the second convolution's zero padding: stored as zeros, not computed).  The
// 3x3 stage is conv3x3_c128_kernel's with a 16-position row pitch: columns 14 and 15 of every MFMA row read past the tile row
// and are discarded (12.5 % of its MFMAs).  The same MFMAs in the same order on the same 16-bit intermediate as the two
// kernels: bit-identical output.
// ---------------------------------------------------------------------------
constexpr int kFeOW = 14;                                      // output columns per workgroup
constexpr int kFeRows = kTH + 2;                               // rows of the intermediate tile
constexpr int kFeM7W = 16 + 2 * kR;                            // 22: pitch of the 7x7 stage's input halo
constexpr int kFeM7Pos = (kFeRows + 2 * kR) * kFeM7W;          // 16 x 22 positions
constexpr int kFeF1Pos = kFeRows * 16 + 2;                     // + 2: what the discarded columns of the last row read
constexpr int kFeLds = kFeF1Pos * kC3Stride + kFeM7Pos * 16;

template <typename T>
__global__ __launch_bounds__(256) void flow_encoder_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w7,
                                                           const float* __restrict__ b7, const uint16_t* __restrict__ w3,
                                                           const float* __restrict__ b3, uint16_t* __restrict__ y,
                                                           int H, int W, int ystride, int yoff) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fes[];          // [f1 tile 162 x 272 B | input halo 352 x 16 B]
  unsigned char* f1 = fes;
  unsigned char* halo = fes + kFeF1Pos * kC3Stride;
  const int e = blockIdx.z, y0 = blockIdx.y * kTH, x0 = blockIdx.x * kFeOW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;

  // ---- stage 1: 7x7, 8 -> 128 channels, on the 10 x 16 positions (y0 - 1 .., x0 - 1 ..); wave w owns channels [32w, 32w + 32)
  {
    cs_u32x4 bf[kSteps][2];
#pragma unroll
    for (int s = 0; s < kSteps; ++s)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        bf[s][nt] = *reinterpret_cast<const cs_u32x4*>(w7 + (static_cast<size_t>(4 * s + lk) * 128 + wave * 32 + nt * 16 + li) * 8);
    const uint16_t* xe = x + static_cast<size_t>(e) * H * W * 8;
    for (int pos = tid; pos < kFeM7Pos; pos += 256) {
      const int hy = y0 - 1 - kR + pos / kFeM7W, hx = x0 - 1 - kR + pos % kFeM7W;
      cs_u32x4 v = {0u, 0u, 0u, 0u};
      if (hy >= 0 && hy < H && hx >= 0 && hx < W) v = *reinterpret_cast<const cs_u32x4*>(xe + (static_cast<size_t>(hy) * W + hx) * 8);
      *reinterpret_cast<cs_u32x4*>(halo + pos * 16) = v;
    }
    if (tid < 2 * 17) *reinterpret_cast<cs_u32x4*>(f1 + kFeRows * 16 * kC3Stride + tid * 16) = cs_u32x4{0u, 0u, 0u, 0u};
    int toff[kSteps];
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
      const int tap = min(4 * s + lk, kTaps - 1);              // taps >= 49 have zero weights, any address does
      toff[s] = ((tap / 7) * kFeM7W + (tap % 7) + li) * 16;
    }
    const float bb0 = b7[wave * 32 + li], bb1 = b7[wave * 32 + 16 + li];
    __syncthreads();
    for (int r = 0; r < kFeRows; ++r) {
      const int gy = y0 - 1 + r;
      cs_v4f d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
      const bool row_in = gy >= 0 && gy < H;                   // (uniform)
      if (row_in) {
        const unsigned char* rowp = halo + r * kFeM7W * 16;
        cs_u32x4 afr[kSteps];
#pragma unroll
        for (int s = 0; s < kSteps; ++s) afr[s] = *reinterpret_cast<const cs_u32x4*>(rowp + toff[s]);
#pragma unroll
        for (int s = 0; s < kSteps; ++s) {
          d0 = cs_mfma<T>(afr[s], bf[s][0], d0);
          d1 = cs_mfma<T>(afr[s], bf[s][1], d1);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {                            // D: column li = channel, rows lk*4 + q = positions of the row
        const int p = lk * 4 + q, gx = x0 - 1 + p;
        const bool in = row_in && gx >= 0 && gx < W;
        unsigned char* dst = f1 + (r * 16 + p) * kC3Stride + (wave * 32 + li) * 2;
        *reinterpret_cast<uint16_t*>(dst) = in ? static_cast<uint16_t>(cs_bits<T>(fmaxf(d0[q] + bb0, 0.0f))) : static_cast<uint16_t>(0);
        *reinterpret_cast<uint16_t*>(dst + 32) = in ? static_cast<uint16_t>(cs_bits<T>(fmaxf(d1[q] + bb1, 0.0f))) : static_cast<uint16_t>(0);
      }
    }
  }
  // ---- stage 2: 3x3, 128 -> 64 channels; wave w owns output channels [16w, 16w + 16)
  const int co0 = wave * 16;
  const uint16_t* wl = w3 + (static_cast<size_t>(co0 + li)) * 128 + lk * 8;
  constexpr size_t tap_stride = static_cast<size_t>(64) * 128;
  cs_u32x4 bcur[4], bnxt[4];
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) bcur[kc] = *reinterpret_cast<const cs_u32x4*>(wl + kc * 32);
  cs_v4f acc[kTH];
#pragma unroll
  for (int py = 0; py < kTH; ++py) acc[py] = cs_v4f{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
#pragma unroll 1
  for (int t = 0; t < 9; ++t) {
    if (t < 8) {
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) bnxt[kc] = *reinterpret_cast<const cs_u32x4*>(wl + (t + 1) * tap_stride + kc * 32);
    }
    const unsigned char* tp = f1 + ((t / 3) * 16 + (t % 3) + li) * kC3Stride + lk * 16;
#pragma unroll
    for (int py = 0; py < kTH; ++py) {
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        const cs_u32x4 a = *reinterpret_cast<const cs_u32x4*>(tp + py * 16 * kC3Stride + kc * 64);
        acc[py] = cs_mfma<T>(a, bcur[kc], acc[py]);
      }
    }
    if (t < 8) {
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) bcur[kc] = bnxt[kc];
    }
  }
  __syncthreads();                                             // the intermediate is consumed: its LDS becomes the output slab
  constexpr int kOutStride = 64 * 2 + 16;
  const float bb = b3 ? b3[co0 + li] : 0.0f;
#pragma unroll
  for (int py = 0; py < kTH; ++py)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<uint16_t*>(f1 + (py * 16 + lk * 4 + q) * kOutStride + (co0 + li) * 2) =
          static_cast<uint16_t>(cs_bits<T>(fmaxf(acc[py][q] + bb, 0.0f)));
  __syncthreads();
  for (int id = tid; id < kTH * 16 * 8; id += 256) {           // 8 chunks of 16 B per pixel
    const int p = id >> 3, c = id & 7;
    const int px = p & 15, gy = y0 + (p >> 4), gx = x0 + px;
    if (px < kFeOW && gy < H && gx < W)
      *reinterpret_cast<cs_u32x4*>(y + ((static_cast<size_t>(e) * H + gy) * W + gx) * ystride + yoff + c * 8) =
          *reinterpret_cast<const cs_u32x4*>(f1 + p * kOutStride + c * 16);
  }
}

}  // namespace

#ifdef PVO_CONV_PROBE
extern "C" int pvo_debug_conv_probe(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_conv_probe), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

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

extern "C" int pvo_flow_encoder(const void* x, const void* w7_taps, const float* bias7, const void* w3_taps, const float* bias3,
                                void* y, int E, int H, int W, int ystride, int yoff, int dtype, void* stream) {
  if (E < 0 || H < 0 || W < 0) return PVO_EINVAL;
  if (ystride == 0) ystride = 64;
  if (ystride < yoff + 64 || yoff < 0 || (ystride & 7) || (yoff & 7)) return PVO_EINVAL;
  if (E == 0 || H == 0 || W == 0) return PVO_OK;
  if (!x || !w7_taps || !bias7 || !w3_taps || !y || E > 65535) return PVO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w7_taps) | reinterpret_cast<uintptr_t>(w3_taps) | reinterpret_cast<uintptr_t>(y)) & 15)
    return PVO_EINVAL;
  hipStream_t st = pvo_stream(stream);
  const dim3 grid((W + kFeOW - 1) / kFeOW, (H + kTH - 1) / kTH, E);
  const uint16_t *xp = static_cast<const uint16_t*>(x), *w7 = static_cast<const uint16_t*>(w7_taps), *w3 = static_cast<const uint16_t*>(w3_taps);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(flow_encoder_kernel<pvo_half>), hipFuncAttributeMaxDynamicSharedMemorySize, kFeLds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(flow_encoder_kernel<pvo_bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, kFeLds) != hipSuccess)
      return PVO_ELAUNCH;
    attr_set = true;
  }
  if (dtype == PVO_F16)
    hipLaunchKernelGGL(flow_encoder_kernel<pvo_half>, grid, dim3(256), kFeLds, st, xp, w7, bias7, w3, bias3, static_cast<uint16_t*>(y), H, W, ystride, yoff);
  else if (dtype == PVO_BF16)
    hipLaunchKernelGGL(flow_encoder_kernel<pvo_bf16>, grid, dim3(256), kFeLds, st, xp, w7, bias7, w3, bias3, static_cast<uint16_t*>(y), H, W, ystride, yoff);
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
                                int E, int H, int W, int Cout, int relu, int ystride, int yoff, int dtype, void* stream) {
  if (E < 0 || H < 0 || W < 0) return PVO_EINVAL;
  if (Cout != 64 && (Cout <= 0 || (Cout & 127))) return PVO_EUNSUPPORTED;
  if (ystride == 0) ystride = Cout;
  if (ystride < yoff + Cout || yoff < 0 || (ystride & 7) || (yoff & 7)) return PVO_EINVAL;
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
    if (dtype == PVO_F16) hipLaunchKernelGGL((conv3x3_c128_kernel<pvo_half, 1>), grid, dim3(256), lds, st, xp, wp, bias, yp, H, W, Cout, relu, ystride, yoff);
    else if (dtype == PVO_BF16) hipLaunchKernelGGL((conv3x3_c128_kernel<pvo_bf16, 1>), grid, dim3(256), lds, st, xp, wp, bias, yp, H, W, Cout, relu, ystride, yoff);
    else return PVO_EUNSUPPORTED;
  } else {
    dim3 grid(ntx * (Cout / 128), (H + kTH - 1) / kTH, E);
    if (dtype == PVO_F16) hipLaunchKernelGGL((conv3x3_c128_kernel<pvo_half, 2>), grid, dim3(256), lds, st, xp, wp, bias, yp, H, W, Cout, relu, ystride, yoff);
    else if (dtype == PVO_BF16) hipLaunchKernelGGL((conv3x3_c128_kernel<pvo_bf16, 2>), grid, dim3(256), lds, st, xp, wp, bias, yp, H, W, Cout, relu, ystride, yoff);
    else return PVO_EUNSUPPORTED;
  }
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

static int launch_big(const void* x, const void* w_taps, const float* bias, void* y,
                      int E, int H, int W, int Cin, int Cout, int relu, int ystride, int yoff, int dtype, void* stream,
                      const BigEpi& ep) {
  if (E < 0 || H < 0 || W < 0) return PVO_EINVAL;
  if (Cin <= 0 || (Cin & 31) || Cout <= 0 || (Cout & 127)) return PVO_EUNSUPPORTED;
  if (ystride == 0) ystride = Cout;
  if (ystride < yoff + Cout || yoff < 0 || (ystride & 7) || (yoff & 7)) return PVO_EINVAL;
  if (E == 0 || H == 0 || W == 0) return PVO_OK;
  if (!x || !w_taps || !y || E > 65535) return PVO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_taps) | reinterpret_cast<uintptr_t>(y)) & 15) return PVO_EINVAL;
  if (static_cast<long long>(H) * W * Cin > 0x7fffffffLL) return PVO_EUNSUPPORTED;
  hipStream_t st = pvo_stream(stream);
  const int ntx = (W + kBT - 1) / kBT;
  const size_t lds = 67584;      // two halo buffers (67200 B) / the fp32 epilogue slab (128 px x 132 floats = 67584 B)
  dim3 grid(ntx * (Cout / 128), (H + kBT - 1) / kBT, E);
  const uint16_t* xp = static_cast<const uint16_t*>(x);
  const uint16_t* wp = static_cast<const uint16_t*>(w_taps);
  uint16_t* yp = static_cast<uint16_t*>(y);
  static bool attr_set[4] = {false, false, false, false};  // hipFuncSetAttribute once per process, not per launch
  auto go = [&](auto kernel, int slot) -> int {
    if (!attr_set[slot]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) return PVO_ELAUNCH;
      attr_set[slot] = true;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, xp, wp, bias, yp, H, W, Cin, Cout, relu, ystride, yoff, ep);
    return PVO_OK;
  };
  const bool heads = ep.mode == 3;
  int rc;
  if (dtype == PVO_F16) rc = heads ? go(conv3x3_big_kernel<pvo_half, true>, 2) : go(conv3x3_big_kernel<pvo_half, false>, 0);
  else if (dtype == PVO_BF16) rc = heads ? go(conv3x3_big_kernel<pvo_bf16, true>, 3) : go(conv3x3_big_kernel<pvo_bf16, false>, 1);
  else return PVO_EUNSUPPORTED;
  if (rc != PVO_OK) return rc;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_conv3x3(const void* x, const void* w_taps, const float* bias, void* y,
                           int E, int H, int W, int Cin, int Cout, int relu, int ystride, int yoff, int dtype, void* stream) {
  return launch_big(x, w_taps, bias, y, E, H, W, Cin, Cout, relu, ystride, yoff, dtype, stream, BigEpi{});
}

extern "C" int pvo_conv3x3_heads(const void* x, const void* w1_taps, const float* bias1, const void* w2_frags, float* z,
                                 int E, int H, int W, int dtype, void* stream) {
  if (!bias1 || !w2_frags || !z || (reinterpret_cast<uintptr_t>(w2_frags) & 15)) return PVO_EINVAL;
  BigEpi ep{};
  ep.mode = 3; ep.w2f = static_cast<const uint16_t*>(w2_frags); ep.z = z;
  return launch_big(x, w1_taps, bias1, z, E, H, W, 128, 512, 1, 0, 0, dtype, stream, ep);      // (y is not written in this mode)
}

// the ConvGRU input [first(128) | cf(cf_channels)] as two segments: `first` = net (gates) or r*net (candidate), `cf` = the
// encoders' output written side by side (relu(corr features) | relu(flow features)) by their own convolutions
static int seg_setup(BigEpi& ep, const void* first, const void* cf, int cf_channels) {
  if (!first || !cf) return PVO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(first) | reinterpret_cast<uintptr_t>(cf)) & 15) return PVO_EINVAL;
  if (cf_channels <= 0 || (cf_channels & 31)) return PVO_EUNSUPPORTED;
  ep.nseg = 2;
  ep.seg_p[0] = static_cast<const uint16_t*>(first); ep.seg_stride[0] = 128; ep.seg_chunks[0] = 4;
  ep.seg_p[1] = static_cast<const uint16_t*>(cf); ep.seg_stride[1] = cf_channels; ep.seg_chunks[1] = cf_channels >> 5;
  return PVO_OK;
}

extern "C" int pvo_gru_conv_gates(const void* net, const void* cf, int cf_channels, const void* w_taps, const float* g,
                                  const void* P_zr, const int* p_slots, void* Z, void* RN, int E, int H, int W, int dtype, void* stream) {
  if (!g || !P_zr || !RN) return PVO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(P_zr) | reinterpret_cast<uintptr_t>(RN)) & 15) return PVO_EINVAL;
  BigEpi ep{};
  const int rc = seg_setup(ep, net, cf, cf_channels);
  if (rc != PVO_OK) return rc;
  ep.mode = 1; ep.g = g; ep.P = static_cast<const uint16_t*>(P_zr); ep.p_slots = p_slots; ep.net = static_cast<const uint16_t*>(net);
  ep.y2 = static_cast<uint16_t*>(RN);
  return launch_big(net, w_taps, nullptr, Z, E, H, W, 128 + cf_channels, 256, 0, 0, 0, dtype, stream, ep);
}

extern "C" int pvo_gru_conv_candidate(const void* RN, const void* cf, int cf_channels, const void* w_taps, const float* g,
                                      const void* P_q, const int* p_slots, const void* Z, const void* net, void* net_out,
                                      int E, int H, int W, int dtype, void* stream) {
  if (!g || !P_q || !net || !Z) return PVO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(P_q) | reinterpret_cast<uintptr_t>(net) | reinterpret_cast<uintptr_t>(Z)) & 15) return PVO_EINVAL;
  BigEpi ep{};
  const int rc = seg_setup(ep, RN, cf, cf_channels);
  if (rc != PVO_OK) return rc;
  ep.mode = 2; ep.g = g; ep.P = static_cast<const uint16_t*>(P_q); ep.p_slots = p_slots; ep.net = static_cast<const uint16_t*>(net);
  ep.Z = static_cast<const uint16_t*>(Z);
  return launch_big(RN, w_taps, nullptr, net_out, E, H, W, 128 + cf_channels, 128, 0, 0, 0, dtype, stream, ep);
}
