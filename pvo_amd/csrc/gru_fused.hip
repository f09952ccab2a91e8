// gru_fused.hip — two layers of the update operator that are neither a plain convolution nor plain element-wise work.
//
//   seg_mean    GraphAgg's scatter_mean over the edges sharing a source frame (VO_Module/droid_slam/droid_net.py:83-87),
//               with conv1's bias + ReLU applied as the input is read
//   heads_out   the second stage of the four output heads, 4 x (ReLU + Conv3x3(128 -> 2)) (droid_net.py:184-210)
// Layout: channels-last fp16/bf16 rows ([E, H*W, C]); 8 channels (16 B) per thread per access; arithmetic in fp32.
#include "common.h"
#include "graph_post.h"
#include <stdlib.h>

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct H8 {   // 8 x 16-bit <-> 8 x float
  static __device__ __forceinline__ void unpack(u32x4 v, float f[8]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[2 * k] = Elem<T>::to_f32(from_bits(v[k] & 0xffffu));
      f[2 * k + 1] = Elem<T>::to_f32(from_bits(v[k] >> 16));
    }
  }
  static __device__ __forceinline__ u32x4 pack(const float f[8]) {
    u32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = to_bits(Elem<T>::from_f32(f[2 * k])) | (to_bits(Elem<T>::from_f32(f[2 * k + 1])) << 16);
    return v;
  }
  static __device__ __forceinline__ typename Elem<T>::store_t from_bits(uint32_t b);
  static __device__ __forceinline__ uint32_t to_bits(typename Elem<T>::store_t s);
};
template <> __device__ __forceinline__ _Float16 H8<pvo_half>::from_bits(uint32_t b) {
  union { uint16_t u; _Float16 h; } c; c.u = static_cast<uint16_t>(b); return c.h;
}
template <> __device__ __forceinline__ uint32_t H8<pvo_half>::to_bits(_Float16 s) {
  union { uint16_t u; _Float16 h; } c; c.h = s; return c.u;
}
template <> __device__ __forceinline__ uint16_t H8<pvo_bf16>::from_bits(uint32_t b) { return static_cast<uint16_t>(b); }
template <> __device__ __forceinline__ uint32_t H8<pvo_bf16>::to_bits(uint16_t s) { return s; }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// 8 consecutive floats (32-byte aligned offset) as two 16-byte loads; p == nullptr -> zeros.
// (Eight scalar loads per 16 bytes of payload made these kernels address-unit bound: 3.3x slower than a copy.)
__device__ __forceinline__ void load8f(const float* __restrict__ p, float f[8]) {
  if (p == nullptr) {
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = 0.0f;
    return;
  }
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// out[k, p, c] = mean over edges e in [ptr[k], ptr[k+1]) of x[idx[e], p, c]   (GraphAgg's scatter_mean, droid_net.py:87)
template <typename T>
__global__ __launch_bounds__(256) void seg_mean_kernel(const uint16_t* __restrict__ x, const int* __restrict__ ptr,
                                                       const int* __restrict__ idx, const float* __restrict__ in_bias,
                                                       uint16_t* __restrict__ out, int HW, int C) {
  const int k = blockIdx.y;
  const int e0 = ptr[k], e1 = ptr[k + 1];
  const int cpr = C >> 3;
  const long long per = static_cast<long long>(HW) * cpr;
  const float inv = 1.0f / static_cast<float>(max(e1 - e0, 1));
  for (long long id = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; id < per; id += static_cast<long long>(gridDim.x) * 256) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float bb[8];
    if (in_bias) load8f(in_bias + static_cast<int>(id % cpr) * 8, bb);
    // four edges' rows are requested before the first is consumed (a load -> add -> load loop left one 16-byte request in
    // flight per thread: 27 us for 28 MB); the sum still runs in edge order
    for (int o = e0; o < e1; o += 4) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        v[u] = *reinterpret_cast<const u32x4*>(x + (static_cast<long long>(idx[min(o + u, e1 - 1)]) * per + id) * 8);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (o + u >= e1) break;
        float f[8];
        H8<T>::unpack(v[u], f);
        if (in_bias) {    // x is a bias-free convolution output: relu(x + b), rounded to the storage type as a separate pass would
#pragma unroll
          for (int q = 0; q < 8; ++q) f[q] = Elem<T>::to_f32(Elem<T>::from_f32(fmaxf(f[q] + bb[q], 0.0f)));
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += f[q];
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] *= inv;
    *reinterpret_cast<u32x4*>(out + (static_cast<long long>(k) * per + id) * 8) = H8<T>::pack(acc);
  }
}

inline unsigned grid_for(long long items) {
  long long b = (items + 255) / 256;
  return static_cast<unsigned>(b < 1 ? 1 : (b > 256 * 32 ? 256 * 32 : b));
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

#define GRU_DISPATCH(dtype, CALL_H, CALL_B)        \
  do {                                             \
    if ((dtype) == PVO_F16) { CALL_H; }            \
    else if ((dtype) == PVO_BF16) { CALL_B; }      \
    else return PVO_EUNSUPPORTED;                  \
  } while (0)

extern "C" int pvo_segment_mean(const void* x, const int* seg_ptr, const int* seg_idx, const float* in_bias, void* out,
                                int K, int HW, int C, int dtype, void* stream) {
  if (K < 0 || HW < 0 || C <= 0 || (C & 7)) return PVO_EINVAL;
  if (K == 0 || HW == 0) return PVO_OK;
  if (!x || !seg_ptr || !seg_idx || !out || !aligned16(x) || !aligned16(out) || K > 65535) return PVO_EINVAL;
  hipStream_t st = pvo_stream(stream);
  const long long per = static_cast<long long>(HW) * (C >> 3);
  dim3 grid(static_cast<unsigned>((per + 255) / 256 > 1024 ? 1024 : (per + 255) / 256), K);
  GRU_DISPATCH(dtype,
    hipLaunchKernelGGL(seg_mean_kernel<pvo_half>, grid, dim3(256), 0, st, static_cast<const uint16_t*>(x), seg_ptr, seg_idx, in_bias, static_cast<uint16_t*>(out), HW, C),
    hipLaunchKernelGGL(seg_mean_kernel<pvo_bf16>, grid, dim3(256), 0, st, static_cast<const uint16_t*>(x), seg_ptr, seg_idx, in_bias, static_cast<uint16_t*>(out), HW, C));
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

// ---------------------------------------------------------------------------
// heads_out: second stage of the four output heads in one kernel.
//   reference: delta / delta_dy / weight / delta_mask = Conv3x3(128->2)(ReLU(Conv3x3(128->128)(net)))   droid_net.py:184-210
//   input  h1 [E,H,W,512] = the four first-stage convolutions side by side, WITHOUT bias and ReLU (a bias-free MIOpen
//          conv); bias1 [512] f32 and the ReLU are applied while the tile is staged, zero padding outside the image
//   w2     [4 heads][2 outputs][9 taps][128 channels] 16-bit, bias2 [8] f32
//   output y [E,H,W,8] (head-major: delta, delta_dy, weight, delta_mask)
// A 512->8 convolution is a poor fit for an implicit-GEMM kernel (N = 8 of a 128/256-wide tile: 57 TFLOP/s measured);
// here one workgroup owns an 8x16 pixel tile and, head by head, stages the 10x18 halo (128 channels) in LDS with a padded
// row stride while the next head's loads are in flight; each wave multiplies two tile rows on the matrix cores
// (v_mfma_f32_16x16x32: 16 pixels x 32 channels x 16 columns of which 2 are the head's outputs).
// ---------------------------------------------------------------------------
namespace {


constexpr int kHT = 8, kWT = 16;                 // pixel tile
constexpr int kHaloW = kWT + 2, kHaloPos = (kHT + 2) * kHaloW;   // 180 positions
constexpr int kPosStride = 128 * 2 + 16;         // bytes per halo position (padded against b128 bank conflicts)

typedef float ho_v4f __attribute__((ext_vector_type(4)));
typedef _Float16 ho_v8h __attribute__((ext_vector_type(8)));
typedef __bf16 ho_v8b __attribute__((ext_vector_type(8)));
template <typename T> __device__ __forceinline__ ho_v4f ho_mfma(u32x4 a, u32x4 b, ho_v4f c);
template <> __device__ __forceinline__ ho_v4f ho_mfma<pvo_half>(u32x4 a, u32x4 b, ho_v4f c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ho_v8h, a), __builtin_bit_cast(ho_v8h, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ ho_v4f ho_mfma<pvo_bf16>(u32x4 a, u32x4 b, ho_v4f c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ho_v8b, a), __builtin_bit_cast(ho_v8b, b), c, 0, 0, 0);
}

// History (all measured on MI355X, S-B): packed-dot versions (v_dot2c, weights through LDS, SGPRs, ...) sat at 72-86 us
// whatever their memory schedule: probe builds showed 31 us of HBM streaming, 20 us of staging and the rest dot issue.
// Matrix cores + per-tap global weight fragments: 62 us (20 us of it the fragment loads); weights through LDS: 44 us.
template <typename T>
__global__ __launch_bounds__(256) void heads_out_kernel(const uint16_t* __restrict__ h1, const float* __restrict__ bias1,
                                                        const uint32_t* __restrict__ w2, const float* __restrict__ bias2,
                                                        uint16_t* __restrict__ y, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xs = smem;                                   // [180][272 B]; reused for the final transpose
  unsigned char* wl = smem + kHaloPos * kPosStride;           // this head's second-stage weights [2][9][128] 16-bit
  unsigned char* wzero = wl + 2 * 9 * 256;                    // 16 zero bytes: the B fragment of the 14 padding columns
  const int e = blockIdx.z;
  const int y0 = blockIdx.y * kHT, x0 = blockIdx.x * kWT;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, lk = lane >> 4;
  constexpr int kIters = (kHaloPos * 16 + 255) / 256;         // 12 independent 16-byte loads in flight per thread
  const int ch = tid & 15;

  // halo positions of this thread: fixed over the heads, so the offsets are computed once (-1: outside the image)
  const uint16_t* __restrict__ h1e = h1 + static_cast<size_t>(e) * H * W * 512;   // this edge's image (32-bit offsets below)
  int off[kIters];
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    const int pos = (tid >> 4) + 16 * it;
    off[it] = -1;
    if (pos < kHaloPos) {
      const int hy = y0 - 1 + pos / kHaloW, hx = x0 - 1 + pos % kHaloW;
      if (hy >= 0 && hy < H && hx >= 0 && hx < W) off[it] = (hy * W + hx) * 512 + ch * 8;
    }
  }
  // software pipeline: the global loads of head h+1 are in flight while head h is being computed from LDS
  u32x4 raw[kIters];
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    raw[it] = u32x4{0u, 0u, 0u, 0u};
    if (off[it] >= 0) raw[it] = *reinterpret_cast<const u32x4*>(h1e + off[it]);
  }
  u32x4 wr0 = *reinterpret_cast<const u32x4*>(w2 + tid * 4), wr1 = {0u, 0u, 0u, 0u};
  if (tid < 32) wr1 = *reinterpret_cast<const u32x4*>(w2 + (256 + tid) * 4);
  if (tid == 0) *reinterpret_cast<u32x4*>(wzero) = u32x4{0u, 0u, 0u, 0u};
  float acc[32];                                              // [head][tile row of the wave][4 pixels] of D column li
#pragma unroll
  for (int head = 0; head < 4; ++head) {
    __syncthreads();                                          // previous head's tile fully consumed
    {
      float bb[8];
      load8f(bias1 + head * 128 + ch * 8, bb);

#pragma unroll
      for (int it = 0; it < kIters; ++it) {
        const int pos = (tid >> 4) + 16 * it;
        if (pos < kHaloPos) {
          u32x4 v = {0u, 0u, 0u, 0u};
          if (off[it] >= 0) {                                 // zero padding is applied AFTER bias + ReLU
            float f[8];
            H8<T>::unpack(raw[it], f);
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = fmaxf(f[k] + bb[k], 0.0f);
            v = H8<T>::pack(f);
          }
          *reinterpret_cast<u32x4*>(xs + pos * kPosStride + ch * 16) = v;
        }
      }
      *reinterpret_cast<u32x4*>(wl + tid * 16) = wr0;
      if (tid < 32) *reinterpret_cast<u32x4*>(wl + (256 + tid) * 16) = wr1;
    }
    __syncthreads();
    if (head < 3) {
#pragma unroll
      for (int it = 0; it < kIters; ++it)
        if (off[it] >= 0) raw[it] = *reinterpret_cast<const u32x4*>(h1e + off[it] + (head + 1) * 128);
      wr0 = *reinterpret_cast<const u32x4*>(w2 + (head + 1) * 1152 + tid * 4);
      if (tid < 32) wr1 = *reinterpret_cast<const u32x4*>(w2 + (head + 1) * 1152 + (256 + tid) * 4);
    }
    // matrix cores: D[16 pixels of a tile row][16 columns, 2 real = the head's outputs] += A[16 px][32 ch] B[32 ch][16]
    // per (tap, 32-channel chunk).  7/8 of the columns multiply zeros - still 4x cheaper than 576 v_dot2c per thread
    // (the packed-dot version sat at 75 us whatever its memory schedule was: it was dot-issue bound).
    // wave w owns tile rows 2w, 2w+1; lane: pixel li = lane & 15, channel group lk = lane >> 4 (8 channels).
    ho_v4f d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    // B fragments come from the head's weights in LDS ([out][tap][128 ch]); columns >= 2 read a zero row (a per-tap
    // global load of the fragments cost 20 us of the 62: measured with probe builds)
    const unsigned char* wrow = li < 2 ? wl + li * 9 * 256 + lk * 16 : wzero;
    const int wstep = li < 2 ? 1 : 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      u32x4 bf[4];
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        bf[kc] = *reinterpret_cast<const u32x4*>(wrow + wstep * (t * 256 + kc * 64));
      }
      const unsigned char* xp0 = xs + ((2 * wave + t / 3) * kHaloW + (li + t % 3)) * kPosStride + lk * 16;
      const unsigned char* xp1 = xp0 + kHaloW * kPosStride;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        d0 = ho_mfma<T>(*reinterpret_cast<const u32x4*>(xp0 + kc * 64), bf[kc], d0);
        d1 = ho_mfma<T>(*reinterpret_cast<const u32x4*>(xp1 + kc * 64), bf[kc], d1);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc[head * 8 + r] = d0[r]; acc[head * 8 + 4 + r] = d1[r]; }
  }
  __syncthreads();                                            // the last tile is consumed: its LDS carries the transpose
  float* ys = reinterpret_cast<float*>(xs);                   // [128 px][8]
  if (li < 2) {                                               // D column li = output li; rows lk*4 + r = pixels of the tile row
#pragma unroll
    for (int head = 0; head < 4; ++head)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ys[((2 * wave) * 16 + lk * 4 + r) * 8 + head * 2 + li] = acc[head * 8 + r];
        ys[((2 * wave + 1) * 16 + lk * 4 + r) * 8 + head * 2 + li] = acc[head * 8 + 4 + r];
      }
  }
  __syncthreads();
  if (tid < 128) {                                            // add bias, one 16-byte store per pixel
    const int gy = y0 + (tid >> 4), gx = x0 + (tid & 15);
    if (gy < H && gx < W) {
      float f[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = ys[tid * 8 + k] + bias2[k];
      *reinterpret_cast<u32x4*>(y + ((static_cast<size_t>(e) * H + gy) * W + gx) * 8) = H8<T>::pack(f);
    }
  }
}

// heads, second stage after pvo_conv3x3_heads: y[p][2 head + o] = bias2 + sum over the nine taps t of z[p + t][head][2 t + o]
// (zero padding: a neighbour outside the image contributes nothing).  One thread per (pixel, head), nine 8-byte reads.
template <typename T>
__global__ __launch_bounds__(256) void heads_gather_kernel(const float* __restrict__ z, const float* __restrict__ bias2,
                                                           uint16_t* __restrict__ y, int H, int W, long long total) {
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= total) return;
  const int head = static_cast<int>(idx & 3);
  const long long p = idx >> 2;
  const int px = static_cast<int>(p % W), py = static_cast<int>((p / W) % H);
  float a0 = bias2[2 * head], a1 = bias2[2 * head + 1];
  // all nine reads are requested first (an out-of-image neighbour reads the centre pixel's entry and is not added)
  float2 v[9];
  bool ok[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int qy = py + t / 3 - 1, qx = px + t % 3 - 1;
    ok[t] = qy >= 0 && qy < H && qx >= 0 && qx < W;
    const long long q = ok[t] ? p + (t / 3 - 1) * W + (t % 3 - 1) : p;
    v[t] = *reinterpret_cast<const float2*>(z + q * 72 + head * 18 + 2 * t);
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
    if (ok[t]) { a0 += v[t].x; a1 += v[t].y; }
  const uint32_t lo = H8<T>::to_bits(Elem<T>::from_f32(a0)), hi = H8<T>::to_bits(Elem<T>::from_f32(a1));
  *reinterpret_cast<uint32_t*>(y + p * 8 + 2 * head) = lo | (hi << 16);
}

// The same through LDS: a workgroup owns 8 x 16 pixels and stages their 10 x 18 halo of z rows (288 B each) with whole 16-byte
// reads - consecutive lanes along a row: every byte of z is fetched once (1.4x with the halo) - where the kernel above sends nine
// 8-byte reads per thread that each touch a different 72-byte segment of a row.  Same sums in the same order: bit-identical.
constexpr int kHgTH = 8, kHgTW = 16, kHgPos = (kHgTH + 2) * (kHgTW + 2), kHgRow = 288 + 16;      // (+16: rows of one pixel column land on different banks)

// POST: the pixel's four heads sit in four consecutive lanes (delta | delta_dy | weight logits | delta_mask, the order pvo_graph_post
// reads them in); the lane of head 0 collects the four packed pairs - rounded to 16 bits exactly as they are stored - and runs
// graph_post_pixel (graph_post.h) on them: FactorGraph.update's mask / target / weight arithmetic without a launch of its own
// (4.6 us + a kernel boundary per graph update at S-B).  Without the panoptic vote only - that needs the edge's whole histogram first.
template <typename T, bool POST>
__global__ __launch_bounds__(256) void heads_gather_tiled_kernel(const float* __restrict__ z, const float* __restrict__ bias2,
                                                                 uint16_t* __restrict__ y, int H, int W, GraphPostArgs gp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hgs[];
  const int e = blockIdx.z, y0 = blockIdx.y * kHgTH, x0 = blockIdx.x * kHgTW, tid = threadIdx.x;
  const float* ze = z + static_cast<size_t>(e) * H * W * 72;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  constexpr int kChunks = kHgPos * 18, kIter = (kChunks + 255) / 256;              // 16-byte chunks of the halo: 18 per position
  u32x4 stg[kIter];
#pragma unroll
  for (int it = 0; it < kIter; ++it) {
    const int id = tid + 256 * it, pos = id / 18, c = id - pos * 18;
    const int hy = y0 - 1 + pos / (kHgTW + 2), hx = x0 - 1 + pos % (kHgTW + 2);
    stg[it] = u32x4{0u, 0u, 0u, 0u};
    if (id < kChunks && hy >= 0 && hy < H && hx >= 0 && hx < W)
      stg[it] = *reinterpret_cast<const u32x4*>(ze + (static_cast<size_t>(hy) * W + hx) * 72 + c * 4);
  }
#pragma unroll
  for (int it = 0; it < kIter; ++it) {
    const int id = tid + 256 * it, pos = id / 18, c = id - pos * 18;
    if (id < kChunks) *reinterpret_cast<u32x4*>(hgs + pos * kHgRow + c * 16) = stg[it];
  }
  __syncthreads();
  const int head = tid & 3;
  const float b0 = bias2[2 * head], b1 = bias2[2 * head + 1];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int pl = (tid >> 2) + 64 * pass, ly = pl >> 4, lx = pl & 15;             // pixel of the tile
    const int py = y0 + ly, px = x0 + lx;
    if (py >= H || px >= W) continue;
    float a0 = b0, a1 = b1;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int qy = py + t / 3 - 1, qx = px + t % 3 - 1;
      if (qy >= 0 && qy < H && qx >= 0 && qx < W) {
        const float2 v = *reinterpret_cast<const float2*>(hgs + ((ly + t / 3) * (kHgTW + 2) + lx + t % 3) * kHgRow + (head * 18 + 2 * t) * 4);
        a0 += v.x; a1 += v.y;
      }
    }
    const uint32_t lo = H8<T>::to_bits(Elem<T>::from_f32(a0)), hi = H8<T>::to_bits(Elem<T>::from_f32(a1));
    const uint32_t mine = (lo & 0xffffu) | (hi << 16);
    *reinterpret_cast<uint32_t*>(y + ((static_cast<size_t>(e) * H + py) * W + px) * 8 + 2 * head) = mine;
    if (POST) {
      // (the four lanes of a pixel are all here or all gone: `continue` above depends on the pixel only)
      const int quad = (tid & 63) & ~3;
      uint4 q;
      q.x = __shfl(mine, quad); q.y = __shfl(mine, quad + 1); q.z = __shfl(mine, quad + 2); q.w = __shfl(mine, quad + 3);
      if (head == 0) {
        const int pix = py * W + px;
        graph_post_pixel<T>(e * (H * W) + pix, e, pix, q, gp, H * W, W, nullptr, nullptr, nullptr, 0, 0.0f);
      }
    }
  }
}

}  // namespace

static int heads_gather_launch(const float* z, const float* bias2, void* y, const GraphPostArgs* post, int* fused,
                               int E, int H, int W, int dtype, void* stream) {
  if (fused) *fused = 0;
  if (E < 0 || H < 0 || W < 0) return PVO_EINVAL;
  if (E == 0 || H == 0 || W == 0) return PVO_OK;
  if (!z || !bias2 || !y || (reinterpret_cast<uintptr_t>(z) & 7) || (reinterpret_cast<uintptr_t>(y) & 3)) return PVO_EINVAL;
  hipStream_t st = pvo_stream(stream);
  const bool flat = pvo_knob(PVO_KNOB_HEADS_GATHER_FLAT) != 0;      // (pvo_debug_config: a test compares the two forms bit for bit)
  if (!flat && (reinterpret_cast<uintptr_t>(z) & 15) == 0 && E <= 65535) {
    const dim3 grid((W + kHgTW - 1) / kHgTW, (H + kHgTH - 1) / kHgTH, E);
    constexpr size_t lds = static_cast<size_t>(kHgPos) * kHgRow;                      // 54720 B
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(heads_gather_tiled_kernel<pvo_half, false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(heads_gather_tiled_kernel<pvo_bf16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(heads_gather_tiled_kernel<pvo_half, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(heads_gather_tiled_kernel<pvo_bf16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess)
        return PVO_ELAUNCH;
      attr_set = true;
    }
    if (dtype != PVO_F16 && dtype != PVO_BF16) return PVO_EUNSUPPORTED;
    const bool with_post = post != nullptr && static_cast<long long>(E) * H * W < (1LL << 31);
    const GraphPostArgs gp = with_post ? *post : GraphPostArgs{};
    uint16_t* yp = static_cast<uint16_t*>(y);
    if (with_post) {
      if (dtype == PVO_F16) hipLaunchKernelGGL((heads_gather_tiled_kernel<pvo_half, true>), grid, dim3(256), lds, st, z, bias2, yp, H, W, gp);
      else hipLaunchKernelGGL((heads_gather_tiled_kernel<pvo_bf16, true>), grid, dim3(256), lds, st, z, bias2, yp, H, W, gp);
      if (fused) *fused = 1;
    } else {
      if (dtype == PVO_F16) hipLaunchKernelGGL((heads_gather_tiled_kernel<pvo_half, false>), grid, dim3(256), lds, st, z, bias2, yp, H, W, gp);
      else hipLaunchKernelGGL((heads_gather_tiled_kernel<pvo_bf16, false>), grid, dim3(256), lds, st, z, bias2, yp, H, W, gp);
    }
    PVO_CHECK_LAUNCH();
    return PVO_OK;
  }
  const long long total = static_cast<long long>(E) * H * W * 4;
  const dim3 grid(static_cast<unsigned>((total + 255) / 256));
  if (dtype == PVO_F16) hipLaunchKernelGGL(heads_gather_kernel<pvo_half>, grid, dim3(256), 0, st, z, bias2, static_cast<uint16_t*>(y), H, W, total);
  else if (dtype == PVO_BF16) hipLaunchKernelGGL(heads_gather_kernel<pvo_bf16>, grid, dim3(256), 0, st, z, bias2, static_cast<uint16_t*>(y), H, W, total);
  else return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_heads_gather(const float* z, const float* bias2, void* y, int E, int H, int W, int dtype, void* stream) {
  return heads_gather_launch(z, bias2, y, nullptr, nullptr, E, H, W, dtype, stream);
}

int pvo_internal_heads_gather_post(const float* z, const float* bias2, void* y, const GraphPostArgs* post, int* fused,
                                   int E, int H, int W, int dtype, void* stream) {
  return heads_gather_launch(z, bias2, y, post, fused, E, H, W, dtype, stream);
}

extern "C" int pvo_heads_out(const void* h1, const float* bias1, const void* w2, const float* bias2, void* y,
                             int E, int H, int W, int dtype, void* stream) {
  if (E < 0 || H < 0 || W < 0) return PVO_EINVAL;
  if (E == 0 || H == 0 || W == 0) return PVO_OK;
  if (!h1 || !bias1 || !w2 || !bias2 || !y || !aligned16(h1) || !aligned16(w2) || !aligned16(y) ||
      (reinterpret_cast<uintptr_t>(bias1) & 15) || E > 65535) return PVO_EINVAL;
  if (static_cast<long long>(H) * W * 512 > 0x7fffffffLL) return PVO_EUNSUPPORTED;   // per-image offsets are 32-bit
  hipStream_t st = pvo_stream(stream);
  const size_t lds = static_cast<size_t>(kHaloPos) * kPosStride + 2 * 9 * 256 + 16;   // 53584 B
  dim3 grid((W + kWT - 1) / kWT, (H + kHT - 1) / kHT, E);
  if (dtype == PVO_F16) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(heads_out_kernel<pvo_half>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) return PVO_ELAUNCH;
    hipLaunchKernelGGL(heads_out_kernel<pvo_half>, grid, dim3(256), lds, st, static_cast<const uint16_t*>(h1), bias1, static_cast<const uint32_t*>(w2), bias2, static_cast<uint16_t*>(y), H, W);
  } else if (dtype == PVO_BF16) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(heads_out_kernel<pvo_bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) return PVO_ELAUNCH;
    hipLaunchKernelGGL(heads_out_kernel<pvo_bf16>, grid, dim3(256), lds, st, static_cast<const uint16_t*>(h1), bias1, static_cast<const uint32_t*>(w2), bias2, static_cast<uint16_t*>(y), H, W);
  } else {
    return PVO_EUNSUPPORTED;
  }
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}
