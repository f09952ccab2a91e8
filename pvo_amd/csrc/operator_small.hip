// operator_small.hip — the small layers of the update operator that used to go to MIOpen / hipBLASLt.
//
//   pvo_gate_context   g[e, 0:384] = Wg glo[e] + bg, glo[e] = sum of the per-chunk partial means pvo_gru_glo_fused wrote:
//                      the three 1x1 "global context" convolutions of the ConvGRU (convz_glo | convr_glo | convq_glo,
//                      VO_Module/droid_slam/modules/gru.py:13-15,26-30) as one tiny fp32 GEMV per edge
//   pvo_eta_head       GraphAgg's eta head, Conv2d(128,1,3,padding=1) + Softplus, x0.01 (droid_net.py:72-74,93-95), and
//                      optionally FactorGraph's damping bookkeeping (factor_graph.py:281-297) in the same launch
//   pvo_conv1x1_c128   y = x W^T + b for a 128-channel channels-last tensor: GraphAgg.upmask_disp, Conv2d(128,576,1)
//                      (droid_net.py:76-77,93)
//   pvo_corr_encode    relu(W corr + b) for an already sampled 196-channel correlation tensor: corr_encoder[0:2],
//                      Conv2d(196,128,1) + ReLU (droid_net.py:172-175).  The factor graph's resident volume pool goes
//                      through pvo_corr_lookup_encode_tiled instead, where the 196 channels never reach HBM.
//   pvo_segment_hist   per (edge, panoptic segment) pixel counts for the dynamic-segment vote (factor_graph.py:256-276)
#include "common.h"
#include "conv1x1_tile.h"
#include "glo_tile.h"

namespace {

typedef uint32_t os_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t os_u32x2 __attribute__((ext_vector_type(2)));
typedef float os_v4f __attribute__((ext_vector_type(4)));
typedef _Float16 os_v8h __attribute__((ext_vector_type(8)));
typedef __bf16 os_v8b __attribute__((ext_vector_type(8)));

template <typename T> __device__ __forceinline__ os_v4f os_mfma(os_u32x4 a, os_u32x4 b, os_v4f c);
template <> __device__ __forceinline__ os_v4f os_mfma<pvo_half>(os_u32x4 a, os_u32x4 b, os_v4f c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(os_v8h, a), __builtin_bit_cast(os_v8h, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ os_v4f os_mfma<pvo_bf16>(os_u32x4 a, os_u32x4 b, os_v4f c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(os_v8b, a), __builtin_bit_cast(os_v8b, b), c, 0, 0, 0);
}
template <typename T> __device__ __forceinline__ float os_val(uint32_t b);
template <> __device__ __forceinline__ float os_val<pvo_half>(uint32_t b) {
  union { uint16_t u; _Float16 h; } c; c.u = static_cast<uint16_t>(b); return static_cast<float>(c.h);
}
template <> __device__ __forceinline__ float os_val<pvo_bf16>(uint32_t b) { return pvo_bf16_to_f32(static_cast<uint16_t>(b)); }
template <typename T> __device__ __forceinline__ uint32_t os_bits(float x);
template <> __device__ __forceinline__ uint32_t os_bits<pvo_half>(float x) {
  union { _Float16 h; uint16_t u; } c; c.h = static_cast<_Float16>(x); return c.u;
}
template <> __device__ __forceinline__ uint32_t os_bits<pvo_bf16>(float x) { return pvo_f32_to_bf16(x); }
template <typename T> __device__ __forceinline__ void os_unpack8(os_u32x4 v, float f[8]) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) { f[2 * k] = os_val<T>(w[k] & 0xffffu); f[2 * k + 1] = os_val<T>(w[k] >> 16); }
}

// ---------------------------------------------------------------------------
// gate context: one workgroup (384 threads, one per output) per edge; the 128-term dot products run as four independent
// partial sums with the weight loads of 16 terms in flight (a 128-thread version with three dependent chains per thread
// and loads issued one term at a time took 14.5 us for these 1.8 MFLOP)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(384) void gate_context_kernel(const float* __restrict__ part, const float* __restrict__ wg_t,
                                                           const float* __restrict__ gb, float* __restrict__ g, int chunks) {
  __shared__ float glo[128];
  glt::gate_context_384(glo, part, wg_t, gb, g, chunks, blockIdx.x);      // (glo_tile.h)
}

// ---------------------------------------------------------------------------
// eta head: 16 lanes per pixel (8 channels each), 9 taps from global (the K-frame tensor is L2 resident), row reduction by
// DPP.  Row r of the output belongs to frame frame[r] and reads image pos[r] of x (pos[r] < 0: no image - the frame only
// carries inactive edges and keeps its stored damping).  frame == nullptr: plain head, eta[r] = 0.01 softplus(conv(x[r]) + b).
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float os_dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
  return v + __int_as_float(moved);
}

template <typename T>
__global__ __launch_bounds__(256) void eta_head_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ wt,
                                                       const float* __restrict__ bias, const long long* __restrict__ frame,
                                                       const int* __restrict__ pos, float* __restrict__ damping,
                                                       float* __restrict__ eta, int H, int W, float EP, float eta_scale) {
  const int r = blockIdx.y;
  const int HW = H * W;
  const int pix = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  const int k = frame ? pos[r] : r;
  if (pix >= HW) return;
  float e;
  if (k >= 0) {
    const int py = pix / W, px = pix - py * W;
    const uint16_t* xe = x + static_cast<size_t>(k) * HW * 128 + l * 8;
    float acc = 0.0f;
    // the nine taps' activations and weights are requested before the first product (an out-of-image tap reads the centre
    // pixel and is skipped): the tensor is L2 resident, so the kernel's time was nine dependent round trips
    os_u32x4 av[9], wv[9];
    bool ok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
      ok[t] = yy >= 0 && yy < H && xx >= 0 && xx < W;
      const int q = ok[t] ? yy * W + xx : pix;
      av[t] = *reinterpret_cast<const os_u32x4*>(xe + static_cast<size_t>(q) * 128);
      wv[t] = *reinterpret_cast<const os_u32x4*>(wt + t * 128 + l * 8);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (ok[t]) {
        float a[8], w[8];
        os_unpack8<T>(av[t], a);
        os_unpack8<T>(wv[t], w);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = fmaf(a[q], w[q], acc);
      }
    }
    acc = os_dpp_add<0xB1>(acc);     // 16-lane row sum: quad_perm xor 1, xor 2, row_half_mirror, row_mirror
    acc = os_dpp_add<0x4E>(acc);
    acc = os_dpp_add<0x141>(acc);
    acc = os_dpp_add<0x140>(acc);
    const float v = acc + bias[0];
    const float sp = v > 20.0f ? v : log1pf(expf(v));            // torch softplus, beta = 1, threshold = 20
    e = __fmul_rn(0.01f, sp);
  } else {
    e = damping[frame[r] * HW + pix];
  }
  if (l != 0) return;
  if (frame) {
    if (k >= 0) damping[frame[r] * HW + pix] = e;
    eta[static_cast<size_t>(r) * HW + pix] = __fadd_rn(__fmul_rn(eta_scale, e), EP);
  } else {
    eta[static_cast<size_t>(r) * HW + pix] = e;
  }
}

// ---------------------------------------------------------------------------
// 1x1 convolution of a 128-channel tensor: workgroup = 64 rows x 192 output channels; the 64 x 128 input tile sits in LDS
// (272-byte row stride), wave w owns 48 output channels (3 column tiles) with its 12 weight fragments in registers.
// ---------------------------------------------------------------------------
constexpr int kOsStride = 272;

template <typename T>
__global__ __launch_bounds__(256) void conv1x1_c128_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ wt,
                                                           const float* __restrict__ bias, uint16_t* __restrict__ y,
                                                           long long rows, int Cout, int relu) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[c1t::kTileBytes];
  c1t::conv1x1_c128_tile<T>(tile, x, wt, bias, y, rows, Cout, relu, blockIdx.x, blockIdx.y);      // (conv1x1_tile.h)
}

// ---------------------------------------------------------------------------
// correlation encoder on a sampled tensor: rows of 196 16-bit channels (392 bytes: 8-byte aligned) are staged into LDS
// rows of 224 channels (zero padded; 464-byte stride), then 64 rows x 128 outputs on the matrix cores (K = 224: 7 chunks)
// against the same zero-padded [128][224] weight matrix pvo_corr_lookup_encode_tiled reads.
// ---------------------------------------------------------------------------
constexpr int kCeStride = 464;

template <typename T>
__global__ __launch_bounds__(256) void corr_encode_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ wt,
                                                          const float* __restrict__ bias, uint16_t* __restrict__ y, long long rows) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[64 * kCeStride];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const long long r0 = static_cast<long long>(blockIdx.x) * 64;
  for (int id = tid; id < 64 * 56; id += 256) {             // 56 pieces of 8 bytes per padded row
    const int px = id / 56, c = id - px * 56;
    os_u32x2 v = {0u, 0u};
    if (c < 49 && r0 + px < rows) v = *reinterpret_cast<const os_u32x2*>(x + static_cast<size_t>(r0 + px) * 196 + c * 4);
    *reinterpret_cast<os_u32x2*>(tile + px * kCeStride + c * 8) = v;
  }
  os_v4f d[4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g) { d[g][0] = os_v4f{0.f, 0.f, 0.f, 0.f}; d[g][1] = os_v4f{0.f, 0.f, 0.f, 0.f}; }
  __syncthreads();
#pragma unroll 1
  for (int kc = 0; kc < 7; ++kc) {
    os_u32x4 b0 = *reinterpret_cast<const os_u32x4*>(wt + static_cast<size_t>(wave * 32 + li) * 224 + kc * 32 + lk * 8);
    os_u32x4 b1 = *reinterpret_cast<const os_u32x4*>(wt + static_cast<size_t>(wave * 32 + 16 + li) * 224 + kc * 32 + lk * 8);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const os_u32x4 a = *reinterpret_cast<const os_u32x4*>(tile + (g * 16 + li) * kCeStride + kc * 64 + lk * 16);
      d[g][0] = os_mfma<T>(a, b0, d[g][0]);
      d[g][1] = os_mfma<T>(a, b1, d[g][1]);
    }
  }
  __syncthreads();
  const float bb0 = bias ? bias[wave * 32 + li] : 0.0f, bb1 = bias ? bias[wave * 32 + 16 + li] : 0.0f;
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      unsigned char* row = tile + (g * 16 + lk * 4 + r) * kOsStride;
      *reinterpret_cast<uint16_t*>(row + (wave * 32 + li) * 2) = static_cast<uint16_t>(os_bits<T>(fmaxf(d[g][0][r] + bb0, 0.0f)));
      *reinterpret_cast<uint16_t*>(row + (wave * 32 + 16 + li) * 2) = static_cast<uint16_t>(os_bits<T>(fmaxf(d[g][1][r] + bb1, 0.0f)));
    }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int id = tid + 256 * it, px = id >> 4, c = id & 15;
    if (r0 + px < rows)
      *reinterpret_cast<os_u32x4*>(y + static_cast<size_t>(r0 + px) * 128 + c * 8) = *reinterpret_cast<const os_u32x4*>(tile + px * kOsStride + c * 16);
  }
}

// ---------------------------------------------------------------------------
// segment histogram for the panoptic vote: tot[e, s] = pixels of segment s on edge e, dyn[e, s] = those whose UPDATED
// mask (raw_mask + delta_mask) is dynamic on either channel (factor_graph.py:252-261)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void segment_hist_kernel(const int* __restrict__ segm, const float2* __restrict__ raw_mask,
                                                           const uint16_t* __restrict__ heads, int* __restrict__ tot,
                                                           int* __restrict__ dyn, int E, int HW, int S, float dy_thresh) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const bool in = idx < E * HW;
  int key = -1;
  bool d = false;
  if (in) {
    const int e = idx / HW;
    int s = segm[idx];
    s = s < 0 ? 0 : (s >= S ? S - 1 : s);
    const uint32_t q = *reinterpret_cast<const uint32_t*>(heads + static_cast<size_t>(idx) * 8 + 6);
    const float2 rm = raw_mask[idx];
    const float m0 = rm.x + os_val<T>(q & 0xffffu), m1 = rm.y + os_val<T>(q >> 16);
    d = !(1.0f / (1.0f + expf(-m0)) >= dy_thresh) || !(1.0f / (1.0f + expf(-m1)) >= dy_thresh);
    key = e * S + s;
  }
  // wave-aggregated counts: a wave's 64 consecutive pixels lie in two or three segments, so one atomic per (wave, segment)
  // instead of one per pixel - the per-pixel form serialised ~3000 atomics on each of an edge's ~10 addresses (61 us per launch
  // in the full-sequence run of bench.py).  Integer sums: the table is the same whatever the order.
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(in);
  while (todo) {
    const int leader = __ffsll(static_cast<long long>(todo)) - 1;
    const int k = __shfl(key, leader);
    const unsigned long long same = __ballot(in && key == k);
    const unsigned long long dyn_same = __ballot(in && key == k && d);
    if (lane == leader) {
      atomicAdd(tot + k, __popcll(same));
      if (dyn_same) atomicAdd(dyn + k, __popcll(dyn_same));
    }
    todo &= ~same;
  }
}

}  // namespace

extern "C" int pvo_gate_context(const float* glo_part, const float* wg_t, const float* g_bias, float* g,
                                int E, int chunks, void* stream) {
  if (E < 0 || chunks <= 0) return PVO_EINVAL;
  if (E == 0) return PVO_OK;
  if (!glo_part || !wg_t || !g_bias || !g) return PVO_EINVAL;
  hipLaunchKernelGGL(gate_context_kernel, dim3(E), dim3(384), 0, pvo_stream(stream), glo_part, wg_t, g_bias, g, chunks);
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_eta_head(const void* x, const void* w_taps, const float* bias, const int64_t* frame, const int* pos,
                            float* damping, float* eta, int R, int H, int W, float EP, float eta_scale, int dtype, void* stream) {
  if (R < 0 || H < 0 || W < 0) return PVO_EINVAL;
  if (R == 0 || H == 0 || W == 0) return PVO_OK;
  if (!x || !w_taps || !bias || !eta || R > 65535) return PVO_EINVAL;
  if (frame && (!pos || !damping)) return PVO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_taps)) & 15) return PVO_EINVAL;
  hipStream_t st = pvo_stream(stream);
  const dim3 grid((H * W + 15) / 16, R);
  if (dtype == PVO_F16)
    hipLaunchKernelGGL(eta_head_kernel<pvo_half>, grid, dim3(256), 0, st, static_cast<const uint16_t*>(x), static_cast<const uint16_t*>(w_taps),
                       bias, reinterpret_cast<const long long*>(frame), pos, damping, eta, H, W, EP, eta_scale);
  else if (dtype == PVO_BF16)
    hipLaunchKernelGGL(eta_head_kernel<pvo_bf16>, grid, dim3(256), 0, st, static_cast<const uint16_t*>(x), static_cast<const uint16_t*>(w_taps),
                       bias, reinterpret_cast<const long long*>(frame), pos, damping, eta, H, W, EP, eta_scale);
  else return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_conv1x1_c128(const void* x, const void* w, const float* bias, void* y, long long rows, int Cout, int relu,
                                int dtype, void* stream) {
  if (rows < 0) return PVO_EINVAL;
  if (Cout <= 0 || Cout % 192) return PVO_EUNSUPPORTED;
  if (rows == 0) return PVO_OK;
  if (!x || !w || !y || rows > (1LL << 31) - 64) return PVO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(y)) & 15) return PVO_EINVAL;
  hipStream_t st = pvo_stream(stream);
  const dim3 grid(static_cast<unsigned>((rows + 63) / 64), Cout / 192);
  if (dtype == PVO_F16)
    hipLaunchKernelGGL(conv1x1_c128_kernel<pvo_half>, grid, dim3(256), 0, st, static_cast<const uint16_t*>(x), static_cast<const uint16_t*>(w), bias, static_cast<uint16_t*>(y), rows, Cout, relu);
  else if (dtype == PVO_BF16)
    hipLaunchKernelGGL(conv1x1_c128_kernel<pvo_bf16>, grid, dim3(256), 0, st, static_cast<const uint16_t*>(x), static_cast<const uint16_t*>(w), bias, static_cast<uint16_t*>(y), rows, Cout, relu);
  else return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_corr_encode(const void* corr, const void* enc_weight, const float* enc_bias, void* y, long long rows,
                               int dtype, void* stream) {
  if (rows < 0) return PVO_EINVAL;
  if (rows == 0) return PVO_OK;
  if (!corr || !enc_weight || !y || rows > (1LL << 31) - 64) return PVO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(corr) & 7) || ((reinterpret_cast<uintptr_t>(enc_weight) | reinterpret_cast<uintptr_t>(y)) & 15)) return PVO_EINVAL;
  hipStream_t st = pvo_stream(stream);
  const dim3 grid(static_cast<unsigned>((rows + 63) / 64));
  if (dtype == PVO_F16)
    hipLaunchKernelGGL(corr_encode_kernel<pvo_half>, grid, dim3(256), 0, st, static_cast<const uint16_t*>(corr), static_cast<const uint16_t*>(enc_weight), enc_bias, static_cast<uint16_t*>(y), rows);
  else if (dtype == PVO_BF16)
    hipLaunchKernelGGL(corr_encode_kernel<pvo_bf16>, grid, dim3(256), 0, st, static_cast<const uint16_t*>(corr), static_cast<const uint16_t*>(enc_weight), enc_bias, static_cast<uint16_t*>(y), rows);
  else return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

// `own_gap`: tot and dyn were carved from ONE workspace by the caller (pvo_graph_update, update_exec.hip carve_up) and the bytes between
// them are that workspace's padding - one fill clears both tables.  The public entry point never assumes that: two tensors of a caching
// allocator may have a live stranger between them.
static int segment_hist_impl(const int* segm, const float* raw_mask, const void* heads, int* tot, int* dyn,
                             int E, int HW, int S, float dy_thresh, int dtype, void* stream, bool own_gap) {
  if (E < 0 || HW < 0 || S <= 0) return PVO_EINVAL;
  const long long n = static_cast<long long>(E) * HW;
  if (n == 0) return PVO_OK;
  if (!segm || !raw_mask || !heads || !tot || !dyn || n >= (1LL << 31)) return PVO_EINVAL;
  hipStream_t st = pvo_stream(stream);
  const size_t tbytes = sizeof(int) * static_cast<size_t>(E) * S;
  const char *t0 = reinterpret_cast<const char*>(tot), *d0 = reinterpret_cast<const char*>(dyn);
  if (own_gap && d0 >= t0 + tbytes && static_cast<size_t>(d0 - t0) <= tbytes + 4096) {
    if (hipMemsetAsync(tot, 0, static_cast<size_t>(d0 - t0) + tbytes, st) != hipSuccess) return PVO_ELAUNCH;
  } else {
    if (hipMemsetAsync(tot, 0, tbytes, st) != hipSuccess) return PVO_ELAUNCH;
    if (hipMemsetAsync(dyn, 0, tbytes, st) != hipSuccess) return PVO_ELAUNCH;
  }
  const dim3 grid(static_cast<unsigned>((n + 255) / 256));
  if (dtype == PVO_F16)
    hipLaunchKernelGGL(segment_hist_kernel<pvo_half>, grid, dim3(256), 0, st, segm, reinterpret_cast<const float2*>(raw_mask), static_cast<const uint16_t*>(heads), tot, dyn, E, HW, S, dy_thresh);
  else if (dtype == PVO_BF16)
    hipLaunchKernelGGL(segment_hist_kernel<pvo_bf16>, grid, dim3(256), 0, st, segm, reinterpret_cast<const float2*>(raw_mask), static_cast<const uint16_t*>(heads), tot, dyn, E, HW, S, dy_thresh);
  else return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_segment_hist(const int* segm, const float* raw_mask, const void* heads, int* tot, int* dyn,
                                int E, int HW, int S, float dy_thresh, int dtype, void* stream) {
  return segment_hist_impl(segm, raw_mask, heads, tot, dyn, E, HW, S, dy_thresh, dtype, stream, false);
}

// internal (update_exec.hip): both tables lie in pvo_graph_update's workspace, separated by its alignment padding only
int pvo_segment_hist_ws(const int* segm, const float* raw_mask, const void* heads, int* tot, int* dyn,
                        int E, int HW, int S, float dy_thresh, int dtype, void* stream) {
  return segment_hist_impl(segm, raw_mask, heads, tot, dyn, E, HW, S, dy_thresh, dtype, stream, true);
}
