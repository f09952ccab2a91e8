// encoder_ops.hip — what follows every convolution of the per-frame encoders, as ONE kernel per layer.
//
// The reference's BasicEncoder (VO_Module/droid_slam/modules/extractor.py:116-201; ResidualBlock :6-56) runs once per frame in
// MotionFilter.track (motion_filter.py:52-60) under fp16 autocast.  Per convolution PyTorch issues: the convolution (MIOpen), the
// bias add, instance norm as batch_norm_collect_statistics + a 256-element kernel for invstd + batch_norm_transform_input, ReLU, and
// per residual block an add and another ReLU - ~95 kernels of 4-20 us per network even when replayed from a HIP graph, 0.9 ms of a
// tracked frame's 1.4 ms (bench.py `sequence`, profiles/r05_sequence_timeline.txt).  pvo_bias_norm_act is everything between two
// convolutions:
//     t = round16(x + bias[c])                                   (the convolution's bias, as `conv(x) + b` rounds it)
//     t = round16((t - mean) * rsqrt(var + eps))   if norm        (InstanceNorm2d, affine = False: biased variance over the plane,
//                                                                 fp32 statistics - what batch_norm computes for a 16-bit input)
//     t = max(t, 0)                                if relu_inner
//     t = round16(residual + t)                    if residual   (ResidualBlock's x + y)
//     t = max(t, 0)                                if relu_outer
// x, residual, y: [N, C, HW] planes (NCHW, contiguous), 16-bit; one workgroup per plane: the plane (6-97 KB) is read from L2 three
// times (mean, variance about the mean, output) - the statistics are two-pass, not E[x^2] - E[x]^2.
#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ float eo_val(uint16_t b);
template <> __device__ __forceinline__ float eo_val<pvo_half>(uint16_t b) {
  union { uint16_t u; _Float16 h; } c; c.u = b; return static_cast<float>(c.h);
}
template <> __device__ __forceinline__ float eo_val<pvo_bf16>(uint16_t b) { return pvo_bf16_to_f32(b); }
template <typename T> __device__ __forceinline__ uint16_t eo_bits(float x);
template <> __device__ __forceinline__ uint16_t eo_bits<pvo_half>(float x) {
  union { _Float16 h; uint16_t u; } c; c.h = static_cast<_Float16>(x); return c.u;
}
template <> __device__ __forceinline__ uint16_t eo_bits<pvo_bf16>(float x) { return pvo_f32_to_bf16(x); }
template <typename T> __device__ __forceinline__ float eo_round(float x) { return eo_val<T>(eo_bits<T>(x)); }

// sum over the workgroup (every thread gets it); `red` holds one float per wave
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = pvo_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();                                    // (`red` may still be read from the previous sum)
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.0f;
  for (int w = 0; w < nw; ++w) s += red[w];
  return s;
}

template <typename T>
__global__ void bias_norm_act_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ bias, const uint16_t* __restrict__ residual,
                                     uint16_t* __restrict__ y, int C, int HW, int norm, float eps, int relu_inner, int relu_outer) {
  __shared__ float red[16];
  const long long plane = blockIdx.x;
  const int c = static_cast<int>(plane % C);
  const uint16_t* xp = x + plane * HW;
  const uint16_t* rp = residual ? residual + plane * HW : nullptr;
  uint16_t* yp = y + plane * HW;
  const float b = bias ? eo_val<T>(bias[c]) : 0.0f;
  const bool pairs = (HW & 1) == 0;                   // planes of an even pixel count are read / written two values at a time
  float mean = 0.0f, invstd = 1.0f;
  if (norm) {
    float s = 0.0f;
    if (pairs) {
      for (int i = threadIdx.x; i < HW / 2; i += blockDim.x) {
        const uint32_t v = reinterpret_cast<const uint32_t*>(xp)[i];
        s += eo_round<T>(eo_val<T>(static_cast<uint16_t>(v & 0xffffu)) + b) + eo_round<T>(eo_val<T>(static_cast<uint16_t>(v >> 16)) + b);
      }
    } else {
      for (int i = threadIdx.x; i < HW; i += blockDim.x) s += eo_round<T>(eo_val<T>(xp[i]) + b);
    }
    mean = block_sum(s, red) / static_cast<float>(HW);
    float q = 0.0f;
    if (pairs) {
      for (int i = threadIdx.x; i < HW / 2; i += blockDim.x) {
        const uint32_t v = reinterpret_cast<const uint32_t*>(xp)[i];
        const float d0 = eo_round<T>(eo_val<T>(static_cast<uint16_t>(v & 0xffffu)) + b) - mean;
        const float d1 = eo_round<T>(eo_val<T>(static_cast<uint16_t>(v >> 16)) + b) - mean;
        q += d0 * d0 + d1 * d1;
      }
    } else {
      for (int i = threadIdx.x; i < HW; i += blockDim.x) { const float d = eo_round<T>(eo_val<T>(xp[i]) + b) - mean; q += d * d; }
    }
    invstd = 1.0f / sqrtf(block_sum(q, red) / static_cast<float>(HW) + eps);
  }
  auto finish = [&](uint16_t xv, uint16_t rv) -> uint16_t {
    float t = eo_round<T>(eo_val<T>(xv) + b);
    if (norm) t = eo_round<T>((t - mean) * invstd);
    if (relu_inner) t = fmaxf(t, 0.0f);
    if (rp) t = eo_round<T>(eo_val<T>(rv) + t);
    if (relu_outer) t = fmaxf(t, 0.0f);
    return eo_bits<T>(t);
  };
  if (pairs) {
    for (int i = threadIdx.x; i < HW / 2; i += blockDim.x) {
      const uint32_t v = reinterpret_cast<const uint32_t*>(xp)[i];
      const uint32_t r = rp ? reinterpret_cast<const uint32_t*>(rp)[i] : 0u;
      const uint32_t lo = finish(static_cast<uint16_t>(v & 0xffffu), static_cast<uint16_t>(r & 0xffffu));
      const uint32_t hi = finish(static_cast<uint16_t>(v >> 16), static_cast<uint16_t>(r >> 16));
      reinterpret_cast<uint32_t*>(yp)[i] = lo | (hi << 16);
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += blockDim.x) yp[i] = finish(xp[i], rp ? rp[i] : static_cast<uint16_t>(0));
  }
}

// The same for LARGE planes (the encoders' first layers: 32 planes of 120 x 404 = 48 480 pixels per image): one workgroup per plane is 32
// workgroups on 256 compute units, each walking its plane three times - 23 us per layer, latency.  Here a plane is cut into S slices:
// bna_stats_kernel leaves (mean, sum of squared deviations about it) of every slice in `ws`, bna_apply_kernel combines a plane's S pairs
// in index order (Chan et al.: exact in exact arithmetic, fp32 here - the statistics differ from the one-workgroup kernel's in the last
// bits, deterministically) and finishes its slice.  Without normalisation only the second kernel runs.
template <typename T>
__global__ __launch_bounds__(256) void bna_stats_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ bias, float* __restrict__ ws,
                                                        int C, int HW, int S, int L) {
  __shared__ float red[16];
  const long long plane = blockIdx.y;
  const int s = blockIdx.x, lo = s * L, hi = min(HW, lo + L), n = max(hi - lo, 0);
  const uint16_t* xp = x + plane * HW;
  const float b = bias ? eo_val<T>(bias[static_cast<int>(plane % C)]) : 0.0f;
  float sum = 0.0f;
  for (int i = lo + threadIdx.x; i < hi; i += 256) sum += eo_round<T>(eo_val<T>(xp[i]) + b);
  const float mean = n > 0 ? block_sum(sum, red) / static_cast<float>(n) : 0.0f;
  float q = 0.0f;
  for (int i = lo + threadIdx.x; i < hi; i += 256) { const float d = eo_round<T>(eo_val<T>(xp[i]) + b) - mean; q += d * d; }
  q = block_sum(q, red);
  if (threadIdx.x == 0) { ws[(plane * S + s) * 2] = mean; ws[(plane * S + s) * 2 + 1] = q; }
}

template <typename T>
__global__ __launch_bounds__(256) void bna_apply_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ bias, const uint16_t* __restrict__ residual,
                                                        uint16_t* __restrict__ y, const float* __restrict__ ws, int C, int HW, int S, int L, int norm,
                                                        float eps, int relu_inner, int relu_outer) {
  const long long plane = blockIdx.y;
  const int s = blockIdx.x, lo = s * L, hi = min(HW, lo + L);
  const uint16_t* xp = x + plane * HW;
  const uint16_t* rp = residual ? residual + plane * HW : nullptr;
  uint16_t* yp = y + plane * HW;
  const float b = bias ? eo_val<T>(bias[static_cast<int>(plane % C)]) : 0.0f;
  float mean = 0.0f, invstd = 1.0f;
  if (norm) {
    float cnt = 0.0f, m2 = 0.0f;                       // (every thread the same S steps: no communication)
    for (int k = 0; k < S; ++k) {
      const float nk = static_cast<float>(max(min(HW, (k + 1) * L) - k * L, 0));
      if (nk <= 0.0f) continue;
      const float mk = ws[(plane * S + k) * 2], qk = ws[(plane * S + k) * 2 + 1];
      const float tot = cnt + nk, delta = mk - mean;
      mean += delta * (nk / tot);
      m2 += qk + delta * delta * (cnt * nk / tot);
      cnt = tot;
    }
    invstd = 1.0f / sqrtf(m2 / static_cast<float>(HW) + eps);
  }
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    float t = eo_round<T>(eo_val<T>(xp[i]) + b);
    if (norm) t = eo_round<T>((t - mean) * invstd);
    if (relu_inner) t = fmaxf(t, 0.0f);
    if (rp) t = eo_round<T>(eo_val<T>(rp[i]) + t);
    if (relu_outer) t = fmaxf(t, 0.0f);
    yp[i] = eo_bits<T>(t);
  }
}

// The encoders' last layer, Conv2d(128, output_dim, 1) (extractor.py:139,199), on NCHW planes: y[n][co][p] = round16(round16(sum_ci
// w[co][ci] x[n][ci][p]) + bias[co]).  MIOpen runs it as an NHWC implicit GEMM that splits K over workgroups and adds the partial sums
// with atomics: three consecutive calls on the same input gave three different feature maps (tools/determinism_probe.py), and with
// them every run of a sequence its own trajectory.  Here one thread owns 4 pixels x 4 output channels and adds the 128 products in
// index order in fp32 - the same bits every time; no layout transposes (MIOpen: three around its kernel).
constexpr int kC1Pix = 64, kC1Co = 64, kC1Ci = 32;
template <typename T>
__global__ __launch_bounds__(256) void conv1x1_planes_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, const uint16_t* __restrict__ bias,
                                                             uint16_t* __restrict__ y, int Cin, int Cout, int Win, int HWin, int Wout, int HWout, int stride) {
  __shared__ uint16_t xs[kC1Ci][kC1Pix];
  __shared__ uint16_t ws[kC1Ci][kC1Co + 4];                       // [ci][co]: a thread's four output channels are adjacent
  __shared__ int src[kC1Pix];                                     // input offset of the tile's output pixels (stride: every s-th row / column)
  const int p0 = blockIdx.x * kC1Pix, co0 = blockIdx.y * kC1Co;
  const long long n = blockIdx.z;
  const uint16_t* xn = x + n * Cin * static_cast<long long>(HWin);
  const int tp = (threadIdx.x & 15) * 4, tc = (threadIdx.x >> 4) * 4;
  if (threadIdx.x < kC1Pix) {
    const int p = p0 + threadIdx.x;
    src[threadIdx.x] = p < HWout ? (p / Wout) * stride * Win + (p % Wout) * stride : -1;
  }
  float acc[4][4] = {};
  for (int c0 = 0; c0 < Cin; c0 += kC1Ci) {
    __syncthreads();
    for (int i = threadIdx.x; i < kC1Ci * kC1Pix; i += 256) {
      const int ci = i / kC1Pix, p = i % kC1Pix;
      const int o = src[p];
      xs[ci][p] = o >= 0 ? xn[static_cast<long long>(c0 + ci) * HWin + o] : static_cast<uint16_t>(0);
    }
    for (int i = threadIdx.x; i < kC1Co * kC1Ci; i += 256) {
      const int co = i / kC1Ci, ci = i % kC1Ci;
      ws[ci][co] = w[static_cast<long long>(co0 + co) * Cin + c0 + ci];
    }
    __syncthreads();
#pragma unroll 4
    for (int ci = 0; ci < kC1Ci; ++ci) {
      float xv[4], wv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { xv[k] = eo_val<T>(xs[ci][tp + k]); wv[k] = eo_val<T>(ws[ci][tc + k]); }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(wv[a], xv[b], acc[a][b]);
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int co = co0 + tc + a;
    const float bv = bias ? eo_val<T>(bias[co]) : 0.0f;
    uint16_t* yr = y + (n * Cout + co) * static_cast<long long>(HWout) + p0 + tp;
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (p0 + tp + b < HWout) yr[b] = eo_bits<T>(eo_round<T>(acc[a][b]) + bv);
  }
}

// A frame as the stream hands it over - [3][H][W] BGR, 0..255, int32 / uint8 / float32 - to the encoders' input: RGB, ((v / 255) - mean) /
// std in fp32 (the operations and their order of motion_filter.py:52-54), rounded to the 16-bit type.  One launch for the six element-wise
// kernels PyTorch issues (flip, float, divide, subtract, divide, cast) in front of EACH encoder.
template <typename T, typename IN>
__global__ void frame_normalise_kernel(const IN* __restrict__ img, uint16_t* __restrict__ out, int HW, float m0, float m1, float m2, float s0, float s1, float s2) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = static_cast<float>(img[static_cast<size_t>(2 - c) * HW + p]) / 255.0f;
    out[static_cast<size_t>(c) * HW + p] = eo_bits<T>((v - mean[c]) / stdv[c]);
  }
}

}  // namespace

extern "C" int pvo_conv1x1_planes(const void* x, const void* w, const void* bias, void* y, int N, int Cin, int Cout, int Hin, int Win, int stride,
                                  int dtype, void* stream) {
  if (N < 0 || Cin <= 0 || Cout <= 0 || Hin < 0 || Win < 0 || stride < 1) return PVO_EINVAL;
  if (N == 0 || Hin == 0 || Win == 0) return PVO_OK;
  if (!x || !w || !y) return PVO_EINVAL;
  if ((Cin % kC1Ci) != 0 || (Cout % kC1Co) != 0 || N > 65535) return PVO_EUNSUPPORTED;
  const int Hout = (Hin - 1) / stride + 1, Wout = (Win - 1) / stride + 1;
  const dim3 grid((Hout * Wout + kC1Pix - 1) / kC1Pix, Cout / kC1Co, N);
  hipStream_t st = pvo_stream(stream);
  if (dtype == PVO_F16)
    hipLaunchKernelGGL(conv1x1_planes_kernel<pvo_half>, grid, dim3(256), 0, st, static_cast<const uint16_t*>(x), static_cast<const uint16_t*>(w),
                       static_cast<const uint16_t*>(bias), static_cast<uint16_t*>(y), Cin, Cout, Win, Hin * Win, Wout, Hout * Wout, stride);
  else if (dtype == PVO_BF16)
    hipLaunchKernelGGL(conv1x1_planes_kernel<pvo_bf16>, grid, dim3(256), 0, st, static_cast<const uint16_t*>(x), static_cast<const uint16_t*>(w),
                       static_cast<const uint16_t*>(bias), static_cast<uint16_t*>(y), Cin, Cout, Win, Hin * Win, Wout, Hout * Wout, stride);
  else return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

// in_kind: 0 = int32, 1 = uint8, 2 = float32
extern "C" int pvo_frame_normalise(const void* img, void* out, int H, int W, const float* mean3, const float* std3, int in_kind, int dtype, void* stream) {
  if (H < 0 || W < 0 || !mean3 || !std3) return PVO_EINVAL;
  if (H == 0 || W == 0) return PVO_OK;
  if (!img || !out) return PVO_EINVAL;
  const int HW = H * W;
  const dim3 grid((HW + 255) / 256);
  hipStream_t st = pvo_stream(stream);
#define PVO_FN_LAUNCH(T, IN) hipLaunchKernelGGL((frame_normalise_kernel<T, IN>), grid, dim3(256), 0, st, static_cast<const IN*>(img), static_cast<uint16_t*>(out), HW, \
                                                mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2])
  if (dtype == PVO_F16) {
    if (in_kind == 0) PVO_FN_LAUNCH(pvo_half, int); else if (in_kind == 1) PVO_FN_LAUNCH(pvo_half, uint8_t); else if (in_kind == 2) PVO_FN_LAUNCH(pvo_half, float); else return PVO_EUNSUPPORTED;
  } else if (dtype == PVO_BF16) {
    if (in_kind == 0) PVO_FN_LAUNCH(pvo_bf16, int); else if (in_kind == 1) PVO_FN_LAUNCH(pvo_bf16, uint8_t); else if (in_kind == 2) PVO_FN_LAUNCH(pvo_bf16, float); else return PVO_EUNSUPPORTED;
  } else return PVO_EUNSUPPORTED;
#undef PVO_FN_LAUNCH
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

// slices per plane of the split form (0: planes of this size keep the one-workgroup kernel)
extern "C" int pvo_bias_norm_act_slices(int HW) { return HW >= 16384 ? 16 : (HW >= 8192 ? 4 : 0); }

extern "C" int pvo_bias_norm_act_split(const void* x, const void* bias, const void* residual, void* y, long long planes, int C, int HW,
                                       int norm, float eps, int relu_inner, int relu_outer, int dtype, float* ws, size_t ws_floats, void* stream) {
  if (planes < 0 || C <= 0 || HW < 0 || (planes % C) != 0) return PVO_EINVAL;
  if (planes == 0 || HW == 0) return PVO_OK;
  const int S = pvo_bias_norm_act_slices(HW);
  if (!x || !y || planes > 65535 || S == 0) return PVO_EINVAL;
  if (norm && (!ws || ws_floats < static_cast<size_t>(planes) * S * 2)) return PVO_EWORKSPACE;
  const int L = (HW + S - 1) / S;
  const dim3 grid(S, static_cast<unsigned>(planes));
  hipStream_t st = pvo_stream(stream);
#define PVO_BNA_SPLIT(T)                                                                                                                        \
  do {                                                                                                                                          \
    if (norm) hipLaunchKernelGGL(bna_stats_kernel<T>, grid, dim3(256), 0, st, static_cast<const uint16_t*>(x), static_cast<const uint16_t*>(bias), ws, C, HW, S, L); \
    hipLaunchKernelGGL(bna_apply_kernel<T>, grid, dim3(256), 0, st, static_cast<const uint16_t*>(x), static_cast<const uint16_t*>(bias),        \
                       static_cast<const uint16_t*>(residual), static_cast<uint16_t*>(y), ws, C, HW, S, L, norm, eps, relu_inner, relu_outer);   \
  } while (0)
  if (dtype == PVO_F16) PVO_BNA_SPLIT(pvo_half);
  else if (dtype == PVO_BF16) PVO_BNA_SPLIT(pvo_bf16);
  else return PVO_EUNSUPPORTED;
#undef PVO_BNA_SPLIT
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_bias_norm_act(const void* x, const void* bias, const void* residual, void* y, long long planes, int C, int HW,
                                 int norm, float eps, int relu_inner, int relu_outer, int dtype, void* stream) {
  if (planes < 0 || C <= 0 || HW < 0 || (planes % C) != 0) return PVO_EINVAL;
  if (planes == 0 || HW == 0) return PVO_OK;
  if (!x || !y || planes > 0x7fffffffLL) return PVO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 3) return PVO_EINVAL;
  const int threads = HW >= 16384 ? 1024 : (HW >= 2048 ? 512 : 256);
  const dim3 grid(static_cast<unsigned>(planes));
  hipStream_t st = pvo_stream(stream);
  if (dtype == PVO_F16)
    hipLaunchKernelGGL(bias_norm_act_kernel<pvo_half>, grid, dim3(threads), 0, st, static_cast<const uint16_t*>(x), static_cast<const uint16_t*>(bias),
                       static_cast<const uint16_t*>(residual), static_cast<uint16_t*>(y), C, HW, norm, eps, relu_inner, relu_outer);
  else if (dtype == PVO_BF16)
    hipLaunchKernelGGL(bias_norm_act_kernel<pvo_bf16>, grid, dim3(threads), 0, st, static_cast<const uint16_t*>(x), static_cast<const uint16_t*>(bias),
                       static_cast<const uint16_t*>(residual), static_cast<uint16_t*>(y), C, HW, norm, eps, relu_inner, relu_outer);
  else return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}
