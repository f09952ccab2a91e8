// gru_fused.hip — the element-wise half of the ConvGRU update operator, fused for gfx950.
//
// The reference's ConvGRU.forward (VO_Module/droid_slam/modules/gru.py:19-32) is, around its
// three 3x3 convolutions, a chain of ~14 element-wise / concat launches over 28-100 MB tensors
// (cat, cat, sigmoid, mul, mean, add, sigmoid, mul, cat, add, tanh, mul, mul, add) — 45 % of the
// GPU time of a graph update once the convolutions run on MIOpen's MFMA kernels.  These four
// kernels replace that chain; the convolutions stay in MIOpen:
//
//   gru_glo      glo[e,c]   = mean_p( sigmoid(wn[e,p,c]) * net[e,p,c] )              gru.py:23-24
//   gru_assemble X[e,p,:]   = [ net | inp | relu(corr_feat) | relu(flow_feat) ]     gru.py:20-21 + encoder ReLUs
//   gru_gate     z = sigmoid(zr[..,:128] + gz);  X[e,p,:128] = sigmoid(zr[..,128:] + gr) * net   gru.py:26-28
//   gru_out      net = (1-z)*net + z*tanh(q + gq)                                    gru.py:28-31
//
// X is ONE persistent 448-channel channels-last buffer: the z/r convolution reads it, gru_gate
// then overwrites its first 128 channels with r*net, and the q convolution reads the same buffer,
// so neither torch.cat of gru.py:20-21,28 is materialised twice.
// Layout: channels-last fp16/bf16 rows ([E, H*W, C]); 8 channels (16 B) per thread per access.
// Arithmetic in fp32, one rounding to the storage type per output.
#include "common.h"

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

// glo[e,c] += (1/HW) * sum over this block's pixel chunk; glo zeroed by the host wrapper
template <typename T>
__global__ __launch_bounds__(256) void gru_glo_kernel(const uint16_t* __restrict__ wn, const uint16_t* __restrict__ net,
                                                      const float* __restrict__ bias, float* __restrict__ glo, int HW, int C, int chunk) {
  __shared__ float red[16][129];
  const int e = blockIdx.y;
  const int cg = threadIdx.x & 15, pl = threadIdx.x >> 4;      // C == 128: 16 groups of 8 channels
  const int p0 = blockIdx.x * chunk, p1 = min(p0 + chunk, HW);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float bw[8];
  load8f(bias ? bias + cg * 8 : nullptr, bw);
  for (int p = p0 + pl; p < p1; p += 16) {
    const long long o = (static_cast<long long>(e) * HW + p) * C + cg * 8;
    float a[8], b[8];
    H8<T>::unpack(*reinterpret_cast<const u32x4*>(wn + o), a);
    H8<T>::unpack(*reinterpret_cast<const u32x4*>(net + o), b);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += sigmoidf_(a[k] + bw[k]) * b[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[pl][cg * 8 + k] = acc[k];
  __syncthreads();
  if (threadIdx.x < 128) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += red[r][threadIdx.x];
    atomicAdd(glo + static_cast<long long>(e) * C + threadIdx.x, s / static_cast<float>(HW));
  }
}

// X [E*HW, 448] <- [net(128) | inp(128) | relu(cf)(128) | relu(ff)(64)]
template <typename T>
__global__ __launch_bounds__(256) void gru_assemble_kernel(const uint16_t* __restrict__ net, const uint16_t* __restrict__ inp,
                                                           const uint16_t* __restrict__ cf, const uint16_t* __restrict__ ff,
                                                           const float* __restrict__ bc, const float* __restrict__ bf,
                                                           uint16_t* __restrict__ X, long long rows, int with_inp) {
  // with_inp == 0: X has 320 channels [net | relu(cf) | relu(ff)] (the inp block's convolution is precomputed)
  const unsigned cpr = with_inp ? 56u : 40u;                  // 16-byte chunks per row of X
  const unsigned total = static_cast<unsigned>(rows) * cpr;
  for (unsigned id = blockIdx.x * 256u + threadIdx.x; id < total; id += gridDim.x * 256u) {
    const unsigned row = id / cpr;
    const int chx = static_cast<int>(id - row * cpr);          // chunk inside X
    const int ch = (with_inp || chx < 16) ? chx : chx + 16;    // chunk in the 448-channel numbering
    u32x4 v;
    if (ch < 16) v = *reinterpret_cast<const u32x4*>(net + static_cast<size_t>(row) * 128 + ch * 8);
    else if (ch < 32) v = *reinterpret_cast<const u32x4*>(inp + static_cast<size_t>(row) * 128 + (ch - 16) * 8);
    else {
      v = (ch < 48) ? *reinterpret_cast<const u32x4*>(cf + static_cast<size_t>(row) * 128 + (ch - 32) * 8)
                    : *reinterpret_cast<const u32x4*>(ff + static_cast<size_t>(row) * 64 + (ch - 48) * 8);
      float f[8];
      H8<T>::unpack(v, f);
      float bb[8];
      load8f((ch < 48) ? (bc ? bc + (ch - 32) * 8 : nullptr) : (bf ? bf + (ch - 48) * 8 : nullptr), bb);
#pragma unroll
      for (int k = 0; k < 8; ++k) f[k] = fmaxf(f[k] + bb[k], 0.0f);
      v = H8<T>::pack(f);
    }
    *reinterpret_cast<u32x4*>(X + static_cast<size_t>(row) * (cpr * 8) + chx * 8) = v;
  }
}

// zr [E*HW, 256], g [E, 384] fp32 (z | r | q context), net [E*HW,128]; writes Z [E*HW,128] and X[:, :128] = r*net
template <typename T>
__global__ __launch_bounds__(256) void gru_gate_kernel(const uint16_t* __restrict__ zr, const float* __restrict__ g,
                                                       const uint16_t* __restrict__ net, uint16_t* __restrict__ Z,
                                                       uint16_t* __restrict__ X, long long rows, int HW,
                                                       const uint16_t* __restrict__ P, int xc) {
  const unsigned total = static_cast<unsigned>(rows) * 16u;
  for (unsigned id = blockIdx.x * 256u + threadIdx.x; id < total; id += gridDim.x * 256u) {
    const size_t row = id >> 4;
    const int ch = static_cast<int>(id & 15u);
    const int e = static_cast<int>((id >> 4) / static_cast<unsigned>(HW));
    float a[8], b[8], n[8], z[8], rn[8];
    H8<T>::unpack(*reinterpret_cast<const u32x4*>(zr + row * 256 + ch * 8), a);
    H8<T>::unpack(*reinterpret_cast<const u32x4*>(zr + row * 256 + 128 + ch * 8), b);
    H8<T>::unpack(*reinterpret_cast<const u32x4*>(net + row * 128 + ch * 8), n);
    float gz[8], gr[8];
    load8f(g + static_cast<long long>(e) * 384 + ch * 8, gz);
    load8f(g + static_cast<long long>(e) * 384 + 128 + ch * 8, gr);
    if (P) {   // precomputed convolution of the (static) inp block, [rows, 256]
      float pz[8], pr[8];
      H8<T>::unpack(*reinterpret_cast<const u32x4*>(P + row * 256 + ch * 8), pz);
      H8<T>::unpack(*reinterpret_cast<const u32x4*>(P + row * 256 + 128 + ch * 8), pr);
#pragma unroll
      for (int k = 0; k < 8; ++k) { gz[k] += pz[k]; gr[k] += pr[k]; }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      z[k] = sigmoidf_(a[k] + gz[k]);
      rn[k] = sigmoidf_(b[k] + gr[k]) * n[k];
    }
    *reinterpret_cast<u32x4*>(Z + row * 128 + ch * 8) = H8<T>::pack(z);
    *reinterpret_cast<u32x4*>(X + row * xc + ch * 8) = H8<T>::pack(rn);
  }
}

// net_out = (1-z)*net + z*tanh(q + gq)
template <typename T>
__global__ __launch_bounds__(256) void gru_out_kernel(const uint16_t* __restrict__ q, const float* __restrict__ g,
                                                      const uint16_t* __restrict__ Z, const uint16_t* __restrict__ net,
                                                      uint16_t* __restrict__ out, long long rows, int HW,
                                                      const uint16_t* __restrict__ P) {
  const unsigned total = static_cast<unsigned>(rows) * 16u;
  for (unsigned id = blockIdx.x * 256u + threadIdx.x; id < total; id += gridDim.x * 256u) {
    const size_t row = id >> 4;
    const int ch = static_cast<int>(id & 15u);
    const int e = static_cast<int>((id >> 4) / static_cast<unsigned>(HW));
    float a[8], z[8], n[8], o[8];
    H8<T>::unpack(*reinterpret_cast<const u32x4*>(q + row * 128 + ch * 8), a);
    H8<T>::unpack(*reinterpret_cast<const u32x4*>(Z + row * 128 + ch * 8), z);
    H8<T>::unpack(*reinterpret_cast<const u32x4*>(net + row * 128 + ch * 8), n);
    float gq[8];
    load8f(g + static_cast<long long>(e) * 384 + 256 + ch * 8, gq);
    if (P) {
      float pq[8];
      H8<T>::unpack(*reinterpret_cast<const u32x4*>(P + row * 128 + ch * 8), pq);
#pragma unroll
      for (int k = 0; k < 8; ++k) gq[k] += pq[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (1.0f - z[k]) * n[k] + z[k] * tanhf(a[k] + gq[k]);
    *reinterpret_cast<u32x4*>(out + row * 128 + ch * 8) = H8<T>::pack(o);
  }
}

// x[row, c] = act(x[row, c] + bias[c]) in place; C % 8 == 0.  Four independent 16-byte chunks per thread,
// all loads issued before the first store (in-place, so the compiler may not reorder them itself).
template <typename T>
__global__ __launch_bounds__(256) void bias_act_kernel(uint16_t* __restrict__ x, const float* __restrict__ bias,
                                                       long long rows, int C, int relu) {
  const unsigned cpr = static_cast<unsigned>(C) >> 3;
  const unsigned total = static_cast<unsigned>(rows) * cpr;
  const unsigned stride = gridDim.x * 256u;
  for (unsigned base = blockIdx.x * 256u + threadIdx.x; base < total; base += 4u * stride) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned id = base + u * stride;
      if (id < total) v[u] = *reinterpret_cast<const u32x4*>(x + static_cast<size_t>(id) * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned id = base + u * stride;
      if (id < total) {
        const int ch = static_cast<int>(id % cpr);
        float f[8], bb[8];
        H8<T>::unpack(v[u], f);
        load8f(bias ? bias + ch * 8 : nullptr, bb);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          f[k] += bb[k];
          if (relu) f[k] = fmaxf(f[k], 0.0f);
        }
        *reinterpret_cast<u32x4*>(x + static_cast<size_t>(id) * 8) = H8<T>::pack(f);
      }
    }
  }
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
    for (int o = e0; o < e1; ++o) {
      float f[8];
      H8<T>::unpack(*reinterpret_cast<const u32x4*>(x + (static_cast<long long>(idx[o]) * per + id) * 8), f);
      if (in_bias) {      // x is a bias-free convolution output: relu(x + b), rounded to the storage type as a separate pass would
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = Elem<T>::to_f32(Elem<T>::from_f32(fmaxf(f[q] + bb[q], 0.0f)));
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += f[q];
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

extern "C" int pvo_gru_glo(const void* wn, const void* net, const float* w_bias, float* glo, int E, int HW, int C, int dtype,
                           void* stream) {
  if (E < 0 || HW < 0 || C != 128) return PVO_EINVAL;
  if (E == 0 || HW == 0) return PVO_OK;
  if (!wn || !net || !glo || !aligned16(wn) || !aligned16(net) || E > 65535) return PVO_EINVAL;
  hipStream_t st = pvo_stream(stream);
  if (hipMemsetAsync(glo, 0, sizeof(float) * static_cast<size_t>(E) * C, st) != hipSuccess) return PVO_ELAUNCH;
  const int chunk = 512;
  dim3 grid((HW + chunk - 1) / chunk, E);
  GRU_DISPATCH(dtype,
    hipLaunchKernelGGL(gru_glo_kernel<pvo_half>, grid, dim3(256), 0, st, static_cast<const uint16_t*>(wn), static_cast<const uint16_t*>(net), w_bias, glo, HW, C, chunk),
    hipLaunchKernelGGL(gru_glo_kernel<pvo_bf16>, grid, dim3(256), 0, st, static_cast<const uint16_t*>(wn), static_cast<const uint16_t*>(net), w_bias, glo, HW, C, chunk));
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_gru_assemble(const void* net, const void* inp, const void* corr_feat, const void* flow_feat,
                                const float* corr_bias, const float* flow_bias,
                                void* X, long long rows, int with_inp, int dtype, void* stream) {
  if (rows < 0) return PVO_EINVAL;
  if (rows == 0) return PVO_OK;
  if (!net || (with_inp && !inp) || !corr_feat || !flow_feat || !X) return PVO_EINVAL;
  if (!inp) inp = net;   // unused when with_inp == 0
  if (rows * 56 >= (1LL << 31)) return PVO_EUNSUPPORTED;
  if (!aligned16(net) || !aligned16(inp) || !aligned16(corr_feat) || !aligned16(flow_feat) || !aligned16(X)) return PVO_EINVAL;
  hipStream_t st = pvo_stream(stream);
  const unsigned gsz = grid_for(rows * 56);
  GRU_DISPATCH(dtype,
    hipLaunchKernelGGL(gru_assemble_kernel<pvo_half>, dim3(gsz), dim3(256), 0, st, static_cast<const uint16_t*>(net), static_cast<const uint16_t*>(inp), static_cast<const uint16_t*>(corr_feat), static_cast<const uint16_t*>(flow_feat), corr_bias, flow_bias, static_cast<uint16_t*>(X), rows, with_inp),
    hipLaunchKernelGGL(gru_assemble_kernel<pvo_bf16>, dim3(gsz), dim3(256), 0, st, static_cast<const uint16_t*>(net), static_cast<const uint16_t*>(inp), static_cast<const uint16_t*>(corr_feat), static_cast<const uint16_t*>(flow_feat), corr_bias, flow_bias, static_cast<uint16_t*>(X), rows, with_inp));
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_gru_gate(const void* zr, const float* g, const void* net, void* Z, void* X,
                            const void* P_zr, int x_channels, int E, int HW, int dtype, void* stream) {
  if (x_channels != 448 && x_channels != 320) return PVO_EINVAL;
  if (P_zr && !aligned16(P_zr)) return PVO_EINVAL;
  if (E < 0 || HW < 0) return PVO_EINVAL;
  const long long rows = static_cast<long long>(E) * HW;
  if (rows == 0) return PVO_OK;
  if (rows * 16 >= (1LL << 31)) return PVO_EUNSUPPORTED;
  if (!zr || !g || !net || !Z || !X || !aligned16(zr) || !aligned16(net) || !aligned16(Z) || !aligned16(X)) return PVO_EINVAL;
  hipStream_t st = pvo_stream(stream);
  const unsigned gsz = grid_for(rows * 16);
  GRU_DISPATCH(dtype,
    hipLaunchKernelGGL(gru_gate_kernel<pvo_half>, dim3(gsz), dim3(256), 0, st, static_cast<const uint16_t*>(zr), g, static_cast<const uint16_t*>(net), static_cast<uint16_t*>(Z), static_cast<uint16_t*>(X), rows, HW, static_cast<const uint16_t*>(P_zr), x_channels),
    hipLaunchKernelGGL(gru_gate_kernel<pvo_bf16>, dim3(gsz), dim3(256), 0, st, static_cast<const uint16_t*>(zr), g, static_cast<const uint16_t*>(net), static_cast<uint16_t*>(Z), static_cast<uint16_t*>(X), rows, HW, static_cast<const uint16_t*>(P_zr), x_channels));
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_gru_out(const void* q, const float* g, const void* Z, const void* net, void* net_out,
                           const void* P_q, int E, int HW, int dtype, void* stream) {
  if (P_q && !aligned16(P_q)) return PVO_EINVAL;
  if (E < 0 || HW < 0) return PVO_EINVAL;
  const long long rows = static_cast<long long>(E) * HW;
  if (rows == 0) return PVO_OK;
  if (rows * 16 >= (1LL << 31)) return PVO_EUNSUPPORTED;
  if (!q || !g || !Z || !net || !net_out || !aligned16(q) || !aligned16(Z) || !aligned16(net) || !aligned16(net_out)) return PVO_EINVAL;
  hipStream_t st = pvo_stream(stream);
  const unsigned gsz = grid_for(rows * 16);
  GRU_DISPATCH(dtype,
    hipLaunchKernelGGL(gru_out_kernel<pvo_half>, dim3(gsz), dim3(256), 0, st, static_cast<const uint16_t*>(q), g, static_cast<const uint16_t*>(Z), static_cast<const uint16_t*>(net), static_cast<uint16_t*>(net_out), rows, HW, static_cast<const uint16_t*>(P_q)),
    hipLaunchKernelGGL(gru_out_kernel<pvo_bf16>, dim3(gsz), dim3(256), 0, st, static_cast<const uint16_t*>(q), g, static_cast<const uint16_t*>(Z), static_cast<const uint16_t*>(net), static_cast<uint16_t*>(net_out), rows, HW, static_cast<const uint16_t*>(P_q)));
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_bias_act(void* x, const float* bias, long long rows, int C, int relu, int dtype, void* stream) {
  if (rows < 0 || C <= 0 || (C & 7)) return PVO_EINVAL;
  if (rows == 0) return PVO_OK;
  if (!x || !aligned16(x)) return PVO_EINVAL;
  if (rows * (C >> 3) >= (1LL << 31)) return PVO_EUNSUPPORTED;
  hipStream_t st = pvo_stream(stream);
  const unsigned gsz = grid_for((rows * (C >> 3) + 3) / 4);
  GRU_DISPATCH(dtype,
    hipLaunchKernelGGL(bias_act_kernel<pvo_half>, dim3(gsz), dim3(256), 0, st, static_cast<uint16_t*>(x), bias, rows, C, relu),
    hipLaunchKernelGGL(bias_act_kernel<pvo_bf16>, dim3(gsz), dim3(256), 0, st, static_cast<uint16_t*>(x), bias, rows, C, relu));
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

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

}  // namespace

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
