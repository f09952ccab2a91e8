// graph_glue.hip — the arithmetic FactorGraph.update does around the update operator, as two kernels.
//
// reference VO_Module/droid_slam/factor_graph.py:231-306 (segm_filter == False):
//   motion   :233-237  motn = clamp(cat[target-coords0, target-coords0+delta_dy, target-coords1, raw_mask], +-64)
//                      permuted to channels-first; here written channels-last in the operator's 16-bit dtype
//   post     :249-306  target = coords1 + delta[..., 0:2]; raw_mask += delta_m; bin = sigmoid(raw_mask) >= 0.5;
//                      delta_dy = delta[..., 2:4] * (1 - bin); weight = sigmoid(weight + 10 (1 - bin));
//                      full_flow = coords1 + delta_dy - coords0; and the [E,2,H,W] target / weight the BA reads
// In PyTorch this is ~25 element-wise / cat / permute launches over 1-2 MB tensors per graph update (launch-bound);
// the head outputs are read straight from the [E,H,W,8] tensor heads_out writes (delta | delta_dy | weight | delta_mask).
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void graph_motion_kernel(const float2* __restrict__ target, const float2* __restrict__ coords1,
                                                           const float2* __restrict__ delta_dy, const float2* __restrict__ raw_mask,
                                                           uint16_t* __restrict__ motn, int E, int HW, int W) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= E * HW) return;
  const int pix = idx % HW;
  const float x0 = static_cast<float>(pix % W), y0 = static_cast<float>(pix / W);
  const float2 t = target[idx], c1 = coords1[idx], d = delta_dy[idx], m = raw_mask[idx];
  float f[8] = {t.x - x0, t.y - y0, t.x - x0 + d.x, t.y - y0 + d.y, t.x - c1.x, t.y - c1.y, m.x, m.y};
  uint32_t o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float a = fminf(fmaxf(f[2 * k], -64.0f), 64.0f), b = fminf(fmaxf(f[2 * k + 1], -64.0f), 64.0f);
    uint32_t lo, hi;
    if constexpr (sizeof(typename Elem<T>::store_t) == 2 && __is_same(T, pvo_half)) {
      union { _Float16 h; uint16_t u; } ca, cb; ca.h = static_cast<_Float16>(a); cb.h = static_cast<_Float16>(b);
      lo = ca.u; hi = cb.u;
    } else {
      lo = pvo_f32_to_bf16(a); hi = pvo_f32_to_bf16(b);
    }
    o[k] = lo | (hi << 16);
  }
  *reinterpret_cast<uint4*>(motn + static_cast<size_t>(idx) * 8) = make_uint4(o[0], o[1], o[2], o[3]);
}

template <typename T>
__device__ __forceinline__ float h2f_(uint32_t bits) {
  if constexpr (__is_same(T, pvo_half)) { union { uint16_t u; _Float16 h; } c; c.u = static_cast<uint16_t>(bits); return static_cast<float>(c.h); }
  else return pvo_bf16_to_f32(static_cast<uint16_t>(bits));
}

template <typename T>
__global__ __launch_bounds__(256) void graph_post_kernel(const float2* __restrict__ coords1, const uint16_t* __restrict__ y8,
                                                         float2* __restrict__ raw_mask, float2* __restrict__ target,
                                                         float2* __restrict__ delta_dy, float2* __restrict__ weight,
                                                         float* __restrict__ target_ba, float* __restrict__ weight_ba,
                                                         float2* __restrict__ full_flow, int E, int HW, int W, float dy_thresh,
                                                         const int* __restrict__ segm, const int* __restrict__ vote_tot,
                                                         const int* __restrict__ vote_dyn, int S, float vote_thresh) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= E * HW) return;
  const int e = idx / HW, pix = idx - e * HW;
  const float x0 = static_cast<float>(pix % W), y0 = static_cast<float>(pix / W);
  const uint4 q = *reinterpret_cast<const uint4*>(y8 + static_cast<size_t>(idx) * 8);
  const float d0 = h2f_<T>(q.x & 0xffffu), d1 = h2f_<T>(q.x >> 16);      // delta
  const float g0 = h2f_<T>(q.y & 0xffffu), g1 = h2f_<T>(q.y >> 16);      // delta_dy (raw)
  const float w0 = h2f_<T>(q.z & 0xffffu), w1 = h2f_<T>(q.z >> 16);      // weight logits
  const float m0 = h2f_<T>(q.w & 0xffffu), m1 = h2f_<T>(q.w >> 16);      // delta_mask
  const float2 c1 = coords1[idx];
  float2 rm = raw_mask[idx];
  rm.x += m0; rm.y += m1;
  raw_mask[idx] = rm;
  float b0 = (1.0f / (1.0f + expf(-rm.x)) >= dy_thresh) ? 1.0f : 0.0f;    // 1: static, 0: dynamic
  float b1 = (1.0f / (1.0f + expf(-rm.y)) >= dy_thresh) ? 1.0f : 0.0f;
  if (segm) {      // panoptic vote (factor_graph.py:256-276): a segment (id != 0) whose dynamic fraction on this edge exceeds the threshold is forced dynamic
    int sg = segm[idx];
    sg = sg < 0 ? 0 : (sg >= S ? S - 1 : sg);
    if (sg != 0) {
      const float tot = static_cast<float>(vote_tot[static_cast<size_t>(e) * S + sg]), dyn = static_cast<float>(vote_dyn[static_cast<size_t>(e) * S + sg]);
      if (dyn / fmaxf(tot, 1.0f) > vote_thresh) { b0 = 0.0f; b1 = 0.0f; }
    }
  }
  const float2 tg = {c1.x + d0, c1.y + d1};
  const float2 dd = {g0 * (1.0f - b0), g1 * (1.0f - b1)};
  const float2 wt = {1.0f / (1.0f + expf(-(w0 + (1.0f - b0) * 10.0f))), 1.0f / (1.0f + expf(-(w1 + (1.0f - b1) * 10.0f)))};
  target[idx] = tg; delta_dy[idx] = dd; weight[idx] = wt;
  full_flow[idx] = {c1.x + dd.x - x0, c1.y + dd.y - y0};
  const size_t ob = static_cast<size_t>(e) * 2 * HW + pix;
  target_ba[ob] = tg.x; target_ba[ob + HW] = tg.y;
  weight_ba[ob] = wt.x; weight_ba[ob + HW] = wt.y;
}

}  // namespace


extern "C" int pvo_graph_motion(const float* target, const float* coords1, const float* delta_dy, const float* raw_mask,
                                void* motn, int E, int H, int W, int dtype, void* stream) {
  if (E < 0 || H < 0 || W < 0) return PVO_EINVAL;
  const long long n = static_cast<long long>(E) * H * W;
  if (n == 0) return PVO_OK;
  if (!target || !coords1 || !delta_dy || !raw_mask || !motn || n >= (1LL << 31)) return PVO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(motn) & 15) || ((reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(coords1) |
       reinterpret_cast<uintptr_t>(delta_dy) | reinterpret_cast<uintptr_t>(raw_mask)) & 7)) return PVO_EINVAL;
  hipStream_t st = pvo_stream(stream);
  const dim3 grid(static_cast<unsigned>((n + 255) / 256));
  auto f2 = [](const float* p) { return reinterpret_cast<const float2*>(p); };
  if (dtype == PVO_F16)
    hipLaunchKernelGGL(graph_motion_kernel<pvo_half>, grid, dim3(256), 0, st, f2(target), f2(coords1), f2(delta_dy), f2(raw_mask), static_cast<uint16_t*>(motn), E, H * W, W);
  else if (dtype == PVO_BF16)
    hipLaunchKernelGGL(graph_motion_kernel<pvo_bf16>, grid, dim3(256), 0, st, f2(target), f2(coords1), f2(delta_dy), f2(raw_mask), static_cast<uint16_t*>(motn), E, H * W, W);
  else return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_graph_post(const float* coords1, const void* heads, float* raw_mask, float* target, float* delta_dy,
                              float* weight, float* target_ba, float* weight_ba, float* full_flow,
                              int E, int H, int W, float dy_thresh, const int* segm, const int* vote_tot, const int* vote_dyn,
                              int max_segments, float vote_thresh, int dtype, void* stream) {
  if (E < 0 || H < 0 || W < 0) return PVO_EINVAL;
  const long long n = static_cast<long long>(E) * H * W;
  if (n == 0) return PVO_OK;
  if (!coords1 || !heads || !raw_mask || !target || !delta_dy || !weight || !target_ba || !weight_ba || !full_flow || n >= (1LL << 31)) return PVO_EINVAL;
  if (reinterpret_cast<uintptr_t>(heads) & 15) return PVO_EINVAL;
  if (segm && (!vote_tot || !vote_dyn || max_segments <= 0)) return PVO_EINVAL;
  hipStream_t st = pvo_stream(stream);
  const dim3 grid(static_cast<unsigned>((n + 255) / 256));
  auto f2 = [](float* p) { return reinterpret_cast<float2*>(p); };
  if (dtype == PVO_F16)
    hipLaunchKernelGGL(graph_post_kernel<pvo_half>, grid, dim3(256), 0, st, reinterpret_cast<const float2*>(coords1), static_cast<const uint16_t*>(heads), f2(raw_mask), f2(target), f2(delta_dy), f2(weight), target_ba, weight_ba, f2(full_flow), E, H * W, W, dy_thresh, segm, vote_tot, vote_dyn, max_segments, vote_thresh);
  else if (dtype == PVO_BF16)
    hipLaunchKernelGGL(graph_post_kernel<pvo_bf16>, grid, dim3(256), 0, st, reinterpret_cast<const float2*>(coords1), static_cast<const uint16_t*>(heads), f2(raw_mask), f2(target), f2(delta_dy), f2(weight), target_ba, weight_ba, f2(full_flow), E, H * W, W, dy_thresh, segm, vote_tot, vote_dyn, max_segments, vote_thresh);
  else return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}
