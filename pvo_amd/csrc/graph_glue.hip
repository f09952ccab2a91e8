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
#include "graph_post.h"

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
__global__ __launch_bounds__(256) void graph_post_kernel(GraphPostArgs g, const uint16_t* __restrict__ y8, int E, int HW, int W,
                                                         const int* __restrict__ segm, const int* __restrict__ vote_tot,
                                                         const int* __restrict__ vote_dyn, int S, float vote_thresh) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= E * HW) return;
  const int e = idx / HW, pix = idx - e * HW;
  const uint4 q = *reinterpret_cast<const uint4*>(y8 + static_cast<size_t>(idx) * 8);
  graph_post_pixel<T>(idx, e, pix, q, g, HW, W, segm, vote_tot, vote_dyn, S, vote_thresh);      // (graph_post.h)
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
  const GraphPostArgs g = {reinterpret_cast<const float2*>(coords1), f2(raw_mask), f2(target), f2(delta_dy), f2(weight), target_ba, weight_ba, f2(full_flow), dy_thresh};
  if (dtype == PVO_F16)
    hipLaunchKernelGGL(graph_post_kernel<pvo_half>, grid, dim3(256), 0, st, g, static_cast<const uint16_t*>(heads), E, H * W, W, segm, vote_tot, vote_dyn, max_segments, vote_thresh);
  else if (dtype == PVO_BF16)
    hipLaunchKernelGGL(graph_post_kernel<pvo_bf16>, grid, dim3(256), 0, st, g, static_cast<const uint16_t*>(heads), E, H * W, W, segm, vote_tot, vote_dyn, max_segments, vote_thresh);
  else return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}
