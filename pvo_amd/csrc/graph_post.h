// graph_post.h — one pixel of FactorGraph.update's arithmetic behind the update operator (factor_graph.py:249-306), shared by
// graph_post_kernel (graph_glue.hip) and the output-head gather that does it in its epilogue (gru_fused.hip): one text, the same
// operations in the same order wherever it runs.
#pragma once
#include "common.h"

struct GraphPostArgs {
  const float2* coords1;
  float2* raw_mask; float2* target; float2* delta_dy; float2* weight;
  float* target_ba; float* weight_ba;      // [E,2,HW] the BA's layout
  float2* full_flow;
  float dy_thresh;
};

template <typename T>
__device__ __forceinline__ float gp_h2f(uint32_t bits) {
  if constexpr (__is_same(T, pvo_half)) { union { uint16_t u; _Float16 h; } c; c.u = static_cast<uint16_t>(bits); return static_cast<float>(c.h); }
  else return pvo_bf16_to_f32(static_cast<uint16_t>(bits));
}

// q = the pixel's eight 16-bit head outputs (delta | delta_dy | weight logits | delta_mask), idx = e * HW + pix
template <typename T>
__device__ __forceinline__ void graph_post_pixel(int idx, int e, int pix, uint4 q, const GraphPostArgs& g, int HW, int W,
                                                 const int* __restrict__ segm, const int* __restrict__ vote_tot,
                                                 const int* __restrict__ vote_dyn, int S, float vote_thresh) {
  const float x0 = static_cast<float>(pix % W), y0 = static_cast<float>(pix / W);
  const float d0 = gp_h2f<T>(q.x & 0xffffu), d1 = gp_h2f<T>(q.x >> 16);      // delta
  const float g0 = gp_h2f<T>(q.y & 0xffffu), g1 = gp_h2f<T>(q.y >> 16);      // delta_dy (raw)
  const float w0 = gp_h2f<T>(q.z & 0xffffu), w1 = gp_h2f<T>(q.z >> 16);      // weight logits
  const float m0 = gp_h2f<T>(q.w & 0xffffu), m1 = gp_h2f<T>(q.w >> 16);      // delta_mask
  const float2 c1 = g.coords1[idx];
  float2 rm = g.raw_mask[idx];
  rm.x += m0; rm.y += m1;
  g.raw_mask[idx] = rm;
  float b0 = (1.0f / (1.0f + expf(-rm.x)) >= g.dy_thresh) ? 1.0f : 0.0f;    // 1: static, 0: dynamic
  float b1 = (1.0f / (1.0f + expf(-rm.y)) >= g.dy_thresh) ? 1.0f : 0.0f;
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
  g.target[idx] = tg; g.delta_dy[idx] = dd; g.weight[idx] = wt;
  g.full_flow[idx] = {c1.x + dd.x - x0, c1.y + dd.y - y0};
  const size_t ob = static_cast<size_t>(e) * 2 * HW + pix;
  g.target_ba[ob] = tg.x; g.target_ba[ob + HW] = tg.y;
  g.weight_ba[ob] = wt.x; g.weight_ba[ob + HW] = wt.y;
}

// gru_fused.hip: pvo_heads_gather with graph_post_pixel as its epilogue (post != nullptr and the LDS-tiled form applies: *fused = 1;
// otherwise the plain gather, *fused = 0 and the caller runs pvo_graph_post).  Not part of the C ABI.
int pvo_internal_heads_gather_post(const float* z, const float* bias2, void* y, const GraphPostArgs* post, int* fused,
                                   int E, int H, int W, int dtype, void* stream) __attribute__((visibility("hidden")));
