// update_exec.hip — native executor of one factor-graph update.
//
// The reference issues one graph update (VO_Module/droid_slam/factor_graph.py:227-307) as ~150 PyTorch / cuDNN launches
// from Python: reproject (lietorch chain), 4 lookups + cat, ~25 convolutions with their element-wise glue
// (droid_net.py:256-314, modules/gru.py:19-32), the mask / weight arithmetic, and droid_backends.ba with its host round
// trips.  Here the whole update is ONE C call that enqueues ~33 hand-written kernels on the caller's stream (the
// aggregation branch on a second stream, forked and joined with events): no Python, no MIOpen / hipBLASLt, no allocation,
// no host synchronisation.  Because every launch comes from this one function with caller-owned buffers, the call can
// also be captured into a HIP graph by the caller.
//
//   pvo_update_operator   DynamicUpdateModule.forward (droid_net.py:256-314) on the 16-bit inference path
//   pvo_graph_update      FactorGraph.update (factor_graph.py:227-307): reproject -> motion features -> operator ->
//                         (panoptic vote) -> mask / weight glue -> eta / damping -> dense BA x itrs -> depth clamp
#include "common.h"
#include "graph_post.h"
#include <stdlib.h>

// operator_small.hip: pvo_segment_hist for two tables carved from one workspace (one fill for both; not part of the C ABI)
int pvo_segment_hist_ws(const int* segm, const float* raw_mask, const void* heads, int* tot, int* dyn,
                        int E, int HW, int S, float dy_thresh, int dtype, void* stream);

namespace {

struct SideCtx { hipStream_t side; hipEvent_t fork, join, mid; bool ok; };

// one side stream + two events per device, created on first use (streams and events are host objects: the library
// still allocates no device memory)
// (events with hipEventReleaseToDevice instead of the default system-scope release: the ~6.5 us a record or a satisfied
// wait costs on the launch stream did not change)
// The library's fork / join / mid events carry NO system-scope fence (hipEventDisableSystemFence): they order work between two
// streams of one device, where the producing kernel's own end-of-kernel release and the consuming kernel's acquire already
// make the data visible; the default flags add a system-scope cache writeback + invalidate to every record.  Measured
// (tools/sched_bisect.py, profiles/r03_sched_bisect.txt): bit-identical results over 1500 two-update runs either way, the update
// 1 % shorter without the fences - and cache-maintenance operations in flight beside a resident kernel of ANOTHER queue are
// exactly what made the BA irreproducible in the overlapped arrangements.
constexpr unsigned g_event_flags = hipEventDisableTiming | hipEventDisableSystemFence;
SideCtx g_side_ctx[64] = {};
SideCtx* side_ctx() {
  SideCtx* ctx = g_side_ctx;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  SideCtx& c = ctx[dev];
  if (!c.ok) {
    // HIGH priority: in both places it is used the side stream carries the chain the launch stream ends up waiting for
    // (GraphAgg beside the heads: conv1 84 us instead of 105 when its workgroups are dispatched first; update 757 -> 728 us)
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return nullptr;
    if (hipStreamCreateWithPriority(&c.side, hipStreamNonBlocking, greatest) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&c.fork, g_event_flags) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&c.join, g_event_flags) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&c.mid, g_event_flags) != hipSuccess) return nullptr;
    c.ok = true;
  }
  return &c;
}

size_t al(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

// The gate context of the NEXT update, computed ahead inside this update's pose solves (pvo_graph_update_args.context_ahead):
// what it was computed for.  One record per device; any call that writes an operator workspace's context buffers drops it.
struct ContextAhead { const void *ws, *weights, *glo_w, *gate_wt, *net; int E, H, W, dtype; bool valid; };
ContextAhead g_ctx_ahead[64] = {};
ContextAhead* ctx_ahead_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  return &g_ctx_ahead[dev];
}

// Measurement hook (bench.py): HIP events around one stage of the update, recorded on the launch stream itself, so that a
// kernel's duration INSIDE the timed steps can be read without a profiler.  Disarmed (stage -1) it costs one compare.
// `every` > 1 samples one occurrence in `every` (the events cost the launch stream ~7 us per occurrence, 2 % of the bench's step).
struct Probe { int stage; int cap; int limit; int n; int every; int seen; hipEvent_t* ev; };
Probe g_probe = {-1, 0, 0, 0, 1, 0, nullptr};

inline void probe_mark(int stage, int which, void* stream) {
  if (g_probe.stage != stage || g_probe.n >= g_probe.limit) return;
  if (g_probe.seen % g_probe.every == 0) {
    (void)hipEventRecord(g_probe.ev[2 * g_probe.n + which], pvo_stream(stream));
    if (which == 1) ++g_probe.n;
  }
  if (which == 1) ++g_probe.seen;
}

struct OpWs {
  char *c1, *f1, *CF, *Z, *RN, *h1, *a1, *am, *a2, *P_zr, *P_q;
  float *part, *g;
  size_t bytes;
};

OpWs carve_op(void* base, int E, int K, int H, int W, int chunks) {
  OpWs w{};
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) { char* r = p ? p + off : nullptr; off += al(bytes); return r; };
  const size_t px = static_cast<size_t>(E) * H * W, kpx = static_cast<size_t>(K) * H * W;
  w.c1 = take(px * 128 * 2); w.f1 = take(px * 128 * 2); w.CF = take(px * 192 * 2);
  w.Z = take(px * 128 * 2); w.RN = take(px * 128 * 2); w.h1 = take(px * 72 * 4);      // z [px][4 heads][18] f32 (pvo_conv3x3_heads)
  w.a1 = take(px * 128 * 2); w.am = take(kpx * 128 * 2); w.a2 = take(kpx * 128 * 2);
  w.P_zr = take(px * 256 * 2); w.P_q = take(px * 128 * 2);
  w.part = reinterpret_cast<float*>(take(static_cast<size_t>(E) * chunks * 128 * 4));
  w.g = reinterpret_cast<float*>(take(static_cast<size_t>(E) * 384 * 4));
  w.bytes = off;
  return w;
}

struct UpWs {
  float *coords, *valid, *eta;
  char *motion, *heads, *upmask;
  int *vote_tot, *vote_dyn;
  void* op;
  size_t bytes;
};

UpWs carve_up(void* base, int E, int K, int R, int H, int W, int S, size_t op_bytes) {
  UpWs w{};
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) { char* r = p ? p + off : nullptr; off += al(bytes); return r; };
  const size_t px = static_cast<size_t>(E) * H * W;
  w.coords = reinterpret_cast<float*>(take(px * 2 * 4));
  w.valid = reinterpret_cast<float*>(take(px * 4));
  w.eta = reinterpret_cast<float*>(take(static_cast<size_t>(R > 0 ? R : 1) * H * W * 4));
  w.motion = take(px * 8 * 2);
  w.heads = take(px * 8 * 2);
  w.upmask = take(static_cast<size_t>(K) * H * W * 576 * 2);
  w.vote_tot = reinterpret_cast<int*>(take(static_cast<size_t>(E) * (S > 0 ? S : 0) * 4 + 4));
  w.vote_dyn = reinterpret_cast<int*>(take(static_cast<size_t>(E) * (S > 0 ? S : 0) * 4 + 4));
  w.op = take(op_bytes);
  w.bytes = off;
  return w;
}

void* ws_base(void* workspace) {
  return reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
}

// (Rounds 2-4 kept alternative stream arrangements of the BA, a buffer tap and three trunk orders in this file behind
// PVO_SCHED_DEBUG / environment switches - the experiments of DESIGN.md section 5 and profiles/r04_*.  They were measured, none
// shipped, and round 5 removed them from the product source: `git show f406a59:pvo_amd/csrc/update_exec.hip` is the tree
// tools/sched_bisect.py and those profiles refer to.)

#define RUN(call)                    \
  do {                               \
    const int rc_ = (call);          \
    if (rc_ != PVO_OK) return rc_;   \
  } while (0)

// everything of the operator up to the new hidden state; the two branches that read it follow in run_heads / run_agg.
// With PVO_OP_ENC_SIDE_STREAM the work that does not depend on the correlation features - the global-context reduction of
// `net`, the gate context, the flow encoder - runs on the side stream beside the HBM-bound lookup and corr_encoder[2].
struct MotionJob {           // pvo_graph_update's motion features: only the flow encoder reads them, so they go with it
  const float *target, *coords, *delta_dy, *raw_mask; void* motion;
};

int run_trunk(const pvo_update_weights* w, const pvo_operator_args* a, OpWs& b, void* stream, const void** P_zr, const void** P_q,
              const MotionJob* mj, bool context_ready = false) {
  const int E = a->E, H = a->H, W = a->W, dt = w->dtype;
  const long long rows = static_cast<long long>(E) * H * W;
  hipStream_t st = pvo_stream(stream);
  SideCtx* sc = (w->flags & PVO_OP_ENC_SIDE_STREAM) && !(w->flags & PVO_OP_SINGLE_STREAM) ? side_ctx() : nullptr;
  // two chains meet in front of the gate convolution: the launch stream runs lookup + corr_encoder[0] -> corr_encoder[2]
  // (the latency-bound lookup first) and then the global context of `net` -> gate context; the side stream the motion
  // features -> flow_encoder.  Balanced on rocprofv3 --kernel-trace (tools/update_timeline.sh): with the context pair on
  // the side stream that chain ended ~55 us after corr_encoder[2]; a third stream for it changed nothing.
  void* s2 = sc ? static_cast<void*>(sc->side) : stream;
  void* s3 = stream;
  if (sc) {
    if (hipEventRecord(sc->fork, st) != hipSuccess) return PVO_ELAUNCH;
    if (hipStreamWaitEvent(sc->side, sc->fork, 0) != hipSuccess) return PVO_ELAUNCH;
  }
  if (a->levels[0]) {
    probe_mark(PVO_STAGE_EMPTY, 0, stream);      // (calibration: an event pair around nothing, at the lookup's place in the stream)
    probe_mark(PVO_STAGE_EMPTY, 1, stream);
    probe_mark(PVO_STAGE_LOOKUP, 0, stream);
    RUN(pvo_corr_lookup_encode_tiled(a->levels, a->coords, w->enc0_w, w->enc0_b, b.c1, E, H, W, dt, a->slots, a->num_slots, stream));
    probe_mark(PVO_STAGE_LOOKUP, 1, stream);
  } else {
    if (!a->corr) return PVO_EINVAL;
    RUN(pvo_corr_encode(a->corr, w->enc0_w, w->enc0_b, b.c1, rows, dt, stream));
  }
  if (mj) RUN(pvo_graph_motion(mj->target, mj->coords, mj->delta_dy, mj->raw_mask, mj->motion, E, H, W, dt, s2));
  // The flow encoder as two kernels.  pvo_flow_encoder (one kernel, the 128-channel intermediate in LDS) is bit-identical and
  // shorter alone, but this stretch of the update is throughput-bound: beside it corr_encoder[2] on the launch stream takes 74
  // instead of 41 us (its redundant 7x7 halo work occupies the matrix cores) and the gate convolution starts 8 us LATER
  // (226.9 -> 225.2 keyframe updates/s, round 3).  Also measured and not shipped (round 4, profiles/r04_update_timeline_*): part
  // of flow_encoder[2]'s edges on the launch stream (no gain at 20 / 33 / 45 %), the 7x7 first and alone on the launch stream,
  // the fork behind the lookup (both slower: whatever runs beside the lookup pays for the memory request rate it saturates).
  RUN(pvo_conv7x7_c8(a->motion, w->fenc0_w, w->fenc0_b, b.f1, E, H, W, dt, s2));
  RUN(pvo_conv3x3_c128(b.f1, w->fenc2_w, w->fenc2_b, b.CF, E, H, W, 64, 1, 192, 128, dt, s2));
  if (sc && hipEventRecord(sc->join, sc->side) != hipSuccess) return PVO_ELAUNCH;
  // the encoders' second layers write relu(features + bias) side by side: CF = [corr features (128) | flow features (64)]
  if (w->flags & PVO_OP_CONV128_WIDE)
    RUN(pvo_conv3x3(b.c1, w->cenc2_w, w->cenc2_b, b.CF, E, H, W, 128, 128, 1, 192, 0, dt, stream));
  else
    RUN(pvo_conv3x3_c128(b.c1, w->cenc2_w, w->cenc2_b, b.CF, E, H, W, 128, 1, 192, 0, dt, stream));
  if (!context_ready) {           // (else b.g already holds it: computed inside the previous update's pose solves)
    RUN(pvo_gru_glo_fused(a->net, w->glo_w, w->glo_b, b.part, E, H * W, dt, s3));
    RUN(pvo_gate_context(b.part, w->gate_wt, w->gate_b, b.g, E, pvo_gru_glo_chunks(H * W), s3));
  }
  *P_zr = a->P_zr; *P_q = a->P_q;
  if (!a->P_zr || !a->P_q) {       // static-input term not cached by the caller: conv(W[:, inp], inp) for this call
    if (!a->inp) return PVO_EINVAL;
    RUN(pvo_conv3x3(a->inp, w->zr_inp_w, nullptr, b.P_zr, E, H, W, 128, 256, 0, 0, 0, dt, stream));
    RUN(pvo_conv3x3(a->inp, w->q_inp_w, nullptr, b.P_q, E, H, W, 128, 128, 0, 0, 0, dt, stream));
    *P_zr = b.P_zr; *P_q = b.P_q;
  }
  if (sc && hipStreamWaitEvent(st, sc->join, 0) != hipSuccess) return PVO_ELAUNCH;
  probe_mark(PVO_STAGE_GATES, 0, stream);
  const int* ps = (a->static_by_slot && a->P_zr && a->P_q) ? a->slots : nullptr;
  if (a->static_by_slot && !a->slots) return PVO_EINVAL;
  RUN(pvo_gru_conv_gates(a->net, b.CF, 192, w->zr_w, b.g, *P_zr, ps, b.Z, b.RN, E, H, W, dt, stream));
  probe_mark(PVO_STAGE_GATES, 1, stream);
  probe_mark(PVO_STAGE_CANDIDATE, 0, stream);
  RUN(pvo_gru_conv_candidate(b.RN, b.CF, 192, w->q_w, b.g, *P_q, ps, b.Z, a->net, a->net_out, E, H, W, dt, stream));
  probe_mark(PVO_STAGE_CANDIDATE, 1, stream);
  return PVO_OK;
}

// post != nullptr: the graph update's mask / target / weight arithmetic (pvo_graph_post without the panoptic vote) runs as the
// gather's epilogue where the gather's tiled form applies; *post_fused says whether it did
int run_heads(const pvo_update_weights* w, const pvo_operator_args* a, OpWs& b, void* stream,
              const GraphPostArgs* post = nullptr, int* post_fused = nullptr) {
  const int E = a->E, H = a->H, W = a->W, dt = w->dtype;
  // (b.h1 holds z [E,H,W,4,18] f32: the hidden tensor itself stays in the first launch's LDS)
  RUN(pvo_conv3x3_heads(a->net_out, w->heads1_w, w->heads1_b, w->heads2_w, reinterpret_cast<float*>(b.h1), E, H, W, dt, stream));
  RUN(pvo_internal_heads_gather_post(reinterpret_cast<const float*>(b.h1), w->heads2_b, a->heads, post, post_fused, E, H, W, dt, stream));
  return PVO_OK;
}

int run_upmask(const pvo_update_weights* w, const pvo_operator_args* a, OpWs& b, void* stream) {
  if (a->K <= 0 || !a->upmask) return PVO_OK;
  return pvo_conv1x1_c128(b.a2, w->up_w, w->up_b, a->upmask, static_cast<long long>(a->K) * a->H * a->W, 576, 0, w->dtype, stream);
}

int run_agg(const pvo_update_weights* w, const pvo_operator_args* a, OpWs& b, void* stream, SideCtx* mark = nullptr, bool with_upmask = true) {
  const int E = a->E, H = a->H, W = a->W, K = a->K, dt = w->dtype;
  if (K <= 0) return PVO_OK;
  if (w->flags & PVO_OP_CONV128_WIDE)
    RUN(pvo_conv3x3(a->net_out, w->agg1_w, nullptr, b.a1, E, H, W, 128, 128, 0, 0, 0, dt, stream));
  else
    RUN(pvo_conv3x3_c128(a->net_out, w->agg1_w, nullptr, b.a1, E, H, W, 128, 0, 0, 0, dt, stream));
  // conv1's bias + ReLU are applied by the mean kernel as it reads
  RUN(pvo_segment_mean(b.a1, a->seg_ptr, a->seg_idx, w->agg1_b, b.am, K, H * W, 128, dt, stream));
  RUN(pvo_conv3x3_c128(b.am, w->agg2_w, w->agg2_b, b.a2, K, H, W, 128, 1, 0, 0, dt, stream));
  if (a->eta)
    RUN(pvo_eta_head(b.a2, w->eta_w, w->eta_b, a->eta_frame, a->eta_pos, a->damping, a->eta, a->eta_frame ? a->R : K, H, W, a->EP, a->eta_scale, dt, stream));
  if (mark && hipEventRecord(mark->mid, pvo_stream(stream)) != hipSuccess) return PVO_ELAUNCH;
  // INVARIANT pvo_graph_update relies on (it waits for `mid` and has no second join): with with_upmask == false NOTHING is
  // enqueued on this stream behind `mid`.  Whatever is added below this line must be waited for by the callers that pass a
  // `mark` (run_operator's `pending` users), or go in front of the record.
  if (with_upmask) RUN(run_upmask(w, a, b, stream));
  return PVO_OK;
}

int check_op(const pvo_update_weights* w, const pvo_operator_args* a) {
  if (!w || !a) return PVO_EINVAL;
  if (w->dtype != PVO_F16 && w->dtype != PVO_BF16) return PVO_EUNSUPPORTED;
  if (a->E < 0 || a->H <= 0 || a->W <= 0 || a->K < 0) return PVO_EINVAL;
  if (a->E > 0 && (!a->net || !a->net_out || !a->heads)) return PVO_EINVAL;
  if (a->K > 0 && (!a->seg_ptr || !a->seg_idx)) return PVO_EINVAL;
  return PVO_OK;
}

// trunk on `stream`, then the aggregation branch on the side stream beside the heads on `stream`.  join_now = false
// leaves the join to the caller (pvo_graph_update joins only in front of the BA, so the K-frame kernels of the
// aggregation branch, which occupy a fraction of the chip, also overlap the mask / weight glue).
int run_operator(const pvo_update_weights* w, const pvo_operator_args* a, OpWs& b, void* stream, SideCtx** pending,
                 const MotionJob* mj = nullptr, bool upmask_on_side = true, bool context_ready = false,
                 const GraphPostArgs* post = nullptr, int* post_fused = nullptr) {
  const void *P_zr, *P_q;
  RUN(run_trunk(w, a, b, stream, &P_zr, &P_q, mj, context_ready));
  hipStream_t st = pvo_stream(stream);
  SideCtx* sc = (a->K > 0 && !(w->flags & PVO_OP_SINGLE_STREAM)) ? side_ctx() : nullptr;
  if (sc) {
    if (hipEventRecord(sc->fork, st) != hipSuccess) return PVO_ELAUNCH;
    if (hipStreamWaitEvent(sc->side, sc->fork, 0) != hipSuccess) return PVO_ELAUNCH;
    RUN(run_agg(w, a, b, sc->side, sc, upmask_on_side));
    if (hipEventRecord(sc->join, sc->side) != hipSuccess) return PVO_ELAUNCH;
    RUN(run_heads(w, a, b, stream, post, post_fused));
    *pending = sc;
  } else {
    RUN(run_heads(w, a, b, stream, post, post_fused));
    RUN(run_agg(w, a, b, stream, nullptr, upmask_on_side));
    *pending = nullptr;
  }
  return PVO_OK;
}

int join(SideCtx* sc, void* stream) {
  if (sc && hipStreamWaitEvent(pvo_stream(stream), sc->join, 0) != hipSuccess) return PVO_ELAUNCH;
  return PVO_OK;
}

__global__ __launch_bounds__(256) void clamp_min_kernel(float* __restrict__ x, long long n, float lo) {
  const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) { const float v = x[i]; x[i] = (v < lo) ? lo : v; }      // NaN stays NaN, as in torch.clamp
}

}  // namespace

extern "C" size_t pvo_operator_workspace_bytes(int E, int K, int H, int W) {
  if (E < 0 || K < 0 || H <= 0 || W <= 0) return 0;
  return carve_op(nullptr, E, K, H, W, pvo_gru_glo_chunks(H * W)).bytes + 256;
}

extern "C" int pvo_update_operator(const pvo_update_weights* w, const pvo_operator_args* a,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  RUN(check_op(w, a));
  if (a->E == 0) return PVO_OK;
  if (!workspace || workspace_bytes < pvo_operator_workspace_bytes(a->E, a->K, a->H, a->W)) return PVO_EWORKSPACE;
  if (!a->motion) return PVO_EINVAL;
  OpWs b = carve_op(ws_base(workspace), a->E, a->K, a->H, a->W, pvo_gru_glo_chunks(a->H * a->W));
  if (ContextAhead* ca = ctx_ahead_slot()) ca->valid = false;
  SideCtx* pending = nullptr;
  RUN(run_operator(w, a, b, stream, &pending));
  return join(pending, stream);
}

extern "C" size_t pvo_graph_update_workspace_bytes(int E, int K, int R, int H, int W, int max_segments) {
  if (E < 0 || K < 0 || R < 0 || H <= 0 || W <= 0 || max_segments < 0) return 0;
  return carve_up(nullptr, E, K, R, H, W, max_segments, pvo_operator_workspace_bytes(E, K, H, W)).bytes + 256;
}

extern "C" int pvo_graph_update(const pvo_update_weights* w, const pvo_graph_update_args* u,
                                void* workspace, size_t workspace_bytes, void* stream) {
  if (!w || !u) return PVO_EINVAL;
  pvo_operator_args a = u->op;
  a.net_out = a.net_out ? a.net_out : const_cast<void*>(a.net);
  a.heads = reinterpret_cast<void*>(1);        // supplied from the workspace below
  RUN(check_op(w, &a));
  const int E = a.E, H = a.H, W = a.W, K = a.K, R = a.R, dt = w->dtype, HW = H * W;
  if (E == 0) return PVO_OK;
  if (!a.levels[0]) return PVO_EUNSUPPORTED;           // the factor graph's volumes live in the tiled pool
  if (!u->poses || !u->disps || !u->intrinsics || !u->ii || !u->jj || !u->target || !u->delta_dy || !u->raw_mask ||
      !u->weight || !u->full_flow || !u->target_ba || !u->weight_ba || !u->ii_ba || !u->jj_ba || !u->sys || !u->ba_ws)
    return PVO_EINVAL;
  if (K <= 0 || R <= 0 || !a.eta_frame || !a.eta_pos || !a.damping || u->n_in < 0) return PVO_EINVAL;
  const int S = u->segm ? u->max_segments : 0;
  if (u->segm && S <= 0) return PVO_EINVAL;
  if (!workspace || workspace_bytes < pvo_graph_update_workspace_bytes(E, K, R, H, W, S)) return PVO_EWORKSPACE;
  const size_t op_bytes = pvo_operator_workspace_bytes(E, K, H, W);
  UpWs s = carve_up(ws_base(workspace), E, K, R, H, W, S, op_bytes);
  OpWs b = carve_op(ws_base(s.op), E, K, H, W, pvo_gru_glo_chunks(HW));
  hipStream_t st = pvo_stream(stream);

  probe_mark(PVO_STAGE_UPDATE, 0, stream);
  // factor_graph.py:231-237: reprojection and motion features
  // (the motion features ride on the reprojection: 20-24 us of the side chain beside the lookup otherwise)
  RUN(pvo_reproject_motion(u->poses, u->disps, u->intrinsics, u->ii, u->jj, s.coords, s.valid, u->target, u->delta_dy, u->raw_mask,
                           s.motion, E, H, W, dt, stream));
  a.coords = s.coords; a.corr = nullptr; a.motion = s.motion; a.heads = s.heads;
  a.eta = u->op.eta ? u->op.eta : s.eta;          // (a caller that runs the BA itself - edge sharding - supplies the buffer)
  if (u->want_upmask && !a.upmask) a.upmask = s.upmask;
  // the gate context may already be in the workspace: computed ahead by the previous call, for exactly this input
  const bool ahead_off = pvo_knob(PVO_KNOB_NO_RIDERS) != 0;      // (pvo_debug_config: tests compare with and without the riders)
  ContextAhead* ca = ctx_ahead_slot();
  const bool ctx_ready = ca && ca->valid && u->context_ready && !ahead_off && ca->ws == workspace && ca->weights == w &&
                         ca->glo_w == w->glo_w && ca->gate_wt == w->gate_wt && ca->net == a.net && ca->E == E && ca->H == H &&
                         ca->W == W && ca->dtype == dt;
  if (ca) ca->valid = false;
  SideCtx* pending = nullptr;
  // :249-306: mask update, (panoptic vote), weights, targets in the BA's layout, full flow.  Without the vote it is the epilogue of
  // the output heads' gather (graph_post.h; pvo_debug_config(PVO_KNOB_POST_SEPARATE) keeps the launch of its own: a test compares)
  float* const tba = u->target_ba + static_cast<size_t>(u->n_in) * 2 * HW;
  float* const wba = u->weight_ba + static_cast<size_t>(u->n_in) * 2 * HW;
  const GraphPostArgs gpa = {reinterpret_cast<const float2*>(s.coords), reinterpret_cast<float2*>(u->raw_mask), reinterpret_cast<float2*>(u->target),
                             reinterpret_cast<float2*>(u->delta_dy), reinterpret_cast<float2*>(u->weight), tba, wba,
                             reinterpret_cast<float2*>(u->full_flow), u->dy_thresh};
  const bool post_in_gather = !u->segm && pvo_knob(PVO_KNOB_POST_SEPARATE) == 0;
  int post_fused = 0;
  RUN(run_operator(w, &a, b, stream, &pending, nullptr, false, ctx_ready, post_in_gather ? &gpa : nullptr, &post_fused));
  if (u->segm)
    RUN(pvo_segment_hist_ws(u->segm, u->raw_mask, s.heads, s.vote_tot, s.vote_dyn, E, HW, S, u->dy_thresh, dt, stream));
  if (!post_fused)
    RUN(pvo_graph_post(s.coords, s.heads, u->raw_mask, u->target, u->delta_dy, u->weight, tba, wba,
                       u->full_flow, E, H, W, u->dy_thresh, u->segm, u->segm ? s.vote_tot : nullptr, u->segm ? s.vote_dyn : nullptr,
                       S, u->vote_thresh, dt, stream));
  const bool rider_off = pvo_knob(PVO_KNOB_NO_RIDERS) != 0;
  // (the rider has preconditions the stand-alone convolution does not: at most 2^24 rows and 16-byte aligned operands,
  // pvo_ba_finish_riders - beyond them the mask is computed the old way, on this stream, instead of failing the update)
  const bool rider_fits = static_cast<long long>(K) * HW <= (1LL << 24) &&
                          !((reinterpret_cast<uintptr_t>(b.a2) | reinterpret_cast<uintptr_t>(w->up_w) | reinterpret_cast<uintptr_t>(a.upmask)) & 15);
  const bool mask_rides = u->itrs > 0 && a.K > 0 && a.upmask && !rider_off && rider_fits;
  // ... and so does the gate context of the NEXT update (a function of this update's net_out and the weights): its partial
  // means in the first solve, the 1x1 context convolutions in the second
  const bool context_ahead = u->context_ahead && u->itrs >= 2 && !u->motion_only && !ahead_off;
  // The upsampling mask - which nothing here reads - rides in the dispatch of the first pose solve (one workgroup solves, the
  // mask convolution's workgroups fill the idle chip: pvo_ba_finish_conv1x1), unless there is no solve to ride.  Before round 3
  // (and still without a BA iteration): The aggregation branch ends at the eta head on the side stream (`mid`); the mask
  // is computed on THIS stream between the mask / weight glue and the BA.  By then `mid` is long recorded (a satisfied wait
  // costs ~6 us on the stream, one that has to be woken ~19), the convolution takes 17 us alone instead of 27 beside the
  // glue kernels, and the BA starts ~20 us earlier than behind a join of the whole branch - while still running with the
  // side stream idle.  (Letting the BA run beside the mask convolution hides those 17 us too, but one of ~290 repeated
  // two-update runs then differed in the last bits of the poses; starting the BA's edge-block assembly before `mid`, with
  // the wait in front of the Schur step only, made 13 of 13 differ - no buffer is shared (addresses checked), agent-scope
  // loads of eta changed nothing.  Round 3, tools/sched_bisect.py: not a race in this code - BA kernels co-resident with another
  // hardware queue's kernel while cache write-back / invalidate operations are in flight read stale 64-byte sectors or lose
  // 8-byte atomics on this platform; alone on the device 0 of 3000 runs differ.  DESIGN 7g, profiles/r03_sched_bisect.txt,
  // tests/...::test_native_updates_are_reproducible.)
  if (pending && hipStreamWaitEvent(st, pending->mid, 0) != hipSuccess) return PVO_ELAUNCH;
  if (!mask_rides) RUN(run_upmask(w, &a, b, stream));
  // :302 dense bundle adjustment on [inactive | active] edges, planned by the caller (pvo_ba_plan) for this edge set
  const int Eb = u->n_in + E;
  probe_mark(PVO_STAGE_BA, 0, stream);
  // `sys` is zero on entry (contract, see the header) and every solve leaves it zero: no memset between updates.  The
  // depth clamp of depth_video.py:214 rides on the last back-substitution.
  for (int it = 0; it < u->itrs; ++it) {
    const bool last = it + 1 == u->itrs;
    RUN(pvo_ba_local(u->poses, u->disps, u->intrinsics, u->target_ba, u->weight_ba, a.eta, u->ii_ba, u->jj_ba, Eb, u->nframes,
                     H, W, R, u->t0, u->t1, (u->motion_only ? 1 : 0) | 2, u->sys, u->ba_ws, u->ba_ws_bytes, stream));
    pvo_ba_riders jobs{};
    if (mask_rides && it == 0) {
      jobs.cx = b.a2; jobs.cw = w->up_w; jobs.cbias = w->up_b; jobs.cy = a.upmask;
      jobs.crows = static_cast<long long>(K) * HW; jobs.cCout = 576; jobs.cdtype = dt;
    }
    if (context_ahead && it == 0) {          // the next update's input is this update's net_out
      jobs.gnet = a.net_out; jobs.gw = w->glo_w; jobs.gbias = w->glo_b; jobs.gpart = b.part; jobs.gE = E; jobs.gHW = HW; jobs.gdtype = dt;
    }
    if (context_ahead && it == 1) {
      jobs.xpart = b.part; jobs.xwt = w->gate_wt; jobs.xbias = w->gate_b; jobs.xg = b.g; jobs.xE = E; jobs.xchunks = pvo_gru_glo_chunks(HW);
    }
    RUN(pvo_ba_finish_riders(u->poses, u->disps, u->sys, u->ii_ba, u->jj_ba, Eb, u->nframes, H, W, u->t0, u->t1, u->lm, u->ep,
                             u->motion_only, last && !u->motion_only ? u->clamp_frames : 0, u->disp_min,
                             nullptr, nullptr, 0, nullptr, u->ba_ws, u->ba_ws_bytes, &jobs, stream));
  }
  probe_mark(PVO_STAGE_BA, 1, stream);
  if (u->clamp_frames > 0 && (u->itrs == 0 || u->motion_only)) {      // (no back-substitution ran: clamp on its own)
    const long long n = static_cast<long long>(u->clamp_frames) * HW;
    hipLaunchKernelGGL(clamp_min_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, u->disps, n, u->disp_min);
    PVO_CHECK_LAUNCH();
  }
  // No second join here: the side stream's last work is the eta head, `mid` was recorded behind it, and this stream has waited
  // for `mid` above (the branch's `join` event is recorded right behind `mid` with nothing in between).  A satisfied wait
  // still costs the launch stream ~6 us (tools/update_timeline.sh).
  if (ca && context_ahead)
    *ca = ContextAhead{workspace, w, w->glo_w, w->gate_wt, a.net_out, E, H, W, dt, true};
  probe_mark(PVO_STAGE_UPDATE, 1, stream);
  return PVO_OK;
}

extern "C" int pvo_side_stream(void** stream_out) {
  if (!stream_out) return PVO_EINVAL;
  SideCtx* sc = side_ctx();
  if (!sc) return PVO_ELAUNCH;
  *stream_out = static_cast<void*>(sc->side);
  return PVO_OK;
}

extern "C" int pvo_probe_arm_every(int stage, int capacity, int every) {
  if (capacity < 0 || every < 1) return PVO_EINVAL;
  if (capacity > g_probe.cap) {
    hipEvent_t* ev = static_cast<hipEvent_t*>(realloc(g_probe.ev, sizeof(hipEvent_t) * 2 * capacity));
    if (!ev) return PVO_EINVAL;
    for (int i = 2 * g_probe.cap; i < 2 * capacity; ++i)
      if (hipEventCreate(&ev[i]) != hipSuccess) return PVO_ELAUNCH;
    g_probe.ev = ev; g_probe.cap = capacity;
  }
  g_probe.n = 0; g_probe.seen = 0; g_probe.limit = capacity; g_probe.every = every;
  g_probe.stage = capacity > 0 ? stage : -1;
  return PVO_OK;
}

extern "C" int pvo_probe_arm(int stage, int capacity) { return pvo_probe_arm_every(stage, capacity, 1); }

extern "C" int pvo_probe_read(float* ms_host, int max_n) {
  const int n = g_probe.n < max_n ? g_probe.n : max_n;
  for (int i = 0; i < n; ++i) {
    if (hipEventSynchronize(g_probe.ev[2 * i + 1]) != hipSuccess) return -1;
    if (hipEventElapsedTime(&ms_host[i], g_probe.ev[2 * i], g_probe.ev[2 * i + 1]) != hipSuccess) return -1;
  }
  g_probe.stage = -1;
  return n;
}
