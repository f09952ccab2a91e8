// capi_misc.hip — error strings / version of libpvo_hip.  (The shader-clock and memory-request probes bench.py reports moved to
// probe_tools.hip -> libpvo_probe.so in round 4: measurement kernels are not part of the product library.)
#include "common.h"

extern "C" const char* pvo_strerror(int code) {
  switch (code) {
    case PVO_OK: return "ok";
    case PVO_EINVAL: return "invalid argument";
    case PVO_ELAUNCH: return "HIP launch error";
    case PVO_EWORKSPACE: return "workspace too small";
    case PVO_EUNSUPPORTED: return "unsupported size or configuration";
    default: return "unknown error";
  }
}

// 101: pvo_graph_update_args grew by context_ahead / context_ready (round 3) - a caller built against a 100 header passes a
// shorter struct; pvo_graph_update_args_size() lets any caller compare its sizeof with the library's before the first call
// 102: pvo_ba_pack / pvo_ba_finish_packed / pvo_ba_last_partition added, the BA workspace grew (round 4)
// 103: pvo_debug_config / pvo_knob replace the library's environment switches (round 5)
extern "C" int pvo_version(void) { return PVO_ABI_VERSION; }

static int g_knobs[PVO_KNOB_COUNT] = {};
extern "C" int pvo_debug_config(int knob, int value) {
  if (knob < 0 || knob >= PVO_KNOB_COUNT) return PVO_EINVAL;
  g_knobs[knob] = value;
  return PVO_OK;
}
extern "C" int pvo_knob(int knob) { return (knob < 0 || knob >= PVO_KNOB_COUNT) ? 0 : g_knobs[knob]; }
extern "C" size_t pvo_graph_update_args_size(void) { return sizeof(pvo_graph_update_args); }

static thread_local int g_last_hip_error = 0;
extern "C" void pvo_note_hip_error(int code) { g_last_hip_error = code; }
extern "C" const char* pvo_last_hip_error(void) {
  return g_last_hip_error ? hipGetErrorString(static_cast<hipError_t>(g_last_hip_error)) : "no HIP error recorded";
}

// ---------------------------------------------------------------------------------------------------------------------------------
// pvo_proximity_select: the greedy edge selection of FactorGraph.add_proximity_factors (VO_Module/droid_slam/factor_graph.py:372-429)
// on the HOST, from the distance matrix the device produced.  No device work: it is here because the selection sits on the host's
// critical path of every keyframe - between the arrival of the distances and the launch of the first graph update the device has nothing
// queued - and as numpy it cost 0.2 ms per frontend window and ~15 ms per global graph (a Python loop over the accepted edges).
// Semantics = pvo_amd/factor_graph.py's array form, which tests pin to the reference's loops on recorded matrices:
//   D[a][b] is the distance of frames (t0 + a, t1 + b);  cells with (i - rad < j) or !(D <= 100) are out;  around every existing edge
//   (i, j) with |i - j| > 2 the diamond |di| + |dj| <= min(|i - j| - 2, nms) is out;  the temporal neighbours (i, j), i < j <= i + rad, are
//   taken both ways first;  then the remaining cells in ascending distance (stable: ties by flat index) while <= thresh, each accepted
//   edge taken both ways and suppressing its diamond.
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>
extern "C" int pvo_proximity_select(const float* dist, int ni, int nj, int t0, int t1, int rad, int nms, double thresh,
                                    const long long* have_i, const long long* have_j, int n_have,
                                    long long* out_i, long long* out_j, int max_out, int* n_out) {
  if (!dist || !out_i || !out_j || !n_out || ni <= 0 || nj <= 0 || rad < 0 || nms < 0 || n_have < 0 || (n_have > 0 && (!have_i || !have_j)))
    return PVO_EINVAL;
  const double inf = std::numeric_limits<double>::infinity();
  const int t = t0 + ni;                                  // frames [t0, t) x [t1, t): both ranges end at the video's counter
  if (t1 + nj != t) return PVO_EINVAL;
  std::vector<double> D(static_cast<size_t>(ni) * nj);
  for (int a = 0; a < ni; ++a)
    for (int b = 0; b < nj; ++b) {
      const double d = static_cast<double>(dist[static_cast<size_t>(a) * nj + b]);
      D[static_cast<size_t>(a) * nj + b] = ((t0 + a) - rad < (t1 + b) || !(d <= 100.0)) ? inf : d;
    }
  auto suppress = [&](long long i, long long j) {
    const long long gap = (i > j ? i - j : j - i) - 2;
    const int r = static_cast<int>(std::max<long long>(std::min<long long>(gap, nms), 0));
    for (int di = -r; di <= r; ++di)
      for (int dj = -(r - std::abs(di)); dj <= r - std::abs(di); ++dj) {
        const long long a = i + di - t0, b = j + dj - t1;
        if (a >= 0 && a < ni && b >= 0 && b < nj) D[static_cast<size_t>(a) * nj + b] = inf;
      }
  };
  for (int k = 0; k < n_have; ++k) {
    const long long gap = have_i[k] > have_j[k] ? have_i[k] - have_j[k] : have_j[k] - have_i[k];
    if (gap > 2) suppress(have_i[k], have_j[k]);
  }
  int n = 0;
  auto take = [&](long long i, long long j) -> bool {
    if (n + 2 > max_out) return false;
    out_i[n] = i; out_j[n] = j; out_i[n + 1] = j; out_j[n + 1] = i;
    n += 2;
    return true;
  };
  for (int i = t0; i < t; ++i)
    for (int j = i + 1; j < std::min(i + rad + 1, t); ++j)
      if (!take(i, j)) return PVO_EWORKSPACE;
  std::vector<int> order;
  order.reserve(D.size());
  const double th = thresh;
  for (size_t k = 0; k < D.size(); ++k)
    if (D[k] <= th) order.push_back(static_cast<int>(k));
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return D[x] < D[y]; });
  for (int k : order) {
    if (!(D[k] <= th)) continue;                          // suppressed since the sort
    const long long i = t0 + k / nj, j = t1 + k % nj;
    if (!take(i, j)) return PVO_EWORKSPACE;
    suppress(i, j);
  }
  *n_out = n;
  return PVO_OK;
}
