// ba.hip — dense bundle adjustment for gfx950 (MI355X), device-resident end to end.
//
// Replaces ba_cuda and everything under it (reference VO_Module/src/droid_kernels.cu:
// projective_transform_kernel :177-403, accum_cuda/accum_kernel :833-853,927-977,
// EEt6x6/Ev6x1/EvT6x1 :980-1094, SparseBlock :1096-1198, schur_block :1201-1290,
// pose/disp retraction :856-925, ba_cuda :1293-1410).
//
// The reference runs ~12 launches and >=8 device<->host copies per iteration: the
// CSR index lists, the (a,b,k) Schur triple list and the Eigen sparse LLT are all
// built / solved on the host.  Here nothing leaves the device:
//
//   plan     (once per call, 1 workgroup)  kx = unique([t0,t1) U ii) as a presence
//            bitmap + scan, per-depth-frame CSR of outgoing edges.
//   per Gauss-Newton step, 5 launches:
//   assemble grid (pixel chunk, edge): Jacobians per pixel in registers; the 78+12
//            pose-block sums are wave-shuffle reduced (no 256-float LDS tree per scalar)
//            and added into the dense fp64 pose system with fp64 atomics; Eii/Eij/Cii/bz
//            stored for the depth elimination.
//   depth    grid (pixel chunk, depth frame): C, w, Q = 1/(C+eta) and the window rows Ei.
//   schur    grid (pixel chunk, depth frame): all rows coupling this depth frame to free
//            poses form M; (M Q) M^T and M (Q w) are accumulated by fp32 MFMA
//            (v_mfma_f32_16x16x4_f32, exact fp32 FMA chains, K = pixels) and atomically
//            SUBTRACTED from the system.
//   solve    one workgroup: damping, fp64 Cholesky (LDS when (6P)^2 fits, global
//            otherwise), forward/back substitution, dx, pose retraction.
//   backsub  grid (pixel chunk, depth frame): dz = Q (w - sum_r E_r^T dx), disps += dz.
//
// `sys` = [(6P)^2 row-major A-S | 6P rhs] in fp64 is also the message an edge-sharded
// multi-GPU run all-reduces between `schur` and `solve` (pvo_ba_local / pvo_ba_finish).
//
// Semantics kept from the reference: MIN_DEPTH 0.25 on the TARGET depth only, weights
// scaled by 1e-3, poses below t0 fixed, fp64 solve with `diag += ep + lm*diag`, zero
// update when the factorisation fails, EvT6x1's skip of window pose 0 in the depth
// back-substitution (:1084).  Deviation: expSE3 uses xi[5], not xi[45] (:154).
#include "se3.h"
#include "conv1x1_tile.h"
#include "glo_tile.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr float kMinDepth = 0.25f;

// The reduced pose system is accumulated in 64-bit FIXED POINT (units of 2^-28): integer atomics commute, so the sums - and
// with them poses and depths - are bitwise reproducible from run to run, and identical whether the edges sit on one GPU or
// are sharded over several (fp64 atomics, used before, made the last bits order dependent).  Every addend is an fp32
// partial sum (|v| <~ 1e8 at most: w J^2 over a chunk of pixels), so 2^-28 = 3.7e-9 absolute resolution is far below its
// own rounding error, and |sum| < 2^35 leaves 7 bits of headroom.  Non-finite or out-of-range addends raise meta[4]; the
// solve then takes the reference's failure path (zero update, droid_kernels.cu:1186-1189).
constexpr double kFix = 268435456.0, kInvFix = 1.0 / 268435456.0;
__device__ __forceinline__ void fix_add(long long* sys, long long idx, double v, int* meta) {
  if (!(fabs(v) < 3.0e10)) { meta[4] = 1; return; }
#ifdef PVO_BA_NO_ATOMICS                 // (timing experiment only - tools/ba_kernel_timeline.py PROBE_DEFS: the sums are lost)
  if (idx >= 0) return;
#endif
  atomicAdd(reinterpret_cast<unsigned long long*>(sys + idx), static_cast<unsigned long long>(__double2ll_rn(v * kFix)));
}
constexpr int kDealEdges = 2;           // edges (consecutive in the plan's by-source order) per chunk-sum workgroup of the Schur kernel
constexpr int kPPT = 2;                 // pixels per thread in assemble (1: 432 workgroups, measured slower - profiles/r03_ba_ablation.txt)
constexpr int kChunkA = 256 * kPPT;     // pixels per assemble workgroup
constexpr int kMaxSep = 12;                                      // separator poses of the partitioned solve (beyond: not partitioned)
constexpr int kXchgDoubles = 2 + 36 * kMaxSep * kMaxSep + 12 * kMaxSep;
constexpr int kLdsCholMax = 126;        // (6P) up to which the fp64 system lives DENSE in LDS (126*127*8 + 21*27*8 + 208 = 132.7 KB of the 143 KB the
                                        // solve kernel's static tables leave); beyond, the compact envelope form


// tools/ba_kernel_timeline.py builds this file with -DPVO_BA_PROBE=3: every workgroup of the assembly / Schur / back-substitution
// kernels stamps the constant-rate clock (10 ns) at its phases: g_ba_wg_probe[(kernel * 4096 + workgroup) * 8 + slot]
#if defined(PVO_BA_PROBE) && PVO_BA_PROBE == 3
__device__ unsigned long long* g_ba_wg_probe = nullptr;
#define BA_WG_PROBE(kern, slot)                                                                                         \
  do {                                                                                                                  \
    if (g_ba_wg_probe && threadIdx.x == 0 && blockIdx.z == 0) {                                                         \
      const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                                         \
      if (wg_ < 4096u) g_ba_wg_probe[((kern) * 4096u + wg_) * 8u + (slot)] = wall_clock64();                            \
    }                                                                                                                   \
  } while (0)
#else
#define BA_WG_PROBE(kern, slot)
#endif

struct Plan {            // int region of the workspace
  int* kidx;             // [F]   frame -> depth index, -1 if none
  int* kx;               // [F]   depth index -> frame
  int* eptr;             // [F+1] CSR over depth index -> edges (ascending edge id)
  int* eidx;             // [E]
  int* meta;             // [8]   0:K 1:status(non-SPD) 2:eta mismatch 3:row table overflow 4:non-finite / out-of-range system entry
  int* env;              // [P]   numeric envelope of a system too large for the dense LDS path (ba_env_kernel; INT_MAX between solves)
};

struct Ws {
  Plan plan;
  float *Eii, *Eij;      // [E][6][HW]
  float *Cii, *bz;       // [E][HW]
  float *Ei;             // [P][6][HW]
  float *Q, *w;          // [F'][HW]   (F' = min(F, P+E) rows)
  float *dx;             // [P][6]
  float *part;           // [E][assembly chunks][90]: per-chunk sums of an edge's pose blocks and gradients (depth BA)
  long long* sys;        // [(6P)^2 + 6P] fixed point
  double* chol;          // [(6P)^2 + 6P] scratch for the global-memory factorisation
  double* xchg;          // partitioned pose solve: [2 doubles = 4 ints: flags, split | separator terms | separator solution] (kXchgDoubles)
  double* ldiag;         // blocked multi-workgroup solve (big dense-ish systems): factored diagonal blocks [n / 48 + 1][48 x 48 + 48]
  double* xvec;          // ... and its solution [6P]
  float* Mrg;            // [E][6][HW]: the summed Eij rows of edges that share source AND target frame (Schur kernel), at the first one's index
  // dense windows (P <= kDenseMaxPoses), two-stage Schur sums: per (depth frame, 256-pixel chunk) the tile-pair sums of the chunk,
  // the frame's row -> system-entry table and its tile count (0 = this frame went the atomic way)
  float* spart;          // [kSchurStageFrames(P)][ceil(HW/256)][kSchurStagePairs][256]
  short* srow;           // [kSchurStageFrames(P)][128]
  int* sT;               // [kSchurStageFrames(P)]
  size_t bytes;
};
constexpr int kSchurStagePairs = 36;     // tile pairs of up to 8 row tiles
__host__ __device__ __forceinline__ int schur_stage_frames(int P) { return (P > 0 && P <= 29) ? P + 8 : 0; }

__host__ size_t align_up(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

__host__ Ws carve(void* base, int E, int P, int F, int HW) {
  Ws w{};
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) { char* r = p ? p + off : nullptr; off += align_up(bytes); return r; };
  const int Kmax = (F < P + E) ? F : (P + E);
  const size_t n6 = static_cast<size_t>(6) * (P > 0 ? P : 0);
  w.plan.kidx = reinterpret_cast<int*>(take(sizeof(int) * (F + 1)));
  w.plan.kx = reinterpret_cast<int*>(take(sizeof(int) * (F + 1)));
  w.plan.eptr = reinterpret_cast<int*>(take(sizeof(int) * (F + 2)));
  w.plan.eidx = reinterpret_cast<int*>(take(sizeof(int) * (E + 1)));
  w.plan.meta = reinterpret_cast<int*>(take(sizeof(int) * 8));
  w.plan.env = reinterpret_cast<int*>(take(sizeof(int) * ((P > 0 ? P : 0) + 1)));
  w.Eii = reinterpret_cast<float*>(take(sizeof(float) * static_cast<size_t>(E) * 6 * HW));
  w.Eij = reinterpret_cast<float*>(take(sizeof(float) * static_cast<size_t>(E) * 6 * HW));
  w.Cii = reinterpret_cast<float*>(take(sizeof(float) * static_cast<size_t>(E) * HW));
  w.bz = reinterpret_cast<float*>(take(sizeof(float) * static_cast<size_t>(E) * HW));
  w.Ei = reinterpret_cast<float*>(take(sizeof(float) * static_cast<size_t>(P > 0 ? P : 0) * 6 * HW));
  w.Q = reinterpret_cast<float*>(take(sizeof(float) * static_cast<size_t>(Kmax) * HW));
  w.w = reinterpret_cast<float*>(take(sizeof(float) * static_cast<size_t>(Kmax) * HW));
  w.dx = reinterpret_cast<float*>(take(sizeof(float) * (n6 + 8)));
  w.part = reinterpret_cast<float*>(take(sizeof(float) * static_cast<size_t>(E) * ((HW + kChunkA - 1) / kChunkA) * 90 + 16));
  w.sys = reinterpret_cast<long long*>(take(sizeof(double) * (n6 * n6 + n6 + 8)));
  // (both at every size: a packed envelope message - pvo_ba_finish_packed - is factorised from the compact image whatever P is)
  w.chol = reinterpret_cast<double*>(take(sizeof(double) * (n6 * n6 + n6 + 27 * (n6 / 6) + 32)));
  w.xchg = reinterpret_cast<double*>(take(sizeof(double) * kXchgDoubles));
  w.ldiag = reinterpret_cast<double*>(take(sizeof(double) * (n6 / 48 + 2) * (48 * 48 + 48)));
  w.xvec = reinterpret_cast<double*>(take(sizeof(double) * (n6 + 8)));
  w.Mrg = reinterpret_cast<float*>(take(sizeof(float) * static_cast<size_t>(E) * 6 * HW));
  {
    const size_t ks = static_cast<size_t>(schur_stage_frames(P));
    w.spart = reinterpret_cast<float*>(take(sizeof(float) * ks * ((HW + 255) / 256) * kSchurStagePairs * 256));
    w.srow = reinterpret_cast<short*>(take(sizeof(short) * ks * 128));
    w.sT = reinterpret_cast<int*>(take(sizeof(int) * (ks + 1)));
  }
  w.bytes = off;
  return w;
}

// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_plan_kernel(
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, Plan pl,
    int E, int F, int t0, int t1, int K_eta, int motion_only) {
  __shared__ int seg[257];
  const int tid = threadIdx.x;
  for (int b = tid; b < t1 - t0; b += 256) pl.env[b] = 0x7fffffff;
  // presence bitmap
  for (int f = tid; f < F; f += 256) pl.kidx[f] = (f >= t0 && f < t1) ? 1 : 0;
  __syncthreads();
  for (int e = tid; e < E; e += 256) {
    const long long f = ii[e];
    if (f >= 0 && f < F) pl.kidx[f] = 1;   // benign race: everyone writes 1
  }
  __syncthreads();
  // exclusive scan of the bitmap, one contiguous segment per thread
  const int per = (F + 255) / 256;
  const int lo = min(tid * per, F), hi = min(lo + per, F);
  int cnt = 0;
  for (int f = lo; f < hi; ++f) cnt += pl.kidx[f];
  seg[tid + 1] = cnt;
  if (tid == 0) seg[0] = 0;
  __syncthreads();
  if (tid == 0) for (int i = 1; i <= 256; ++i) seg[i] += seg[i - 1];
  __syncthreads();
  int run = seg[tid];
  for (int f = lo; f < hi; ++f) {
    if (pl.kidx[f]) { pl.kx[run] = f; pl.kidx[f] = run; ++run; } else pl.kidx[f] = -1;
  }
  const int K = seg[256];
  __syncthreads();
  // CSR of edges by depth index (= source frame), ascending edge id inside a row
  for (int k = tid; k < K; k += 256) {
    const int f = pl.kx[k];
    int c = 0;
    for (int e = 0; e < E; ++e) c += (ii[e] == f) ? 1 : 0;
    pl.eptr[k + 1] = c;
  }
  if (tid == 0) pl.eptr[0] = 0;
  __syncthreads();
  if (tid == 0) for (int k = 1; k <= K; ++k) pl.eptr[k] += pl.eptr[k - 1];
  __syncthreads();
  for (int k = tid; k < K; k += 256) {
    const int f = pl.kx[k];
    int o = pl.eptr[k];
    for (int e = 0; e < E; ++e) if (ii[e] == f) pl.eidx[o++] = e;
  }
  if (tid == 0) {
    pl.meta[0] = K;
    pl.meta[1] = 0;
    pl.meta[3] = 0;
    pl.meta[4] = 0;
    // eta must have one row per depth frame (or one row, broadcast).  Anything else is a caller error the host cannot see
    // without a synchronisation (K is found here): flagged, and the step becomes a NO-OP - the Schur grid is sized by the caller's
    // row count, so depth frames beyond it would keep stale Q / w; the solve reports failure (dx = 0, poses untouched) and the
    // back-substitution leaves every depth map alone (status_out[0] = 1, [2] = 1).
    pl.meta[2] = (!motion_only && K_eta != K && K_eta != 1) ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------
// assemble
// ---------------------------------------------------------------------------
struct EdgeGeom { Pose G; float fx, fy, cx, cy; };

// one pixel of projective_transform_kernel (droid_kernels.cu:266-357): accumulates the
// upper triangle h[78] of [Ji Jj]^T W [Ji Jj], the gradients vi/vj, and returns the
// depth-coupling terms.
__device__ __forceinline__ void pixel_terms(const EdgeGeom& g, float u, float v, float disp,
                                            float tu, float tv, float wgu, float wgv,
                                            float (&h)[78], float (&vi)[6], float (&vj)[6],
                                            float (&eii)[6], float (&eij)[6], float& cii, float& bzz) {
  float Xi[4] = {(u - g.cx) / g.fx, (v - g.cy) / g.fy, 1.0f, disp};
  float Xj[4];
  act4(g.G, Xi, Xj);
  const float x = Xj[0], y = Xj[1], hh = Xj[3];
  const bool ok = !(Xj[2] < kMinDepth);
  const float d = ok ? 1.0f / Xj[2] : 0.0f;
  const float d2 = d * d;
  // droid_kernels.cu:290-291 write `.001 * weight`: a DOUBLE literal, so the product is formed in double and rounded to
  // float once - not the same number as 0.001f * w (0.001f != 0.001).  Followed literally; `1.0 / Xj[2]` (:287) equals the
  // correctly rounded float division (double rounding of a quotient is innocuous at 53 >= 2 * 24 + 2 bits).
  const float wu = ok ? static_cast<float>(0.001 * static_cast<double>(wgu)) : 0.0f;
  const float wv = ok ? static_cast<float>(0.001 * static_cast<double>(wgv)) : 0.0f;
  const float ru = tu - (g.fx * d * x + g.cx);
  const float rv = tv - (g.fy * d * y + g.cy);
  float J[12];   // [Ji | Jj]
  float Jz;
  // ---- x residual
  J[6] = g.fx * (hh * d); J[7] = 0.0f; J[8] = g.fx * (-x * hh * d2);
  J[9] = g.fx * (-x * y * d2); J[10] = g.fx * (1.0f + x * x * d2); J[11] = g.fx * (-y * d);
  Jz = g.fx * (g.G.t.x * d - g.G.t.z * (x * d2));
  adjT(g.G, &J[6], &J[0]);
#pragma unroll
  for (int n = 0; n < 6; ++n) J[n] = -J[n];
  {
    int l = 0;
#pragma unroll
    for (int n = 0; n < 12; ++n) {
      const float wj = wu * J[n];
#pragma unroll
      for (int m = 0; m <= n; ++m) { h[l] += wj * J[m]; ++l; }
    }
  }
#pragma unroll
  for (int n = 0; n < 6; ++n) {
    vi[n] += wu * ru * J[n];
    vj[n] += wu * ru * J[6 + n];
    eii[n] = wu * Jz * J[n];
    eij[n] = wu * Jz * J[6 + n];
  }
  cii = wu * Jz * Jz;
  bzz = wu * ru * Jz;
  // ---- y residual
  J[6] = 0.0f; J[7] = g.fy * (hh * d); J[8] = g.fy * (-y * hh * d2);
  J[9] = g.fy * (-1.0f - y * y * d2); J[10] = g.fy * (x * y * d2); J[11] = g.fy * (x * d);
  Jz = g.fy * (g.G.t.y * d - g.G.t.z * (y * d2));
  adjT(g.G, &J[6], &J[0]);
#pragma unroll
  for (int n = 0; n < 6; ++n) J[n] = -J[n];
  {
    int l = 0;
#pragma unroll
    for (int n = 0; n < 12; ++n) {
      const float wj = wv * J[n];
#pragma unroll
      for (int m = 0; m <= n; ++m) { h[l] += wj * J[m]; ++l; }
    }
  }
#pragma unroll
  for (int n = 0; n < 6; ++n) {
    vi[n] += wv * rv * J[n];
    vj[n] += wv * rv * J[6 + n];
    eii[n] += wv * Jz * J[n];
    eij[n] += wv * Jz * J[6 + n];
  }
  cii += wv * Jz * Jz;
  bzz += wv * rv * Jz;
}

// Entry t of an edge's 90 sums - upper triangle of [Ji Jj]^T W [Ji Jj] (78, droid_kernels.cu:309-315 ordering), vi (6), vj (6) -
// added to the pose system in fixed point.
__device__ __forceinline__ void pose_block_scatter(int t, double val, int pi, int pj, int P, long long* __restrict__ sys, int* __restrict__ meta) {
  const bool iok = pi >= 0 && pi < P, jok = pj >= 0 && pj < P;
  const int n6 = 6 * P;
  if (t < 78) {
    // invert l -> (n, m), m <= n
    int n = 0, base = 0;
    while (base + n + 1 <= t) { base += n + 1; ++n; }
    const int m = t - base;
    if (n < 6) {                                  // (ii,ii), symmetric
      if (iok) {
        fix_add(sys, static_cast<long long>(6 * pi + n) * n6 + 6 * pi + m, val, meta);
        if (n != m) fix_add(sys, static_cast<long long>(6 * pi + m) * n6 + 6 * pi + n, val, meta);
      }
    } else if (m < 6) {                           // (ii,jj)[m][n-6] and (jj,ii)[n-6][m]
      // only the LOWER block triangle of the system is ever read (the solve factorises it; an edge-sharded run all-reduces
      // exactly those blocks): of the two mirrored off-diagonal blocks the one above the diagonal is not written.  Nothing at
      // window size; a global bundle adjustment's Schur kernel is bound by these atomics (round 4)
      if (iok && jok) {
        if (pi > pj) fix_add(sys, static_cast<long long>(6 * pi + m) * n6 + 6 * pj + (n - 6), val, meta);
        else fix_add(sys, static_cast<long long>(6 * pj + (n - 6)) * n6 + 6 * pi + m, val, meta);
      }
    } else {                                      // (jj,jj), symmetric
      if (jok) {
        fix_add(sys, static_cast<long long>(6 * pj + n - 6) * n6 + 6 * pj + m - 6, val, meta);
        if (n != m) fix_add(sys, static_cast<long long>(6 * pj + m - 6) * n6 + 6 * pj + n - 6, val, meta);
      }
    }
  } else if (t < 84) {
    if (iok) fix_add(sys, static_cast<long long>(n6) * n6 + 6 * pi + (t - 78), val, meta);
  } else {
    if (jok) fix_add(sys, static_cast<long long>(n6) * n6 + 6 * pj + (t - 84), val, meta);
  }
}

// The (at most two) entries of the pose system pose_block_scatter adds entry t of edge (pi -> pj) to; -1 = none.  Same case
// analysis, kept beside it.
__device__ __forceinline__ void pose_block_targets(int t, int pi, int pj, int P, long long idx[2]) {
  const bool iok = pi >= 0 && pi < P, jok = pj >= 0 && pj < P;
  const long long n6 = 6 * P;
  idx[0] = idx[1] = -1;
  if (t < 78) {
    int n = 0, base = 0;
    while (base + n + 1 <= t) { base += n + 1; ++n; }
    const int m = t - base;
    if (n < 6) {
      if (iok) { idx[0] = (6 * pi + n) * n6 + 6 * pi + m; if (n != m) idx[1] = (6 * pi + m) * n6 + 6 * pi + n; }
    } else if (m < 6) {
      if (iok && jok) idx[0] = (pi > pj) ? (6 * pi + m) * n6 + 6 * pj + (n - 6) : (6 * pj + (n - 6)) * n6 + 6 * pi + m;
    } else {
      if (jok) { idx[0] = (6 * pj + n - 6) * n6 + 6 * pj + m - 6; if (n != m) idx[1] = (6 * pj + m - 6) * n6 + 6 * pj + n - 6; }
    }
  } else if (t < 84) {
    if (iok) idx[0] = n6 * n6 + 6 * pi + (t - 78);
  } else {
    if (jok) idx[0] = n6 * n6 + 6 * pj + (t - 84);
  }
}

__global__ __launch_bounds__(256) void ba_assemble_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intr,
    const float* __restrict__ targets, const float* __restrict__ weights,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj,
    float* __restrict__ Eii, float* __restrict__ Eij, float* __restrict__ Cii, float* __restrict__ bz,
    long long* __restrict__ sys, int* __restrict__ meta, int HW, int wd, int t0, int P, int motion_only, float* __restrict__ part) {
  // Every (edge, pixel chunk) workgroup ends with 90 sums.  Added to `sys` right here they are 216 x 90 memory-side atomics on
  // ~1800 addresses at S-B - each address serialises the ~30 workgroups that hit it, 5 of this kernel's 14 us (ablation build).
  // In a depth BA (`part`) the chunk sums are stored instead and the Schur kernel, which follows anyway, adds them up per edge:
  // a sixth of the atomics, issued while its own work starts.
  __shared__ float red[4][90];
  BA_WG_PROBE(0, 0);
  const int e = blockIdx.y;
  const int ix = static_cast<int>(ii[e]), jx = static_cast<int>(jj[e]);
  EdgeGeom g;
  g.G = rel_pose(load_pose(poses + 7 * static_cast<long long>(ix)), load_pose(poses + 7 * static_cast<long long>(jx)));
  g.fx = intr[0]; g.fy = intr[1]; g.cx = intr[2]; g.cy = intr[3];

  float h[78], vi[6], vj[6];
#pragma unroll
  for (int l = 0; l < 78; ++l) h[l] = 0.0f;
#pragma unroll
  for (int n = 0; n < 6; ++n) { vi[n] = 0.0f; vj[n] = 0.0f; }

  const float* __restrict__ d_i = disps + static_cast<long long>(ix) * HW;
  const float* __restrict__ tg = targets + static_cast<long long>(e) * 2 * HW;
  const float* __restrict__ wg = weights + static_cast<long long>(e) * 2 * HW;
#pragma unroll
  for (int s = 0; s < kPPT; ++s) {
    const int k = blockIdx.x * kChunkA + s * 256 + threadIdx.x;
    if (k < HW) {
      const int i = k / wd, j = k - i * wd;
      float eii[6], eij[6], cii, bzz;
      pixel_terms(g, static_cast<float>(j), static_cast<float>(i), d_i[k], tg[k], tg[HW + k], wg[k], wg[HW + k],
                  h, vi, vj, eii, eij, cii, bzz);
      if (!motion_only) {
        const long long eb = static_cast<long long>(e) * 6 * HW + k;
#pragma unroll
        for (int n = 0; n < 6; ++n) { Eii[eb + static_cast<long long>(n) * HW] = eii[n]; Eij[eb + static_cast<long long>(n) * HW] = eij[n]; }
        Cii[static_cast<long long>(e) * HW + k] = cii;
        bz[static_cast<long long>(e) * HW + k] = bzz;
      }
    }
  }
  BA_WG_PROBE(0, 1);                     // pixel terms done (stores issued)
  // 90 sums per wave.  Ninety separate wave reductions (6 DPP steps each) were 5.2 of this kernel's 9.5 us at S-B
  // (profiles/r04_ba_kernel_timeline.txt); a reduce-SCATTER does the same work in a third of the instructions: the values sit in
  // 96 registers, and at every step a lane and its partner each keep one half of the array and add the other's copy of it -
  // 48, 24, 12, 6, 3 additions, then one plain exchange.  Partners: lane ^ 32 and lane ^ 16 by v_permlane32_swap /
  // v_permlane16_swap (gfx950: the two halves change places in one instruction, no select), then the DPP mirrors inside a row
  // (i <-> 15 - i, i <-> 7 - i, i <-> 3 - i) and quad_perm [1,0,3,2]; which half a lane keeps is its bit 5, 4, 3, 2, 1 in turn,
  // so lane L ends with the sums of entries [48 b5 + 24 b4 + 12 b3 + 6 b2 + 3 b1, + 3).  A fixed order of additions, as before.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {
    float c[96];
#pragma unroll
    for (int l = 0; l < 78; ++l) c[l] = h[l];
#pragma unroll
    for (int n = 0; n < 6; ++n) { c[78 + n] = vi[n]; c[84 + n] = vj[n]; c[90 + n] = 0.0f; }
    auto fbits = [](float x) { return __builtin_bit_cast(unsigned, x); };
    auto bitsf = [](unsigned x) { return __builtin_bit_cast(float, x); };
#pragma unroll
    for (int q = 0; q < 48; ++q) {
      const auto r = __builtin_amdgcn_permlane32_swap(fbits(c[q]), fbits(c[q + 48]), false, false);
      c[q] = bitsf(r[0]) + bitsf(r[1]);
    }
#pragma unroll
    for (int q = 0; q < 24; ++q) {
      const auto r = __builtin_amdgcn_permlane16_swap(fbits(c[q]), fbits(c[q + 24]), false, false);
      c[q] = bitsf(r[0]) + bitsf(r[1]);
    }
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      const float keep = b3 ? c[q + 12] : c[q], send = b3 ? c[q] : c[q + 12];
      c[q] = keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x140, 0xf, 0xf, true));      // row_mirror
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const float keep = b2 ? c[q + 6] : c[q], send = b2 ? c[q] : c[q + 6];
      c[q] = keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x141, 0xf, 0xf, true));      // row_half_mirror
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float keep = b1 ? c[q + 3] : c[q], send = b1 ? c[q] : c[q + 3];
      c[q] = keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x1b, 0xf, 0xf, true));       // quad_perm [3,2,1,0]
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
      c[q] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c[q]), 0xb1, 0xf, 0xf, true));             // quad_perm [1,0,3,2]
    const int start = ((lane >> 5) & 1) * 48 + ((lane >> 4) & 1) * 24 + ((lane >> 3) & 1) * 12 + ((lane >> 2) & 1) * 6 + ((lane >> 1) & 1) * 3;
    if (!(lane & 1)) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
        if (start + q < 90) red[wave][start + q] = c[q];
    }
  }
  __syncthreads();
  BA_WG_PROBE(0, 2);                     // wave reductions done
  const int t = threadIdx.x;
  if (t < 90) {
    const float sum = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    if (part) part[(static_cast<long long>(e) * gridDim.x + blockIdx.x) * 90 + t] = sum;      // summed per edge by the Schur kernel
    else pose_block_scatter(t, static_cast<double>(sum), ix - t0, jx - t0, P, sys, meta);
  }
  BA_WG_PROBE(0, 3);
}

// ---------------------------------------------------------------------------
// schur: depth elimination for one (pixel chunk, depth frame)
// ---------------------------------------------------------------------------
// Row r of depth frame k: r == -1 is the window-pose row (Ei[k], pose kx[k]-t0),
// r >= 0 is outgoing edge eidx[eptr[k]+r] (Eij, pose jj-t0).
struct RowRef { const float* base; int pose; };

__device__ __forceinline__ RowRef row_of(int r, int k, const Plan& pl, const float* Ei, const float* Eij,
                                         const int64_t* jj, int HW, int t0, int P) {
  RowRef R;
  if (r < 0) {
    const int p = pl.kx[k] - t0;
    R.pose = (p >= 0 && p < P) ? p : -1;
    R.base = Ei + static_cast<long long>(p >= 0 && p < P ? p : 0) * 6 * HW;
  } else {
    const int e = pl.eidx[pl.eptr[k] + r];
    const int p = static_cast<int>(jj[e]) - t0;
    R.pose = (p >= 0 && p < P) ? p : -1;
    R.base = Eij + static_cast<long long>(e) * 6 * HW;
  }
  return R;
}

// ---- depth: C, w, Q and the window-pose rows Ei for one (pixel, depth frame) ----------
// (accum_cuda x3 + the Q expression of ba_cuda, droid_kernels.cu:1374-1378).  Runs as the first phase of the Schur
// kernel: every workgroup prepares Q, w and Ei for exactly the pixels it is about to reduce.
__device__ __forceinline__ void depth_pixel(const Plan& pl, int k, int x, const float* __restrict__ eta, int K_eta,
                                            const float* __restrict__ Eii, const float* __restrict__ Cii, const float* __restrict__ bz,
                                            float* __restrict__ Ei, float* __restrict__ Q, float* __restrict__ w, int HW, int t0, int P,
                                            const int* __restrict__ edges, int deg, int pself) {
  // `edges` = this depth frame's out-edges in LDS (round 4): read from the plan in global memory inside this loop, every
  // iteration was two dependent round trips (eidx[o], then the rows of edge e) - 6 x 2 of them in front of the first product
  const bool self_in = pself >= 0 && pself < P;
  float C = 0.0f, ww = 0.0f, ei[6] = {0, 0, 0, 0, 0, 0};
  // (six edges' 48 loads in flight: a frame of a real window has ~18 out-edges and this loop was a third of the kernel at unroll 2)
#pragma unroll 6
  for (int o = 0; o < deg; ++o) {
    const int e = edges[o];
    C += Cii[static_cast<long long>(e) * HW + x];
    ww += bz[static_cast<long long>(e) * HW + x];
    if (self_in) {
#pragma unroll
      for (int n = 0; n < 6; ++n) ei[n] += Eii[(static_cast<long long>(e) * 6 + n) * HW + x];
    }
  }
  // K_eta == 1 broadcasts; a row-count mismatch is flagged in meta[2] and clamped here
  const float et = eta[static_cast<long long>(k < K_eta ? k : K_eta - 1) * HW + x];
  Q[static_cast<long long>(k) * HW + x] = 1.0f / (C + et);           // droid_kernels.cu:1376
  w[static_cast<long long>(k) * HW + x] = ww;
  if (self_in) {
#pragma unroll
    for (int n = 0; n < 6; ++n) Ei[(static_cast<long long>(pself) * 6 + n) * HW + x] = ei[n];
  }
}

// For depth frame k, M stacks the 6-row blocks that couple it to window poses: Ei[k] (its own
// pose) and Eij[e] for every outgoing edge whose target pose is free, plus one extra row w_k,
// so that the rhs correction M (Q w) falls out of the same product.  One workgroup owns
// (512 pixels, k); each wave reduces its 128 pixels with v_mfma_f32_16x16x4_f32 — exact fp32
// FMA chains, K = pixels — one 16x16 tile pair at a time (row tiles re-read from L2, which
// keeps any graph degree in one code path).  Wave partials meet in LDS and leave as fp64
// atomics on the dense pose system: S is SUBTRACTED (A - S, v - E Q w; :1382).  Up to 4 row
// tiles (10 free neighbours) all tile pairs accumulate in one pass; beyond that one pair at a time.
// The reference enumerates (a,b,k) triples on the host and launches one 256-thread block per
// triple with 36 LDS tree reductions each (schur_block :1201-1290, EEt6x6 :980-1035).
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kSchurPix = 256;           // pixels per workgroup (a quarter per wave); 512: 48 workgroups at S-B, 19.9 us; 256: 96, 18.1 us
constexpr int kMaxRows = 1024;           // rows of M the LDS row table can describe
constexpr int kFastTiles = 4;            // up to 4 row tiles (63 rows + w) accumulate in one pass

// Row pointers pass through an LDS table: typed as GLOBAL-address-space pointers, so that what is loaded through them is a
// global_load (a plain `const float*` read back from LDS is a generic pointer and every access a flat_load: 112 of them in
// this kernel, each counting against both vmcnt and lgkmcnt)
typedef const float __attribute__((address_space(1))) gfloat;

// The LDS row table says, per row of M, where the row lives - a source array and a row index in it, packed into one int - and which
// entry of the pose system its sums go to (a short).  6 KB for kMaxRows rows instead of the 12 KB that pointers + ints took: with the
// 40 KB of `red` the workgroup stays under a third of the CU's 160 KB (three workgroups per CU instead of two; a frontend window
// with its inactive edges launches 550-700 of them).
constexpr int kRowEij = 0, kRowMrg = 1, kRowEi = 2, kRowW = 3;
__device__ __forceinline__ int row_code(int src, long long row) { return (src << 28) | static_cast<int>(row); }
struct RowTab {
  const int* code;           // LDS: (source << 28) | row index, -1 = padding
  const short* out;          // LDS: 6 * pose + comp for an M row, -2 for the w row, -1 padding
  const float* base[4];      // Eij, Mrg, Ei, w
  int HW;
  __device__ __forceinline__ gfloat* ptr(int r) const {
    const int c = code[r];
    if (c < 0) return nullptr;
    const int src = c >> 28;
    const float* b = src == kRowEij ? base[0] : (src == kRowMrg ? base[1] : (src == kRowEi ? base[2] : base[3]));
    return (gfloat*)(b + static_cast<long long>(c & 0x0fffffff) * HW);
  }
};

template <bool VEC4>
__device__ __forceinline__ f32x4 load4(gfloat* __restrict__ row, int p, int HW) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (row == nullptr) return v;
  if (VEC4) {
    if (p + 3 < HW) return *reinterpret_cast<const f32x4 __attribute__((address_space(1)))*>(row + p);
  }
  if (p < HW) v.x = row[p];
  if (p + 1 < HW) v.y = row[p + 1];
  if (p + 2 < HW) v.z = row[p + 2];
  if (p + 3 < HW) v.w = row[p + 3];
  return v;
}

// scatter one reduced 16x16 tile (ti,tj) of -S into the pose system
template <typename V>
__device__ __forceinline__ void scatter_tile(V v, int reg, int l, int ti, int tj, const short* rowout,
                                             long long* __restrict__ sys, int n6, int* meta) {
  // D[i][j]: i = 4*(lane>>4)+reg (row in tile ti), j = lane&15 (row in tile tj)
  const int oi = rowout[ti * 16 + 4 * (l >> 4) + reg];
  const int oj = rowout[tj * 16 + (l & 15)];
  if (oi < 0) return;
  const double val = -static_cast<double>(v);
  if (oj >= 0) {
    // lower block triangle only (see pose_block_scatter): entry (oi, oj) if its block row is not above its block column, and
    // the mirrored entry of an off-diagonal tile pair under the same rule - inside a diagonal 6 x 6 block both are kept
    const int bi = oi / 6, bj = oj / 6;
    if (bi >= bj) fix_add(sys, static_cast<long long>(oi) * n6 + oj, val, meta);
    if (ti != tj && bj >= bi) fix_add(sys, static_cast<long long>(oj) * n6 + oi, val, meta);   // mirrored tile
  } else if (oj == -2) {
    fix_add(sys, static_cast<long long>(n6) * n6 + oi, val, meta);                 // rhs: - E (Q w)
  }
}

// all T(T+1)/2 tile pairs in ONE pass over the pixels (rows read once, accumulators static)
template <int T, bool VEC4, int PIX>
__device__ __forceinline__ void schur_pass(const RowTab& rt,
                                           gfloat* __restrict__ qrow, float* red /*[4][NT*4][64]*/,
                                           long long* __restrict__ sys, int HW, int n6, int pix_base, int* meta) {
  constexpr int NT = T * (T + 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int idx = lane & 15, kq = lane >> 4;
  gfloat* rows[T];
#pragma unroll
  for (int t = 0; t < T; ++t) rows[t] = rt.ptr(t * 16 + idx);
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int s = 0; s < PIX / 64; ++s) {
    const int p = pix_base + s * 16 + 4 * kq;
    const f32x4 q = load4<VEC4>(qrow, p, HW);
    f32x4 b[T], a[T];
#pragma unroll
    for (int t = 0; t < T; ++t) { b[t] = load4<VEC4>(rows[t], p, HW); a[t] = b[t] * q; }
    int n = 0;
#pragma unroll
    for (int ti = 0; ti < T; ++ti)
#pragma unroll
      for (int tj = ti; tj < T; ++tj) {
        // K index of step c = pixel p + c of lane group kq (the same pixel for A and B)
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti].x, b[tj].x, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti].y, b[tj].y, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti].z, b[tj].z, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti].w, b[tj].w, acc[n], 0, 0, 0);
        ++n;
      }
  }
  BA_WG_PROBE(1, 5);                     // products done (wave 0)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float* r = red + (static_cast<size_t>(wave) * NT + t) * 256;
    r[lane] = acc[t].x; r[64 + lane] = acc[t].y; r[128 + lane] = acc[t].z; r[192 + lane] = acc[t].w;
  }
  __syncthreads();
  BA_WG_PROBE(1, 6);                     // all four waves' products in LDS
  int n = 0;
#pragma unroll
  for (int ti = 0; ti < T; ++ti)
#pragma unroll
    for (int tj = ti; tj < T; ++tj) {
      const float* r0 = red + static_cast<size_t>(n) * 256 + tid;
      const float v = (r0[0] + r0[static_cast<size_t>(NT) * 256]) +
                      (r0[static_cast<size_t>(2 * NT) * 256] + r0[static_cast<size_t>(3 * NT) * 256]);
      scatter_tile(v, tid >> 6, tid & 63, ti, tj, rt.out, sys, n6, meta);
      ++n;
    }
}

// Row tile ti against the CNT row tiles tj0 .. tj0 + CNT - 1 (tj0 >= ti) in one pass over the pixels: what a depth frame with more
// than kFastTiles row tiles (> 9 free neighbours: a frontend window with its inactive edges, found by the full-sequence run of
// bench.py - the replayed S-B / S-A windows have no inactive edges and never came here) used to do one tile PAIR at a time, two
// workgroup barriers per pair: 15 to 36 rounds, 118 us per launch.  Per pair the SAME chain of MFMAs in the same order and the
// same (w0 + w1) + (w2 + w3) reduction as every other path: bit-identical sums.
template <int CNT, bool VEC4, int PIX>
__device__ __forceinline__ void schur_rowpass(const RowTab& rt, gfloat* __restrict__ qrow, float* red,
                                              long long* __restrict__ sys, int HW, int n6, int pix_base, int* meta, int ti, int tj0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int idx = lane & 15, kq = lane >> 4;
  gfloat* __restrict__ ra = rt.ptr(ti * 16 + idx);
  gfloat* rb[CNT];
#pragma unroll
  for (int t = 0; t < CNT; ++t) rb[t] = rt.ptr((tj0 + t) * 16 + idx);
  f32x4 acc[CNT];
#pragma unroll
  for (int t = 0; t < CNT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int s = 0; s < PIX / 64; ++s) {
    const int p = pix_base + s * 16 + 4 * kq;
    const f32x4 q = load4<VEC4>(qrow, p, HW);
    const f32x4 a0 = load4<VEC4>(ra, p, HW);
    const f32x4 a = a0 * q;
    f32x4 b[CNT];
#pragma unroll
    for (int t = 0; t < CNT; ++t) b[t] = load4<VEC4>(rb[t], p, HW);
#pragma unroll
    for (int t = 0; t < CNT; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[t].x, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[t].y, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[t].z, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[t].w, acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < CNT; ++t) {
    float* r = red + (static_cast<size_t>(wave) * CNT + t) * 256;
    r[lane] = acc[t].x; r[64 + lane] = acc[t].y; r[128 + lane] = acc[t].z; r[192 + lane] = acc[t].w;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < CNT; ++t) {
    const float* r0 = red + static_cast<size_t>(t) * 256 + tid;
    const float v = (r0[0] + r0[static_cast<size_t>(CNT) * 256]) + (r0[static_cast<size_t>(2 * CNT) * 256] + r0[static_cast<size_t>(3 * CNT) * 256]);
    scatter_tile(v, tid >> 6, tid & 63, ti, tj0 + t, rt.out, sys, n6, meta);
  }
  __syncthreads();                        // `red` is rewritten by the next pass
}

// schur_rowpass with the depth frame's rows STAGED IN LDS (round 6: dense frontend windows).  Streaming from global memory, every row
// tile is read once per row pass it takes part in: 26 frames x 6 chunks with 7 row tiles re-read 1.1 MB of a (frame, chunk)'s 211 KB
// of rows - 170 MB per launch, the whole of its 75 us (tools/ba_kernel_timeline.py at NF=26 RAD=8 HT=30 WD=101: tile pairs 38-52 us
// per workgroup against ~3 us of MFMA).  Staged once per (frame, 256-pixel chunk) the pairs read LDS.  Same operands, same chain of
// MFMAs in the same order, same (w0 + w1) + (w2 + w3) reduction: bit-identical to the streaming form at the same chunk size.
constexpr int kLdsRowPad = 4;            // floats between rows: 16-byte alignment of every row, rows 8 apart share a bank group (2-way)
template <int CNT, int PIX>
__device__ __forceinline__ void schur_rowpass_lds(const float* L, const float* Lq, int nrows, const short* rowout, float* red,
                                                  long long* __restrict__ sys, int n6, int* meta, int ti, int tj0,
                                                  float* __restrict__ part_out, int pair0) {
  constexpr int pitch = PIX + kLdsRowPad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int idx = lane & 15, kq = lane >> 4;
  const int ra_row = ti * 16 + idx;
  const float* ra = L + static_cast<size_t>(ra_row < nrows ? ra_row : 0) * pitch;
  const bool oka = ra_row < nrows;
  const float* rb[CNT];
  bool okb[CNT];
#pragma unroll
  for (int t = 0; t < CNT; ++t) {
    const int r = (tj0 + t) * 16 + idx;
    okb[t] = r < nrows;
    rb[t] = L + static_cast<size_t>(okb[t] ? r : 0) * pitch;
  }
  f32x4 acc[CNT];
#pragma unroll
  for (int t = 0; t < CNT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int s = 0; s < PIX / 64; ++s) {
    const int p = wave * (PIX / 4) + s * 16 + 4 * kq;                    // (inside the chunk)
    const f32x4 q = *reinterpret_cast<const f32x4*>(Lq + p);
    const f32x4 a0 = oka ? *reinterpret_cast<const f32x4*>(ra + p) : zero;
    const f32x4 a = a0 * q;
    f32x4 b[CNT];
#pragma unroll
    for (int t = 0; t < CNT; ++t) b[t] = okb[t] ? *reinterpret_cast<const f32x4*>(rb[t] + p) : zero;
#pragma unroll
    for (int t = 0; t < CNT; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[t].x, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[t].y, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[t].z, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[t].w, acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < CNT; ++t) {
    float* r = red + (static_cast<size_t>(wave) * CNT + t) * 256;
    r[lane] = acc[t].x; r[64 + lane] = acc[t].y; r[128 + lane] = acc[t].z; r[192 + lane] = acc[t].w;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < CNT; ++t) {
    const float* r0 = red + static_cast<size_t>(t) * 256 + tid;
    const float v = (r0[0] + r0[static_cast<size_t>(CNT) * 256]) + (r0[static_cast<size_t>(2 * CNT) * 256] + r0[static_cast<size_t>(3 * CNT) * 256]);
    // two-stage sums (dense windows): the chunk's tile-pair sum goes to the workspace, ba_schur_reduce_kernel adds a frame's chunks and
    // issues ONE fixed-point atomic per entry and frame - with one per chunk the 3.3 M memory-side atomics of a window's launch were
    // half of its 93 us (tools/ba_kernel_timeline.py with PROBE_DEFS=-DPVO_BA_NO_ATOMICS)
    if (part_out) part_out[static_cast<size_t>(pair0 + t) * 256 + tid] = v;
    else scatter_tile(v, tid >> 6, tid & 63, ti, tj0 + t, rowout, sys, n6, meta);
  }
  __syncthreads();                        // `red` is rewritten by the next pass
}

template <bool VEC4, int PIX>
__device__ __forceinline__ void ba_schur_body(
    const Plan& pl, const int64_t* __restrict__ jj, const float* __restrict__ eta, int K_eta,
    const float* __restrict__ Eii, const float* __restrict__ Cii, const float* __restrict__ bz,
    float* __restrict__ Ei, const float* __restrict__ Eij,
    float* __restrict__ Q, float* __restrict__ w, long long* __restrict__ sys,
    int HW, int t0, int P, const float* __restrict__ part, const int64_t* __restrict__ ii, int E, int chunksA, int deal_rows,
    float* __restrict__ Mrg, int depth_done, int lds_floats, float* __restrict__ spart, short* __restrict__ srow, int* __restrict__ sT) {
  __shared__ int rowcode[kMaxRows];       // see RowTab
  __shared__ short rowout[kMaxRows];
  __shared__ int nrows_s;
  __shared__ float red[4 * (kFastTiles * (kFastTiles + 1) / 2) * 256];   // 40 KB
  // the assembly's chunk sums, per edge, are added up and scattered by workgroups of their OWN: the last `deal_rows` rows of the
  // grid (at most two edges each).  Dealt over the depth frames' workgroups, as in round 3, the 36 of 96 that had an edge at S-B
  // started their own work 3.5 us late - and the kernel ends with its slowest workgroup (profiles/r04_ba_kernel_timeline.txt).
  // gridDim.z slices: the row-tile passes of a depth frame with MANY neighbours (more than kFastTiles row tiles: a frontend window
  // with its inactive edges) are dealt over the slices, ti = z, z + gridDim.z, ...; everything else is slice 0's and the other
  // slices leave at once.  Every slice that stays runs the depth phase itself (identical values to identical addresses - the
  // rows a slice reads are the ones it wrote) and every tile pair is still summed by ONE workgroup in the fixed order.
  const int zi = blockIdx.z, Z = gridDim.z;
  if (static_cast<int>(blockIdx.y) >= static_cast<int>(gridDim.y) - deal_rows) {
    if (zi != 0) return;
    if (part && threadIdx.x < 90) {
      // kDealEdges edges per workgroup, CONSECUTIVE in the plan's by-source order: an edge's (ii, ii) block and vi entries land on
      // the same addresses as its neighbours' (same source frame), so their fixed-point addends are summed in a register and
      // leave as one atomic when the address changes - integer sums: the system is bit for bit what one atomic per edge gave.
      // (A real window's 342 edges were 30 780 memory-side atomics, up to ~33 deep on a diagonal entry: half of the launch's
      // 105 us, bench.py `sequence`; two edges a workgroup apart in the edge list shared nothing.)
      const int t = threadIdx.x;
      const int wgi = (blockIdx.y - (gridDim.y - deal_rows)) * gridDim.x + blockIdx.x;
      // (the plan lists only edges whose source frame lies in [0, F): eidx[eptr[K] ..) is not initialised - bound by the plan's count)
      const int Epl = pl.eptr[pl.meta[0]];
      const int p0 = wgi * kDealEdges, p1 = (p0 + kDealEdges < Epl) ? p0 + kDealEdges : Epl;
      long long pend_idx[2] = {-1, -1}, pend_acc[2] = {0, 0};
      for (int pos = p0; pos < p1; ++pos) {
        const int e = pl.eidx[pos];
        double val = 0.0;
        for (int c = 0; c < chunksA; ++c) val += static_cast<double>(part[(static_cast<long long>(e) * chunksA + c) * 90 + t]);
        if (!(fabs(val) < 3.0e10)) { pl.meta[4] = 1; continue; }       // (fix_add's range check)
        const long long q = __double2ll_rn(val * kFix);
        long long idx[2];
        pose_block_targets(t, static_cast<int>(ii[e]) - t0, static_cast<int>(jj[e]) - t0, P, idx);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          if (idx[sl] < 0) continue;
          if (idx[sl] == pend_idx[sl]) { pend_acc[sl] += q; continue; }
          if (pend_idx[sl] >= 0) atomicAdd(reinterpret_cast<unsigned long long*>(sys + pend_idx[sl]), static_cast<unsigned long long>(pend_acc[sl]));
          pend_idx[sl] = idx[sl]; pend_acc[sl] = q;
        }
      }
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
        if (pend_idx[sl] >= 0) atomicAdd(reinterpret_cast<unsigned long long*>(sys + pend_idx[sl]), static_cast<unsigned long long>(pend_acc[sl]));
    }
    return;
  }
  const int k = blockIdx.y;
  BA_WG_PROBE(1, 0);
  if (k >= pl.meta[0]) return;
  BA_WG_PROBE(1, 1);                     // chunk sums of the assembly added, meta read
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n6 = 6 * P;
  // this depth frame's out-edges and their target poses, fetched ONCE and in parallel into LDS: eptr -> eidx -> jj is a chain
  // of three dependent loads that the depth phase and the row table used to walk per edge (12 + 12 serial round trips at S-B)
  __shared__ int s_edge[256], s_pose[256];
  __shared__ int s_lead[64], s_merged[64], any_merged_s;      // (the ballot path below: at most 64 out-edges)
  const int e0 = pl.eptr[k], deg_all = pl.eptr[k + 1] - e0;
  const int pself = pl.kx[k] - t0;
  const bool in_lds = deg_all <= 256;
  if (zi > 0 && 6 * deg_all + 7 <= 16 * kFastTiles) return;      // at most kFastTiles row tiles whatever the poses: slice 0 alone
  if (in_lds && tid < deg_all) {
    const int e = pl.eidx[e0 + tid];
    s_edge[tid] = e;
    s_pose[tid] = static_cast<int>(jj[e]) - t0;
  }
  __syncthreads();
  BA_WG_PROBE(1, 2);                     // edge list in LDS
  // The depth phase needs the edge list only.  Slice 0 issues it BEFORE wave 0 builds the row table, so the table's ~1 us of LDS
  // work hides behind the depth phase's loads (as up to round 4); ONE barrier behind both publishes table and rows together.
  // (S-B's launch: 18.2 us with the table in front and a 64-step follower scan, 17.3 now, 16.2 with round 4's kernel -
  // profiles/r05_schur_sweep.txt, block 6.)
  auto depth_phase = [&]() {
#pragma unroll
    for (int h = 0; h < PIX / 256; ++h) {
      const int x = blockIdx.x * PIX + h * 256 + tid;
      if (x < HW) depth_pixel(pl, k, x, eta, K_eta, Eii, Cii, bz, Ei, Q, w, HW, t0, P, in_lds ? s_edge : pl.eidx + e0, deg_all, pself);
    }
  };
  // (every slice that gets here may have to stay - more than kFastTiles row tiles unless targets merge or are fixed - and then needs
  // the rows: it issues the depth phase in front of the table like slice 0 instead of behind it; a slice that leaves after all has
  // written the same values to the same addresses as slice 0.)
  // depth_done: ba_depth_kernel ran in front of this launch (windows on 512-pixel chunks, round 6): at a real frontend window every
  // one of the four slices re-read the frame's 16-18 edges x 8 rows here - 164 MB per launch instead of 41, and 20 us at the head of
  // every workgroup (tools/ba_kernel_timeline.py, NF=26 RAD=8 HT=30 WD=101)
  if (!depth_done) depth_phase();
  if (in_lds && deg_all <= 64) {
    // row table by wave 0, one lane per out-edge: the position of an edge's six rows = the number of free target poses before it
    // (ballot + popcount) - the same order as the sequential walk below, which took 1.9 us of this kernel at S-B
    // Round 5: edges of this frame that share their TARGET pose (a frontend window keeps every aged-out edge as an inactive one,
    // factor_graph.py:281-289: the same frame pair several times - 18 out-edges to ~7 distinct poses) contribute row blocks that
    // only ever meet as their SUM: S gets (sum_e E_e) Q (sum_e E_e)^T for that pose.  The first edge of each target (the "leader",
    // in edge order) stands for the group; a leader with followers reads the group's summed rows from `Mrg` (written below by
    // this workgroup for its own pixels, in edge order), one without reads its own Eij rows as before.  Rows 6 x 18 + 7 -> 6 x 7 + 7:
    // four row tiles, the one-pass path.  Frames without repeated targets take exactly the path they took (bit-identical).
    if (tid < 64) {
      const bool free_pose = tid < deg_all && s_pose[tid] >= 0 && s_pose[tid] < P;      // fixed target pose: drops out (:1125, :1227)
      int lead = tid;
      if (free_pose) {
        const int p = s_pose[tid];
        for (int u = 0; u < tid; ++u) if (s_pose[u] == p) { lead = u; break; }
      }
      s_lead[tid] = free_pose ? lead : -1;
      const bool is_lead = free_pose && lead == tid;
      const unsigned long long mask = __ballot(is_lead);
      // followers per leader: a leader has some iff another lane names it
      const int named = (free_pose && lead != tid) ? lead : -1;
      unsigned long long has_f = 0ull;
      for (int u = 1; u < deg_all; ++u) {          // (lane 0 leads itself; uniform bound: deg_all is the workgroup's)
        const int lu = __shfl(named, u);
        if (lu >= 0) has_f |= 1ull << lu;
      }
      const bool self = pself >= 0 && pself < P;
      const int base = self ? 6 : 0;
      const bool merged = is_lead && ((has_f >> tid) & 1ull);
      s_merged[tid] = merged ? 1 : 0;
      if (is_lead) {
        const int r = base + 6 * __popcll(mask & ((1ull << tid) - 1ull));
        const int e = s_edge[tid], p = s_pose[tid];
        const int src = merged ? kRowMrg : kRowEij;
#pragma unroll
        for (int n = 0; n < 6; ++n) { rowcode[r + n] = row_code(src, static_cast<long long>(e) * 6 + n); rowout[r + n] = static_cast<short>(6 * p + n); }
      }
      if (tid == 0) {
        if (self) {
#pragma unroll
          for (int n = 0; n < 6; ++n) { rowcode[n] = row_code(kRowEi, static_cast<long long>(pself) * 6 + n); rowout[n] = static_cast<short>(6 * pself + n); }
        }
        int r = base + 6 * __popcll(mask);
        if (r > 0) { rowcode[r] = row_code(kRowW, k); rowout[r] = -2; ++r; }
        const int padded = (r + 15) & ~15;
        for (int q = r; q < padded; ++q) { rowcode[q] = -1; rowout[q] = -1; }
        nrows_s = r;
        any_merged_s = has_f != 0ull ? 1 : 0;
      }
    }
  } else if (tid == 0) {                  // any degree: sequential
    any_merged_s = 0;
    int r = 0;
    if (pself >= 0 && pself < P) {
      for (int n = 0; n < 6; ++n) { rowcode[r] = row_code(kRowEi, static_cast<long long>(pself) * 6 + n); rowout[r] = static_cast<short>(6 * pself + n); ++r; }
    }
    for (int o = 0; o < deg_all; ++o) {
      const int e = in_lds ? s_edge[o] : pl.eidx[e0 + o];
      const int p = in_lds ? s_pose[o] : static_cast<int>(jj[e]) - t0;
      if (p < 0 || p >= P) continue;      // fixed target pose: drops out (:1125, :1227)
      if (r + 7 > kMaxRows) { pl.meta[3] = 1; break; }   // > 169 free neighbours of one frame: flagged
      for (int n = 0; n < 6; ++n) { rowcode[r] = row_code(kRowEij, static_cast<long long>(e) * 6 + n); rowout[r] = static_cast<short>(6 * p + n); ++r; }
    }
    if (r > 0) { rowcode[r] = row_code(kRowW, k); rowout[r] = -2; ++r; }
    const int padded = (r + 15) & ~15;
    for (int q = r; q < padded; ++q) { rowcode[q] = -1; rowout[q] = -1; }
    nrows_s = r;
  }
  // Slice 0 has issued its depth phase in front of the table (above); the other slices need the tile count first - most of them
  // leave here - and run the depth phase now.
  if (zi > 0) {
    __syncthreads();
    if (((nrows_s + 15) >> 4) <= kFastTiles) return;               // (uniform: nrows_s is the workgroup's)
  }
  BA_WG_PROBE(1, 3);                     // row table built, depth phase issued
  __syncthreads();
  const int nrows = nrows_s;
  const int T = (nrows + 15) >> 4;
  if (in_lds && deg_all <= 64 && any_merged_s) {        // the summed rows of every target group with more than one edge, this workgroup's pixels
#pragma unroll
    for (int h = 0; h < PIX / 256; ++h) {
      const int x = blockIdx.x * PIX + h * 256 + tid;
      if (x >= HW) continue;
      for (int L = 0; L < deg_all; ++L) {
        if (!s_merged[L]) continue;                       // (uniform over the workgroup)
        float m[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int t = L; t < deg_all; ++t) {
          if (s_lead[t] != L) continue;
          const int e = s_edge[t];
#pragma unroll
          for (int n = 0; n < 6; ++n) m[n] += Eij[(static_cast<long long>(e) * 6 + n) * HW + x];
        }
        const int el = s_edge[L];
#pragma unroll
        for (int n = 0; n < 6; ++n) Mrg[(static_cast<long long>(el) * 6 + n) * HW + x] = m[n];
      }
    }
    __syncthreads();                      // (uniform: any_merged_s is the workgroup's)
  }
  BA_WG_PROBE(1, 4);                     // depth rows (and merged rows) stored
  const bool lds_path = lds_floats > 0 && T > kFastTiles && static_cast<long long>(nrows) * (PIX + kLdsRowPad) + PIX <= lds_floats;
  const bool staged = lds_path && spart != nullptr && T * (T + 1) / 2 <= kSchurStagePairs;
  if (sT != nullptr && blockIdx.x == 0 && zi == 0) {          // what ba_schur_reduce_kernel needs to know about this frame
    if (tid == 0) sT[k] = staged ? T : 0;
    if (staged && tid < 16 * T) srow[k * 128 + tid] = rowout[tid];
  }
  if (nrows == 0) return;

  gfloat* __restrict__ qrow = (gfloat*)(Q + static_cast<long long>(k) * HW);
  const RowTab rt = {rowcode, rowout, {Eij, Mrg, Ei, w}, HW};
  const int pix_base = blockIdx.x * PIX + wave * (PIX / 4);

  switch (T) {
    case 1: schur_pass<1, VEC4, PIX>(rt, qrow, red, sys, HW, n6, pix_base, pl.meta); BA_WG_PROBE(1, 7); return;
    case 2: schur_pass<2, VEC4, PIX>(rt, qrow, red, sys, HW, n6, pix_base, pl.meta); BA_WG_PROBE(1, 7); return;
    case 3: schur_pass<3, VEC4, PIX>(rt, qrow, red, sys, HW, n6, pix_base, pl.meta); BA_WG_PROBE(1, 7); return;
    case 4: schur_pass<4, VEC4, PIX>(rt, qrow, red, sys, HW, n6, pix_base, pl.meta); BA_WG_PROBE(1, 7); return;
    default: break;
  }
  // any degree: one ROW TILE against up to eight others per pass (row tiles re-read from L2 once per pass)
  constexpr int kPassTiles = 8;           // 8 x 4 x 256 floats of `red` = 32 KB
  if (lds_path) {
    // (round 6) the frame's rows of this chunk + Q, once into LDS (dynamic segment; the host asks for it only on 256-pixel chunks of
    // a dense window and launches ONE slice then); a frame whose rows do not fit streams them as before, a few lines down
    extern __shared__ __attribute__((aligned(16))) float dyn_rows[];
    constexpr int pitch = PIX + kLdsRowPad;
    float* Lq = dyn_rows + static_cast<size_t>(nrows) * pitch;
    const int x0 = blockIdx.x * PIX;
    // (eight rows' loads in flight per wave before the first store: one row at a time the staging was 26 serial round trips to HBM
    // per wave - 50 of the workgroup's 60 us)
    static_assert(PIX == 256 || PIX == 512 || PIX == 1024, "chunk");
    for (int r0 = wave; r0 < nrows; r0 += 32) {
      f32x4 v[8][PIX / 256];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + 4 * u;
        gfloat* src = rt.ptr(r < nrows ? r : r0);
#pragma unroll
        for (int c = 0; c < PIX / 256; ++c) v[u][c] = load4<VEC4>(src, x0 + c * 256 + lane * 4, HW);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + 4 * u;
        if (r < nrows) {
#pragma unroll
          for (int c = 0; c < PIX / 256; ++c)
            *reinterpret_cast<f32x4*>(dyn_rows + static_cast<size_t>(r) * pitch + c * 256 + lane * 4) = v[u][c];
        }
      }
    }
    if (wave == 0) {
#pragma unroll
      for (int c = 0; c < PIX / 256; ++c) {
        const int px = c * 256 + lane * 4;
        *reinterpret_cast<f32x4*>(Lq + px) = load4<VEC4>(qrow, x0 + px, HW);
      }
    }
    __syncthreads();
    BA_WG_PROBE(1, 5);                   // (LDS form: rows staged)
    float* part_out = staged ? spart + (static_cast<size_t>(k) * gridDim.x + blockIdx.x) * kSchurStagePairs * 256 : nullptr;
    for (int ti = 0; ti < T; ++ti) {
      if (ti == 1) BA_WG_PROBE(1, 6);    // (LDS form: first row pass - the longest: T tile pairs - done)
      const int ph = ti % (2 * Z);
      if ((ph < Z ? ph : 2 * Z - 1 - ph) != zi) continue;
      for (int tj0 = ti; tj0 < T; tj0 += kPassTiles) {
        const int cnt = (T - tj0 < kPassTiles) ? T - tj0 : kPassTiles;
        const int pair0 = ti * T - ti * (ti - 1) / 2 + (tj0 - ti);      // index of pair (ti, tj0) among the T (T + 1) / 2
        switch (cnt) {
          case 1: schur_rowpass_lds<1, PIX>(dyn_rows, Lq, nrows, rowout, red, sys, n6, pl.meta, ti, tj0, part_out, pair0); break;
          case 2: schur_rowpass_lds<2, PIX>(dyn_rows, Lq, nrows, rowout, red, sys, n6, pl.meta, ti, tj0, part_out, pair0); break;
          case 3: schur_rowpass_lds<3, PIX>(dyn_rows, Lq, nrows, rowout, red, sys, n6, pl.meta, ti, tj0, part_out, pair0); break;
          case 4: schur_rowpass_lds<4, PIX>(dyn_rows, Lq, nrows, rowout, red, sys, n6, pl.meta, ti, tj0, part_out, pair0); break;
          case 5: schur_rowpass_lds<5, PIX>(dyn_rows, Lq, nrows, rowout, red, sys, n6, pl.meta, ti, tj0, part_out, pair0); break;
          case 6: schur_rowpass_lds<6, PIX>(dyn_rows, Lq, nrows, rowout, red, sys, n6, pl.meta, ti, tj0, part_out, pair0); break;
          case 7: schur_rowpass_lds<7, PIX>(dyn_rows, Lq, nrows, rowout, red, sys, n6, pl.meta, ti, tj0, part_out, pair0); break;
          default: schur_rowpass_lds<8, PIX>(dyn_rows, Lq, nrows, rowout, red, sys, n6, pl.meta, ti, tj0, part_out, pair0); break;
        }
      }
    }
    BA_WG_PROBE(1, 7);
    return;
  }
  // row tile ti costs T - ti tile pairs: the slices take them in a SNAKE (0 1 2 3 3 2 1 0 0 1 ...), so that every slice gets the same
  // number of pairs (T = 7, four slices: 7 | 6 + 1 | 5 + 2 | 4 + 3; dealt ti = z, z + 4, .. slice 0 had 10 and slice 3 had 4, and the
  // launch ends with its slowest slice).  Every pair is still one workgroup's chain in the same order: bit-identical.
  for (int ti = 0; ti < T; ++ti) {
    const int ph = ti % (2 * Z);
    if ((ph < Z ? ph : 2 * Z - 1 - ph) != zi) continue;
    for (int tj0 = ti; tj0 < T; tj0 += kPassTiles) {
      const int cnt = (T - tj0 < kPassTiles) ? T - tj0 : kPassTiles;
      switch (cnt) {
        case 1: schur_rowpass<1, VEC4, PIX>(rt, qrow, red, sys, HW, n6, pix_base, pl.meta, ti, tj0); break;
        case 2: schur_rowpass<2, VEC4, PIX>(rt, qrow, red, sys, HW, n6, pix_base, pl.meta, ti, tj0); break;
        case 3: schur_rowpass<3, VEC4, PIX>(rt, qrow, red, sys, HW, n6, pix_base, pl.meta, ti, tj0); break;
        case 4: schur_rowpass<4, VEC4, PIX>(rt, qrow, red, sys, HW, n6, pix_base, pl.meta, ti, tj0); break;
        case 5: schur_rowpass<5, VEC4, PIX>(rt, qrow, red, sys, HW, n6, pix_base, pl.meta, ti, tj0); break;
        case 6: schur_rowpass<6, VEC4, PIX>(rt, qrow, red, sys, HW, n6, pix_base, pl.meta, ti, tj0); break;
        case 7: schur_rowpass<7, VEC4, PIX>(rt, qrow, red, sys, HW, n6, pix_base, pl.meta, ti, tj0); break;
        default: schur_rowpass<8, VEC4, PIX>(rt, qrow, red, sys, HW, n6, pix_base, pl.meta, ti, tj0); break;
      }
    }
  }
  BA_WG_PROBE(1, 7);
}

template <bool VEC4, int PIX>
__global__ __launch_bounds__(256) void ba_schur_mfma_kernel(
    Plan pl, const int64_t* __restrict__ jj, const float* __restrict__ eta, int K_eta,
    const float* __restrict__ Eii, const float* __restrict__ Cii, const float* __restrict__ bz,
    float* __restrict__ Ei, const float* __restrict__ Eij,
    float* __restrict__ Q, float* __restrict__ w, long long* __restrict__ sys,
    int HW, int t0, int P, const float* __restrict__ part, const int64_t* __restrict__ ii, int E, int chunksA, int deal_rows,
    float* __restrict__ Mrg, int depth_done, int lds_floats, float* __restrict__ spart, short* __restrict__ srow, int* __restrict__ sT) {
  ba_schur_body<VEC4, PIX>(pl, jj, eta, K_eta, Eii, Cii, bz, Ei, Eij, Q, w, sys, HW, t0, P, part, ii, E, chunksA, deal_rows, Mrg, depth_done, lds_floats,
                           spart, srow, sT);
}

// second stage of the dense-window Schur sums: tile pair p of depth frame k = the sum over the frame's chunks (fixed order, fp64: the
// addends are fp32) -> one fixed-point atomic per entry.  Integer sums across frames as before: bitwise reproducible, and an
// edge-sharded run gets the whole graph's bits (a frame's edges, hence its chunks, live on one rank).
__global__ __launch_bounds__(256) void ba_schur_reduce_kernel(Plan pl, const float* __restrict__ spart, const short* __restrict__ srow,
                                                              const int* __restrict__ sT, long long* __restrict__ sys, int gx, int n6) {
  const int k = blockIdx.y, p = blockIdx.x, tid = threadIdx.x;
  if (k >= pl.meta[0]) return;
  const int T = sT[k];
  if (T == 0 || p >= T * (T + 1) / 2) return;
  __shared__ short rowout[128];
  if (tid < 128) rowout[tid] = tid < 16 * T ? srow[k * 128 + tid] : static_cast<short>(-1);
  __syncthreads();
  int ti = 0, base = 0;
  while (base + (T - ti) <= p) { base += T - ti; ++ti; }
  const int tj = ti + (p - base);
  double v = 0.0;
  const float* sp = spart + (static_cast<size_t>(k) * gx * kSchurStagePairs + p) * 256 + tid;
  const size_t step = static_cast<size_t>(kSchurStagePairs) * 256;
  for (int x0 = 0; x0 < gx; x0 += 8) {                     // (eight chunks' loads in flight; the additions keep the chunk order)
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = sp[static_cast<size_t>(x0 + u < gx ? x0 + u : x0) * step];
#pragma unroll
    for (int u = 0; u < 8; ++u) if (x0 + u < gx) v += static_cast<double>(a[u]);
  }
  scatter_tile(v, tid >> 6, tid & 63, ti, tj, rowout, sys, n6, pl.meta);
}

// the depth phase of the Schur kernel as a launch of its own (one thread per pixel and depth frame): C, w, Q and the frame's own pose
// rows Ei, in exactly the order depth_pixel states - the values are bit for bit what the fused form writes.  Used in front of the
// 512- / 1024-pixel Schur grids (windows beyond ~15 poses), where the Schur kernel's z-slices would each repeat it.
__global__ __launch_bounds__(256) void ba_depth_kernel(
    Plan pl, const float* __restrict__ eta, int K_eta, const float* __restrict__ Eii, const float* __restrict__ Cii,
    const float* __restrict__ bz, float* __restrict__ Ei, float* __restrict__ Q, float* __restrict__ w, int HW, int t0, int P) {
  const int k = blockIdx.y;
  if (k >= pl.meta[0]) return;
  __shared__ int s_edge[256];
  const int tid = threadIdx.x;
  const int e0 = pl.eptr[k], deg_all = pl.eptr[k + 1] - e0;
  const int pself = pl.kx[k] - t0;
  const bool in_lds = deg_all <= 256;
  if (in_lds && tid < deg_all) s_edge[tid] = pl.eidx[e0 + tid];
  __syncthreads();
  const int x = blockIdx.x * 256 + tid;
  if (x < HW) depth_pixel(pl, k, x, eta, K_eta, Eii, Cii, bz, Ei, Q, w, HW, t0, P, in_lds ? s_edge : pl.eidx + e0, deg_all, pself);
}

// ---------------------------------------------------------------------------
// solve: damping + fp64 Cholesky + substitution + pose retraction, one workgroup
// ---------------------------------------------------------------------------
// Blocked LL^T (6x6 blocks: the natural granularity of the pose system) of the SPD matrix in A, rhs carried as
// row n of A (so the forward substitution is just one more row of every panel / trailing update).
//   per block column kb:  (a) every thread factors the 6x6 diagonal block redundantly in registers (21 broadcast LDS reads,
//                             no communication);  (b) one thread per remaining row solves its 1x6 panel in place;
//                         barrier;  (c) rank-6 trailing update over a 16x16 thread grid;  barrier.
// 2 barriers per 6 columns (the column-at-a-time version needed 6, and a serial fp64 sqrt/div in front of each).
// Ld[kb][27] keeps the factored diagonal blocks (row-major lower, 21) and their reciprocal diagonals (6) for the back substitution.
// 1/sqrt(d) in fp64: hardware estimate (v_rsq_f64) + two Newton steps; the library sqrt and divide are ~30-instruction
// software sequences each and sit on the serial critical path of the factorisation
// tools/ba_solve_timeline.py builds this file with -DPVO_BA_PROBE: clock stamps of the solve kernel's phases
#ifdef PVO_BA_PROBE
__device__ unsigned long long* g_ba_probe = nullptr;
#define BA_PROBE(slot) do { if (g_ba_probe && threadIdx.x == 0) g_ba_probe[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define BA_PROBE(slot)
#endif
// ... and (-DPVO_BA_PROBE=2) of one block column (P / 2) of the pipelined factorisation's wave 0, slots 8..: every stamp is a
// scalar load + s_waitcnt lgkmcnt(0), which drains the LDS queue - the split it shows is of a step ~30 % longer than the real one
#if defined(PVO_BA_PROBE) && PVO_BA_PROBE == 2
#define BA_PROBE_STEP(slot) do { if (g_ba_probe && lane == 0 && kb == P / 2) g_ba_probe[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define BA_PROBE_STEP(slot)
#endif

__device__ __forceinline__ double rsqrt_nr(double d) {
  double y = __builtin_amdgcn_rsq(d);
  y = y * (1.5 - 0.5 * d * y * y);
  y = y * (1.5 - 0.5 * d * y * y);
  return y;
}

__device__ __forceinline__ bool chol6(const double D[21], double L[21], double rdiag[6]) {
  // index (r,c), r >= c:  r*(r+1)/2 + c
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = D[j * (j + 1) / 2 + j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[j * (j + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
    ok = ok && (d > 0.0);
    const double rs = ok ? rsqrt_nr(d) : 0.0;
    rdiag[j] = rs;
    L[j * (j + 1) / 2 + j] = d * rs;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double v = D[i * (i + 1) / 2 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
      L[i * (i + 1) / 2 + j] = v * rs;
    }
  }
  return ok;
}

// ENVELOPE (skyline) form: first[ib] = first block column of block row ib that holds a non-zero.  Cholesky creates no
// fill outside the envelope, so every panel / trailing-update / back-substitution term that lies outside it is an exact
// zero and is skipped - same bits as the dense factorisation, O(P * band^2) block operations instead of O(P^3).  The
// reduced pose system of a keyframe graph is block-banded up to its loop closures (two poses couple only through a depth
// frame both observe): at 63 free poses and temporal radius 3 the envelope holds 12 % of the matrix.  (The reference
// uses a sparse LLT for the same reason, droid_kernels.cu:1178-1184.)
// The matrix behind an accessor A(i, c): (n+1) x n, row n = rhs.  Two storages:
//   DenseMat  row-major, in LDS (up to 22 free poses) or in global memory (anything, slow: every access an L2 round trip);
//   EnvMat    only the blocks inside the envelope, block row b = blocks first[b] .. b of 36 doubles each at rowbase[b], + the
//             rhs - what lets a 63-pose system of a radius-3 graph (envelope 12 % of the matrix: 121 KB) live in LDS.
struct MatRow {            // one row of either storage: element c at base[c + k * (c / 6)]
  double* base; int k;
  __device__ __forceinline__ double& operator()(int c) const { return base[c + k * (c / 6)]; }
};
template <typename Index>      // int for the LDS copy (32-bit address arithmetic), long long for a matrix in global memory
struct DenseMat {
  double* p; Index n;
  __device__ __forceinline__ MatRow row(int i) const { return MatRow{p + i * n, 0}; }
  __device__ __forceinline__ Index rowstep() const { return n; }      // distance between the same column of rows i and i + 1
  __device__ __forceinline__ int blockstep() const { return 6; }       // ... between the same entry of blocks (ib, cb) and (ib, cb + 1)
  // six contiguous (16-byte aligned) entries: row r of 6 x 6 block (ib, cb); the rhs entries of block cb
  __device__ __forceinline__ double* brow(int ib, int r, int cb) const { return p + static_cast<Index>(6 * ib + r) * n + 6 * cb; }
  __device__ __forceinline__ double* yrow(int cb) const { return p + n * n + 6 * cb; }
};
struct EnvMat {
  double* blk; double* rhs; const int* rowoff; int n;      // rowoff[ib] = rowbase[ib] - 36 first[ib]: block (ib, cb) at rowoff[ib] + 36 cb
  __device__ __forceinline__ MatRow row(int i) const {
    if (i == n) return MatRow{rhs, 0};
    const int ib = i / 6;                                  // 6 x 6 blocks, row-major inside
    return MatRow{blk + rowoff[ib] + (i - 6 * ib) * 6, 30};
  }
  __device__ __forceinline__ int rowstep() const { return 6; }         // ... inside one block row
  __device__ __forceinline__ int blockstep() const { return 36; }
  __device__ __forceinline__ double* brow(int ib, int r, int cb) const { return blk + rowoff[ib] + 36 * cb + 6 * r; }
  __device__ __forceinline__ double* yrow(int cb) const { return rhs + 6 * cb; }
};

// Inclusive scan (sum or max) of v[0..P) in place by the whole workgroup, 256 entries at a time with a carry: the serial
// thread-0 loops this replaces were 50 k cycles of LDS round trips per 63-pose solve.
__device__ __forceinline__ int* scan_scratch() {             // one copy for both instantiations of block_scan (LDS is scarce in the solve)
  __shared__ int scratch[2 * 256 + 1];
  return scratch;
}
template <bool MAX>
__device__ __forceinline__ void block_scan(int* v, int P) {
  int (*buf)[256] = reinterpret_cast<int (*)[256]>(scan_scratch());
  int& carry_s = scan_scratch()[512];
  const int tid = threadIdx.x;
  if (tid == 0) carry_s = MAX ? -0x7fffffff : 0;
  __syncthreads();
  for (int base = 0; base < P; base += 256) {
    const int i = base + tid;
    int x = i < P ? v[i] : (MAX ? -0x7fffffff : 0);
    int cur = 0;
    buf[0][tid] = x;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      int y = buf[cur][tid];
      if (tid >= d) { const int z = buf[cur][tid - d]; y = MAX ? (z > y ? z : y) : y + z; }
      buf[cur ^ 1][tid] = y;
      cur ^= 1;
      __syncthreads();
    }
    const int c = carry_s;
    const int r = MAX ? (c > buf[cur][tid] ? c : buf[cur][tid]) : c + buf[cur][tid];
    if (i < P) v[i] = r;
    __syncthreads();
    if (tid == 255) carry_s = r;
    __syncthreads();
  }
}

// reach[kb] = last block row whose envelope reaches block column kb: the rows below it (except the rhs) take no part in
// step kb, and scanning them - 23 loop trips per thread and step at 63 poses, each an LDS lookup to find out - cost more than
// the arithmetic (tools/ba_solve_timeline.py: 14 k cycles per block column against 4.5 k at 7 poses)
__device__ __forceinline__ void envelope_reach(const int* first, int* reach, int P) {
  if (P <= 12) {                                           // (nothing to skip in a window-sized system)
    for (int b = threadIdx.x; b < P; b += blockDim.x) reach[b] = P - 1;
    __syncthreads();
    return;
  }
  for (int b = threadIdx.x; b < P; b += blockDim.x) reach[b] = b;
  __syncthreads();
  for (int b = threadIdx.x; b < P; b += blockDim.x) atomicMax(&reach[first[b]], b);
  __syncthreads();
  block_scan<true>(reach, P);
}

template <class Mat>
__device__ __forceinline__ void chol_solve_blocked(Mat A, double* Ld, int n, int* fail_flag, const int* first, const int* reach) {
  // On return row n holds the solution x.
  const int tid = threadIdx.x, nt = blockDim.x;
  const int tx = tid & 15, ty = tid >> 4, nty = nt >> 4;
  const int P = n / 6;
  const MatRow Y = A.row(n);
  for (int kb = 0; kb < P; ++kb) {
    const int j0 = 6 * kb;
    const int iend = 6 * reach[kb] + 5;                    // last matrix row active in this step
    double D[21], L[21], rd[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double* rp = &A.row(j0 + r)(j0);               // (six entries of one block row are contiguous in either storage)
#pragma unroll
      for (int c = 0; c <= r; ++c) D[r * (r + 1) / 2 + c] = rp[c];
    }
    const bool ok = chol6(D, L, rd);
    if (!ok) { if (tid == 0) *fail_flag = 1; __syncthreads(); return; }      // uniform: every thread saw the same block
    if (tid == 0) {
#pragma unroll
      for (int q = 0; q < 21; ++q) Ld[kb * 27 + q] = L[q];
#pragma unroll
      for (int q = 0; q < 6; ++q) Ld[kb * 27 + 21 + q] = rd[q];
    }
    // (b) panel rows j0+6 .. iend and the rhs row n, those whose envelope reaches this block column
    for (int ii = j0 + 6 + tid; ii <= iend + 1; ii += nt) {
      const int i = ii <= iend ? ii : n;
      if (i < n && first[i / 6] > kb) continue;
      double* rp = &A.row(i)(j0);
      double x[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double v = rp[c];
#pragma unroll
        for (int k = 0; k < c; ++k) v = fma(-x[k], L[c * (c + 1) / 2 + k], v);
        x[c] = v * rd[c];
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) rp[c] = x[c];
    }
    __syncthreads();
    // (c) trailing update, rows i in (j0+6 .. iend] and n, columns c in (j0+6 .. min(i, iend)]
    for (int ii = j0 + 6 + ty; ii <= iend + 1; ii += nty) {
      const int i = ii <= iend ? ii : n;
      if (i < n && first[i / 6] > kb) continue;
      const MatRow R = A.row(i);
      double li[6];
      {
        const double* rp = &R(j0);
#pragma unroll
        for (int t = 0; t < 6; ++t) li[t] = rp[t];
      }
      const int cmax = (i < n) ? i : iend;
      for (int c = j0 + 6 + tx; c <= cmax; c += 16) {
        if (first[c / 6] > kb) continue;
        const double* cp = &A.row(c)(j0);
        double& dst = R(c);
        double acc = dst;
#pragma unroll
        for (int t = 0; t < 6; ++t) acc = fma(-li[t], cp[t], acc);
        dst = acc;
      }
    }
    __syncthreads();
  }
  BA_PROBE(2);
  // back substitution L^T x = y (y in row n), block rows from the bottom; x overwrites y.  Right-looking: once the six
  // unknowns of block kb are known, every earlier entry takes its update y[i] -= sum_c L[j0+c][i] x[c] independently, so
  // a block step is one redundant 6x6 triangular solve per thread (registers, reciprocal diagonals, no division), one
  // parallel update and ONE barrier.  (The left-looking form - dot products over the rows below, 36 wave shuffles of
  // doubles, a serial thread-0 solve with six fp64 divisions, two barriers - cost 20 of the solve's 37 us at P = 7:
  // measured with ablation builds.)
  for (int kb = P - 1; kb >= 0; --kb) {
    const int j0 = 6 * kb;
    const double* L = Ld + kb * 27;
    double x[6];
#pragma unroll
    for (int c = 5; c >= 0; --c) {
      double v = Y(j0 + c);
#pragma unroll
      for (int k = c + 1; k < 6; ++k) v = fma(-L[k * (k + 1) / 2 + c], x[k], v);
      x[c] = v * L[21 + c];                                  // reciprocal diagonal
    }
    __syncthreads();                                         // everyone has read y[j0..j0+6) before it is overwritten
    if (tid < 6) Y(j0 + tid) = x[tid];
    const MatRow R0 = A.row(j0);                             // rows j0 .. j0+5 of one block row: 6 entries apart
    for (int i = 6 * first[kb] + tid; i < j0; i += nt) {     // row block kb of L is zero left of its envelope
      const double* lp = &R0(i);
      double v = Y(i);
#pragma unroll
      for (int c = 0; c < 6; ++c) v = fma(-lp[c * A.rowstep()], x[c], v);
      Y(i) = v;
    }
    __syncthreads();
  }
}

// ---- the same factorisation by ONE wave, without workgroup barriers (matrices that live in LDS) ------------------------------
// The block steps of a Cholesky factorisation are a serial chain; with four waves every step paid two s_barriers, a scan of
// the candidate rows by 256 threads and element-wise LDS traffic through the row accessor (tools/ba_solve_timeline.py: 3.8 k
// cycles per block column at 7 poses, 7.6 k at 63).  One wave needs no barrier - LDS operations of a wave are performed in
// program order, only the COMPILER has to be kept from reordering them (wave_lds_sync) - and the step's work is small: at
// most a few hundred 6-element row tasks.  Per block column kb:
//   (a) every lane factors the 6 x 6 diagonal block redundantly in registers (as before);
//   (b) the active block rows (envelope reaches column kb) are compacted into a list by ballot;
//   (c) panel: one lane per row of the active blocks and the rhs row;
//   (d) trailing update: one lane per (block pair, row): six entries of the pair's block, 36 FMAs, operands as 16-byte reads.
// Every entry sees exactly the operations of chol_solve_blocked in the same order (t = 0..5 within a step, steps ascending),
// so the result is bit-identical to it - and to the dense factorisation, for the reason given at EnvMat.
constexpr int kMaxEnvBlocks = 2048;      // poses the envelope table covers (beyond: dense, first = 0)
constexpr int kMaxActiveRows = 512;      // active block rows of one step (each owns a stored block: the LDS budget holds < 490)

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void ld6(const double* p, double (&v)[6]) {
  const double2 a = reinterpret_cast<const double2*>(p)[0], b = reinterpret_cast<const double2*>(p)[1], c = reinterpret_cast<const double2*>(p)[2];
  v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
}
__device__ __forceinline__ void st6(double* p, const double (&v)[6]) {
  reinterpret_cast<double2*>(p)[0] = double2{v[0], v[1]};
  reinterpret_cast<double2*>(p)[1] = double2{v[2], v[3]};
  reinterpret_cast<double2*>(p)[2] = double2{v[4], v[5]};
}

template <class Mat>
__device__ __forceinline__ void chol_solve_wave(Mat A, double* Ld, int n, int* fail_flag, const int* first, const int* reach, unsigned short* rows) {
  // executed by the 64 lanes of ONE wave; on return the rhs holds the solution x
  const int lane = threadIdx.x & 63;
  const int P = n / 6;
  for (int kb = 0; kb < P; ++kb) {
    double D[21], L[21], rd[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double* rp = A.brow(kb, r, kb);
#pragma unroll
      for (int c = 0; c <= r; ++c) D[r * (r + 1) / 2 + c] = rp[c];
    }
    const bool ok = chol6(D, L, rd);
    if (!ok) { if (lane == 0) *fail_flag = 1; return; }                // uniform: every lane saw the same block
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 21; ++q) Ld[kb * 27 + q] = L[q];
#pragma unroll
      for (int q = 0; q < 6; ++q) Ld[kb * 27 + 21 + q] = rd[q];
    }
    // (b) the block rows below kb whose envelope reaches this block column
    const int lo = kb + 1, hi = reach[kb];
    int m = 0;
    for (int base = lo; base <= hi; base += 64) {
      const int ib = base + lane;
      const bool act = ib <= hi && first[ib] <= kb;
      const unsigned long long mask = __ballot(act);
      const int at = m + __popcll(mask & ((1ull << lane) - 1ull));
      if (act && at < kMaxActiveRows) rows[at] = static_cast<unsigned short>(ib);
      m += __popcll(mask);
    }
    if (m > kMaxActiveRows) { if (lane == 0) *fail_flag = 1; return; }   // (cannot happen for a matrix that fits the LDS budget: every active row owns a stored block)
    wave_lds_sync();
    // (c) panel: x = row L^-T, in place; task 6 m is the rhs row
    for (int t = lane; t <= 6 * m; t += 64) {
      const int q = t / 6;
      double* rp = (t < 6 * m) ? A.brow(rows[q], t - 6 * q, kb) : A.yrow(kb);
      double v[6], x[6];
      ld6(rp, v);
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double u = v[c];
#pragma unroll
        for (int k = 0; k < c; ++k) u = fma(-x[k], L[c * (c + 1) / 2 + k], u);
        x[c] = u * rd[c];
      }
      st6(rp, x);
    }
    wave_lds_sync();
    // (d) trailing update: block (a, b) of the active list, a >= b, row r: A(a,b)[r][:] -= L(a,kb)[r][:] L(b,kb)^T; then
    // one task per active block for the rhs row
    const int npair = m * (m + 1) / 2;
    const int ntask = 6 * npair + m;
    for (int t = lane; t < ntask; t += 64) {
      double li[6];
      double* dst;
      int cb, cmax = 5;
      if (t < 6 * npair) {
        const int pr = t / 6, r = t - 6 * pr;
        int a = static_cast<int>((sqrtf(8.0f * static_cast<float>(pr) + 1.0f) - 1.0f) * 0.5f);
        while ((a + 1) * (a + 2) / 2 <= pr) ++a;
        while (a * (a + 1) / 2 > pr) --a;
        const int b = pr - a * (a + 1) / 2;
        const int ib = rows[a];
        cb = rows[b];
        ld6(A.brow(ib, r, kb), li);
        dst = A.brow(ib, r, cb);
        if (a == b) cmax = r;                                          // lower triangle of a diagonal block
      } else {
        cb = rows[t - 6 * npair];
        ld6(A.yrow(kb), li);
        dst = A.yrow(cb);
      }
      double acc[6];
      ld6(dst, acc);
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        double cp[6];
        ld6(A.brow(cb, c, kb), cp);
        double u = acc[c];
#pragma unroll
        for (int q = 0; q < 6; ++q) u = fma(-li[q], cp[q], u);
        acc[c] = (c <= cmax) ? u : acc[c];
      }
      st6(dst, acc);
    }
    wave_lds_sync();
  }
  BA_PROBE(2);
  // back substitution L^T x = y, block rows from the bottom (right-looking, as in chol_solve_blocked)
  for (int kb = P - 1; kb >= 0; --kb) {
    const double* L = Ld + kb * 27;
    double* y = A.yrow(kb);
    double x[6];
#pragma unroll
    for (int c = 5; c >= 0; --c) {
      double v = y[c];
#pragma unroll
      for (int k = c + 1; k < 6; ++k) v = fma(-L[k * (k + 1) / 2 + c], x[k], v);
      x[c] = v * L[21 + c];                                            // reciprocal diagonal
    }
    wave_lds_sync();                                                   // every lane has read y before it is overwritten
    if (lane < 6) y[lane] = x[lane];
    for (int i = 6 * first[kb] + lane; i < 6 * kb; i += 64) {          // row block kb of L is zero left of its envelope
      const int cbk = i / 6, c2 = i - 6 * cbk;
      double* yp = A.yrow(cbk) + c2;
      double v = *yp;
#pragma unroll
      for (int c = 0; c < 6; ++c) v = fma(-A.brow(kb, c, cbk)[c2], x[c], v);
      *yp = v;
    }
    wave_lds_sync();
  }
}

// ---- ... and by FOUR waves in a look-ahead pipeline --------------------------------------------------------------------------
// What bounds the two forms above is the serial chain  6 x 6 Cholesky of block (kb, kb) -> panel -> update of block column
// kb + 1 -> next 6 x 6 Cholesky; the rest of a step's trailing update (all block columns beyond kb + 1) and the whole forward
// substitution of the right-hand side are not on it.  Here wave 0 walks the chain and nothing else:
//   wave 0    step kb: factor the diagonal block, panel; make sure the workers have finished step kb - 1; publish
//             (`panel` = kb); update block column kb + 1 (the look-ahead); next step.
//   wave 1    the right-hand side (its panel solve and its updates, all steps in order), then pair tasks
//   waves 2-3 pair tasks of step kb for the block columns beyond kb + 1, as soon as `panel` >= kb; publish `done[w]` = kb.
// Order of the operations an entry sees (= bit-identity with chol_solve_blocked): a worker starts step kb when `panel` >= kb,
// which wave 0 publishes only after EVERY worker has finished step kb - 1 (two steps' updates of one entry may belong to two
// different waves); wave 0's look-ahead of step kb comes after the same wait.  Waves of one workgroup are resident together,
// so waiting on a flag in LDS cannot deadlock; LDS operations of a wave are performed in order, so data written before a flag
// are visible to whoever sees the flag (the fences only pin the compiler and the wait counters).  Waits are bounded: a wave
// that gives up sets `abort` and the solve reports failure instead of hanging the device.
// Wave 0's step: the active rows of every step are listed ONCE, before the factorisation, by the whole workgroup (pipe_lists:
// they depend on the envelope only); a step's lookups (list, row offsets) are made during the step before; the panel's
// operand rows and the workers' flags are loaded BEFORE the 6 x 6 Cholesky; a lane keeps its panel row in registers for the
// look-ahead update (it is that update's left operand).
// Measured (tools/ba_solve_timeline.py, 63 free poses): 4.1 k cycles per block column against 6.7 k for one wave doing
// everything.  A lone wave pays ~10-16 cycles of issue per LDS instruction and ~8 per fp64 operation, and ~150-200 per
// dependent LDS round trip, so what is left is the 6 x 6 Cholesky (~1.0 k), the look-ahead (~1.5 k: 24 16-byte loads, 36
// FMAs, 3 stores per lane) and the panel.  Tried on top of this and slower: the next diagonal block handed over through 42
// v_readlane instead of LDS, and the factored block stored by six panel lanes instead of 27 stores of lane 0 (+5 %); the
// look-ahead split into wave 0 (diagonal block, one entry per lane) and a worker (rest of the column, own flag) (+17 %: two
// flag hand-overs on the chain cost more than the arithmetic they take off it).
constexpr int kPipeSpinLimit = 1 << 20;
constexpr int kPipeListMax = 2 * kMaxActiveRows;      // sub-diagonal blocks inside the envelope (every one is a stored block: < 494 in LDS)
struct __attribute__((aligned(16))) PipeCtl { int panel; int done[3]; int abort; int total; };

__device__ __forceinline__ int lds_peek(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void pipe_post(int* flag, int v) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool pipe_wait_panel(PipeCtl* ctl, int v) {
  int spins = 0;
  while (lds_peek(&ctl->panel) < v) {
    if (lds_peek(&ctl->abort) || ++spins > kPipeSpinLimit) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return true;
}
__device__ __forceinline__ int pipe_done_min(PipeCtl* ctl) {            // three reads in flight, one round trip
  const int a = lds_peek(&ctl->done[0]), b = lds_peek(&ctl->done[1]), c = lds_peek(&ctl->done[2]);
  return min(a, min(b, c));
}
__device__ __forceinline__ bool pipe_wait_done(PipeCtl* ctl, int seen, int v) {
  int spins = 0;
  while (seen < v) {
    if (lds_peek(&ctl->abort) || ++spins > kPipeSpinLimit) return false;
    __builtin_amdgcn_s_sleep(1);
    seen = pipe_done_min(ctl);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return true;
}

// Lists of every step's active block rows (ib > kb with first[ib] <= kb), ascending, in compressed-column form:
// rowlist[colptr[kb] .. colptr[kb + 1]).  `reach` is consumed: on return it holds colptr (P + 1 entries; P < kMaxEnvBlocks).
// Whole workgroup; returns false (everything untouched) when the lists do not fit.
__device__ __forceinline__ bool pipe_lists(const int* first, int* reach, int P, unsigned short* rowlist, PipeCtl* ctl) {
  int hi_keep[kMaxEnvBlocks / 256], cnt_keep[kMaxEnvBlocks / 256];
  int mine = 0;
#pragma unroll
  for (int q = 0; q < kMaxEnvBlocks / 256; ++q) {
    const int kb = q * 256 + threadIdx.x;
    int cnt = 0, hi = -1;
    if (kb < P) {
      hi = reach[kb];
      for (int ib = kb + 1; ib <= hi; ++ib) cnt += first[ib] <= kb ? 1 : 0;
    }
    hi_keep[q] = hi; cnt_keep[q] = cnt; mine += cnt;
  }
  if (threadIdx.x == 0) ctl->total = 0;
  __syncthreads();
  if (mine) atomicAdd(&ctl->total, mine);
  __syncthreads();
  if (ctl->total > kPipeListMax || P + 1 > kMaxEnvBlocks) return false;
#pragma unroll
  for (int q = 0; q < kMaxEnvBlocks / 256; ++q) {
    const int kb = q * 256 + threadIdx.x;
    if (kb < P) reach[kb] = cnt_keep[q];
  }
  __syncthreads();
  block_scan<false>(reach, P);                                            // inclusive
#pragma unroll
  for (int q = 0; q < kMaxEnvBlocks / 256; ++q) {
    const int kb = q * 256 + threadIdx.x;
    if (kb >= P) continue;
    int at = reach[kb] - cnt_keep[q];
    for (int ib = kb + 1; ib <= hi_keep[q]; ++ib)
      if (first[ib] <= kb) rowlist[at++] = static_cast<unsigned short>(ib);
  }
  __syncthreads();
  // inclusive -> colptr: shift by one entry (through registers: in place)
  int incl[kMaxEnvBlocks / 256];
#pragma unroll
  for (int q = 0; q < kMaxEnvBlocks / 256; ++q) {
    const int kb = q * 256 + threadIdx.x;
    incl[q] = kb < P ? reach[kb] : 0;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kMaxEnvBlocks / 256; ++q) {
    const int kb = q * 256 + threadIdx.x;
    if (kb < P) reach[kb + 1] = incl[q];
  }
  if (threadIdx.x == 0) reach[0] = 0;
  __syncthreads();
  return true;
}

// acc (row r of block (ib, cb)) -= li (row r of L(ib, kb)) times L(cb, kb)^T, whose rows start at cprow0 and lie rstep apart;
// entries beyond column cmax stay (diagonal blocks: lower triangle)
__device__ __forceinline__ void pair_update(const double* cprow0, int rstep, const double (&li)[6], int cmax, double (&acc)[6]) {
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double cp[6];
    ld6(cprow0 + c * rstep, cp);
    double u = acc[c];
#pragma unroll
    for (int q = 0; q < 6; ++q) u = fma(-li[q], cp[q], u);
    acc[c] = (c <= cmax) ? u : acc[c];
  }
}
template <class Mat>
__device__ __forceinline__ void pair_task(const Mat& A, int kb, int ib, int r, int cb, int cmax) {
  double li[6], acc[6];
  ld6(A.brow(ib, r, kb), li);
  double* dst = A.brow(ib, r, cb);
  ld6(dst, acc);
  pair_update(A.brow(cb, 0, kb), static_cast<int>(A.rowstep()), li, cmax, acc);
  st6(dst, acc);
}
__device__ __forceinline__ void panel_row(const double (&v)[6], const double (&L)[21], const double (&rd)[6], double (&x)[6]) {
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    double u = v[c];
#pragma unroll
    for (int k = 0; k < c; ++k) u = fma(-x[k], L[c * (c + 1) / 2 + k], u);
    x[c] = u * rd[c];
  }
}

// Block columns [kb0, kb1) of the factorisation (and of the right-hand side's forward substitution): the whole of it is
// (0, P); the partitioned solve (ba_solve_twin_kernel) stops in between, and continues with the flags where they stand -
// `panel` and `done[]` hold kb0 - 1 when a range ends at kb0.  Executed by the four waves of the workgroup, after pipe_lists;
// the caller puts a barrier behind it.
template <class Mat>
__device__ __forceinline__ void chol_pipe_range(Mat A, double* Ld, int P, int* fail_flag, const int* colptr,
                                const unsigned short* rowlist, PipeCtl* ctl, int kb0, int kb1) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rstep = static_cast<int>(A.rowstep()), bstep = A.blockstep();
#define PIPE_GIVE_UP() do { if (lane == 0) { *fail_flag = 1; __hip_atomic_store(&ctl->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } return; } while (0)
  if (kb0 >= kb1) return;
  if (wave == 0) {
    const int q0 = lane / 6, r0 = lane - 6 * q0;
    int m = colptr[kb0 + 1] - colptr[kb0];
    const unsigned short* rows = rowlist + colptr[kb0];
    bool one_trip = 6 * m <= 64, mine = one_trip && lane < 6 * m;
    double* rp0 = A.brow(mine ? rows[q0] : kb0, r0, kb0);
    double* dptr = A.brow(kb0, 0, kb0);
    bool ahead = m > 0 && rows[0] == kb0 + 1;
    for (int kb = kb0; kb < kb1; ++kb) {
      const bool more = kb + 1 < P;
      BA_PROBE_STEP(8);
      double D[21], L[21], rd[6], v0[6], x0[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double v[6];
        ld6(dptr + r * rstep, v);
#pragma unroll
        for (int c = 0; c <= r; ++c) D[r * (r + 1) / 2 + c] = v[c];
      }
      if (one_trip) ld6(rp0, v0);                                          // (idle lanes read some valid row)
      const int seen = kb >= 1 ? pipe_done_min(ctl) : 0;                   // the workers' progress, read early
      const int nbeg = more ? colptr[kb + 1] : 0;                          // next step, first lookup
      const int nm = more ? colptr[kb + 2] - nbeg : 0;
      BA_PROBE_STEP(9);
      const bool ok = chol6(D, L, rd);
      if (!ok) PIPE_GIVE_UP();                                             // uniform: every lane saw the same block
      BA_PROBE_STEP(10);
      const unsigned short* nrows = rowlist + nbeg;                        // next step, second lookup
      const bool n_one = 6 * nm <= 64, n_mine = n_one && lane < 6 * nm;
      const int nib0 = n_mine ? nrows[q0] : (more ? kb + 1 : kb);
      const bool nahead = nm > 0 && nrows[0] == kb + 2;
      if (one_trip) {
        panel_row(v0, L, rd, x0);
        if (mine) st6(rp0, x0);
      } else {
        for (int t = lane; t < 6 * m; t += 64) {
          const int q = t / 6;
          double* rp = A.brow(rows[q], t - 6 * q, kb);
          double v[6], x[6];
          ld6(rp, v);
          panel_row(v, L, rd, x);
          st6(rp, x);
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 21; ++q) Ld[kb * 27 + q] = L[q];
#pragma unroll
        for (int q = 0; q < 6; ++q) Ld[kb * 27 + 21 + q] = rd[q];
      }
      BA_PROBE_STEP(11);
      double* nrp0 = A.brow(nib0, r0, more ? kb + 1 : kb);                 // next step, third lookup (the rows' offsets)
      double* ndptr = more ? A.brow(kb + 1, 0, kb + 1) : dptr;
      // nobody may start step kb before every worker has finished step kb - 1; the same wait covers the look-ahead below
      // (block column kb + 1 must have seen the workers' step kb - 1 before this step's update and the next Cholesky)
      if (kb >= 1 && !pipe_wait_done(ctl, seen, kb - 1)) PIPE_GIVE_UP();
      BA_PROBE_STEP(12);
      pipe_post(&ctl->panel, kb);
      BA_PROBE_STEP(13);
      if (ahead) {                                                         // look-ahead: pairs (a, 0) of the list; L(kb + 1, kb) sits left of the next diagonal block
        if (one_trip) {
          if (mine) {
            double* dst = rp0 + bstep;
            double acc[6];
            ld6(dst, acc);
            pair_update(ndptr - bstep, rstep, x0, q0 == 0 ? r0 : 5, acc);
            st6(dst, acc);
          }
        } else {
          for (int t = lane; t < 6 * m; t += 64) {
            const int a = t / 6, r = t - 6 * a;
            pair_task(A, kb, rows[a], r, kb + 1, a == 0 ? r : 5);
          }
        }
      }
      wave_lds_sync();
      BA_PROBE_STEP(14);
      m = nm; rows = nrows; one_trip = n_one; mine = n_mine; rp0 = nrp0; dptr = ndptr; ahead = nahead;
    }
    // the right-hand side's forward substitution is wave 1's: wait for its last step
    if (!pipe_wait_done(ctl, pipe_done_min(ctl), kb1 - 1)) PIPE_GIVE_UP();
    wave_lds_sync();
    return;
  }
  // ---- workers
  const int slot = wave == 1 ? 2 : wave - 2;                               // pair tasks go to waves 2, 3 first: wave 1 has the rhs
  for (int kb = kb0; kb < kb1; ++kb) {
    const int beg = colptr[kb], m = colptr[kb + 1] - beg;
    const unsigned short* rows = rowlist + beg;
    if (!pipe_wait_panel(ctl, kb)) PIPE_GIVE_UP();
    if (wave == 1) {
      // rhs: y(kb) <- L(kb,kb)^-1 y(kb) (every lane redundantly, lane 0 stores), then y(cb) -= L(cb, kb) y(kb) per active row
      const double* Lp = Ld + kb * 27;
      double L[21], rd[6];
#pragma unroll
      for (int q = 0; q < 21; ++q) L[q] = Lp[q];
#pragma unroll
      for (int q = 0; q < 6; ++q) rd[q] = Lp[21 + q];
      double* y = A.yrow(kb);
      double v[6], x[6];
      ld6(y, v);
      panel_row(v, L, rd, x);
      if (lane == 0) st6(y, x);
      for (int a = lane; a < m; a += 64) {
        const int cb = rows[a];
        double* dst = A.yrow(cb);
        double acc[6];
        ld6(dst, acc);
        pair_update(A.brow(cb, 0, kb), rstep, x, 5, acc);
        st6(dst, acc);
      }
    }
    const int b0 = (m > 0 && rows[0] == kb + 1) ? 1 : 0;                   // block column kb + 1 is wave 0's
    const int mm = m - b0;
    const int ntask = 6 * (mm * (mm + 1) / 2);
    for (int t = slot * 64 + lane; t < ntask; t += 192) {
      const int pr = t / 6, r = t - 6 * pr;
      int a = static_cast<int>((sqrtf(8.0f * static_cast<float>(pr) + 1.0f) - 1.0f) * 0.5f);
      while ((a + 1) * (a + 2) / 2 <= pr) ++a;
      while (a * (a + 1) / 2 > pr) --a;
      const int b = pr - a * (a + 1) / 2;
      pair_task(A, kb, rows[a + b0], r, rows[b + b0], a == b ? r : 5);
    }
    pipe_post(&ctl->done[wave - 1], kb);
  }
#undef PIPE_GIVE_UP
}

// Back substitution L^T x = y by wave 0, block rows kb_top .. 0, the arithmetic of chol_solve_blocked.  One wave, LDS
// operations in program order, so no fence inside: a step's operands that do not depend on x (this lane's column of
// L and its y entry, the next step's factored diagonal block) are in flight while the 6 x 6 triangular solve runs.
// Block rows >= given_from hold their solution already (the partitioned solve: the separator's unknowns arrive from the
// other workgroup); they only update the rows below given_from.  given_from > kb_top: the ordinary substitution.
// Stops after block row kb_bot (every row above it has then received all its updates from the rows processed).
template <class Mat>
__device__ __forceinline__ void pipe_backsub(Mat A, const double* Ld, const int* first, int kb_top, int given_from, int kb_bot = 0) {
  if (threadIdx.x >= 64 || kb_top < kb_bot) return;
  const int lane = threadIdx.x & 63;
  const int rstep = static_cast<int>(A.rowstep());
  double Lc[27];
#pragma unroll
  for (int q = 0; q < 27; ++q) Lc[q] = Ld[kb_top * 27 + q];
  for (int kb = kb_top; kb >= kb_bot; --kb) {
    double* y = A.yrow(kb);
    double yv[6], x[6], Ln[27];
    ld6(y, yv);
    const int ibeg = 6 * first[kb];
    const bool given = kb >= given_from;
    const int iend = 6 * (given ? given_from : kb);
    if (kb > 0) {
#pragma unroll
      for (int q = 0; q < 27; ++q) Ln[q] = Ld[(kb - 1) * 27 + q];
    }
#pragma unroll
    for (int c = 5; c >= 0; --c) {
      double v = yv[c];
#pragma unroll
      for (int k = c + 1; k < 6; ++k) v = fma(-Lc[k * (k + 1) / 2 + c], x[k], v);
      x[c] = given ? yv[c] : v * Lc[21 + c];                               // reciprocal diagonal
    }
    asm volatile("" ::: "memory");
    if (lane < 6 && !given) y[lane] = x[lane];
    asm volatile("" ::: "memory");
    for (int i = ibeg + lane; i < iend; i += 64) {                         // row block kb of L is zero left of its envelope
      const int cbk = i / 6, c2 = i - 6 * cbk;
      double* yp = A.yrow(cbk) + c2;
      const double* lp = A.brow(kb, 0, cbk) + c2;
      double v = *yp;
#pragma unroll
      for (int c = 0; c < 6; ++c) v = fma(-lp[c * rstep], x[c], v);
      *yp = v;
    }
    asm volatile("" ::: "memory");                                         // (compiler only: the next step reads what other lanes wrote)
    if (kb > 0) {
#pragma unroll
      for (int q = 0; q < 27; ++q) Lc[q] = Ln[q];
    }
  }
  wave_lds_sync();
}

template <class Mat>
__device__ __forceinline__ void chol_solve_pipe(Mat A, double* Ld, int n, int* fail_flag, const int* first, const int* colptr,
                                const unsigned short* rowlist, PipeCtl* ctl) {
  // executed by the four waves of the workgroup, after pipe_lists; on return (and after the caller's barrier) the rhs holds x
  const int P = n / 6;
  chol_pipe_range(A, Ld, P, fail_flag, colptr, rowlist, ctl, 0, P);
  if (threadIdx.x < 64 && !lds_peek(&ctl->abort)) {
    BA_PROBE(2);
    pipe_backsub(A, Ld, first, P - 1, P);
  }
}


// ---- systems beyond the dense LDS path (more than 22 free poses: the global bundle adjustment) --------------------------
// Two multi-workgroup kernels prepare the one-workgroup solve (inside it, reading 1.1 MB of fixed point through a single CU
// was 429 k of its 1479 k cycles at 63 poses - tools/ba_solve_timeline.py):
//   ba_env_kernel      numeric envelope: env[b] = first block column with a non-zero in block row b (atomicMin; the table is
//                      INT_MAX between solves).  Numeric, not structural: in an edge-sharded BA the system is the all-reduced
//                      one, whose couplings this rank's edge list does not know.
//   ba_prepare_kernel  fixed point -> fp64 with the damping, `sys` zeroed, into the layout the solve will use: compact
//                      envelope blocks if they fit the LDS budget, else dense.
// env_layout: offsets of the compact layout (block row b = blocks first[b] .. b); returns the number of doubles of all
// blocks.  Executed by one thread.
// first[] from the numeric envelope and the offsets of the compact layout (block row b = blocks first[b] .. b); returns the
// number of doubles of all blocks, or INT_MAX when that does not fit an int.  Whole workgroup; valid after it returns.
__device__ __forceinline__ int env_layout(const int* __restrict__ env, int P, int* first, int* rowbase, int* total_s) {
  for (int b = threadIdx.x; b < P; b += blockDim.x) {
    const int e = env[b];
    const int f = e < b ? (e < 0 ? 0 : e) : b;             // clamped to [0, b] on BOTH sides, as pvo_ba_packed_elems sizes the message
    first[b] = f;
    rowbase[b] = (b - f + 1) * 36;                         // sizes, scanned below
  }
  __syncthreads();
  // (sizes up to 36 * 2048 per row: the running sum can overflow an int only far beyond any LDS budget - clamp)
  block_scan<false>(rowbase, P);
  if (threadIdx.x == 0) *total_s = (P > 0 && rowbase[P - 1] >= 0) ? rowbase[P - 1] : 0x7fffffff;
  __syncthreads();
  for (int b = threadIdx.x; b < P; b += blockDim.x) {      // inclusive -> exclusive
    const int sz = (b - first[b] + 1) * 36;
    rowbase[b] -= sz;
  }
  __syncthreads();
  return *total_s;
}
// doubles of LDS the compact path needs: blocks + rhs + Ld (27 per pose) + slack, and (as ints) the offset table
__device__ __host__ __forceinline__ long long env_lds_bytes(long long blocks, int n, int P) {
  return 16 + 8 * (blocks + n + 27LL * P + 24) + 4LL * (P + 2);
}

__global__ __launch_bounds__(256) void ba_env_kernel(const long long* __restrict__ sys, int* __restrict__ env, int n) {
  // the minima of the few block rows a workgroup's 2048 entries span are formed in LDS first: one atomic per entry on ~P
  // global addresses was 69 us of a 64-keyframe step (15 k non-zeros below the diagonal, ~250 per address)
  constexpr int kSlots = 64;
  __shared__ int smin[kSlots];
  const int NN = n * n;
  const int base = blockIdx.x * 2048;
  const int br0 = (base / n) / 6;
  if (threadIdx.x < kSlots) smin[threadIdx.x] = 0x7fffffff;
  __syncthreads();
  long long raw[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int idx = base + u * 256 + threadIdx.x;
    raw[u] = idx < NN ? sys[idx] : 0;
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int idx = base + u * 256 + threadIdx.x;
    if (idx >= NN || raw[u] == 0) continue;
    const int r = idx / n, c = idx - r * n;
    if (c >= r) continue;
    const int slot = r / 6 - br0, cb = c / 6;
    if (slot < kSlots) { if (cb < smin[slot]) atomicMin(&smin[slot], cb); }      // (the plain read only skips atomics that cannot lower the minimum)
    else atomicMin(&env[r / 6], cb);
  }
  __syncthreads();
  if (threadIdx.x < kSlots && smin[threadIdx.x] != 0x7fffffff) atomicMin(&env[br0 + threadIdx.x], smin[threadIdx.x]);
}

// The partition of the pose chain (ba_solve_twin_kernel): poses [0, m) are eliminated by one workgroup, poses [s, P) - in
// reverse order - by another, at the same time; the separator [m, s) = every pose that couples with one below m (s =
// reach[m - 1] + 1) comes last.  Chosen to make the longer chain + the separator shortest; 0 / 0 = not partitioned: separator
// wider than kMaxSep (loop closures), a chain less than a quarter shorter, or an image that does not fit the LDS budget.
// One workgroup, after env_layout (first, exclusive row offsets) and envelope_reach.
__device__ __forceinline__ void choose_partition(const int* first, const int* reach, const int* rowbase, int P, int blocks,
                                                 long long lds_budget, int* xchg_i) {
  __shared__ int best_s, colsum_s;
  if (threadIdx.x == 0) { best_s = 0x7fffffff; colsum_s = 0; }
  __syncthreads();
  for (int m = 1 + threadIdx.x; m < P; m += blockDim.x) {
    const int sp = reach[m - 1] + 1, w = sp - m;
    if (w < 1 || w > kMaxSep || P - sp < 2 || m < 2) continue;
    const int cost = (m > P - sp ? m : P - sp) + w;
    atomicMin(&best_s, cost * 4096 + m);
  }
  __syncthreads();
  int m = 0, sp = 0;
  if (best_s != 0x7fffffff && P < 4096) {
    m = best_s & 4095; sp = reach[m - 1] + 1;
    if (4 * (best_s >> 12) > 3 * P) m = 0;
  }
  if (m) {                                                 // sizes of the two images (bounds: the separator rows are stored dense)
    int mine = 0;
    for (int g = m + threadIdx.x; g < P; g += blockDim.x) mine += reach[g] - g + 1;
    if (mine) atomicAdd(&colsum_s, mine);
    __syncthreads();
    const int w = sp - m;
    const long long b0 = rowbase[sp] + 36LL * w * w, b1 = 36LL * (colsum_s + w * w);
    const long long tail = 16LL * (P + 2);                 // the global tables the loaders keep beside the image
    const bool fits = env_lds_bytes(b0, 6 * sp, sp) + tail <= lds_budget && env_lds_bytes(b1, 6 * (P - m), P - m) + tail <= lds_budget &&
                      b0 / 36 - sp <= kPipeListMax && b1 / 36 - (P - m) <= kPipeListMax && blocks > 0;
    if (!fits) m = 0;
  }
  if (threadIdx.x == 0) { xchg_i[0] = 0; xchg_i[1] = 0; xchg_i[2] = m; xchg_i[3] = m ? sp : 0; }
}

__global__ __launch_bounds__(256) void ba_prepare_kernel(long long* __restrict__ sys, double* __restrict__ out, const int* __restrict__ env,
                                                         int n, float lm, float ep, long long lds_budget, int* __restrict__ xchg_i) {
  __shared__ int first[kMaxEnvBlocks], rowbase[kMaxEnvBlocks + 1];
  __shared__ int blocks_s;
  const int P = n / 6;
  const int blocks = env_layout(env, P, first, rowbase, &blocks_s);
  const bool compact = env_lds_bytes(blocks, n, P) <= lds_budget;
  if (xchg_i && blockIdx.x == 0) {                         // (block-uniform)
    __shared__ int reach[kMaxEnvBlocks];
    if (compact && P > 12) {
      envelope_reach(first, reach, P);
      choose_partition(first, reach, rowbase, P, blocks, lds_budget, xchg_i);
    } else if (threadIdx.x == 0) {
      xchg_i[0] = 0; xchg_i[1] = 0; xchg_i[2] = 0; xchg_i[3] = 0;
    }
  }
  const int N = n * n + n;
  const int base = blockIdx.x * 2048;
  long long raw[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int idx = base + u * 256 + threadIdx.x;
    raw[u] = idx < N ? sys[idx] : 0;
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int idx = base + u * 256 + threadIdx.x;
    if (idx >= N) continue;
    double v = static_cast<double>(raw[u]) * kInvFix;       // fixed point -> fp64
    sys[idx] = 0;                                           // ready for the next Gauss-Newton step's accumulation
    if (idx >= n * n) {                                     // rhs
      out[compact ? blocks + (idx - n * n) : idx] = v;
      continue;
    }
    const int r = idx / n, c = idx - r * n;
    if (r == c) v += static_cast<double>(ep) + static_cast<double>(lm) * v;   // droid_kernels.cu:1176
    if (!compact) { out[idx] = v; continue; }
    const int rb = r / 6, cb = c / 6;
    if (cb > rb || cb < first[rb]) continue;                // upper triangle / outside the envelope (exact zeros)
    out[rowbase[rb] + (cb - first[rb]) * 36 + (r - 6 * rb) * 6 + (c - 6 * cb)] = v;
  }
}

// ---- the edge-sharded step's message: the STRUCTURAL envelope of the pose system, packed -----------------------------------
// Between pvo_ba_local and pvo_ba_finish an edge-sharded run all-reduces the lower-triangle blocks (b, first_s[b] .. b) and the
// right-hand side (parallel.py: envelope_structure - derived from the whole graph's edge list, the same table on every rank).
// ba_pack_kernel gathers exactly those entries of the dense fixed-point system into one contiguous int64 message, in the
// compact layout of env_layout(first_s) - block row b = blocks first_s[b] .. b of 36 entries, then the 6P right-hand side - and
// leaves `sys` zeroed; after the all-reduce ba_env_packed_kernel / ba_prepare_packed_kernel read the MESSAGE (124 KB at 63
// poses) instead of the dense system (1.15 MB): numeric envelope, fp64 + damping into the solve's layout, partition.  No
// index tensors, no scatter back, nothing of torch's between the two library calls and the collective.
__global__ __launch_bounds__(256) void ba_pack_kernel(long long* __restrict__ sys, const int* __restrict__ first_s, long long* __restrict__ msg, int n) {
  __shared__ int first[kMaxEnvBlocks], rowbase[kMaxEnvBlocks + 1];
  __shared__ int blocks_s;
  const int P = n / 6;
  const int blocks = env_layout(first_s, P, first, rowbase, &blocks_s);
  const int N = n * n + n;
  const int base = blockIdx.x * 2048;
  long long raw[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int idx = base + u * 256 + threadIdx.x;
    raw[u] = idx < N ? sys[idx] : 0;
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int idx = base + u * 256 + threadIdx.x;
    if (idx >= N) continue;
    sys[idx] = 0;
    if (idx >= n * n) { msg[blocks + (idx - n * n)] = raw[u]; continue; }
    const int r = idx / n, c = idx - r * n;
    const int rb = r / 6, cb = c / 6;
    if (cb > rb || cb < first[rb]) continue;                // upper triangle (never written) / structural zeros
    msg[rowbase[rb] + (cb - first[rb]) * 36 + (r - 6 * rb) * 6 + (c - 6 * cb)] = raw[u];
  }
}

// One workgroup per BLOCK ROW of the message (its blocks are contiguous: no search for the row of an entry; a flat split of the
// message with a binary search per entry made these two kernels twice as long as their dense counterparts).
__global__ __launch_bounds__(256) void ba_env_packed_kernel(const long long* __restrict__ msg, const int* __restrict__ first_s, int* __restrict__ env, int n) {
  __shared__ int first[kMaxEnvBlocks], rowbase[kMaxEnvBlocks + 1];
  __shared__ int blocks_s, smin;
  const int P = n / 6;
  env_layout(first_s, P, first, rowbase, &blocks_s);
  const int rb = blockIdx.x;
  if (threadIdx.x == 0) smin = 0x7fffffff;
  __syncthreads();
  const int f = first[rb], nsub = (rb - f) * 36;                         // entries left of the diagonal block
  const long long* row = msg + rowbase[rb];
  for (int base = 0; base < nsub; base += 8 * 256) {
    long long raw[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int k = base + u * 256 + threadIdx.x; raw[u] = k < nsub ? row[k] : 0; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = base + u * 256 + threadIdx.x;
      if (k < nsub && raw[u] != 0) { const int cb = f + k / 36; if (cb < smin) atomicMin(&smin, cb); }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && smin != 0x7fffffff) env[rb] = smin;             // (this workgroup alone writes env[rb]; INT_MAX between solves)
}

// grid: P block rows + one workgroup for the right-hand side
__global__ __launch_bounds__(256) void ba_prepare_packed_kernel(const long long* __restrict__ msg, const int* __restrict__ first_s, double* __restrict__ out,
                                                                const int* __restrict__ env, int n, float lm, float ep, long long lds_budget,
                                                                int* __restrict__ xchg_i) {
  __shared__ int sfirst[kMaxEnvBlocks], srow[kMaxEnvBlocks + 1];       // the message's (structural) layout
  __shared__ int first[kMaxEnvBlocks], rowbase[kMaxEnvBlocks + 1];     // the solve's (numeric) layout
  __shared__ int sblocks_s, blocks_s;
  const int P = n / 6;
  const int sblocks = env_layout(first_s, P, sfirst, srow, &sblocks_s);
  const int blocks = env_layout(env, P, first, rowbase, &blocks_s);
  const bool compact = env_lds_bytes(blocks, n, P) <= lds_budget;
  if (xchg_i && blockIdx.x == 0) {                         // (block-uniform)
    __shared__ int reach[kMaxEnvBlocks];
    if (compact && P > 12) {
      envelope_reach(first, reach, P);
      choose_partition(first, reach, rowbase, P, blocks, lds_budget, xchg_i);
    } else if (threadIdx.x == 0) {
      xchg_i[0] = 0; xchg_i[1] = 0; xchg_i[2] = 0; xchg_i[3] = 0;
    }
  }
  const int rb = blockIdx.x;
  if (rb == P) {                                            // rhs
    for (int i = threadIdx.x; i < n; i += 256)
      out[compact ? blocks + i : static_cast<long long>(n) * n + i] = static_cast<double>(msg[sblocks + i]) * kInvFix;
    return;
  }
  const int sf = sfirst[rb], f = first[rb], nrow = (rb - sf + 1) * 36;
  const long long* row = msg + srow[rb];
  for (int base = 0; base < nrow; base += 8 * 256) {
    long long raw[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int k = base + u * 256 + threadIdx.x; raw[u] = k < nrow ? row[k] : 0; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = base + u * 256 + threadIdx.x;
      if (k >= nrow) continue;
      double v = static_cast<double>(raw[u]) * kInvFix;
      const int q = k / 36, cb = sf + q, e = k - 36 * q, ri = e / 6, ci = e - 6 * ri;
      if (rb == cb && ri == ci) v += static_cast<double>(ep) + static_cast<double>(lm) * v;   // droid_kernels.cu:1176
      if (!compact) { out[static_cast<long long>(6 * rb + ri) * n + 6 * cb + ci] = v; continue; }
      if (cb < f) continue;                                 // inside the structural envelope, outside the numeric one: an exact zero
      out[rowbase[rb] + (cb - f) * 36 + e] = v;
    }
  }
}

// The pose solve is ONE workgroup for ~20 us (a window) to ~170 us (63 free poses) while the rest of the chip idles - and
// nothing else may run beside the bundle adjustment on another queue.  Work that is independent of it can ride in
// the SAME dispatch instead: workgroups 1 .. of this launch run up to three jobs that share no buffer with the solve -
//   a 1x1 convolution of a 128-channel tensor (conv1x1_tile.h): GraphAgg's upsampling mask in pvo_graph_update, 17 us on the
//   launch stream otherwise;
//   the two halves of the ConvGRU's global context (glo_tile.h) of the NEXT update, which depend on the hidden state only:
//   the partial means in the first solve's dispatch, the gate context in the second's (it reads what the first wrote - a
//   kernel boundary in between, no fence inside a dispatch).
struct Riders {
  const uint16_t *cx, *cw; const float* cbias; uint16_t* cy; long long crows; int cCout, cdtype, crow_blocks, cblocks;
  const uint16_t *gnet, *gw; const float* gbias; float* gpart; int gHW, gchunks, gdtype, gblocks;
  const float *xpart, *xwt, *xbias; float* xg; int xchunks, xblocks;
};

__device__ __forceinline__ void ride(unsigned char* smem, const Riders r, int b) {
  if (b < r.cblocks) {
    const int cb = b / r.crow_blocks;
    const long long rb = b - cb * r.crow_blocks;
    if (r.cdtype == PVO_F16) c1t::conv1x1_c128_tile<pvo_half>(smem, r.cx, r.cw, r.cbias, r.cy, r.crows, r.cCout, 0, rb, cb);
    else c1t::conv1x1_c128_tile<pvo_bf16>(smem, r.cx, r.cw, r.cbias, r.cy, r.crows, r.cCout, 0, rb, cb);
    return;
  }
  b -= r.cblocks;
  if (b < r.gblocks) {
    const int e = b / r.gchunks, ci = b - e * r.gchunks;
    if (r.gdtype == PVO_F16) glt::glo_partial_means<pvo_half>(smem, r.gnet, r.gw, r.gbias, r.gpart, r.gHW, 256, e, ci, r.gchunks);
    else glt::glo_partial_means<pvo_bf16>(smem, r.gnet, r.gw, r.gbias, r.gpart, r.gHW, 256, e, ci, r.gchunks);
    return;
  }
  b -= r.gblocks;
  if (b < r.xblocks) glt::gate_context_256(reinterpret_cast<float*>(smem), r.xpart, r.xwt, r.xbias, r.xg, r.xchunks, b);
}

// (experiment hook, tools/variant.py: -DPVO_SOLVE_ATTR='__attribute__((amdgpu_waves_per_eu(3,4)))' caps the registers of the fused
// solve + riders kernel for three workgroups per CU - measured within noise of the default)
#ifndef PVO_SOLVE_ATTR
#define PVO_SOLVE_ATTR
#endif
__global__ __launch_bounds__(256) PVO_SOLVE_ATTR void ba_solve_kernel(
    long long* __restrict__ sys, double* __restrict__ chol_global, float* __restrict__ poses,
    float* __restrict__ dx_ws, float* __restrict__ dx_out, int* __restrict__ meta, int* __restrict__ status_out,
    int P, int t0, float lm, float ep, int use_lds, int* __restrict__ env, long long lds_budget, int solver, Riders riders) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [16 B flags | fp64 matrix + rhs | ...]
  if (blockIdx.x > 0) {                                                  // rider workgroups (launched only with riders)
    ride(smem, riders, blockIdx.x - 1);
    return;
  }
  int& fail = *reinterpret_cast<int*>(smem);
  const int n = 6 * P;
  __shared__ int first[kMaxEnvBlocks];                                           // envelope: first non-zero block column per block row
  __shared__ int reach[kMaxEnvBlocks];                                           // ... and last block row that reaches a block column
  __shared__ unsigned short act_rows[2][kMaxActiveRows];                         // lists of a step's active block rows (wave / pipe solvers)
  __shared__ PipeCtl pipe_ctl;
  __shared__ int blocks_s;
  BA_PROBE(0);
  if (threadIdx.x == 0) {
    fail = 0;
    pipe_ctl.panel = -1; pipe_ctl.done[0] = pipe_ctl.done[1] = pipe_ctl.done[2] = -1; pipe_ctl.abort = 0; pipe_ctl.total = 0;
  }
  double* xrow = nullptr;                                                        // where the solution ends up
  if (!use_lds) {
    // prepared by ba_env_kernel + ba_prepare_kernel
    int* rowbase = nullptr;
    {
      // the offset table sits behind the doubles of the compact layout; its position depends on the block count, which
      // thread 0 computes into a scratch copy first
      int* tmp = reinterpret_cast<int*>(smem + 16);
      env_layout(env, P, first, tmp, &blocks_s);
    }
    const int blocks = blocks_s;
    const bool compact = env_lds_bytes(blocks, n, P) <= lds_budget;
    if (compact) {
      double* blk = reinterpret_cast<double*>(smem + 16);
      double* rhs = blk + blocks;
      double* Ld = rhs + n;
      rowbase = reinterpret_cast<int*>(Ld + 27 * P + 24);
      // (the offsets computed into the scratch copy move to their final place behind the doubles; the scratch is about to be
      // overwritten by the matrix, so they travel through registers)
      int rb_keep[kMaxEnvBlocks / 256];
#pragma unroll
      for (int q = 0; q < kMaxEnvBlocks / 256; ++q) {
        const int b = q * 256 + threadIdx.x;
        rb_keep[q] = b < P ? reinterpret_cast<const int*>(smem + 16)[b] : 0;
      }
      __syncthreads();
      {
        // 124 KB at 63 poses through one workgroup: sixteen 16-byte loads in flight per thread (one 8-byte load at a time
        // was 57 k cycles of latency)
        const int nd2 = (blocks + n) >> 1;                                          // blocks is a multiple of 36: pairs cover blocks + n but for an odd n
        const double2* src = reinterpret_cast<const double2*>(chol_global);
        double2* dst = reinterpret_cast<double2*>(blk);
        // (sixteen per thread, every load unconditional on a clamped index: a guarded `v[u] = src[i]` sends the array to scratch)
        for (int base = 0; base < nd2; base += 16 * blockDim.x) {
          double2 v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) { const int i = base + u * blockDim.x + threadIdx.x; v[u] = src[i < nd2 ? i : nd2 - 1]; }
#pragma unroll
          for (int u = 0; u < 16; ++u) { const int i = base + u * blockDim.x + threadIdx.x; if (i < nd2) dst[i] = v[u]; }
        }
        if (((blocks + n) & 1) && threadIdx.x == 0) blk[blocks + n - 1] = chol_global[blocks + n - 1];
      }
#pragma unroll
      for (int q = 0; q < kMaxEnvBlocks / 256; ++q) {
        const int b = q * 256 + threadIdx.x;
        if (b < P) rowbase[b] = rb_keep[q] - 36 * first[b];                        // (as EnvMat::rowoff)
      }
      for (int b = threadIdx.x; b < P; b += blockDim.x) env[b] = 0x7fffffff;
      __syncthreads();
      envelope_reach(first, reach, P);
      BA_PROBE(1);
      if (solver == 2 && pipe_lists(first, reach, P, act_rows[0], &pipe_ctl))
        chol_solve_pipe(EnvMat{blk, rhs, rowbase, n}, Ld, n, &fail, first, reach, act_rows[0], &pipe_ctl);
      else if (solver >= 1) { if (threadIdx.x < 64) chol_solve_wave(EnvMat{blk, rhs, rowbase, n}, Ld, n, &fail, first, reach, act_rows[0]); }
      else chol_solve_blocked(EnvMat{blk, rhs, rowbase, n}, Ld, n, &fail, first, reach);
      xrow = rhs;
    } else {
      for (int b = threadIdx.x; b < P; b += blockDim.x) env[b] = 0x7fffffff;
      double* A = chol_global;
      __syncthreads();
      envelope_reach(first, reach, P);
      BA_PROBE(1);
      chol_solve_blocked(DenseMat<long long>{A, n}, A + static_cast<long long>(n) * n + n, n, &fail, first, reach);
      xrow = A + static_cast<long long>(n) * n;
    }
  } else {
    double* A = reinterpret_cast<double*>(smem + 16);                              // (n+1) x n: system, then the rhs row
    for (int b = threadIdx.x; b < P; b += blockDim.x) first[b] = b;
    __syncthreads();
    // (eight loads in flight per thread: one at a time, behind the zeroing store of the previous one, the 1806 entries of a
    // 7-pose system cost eight serial L2 round trips - a third of this kernel)
    const int N = n * n + n;
    for (int base = 0; base < N; base += 8 * blockDim.x) {
      long long raw[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * blockDim.x + threadIdx.x;
        raw[u] = idx < N ? sys[idx] : 0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * blockDim.x + threadIdx.x;
        if (idx >= N) continue;
        double v = static_cast<double>(raw[u]) * kInvFix;       // fixed point -> fp64
        sys[idx] = 0LL;                                // ready for the next Gauss-Newton step's accumulation
        if (idx < n * n) {
          const int r = idx / n, c = idx - r * n;
          if (r == c) v += static_cast<double>(ep) + static_cast<double>(lm) * v;   // droid_kernels.cu:1176
          if (raw[u] != 0 && c < r) atomicMin(&first[r / 6], c / 6);
        }
        A[idx] = v;
      }
    }
    __syncthreads();
    envelope_reach(first, reach, P);
    BA_PROBE(1);
    if (solver == 2 && pipe_lists(first, reach, P, act_rows[0], &pipe_ctl))
      chol_solve_pipe(DenseMat<int>{A, n}, A + n * n + n, n, &fail, first, reach, act_rows[0], &pipe_ctl);
    else if (solver >= 1) { if (threadIdx.x < 64) chol_solve_wave(DenseMat<int>{A, n}, A + n * n + n, n, &fail, first, reach, act_rows[0]); }
    else chol_solve_blocked(DenseMat<int>{A, n}, A + n * n + n, n, &fail, first, reach);
    xrow = A + n * n;
  }
  __syncthreads();
  BA_PROBE(4);
  const int failed = fail | meta[4] | meta[2];      // (meta[2]: eta's row count != K - the whole step is a no-op, see ba_plan_kernel)
  for (int idx = threadIdx.x; idx < n; idx += blockDim.x) {
    const float v = failed ? 0.0f : static_cast<float>(xrow[idx]);    // zeros on failure (:1186-1189)
    dx_ws[idx] = v;
    if (dx_out) dx_out[idx] = v;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < P; p += blockDim.x) {               // pose_retr_kernel (:877-910)
    float xi[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) xi[c] = dx_ws[6 * p + c];
    float* ps = poses + 7 * static_cast<long long>(t0 + p);
    const Pose T = retract(xi, load_pose(ps));
    ps[0] = T.t.x; ps[1] = T.t.y; ps[2] = T.t.z;
    ps[3] = T.q.x; ps[4] = T.q.y; ps[5] = T.q.z; ps[6] = T.q.w;
  }
  __syncthreads();
  BA_PROBE(5);
  if (threadIdx.x == 0) {
    meta[4] = 0;
    if (failed) meta[1] = 1;
    if (status_out) { status_out[0] = meta[1]; status_out[1] = meta[0]; status_out[2] = meta[2]; status_out[3] = meta[3]; }
  }
}

// ---- the PARTITIONED pose solve (systems in the compact envelope form: the global bundle adjustment) ----------------------
// The factorisation is a serial chain of P block columns (chol_solve_pipe: ~4 k cycles each), replicated on every rank of an
// edge-sharded run.  A keyframe graph's pose system is block-banded away from its loop closures, and a banded chain can be
// eliminated from BOTH ends at once (a "twisted" factorisation, the two-way case of nested dissection): ba_prepare_kernel
// picks poses m <= s (choose_partition) such that nothing below m couples with anything from s on; then
//   workgroup 0   loads block rows [0, s), eliminates [0, m), waits for workgroup 1's terms on the separator [m, s), adds them,
//                 eliminates the separator, substitutes back, hands the separator's solution over, retracts poses [0, s);
//   workgroup 1   loads block rows [m, P) REVERSED (local row l = pose P - 1 - l, blocks transposed: eliminating from the far
//                 end is the ordinary factorisation of the reversed matrix, whose envelope is the column envelope `reach`),
//                 with the separator x separator blocks and the separator's right-hand side ZERO, eliminates its interior
//                 [s, P) - what is left in the separator's blocks is exactly its Schur-complement term -, writes that to
//                 global memory, waits for the separator's solution, substitutes back, retracts poses [s, P).
// Both run chol_pipe_range / pipe_backsub, i.e. the pipelined factorisation of the whole chain on their part; the order
// of elimination differs from the one-chain solve, so the result agrees with it to fp64 rounding, not bit for bit - and it
// is the same on every rank, because every rank factorises the same (all-reduced, integer) system with the same partition.
// Hand-over through global memory: data, agent-scope release fence, flag; the reader spins on the flag (bounded), acquires.
// Workgroups 0 and 1 of a dispatch are resident together (the first two to be placed), so the wait cannot deadlock; if a
// limit is hit anyway the solve reports failure (zero update) instead of hanging.
// (-DPVO_BA_PROBE, tools/ba_solve_timeline.py: the constant-rate 100 MHz clock - the two workgroups sit on different CUs -
// at the phases of either part, slots 32 + 16 * part + k)
#ifdef PVO_BA_PROBE
#define TWIN_PROBE(k) do { if (g_ba_probe && threadIdx.x == 0) g_ba_probe[32 + 16 * blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TWIN_PROBE(k)
#endif
constexpr int kTwinSpinLimit = 1 << 18;
__device__ __forceinline__ int twin_wait(int* flag) {                      // thread 0: 1 = data ready, 2 = the other side failed / timeout
  int v = 0, spins = 0;
  while ((v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
    if (++spins > kTwinSpinLimit) return 2;
    __builtin_amdgcn_s_sleep(8);
  }
  return v;
}

__global__ __launch_bounds__(256) void ba_solve_twin_kernel(
    double* __restrict__ chol_global, float* __restrict__ poses, float* __restrict__ dx_ws, float* __restrict__ dx_out,
    int* __restrict__ meta, int* __restrict__ status_out, int P, int t0, int* __restrict__ env, long long lds_budget,
    double* __restrict__ xchg, Riders riders) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (blockIdx.x > 1) {
    ride(smem, riders, blockIdx.x - 2);
    return;
  }
  int* xi = reinterpret_cast<int*>(xchg);                                  // 0: terms ready  1: solution ready  2: m  3: s
  const int m = xi[2], s = xi[3];
  const bool split = m > 0, bottom = blockIdx.x == 1;
  if (bottom && !split) return;
  int& fail = *reinterpret_cast<int*>(smem);
  __shared__ int first[kMaxEnvBlocks];
  __shared__ int reach[kMaxEnvBlocks];
  __shared__ unsigned short act_rows[2][kMaxActiveRows];
  __shared__ PipeCtl pipe_ctl;
  __shared__ int blocks_s, got_s;
  const int meta4 = meta[4] | meta[2];      // (meta[2]: eta's row count != K - the whole step is a no-op)
  if (threadIdx.x == 0) {
    fail = 0;
    pipe_ctl.panel = -1; pipe_ctl.done[0] = pipe_ctl.done[1] = pipe_ctl.done[2] = -1; pipe_ctl.abort = 0; pipe_ctl.total = 0;
  }
  const int n = 6 * P;
  TWIN_PROBE(0);
  // the global tables: first[] and the offsets of the compact layout ba_prepare_kernel wrote, the column envelope
  int* tmp = reinterpret_cast<int*>(smem + 16);
  env_layout(env, P, first, tmp, &blocks_s);
  const int blocks = blocks_s;
  if (!split) {
    // one chain (what ba_solve_kernel does with a compact system)
    double* blk = reinterpret_cast<double*>(smem + 16);
    double* rhs = blk + blocks;
    double* Ld = rhs + n;
    int* rowbase = reinterpret_cast<int*>(Ld + 27 * P + 24);
    const bool compact = env_lds_bytes(blocks, n, P) <= lds_budget;
    double* xrow;
    if (compact) {
      int rb_keep[kMaxEnvBlocks / 256];
#pragma unroll
      for (int q = 0; q < kMaxEnvBlocks / 256; ++q) {
        const int b = q * 256 + threadIdx.x;
        rb_keep[q] = b < P ? tmp[b] : 0;
      }
      __syncthreads();
      const int nd2 = (blocks + n) >> 1;
      const double2* src = reinterpret_cast<const double2*>(chol_global);
      double2* dst = reinterpret_cast<double2*>(blk);
      for (int base = 0; base < nd2; base += 16 * blockDim.x) {
        double2 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int i = base + u * blockDim.x + threadIdx.x; v[u] = src[i < nd2 ? i : nd2 - 1]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int i = base + u * blockDim.x + threadIdx.x; if (i < nd2) dst[i] = v[u]; }
      }
      if (((blocks + n) & 1) && threadIdx.x == 0) blk[blocks + n - 1] = chol_global[blocks + n - 1];
#pragma unroll
      for (int q = 0; q < kMaxEnvBlocks / 256; ++q) {
        const int b = q * 256 + threadIdx.x;
        if (b < P) rowbase[b] = rb_keep[q] - 36 * first[b];
      }
      for (int b = threadIdx.x; b < P; b += blockDim.x) env[b] = 0x7fffffff;
      __syncthreads();
      envelope_reach(first, reach, P);
      if (pipe_lists(first, reach, P, act_rows[0], &pipe_ctl))
        chol_solve_pipe(EnvMat{blk, rhs, rowbase, n}, Ld, n, &fail, first, reach, act_rows[0], &pipe_ctl);
      else if (threadIdx.x < 64) chol_solve_wave(EnvMat{blk, rhs, rowbase, n}, Ld, n, &fail, first, reach, act_rows[0]);
      xrow = rhs;
    } else {
      for (int b = threadIdx.x; b < P; b += blockDim.x) env[b] = 0x7fffffff;
      double* A = chol_global;
      __syncthreads();
      envelope_reach(first, reach, P);
      chol_solve_blocked(DenseMat<long long>{A, n}, A + static_cast<long long>(n) * n + n, n, &fail, first, reach);
      xrow = A + static_cast<long long>(n) * n;
    }
    __syncthreads();
    const int failed = fail | meta4;
    for (int idx = threadIdx.x; idx < n; idx += blockDim.x) {
      const float v = failed ? 0.0f : static_cast<float>(xrow[idx]);
      dx_ws[idx] = v;
      if (dx_out) dx_out[idx] = v;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
      float xi6[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) xi6[c] = dx_ws[6 * p + c];
      float* ps = poses + 7 * static_cast<long long>(t0 + p);
      const Pose T = retract(xi6, load_pose(ps));
      ps[0] = T.t.x; ps[1] = T.t.y; ps[2] = T.t.z;
      ps[3] = T.q.x; ps[4] = T.q.y; ps[5] = T.q.z; ps[6] = T.q.w;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      meta[4] = 0;
      if (failed) meta[1] = 1;
      if (status_out) { status_out[0] = meta[1]; status_out[1] = meta[0]; status_out[2] = meta[2]; status_out[3] = meta[3]; }
    }
    return;
  }

  // ---- partitioned
  envelope_reach(first, reach, P);                                         // column envelope of the whole system
  const int w = s - m, Pl = bottom ? P - m : s, nl = 6 * Pl;
  // the global tables move to the end of the dynamic segment (the image grows from its start; choose_partition left room)
  int* tail = reinterpret_cast<int*>(smem + ((lds_budget - 16LL * (P + 2)) & ~15LL));
  int* gfirst = tail, *grow = tail + (P + 2), *lrow = tail + 2 * (P + 2), *greach = tail + 3 * (P + 2);
  {
    int keep[4][kMaxEnvBlocks / 256];
#pragma unroll
    for (int q = 0; q < kMaxEnvBlocks / 256; ++q) {
      const int b = q * 256 + threadIdx.x;
      keep[0][q] = b < P ? first[b] : 0; keep[1][q] = b < P ? tmp[b] : 0; keep[2][q] = b < P ? reach[b] : 0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kMaxEnvBlocks / 256; ++q) {
      const int b = q * 256 + threadIdx.x;
      if (b < P) { gfirst[b] = keep[0][q]; grow[b] = keep[1][q]; greach[b] = keep[2][q]; }
    }
    if (threadIdx.x == 0) grow[P] = blocks;
    __syncthreads();
  }
  // this part's envelope: the separator's rows are dense inside the separator (the other part's elimination fills them)
  for (int l = threadIdx.x; l < Pl; l += blockDim.x) {
    int f;
    if (!bottom) f = l < m ? gfirst[l] : min(gfirst[l], m);
    else { const int g = P - 1 - l; f = P - 1 - greach[g]; if (l >= P - s) f = min(f, P - s); }
    first[l] = f;
  }
  __syncthreads();
  const int lblocks = env_layout(first, Pl, first, lrow, &blocks_s);       // (reads first[b] as the envelope, writes min(first[b], b) back)
  double* blk = reinterpret_cast<double*>(smem + 16);
  double* rhs = blk + lblocks;
  double* Ld = rhs + nl;
  int* rowoff = reinterpret_cast<int*>(Ld + 27 * Pl + 24);
  const bool room = reinterpret_cast<unsigned char*>(rowoff + Pl + 2) <= reinterpret_cast<unsigned char*>(tail);
  if (!room) {                                                             // (choose_partition checked a bound of this: not reached)
    if (threadIdx.x == 0) fail = 1;
  } else {
    for (int i = threadIdx.x; i < lblocks + nl; i += blockDim.x) blk[i] = 0.0;
    for (int l = threadIdx.x; l < Pl; l += blockDim.x) rowoff[l] = lrow[l] - 36 * first[l];
  }
  __syncthreads();
  TWIN_PROBE(1);
  if (room) {
    // source: whole block rows of the compact image, [0, s) for the top part, [s, P) for the bottom part.  Rows [0, m) keep
    // their layout: a straight copy.  Everything else is placed pair by pair (offsets are multiples of 36, so a 16-byte pair
    // never straddles a block): sixteen loads in flight per thread, the row search and the index arithmetic behind them.
    const double2* src2 = reinterpret_cast<const double2*>(chol_global);
    if (!bottom) {
      const int nd2 = grow[m] >> 1;
      double2* dst2 = reinterpret_cast<double2*>(blk);
      for (int base = 0; base < nd2; base += 16 * blockDim.x) {
        double2 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int i = base + u * blockDim.x + threadIdx.x; v[u] = src2[i < nd2 ? i : nd2 - 1]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int i = base + u * blockDim.x + threadIdx.x; if (i < nd2) dst2[i] = v[u]; }
      }
    }
    const int p0 = (bottom ? grow[s] : grow[m]) >> 1, p1 = (bottom ? blocks : grow[s]) >> 1;
    for (int base = p0; base < p1; base += 16 * blockDim.x) {
      double2 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { const int i = base + u * blockDim.x + threadIdx.x; v[u] = src2[i < p1 ? i : p1 - 1]; }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int i2 = base + u * blockDim.x + threadIdx.x;
        if (i2 >= p1) continue;
        const int i = 2 * i2;
        int lo = bottom ? s : m, hi = bottom ? P : s;                      // row rb with grow[rb] <= i < grow[rb + 1]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (grow[mid] <= i) lo = mid; else hi = mid; }
        const int rb = lo, k = i - grow[rb], kb36 = k / 36, e = k - 36 * kb36, cb = gfirst[rb] + kb36;
        if (!bottom) {
          *reinterpret_cast<double2*>(blk + rowoff[rb] + 36 * cb + e) = v[u];
        } else {
          const int ri = e / 6, ci = e - 6 * ri;                           // (e even: ci and ci + 1 lie in one row)
          const int l = P - 1 - cb, lp = P - 1 - rb;                       // block (rb, cb) -> block (l, lp), transposed
          double* dst = blk + rowoff[l] + 36 * lp + 6 * ci + ri;
          dst[0] = v[u].x; dst[6] = v[u].y;
        }
      }
    }
    if (!bottom) { for (int i = threadIdx.x; i < 6 * s; i += blockDim.x) rhs[i] = chol_global[blocks + i]; }
    else { for (int i = threadIdx.x; i < 6 * (P - s); i += blockDim.x) { const int l = i / 6; rhs[i] = chol_global[blocks + 6 * (P - 1 - l) + (i - 6 * l)]; } }
  }
  __syncthreads();
  TWIN_PROBE(2);
  bool lists = false;
  if (room) {
    envelope_reach(first, reach, Pl);
    lists = pipe_lists(first, reach, Pl, act_rows[0], &pipe_ctl);
    if (!lists && threadIdx.x == 0) fail = 1;
  }
  __syncthreads();
  const EnvMat A{blk, rhs, rowoff, nl};
  TWIN_PROBE(3);
  double* cterm = xchg + 2;                                                // [w][w][36] rows a >= c, then the rhs [6 w], then x [6 w]
  double* crhs = cterm + 36 * kMaxSep * kMaxSep;
  double* xsep = crhs + 6 * kMaxSep;
  if (!bottom) {
    if (!fail) chol_pipe_range(A, Ld, Pl, &fail, reach, act_rows[0], &pipe_ctl, 0, m);
    __syncthreads();
    TWIN_PROBE(4);
    if (threadIdx.x == 0) got_s = twin_wait(&xi[0]);                       // (always: workgroup 1 reads `env`, reset below)
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (got_s != 1 && threadIdx.x == 0) fail = 1;
    TWIN_PROBE(5);
    for (int b = threadIdx.x; b < P; b += blockDim.x) env[b] = 0x7fffffff;
    __syncthreads();
    if (!fail) {
      for (int t = threadIdx.x; t < 36 * w * w; t += blockDim.x) {
        const int pr = t / 36, e = t - 36 * pr, a = pr / w, c = pr - a * w;
        if (c > a) continue;
        blk[rowoff[m + a] + 36 * (m + c) + e] += cterm[(a * kMaxSep + c) * 36 + e];
      }
      for (int t = threadIdx.x; t < 6 * w; t += blockDim.x) rhs[6 * m + t] += crhs[t];
    }
    __syncthreads();
    TWIN_PROBE(6);
    if (!fail) chol_pipe_range(A, Ld, Pl, &fail, reach, act_rows[0], &pipe_ctl, m, s);
    __syncthreads();
    TWIN_PROBE(7);
    if (!fail) pipe_backsub(A, Ld, first, Pl - 1, Pl, m);                   // the separator's unknowns first: the other part waits for them
    __syncthreads();
    TWIN_PROBE(8);
    if (threadIdx.x < 6 * w) xsep[threadIdx.x] = fail ? 0.0 : rhs[6 * m + threadIdx.x];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&xi[1], fail ? 2 : 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    TWIN_PROBE(9);
    if (!fail) pipe_backsub(A, Ld, first, m - 1, Pl);
  } else {
    if (!fail) chol_pipe_range(A, Ld, Pl, &fail, reach, act_rows[0], &pipe_ctl, 0, P - s);
    __syncthreads();
    TWIN_PROBE(4);
    if (!fail) {
      for (int t = threadIdx.x; t < 36 * w * w; t += blockDim.x) {
        const int pr = t / 36, e = t - 36 * pr, a = pr / w, c = pr - a * w;
        if (c > a) continue;
        const int ri = e / 6, ci = e - 6 * ri;
        const int l = P - 1 - (m + c), lp = P - 1 - (m + a);               // global block (m + a, m + c), a >= c  <-  local (l, lp), l >= lp
        cterm[(a * kMaxSep + c) * 36 + e] = blk[rowoff[l] + 36 * lp + (a == c ? 6 * ri + ci : 6 * ci + ri)];
      }
      for (int t = threadIdx.x; t < 6 * w; t += blockDim.x) { const int a = t / 6; crhs[t] = rhs[6 * (P - 1 - (m + a)) + (t - 6 * a)]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_store(&xi[0], fail ? 2 : 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      TWIN_PROBE(5);
      got_s = twin_wait(&xi[1]);
    }
    __syncthreads();
    TWIN_PROBE(6);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (got_s != 1 && threadIdx.x == 0) fail = 1;
    __syncthreads();
    if (!fail) {
      for (int t = threadIdx.x; t < 6 * w; t += blockDim.x) { const int a = t / 6; rhs[6 * (P - 1 - (m + a)) + (t - 6 * a)] = xsep[t]; }
    }
    __syncthreads();
    if (!fail) pipe_backsub(A, Ld, first, Pl - 1, P - s);
  }
  __syncthreads();
  TWIN_PROBE(10);
  const int failed = fail | meta4;
  const int g0 = bottom ? s : 0, g1 = bottom ? P : s;                      // the poses this part owns
  for (int idx = 6 * g0 + threadIdx.x; idx < 6 * g1; idx += blockDim.x) {
    const int g = idx / 6, c = idx - 6 * g;
    const float v = failed ? 0.0f : static_cast<float>(rhs[bottom ? 6 * (P - 1 - g) + c : idx]);
    dx_ws[idx] = v;
    if (dx_out) dx_out[idx] = v;
  }
  __syncthreads();
  for (int p = g0 + threadIdx.x; p < g1; p += blockDim.x) {               // pose_retr_kernel (:877-910)
    float xi6[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) xi6[c] = dx_ws[6 * p + c];
    float* ps = poses + 7 * static_cast<long long>(t0 + p);
    const Pose T = retract(xi6, load_pose(ps));
    ps[0] = T.t.x; ps[1] = T.t.y; ps[2] = T.t.z;
    ps[3] = T.q.x; ps[4] = T.q.y; ps[5] = T.q.z; ps[6] = T.q.w;
  }
  __syncthreads();
  TWIN_PROBE(11);
  if (threadIdx.x == 0 && !bottom) {
    meta[4] = 0;
    if (failed) meta[1] = 1;
    if (status_out) { status_out[0] = meta[1]; status_out[1] = meta[0]; status_out[2] = meta[2]; status_out[3] = meta[3]; }
  }
}

// ---- the DENSE pose solve on the fp64 matrix cores (round 6): windows up to kDenseMaxPoses free poses, one launch ---------------
// A frontend window's pose system is DENSE: 21-25 free poses whose ~400 inactive edges couple almost every pair
// (factor_graph.py:281-291; bench.py `sequence.ba_windows_sampled`).  The envelope forms above gain nothing there and paid for it:
// beyond 21 poses env + prepare + partitioned solve = three launches, 16 + 100 us per Gauss-Newton step, the partition finding no
// separator and one workgroup walking a 25-column chain at ~9 k cycles per block column - most of them the rank-6 trailing update
// read and written through LDS (27 LDS operations per 36 multiply-adds).
// Here the whole lower triangle lives in the REGISTERS of one workgroup, as 16 x 16 accumulator tiles of
// v_mfma_f64_16x16x4_f64 dealt round-robin over the four waves (a 26-pose system: 55 tiles, 14 per wave, 112 VGPRs):
//   per step of EIGHT columns (20 steps for 156 rows; a panel never straddles a tile column):
//     1. the tiles of the panel's tile column write its eight columns to the panel's place in LDS (+ the 8 x 8 diagonal block apart)
//     2. one lane per row: the 8 x 8 Cholesky redundantly in registers, then its own row of the panel (forward substitution
//        against the diagonal block) - in place; the finished panels ARE the factor L, kept for the back-substitution
//     3. every tile right of / below the panel: C -= L_rows(i) L_rows(j)^T, two MFMAs (K = 8), operands straight from the panel
//   two barriers per step; the trailing matrix never touches LDS.  The right-hand side rides as row n of the matrix (as in the
//   forms above), so the forward substitution is part of step 2; the back-substitution is wave 0's: one lane per column, the
//   panel's eight unknowns by v_readlane, no LDS exchange of the vector.
// Loads the fixed-point system itself (lower block triangle + right-hand side; zeroes ALL of it for the next step), damping as
// droid_kernels.cu:1176, failure -> zero update (:1186-1189).  Deterministic: a fixed order of fp64 operations on an integer
// system, so every rank of an edge-sharded run that takes this path gets the same bits; against the envelope forms the order of
// the additions differs: equal to fp64 rounding, not bit for bit (tests: oracle parity at 1e-4 in fp32 outputs, and agreement
// with the pipelined form to 4e-7 relative).
constexpr int kDenseMaxPoses = 29;      // N = 16 ceil((6 P + 1) / 16) <= 176: 66 tiles (17 per wave), 129.5 KB of panels in LDS
typedef double pvo_d4 __attribute__((ext_vector_type(4)));

__host__ __device__ __forceinline__ int dense_N(int n) { return ((n + 1 + 15) / 16) * 16; }
__host__ __device__ __forceinline__ int dense_panel_off(int kb8, int N) { return kb8 * N - 4 * kb8 * (kb8 - 1); }      // rows in front of panel kb8
__host__ __forceinline__ size_t dense_lds_bytes(int n) {
  const int N = dense_N(n), nb = N / 8;
  return 16 + sizeof(double) * (8 * static_cast<size_t>(dense_panel_off(nb, N)) + 64 + 2 * static_cast<size_t>(N));
}

#ifdef PVO_BA_PROBE
#define DENSE_STEP_PROBE(slot) do { if (g_ba_probe && threadIdx.x == 0 && kb8 == nb / 2) g_ba_probe[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define DENSE_STEP_PROBE(slot)
#endif
// element (row t of a panel, column k of its eight) sits at t * 8 + (k ^ ((t >> 1) & 7)): the matrix-core operand reads take ONE
// column of 16 consecutive rows per 16 lanes - at a plain 64-byte row pitch that is an 8-way bank conflict on every read
__device__ __forceinline__ int dense_swz(int t, int k) { return t * 8 + (k ^ ((t >> 1) & 7)); }
// 1 / sqrt(d): hardware estimate + ONE Newton step (the estimate carries ~26 bits: ~1.5 ulp after the step; the outputs are fp32).
// The second step of rsqrt_nr was 4 of the ~12 dependent fp64 operations per column on this kernel's critical chain.
__device__ __forceinline__ double rsqrt_nr1(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  return y * (1.5 - 0.5 * d * y * y);
}

template <int SLOTS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void ba_solve_dense_kernel(
    long long* __restrict__ sys, const long long* __restrict__ msg, const int* __restrict__ first_s,
    float* __restrict__ poses, float* __restrict__ dx_ws, float* __restrict__ dx_out,
    int* __restrict__ meta, int* __restrict__ status_out, int P, int t0, float lm, float ep, Riders riders) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [16 B flags | panels | diagonal block | 1 / L_jj | x]
  if (blockIdx.x > 0) {                                                  // rider workgroups (launched only with riders)
    ride(smem, riders, blockIdx.x - 1);
    return;
  }
  // msg != NULL: the system arrives as the packed envelope message of an edge-sharded step (ba_pack_kernel's layout: block row b =
  // blocks first[b] .. b of 36, then the right-hand side) instead of the dense image - same values into the same tiles, so a
  // sharded run gets the bits of the whole graph on one GPU at these sizes too
  __shared__ int mfirst[kDenseMaxPoses + 1], mbase[kDenseMaxPoses + 2];
  if (msg) {
    if (threadIdx.x == 0) {
      int run = 0;
      for (int b = 0; b < P; ++b) {
        const int e = first_s[b];
        const int f = e < b ? (e < 0 ? 0 : e) : b;                       // (clamped as env_layout does)
        mfirst[b] = f; mbase[b] = run; run += (b - f + 1) * 36;
      }
      mbase[P] = run;
    }
    __syncthreads();
  }
  int& fail = *reinterpret_cast<int*>(smem);
  const int n = 6 * P, N = dense_N(n), T = N / 16, nb = (n + 7) / 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform: the tile tables below live in scalar registers)
  const int lr = lane >> 4, lc = lane & 15;                              // C / D: row lr + 4 r, column lc; A / B: index lc, k = lr
  double* Lp = reinterpret_cast<double*>(smem + 16);
  double* Dg = Lp + 8 * dense_panel_off(N / 8, N);
  double* rdg = Dg + 64;
  double* xs = rdg + N;
  if (tid == 0) fail = 0;
  BA_PROBE(0);
#ifdef PVO_BA_PROBE
  if (g_ba_probe && tid == 0) g_ba_probe[15] = 0xD15E;                    // (tools/ba_solve_timeline.py: the step stamps below are this kernel's)
#endif

  // ---- this wave's tiles: the lower triangle's tiles in COLUMN-major order (tile column 0 top to bottom, then column 1, ...),
  // q = 4 slot + wave.  The tiles a step still has to update are then a SUFFIX of every wave's slots and the tiles of one tile
  // column a short run of them: the unrolled slot code is entered through one jump instead of one branch per slot
  int tti[SLOTS], ttj[SLOTS];
  const int ntiles = T * (T + 1) / 2;
  auto colstart = [&](int c) { return c * T - c * (c - 1) / 2; };         // index of tile (c, c)
  auto first_slot = [&](int c) { const int q0 = colstart(c) - wave; return q0 > 0 ? (q0 + 3) >> 2 : 0; };      // this wave's first slot with tj >= c
#pragma unroll
  for (int sl = 0; sl < SLOTS; ++sl) {
    const int q = 4 * sl + wave;
    int tj = 0;
    while (tj + 1 < T && colstart(tj + 1) <= q) ++tj;
    tti[sl] = q < ntiles ? tj + (q - colstart(tj)) : -1;
    ttj[sl] = tj;
  }
  const int nsl = ntiles > wave ? (ntiles - wave + 3) >> 2 : 0;           // this wave's slots in use
  // ---- load: fixed point -> fp64 (+ damping on the diagonal, droid_kernels.cu:1176) straight into the accumulator tiles.  Every
  // load unconditional on a clamped index (a guarded load is a branch of its own; these are 32 in flight per lane)
  pvo_d4 C[SLOTS];
  auto load_tiles = [&](auto from_message) {
    constexpr bool MSG = decltype(from_message)::value;
    const long long* __restrict__ src = MSG ? msg : sys;
    constexpr int B = MSG ? 4 : SLOTS;       // slots per batch of loads: the dense image's addresses are cheap - all 4 SLOTS loads in flight at once (a round trip to the
                                             // memory side, where the atomics left the system, is ~3 us: four batches were 14 us of this kernel)
#pragma unroll
    for (int s0 = 0; s0 < SLOTS; s0 += B) {
      long long raw[B][4];
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const int sl = s0 + u < SLOTS ? s0 + u : SLOTS - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * tti[sl] + lr + 4 * r, col = 16 * ttj[sl] + lc;
          int idx = -1;
          const bool in = tti[sl] >= 0 && col < n;
          if (MSG) {
            const int rw = row < n ? row : 0;
            const int rb = rw / 6, cb = col / 6;
            const int body = mbase[rb] + (cb - mfirst[rb]) * 36 + (rw - 6 * rb) * 6 + (col - 6 * cb);
            idx = (in && row < n && col <= row && cb >= mfirst[rb]) ? body : ((in && row == n) ? mbase[P] + col : -1);
          } else {
            idx = (in && row < n && col <= row) ? row * n + col : ((in && row == n) ? n * n + col : -1);
          }
          const long long v = src[idx < 0 ? 0 : idx];
          raw[u][r] = idx < 0 ? 0LL : v;
        }
      }
#pragma unroll
      for (int u = 0; u < B; ++u) {
        if (s0 + u >= SLOTS) continue;
        const int sl = s0 + u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * tti[sl] + lr + 4 * r, col = 16 * ttj[sl] + lc;
          double v = static_cast<double>(raw[u][r]) * kInvFix;
          if (row == col && row < n) v += static_cast<double>(ep) + static_cast<double>(lm) * v;
          C[sl][r] = v;
        }
      }
    }
  };
  if (msg) load_tiles(std::true_type{}); else load_tiles(std::false_type{});
  // the first panel's columns out of the accumulators (afterwards every step extracts the NEXT panel behind its own update)
  auto extract = [&](int kb8x) {
    const int tcx = kb8x >> 1, hx = kb8x & 1, c0x = 8 * kb8x;
    double* panx = Lp + 8 * dense_panel_off(kb8x, N);
    const int e0 = first_slot(tcx), e1x = first_slot(tcx + 1), e1 = e1x < nsl ? e1x : nsl;
    auto one = [&](const pvo_d4& Cs, int ti) {
      if ((lc >> 3) == hx) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * ti + lr + 4 * r;
          if (row >= c0x) {
            panx[dense_swz(row - c0x, lc & 7)] = Cs[r];
            if (row < c0x + 8) Dg[(row - c0x) * 8 + (lc & 7)] = Cs[r];
          }
        }
      }
    };
    switch (e0) {                                                         // (the column's slots are consecutive: one jump, then straight down)
#define PVO_DENSE_CASE(K) case K: if (K < SLOTS) { if (K >= e1) break; one(C[K < SLOTS ? K : 0], tti[K < SLOTS ? K : 0]); } [[fallthrough]];
      PVO_DENSE_CASE(0) PVO_DENSE_CASE(1) PVO_DENSE_CASE(2) PVO_DENSE_CASE(3) PVO_DENSE_CASE(4) PVO_DENSE_CASE(5)
      PVO_DENSE_CASE(6) PVO_DENSE_CASE(7) PVO_DENSE_CASE(8) PVO_DENSE_CASE(9) PVO_DENSE_CASE(10) PVO_DENSE_CASE(11)
      PVO_DENSE_CASE(12) PVO_DENSE_CASE(13) PVO_DENSE_CASE(14) PVO_DENSE_CASE(15) PVO_DENSE_CASE(16)
#undef PVO_DENSE_CASE
      default: break;
    }
  };
  extract(0);
  // (every entry the assembly / Schur kernels or an all-reduce may have written: ready for the next step's accumulation.  Behind a
  // barrier: a wave with few tiles would otherwise zero entries another wave has not loaded yet)
  __syncthreads();
  if (!msg) for (int idx = tid; idx < n * n + n; idx += 256) sys[idx] = 0LL;      // (ba_pack_kernel has zeroed the image a message came from)
  BA_PROBE(1);

  // ---- factorisation, eight columns per step.  At the head of a step the panel's columns (as the updates so far left them) and
  // its diagonal block are in LDS.
  for (int kb8 = 0; kb8 < nb; ++kb8) {
    const int tc = kb8 >> 1, h = kb8 & 1, c0 = 8 * kb8;
    double* pan = Lp + 8 * dense_panel_off(kb8, N);                      // rows c0 .. N-1 of columns c0 .. c0+7, 8 doubles per row (swizzled)
    DENSE_STEP_PROBE(8);
    // 2. one lane per row of the panel
    const int nv = (n - c0 < 8) ? n - c0 : 8;                           // columns of this panel that belong to the system (the rest: padding)
    if (tid < N - c0) {
      // the 8 x 8 Cholesky IN PLACE on the block's lower triangle; a diagonal slot ends up holding 1 / L_jj (L_jj itself is never
      // needed: the diagonal rows leave as d * rs through the row formula below) - registers are what bounds this kernel's occupancy
      double L[36], a[8];
      const int sw = (tid >> 1) & 7;
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = pan[tid * 8 + (j ^ sw)];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) L[i * (i + 1) / 2 + j] = Dg[i * 8 + j];
      }
      bool ok = true;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        double d = L[j * (j + 1) / 2 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j * (j + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
        const bool real = j < nv;
        ok = ok && (!real || d > 0.0);
        const double rs = (real && ok) ? rsqrt_nr1(d) : 0.0;
        L[j * (j + 1) / 2 + j] = rs;
#pragma unroll
        for (int i = j + 1; i < 8; ++i) {
          double v = L[i * (i + 1) / 2 + j];
#pragma unroll
          for (int k = 0; k < j; ++k) v -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
          L[i * (i + 1) / 2 + j] = v * rs;
        }
        // this row's entry of column j as soon as the column's reciprocal pivot exists (the row's chain runs beside the block's)
        double v = a[j];
#pragma unroll
        for (int k = 0; k < j; ++k) v -= a[k] * L[j * (j + 1) / 2 + k];
        v *= rs;
        if (tid < 8 && j > tid) v = 0.0;                                 // (the diagonal block's rows: L is lower triangular)
        a[j] = v;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) pan[tid * 8 + (j ^ sw)] = a[j];
      if (tid < 8) {
        double r = L[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) r = (tid == j) ? L[j * (j + 1) / 2 + j] : r;
        rdg[c0 + tid] = r;
      }
      if (tid == 0 && !ok) fail = 1;
    }
    DENSE_STEP_PROBE(9);
    __syncthreads();
    DENSE_STEP_PROBE(10);
    // 3. trailing update C -= L_rows(i) L_rows(j)^T of every tile with a column beyond the panel: this wave's slots from `u0` on,
    // two slots per block of straight-line code (eight operand reads in flight, the two tiles' MFMA pairs interleaved), entered
    // through one jump
    {
      const int u0 = first_slot(h == 0 ? tc : tc + 1);
      auto pair = [&](auto kc) {
        constexpr int K = decltype(kc)::value;
        constexpr int K1 = K + 1 < SLOTS ? K + 1 : K;
        const bool on0 = K >= u0 && K < nsl, on1 = K + 1 < SLOTS && K + 1 >= u0 && K + 1 < nsl;
        const int ta0 = on0 ? 16 * tti[K] + lc - c0 : 0, tb0 = on0 ? 16 * ttj[K] + lc - c0 : 0;
        const int ta1 = on1 ? 16 * tti[K1] + lc - c0 : 0, tb1 = on1 ? 16 * ttj[K1] + lc - c0 : 0;
        double a00 = pan[dense_swz(ta0, lr)], b00 = pan[dense_swz(tb0, lr)], a01 = pan[dense_swz(ta0, 4 + lr)], b01 = pan[dense_swz(tb0, 4 + lr)];
        double a10 = pan[dense_swz(ta1, lr)], b10 = pan[dense_swz(tb1, lr)], a11 = pan[dense_swz(ta1, 4 + lr)], b11 = pan[dense_swz(tb1, 4 + lr)];
        if (!on0) { a00 = 0.0; a01 = 0.0; }                              // (a slot below u0 in its pair / beyond this wave's tiles: C += 0)
        if (!on1) { a10 = 0.0; a11 = 0.0; }
        C[K] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a00, b00, C[K], 0, 0, 0);
        if (K + 1 < SLOTS) C[K1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a10, b10, C[K1], 0, 0, 0);
        C[K] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a01, b01, C[K], 0, 0, 0);
        if (K + 1 < SLOTS) C[K1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a11, b11, C[K1], 0, 0, 0);
      };
      switch (u0 >> 1) {
#define PVO_DENSE_PAIR(K) case K / 2: if (K < SLOTS) { if (K >= nsl) break; pair(std::integral_constant<int, (K < SLOTS ? K : 0)>{}); } [[fallthrough]];
        PVO_DENSE_PAIR(0) PVO_DENSE_PAIR(2) PVO_DENSE_PAIR(4) PVO_DENSE_PAIR(6) PVO_DENSE_PAIR(8) PVO_DENSE_PAIR(10) PVO_DENSE_PAIR(12)
        PVO_DENSE_PAIR(14) PVO_DENSE_PAIR(16)
#undef PVO_DENSE_PAIR
        default: break;
      }
    }
    DENSE_STEP_PROBE(11);
    if (kb8 + 1 < nb) extract(kb8 + 1);
    DENSE_STEP_PROBE(12);
    DENSE_STEP_PROBE(13);
    __syncthreads();
    DENSE_STEP_PROBE(14);
  }
  BA_PROBE(2);

  // ---- back-substitution L^T x = y (y = row n of L), wave 0: lane holds columns lane, lane + 64, lane + 128
  if (wave == 0) {
    double z[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int c = lane + 64 * m;
      z[m] = c < n ? Lp[8 * dense_panel_off(c >> 3, N) + dense_swz(n - 8 * (c >> 3), c & 7)] : 0.0;
    }
    for (int kb8 = nb - 1; kb8 >= 0; --kb8) {
      const int c0 = 8 * kb8, m0 = c0 >> 6, l0 = c0 & 63;
      const int nv = (n - c0 < 8) ? n - c0 : 8;
      const double* pan = Lp + 8 * dense_panel_off(kb8, N);
      // everything this step reads from LDS is independent of the unknowns: requested up front
      double Lk[28], rdv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        rdv[j] = rdg[c0 + j];
#pragma unroll
        for (int i = j + 1; i < 8; ++i) Lk[i * (i - 1) / 2 + j] = pan[dense_swz(i, j)];      // L[c0 + i][c0 + j]
      }
      const double zsel = m0 == 0 ? z[0] : (m0 == 1 ? z[1] : z[2]);
      double x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int lo = __builtin_amdgcn_readlane(static_cast<int>(__double_as_longlong(zsel) & 0xffffffffLL), l0 + j);
        const int hi = __builtin_amdgcn_readlane(static_cast<int>(__double_as_longlong(zsel) >> 32), l0 + j);
        x[j] = __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
      }
#pragma unroll
      for (int j = 7; j >= 0; --j) {
        double v = x[j];
#pragma unroll
        for (int i = 7; i > j; --i) v -= Lk[i * (i - 1) / 2 + j] * x[i];
        x[j] = (j < nv) ? v * rdv[j] : 0.0;
      }
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const int c = lane + 64 * m;
        if (c < c0) {
          const double* pc = Lp + 8 * dense_panel_off(c >> 3, N);
          double v = z[m];
#pragma unroll
          for (int i = 0; i < 8; ++i) v -= pc[dense_swz(c0 + i - 8 * (c >> 3), c & 7)] * x[i];          // L[c0 + i][c]
          z[m] = v;
        } else if (c < c0 + 8) {
          double v = x[0];
#pragma unroll
          for (int j = 1; j < 8; ++j) v = (c - c0 == j) ? x[j] : v;
          z[m] = v;
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int c = lane + 64 * m;
      if (c < n) xs[c] = z[m];
    }
  }
  __syncthreads();
  BA_PROBE(4);
  const int failed = fail | meta[4] | meta[2];      // (meta[2]: eta's row count != K - the whole step is a no-op, see ba_plan_kernel)
  for (int idx = tid; idx < n; idx += 256) {
    double xv = xs[idx];
    const float v = (failed || !(xv == xv)) ? 0.0f : static_cast<float>(xv);    // zeros on failure (:1186-1189)
    dx_ws[idx] = v;
    if (dx_out) dx_out[idx] = v;
  }
  __syncthreads();
  for (int p = tid; p < P; p += 256) {               // pose_retr_kernel (:877-910)
    float xi[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) xi[c] = dx_ws[6 * p + c];
    float* ps = poses + 7 * static_cast<long long>(t0 + p);
    const Pose Tn = retract(xi, load_pose(ps));
    ps[0] = Tn.t.x; ps[1] = Tn.t.y; ps[2] = Tn.t.z;
    ps[3] = Tn.q.x; ps[4] = Tn.q.y; ps[5] = Tn.q.z; ps[6] = Tn.q.w;
  }
  __syncthreads();
  BA_PROBE(5);
  if (tid == 0) {
    meta[4] = 0;
    if (failed) meta[1] = 1;
    if (status_out) { status_out[0] = meta[1]; status_out[1] = meta[0]; status_out[2] = meta[2]; status_out[3] = meta[3]; }
  }
}

// ---- BLOCKED pose solve over MANY workgroups (round 6): big systems that are not narrow-banded - the backend's global graph ---------
// A global bundle adjustment over ~85 keyframes connected by proximity (10-15 edges per frame) has an envelope of several hundred KB:
// it fits no compute unit's LDS, the partition finds no separator, and the one-workgroup factorisation then ran out of global memory
// - 1.8 ms per Gauss-Newton step, 38 of them in Droid.terminate (bench.py `sequence`, profiles/r06_sequence_timeline.txt).  At that
// point sparsity is not worth a CU's patience: the DENSE right-looking factorisation in 48 x 48 blocks, three launches per block column,
//   dense_panel_kernel   every workgroup factors the 48 x 48 diagonal block redundantly (one wave, rows in registers) and solves its own
//                        row block of the panel against it (one thread per row; the right-hand side rides as row n, so this is the
//                        forward substitution too); the diagonal block's workgroup stores the factor + reciprocal pivots apart
//   dense_update_kernel  A_ij -= L_ik L_jk^T, one workgroup per block pair of the trailing lower triangle (3 x 3 outputs per thread)
//   dense_back_kernel    back-substitution, block column by block column from the end: x_k from the stored diagonal factor (redundantly
//                        per workgroup), y_j -= L_kj^T x_k by the workgroup of column block j
// and dense_finish_kernel (dx, retraction, status; the riders ride here).  ~3 n / 48 + 3 launches, no host synchronisation; 85 poses:
// ~35 launches.  Same damping, failure -> zero update, fixed order of operations (bitwise reproducible); agrees with the other forms
// to fp64 rounding.  Chosen on the host by size and edge density (ba_finish_impl); never for a packed message.
constexpr int kNB = 48;
__global__ __launch_bounds__(256) void dense_panel_kernel(double* __restrict__ A, double* __restrict__ Ldiag, int n, int kb, int* __restrict__ meta) {
  // (everything in LDS, runtime loops: the first version kept a row per lane in 48 registers with every loop unrolled - 512 registers
  // and 8 KB of scratch per thread, 131 us per launch)
  __shared__ double Lk[kNB][kNB + 1];       // the diagonal block: lower triangle, factored in place (the diagonal itself stays, sqrt in ldg)
  __shared__ double Xs[kNB][kNB + 1];       // this workgroup's rows of the panel
  __shared__ double rinv[kNB], ldg[kNB];
  const int tid = threadIdx.x;
  const int c0 = kNB * kb, nc = (n - c0 < kNB) ? n - c0 : kNB;         // valid columns of this block column
  const int rb = kb + blockIdx.x, r0 = kNB * rb;                        // this workgroup's row block
  for (int t = tid; t < kNB * kNB; t += 256) {
    const int r = t / kNB, c = t - r * kNB;
    Lk[r][c] = (r < nc && c <= r) ? A[static_cast<size_t>(c0 + r) * n + c0 + c] : 0.0;
    const int rr = r0 + r;
    Xs[r][c] = (rr <= n && c < nc) ? A[static_cast<size_t>(rr) * n + c0 + c] : 0.0;
  }
  __syncthreads();
  // ---- factorisation of the diagonal block AND substitution of this workgroup's panel rows together, EIGHT columns per step (two
  // barriers per step; one column per step with a thread per entry was 48 x 2 barriers + a 48-step substitution: 53 us per launch):
  //   (a) one thread per row - rows of the diagonal block from the step's first column on, and the 48 panel rows -: the 8 x 8 Cholesky of
  //       the step's diagonal sub-block redundantly in registers, then the row's eight entries by forward substitution against it;
  //   (b) every entry right of the step's columns: E[r][c] -= sum_k row_r[k] L_c[k], a thread per (row, column stripe).
  // (The diagonal block's workgroup and every panel workgroup do the same arithmetic on the diagonal block: redundant, and identical.)
  bool ok = true;
  for (int j0 = 0; j0 < nc; j0 += 8) {
    const int nv = (nc - j0 < 8) ? nc - j0 : 8;
    double L[36], av[8];
    if (tid < 2 * kNB) {
      const bool isx = tid >= kNB;
      const int t = isx ? tid - kNB : tid;
      if (isx || (t >= j0 && t < nc)) {
        double (*Row)[kNB + 1] = isx ? Xs : Lk;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
          for (int j = 0; j <= i; ++j) L[i * (i + 1) / 2 + j] = Lk[j0 + i][j0 + j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) av[j] = Row[t][j0 + j];
      }
    }
    __syncthreads();                       // (the sub-block's own rows are rewritten below by their threads: everybody has read them)
    if (tid < 2 * kNB) {
      const bool isx = tid >= kNB;
      const int t = isx ? tid - kNB : tid;
      if (isx || (t >= j0 && t < nc)) {
        double (*Row)[kNB + 1] = isx ? Xs : Lk;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          double d = L[j * (j + 1) / 2 + j];
#pragma unroll
          for (int k = 0; k < j; ++k) d -= L[j * (j + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
          const bool real = j < nv;
          ok = ok && (!real || d > 0.0);
          const double rs = (real && ok) ? rsqrt_nr(d) : 0.0;
          if (!isx && t == j0 + j) { rinv[j0 + j] = rs; ldg[j0 + j] = d * rs; }
          L[j * (j + 1) / 2 + j] = rs;
#pragma unroll
          for (int i = j + 1; i < 8; ++i) {
            double v = L[i * (i + 1) / 2 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
            L[i * (i + 1) / 2 + j] = v * rs;
          }
          double v = av[j];
#pragma unroll
          for (int k = 0; k < j; ++k) v -= av[k] * L[j * (j + 1) / 2 + k];
          av[j] = v * rs;
        }
        // (a row of the diagonal sub-block itself: entry j == its own column is d * rs = the pivot's square root through the same formula;
        // entries right of its diagonal are not part of L - they are never read again)
#pragma unroll
        for (int j = 0; j < 8; ++j) if (j < nv && (isx || j0 + j <= t)) Row[t][j0 + j] = av[j];
      }
    }
    __syncthreads();
    const int c1 = j0 + 8;
    if (c1 < nc) {
      // rows: the diagonal block's rows r >= c1 (columns c1 .. r), then the 48 panel rows (columns c1 .. nc - 1); 16 column stripes
      const int ty = tid >> 4, tx = tid & 15;
      for (int r = c1 + ty; r < nc + kNB; r += 16) {
        const bool isx = r >= nc;
        double (*Row)[kNB + 1] = isx ? Xs : Lk;
        const int rr = isx ? r - nc : r;
        double a8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a8[k] = Row[rr][j0 + k];
        const int cend = isx ? nc - 1 : rr;
        for (int c = c1 + tx; c <= cend; c += 16) {
          double v = Row[rr][c];
#pragma unroll
          for (int k = 0; k < 8; ++k) v -= a8[k] * Lk[c][j0 + k];
          Row[rr][c] = v;
        }
      }
    }
    __syncthreads();
  }
  if (tid == kNB && !ok) meta[4] = 1;      // non-SPD (the first panel-row thread factors every sub-block): the finish kernel writes the zero update
  if (tid >= nc && tid < kNB) { rinv[tid] = 0.0; ldg[tid] = 0.0; }
  __syncthreads();
  for (int t = tid; t < kNB * kNB; t += 256) {
    const int r = t / kNB, c = t - r * kNB;
    const int rr = r0 + r;
    if (rr <= n && c < nc && (rb > kb || r >= nc)) A[static_cast<size_t>(rr) * n + c0 + c] = Xs[r][c];
  }
  if (blockIdx.x == 0) {                                                 // the diagonal block's workgroup keeps the factor for the back-substitution
    double* Ld = Ldiag + static_cast<size_t>(kb) * (kNB * kNB + kNB);
    for (int t = tid; t < kNB * kNB; t += 256) {
      const int r = t / kNB, c = t - r * kNB;
      Ld[t] = (r < nc && c < r) ? Lk[r][c] : ((r == c && r < nc) ? ldg[r] : 0.0);
    }
    if (tid < kNB) Ld[kNB * kNB + tid] = rinv[tid];
  }
}

__global__ __launch_bounds__(256) void dense_update_kernel(double* __restrict__ A, int n, int kb, int nrb) {
  __shared__ double Li[kNB][kNB + 1], Lj[kNB][kNB + 1];
  // block pair p -> (i, j), kb < j <= i < nrb
  const int m = nrb - kb - 1;
  int p = blockIdx.x, jj = 0;
  while (p >= m - jj) { p -= m - jj; ++jj; }
  const int bj = kb + 1 + jj, bi = bj + p;
  const int tid = threadIdx.x;
  const int c0 = kNB * kb, nc = (n - c0 < kNB) ? n - c0 : kNB;
  for (int t = tid; t < kNB * kNB; t += 256) {
    const int r = t / kNB, c = t - r * kNB;
    const int ri = kNB * bi + r, rj = kNB * bj + r;
    Li[r][c] = (ri <= n && c < nc) ? A[static_cast<size_t>(ri) * n + c0 + c] : 0.0;
    Lj[r][c] = (rj < n && c < nc) ? A[static_cast<size_t>(rj) * n + c0 + c] : 0.0;        // (columns of the target block: rows rj < n)
  }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;                  // 3 x 3 outputs: rows 3 ty .., columns 3 tx ..
  double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll 4
  for (int k = 0; k < kNB; ++k) {
    double a[3], b[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) { a[u] = Li[3 * ty + u][k]; b[u] = Lj[3 * tx + u][k]; }
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int v = 0; v < 3; ++v) acc[u][v] += a[u] * b[v];
  }
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int rr = kNB * bi + 3 * ty + u;
    if (rr > n) continue;
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int cc = kNB * bj + 3 * tx + v;
      if (cc < n && (bi > bj || cc <= rr)) A[static_cast<size_t>(rr) * n + cc] -= acc[u][v];
    }
  }
}

__global__ __launch_bounds__(256) void dense_back_kernel(const double* __restrict__ A, const double* __restrict__ Ldiag, double* __restrict__ yrow,
                                                         double* __restrict__ xvec, int n, int kb) {
  __shared__ double Lk[kNB][kNB + 1];
  __shared__ double s[kNB], xk[kNB], rinv[kNB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int c0 = kNB * kb, nc = (n - c0 < kNB) ? n - c0 : kNB;
  const double* Ld = Ldiag + static_cast<size_t>(kb) * (kNB * kNB + kNB);
  for (int t = tid; t < kNB * kNB; t += 256) Lk[t / kNB][t % kNB] = Ld[t];
  if (tid < kNB) { rinv[tid] = Ld[kNB * kNB + tid]; s[tid] = tid < nc ? yrow[c0 + tid] : 0.0; }
  __syncthreads();
  if (tid < 64) {                                        // L_kk^T x_k = s, by one wave: x_j, then s_t -= L[j][t] x_j for t < j
    for (int j = nc - 1; j >= 0; --j) {
      if (lane == j) xk[j] = s[j] * rinv[j];
      const double xj = xk[j];                           // (the wave's own LDS write just above)
      if (lane < j) s[lane] -= Lk[j][lane] * xj;
    }
  }
  __syncthreads();
  const int bj = blockIdx.x;                             // column block j < kb: y_j -= L[rows of block kb][columns of block j]^T x_k;  bj == kb: keep x_k
  if (bj == kb) {
    if (tid < nc) xvec[c0 + tid] = xk[tid];
    return;
  }
  if (tid < kNB) {
    const int col = kNB * bj + tid;
    double v = yrow[col];
    for (int r = 0; r < nc; ++r) v -= A[static_cast<size_t>(c0 + r) * n + col] * xk[r];
    yrow[col] = v;
  }
}

__global__ __launch_bounds__(256) void dense_finish_kernel(const double* __restrict__ xvec, float* __restrict__ poses, float* __restrict__ dx_ws,
                                                           float* __restrict__ dx_out, int* __restrict__ meta, int* __restrict__ status_out,
                                                           int P, int t0, Riders riders) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (blockIdx.x > 0) {                                                  // rider workgroups (launched only with riders)
    ride(smem, riders, blockIdx.x - 1);
    return;
  }
  const int n = 6 * P, tid = threadIdx.x;
  const int failed = meta[4] | meta[2];
  for (int idx = tid; idx < n; idx += 256) {
    const double xv = xvec[idx];
    const float v = (failed || !(xv == xv)) ? 0.0f : static_cast<float>(xv);    // zeros on failure (:1186-1189)
    dx_ws[idx] = v;
    if (dx_out) dx_out[idx] = v;
  }
  __syncthreads();
  for (int p = tid; p < P; p += 256) {               // pose_retr_kernel (:877-910)
    float xi[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) xi[c] = dx_ws[6 * p + c];
    float* ps = poses + 7 * static_cast<long long>(t0 + p);
    const Pose Tn = retract(xi, load_pose(ps));
    ps[0] = Tn.t.x; ps[1] = Tn.t.y; ps[2] = Tn.t.z;
    ps[3] = Tn.q.x; ps[4] = Tn.q.y; ps[5] = Tn.q.z; ps[6] = Tn.q.w;
  }
  __syncthreads();
  if (tid == 0) {
    meta[4] = 0;
    if (failed) meta[1] = 1;
    if (status_out) { status_out[0] = meta[1]; status_out[1] = meta[0]; status_out[2] = meta[2]; status_out[3] = meta[3]; }
  }
}

// ---------------------------------------------------------------------------
// backsub: dz = Q (w - sum_r E_r^T dx[pose(r)]), disps += dz
// ---------------------------------------------------------------------------
__device__ __forceinline__ void ba_backsub_body(
    const Plan& pl, const int64_t* __restrict__ jj, const float* __restrict__ Ei, const float* __restrict__ Eij,
    const float* __restrict__ Q, const float* __restrict__ w, const float* __restrict__ dx,
    float* __restrict__ disps, float* __restrict__ dz_out, int dz_rows, int HW, int t0, int P, int flags,
    int clamp_frames, float disp_min) {
  const int k = blockIdx.y;
  const int x = blockIdx.x * 256 + threadIdx.x;
  // (round 4) the depth frame's out-edges, their target poses and the pose updates those rows multiply, once per workgroup in LDS:
  // row_of() walked eptr -> eidx -> jj per row and pixel, three dependent loads in front of every row's six products
  __shared__ int s_edge[256], s_pose[256];
  __shared__ float s_dx[256][6];
  const bool live = k < pl.meta[0] && !pl.meta[2];        // (uniform; meta[2]: eta's row count != K - nothing is updated)
  const int e0 = live ? pl.eptr[k] : 0, deg = live ? pl.eptr[k + 1] - e0 : 0;
  const bool in_lds = deg <= 255;
  const int lo = (flags & 1) ? 0 : 1;                     // EvT6x1_kernel returns early for pose index <= 0 (:1084): window pose 0 never reaches dz
  if (live && in_lds && static_cast<int>(threadIdx.x) <= deg) {
    const int r = static_cast<int>(threadIdx.x) - 1;      // row -1 = the frame's own pose row (Ei), rows 0.. = its out-edges (Eij)
    int e = 0, p;
    if (r < 0) p = pl.kx[k] - t0;
    else { e = pl.eidx[e0 + r]; p = static_cast<int>(jj[e]) - t0; }
    const bool ok = p >= 0 && p < P && p >= lo;
    s_edge[threadIdx.x] = e; s_pose[threadIdx.x] = ok ? p : -1;
#pragma unroll
    for (int n = 0; n < 6; ++n) s_dx[threadIdx.x][n] = ok ? dx[6 * p + n] : 0.0f;
  }
  BA_WG_PROBE(2, 0);
  __syncthreads();
  BA_WG_PROBE(2, 1);
  if (x >= HW) return;
  // clamp_frames > 0: disps[:clamp_frames].clamp_(min=disp_min) in the same launch (depth_video.py:214) - frames this BA
  // does not optimise here, by frame index; optimised ones below, after their update
  if (k < clamp_frames && pl.kidx[k] < 0) {
    const float v = disps[static_cast<long long>(k) * HW + x];
    if (v < disp_min) disps[static_cast<long long>(k) * HW + x] = disp_min;            // NaN stays NaN, as in torch.clamp
  }
  if (!live) return;
  float acc = 0.0f;
  if (in_lds) {
    for (int q = 0; q <= deg; ++q) {
      const int p = s_pose[q];
      if (p < 0) continue;
      const float* base = (q == 0) ? Ei + static_cast<long long>(p) * 6 * HW : Eij + static_cast<long long>(s_edge[q]) * 6 * HW;
      float sacc = 0.0f;
#pragma unroll
      for (int n = 0; n < 6; ++n) sacc += base[static_cast<long long>(n) * HW + x] * s_dx[q][n];
      acc += sacc;
    }
  } else {
    for (int r = -1; r < deg; ++r) {
      const RowRef R = row_of(r, k, pl, Ei, Eij, jj, HW, t0, P);
      if (R.pose < lo) continue;
      float sacc = 0.0f;
#pragma unroll
      for (int n = 0; n < 6; ++n) sacc += R.base[static_cast<long long>(n) * HW + x] * dx[6 * R.pose + n];
      acc += sacc;
    }
  }
  const float dz = Q[static_cast<long long>(k) * HW + x] * (w[static_cast<long long>(k) * HW + x] - acc);
  float d = disps[static_cast<long long>(pl.kx[k]) * HW + x] + dz;  // disp_retr_kernel (:912-925)
  if (pl.kx[k] < clamp_frames && d < disp_min) d = disp_min;
  disps[static_cast<long long>(pl.kx[k]) * HW + x] = d;
  if (dz_out && k < dz_rows) dz_out[static_cast<long long>(k) * HW + x] = dz;
  BA_WG_PROBE(2, 2);
}

__global__ __launch_bounds__(256) void ba_backsub_kernel(
    Plan pl, const int64_t* __restrict__ jj, const float* __restrict__ Ei, const float* __restrict__ Eij,
    const float* __restrict__ Q, const float* __restrict__ w, const float* __restrict__ dx,
    float* __restrict__ disps, float* __restrict__ dz_out, int dz_rows, int HW, int t0, int P, int flags,
    int clamp_frames, float disp_min) {
  ba_backsub_body(pl, jj, Ei, Eij, Q, w, dx, disps, dz_out, dz_rows, HW, t0, P, flags, clamp_frames, disp_min);
}

int check_common(int E, int F, int ht, int wd, int t0, int t1) {
  if (E < 0 || F <= 0 || ht <= 0 || wd <= 0 || t0 < 0 || t1 < t0 || t1 > F) return PVO_EINVAL;
  if (E > 65535) return PVO_EUNSUPPORTED;
  return PVO_OK;
}

}  // namespace

#if defined(PVO_BA_PROBE) && PVO_BA_PROBE == 3
extern "C" int pvo_debug_ba_wg_probe(void* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_ba_wg_probe), &buf, sizeof(buf)) == hipSuccess ? 0 : 1; }
#endif
#ifdef PVO_BA_PROBE
extern "C" int pvo_debug_ba_probe(void* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_ba_probe), &buf, sizeof(buf)) == hipSuccess ? 0 : 1; }
#endif



extern "C" size_t pvo_ba_workspace_bytes(int E, int P, int nframes, int HW) {
  if (E < 0 || P < 0 || nframes < 0 || HW < 0) return 0;
  return carve(nullptr, E, P, nframes, HW).bytes + 256;
}

static inline void* ws_base(void* workspace) {
  return reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
}

extern "C" int pvo_ba_plan(const int64_t* ii, const int64_t* jj, int E, int nframes, int HW,
                           int K_eta, int t0, int t1, void* workspace, size_t workspace_bytes,
                           void* stream) {
  if (E < 0 || nframes <= 0 || t0 < 0 || t1 < t0 || t1 > nframes || !workspace) return PVO_EINVAL;
  if (E > 0 && (!ii || !jj)) return PVO_EINVAL;
  const int P = t1 - t0;
  if (workspace_bytes < pvo_ba_workspace_bytes(E, P, nframes, HW)) return PVO_EWORKSPACE;
  Ws w = carve(ws_base(workspace), E, P, nframes, HW);
  hipLaunchKernelGGL(ba_plan_kernel, dim3(1), dim3(256), 0, pvo_stream(stream),
                     ii, jj, w.plan, E, nframes, t0, t1, K_eta, K_eta < 0 ? 1 : 0);
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_ba_local(const float* poses, const float* disps, const float* intrinsics,
                            const float* targets, const float* weights, const float* eta,
                            const int64_t* ii, const int64_t* jj,
                            int E, int nframes, int ht, int wd, int K_eta, int t0, int t1,
                            int motion_only, void* sys_,
                            void* workspace, size_t workspace_bytes, void* stream) {
  long long* sys = static_cast<long long*>(sys_);
  int rc = check_common(E, nframes, ht, wd, t0, t1);
  if (rc != PVO_OK) return rc;
  const int P = t1 - t0, HW = ht * wd;
  if (!poses || !disps || !intrinsics || !sys || !workspace) return PVO_EINVAL;
  if (E > 0 && (!targets || !weights || !ii || !jj)) return PVO_EINVAL;
  if (!motion_only && !eta) return PVO_EINVAL;
  if (P > 5400 || nframes >= (1 << 28)) return PVO_EUNSUPPORTED;      // (the Schur kernel's row table: 6 P + 5 in a short, row indices in 28 bits)
  if (workspace_bytes < pvo_ba_workspace_bytes(E, P, nframes, HW)) return PVO_EWORKSPACE;
  Ws w = carve(ws_base(workspace), E, P, nframes, HW);
  hipStream_t st = pvo_stream(stream);
  // pvo_ba_finish leaves `sys` zeroed after reading it; callers that chain local -> finish -> local on the same buffer
  // say so with bit 1 of motion_only and save the memset
  const size_t n6 = static_cast<size_t>(6) * P;
  const bool clean = (motion_only & 2) != 0;
  const bool only_assemble = (motion_only & 4) != 0, only_schur = (motion_only & 8) != 0;   // (halves of this call: schedule experiments)
  motion_only &= 1;
  if (!clean && !only_schur && hipMemsetAsync(sys, 0, sizeof(long long) * (n6 * n6 + n6), st) != hipSuccess) return PVO_ELAUNCH;
  if (E == 0) return PVO_OK;
  const bool two_stage = !motion_only;      // the chunk sums go through the Schur kernel (there is none in a motion-only BA)
  const int chunksA = (HW + kChunkA - 1) / kChunkA;
  if (!only_schur)
  hipLaunchKernelGGL(ba_assemble_kernel, dim3((HW + kChunkA - 1) / kChunkA, E), dim3(256), 0, st,
                     poses, disps, intrinsics, targets, weights, ii, jj, w.Eii, w.Eij, w.Cii, w.bz,
                     sys, w.plan.meta, HW, wd, t0, P, motion_only, two_stage ? w.part : nullptr);
  PVO_CHECK_LAUNCH();
  if (!motion_only && !only_assemble) {
    const int Kmax = (nframes < P + E) ? nframes : (P + E);
    // pixels per workgroup: 256 while the grid fits the chip's resident slots (two 55 KB workgroups per CU); a global bundle
    // adjustment (64 keyframes x 12 chunks: 768 workgroups in two rounds, every one issuing ~1000 atomics into the same 63 x 63
    // blocks) takes bigger chunks - fewer, longer workgroups, a fraction of the atomics
    // (a function of the map size and the POSE WINDOW only - never of this rank's edge count: the chunking fixes the fp32
    // partial sums, and an edge-sharded run must form the same ones as the whole graph; t0 / t1 are the same on every rank.
    // Depth frames optimised = the window's frames + the few source frames in front of it, so P + 1 stands for their number.
    // Until round 5 this was the BUFFER length: a 1024-frame video buffer - the reference driver's default - put every window-
    // sized BA on 1024-pixel chunks, 3 x K workgroups, 443 us instead of 20: found by the full-sequence run of bench.py)
    const int frames_opt = (nframes < P + 1) ? nframes : (P + 1);
    const long long wg256 = static_cast<long long>((HW + 255) / 256) * frames_opt;
    // (256-pixel chunks only up to ~190 workgroups, i.e. windows of up to 15 poses at 12 chunks a frame: a frontend window of the
    // full-sequence run - 21-25 poses, frames with 18 neighbours - took 86.5 us with 256-pixel chunks and 73.4 with 512: half the
    // tile-pair atomics and half the workgroups that each redo row table and depth phase per z-slice; 2 / 8 slices and 1024-pixel
    // chunks were all slower, tools/_probe sweep of a dumped window)
    // Round 6: a DENSE window (up to kDenseMaxPoses poses whose 256-pixel grid would not fit: the frontend's window with its
    // inactive edges) stays on 256-pixel chunks and ONE slice, with the dynamic LDS segment for a frame's rows (schur_rowpass_lds);
    // its depth phase is ba_depth_kernel's.  By P and the map size only, like the chunk rule: the same on every rank.
    const bool dense_window = P <= kDenseMaxPoses && wg256 > 192;
    const int pix = (wg256 <= 192 || dense_window) ? 256 : (wg256 <= 1024 ? 512 : 1024);
    const int gx = (HW + pix - 1) / pix;
    const int deal_rows = two_stage ? (E + kDealEdges * gx - 1) / (kDealEdges * gx) : 0;      // workgroups that add up the assembly's chunk sums: kDealEdges edges each
    // rows of the grid = depth frames: the caller's eta has one row per depth frame (K_eta == K, checked by the plan kernel and
    // reported in the status words), so that is the count; only a broadcast eta (one row) leaves the host with the bound P + E -
    // which at a real window's 48 + 400 edges meant 454 grid rows x 12 chunks x 4 slices for 27 depth frames
    const int Kgrid = (K_eta > 1 && K_eta <= Kmax) ? K_eta : Kmax;
    const dim3 sgrid(gx, Kgrid + deal_rows, dense_window ? 1 : 4);      // (z: slices for the row-tile passes of many-neighbour frames, see ba_schur_body)
    const int depth_done = (pix != 256 || dense_window) ? 1 : 0;       // (by map size and pose window only: the same on every rank of a sharded run)
    // dynamic LDS of the dense-window form: what the CU's 160 KB leave beside the kernel's static tables (49.7 KB)
    constexpr int kSchurRowsLds = 110 * 1024;
    const size_t sdyn = dense_window ? kSchurRowsLds : 0;
    const int lds_floats = static_cast<int>(sdyn / sizeof(float));
    if (dense_window) {
      static bool schur_attr_set = false;
      if (!schur_attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(ba_schur_mfma_kernel<true, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, kSchurRowsLds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(ba_schur_mfma_kernel<false, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, kSchurRowsLds) != hipSuccess)
          return PVO_ELAUNCH;
        schur_attr_set = true;
      }
    }
    if (depth_done) {
      hipLaunchKernelGGL(ba_depth_kernel, dim3((HW + 255) / 256, Kgrid), dim3(256), 0, st, w.plan, eta, K_eta, w.Eii, w.Cii, w.bz, w.Ei, w.Q, w.w, HW, t0, P);
      PVO_CHECK_LAUNCH();
    }
    const bool stage2 = dense_window && Kgrid <= schur_stage_frames(P);       // (the workspace holds the chunk sums of that many frames)
#define PVO_SCHUR_LAUNCH(V, PX) hipLaunchKernelGGL((ba_schur_mfma_kernel<V, PX>), sgrid, dim3(256), (PX == 256 ? sdyn : 0), st, w.plan, jj, eta, K_eta, w.Eii, w.Cii, w.bz, w.Ei, \
                                                   w.Eij, w.Q, w.w, sys, HW, t0, P, two_stage ? w.part : nullptr, ii, E, chunksA, deal_rows, w.Mrg, depth_done, (PX == 256 ? lds_floats : 0), \
                                                   (stage2 && PX == 256) ? w.spart : nullptr, w.srow, stage2 ? w.sT : nullptr)
    if ((HW & 3) == 0) { if (pix == 256) PVO_SCHUR_LAUNCH(true, 256); else if (pix == 512) PVO_SCHUR_LAUNCH(true, 512); else PVO_SCHUR_LAUNCH(true, 1024); }
    else { if (pix == 256) PVO_SCHUR_LAUNCH(false, 256); else if (pix == 512) PVO_SCHUR_LAUNCH(false, 512); else PVO_SCHUR_LAUNCH(false, 1024); }
#undef PVO_SCHUR_LAUNCH
    PVO_CHECK_LAUNCH();
    if (stage2) {
      hipLaunchKernelGGL(ba_schur_reduce_kernel, dim3(kSchurStagePairs, Kgrid), dim3(256), 0, st, w.plan, w.spart, w.srow, w.sT, sys, gx, 6 * P);
      PVO_CHECK_LAUNCH();
    }
  }
  return PVO_OK;
}

extern "C" int pvo_ba_finish(float* poses, float* disps, void* sys_,
                             const int64_t* ii, const int64_t* jj,
                             int E, int nframes, int ht, int wd, int t0, int t1,
                             float lm, float ep, int motion_only, int clamp_frames, float disp_min,
                             float* dx_out, float* dz_out, int dz_rows, int* status_out,
                             void* workspace, size_t workspace_bytes, void* stream) {
  return pvo_ba_finish_riders(poses, disps, sys_, ii, jj, E, nframes, ht, wd, t0, t1, lm, ep, motion_only, clamp_frames, disp_min,
                              dx_out, dz_out, dz_rows, status_out, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int pvo_ba_finish_conv1x1(float* poses, float* disps, void* sys_,
                                     const int64_t* ii, const int64_t* jj,
                                     int E, int nframes, int ht, int wd, int t0, int t1,
                                     float lm, float ep, int motion_only, int clamp_frames, float disp_min,
                                     float* dx_out, float* dz_out, int dz_rows, int* status_out,
                                     void* workspace, size_t workspace_bytes,
                                     const void* cx, const void* cw, const float* cbias, void* cy, long long crows, int cCout, int cdtype,
                                     void* stream) {
  pvo_ba_riders r{};
  r.cx = cx; r.cw = cw; r.cbias = cbias; r.cy = cy; r.crows = crows; r.cCout = cCout; r.cdtype = cdtype;
  return pvo_ba_finish_riders(poses, disps, sys_, ii, jj, E, nframes, ht, wd, t0, t1, lm, ep, motion_only, clamp_frames, disp_min,
                              dx_out, dz_out, dz_rows, status_out, workspace, workspace_bytes, cy ? &r : nullptr, stream);
}

static int ba_finish_impl(float* poses, float* disps, void* sys_, const long long* msg, const int* first_s,
                          const int64_t* ii, const int64_t* jj,
                          int E, int nframes, int ht, int wd, int t0, int t1,
                          float lm, float ep, int motion_only, int clamp_frames, float disp_min,
                          float* dx_out, float* dz_out, int dz_rows, int* status_out,
                          void* workspace, size_t workspace_bytes,
                          const pvo_ba_riders* jobs, void* stream);

extern "C" int pvo_ba_finish_riders(float* poses, float* disps, void* sys_,
                                    const int64_t* ii, const int64_t* jj,
                                    int E, int nframes, int ht, int wd, int t0, int t1,
                                    float lm, float ep, int motion_only, int clamp_frames, float disp_min,
                                    float* dx_out, float* dz_out, int dz_rows, int* status_out,
                                    void* workspace, size_t workspace_bytes,
                                    const pvo_ba_riders* jobs, void* stream) {
  if (!sys_) return PVO_EINVAL;
  return ba_finish_impl(poses, disps, sys_, nullptr, nullptr, ii, jj, E, nframes, ht, wd, t0, t1, lm, ep, motion_only, clamp_frames, disp_min,
                        dx_out, dz_out, dz_rows, status_out, workspace, workspace_bytes, jobs, stream);
}

extern "C" size_t pvo_ba_packed_elems(const int* first_host, int P) {
  if (!first_host || P < 0) return 0;
  size_t blocks = 0;
  for (int b = 0; b < P; ++b) {
    const int f = first_host[b] < b ? (first_host[b] < 0 ? 0 : first_host[b]) : b;
    blocks += static_cast<size_t>(b - f + 1) * 36;
  }
  return blocks + static_cast<size_t>(6) * P;
}

extern "C" int pvo_ba_pack(void* sys_, const int* first_s, void* msg, int P, void* stream) {
  if (!sys_ || !first_s || !msg || P < 0) return PVO_EINVAL;
  if (P > kMaxEnvBlocks) return PVO_EUNSUPPORTED;
  if (P == 0) return PVO_OK;
  const int n6 = 6 * P;
  hipLaunchKernelGGL(ba_pack_kernel, dim3((n6 * n6 + n6 + 2047) / 2048), dim3(256), 0, pvo_stream(stream),
                     static_cast<long long*>(sys_), first_s, static_cast<long long*>(msg), n6);
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_ba_finish_packed(float* poses, float* disps, const void* msg, const int* first_s,
                                    const int64_t* ii, const int64_t* jj,
                                    int E, int nframes, int ht, int wd, int t0, int t1,
                                    float lm, float ep, int motion_only, int clamp_frames, float disp_min,
                                    float* dx_out, float* dz_out, int dz_rows, int* status_out,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  if (!msg || !first_s) return PVO_EINVAL;
  return ba_finish_impl(poses, disps, nullptr, static_cast<const long long*>(msg), first_s, ii, jj, E, nframes, ht, wd, t0, t1, lm, ep, motion_only,
                        clamp_frames, disp_min, dx_out, dz_out, dz_rows, status_out, workspace, workspace_bytes, nullptr, stream);
}

static int ba_finish_impl(float* poses, float* disps, void* sys_, const long long* msg, const int* first_s,
                          const int64_t* ii, const int64_t* jj,
                          int E, int nframes, int ht, int wd, int t0, int t1,
                          float lm, float ep, int motion_only, int clamp_frames, float disp_min,
                          float* dx_out, float* dz_out, int dz_rows, int* status_out,
                          void* workspace, size_t workspace_bytes,
                          const pvo_ba_riders* jobs, void* stream) {
  Riders rider{};
  size_t rider_lds = 0;
  if (jobs && jobs->cy && jobs->crows > 0) {
    if (!jobs->cx || !jobs->cw || jobs->cCout <= 0 || jobs->cCout % 192) return PVO_EINVAL;
    if (jobs->cdtype != PVO_F16 && jobs->cdtype != PVO_BF16) return PVO_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(jobs->cx) | reinterpret_cast<uintptr_t>(jobs->cw) | reinterpret_cast<uintptr_t>(jobs->cy)) & 15) return PVO_EINVAL;
    if (jobs->crows > (1LL << 24)) return PVO_EUNSUPPORTED;
    rider.cx = static_cast<const uint16_t*>(jobs->cx); rider.cw = static_cast<const uint16_t*>(jobs->cw); rider.cbias = jobs->cbias;
    rider.cy = static_cast<uint16_t*>(jobs->cy); rider.crows = jobs->crows; rider.cCout = jobs->cCout; rider.cdtype = jobs->cdtype;
    rider.crow_blocks = static_cast<int>((jobs->crows + 63) / 64);
    rider.cblocks = rider.crow_blocks * (jobs->cCout / 192);
    rider_lds = c1t::kTileBytes;
  }
  if (jobs && jobs->gpart && jobs->gE > 0 && jobs->gHW > 0) {
    if (!jobs->gnet || !jobs->gw || jobs->gE > 65535) return PVO_EINVAL;
    if (jobs->gdtype != PVO_F16 && jobs->gdtype != PVO_BF16) return PVO_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(jobs->gnet) | reinterpret_cast<uintptr_t>(jobs->gw)) & 15) return PVO_EINVAL;
    rider.gnet = static_cast<const uint16_t*>(jobs->gnet); rider.gw = static_cast<const uint16_t*>(jobs->gw); rider.gbias = jobs->gbias;
    rider.gpart = jobs->gpart; rider.gHW = jobs->gHW; rider.gchunks = (jobs->gHW + 255) / 256; rider.gdtype = jobs->gdtype;
    rider.gblocks = jobs->gE * rider.gchunks;
    if (rider_lds < static_cast<size_t>(glt::kTileBytes)) rider_lds = glt::kTileBytes;
  }
  if (jobs && jobs->xg && jobs->xE > 0) {
    if (!jobs->xpart || !jobs->xwt || !jobs->xbias || jobs->xchunks <= 0) return PVO_EINVAL;
    rider.xpart = jobs->xpart; rider.xwt = jobs->xwt; rider.xbias = jobs->xbias; rider.xg = jobs->xg;
    rider.xchunks = jobs->xchunks; rider.xblocks = jobs->xE;
    if (rider_lds < 512) rider_lds = 512;
  }
  const int rider_blocks = rider.cblocks + rider.gblocks + rider.xblocks;
  int rc = check_common(E, nframes, ht, wd, t0, t1);
  if (rc != PVO_OK) return rc;
  if (clamp_frames < 0 || clamp_frames > nframes) return PVO_EINVAL;
  const int P = t1 - t0, HW = ht * wd;
  long long* sys = static_cast<long long*>(sys_);
  if (!poses || !disps || (!sys && !msg) || !workspace) return PVO_EINVAL;
  if (P > kMaxEnvBlocks) return PVO_EUNSUPPORTED;           // (a 12288^2 fp64 system: 1.2 GB)
  if (workspace_bytes < pvo_ba_workspace_bytes(E, P, nframes, HW)) return PVO_EWORKSPACE;
  Ws w = carve(ws_base(workspace), E, P, nframes, HW);
  hipStream_t st = pvo_stream(stream);
  const int n6 = 6 * P;
  // Round 6: a window's system (up to kDenseMaxPoses free poses; dense image or packed message) is factorised DENSE in the registers of one
  // workgroup on the fp64 matrix cores (ba_solve_dense_kernel) - one launch where the envelope forms took three beyond 21 poses.
  // pvo_debug_config(PVO_KNOB_BA_SOLVER, 1..4) selects the older forms (tests compare them), 5 names this one.
  const int solver_env = pvo_knob(PVO_KNOB_BA_SOLVER) - 1;      // (pvo_debug_config: -1 = the choice by size below)
  if (P > 0 && P <= kDenseMaxPoses && (solver_env < 0 || solver_env == 4)) {
    const size_t dl = dense_lds_bytes(n6);
    const size_t lds_d = dl > rider_lds ? dl : rider_lds;
    const int Td = dense_N(n6) / 16;                            // tiles per side: 4 slots per wave up to 5 (15 tiles), 14 up to 10 (55: 26 poses), 17 up to 11
    static bool dense_attr_set = false;
    if (!dense_attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(ba_solve_dense_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 142000) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(ba_solve_dense_kernel<14>), hipFuncAttributeMaxDynamicSharedMemorySize, 142000) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(ba_solve_dense_kernel<17>), hipFuncAttributeMaxDynamicSharedMemorySize, 142000) != hipSuccess)
        return PVO_ELAUNCH;
      dense_attr_set = true;
    }
    if (Td <= 5)
      hipLaunchKernelGGL(ba_solve_dense_kernel<4>, dim3(1 + rider_blocks), dim3(256), lds_d, st, sys, msg, first_s, poses, w.dx, dx_out, w.plan.meta, status_out,
                         P, t0, lm, ep, rider);
    else if (Td <= 10)
      hipLaunchKernelGGL(ba_solve_dense_kernel<14>, dim3(1 + rider_blocks), dim3(256), lds_d, st, sys, msg, first_s, poses, w.dx, dx_out, w.plan.meta, status_out,
                         P, t0, lm, ep, rider);
    else
      hipLaunchKernelGGL(ba_solve_dense_kernel<17>, dim3(1 + rider_blocks), dim3(256), lds_d, st, sys, msg, first_s, poses, w.dx, dx_out, w.plan.meta, status_out,
                         P, t0, lm, ep, rider);
    PVO_CHECK_LAUNCH();
    if (!motion_only && E + P > 0) {
      const int Kmax = (nframes < P + E) ? nframes : (P + E);
      hipLaunchKernelGGL(ba_backsub_kernel, dim3((HW + 255) / 256, Kmax > clamp_frames ? Kmax : clamp_frames), dim3(256), 0, st,
                         w.plan, jj, w.Ei, w.Eij, w.Q, w.w, w.dx, disps, dz_out, dz_rows, HW, t0, P, 0, clamp_frames, disp_min);
      PVO_CHECK_LAUNCH();
    }
    return PVO_OK;
  }
  const int use_lds = n6 <= kLdsCholMax && !msg;               // (a packed message is factorised from the compact image at every size)
  constexpr size_t kSolveLdsMax = 142000;      // dynamic LDS of the solve: the CU's 163840 B minus its 20528 B of static tables (envelope, reach, active rows, scan buffers)
  size_t lds = use_lds ? 16 + sizeof(double) * (static_cast<size_t>(n6) * n6 + n6 + 27 * P + 24) : kSolveLdsMax;
  if (lds < rider_lds) lds = rider_lds;                                   // (the riders' tiles live in the dynamic segment)
  if (lds > 48 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(ba_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(kSolveLdsMax)) != hipSuccess) return PVO_ELAUNCH;
      attr_set = true;
    }
  }
  // Three bit-identical factorisations of a system that lives in LDS (chol_solve_blocked: four waves and barriers;
  // chol_solve_wave: one wave, none; chol_solve_pipe: wave 0 on the critical chain, three worker waves behind it).  Measured
  // with tools/ba_solve_timeline.py, cycles of the whole kernel: 7 free poses 44.8 k blocked / 48.0 k wave / 52.1 k pipe,
  // 12: 87.5 k blocked / 86.0 k pipe, 21: 168 k / 157 k, 63: 578 k blocked / 485 k wave / 364 k pipe - the pipeline wins once the
  // envelope makes most of a step's candidate rows inactive.  pvo_debug_config(PVO_KNOB_BA_SOLVER, ..) overrides (tests compare
  // the three bit for bit).
  // Beyond the dense LDS path a fourth form, the PARTITIONED solve (ba_solve_twin_kernel: two workgroups eliminate the pose
  // chain from both ends, tools/ba_solve_timeline.py), is the default; its result equals the others' to fp64 rounding.
  // Round 6: a big system that is not narrow-banded - more than 8 edges per pose, i.e. a global graph connected by proximity rather than
  // a keyframe chain - is factorised DENSE in 48 x 48 blocks over many workgroups (dense_panel / dense_update / dense_back kernels).
  // By P and this call's E: used for the dense image only (an edge-sharded step arrives as a packed message and keeps the forms every
  // rank chooses alike).  pvo_debug_config(PVO_KNOB_BA_SOLVER, 6) forces it at any size beyond the LDS path.
  if (!msg && !use_lds && P > 0 && ((solver_env < 0 && static_cast<long long>(E) > 8LL * P) || solver_env == 5)) {
    // (no envelope pass: ba_prepare_kernel with no LDS budget writes the dense row-major image whatever `env` holds - INT_MAX between solves)
    hipLaunchKernelGGL(ba_prepare_kernel, dim3((n6 * n6 + n6 + 2047) / 2048), dim3(256), 0, st, sys, w.chol, w.plan.env, n6, lm, ep,
                       0LL /* no LDS budget: the dense row-major image */, static_cast<int*>(nullptr));
    PVO_CHECK_LAUNCH();
    if (hipMemsetAsync(w.xchg, 0, 16, st) != hipSuccess) return PVO_ELAUNCH;                                       // (pvo_ba_last_partition: no partition)
    const int nkb = (n6 + kNB - 1) / kNB, nrb = (n6 + 1 + kNB - 1) / kNB;
    for (int kb = 0; kb < nkb; ++kb) {
      hipLaunchKernelGGL(dense_panel_kernel, dim3(nrb - kb), dim3(256), 0, st, w.chol, w.ldiag, n6, kb, w.plan.meta);
      const int m = nrb - kb - 1;
      if (m > 0) hipLaunchKernelGGL(dense_update_kernel, dim3(m * (m + 1) / 2), dim3(256), 0, st, w.chol, n6, kb, nrb);
    }
    PVO_CHECK_LAUNCH();
    double* yrow = w.chol + static_cast<size_t>(n6) * n6;
    for (int kb = nkb - 1; kb >= 0; --kb)
      hipLaunchKernelGGL(dense_back_kernel, dim3(kb + 1), dim3(256), 0, st, w.chol, w.ldiag, yrow, w.xvec, n6, kb);
    PVO_CHECK_LAUNCH();
    if (rider_lds > 48 * 1024) {
      static bool fin_attr_set = false;
      if (!fin_attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(dense_finish_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kSolveLdsMax)) != hipSuccess)
          return PVO_ELAUNCH;
        fin_attr_set = true;
      }
    }
    hipLaunchKernelGGL(dense_finish_kernel, dim3(1 + rider_blocks), dim3(256), rider_lds, st, w.xvec, poses, w.dx, dx_out, w.plan.meta, status_out, P, t0, rider);
    PVO_CHECK_LAUNCH();
    if (!motion_only && E + P > 0) {
      const int Kmax = (nframes < P + E) ? nframes : (P + E);
      hipLaunchKernelGGL(ba_backsub_kernel, dim3((HW + 255) / 256, Kmax > clamp_frames ? Kmax : clamp_frames), dim3(256), 0, st,
                         w.plan, jj, w.Ei, w.Eij, w.Q, w.w, w.dx, disps, dz_out, dz_rows, HW, t0, P, 0, clamp_frames, disp_min);
      PVO_CHECK_LAUNCH();
    }
    return PVO_OK;
  }
  const int solver_pick = (solver_env >= 0 && solver_env < 4) ? solver_env : (use_lds ? (P > 12 ? 2 : 0) : 3);      // 0 blocked | 1 wave | 2 pipe | 3 partitioned
  const bool twin = solver_pick == 3 && !use_lds;
  const int solver_wave = solver_pick == 3 ? 2 : solver_pick;
  if (msg) {
    if (P == 0) return PVO_OK;
    hipLaunchKernelGGL(ba_env_packed_kernel, dim3(P), dim3(256), 0, st, msg, first_s, w.plan.env, n6);
    PVO_CHECK_LAUNCH();
    hipLaunchKernelGGL(ba_prepare_packed_kernel, dim3(P + 1), dim3(256), 0, st, msg, first_s, w.chol, w.plan.env, n6, lm, ep,
                       static_cast<long long>(kSolveLdsMax), twin ? reinterpret_cast<int*>(w.xchg) : nullptr);
    PVO_CHECK_LAUNCH();
  } else if (!use_lds) {
    hipLaunchKernelGGL(ba_env_kernel, dim3((n6 * n6 + 2047) / 2048), dim3(256), 0, st, sys, w.plan.env, n6);
    PVO_CHECK_LAUNCH();
    hipLaunchKernelGGL(ba_prepare_kernel, dim3((n6 * n6 + n6 + 2047) / 2048), dim3(256), 0, st, sys, w.chol, w.plan.env, n6, lm, ep,
                       static_cast<long long>(kSolveLdsMax), twin ? reinterpret_cast<int*>(w.xchg) : nullptr);
    PVO_CHECK_LAUNCH();
  }
  if (!use_lds && !twin && hipMemsetAsync(w.xchg, 0, 16, st) != hipSuccess) return PVO_ELAUNCH;      // (pvo_ba_last_partition: one chain)
  if (twin) {
    static bool twin_attr_set = false;
    if (!twin_attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(ba_solve_twin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              static_cast<int>(kSolveLdsMax)) != hipSuccess) return PVO_ELAUNCH;
      twin_attr_set = true;
    }
    hipLaunchKernelGGL(ba_solve_twin_kernel, dim3(2 + rider_blocks), dim3(256), lds, st, w.chol, poses, w.dx, dx_out,
                       w.plan.meta, status_out, P, t0, w.plan.env, static_cast<long long>(kSolveLdsMax), w.xchg, rider);
  } else {
    hipLaunchKernelGGL(ba_solve_kernel, dim3(1 + rider_blocks), dim3(256), lds, st, sys, w.chol, poses, w.dx, dx_out,
                       w.plan.meta, status_out, P, t0, lm, ep, use_lds, w.plan.env, static_cast<long long>(kSolveLdsMax), solver_wave, rider);
  }
  PVO_CHECK_LAUNCH();
  if (!motion_only && E + P > 0) {
    const int Kmax = (nframes < P + E) ? nframes : (P + E);
    const int flags = 0;
    hipLaunchKernelGGL(ba_backsub_kernel, dim3((HW + 255) / 256, Kmax > clamp_frames ? Kmax : clamp_frames), dim3(256), 0, st,
                       w.plan, jj, w.Ei, w.Eij, w.Q, w.w, w.dx, disps, dz_out, dz_rows, HW, t0, P, flags, clamp_frames, disp_min);
    PVO_CHECK_LAUNCH();
  }
  return PVO_OK;
}

// diagnostic (tests, tools): the partition the last pvo_ba_finish on this workspace chose - out[0] = m, out[1] = s, both 0 when the
// pose chain was solved in one piece.  Synchronises the stream.
extern "C" int pvo_ba_last_partition(void* workspace, size_t workspace_bytes, int E, int P, int nframes, int HW, int* out, void* stream) {
  if (!workspace || !out || E < 0 || P < 0) return PVO_EINVAL;
  if (workspace_bytes < pvo_ba_workspace_bytes(E, P, nframes, HW)) return PVO_EWORKSPACE;
  out[0] = out[1] = 0;
  if (6 * P <= kLdsCholMax) return PVO_OK;
  {
    const int k = pvo_knob(PVO_KNOB_BA_SOLVER);
    if (P <= kDenseMaxPoses && (k == 0 || k == 5)) return PVO_OK;      // (the dense solve: nothing to partition; a packed message never asks here)
  }
  Ws w = carve(ws_base(workspace), E, P, nframes, HW);
  int host[4] = {0, 0, 0, 0};
  if (hipMemcpyAsync(host, w.xchg, sizeof(host), hipMemcpyDeviceToHost, pvo_stream(stream)) != hipSuccess) return PVO_ELAUNCH;
  if (hipStreamSynchronize(pvo_stream(stream)) != hipSuccess) return PVO_ELAUNCH;
  out[0] = host[2]; out[1] = host[3];
  return PVO_OK;
}

extern "C" int pvo_ba(float* poses, float* disps, const float* intrinsics,
                      const float* targets, const float* weights, const float* eta,
                      const int64_t* ii, const int64_t* jj,
                      int E, int nframes, int ht, int wd, int K_eta,
                      int t0, int t1, int iterations, float lm, float ep, int motion_only,
                      float* dx_out, float* dz_out, int dz_rows, int* status_out,
                      void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_common(E, nframes, ht, wd, t0, t1);
  if (rc != PVO_OK) return rc;
  if (iterations < 0) return PVO_EINVAL;
  const int P = t1 - t0, HW = ht * wd;
  if (!workspace) return PVO_EINVAL;
  if (workspace_bytes < pvo_ba_workspace_bytes(E, P, nframes, HW)) return PVO_EWORKSPACE;
  Ws w = carve(ws_base(workspace), E, P, nframes, HW);
  hipStream_t st = pvo_stream(stream);
  hipLaunchKernelGGL(ba_plan_kernel, dim3(1), dim3(256), 0, st, ii, jj, w.plan, E, nframes, t0, t1,
                     K_eta, motion_only);
  PVO_CHECK_LAUNCH();
  for (int it = 0; it < iterations; ++it) {
    rc = pvo_ba_local(poses, disps, intrinsics, targets, weights, eta, ii, jj, E, nframes, ht, wd, K_eta,
                      t0, t1, (motion_only ? 1 : 0) | (it > 0 ? 2 : 0), w.sys, workspace, workspace_bytes, stream);
    if (rc != PVO_OK) return rc;
    rc = pvo_ba_finish(poses, disps, w.sys, ii, jj, E, nframes, ht, wd, t0, t1, lm, ep, motion_only, 0, 0.0f,
                       dx_out, dz_out, dz_rows, status_out, workspace, workspace_bytes, stream);
    if (rc != PVO_OK) return rc;
  }
  return PVO_OK;
}
