// geom.hip — reprojection-family kernels: frame_distance, projmap, iproj,
// depth_filter and the non-Jacobian projective transform ("reproject").
//
// Reference: VO_Module/src/droid_kernels.cu:406-495 (projmap), 497-636
// (frame_distance), 640-754 (depth_filter), 758-829 (iproj) and
// VO_Module/droid_slam/geom/projective_ops.py:21-130 (projective_transform).
//
// Launch shape (MI355X): per-pixel kernels are flat 1-D grids over (edge, pixel)
// so that a 36-edge graph already gives 432 workgroups; the reference launches one
// 256-thread block per edge.  The relative pose is recomputed per thread from 14
// scalar-cached floats instead of going through LDS and two barriers.
// frame_distance keeps one workgroup per pair but reduces with wave shuffles.
#include "se3.h"

namespace {

constexpr float kMinDepthNative = 0.25f;  // droid_kernels.cu:26  MIN_DEPTH
constexpr float kMinDepthPy = 0.2f;       // projective_ops.py:6  MIN_DEPTH

struct Intr { float fx, fy, cx, cy; };
__device__ __forceinline__ Intr load_intr(const float* p) { return {p[0], p[1], p[2], p[3]}; }

// one direction of the frame distance for the pair (ix -> jx), computed by the 256 threads `tid` of one group; `red` is
// that group's [3][4] reduction buffer.  Returns the distance in the group's thread 0 (droid_kernels.cu:497-636).
__device__ __forceinline__ float pair_distance(const float* __restrict__ poses, const float* __restrict__ disps, const Intr K,
                                               int ix, int jx, int HW, int wd, float beta, int tid, float (*red)[4]) {
  const Pose G = rel_pose(load_pose(poses + 7 * static_cast<long long>(ix)), load_pose(poses + 7 * static_cast<long long>(jx)));
  const float* __restrict__ d_i = disps + static_cast<long long>(ix) * HW;

  float accum = 0.f, valid = 0.f, total = 0.f;
  for (int k = tid; k < HW; k += 256) {
    const int i = k / wd, j = k - i * wd;
    const float u = static_cast<float>(j), v = static_cast<float>(i);
    float Xi[4] = {(u - K.cx) / K.fx, (v - K.cy) / K.fy, 1.0f, d_i[k]};
    float Xj[4];
    act4(G, Xi, Xj);
    float du = K.fx * (Xj[0] / Xj[2]) + K.cx - u;
    float dv = K.fy * (Xj[1] / Xj[2]) + K.cy - v;
    float d = sqrtf(du * du + dv * dv);
    total += beta;
    if (Xj[2] > kMinDepthNative) { accum += beta * d; valid += beta; }
    // translation-only flow (droid_kernels.cu:597-615)
    Xj[0] = Xi[0] + Xi[3] * G.t.x;
    Xj[1] = Xi[1] + Xi[3] * G.t.y;
    Xj[2] = Xi[2] + Xi[3] * G.t.z;
    du = K.fx * (Xj[0] / Xj[2]) + K.cx - u;
    dv = K.fy * (Xj[1] / Xj[2]) + K.cy - v;
    d = sqrtf(du * du + dv * dv);
    total += (1.0f - beta);
    if (Xj[2] > kMinDepthNative) { accum += (1.0f - beta) * d; valid += (1.0f - beta); }
  }
  accum = pvo_wave_sum(accum); valid = pvo_wave_sum(valid); total = pvo_wave_sum(total);
  const int wave = tid >> 6;
  if ((tid & 63) == 0) { red[0][wave] = accum; red[1][wave] = valid; red[2][wave] = total; }
  __syncthreads();
  const float a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  const float va = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const float to = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
  // droid_kernels.cu:634 (the 1e-8 literal makes the comparison double precision)
  return (static_cast<double>(va) / (static_cast<double>(to) + 1e-8) < 0.75) ? 1000.0f : a / va;
}

__global__ __launch_bounds__(256) void frame_distance_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intrinsics,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, float* __restrict__ dist,
    int HW, int wd, float beta) {
  const int m = blockIdx.x;
  __shared__ float red[3][4];
  const float d = pair_distance(poses, disps, load_intr(intrinsics), static_cast<int>(ii[m]), static_cast<int>(jj[m]), HW, wd, beta,
                                threadIdx.x, red);
  if (threadIdx.x == 0) dist[m] = d;
}

// 0.5 * (distance(ii -> jj) + distance(jj -> ii)) in one launch (depth_video.py:183-193 runs the kernel twice and averages
// with two more element-wise kernels): threads 0..255 of a workgroup take one direction, 256..511 the other, each exactly as
// frame_distance_kernel does, so the result equals the two-launch formulation bit for bit.
__global__ __launch_bounds__(512) void frame_distance_bidir_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intrinsics,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, float* __restrict__ dist,
    int HW, int wd, float beta) {
  const int m = blockIdx.x;
  const int g = threadIdx.x >> 8, tid = threadIdx.x & 255;
  __shared__ float red[2][3][4];
  __shared__ float both[2];
  const int a = static_cast<int>(ii[m]), b = static_cast<int>(jj[m]);
  const float d = pair_distance(poses, disps, load_intr(intrinsics), g ? b : a, g ? a : b, HW, wd, beta, tid, red[g]);
  if (tid == 0) both[g] = d;
  __syncthreads();
  if (threadIdx.x == 0) dist[m] = __fmul_rn(0.5f, __fadd_rn(both[0], both[1]));
}

__global__ __launch_bounds__(256) void projmap_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intrinsics,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj,
    float* __restrict__ coords, float* __restrict__ valid, int HW, int wd) {
  const int e = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= HW) return;
  const int ix = static_cast<int>(ii[e]), jx = static_cast<int>(jj[e]);
  const Intr K = load_intr(intrinsics);
  const Pose G = rel_pose(load_pose(poses + 7 * static_cast<long long>(ix)), load_pose(poses + 7 * static_cast<long long>(jx)));
  const int i = k / wd, j = k - i * wd;
  const float u = static_cast<float>(j), v = static_cast<float>(i);
  float Xi[4] = {(u - K.cx) / K.fx, (v - K.cy) / K.fy, 1.0f, disps[static_cast<long long>(ix) * HW + k]};
  float Xj[4];
  act4(G, Xi, Xj);
  float cu = u, cv = v;
  if (Xj[2] > 0.01f) {   // droid_kernels.cu:487 (0.01 is a double literal there; 0.01f rounds the same side for float Z)
    cu = K.fx * (Xj[0] / Xj[2]) + K.cx;
    cv = K.fy * (Xj[1] / Xj[2]) + K.cy;
  }
  float* c = coords + (static_cast<long long>(e) * HW + k) * 3;
  c[0] = cu; c[1] = cv; c[2] = 0.0f;
  valid[static_cast<long long>(e) * HW + k] = (Xj[2] > kMinDepthNative) ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(256) void iproj_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intrinsics,
    float* __restrict__ points, int HW, int wd) {
  const int n = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= HW) return;
  const Intr K = load_intr(intrinsics);
  const Pose G = load_pose(poses + 7 * static_cast<long long>(n));
  const int i = k / wd, j = k - i * wd;
  float Xi[4] = {(static_cast<float>(j) - K.cx) / K.fx, (static_cast<float>(i) - K.cy) / K.fy, 1.0f,
                 disps[static_cast<long long>(n) * HW + k]};
  float Xj[4];
  act4(G, Xi, Xj);   // NB: the reference applies the pose itself here, not its inverse (droid_kernels.cu:822)
  float* p = points + (static_cast<long long>(n) * HW + k) * 3;
  p[0] = Xj[0] / Xj[3]; p[1] = Xj[1] / Xj[3]; p[2] = Xj[2] / Xj[3];
}

__global__ __launch_bounds__(256) void depth_filter_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intrinsics,
    const int64_t* __restrict__ inds, const float* __restrict__ thresh, float* __restrict__ counter,
    int nframes, int ht, int wd) {
  const int HW = ht * wd;
  const int b = blockIdx.z;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= HW) return;
  const int ix = static_cast<int>(inds[b]);
  const Intr K = load_intr(intrinsics);
  const Pose Gi = load_pose(poses + 7 * static_cast<long long>(ix));
  const float t = thresh[b];
  const int i = k / wd, j = k - i * wd;
  const float di = disps[static_cast<long long>(ix) * HW + k];
  const float Xi[4] = {(static_cast<float>(j) - K.cx) / K.fx, (static_cast<float>(i) - K.cy) / K.fy, 1.0f, di};
  // the reference votes with one atomicAdd per neighbour view (grid.y = 6); the six
  // votes are summed in a register here and stored once.
  float votes = 0.f;
#pragma unroll
  for (int neigh = 0; neigh < 6; ++neigh) {
    const int jx = (neigh < 3) ? ix - neigh - 1 : ix + neigh;   // droid_kernels.cu:674
    if (jx < 0 || jx >= nframes) continue;
    const Pose G = rel_pose(Gi, load_pose(poses + 7 * static_cast<long long>(jx)));
    float Xj[4];
    act4(G, Xi, Xj);
    const float uj = K.fx * (Xj[0] / Xj[2]) + K.cx;
    const float vj = K.fy * (Xj[1] / Xj[2]) + K.cy;
    const float dj = Xj[3] / Xj[2];
    const int u0 = pvo_floor_to_int(uj), v0 = pvo_floor_to_int(vj);
    if (u0 >= 0 && v0 >= 0 && u0 < wd - 1 && v0 < ht - 1) {
      const float* dm = disps + static_cast<long long>(jx) * HW;
      const float d00 = dm[v0 * wd + u0], d01 = dm[v0 * wd + u0 + 1];
      const float d10 = dm[(v0 + 1) * wd + u0], d11 = dm[(v0 + 1) * wd + u0 + 1];
      // droid_kernels.cu:748-751: double-precision reciprocal differences
      const double idj = 1.0 / static_cast<double>(dj);
      if (fabs(idj - 1.0 / static_cast<double>(d00)) < t) votes += 1.0f;
      else if (fabs(idj - 1.0 / static_cast<double>(d01)) < t) votes += 1.0f;
      else if (fabs(idj - 1.0 / static_cast<double>(d10)) < t) votes += 1.0f;
      else if (fabs(idj - 1.0 / static_cast<double>(d11)) < t) votes += 1.0f;
    }
  }
  counter[static_cast<long long>(b) * HW + k] = votes;
}

// projective_ops.py:102-130 with jacobian=False; per-frame intrinsics [nframes,4]
__global__ __launch_bounds__(256) void reproject_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intrinsics,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj,
    float* __restrict__ coords, float* __restrict__ valid, int HW, int wd) {
  const int e = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= HW) return;
  const int ix = static_cast<int>(ii[e]), jx = static_cast<int>(jj[e]);
  const Intr Ki = load_intr(intrinsics + 4 * static_cast<long long>(ix));
  const Intr Kj = load_intr(intrinsics + 4 * static_cast<long long>(jx));
  const Pose G = rel_pose(load_pose(poses + 7 * static_cast<long long>(ix)), load_pose(poses + 7 * static_cast<long long>(jx)));
  const int i = k / wd, j = k - i * wd;
  float X0[4] = {(static_cast<float>(j) - Ki.cx) / Ki.fx, (static_cast<float>(i) - Ki.cy) / Ki.fy, 1.0f,
                 disps[static_cast<long long>(ix) * HW + k]};
  float X1[4];
  act4(G, X0, X1);
  float Z = X1[2];
  Z = (Z < 0.5f * kMinDepthPy) ? 1.0f : Z;   // projective_ops.py:48
  const float d = 1.0f / Z;
  float2 c;
  c.x = Kj.fx * (X1[0] * d) + Kj.cx;
  c.y = Kj.fy * (X1[1] * d) + Kj.cy;
  *reinterpret_cast<float2*>(coords + (static_cast<long long>(e) * HW + k) * 2) = c;
  valid[static_cast<long long>(e) * HW + k] = (X1[2] > kMinDepthPy && X0[2] > kMinDepthPy) ? 1.0f : 0.0f;
}

}  // namespace

#define PVO_REQ(c) do { if (!(c)) return PVO_EINVAL; } while (0)

extern "C" int pvo_frame_distance(const float* poses, const float* disps, const float* intrinsics,
                                  const int64_t* ii, const int64_t* jj, float* dist,
                                  int M, int ht, int wd, float beta, void* stream) {
  PVO_REQ(M >= 0 && ht >= 0 && wd >= 0);
  if (M == 0) return PVO_OK;
  PVO_REQ(poses && disps && intrinsics && ii && jj && dist);
  hipLaunchKernelGGL(frame_distance_kernel, dim3(M), dim3(256), 0, pvo_stream(stream),
                     poses, disps, intrinsics, ii, jj, dist, ht * wd, wd, beta);
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_frame_distance_bidirectional(const float* poses, const float* disps, const float* intrinsics,
                                                const int64_t* ii, const int64_t* jj, float* dist,
                                                int M, int ht, int wd, float beta, void* stream) {
  PVO_REQ(M >= 0 && ht >= 0 && wd >= 0);
  if (M == 0) return PVO_OK;
  PVO_REQ(poses && disps && intrinsics && ii && jj && dist);
  hipLaunchKernelGGL(frame_distance_bidir_kernel, dim3(M), dim3(512), 0, pvo_stream(stream),
                     poses, disps, intrinsics, ii, jj, dist, ht * wd, wd, beta);
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_projmap(const float* poses, const float* disps, const float* intrinsics,
                           const int64_t* ii, const int64_t* jj, float* coords, float* valid,
                           int E, int ht, int wd, void* stream) {
  PVO_REQ(E >= 0 && ht >= 0 && wd >= 0);
  if (E == 0 || ht * wd == 0) return PVO_OK;
  PVO_REQ(poses && disps && intrinsics && ii && jj && coords && valid && E <= 65535);
  hipLaunchKernelGGL(projmap_kernel, dim3((ht * wd + 255) / 256, E), dim3(256), 0, pvo_stream(stream),
                     poses, disps, intrinsics, ii, jj, coords, valid, ht * wd, wd);
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_iproj(const float* poses, const float* disps, const float* intrinsics,
                         float* points, int N, int ht, int wd, void* stream) {
  PVO_REQ(N >= 0 && ht >= 0 && wd >= 0);
  if (N == 0 || ht * wd == 0) return PVO_OK;
  PVO_REQ(poses && disps && intrinsics && points && N <= 65535);
  hipLaunchKernelGGL(iproj_kernel, dim3((ht * wd + 255) / 256, N), dim3(256), 0, pvo_stream(stream),
                     poses, disps, intrinsics, points, ht * wd, wd);
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_depth_filter(const float* poses, const float* disps, const float* intrinsics,
                                const int64_t* ix, const float* thresh, float* counter,
                                int N, int nframes, int ht, int wd, void* stream) {
  PVO_REQ(N >= 0 && ht >= 0 && wd >= 0 && nframes >= 0);
  if (N == 0 || ht * wd == 0) return PVO_OK;
  PVO_REQ(poses && disps && intrinsics && ix && thresh && counter && N <= 65535);
  hipLaunchKernelGGL(depth_filter_kernel, dim3((ht * wd + 255) / 256, 1, N), dim3(256), 0, pvo_stream(stream),
                     poses, disps, intrinsics, ix, thresh, counter, nframes, ht, wd);
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

template <typename T> __device__ __forceinline__ uint32_t bits16(float x);
template <> __device__ __forceinline__ uint32_t bits16<pvo_half>(float x) {
  union { _Float16 h; uint16_t u; } c; c.h = static_cast<_Float16>(x); return c.u;
}
template <> __device__ __forceinline__ uint32_t bits16<pvo_bf16>(float x) { return pvo_f32_to_bf16(x); }

// reproject + FactorGraph.update's motion features (graph_glue.hip: graph_motion_kernel, the same arithmetic on the same
// values) in one pass: alone at the head of a graph update this costs what the reprojection costs, while the separate motion
// kernel ran for 20-24 us on the side stream beside the correlation lookup, which saturates the memory system's request rate.
template <typename T>
__global__ __launch_bounds__(256) void reproject_motion_kernel(
    const float* __restrict__ poses, const float* __restrict__ disps, const float* __restrict__ intrinsics,
    const int64_t* __restrict__ ii, const int64_t* __restrict__ jj,
    float* __restrict__ coords, float* __restrict__ valid,
    const float2* __restrict__ target, const float2* __restrict__ delta_dy, const float2* __restrict__ raw_mask,
    uint16_t* __restrict__ motn, int HW, int wd) {
  const int e = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= HW) return;
  const long long idx = static_cast<long long>(e) * HW + k;
  const float2 t = target[idx], dd = delta_dy[idx], m = raw_mask[idx];
  const int ix = static_cast<int>(ii[e]), jx = static_cast<int>(jj[e]);
  const Intr Ki = load_intr(intrinsics + 4 * static_cast<long long>(ix));
  const Intr Kj = load_intr(intrinsics + 4 * static_cast<long long>(jx));
  const Pose G = rel_pose(load_pose(poses + 7 * static_cast<long long>(ix)), load_pose(poses + 7 * static_cast<long long>(jx)));
  const int i = k / wd, j = k - i * wd;
  float X0[4] = {(static_cast<float>(j) - Ki.cx) / Ki.fx, (static_cast<float>(i) - Ki.cy) / Ki.fy, 1.0f,
                 disps[static_cast<long long>(ix) * HW + k]};
  float X1[4];
  act4(G, X0, X1);
  float Z = X1[2];
  Z = (Z < 0.5f * kMinDepthPy) ? 1.0f : Z;   // projective_ops.py:48
  const float d = 1.0f / Z;
  float2 c;
  c.x = Kj.fx * (X1[0] * d) + Kj.cx;
  c.y = Kj.fy * (X1[1] * d) + Kj.cy;
  *reinterpret_cast<float2*>(coords + idx * 2) = c;
  valid[idx] = (X1[2] > kMinDepthPy && X0[2] > kMinDepthPy) ? 1.0f : 0.0f;
  // factor_graph.py:233-237: motn = clamp(cat[target - coords0, target - coords0 + delta_dy, target - coords1, raw_mask], +-64)
  const float x0 = static_cast<float>(j), y0 = static_cast<float>(i);
  const float f[8] = {t.x - x0, t.y - y0, t.x - x0 + dd.x, t.y - y0 + dd.y, t.x - c.x, t.y - c.y, m.x, m.y};
  uint32_t o[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float a = fminf(fmaxf(f[2 * q], -64.0f), 64.0f), b = fminf(fmaxf(f[2 * q + 1], -64.0f), 64.0f);
    o[q] = bits16<T>(a) | (bits16<T>(b) << 16);
  }
  *reinterpret_cast<uint4*>(motn + idx * 8) = make_uint4(o[0], o[1], o[2], o[3]);
}

extern "C" int pvo_reproject_motion(const float* poses, const float* disps, const float* intrinsics,
                                    const int64_t* ii, const int64_t* jj, float* coords, float* valid,
                                    const float* target, const float* delta_dy, const float* raw_mask, void* motn,
                                    int E, int ht, int wd, int dtype, void* stream) {
  PVO_REQ(E >= 0 && ht >= 0 && wd >= 0);
  if (E == 0 || ht * wd == 0) return PVO_OK;
  PVO_REQ(poses && disps && intrinsics && ii && jj && coords && valid && target && delta_dy && raw_mask && motn && E <= 65535);
  PVO_REQ(!(reinterpret_cast<uintptr_t>(motn) & 15) && !((reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(delta_dy) |
           reinterpret_cast<uintptr_t>(raw_mask) | reinterpret_cast<uintptr_t>(coords)) & 7));
  const dim3 grid((ht * wd + 255) / 256, E);
  auto f2 = [](const float* p) { return reinterpret_cast<const float2*>(p); };
  if (dtype == PVO_F16)
    hipLaunchKernelGGL(reproject_motion_kernel<pvo_half>, grid, dim3(256), 0, pvo_stream(stream), poses, disps, intrinsics, ii, jj, coords, valid,
                       f2(target), f2(delta_dy), f2(raw_mask), static_cast<uint16_t*>(motn), ht * wd, wd);
  else if (dtype == PVO_BF16)
    hipLaunchKernelGGL(reproject_motion_kernel<pvo_bf16>, grid, dim3(256), 0, pvo_stream(stream), poses, disps, intrinsics, ii, jj, coords, valid,
                       f2(target), f2(delta_dy), f2(raw_mask), static_cast<uint16_t*>(motn), ht * wd, wd);
  else return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_reproject(const float* poses, const float* disps, const float* intrinsics,
                             const int64_t* ii, const int64_t* jj, float* coords, float* valid,
                             int E, int ht, int wd, void* stream) {
  PVO_REQ(E >= 0 && ht >= 0 && wd >= 0);
  if (E == 0 || ht * wd == 0) return PVO_OK;
  PVO_REQ(poses && disps && intrinsics && ii && jj && coords && valid && E <= 65535);
  hipLaunchKernelGGL(reproject_kernel, dim3((ht * wd + 255) / 256, E), dim3(256), 0, pvo_stream(stream),
                     poses, disps, intrinsics, ii, jj, coords, valid, ht * wd, wd);
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}
