// se3_ops.hip — the SE3 subset of lietorch the VO path uses, as fused element-wise kernels.
//
// Replaces the lietorch CUDA element-wise kernels (reference VO_Module/thirdparty/lietorch/lietorch/src/lietorch_gpu.cu:21-296;
// group maths lietorch/include/se3.h:36-56,84-86,124-142, so3.h:55-60,115-208, EPS = 1e-6 common.h:7): one thread per group
// element, fp32 and fp64.  lietorch broadcasts by MATERIALISING the pose once per pixel (`broadcast_inputs` -> .repeat,
// broadcasting.py:27-29); here an operand with fewer elements is indexed i / rep (a pose per edge acting on H*W points),
// so nothing is copied.  Forward only: autograd goes through the PyTorch formulation in pvo_amd/geom/se3.py.
//   data layout [n,7] = (tx,ty,tz, qx,qy,qz,qw); tangent (tau, phi).
#include "common.h"

namespace {

template <typename F> struct V3 { F x, y, z; };
template <typename F> struct Q4 { F x, y, z, w; };

template <typename F> __device__ __forceinline__ V3<F> cross(V3<F> a, V3<F> b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename F> __device__ __forceinline__ V3<F> qrot(Q4<F> q, V3<F> v) {
  const V3<F> qv = {q.x, q.y, q.z};
  V3<F> uv = cross(qv, v);
  uv = {F(2) * uv.x, F(2) * uv.y, F(2) * uv.z};
  const V3<F> c = cross(qv, uv);
  return {v.x + q.w * uv.x + c.x, v.y + q.w * uv.y + c.y, v.z + q.w * uv.z + c.z};
}
template <typename F> __device__ __forceinline__ Q4<F> qmul(Q4<F> a, Q4<F> b) {
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
template <typename F> __device__ __forceinline__ Q4<F> qconj(Q4<F> q) { return {-q.x, -q.y, -q.z, q.w}; }

constexpr double kEps = 1e-6;
enum { OP_EXP = 0, OP_LOG = 1, OP_INV = 2, OP_MUL = 3, OP_ACT4 = 4, OP_ACT3 = 5, OP_ADJ = 6, OP_ADJT = 7 };

template <typename F> __device__ __forceinline__ void se3_exp(const F* xi, F* out) {
  const V3<F> tau = {xi[0], xi[1], xi[2]}, phi = {xi[3], xi[4], xi[5]};
  const F th2 = phi.x * phi.x + phi.y * phi.y + phi.z * phi.z, th = sqrt(th2);
  const bool small = th < F(kEps);
  const F ths = small ? F(1) : th, th2s = small ? F(1) : th2;
  const F imag = small ? F(0.5) - th2 / F(48) + th2 * th2 / F(3840) : sin(F(0.5) * ths) / ths;
  const F real = small ? F(1) - th2 / F(8) + th2 * th2 / F(384) : cos(F(0.5) * ths);
  const F c1 = small ? F(0.5) - th2 / F(24) : (F(1) - cos(ths)) / th2s;
  const F c2 = small ? F(1) / F(6) - th2 / F(120) : (ths - sin(ths)) / (th2s * ths);
  const V3<F> pt = cross(phi, tau), ppt = cross(phi, pt);
  out[0] = tau.x + c1 * pt.x + c2 * ppt.x; out[1] = tau.y + c1 * pt.y + c2 * ppt.y; out[2] = tau.z + c1 * pt.z + c2 * ppt.z;
  out[3] = imag * phi.x; out[4] = imag * phi.y; out[5] = imag * phi.z; out[6] = real;
}

template <typename F> __device__ __forceinline__ void se3_log(const F* g, F* out) {
  const V3<F> t = {g[0], g[1], g[2]}, v = {g[3], g[4], g[5]};
  const F w = g[6];
  const F n2 = v.x * v.x + v.y * v.y + v.z * v.z;
  const bool smallq = n2 < F(kEps * kEps);
  const F n = sqrt(smallq ? F(1) : n2);
  const F ws = fabs(w) < F(kEps) ? F(kEps) : w;
  F big = F(2) * atan(n / ws) / n;
  if (fabs(w) < F(kEps)) big = (w > F(0) ? F(3.14159265358979323846) : -F(3.14159265358979323846)) / n;
  const F sm = F(2) / w - (F(2) / F(3)) * n2 / (w * w * w);
  const F s = smallq ? sm : big;
  const V3<F> phi = {s * v.x, s * v.y, s * v.z};
  const F th2 = phi.x * phi.x + phi.y * phi.y + phi.z * phi.z, th = sqrt(th2);
  const bool small = th < F(kEps);
  const F ths = small ? F(1) : th, half = F(0.5) * ths;
  const F c2 = small ? F(1) / F(12) : (F(1) - ths * cos(half) / (F(2) * sin(half))) / (ths * ths);
  const V3<F> pt = cross(phi, t), ppt = cross(phi, pt);
  out[0] = t.x - F(0.5) * pt.x + c2 * ppt.x; out[1] = t.y - F(0.5) * pt.y + c2 * ppt.y; out[2] = t.z - F(0.5) * pt.z + c2 * ppt.z;
  out[3] = phi.x; out[4] = phi.y; out[5] = phi.z;
}

template <typename F>
__global__ __launch_bounds__(256) void se3_unary_kernel(int op, const F* __restrict__ x, F* __restrict__ y, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  if (op == OP_EXP) {
    F xi[6], o[7];
#pragma unroll
    for (int k = 0; k < 6; ++k) xi[k] = x[i * 6 + k];
    se3_exp(xi, o);
#pragma unroll
    for (int k = 0; k < 7; ++k) y[i * 7 + k] = o[k];
  } else if (op == OP_LOG) {
    F g[7], o[6];
#pragma unroll
    for (int k = 0; k < 7; ++k) g[k] = x[i * 7 + k];
    se3_log(g, o);
#pragma unroll
    for (int k = 0; k < 6; ++k) y[i * 6 + k] = o[k];
  } else {                                         // inverse
    const V3<F> t = {x[i * 7], x[i * 7 + 1], x[i * 7 + 2]};
    const Q4<F> qi = qconj(Q4<F>{x[i * 7 + 3], x[i * 7 + 4], x[i * 7 + 5], x[i * 7 + 6]});
    const V3<F> r = qrot(qi, t);
    y[i * 7] = -r.x; y[i * 7 + 1] = -r.y; y[i * 7 + 2] = -r.z;
    y[i * 7 + 3] = qi.x; y[i * 7 + 4] = qi.y; y[i * 7 + 5] = qi.z; y[i * 7 + 6] = qi.w;
  }
}

// element i reads a[i / rep_a] (a group element) and b[i / rep_b]
template <typename F>
__global__ __launch_bounds__(256) void se3_binary_kernel(int op, const F* __restrict__ a, long long rep_a, const F* __restrict__ b,
                                                         long long rep_b, F* __restrict__ y, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const F* g = a + (i / rep_a) * 7;
  const V3<F> t = {g[0], g[1], g[2]};
  const Q4<F> q = {g[3], g[4], g[5], g[6]};
  const long long ib = i / rep_b;
  if (op == OP_MUL) {
    const F* h = b + ib * 7;
    const V3<F> r = qrot(q, V3<F>{h[0], h[1], h[2]});
    const Q4<F> qq = qmul(q, Q4<F>{h[3], h[4], h[5], h[6]});
    y[i * 7] = t.x + r.x; y[i * 7 + 1] = t.y + r.y; y[i * 7 + 2] = t.z + r.z;
    y[i * 7 + 3] = qq.x; y[i * 7 + 4] = qq.y; y[i * 7 + 5] = qq.z; y[i * 7 + 6] = qq.w;
  } else if (op == OP_ACT4) {
    const F* p = b + ib * 4;
    const V3<F> r = qrot(q, V3<F>{p[0], p[1], p[2]});
    y[i * 4] = r.x + t.x * p[3]; y[i * 4 + 1] = r.y + t.y * p[3]; y[i * 4 + 2] = r.z + t.z * p[3]; y[i * 4 + 3] = p[3];
  } else if (op == OP_ACT3) {
    const F* p = b + ib * 3;
    const V3<F> r = qrot(q, V3<F>{p[0], p[1], p[2]});
    y[i * 3] = r.x + t.x; y[i * 3 + 1] = r.y + t.y; y[i * 3 + 2] = r.z + t.z;
  } else if (op == OP_ADJ) {
    const F* x = b + ib * 6;
    const V3<F> rphi = qrot(q, V3<F>{x[3], x[4], x[5]}), rt = qrot(q, V3<F>{x[0], x[1], x[2]}), c = cross(t, rphi);
    y[i * 6] = rt.x + c.x; y[i * 6 + 1] = rt.y + c.y; y[i * 6 + 2] = rt.z + c.z;
    y[i * 6 + 3] = rphi.x; y[i * 6 + 4] = rphi.y; y[i * 6 + 5] = rphi.z;
  } else {                                         // adjT
    const F* x = b + ib * 6;
    const Q4<F> qi = qconj(q);
    const V3<F> at = {x[0], x[1], x[2]};
    const V3<F> r0 = qrot(qi, at), r1 = qrot(qi, V3<F>{x[3], x[4], x[5]}), r2 = qrot(qi, cross(at, t));
    y[i * 6] = r0.x; y[i * 6 + 1] = r0.y; y[i * 6 + 2] = r0.z;
    y[i * 6 + 3] = r1.x + r2.x; y[i * 6 + 4] = r1.y + r2.y; y[i * 6 + 5] = r1.z + r2.z;
  }
}

}  // namespace

extern "C" int pvo_se3_unary(int op, const void* x, void* y, long long n, int dtype, void* stream) {
  if (n < 0 || op < OP_EXP || op > OP_INV) return PVO_EINVAL;
  if (n == 0) return PVO_OK;
  if (!x || !y) return PVO_EINVAL;
  const dim3 grid(static_cast<unsigned>((n + 255) / 256));
  if (dtype == PVO_F32)
    hipLaunchKernelGGL(se3_unary_kernel<float>, grid, dim3(256), 0, pvo_stream(stream), op, static_cast<const float*>(x), static_cast<float*>(y), n);
  else if (dtype == PVO_F64)
    hipLaunchKernelGGL(se3_unary_kernel<double>, grid, dim3(256), 0, pvo_stream(stream), op, static_cast<const double*>(x), static_cast<double*>(y), n);
  else return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_se3_binary(int op, const void* a, long long rep_a, const void* b, long long rep_b, void* y, long long n,
                              int dtype, void* stream) {
  if (n < 0 || op < OP_MUL || op > OP_ADJT || rep_a <= 0 || rep_b <= 0) return PVO_EINVAL;
  if (n == 0) return PVO_OK;
  if (!a || !b || !y) return PVO_EINVAL;
  const dim3 grid(static_cast<unsigned>((n + 255) / 256));
  if (dtype == PVO_F32)
    hipLaunchKernelGGL(se3_binary_kernel<float>, grid, dim3(256), 0, pvo_stream(stream), op, static_cast<const float*>(a), rep_a,
                       static_cast<const float*>(b), rep_b, static_cast<float*>(y), n);
  else if (dtype == PVO_F64)
    hipLaunchKernelGGL(se3_binary_kernel<double>, grid, dim3(256), 0, pvo_stream(stream), op, static_cast<const double*>(a), rep_a,
                       static_cast<const double*>(b), rep_b, static_cast<double*>(y), n);
  else return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}
