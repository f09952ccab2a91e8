// se3_ops.hip — the SE3 subset of lietorch the VO path uses, as fused element-wise kernels.
//
// Replaces the lietorch CUDA element-wise kernels (reference VO_Module/thirdparty/lietorch/lietorch/src/lietorch_gpu.cu:21-296;
// group maths lietorch/include/se3.h:36-56,84-86,124-142, so3.h:55-60,115-208, EPS = 1e-6 common.h:7): one thread per group
// element, fp32 and fp64.  lietorch broadcasts by MATERIALISING the pose once per pixel (`broadcast_inputs` -> .repeat,
// broadcasting.py:27-29); here an operand with fewer elements is indexed i / rep (a pose per edge acting on H*W points),
// so nothing is copied.
// Backward (round 4; lietorch: hand-written backward kernels, lietorch_gpu.cu:21-296 / groups.py:141-178): pvo_se3_vjp, the
// vector-Jacobian product of every operation in AMBIENT coordinates (the 7 / 6 / 4 / 3 numbers of each operand) - what
// torch.autograd computes for the PyTorch formulation of the same formulas in pvo_amd/geom/se3.py, which stays the reference
// the kernels are tested against.  Not derived by hand: the forward templates below are instantiated with a forward-mode dual
// number and evaluated once per input component (at most 14 evaluations of a few dozen flops; the operations are launch- and
// memory-bound), so a derivative can only be wrong if the forward formula is.
//   data layout [n,7] = (tx,ty,tz, qx,qy,qz,qw); tangent (tau, phi).
#include "common.h"
#include <type_traits>

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

// forward-mode dual number over F (value, derivative along ONE seeded input direction)
template <typename F> struct Dual {
  F v, d;
  __device__ __forceinline__ Dual() : v(0), d(0) {}
  __device__ __forceinline__ Dual(F v_) : v(v_), d(0) {}
  __device__ __forceinline__ Dual(F v_, F d_) : v(v_), d(d_) {}
  template <typename C, typename = typename std::enable_if<std::is_arithmetic<C>::value && !std::is_same<C, F>::value>::type>
  __device__ __forceinline__ explicit Dual(C c) : v(static_cast<F>(c)), d(0) {}      // literals: F(2), F(0.5), F(kEps)
};
template <typename F> struct ScalarOf { using type = F; };
template <typename F> struct ScalarOf<Dual<F>> { using type = F; };
using ::sqrt; using ::sin; using ::cos; using ::atan; using ::fabs;      // (the overloads below must not hide the float / double ones)
#define DU __device__ __forceinline__
template <typename F> DU Dual<F> operator+(Dual<F> a, Dual<F> b) { return {a.v + b.v, a.d + b.d}; }
template <typename F> DU Dual<F> operator-(Dual<F> a, Dual<F> b) { return {a.v - b.v, a.d - b.d}; }
template <typename F> DU Dual<F> operator-(Dual<F> a) { return {-a.v, -a.d}; }
template <typename F> DU Dual<F> operator*(Dual<F> a, Dual<F> b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
template <typename F> DU Dual<F> operator/(Dual<F> a, Dual<F> b) { const F q = a.v / b.v; return {q, (a.d - q * b.d) / b.v}; }
template <typename F> DU bool operator<(Dual<F> a, Dual<F> b) { return a.v < b.v; }
template <typename F> DU bool operator>(Dual<F> a, Dual<F> b) { return a.v > b.v; }
template <typename F> DU Dual<F> sqrt(Dual<F> a) { const F r = sqrt(a.v); return {r, a.d / (F(2) * r)}; }
template <typename F> DU Dual<F> sin(Dual<F> a) { return {sin(a.v), cos(a.v) * a.d}; }
template <typename F> DU Dual<F> cos(Dual<F> a) { return {cos(a.v), -sin(a.v) * a.d}; }
template <typename F> DU Dual<F> atan(Dual<F> a) { return {atan(a.v), a.d / (F(1) + a.v * a.v)}; }
template <typename F> DU Dual<F> fabs(Dual<F> a) { return a.v < F(0) ? Dual<F>{-a.v, -a.d} : a; }
#undef DU

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

// ---- the remaining operations as templates over the scalar (F or Dual<F>): inputs / outputs as small arrays ----------------
template <typename F> __device__ __forceinline__ void se3_inv(const F* g, F* out) {
  const V3<F> t = {g[0], g[1], g[2]};
  const Q4<F> qi = qconj(Q4<F>{g[3], g[4], g[5], g[6]});
  const V3<F> r = qrot(qi, t);
  out[0] = -r.x; out[1] = -r.y; out[2] = -r.z; out[3] = qi.x; out[4] = qi.y; out[5] = qi.z; out[6] = qi.w;
}
// op in {MUL, ACT4, ACT3, ADJ, ADJT}: g [7] group element, x the second operand ([7] / [4] / [3] / [6]), out [7] / [4] / [3] / [6]
template <typename F> __device__ __forceinline__ void se3_bin(int op, const F* g, const F* x, F* out) {
  const V3<F> t = {g[0], g[1], g[2]};
  const Q4<F> q = {g[3], g[4], g[5], g[6]};
  if (op == OP_MUL) {
    const V3<F> r = qrot(q, V3<F>{x[0], x[1], x[2]});
    const Q4<F> qq = qmul(q, Q4<F>{x[3], x[4], x[5], x[6]});
    out[0] = t.x + r.x; out[1] = t.y + r.y; out[2] = t.z + r.z; out[3] = qq.x; out[4] = qq.y; out[5] = qq.z; out[6] = qq.w;
  } else if (op == OP_ACT4) {
    const V3<F> r = qrot(q, V3<F>{x[0], x[1], x[2]});
    out[0] = r.x + t.x * x[3]; out[1] = r.y + t.y * x[3]; out[2] = r.z + t.z * x[3]; out[3] = x[3];
  } else if (op == OP_ACT3) {
    const V3<F> r = qrot(q, V3<F>{x[0], x[1], x[2]});
    out[0] = r.x + t.x; out[1] = r.y + t.y; out[2] = r.z + t.z;
  } else if (op == OP_ADJ) {
    const V3<F> rphi = qrot(q, V3<F>{x[3], x[4], x[5]}), rt = qrot(q, V3<F>{x[0], x[1], x[2]}), c = cross(t, rphi);
    out[0] = rt.x + c.x; out[1] = rt.y + c.y; out[2] = rt.z + c.z; out[3] = rphi.x; out[4] = rphi.y; out[5] = rphi.z;
  } else {                                         // adjT
    const Q4<F> qi = qconj(q);
    const V3<F> at = {x[0], x[1], x[2]};
    const V3<F> r0 = qrot(qi, at), r1 = qrot(qi, V3<F>{x[3], x[4], x[5]}), r2 = qrot(qi, cross(at, t));
    out[0] = r0.x; out[1] = r0.y; out[2] = r0.z; out[3] = r1.x + r2.x; out[4] = r1.y + r2.y; out[5] = r1.z + r2.z;
  }
}
__host__ __device__ __forceinline__ int se3_nb(int op) { return op == OP_MUL ? 7 : (op == OP_ACT4 ? 4 : (op == OP_ACT3 ? 3 : 6)); }      // second operand = output size
__host__ __device__ __forceinline__ int se3_nin(int op) { return op == OP_EXP ? 6 : 7; }
__host__ __device__ __forceinline__ int se3_nout(int op) { return op == OP_EXP ? 7 : (op == OP_LOG ? 6 : 7); }

// Vector-Jacobian products.  Unary: gx[i, :] = gy[i, :] . d op(x[i]) / dx.  Binary: the per-ELEMENT products ga [n,7] and gb [n,nb]
// (element i reads a[i / rep_a], b[i / rep_b] as in the forward kernel); the caller sums ga / gb over the repeats of a broadcast
// operand.  One forward evaluation on dual numbers per input component.
template <typename F>
__global__ __launch_bounds__(256) void se3_unary_vjp_kernel(int op, const F* __restrict__ x, const F* __restrict__ gy, F* __restrict__ gx, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const int ni = se3_nin(op), no = se3_nout(op);
  F xv[7], g[7];
  for (int k = 0; k < ni; ++k) xv[k] = x[i * ni + k];
  for (int k = 0; k < no; ++k) g[k] = gy[i * no + k];
#pragma unroll 1
  for (int j = 0; j < ni; ++j) {
    Dual<F> in[7], out[7];
    for (int k = 0; k < 7; ++k) in[k] = Dual<F>(k < ni ? xv[k] : F(0), k == j ? F(1) : F(0));
    if (op == OP_EXP) se3_exp(in, out); else if (op == OP_LOG) se3_log(in, out); else se3_inv(in, out);
    F acc = 0;
    for (int k = 0; k < no; ++k) acc += g[k] * out[k].d;
    gx[i * ni + j] = acc;
  }
}
template <typename F>
__global__ __launch_bounds__(256) void se3_binary_vjp_kernel(int op, const F* __restrict__ a, long long rep_a, const F* __restrict__ b, long long rep_b,
                                                             const F* __restrict__ gy, F* __restrict__ ga, F* __restrict__ gb, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const int nb = se3_nb(op);
  F av[7], bv[7], g[7];
  for (int k = 0; k < 7; ++k) av[k] = a[(i / rep_a) * 7 + k];
  for (int k = 0; k < nb; ++k) { bv[k] = b[(i / rep_b) * nb + k]; g[k] = gy[i * nb + k]; }
#pragma unroll 1
  for (int j = 0; j < 7 + nb; ++j) {
    Dual<F> A[7], B[7], out[7];
    for (int k = 0; k < 7; ++k) A[k] = Dual<F>(av[k], k == j ? F(1) : F(0));
    for (int k = 0; k < 7; ++k) B[k] = Dual<F>(k < nb ? bv[k] : F(0), k + 7 == j ? F(1) : F(0));
    se3_bin(op, A, B, out);
    F acc = 0;
    for (int k = 0; k < nb; ++k) acc += g[k] * out[k].d;
    if (j < 7) { if (ga) ga[i * 7 + j] = acc; } else if (gb) gb[i * nb + (j - 7)] = acc;
  }
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
    F g[7], o[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) g[k] = x[i * 7 + k];
    se3_inv(g, o);
#pragma unroll
    for (int k = 0; k < 7; ++k) y[i * 7 + k] = o[k];
  }
}

// element i reads a[i / rep_a] (a group element) and b[i / rep_b]
template <typename F>
__global__ __launch_bounds__(256) void se3_binary_kernel(int op, const F* __restrict__ a, long long rep_a, const F* __restrict__ b,
                                                         long long rep_b, F* __restrict__ y, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const int nb = se3_nb(op);
  F g[7], x[7], o[7];
  for (int k = 0; k < 7; ++k) g[k] = a[(i / rep_a) * 7 + k];
  for (int k = 0; k < nb; ++k) x[k] = b[(i / rep_b) * nb + k];
  se3_bin(op, g, x, o);
  for (int k = 0; k < nb; ++k) y[i * nb + k] = o[k];
}

// ---- projective_transform (VO_Module/droid_slam/geom/projective_ops.py:106-130: iproj :21-41, actp :76-103, proj :44-73) ----------
// The training path's reprojection i -> j with its closed-form Jacobians, ONE kernel per direction instead of the ~90 element-wise
// operators of the PyTorch formulation (pvo_amd/geom/projective_ops.py, which stays the reference these are tested against and
// what CPU tensors use).  Forward: one thread per (batch, edge, pixel).  Backward: the same template on dual numbers, one
// evaluation per input component (7 + 7 pose numbers, the pixel's inverse depth) - ambient-coordinate vector-Jacobian products,
// as for the group operations above; the pose gradients of a workgroup's pixels are summed in LDS before the atomics.
//   out[0..2]  x, y, (inverse depth in frame j)      out[3..14]  Jj [2][6]      out[15..26]  Ji [2][6]      out[27..28]  Jz [2]
// returns Z (the validity test Z > MIN_DEPTH is the caller's: :113)
constexpr double kMinDepth = 0.2;
template <typename F, typename S>
__device__ __forceinline__ F proj_terms(const F* Pi, const F* Pj, F d, S u, S v, const S* Ki, const S* Kj, bool jac, F* out) {
  F Pinv[7], G[7], X0[4], X1[4];
  se3_inv(Pi, Pinv);
  se3_bin(OP_MUL, Pj, Pinv, G);
  X0[0] = F((u - Ki[2]) / Ki[0]); X0[1] = F((v - Ki[3]) / Ki[1]); X0[2] = F(S(1)); X0[3] = d;
  se3_bin(OP_ACT4, G, X0, X1);
  const F X = X1[0], Y = X1[1], Z = X1[2], W = X1[3];
  const F fx = F(Kj[0]), fy = F(Kj[1]);
  const F dinv = F(S(1)) / (Z < F(S(0.5 * kMinDepth)) ? F(S(1)) : Z);
  out[0] = fx * (X * dinv) + F(Kj[2]);
  out[1] = fy * (Y * dinv) + F(Kj[3]);
  out[2] = W * dinv;
  if (!jac) return Z;
  const F fxd = fx * dinv, fyd = fy * dinv;
  const F gx = -(fx * X * dinv * dinv), gy = -(fy * Y * dinv * dinv);
  const F o = F(S(0));
  F* Jj = out + 3;
  Jj[0] = fxd * W; Jj[1] = o; Jj[2] = gx * W; Jj[3] = gx * Y; Jj[4] = fxd * Z - gx * X; Jj[5] = -(fxd * Y);
  Jj[6] = o; Jj[7] = fyd * W; Jj[8] = gy * W; Jj[9] = gy * Y - fyd * Z; Jj[10] = -(gy * X); Jj[11] = fyd * X;
  F* Ji = out + 15;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    F a[6];
    se3_bin(OP_ADJT, G, Jj + 6 * r, a);
#pragma unroll
    for (int k = 0; k < 6; ++k) Ji[6 * r + k] = -a[k];
  }
  out[27] = fxd * G[0] + gx * G[2];
  out[28] = fyd * G[1] + gy * G[2];
  return Z;
}

// poses [B,P,7], depths [B,P,HW], intr [B,P,4], ii / jj [N]; x1 [B,N,HW,nx] (nx = 2 or 3), valid [B,N,HW], Jj / Ji [B,N,HW,2,6], Jz [B,N,HW,2]
template <typename F>
__global__ __launch_bounds__(256) void proj_fwd_kernel(const F* __restrict__ poses, const F* __restrict__ depths, const F* __restrict__ intr,
                                                       const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, int P, int N, int HW, int wd, int nx,
                                                       F* __restrict__ x1, F* __restrict__ valid, F* __restrict__ Ji, F* __restrict__ Jj, F* __restrict__ Jz) {
  const int px = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y, b = blockIdx.z;
  if (px >= HW) return;
  // frame indices as torch indexing reads them: a negative index counts from the end; anything still outside [0, P) - which the
  // PyTorch formulation answers with an IndexError - leaves this edge NaN / invalid instead of reading beside the buffers
  long long i = ii[n], j = jj[n];
  i += (i < 0) ? P : 0; j += (j < 0) ? P : 0;
  if (i < 0 || i >= P || j < 0 || j >= P) {
    const long long row = (static_cast<long long>(b) * N + n) * HW + px;
    const F nan = F(0) / F(0);
    for (int k = 0; k < nx; ++k) x1[row * nx + k] = nan;
    valid[row] = F(0);
    if (Jj) {
      for (int k = 0; k < 12; ++k) { Jj[row * 12 + k] = nan; Ji[row * 12 + k] = nan; }
      Jz[row * 2] = nan; Jz[row * 2 + 1] = nan;
    }
    return;
  }
  const F* Pi = poses + (static_cast<long long>(b) * P + i) * 7;
  const F* Pj = poses + (static_cast<long long>(b) * P + j) * 7;
  F pi[7], pj[7], Ki[4], Kj[4], out[29];
#pragma unroll
  for (int k = 0; k < 7; ++k) { pi[k] = Pi[k]; pj[k] = Pj[k]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) { Ki[k] = intr[(static_cast<long long>(b) * P + i) * 4 + k]; Kj[k] = intr[(static_cast<long long>(b) * P + j) * 4 + k]; }
  const F d = depths[(static_cast<long long>(b) * P + i) * HW + px];
  const F Z = proj_terms<F, F>(pi, pj, d, F(px % wd), F(px / wd), Ki, Kj, Jj != nullptr, out);
  const long long row = (static_cast<long long>(b) * N + n) * HW + px;
  for (int k = 0; k < nx; ++k) x1[row * nx + k] = out[k];
  valid[row] = Z > F(kMinDepth) ? F(1) : F(0);
  if (Jj) {
#pragma unroll
    for (int k = 0; k < 12; ++k) { Jj[row * 12 + k] = out[3 + k]; Ji[row * 12 + k] = out[15 + k]; }
    Jz[row * 2] = out[27]; Jz[row * 2 + 1] = out[28];
  }
}

// g_* : the incoming gradients of x1 / Ji / Jj / Jz (NULL = none); gposes [B,P,7] and gdepths [B,P,HW] are ACCUMULATED into (zeroed by the caller)
template <typename F>
__global__ __launch_bounds__(256) void proj_vjp_kernel(const F* __restrict__ poses, const F* __restrict__ depths, const F* __restrict__ intr,
                                                       const int64_t* __restrict__ ii, const int64_t* __restrict__ jj, int P, int N, int HW, int wd, int nx,
                                                       const F* __restrict__ g_x1, const F* __restrict__ g_Ji, const F* __restrict__ g_Jj, const F* __restrict__ g_Jz,
                                                       F* __restrict__ gposes, F* __restrict__ gdepths) {
  __shared__ F red[4][14];
  const int px = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y, b = blockIdx.z;
  long long i = ii[n], j = jj[n];                           // (as in the forward kernel: negatives wrap, an edge outside [0, P) adds nothing)
  i += (i < 0) ? P : 0; j += (j < 0) ? P : 0;
  if (i < 0 || i >= P || j < 0 || j >= P) return;           // (uniform over the workgroup: n is blockIdx.y)
  const bool jac = g_Ji || g_Jj || g_Jz;
  F s[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) s[k] = F(0);
  if (px < HW) {
    F pi[7], pj[7], Ki[4], Kj[4], g[29];
#pragma unroll
    for (int k = 0; k < 7; ++k) { pi[k] = poses[(static_cast<long long>(b) * P + i) * 7 + k]; pj[k] = poses[(static_cast<long long>(b) * P + j) * 7 + k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { Ki[k] = intr[(static_cast<long long>(b) * P + i) * 4 + k]; Kj[k] = intr[(static_cast<long long>(b) * P + j) * 4 + k]; }
    const F d = depths[(static_cast<long long>(b) * P + i) * HW + px];
    const long long row = (static_cast<long long>(b) * N + n) * HW + px;
#pragma unroll
    for (int k = 0; k < 29; ++k) g[k] = F(0);
    if (g_x1) for (int k = 0; k < nx; ++k) g[k] = g_x1[row * nx + k];
    if (g_Jj) for (int k = 0; k < 12; ++k) g[3 + k] = g_Jj[row * 12 + k];
    if (g_Ji) for (int k = 0; k < 12; ++k) g[15 + k] = g_Ji[row * 12 + k];
    if (g_Jz) { g[27] = g_Jz[row * 2]; g[28] = g_Jz[row * 2 + 1]; }
    const F u = F(px % wd), v = F(px / wd);
#pragma unroll 1
    for (int q = 0; q < 15; ++q) {
      Dual<F> A[7], Bq[7], out[29];
#pragma unroll
      for (int k = 0; k < 7; ++k) { A[k] = Dual<F>(pi[k], k == q ? F(1) : F(0)); Bq[k] = Dual<F>(pj[k], k + 7 == q ? F(1) : F(0)); }
      const Dual<F> dd(d, q == 14 ? F(1) : F(0));
      proj_terms<Dual<F>, F>(A, Bq, dd, u, v, Ki, Kj, jac, out);
      F acc = F(0);
      const int nout = jac ? 29 : 3;
      for (int k = 0; k < nout; ++k) acc += g[k] * out[k].d;
      s[q] = acc;
    }
    // (i == j: both seeds move the same pose; the two partial derivatives add, which the two accumulations below do)
    atomicAdd(&gdepths[(static_cast<long long>(b) * P + i) * HW + px], s[14]);
  }
  // the 14 pose sums of this workgroup's pixels: wave shuffle, then across the four waves through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 14; ++k) {
    F v = s[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 14) {
    const F v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    const int k = threadIdx.x;
    F* dst = gposes + (static_cast<long long>(b) * P + (k < 7 ? i : j)) * 7 + (k < 7 ? k : k - 7);
    atomicAdd(dst, v);
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

// Backward of pvo_se3_unary / pvo_se3_binary (ambient-coordinate vector-Jacobian products; see the head of this file).
//   unary : gx [n, in] = gy [n, out] . J          binary: ga [n,7], gb [n,nb] PER OUTPUT ELEMENT (either may be NULL); the caller sums
//   them over the `rep` repeats of a broadcast operand (element i belongs to a[i / rep_a], b[i / rep_b]).
extern "C" int pvo_se3_unary_vjp(int op, const void* x, const void* gy, void* gx, long long n, int dtype, void* stream) {
  if (n < 0 || op < OP_EXP || op > OP_INV) return PVO_EINVAL;
  if (n == 0) return PVO_OK;
  if (!x || !gy || !gx) return PVO_EINVAL;
  const dim3 grid(static_cast<unsigned>((n + 255) / 256));
  if (dtype == PVO_F32)
    hipLaunchKernelGGL(se3_unary_vjp_kernel<float>, grid, dim3(256), 0, pvo_stream(stream), op, static_cast<const float*>(x), static_cast<const float*>(gy), static_cast<float*>(gx), n);
  else if (dtype == PVO_F64)
    hipLaunchKernelGGL(se3_unary_vjp_kernel<double>, grid, dim3(256), 0, pvo_stream(stream), op, static_cast<const double*>(x), static_cast<const double*>(gy), static_cast<double*>(gx), n);
  else return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_se3_binary_vjp(int op, const void* a, long long rep_a, const void* b, long long rep_b, const void* gy,
                                  void* ga, void* gb, long long n, int dtype, void* stream) {
  if (n < 0 || op < OP_MUL || op > OP_ADJT || rep_a <= 0 || rep_b <= 0) return PVO_EINVAL;
  if (n == 0) return PVO_OK;
  if (!a || !b || !gy || (!ga && !gb)) return PVO_EINVAL;
  const dim3 grid(static_cast<unsigned>((n + 255) / 256));
  if (dtype == PVO_F32)
    hipLaunchKernelGGL(se3_binary_vjp_kernel<float>, grid, dim3(256), 0, pvo_stream(stream), op, static_cast<const float*>(a), rep_a, static_cast<const float*>(b), rep_b,
                       static_cast<const float*>(gy), static_cast<float*>(ga), static_cast<float*>(gb), n);
  else if (dtype == PVO_F64)
    hipLaunchKernelGGL(se3_binary_vjp_kernel<double>, grid, dim3(256), 0, pvo_stream(stream), op, static_cast<const double*>(a), rep_a, static_cast<const double*>(b), rep_b,
                       static_cast<const double*>(gy), static_cast<double*>(ga), static_cast<double*>(gb), n);
  else return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

// projective_transform forward / backward (see proj_terms above).  Jj == NULL: coordinates and validity only.
extern "C" int pvo_proj_transform(const void* poses, const void* depths, const void* intr, const int64_t* ii, const int64_t* jj,
                                  int B, int P, int N, int ht, int wd, int nx, void* x1, void* valid, void* Ji, void* Jj, void* Jz,
                                  int dtype, void* stream) {
  if (B < 0 || P <= 0 || N < 0 || ht <= 0 || wd <= 0 || (nx != 2 && nx != 3)) return PVO_EINVAL;
  if (B == 0 || N == 0) return PVO_OK;
  if (!poses || !depths || !intr || !ii || !jj || !x1 || !valid) return PVO_EINVAL;
  if ((Jj != nullptr) != (Ji != nullptr) || (Jj != nullptr) != (Jz != nullptr)) return PVO_EINVAL;
  if (N > 65535 || B > 65535) return PVO_EUNSUPPORTED;      // an edge / batch index is a grid coordinate (include/pvo_hip.h "Limits")
  const int HW = ht * wd;
  const dim3 grid((HW + 255) / 256, N, B);
  if (dtype == PVO_F32)
    hipLaunchKernelGGL(proj_fwd_kernel<float>, grid, dim3(256), 0, pvo_stream(stream), static_cast<const float*>(poses), static_cast<const float*>(depths),
                       static_cast<const float*>(intr), ii, jj, P, N, HW, wd, nx, static_cast<float*>(x1), static_cast<float*>(valid),
                       static_cast<float*>(Ji), static_cast<float*>(Jj), static_cast<float*>(Jz));
  else if (dtype == PVO_F64)
    hipLaunchKernelGGL(proj_fwd_kernel<double>, grid, dim3(256), 0, pvo_stream(stream), static_cast<const double*>(poses), static_cast<const double*>(depths),
                       static_cast<const double*>(intr), ii, jj, P, N, HW, wd, nx, static_cast<double*>(x1), static_cast<double*>(valid),
                       static_cast<double*>(Ji), static_cast<double*>(Jj), static_cast<double*>(Jz));
  else return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

// gposes [B,P,7] and gdepths [B,P,ht*wd] must be ZERO on entry (the products of all edges and pixels are added into them)
extern "C" int pvo_proj_transform_vjp(const void* poses, const void* depths, const void* intr, const int64_t* ii, const int64_t* jj,
                                      int B, int P, int N, int ht, int wd, int nx, const void* g_x1, const void* g_Ji, const void* g_Jj, const void* g_Jz,
                                      void* gposes, void* gdepths, int dtype, void* stream) {
  if (B < 0 || P <= 0 || N < 0 || ht <= 0 || wd <= 0 || (nx != 2 && nx != 3)) return PVO_EINVAL;
  if (B == 0 || N == 0) return PVO_OK;
  if (!poses || !depths || !intr || !ii || !jj || !gposes || !gdepths) return PVO_EINVAL;
  if (!g_x1 && !g_Ji && !g_Jj && !g_Jz) return PVO_OK;
  if (N > 65535 || B > 65535) return PVO_EUNSUPPORTED;
  const int HW = ht * wd;
  const dim3 grid((HW + 255) / 256, N, B);
  if (dtype == PVO_F32)
    hipLaunchKernelGGL(proj_vjp_kernel<float>, grid, dim3(256), 0, pvo_stream(stream), static_cast<const float*>(poses), static_cast<const float*>(depths),
                       static_cast<const float*>(intr), ii, jj, P, N, HW, wd, nx, static_cast<const float*>(g_x1), static_cast<const float*>(g_Ji),
                       static_cast<const float*>(g_Jj), static_cast<const float*>(g_Jz), static_cast<float*>(gposes), static_cast<float*>(gdepths));
  else if (dtype == PVO_F64)
    hipLaunchKernelGGL(proj_vjp_kernel<double>, grid, dim3(256), 0, pvo_stream(stream), static_cast<const double*>(poses), static_cast<const double*>(depths),
                       static_cast<const double*>(intr), ii, jj, P, N, HW, wd, nx, static_cast<const double*>(g_x1), static_cast<const double*>(g_Ji),
                       static_cast<const double*>(g_Jj), static_cast<const double*>(g_Jz), static_cast<double*>(gposes), static_cast<double*>(gdepths));
  else return PVO_EUNSUPPORTED;
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}
