// se3.h — minimal SE(3) arithmetic used by the reprojection / BA kernels.
// Pose storage follows the reference: 7 floats, translation (tx,ty,tz) then unit
// quaternion (qx,qy,qz,qw); poses are world-to-camera.
// Math restated from the closed forms the reference uses
// (VO_Module/src/droid_kernels.cu:58-176, 856-874; lietorch include/se3.h, so3.h).
#pragma once
#include "common.h"

struct Quat { float x, y, z, w; };
struct Vec3 { float x, y, z; };
struct Pose { Vec3 t; Quat q; };

__device__ __forceinline__ Pose load_pose(const float* p) {
  Pose P; P.t = {p[0], p[1], p[2]}; P.q = {p[3], p[4], p[5], p[6]}; return P;
}

__device__ __forceinline__ Vec3 cross3(const Vec3 a, const Vec3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// rotate v by unit quaternion q:  v + w*uv + q_xyz x uv,  uv = 2 (q_xyz x v)
__device__ __forceinline__ Vec3 rotate(const Quat q, const Vec3 v) {
  const Vec3 qv = {q.x, q.y, q.z};
  Vec3 uv = cross3(qv, v);
  uv = {2.0f * uv.x, 2.0f * uv.y, 2.0f * uv.z};
  const Vec3 c = cross3(qv, uv);
  return {v.x + q.w * uv.x + c.x, v.y + q.w * uv.y + c.y, v.z + q.w * uv.z + c.z};
}

__device__ __forceinline__ Quat qconj(const Quat q) { return {-q.x, -q.y, -q.z, q.w}; }

// Hamilton product a*b
__device__ __forceinline__ Quat qmul(const Quat a, const Quat b) {
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x,
          a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}

// Gij = Gj * Gi^-1  (droid_kernels.cu:96-107 relSE3)
__device__ __forceinline__ Pose rel_pose(const Pose Gi, const Pose Gj) {
  Pose G;
  G.q = qmul(Gj.q, qconj(Gi.q));
  const Vec3 r = rotate(G.q, Gi.t);
  G.t = {Gj.t.x - r.x, Gj.t.y - r.y, Gj.t.z - r.z};
  return G;
}

// homogeneous point action [X,Y,Z,d] -> [R X + d t, d]  (droid_kernels.cu:70-77 actSE3)
__device__ __forceinline__ void act4(const Pose G, const float X[4], float Y[4]) {
  const Vec3 r = rotate(G.q, {X[0], X[1], X[2]});
  Y[0] = r.x + X[3] * G.t.x;
  Y[1] = r.y + X[3] * G.t.y;
  Y[2] = r.z + X[3] * G.t.z;
  Y[3] = X[3];
}

// Y = Ad(G)^T-style pull-back of a 6-covector (tau, phi) (droid_kernels.cu:79-94 adjSE3;
// lietorch SE3::adjT)
__device__ __forceinline__ void adjT(const Pose G, const float X[6], float Y[6]) {
  const Quat qi = qconj(G.q);
  const Vec3 a = rotate(qi, {X[0], X[1], X[2]});
  const Vec3 b = rotate(qi, {X[3], X[4], X[5]});
  const Vec3 u = {G.t.z * X[1] - G.t.y * X[2], G.t.x * X[2] - G.t.z * X[0], G.t.y * X[0] - G.t.x * X[1]};
  const Vec3 v = rotate(qi, u);
  Y[0] = a.x; Y[1] = a.y; Y[2] = a.z;
  Y[3] = b.x + v.x; Y[4] = b.y + v.y; Y[5] = b.z + v.z;
}

// quaternion exponential of a rotation vector (droid_kernels.cu:110-132 expSO3)
__device__ __forceinline__ Quat exp_so3(const Vec3 phi) {
  const float th2 = phi.x * phi.x + phi.y * phi.y + phi.z * phi.z;
  const float th4 = th2 * th2;
  const float th = sqrtf(th2);
  float imag, real;
  if (th2 < 1e-8f) {
    imag = 0.5f - (1.0f / 48.0f) * th2 + (1.0f / 3840.0f) * th4;
    real = 1.0f - (1.0f / 8.0f) * th2 + (1.0f / 384.0f) * th4;
  } else {
    imag = sinf(0.5f * th) / th;
    real = cosf(0.5f * th);
  }
  return {imag * phi.x, imag * phi.y, imag * phi.z, real};
}

// SE3 exponential (droid_kernels.cu:147-176 expSE3).  NOTE: the reference reads
// xi[45] for the z rotation component at :154 (an out-of-bounds typo; upstream
// DROID-SLAM has xi[5]).  This build implements the evident intent, xi[5].
__device__ __forceinline__ Pose exp_se3(const float xi[6]) {
  Pose G;
  const Vec3 phi = {xi[3], xi[4], xi[5]};
  G.q = exp_so3(phi);
  Vec3 tau = {xi[0], xi[1], xi[2]};
  const float th2 = phi.x * phi.x + phi.y * phi.y + phi.z * phi.z;
  const float th = sqrtf(th2);
  G.t = tau;
  if (th > 1e-4f) {
    const float a = (1.0f - cosf(th)) / th2;
    tau = cross3(phi, tau);
    G.t = {G.t.x + a * tau.x, G.t.y + a * tau.y, G.t.z + a * tau.z};
    const float b = (th - sinf(th)) / (th * th2);
    tau = cross3(phi, tau);
    G.t = {G.t.x + b * tau.x, G.t.y + b * tau.y, G.t.z + b * tau.z};
  }
  return G;
}

// retraction T <- Exp(xi) * T  (droid_kernels.cu:856-874 retrSE3)
__device__ __forceinline__ Pose retract(const float xi[6], const Pose T) {
  const Pose d = exp_se3(xi);
  Pose R;
  R.q = qmul(d.q, T.q);
  const Vec3 r = rotate(d.q, T.t);
  R.t = {r.x + d.t.x, r.y + d.t.y, r.z + d.t.z};
  return R;
}
