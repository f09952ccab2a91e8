/*
 * oracle_ba.c — CPU restatement of the reference's native bundle adjustment and
 * reprojection kernels.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Follows, step for step:
 *   actSO3/actSE3/adjSE3/relSE3/expSO3/expSE3/retrSE3   droid_kernels.cu:58-176, 856-874
 *   projective_transform_kernel (BA assembly)           droid_kernels.cu:177-403
 *   projmap_kernel                                      droid_kernels.cu:406-495
 *   frame_distance_kernel                               droid_kernels.cu:497-636
 *   depth_filter_kernel                                 droid_kernels.cu:640-754
 *   iproj_kernel                                        droid_kernels.cu:758-829
 *   accum_cuda / accum_kernel                           droid_kernels.cu:833-853, 927-977
 *   EEt6x6 / Ev6x1 / EvT6x1 kernels                     droid_kernels.cu:980-1094
 *   SparseBlock (update_lhs/rhs, operator-, solve)      droid_kernels.cu:1096-1198
 *   schur_block                                         droid_kernels.cu:1201-1290
 *   ba_cuda                                             droid_kernels.cu:1293-1410
 *   pops.projective_transform (jacobian=False)          geom/projective_ops.py:102-130
 *
 * PARITY PIN STATUS: droid_kernels.cu cannot be built here (no nvcc, and the
 * vendored Eigen lacks Eigen/Core), and the reference has no test vectors for this
 * path.  The restatement is pinned by the reference's second, independent BA
 * implementation geom/ba.py + geom/chol.py + geom/projective_ops.py, imported from
 * /root/reference (fixtures under tests/golden, generator tests/golden/gen_golden.py)
 * on inputs where the two reference paths coincide (all depths > 0.25, shared
 * intrinsics, t0 == fixedp), plus SE3 group identities (lietorch run_tests.py:16-54).
 *
 * Precision model: per-pixel quantities are fp32 exactly as the kernels compute
 * them; every SUM over pixels / edges is carried in fp64 (the reference's fp32
 * block-reduction order is a property of its launch shape, not of the algorithm);
 * the pose system is assembled and factorised in fp64 as SparseBlock does on the
 * host; results are cast to fp32 where the reference stores fp32.
 *
 * Deliberate deviation: expSE3 reads xi[45] at droid_kernels.cu:154 (out of bounds;
 * upstream DROID-SLAM has xi[5]).  `xi45_zero` != 0 reproduces "that read returned
 * 0"; the default (0) uses xi[5], the evident intent, which is what the build ships.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MIN_DEPTH 0.25f

/* ---------------- SE3 helpers (fp32, as on the device) ---------------- */
static void actSO3(const float* q, const float* X, float* Y) {
  float uv[3];
  uv[0] = 2.0f * (q[1] * X[2] - q[2] * X[1]);
  uv[1] = 2.0f * (q[2] * X[0] - q[0] * X[2]);
  uv[2] = 2.0f * (q[0] * X[1] - q[1] * X[0]);
  Y[0] = X[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  Y[1] = X[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  Y[2] = X[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}

static void actSE3(const float* t, const float* q, const float* X, float* Y) {
  actSO3(q, X, Y);
  Y[3] = X[3];
  Y[0] += X[3] * t[0];
  Y[1] += X[3] * t[1];
  Y[2] += X[3] * t[2];
}

static void adjSE3(const float* t, const float* q, const float* X, float* Y) {
  float qinv[4] = {-q[0], -q[1], -q[2], q[3]};
  float u[3], v[3];
  actSO3(qinv, &X[0], &Y[0]);
  actSO3(qinv, &X[3], &Y[3]);
  u[0] = t[2] * X[1] - t[1] * X[2];
  u[1] = t[0] * X[2] - t[2] * X[0];
  u[2] = t[1] * X[0] - t[0] * X[1];
  actSO3(qinv, u, v);
  Y[3] += v[0]; Y[4] += v[1]; Y[5] += v[2];
}

static void relSE3(const float* ti, const float* qi, const float* tj, const float* qj, float* tij, float* qij) {
  qij[0] = -qj[3] * qi[0] + qj[0] * qi[3] - qj[1] * qi[2] + qj[2] * qi[1];
  qij[1] = -qj[3] * qi[1] + qj[1] * qi[3] - qj[2] * qi[0] + qj[0] * qi[2];
  qij[2] = -qj[3] * qi[2] + qj[2] * qi[3] - qj[0] * qi[1] + qj[1] * qi[0];
  qij[3] = qj[3] * qi[3] + qj[0] * qi[0] + qj[1] * qi[1] + qj[2] * qi[2];
  actSO3(qij, ti, tij);
  tij[0] = tj[0] - tij[0];
  tij[1] = tj[1] - tij[1];
  tij[2] = tj[2] - tij[2];
}

static void expSO3(const float* phi, float* q) {
  float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  float theta_p4 = theta_sq * theta_sq;
  float theta = sqrtf(theta_sq);
  float imag, real;
  if (theta_sq < 1e-8) {
    imag = (float)(0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_p4);
    real = (float)(1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_p4);
  } else {
    imag = sinf(0.5f * theta) / theta;
    real = cosf(0.5f * theta);
  }
  q[0] = imag * phi[0]; q[1] = imag * phi[1]; q[2] = imag * phi[2]; q[3] = real;
}

static void crossInplace(const float* a, float* b) {
  float x[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  b[0] = x[0]; b[1] = x[1]; b[2] = x[2];
}

static void expSE3(const float* xi, float* t, float* q, int xi45_zero) {
  float tau[3] = {xi[0], xi[1], xi[2]};
  float phi[3] = {xi[3], xi[4], xi45_zero ? 0.0f : xi[5]};
  float theta_sq, theta;
  expSO3(xi + 3, q);
  theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  theta = sqrtf(theta_sq);
  t[0] = tau[0]; t[1] = tau[1]; t[2] = tau[2];
  if (theta > 1e-4) {
    float a = (1 - cosf(theta)) / theta_sq, b;
    crossInplace(phi, tau);
    t[0] += a * tau[0]; t[1] += a * tau[1]; t[2] += a * tau[2];
    b = (theta - sinf(theta)) / (theta * theta_sq);
    crossInplace(phi, tau);
    t[0] += b * tau[0]; t[1] += b * tau[1]; t[2] += b * tau[2];
  }
}

static void retrSE3(const float* xi, const float* t, const float* q, float* t1, float* q1, int xi45_zero) {
  float dt[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 1};
  expSE3(xi, dt, dq, xi45_zero);
  q1[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
  q1[1] = dq[3] * q[1] + dq[1] * q[3] + dq[2] * q[0] - dq[0] * q[2];
  q1[2] = dq[3] * q[2] + dq[2] * q[3] + dq[0] * q[1] - dq[1] * q[0];
  q1[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
  actSO3(dq, t, t1);
  t1[0] += dt[0]; t1[1] += dt[1]; t1[2] += dt[2];
}

/* exported so tests can check group identities on the helpers themselves */
void oracle_relSE3(const float* pi, const float* pj, float* pij) { relSE3(pi, pi + 3, pj, pj + 3, pij, pij + 3); }
void oracle_actSE3(const float* p, const float* X, float* Y) { actSE3(p, p + 3, X, Y); }
void oracle_adjSE3(const float* p, const float* X, float* Y) { adjSE3(p, p + 3, X, Y); }
void oracle_retrSE3(const float* xi, const float* p, float* p1, int xi45_zero) { retrSE3(xi, p, p + 3, p1, p1 + 3, xi45_zero); }

/* ---------------- per-pixel kernels ---------------- */

/* droid_kernels.cu:497-636 */
void oracle_frame_distance(const float* poses, const float* disps, const float* intr,
                           const int64_t* ii, const int64_t* jj, float* dist,
                           int M, int ht, int wd, float beta) {
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  for (int m = 0; m < M; m++) {
    const int ix = (int)ii[m], jx = (int)jj[m];
    float tij[3], qij[4];
    double accum = 0, valid = 0, total = 0;
    relSE3(poses + 7 * ix, poses + 7 * ix + 3, poses + 7 * jx, poses + 7 * jx + 3, tij, qij);
    for (int k = 0; k < ht * wd; k++) {
      const int i = k / wd, j = k % wd;
      const float u = (float)j, v = (float)i;
      float Xi[4], Xj[4], du, dv, d;
      Xi[0] = (u - cx) / fx; Xi[1] = (v - cy) / fy; Xi[2] = 1; Xi[3] = disps[(long long)ix * ht * wd + k];
      actSE3(tij, qij, Xi, Xj);
      du = fx * (Xj[0] / Xj[2]) + cx - u;
      dv = fy * (Xj[1] / Xj[2]) + cy - v;
      d = sqrtf(du * du + dv * dv);
      total += beta;
      if (Xj[2] > MIN_DEPTH) { accum += beta * d; valid += beta; }
      Xj[0] = Xi[0] + Xi[3] * tij[0];
      Xj[1] = Xi[1] + Xi[3] * tij[1];
      Xj[2] = Xi[2] + Xi[3] * tij[2];
      du = fx * (Xj[0] / Xj[2]) + cx - u;
      dv = fy * (Xj[1] / Xj[2]) + cy - v;
      d = sqrtf(du * du + dv * dv);
      total += (1 - beta);
      if (Xj[2] > MIN_DEPTH) { accum += (1 - beta) * d; valid += (1 - beta); }
    }
    dist[m] = (valid / (total + 1e-8) < 0.75) ? 1000.0f : (float)(accum / valid);
  }
}

/* droid_kernels.cu:406-495; coords [E,ht,wd,3] (third channel stays 0), valid [E,ht,wd,1] */
void oracle_projmap(const float* poses, const float* disps, const float* intr,
                    const int64_t* ii, const int64_t* jj, float* coords, float* valid,
                    int E, int ht, int wd) {
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const int HW = ht * wd;
  for (int e = 0; e < E; e++) {
    const int ix = (int)ii[e], jx = (int)jj[e];
    float tij[3], qij[4];
    relSE3(poses + 7 * ix, poses + 7 * ix + 3, poses + 7 * jx, poses + 7 * jx + 3, tij, qij);
    for (int k = 0; k < HW; k++) {
      const float u = (float)(k % wd), v = (float)(k / wd);
      float Xi[4] = {(u - cx) / fx, (v - cy) / fy, 1, disps[(long long)ix * HW + k]}, Xj[4];
      float* c = coords + ((long long)e * HW + k) * 3;
      actSE3(tij, qij, Xi, Xj);
      c[0] = u; c[1] = v; c[2] = 0;
      if (Xj[2] > 0.01) {
        c[0] = fx * (Xj[0] / Xj[2]) + cx;
        c[1] = fy * (Xj[1] / Xj[2]) + cy;
      }
      valid[(long long)e * HW + k] = (Xj[2] > MIN_DEPTH) ? 1.0f : 0.0f;
    }
  }
}

/* droid_kernels.cu:758-829 */
void oracle_iproj(const float* poses, const float* disps, const float* intr, float* points, int N, int ht, int wd) {
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const int HW = ht * wd;
  for (int n = 0; n < N; n++)
    for (int k = 0; k < HW; k++) {
      float Xi[4] = {((float)(k % wd) - cx) / fx, ((float)(k / wd) - cy) / fy, 1, disps[(long long)n * HW + k]}, Xj[4];
      float* p = points + ((long long)n * HW + k) * 3;
      actSE3(poses + 7 * n, poses + 7 * n + 3, Xi, Xj);
      p[0] = Xj[0] / Xj[3]; p[1] = Xj[1] / Xj[3]; p[2] = Xj[2] / Xj[3];
    }
}

/* droid_kernels.cu:640-754; counter [N,ht,wd] zeroed here (torch::zeros at :1478) */
void oracle_depth_filter(const float* poses, const float* disps, const float* intr,
                         const int64_t* inds, const float* thresh, float* counter,
                         int N, int nframes, int ht, int wd) {
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const int HW = ht * wd;
  memset(counter, 0, sizeof(float) * (size_t)N * HW);
  for (int b = 0; b < N; b++)
    for (int neigh = 0; neigh < 6; neigh++) {
      const int ix = (int)inds[b];
      const int jx = (neigh < 3) ? ix - neigh - 1 : ix + neigh;
      const float t = thresh[b];
      float tij[3], qij[4];
      if (jx < 0 || jx >= nframes) continue;
      relSE3(poses + 7 * ix, poses + 7 * ix + 3, poses + 7 * jx, poses + 7 * jx + 3, tij, qij);
      for (int k = 0; k < HW; k++) {
        const int i = k / wd, j = k % wd;
        float Xi[4] = {((float)j - cx) / fx, ((float)i - cy) / fy, 1, disps[(long long)ix * HW + k]}, Xj[4];
        float uj, vj, dj;
        int u0, v0;
        actSE3(tij, qij, Xi, Xj);
        uj = fx * (Xj[0] / Xj[2]) + cx;
        vj = fy * (Xj[1] / Xj[2]) + cy;
        dj = Xj[3] / Xj[2];
        if (!(uj == uj) || !(vj == vj)) continue;  /* static_cast<int>(NaN) is 0 on the device; NaN never passes :736 usefully */
        u0 = (int)floorf(fminf(fmaxf(uj, -1e9f), 1e9f));
        v0 = (int)floorf(fminf(fmaxf(vj, -1e9f), 1e9f));
        if (u0 >= 0 && v0 >= 0 && u0 < wd - 1 && v0 < ht - 1) {
          const float* dm = disps + (long long)jx * HW;
          const float d00 = dm[v0 * wd + u0], d01 = dm[v0 * wd + u0 + 1];
          const float d10 = dm[(v0 + 1) * wd + u0], d11 = dm[(v0 + 1) * wd + u0 + 1];
          if (fabs(1.0 / dj - 1.0 / d00) < t) counter[(long long)b * HW + k] += 1.0f;
          else if (fabs(1.0 / dj - 1.0 / d01) < t) counter[(long long)b * HW + k] += 1.0f;
          else if (fabs(1.0 / dj - 1.0 / d10) < t) counter[(long long)b * HW + k] += 1.0f;
          else if (fabs(1.0 / dj - 1.0 / d11) < t) counter[(long long)b * HW + k] += 1.0f;
        }
      }
    }
}

/* geom/projective_ops.py:102-130, jacobian=False; intrinsics [nframes,4] */
void oracle_reproject(const float* poses, const float* disps, const float* intr,
                      const int64_t* ii, const int64_t* jj, float* coords, float* valid,
                      int E, int ht, int wd) {
  const int HW = ht * wd;
  for (int e = 0; e < E; e++) {
    const int ix = (int)ii[e], jx = (int)jj[e];
    const float* Ki = intr + 4 * ix;
    const float* Kj = intr + 4 * jx;
    float tij[3], qij[4];
    relSE3(poses + 7 * ix, poses + 7 * ix + 3, poses + 7 * jx, poses + 7 * jx + 3, tij, qij);
    for (int k = 0; k < HW; k++) {
      float X0[4] = {((float)(k % wd) - Ki[2]) / Ki[0], ((float)(k / wd) - Ki[3]) / Ki[1], 1, disps[(long long)ix * HW + k]}, X1[4];
      float Z, d;
      actSE3(tij, qij, X0, X1);
      Z = X1[2];
      if (Z < 0.5f * 0.2f) Z = 1.0f;
      d = 1.0f / Z;
      coords[((long long)e * HW + k) * 2 + 0] = Kj[0] * (X1[0] * d) + Kj[2];
      coords[((long long)e * HW + k) * 2 + 1] = Kj[1] * (X1[1] * d) + Kj[3];
      valid[(long long)e * HW + k] = (X1[2] > 0.2f && X0[2] > 0.2f) ? 1.0f : 0.0f;
    }
  }
}

/* ---------------- bundle adjustment ---------------- */

/* dense LLT in fp64, in place on the lower triangle; returns 0 on success
 * (SimplicialLLT reports NumericalIssue on a non-positive pivot, droid_kernels.cu:1181) */
static int chol_solve(double* A, double* b, int n) {
  for (int j = 0; j < n; j++) {
    double d = A[j * n + j];
    for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0.0)) return 1;
    d = sqrt(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= A[i * n + k] * b[k];
    b[i] = s / A[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = b[i];
    for (int k = i + 1; k < n; k++) s -= A[k * n + i] * b[k];
    b[i] = s / A[i * n + i];
  }
  return 0;
}

static int cmp_i64(const void* a, const void* b) {
  int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return (x > y) - (x < y);
}

/* One edge of projective_transform_kernel (droid_kernels.cu:238-357): per-pixel fp32
 * Jacobians; Hessian/gradient sums in fp64.  h[78] upper-triangle order of :309-315,
 * Eii/Eij [6][HW], Cii/bz [HW]. */
static void assemble_edge(const float* target, const float* weight, const float* poses, const float* disps,
                          const float* intr, int ix, int jx, int ht, int wd,
                          double* h, double* vi, double* vj, float* Eii, float* Eij, float* Cii, float* bz) {
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const int HW = ht * wd;
  float tij[3], qij[4];
  relSE3(poses + 7 * ix, poses + 7 * ix + 3, poses + 7 * jx, poses + 7 * jx + 3, tij, qij);
  for (int l = 0; l < 78; l++) h[l] = 0;
  for (int n = 0; n < 6; n++) { vi[n] = 0; vj[n] = 0; }
  for (int k = 0; k < HW; k++) {
    const int i = k / wd, j = k % wd;
    const float u = (float)j, v = (float)i;
    float Xi[4], Xj[4], Jx[12], Jz;
    float* Ji = &Jx[0];
    float* Jj = &Jx[6];
    int l;
    Xi[0] = (u - cx) / fx; Xi[1] = (v - cy) / fy; Xi[2] = 1; Xi[3] = disps[(long long)ix * HW + k];
    actSE3(tij, qij, Xi, Xj);
    {
      const float x = Xj[0], y = Xj[1], hh = Xj[3];
      /* :287,290-291 are written with double literals (`1.0 / Xj[2]`, `.001 * weight`): the quotient and the products
       * are formed in double and rounded to float once.  For the quotient that equals the float division (double
       * rounding of a correctly rounded quotient is innocuous at 53 >= 2*24+2 bits); for the weights it does NOT
       * (.001 and .001f are different numbers), so the text is followed literally. */
      const float d = (Xj[2] < MIN_DEPTH) ? 0.0f : (float)(1.0 / (double)Xj[2]);
      const float d2 = d * d;
      const float wu = (Xj[2] < MIN_DEPTH) ? 0.0f : (float)(.001 * (double)weight[0 * HW + k]);
      const float wv = (Xj[2] < MIN_DEPTH) ? 0.0f : (float)(.001 * (double)weight[1 * HW + k]);
      const float ru = target[0 * HW + k] - (fx * d * x + cx);
      const float rv = target[1 * HW + k] - (fy * d * y + cy);

      Jj[0] = fx * (hh * d); Jj[1] = fx * 0; Jj[2] = fx * (-x * hh * d2);
      Jj[3] = fx * (-x * y * d2); Jj[4] = fx * (1 + x * x * d2); Jj[5] = fx * (-y * d);
      Jz = fx * (tij[0] * d - tij[2] * (x * d2));
      adjSE3(tij, qij, Jj, Ji);
      for (int n = 0; n < 6; n++) Ji[n] *= -1;
      l = 0;
      for (int n = 0; n < 12; n++)
        for (int m = 0; m <= n; m++) { h[l] += (double)(wu * Jx[n] * Jx[m]); l++; }
      for (int n = 0; n < 6; n++) {
        vi[n] += (double)(wu * ru * Ji[n]);
        vj[n] += (double)(wu * ru * Jj[n]);
        Eii[n * HW + k] = wu * Jz * Ji[n];
        Eij[n * HW + k] = wu * Jz * Jj[n];
      }
      Cii[k] = wu * Jz * Jz;
      bz[k] = wu * ru * Jz;

      Jj[0] = fy * 0; Jj[1] = fy * (hh * d); Jj[2] = fy * (-y * hh * d2);
      Jj[3] = fy * (-1 - y * y * d2); Jj[4] = fy * (x * y * d2); Jj[5] = fy * (x * d);
      Jz = fy * (tij[1] * d - tij[2] * (y * d2));
      adjSE3(tij, qij, Jj, Ji);
      for (int n = 0; n < 6; n++) Ji[n] *= -1;
      l = 0;
      for (int n = 0; n < 12; n++)
        for (int m = 0; m <= n; m++) { h[l] += (double)(wv * Jx[n] * Jx[m]); l++; }
      for (int n = 0; n < 6; n++) {
        vi[n] += (double)(wv * rv * Ji[n]);
        vj[n] += (double)(wv * rv * Jj[n]);
        Eii[n * HW + k] += wv * Jz * Ji[n];
        Eij[n * HW + k] += wv * Jz * Jj[n];
      }
      Cii[k] += wv * Jz * Jz;
      bz[k] += wv * rv * Jz;
    }
  }
}

/* The first launch of ba_cuda on its own (droid_kernels.cu:1346-1349): Hs [4,E,6,6] (ii, ij, ji, jj blocks as :386-398
 * scatter them), vs [2,E,6], Eii/Eij [E,6,HW], Cii/bz [E,HW].  Per-pixel outputs are fp32 exactly as the kernel forms
 * them; the 90 sums are fp64 here (the kernel: fp32 thread partials + a 256-leaf tree).  Pinned by
 * tests/golden/ba_assemble_kernel.npz (the kernel's own text run on the host). */
void oracle_ba_assemble(const float* poses, const float* disps, const float* intr, const float* targets,
                        const float* weights, const int64_t* ii, const int64_t* jj, int E, int ht, int wd,
                        double* Hs, double* vs, float* Eii, float* Eij, float* Cii, float* bz) {
  const int HW = ht * wd;
  for (int e = 0; e < E; e++) {
    double h[78], vi[6], vj[6];
    int l = 0;
    assemble_edge(targets + (long long)e * 2 * HW, weights + (long long)e * 2 * HW, poses, disps, intr,
                  (int)ii[e], (int)jj[e], ht, wd, h, vi, vj, Eii + (long long)e * 6 * HW, Eij + (long long)e * 6 * HW,
                  Cii + (long long)e * HW, bz + (long long)e * HW);
    for (int n = 0; n < 6; n++) { vs[(0 * E + e) * 6 + n] = vi[n]; vs[(1 * E + e) * 6 + n] = vj[n]; }
    for (int n = 0; n < 12; n++)
      for (int m = 0; m <= n; m++, l++) {
        if (n < 6 && m < 6) { Hs[((0LL * E + e) * 6 + n) * 6 + m] = h[l]; Hs[((0LL * E + e) * 6 + m) * 6 + n] = h[l]; }
        else if (n >= 6 && m < 6) { Hs[((1LL * E + e) * 6 + m) * 6 + (n - 6)] = h[l]; Hs[((2LL * E + e) * 6 + (n - 6)) * 6 + m] = h[l]; }
        else { Hs[((3LL * E + e) * 6 + (n - 6)) * 6 + (m - 6)] = h[l]; Hs[((3LL * E + e) * 6 + (m - 6)) * 6 + (n - 6)] = h[l]; }
      }
  }
}

/* ba_cuda (droid_kernels.cu:1293-1410).
 * poses [nframes,7], disps [nframes,ht,wd] updated in place; targets/weights [E,2,ht,wd];
 * eta [K_eta,ht,wd] (K_eta == K or 1); dx_out [P,6]; dz_out [K,HW] or NULL;
 * sys_out (optional, may be NULL): the LAST iteration's undamped reduced system,
 * (6P)^2 row-major A-S followed by 6P rhs, fp64 — what an edge-sharded run all-reduces.
 * evt_skip_first != 0 reproduces EvT6x1_kernel's `idx <= 0` early return (:1084), which drops
 * window pose 0 from the depth back-substitution; 0 gives the exact Schur back-substitution
 * (what geom/ba.py computes) and is used only to pin this restatement against that file.
 * dx_in (optional): skip the solve and apply this pose update instead (used by the 2-rank gloo test, where
 * the reduced system is all-reduced and solved outside).
 * returns K (number of depth maps optimised) or a negative error. status_out[0]=1 when a
 * factorisation failed. */
int oracle_ba(float* poses, float* disps, const float* intr, const float* targets, const float* weights,
              const float* eta, const int64_t* ii, const int64_t* jj,
              int E, int nframes, int ht, int wd, int K_eta, int t0, int t1, int iterations,
              float lm, float ep, int motion_only, float* dx_out, float* dz_out, double* sys_out,
              int* status_out, int xi45_zero, int evt_skip_first, const float* dx_in) {
  const int HW = ht * wd, P = t1 - t0, n6 = 6 * P;
  int K = 0, fail_any = 0;
  int64_t* kx = (int64_t*)malloc(sizeof(int64_t) * (size_t)(P + E + 1));
  int* kidx = (int*)malloc(sizeof(int) * (size_t)(nframes + 1));
  /* kx = unique(cat(ts, ii)), kk = inverse (droid_kernels.cu:1314-1322) */
  {
    int cnt = 0;
    for (int p = 0; p < P; p++) kx[cnt++] = t0 + p;
    for (int e = 0; e < E; e++) kx[cnt++] = ii[e];
    qsort(kx, cnt, sizeof(int64_t), cmp_i64);
    for (int c = 0; c < cnt; c++) if (c == 0 || kx[c] != kx[c - 1]) kx[K++] = kx[c];
    for (int f = 0; f <= nframes; f++) kidx[f] = -1;
    for (int k = 0; k < K; k++) if (kx[k] >= 0 && kx[k] < nframes) kidx[kx[k]] = k;
  }
  if (!motion_only && !(K_eta == K || K_eta == 1)) { free(kx); free(kidx); return -2; }

  double* Hs = (double*)malloc(sizeof(double) * (size_t)E * 78);
  double* vis = (double*)malloc(sizeof(double) * (size_t)E * 12);
  float* Eii = (float*)malloc(sizeof(float) * (size_t)E * 6 * HW);
  float* Eij = (float*)malloc(sizeof(float) * (size_t)E * 6 * HW);
  float* Cii = (float*)malloc(sizeof(float) * (size_t)E * HW);
  float* wi = (float*)malloc(sizeof(float) * (size_t)E * HW);
  float* C = (float*)malloc(sizeof(float) * (size_t)K * HW);
  float* w = (float*)malloc(sizeof(float) * (size_t)K * HW);
  float* Q = (float*)malloc(sizeof(float) * (size_t)K * HW);
  float* Ei = (float*)malloc(sizeof(float) * (size_t)(P > 0 ? P : 1) * 6 * HW);
  float* dw = (float*)malloc(sizeof(float) * (size_t)(P + E) * HW);
  double* A = (double*)malloc(sizeof(double) * (size_t)(n6 > 0 ? n6 * n6 : 1));
  double* b = (double*)malloc(sizeof(double) * (size_t)(n6 > 0 ? n6 : 1));
  float* dx = (float*)calloc((size_t)(n6 > 0 ? n6 : 1), sizeof(float));
  float* dz = (float*)calloc((size_t)K * HW + 1, sizeof(float));

  for (int itr = 0; itr < iterations; itr++) {
    for (int e = 0; e < E; e++)
      assemble_edge(targets + (long long)e * 2 * HW, weights + (long long)e * 2 * HW, poses, disps, intr,
                    (int)ii[e], (int)jj[e], ht, wd, Hs + e * 78, vis + e * 12, vis + e * 12 + 6,
                    Eii + (long long)e * 6 * HW, Eij + (long long)e * 6 * HW, Cii + (long long)e * HW, wi + (long long)e * HW);

    /* pose x pose block: SparseBlock A(P,6), update_lhs/update_rhs (droid_kernels.cu:1354-1361).
     * Hs layout of :385-397: blocks (ii,ii) (ii,jj) (jj,ii) (jj,jj); the reference stores fp32 sums. */
    for (int k = 0; k < n6 * n6; k++) A[k] = 0;
    for (int k = 0; k < n6; k++) b[k] = 0;
    for (int e = 0; e < E; e++) {
      const int pi = (int)ii[e] - t0, pj = (int)jj[e] - t0;
      int l = 0;
      for (int n = 0; n < 12; n++)
        for (int m = 0; m <= n; m++, l++) {
          const double val = (double)(float)Hs[e * 78 + l];
          if (n < 6 && m < 6) {                       /* Hs[0]: (ii,ii), symmetric */
            if (pi >= 0 && pi < P) { A[(6 * pi + n) * n6 + 6 * pi + m] += val; if (n != m) A[(6 * pi + m) * n6 + 6 * pi + n] += val; }
          } else if (n >= 6 && m < 6) {               /* Hs[1][m][n-6]: (ii,jj) ; Hs[2][n-6][m]: (jj,ii) */
            if (pi >= 0 && pj >= 0 && pi < P && pj < P) {
              A[(6 * pi + m) * n6 + 6 * pj + (n - 6)] += val;
              A[(6 * pj + (n - 6)) * n6 + 6 * pi + m] += val;
            }
          } else {                                    /* Hs[3]: (jj,jj) */
            if (pj >= 0 && pj < P) { A[(6 * pj + n - 6) * n6 + 6 * pj + m - 6] += val; if (n != m) A[(6 * pj + m - 6) * n6 + 6 * pj + n - 6] += val; }
          }
        }
      for (int n = 0; n < 6; n++) {
        if (pi >= 0 && pi < P) b[6 * pi + n] += (double)(float)vis[e * 12 + n];
        if (pj >= 0 && pj < P) b[6 * pj + n] += (double)(float)vis[e * 12 + 6 + n];
      }
    }

    if (!motion_only) {
      /* C, w = accum(Cii, wi by ii over kx); Q = 1/(C+eta)  (:1374-1376) */
      for (long long k = 0; k < (long long)K * HW; k++) { C[k] = 0; w[k] = 0; }
      for (int e = 0; e < E; e++) {
        const int k = kidx[ii[e]];
        for (int x = 0; x < HW; x++) { C[(long long)k * HW + x] += Cii[(long long)e * HW + x]; w[(long long)k * HW + x] += wi[(long long)e * HW + x]; }
      }
      for (int k = 0; k < K; k++)
        for (int x = 0; x < HW; x++)
          Q[(long long)k * HW + x] = 1.0f / (C[(long long)k * HW + x] + eta[(long long)(K_eta == 1 ? 0 : k) * HW + x]);
      /* Ei = accum(Eii by ii over ts)  (:1378) */
      for (long long k = 0; k < (long long)P * 6 * HW; k++) Ei[k] = 0;
      for (int e = 0; e < E; e++) {
        const int p = (int)ii[e] - t0;
        if (p >= 0 && p < P)
          for (int x = 0; x < 6 * HW; x++) Ei[(long long)p * 6 * HW + x] += Eii[(long long)e * 6 * HW + x];
      }
      /* schur_block (:1201-1290) over E_rows = [Ei (P rows); Eij (E rows)],
       * ii_exp=[ts;ii], jj_exp=[ts;jj], kk_exp = index of ii_exp in kx.
       * S[(a,b)] += (E_n1 * Q_k) E_n2^T for rows n1 (pose a), n2 (pose b) sharing depth frame k;
       * v[pose(n)] += E_n (Q_k * w_k);  then A <- A - S, b <- b - v (:1382). */
      const int R = P + E;
      for (int n1 = 0; n1 < R; n1++) {
        const int a = (n1 < P) ? n1 : (int)jj[n1 - P] - t0;
        const int k1 = (n1 < P) ? kidx[t0 + n1] : kidx[ii[n1 - P]];
        const float* E1 = (n1 < P) ? Ei + (long long)n1 * 6 * HW : Eij + (long long)(n1 - P) * 6 * HW;
        if (a < 0 || a >= P) continue;      /* :1227 (j >= t0 && j <= t1; j == t1 would index out of range) */
        for (int n2 = 0; n2 < R; n2++) {
          const int bpose = (n2 < P) ? n2 : (int)jj[n2 - P] - t0;
          const int k2 = (n2 < P) ? kidx[t0 + n2] : kidx[ii[n2 - P]];
          const float* E2 = (n2 < P) ? Ei + (long long)n2 * 6 * HW : Eij + (long long)(n2 - P) * 6 * HW;
          double dS[6][6];
          if (bpose < 0 || bpose >= P || k1 != k2) continue;
          memset(dS, 0, sizeof(dS));
          for (int x = 0; x < HW; x++) {
            const float q = Q[(long long)k1 * HW + x];
            for (int n = 0; n < 6; n++) {
              const float ei = E1[n * HW + x] * q;
              for (int m = 0; m < 6; m++) dS[n][m] += (double)(ei * E2[m * HW + x]);
            }
          }
          for (int n = 0; n < 6; n++)
            for (int m = 0; m < 6; m++) A[(6 * a + n) * n6 + 6 * bpose + m] -= (double)(float)dS[n][m];
        }
        {
          double bb[6] = {0, 0, 0, 0, 0, 0};
          for (int x = 0; x < HW; x++) {
            const float qw = Q[(long long)k1 * HW + x] * w[(long long)k1 * HW + x];
            for (int n = 0; n < 6; n++) bb[n] += (double)(qw * E1[n * HW + x]);
          }
          for (int n = 0; n < 6; n++) b[6 * a + n] -= (double)(float)bb[n];
        }
      }
    }

    if (sys_out) {
      memcpy(sys_out, A, sizeof(double) * (size_t)n6 * n6);
      memcpy(sys_out + (size_t)n6 * n6, b, sizeof(double) * (size_t)n6);
    }

    /* solve (:1171-1192): diag += ep + lm*diag; LLT; zeros on failure */
    for (int k = 0; k < n6; k++) A[k * n6 + k] += (double)ep + (double)lm * A[k * n6 + k];
    if (dx_in) {                      /* edge-sharded test harness: the globally solved dx is handed in */
      for (int k = 0; k < n6; k++) dx[k] = dx_in[k];
    } else if (n6 > 0 && chol_solve(A, b, n6) == 0) {
      for (int k = 0; k < n6; k++) dx[k] = (float)b[k];
    } else {
      for (int k = 0; k < n6; k++) dx[k] = 0.0f;
      if (n6 > 0) fail_any = 1;
    }

    if (!motion_only) {
      /* dw = EvT (:1074-1094): rows whose pose index is <= 0 or >= P are skipped (left 0) */
      const int R = P + E;
      for (long long k = 0; k < (long long)R * HW; k++) dw[k] = 0;
      for (int n = 0; n < R; n++) {
        const int a = (n < P) ? n : (int)jj[n - P] - t0;
        const float* En = (n < P) ? Ei + (long long)n * 6 * HW : Eij + (long long)(n - P) * 6 * HW;
        if (a >= P || (evt_skip_first ? a <= 0 : a < 0)) continue;
        for (int x = 0; x < HW; x++) {
          float s = 0;
          for (int m = 0; m < 6; m++) s += En[m * HW + x] * dx[6 * a + m];
          dw[(long long)n * HW + x] = s;
        }
      }
      /* dz = Q * (w - accum(dw by ii_exp over kx))  (:1393) */
      for (long long k = 0; k < (long long)K * HW; k++) dz[k] = 0;
      for (int n = 0; n < R; n++) {
        const int k = (n < P) ? kidx[t0 + n] : kidx[ii[n - P]];
        for (int x = 0; x < HW; x++) dz[(long long)k * HW + x] += dw[(long long)n * HW + x];
      }
      for (long long k = 0; k < (long long)K * HW; k++) dz[k] = Q[k] * (w[k] - dz[k]);
    }

    /* pose_retr_kernel (:877-910) */
    for (int p = 0; p < P; p++) {
      float t1v[3], q1v[4];
      float* ps = poses + 7 * (t0 + p);
      retrSE3(dx + 6 * p, ps, ps + 3, t1v, q1v, xi45_zero);
      ps[0] = t1v[0]; ps[1] = t1v[1]; ps[2] = t1v[2];
      ps[3] = q1v[0]; ps[4] = q1v[1]; ps[5] = q1v[2]; ps[6] = q1v[3];
    }
    /* disp_retr_kernel (:912-925) */
    if (!motion_only)
      for (int k = 0; k < K; k++)
        for (int x = 0; x < HW; x++) disps[(long long)kx[k] * HW + x] += dz[(long long)k * HW + x];
  }

  if (dx_out) memcpy(dx_out, dx, sizeof(float) * (size_t)n6);
  if (dz_out && !motion_only) memcpy(dz_out, dz, sizeof(float) * (size_t)K * HW);
  if (status_out) status_out[0] = fail_any;
  free(kx); free(kidx); free(Hs); free(vis); free(Eii); free(Eij); free(Cii); free(wi);
  free(C); free(w); free(Q); free(Ei); free(dw); free(A); free(b); free(dx); free(dz);
  return K;
}
