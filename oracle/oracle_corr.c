/*
 * oracle_corr.c — CPU restatement of the reference correlation path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pvo_amd/ may import, link or call this
 * file; it exists so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg can check / time the HIP path against the reference algorithm.
 *
 * Restates, loop for loop:
 *   corr_index_forward_kernel   reference VO_Module/src/correlation_kernels.cu:19-70
 *   corr_index_backward_kernel  reference VO_Module/src/correlation_kernels.cu:73-124
 *   CorrBlock.corr + pyramid    reference VO_Module/droid_slam/modules/corr.py:24-38,63-71
 *   CorrBlock.__call__          reference VO_Module/droid_slam/modules/corr.py:40-50
 *   altcorr_forward_kernel      reference VO_Module/src/altcorr_kernel.cu:27-149
 *   altcorr_backward_kernel     reference VO_Module/src/altcorr_kernel.cu:152-286
 *
 * PARITY PIN STATUS: the reference CUDA kernels cannot be built in this image
 * (no nvcc; torch's hipify output does not compile against torch 2.10 because of
 * AT_DISPATCH(volume.type())), and the reference holds no golden
 * vectors for this path.  This restatement is pinned by (a) the reference's own
 * Python (modules/corr.py imported from /root/reference: volume + pyramid, fixtures
 * in tests/golden), (b) an independent bilinear-sampling formulation
 * (torch.nn.functional.grid_sample) for the lookup.  "parity unpinned" applies to
 * the per-op rounding order of the 16-bit lookup, which follows the kernel text.
 *
 * Arithmetic model (what the reference's CUDA build computes):
 *   half    c10::Half operators: float op, then round to half, per operator.
 *   float   `corr += s * w` is contracted to one FMA by nvcc (default -fmad=true):
 *           contract != 0 selects fmaf(), contract == 0 the two-rounding form.
 *   double  same with fma().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { O_F32 = 0, O_F16 = 1, O_BF16 = 2, O_F64 = 3 };

/* ---------- software 16-bit float conversions (round to nearest even) ---------- */
static float h2f(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu, u;
  if (e == 0) {
    if (m == 0) u = sign;
    else {
      int sh = 0;
      while (!(m & 0x400u)) { m <<= 1; sh++; }
      m &= 0x3ffu;
      u = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
    }
  } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
  else u = sign | ((e + 112u) << 23) | (m << 13);
  float f; memcpy(&f, &u, 4); return f;
}

static uint16_t f2h(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);          /* NaN */
  if (ax >= 0x47800000u) return (uint16_t)(sign | 0x7c00u);         /* overflow -> inf (>= 65536) */
  if (ax < 0x38800000u) {                                           /* subnormal half or zero */
    if (ax < 0x33000000u) return (uint16_t)sign;                    /* < 2^-25 -> 0 */
    uint32_t e = ax >> 23;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;
    int shift = 113 - (int)e + 13;                                  /* bits to drop */
    uint32_t r = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (r & 1u))) r++;
    return (uint16_t)(sign | r);
  }
  {
    uint32_t e = (ax >> 23) - 112u;
    uint32_t m = ax & 0x7fffffu;
    uint32_t r = (e << 10) | (m >> 13);
    uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;         /* may carry into exponent / inf: correct */
    return (uint16_t)(sign | r);
  }
}

static float b2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2b(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

uint16_t oracle_f32_to_f16(float f) { return f2h(f); }
float oracle_f16_to_f32(uint16_t h) { return h2f(h); }
uint16_t oracle_f32_to_bf16(float f) { return f2b(f); }

/* scalar value in "float/double holding a representable value" form */
static double ld(const void* p, long long i, int dt) {
  switch (dt) {
    case O_F32: return ((const float*)p)[i];
    case O_F16: return h2f(((const uint16_t*)p)[i]);
    case O_BF16: return b2f(((const uint16_t*)p)[i]);
    default: return ((const double*)p)[i];
  }
}
static void st(void* p, long long i, int dt, double v) {
  switch (dt) {
    case O_F32: ((float*)p)[i] = (float)v; break;
    case O_F16: ((uint16_t*)p)[i] = f2h((float)v); break;
    case O_BF16: ((uint16_t*)p)[i] = f2b((float)v); break;
    default: ((double*)p)[i] = v; break;
  }
}
/* scalar_t(w) for a float weight */
static double cast_w(float w, int dt) {
  switch (dt) {
    case O_F16: return h2f(f2h(w));
    case O_BF16: return b2f(f2b(w));
    default: return w;
  }
}
/* acc += s * w in scalar_t arithmetic */
static double acc_step(double acc, double s, double w, int dt, int contract) {
  switch (dt) {
    case O_F32: {
      float a = (float)acc, sf = (float)s, wf = (float)w;
      if (contract) return fmaf(sf, wf, a);
      { volatile float p = sf * wf; return (float)(a + p); }
    }
    case O_F16: { float p = h2f(f2h((float)s * (float)w)); return h2f(f2h((float)acc + p)); }
    case O_BF16: { float p = b2f(f2b((float)s * (float)w)); return b2f(f2b((float)acc + p)); }
    default:
      if (contract) return fma(s, w, acc);
      { volatile double p = s * w; return acc + p; }
  }
}

static int floor_to_int(float x) {
  float f = floorf(x);
  if (f != f) return 0;
  if (f < -1073741824.0f) f = -1073741824.0f;
  if (f > 1073741824.0f) f = 1073741824.0f;
  return (int)f;
}

static int within(int h, int w, int H, int W) { return h >= 0 && h < H && w >= 0 && w < W; }

static size_t esize(int dt) { return dt == O_F64 ? 8 : (dt == O_F32 ? 4 : 2); }

/* correlation_kernels.cu:19-70 — scatter form, exactly the reference's loop nest.
 * volume [N,h1,w1,h2,w2], coords [N,2,h1,w1], corr [N,rd,rd,h1,w1] (zeroed here, as
 * torch::zeros at :142-143). `cscale` multiplies the coordinates first (1/2^level). */
int oracle_corr_index_forward(const void* volume, const float* coords, void* corr,
                              int N, int h1, int w1, int h2, int w2, int r, int dt, int contract) {
  const int rd = 2 * r + 1;
  const long long HW = (long long)h1 * w1;
  memset(corr, 0, (size_t)N * rd * rd * HW * esize(dt));
  for (int n = 0; n < N; n++)
    for (int y = 0; y < h1; y++)
      for (int x = 0; x < w1; x++) {
        float x0 = coords[((long long)n * 2 + 0) * HW + (long long)y * w1 + x];
        float y0 = coords[((long long)n * 2 + 1) * HW + (long long)y * w1 + x];
        float dx = x0 - floorf(x0), dy = y0 - floorf(y0);
        for (int i = 0; i < rd + 1; i++)
          for (int j = 0; j < rd + 1; j++) {
            int x1 = floor_to_int(x0) - r + i;
            int y1 = floor_to_int(y0) - r + j;
            if (!within(y1, x1, h2, w2)) continue;
            double s = ld(volume, ((((long long)n * h1 + y) * w1 + x) * h2 + y1) * w2 + x1, dt);
#define CIDX(a, b) ((((long long)n * rd + (a)) * rd + (b)) * HW + (long long)y * w1 + x)
            if (i > 0 && j > 0) {
              long long k = CIDX(i - 1, j - 1);
              st(corr, k, dt, acc_step(ld(corr, k, dt), s, cast_w(dx * dy, dt), dt, contract));
            }
            if (i > 0 && j < rd) {
              long long k = CIDX(i - 1, j);
              st(corr, k, dt, acc_step(ld(corr, k, dt), s, cast_w(dx * (1.0f - dy), dt), dt, contract));
            }
            if (i < rd && j > 0) {
              long long k = CIDX(i, j - 1);
              st(corr, k, dt, acc_step(ld(corr, k, dt), s, cast_w((1.0f - dx) * dy, dt), dt, contract));
            }
            if (i < rd && j < rd) {
              long long k = CIDX(i, j);
              st(corr, k, dt, acc_step(ld(corr, k, dt), s, cast_w((1.0f - dx) * (1.0f - dy), dt), dt, contract));
            }
#undef CIDX
          }
      }
  return 0;
}

/* correlation_kernels.cu:73-124. corr_grad [N,rd,rd,h1,w1] -> volume_grad [N,h1,w1,h2,w2]. */
int oracle_corr_index_backward(const float* coords, const void* corr_grad, void* volume_grad,
                               int N, int h1, int w1, int h2, int w2, int r, int dt, int contract) {
  const int rd = 2 * r + 1;
  const long long HW = (long long)h1 * w1;
  memset(volume_grad, 0, (size_t)N * HW * h2 * w2 * esize(dt));
  for (int n = 0; n < N; n++)
    for (int y = 0; y < h1; y++)
      for (int x = 0; x < w1; x++) {
        float x0 = coords[((long long)n * 2 + 0) * HW + (long long)y * w1 + x];
        float y0 = coords[((long long)n * 2 + 1) * HW + (long long)y * w1 + x];
        float dx = x0 - floorf(x0), dy = y0 - floorf(y0);
        for (int i = 0; i < rd + 1; i++)
          for (int j = 0; j < rd + 1; j++) {
            int x1 = floor_to_int(x0) - r + i;
            int y1 = floor_to_int(y0) - r + j;
            if (!within(y1, x1, h2, w2)) continue;
            double g = 0.0;
#define GIDX(a, b) ((((long long)n * rd + (a)) * rd + (b)) * HW + (long long)y * w1 + x)
            if (i > 0 && j > 0) g = acc_step(g, ld(corr_grad, GIDX(i - 1, j - 1), dt), cast_w(dx * dy, dt), dt, contract);
            if (i > 0 && j < rd) g = acc_step(g, ld(corr_grad, GIDX(i - 1, j), dt), cast_w(dx * (1.0f - dy), dt), dt, contract);
            if (i < rd && j > 0) g = acc_step(g, ld(corr_grad, GIDX(i, j - 1), dt), cast_w((1.0f - dx) * dy, dt), dt, contract);
            if (i < rd && j < rd) g = acc_step(g, ld(corr_grad, GIDX(i, j), dt), cast_w((1.0f - dx) * (1.0f - dy), dt), dt, contract);
#undef GIDX
            {
              long long k = ((((long long)n * h1 + y) * w1 + x) * h2 + y1) * w2 + x1;
              /* volume_grad += g  (scalar_t add onto the zero-initialised tensor) */
              st(volume_grad, k, dt, acc_step(ld(volume_grad, k, dt), g, 1.0, dt, 0));
            }
          }
      }
  return 0;
}

/* CorrBlock.__call__ (corr.py:40-50): levels sampled at coords/2^l, concatenated on
 * the channel axis.  coords_nhw2 [N,h1,w1,2] (the layout FactorGraph hands over);
 * volumes[l] is level l, [N,h1,w1,h2>>l,w2>>l]; out [N,L*rd*rd,h1,w1]. */
int oracle_corr_pyramid_lookup(const void* const* volumes, const float* coords_nhw2, void* out,
                               int N, int h1, int w1, int h2, int w2, int nlev, int r, int dt, int contract) {
  const int rd = 2 * r + 1;
  const long long HW = (long long)h1 * w1;
  float* c = (float*)malloc(sizeof(float) * (size_t)N * 2 * HW);
  void* tmp = malloc(esize(dt) * (size_t)N * rd * rd * HW);
  if (!c || !tmp) { free(c); free(tmp); return 1; }
  for (int l = 0; l < nlev; l++) {
    const float div = (float)(1 << l);
    for (int n = 0; n < N; n++)
      for (long long p = 0; p < HW; p++) {
        c[((long long)n * 2 + 0) * HW + p] = coords_nhw2[((long long)n * HW + p) * 2 + 0] / div;
        c[((long long)n * 2 + 1) * HW + p] = coords_nhw2[((long long)n * HW + p) * 2 + 1] / div;
      }
    oracle_corr_index_forward(volumes[l], c, tmp, N, h1, w1, h2 >> l, w2 >> l, r, dt, contract);
    for (int n = 0; n < N; n++)
      memcpy((char*)out + esize(dt) * (((size_t)n * nlev + l) * rd * rd * HW),
             (char*)tmp + esize(dt) * ((size_t)n * rd * rd * HW), esize(dt) * (size_t)rd * rd * HW);
  }
  free(c); free(tmp);
  return 0;
}

/* CorrBlock.corr + pyramid (corr.py:24-38, 63-71).
 * fmap1,fmap2 [N,C,H,W] in dt; levels[l] [N,H,W,H>>l,W>>l] in dt.
 *   level0 = round_dt( sum_c (f1/4)*(f2/4) )  with the sum carried in fp32 over c
 *            ascending (torch.matmul's accumulation order is unspecified: compare with
 *            a tolerance, see tests), products of the dt-rounded quotients;
 *   level l+1 = round_dt( (a+b+c+d) * 0.25 ) of the ROUNDED level l, floor sizes
 *            (F.avg_pool2d(2, stride=2) accumulates a 16-bit input in fp32). */
int oracle_corr_build(const void* fmap1, const void* fmap2, void* const* levels,
                      int N, int C, int H, int W, int nlev, int dt) {
  const long long HW = (long long)H * W;
  float* a = (float*)malloc(sizeof(float) * (size_t)C * HW);
  float* b = (float*)malloc(sizeof(float) * (size_t)C * HW);
  if (!a || !b) { free(a); free(b); return 1; }
  for (int n = 0; n < N; n++) {
    for (long long k = 0; k < C * HW; k++) {
      /* fmap / 4.0 in dt (exact for normal values) */
      a[k] = (float)cast_w((float)ld(fmap1, (long long)n * C * HW + k, dt) / 4.0f, dt);
      b[k] = (float)cast_w((float)ld(fmap2, (long long)n * C * HW + k, dt) / 4.0f, dt);
    }
    for (long long p1 = 0; p1 < HW; p1++)
      for (long long p2 = 0; p2 < HW; p2++) {
        float acc = 0.f;
        for (int c = 0; c < C; c++) acc = fmaf(a[c * HW + p1], b[c * HW + p2], acc);
        st(levels[0], ((long long)n * HW + p1) * HW + p2, dt, acc);
      }
  }
  free(a); free(b);
  int h = H, w = W;
  for (int l = 1; l < nlev; l++) {
    const int h2 = h / 2, w2 = w / 2;
    for (long long pl = 0; pl < (long long)N * HW; pl++)
      for (int y = 0; y < h2; y++)
        for (int x = 0; x < w2; x++) {
          const long long s = pl * h * w;
          float v00 = (float)ld(levels[l - 1], s + (long long)(2 * y) * w + 2 * x, dt);
          float v01 = (float)ld(levels[l - 1], s + (long long)(2 * y) * w + 2 * x + 1, dt);
          float v10 = (float)ld(levels[l - 1], s + (long long)(2 * y + 1) * w + 2 * x, dt);
          float v11 = (float)ld(levels[l - 1], s + (long long)(2 * y + 1) * w + 2 * x + 1, dt);
          /* ATen avg_pool2d: sum in row-major window order, then divide by the pool size */
          float sum = ((v00 + v01) + v10) + v11;
          st(levels[l], pl * h2 * w2 + (long long)y * w2 + x, dt, sum / 4.0f);
        }
    h = h2; w = w2;
  }
  return 0;
}

/* altcorr_forward_kernel (altcorr_kernel.cu:27-149), fp32.  fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C],
 * coords [B,S,H1,W1,2], corr [B,S,rd*rd,H1,W1] (zeroed here, torch::zeros at :302).
 * Keeps the reference's accumulation structure: 32-channel slabs outermost, each slab's partial dot
 * product scattered with the four bilinear weights (channel = iy + rd*ix). */
int oracle_altcorr_forward(const float* f1, const float* f2, const float* coords, float* corr,
                           int B, int S, int H1, int W1, int H2, int W2, int C, int r) {
  const int rd = 2 * r + 1;
  const long long HW = (long long)H1 * W1;
  memset(corr, 0, sizeof(float) * (size_t)B * S * rd * rd * HW);
  for (int b = 0; b < B; b++)
    for (int c0 = 0; c0 < C; c0 += 32)
      for (int h1 = 0; h1 < H1; h1++)
        for (int w1 = 0; w1 < W1; w1++)
          for (int n = 0; n < S; n++) {
            const float* cp = coords + ((((long long)b * S + n) * H1 + h1) * W1 + w1) * 2;
            const float x = cp[0], y = cp[1];
            const float dx = x - floorf(x), dy = y - floorf(y);
            float* out = corr + (((long long)b * S + n) * rd * rd) * HW + (long long)h1 * W1 + w1;
            for (int iy = 0; iy < rd + 1; iy++)
              for (int ix = 0; ix < rd + 1; ix++) {
                const int h2 = floor_to_int(y) - r + iy, w2 = floor_to_int(x) - r + ix;
                float s = 0.0f;
                if (within(h2, w2, H2, W2)) {
                  const float* a = f1 + (((long long)b * H1 + h1) * W1 + w1) * C;
                  const float* q = f2 + (((long long)b * H2 + h2) * W2 + w2) * C;
                  for (int k = c0; k < c0 + 32 && k < C; k++) s = fmaf(a[k], q[k], s);
                }
                /* `nw = s * w; corr += nw` (:113-135) is one fused multiply-add under nvcc's default -fmad=true, like the dot
                 * product above: tests/golden/altcorr_kernel.npz (the kernel text run on the host with and without
                 * contraction) pins this form bit for bit */
                float* o;
                if (iy > 0 && ix > 0) { o = out + (long long)((iy - 1) + rd * (ix - 1)) * HW; *o = fmaf(s, dy * dx, *o); }
                if (iy > 0 && ix < rd) { o = out + (long long)((iy - 1) + rd * ix) * HW; *o = fmaf(s, dy * (1 - dx), *o); }
                if (iy < rd && ix > 0) { o = out + (long long)(iy + rd * (ix - 1)) * HW; *o = fmaf(s, (1 - dy) * dx, *o); }
                if (iy < rd && ix < rd) { o = out + (long long)(iy + rd * ix) * HW; *o = fmaf(s, (1 - dy) * (1 - dx), *o); }
              }
          }
  return 0;
}

/* altcorr_backward_kernel (altcorr_kernel.cu:152-286), fp32.  Sums carried in fp64 (the reference's
 * atomicAdd order is not defined). coords_grad is all zeros in the reference (:340) and is not produced. */
int oracle_altcorr_backward(const float* f1, const float* f2, const float* coords, const float* cg,
                            float* g1, float* g2, int B, int S, int H1, int W1, int H2, int W2, int C, int r) {
  const int rd = 2 * r + 1;
  const long long HW = (long long)H1 * W1;
  double* a1 = (double*)calloc((size_t)B * HW * C, sizeof(double));
  double* a2 = (double*)calloc((size_t)B * H2 * W2 * C + 1, sizeof(double));
  if (!a1 || !a2) { free(a1); free(a2); return 1; }
  for (int b = 0; b < B; b++)
    for (long long p = 0; p < HW; p++)
      for (int n = 0; n < S; n++) {
        const float* cp = coords + (((long long)b * S + n) * HW + p) * 2;
        const float x = cp[0], y = cp[1];
        const float dx = x - floorf(x), dy = y - floorf(y);
        const float* gp = cg + (((long long)b * S + n) * rd * rd) * HW + p;
        for (int iy = 0; iy < rd + 1; iy++)
          for (int ix = 0; ix < rd + 1; ix++) {
            const int h2 = floor_to_int(y) - r + iy, w2 = floor_to_int(x) - r + ix;
            float g = 0.0f;
            if (!within(h2, w2, H2, W2)) continue;
            if (iy > 0 && ix > 0) g += gp[(long long)((iy - 1) + rd * (ix - 1)) * HW] * dy * dx;
            if (iy > 0 && ix < rd) g += gp[(long long)((iy - 1) + rd * ix) * HW] * dy * (1 - dx);
            if (iy < rd && ix > 0) g += gp[(long long)(iy + rd * (ix - 1)) * HW] * (1 - dy) * dx;
            if (iy < rd && ix < rd) g += gp[(long long)(iy + rd * ix) * HW] * (1 - dy) * (1 - dx);
            for (int k = 0; k < C; k++) {
              a1[((long long)b * HW + p) * C + k] += (double)(g * f2[(((long long)b * H2 + h2) * W2 + w2) * C + k]);
              a2[(((long long)b * H2 + h2) * W2 + w2) * C + k] += (double)(g * f1[((long long)b * HW + p) * C + k]);
            }
          }
      }
  for (long long i = 0; i < (long long)B * HW * C; i++) g1[i] = (float)a1[i];
  for (long long i = 0; i < (long long)B * H2 * W2 * C; i++) g2[i] = (float)a2[i];
  free(a1); free(a2);
  return 0;
}
