// altcorr.hip — volume-free correlation lookup ("alternate" correlation) for gfx950.
//
// Replaces altcorr_forward_kernel / altcorr_backward_kernel
// (reference VO_Module/src/altcorr_kernel.cu:27-149, 152-286; hosts :290-356), the memory-light
// path AltCorrBlock uses for global BA (modules/corr.py:74-139, droid_backend.py:31-39).
//   fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C] channels-last fp32, coords [B,S,H1,W1,2]
//   corr  [B,S,(2r+1)^2,H1,W1], channel = iy + (2r+1)*ix  (x offset major, as the volume lookup)
//
// The reference uses 32-thread blocks (4x8 pixels), 32-channel slabs through shared memory and
// one global read-modify-write per tap per slab.  Here (r == 3):
//   forward : one WAVE per (pixel, sample): the 8x8 tap window is exactly 64 lanes.  Each lane
//             dots its tap's 128-channel row (contiguous 512 B) with the source row, neighbours
//             come from three shuffles, 49 lanes emit one output each; a workgroup covers 64
//             consecutive pixels and writes each channel row as one 256-byte segment through LDS.
//   backward: one wave per (pixel, sample), lanes over CHANNELS: per tap the scalar g (4 bilinear
//             terms of corr_grad) is wave-uniform, fmap1_grad accumulates in registers with no
//             reduction, fmap2_grad gets one coalesced row of atomicAdds per tap.
// fp32 only (AltCorrBlock casts to float, corr.py:120; the reference's backward is float-only,
// :345).  Accumulation order differs from the reference's slab order -> compared with a tolerance.
#include "common.h"

namespace {

constexpr int kPix = 64;        // pixels per workgroup
constexpr int kPad = 65;        // LDS row stride (floats)

__global__ __launch_bounds__(256) void altcorr_fwd_r3_kernel(
    const float* __restrict__ f1, const float* __restrict__ f2, const float* __restrict__ coords,
    float* __restrict__ corr, int S, int H1, int W1, int H2, int W2, int C) {
  __shared__ float stage[49 * kPad];
  const int HW = H1 * W1;
  const int bn = blockIdx.y;                       // b * S + n
  const int b = bn / S;
  const int pix0 = blockIdx.x * kPix;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int iy = lane >> 3, ix = lane & 7;

  for (int q = 0; q < 16; ++q) {
    const int p = wave * 16 + q;
    const int pix = pix0 + p;
    if (pix >= HW) break;                          // wave-uniform
    const float2 c = *reinterpret_cast<const float2*>(coords + (static_cast<long long>(bn) * HW + pix) * 2);
    const float fxl = floorf(c.x), fyl = floorf(c.y);
    const float dx = c.x - fxl, dy = c.y - fyl;
    const int w2 = pvo_floor_to_int(c.x) - 3 + ix;
    const int h2 = pvo_floor_to_int(c.y) - 3 + iy;
    float s = 0.0f;
    if (h2 >= 0 && h2 < H2 && w2 >= 0 && w2 < W2) {
      const float* __restrict__ a = f1 + (static_cast<long long>(b) * HW + pix) * C;          // wave-uniform row
      const float* __restrict__ r = f2 + ((static_cast<long long>(b) * H2 + h2) * W2 + w2) * C;
      if ((C & 3) == 0) {
        for (int k = 0; k < C; k += 4) {
          const float4 av = *reinterpret_cast<const float4*>(a + k);
          const float4 rv = *reinterpret_cast<const float4*>(r + k);
          s = fmaf(av.x, rv.x, s); s = fmaf(av.y, rv.y, s); s = fmaf(av.z, rv.z, s); s = fmaf(av.w, rv.w, s);
        }
      } else {
        for (int k = 0; k < C; ++k) s = fmaf(a[k], r[k], s);
      }
    }
    const float s_e = __shfl_down(s, 1, 64);       // tap (iy, ix+1)
    const float s_s = __shfl_down(s, 8, 64);       // tap (iy+1, ix)
    const float s_se = __shfl_down(s, 9, 64);      // tap (iy+1, ix+1)
    if (iy < 7 && ix < 7) {
      // altcorr_kernel.cu:104-137: cell (iy,ix) <- se of (iy,ix), sw of (iy,ix+1), ne of (iy+1,ix), nw of (iy+1,ix+1)
      float o = s * ((1.0f - dy) * (1.0f - dx));
      o += s_e * ((1.0f - dy) * dx);
      o += s_s * (dy * (1.0f - dx));
      o += s_se * (dy * dx);
      stage[(iy + 7 * ix) * kPad + p] = o;
    }
  }
  __syncthreads();
  const int npix = min(kPix, HW - pix0);
  float* out = corr + static_cast<long long>(bn) * 49 * HW + pix0;
  for (int idx = threadIdx.x; idx < 49 * kPix; idx += 256) {
    const int ch = idx >> 6, cpx = idx & 63;
    if (cpx < npix) out[static_cast<long long>(ch) * HW + cpx] = stage[ch * kPad + cpx];
  }
}

// any radius: one thread per (pixel, sample), gather form
__global__ __launch_bounds__(256) void altcorr_fwd_generic_kernel(
    const float* __restrict__ f1, const float* __restrict__ f2, const float* __restrict__ coords,
    float* __restrict__ corr, int S, int H1, int W1, int H2, int W2, int C, int r) {
  const int HW = H1 * W1;
  const int bn = blockIdx.y, b = bn / S;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= HW) return;
  const int rd = 2 * r + 1;
  const float cx = coords[(static_cast<long long>(bn) * HW + pix) * 2], cy = coords[(static_cast<long long>(bn) * HW + pix) * 2 + 1];
  const float dx = cx - floorf(cx), dy = cy - floorf(cy);
  const int x0 = pvo_floor_to_int(cx) - r, y0 = pvo_floor_to_int(cy) - r;
  const float* a = f1 + (static_cast<long long>(b) * HW + pix) * C;
  for (int ax = 0; ax < rd; ++ax)
    for (int by = 0; by < rd; ++by) {
      float o = 0.0f;
      for (int t = 0; t < 4; ++t) {
        const int h2 = y0 + by + (t >> 1), w2 = x0 + ax + (t & 1);
        if (h2 < 0 || h2 >= H2 || w2 < 0 || w2 >= W2) continue;
        const float* rr = f2 + ((static_cast<long long>(b) * H2 + h2) * W2 + w2) * C;
        float s = 0.0f;
        for (int k = 0; k < C; ++k) s = fmaf(a[k], rr[k], s);
        const float wy = (t >> 1) ? dy : (1.0f - dy), wx = (t & 1) ? dx : (1.0f - dx);
        o += s * (wy * wx);
      }
      corr[(static_cast<long long>(bn) * rd * rd + by + rd * ax) * HW + pix] = o;
    }
}

// backward: lanes over channels (2 per lane for C = 128), one wave per (pixel, sample)
__global__ __launch_bounds__(256) void altcorr_bwd_kernel(
    const float* __restrict__ f1, const float* __restrict__ f2, const float* __restrict__ coords,
    const float* __restrict__ cg, float* __restrict__ g1, float* __restrict__ g2,
    int S, int H1, int W1, int H2, int W2, int C, int r) {
  const int HW = H1 * W1;
  const int lane = threadIdx.x & 63;
  const int pix = blockIdx.x * 4 + (threadIdx.x >> 6);          // one wave per source pixel, all samples inside
  const int b = blockIdx.y;
  if (pix >= HW) return;
  const int rd = 2 * r + 1;
  const float* a = f1 + (static_cast<long long>(b) * HW + pix) * C;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + lane;
    const bool cok = c < C;
    const float av = cok ? a[c] : 0.0f;
    float acc1 = 0.0f;
    for (int n = 0; n < S; ++n) {
      const long long bn = static_cast<long long>(b) * S + n;
      const float cx = coords[(bn * HW + pix) * 2], cy = coords[(bn * HW + pix) * 2 + 1];
      const float dx = cx - floorf(cx), dy = cy - floorf(cy);
      const int x0 = pvo_floor_to_int(cx) - r, y0 = pvo_floor_to_int(cy) - r;
      const float* gp = cg + bn * rd * rd * HW + pix;
      for (int iy = 0; iy <= rd; ++iy)
        for (int ix = 0; ix <= rd; ++ix) {
          const int h2 = y0 + iy, w2 = x0 + ix;
          if (h2 < 0 || h2 >= H2 || w2 < 0 || w2 >= W2) continue;       // wave-uniform
          // altcorr_kernel.cu:232-243
          float g = 0.0f;
          if (iy > 0 && ix > 0)   g += gp[static_cast<long long>((iy - 1) + rd * (ix - 1)) * HW] * dy * dx;
          if (iy > 0 && ix < rd)  g += gp[static_cast<long long>((iy - 1) + rd * ix) * HW] * dy * (1.0f - dx);
          if (iy < rd && ix > 0)  g += gp[static_cast<long long>(iy + rd * (ix - 1)) * HW] * (1.0f - dy) * dx;
          if (iy < rd && ix < rd) g += gp[static_cast<long long>(iy + rd * ix) * HW] * (1.0f - dy) * (1.0f - dx);
          const long long ro = ((static_cast<long long>(b) * H2 + h2) * W2 + w2) * C;
          if (cok) {
            acc1 = fmaf(g, f2[ro + c], acc1);
            atomicAdd(g2 + ro + c, g * av);
          }
        }
    }
    if (cok) g1[(static_cast<long long>(b) * HW + pix) * C + c] = acc1;
  }
}

}  // namespace

extern "C" int pvo_altcorr_forward(const void* fmap1, const void* fmap2, const float* coords, void* corr,
                                   int B, int S, int H1, int W1, int H2, int W2, int C,
                                   int radius, int dtype, void* stream) {
  if (B < 0 || S < 0 || H1 < 0 || W1 < 0 || H2 < 0 || W2 < 0 || C <= 0 || radius < 0) return PVO_EINVAL;
  if (dtype != PVO_F32) return PVO_EUNSUPPORTED;
  if (B == 0 || S == 0 || H1 * W1 == 0) return PVO_OK;
  if (!fmap1 || !coords || !corr || (!fmap2 && H2 * W2 > 0)) return PVO_EINVAL;
  if (static_cast<long long>(B) * S > 65535) return PVO_EUNSUPPORTED;
  hipStream_t st = pvo_stream(stream);
  const int HW = H1 * W1;
  const bool al = (((reinterpret_cast<uintptr_t>(fmap1) | reinterpret_cast<uintptr_t>(fmap2)) & 15) == 0);
  if (radius == 3 && ((C & 3) != 0 || al))
    hipLaunchKernelGGL(altcorr_fwd_r3_kernel, dim3((HW + kPix - 1) / kPix, B * S), dim3(256), 0, st,
                       static_cast<const float*>(fmap1), static_cast<const float*>(fmap2), coords,
                       static_cast<float*>(corr), S, H1, W1, H2, W2, C);
  else
    hipLaunchKernelGGL(altcorr_fwd_generic_kernel, dim3((HW + 255) / 256, B * S), dim3(256), 0, st,
                       static_cast<const float*>(fmap1), static_cast<const float*>(fmap2), coords,
                       static_cast<float*>(corr), S, H1, W1, H2, W2, C, radius);
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

extern "C" int pvo_altcorr_backward(const void* fmap1, const void* fmap2, const float* coords,
                                    const void* corr_grad, void* fmap1_grad, void* fmap2_grad,
                                    int B, int S, int H1, int W1, int H2, int W2, int C,
                                    int radius, int dtype, void* stream) {
  if (B < 0 || S < 0 || H1 < 0 || W1 < 0 || H2 < 0 || W2 < 0 || C <= 0 || radius < 0) return PVO_EINVAL;
  if (dtype != PVO_F32) return PVO_EUNSUPPORTED;
  if (B == 0 || H1 * W1 == 0) return PVO_OK;
  if (!fmap1 || !fmap1_grad || (S > 0 && (!coords || !corr_grad)) || (H2 * W2 > 0 && (!fmap2 || !fmap2_grad))) return PVO_EINVAL;
  if (B > 65535) return PVO_EUNSUPPORTED;
  hipStream_t st = pvo_stream(stream);
  const int HW = H1 * W1;
  // fmap2_grad accumulates with atomics: zero it here so the caller may hand over torch.empty
  if (H2 * W2 > 0 && hipMemsetAsync(fmap2_grad, 0, sizeof(float) * static_cast<size_t>(B) * H2 * W2 * C, st) != hipSuccess)
    return PVO_ELAUNCH;
  hipLaunchKernelGGL(altcorr_bwd_kernel, dim3((HW + 3) / 4, B), dim3(256), 0, st,
                     static_cast<const float*>(fmap1), static_cast<const float*>(fmap2), coords,
                     static_cast<const float*>(corr_grad), static_cast<float*>(fmap1_grad),
                     static_cast<float*>(fmap2_grad), S, H1, W1, H2, W2, C, radius);
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}
