// corr_lookup.hip — correlation-pyramid lookup for gfx950 (MI355X).
//
// Replaces corr_index_forward_kernel / corr_index_backward_kernel
// (reference VO_Module/src/correlation_kernels.cu:19-70, 73-124) and the Python
// per-level loop + torch.cat of CorrBlock.__call__ (modules/corr.py:40-50).
//
// Design (not a translation of the reference's 16x16 one-thread-per-pixel scatter):
//   * r == 3 fast path: a 256-thread workgroup owns a strip of 64 consecutive pixels
//     of one edge.  8 lanes cooperate on one pixel: lane `r` fetches tap ROW r of the
//     8x8 window as one 16/32-byte segment (dwordx4+dword for 16-bit volumes), the
//     neighbouring row comes from lane r+1 by a wave shuffle, and the lane produces
//     the 7 outputs of its row in the reference's accumulation order.  All pyramid
//     levels are gathered in the same launch; results are transposed through LDS so
//     that every output channel row is written as one contiguous 64-pixel segment
//     (the reference does 49 read-modify-writes per pixel on a pre-zeroed tensor).
//   * generic path (any radius, fp64): one thread per pixel, gather form.
//
// Numerics: bit-exact with the reference's arithmetic model.
//   fp16/bf16: every product and every sum is formed in fp32 and rounded to the
//              storage type (c10::Half operators), weights are fp32 products rounded
//              to the storage type; accumulation order per output cell is taps
//              (a,b), (a,b+1), (a+1,b), (a+1,b+1)   [a: x offset, b: y offset].
//   fp32/fp64: acc = fma(s, w, acc) — nvcc contracts `corr += s * w` (default
//              -fmad=true), so the CUDA reference path is an FMA chain.
// Floating-point contraction is OFF for this file; the FMAs are explicit.
#pragma clang fp contract(off)

#include "common.h"

namespace {

constexpr int kMaxLevels = 4;
constexpr int kStrip = 64;          // pixels per workgroup (two passes of 32)
constexpr int kEncStrip = 32;       // ... of the fused lookup + encoder: one pass; twice the workgroups at half the LDS each fill the
                                    // CUs' last round (1728 workgroups of 64 pixels were 6.75 per CU at 5 resident: two rounds, the
                                    // second a third full - in-kernel clock stamps, tools/lookup_timeline.py)
constexpr int kStripPad = 66;       // LDS row stride in elements (keeps 4-byte alignment, spreads banks)
constexpr int kEncK = 224;          // ENC: 196 lookup channels padded to 7 MFMA k-steps of 32
constexpr int kEncStride = 232;     // ENC: staging row of a pixel in elements (464 B: 16-byte aligned, odd multiple of 16 B)
constexpr int kEncOutStride = 136;  // ENC: output slab row in elements (272 B)
constexpr int kPixPad = 4;          // channels-last staging: row of nch + 4 elements per pixel (8-byte aligned rows)

struct LookupLevel {
  const void* vol;    // [N*HW planes][h2][w2]
  long long total;    // elements in the level tensor
  int h2, w2;
  float scale;        // 1 / 2^level (exact)
  int tw;             // tiled layout only: 8x8 tiles per row of a plane
  long long plane_elems;  // tiled layout only: elements per plane = th * tw * 64
};

struct LookupArgs {
  LookupLevel lv[kMaxLevels];
  const float* coords;
  void* out;
  int nlev;
  int coords_interleaved;  // 0: [N,2,h1,w1]   1: [N,h1,w1,2]
  int out_channels_last;   // 0: out [N,nch,h1,w1]   1: out [N,h1,w1,nch] (what the NHWC convolutions read)
  const int* slots;        // optional: volume of edge n lives in slot slots[n] of a pool (NULL: slot n)
  const void* enc_w;       // ENC kernels: 1x1 encoder weights [128 outputs][224 = 196 + 28 zero] in the volume dtype
  const float* enc_b;      // ENC kernels: encoder bias f32 [128]
  int HW;                  // h1*w1
  int N;
};

template <typename T> struct Arith;  // value domain = float holding a representable value
template <> struct Arith<float> {
  static __device__ __forceinline__ float rnd(float x) { return x; }
  static __device__ __forceinline__ float step(float acc, float s, float w) { return fmaf(s, w, acc); }
};
template <> struct Arith<pvo_half> {
  // The fp32 value is made opaque before the conversion: hipcc otherwise fuses
  // `half(a*b)` into v_fma_mixlo_f16, which rounds the exact product ONCE to fp16;
  // the reference rounds to fp32 first and then to fp16 (scalar_t(dx*dy)).
  static __device__ __forceinline__ float rnd(float x) {
    asm volatile("" : "+v"(x));
    return static_cast<float>(static_cast<_Float16>(x));
  }
  static __device__ __forceinline__ float step(float acc, float s, float w) { return rnd(acc + rnd(s * w)); }
};
template <> struct Arith<pvo_bf16> {
  static __device__ __forceinline__ float rnd(float x) { return pvo_bf16_to_f32(pvo_f32_to_bf16(x)); }
  static __device__ __forceinline__ float step(float acc, float s, float w) { return rnd(acc + rnd(s * w)); }
};

struct __attribute__((packed, aligned(4))) U4A4 { uint32_t x, y, z, w; };

__device__ __forceinline__ float h16_to_f32(uint32_t bits, pvo_half*) {
  union { uint16_t u; _Float16 h; } c; c.u = static_cast<uint16_t>(bits);
  return static_cast<float>(c.h);
}
__device__ __forceinline__ float h16_to_f32(uint32_t bits, pvo_bf16*) {
  return pvo_bf16_to_f32(static_cast<uint16_t>(bits));
}

// Fetch 8 consecutive 16-bit elements g..g+7 of a level tensor as 4 packed dwords
// (element 2k in the low half of u[k]).  Elements whose mask bit is clear are never
// dereferenced outside [0,total); their lanes of u[] hold don't-care bits.
__device__ __forceinline__ void fetch_row8_16(const uint16_t* base, long long g, long long total,
                                             uint32_t mask, uint32_t u[4]) {
  const long long ga = g & ~1LL;
  u[0] = u[1] = u[2] = u[3] = 0u;
  if (mask == 0) return;
  if (ga >= 0 && ga + 10 <= total) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(base + ga);
    const U4A4 q = *reinterpret_cast<const U4A4*>(p);
    const uint32_t q4 = p[4];
    const uint32_t sh = static_cast<uint32_t>(g & 1) * 2u;  // byte shift
    u[0] = __builtin_amdgcn_alignbyte(q.y, q.x, sh);
    u[1] = __builtin_amdgcn_alignbyte(q.z, q.y, sh);
    u[2] = __builtin_amdgcn_alignbyte(q.w, q.z, sh);
    u[3] = __builtin_amdgcn_alignbyte(q4, q.w, sh);
  } else {  // first / last few elements of the tensor: guarded element loads
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t e = 0u;
      if ((mask >> j) & 1u) e = base[g + j];
      u[j >> 1] |= e << ((j & 1) * 16);
    }
  }
}

// Tiled level layout [plane][th][tw][8][8] (16-bit elements: one tile = one 128-byte line, so an 8x8 tap
// window touches at most 4 lines instead of 8 row segments in 8 different lines).  Fetch row iy, columns
// ix..ix+7: they live in row (iy & 7) of tiles c0 = ix >> 3 and c0 + 1; two aligned 16-byte loads and a
// funnel shift by (ix & 7) elements.  Elements outside [0, w2) hold don't-care bits (masked by the caller).
typedef uint32_t lk_u32x4 __attribute__((ext_vector_type(4)));
// The two halves are separate so that a lane can have the loads of all pyramid levels in flight before it consumes
// the first one (with the fetch inside the level loop the compiler waits after every level: two loads in flight per
// lane, the HBM latency exposed eight times per workgroup).  Addresses are clamped into the plane instead of
// predicated: no exec-mask branches between the loads; whatever a clamped load returns is masked by the caller.
__device__ __forceinline__ void tiled_issue(const uint16_t* base, long long pbase, int tw, int h2, int iy, int ix,
                                            lk_u32x4& A, lk_u32x4& B) {
  const int c0 = ix >> 3;
  const int iyc = min(max(iy, 0), h2 - 1);
  const int ca = min(max(c0, 0), tw - 1), cb = min(max(c0 + 1, 0), tw - 1);
  const uint16_t* rowp = base + pbase + static_cast<long long>(iyc >> 3) * tw * 64 + (iyc & 7) * 8;
  A = *reinterpret_cast<const lk_u32x4*>(rowp + ca * 64);
  B = *reinterpret_cast<const lk_u32x4*>(rowp + cb * 64);
}

__device__ __forceinline__ void tiled_finish(lk_u32x4 A, lk_u32x4 B, int ix, uint32_t u[4]) {
  const int s = ix & 7;
  // funnel shift of the 8 dwords (A:B) by s elements, written with scalars only (local arrays of the
  // selects end up in scratch memory: 64 B/lane and 5x the HBM write traffic, measured)
  const bool by2 = (s & 4) != 0, by1 = (s & 2) != 0;
  const uint32_t p0 = by2 ? A.z : A.x, p1 = by2 ? A.w : A.y, p2 = by2 ? B.x : A.z, p3 = by2 ? B.y : A.w,
                 p4 = by2 ? B.z : B.x, p5 = by2 ? B.w : B.y;
  const uint32_t q0 = by1 ? p1 : p0, q1 = by1 ? p2 : p1, q2 = by1 ? p3 : p2, q3 = by1 ? p4 : p3, q4 = by1 ? p5 : p4;
  const uint32_t sh = static_cast<uint32_t>(s & 1) * 2u;
  u[0] = __builtin_amdgcn_alignbyte(q1, q0, sh);
  u[1] = __builtin_amdgcn_alignbyte(q2, q1, sh);
  u[2] = __builtin_amdgcn_alignbyte(q3, q2, sh);
  u[3] = __builtin_amdgcn_alignbyte(q4, q3, sh);
}

// value of lane+1 within a 16-lane row (DPP row_shl:1; the last lane of a row reads 0): the next tap row of the same
// pixel for the 8-lane groups used here, without the LDS crossbar a __shfl_down goes through
__device__ __forceinline__ uint32_t next_lane(uint32_t x) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x101, 0xf, 0xf, true));
}

__device__ __forceinline__ void fetch_row8_32(const float* base, long long g, uint32_t mask, float v[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    v[j] = 0.0f;
    if ((mask >> j) & 1u) v[j] = base[g + j];
  }
}

template <typename T>
__device__ __forceinline__ void unpack8(const uint32_t u[4], float v[8]) {
  T* tag = nullptr;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[2 * k] = h16_to_f32(u[k] & 0xffffu, tag);
    v[2 * k + 1] = h16_to_f32(u[k] >> 16, tag);
  }
}

// ---------------------------------------------------------------------------
// r == 3 fast path.
// ---------------------------------------------------------------------------
typedef float lk_v4f __attribute__((ext_vector_type(4)));
typedef _Float16 lk_v8h __attribute__((ext_vector_type(8)));
typedef __bf16 lk_v8b __attribute__((ext_vector_type(8)));
template <typename T> __device__ __forceinline__ lk_v4f lk_mfma(lk_u32x4 x, lk_u32x4 w, lk_v4f c);
template <> __device__ __forceinline__ lk_v4f lk_mfma<pvo_half>(lk_u32x4 x, lk_u32x4 w, lk_v4f c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(lk_v8h, x), __builtin_bit_cast(lk_v8h, w), c, 0, 0, 0);
}
template <> __device__ __forceinline__ lk_v4f lk_mfma<pvo_bf16>(lk_u32x4 x, lk_u32x4 w, lk_v4f c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lk_v8b, x), __builtin_bit_cast(lk_v8b, w), c, 0, 0, 0);
}
template <> __device__ __forceinline__ lk_v4f lk_mfma<float>(lk_u32x4, lk_u32x4, lk_v4f c) { return c; }   // never instantiated with ENC

// ENC: the lookup's 196 channels never leave the workgroup; they feed the update operator's first correlation-encoder
// layer, relu(W corr + b) with W [128,196] (droid_net.py:172-175: Conv2d(196,128,1) + ReLU), on the matrix cores, and the
// 128 encoded channels are written instead: 28 MB of output instead of 43 MB, and the 1x1 convolution's 27 us + 9 us
// (conv, bias/ReLU pass) with their 43 + 28 + 56 MB of traffic disappear.
// tools/lookup_timeline.py builds this file with -DPVO_LOOKUP_PROBE: every workgroup records shader-clock stamps of its phases
#ifdef PVO_LOOKUP_PROBE
__device__ unsigned long long* g_lk_probe = nullptr;
#define LK_PROBE(slot)                                                                                              \
  do {                                                                                                              \
    if (g_lk_probe && threadIdx.x == 0)                                                                             \
      g_lk_probe[(static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define LK_PROBE(slot)
#endif

#define LK_SYNC() __syncthreads()

template <typename T, bool TILED, bool ENC>
__device__ __forceinline__ void corr_lookup_r3_body(const LookupArgs& a) {
  using S = typename Elem<T>::store_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  S* stage = reinterpret_cast<S*>(smem_raw);  // [nlev*49][kStripPad]

  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  constexpr int STRIP = ENC ? kEncStrip : kStrip;
  const int pix0 = blockIdx.x * STRIP;
  const int row = tid & 7;        // tap row (y offset index) handled by this lane
  const int HW = a.HW;
  const int cl_stride = ENC ? kEncStride : a.nlev * 49 + kPixPad;   // channels-last staging is pixel-major
  if constexpr (ENC) {     // zero the k-padding columns 196..231 of every pixel row (0 x garbage could be NaN)
    uint32_t* z = reinterpret_cast<uint32_t*>(smem_raw) + (tid >> 2) * (kEncStride / 2) + 98 + (tid & 3);
    if ((tid >> 2) < STRIP) {
#pragma unroll
      for (int q = 0; q < 5; ++q)
        if (98 + (tid & 3) + 4 * q < kEncStride / 2) z[4 * q] = 0u;
    }
  }

  LK_PROBE(0);
#pragma unroll 1
  for (int pass = 0; pass < STRIP / 32; ++pass) {
    if (pass == 1) LK_PROBE(1);
    const int p = pass * 32 + (tid >> 3);
    const int pix = pix0 + p;
    const bool pix_ok = pix < HW;
    float x0 = 0.f, y0 = 0.f;
    if (pix_ok) {
      if (a.coords_interleaved) {
        const float2 c = *reinterpret_cast<const float2*>(a.coords + (static_cast<long long>(n) * HW + pix) * 2);
        x0 = c.x; y0 = c.y;
      } else {
        x0 = a.coords[(static_cast<long long>(n) * 2) * HW + pix];
        y0 = a.coords[(static_cast<long long>(n) * 2 + 1) * HW + pix];
      }
    }
    const long long plane = static_cast<long long>(a.slots ? a.slots[n] : n) * HW + (pix_ok ? pix : pix0);

    lk_u32x4 RA[kMaxLevels], RB[kMaxLevels];
    if constexpr (TILED && sizeof(S) == 2) {
#pragma unroll
      for (int l = 0; l < kMaxLevels; ++l) {
        if (l < a.nlev) {
          const LookupLevel L = a.lv[l];
          const int ixp = pvo_floor_to_int(x0 * L.scale) - 3, iyp = pvo_floor_to_int(y0 * L.scale) - 3 + row;
          tiled_issue(reinterpret_cast<const uint16_t*>(L.vol), plane * L.plane_elems, L.tw, L.h2, iyp, ixp, RA[l], RB[l]);
        }
      }
    }

#pragma unroll
    for (int l = 0; l < kMaxLevels; ++l) {
      if (l >= a.nlev) break;
      const LookupLevel L = a.lv[l];
      const float xs = x0 * L.scale;   // == x0 / 2^l exactly (corr.py:47)
      const float ys = y0 * L.scale;
      const float dx = xs - floorf(xs);
      const float dy = ys - floorf(ys);
      const int ix = pvo_floor_to_int(xs) - 3;
      const int iy = pvo_floor_to_int(ys) - 3 + row;
      const bool rowok = pix_ok && iy >= 0 && iy < L.h2;
      uint32_t mask = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        mask |= (rowok && static_cast<unsigned>(ix + j) < static_cast<unsigned>(L.w2)) ? (1u << j) : 0u;
      const long long g = (plane * L.h2 + iy) * L.w2 + ix;

      // own tap row, and the next one (y offset row+1) from lane+1 of this 8-lane group
      float v[8], vn[8];
      const uint32_t maskn = next_lane(mask);
      if constexpr (__is_same(T, pvo_half)) {
        // fp16: NATIVE packed arithmetic, two outputs per instruction.  The reference computes every product and sum in
        // fp32 and rounds to fp16 (c10::Half operators); fp32 carries 24 >= 2 * 11 + 2 significand bits, so that double
        // rounding is innocuous and v_pk_mul_f16 / v_pk_add_f16 (one rounding of the exact result each) return the same
        // bits for every finite input - the bit-exactness tests against the oracle are unchanged.  (The scalar form below -
        // mul, round, add, round with four fp32 <-> fp16 conversions per tap - made this kernel VALU-bound: a back-to-back
        // run from the Infinity Cache took 43 us against 49 us from HBM.)  Products and sums stay separate instructions
        // (contraction is off in this file); a packed FMA would round once.
        typedef _Float16 lk_h2 __attribute__((ext_vector_type(2)));
        uint32_t u[5], un[5];
        if constexpr (TILED)
          tiled_finish(RA[l], RB[l], ix, u);
        else
          fetch_row8_16(reinterpret_cast<const uint16_t*>(L.vol), g, L.total, mask, u);
#pragma unroll
        for (int k = 0; k < 4; ++k) un[k] = next_lane(u[k]);
        u[4] = 0u; un[4] = 0u;
        auto bc = [](float w) { const _Float16 h = static_cast<_Float16>(w); return lk_h2{h, h}; };      // w is already an fp16 value
        const lk_h2 W00 = bc(Arith<T>::rnd((1.0f - dx) * (1.0f - dy))), W01 = bc(Arith<T>::rnd((1.0f - dx) * dy));
        const lk_h2 W10 = bc(Arith<T>::rnd(dx * (1.0f - dy))), W11 = bc(Arith<T>::rnd(dx * dy));
        // per-half select masks: pair k = taps (2k, 2k+1) [aligned] and (2k+1, 2k+2) [shifted by one tap]
        auto pairmask = [](uint32_t m, int k) {
          const uint32_t e0 = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(m), 2 * k, 1));
          const uint32_t e1 = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(m), 2 * k + 1, 1));
          return __builtin_amdgcn_perm(e1, e0, 0x05040100u);           // (e1.lo16 << 16) | e0.lo16
        };
        // wave-uniform: interior windows need no selects (lanes of tap row 7 produce no output: their maskn belongs to a neighbour)
        const bool full = __all(static_cast<int>(row == 7 || (mask == 0xffu && maskn == 0xffu)));
        if (row < 7) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t ub = __builtin_amdgcn_alignbyte(u[k + 1], u[k], 2), unb = __builtin_amdgcn_alignbyte(un[k + 1], un[k], 2);
            const lk_h2 A = __builtin_bit_cast(lk_h2, u[k]), An = __builtin_bit_cast(lk_h2, un[k]);
            const lk_h2 B = __builtin_bit_cast(lk_h2, ub), Bn = __builtin_bit_cast(lk_h2, unb);
            lk_h2 acc = {static_cast<_Float16>(0.0f), static_cast<_Float16>(0.0f)};
            if (full) {
              acc = acc + A * W00;
              acc = acc + An * W01;
              acc = acc + B * W10;
              acc = acc + Bn * W11;
            } else {
              // order: taps (ax,row) (ax,row+1) (ax+1,row) (ax+1,row+1); skipped taps leave acc untouched
              const uint32_t mA = pairmask(mask, k), mAn = pairmask(maskn, k);
              const uint32_t mB = __builtin_amdgcn_alignbyte(k < 3 ? pairmask(mask, k + 1) : 0u, mA, 2);
              const uint32_t mBn = __builtin_amdgcn_alignbyte(k < 3 ? pairmask(maskn, k + 1) : 0u, mAn, 2);
              auto sel = [](uint32_t m, lk_h2 t, lk_h2 a) {
                return __builtin_bit_cast(lk_h2, (__builtin_bit_cast(uint32_t, t) & m) | (__builtin_bit_cast(uint32_t, a) & ~m));
              };
              acc = sel(mA, acc + A * W00, acc);
              acc = sel(mAn, acc + An * W01, acc);
              acc = sel(mB, acc + B * W10, acc);
              acc = sel(mBn, acc + Bn * W11, acc);
            }
            const uint32_t bits = __builtin_bit_cast(uint32_t, acc);
            const int ch = l * 49 + (2 * k) * 7 + row;
            reinterpret_cast<uint16_t*>(stage)[(ENC || a.out_channels_last) ? p * cl_stride + ch : ch * kStripPad + p] = static_cast<uint16_t>(bits & 0xffffu);
            if (k < 3)
              reinterpret_cast<uint16_t*>(stage)[(ENC || a.out_channels_last) ? p * cl_stride + ch + 7 : (ch + 7) * kStripPad + p] = static_cast<uint16_t>(bits >> 16);
          }
        }
        continue;
      } else if constexpr (sizeof(S) == 2) {
        uint32_t u[4], un[4];
        if constexpr (TILED)
          tiled_finish(RA[l], RB[l], ix, u);
        else
          fetch_row8_16(reinterpret_cast<const uint16_t*>(L.vol), g, L.total, mask, u);
#pragma unroll
        for (int k = 0; k < 4; ++k) un[k] = next_lane(u[k]);
        unpack8<T>(u, v);
        unpack8<T>(un, vn);
      } else {
        fetch_row8_32(reinterpret_cast<const float*>(L.vol), g, mask, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) vn[j] = __shfl_down(v[j], 1, 64);
      }

      // weights: fp32 products rounded to the storage type (correlation_kernels.cu:56-65)
      const float w11 = Arith<T>::rnd(dx * dy);
      const float w10 = Arith<T>::rnd(dx * (1.0f - dy));
      const float w01 = Arith<T>::rnd((1.0f - dx) * dy);
      const float w00 = Arith<T>::rnd((1.0f - dx) * (1.0f - dy));

      if (row < 7) {
#pragma unroll
        for (int ax = 0; ax < 7; ++ax) {
          // order: taps (ax,row) (ax,row+1) (ax+1,row) (ax+1,row+1); skipped taps leave acc untouched
          float acc = 0.0f, t;
          t = Arith<T>::step(acc, v[ax], w00);      acc = ((mask >> ax) & 1u) ? t : acc;
          t = Arith<T>::step(acc, vn[ax], w01);     acc = ((maskn >> ax) & 1u) ? t : acc;
          t = Arith<T>::step(acc, v[ax + 1], w10);  acc = ((mask >> (ax + 1)) & 1u) ? t : acc;
          t = Arith<T>::step(acc, vn[ax + 1], w11); acc = ((maskn >> (ax + 1)) & 1u) ? t : acc;
          const int ch = l * 49 + ax * 7 + row;
          stage[(ENC || a.out_channels_last) ? p * cl_stride + ch : ch * kStripPad + p] = Elem<T>::from_f32(acc);
        }
      }
    }
  }
  LK_PROBE(2);
  LK_SYNC();
  LK_PROBE(3);

  if constexpr (ENC && sizeof(S) == 2) {
    // encoded[px][n] = relu(sum_k corr[px][k] W[n][k] + b[n]); wave w owns outputs [32w, 32w+32), 16 at a time so that
    // only 7 weight fragments are live (they are fetched here, after the gather phase, to keep its occupancy)
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const uint16_t* W16 = reinterpret_cast<const uint16_t*>(a.enc_w);
    unsigned char* oslab = smem_raw;                                     // [64 px][272 B], over the staging once it is consumed
    constexpr int MT = STRIP / 16;                                       // 16-pixel M-tiles
    lk_v4f dd[2][MT];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int nn = wave * 32 + nt * 16 + li;
      lk_u32x4 wf[kEncK / 32];
#pragma unroll
      for (int ks = 0; ks < kEncK / 32; ++ks)
        wf[ks] = *reinterpret_cast<const lk_u32x4*>(W16 + static_cast<size_t>(nn) * kEncK + ks * 32 + lk * 8);
#pragma unroll
      for (int g = 0; g < MT; ++g) {
        lk_v4f d = {0.f, 0.f, 0.f, 0.f};
        const unsigned char* arow = smem_raw + (g * 16 + li) * (kEncStride * 2) + lk * 16;
#pragma unroll
        for (int ks = 0; ks < kEncK / 32; ++ks)
          d = lk_mfma<T>(*reinterpret_cast<const lk_u32x4*>(arow + ks * 64), wf[ks], d);
        dd[nt][g] = d;
      }
      asm volatile("" ::: "memory");      // keep the second tile's weight loads below the first tile's MFMAs (registers)
    }
    LK_PROBE(4);
    LK_SYNC();                                                        // every wave has read its A fragments
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int nn = wave * 32 + nt * 16 + li;
      const float bn = a.enc_b[nn];
#pragma unroll
      for (int g = 0; g < MT; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r)                                   // D rows lk*4 + r = pixels, column li = output nn
          *reinterpret_cast<S*>(oslab + (g * 16 + lk * 4 + r) * (kEncOutStride * 2) + nn * 2) =
              Elem<T>::from_f32(fmaxf(dd[nt][g][r] + bn, 0.0f));
    }
    LK_SYNC();
    const int npix = min(STRIP, HW - pix0);
    S* o = reinterpret_cast<S*>(a.out) + (static_cast<long long>(n) * HW + pix0) * 128;
#pragma unroll
    for (int it = 0; it < STRIP / 16; ++it) {                         // STRIP px x 16 chunks of 16 B: whole 256-byte pixel rows
      const int id = tid + 256 * it, px = id >> 4, c = id & 15;
      if (px < npix)
        *reinterpret_cast<lk_u32x4*>(o + static_cast<long long>(px) * 128 + c * 8) =
            *reinterpret_cast<const lk_u32x4*>(oslab + px * (kEncOutStride * 2) + c * 16);
    }
    LK_PROBE(5);
#ifdef PVO_LOOKUP_PROBE
    if (g_lk_probe && threadIdx.x == 0)
      g_lk_probe[(static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 8 + 6] =
          (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg(20 | (31 << 11))) << 32) | __builtin_amdgcn_s_getreg(4 | (31 << 11));
#endif
    return;
  }
  // coalesced write-out: channel rows of 64 pixels
  const int nch = a.nlev * 49;
  S* outp = reinterpret_cast<S*>(a.out) + static_cast<long long>(n) * nch * HW;
  const int npix = min(kStrip, HW - pix0);
  if (a.out_channels_last) {
    // the strip's outputs are one contiguous run of npix*nch elements, staged pixel-major.  16-bit: 8-byte
    // (4-channel) stores, nch = 49*nlev so a pixel is a whole number of them only when nlev is even; a wave then
    // writes 512 contiguous, 512-byte aligned bytes per instruction (2-byte stores cost 5x the HBM write traffic:
    // partial-line writes, measured with WRITE_SIZE).
    S* o = outp + static_cast<long long>(pix0) * nch;
    if constexpr (sizeof(S) == 2) {
      if ((nch & 3) == 0 && (reinterpret_cast<uintptr_t>(o) & 7) == 0) {
        const int qpp = nch >> 2;                              // 8-byte chunks per pixel
        for (int idx = tid; idx < npix * qpp; idx += 256) {
          const int c = idx / qpp, q = idx - c * qpp;
          *reinterpret_cast<uint2*>(o + static_cast<long long>(idx) * 4) =
              *reinterpret_cast<const uint2*>(&stage[c * cl_stride + q * 4]);
        }
        return;
      }
    }
    for (int idx = tid; idx < npix * nch; idx += 256) {
      const int c = idx / nch, ch = idx - c * nch;
      o[idx] = stage[c * cl_stride + ch];
    }
    return;
  }
  if constexpr (sizeof(S) == 2) {
    if (((HW | pix0) & 1) == 0 && (npix & 1) == 0 && ((reinterpret_cast<uintptr_t>(a.out) & 3) == 0)) {
      const int half = npix >> 1;
      for (int idx = tid; idx < nch * 32; idx += 256) {
        const int ch = idx >> 5, c = idx & 31;
        if (c < half) {
          const uint32_t val = *reinterpret_cast<const uint32_t*>(&stage[ch * kStripPad + 2 * c]);
          *reinterpret_cast<uint32_t*>(&outp[static_cast<long long>(ch) * HW + pix0 + 2 * c]) = val;
        }
      }
      return;
    }
  }
  for (int idx = tid; idx < nch * kStrip; idx += 256) {
    const int ch = idx >> 6, c = idx & 63;
    if (c < npix) outp[static_cast<long long>(ch) * HW + pix0 + c] = stage[ch * kStripPad + c];
  }
}

template <typename T, bool TILED, bool ENC>
__global__ __launch_bounds__(256) void corr_lookup_r3_kernel(LookupArgs a) { corr_lookup_r3_body<T, TILED, ENC>(a); }

// The fused lookup + encoder as its own entry: left alone the register allocator spends 97 VGPRs + 16 AGPRs on it (four
// waves per SIMD, 3.3 workgroups resident per CU on average by the in-kernel clock stamps); asked for six to eight waves
// it needs 56 and no scratch.
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void corr_lookup_r3_enc_kernel(LookupArgs a) {
  corr_lookup_r3_body<T, true, true>(a);
}

// ---------------------------------------------------------------------------
// generic path: any radius, also fp64.  One thread per pixel, gather form, same
// accumulation order and rounding model.
// ---------------------------------------------------------------------------
template <typename T> struct GArith {  // 16/32-bit types reuse Arith via float domain
  using val_t = float;
  static __device__ __forceinline__ float load(const void* p, long long i) {
    return Elem<T>::to_f32(reinterpret_cast<const typename Elem<T>::store_t*>(p)[i]);
  }
  static __device__ __forceinline__ void store(void* p, long long i, float v) {
    reinterpret_cast<typename Elem<T>::store_t*>(p)[i] = Elem<T>::from_f32(v);
  }
  static __device__ __forceinline__ float rnd(float x) { return Arith<T>::rnd(x); }
  static __device__ __forceinline__ float step(float acc, float s, float w) { return Arith<T>::step(acc, s, w); }
};
template <> struct GArith<double> {
  using val_t = double;
  static __device__ __forceinline__ double load(const void* p, long long i) { return reinterpret_cast<const double*>(p)[i]; }
  static __device__ __forceinline__ void store(void* p, long long i, double v) { reinterpret_cast<double*>(p)[i] = v; }
  static __device__ __forceinline__ double rnd(float x) { return static_cast<double>(x); }
  static __device__ __forceinline__ double step(double acc, double s, double w) { return fma(s, w, acc); }
};

template <typename T>
__global__ __launch_bounds__(256) void corr_lookup_generic_kernel(LookupArgs a, int r) {
  using G = GArith<T>;
  using V = typename G::val_t;
  const int HW = a.HW;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (pix >= HW) return;
  float x0, y0;
  if (a.coords_interleaved) {
    x0 = a.coords[(static_cast<long long>(n) * HW + pix) * 2];
    y0 = a.coords[(static_cast<long long>(n) * HW + pix) * 2 + 1];
  } else {
    x0 = a.coords[(static_cast<long long>(n) * 2) * HW + pix];
    y0 = a.coords[(static_cast<long long>(n) * 2 + 1) * HW + pix];
  }
  const int rd = 2 * r + 1;
  const long long plane = static_cast<long long>(a.slots ? a.slots[n] : n) * HW + pix;
  for (int l = 0; l < a.nlev; ++l) {
    const LookupLevel L = a.lv[l];
    const float xs = x0 * L.scale, ys = y0 * L.scale;
    const float dx = xs - floorf(xs), dy = ys - floorf(ys);
    const int ix = pvo_floor_to_int(xs) - r, iy = pvo_floor_to_int(ys) - r;
    const V w11 = G::rnd(dx * dy), w10 = G::rnd(dx * (1.0f - dy));
    const V w01 = G::rnd((1.0f - dx) * dy), w00 = G::rnd((1.0f - dx) * (1.0f - dy));
    const long long pbase = plane * L.h2 * L.w2;
    for (int ax = 0; ax < rd; ++ax) {
      for (int by = 0; by < rd; ++by) {
        V acc = 0;
        const int xa = ix + ax, yb = iy + by;
        const bool x0ok = xa >= 0 && xa < L.w2, x1ok = xa + 1 >= 0 && xa + 1 < L.w2;
        const bool y0ok = yb >= 0 && yb < L.h2, y1ok = yb + 1 >= 0 && yb + 1 < L.h2;
        if (x0ok && y0ok) acc = G::step(acc, G::load(L.vol, pbase + static_cast<long long>(yb) * L.w2 + xa), w00);
        if (x0ok && y1ok) acc = G::step(acc, G::load(L.vol, pbase + static_cast<long long>(yb + 1) * L.w2 + xa), w01);
        if (x1ok && y0ok) acc = G::step(acc, G::load(L.vol, pbase + static_cast<long long>(yb) * L.w2 + xa + 1), w10);
        if (x1ok && y1ok) acc = G::step(acc, G::load(L.vol, pbase + static_cast<long long>(yb + 1) * L.w2 + xa + 1), w11);
        const long long ch = static_cast<long long>(l) * rd * rd + ax * rd + by;
        if (a.out_channels_last) G::store(a.out, (static_cast<long long>(n) * HW + pix) * (a.nlev * rd * rd) + ch, acc);
        else G::store(a.out, (static_cast<long long>(n) * a.nlev * rd * rd + ch) * HW + pix, acc);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// backward: volume_grad[n,y,x,:,:] written densely by one wave per plane
// (zero outside the (2r+2)^2 window), so no pre-zeroing and no read-modify-write.
// g accumulates in the reference's order (correlation_kernels.cu:107-118).
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void corr_lookup_backward_kernel(
    const float* __restrict__ coords, const void* __restrict__ corr_grad, void* __restrict__ vol_grad,
    int N, int HW, int h2, int w2, int r) {
  using G = GArith<T>;
  using V = typename G::val_t;
  const int lane = threadIdx.x & 63;
  const long long plane = static_cast<long long>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (plane >= static_cast<long long>(N) * HW) return;
  const int n = static_cast<int>(plane / HW);
  const int pix = static_cast<int>(plane - static_cast<long long>(n) * HW);
  const float x0 = coords[(static_cast<long long>(n) * 2) * HW + pix];
  const float y0 = coords[(static_cast<long long>(n) * 2 + 1) * HW + pix];
  const float dx = x0 - floorf(x0), dy = y0 - floorf(y0);
  const int ix = pvo_floor_to_int(x0) - r, iy = pvo_floor_to_int(y0) - r;
  const int rd = 2 * r + 1;
  const V w11 = G::rnd(dx * dy), w10 = G::rnd(dx * (1.0f - dy));
  const V w01 = G::rnd((1.0f - dx) * dy), w00 = G::rnd((1.0f - dx) * (1.0f - dy));
  const long long gbase = static_cast<long long>(n) * rd * rd * HW + pix;  // + (i*rd+j)*HW
  const long long obase = plane * h2 * w2;
  const int P2 = h2 * w2;
  for (int e = lane; e < P2; e += 64) {
    const int y1 = e / w2, x1 = e - y1 * w2;
    const int i = x1 - ix, j = y1 - iy;   // tap indices in [0, rd]
    V g = 0;
    if (i >= 0 && i <= rd && j >= 0 && j <= rd) {
      if (i > 0 && j > 0)   g = G::step(g, G::load(corr_grad, gbase + static_cast<long long>((i - 1) * rd + (j - 1)) * HW), w11);
      if (i > 0 && j < rd)  g = G::step(g, G::load(corr_grad, gbase + static_cast<long long>((i - 1) * rd + j) * HW), w10);
      if (i < rd && j > 0)  g = G::step(g, G::load(corr_grad, gbase + static_cast<long long>(i * rd + (j - 1)) * HW), w01);
      if (i < rd && j < rd) g = G::step(g, G::load(corr_grad, gbase + static_cast<long long>(i * rd + j) * HW), w00);
    }
    G::store(vol_grad, obase + e, g);
  }
}

template <typename T>
int launch_lookup(const LookupArgs& a, int radius, hipStream_t st) {
  if (a.N == 0 || a.HW == 0) return PVO_OK;
  if (radius < 0) {  // forced generic path, radius encoded as -(r+1)
    radius = -radius - 1;
    dim3 grid((a.HW + 255) / 256, a.N);
    hipLaunchKernelGGL(corr_lookup_generic_kernel<T>, grid, dim3(256), 0, st, a, radius);
  } else if (radius == 3) {
    const size_t lds = static_cast<size_t>(max(a.nlev * 49 * kStripPad, kStrip * (a.nlev * 49 + kPixPad))) *
                       sizeof(typename Elem<T>::store_t);
    dim3 grid((a.HW + kStrip - 1) / kStrip, a.N);
    if (a.lv[0].plane_elems != 0) {
      if constexpr (sizeof(typename Elem<T>::store_t) == 2) {
        if (a.enc_w) {
          const size_t lds_enc = static_cast<size_t>(kEncStrip) * kEncStride * 2;   // 14.8 KB; the output slab reuses it
          const dim3 grid_enc((a.HW + kEncStrip - 1) / kEncStrip, a.N);
          hipLaunchKernelGGL(corr_lookup_r3_enc_kernel<T>, grid_enc, dim3(256), lds_enc, st, a);
        } else {
          hipLaunchKernelGGL((corr_lookup_r3_kernel<T, true, false>), grid, dim3(256), lds, st, a);
        }
      } else {
        return PVO_EUNSUPPORTED;
      }
    } else {
      hipLaunchKernelGGL((corr_lookup_r3_kernel<T, false, false>), grid, dim3(256), lds, st, a);
    }
  } else {
    dim3 grid((a.HW + 255) / 256, a.N);
    hipLaunchKernelGGL(corr_lookup_generic_kernel<T>, grid, dim3(256), 0, st, a, radius);
  }
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

int launch_lookup_f64(const LookupArgs& a, int radius, hipStream_t st) {
  if (a.N == 0 || a.HW == 0) return PVO_OK;
  if (radius < 0) radius = -radius - 1;
  dim3 grid((a.HW + 255) / 256, a.N);
  hipLaunchKernelGGL(corr_lookup_generic_kernel<double>, grid, dim3(256), 0, st, a, radius);
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

int dispatch_lookup(const LookupArgs& a, int radius, int dtype, hipStream_t st) {
  switch (dtype) {
    case PVO_F32: return launch_lookup<float>(a, radius, st);
    case PVO_F16: return launch_lookup<pvo_half>(a, radius, st);
    case PVO_BF16: return launch_lookup<pvo_bf16>(a, radius, st);
    case PVO_F64: return launch_lookup_f64(a, radius, st);
    default: return PVO_EINVAL;
  }
}

}  // namespace

#ifdef PVO_LOOKUP_PROBE
extern "C" int pvo_debug_lookup_probe(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_lk_probe), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int pvo_corr_index_forward(const void* volume, const float* coords, void* corr,
                                      int N, int h1, int w1, int h2, int w2,
                                      int radius, int dtype, void* stream) {
  if (N < 0 || h1 < 0 || w1 < 0 || h2 < 0 || w2 < 0 || radius < 0) return PVO_EINVAL;
  if (N == 0 || h1 == 0 || w1 == 0) return PVO_OK;
  if (!volume && h2 > 0 && w2 > 0) return PVO_EINVAL;
  if (!coords || !corr) return PVO_EINVAL;
  if (N > 65535) return PVO_EUNSUPPORTED;
  LookupArgs a{};
  a.lv[0] = LookupLevel{volume, static_cast<long long>(N) * h1 * w1 * h2 * w2, h2, w2, 1.0f};
  a.coords = coords; a.out = corr; a.nlev = 1; a.coords_interleaved = 0; a.HW = h1 * w1; a.N = N;
  // the 16-bit fast path needs a 4-byte aligned base; odd views use the generic kernel
  const bool force_generic = (dtype == PVO_F16 || dtype == PVO_BF16) && (reinterpret_cast<uintptr_t>(volume) & 3);
  return dispatch_lookup(a, force_generic ? -radius - 1 : radius, dtype, pvo_stream(stream));
}

extern "C" int pvo_corr_pyramid_lookup(const void* const* volumes_host, const float* coords, void* out,
                                       int N, int h1, int w1, int h2, int w2,
                                       int num_levels, int radius, int dtype, int out_channels_last,
                                       const int* slots, int num_slots, void* stream) {
  if (N < 0 || h1 < 0 || w1 < 0 || h2 < 0 || w2 < 0 || radius < 0) return PVO_EINVAL;
  if (num_levels < 1 || num_levels > kMaxLevels || !volumes_host) return PVO_EINVAL;
  if (N == 0 || h1 == 0 || w1 == 0) return PVO_OK;
  if (!coords || !out) return PVO_EINVAL;
  if (N > 65535) return PVO_EUNSUPPORTED;
  LookupArgs a{};
  bool aligned = true;
  for (int l = 0; l < num_levels; ++l) {
    const int hl = h2 >> l, wl = w2 >> l;
    if (!volumes_host[l] && hl > 0 && wl > 0) return PVO_EINVAL;
    a.lv[l] = LookupLevel{volumes_host[l], static_cast<long long>(slots ? num_slots : N) * h1 * w1 * hl * wl, hl, wl,
                          1.0f / static_cast<float>(1 << l)};
    aligned = aligned && ((reinterpret_cast<uintptr_t>(volumes_host[l]) & 3) == 0);
  }
  a.coords = coords; a.out = out; a.nlev = num_levels; a.coords_interleaved = 1; a.HW = h1 * w1; a.N = N;
  a.out_channels_last = out_channels_last ? 1 : 0;
  a.slots = slots;
  if (slots && num_slots <= 0) return PVO_EINVAL;
  if (!aligned && (dtype == PVO_F16 || dtype == PVO_BF16)) return PVO_EINVAL;
  return dispatch_lookup(a, radius, dtype, pvo_stream(stream));
}

extern "C" int pvo_corr_pyramid_lookup_tiled(const void* const* volumes_host, const float* coords, void* out,
                                             int N, int h1, int w1, int h2, int w2,
                                             int num_levels, int dtype, int out_channels_last,
                                             const int* slots, int num_slots, void* stream) {
  if (N < 0 || h1 < 0 || w1 < 0 || h2 < 0 || w2 < 0) return PVO_EINVAL;
  if (num_levels < 1 || num_levels > kMaxLevels || !volumes_host) return PVO_EINVAL;
  if (dtype != PVO_F16 && dtype != PVO_BF16) return PVO_EUNSUPPORTED;
  if (N == 0 || h1 == 0 || w1 == 0) return PVO_OK;
  if (!coords || !out) return PVO_EINVAL;
  if (N > 65535) return PVO_EUNSUPPORTED;
  if (slots && num_slots <= 0) return PVO_EINVAL;
  LookupArgs a{};
  for (int l = 0; l < num_levels; ++l) {
    const int hl = h2 >> l, wl = w2 >> l;
    if (hl <= 0 || wl <= 0) return PVO_EUNSUPPORTED;
    if (!volumes_host[l] || (reinterpret_cast<uintptr_t>(volumes_host[l]) & 15)) return PVO_EINVAL;
    const int th = (hl + 7) >> 3, tw = (wl + 7) >> 3;
    const long long pe = static_cast<long long>(th) * tw * 64;
    a.lv[l] = LookupLevel{volumes_host[l], static_cast<long long>(slots ? num_slots : N) * h1 * w1 * pe, hl, wl,
                          1.0f / static_cast<float>(1 << l), tw, pe};
  }
  a.coords = coords; a.out = out; a.nlev = num_levels; a.coords_interleaved = 1; a.HW = h1 * w1; a.N = N;
  a.out_channels_last = out_channels_last ? 1 : 0;
  a.slots = slots;
  return dispatch_lookup(a, 3, dtype, pvo_stream(stream));
}

extern "C" int pvo_corr_lookup_encode_tiled(const void* const* volumes_host, const float* coords,
                                            const void* enc_weight, const float* enc_bias, void* out,
                                            int N, int h1, int w1, int dtype,
                                            const int* slots, int num_slots, void* stream) {
  if (N < 0 || h1 < 0 || w1 < 0 || !volumes_host) return PVO_EINVAL;
  if (dtype != PVO_F16 && dtype != PVO_BF16) return PVO_EUNSUPPORTED;
  if (N == 0 || h1 == 0 || w1 == 0) return PVO_OK;
  if (!coords || !out || !enc_weight || !enc_bias) return PVO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(enc_weight) | reinterpret_cast<uintptr_t>(out)) & 15) return PVO_EINVAL;
  if (N > 65535) return PVO_EUNSUPPORTED;
  if (slots && num_slots <= 0) return PVO_EINVAL;
  LookupArgs a{};
  for (int l = 0; l < kMaxLevels; ++l) {
    const int hl = h1 >> l, wl = w1 >> l;
    if (hl <= 0 || wl <= 0) return PVO_EUNSUPPORTED;
    if (!volumes_host[l] || (reinterpret_cast<uintptr_t>(volumes_host[l]) & 15)) return PVO_EINVAL;
    const int th = (hl + 7) >> 3, tw = (wl + 7) >> 3;
    const long long pe = static_cast<long long>(th) * tw * 64;
    a.lv[l] = LookupLevel{volumes_host[l], static_cast<long long>(slots ? num_slots : N) * h1 * w1 * pe, hl, wl,
                          1.0f / static_cast<float>(1 << l), tw, pe};
  }
  a.coords = coords; a.out = out; a.nlev = kMaxLevels; a.coords_interleaved = 1; a.HW = h1 * w1; a.N = N;
  a.out_channels_last = 1;
  a.slots = slots;
  a.enc_w = enc_weight; a.enc_b = enc_bias;
  return dispatch_lookup(a, 3, dtype, pvo_stream(stream));
}

extern "C" int pvo_corr_index_backward(const float* coords, const void* corr_grad, void* volume_grad,
                                       int N, int h1, int w1, int h2, int w2,
                                       int radius, int dtype, void* stream) {
  if (N < 0 || h1 < 0 || w1 < 0 || h2 < 0 || w2 < 0 || radius < 0) return PVO_EINVAL;
  const long long planes = static_cast<long long>(N) * h1 * w1;
  if (planes == 0 || h2 == 0 || w2 == 0) return PVO_OK;
  if (!coords || !corr_grad || !volume_grad) return PVO_EINVAL;
  const long long nblk = (planes + 3) / 4;
  if (nblk > 0x7fffffffLL) return PVO_EUNSUPPORTED;
  hipStream_t st = pvo_stream(stream);
  dim3 grid(static_cast<unsigned>(nblk)), block(256);
  switch (dtype) {
    case PVO_F32: hipLaunchKernelGGL(corr_lookup_backward_kernel<float>, grid, block, 0, st, coords, corr_grad, volume_grad, N, h1 * w1, h2, w2, radius); break;
    case PVO_F16: hipLaunchKernelGGL(corr_lookup_backward_kernel<pvo_half>, grid, block, 0, st, coords, corr_grad, volume_grad, N, h1 * w1, h2, w2, radius); break;
    case PVO_BF16: hipLaunchKernelGGL(corr_lookup_backward_kernel<pvo_bf16>, grid, block, 0, st, coords, corr_grad, volume_grad, N, h1 * w1, h2, w2, radius); break;
    case PVO_F64: hipLaunchKernelGGL(corr_lookup_backward_kernel<double>, grid, block, 0, st, coords, corr_grad, volume_grad, N, h1 * w1, h2, w2, radius); break;
    default: return PVO_EINVAL;
  }
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}
