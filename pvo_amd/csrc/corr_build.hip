// corr_build.hip — all-pairs correlation volume + its 4-level pyramid in one launch.
//
// Replaces CorrBlock.corr (torch.matmul on fmap/4) and the three F.avg_pool2d passes of
// CorrBlock.__init__ (reference VO_Module/droid_slam/modules/corr.py:24-38, 63-71).
// The reference writes level 0 with cuBLAS and then re-reads every level to pool it;
// here each output tile is produced once on the matrix cores and all levels are written
// from the accumulators: 1.33x the volume in HBM writes, no re-reads.
//
// 16-bit path (fp16 / bf16, C in {16,32,64,128}), channels-last features [N,H,W,C]:
//   workgroup = 128 source pixels (p1)  x  one 8x32 PATCH of target pixels (p2).
//   The patch is 8-aligned in both axes, so every 2x2 / 4x4 / 8x8 pooling window is
//   complete inside a workgroup.  fmap2's patch rows are staged once in LDS (row stride
//   padded by 16 B against bank conflicts) and shared by 4 waves; each wave owns 32 source
//   pixels, loads its fmap1 fragments straight from global (8 x 16 B per lane), and runs
//   8 (k) x 8 (patch rows) v_mfma_f32_32x32x16_{f16,bf16}.  GEMM column n maps to patch
//   pixel (y = n/32, x = n%32): horizontal pooling partners are adjacent lanes (DPP quad
//   permutes), vertical partners are adjacent accumulator tiles of the same lane.
//   level 0 = round16(acc / 16); level l+1 = round16(mean of the 4 ROUNDED level-l values),
//   exactly what pooling the stored fp16 tensor gives (floor sizes: partial windows at the
//   right / bottom border are dropped, as avg_pool2d does).
// generic path (fp32 / fp64 / odd C, NCHW or NHWC): LDS-tiled FMA GEMM + one pooling launch
//   per level.  Used by the fp32 training configuration and as the layout-agnostic fallback.
#include "common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef __bf16 v8b __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kPatchH = 8, kPatchW = 32, kPatchPix = kPatchH * kPatchW;  // 256 target pixels
constexpr int kTileM = 128;                                              // source pixels per workgroup
constexpr int kMaxC = 128;

struct BuildArgs {
  const void* f1; const void* f2;   // [N,H,W,C]
  void* lv[4];                      // level l: [N, HW, H>>l, W>>l]
  int N, C, H, W, nlev;
  const int* out_slots;             // optional: edge n is written to slot out_slots[n] of the level tensors
  int tiled;                        // level l stored [N, HW, ceil(Hl/8), ceil(Wl/8), 8, 8] (see corr_lookup.hip)
};

template <typename T> struct Cvt16;
template <> struct Cvt16<pvo_half> {
  static __device__ __forceinline__ uint32_t bits(float x) {
    union { _Float16 h; uint16_t u; } c; c.h = static_cast<_Float16>(x); return c.u;
  }
  static __device__ __forceinline__ float val(uint32_t b) {
    union { _Float16 h; uint16_t u; } c; c.u = static_cast<uint16_t>(b); return static_cast<float>(c.h);
  }
  static __device__ __forceinline__ v16f mfma(u32x4 a, u32x4 b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
  }
};
template <> struct Cvt16<pvo_bf16> {
  static __device__ __forceinline__ uint32_t bits(float x) { return pvo_f32_to_bf16(x); }
  static __device__ __forceinline__ float val(uint32_t b) { return pvo_bf16_to_f32(static_cast<uint16_t>(b)); }
  static __device__ __forceinline__ v16f mfma(u32x4 a, u32x4 b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8b, a), __builtin_bit_cast(v8b, b), c, 0, 0, 0);
  }
};

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// pooled = round16(((a + b) + c) + d) / 4), ATen avg_pool2d window order (kh, kw)
template <typename T>
__device__ __forceinline__ float pool4(float a, float b, float c, float d) {
  float s = ((a + b) + c) + d;
  s *= 0.25f;
  asm volatile("" : "+v"(s));           // keep the fp32 rounding before the 16-bit one (no mixlo fusion)
  return Cvt16<T>::val(Cvt16<T>::bits(s));
}

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void corr_build_mfma_kernel(BuildArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int C = a.C, H = a.H, W = a.W, HW = H * W;
  const int rowB = C * 2 + 16;                       // padded LDS row stride (bytes)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.z;
  const int npw = (W + kPatchW - 1) / kPatchW;
  const int py = blockIdx.y / npw, px = blockIdx.y - py * npw;
  const int y2_0 = py * kPatchH, x2_0 = px * kPatchW;
  const int m0 = blockIdx.x * kTileM + wave * 32;

  // ---- stage the fmap2 patch: 256 pixel rows of C halves
  const uint16_t* f2 = reinterpret_cast<const uint16_t*>(a.f2) + static_cast<long long>(n) * HW * C;
  // C is a power of two here (16..128): cpr = C/8 chunks of 16 B per pixel row.  All loads of a
  // thread are issued before the first LDS store (16 independent 16-byte loads in flight);
  // a load -> store -> load loop serialises 16 memory round trips per workgroup.
  const int cpr = C >> 3, lcpr = 31 - __builtin_clz(cpr);
  {
    const int ch = tid & (cpr - 1);
    const int q0 = tid >> lcpr, qstep = 256 >> lcpr;   // pixel of iteration it: q0 + it * qstep
    u32x4 stg[kMaxC / 8];
#pragma unroll
    for (int it = 0; it < kMaxC / 8; ++it) {
      stg[it] = u32x4{0u, 0u, 0u, 0u};
      if (it < cpr) {
        const int q = q0 + it * qstep;
        const int y2 = y2_0 + (q >> 5), x2 = x2_0 + (q & 31);
        if (y2 < H && x2 < W) stg[it] = *reinterpret_cast<const u32x4*>(f2 + (static_cast<long long>(y2) * W + x2) * C + ch * 8);
      }
    }
#pragma unroll
    for (int it = 0; it < kMaxC / 8; ++it)
      if (it < cpr) *reinterpret_cast<u32x4*>(smem + (q0 + it * qstep) * rowB + ch * 16) = stg[it];
  }

  // ---- this lane's fmap1 fragments (row i = lane&31 of the wave's 32 source pixels)
  const int kg = lane >> 5;
  const int p1_frag = m0 + (lane & 31);
  const uint16_t* f1row = reinterpret_cast<const uint16_t*>(a.f1) + (static_cast<long long>(n) * HW + min(p1_frag, HW - 1)) * C;
  u32x4 afrag[kMaxC / 16];
  const int ksteps = C >> 4;
#pragma unroll
  for (int s = 0; s < kMaxC / 16; ++s) {
    afrag[s] = u32x4{0u, 0u, 0u, 0u};
    if (s < ksteps && p1_frag < HW) afrag[s] = *reinterpret_cast<const u32x4*>(f1row + s * 16 + kg * 8);
  }
  __syncthreads();

  v16f acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

#pragma unroll
  for (int s = 0; s < kMaxC / 16; ++s) {
    if (s < ksteps) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const u32x4 b = *reinterpret_cast<const u32x4*>(smem + (t * 32 + (lane & 31)) * rowB + (s * 16 + kg * 8) * 2);
        acc[t] = Cvt16<T>::mfma(afrag[s], b, acc[t]);
      }
    }
  }

  // ---- epilogue: every level is formed in registers, transposed through a per-wave LDS
  // slab and leaves as 16-byte row segments (a 2-byte store per value is store-issue bound).
  // Two passes of 16 source pixels each through a HALF-size slab: with slabs for all 32 rows (94 KB) the kernel ran one
  // workgroup per CU; 47 KB (< the 70 KB fmap2 patch) lets two reside, and one can compute while the other drains its stores.
  // The slabs are per wave: between fill and drain only the wave itself has to be ordered (no workgroup barrier).
  __syncthreads();                                   // the fmap2 patch is dead: its LDS is reused
  constexpr int RS0 = kPatchPix * 2 + 16, RS1 = 64 * 2 + 16, RS2 = 16 * 2 + 16, RS3 = 16;
  constexpr int kHalfRows = 16;
  unsigned char* slab0 = smem + wave * (kHalfRows * RS0);
  unsigned char* slab1 = smem + 4 * (kHalfRows * RS0) + wave * (kHalfRows * RS1);
  unsigned char* slab2 = smem + 4 * (kHalfRows * (RS0 + RS1)) + wave * (kHalfRows * RS2);
  unsigned char* slab3 = smem + 4 * (kHalfRows * (RS0 + RS1 + RS2)) + wave * (kHalfRows * RS3);
  const int x2l = lane & 31;
  const long long plane00 = static_cast<long long>(a.out_slots ? a.out_slots[n] : n) * HW + m0;   // plane of this wave's row 0
  const int rows_all = min(32, HW - m0);                           // valid source pixels of this wave
  // aligned: every level's row segments are 16-byte (level 3: 8-byte) aligned and complete
  const bool aligned = ((W & 63) == 0) && ((H & 7) == 0) &&
      (((reinterpret_cast<uintptr_t>(a.lv[0]) | reinterpret_cast<uintptr_t>(a.lv[1]) |
         reinterpret_cast<uintptr_t>(a.lv[2]) | reinterpret_cast<uintptr_t>(a.lv[3])) & 15) == 0 || a.nlev < 4);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int r = 8 * half + rr;
    const int mrow = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);       // row inside this half: (r & 3) + 8 (r >> 2) + 4 (lane >> 5) - 16 half
    float v0[8];                                     // rounded level-0 values, one per patch row
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float sc = acc[t][r] * 0.0625f;                // (f1/4).(f2/4)
      asm volatile("" : "+v"(sc));
      const uint32_t bb = Cvt16<T>::bits(sc);
      v0[t] = Cvt16<T>::val(bb);
      *reinterpret_cast<uint16_t*>(slab0 + mrow * RS0 + (t * 32 + x2l) * 2) = static_cast<uint16_t>(bb);
    }
    // level 1: patch rows (2q, 2q+1), lanes (x, x^1); level 2: rows (2q,2q+1) of level 1, lanes (x, x^2);
    // level 3: the two level-2 rows, lanes (x, x^4).  Lanes that do not own a result compute don't-cares.
    float v1[4], v2[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v1[q] = pool4<T>(v0[2 * q], dpp_f<0xB1>(v0[2 * q]), v0[2 * q + 1], dpp_f<0xB1>(v0[2 * q + 1]));
      if ((lane & 1) == 0)
        *reinterpret_cast<uint16_t*>(slab1 + mrow * RS1 + (q * 16 + (x2l >> 1)) * 2) = static_cast<uint16_t>(Cvt16<T>::bits(v1[q]));
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      v2[q] = pool4<T>(v1[2 * q], dpp_f<0x4E>(v1[2 * q]), v1[2 * q + 1], dpp_f<0x4E>(v1[2 * q + 1]));
      if ((lane & 3) == 0)
        *reinterpret_cast<uint16_t*>(slab2 + mrow * RS2 + (q * 8 + (x2l >> 2)) * 2) = static_cast<uint16_t>(Cvt16<T>::bits(v2[q]));
    }
    {
      const float n0 = __shfl_xor(v2[0], 4, 64), n1 = __shfl_xor(v2[1], 4, 64);
      const float v3 = pool4<T>(v2[0], n0, v2[1], n1);
      if ((lane & 7) == 0)
        *reinterpret_cast<uint16_t*>(slab3 + mrow * RS3 + (x2l >> 3) * 2) = static_cast<uint16_t>(Cvt16<T>::bits(v3));
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const long long plane0 = plane00 + kHalfRows * half;             // plane of this half's row 0
  const int rows_ok = min(kHalfRows, rows_all - kHalfRows * half);   // valid source pixels of this half
  if (rows_ok > 0) {
  if (a.nlev == 4 && a.tiled) {
    // 8x8-tiled planes, any map size: level l is [ceil(Hl/8)][ceil(Wl/8)] tiles of 8x8 (Hl = H >> l).  The patch is one
    // tile row high and 8-aligned, so its level-0 part is 4 whole tiles = 512 contiguous bytes per source pixel, and the
    // coarser levels fill half / quarter tiles.  Whole tile rows are written wherever the TILE exists; elements of a tile
    // beyond (Hl, Wl) - zero-padded target pixels, partial pooling windows - are padding no reader treats as valid.
    const int tw0 = (W + 7) >> 3, tw1 = ((W >> 1) + 7) >> 3, tw2 = ((W >> 2) + 7) >> 3, tw3 = ((W >> 3) + 7) >> 3;
    const int th1 = ((H >> 1) + 7) >> 3, th2 = ((H >> 2) + 7) >> 3, th3 = ((H >> 3) + 7) >> 3;
    const long long pe0 = static_cast<long long>((H + 7) >> 3) * tw0 * 64, pe1 = static_cast<long long>(th1) * tw1 * 64,
                    pe2 = static_cast<long long>(th2) * tw2 * 64, pe3 = static_cast<long long>(th3) * tw3 * 64;
    uint16_t* L0 = reinterpret_cast<uint16_t*>(a.lv[0]);
#pragma unroll 4
    for (int i = 0; i < 8; ++i) {                    // level 0: 16 rows x 4 tiles x 8 tile rows
      const int id = lane + 64 * i;
      const int mrow = id >> 5, c = (id >> 3) & 3, t = id & 7;
      if (mrow < rows_ok && (x2_0 >> 3) + c < tw0) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(slab0 + mrow * RS0 + (t * 32 + c * 8) * 2);
        *reinterpret_cast<u32x4*>(L0 + (plane0 + mrow) * pe0 + (static_cast<long long>(y2_0 >> 3) * tw0 + (x2_0 >> 3) + c) * 64 + t * 8) = v;
      }
    }
    uint16_t* L1 = reinterpret_cast<uint16_t*>(a.lv[1]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {                    // level 1: 16 rows x 2 tiles x 4 tile rows
      const int id = lane + 64 * i;
      const int mrow = id >> 3, c = (id >> 2) & 1, q = id & 3;
      if (mrow < rows_ok && (x2_0 >> 4) + c < tw1 && (y2_0 >> 4) < th1) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(slab1 + mrow * RS1 + (q * 16 + c * 8) * 2);
        *reinterpret_cast<u32x4*>(L1 + (plane0 + mrow) * pe1 + (static_cast<long long>(y2_0 >> 4) * tw1 + (x2_0 >> 4) + c) * 64 +
                                  (((y2_0 >> 1) & 7) + q) * 8) = v;
      }
    }
    uint16_t* L2 = reinterpret_cast<uint16_t*>(a.lv[2]);
    if (lane < 32) {                                 // level 2: 16 rows x 2 tile rows of one tile
      const int mrow = lane >> 1, q = lane & 1;
      if (mrow < rows_ok && (x2_0 >> 5) < tw2 && (y2_0 >> 5) < th2) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(slab2 + mrow * RS2 + q * 16);
        *reinterpret_cast<u32x4*>(L2 + (plane0 + mrow) * pe2 + (static_cast<long long>(y2_0 >> 5) * tw2 + (x2_0 >> 5)) * 64 +
                                  (((y2_0 >> 2) & 7) + q) * 8) = v;
      }
    }
    uint16_t* L3 = reinterpret_cast<uint16_t*>(a.lv[3]);
    if (lane < kHalfRows && lane < rows_ok && (x2_0 >> 6) < tw3 && (y2_0 >> 6) < th3) {      // level 3: 4 values = half a tile row
      const uint2 v = *reinterpret_cast<const uint2*>(slab3 + lane * RS3);
      *reinterpret_cast<uint2*>(L3 + (plane0 + lane) * pe3 + (static_cast<long long>(y2_0 >> 6) * tw3 + (x2_0 >> 6)) * 64 +
                                ((y2_0 >> 3) & 7) * 8 + ((x2_0 >> 3) & 7)) = v;
    }
  } else if (aligned && a.nlev == 4) {
    uint16_t* L0 = reinterpret_cast<uint16_t*>(a.lv[0]);
#pragma unroll 4
    for (int i = 0; i < 8; ++i) {                    // level 0: 16 rows x 8 patch rows x 4 chunks of 8
      const int id = lane + 64 * i;
      const int mrow = id >> 5, t = (id >> 2) & 7, c = id & 3;
      if (mrow < rows_ok) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(slab0 + mrow * RS0 + (t * 32 + c * 8) * 2);
        *reinterpret_cast<u32x4*>(L0 + ((plane0 + mrow) * H + (y2_0 + t)) * W + x2_0 + c * 8) = v;
      }
    }
    {
      uint16_t* L1 = reinterpret_cast<uint16_t*>(a.lv[1]);
      const int H1 = H >> 1, W1 = W >> 1;
#pragma unroll
      for (int i = 0; i < 2; ++i) {                  // level 1: 16 rows x 4 rows x 2 chunks of 8
        const int id = lane + 64 * i;
        const int mrow = id >> 3, q = (id >> 1) & 3, c = id & 1;
        if (mrow < rows_ok) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(slab1 + mrow * RS1 + (q * 16 + c * 8) * 2);
          *reinterpret_cast<u32x4*>(L1 + ((plane0 + mrow) * H1 + ((y2_0 >> 1) + q)) * W1 + (x2_0 >> 1) + c * 8) = v;
        }
      }
      uint16_t* L2 = reinterpret_cast<uint16_t*>(a.lv[2]);
      const int H2 = H >> 2, W2 = W >> 2;
      if (lane < 32) {                               // level 2: 16 rows x 2 rows x 1 chunk of 8
        const int mrow = lane >> 1, q = lane & 1;
        if (mrow < rows_ok) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(slab2 + mrow * RS2 + q * 16);
          *reinterpret_cast<u32x4*>(L2 + ((plane0 + mrow) * H2 + ((y2_0 >> 2) + q)) * W2 + (x2_0 >> 2)) = v;
        }
      }
      uint16_t* L3 = reinterpret_cast<uint16_t*>(a.lv[3]);
      const int H3 = H >> 3, W3 = W >> 3;
      if (lane < kHalfRows && lane < rows_ok) {      // level 3: 16 rows x 4 values (8 bytes)
        const uint2 v = *reinterpret_cast<const uint2*>(slab3 + lane * RS3);
        *reinterpret_cast<uint2*>(L3 + ((plane0 + lane) * H3 + (y2_0 >> 3)) * W3 + (x2_0 >> 3)) = v;
      }
    }
  } else {
    // ragged shapes: element stores, consecutive lanes along x
    const unsigned char* slabs[4] = {slab0, slab1, slab2, slab3};
    const int RS[4] = {RS0, RS1, RS2, RS3};
    for (int l = 0; l < a.nlev; ++l) {
      uint16_t* L = reinterpret_cast<uint16_t*>(a.lv[l]);
      const int Hl = H >> l, Wl = W >> l;
      const int ph = kPatchH >> l, pw = kPatchW >> l;          // patch extent at this level
      const int y0 = y2_0 >> l, x0 = x2_0 >> l;
      for (int id = lane; id < kHalfRows * ph * pw; id += 64) {
        const int mrow = id / (ph * pw);
        const int rem = id - mrow * (ph * pw);
        const int yy = rem / pw, xx = rem - yy * pw;
        if (mrow < rows_ok && y0 + yy < Hl && x0 + xx < Wl)
          L[((plane0 + mrow) * Hl + (y0 + yy)) * Wl + x0 + xx] =
              *reinterpret_cast<const uint16_t*>(slabs[l] + mrow * RS[l] + (yy * pw + xx) * 2);
      }
    }
  }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // the slab is free again once this wave has read it
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ---------------------------------------------------------------------------
// generic path
// ---------------------------------------------------------------------------
template <typename T> struct GVal {   // 16/32-bit: float domain
  using acc_t = float;
  static __device__ __forceinline__ float ld(const void* p, long long i) { return Elem<T>::to_f32(reinterpret_cast<const typename Elem<T>::store_t*>(p)[i]); }
  static __device__ __forceinline__ void st(void* p, long long i, float v) { reinterpret_cast<typename Elem<T>::store_t*>(p)[i] = Elem<T>::from_f32(v); }
  static __device__ __forceinline__ float rnd(float v) { return Elem<T>::to_f32(Elem<T>::from_f32(v)); }
};
template <> struct GVal<double> {
  using acc_t = double;
  static __device__ __forceinline__ double ld(const void* p, long long i) { return reinterpret_cast<const double*>(p)[i]; }
  static __device__ __forceinline__ void st(void* p, long long i, double v) { reinterpret_cast<double*>(p)[i] = v; }
  static __device__ __forceinline__ double rnd(double v) { return v; }
};

// level0[n,p1,p2] = sum_c (f1/4)(f2/4); 64x64 output tile, 16-deep k slices in LDS, 4x4 per thread.
// sc/sp: element strides of the channel / pixel axes (NCHW: HW,1   NHWC: 1,C).
template <typename T>
__global__ __launch_bounds__(256) void corr_gemm_generic_kernel(const void* f1, const void* f2, void* out,
                                                                int C, int HW, long long sc, long long sp) {
  using A = typename GVal<T>::acc_t;
  __shared__ A sa[16][64 + 1];
  __shared__ A sb[16][64 + 1];
  const int n = blockIdx.z;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const long long fbase = static_cast<long long>(n) * C * HW;
  A acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0;
  for (int k0 = 0; k0 < C; k0 += 16) {
    for (int id = threadIdx.x; id < 16 * 64; id += 256) {
      const int kk = id >> 6, pp = id & 63;
      const int c = k0 + kk;
      A va = 0, vb = 0;
      if (c < C && m0 + pp < HW) va = GVal<T>::rnd(GVal<T>::ld(f1, fbase + c * sc + (m0 + pp) * sp) / static_cast<A>(4));
      if (c < C && n0 + pp < HW) vb = GVal<T>::rnd(GVal<T>::ld(f2, fbase + c * sc + (n0 + pp) * sp) / static_cast<A>(4));
      sa[kk][pp] = va; sb[kk][pp] = vb;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      A av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = sa[kk][ty * 4 + i]; bv[i] = sb[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fma(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p1 = m0 + ty * 4 + i, p2 = n0 + tx * 4 + j;
      if (p1 < HW && p2 < HW) GVal<T>::st(out, (static_cast<long long>(n) * HW + p1) * HW + p2, acc[i][j]);
    }
}

// one pooling level: out[pl, y, x] = mean of the 2x2 window of in[pl]; planes = N*HW
template <typename T>
__global__ __launch_bounds__(256) void corr_pool_generic_kernel(const void* in, void* out, long long planes, int h, int w) {
  using A = typename GVal<T>::acc_t;
  const int h2 = h >> 1, w2 = w >> 1;
  const long long total = planes * h2 * w2;
  for (long long id = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; id < total; id += static_cast<long long>(gridDim.x) * 256) {
    const long long pl = id / (h2 * w2);
    const int rem = static_cast<int>(id - pl * (h2 * w2));
    const int y = rem / w2, x = rem - y * w2;
    const long long s = pl * h * w + static_cast<long long>(2 * y) * w + 2 * x;
    const A v = ((GVal<T>::ld(in, s) + GVal<T>::ld(in, s + 1)) + GVal<T>::ld(in, s + w)) + GVal<T>::ld(in, s + w + 1);
    GVal<T>::st(out, id, v * static_cast<A>(0.25));
  }
}

template <typename T>
int build_generic(const void* f1, const void* f2, void* const* lv, int N, int C, int H, int W, int nlev,
                  int channels_last, hipStream_t st) {
  const int HW = H * W;
  const long long sc = channels_last ? 1 : HW, sp = channels_last ? C : 1;
  dim3 grid((HW + 63) / 64, (HW + 63) / 64, N);
  hipLaunchKernelGGL(corr_gemm_generic_kernel<T>, grid, dim3(256), 0, st, f1, f2, lv[0], C, HW, sc, sp);
  PVO_CHECK_LAUNCH();
  int h = H, w = W;
  for (int l = 1; l < nlev; ++l) {
    const long long total = static_cast<long long>(N) * HW * (h >> 1) * (w >> 1);
    if (total > 0) {
      const unsigned blocks = static_cast<unsigned>(total / 256 + 1 < 65535 * 16 ? total / 256 + 1 : 65535 * 16);
      hipLaunchKernelGGL(corr_pool_generic_kernel<T>, dim3(blocks), dim3(256), 0, st, lv[l - 1], lv[l],
                         static_cast<long long>(N) * HW, h, w);
      PVO_CHECK_LAUNCH();
    }
    h >>= 1; w >>= 1;
  }
  return PVO_OK;
}

template <typename T>
int build_mfma(const BuildArgs& a, hipStream_t st) {
  const int HW = a.H * a.W;
  const size_t lds_in = static_cast<size_t>(kPatchPix) * (a.C * 2 + 16);
  const size_t lds_out = 4 * 16 * static_cast<size_t>((kPatchPix * 2 + 16) + (64 * 2 + 16) + (16 * 2 + 16) + 16);      // half-size slabs: two passes
  const size_t lds = lds_in > lds_out ? lds_in : lds_out;
  if (lds > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(corr_build_mfma_kernel<T>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) return PVO_ELAUNCH;
  }
  const int npatch = ((a.H + kPatchH - 1) / kPatchH) * ((a.W + kPatchW - 1) / kPatchW);
  dim3 grid((HW + kTileM - 1) / kTileM, npatch, a.N);
  hipLaunchKernelGGL(corr_build_mfma_kernel<T>, grid, dim3(256), lds, st, a);
  PVO_CHECK_LAUNCH();
  return PVO_OK;
}

}  // namespace

extern "C" int pvo_corr_build_tiled(const void* fmap1, const void* fmap2, void* const* levels_host,
                                    int N, int C, int H, int W, int dtype, const int* out_slots, void* stream) {
  if (N < 0 || C <= 0 || H < 0 || W < 0 || !levels_host) return PVO_EINVAL;
  if (N == 0 || H == 0 || W == 0) return PVO_OK;
  if (!fmap1 || !fmap2 || N > 65535) return PVO_EINVAL;
  uintptr_t al = reinterpret_cast<uintptr_t>(fmap1) | reinterpret_cast<uintptr_t>(fmap2);
  for (int l = 0; l < 4; ++l) {
    if (!levels_host[l]) return PVO_EINVAL;
    al |= reinterpret_cast<uintptr_t>(levels_host[l]);
  }
  // the tiled writer is the matrix-core kernel's aligned epilogue
  if ((dtype != PVO_F16 && dtype != PVO_BF16) || C < 16 || C > kMaxC || (C & (C - 1)) || W < 8 || H < 8 || (al & 15))
    return PVO_EUNSUPPORTED;
  BuildArgs a{};
  a.f1 = fmap1; a.f2 = fmap2; a.N = N; a.C = C; a.H = H; a.W = W; a.nlev = 4; a.out_slots = out_slots; a.tiled = 1;
  for (int l = 0; l < 4; ++l) a.lv[l] = levels_host[l];
  hipStream_t st = pvo_stream(stream);
  return dtype == PVO_F16 ? build_mfma<pvo_half>(a, st) : build_mfma<pvo_bf16>(a, st);
}

extern "C" int pvo_corr_build(const void* fmap1, const void* fmap2, void* const* levels_host,
                              int N, int C, int H, int W, int num_levels, int dtype, int channels_last,
                              const int* out_slots, void* stream) {
  if (N < 0 || C <= 0 || H < 0 || W < 0 || num_levels < 1 || num_levels > 4 || !levels_host) return PVO_EINVAL;
  if (N == 0 || H == 0 || W == 0) return PVO_OK;
  if (!fmap1 || !fmap2 || N > 65535) return PVO_EINVAL;
  for (int l = 0; l < num_levels; ++l)
    if (!levels_host[l] && (H >> l) > 0 && (W >> l) > 0) return PVO_EINVAL;
  hipStream_t st = pvo_stream(stream);
  const bool fast = channels_last && (dtype == PVO_F16 || dtype == PVO_BF16) && C >= 16 && C <= kMaxC && (C & (C - 1)) == 0 &&
                    ((reinterpret_cast<uintptr_t>(fmap1) | reinterpret_cast<uintptr_t>(fmap2)) & 15) == 0;
  if (fast) {
    BuildArgs a{};
    a.f1 = fmap1; a.f2 = fmap2; a.N = N; a.C = C; a.H = H; a.W = W; a.nlev = num_levels; a.out_slots = out_slots;
    for (int l = 0; l < num_levels; ++l) a.lv[l] = levels_host[l];
    return dtype == PVO_F16 ? build_mfma<pvo_half>(a, st) : build_mfma<pvo_bf16>(a, st);
  }
  if (out_slots) return PVO_EUNSUPPORTED;   // slot-pool output is a feature of the matrix-core path
  switch (dtype) {
    case PVO_F32: return build_generic<float>(fmap1, fmap2, levels_host, N, C, H, W, num_levels, channels_last, st);
    case PVO_F16: return build_generic<pvo_half>(fmap1, fmap2, levels_host, N, C, H, W, num_levels, channels_last, st);
    case PVO_BF16: return build_generic<pvo_bf16>(fmap1, fmap2, levels_host, N, C, H, W, num_levels, channels_last, st);
    case PVO_F64: return build_generic<double>(fmap1, fmap2, levels_host, N, C, H, W, num_levels, channels_last, st);
    default: return PVO_EINVAL;
  }
}
