// Stand-alone reproducer (no code of the library, no torch): packed-FP32 VALU instructions of one kernel give WRONG RESULTS while a
// kernel of ANOTHER stream that issues MFMA instructions is resident on the same SIMDs.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_probe/pkfp32_mfma_repro tools/pkfp32_mfma_repro.hip      (here)
//   tools/_probe/pkfp32_mfma_repro [iterations]                                                       (on the GPU box)
//
// Why it exists (profiles/r04_coresidency.md): the library's bundle adjustment was bit-for-bit reproducible
// alone on the device and NOT when a convolution of the library ran beside it on the side stream.  Round 4 narrowed it down inside
// the library (tools/sched_bisect.py): the kernels are correctly ordered (clock stamps), agent-scope (sc1) loads / stores of every
// buffer they exchange change nothing, a fill or a plain streaming kernel beside the BA is harmless, ANY MFMA kernel beside it - even
// three workgroups - is not, what goes wrong are single per-wave accumulators of the assembly kernel (v_pk_fma_f32 chains), and the
// same sources compiled with `-target-feature -packed-fp32-ops` (no v_pk_*_f32) never differ in 2100 runs of the arrangements
// that differed in 100 % of the runs before.  This program keeps only those two ingredients:
//   victim    (stream A): every thread runs chains of v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on ~200 VGPRs, from inputs that
//                         depend only on its index, and stores a digest;  `scalar` variants use v_fma_f32 for the same arithmetic
//   aggressor (stream B): a loop of v_mfma_f32_32x32x16_f16 on registers (no memory traffic to speak of) | a plain VALU loop
// and compares every launch of the victim with its first launch made ALONE on the device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int kAcc = 88;        // packed accumulators per thread: 176 VGPRs of state + operands (the assembly kernel of the library: 220)

template <bool PACKED>
__global__ __launch_bounds__(256) void victim(float* out, int rounds) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  f2 acc[kAcc];
#pragma unroll
  for (int k = 0; k < kAcc; ++k) acc[k] = f2{0.001f * static_cast<float>((gid * 7 + k) & 1023), 0.002f * static_cast<float>((gid * 13 + 3 * k) & 511)};
  f2 a = {1.0f + 1e-3f * static_cast<float>(gid & 63), 1.0f - 1e-3f * static_cast<float>(gid & 31)};
  f2 b = {1e-4f * static_cast<float>(gid & 255), -1e-4f * static_cast<float>(gid & 127)};
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int k = 0; k < kAcc; ++k) {
      if (PACKED) {
        acc[k] = __builtin_elementwise_fma(acc[k], a, b);                   // v_pk_fma_f32
        acc[k] = acc[k] * f2{0.999f, 1.001f} + acc[(k + 1) % kAcc] * 1e-3f;   // v_pk_mul_f32 / v_pk_fma_f32
      } else {
        // the same arithmetic, one component at a time, kept scalar by opaque copies (v_fma_f32 / v_mul_f32)
        float x = acc[k].x, y = acc[k].y;
        x = __builtin_fmaf(x, a.x, b.x); asm volatile("" : "+v"(x));
        y = __builtin_fmaf(y, a.y, b.y); asm volatile("" : "+v"(y));
        float nx = acc[(k + 1) % kAcc].x, ny = acc[(k + 1) % kAcc].y;
        x = __builtin_fmaf(nx, 1e-3f, x * 0.999f); asm volatile("" : "+v"(x));
        y = __builtin_fmaf(ny, 1e-3f, y * 1.001f); asm volatile("" : "+v"(y));
        acc[k] = f2{x, y};
      }
    }
    a = a * f2{1.00001f, 0.99999f};
  }
  // digest: every accumulator is stored (a corrupted register shows as one wrong word)
#pragma unroll
  for (int k = 0; k < kAcc; ++k) { out[(static_cast<size_t>(gid) * kAcc + k) * 2] = acc[k].x; out[(static_cast<size_t>(gid) * kAcc + k) * 2 + 1] = acc[k].y; }
}

template <bool MFMA>
__global__ __launch_bounds__(256) void aggressor(float* sink, int rounds) {
  const int lane = threadIdx.x & 63;
  if (MFMA) {
    h8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = static_cast<_Float16>(0.01f * static_cast<float>((lane + i) & 15)); b[i] = static_cast<_Float16>(0.02f * static_cast<float>((lane * 3 + i) & 7)); }
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int r = 0; r < rounds; ++r) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.678f) sink[blockIdx.x] = s;
  } else {
    float x = 1.0f + 1e-3f * static_cast<float>(lane), y = 0.5f;
    for (int r = 0; r < rounds * 16; ++r) { x = __builtin_fmaf(x, 0.9999f, y); asm volatile("" : "+v"(x)); y = __builtin_fmaf(y, 1.0001f, 1e-6f); asm volatile("" : "+v"(y)); }
    if (x == 12345.678f) sink[blockIdx.x] = x + y;
  }
}

struct Stat { long long launches, bad_launches, bad_words; };

template <bool PACKED>
static Stat run(int iters, int aggr /*0 none, 1 MFMA, 2 plain VALU*/, int prio_b, bool same_stream) {
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, prio_b));
  const int wg = 216, n = wg * 256 * kAcc * 2;          // the assembly kernel's grid: 216 workgroups, fewer than compute units
  float *out, *ref, *sink; unsigned long long* diff;
  CK(hipMalloc(&out, sizeof(float) * n)); CK(hipMalloc(&ref, sizeof(float) * n)); CK(hipMalloc(&sink, 4096 * 4)); CK(hipMalloc(&diff, 16));
  CK(hipMemset(diff, 0, 16));
  victim<PACKED><<<wg, 256, 0, a>>>(ref, 40);          // the reference: alone on the device
  CK(hipDeviceSynchronize());
  Stat s{0, 0, 0};
  std::vector<float> h_ref(n), h_out(n);
  CK(hipMemcpy(h_ref.data(), ref, sizeof(float) * n, hipMemcpyDeviceToHost));
  for (int it = 0; it < iters; ++it) {
    if (aggr) {
      hipStream_t sb = same_stream ? a : b;
      if (aggr == 1) aggressor<true><<<512, 256, 0, sb>>>(sink, 1500); else aggressor<false><<<512, 256, 0, sb>>>(sink, 1500);
    }
    victim<PACKED><<<wg, 256, 0, a>>>(out, 40);
    CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
    CK(hipMemcpy(h_out.data(), out, sizeof(float) * n, hipMemcpyDeviceToHost));
    long long bad = 0;
    for (int i = 0; i < n; ++i) bad += (reinterpret_cast<unsigned*>(h_out.data())[i] != reinterpret_cast<unsigned*>(h_ref.data())[i]);
    s.launches++; s.bad_launches += bad > 0; s.bad_words += bad;
    if (bad && s.bad_launches <= 3) {
      int shown = 0;
      for (int i = 0; i < n && shown < 6; ++i)
        if (reinterpret_cast<unsigned*>(h_out.data())[i] != reinterpret_cast<unsigned*>(h_ref.data())[i]) {
          const int gid = i / (2 * kAcc);
          printf("      launch %d: workgroup %d wave %d lane %d accumulator %d.%c = %.9g (alone: %.9g)\n", it, gid / 256, (gid & 255) >> 6, gid & 63,
                 (i / 2) % kAcc, "xy"[i & 1], h_out[i], h_ref[i]);
          ++shown;
        }
    }
  }
  CK(hipFree(out)); CK(hipFree(ref)); CK(hipFree(sink)); CK(hipFree(diff)); CK(hipStreamDestroy(a)); CK(hipStreamDestroy(b));
  return s;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 300;
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s (%s), %d CUs; %d launches of the victim per line, each compared bit for bit with its launch alone on the device\n", p.name, p.gcnArchName,
         p.multiProcessorCount, iters);
  struct Case { const char* name; bool packed; int aggr; int prio; bool same; } cases[] = {
      {"packed-FP32 victim, alone", true, 0, 0, false},
      {"packed-FP32 victim, MFMA kernel on another stream", true, 1, 0, false},
      {"packed-FP32 victim, MFMA kernel on another (high-priority) stream", true, 1, -1, false},
      {"packed-FP32 victim, plain-VALU kernel on another stream", true, 2, 0, false},
      {"packed-FP32 victim, MFMA kernel in front of it on the SAME stream", true, 1, 0, true},
      {"scalar-FP32 victim (same arithmetic, no v_pk_*), MFMA kernel on another stream", false, 1, 0, false},
      {"packed-FP32 victim, alone (again)", true, 0, 0, false},
  };
  long long bad_alone = 0, bad_pk_mfma = 0, bad_other = 0;
  for (const Case& c : cases) {
    const Stat s = c.packed ? run<true>(iters, c.aggr, c.prio, c.same) : run<false>(iters, c.aggr, c.prio, c.same);
    printf("%-84s: %lld of %lld launches differ (%lld words)\n", c.name, s.bad_launches, s.launches, s.bad_words);
    if (c.aggr == 0 || c.same) bad_alone += s.bad_launches; else if (c.packed && c.aggr == 1) bad_pk_mfma += s.bad_launches; else bad_other += s.bad_launches;
  }
  printf("verdict: %s\n", bad_alone ? "differences without a co-resident kernel: this program is broken"
         : (bad_pk_mfma && !bad_other) ? "v_pk_*_f32 results are corrupted by MFMA instructions of a co-resident kernel of another stream: platform erratum, reproduced without library code"
         : bad_pk_mfma ? "differences beside an MFMA kernel, but not only for the packed-FP32 victim: see the lines above"
                       : "no differences: this skeleton does not reproduce the library's failures");
  return 0;
}
