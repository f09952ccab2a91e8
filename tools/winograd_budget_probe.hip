// What a Winograd F(2x2, 3x3) form of the wide convolution (conv3x3_big_kernel) could reach on gfx950: the instruction budget of its
// main loop, MEASURED.  Stand-alone (no library code, no torch):
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_probe/winograd_budget_probe tools/winograd_budget_probe.hip     (here)
//   tools/_probe/winograd_budget_probe                                                                     (on the GPU box)
//
// The direct kernel's wave spends a 32-channel chunk of its 16 x 16-pixel x 128-output tile on 144 v_mfma_f32_32x32x16_f16 (9 taps x 8
// accumulator tiles x 2 k-steps; 128 accumulator registers, two workgroups per CU).  F(2x2, 3x3) needs 2.25 x fewer multiplications -
// 64 MFMAs per wave and chunk (16 transform positions x 2 output tiles x 2 k-steps on a quarter of the rows) - but the products of
// the 16 positions are 16 accumulator tiles per output tile: 512 registers per wave for the workgroup's tile, i.e. the whole
// register file.  They can only live for ONE chunk: the output transform Y = A^T (U . V) A, which is linear, has to run inside the K
// loop - per chunk and lane 896 additions fold the 512 product values into the 128 output-domain accumulators (4 + 2 .. 4 + 4 per
// value and position row), and the input transform B^T d B adds ~256 more per thread.  This program times exactly those instruction
// streams on registers only - no LDS, no global memory, no dependence of an addition on an MFMA still in flight: every simplification
// favours Winograd -
//   direct       144 MFMAs per iteration, 8 accumulator tiles, 2 workgroups per CU          (the shipped kernel's chunk)
//   winograd     64 MFMAs + 1152 v_add_f32 per iteration, 18 additions behind every MFMA, 1 workgroup per CU (256 + 128 registers)
//   winograd_pk  the same with v_pk_add_f32 (576 packed additions) - NOT available to the library: packed FP32 beside another stream's
//                MFMA kernel returns wrong values on this part (profiles/r04_coresidency.md), the library is built without it
// and prints the time per iteration and the chunk tiles a CU finishes per microsecond.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256, 2) void direct_loop(float* out, int iters) {
  f16v acc[8];
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = static_cast<_Float16>(0.001f * (threadIdx.x + i)); b[i] = static_cast<_Float16>(0.002f * (threadIdx.x + 2 * i)); }
  for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 18; ++s) {          // 9 taps x 2 k-steps
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
    }
  }
  float s = 0.0f;
  for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool PACKED>
__global__ __launch_bounds__(256, 1) void winograd_loop(float* out, int iters) {
  f16v prod[8];                              // the 4 positions of one transform row x 2 output tiles
  f16v yacc[8];                              // output-domain accumulators: 2 x 2 outputs x 2 output tiles
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = static_cast<_Float16>(0.001f * (threadIdx.x + i)); b[i] = static_cast<_Float16>(0.002f * (threadIdx.x + 2 * i)); }
  for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) { prod[t][r] = 0.0f; yacc[t][r] = 0.001f * r; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int row = 0; row < 4; ++row) {      // transform rows: 16 MFMAs each, 288 additions behind them
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        prod[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, prod[m & 7], 0, 0, 0);
        // 18 additions behind every MFMA, on the accumulators of the OUTPUT domain only (no wait for a product)
        if (PACKED) {
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            const int t = (m + q) & 7, r = (2 * q + 2 * m) & 14;
            f2 v = {yacc[t][r], yacc[t][r + 1]}, w = {yacc[(t + 1) & 7][r], yacc[(t + 1) & 7][r + 1]};
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(w));
            yacc[t][r] = v[0]; yacc[t][r + 1] = v[1];
          }
        } else {
#pragma unroll
          for (int q = 0; q < 18; ++q) {
            const int t = (m + q) & 7, r = (q + 3 * m) & 15;
            float v = yacc[t][r];
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(yacc[(t + 1) & 7][r]));
            yacc[t][r] = v;
          }
        }
      }
    }
  }
  float s = 0.0f;
  for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += prod[t][r] + yacc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
static double run(K kernel, int grid, int iters, float* out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, out, 200);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, out, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0.0f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / iters;                   // microseconds per iteration of every resident workgroup
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  float* out;
  CK(hipMalloc(&out, sizeof(float) * 256 * 4 * cus));
  const int iters = 4000;
  const double d = run(direct_loop, 2 * cus, iters, out);
  const double w = run(winograd_loop<false>, cus, iters, out);
  const double wp = run(winograd_loop<true>, cus, iters, out);
  printf("%s, %d CUs; one iteration = one 32-channel chunk of a 16 x 16-pixel x 128-output workgroup tile\n", p.name, cus);
  printf("direct       144 MFMA per wave, 2 workgroups / CU : %.3f us per iteration -> %.3f chunk tiles per us and CU (%.1f TFLOP/s of direct-convolution work on the chip)\n",
         d, 2.0 / d, 2.0 / d * cus * 2.0 * 256 * 128 * 32 * 9 / 1e6);
  printf("winograd      64 MFMA + 1152 v_add_f32, 1 workgroup / CU : %.3f us per iteration -> %.3f chunk tiles per us and CU (x %.2f of direct)\n", w, 1.0 / w, (1.0 / w) / (2.0 / d));
  printf("winograd_pk   64 MFMA +  576 v_pk_add_f32, 1 workgroup / CU : %.3f us per iteration -> %.3f chunk tiles per us and CU (x %.2f of direct)\n", wp, 1.0 / wp, (1.0 / wp) / (2.0 / d));
  return 0;
}
