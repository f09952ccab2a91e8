// probe_tools.hip — MEASUREMENT kernels (libpvo_probe.so, include/pvo_probe.h): the shader-clock probe and the memory-request probe
// that bench.py / tools/mem_probe.py report.  Built beside libpvo_hip.so, not part of it: nothing of the product calls them.
#include "common.h"
#include "../../include/pvo_probe.h"

// One wave runs a dependent chain of `iters` x 64 v_fma_f32 and reports how many shader cycles (s_memtime) and how many
// 10 ns ticks of the constant 100 MHz counter (s_memrealtime) it took: launched on a second stream beside a kernel, the
// ratio is the clock the chip sustains under that kernel's load (MI355X lowers the clock to hold its power budget).
namespace {
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, int iters) {
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  float a = static_cast<float>(threadIdx.x), b = 1.0001f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 64; ++k) a = __builtin_fmaf(a, b, 0.5f);
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = static_cast<unsigned long long>(a); }
}
}  // namespace

extern "C" int pvo_clock_probe(void* out3_u64, int iters, void* stream) {
  if (!out3_u64 || iters <= 0) return PVO_EINVAL;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, pvo_stream(stream), static_cast<unsigned long long*>(out3_u64), iters);
  return hipGetLastError() == hipSuccess ? PVO_OK : PVO_ELAUNCH;
}

// Memory-request probe: groups of `lanes` lanes (8: a 128-byte line, 4: a 64-byte half line) read one RANDOM line of `buf`
// each, 16 bytes per lane, eight independent loads in flight per lane (mode 1 / 2), or consecutive lines (mode 0).  Timed by the
// caller, it gives the rate at which this part's memory system serves SCATTERED lines - the ceiling of the correlation lookup,
// whose windows are a few partial lines each at data-dependent places (measured: ~56 G requests/s whether a request is 64 or
// 128 bytes, i.e. 3.6 TB/s of half lines, 7.1 TB/s of whole lines, 7.4 TB/s streaming).
namespace {
typedef uint32_t mp_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t mp_mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__global__ __launch_bounds__(256) void mem_probe_kernel(const unsigned char* __restrict__ buf, uint32_t nlines, int iters, int lanes, int stream_mode,
                                                        uint32_t* __restrict__ sink) {
  const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
  const uint32_t grp = gid / lanes, sub = gid % lanes, ngrp = gridDim.x * 256u / lanes, bytes = lanes * 16u;
  mp_u32x4 acc = {0u, 0u, 0u, 0u};
  for (int it = 0; it < iters; ++it) {
    mp_u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t idx = (it * 8u + k) * ngrp + grp;
      const uint32_t line = stream_mode ? idx % nlines : mp_mix(idx * 2654435761u + 12345u) % nlines;
      v[k] = *reinterpret_cast<const mp_u32x4*>(buf + static_cast<size_t>(line) * bytes + sub * 16u);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc ^= v[k];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = gid;       // (keeps the loads alive)
}
}  // namespace

extern "C" long long pvo_mem_probe(const void* buf, size_t bytes, int mode, int iters, int blocks, void* sink, void* stream) {
  if (!buf || !sink || iters <= 0 || blocks <= 0 || mode < 0 || mode > 2 || bytes < 4096 || (reinterpret_cast<uintptr_t>(buf) & 127)) return -1;
  const int lanes = mode == 2 ? 4 : 8;
  const uint32_t nlines = static_cast<uint32_t>(bytes / (lanes * 16u));
  hipLaunchKernelGGL(mem_probe_kernel, dim3(blocks), dim3(256), 0, pvo_stream(stream), static_cast<const unsigned char*>(buf), nlines, iters, lanes,
                     mode == 0 ? 1 : 0, static_cast<uint32_t*>(sink));
  if (hipGetLastError() != hipSuccess) return -1;
  return static_cast<long long>(blocks) * 256 / lanes * iters * 8 * (lanes * 16);         // bytes fetched by the launch
}
