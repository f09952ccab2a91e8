// Stand-alone reproducer for the co-residency question of DESIGN.md section 5 (no BA code, no torch, no library of ours).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_probe/coresidency_repro tools/coresidency_repro.hip     (here)
//   tools/_probe/coresidency_repro [iterations]                                                     (on the GPU box)
//
// What profiles/r03_sched_bisect.txt saw inside pvo_graph_update: with a kernel of ANOTHER hardware queue resident beside the
// BA, single 64-byte sectors of the rows one BA kernel stores and the NEXT kernel of the same stream loads (Eii, Eij) held the
// previous iteration's values, and a few 8-byte entries of the atomically accumulated pose system differed - never alone on the
// device, more often the more cache writeback / invalidate operations were in flight.  The diagnosis ("the per-XCD L2 maintenance
// at kernel boundaries / event fences is not atomic against a kernel still running on another queue") rested on the full update.
// This program has only the ingredients:
//   stream A, per iteration:  producer  - 216 workgroups rewrite a 2.6 MB buffer with values that depend on the iteration
//                                          and add fixed-point terms into a small table with 64-bit device-scope atomics;
//                             consumer  - 96 workgroups (the NEXT kernel of the same stream) read the buffer back, count the
//                                          words that do not hold THIS iteration's value (and which iteration they hold), and
//                                          check the table of the PREVIOUS consumer pass;
//   stream B (another HIP stream = another hardware queue), optional: a ~40 us filler kernel over its own buffer, launched so
//                             that it is resident beside both kernels of stream A;
//   optional: an event record (default flags: system-scope release) on stream B every iteration; explicit agent-scope
//             acquire / release fences in the stream-A kernels.
// Modes:  0 = stream A alone                      1 = + fillers on stream B
//         2 = 1 + default-flag event records      3 = 2 + explicit fences in producer / consumer
//         4 = 1 with the fillers on stream A itself (same queue: serialised, the control for "more kernels" alone)
//         5 = the library's side chain: fork event, a 28 MB streaming writer on stream B beside producer + consumer, join
//             event (events without the system-scope fence, as the library creates them)      6 = 5 with default-flag events
// Output: per mode, iterations run, words read, stale words (and of those, how many held the previous iteration's value),
// 64-byte sectors touched by stale words, wrong table entries.  "0 stale" in mode 0 and "> 0" in modes 1-3 = the platform
// property, reproduced without any of our kernels.  "0 everywhere" = this skeleton does not reproduce it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int kWords = 36 * 6 * 3072;        // the size of Eii at S-B: 663 552 floats
constexpr int kTable = 1806;                 // the size of the reduced pose system at 7 free poses
constexpr int kProdWG = 216, kConsWG = 96;

__device__ __forceinline__ unsigned value_of(unsigned iter, unsigned i) { return (iter * 2654435761u) ^ (i * 40503u + 12345u); }

template <bool FENCES>
__global__ __launch_bounds__(256) void producer(unsigned* buf, unsigned long long* table, unsigned iter) {
  if (FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  for (int i = blockIdx.x * 256 + threadIdx.x; i < kWords; i += kProdWG * 256) buf[i] = value_of(iter, i);
  // 90 "sums" per workgroup into the table: every entry receives the same total in every iteration (kProdWG * 90 adds of 1 spread
  // over 1806 entries by a fixed pattern), so the expected table is known without a reference run
  if (threadIdx.x < 90) atomicAdd(&table[(blockIdx.x * 90 + threadIdx.x) % kTable], 1ull);
  if (FENCES) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}

template <bool FENCES>
__global__ __launch_bounds__(256) void consumer(const unsigned* buf, unsigned long long* table, unsigned iter, unsigned long long* stats,
                                                unsigned* log, int log_cap) {
  if (FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  unsigned stale = 0, prev = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < kWords; i += kConsWG * 256) {
    const unsigned v = buf[i];
    if (v != value_of(iter, i)) {
      stale++;
      if (v == value_of(iter - 1, i)) prev++;
      const unsigned long long slot = atomicAdd(&stats[4], 1ull);
      if (slot < static_cast<unsigned long long>(log_cap)) { log[2 * slot] = iter; log[2 * slot + 1] = i; }
    }
  }
  if (stale) { atomicAdd(&stats[0], static_cast<unsigned long long>(stale)); atomicAdd(&stats[1], static_cast<unsigned long long>(prev)); }
  // the table: after `iter + 1` producer passes entry e holds (iter + 1) * (number of (wg, t) pairs that map to e)
  if (blockIdx.x == 0) {
    for (int e = threadIdx.x; e < kTable; e += 256) {
      unsigned long long n = 0;
      for (int k = e; k < kProdWG * 90; k += kTable) n++;
      if (table[e] != n * (iter + 1ull)) atomicAdd(&stats[2], 1ull);
    }
  }
  if (FENCES) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}

// ~40 us of memory + ALU work on its own buffer, enough workgroups to sit on every CU beside stream A's kernels
__global__ __launch_bounds__(256) void filler(float* x, int n, int rounds) {
  for (int r = 0; r < rounds; r++)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
      float v = x[i];
#pragma unroll 8
      for (int k = 0; k < 32; k++) v = v * 1.0001f + 0.5f;
      x[i] = v;
    }
}

// the shape of the upsampling-mask convolution that sits beside the BA in the library's overlapped arrangement: a small input
// (3 MB) read, 28 MB of 16-byte streaming stores
__global__ __launch_bounds__(256) void writer(const uint4* __restrict__ in, uint4* __restrict__ out, int n_in, int n_out) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_out; i += gridDim.x * 256) {
    uint4 v = in[i % n_in];
    v.x += i; v.y ^= v.x * 2654435761u; v.z += v.y; v.w ^= v.z;
    out[i] = v;
  }
}

struct Result { unsigned long long stale, prev, table_bad, logged; std::vector<unsigned> log; double ms; };

static Result run(int mode, int iters) {
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, -1));      // (the library's side stream is a high-priority stream)
  unsigned *buf, *log; unsigned long long *table, *stats; float* fill;
  const int log_cap = 4096, nfill = 16 << 20;      // 64 MB: the filler's buffer, and input + 28 MB output of the writer
  CK(hipMalloc(&buf, sizeof(unsigned) * kWords)); CK(hipMalloc(&table, 8 * kTable)); CK(hipMalloc(&stats, 8 * 8));
  CK(hipMalloc(&log, 8 * log_cap)); CK(hipMalloc(&fill, sizeof(float) * nfill));
  CK(hipMemset(buf, 0, sizeof(unsigned) * kWords)); CK(hipMemset(table, 0, 8 * kTable)); CK(hipMemset(stats, 0, 64));
  CK(hipMemset(log, 0, 8 * log_cap)); CK(hipMemset(fill, 0, sizeof(float) * nfill));
  CK(hipDeviceSynchronize());
  hipEvent_t ev, t0, t1;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));             // default fence flags: system-scope release at the record
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  const bool fences = mode == 3;
  hipEvent_t evF, evJ;                                                  // modes 5 / 6: fork + join like the library's side chain
  CK(hipEventCreateWithFlags(&evF, hipEventDisableTiming | (mode == 5 ? hipEventDisableSystemFence : 0)));
  CK(hipEventCreateWithFlags(&evJ, hipEventDisableTiming | (mode == 5 ? hipEventDisableSystemFence : 0)));
  CK(hipEventRecord(t0, a));
  for (int it = 0; it < iters; it++) {
    if (mode >= 5) {
      // A: (previous consumer) -> record F;  B: wait F, 28 MB writer, record J;  A: producer + consumer beside it, then wait J
      CK(hipEventRecord(evF, a)); CK(hipStreamWaitEvent(b, evF, 0));
      writer<<<2048, 256, 0, b>>>(reinterpret_cast<const uint4*>(fill), reinterpret_cast<uint4*>(fill) + (1 << 18), 1 << 17, 28 * 65536);
      CK(hipEventRecord(evJ, b));
      producer<false><<<kProdWG, 256, 0, a>>>(buf, table, it);
      consumer<false><<<kConsWG, 256, 0, a>>>(buf, table, it, stats, log, log_cap);
      CK(hipStreamWaitEvent(a, evJ, 0));
      if ((it & 255) == 255) { CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b)); }
      continue;
    }
    if (fences) producer<true><<<kProdWG, 256, 0, a>>>(buf, table, it); else producer<false><<<kProdWG, 256, 0, a>>>(buf, table, it);
    if (mode >= 1) filler<<<1024, 256, 0, mode == 4 ? a : b>>>(fill, nfill, 1);
    if (mode == 2 || mode == 3) CK(hipEventRecord(ev, b));
    if (fences) consumer<true><<<kConsWG, 256, 0, a>>>(buf, table, it, stats, log, log_cap);
    else consumer<false><<<kConsWG, 256, 0, a>>>(buf, table, it, stats, log, log_cap);
    if ((it & 255) == 255) { CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b)); }   // keep the two queues within 256 iterations of each other
  }
  CK(hipEventRecord(t1, a));
  CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
  unsigned long long h[8]; CK(hipMemcpy(h, stats, 64, hipMemcpyDeviceToHost));
  Result r{h[0], h[1], h[2], h[4], {}, ms};
  r.log.resize(2 * (h[4] < (unsigned long long)log_cap ? h[4] : log_cap));
  if (!r.log.empty()) CK(hipMemcpy(r.log.data(), log, r.log.size() * 4, hipMemcpyDeviceToHost));
  CK(hipFree(buf)); CK(hipFree(table)); CK(hipFree(stats)); CK(hipFree(log)); CK(hipFree(fill));
  CK(hipStreamDestroy(a)); CK(hipStreamDestroy(b)); CK(hipEventDestroy(ev)); CK(hipEventDestroy(evF)); CK(hipEventDestroy(evJ)); CK(hipEventDestroy(t0)); CK(hipEventDestroy(t1));
  return r;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s (%s), %d CUs; %d iterations per mode; buffer %d words, table %d entries\n", p.name, p.gcnArchName, p.multiProcessorCount,
         iters, kWords, kTable);
  const char* names[] = {"0 stream A alone", "1 + fillers on stream B (another queue)", "2 + default-flag event records on B",
                         "3 + explicit agent-scope fences in A's kernels", "4 fillers on stream A itself (same queue)",
                         "5 fork / 28 MB writer on B / join (no-fence events)", "6 fork / 28 MB writer on B / join (default events)"};
  int bad_alone = 0, bad_beside = 0;
  for (int mode : {0, 1, 2, 3, 4, 5, 6, 0}) {
    Result r = run(mode, iters);
    // distinct 64-byte sectors among the logged stale words
    std::vector<unsigned long long> sect;
    for (size_t k = 0; k + 1 < r.log.size(); k += 2) sect.push_back((static_cast<unsigned long long>(r.log[k]) << 32) | (r.log[k + 1] >> 4));
    size_t distinct = 0;
    if (!sect.empty()) { std::qsort(sect.data(), sect.size(), 8, [](const void* x, const void* y) { auto a = *(const unsigned long long*)x, b = *(const unsigned long long*)y; return (a > b) - (a < b); });
      distinct = 1; for (size_t k = 1; k < sect.size(); k++) distinct += sect[k] != sect[k - 1]; }
    printf("mode %-48s: %.1f us / iteration, words read %.3g, STALE %llu (previous iteration's value: %llu; %zu distinct 64-byte sectors among the first %zu), "
           "wrong table entries %llu\n", names[mode], r.ms * 1e3 / iters, (double)iters * kWords, r.stale, r.prev, distinct, r.log.size() / 2, r.table_bad);
    for (size_t k = 0; k + 1 < r.log.size() && k < 16; k += 2) printf("      stale word: iteration %u, index %u\n", r.log[k], r.log[k + 1]);
    if (mode == 0 || mode == 4) bad_alone += (r.stale || r.table_bad); else bad_beside += (r.stale || r.table_bad);
  }
  printf("verdict: %s\n", bad_alone ? "stale data even WITHOUT a second queue: this skeleton is broken, not the platform"
         : bad_beside ? "stale data only with a kernel of another queue resident: platform property reproduced without BA code"
                      : "no stale data in any mode: the skeleton does not reproduce the BA's co-residency failures");
  return 0;
}
