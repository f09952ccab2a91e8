/*
 * pvo_probe.h - C ABI of libpvo_probe.so: two measurement kernels used by bench.py and tools/ (NOT part of libpvo_hip.so, the
 * product library; moved out of it in round 4).  Same conventions as pvo_hip.h: device pointers, a hipStream_t as void*.
 */
#ifndef PVO_PROBE_H
#define PVO_PROBE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Shader-clock probe: one wave runs iters x 64 dependent v_fma_f32 and writes {s_memtime cycles, s_memrealtime ticks of
 * 10 ns, (unused)} to out3_u64 (device, 3 x uint64).  Launched on a second stream beside a kernel, cycles / (10 ns x
 * ticks) is the clock the chip sustains under that kernel's load (the MFMA-bound convolutions run at ~1.35 GHz on random
 * data, 2.1 GHz on zero-filled operands, 2.4 GHz idle: DESIGN.md section 5). */
int pvo_clock_probe(void* out3_u64, int iters, void* stream);
/* Memory-request probe (measurement): one launch of `blocks` x 256 lanes, each lane 8 x iters independent 16-byte loads from
 * `buf` (device, 128-byte aligned, `bytes` long; use >= 1 GiB so that neither L2 nor the Infinity Cache holds it) -
 * mode 0: consecutive 128-byte lines (streaming), 1: one RANDOM 128-byte line per 8 lanes, 2: one random 64-byte half line per
 * 4 lanes.  Returns the number of bytes the launch fetches (lines x line size), or -1; the caller times it.  The rate of mode 2
 * is the ceiling of the correlation lookup, whose traffic is scattered partial lines.  sink: >= 4 bytes. */
long long pvo_mem_probe(const void* buf, size_t bytes, int mode, int iters, int blocks, void* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif
