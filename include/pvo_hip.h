/*
 * pvo_hip.h — C ABI of libpvo_hip.so, the MI355X (gfx950) implementation of the
 * PVO VO_Module hot path (correlation lookup/build, dense bundle adjustment,
 * frame distance and friends).
 *
 * Every entry point replaces one function of the reference's pybind module
 * `droid_backends` (VO_Module/src/droid.cpp:234-247) or one piece of Python the
 * reference runs around it; the reference location is cited per function.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in `_host`
 *   - tensors are dense, row-major, in the reference's layouts
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every
 *     kernel is enqueued on it, nothing synchronises the device or the host
 *   - return value: PVO_OK (0) or a PVO_E* code; pvo_strerror() names it.
 *     Launch failures are reported from hipGetLastError() after the enqueue.
 *   - no function allocates device memory; scratch comes from the caller
 *     (`workspace`, sized by the matching *_workspace_bytes function)
 */
#ifndef PVO_HIP_H
#define PVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types of the correlation volume / feature maps */
enum { PVO_F32 = 0, PVO_F16 = 1, PVO_BF16 = 2, PVO_F64 = 3 };

enum {
  PVO_OK = 0,
  PVO_EINVAL = 1,     /* bad argument (shape, dtype, null pointer, alignment)   */
  PVO_ELAUNCH = 2,    /* HIP reported a launch error                            */
  PVO_EWORKSPACE = 3, /* workspace too small                                    */
  PVO_EUNSUPPORTED = 4
};

const char* pvo_strerror(int code);
int pvo_version(void);

/* ------------------------------------------------------------------------- */
/* Correlation lookup                                                         */
/* ------------------------------------------------------------------------- */

/* droid_backends.corr_index_forward (droid.cpp:167-175; kernel
 * correlation_kernels.cu:19-70, host :126-155).
 *   volume [N,h1,w1,h2,w2] dtype; coords [N,2,h1,w1] f32 (x plane then y plane)
 *   corr   [N,2r+1,2r+1,h1,w1] dtype  — fully written (no pre-zeroing needed)
 * channel order is the reference's: first index = x offset, second = y offset. */
int pvo_corr_index_forward(const void* volume, const float* coords, void* corr,
                           int N, int h1, int w1, int h2, int w2,
                           int radius, int dtype, void* stream);

/* droid_backends.corr_index_backward (droid.cpp:177-188; kernel
 * correlation_kernels.cu:73-124, host :157-185).
 *   corr_grad [N,2r+1,2r+1,h1,w1]; volume_grad [N,h1,w1,h2,w2] — fully written
 *   (zero where no tap lands), so the caller needs no zero fill. */
int pvo_corr_index_backward(const float* coords, const void* corr_grad, void* volume_grad,
                            int N, int h1, int w1, int h2, int w2,
                            int radius, int dtype, void* stream);

/* CorrBlock.__call__ (modules/corr.py:40-50) as ONE launch: all `num_levels`
 * pyramid levels gathered and written straight into the concatenated tensor.
 *   volumes_host[l] : device pointer of level l, [N,h1,w1,h2>>l,w2>>l]
 *   coords [N,h1,w1,2] f32 (the reference's un-permuted layout, x then y)
 *   out    [N,num_levels*(2r+1)^2,h1,w1] dtype
 * Level l is sampled at coords / 2^l exactly as corr.py:47 does. */
int pvo_corr_pyramid_lookup(const void* const* volumes_host, const float* coords, void* out,
                            int N, int h1, int w1, int h2, int w2,
                            int num_levels, int radius, int dtype, void* stream);

/* ------------------------------------------------------------------------- */
/* Reprojection helpers                                                       */
/* ------------------------------------------------------------------------- */

/* droid_backends.frame_distance (droid.cpp:117-133; droid_kernels.cu:497-636,
 * 1414-1436).  poses [nposes,7] (tx ty tz qx qy qz qw), disps [nframes,ht,wd],
 * intrinsics [4] = fx fy cx cy, ii/jj [M] int64, dist [M]. */
int pvo_frame_distance(const float* poses, const float* disps, const float* intrinsics,
                       const int64_t* ii, const int64_t* jj, float* dist,
                       int M, int ht, int wd, float beta, void* stream);

/* droid_backends.projmap (droid.cpp:136-151; droid_kernels.cu:406-495,1439-1464).
 * coords [E,ht,wd,3] (channel 2 left 0, as the reference does), valid [E,ht,wd,1]. */
int pvo_projmap(const float* poses, const float* disps, const float* intrinsics,
                const int64_t* ii, const int64_t* jj, float* coords, float* valid,
                int E, int ht, int wd, void* stream);

/* droid_backends.iproj (droid.cpp:154-163; droid_kernels.cu:758-829,1494-1517).
 * points [N,ht,wd,3]. */
int pvo_iproj(const float* poses, const float* disps, const float* intrinsics,
              float* points, int N, int ht, int wd, void* stream);

/* droid_backends.depth_filter (droid.cpp:217-231; droid_kernels.cu:640-754,1467-1491).
 * ix [N] int64, thresh [N], counter [N,ht,wd] (must be zero on entry: the six
 * neighbour views vote into it). nframes = disps.size(0). */
int pvo_depth_filter(const float* poses, const float* disps, const float* intrinsics,
                     const int64_t* ix, const float* thresh, float* counter,
                     int N, int nframes, int ht, int wd, void* stream);

/* DepthVideo.reproject -> pops.projective_transform(jacobian=False)
 * (depth_video.py:154-163; geom/projective_ops.py:102-130) with per-frame
 * intrinsics [nframes,4].  coords [E,ht,wd,2], valid [E,ht,wd,1]. */
int pvo_reproject(const float* poses, const float* disps, const float* intrinsics,
                  const int64_t* ii, const int64_t* jj, float* coords, float* valid,
                  int E, int ht, int wd, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PVO_HIP_H */
