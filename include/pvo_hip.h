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
 *   - one HIP runtime per process: a host that also loads PyTorch must load torch's
 *     libamdhip64 before this library
 *   - threading / streams: as the reference (called from the single Python main thread with the GIL held,
 *     droid.cpp has no gil_scoped_release), the library is re-entrant PER DEVICE, not per thread: the
 *     composite calls (pvo_update_operator, pvo_graph_update) fork onto ONE library-owned side stream per device
 *     (pvo_side_stream) with one set of fork / join events, created lazily without a lock, and keep one record per device of
 *     the gate context computed ahead (pvo_graph_update_args.context_ahead).  One host thread per device
 *     issues them, on one launch stream at a time; two graphs may share a device as long as their calls are not
 *     issued concurrently from different threads or interleaved on different launch streams without the caller
 *     ordering those streams.  The plain kernels (lookup, build, ba, geometry) have no shared state and may be
 *     called from any thread on any stream.
 *   - kernels side by side on one device: every kernel of this library may be resident beside kernels of other streams,
 *     pvo_ba included.  That was NOT so before round 4: on MI355X (ROCm 7.0.2) packed-FP32 VALU instructions returned wrong
 *     values while an MFMA kernel of another stream was resident on the compute unit, which made pvo_ba irreproducible
 *     beside the library's own convolutions; the library is therefore built WITHOUT packed-FP32 instructions
 *     (-target-feature -packed-fp32-ops: pvo_amd/build.py, DESIGN.md section 5, profiles/r04_coresidency.md) - keep the
 *     flag when building it by other means.  The converse is untested: a CALLER's kernels that use packed-FP32 arithmetic,
 *     run on another stream, beside this library's matrix-core kernels.
 *
 * Limits (PVO_EUNSUPPORTED beyond them)
 *   - edges / images per call: 65535 (they index a grid's y or z dimension)
 *   - one image: H * W * channels < 2^31 elements
 *   - bundle adjustment: at most 2048 free poses; the reduced pose system is factorised by ONE workgroup in envelope
 *     form (dense in LDS up to 21 free poses, compact envelope blocks in LDS while they fit ~140 KB - 63 poses of a
 *     radius-3 graph use 121 KB - and dense in global memory beyond that, which is slow: O(P band^2) dependent fp64 steps
 *     through L2)
 *   - panoptic segments per frame: `max_segments` of the caller (DepthVideo: 1024 dense labels)
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
/* text of the HIP error behind this thread's most recent PVO_ELAUNCH (hipGetErrorString) */
const char* pvo_last_hip_error(void);
/* ABI version of the library that was loaded; PVO_ABI_VERSION is the one this header describes.  The argument structs below are
 * passed by pointer and have GROWN between versions (100 -> 101: pvo_graph_update_args.context_ahead / context_ready; 101 -> 102: no struct changed - new entry points
 * pvo_ba_pack / pvo_ba_finish_packed / pvo_ba_last_partition / pvo_proj_transform[_vjp], and pvo_ba_workspace_bytes returns more): a caller
 * checks pvo_version() == PVO_ABI_VERSION, or pvo_graph_update_args_size() == sizeof(pvo_graph_update_args), once after loading. */
#define PVO_ABI_VERSION 103
int pvo_version(void);
size_t pvo_graph_update_args_size(void);

/* Test hook (102 -> 103).  Until round 5 eleven environment variables read inside the library selected experiment variants; a
 * process environment is invisible at a call site and two ranks could differ in it.  The experiments are gone from the source and
 * the THREE selections the test-suite needs - it compares bit-identical forms of the same computation - are set by this one call,
 * process-wide, not thread-safe against running calls.  A product caller never needs it: every knob defaults to 0 = "the shipped
 * choice".  Returns PVO_EINVAL for an unknown knob.  (Reference counterpart: none - droid.cpp has no such switches.) */
enum {
  PVO_KNOB_BA_SOLVER = 0,         /* 0 shipped choice by size | 1 blocked | 2 one wave | 3 pipelined | 4 partitioned (two workgroups) | 5 dense on the fp64 matrix cores (<= 29 poses) | 6 dense in 48 x 48 blocks over many workgroups (beyond the LDS path) */
  PVO_KNOB_HEADS_GATHER_FLAT = 1, /* 1: pvo_heads_gather without its LDS-tiled form */
  PVO_KNOB_NO_RIDERS = 2,         /* 1: pvo_graph_update computes the upsampling mask and the next gate context as launches of their own */
  PVO_KNOB_POST_SEPARATE = 3,     /* 1: pvo_graph_update runs pvo_graph_post as a launch of its own instead of as the epilogue of the heads' gather */
  PVO_KNOB_COUNT = 4
};
int pvo_debug_config(int knob, int value);
int pvo_knob(int knob); /* current value (0 for an unknown knob) */

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
 *   out    [N,num_levels*(2r+1)^2,h1,w1] dtype, or [N,h1,w1,num_levels*(2r+1)^2] when
 *          out_channels_last != 0 (the layout the update operator's NHWC convolutions read)
 * Level l is sampled at coords / 2^l exactly as corr.py:47 does.
 * slots (device int[N], may be NULL): the volume of edge n is slot slots[n] of level tensors holding num_slots
 * volumes each — a resident pool, so adding / dropping factor-graph edges never moves a 25 MB volume (the
 * reference re-indexes and torch.cat's the whole pyramid on every edge change, factor_graph.py:135,177). */
int pvo_corr_pyramid_lookup(const void* const* volumes_host, const float* coords, void* out,
                            int N, int h1, int w1, int h2, int w2,
                            int num_levels, int radius, int dtype, int out_channels_last,
                            const int* slots, int num_slots, void* stream);

/* droid_backends.altcorr_forward (droid.cpp:190-200; altcorr_kernel.cu:27-149, host :290-319):
 * volume-free lookup.  fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C] channels-last, coords [B,S,H1,W1,2] f32,
 * corr [B,S,(2r+1)^2,H1,W1] fully written, channel = iy + (2r+1)*ix.  fp32 only (AltCorrBlock casts
 * to float, corr.py:120); other dtypes return PVO_EUNSUPPORTED. */
int pvo_altcorr_forward(const void* fmap1, const void* fmap2, const float* coords, void* corr,
                        int B, int S, int H1, int W1, int H2, int W2, int C,
                        int radius, int dtype, void* stream);

/* droid_backends.altcorr_backward (droid.cpp:202-214; altcorr_kernel.cu:152-286, host :321-356; float
 * only there too).  fmap1_grad [B,H1,W1,C] and fmap2_grad [B,H2,W2,C] are fully written (fmap2_grad is
 * zeroed on the stream before its atomic accumulation).  The reference's third output, coords_grad, is
 * all zeros (:340) and is produced by the Python binding. */
int pvo_altcorr_backward(const void* fmap1, const void* fmap2, const float* coords,
                         const void* corr_grad, void* fmap1_grad, void* fmap2_grad,
                         int B, int S, int H1, int W1, int H2, int W2, int C,
                         int radius, int dtype, void* stream);

/* ------------------------------------------------------------------------- */
/* Correlation volume build                                                   */
/* ------------------------------------------------------------------------- */

/* CorrBlock.corr + the avg-pool pyramid (modules/corr.py:24-38,63-71).
 *   fmap1,fmap2 : features, [N,C,H,W] (channels_last = 0, the reference's layout) or
 *                 [N,H,W,C] (channels_last = 1, the layout AltCorrBlock already uses, corr.py:104)
 *   levels_host[l] : device pointer of level l output [N,H,W,H>>l,W>>l] dtype
 * fp16/bf16 + channels_last + C in {16,32,64,128} + 16-byte aligned features run on the matrix
 * cores with every level written from the accumulators; anything else takes the generic path.
 * out_slots (device int[N], may be NULL): edge n is written into slot out_slots[n] of the level tensors (pool).
 * level0[n,p1,p2] = sum_c (fmap1[n,c,p1]/4)*(fmap2[n,c,p2]/4), fp32 accumulate,
 * rounded to dtype; level l+1 = 2x2 mean of the ROUNDED level l (floor sizes). */
int pvo_corr_build(const void* fmap1, const void* fmap2, void* const* levels_host,
                   int N, int C, int H, int W, int num_levels, int dtype, int channels_last,
                   const int* out_slots, void* stream);

/* 8x8-TILED pyramid for a resident volume pool (no reference counterpart: a storage layout).  Level l of the
 * pool is [slots, h1*w1 planes, ceil(Hl/8), ceil(Wl/8), 8, 8] elements (Hl = H >> l): one tile of 16-bit
 * elements is one 128-byte line, so an 8x8 tap window touches <= 4 lines instead of 8 (measured: -27 % HBM
 * traffic for the lookup).  Values are identical to the row-major pyramid.  Requirements: fp16/bf16, 4 levels,
 * radius 3, channels-last features, C in {16,32,64,128}, H, W >= 8 (any map size: VKITTI2's 30x101 included),
 * 16-byte aligned pointers;
 * anything else returns PVO_EUNSUPPORTED (use the row-major entry points). */
int pvo_corr_build_tiled(const void* fmap1, const void* fmap2, void* const* levels_host,
                         int N, int C, int H, int W, int dtype, const int* out_slots, void* stream);
int pvo_corr_pyramid_lookup_tiled(const void* const* volumes_host, const float* coords, void* out,
                                  int N, int h1, int w1, int h2, int w2, int num_levels, int dtype,
                                  int out_channels_last, const int* slots, int num_slots, void* stream);

/* Lookup + first correlation-encoder layer in one kernel: out[N,h1,w1,128] = relu(W corr + b), where corr is the
 * 196-channel result of pvo_corr_pyramid_lookup_tiled (4 levels, radius 3) and W/b are the weights of the update
 * operator's Conv2d(196,128,1) (droid_net.py:172-175).  The 196 channels never reach HBM.  enc_weight is
 * [128 outputs][224] in `dtype` (input channel k < 196 in the lookup's channel order, 28 zero columns of padding),
 * enc_bias f32 [128].  Same shape/dtype restrictions as the tiled lookup. */
int pvo_corr_lookup_encode_tiled(const void* const* volumes_host, const float* coords,
                                 const void* enc_weight, const float* enc_bias, void* out,
                                 int N, int h1, int w1, int dtype,
                                 const int* slots, int num_slots, void* stream);

/* ------------------------------------------------------------------------- */
/* Update operator (DynamicUpdateModule, droid_net.py:166-314) on the 16-bit     */
/* inference path: every layer is a hand-written kernel, no MIOpen / hipBLASLt   */
/* ------------------------------------------------------------------------- */

/* All feature tensors are channels-last rows [E*H*W, C] of fp16/bf16 (`dtype`), 16-byte aligned.  Convolution filters
 * arrive re-arranged once by the host ("taps": [9 taps (ky*3+kx)][Cout][Cin]); biases are f32.
 *
 * ConvGRU (modules/gru.py:19-32), as issued here:
 *   static-input split   conv(W,[net|inp|corr|flow]) = conv(W[:,dyn],[net|corr|flow]) + conv(W[:,inp], inp); `inp` is
 *                        constant over an edge's life, so P_zr [rows,256] / P_q [rows,128] = conv(W[:,inp], inp) are
 *                        computed once per edge (pvo_conv3x3) and added inside the gate epilogues
 *   pvo_gru_glo_fused    per-chunk partial means of sigmoid(w(net) + b) * net  (gru.py:22-24, 1x1 conv inside)
 *   pvo_gate_context     g[E,384] = [convz_glo | convr_glo | convq_glo](glo) + their biases + the z/r/q conv biases
 *   pvo_gru_conv_gates / pvo_gru_conv_candidate   the two wide 3x3 convolutions with the gate arithmetic as epilogue */

/* glo_part[e, k, c] = (1/HW) * sum over pixels of chunk k (256 pixels) of sigmoid((W net)[c] + b[c]) * net[c], so
 * glo[e,c] = sum_k glo_part[e,k,c]; K = pvo_gru_glo_chunks(HW).  net [E,HW,128], w_weight [128 out][128 in] in `dtype`,
 * w_bias f32 [128] or NULL, glo_part f32 [E,K,128]. */
int pvo_gru_glo_chunks(int HW);
int pvo_gru_glo_fused(const void* net, const void* w_weight, const float* w_bias, float* glo_part,
                      int E, int HW, int dtype, void* stream);
/* g[e, j] = g_bias[j] + sum_c (sum_k glo_part[e,k,c]) * wg_t[c, j];  wg_t f32 [128][384], g_bias f32 [384], g f32 [E,384] */
int pvo_gate_context(const float* glo_part, const float* wg_t, const float* g_bias, float* g,
                     int E, int chunks, void* stream);
/* y[E,H,W,ystride (channels yoff .. yoff+Cout)] = act(conv3x3(x[E,H,W,128], zero padding 1) + bias) on the matrix cores:
 * the 128-input 3x3 convolutions (corr_encoder[2], flow_encoder[2], GraphAgg.conv1/conv2; droid_net.py:79-95,172-180).
 * Cout in {64, 128, 256, 512}; w_taps [9][Cout][128]; bias f32 [Cout] or NULL; relu != 0 applies ReLU.
 * ystride = 0 means a dense output (ystride = Cout, yoff = 0); otherwise the result lands in a channel slice of a wider
 * tensor (the encoders write [corr features | flow features] side by side for the ConvGRU). */
int pvo_conv3x3_c128(const void* x, const void* w_taps, const float* bias, void* y,
                     int E, int H, int W, int Cout, int relu, int ystride, int yoff, int dtype, void* stream);
/* the same for wide layers (implicit GEMM, 16x16 pixel tile x 128 outputs per workgroup): Cin % 32 == 0, Cout % 128 == 0.
 * The filter is read in MFMA-fragment order, [Cout/128][Cin/32 chunks][9 taps][2][2][2][64 lanes][8]: element
 * (cg, cc, t, wn, nt, ks, lane, j) = W[tap t][output cg*128 + wn*64 + nt*32 + (lane & 31)][input cc*32 + ks*16 + (lane >> 5)*8 + j]
 * (one coalesced 1 KB load per fragment, no LDS staging of the filter). */
int pvo_conv3x3(const void* x, const void* w_taps, const float* bias, void* y,
                int E, int H, int W, int Cin, int Cout, int relu, int ystride, int yoff, int dtype, void* stream);
/* The ConvGRU's two large convolutions with the gate arithmetic as their epilogue (modules/gru.py:26-31); the 256 gate
 * pre-activations and the 128 candidate pre-activations never reach HBM.  The input channels are read from two tensors,
 * [net | cf] resp. [RN | cf] (cf [E,H,W,cf_channels], cf_channels % 32 == 0: the encoders' outputs side by side), so no
 * concatenated input is assembled (the torch.cat's of gru.py:20-21,28).
 *   pvo_gru_conv_gates:     Z  = sigmoid(conv3x3([net|cf], w)[:, :128] + g[e, 0:128]   + P_zr[:, :128])
 *                           RN = sigmoid(conv3x3([net|cf], w)[:, 128:] + g[e, 128:256] + P_zr[:, 128:]) * net
 *   pvo_gru_conv_candidate: net_out = (1 - Z) * net + Z * tanh(conv3x3([RN|cf], w) + g[e, 256:384] + P_q)
 * w_taps [9][256 or 128][128 + cf_channels], g f32 [E,384], P_zr [E,H,W,256], P_q / net / Z / RN / net_out [E,H,W,128];
 * net_out may alias net.
 * p_slots (int32 [E] or NULL): the static terms live in a slot pool, P_zr [slots,H,W,256] / P_q [slots,H,W,128], and edge e
 * reads image p_slots[e] - the factor graph keeps them in the slots of its volume pool, so they are written once when an edge
 * is created and never gathered or concatenated when the edge set changes. */
int pvo_gru_conv_gates(const void* net, const void* cf, int cf_channels, const void* w_taps, const float* g,
                       const void* P_zr, const int* p_slots, void* Z, void* RN, int E, int H, int W, int dtype, void* stream);
int pvo_gru_conv_candidate(const void* RN, const void* cf, int cf_channels, const void* w_taps, const float* g,
                           const void* P_q, const int* p_slots, const void* Z, const void* net, void* net_out,
                           int E, int H, int W, int dtype, void* stream);
/* Second stage of the four output heads (droid_net.py:184-210) in one launch: y[E,H,W,8] =
 * Conv3x3(128->2) per head applied to relu(h1[..., head*128:(head+1)*128] + bias1), zero padding.
 * h1 [E,H,W,512] is the bias-free output of the four first-stage convolutions; w2 is
 * [4 heads][2 outputs][9 taps (ky*3+kx)][128 channels] in `dtype`; bias1 [512], bias2 [8] f32. */
/* The same heads WITHOUT the [E,H,W,512] hidden tensor in HBM (what pvo_update_operator / pvo_graph_update run):
 *   pvo_conv3x3_heads  first-stage convolutions (128 -> 4 x 128, w1_taps as for pvo_conv3x3, bias1 [512]) whose epilogue
 *                      keeps relu(hidden) in LDS and multiplies every pixel's 128 hidden channels of a head by all nine
 *                      second-stage tap filters on the matrix cores: z [E,H,W,4,18] f32, z[q][head][2 t + o] =
 *                      W2[head][o][tap t] . hidden_head(q).  w2_frags: the second-stage filter as MFMA B fragments,
 *                      [4 heads][8 k-steps][64 lanes][8]: element (h, ks, lane, j) = W2[h][o][t][ks*16 + (lane>>5)*8 + j]
 *                      for n = lane & 31 = 2 t + o < 18, else 0.
 *   pvo_heads_gather   y[p][2 head + o] = bias2[2 head + o] + sum_t z[p + tap t][head][2 t + o] (zero padding) -> [E,H,W,8]. */
int pvo_conv3x3_heads(const void* x, const void* w1_taps, const float* bias1, const void* w2_frags, float* z,
                      int E, int H, int W, int dtype, void* stream);
int pvo_heads_gather(const float* z, const float* bias2, void* y, int E, int H, int W, int dtype, void* stream);
/* pvo_heads_out: SUPERSEDED by the pair above (round 2); not called by pvo_update_operator / pvo_graph_update.  Kept as a
 * single-layer entry point for tests/test_update_operator.py, which checks the pair against it. */
int pvo_heads_out(const void* h1, const float* bias1, const void* w2, const float* bias2, void* y,
                  int E, int H, int W, int dtype, void* stream);
/* y[E,H,W,128] = relu(conv7x7(x[E,H,W,8], zero padding 3) + bias): the first layer of the update operator's
 * flow encoder (droid_net.py:176-180) on the matrix cores; 16-bit channels-last in and out.  w_taps is the filter
 * re-arranged to [52 taps (ky*7+kx; 49 real + 3 zero)][128 outputs][8 input channels] in `dtype`; bias f32 [128]. */
int pvo_conv7x7_c8(const void* x, const void* w_taps, const float* bias, void* y,
                   int E, int H, int W, int dtype, void* stream);
/* The whole flow encoder (droid_net.py:176-180) in one kernel: y[E,H,W,ystride (channels yoff .. yoff+64)] =
 * relu(conv3x3(relu(conv7x7(x[E,H,W,8]) + bias7), zero padding 1) + bias3); the 128-channel intermediate stays in LDS.
 * Bit-identical to pvo_conv7x7_c8 followed by pvo_conv3x3_c128(Cout = 64, relu = 1).  w7_taps as pvo_conv7x7_c8's,
 * w3_taps [9][64][128] as pvo_conv3x3_c128's; bias7 f32 [128], bias3 f32 [64] or NULL; ystride = 0: dense. */
/* (pvo_flow_encoder: not the default inside the update - two kernels are faster there, PVO_FLOW_ENCODER_FUSED=1 selects it;
 * an entry point of its own for callers that run the flow encoder alone, and for tests) */
int pvo_flow_encoder(const void* x, const void* w7_taps, const float* bias7, const void* w3_taps, const float* bias3,
                     void* y, int E, int H, int W, int ystride, int yoff, int dtype, void* stream);
/* y[rows,128] = relu(W corr + b) for an already sampled correlation tensor corr [rows,196] (channels-last, 8-byte aligned):
 * corr_encoder[0:2] = Conv2d(196,128,1) + ReLU (droid_net.py:172-175) for callers without a resident volume pool (motion
 * filter, global BA with alt-corr).  enc_weight as in pvo_corr_lookup_encode_tiled: [128][224], zero padded. */
int pvo_corr_encode(const void* corr, const void* enc_weight, const float* enc_bias, void* y, long long rows,
                    int dtype, void* stream);
/* out[k] = mean over e in [seg_ptr[k], seg_ptr[k+1]) of x[seg_idx[e]] — GraphAgg's scatter_mean over the edges sharing a
 * source frame (droid_net.py:83-87); in_bias != NULL: x is first mapped through relu(x + in_bias[c]) (the bias + ReLU of
 * the bias-free convolution that produced it).  x [E,HW,C], out [K,HW,C]. */
int pvo_segment_mean(const void* x, const int* seg_ptr, const int* seg_idx, const float* in_bias, void* out,
                     int K, int HW, int C, int dtype, void* stream);
/* GraphAgg's eta head (droid_net.py:72-74,93-95): e = 0.01 * softplus(conv3x3(x, w)[1 channel] + bias), x [K,H,W,128],
 * w_taps [9][128] in `dtype`, bias f32 [1].
 *   frame == NULL: eta[k] = e for the K = R images (what GraphAgg returns).
 *   frame != NULL: FactorGraph's damping bookkeeping as well (factor_graph.py:281-297): row r of eta belongs to frame
 *     frame[r] (int64 [R]) and takes image pos[r] of x (int32 [R]; -1: the frame only carries inactive edges and keeps
 *     its stored damping):  damping[frame[r]] = e (pos[r] >= 0);  eta[r] = eta_scale * damping[frame[r]] + EP
 *     (eta_scale = 0.2 in FactorGraph.update, factor_graph.py:297; 1 in update_lowmem, :352).
 * damping f32 [buffer,H,W] (updated in place), eta f32 [R,H,W]. */
int pvo_eta_head(const void* x, const void* w_taps, const float* bias, const int64_t* frame, const int* pos,
                 float* damping, float* eta, int R, int H, int W, float EP, float eta_scale, int dtype, void* stream);
/* y[rows,Cout] = act(x[rows,128] W^T + b), W [Cout][128] in `dtype`, Cout % 192 == 0: GraphAgg.upmask_disp =
 * Conv2d(128, 576, 1) (droid_net.py:76-77). */
int pvo_conv1x1_c128(const void* x, const void* w, const float* bias, void* y, long long rows, int Cout, int relu,
                     int dtype, void* stream);

/* FactorGraph.update's arithmetic around the update operator (factor_graph.py:231-306).  All [E,H,W,2] tensors are f32.
 *   pvo_graph_motion  motn [E,H,W,8] (`dtype`, channels-last) = clamp([target-coords0 | target-coords0+delta_dy |
 *                     target-coords1 | raw_mask], +-64)                                   (:233-237)
 *   pvo_segment_hist  panoptic vote, counting half (:256-261): tot[e,s] = pixels of segment s on edge e, dyn[e,s] = those
 *                     whose UPDATED mask (raw_mask + delta_mask, read from heads) is dynamic on either channel;
 *                     segm int32 [E,H,W] dense segment labels in [0, max_segments), 0 = no segment; tot/dyn int32
 *                     [E,max_segments] (zeroed here: two fills of exactly E*max_segments*4 bytes, nothing else is written)
 *   pvo_graph_post    heads [E,H,W,8] (`dtype`) = delta | delta_dy | weight logits | delta_mask as pvo_heads_out writes them:
 *                     raw_mask += delta_mask (in place); bin = sigmoid(raw_mask) >= dy_thresh; with segm != NULL, bin = 0
 *                     on both channels where the pixel's segment s != 0 has dyn[e,s] / max(tot[e,s],1) > vote_thresh
 *                     (:262-276); target = coords1 + delta; delta_dy = delta_dy_raw (1-bin); weight = sigmoid(logits +
 *                     10 (1-bin)); full_flow = coords1 + delta_dy - coords0; target_ba / weight_ba [E,2,H,W] are the
 *                     layouts pvo_ba reads                                                (:249-306)
 * target / delta_dy may alias the tensors pvo_graph_motion read. */
/* Per-frame encoders (round 5).  Reference: BasicEncoder / ResidualBlock, VO_Module/droid_slam/modules/extractor.py:6-56,116-201, run
 * once per frame by MotionFilter.track (motion_filter.py:52-60) under fp16 autocast.  No native entry point exists for them in
 * droid.cpp (cuDNN + ATen element-wise kernels do the work); here everything BETWEEN two convolutions of an encoder layer is one
 * kernel: t = x + bias[c]; if norm: instance norm over the plane (biased variance, fp32 statistics, eps); if relu_inner: relu;
 * if residual: t = residual + t; if relu_outer: relu - every step rounded to `dtype` (PVO_F16 / PVO_BF16) as the separate
 * ATen kernels round it.  x, residual (or NULL), y: [planes = N * C, HW] contiguous planes (NCHW), 4-byte aligned; bias [C] in
 * `dtype` or NULL.  y may alias x. */
int pvo_bias_norm_act(const void* x, const void* bias, const void* residual, void* y, long long planes, int C, int HW,
                      int norm, float eps, int relu_inner, int relu_outer, int dtype, void* stream);
/* pvo_bias_norm_act for LARGE planes (the encoders' layers at 1/2 and 1/4 resolution): a plane is cut into pvo_bias_norm_act_slices(HW)
 * slices (0: use pvo_bias_norm_act) - one launch leaves every slice's (mean, squared deviations) in ws (f32, 2 * planes * slices values;
 * only read / written with norm != 0), a second combines a plane's slices in index order and finishes its slice.  Same operations and
 * roundings as pvo_bias_norm_act; the instance-norm statistics differ from its two-pass sums in the last fp32 bits, deterministically.
 * planes <= 65535.  y may alias x. */
int pvo_bias_norm_act_slices(int HW);
int pvo_bias_norm_act_split(const void* x, const void* bias, const void* residual, void* y, long long planes, int C, int HW,
                            int norm, float eps, int relu_inner, int relu_outer, int dtype, float* ws, size_t ws_floats, void* stream);
/* The encoders' 1 x 1 convolutions on NCHW planes, bias included - the last layer Conv2d(128, output_dim, 1) (extractor.py:139,199) and
 * the residual blocks' strided shortcut Conv2d(in, out, 1, stride = 2) (extractor.py:31-33):
 *   y[n][co][oy][ox] = round(round(sum_ci w[co][ci] x[n][ci][oy * stride][ox * stride]) + bias[co]),  products added in index order in
 * fp32 - DETERMINISTIC (the vendor library's implicit GEMM for the last layer's shape splits K over workgroups with atomics: the same frame
 * gave different feature maps, and a sequence a different trajectory, from run to run; its strided 1 x 1 form is four launches).
 * x [N,Cin,Hin,Win], w [Cout,Cin], bias [Cout] or NULL, y [N,Cout,Hout,Wout] with Hout = (Hin - 1) / stride + 1, all `dtype` (PVO_F16 /
 * PVO_BF16).  Cin a multiple of 32, Cout of 64 (PVO_EUNSUPPORTED otherwise: callers keep the library convolution). */
int pvo_conv1x1_planes(const void* x, const void* w, const void* bias, void* y, int N, int Cin, int Cout, int Hin, int Win, int stride,
                       int dtype, void* stream);
/* A frame as the stream hands it over ([3][H][W] BGR 0..255; in_kind 0 = int32, 1 = uint8, 2 = float32) -> the encoders' input [3][H][W] RGB
 * `dtype`: ((v / 255) - mean[c]) / std[c] in fp32, the operations and order of motion_filter.py:52-54, then rounded.  mean3 / std3: HOST
 * pointers to three floats each. */
int pvo_frame_normalise(const void* img, void* out, int H, int W, const float* mean3, const float* std3, int in_kind, int dtype, void* stream);

/* (pvo_graph_motion: inside pvo_graph_update the motion features are written by pvo_reproject_motion since round 3; this
 * entry point serves pvo_update_operator callers that reproject themselves, and tests) */
int pvo_graph_motion(const float* target, const float* coords1, const float* delta_dy, const float* raw_mask,
                     void* motn, int E, int H, int W, int dtype, void* stream);
int pvo_segment_hist(const int* segm, const float* raw_mask, const void* heads, int* tot, int* dyn,
                     int E, int HW, int max_segments, float dy_thresh, int dtype, void* stream);
int pvo_graph_post(const float* coords1, const void* heads, float* raw_mask, float* target, float* delta_dy,
                   float* weight, float* target_ba, float* weight_ba, float* full_flow,
                   int E, int H, int W, float dy_thresh, const int* segm, const int* vote_tot, const int* vote_dyn,
                   int max_segments, float vote_thresh, int dtype, void* stream);

/* ---- the operator and the whole graph update as single calls (update_exec.hip) ------------------------------------ */

enum {
  PVO_OP_CONV128_WIDE = 1,   /* corr_encoder[2] / GraphAgg.conv1 on pvo_conv3x3 instead of pvo_conv3x3_c128 */
  PVO_OP_SINGLE_STREAM = 2,  /* no second stream at all */
  PVO_OP_ENC_SIDE_STREAM = 4 /* flow encoder + global context on the second stream beside the lookup and corr_encoder[2] */
};

/* Device pointers to the update operator's parameters, re-arranged once by the host (pvo_amd/modules/update.py
 * `packed_weights`); parameter names are the reference's (droid_net.py:172-225, gru.py:9-17). */
typedef struct pvo_update_weights {
  int dtype;                                     /* PVO_F16 or PVO_BF16: every 16-bit tensor below and all activations */
  int flags;                                     /* PVO_OP_* */
  const void* enc0_w;   const float* enc0_b;     /* corr_encoder.0: [128][224] zero padded, [128] */
  const void* cenc2_w;  const float* cenc2_b;    /* corr_encoder.2: taps [9][128][128], [128] */
  const void* fenc0_w;  const float* fenc0_b;    /* flow_encoder.0: [52][128][8], [128] */
  const void* fenc2_w;  const float* fenc2_b;    /* flow_encoder.2: taps [9][64][128], [64] */
  const void* glo_w;    const float* glo_b;      /* gru.w: [128][128], [128] */
  const float* gate_wt; const float* gate_b;     /* gru.conv{z,r,q}_glo: f32 [128][384] (transposed), f32 [384] (+ conv{z,r,q} biases) */
  const void* zr_w;     const void* q_w;         /* gru.convz|convr, gru.convq over [net|corr|flow]: taps [9][256][320], [9][128][320] */
  const void* zr_inp_w; const void* q_inp_w;     /* the same filters' `inp` input channels: taps [9][256][128], [9][128][128] */
  const void* heads1_w; const float* heads1_b;   /* delta|delta_dy|weight|delta_mask .0: taps [9][512][128], [512] */
  const void* heads2_w; const float* heads2_b;   /* ... .2: as pvo_conv3x3_heads' w2_frags [4][8][64][8], [8] */
  const void* agg1_w;   const float* agg1_b;     /* agg.conv1: taps [9][128][128], [128] */
  const void* agg2_w;   const float* agg2_b;     /* agg.conv2 */
  const void* eta_w;    const float* eta_b;      /* agg.eta.0: [9][128], [1] */
  const void* up_w;     const float* up_b;       /* agg.upmask_disp.0: [576][128], [576] */
} pvo_update_weights;

/* One call of DynamicUpdateModule.forward(net, inp, corr, flow, ii) (droid_net.py:256-314). */
typedef struct pvo_operator_args {
  int E, H, W;
  /* correlation features: the factor graph's tiled volume pool + coordinates (lookup fused with corr_encoder.0) ... */
  const void* levels[4]; const int* slots; int num_slots;
  const float* coords;        /* [E,H,W,2] f32 */
  const void* corr;           /* ... or, when levels[0] == NULL, a sampled tensor [E,H,W,196] */
  const void* motion;         /* [E,H,W,8] */
  const void* net;            /* [E,H,W,128] hidden state */
  void* net_out;              /* [E,H,W,128] new hidden state (may alias net) */
  const void* inp;            /* [E,H,W,128] context features; read only when P_zr / P_q are NULL */
  const void* P_zr; const void* P_q;   /* cached static-input terms [E,H,W,256], [E,H,W,128], or NULL */
  int static_by_slot;         /* 1: P_zr / P_q are slot pools [num_slots,H,W,C] indexed by `slots` (see pvo_gru_conv_gates) */
  const int* seg_ptr; const int* seg_idx; int K;   /* GraphAgg groups: CSR of edges by source frame (K = 0: no aggregation) */
  void* heads;                /* [E,H,W,8] out: delta | delta_dy | weight logits | delta_mask */
  const int64_t* eta_frame; const int* eta_pos; int R; float* damping; float EP; float eta_scale;   /* see pvo_eta_head */
  float* eta;                 /* [R or K,H,W] f32 out, or NULL */
  void* upmask;               /* [K,H,W,576] out, or NULL */
} pvo_operator_args;

size_t pvo_operator_workspace_bytes(int E, int K, int H, int W);
int pvo_update_operator(const pvo_update_weights* weights, const pvo_operator_args* args,
                        void* workspace, size_t workspace_bytes, void* stream);

/* One call of FactorGraph.update (factor_graph.py:227-307) on a resident tiled volume pool. */
typedef struct pvo_graph_update_args {
  pvo_operator_args op;       /* coords / corr / motion / heads are supplied from the workspace; eta too unless op.eta is set */
  int nframes;
  float* poses; float* disps; const float* intrinsics;      /* [nframes,7], [nframes,H,W], [nframes,4]; poses / disps updated in place */
  const int64_t* ii; const int64_t* jj;                      /* active edges [E] */
  float* target; float* delta_dy; float* raw_mask;           /* [E,H,W,2] f32 state, updated in place */
  float* weight; float* full_flow;                           /* [E,H,W,2] f32 out */
  const int* segm; int max_segments; float vote_thresh;      /* panoptic vote (segm == NULL: off) */
  float dy_thresh;
  int n_in;                                                  /* inactive edges that take part in the BA (use_inactive) */
  float* target_ba; float* weight_ba;                        /* [n_in + E,2,H,W] f32: rows [0,n_in) filled by the caller, the rest here */
  const int64_t* ii_ba; const int64_t* jj_ba;                /* [n_in + E] */
  int t0, t1, itrs, motion_only; float lm, ep;               /* itrs = 0: no BA here (an edge-sharded caller runs it between all-reduces) */
  void* sys; void* ba_ws; size_t ba_ws_bytes;                /* [(6P)^2 + 6P] x 8 bytes, ZERO on entry (every solve leaves it zero again: allocate it
                                                                zeroed once); planned with pvo_ba_plan(ii_ba, jj_ba, ..., K_eta = op.R) */
  int clamp_frames; float disp_min;                          /* disps[:clamp_frames].clamp_(min=disp_min) (depth_video.py:214) */
  int want_upmask;                                           /* compute agg.upmask_disp although FactorGraph.update discards it */
  /* The ConvGRU's gate context is a function of the hidden state and the weights only.  context_ahead = 1: this call also
   * computes the context of op.net_out - the NEXT update's input - inside its two pose solves' dispatches (needs itrs >= 2
   * and a depth BA; ignored otherwise) and leaves it in the workspace.  context_ready = 1: the caller's promise that neither
   * op.net nor the context weights (glo_w, glo_b, gate_wt, gate_b) have been written since the previous pvo_graph_update of
   * this device returned op.net as net_out; the library then uses the stored context if that call was made with context_ahead,
   * the same workspace, weight struct and weight pointers, E, H, W and net_out == this op.net - and computes it as usual if not. */
  int context_ahead, context_ready;
} pvo_graph_update_args;

size_t pvo_graph_update_workspace_bytes(int E, int K, int R, int H, int W, int max_segments);
/* The library's second stream on the current device (created on first use, high priority): the one pvo_update_operator /
 * pvo_graph_update run their side chains on.  A caller with work of its own that may run beside the launch stream and is
 * only needed by the NEXT update (the volumes and static GRU terms of newly added edges, factor_graph.py:106-161) can
 * queue it there instead of creating yet another stream: HIP multiplexes streams onto a handful of hardware queues, and
 * an extra stream can end up sharing a queue with one of the two that must overlap.  Ordering against the launch
 * stream is the caller's business (events).  Writes the hipStream_t to *stream_out. */
int pvo_side_stream(void** stream_out);

/* HOST function (no device work, any thread): the greedy proximity-edge selection of FactorGraph.add_proximity_factors
 * (VO_Module/droid_slam/factor_graph.py:372-429) from a distance matrix.  dist [ni][nj] f32: distance of frames (t0 + a, t1 + b), both
 * ranges ending at the video's counter (t0 + ni == t1 + nj).  Out of the candidates: cells with (i - rad < j) or !(d <= 100), and the
 * diamond |di| + |dj| <= min(|i - j| - 2, nms) around every existing edge (have_i, have_j)[n_have] with |i - j| > 2.  Taken, each in both
 * directions: the temporal neighbours i < j <= i + rad first, then the remaining cells in ascending distance (ties by index) while
 * <= thresh, every accepted edge suppressing its own diamond.  out_i / out_j [max_out] receive *n_out edges (PVO_EWORKSPACE if they do
 * not fit: 2 * (ni * rad + ni * nj) always does). */
int pvo_proximity_select(const float* dist, int ni, int nj, int t0, int t1, int rad, int nms, double thresh,
                         const long long* have_i, const long long* have_j, int n_have,
                         long long* out_i, long long* out_j, int max_out, int* n_out);
/* Measurement hook: HIP events recorded on the launch stream around one stage of the following pvo_graph_update /
 * pvo_update_operator calls (at most `capacity` occurrences), so a benchmark can read a kernel's duration inside its timed
 * steps.  pvo_probe_read waits for the recorded events, writes their elapsed times in milliseconds to HOST memory,
 * disarms the probe and returns the number of samples (or -1). */
enum { PVO_STAGE_LOOKUP = 0, PVO_STAGE_GATES = 1, PVO_STAGE_CANDIDATE = 2, PVO_STAGE_BA = 3, PVO_STAGE_UPDATE = 4,
       PVO_STAGE_EMPTY = 5 /* an event pair around nothing in front of the lookup: what the pair itself adds to a reading */ };
int pvo_probe_arm(int stage, int capacity);
/* The same, sampling one occurrence in `every` (>= 1): the event pair costs the launch stream a few microseconds per occurrence. */
int pvo_probe_arm_every(int stage, int capacity, int every);
int pvo_probe_read(float* ms_host, int max_n);
int pvo_graph_update(const pvo_update_weights* weights, const pvo_graph_update_args* args,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* Reprojection helpers                                                       */
/* ------------------------------------------------------------------------- */

/* droid_backends.frame_distance (droid.cpp:117-133; droid_kernels.cu:497-636,
 * 1414-1436).  poses [nposes,7] (tx ty tz qx qy qz qw), disps [nframes,ht,wd],
 * intrinsics [4] = fx fy cx cy, ii/jj [M] int64, dist [M]. */
int pvo_frame_distance(const float* poses, const float* disps, const float* intrinsics,
                       const int64_t* ii, const int64_t* jj, float* dist,
                       int M, int ht, int wd, float beta, void* stream);
/* DepthVideo.distance(bidirectional=True) (depth_video.py:183-193: two frame_distance calls and 0.5 * (d1 + d2)) in one
 * launch; dist[m] equals that formulation bit for bit. */
int pvo_frame_distance_bidirectional(const float* poses, const float* disps, const float* intrinsics,
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
 * ix [N] int64, thresh [N], counter [N,ht,wd] (overwritten: the six neighbour views
 * are summed in-register instead of six atomicAdd passes). nframes = disps.size(0). */
int pvo_depth_filter(const float* poses, const float* disps, const float* intrinsics,
                     const int64_t* ix, const float* thresh, float* counter,
                     int N, int nframes, int ht, int wd, void* stream);

/* DepthVideo.reproject -> pops.projective_transform(jacobian=False)
 * (depth_video.py:154-163; geom/projective_ops.py:102-130) with per-frame
 * intrinsics [nframes,4].  coords [E,ht,wd,2], valid [E,ht,wd,1]. */
int pvo_reproject(const float* poses, const float* disps, const float* intrinsics,
                  const int64_t* ii, const int64_t* jj, float* coords, float* valid,
                  int E, int ht, int wd, void* stream);
/* pvo_reproject + pvo_graph_motion (below: FactorGraph.update's motion features of the same edges, factor_graph.py:231-237) in
 * one pass, bit-identical to the two calls; motn [E,ht,wd,8] in `dtype`, 16-byte aligned. */
int pvo_reproject_motion(const float* poses, const float* disps, const float* intrinsics,
                         const int64_t* ii, const int64_t* jj, float* coords, float* valid,
                         const float* target, const float* delta_dy, const float* raw_mask, void* motn,
                         int E, int ht, int wd, int dtype, void* stream);

/* ------------------------------------------------------------------------- */
/* SE3 element-wise operations (lietorch subset)                              */
/* ------------------------------------------------------------------------- */

/* The SE3 operations of lietorch the VO path uses (lietorch/groups.py:141-178,199-209; CUDA kernels
 * lietorch/src/lietorch_gpu.cu:21-296), forward, fp32 / fp64 (`dtype` PVO_F32 / PVO_F64), one thread per element.
 * Group elements are [n,7] = (tx,ty,tz, qx,qy,qz,qw), tangents [n,6] = (tau, phi).
 *   pvo_se3_unary : op 0 exp [n,6] -> [n,7];  1 log [n,7] -> [n,6];  2 inv [n,7] -> [n,7]
 *   pvo_se3_binary: op 3 mul (a b) [7],[7] -> [7];  4 act on homogeneous points [7],[4] -> [4];  5 act on points [7],[3] -> [3];
 *                   6 adj [7],[6] -> [6];  7 adjT [7],[6] -> [6]
 *                   element i of the n outputs reads a[i / rep_a] and b[i / rep_b]: an operand with fewer elements is
 *                   broadcast by index (e.g. one pose per edge acting on H*W points: rep_a = H*W, rep_b = 1) where
 *                   lietorch materialises it with .repeat (broadcasting.py:27-29). */
int pvo_se3_unary(int op, const void* x, void* y, long long n, int dtype, void* stream);
int pvo_se3_binary(int op, const void* a, long long rep_a, const void* b, long long rep_b, void* y, long long n,
                   int dtype, void* stream);
/* Backward of the two (round 4; lietorch's counterpart: the *_backward kernels of lietorch_gpu.cu): vector-Jacobian products in the
 * operands' own coordinates, i.e. what autograd gives for the formulas - se3_ops.hip evaluates its forward templates on dual numbers.
 *   pvo_se3_unary_vjp : gx [n, in] from gy [n, out] (in / out as pvo_se3_unary)
 *   pvo_se3_binary_vjp: ga [n,7] and gb [n, size of b's element] PER OUTPUT ELEMENT i (which read a[i / rep_a], b[i / rep_b]); either
 *                       may be NULL; the caller sums them over the repeats of a broadcast operand. */
int pvo_se3_unary_vjp(int op, const void* x, const void* gy, void* gx, long long n, int dtype, void* stream);
int pvo_se3_binary_vjp(int op, const void* a, long long rep_a, const void* b, long long rep_b, const void* gy,
                       void* ga, void* gb, long long n, int dtype, void* stream);

/* projective_transform of the training path (VO_Module/droid_slam/geom/projective_ops.py:106-130: iproj, relative pose, actp, proj and
 * their closed-form Jacobians) as ONE kernel per direction.  poses [B,P,7] (t, xyzw quaternion; world-to-camera), depths [B,P,ht,wd]
 * inverse depth, intr [B,P,4], ii / jj [N] device int64; outputs x1 [B,N,ht,wd,nx] (nx = 2, or 3 with the inverse depth in frame j),
 * valid [B,N,ht,wd] (Z > 0.2, as a number), and - all three or none - Ji, Jj [B,N,ht,wd,2,6], Jz [B,N,ht,wd,2].  dtype PVO_F32 / PVO_F64.
 * _vjp: vector-Jacobian product in ambient coordinates (what torch.autograd returns for the PyTorch formulation,
 * pvo_amd/geom/projective_ops.py); any of the four output gradients may be NULL; gposes [B,P,7] / gdepths [B,P,ht,wd] must be zero on
 * entry and receive the sums over edges and pixels (fp atomics: the order of those sums is not fixed). */
int pvo_proj_transform(const void* poses, const void* depths, const void* intr, const int64_t* ii, const int64_t* jj,
                       int B, int P, int N, int ht, int wd, int nx, void* x1, void* valid, void* Ji, void* Jj, void* Jz,
                       int dtype, void* stream);
int pvo_proj_transform_vjp(const void* poses, const void* depths, const void* intr, const int64_t* ii, const int64_t* jj,
                           int B, int P, int N, int ht, int wd, int nx, const void* g_x1, const void* g_Ji, const void* g_Jj, const void* g_Jz,
                           void* gposes, void* gdepths, int dtype, void* stream);

/* ------------------------------------------------------------------------- */
/* Dense bundle adjustment                                                    */
/* ------------------------------------------------------------------------- */

/* Scratch needed by pvo_ba for a graph of E edges over a pose window of
 * P = t1 - t0 poses, frame ids < nframes, maps of HW pixels. */
size_t pvo_ba_workspace_bytes(int E, int P, int nframes, int HW);

/* droid_backends.ba (droid.cpp:87-114; ba_cuda droid_kernels.cu:1293-1410).
 *   poses [nframes,7] and disps [nframes,ht,wd] are UPDATED IN PLACE
 *   intrinsics [4]; targets, weights [E,2,ht,wd]; eta [K_eta,ht,wd];
 *   ii, jj [E] int64; pose window [t0,t1); `iterations` Gauss-Newton steps
 *   dx_out [t1-t0,6]; dz_out [dz_rows,ht*wd] (may be NULL) receive the last step
 * K (number of depth maps optimised) = |unique([t0..t1) U ii)|; eta must have K
 * rows, or 1 row which is then broadcast; dz_rows >= K unless dz_out is NULL.
 * status_out (device int[4], may be NULL): [0]=0 ok / 1 non-SPD in some iteration
 * (that step's dx is 0, as droid_kernels.cu:1186-1189), [1]=K found on device,
 * [2]=1 if K_eta mismatched K (the call then updates NOTHING and [0] is 1 as well), [3] reserved.
 * The depth back-substitution reproduces EvT6x1_kernel's skip of window pose 0
 * (droid_kernels.cu:1084).  expSE3 uses xi[5] where the reference reads xi[45] (:154).
 * No host synchronisation: the factor-graph index structures are built on the
 * device, the (6P)^2 system is factorised in fp64 by one workgroup.
 * Bitwise reproducible run to run, alone on the device or beside kernels of other streams (see "kernels side by side"
 * at the head of this file: true for a library built with pvo_amd/build.py's flags). */
int pvo_ba(float* poses, float* disps, const float* intrinsics,
           const float* targets, const float* weights, const float* eta,
           const int64_t* ii, const int64_t* jj,
           int E, int nframes, int ht, int wd, int K_eta,
           int t0, int t1, int iterations, float lm, float ep, int motion_only,
           float* dx_out, float* dz_out, int dz_rows, int* status_out,
           void* workspace, size_t workspace_bytes, void* stream);

/* Edge-sharded BA: one Gauss-Newton step split at the point where
 * ranks exchange the reduced pose system.
 *   pvo_ba_local : assemble this rank's edges, eliminate its depth maps, and write
 *                  the rank-local reduced system sys[(6P)*(6P) + 6P] (row-major A-S
 *                  followed by the rhs), no damping applied yet.
 *   -- caller all-reduces (sum) `sys` across ranks AS 64-BIT INTEGERS --
 *   pvo_ba_finish: damp + factorise + solve on every rank (identical input ->
 *                  identical dx), retract poses, back-substitute this rank's dz.
 * `sys` holds 64-bit FIXED-POINT numbers (int64, units of 2^-28): integer sums commute, so the
 * system - and with it every pose and depth - is bitwise reproducible from run to run and
 * identical for any partition of the edges over ranks (the reference's host-side sum of fp32
 * block sums has no such guarantee either way; fp64 atomics, used in round 1, did not).
 * pvo_ba_finish converts to fp64, solves in fp64 (as SparseBlock::solve, droid_kernels.cu:1160-1198)
 * and leaves `sys` ZEROED; bit 1 of pvo_ba_local's motion_only argument says that `sys` is
 * already zero (local -> finish -> local chains), otherwise pvo_ba_local clears it first.
 * clamp_frames > 0 (pvo_ba_finish, depth BA only): disps[:clamp_frames].clamp_(min=disp_min) in the back-substitution
 * launch - DepthVideo.ba's clamp (depth_video.py:214) without a launch of its own.
 * With one rank pvo_ba == plan + iterations x (local, finish).  pvo_ba_plan must run
 * before the first pvo_ba_local of a graph (same workspace); pass K_eta = -1 for a
 * motion-only plan. */
int pvo_ba_plan(const int64_t* ii, const int64_t* jj, int E, int nframes, int HW,
                int K_eta, int t0, int t1, void* workspace, size_t workspace_bytes,
                void* stream);
/* The pose solve beyond 21 free poses (the global bundle adjustment; the reference: Eigen's sparse LLT on the host,
 * droid_kernels.cu:1160-1198) is PARTITIONED when the system is block-banded: two workgroups eliminate the pose chain from
 * both ends at once, the separator - the poses that couple the two parts - last (ba.hip, ba_solve_twin_kernel).  Same
 * result on every rank of an edge-sharded run (same integer system, same partition); against the one-chain solve it agrees
 * to fp64 rounding.  Diagnostic: out[0] = m, out[1] = s of the last pvo_ba_finish on this workspace - poses [0, m) and
 * [s, P) were the two parts, [m, s) the separator; 0, 0 = one chain (loop closures that widen the separator beyond 12
 * poses, or a chain that would not get a quarter shorter).  Synchronises the stream. */
int pvo_ba_last_partition(void* workspace, size_t workspace_bytes, int E, int P, int nframes, int HW, int* out, void* stream);
int pvo_ba_local(const float* poses, const float* disps, const float* intrinsics,
                 const float* targets, const float* weights, const float* eta,
                 const int64_t* ii, const int64_t* jj,
                 int E, int nframes, int ht, int wd, int K_eta, int t0, int t1,
                 int motion_only, void* sys,
                 void* workspace, size_t workspace_bytes, void* stream);
int pvo_ba_finish(float* poses, float* disps, void* sys,
                  const int64_t* ii, const int64_t* jj,
                  int E, int nframes, int ht, int wd, int t0, int t1,
                  float lm, float ep, int motion_only, int clamp_frames, float disp_min,
                  float* dx_out, float* dz_out, int dz_rows, int* status_out,
                  void* workspace, size_t workspace_bytes, void* stream);
/* The message of an edge-sharded step, packed by the library (round 4): first_s (device, int[P]) is the
 * STRUCTURAL envelope of the pose system - first_s[b] = lowest free pose block row b can couple with, derived from the whole
 * graph's edge list, the same table on every rank (pvo_amd/parallel.py: envelope_structure).
 *   pvo_ba_packed_elems(first_host, P)  int64 elements of the message: 36 * sum_b (b - first[b] + 1) + 6P (host-side helper)
 *   pvo_ba_pack(sys, first_s, msg, P)    after pvo_ba_local: the lower-triangle blocks (b, first_s[b] .. b) and the rhs of `sys`
 *                                        gathered into msg (block row after block row, 36 entries per block, then the rhs);
 *                                        `sys` is left ZEROED (the next pvo_ba_local may say so, bit 1 of motion_only)
 *   -- caller all-reduces (sum) msg AS 64-BIT INTEGERS --
 *   pvo_ba_finish_packed(..., msg, first_s, ...)   pvo_ba_finish reading the message instead of the dense system: numeric
 *                                        envelope, fp64 + damping, (partitioned) solve, retraction, back-substitution.
 * Results are bit-identical to all-reducing the dense `sys` and calling pvo_ba_finish (the entries outside the structural
 * envelope are zero on every rank); the message is 124 KB instead of 1.15 MB at 63 poses of a radius-3 graph. */
size_t pvo_ba_packed_elems(const int* first_host, int P);
int pvo_ba_pack(void* sys, const int* first_s, void* msg, int P, void* stream);
int pvo_ba_finish_packed(float* poses, float* disps, const void* msg, const int* first_s,
                         const int64_t* ii, const int64_t* jj,
                         int E, int nframes, int ht, int wd, int t0, int t1,
                         float lm, float ep, int motion_only, int clamp_frames, float disp_min,
                         float* dx_out, float* dz_out, int dz_rows, int* status_out,
                         void* workspace, size_t workspace_bytes, void* stream);

/* pvo_ba_finish with riders: the pose solve is one workgroup, the rest of the chip idles meanwhile, and no other queue may run
 * beside the bundle adjustment - so up to three independent jobs are computed by additional workgroups of the SAME dispatch
 * (a job with a NULL output is absent; none of them may alias a BA buffer):
 *   c*  cy[crows, cCout] = cx[crows, 128] cw^T + cbias, exactly pvo_conv1x1_c128 without ReLU (cCout a multiple of 192; 16-byte
 *       aligned pointers; cdtype PVO_F16 | PVO_BF16).  pvo_graph_update sends GraphAgg's upsampling mask (droid_net.py:76-77,93
 *       - computed by the reference's update module and never read by its factor graph) this way;
 *   g*  gpart = pvo_gru_glo_fused(gnet [gE, gHW, 128], gw, gbias): the partial means of the ConvGRU's global context;
 *   x*  xg = pvo_gate_context(xpart, xwt, xbias) for xE edges.  x* reads what g* wrote, so the two go into DIFFERENT solves:
 *       pvo_graph_update computes the NEXT update's gate context (a function of the hidden state and the weights only) in
 *       the first and second solve of this one (pvo_graph_update_args.context_ahead / context_ready). */
typedef struct pvo_ba_riders {
  const void* cx; const void* cw; const float* cbias; void* cy; long long crows; int cCout; int cdtype;
  const void* gnet; const void* gw; const float* gbias; float* gpart; int gE; int gHW; int gdtype;
  const float* xpart; const float* xwt; const float* xbias; float* xg; int xE; int xchunks;
} pvo_ba_riders;
int pvo_ba_finish_riders(float* poses, float* disps, void* sys,
                         const int64_t* ii, const int64_t* jj,
                         int E, int nframes, int ht, int wd, int t0, int t1,
                         float lm, float ep, int motion_only, int clamp_frames, float disp_min,
                         float* dx_out, float* dz_out, int dz_rows, int* status_out,
                         void* workspace, size_t workspace_bytes,
                         const pvo_ba_riders* riders, void* stream);
/* ... with the convolution job only */
int pvo_ba_finish_conv1x1(float* poses, float* disps, void* sys,
                          const int64_t* ii, const int64_t* jj,
                          int E, int nframes, int ht, int wd, int t0, int t1,
                          float lm, float ep, int motion_only, int clamp_frames, float disp_min,
                          float* dx_out, float* dz_out, int dz_rows, int* status_out,
                          void* workspace, size_t workspace_bytes,
                          const void* cx, const void* cw, const float* cbias, void* cy, long long crows, int cCout, int cdtype,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PVO_HIP_H */
