"""Golden fixtures from the reference's KERNEL TEXT, executed on the host in the build container.

    python tests/golden/gen_kernel_golden.py     (needs /root/reference; never runs on the GPU box)

The reference's native code cannot be built here as it stands (no nvcc; droid_kernels.cu includes an Eigen the
vendored copy does not have).  What this script does instead: at generation time it reads line
ranges of the reference's .cu files FROM WHERE THEY LIE (/root/reference/VO_Module/src), puts a small shim in front
(the CUDA qualifiers defined away, blockIdx / threadIdx as thread-local variables, a grid loop on the host, `__shared__`
as `static`, `__syncthreads()` as a barrier between the real OS threads that play one thread block, atomicAdd as a
mutex-free add - see each shim for why that is enough), compiles the result as host C++ into a scratch directory
under /tmp and runs it on seeded inputs.  Only the INPUTS and OUTPUTS are written to tests/golden/*.npz; the
extracted text is never stored, in any form.

This is NOT "the reference compiled here": g++ stands in for nvcc and the shim for the CUDA execution model.  What it
buys is that the fixtures come from the reference's own statements - index arithmetic, guards, cast points, accumulation
order, c10::Half arithmetic - not from a restatement of them.

  correlation_kernels.cu:13-124   corr_index_forward_kernel / corr_index_backward_kernel (barrier-free)
                                  -> corr_lookup_kernel.npz   fp32 (with and without the FMA contraction nvcc applies
                                     by default), fp16 (c10::Half: float op, then round), radius 3 and 2,
                                     coordinates far outside the volume included
  droid_kernels.cu:26-29, 58-176 (constants and SE3 helpers; warpReduce / blockReduce / GPU_1D_KERNEL_LOOP are
      supplied by the shim, reasons below) and
      :406-495 projmap_kernel, :497-636 frame_distance_kernel, :640-754 depth_filter_kernel, :758-829 iproj_kernel
                                  -> geom_kernels.npz
      :177-403 projective_transform_kernel   -> ba_assemble_kernel.npz  (Hs, vs, Eii, Eij, Cii, bz of one BA step)
  altcorr_kernel.cu:16-286        altcorr_forward_kernel / altcorr_backward_kernel (one warp per block; see _ALTCORR_SHIM)
                                  -> altcorr_kernel.npz

blockReduce (droid_kernels.cu:36-55) relies on the lock-step execution of a warp (`warpReduce` on volatile shared
memory, no barrier): OS threads are not in lock step, so the shim supplies a reduction that performs the SAME additions
in the SAME tree order (128, 64, then 32 ... 1 inside "the warp", every step reading all operands before any write).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/VO_Module/src"
SCRATCH = "/tmp/pvo_kernel_golden_build"


def _lines(path, first, last):
    with open(path) as f:
        src = f.read().split("\n")
    return "\n".join(src[first - 1:last])


_COMMON = r"""
#include <torch/extension.h>
#include <cmath>
#include <cstdlib>
#include <thread>
#include <vector>
#include <mutex>
#include <condition_variable>
#define __global__
#define __device__
#define __forceinline__ inline
#define RestrictPtrTraits DefaultPtrTraits
struct shim_dim3 { int x = 1, y = 1, z = 1; };
static thread_local shim_dim3 blockIdx, blockDim, threadIdx, gridDim;
using std::abs;   // abs(double) as in CUDA's global namespace (depth_filter_kernel, droid_kernels.cu:746-750)
"""


def _load(name, cpp, functions, extra_cflags=()):
    from torch.utils.cpp_extension import load_inline
    os.makedirs(os.path.join(SCRATCH, name), exist_ok=True)
    return load_inline(name=name, cpp_sources=[cpp], functions=functions, extra_cflags=["-O2"] + list(extra_cflags),
                       build_directory=os.path.join(SCRATCH, name), verbose=False)


# ------------------------------------------------------------------------------------------------------------
# correlation lookup: barrier-free kernels, one host loop over the grid
# ------------------------------------------------------------------------------------------------------------
def _corr_module(tag, cflags):
    text = _lines(os.path.join(SRC, "correlation_kernels.cu"), 13, 124)
    cpp = _COMMON + text + r"""
template <typename scalar_t> static void run_fwd(torch::Tensor volume, torch::Tensor coords, torch::Tensor corr, int r) {
  auto v = volume.packed_accessor32<scalar_t,5,torch::DefaultPtrTraits>();
  auto c = coords.packed_accessor32<float,4,torch::DefaultPtrTraits>();
  auto o = corr.packed_accessor32<scalar_t,5,torch::DefaultPtrTraits>();
  const int N = volume.size(0), ht = volume.size(1), wd = volume.size(2);
  blockDim.x = BLOCK; blockDim.y = BLOCK;
  for (int bz = 0; bz < N; bz++) for (int by = 0; by < (ht + BLOCK - 1) / BLOCK; by++) for (int bx = 0; bx < (wd + BLOCK - 1) / BLOCK; bx++)
    for (int ty = 0; ty < BLOCK; ty++) for (int tx = 0; tx < BLOCK; tx++) {
      blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz; threadIdx.x = tx; threadIdx.y = ty;
      corr_index_forward_kernel<scalar_t>(v, c, o, r);
    }
}
template <typename scalar_t> static void run_bwd(torch::Tensor coords, torch::Tensor grad, torch::Tensor vgrad, int r) {
  auto c = coords.packed_accessor32<float,4,torch::DefaultPtrTraits>();
  auto g = grad.packed_accessor32<scalar_t,5,torch::DefaultPtrTraits>();
  auto o = vgrad.packed_accessor32<scalar_t,5,torch::DefaultPtrTraits>();
  const int N = vgrad.size(0), ht = vgrad.size(1), wd = vgrad.size(2);
  blockDim.x = BLOCK; blockDim.y = BLOCK;
  for (int bz = 0; bz < N; bz++) for (int by = 0; by < (ht + BLOCK - 1) / BLOCK; by++) for (int bx = 0; bx < (wd + BLOCK - 1) / BLOCK; bx++)
    for (int ty = 0; ty < BLOCK; ty++) for (int tx = 0; tx < BLOCK; tx++) {
      blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz; threadIdx.x = tx; threadIdx.y = ty;
      corr_index_backward_kernel<scalar_t>(c, g, o, r);
    }
}
// the host wrappers of correlation_kernels.cu:126-187 restated: zero-filled output, dispatch on the volume's type
torch::Tensor forward(torch::Tensor volume, torch::Tensor coords, int64_t radius) {
  auto corr = torch::zeros({volume.size(0), 2*radius+1, 2*radius+1, volume.size(1), volume.size(2)}, volume.options());
  if (volume.scalar_type() == torch::kFloat) run_fwd<float>(volume, coords, corr, radius);
  else if (volume.scalar_type() == torch::kHalf) run_fwd<c10::Half>(volume, coords, corr, radius);
  else if (volume.scalar_type() == torch::kDouble) run_fwd<double>(volume, coords, corr, radius);
  else TORCH_CHECK(false, "dtype");
  return corr;
}
torch::Tensor backward(torch::Tensor volume, torch::Tensor coords, torch::Tensor grad, int64_t radius) {
  auto vgrad = torch::zeros_like(volume);
  if (volume.scalar_type() == torch::kFloat) run_bwd<float>(coords, grad, vgrad, radius);
  else if (volume.scalar_type() == torch::kHalf) run_bwd<c10::Half>(coords, grad, vgrad, radius);
  else if (volume.scalar_type() == torch::kDouble) run_bwd<double>(coords, grad, vgrad, radius);
  else TORCH_CHECK(false, "dtype");
  return vgrad;
}
"""
    return _load("pvo_ref_corr_" + tag, cpp, ["forward", "backward"], cflags)


def corr_cases():
    """seeded inputs: (name, volume fp32, coords [N,2,h1,w1] fp32, radius)"""
    out = []
    g = np.random.default_rng(41)

    def case(name, N, h1, w1, h2, w2, r, spread, scale=1.0):
        vol = (g.standard_normal((N, h1, w1, h2, w2)) * scale).astype(np.float32)
        base = np.stack(np.meshgrid(np.arange(w1), np.arange(h1)), 0).astype(np.float32)
        co = base[None] * np.array([w2 / w1, h2 / h1], np.float32).reshape(1, 2, 1, 1) \
            + g.uniform(-spread, spread, (N, 2, h1, w1)).astype(np.float32)
        out.append((name, vol, co.astype(np.float32), r))

    case("a", 2, 6, 7, 9, 11, 3, 3.0)              # level-0 like: most windows cross a border somewhere
    case("b", 2, 5, 9, 4, 6, 3, 2.0, 4.0)          # a coarse level: the window is larger than the plane
    case("c", 1, 4, 5, 8, 8, 2, 40.0)              # radius 2; most coordinates far outside (all-zero windows)
    # ... and four coordinates as far out as a diverged pose puts them: +-1e7 (static_cast<int>(floor(x)) is still defined
    # there, correlation_kernels.cu:49-50).  Beyond the int range (3e9) and for NaN the conversion is undefined behaviour in C++:
    # x86 returns INT_MIN for both, PTX's cvt.rzi.s32.f32 saturates and maps NaN to 0 - what a host compilation of the text
    # does with them says nothing about the reference, so they stay out of the fixture (tests/test_corr_lookup_gpu.py checks
    # the HIP kernel against the oracle's saturating conversion there).
    co_c = out[-1][2].copy()
    co_c[0, :, 0, 0] = (1.0e7, 3.0)
    co_c[0, :, 0, 1] = (-1.0e7, -1.0e7)
    co_c[0, :, 1, 0] = (4.0, 1.0e7)
    co_c[0, :, 1, 1] = (9999999.5, -9999999.5)
    out[-1] = ("c", out[-1][1], co_c, 2)
    case("d", 1, 3, 4, 7, 5, 3, 1.0)
    # integer coordinates (dx = dy = 0), half-pixel, and the exact borders -r-1, h2+r
    co = out[-1][2].copy()
    co[0, :, 0, :] = np.array([[0.0, 4.0, -4.0, 8.0], [0.0, 6.0, -4.0, 10.0]], np.float32)
    co[0, :, 1, :] = np.array([[0.5, 4.5, -3.5, 7.5], [6.5, 0.5, -0.5, 3.0]], np.float32)
    out[-1] = ("d", out[-1][1], co, 3)
    return out


def gen_corr_lookup_kernel():
    plain = _corr_module("plain", ["-ffp-contract=off"])
    fused = _corr_module("fma", ["-mfma", "-ffp-contract=fast"])   # nvcc's default -fmad=true for float / double
    out = {}
    g = torch.Generator().manual_seed(43)
    for name, vol, co, r in corr_cases():
        v32, c = torch.from_numpy(vol), torch.from_numpy(co)
        rd = 2 * r + 1
        grad = torch.randn(vol.shape[0], rd, rd, vol.shape[1], vol.shape[2], generator=g)
        out[name + "_volume"], out[name + "_coords"], out[name + "_radius"] = vol, co, np.int64(r)
        out[name + "_grad"] = grad.numpy()
        out[name + "_fwd_f32_fma"] = fused.forward(v32, c, r).numpy()
        out[name + "_fwd_f32_nofma"] = plain.forward(v32, c, r).numpy()
        out[name + "_bwd_f32_fma"] = fused.backward(v32, c, grad, r).numpy()
        out[name + "_bwd_f32_nofma"] = plain.backward(v32, c, grad, r).numpy()
        h_a = plain.forward(v32.half(), c, r)
        h_b = fused.forward(v32.half(), c, r)
        assert torch.equal(h_a, h_b), "fp16: the conversion between product and sum leaves nothing to contract"
        out[name + "_fwd_f16"] = h_a.numpy()
        out[name + "_bwd_f16"] = plain.backward(v32.half(), c, grad.half(), r).numpy()
        assert torch.equal(plain.backward(v32.half(), c, grad.half(), r), fused.backward(v32.half(), c, grad.half(), r))
    np.savez_compressed(os.path.join(HERE, "corr_lookup_kernel.npz"), **out)
    print("corr_lookup_kernel.npz:", sorted(k for k in out if k.endswith("_volume")),
          "fma changes fp32 forward in %d of %d values (case a)" % (
              int((out["a_fwd_f32_fma"] != out["a_fwd_f32_nofma"]).sum()), out["a_fwd_f32_fma"].size))


# ------------------------------------------------------------------------------------------------------------
# droid_kernels.cu: kernels with __shared__ + __syncthreads -> one OS thread per CUDA thread of a block
# ------------------------------------------------------------------------------------------------------------
_BLOCK_SHIM = r"""
#define __shared__ static
#define __syncthreads() shim_barrier()
struct ShimBarrier {
  std::mutex m; std::condition_variable cv; int count = 0, gen = 0, n = 1;
  void wait() { std::unique_lock<std::mutex> l(m); int g = gen; if (++count == n) { gen++; count = 0; cv.notify_all(); }
                else cv.wait(l, [&]{ return g != gen; }); }
};
static ShimBarrier shim_bar;
static inline void shim_barrier() { shim_bar.wait(); }
// a block that returns early in some threads (`if (jx < 0) return;` is block-uniform in the text) leaves together
template <typename F> static void shim_launch(shim_dim3 grid, int threads, F body) {
  shim_bar.n = threads;
  for (int bz = 0; bz < grid.z; bz++) for (int by = 0; by < grid.y; by++) for (int bx = 0; bx < grid.x; bx++) {
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back([=]() {
      blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz; blockDim.x = threads; threadIdx.x = t; gridDim = grid;
      body();
    });
    for (auto &th : pool) th.join();
    shim_bar.count = 0;
  }
}
// GPU_1D_KERNEL_LOOP (droid_kernels.cu:32-33: `for (size_t k = threadIdx.x; k<n; k += blockDim.x)`) with a rendezvous
// when a thread ENTERS and when it LEAVES the loop.  frame_distance_kernel has two unsynchronised hand-offs around its
// pixel loop: every thread executes relSE3 into the SHARED tij / qij right before it (:561; relSE3 parks R*ti in tij
// before subtracting, :103-107), and thread 0 swaps the shared `ix` / `jx` right after its own loop (:619-623).  On the
// GPU the eight warps of a block run those statements side by side (and write identical values), so no thread observes an
// intermediate; OS threads start microseconds apart, so without the rendezvous late threads would read a half-written
// tij or the swapped index - a property of this emulation, not of the kernel.  Every thread of a block enters and leaves
// every such loop exactly once, so the barrier counts match.
static inline size_t shim_loop_enter(size_t k) { shim_barrier(); return k; }
static inline bool shim_loop_exit() { shim_barrier(); return false; }
#define GPU_1D_KERNEL_LOOP(k, n) for (size_t k = shim_loop_enter(threadIdx.x); k < (size_t)(n) || shim_loop_exit(); k += blockDim.x)
// atomicAdd(float*): the kernels that use it add from different BLOCKS (depth_filter: blockIdx.y neighbours), which this
// shim runs one after the other - a plain add; the counter values are small integers, order-free
static inline float atomicAdd(float *p, float v) { float o = *p; *p = o + v; return o; }
// blockReduce (droid_kernels.cu:45-55) with warpReduce (:36-43): the same additions in the same tree order; the
// text's last six steps rely on a warp running in lock step (every lane reads its operand before any lane writes)
static void blockReduce(volatile float *sdata) {
  shim_barrier();
  if (threadIdx.x == 0) {
    for (int s = 128; s >= 64; s >>= 1) for (int t = 0; t < s; t++) sdata[t] += sdata[t + s];
    for (int s = 32; s >= 1; s >>= 1) { float tmp[32]; for (int t = 0; t < 32; t++) tmp[t] = sdata[t] + sdata[t + s];
                                        for (int t = 0; t < 32; t++) sdata[t] = tmp[t]; }
  }
  shim_barrier();
}
"""


def _patched(text, old, new):
    """one token of the extracted text replaced at generation time; the substitution must hit exactly once"""
    assert text.count(old) == 1, (old, text.count(old))
    return text.replace(old, new)


def _droid_module(tag="plain", cflags=("-ffp-contract=off",), retr=None):
    """retr: None (the retraction kernels are left out), "xi45_zero" or "xi5" - pose_retr_kernel's `float xi[6]` is a local
    array and expSE3 (:154) reads xi[45], out of bounds.  Two defined readings are generated, each with ONE token of the text
    replaced at generation time: "xi45_zero" pads the local array to 46 zero-initialised floats (`float xi[6]` ->
    `float xi[46] = {}`: the read returns 0), "xi5" replaces `xi[45]` by `xi[5]` (upstream DROID-SLAM's statement)."""
    path = os.path.join(SRC, "droid_kernels.cu")
    helpers = _lines(path, 58, 176)      # actSO3 ... expSE3 (the SE3 helpers)
    retr_text = ""
    if retr is not None:
        retr_text = _lines(path, 856, 925)          # retrSE3, pose_retr_kernel, disp_retr_kernel
        if retr == "xi45_zero":
            retr_text = _patched(retr_text, "float xi[6],", "float xi[46] = {},")
        elif retr == "xi5":
            helpers = _patched(helpers, "xi[45]", "xi[5]")
        else:
            raise ValueError(retr)
    text = "\n".join([
        _lines(path, 26, 29),       # MIN_DEPTH, THREADS, NUM_BLOCKS
        helpers,
        _lines(path, 177, 403),     # projective_transform_kernel
        _lines(path, 406, 495),     # projmap_kernel
        _lines(path, 497, 636),     # frame_distance_kernel
        _lines(path, 640, 754),     # depth_filter_kernel
        _lines(path, 758, 829),     # iproj_kernel
        _lines(path, 833, 853),     # accum_kernel
        _lines(path, 980, 1094),    # EEt6x6_kernel, Ev6x1_kernel, EvT6x1_kernel
        retr_text,
    ])
    retr_wrappers = r"""
// pose_retr_kernel<<<1, THREADS>>>(poses, dx, t0, t1) and disp_retr_kernel<<<kx.size(0), THREADS>>>(disps, dz, kx)
// (droid_kernels.cu:1202-1203, 1393-1394); both update in place, the wrappers return the updated copies
torch::Tensor pose_retr(torch::Tensor poses, torch::Tensor dx, int64_t t0, int64_t t1) {
  auto out = poses.clone();
  shim_dim3 g;
  shim_launch(g, THREADS, [&]() { pose_retr_kernel(A2(out, float), A2(dx, float), (int)t0, (int)t1); });
  return out;
}
torch::Tensor disp_retr(torch::Tensor disps, torch::Tensor dz, torch::Tensor inds) {
  auto out = disps.clone();
  shim_dim3 g; g.x = inds.size(0);
  shim_launch(g, THREADS, [&]() { disp_retr_kernel(A3(out, float), A2(dz, float), A1(inds, long)); });
  return out;
}
""" if retr is not None else ""
    cpp = _COMMON + _BLOCK_SHIM + text + r"""
#define A1(t, T) t.packed_accessor32<T,1,torch::DefaultPtrTraits>()
#define A2(t, T) t.packed_accessor32<T,2,torch::DefaultPtrTraits>()
#define A3(t, T) t.packed_accessor32<T,3,torch::DefaultPtrTraits>()
#define A4(t, T) t.packed_accessor32<T,4,torch::DefaultPtrTraits>()
// host wrappers restated from droid_kernels.cu:1414-1510 (allocation + launch geometry only)
torch::Tensor frame_distance(torch::Tensor poses, torch::Tensor disps, torch::Tensor intr, torch::Tensor ii, torch::Tensor jj, double beta) {
  auto dist = torch::zeros({ii.size(0)}, poses.options());
  shim_dim3 g; g.x = ii.size(0);
  shim_launch(g, THREADS, [&]() { frame_distance_kernel(A2(poses, float), A3(disps, float), A1(intr, float), A1(ii, long), A1(jj, long), A1(dist, float), (float)beta); });
  return dist;
}
std::vector<torch::Tensor> projmap(torch::Tensor poses, torch::Tensor disps, torch::Tensor intr, torch::Tensor ii, torch::Tensor jj) {
  const int n = ii.size(0), ht = disps.size(1), wd = disps.size(2);
  auto coords = torch::zeros({n, ht, wd, 3}, disps.options());
  auto valid = torch::zeros({n, ht, wd, 1}, disps.options());
  shim_dim3 g; g.x = n;
  shim_launch(g, THREADS, [&]() { projmap_kernel(A2(poses, float), A3(disps, float), A1(intr, float), A1(ii, long), A1(jj, long), A4(coords, float), A4(valid, float)); });
  return {coords, valid};
}
torch::Tensor depth_filter(torch::Tensor poses, torch::Tensor disps, torch::Tensor intr, torch::Tensor ix, torch::Tensor thresh) {
  const int n = ix.size(0), ht = disps.size(1), wd = disps.size(2);
  auto counter = torch::zeros({n, ht, wd}, disps.options());
  shim_dim3 g; g.x = n; g.y = 6; g.z = NUM_BLOCKS(ht * wd);
  shim_launch(g, THREADS, [&]() { depth_filter_kernel(A2(poses, float), A3(disps, float), A1(intr, float), A1(ix, long), A1(thresh, float), A3(counter, float)); });
  return counter;
}
torch::Tensor iproj(torch::Tensor poses, torch::Tensor disps, torch::Tensor intr) {
  const int n = disps.size(0), ht = disps.size(1), wd = disps.size(2);
  auto points = torch::zeros({n, ht, wd, 3}, disps.options());
  shim_dim3 g; g.x = n; g.y = NUM_BLOCKS(ht * wd);
  shim_launch(g, THREADS, [&]() { iproj_kernel(A2(poses, float), A3(disps, float), A1(intr, float), A4(points, float)); });
  return points;
}
// the allocation of ba_cuda (droid_kernels.cu:1326-1341) and ONE launch of projective_transform_kernel (:1346-1349)
std::vector<torch::Tensor> ba_assemble(torch::Tensor targets, torch::Tensor weights, torch::Tensor poses, torch::Tensor disps,
                                       torch::Tensor intr, torch::Tensor ii, torch::Tensor jj) {
  const int n = ii.size(0), ht = disps.size(1), wd = disps.size(2);
  auto opts = poses.options();
  auto Hs = torch::zeros({4, n, 6, 6}, opts), vs = torch::zeros({2, n, 6}, opts);
  auto Eii = torch::zeros({n, 6, ht*wd}, opts), Eij = torch::zeros({n, 6, ht*wd}, opts);
  auto Cii = torch::zeros({n, ht*wd}, opts), wi = torch::zeros({n, ht*wd}, opts);
  shim_dim3 g; g.x = n;
  shim_launch(g, THREADS, [&]() { projective_transform_kernel(A4(targets, float), A4(weights, float), A2(poses, float), A3(disps, float),
      A1(intr, float), A1(ii, long), A1(jj, long), A4(Hs, float), A3(vs, float), A3(Eii, float), A3(Eij, float), A2(Cii, float), A2(wi, float)); });
  return {Hs, vs, Eii, Eij, Cii, wi};
}
// bare launches of the Schur-path kernels (geometry of droid_kernels.cu:966-970, 1264-1275, 1384-1388)
torch::Tensor accum(torch::Tensor data, torch::Tensor ptrs, torch::Tensor idxs) {
  auto out = torch::zeros({ptrs.size(0) - 1, data.size(1)}, data.options());
  shim_dim3 g; g.x = ptrs.size(0) - 1;
  shim_launch(g, THREADS, [&]() { accum_kernel(A2(data, float), A1(ptrs, long), A1(idxs, long), A2(out, float)); });
  return out;
}
torch::Tensor EEt6x6(torch::Tensor E, torch::Tensor Q, torch::Tensor idx) {
  auto S = torch::zeros({idx.size(0), 6, 6}, E.options());
  shim_dim3 g; g.x = idx.size(0);
  shim_launch(g, THREADS, [&]() { EEt6x6_kernel(A3(E, float), A2(Q, float), A2(idx, long), A3(S, float)); });
  return S;
}
torch::Tensor Ev6x1(torch::Tensor E, torch::Tensor Q, torch::Tensor w, torch::Tensor idx) {
  auto v = torch::zeros({idx.size(0), 6}, E.options());
  shim_dim3 g; g.x = idx.size(0);
  shim_launch(g, THREADS, [&]() { Ev6x1_kernel(A3(E, float), A2(Q, float), A2(w, float), A2(idx, long), A2(v, float)); });
  return v;
}
torch::Tensor EvT6x1(torch::Tensor E, torch::Tensor x, torch::Tensor idx) {
  auto dw = torch::zeros({idx.size(0), E.size(2)}, E.options());
  shim_dim3 g; g.x = idx.size(0);
  shim_launch(g, THREADS, [&]() { EvT6x1_kernel(A3(E, float), A2(x, float), A1(idx, long), A2(dw, float)); });
  return dw;
}
""" + retr_wrappers
    return _load("pvo_ref_droid_" + tag + ("_" + retr if retr else ""), cpp,
                 ["frame_distance", "projmap", "depth_filter", "iproj", "ba_assemble", "accum", "EEt6x6", "Ev6x1", "EvT6x1"] +
                 (["pose_retr", "disp_retr"] if retr is not None else []),
                 list(cflags) + ["-pthread"])


def _accum(m, data, ix, jx):
    """accum_cuda (droid_kernels.cu:927-977): the host-side CSR construction restated, the kernel is the reference's"""
    ix, jx = ix.tolist(), jx.tolist()
    inds = sorted(range(len(ix)), key=lambda n: ix[n])        # stable; torch::argsort's tie order only permutes fp32 addends
    ptrs, cols, i = [0], [], 0
    for j in range(len(jx)):
        while i < len(ix) and ix[inds[i]] <= jx[j]:
            if ix[inds[i]] == jx[j]:
                cols.append(inds[i])
            i += 1
        ptrs.append(len(cols))
    return m.accum(data.contiguous(), torch.tensor(ptrs, dtype=torch.int64), torch.tensor(cols, dtype=torch.int64))


def ref_ba_step(m, poses, disps, intr, targets, weights, eta, ii, jj, t0, t1, lm, ep):
    """ONE iteration of ba_cuda (droid_kernels.cu:1293-1410, full BA branch) up to dx and dz: the HOST code restated here
    (edge-list augmentation :1314-1322, SparseBlock assembly with duplicates summed in fp64 :1109-1153, schur_block's
    triple loop :1222-1246, damping + solve :1170-1194 with a dense fp64 Cholesky in place of Eigen's sparse LLT), every
    KERNEL the reference's own text.  Retraction (pose_retr_kernel reads xi[45], out of bounds) is not run."""
    P, num = t1 - t0, ii.shape[0]
    ht, wd = disps.shape[1:]
    HW = ht * wd
    ts = torch.arange(t0, t1)
    ii_exp, jj_exp = torch.cat([ts, ii]), torch.cat([ts, jj])
    kx, kk_exp = torch.unique(ii_exp, sorted=True, return_inverse=True)
    Hs, vs, Eii, Eij, Cii, wi = m.ba_assemble(targets, weights, poses, disps, intr, ii, jj)

    def lhs(As, bi, bj):
        A = np.zeros((6 * P, 6 * P), np.float64)
        for n in range(bi.shape[0]):
            i, j = int(bi[n]), int(bj[n])
            if i >= 0 and j >= 0:
                A[6 * i:6 * i + 6, 6 * j:6 * j + 6] += As[n].double().numpy()
        return A

    def rhs(bs, bi):
        b = np.zeros(6 * P, np.float64)
        for n in range(bi.shape[0]):
            i = int(bi[n])
            if i >= 0:
                b[6 * i:6 * i + 6] += bs[n].double().numpy()
        return b

    A = lhs(Hs.reshape(-1, 6, 6), torch.cat([ii, ii, jj, jj]) - t0, torch.cat([ii, jj, ii, jj]) - t0)
    b = rhs(vs.reshape(-1, 6), torch.cat([ii, jj]) - t0)
    C, w = _accum(m, Cii, ii, kx), _accum(m, wi, ii, kx)
    Q = 1.0 / (C + eta.view(-1, HW))
    Ei = _accum(m, Eii.view(num, 6 * HW), ii, ts).view(P, 6, HW)
    E = torch.cat([Ei, Eij], 0).contiguous()
    # schur_block
    graph, index = [[] for _ in range(P)], [[] for _ in range(P)]
    for n in range(ii_exp.shape[0]):
        j, k = int(jj_exp[n]), int(kk_exp[n])
        if t0 <= j < t1:                          # the text has `j <= t1` (:1228): j == t1 would index past `graph`
            graph[j - t0].append(k)
            index[j - t0].append(n)
    il, jl, idx = [], [], []
    for i in range(P):
        for j in range(P):
            for k in range(len(graph[i])):
                for l in range(len(graph[j])):
                    if graph[i][k] == graph[j][l]:
                        il.append(i); jl.append(j); idx.append([index[i][k], index[j][l], graph[i][k]])
    S = m.EEt6x6(E, Q.contiguous(), torch.tensor(idx, dtype=torch.int64).view(-1, 3))
    v = m.Ev6x1(E, Q.contiguous(), w.contiguous(), kk_exp.view(-1, 1).contiguous())
    SA = lhs(S, torch.tensor(il), torch.tensor(jl))
    Sb = rhs(v, jj_exp - t0)
    L = A - SA
    L[np.diag_indices(6 * P)] += ep + lm * np.diag(L)
    try:
        c = np.linalg.cholesky(L)
        x = np.linalg.solve(c.T, np.linalg.solve(c, b - Sb))
    except np.linalg.LinAlgError:
        x = np.zeros(6 * P)
    dx = torch.from_numpy(x.reshape(P, 6)).float()
    dw = m.EvT6x1(E, dx.contiguous(), (jj_exp - t0).contiguous())
    dz = Q * (w - _accum(m, dw, ii_exp, kx))
    return dict(dx=dx.numpy(), dz=dz.numpy(), kx=kx.numpy(), sysA=(A - SA), sysb=(b - Sb), Q=Q.numpy(), w=w.numpy())


def geom_scene(seed, P=7, ht=9, wd=13):
    """poses (w2c, t + xyzw quaternion), smooth disparities, intrinsics: a camera moving forward and sideways"""
    g = np.random.default_rng(seed)
    poses = np.zeros((P, 7), np.float32)
    for k in range(P):
        ang = 0.03 * k * np.array([0.3, 1.0, -0.2])
        th = np.linalg.norm(ang)
        q = np.concatenate([np.sin(th / 2) * ang / max(th, 1e-12), [np.cos(th / 2)]]) if th > 0 else np.array([0, 0, 0, 1.0])
        poses[k, :3] = np.array([0.08 * k, -0.01 * k, 0.05 * k]) + g.normal(0, 0.005, 3)
        poses[k, 3:] = q
    disps = g.uniform(0.2, 1.2, (P, ht, wd)).astype(np.float32)
    disps = (disps + np.roll(disps, 1, 1) + np.roll(disps, 1, 2)) / 3.0
    intr = np.array([wd * 0.8, wd * 0.8, wd / 2.0, ht / 2.0], np.float32)
    return poses, disps.astype(np.float32), intr


def retr_cases():
    """poses and tangent updates for the retraction: the identity update, an update below expSE3's theta > 1e-4 test, one inside
    expSO3's theta^2 < 1e-8 series branch, moderate updates, one near pi, and one whose sixth component dominates (where the
    xi[45] read matters most)"""
    g = np.random.default_rng(71)
    P = 12
    poses, _, _ = geom_scene(72, P, 4, 4)
    poses[:, :3] += g.normal(0, 0.5, (P, 3)).astype(np.float32)
    dx = g.normal(0, 0.15, (P, 6)).astype(np.float32)
    dx[0] = 0.0
    dx[1, 3:] = np.array([2e-5, -3e-5, 1e-5], np.float32)
    dx[2, 3:] = np.array([4e-5, 5e-5, -2e-5], np.float32)            # theta^2 = 4.5e-9 < 1e-8
    dx[3, 3:] = np.array([1.8, -2.2, 1.1], np.float32)               # |phi| = 3.05
    dx[4, 3:] = np.array([0.001, -0.002, 0.9], np.float32)
    dx[5, :3] = 0.0
    return poses.astype(np.float32), dx


def gen_geom_kernels():
    m = _droid_module("plain", ["-ffp-contract=off"])
    mf = _droid_module("fma", ["-mfma", "-ffp-contract=fast"])      # nvcc's default -fmad=true (as for the lookup's fixture)
    retr = {"xi45_zero": _droid_module("plain", ["-ffp-contract=off"], retr="xi45_zero"),
            "xi5": _droid_module("plain", ["-ffp-contract=off"], retr="xi5"),
            "xi5_fma": _droid_module("fma", ["-mfma", "-ffp-contract=fast"], retr="xi5")}
    out = {}
    for name, seed, P, ht, wd in (("a", 51, 7, 9, 13), ("b", 52, 5, 18, 16)):
        poses, disps, intr = geom_scene(seed, P, ht, wd)
        if name == "b":
            disps[1, :6] *= 8.0      # near surface: points that project outside / behind, some frame pairs fall under 0.75 valid
            poses[3, :3] += np.array([0.0, 0.0, 2.5], np.float32)
        ii, jj = np.meshgrid(np.arange(P), np.arange(P), indexing="ij")
        ii, jj = ii.reshape(-1).astype(np.int64), jj.reshape(-1).astype(np.int64)
        tp, td, ti = torch.from_numpy(poses), torch.from_numpy(disps), torch.from_numpy(intr)
        tii, tjj = torch.from_numpy(ii), torch.from_numpy(jj)
        out[name + "_poses"], out[name + "_disps"], out[name + "_intr"] = poses, disps, intr
        out[name + "_ii"], out[name + "_jj"] = ii, jj
        # every output twice: as the text reads with every rounding explicit (no suffix) and with the multiply-adds contracted
        # the way nvcc does by default ("_fma"; g++ -mfma -ffp-contract=fast chooses the contractions, not nvcc: see the header)
        for mod, sfx in ((m, ""), (mf, "_fma")):
            for beta in (0.3, 1.0, 0.0):
                out[name + "_frame_distance_beta%g" % beta + sfx] = mod.frame_distance(tp, td, ti, tii, tjj, beta).numpy()
            co, va = mod.projmap(tp, td, ti, tii, tjj)
            out[name + "_projmap_coords" + sfx], out[name + "_projmap_valid" + sfx] = co.numpy(), va.numpy()
            out[name + "_iproj" + sfx] = mod.iproj(tp, td, ti).numpy()
            ix = np.arange(P, dtype=np.int64)
            for t in (0.005, 0.05):
                th = np.full(P, t, np.float32)
                out[name + "_depth_filter_t%g" % t + sfx] = mod.depth_filter(tp, td, ti, torch.from_numpy(ix), torch.from_numpy(th)).numpy()
    # the retraction (pose_retr_kernel, droid_kernels.cu:877-910) on its own inputs, in both defined readings of xi[45]
    rp, rdx = retr_cases()
    out["retr_poses"], out["retr_dx"] = rp, rdx
    for tag, mod in retr.items():
        out["retr_out_" + tag] = mod.pose_retr(torch.from_numpy(rp), torch.from_numpy(rdx), 0, len(rp)).numpy()
    np.savez_compressed(os.path.join(HERE, "geom_kernels.npz"), **out)
    print("retraction: xi[45] = 0 against xi[5]: max |pose difference| %.3g; fma changes %d of %d values" % (
        np.abs(out["retr_out_xi45_zero"] - out["retr_out_xi5"]).max(), int((out["retr_out_xi5"] != out["retr_out_xi5_fma"]).sum()), out["retr_out_xi5"].size))
    print("contraction changes: projmap %d of %d values, iproj %d, depth_filter counts %d (case b)" % (
        int((out["b_projmap_coords"] != out["b_projmap_coords_fma"]).sum()), out["b_projmap_coords"].size,
        int((out["b_iproj"] != out["b_iproj_fma"]).sum()),
        int((out["b_depth_filter_t0.05"] != out["b_depth_filter_t0.05_fma"]).sum())))
    print("geom_kernels.npz:", {k: v.shape for k, v in out.items() if "frame_distance_beta0.3" in k or "depth_filter_t0.05" in k},
          "fd>=1000:", int((out["b_frame_distance_beta0.3"] >= 1000).sum()),
          "filter counts:", np.unique(out["b_depth_filter_t0.05"]).tolist())

    # one BA assembly (projective_transform_kernel) on a small window
    outb = {}
    for name, seed, P, ht, wd in (("a", 61, 5, 8, 10), ("b", 62, 6, 17, 16)):
        poses, disps, intr = geom_scene(seed, P, ht, wd)
        g = np.random.default_rng(seed + 100)
        e = [(i, j) for i in range(P) for j in range(P) if 0 < abs(i - j) <= 2]
        ii = np.array([a for a, _ in e], np.int64)
        jj = np.array([b for _, b in e], np.int64)
        if name == "b":
            disps[2, 3:9, 2:7] = 14.0      # Z = 1 + disp * t_z < MIN_DEPTH in the target frames behind frame 2
        tp, td, ti = torch.from_numpy(poses), torch.from_numpy(disps), torch.from_numpy(intr)
        co, _ = m.projmap(tp, td, ti, torch.from_numpy(ii), torch.from_numpy(jj))
        targets = (co[..., :2].permute(0, 3, 1, 2) + torch.from_numpy(g.normal(0, 0.3, (len(e), 2, ht, wd)).astype(np.float32))).contiguous()
        weights = torch.from_numpy(g.uniform(0, 1, (len(e), 2, ht, wd)).astype(np.float32))
        res = m.ba_assemble(targets, weights, tp, td, ti, torch.from_numpy(ii), torch.from_numpy(jj))
        for k, v in (("poses", poses), ("disps", disps), ("intr", intr), ("ii", ii), ("jj", jj),
                     ("targets", targets.numpy()), ("weights", weights.numpy())):
            outb[name + "_" + k] = v
        for k, v in zip(("Hs", "vs", "Eii", "Eij", "Cii", "bz"), res):
            outb[name + "_" + k] = v.numpy()
        for k, v in zip(("Hs", "vs", "Eii", "Eij", "Cii", "bz"), mf.ba_assemble(targets, weights, tp, td, ti, torch.from_numpy(ii), torch.from_numpy(jj))):
            outb[name + "_" + k + "_fma"] = v.numpy()
        # the whole step (t0 = 1: frame 0 fixed; edges out of frame 0 make it a depth frame outside the pose window)
        eta = torch.from_numpy(g.uniform(1e-3, 2e-2, (P, ht, wd)).astype(np.float32))   # one row per depth frame 0..P-1
        step = ref_ba_step(m, tp, td, ti, targets, weights, eta, torch.from_numpy(ii), torch.from_numpy(jj), 1, P, 1e-4, 0.1)
        outb[name + "_eta"] = eta.numpy()
        for k, v in step.items():
            outb[name + "_step_" + k] = v
        step_f = ref_ba_step(mf, tp, td, ti, targets, weights, eta, torch.from_numpy(ii), torch.from_numpy(jj), 1, P, 1e-4, 0.1)
        outb[name + "_step_dx_fma"], outb[name + "_step_dz_fma"] = step_f["dx"], step_f["dz"]
        # ... and the end of the iteration (droid_kernels.cu:1393-1397): disp_retr_kernel on the depth frames kx, pose_retr_kernel
        # on poses t0 .. t1 - 1, from the text, in both readings of xi[45]
        tdx, tdz, tkx = torch.from_numpy(step["dx"]), torch.from_numpy(step["dz"]), torch.from_numpy(step["kx"])
        outb[name + "_step_disps"] = retr["xi5"].disp_retr(td, tdz.contiguous(), tkx).numpy()
        for tag in ("xi45_zero", "xi5"):
            outb[name + "_step_poses_" + tag] = retr[tag].pose_retr(tp, tdx.contiguous(), 1, P).numpy()
    np.savez_compressed(os.path.join(HERE, "ba_assemble_kernel.npz"), **outb)
    print("ba_assemble_kernel.npz:", {k: v.shape for k, v in outb.items() if k.startswith("b_") and k[2:] in ("Hs", "vs", "Eii", "Cii")})


# ------------------------------------------------------------------------------------------------------------
# altcorr_kernel.cu: one 4 x 8 thread block = ONE WARP that hands data between its threads through shared memory with
# fewer barriers than OS threads need
# ------------------------------------------------------------------------------------------------------------
_ALTCORR_SHIM = r"""
#define __shared__ static
#define __syncthreads() shim_barrier()
struct ShimBarrier {
  std::mutex m; std::condition_variable cv; int count = 0, gen = 0, n = 1;
  void wait() { std::unique_lock<std::mutex> l(m); int g = gen; if (++count == n) { gen++; count = 0; cv.notify_all(); }
                else cv.wait(l, [&]{ return g != gen; }); }
};
static ShimBarrier shim_bar;
static inline void shim_barrier() { shim_bar.wait(); }
// The 32 threads of a block are one warp, and the text leans on its lock step in four places: (1) the f2 tile of the next
// tap is stored right after the dot products of this one (one __syncthreads per tap, none behind the reads); (2) the f1 tile
// of the next channel slab likewise; (3) x2s / y2s are read by all threads right after each thread stored its own entry;
// (4) backward: f2_grad is accumulated by columns, read by rows and zeroed again without a barrier.  OS threads are not in
// lock step, so the emulation adds rendezvous where EVERY thread of the block passes the same number of times whatever its
// pixel: every call of floor() (each of the four hand-overs is followed by one before the next access), and every
// ASSIGNMENT to a scalar_t lvalue (the tile stores; the initialisations `scalar_t s = 0.0` are constructions, the
// accumulations are +=, neither waits).  scalar_t is therefore not float but a struct around one: same arithmetic, float
// operations on float values.
static inline float shim_floor(float x) { shim_barrier(); return std::floor(x); }
struct Sh {
  float v;
  Sh() = default;
  Sh(const Sh&) = default;
  Sh(float x) : v(x) {}
  Sh(double x) : v(static_cast<float>(x)) {}
  Sh(int x) : v(static_cast<float>(x)) {}
  Sh& operator=(const Sh& o) { shim_barrier(); v = o.v; return *this; }
  Sh& operator+=(const Sh& o) { v += o.v; return *this; }
  explicit operator int() const { return static_cast<int>(v); }
};
static inline Sh operator*(const Sh& a, const Sh& b) { return Sh(a.v * b.v); }
static inline Sh operator-(const Sh& a, const Sh& b) { return Sh(a.v - b.v); }
static inline Sh shim_floor_any(const Sh& a) { return Sh(shim_floor(a.v)); }
static inline float shim_floor_any(float x) { return shim_floor(x); }
#define floor(x) shim_floor_any(x)
static std::mutex shim_atomic_mutex;     // threads of one block do add to the same address (two pixels, one target)
static inline void atomicAdd(Sh* p, const Sh& x) { std::lock_guard<std::mutex> l(shim_atomic_mutex); p->v += x.v; }
template <typename F> static void shim_launch_4x8(int gx, int gy, int gz, F body) {
  shim_bar.n = 32;
  for (int bx = 0; bx < gx; bx++) for (int by = 0; by < gy; by++) for (int bz = 0; bz < gz; bz++) {
    std::vector<std::thread> pool;
    for (int t = 0; t < 32; t++) pool.emplace_back([=]() {
      blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz; blockDim.x = 4; blockDim.y = 8; threadIdx.x = t / 8; threadIdx.y = t % 8;
      body();
    });
    for (auto &th : pool) th.join();
    shim_bar.count = 0;
  }
}
template <int N> static torch::PackedTensorAccessor32<Sh, N, torch::DefaultPtrTraits> sh_acc(torch::Tensor t) {
  TORCH_CHECK(t.scalar_type() == torch::kFloat && t.dim() == N);
  return torch::PackedTensorAccessor32<Sh, N, torch::DefaultPtrTraits>(reinterpret_cast<Sh*>(t.data_ptr<float>()), t.sizes().data(), t.strides().data());
}
"""


def _altcorr_module(tag, cflags):
    text = _lines(os.path.join(SRC, "altcorr_kernel.cu"), 16, 286)      # the four #defines, within_bounds, both kernels
    cpp = _COMMON + _ALTCORR_SHIM + text + r"""
// host wrappers restated from altcorr_kernel.cu:288-356 (allocation + launch geometry)
torch::Tensor forward(torch::Tensor fmap1, torch::Tensor fmap2, torch::Tensor coords, int64_t radius) {
  const int B = coords.size(0), N = coords.size(1), H = coords.size(2), W = coords.size(3), rd = 2 * radius + 1;
  auto corr = torch::zeros({B, N, rd * rd, H, W}, fmap1.options());
  auto a1 = sh_acc<4>(fmap1), a2 = sh_acc<4>(fmap2);
  auto co = coords.packed_accessor32<float,5,torch::DefaultPtrTraits>();
  auto out = sh_acc<5>(corr);
  shim_launch_4x8(B, (H + BLOCK_H - 1) / BLOCK_H, (W + BLOCK_W - 1) / BLOCK_W, [&]() { altcorr_forward_kernel<Sh>(a1, a2, co, out, (int)radius); });
  return corr;
}
std::vector<torch::Tensor> backward(torch::Tensor fmap1, torch::Tensor fmap2, torch::Tensor coords, torch::Tensor corr_grad, int64_t radius) {
  const int B = coords.size(0), N = coords.size(1), H1 = fmap1.size(1), W1 = fmap1.size(2);
  // the text reads coords[b][n][h1][w1] for EVERY thread of a block, pixels beyond the image included (:207-208)
  TORCH_CHECK(H1 % BLOCK_H == 0 && W1 % BLOCK_W == 0, "backward: whole blocks only");
  auto g1 = torch::zeros_like(fmap1), g2 = torch::zeros_like(fmap2);
  auto gc = torch::zeros({B, N, H1, W1, 2}, fmap1.options());
  auto a1 = sh_acc<4>(fmap1), a2 = sh_acc<4>(fmap2), o1 = sh_acc<4>(g1), o2 = sh_acc<4>(g2);
  auto co = sh_acc<5>(coords), cg = sh_acc<5>(corr_grad), og = sh_acc<5>(gc);
  shim_launch_4x8(B, (H1 + BLOCK_H - 1) / BLOCK_H, (W1 + BLOCK_W - 1) / BLOCK_W, [&]() { altcorr_backward_kernel<Sh>(a1, a2, co, cg, o1, o2, og, (int)radius); });
  return {g1, g2, gc};
}
"""
    return _load("pvo_ref_altcorr_" + tag, cpp, ["forward", "backward"], list(cflags) + ["-pthread"])


def gen_altcorr_kernel():
    plain = _altcorr_module("plain", ["-ffp-contract=off"])
    fused = _altcorr_module("fma", ["-mfma", "-ffp-contract=fast"])
    g = np.random.default_rng(61)
    out = {}
    #        name B  S  H1 W1 H2 W2  C  r  spread
    cases = (("a", 1, 2, 5, 9, 6, 7, 64, 3, 2.5),      # ragged: 2 x 2 blocks, three of them partly outside the image
             ("b", 2, 1, 4, 8, 3, 5, 32, 3, 6.0),      # window larger than the plane; one channel slab
             ("c", 1, 2, 4, 8, 7, 9, 96, 2, 1.5))      # radius 2, three slabs
    for name, B, S, H1, W1, H2, W2, C, r, spread in cases:
        f1 = g.standard_normal((B, H1, W1, C)).astype(np.float32)
        f2 = g.standard_normal((B, H2, W2, C)).astype(np.float32)
        base = np.stack(np.meshgrid(np.arange(W1), np.arange(H1)), -1).astype(np.float32)          # [H1,W1,2] (x, y)
        co = base[None, None] * np.array([W2 / W1, H2 / H1], np.float32) + g.uniform(-spread, spread, (B, S, H1, W1, 2)).astype(np.float32)
        co = co.astype(np.float32)
        co[0, 0, 0, 0] = (1.0, 2.0)                    # integer coordinates: dx = dy = 0
        co[0, 0, 0, 1] = (-20.0, 3.5)                  # whole window outside
        rd = 2 * r + 1
        t = torch.from_numpy
        out[name + "_fmap1"], out[name + "_fmap2"], out[name + "_coords"], out[name + "_radius"] = f1, f2, co, np.int64(r)
        out[name + "_fwd_fma"] = fused.forward(t(f1), t(f2), t(co), r).numpy()
        out[name + "_fwd_nofma"] = plain.forward(t(f1), t(f2), t(co), r).numpy()
        if H1 % 4 == 0 and W1 % 8 == 0:
            cg = g.standard_normal((B, S, rd * rd, H1, W1)).astype(np.float32)
            g1, g2, gc = fused.backward(t(f1), t(f2), t(co), t(cg), r)
            assert not gc.any()                          # coords_grad is allocated and never written (altcorr_kernel.cu:340)
            out[name + "_grad"], out[name + "_bwd_fmap1_fma"], out[name + "_bwd_fmap2_fma"] = cg, g1.numpy(), g2.numpy()
    np.savez_compressed(os.path.join(HERE, "altcorr_kernel.npz"), **out)
    print("altcorr_kernel.npz:", sorted(k for k in out if k.endswith("_fmap1")),
          "fma changes %d of %d forward values (case a)" % (int((out["a_fwd_fma"] != out["a_fwd_nofma"]).sum()), out["a_fwd_fma"].size))


if __name__ == "__main__":
    if not os.path.isdir(SRC):
        raise SystemExit("reference tree not present; fixtures can only be generated in the build container")
    which = sys.argv[1:] or ["corr", "geom", "altcorr"]
    if "corr" in which:
        gen_corr_lookup_kernel()
    if "geom" in which:
        gen_geom_kernels()
    if "altcorr" in which:
        gen_altcorr_kernel()
