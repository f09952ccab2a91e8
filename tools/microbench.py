"""Kernel-level timings on the GPU box (not the contract bench; see bench.py).
usage: python tools/microbench.py [lookup] [ba] ..."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pvo_amd import droid_backends as db


def timeit(fn, iters=50, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def lookup():
    dev = torch.device("cuda:0")
    for dt in (torch.float16, torch.float32):
        N, H, W = 36, 48, 64
        pyr = [torch.randn(N, H, W, H >> l, W >> l, device=dev, dtype=dt) for l in range(4)]
        base = torch.stack(torch.meshgrid(torch.arange(W), torch.arange(H), indexing="xy"), -1).float().to(dev)
        coords = (base[None] + torch.randn(N, H, W, 2, device=dev) * 4).contiguous()
        us = timeit(lambda: db.corr_pyramid_lookup(pyr, coords, 3))
        s = pyr[0].element_size()
        alg = N * H * W * (4 * 64 * s + 8 + 196 * s)
        print(f"lookup4 {dt} E={N} {H}x{W}: {us:.1f} us  alg {alg/1e6:.1f} MB -> {alg/us/1e6:.3f} TB/s")
        # per-level reference-style calls
        cp = coords.permute(0, 3, 1, 2).contiguous()
        us1 = timeit(lambda: [db.corr_index_forward(pyr[l], cp / 2 ** l, 3) for l in range(4)])
        print(f"  4x corr_index_forward: {us1:.1f} us")


def ba():
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from test_geom_ba_gpu import _scene
    dev = torch.device("cuda:0")
    s = _scene(0, 8, 48, 64, 3, 1)
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in s.items()}
    poses0, disps0 = d["poses"].clone(), d["disps"].clone()
    def run(iters):
        poses, disps = poses0.clone(), disps0.clone()
        db.ba(poses, disps, d["intr"], d["target"], d["weight"], d["eta"], d["ii"], d["jj"], 1, 8, iters, 1e-4, 0.1, False)
    for it in (1, 2, 8):
        us = timeit(lambda: run(it), iters=30)
        print(f"ba S-B E=36 P=7 48x64 iters={it}: {us:.1f} us per call ({us/it:.1f} us/iter incl. clone+plan)")
    ii2, jj2 = d["jj"], d["ii"]
    us = timeit(lambda: db.frame_distance(poses0, disps0, d["intr"], d["ii"], d["jj"], 0.3))
    print(f"frame_distance M=36: {us:.1f} us")
    intr_all = d["intr"][None].repeat(8, 1).contiguous()
    us = timeit(lambda: db.reproject(poses0, disps0, intr_all, d["ii"], d["jj"]))
    print(f"reproject E=36: {us:.1f} us")


def build():
    dev = torch.device("cuda:0")
    N, C, H, W = 36, 128, 48, 64
    f1 = torch.randn(N, H, W, C, device=dev).half(); f2 = torch.randn(N, H, W, C, device=dev).half()
    us = timeit(lambda: db.corr_build(f1, f2, 4, channels_last=True), iters=20)
    byts = N * (2 * H * W * C * 2 + (H * W) ** 2 * 2 * (1 + .25 + 1 / 16 + 1 / 64))
    print(f"corr_build fp16 E={N}: {us:.1f} us  {N*2*(H*W)**2*C/us/1e6:.1f} TFLOP/s  alg {byts/1e6:.0f} MB -> {byts/us/1e6:.2f} TB/s")
    def ref():
        a = (f1.view(N, H * W, C) / 4.0) @ (f2.view(N, H * W, C) / 4.0).transpose(1, 2)
        a = a.view(N * H * W, 1, H, W); out = [a]
        for i in range(3):
            a = torch.nn.functional.avg_pool2d(a, 2, stride=2); out.append(a)
        return out
    us = timeit(ref, iters=10)
    print(f"  torch matmul + 3x avg_pool2d (reference formulation on this GPU): {us:.1f} us")


if __name__ == "__main__":
    which = sys.argv[1:] or ["lookup"]
    for w in which:
        globals()[w]()
