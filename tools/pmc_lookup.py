"""Workload for the PMC passes: a known-size streaming copy (calibration) + the fused lookup at S-B size.
Run under:  rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv ...   and again with --pmc WRITE_SIZE"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pvo_amd import droid_backends as db
dev = torch.device("cuda:0")
N, H, W = 36, 48, 64
g = torch.Generator(device=dev).manual_seed(0)
pyr = [torch.randn(N, H, W, H >> l, W >> l, device=dev, generator=g).half() for l in range(4)]
base = torch.stack(torch.meshgrid(torch.arange(W), torch.arange(H), indexing="xy"), -1).float().to(dev)
coords = (base[None] + torch.randn(N, H, W, 2, device=dev, generator=g) * 4).contiguous()
big = torch.empty(512 * 1024 * 1024 // 4, device=dev, dtype=torch.float32).normal_()   # 512 MiB > Infinity Cache
dst = torch.empty_like(big)
for _ in range(3):
    dst.copy_(big)            # calibration: reads 512 MiB, writes 512 MiB (float4 vectorised copy kernel)
    torch.cuda.synchronize()
flush = torch.empty(600 * 1024 * 1024, device=dev, dtype=torch.uint8)
for _ in range(5):
    flush.zero_()             # evict the volume from the 256 MiB Infinity Cache between launches
    out = db.corr_pyramid_lookup(pyr, coords, 3)          # row-major planes (kernel <pvo_half, false>)
    torch.cuda.synchronize()
# the resident pool's 8x8-tiled planes (kernel <pvo_half, true>): what FactorGraph.update launches at this shape
from pvo_amd.modules.corr import CorrVolumePool
pool = CorrVolumePool(N, H, W, dev)
f1 = torch.randn(N, H, W, 128, device=dev, generator=g).half()
f2 = torch.randn(N, H, W, 128, device=dev, generator=g).half()
pool.add(f1, f2)
assert pool.tiled
for _ in range(5):
    flush.zero_()
    out = pool(coords[None], channels_last=True)
    torch.cuda.synchronize()
# ... and fused with the first correlation-encoder layer (kernel <pvo_half, true, true>): what the bench step launches
w = db.corr_encoder_weights(torch.randn(128, 196, 1, 1, device=dev, generator=g) * 0.05, torch.half)
bias = torch.randn(128, device=dev, generator=g)
for _ in range(5):
    flush.zero_()
    out = pool.encoded(coords[None], w, bias)
    torch.cuda.synchronize()
print("done", out.shape)
