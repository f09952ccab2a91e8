"""Workload for the counter passes of tools/pmc_kernels.sh: the wide convolution as the ConvGRU gate launch, the fused
lookup, and the dense BA, each at the S-B shape, a few dispatches each (counters are read per dispatch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from pvo_amd import droid_backends as db

dev = torch.device("cuda:0")
E, H, W = 36, 48, 64
cl = torch.channels_last
g = torch.Generator(device=dev).manual_seed(0)
net = torch.tanh(torch.randn(E, 128, H, W, device=dev, generator=g)).half().contiguous(memory_format=cl)
cf = torch.relu(torch.randn(E, 192, H, W, device=dev, generator=g)).half().contiguous(memory_format=cl)
gg = torch.randn(E, 384, device=dev, generator=g)
P_zr = torch.randn(E, 256, H, W, device=dev, generator=g).half().contiguous(memory_format=cl)
tzr = db.conv3x3_weights((torch.randn(256, 320, 3, 3, device=dev, generator=g) * 0.02).half(), torch.half)
for _ in range(6):
    db.gru_conv_gates(net, cf, tzr, gg, P_zr)
torch.cuda.synchronize()
# the heads' first stage (128 -> 4 x 128 with the second stage in its epilogue): the <pvo_half, true> instantiation
w1 = db.conv3x3_weights((torch.randn(512, 128, 3, 3, device=dev, generator=g) * 0.03).half(), torch.half)
b1 = torch.randn(512, device=dev, generator=g) * 0.1
w2 = db.heads2_fragments(torch.randn(4, 2, 9, 128, device=dev, generator=g) * 0.05, torch.half)
b2 = torch.randn(8, device=dev, generator=g)
for _ in range(6):
    db.heads_fused(net, w1, b1, w2, b2)
torch.cuda.synchronize()
# the fused lookup (lookup + corr_encoder[0]) on the resident tiled pool, Infinity Cache evicted between launches
from pvo_amd.modules.corr import CorrVolumePool
pool = CorrVolumePool(E, H, W, dev)
pool.add(torch.randn(E, H, W, 128, device=dev, generator=g).half(), torch.randn(E, H, W, 128, device=dev, generator=g).half())
base = torch.stack(torch.meshgrid(torch.arange(W), torch.arange(H), indexing="xy"), -1).float().to(dev)
coords = (base[None] + torch.randn(E, H, W, 2, device=dev, generator=g) * 4).contiguous()
we = db.corr_encoder_weights(torch.randn(128, 196, 1, 1, device=dev, generator=g) * 0.05, torch.half)
be = torch.randn(128, device=dev, generator=g)
flush = torch.empty(600 * 1024 * 1024, device=dev, dtype=torch.uint8)
for _ in range(6):
    flush.zero_()
    pool.encoded(coords[None], we, be)
    torch.cuda.synchronize()
del flush
from test_geom_ba_gpu import _scene
s = _scene(0, 8, H, W, 3, 1)
d = lambda t: t.to(dev)
for _ in range(4):
    poses, disps = d(s["poses"].clone()), d(s["disps"].clone())
    db.ba(poses, disps, d(s["intr"]), d(s["target"]), d(s["weight"]), d(s["eta"]), d(s["ii"]), d(s["jj"]), 1, 8, 2, 1e-4, 0.1, False)
torch.cuda.synchronize()
# the BA of a real frontend update (round 6): 26 poses, 440 edges (every inactive pair four times + 48 active), 30 x 101 maps -
# ba_schur_mfma_kernel<false, 512> with its row passes / merged targets, ba_solve_dense_kernel<14>
from test_geom_ba_gpu import _frontend_window
s = _frontend_window()
for _ in range(4):
    poses, disps = d(s["poses"].clone()), d(s["disps"].clone())
    db.ba(poses, disps, d(s["intr"]), d(s["target"]), d(s["weight"]), d(s["eta"]), d(s["ii"]), d(s["jj"]), 1, 26, 2, 1e-4, 0.1, False)
torch.cuda.synchronize()
print("done")
