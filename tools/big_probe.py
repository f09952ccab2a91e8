"""Where does conv3x3_big_kernel's time go?  Runs the probe build (tools/_probe/libbig_probe.so, -DPVO_PROBE_BIG) of the
gate convolution at the bench's shape and prints, from the per-workgroup shader-clock stamps: residency per CU (how many
workgroups overlap in time on one CU), cycles per K step under that residency, prologue / epilogue cycles.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DPVO_PROBE_BIG -shared pvo_amd/csrc/conv_small.hip \
          -o tools/_probe/libbig_probe.so
    gpurun -- python tools/big_probe.py
"""
import ctypes, collections, os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_probe", sys.argv[1] if len(sys.argv) > 1 else "libbig_probe.so"))
dev = torch.device("cuda:0")
E, H, W, Cin = 36, 48, 64, 320
vp = ctypes.c_void_p
g = torch.Generator().manual_seed(0)
X = torch.randn(E, H, W, Cin, generator=g).half().to(dev)
wt = (torch.randn(9, 256, Cin, generator=g) * 0.02).half().to(dev)
gg = torch.randn(E, 384, generator=g).to(dev)
P = torch.randn(E, H, W, 256, generator=g).half().to(dev)
net = torch.randn(E, H, W, 128, generator=g).half().to(dev)
Z = torch.empty(E, H, W, 128, dtype=torch.half, device=dev)
RN = torch.empty_like(Z)
f = lib.pvo_gru_conv_gates
f.restype = ctypes.c_int
args = (vp(X.data_ptr()), vp(wt.data_ptr()), vp(gg.data_ptr()), vp(P.data_ptr()), vp(net.data_ptr()), vp(Z.data_ptr()),
        vp(RN.data_ptr()), E, H, W, Cin, 1, vp(0))
for _ in range(3):
    assert f(*args) == 0
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    f(*args)
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) / 20 * 1e3
flop = 2.0 * E * H * W * 9 * Cin * 256
print("gates 320->256: %.1f us back to back = %.3f PFLOP/s" % (us, flop / us / 1e9))

n_wg = 4 * 2 * 3 * E
buf = np.zeros(n_wg * 8, dtype=np.uint64)
assert lib.pvo_big_probe_read(vp(buf.ctypes.data), n_wg * 8) == 0
b = buf.reshape(n_wg, 8).astype(np.int64)
start, loop_end, end, hw, xcc = b[:, 0], b[:, 1], b[:, 2], b[:, 3], b[:, 4] & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
S = Cin // 32 * 9
print("workgroups", n_wg, "| cycles: total median %d, main loop median %d (%.0f per step), epilogue median %d"
      % (np.median(end - start), np.median(loop_end - start), np.median(loop_end - start) / S, np.median(end - loop_end)))
key = xcc * 4096 + se * 256 + sh * 16 + cu
groups = collections.defaultdict(list)
for i in range(n_wg):
    groups[int(key[i])].append(i)
print("distinct CUs seen:", len(groups), "| workgroups per CU: min %d max %d" % (min(map(len, groups.values())), max(map(len, groups.values()))))
peak = []
for k, ids in groups.items():
    ev = sorted([(start[i], 1) for i in ids] + [(end[i], -1) for i in ids])
    cur = best = 0
    for _, d in ev:
        cur += d
        best = max(best, cur)
    peak.append(best)
print("peak concurrent workgroups on a CU: histogram", dict(collections.Counter(peak)))
for x in range(8):
    m = xcc == x
    if m.any():
        print("  xcc %d: %d workgroups, span %d cycles, first start %d" % (x, m.sum(), end[m].max() - start[m].min(), start[m].min() - start.min()))
# steps while alone vs while sharing the CU: split workgroups by whether another one overlapped > 50 % of their life
share = np.zeros(n_wg)
for k, ids in groups.items():
    for i in ids:
        ov = 0
        for j in ids:
            if j != i:
                ov += max(0, min(end[i], end[j]) - max(start[i], start[j]))
        share[i] = ov / max(end[i] - start[i], 1)
for lo, hi in ((0, 0.25), (0.25, 0.75), (0.75, 1.5), (1.5, 9)):
    m = (share >= lo) & (share < hi)
    if m.any():
        print("  overlap %.2f-%.2f: %4d workgroups, %.0f cycles per step" % (lo, hi, m.sum(), np.median((loop_end - start)[m]) / S))
