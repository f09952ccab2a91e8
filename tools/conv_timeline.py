"""Per-workgroup timeline of the wide convolution kernel (conv3x3_big_kernel) from in-kernel clock stamps.

    python tools/conv_timeline.py --build        (here: hipcc -DPVO_CONV_PROBE -> tools/_probe/libpvo_hip.so)
    python tools/conv_timeline.py [E H W]        (GPU box: runs the gate convolution on the probe library)

Every workgroup records s_memtime at entry, when the main loop starts, when it ends and at exit, plus the XCC / CU it ran
on.  The script prints the share of a workgroup's life spent in prologue / main loop / epilogue, the dispatch gaps
between consecutive workgroups of a CU, and how full the chip was over the launch."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBE_DIR = os.path.join(ROOT, "tools", "_probe")
PROBE_LIB = os.path.join(PROBE_DIR, "libpvo_hip.so")

if "--build" in sys.argv:
    from pvo_amd import build
    build.build_hip()
    os.makedirs(PROBE_DIR, exist_ok=True)
    obj = os.path.join(PROBE_DIR, "conv_small.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + build.HIPCC_FLAGS + ["-DPVO_CONV_PROBE", "-c", os.path.join(build.CSRC, "conv_small.hip"), "-o", obj])
    objs = [obj if s == "conv_small.hip" else os.path.join(build.CSRC, s.replace(".hip", ".o")) for s in build.HIP_SOURCES]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROBE_LIB] + objs)
    print(PROBE_LIB)
    sys.exit(0)

import numpy as np
import torch
from pvo_amd import _lib
_lib.LIB_PATH = PROBE_LIB
from pvo_amd import droid_backends as db

dev = torch.device("cuda:0")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
E, H, W = (int(args[0]), int(args[1]), int(args[2])) if len(args) >= 3 else (36, 48, 64)
mode = "plain" if "--plain" in sys.argv else "gates"
cl = torch.channels_last
lib = _lib.load()
lib.pvo_debug_conv_probe.restype = ctypes.c_int
lib.pvo_debug_conv_probe.argtypes = [ctypes.c_void_p]

net = torch.tanh(torch.randn(E, 128, H, W, device=dev)).half().contiguous(memory_format=cl)
cf = torch.relu(torch.randn(E, 192, H, W, device=dev)).half().contiguous(memory_format=cl)
gg = torch.randn(E, 384, device=dev)
P_zr = torch.randn(E, 256, H, W, device=dev).half().contiguous(memory_format=cl)
tzr = db.conv3x3_weights((torch.randn(256, 320, 3, 3, device=dev) * 0.02).half(), torch.half)
x = torch.randn(E, 320, H, W, device=dev).half().contiguous(memory_format=cl)
run = (lambda: db.gru_conv_gates(net, cf, tzr, gg, P_zr)) if mode == "gates" else (lambda: db.conv3x3(x, tzr))
for _ in range(3):
    run()
torch.cuda.synchronize()
nwg = E * ((H + 15) // 16) * ((W + 15) // 16) * 2
buf = torch.zeros(nwg, 8, dtype=torch.int64, device=dev)
assert lib.pvo_debug_conv_probe(buf.data_ptr()) == 0
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(9):                 # the stamps kept are those of the LAST of ten back-to-back launches (clocks settled)
    run()
a.record(); run(); b.record()
torch.cuda.synchronize()
assert lib.pvo_debug_conv_probe(None) == 0
us = a.elapsed_time(b) * 1e3
t = buf.cpu().numpy().astype(np.int64)
t0, t1, t2, t3 = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
hw = t[:, 4] & 0xffffffff
xcc = (t[:, 4] >> 32) & 0xf
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 0x1
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
wc0 = t[:, 5]
# s_memtime ticks are shader cycles, and every XCD has its own counter: durations inside a workgroup come from it, the
# placement of workgroups on the common time axis from s_memrealtime (100 MHz)
tick = 1e-3
life = (t3 - t0) * tick
nsteps = 9 * (320 // 32)
print("%s E=%d %dx%d: %d workgroups, launch %.1f us by events; entries spread over %.1f us" % (mode, E, H, W, nwg, us, (wc0.max() - wc0.min()) * 0.01))
print("workgroup life: median %.1f kcycles (min %.1f max %.1f); prologue %.1f, main loop %.1f (%.0f cycles per step), epilogue %.1f (medians)"
      % (np.median(life), life.min(), life.max(), np.median(t1 - t0) * tick, np.median(t2 - t1) * tick,
         np.median(t2 - t1) / nsteps, np.median(t3 - t2) * tick))
for xc in np.unique(xcc)[:2]:
    m = xcc == xc
    dw = (wc0[m].max() - wc0[m].min()) * 10e-9
    if dw > 0:
        print("XCD %d shader clock during the launch: %.2f GHz (s_memtime vs s_memrealtime between first and last entry)" % (xc, (t0[m].max() - t0[m].min()) / dw / 1e9))
if mode == "gates":
    t6, t7 = t[:, 6], t[:, 7]
    print("gate epilogue (kcycles, medians): operands requested + accumulators through the slab %.1f | finish 2 x 4 pixels per thread %.1f | second half %.1f"
          % (np.median(t6 - t2) * tick, np.median(t7 - t6) * tick, np.median(t3 - t7) * tick))
cus = np.unique(cuid)
print("compute units seen: %d; workgroups per CU min %d max %d" % (len(cus), min((cuid == c).sum() for c in cus), max((cuid == c).sum() for c in cus)))
# the same on the common axis: entry times in units of 10 ns, life converted with the clock the launch averaged
order = np.argsort(wc0)
first = order[: 2 * len(cus)]
rest = order[2 * len(cus):]
print("first %d workgroups to enter: main loop median %.1f kcycles, epilogue %.1f; the rest: main loop %.1f, epilogue %.1f"
      % (len(first), np.median((t2 - t1)[first]) * tick, np.median((t3 - t2)[first]) * tick,
         np.median((t2 - t1)[rest]) * tick if len(rest) else 0, np.median((t3 - t2)[rest]) * tick if len(rest) else 0))
# dispatch gap: on each CU, time from an exit to the next entry after it
gaps = []
for c in cus:
    m = cuid == c
    ends = np.sort(t3[m]); starts = np.sort(t0[m])
    later = starts[2:]                      # the first two fill the empty CU
    for s, e in zip(later, ends):
        gaps.append((s - e) * tick)
if gaps:
    print("exit -> next entry on the same CU: median %.1f kcycles (min %.1f, max %.1f), %d samples" % (np.median(gaps), min(gaps), max(gaps), len(gaps)))
