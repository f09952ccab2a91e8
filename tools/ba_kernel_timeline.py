"""Where the BA's three data-parallel kernels spend their time: per-workgroup stamps of the constant-rate clock (10 ns) at the phases
of ba_assemble_kernel / ba_schur_mfma_kernel / ba_backsub_kernel at S-B (8 keyframes, 36 edges, 48x64).
    python tools/ba_kernel_timeline.py --build       (here: -DPVO_BA_PROBE=3 -> tools/_probe/libpvo_hip_bawg.so)
    python tools/ba_kernel_timeline.py               (GPU box)"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
PROBE_DIR = os.path.join(ROOT, "tools", "_probe")
PROBE_LIB = os.path.join(PROBE_DIR, "libpvo_hip_bawg.so")
if "--build" in sys.argv:
    from pvo_amd import build
    build.build_hip()
    os.makedirs(PROBE_DIR, exist_ok=True)
    obj = os.path.join(PROBE_DIR, "bawg.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + build.HIPCC_FLAGS + ["-DPVO_BA_PROBE=3"] + os.environ.get("PROBE_DEFS", "").split() + ["-c", os.path.join(build.CSRC, "ba.hip"), "-o", obj], stderr=subprocess.DEVNULL)
    objs = [obj if s == "ba.hip" else os.path.join(build.CSRC, s.replace(".hip", ".o")) for s in build.HIP_SOURCES]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROBE_LIB] + objs)
    print(PROBE_LIB); sys.exit(0)
import torch
from pvo_amd import _lib
_lib.LIB_PATH = PROBE_LIB
from pvo_amd import droid_backends as db
from test_geom_ba_gpu import _scene
dev = torch.device("cuda:0")
nf = int(os.environ.get("NF", "8"))
s = _scene(0, nf, int(os.environ.get("HT", "48")), int(os.environ.get("WD", "64")), int(os.environ.get("RAD", "3")), 1)      # RAD 8: frames with 16 neighbours (a frontend window with its inactive edges)
copies = int(os.environ.get("COPIES", "1"))      # COPIES 3, RAD 3: what a frontend window with its inactive edges looks like (450 edges, 18 per frame to 6 poses)
if copies > 1:
    s = dict(s, ii=s["ii"].repeat(copies), jj=s["jj"].repeat(copies), target=s["target"].repeat(copies, 1, 1, 1).contiguous(),
             weight=s["weight"].repeat(copies, 1, 1, 1).contiguous())
print("keyframes %d, edges %d, map %s x %s, radius %s" % (nf, s["ii"].shape[0], os.environ.get("HT", "48"), os.environ.get("WD", "64"), os.environ.get("RAD", "3")))
d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in s.items()}
if os.environ.get("BA_DUMP"):        # a BA of a real run (bench.py --sequence-only with PVO_BENCH_DUMP_BA=<file>): the frontend's window with its inactive edges
    z = torch.load(os.environ["BA_DUMP"])
    nfb = z["poses"].shape[0]
    eta = (0.2 * z["damping"][z["rows"]] + 1e-7).contiguous()
    d = dict(poses=z["poses"].to(dev), disps=z["disps"].to(dev), intr=z["intr"].to(dev), target=z["target"].to(dev).contiguous(), weight=z["weight"].to(dev).contiguous(),
             eta=eta.to(dev), ii=z["ii"].to(dev), jj=z["jj"].to(dev))
    t0_, nf = int(z["t0"]), int(z["t1"])
    print("real window: poses %d..%d, %d edges, %d depth frames, buffer %d" % (t0_, nf, d["ii"].shape[0], eta.shape[0], nfb))
else:
    t0_ = 1
lib = _lib.load()
lib.pvo_debug_ba_wg_probe.restype = ctypes.c_int; lib.pvo_debug_ba_wg_probe.argtypes = [ctypes.c_void_p]
buf = torch.zeros(3 * 4096 * 8, dtype=torch.int64, device=dev)
run = lambda it: db.ba(d["poses"].clone(), d["disps"].clone(), d["intr"], d["target"], d["weight"], d["eta"], d["ii"], d["jj"], t0_, nf, it, 1e-4, 0.1, False)
for _ in range(5):
    run(2)
torch.cuda.synchronize()
assert lib.pvo_debug_ba_wg_probe(buf.data_ptr()) == 0
run(1); torch.cuda.synchronize()
assert lib.pvo_debug_ba_wg_probe(None) == 0
t = buf.cpu().numpy().reshape(3, 4096, 8).astype(np.int64)
def show(name, k, labels, last):
    a = t[k]; live = a[:, 0] > 0
    a = a[live]
    if not len(a):
        print(name, "no stamps"); return
    t0 = a[:, 0].min()
    print("%s: %d workgroups; entries spread over %.2f us; kernel (first entry -> last exit) %.2f us" % (name, len(a), (a[:, 0].max() - t0) * 0.01, (a[:, last][a[:, last] > 0].max() - t0) * 0.01))
    for (i, j, lab) in labels:
        ok = (a[:, i] > 0) & (a[:, j] > 0)
        if ok.any():
            dtt = (a[ok, j] - a[ok, i]) * 0.01
            print("    %-58s median %6.2f us   max %6.2f   (%d workgroups)" % (lab, np.median(dtt), dtt.max(), ok.sum()))
show("ba_assemble_kernel", 0, [(0, 1, "entry -> pixel terms computed, rows stored"), (1, 2, "the waves' 90 sums (reduce-scatter) + barrier"), (2, 3, "chunk sums stored / atomics issued"), (0, 3, "whole workgroup")], 3)
show("ba_schur_mfma_kernel", 1, [(0, 1, "entry -> meta read (the chunk sums are other workgroups' now)"), (1, 2, "edge list -> LDS"), (2, 3, "depth phase issued, row table built (wave 0; slices > 0: table, barrier, depth phase)"),
                                 (3, 4, "barrier: rows and table visible (+ merged rows)"), (4, 5, "fast path: row loads + MFMA, PIX / 64 steps (wave 0)"), (5, 6, "fast path: products -> LDS + barrier"),
                                 (6, 7, "fast path: 4-wave sums + fixed-point atomics"), (4, 7, "all tile pairs (fast path or this slice's row passes)"),
                                 (4, 5, "LDS form (dense window; same slots as the fast path's, other workgroups): rows staged"), (5, 6, "LDS form: first row pass (T pairs)"), (6, 7, "LDS form: remaining row passes"), (0, 7, "whole workgroup (slice 0)")], 7)
show("ba_backsub_kernel", 2, [(0, 1, "rows / dx -> LDS + barrier"), (1, 2, "rows x dx, depth update"), (0, 2, "whole workgroup")], 2)
