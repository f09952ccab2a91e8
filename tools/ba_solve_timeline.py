"""Phase split of ba_solve_kernel (one workgroup: fixed point -> fp64, blocked Cholesky, substitution, retraction) from
in-kernel clock stamps.   python tools/ba_solve_timeline.py --build   |   NF=8 python tools/ba_solve_timeline.py"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
PROBE_DIR = os.path.join(ROOT, "tools", "_probe")
PROBE_LIB = os.path.join(PROBE_DIR, "libpvo_hip_ba%s.so" % os.environ.get("PROBE_TAG", ""))
if "--build" in sys.argv:
    from pvo_amd import build
    build.build_hip()
    os.makedirs(PROBE_DIR, exist_ok=True)
    obj = os.path.join(PROBE_DIR, "ba%s.o" % os.environ.get("PROBE_TAG", ""))
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + build.HIPCC_FLAGS + ["-DPVO_BA_PROBE=%s" % os.environ.get("PROBE_LEVEL", "1")] + os.environ.get("PROBE_DEFS", "").split() + ["-c", os.path.join(build.CSRC, "ba.hip"), "-o", obj])
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
s = _scene(0, nf, int(os.environ.get("HT", "48")), int(os.environ.get("WD", "64")), int(os.environ.get("RAD", "3")), 1)      # RAD=8: a dense window
d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in s.items()}
lib = _lib.load()
lib.pvo_debug_ba_probe.restype = ctypes.c_int; lib.pvo_debug_ba_probe.argtypes = [ctypes.c_void_p]
buf = torch.zeros(64, dtype=torch.int64, device=dev)      # (slot 15: which kernel wrote the step stamps)
run = lambda: db.ba(d["poses"].clone(), d["disps"].clone(), d["intr"], d["target"], d["weight"], d["eta"], d["ii"], d["jj"], 1, nf, 2, 1e-4, 0.1, False)
for _ in range(3):
    run()
torch.cuda.synchronize()
assert lib.pvo_debug_ba_probe(buf.data_ptr()) == 0
run(); torch.cuda.synchronize()
assert lib.pvo_debug_ba_probe(None) == 0
t = buf.cpu().tolist()
print("P = %d free poses: load + convert %.1f k cycles | factorisation %.1f | substitution %.1f | dx out + retraction %.1f | total %.1f"
      % (nf - 1, (t[1] - t[0]) / 1e3, (t[2] - t[1]) / 1e3, (t[4] - t[2]) / 1e3, (t[5] - t[4]) / 1e3, (t[5] - t[0]) / 1e3))
if t[15] == 0xD15E:        # the dense matrix-core solve (ba_solve_dense_kernel): one step of eight columns, the middle one, thread 0
    names = ["8x8 Cholesky + row solve", "barrier", "trailing update (MFMA issue)", "next panel out of the accumulators", "-", "barrier"]
    print("   dense solve, step nb/2: " + " | ".join("%s %d" % (nm, t[9 + i] - t[8 + i]) for i, nm in enumerate(names)) + " | step %d cycles" % (t[14] - t[8]))
if t[14] and t[15] != 0xD15E:
    names = ["operand loads issued", "6x6 Cholesky", "panel + store of the factored block", "publish", "wait for the workers", "look-ahead update"]
    print("   pipelined factorisation, wave 0, block column P/2: " + " | ".join("%s %d" % (nm, t[9 + i] - t[8 + i]) for i, nm in enumerate(names)) + " | step %d cycles" % (t[14] - t[8]))
if t[32]:
    part = db.ba_last_partition(d["ii"].shape[0], nf - 1, nf, s["disps"].shape[1] * s["disps"].shape[2], dev)
    names = ["", "tables, local layout, zero fill", "load", "active-row lists", "own columns", "top: wait for the terms | bottom: terms written", "top: terms added | bottom: wait for x",
             "top: separator columns | bottom: (none)", "top: back substitution", "top: x handed over", "back substitution (bottom) / idle", "dx out + retraction"]
    base = min(t[32], t[48])
    print("   partitioned solve, (m, s) = %s; 100 MHz clock, us since the first workgroup entered:" % (part,))
    for w, nm in ((0, "workgroup 0 (poses [0, s))"), (1, "workgroup 1 (poses [s, P) reversed)")):
        st = [t[32 + 16 * w + k] for k in range(12)]
        print("     %s: entered at %.2f" % (nm, (st[0] - base) / 100.0))
        prev = st[0]
        for k in range(1, 12):
            if st[k]:
                print("        %-62s %7.2f us  (at %.2f)" % (names[k], (st[k] - prev) / 100.0, (st[k] - base) / 100.0)); prev = st[k]
