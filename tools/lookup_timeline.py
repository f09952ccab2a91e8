"""Per-workgroup phase timeline of the fused lookup + encoder kernel (corr_lookup_r3_kernel<half, tiled, enc>) from in-kernel
clock stamps, on the bench's S-B window.

    python tools/lookup_timeline.py --build      (here: hipcc -DPVO_LOOKUP_PROBE -> tools/_probe/libpvo_hip_lk.so)
    python tools/lookup_timeline.py [--cold]      (GPU box)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBE_DIR = os.path.join(ROOT, "tools", "_probe")
PROBE_LIB = os.path.join(PROBE_DIR, "libpvo_hip_lk.so")
if "--build" in sys.argv:
    from pvo_amd import build
    build.build_hip()
    os.makedirs(PROBE_DIR, exist_ok=True)
    obj = os.path.join(PROBE_DIR, "corr_lookup.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + build.HIPCC_FLAGS + ["-DPVO_LOOKUP_PROBE", "-c", os.path.join(build.CSRC, "corr_lookup.hip"), "-o", obj])
    objs = [obj if s == "corr_lookup.hip" else os.path.join(build.CSRC, s.replace(".hip", ".o")) for s in build.HIP_SOURCES]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROBE_LIB] + objs)
    print(PROBE_LIB)
    sys.exit(0)
import numpy as np
import torch
from pvo_amd import _lib
_lib.LIB_PATH = PROBE_LIB
from pvo_amd import droid_backends as db
import bench
dev = torch.device("cuda:0")
video, graph = bench.make_window(dev)
lib = _lib.load()
lib.pvo_debug_lookup_probe.restype = ctypes.c_int
lib.pvo_debug_lookup_probe.argtypes = [ctypes.c_void_p]
coords1, _ = video.reproject(graph.ii, graph.jj)
pw = graph.update_op.packed_weights(torch.float16)
c1 = coords1[0].contiguous()
launch = lambda: db.corr_lookup_encode_tiled(graph.corr.levels, c1, pw.tensors["enc0_w"], pw.tensors["enc0_b"], slots=graph.corr.slots_tensor())
E = len(graph._ii_h)
nwg = E * (bench.H8 * bench.W8 // 32)
buf = torch.zeros(nwg, 8, dtype=torch.int64, device=dev)
flush = torch.zeros(150 * 1024 * 1024, dtype=torch.float32, device=dev)
for _ in range(5):
    launch()
torch.cuda.synchronize()
assert lib.pvo_debug_lookup_probe(buf.data_ptr()) == 0
if "--cold" in sys.argv:
    flush.max()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); launch(); b.record()
torch.cuda.synchronize()
assert lib.pvo_debug_lookup_probe(None) == 0
t = buf.cpu().numpy().astype(np.int64)
k = 1e-3
d = lambda i, j: np.median(t[:, j] - t[:, i]) * k
hw = t[:, 6] & 0xffffffff
cuid = ((((t[:, 6] >> 32) & 0xf) * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xf)
print("%s: %d workgroups, launch %.1f us by events" % ("cold" if "--cold" in sys.argv else "warm", nwg, a.elapsed_time(b) * 1e3))
print("workgroup life (kcycles, medians): total %.1f | pass 0 (coords -> 8 tile loads -> bilinear -> LDS) %.1f | pass 1 %.1f | barrier %.1f | "
      "encoder weights + 56 MFMA %.1f | slab + 256-byte row stores %.1f" % (d(0, 5), d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5)))
cus = np.unique(cuid)
per = [(cuid == c).sum() for c in cus]
print("compute units %d, workgroups per CU %d..%d" % (len(cus), min(per), max(per)))
conc = []
for c in cus[:64]:
    m = cuid == c
    s, e = t[m, 0], t[m, 5]
    grid = np.linspace(s.min(), e.max(), 50)
    conc.append(np.mean([np.sum((s <= g) & (e > g)) for g in grid]))
print("workgroups resident per CU, time average over its busy span: %.2f" % np.mean(conc))
