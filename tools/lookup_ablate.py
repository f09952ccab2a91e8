"""Where does the fused lookup's time go?  Ablation builds of corr_lookup.hip (results are wrong by construction):
    python tools/lookup_ablate.py --build        here: base / loads-only / no-loads libraries under tools/_probe/
    python tools/lookup_ablate.py                GPU: cold (cache flushed) and warm times of each on the S-B window"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBE_DIR = os.path.join(ROOT, "tools", "_probe")
VARIANTS = (("base", []), ("loads_only", ["-DPVO_LK_ABL=1"]), ("no_loads", ["-DPVO_LK_ABL=2"]), ("no_barriers", ["-DPVO_LK_ABL=3"]))
if "--build" in sys.argv:
    from pvo_amd import build
    build.build_hip()
    os.makedirs(PROBE_DIR, exist_ok=True)
    for tag, flags in VARIANTS:
        obj = os.path.join(PROBE_DIR, "lkabl_%s.o" % tag)
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + build.HIPCC_FLAGS + flags + ["-c", os.path.join(build.CSRC, "corr_lookup.hip"), "-o", obj])
        objs = [obj if s == "corr_lookup.hip" else os.path.join(build.CSRC, s.replace(".hip", ".o")) for s in build.HIP_SOURCES]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(PROBE_DIR, "libpvo_hip_lkabl_%s.so" % tag)] + objs)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch
    from pvo_amd import _lib
    _lib.LIB_PATH = os.path.join(PROBE_DIR, "libpvo_hip_lkabl_%s.so" % sys.argv[2])
    from pvo_amd import droid_backends as db
    import bench
    dev = torch.device("cuda:0")
    video, graph = bench.make_window(dev)
    coords1, _ = video.reproject(graph.ii, graph.jj)
    pw = graph.update_op.packed_weights(torch.float16)
    c1 = coords1[0].contiguous()
    launch = lambda: db.corr_lookup_encode_tiled(graph.corr.levels, c1, pw.tensors["enc0_w"], pw.tensors["enc0_b"], slots=graph.corr.slots_tensor())
    flush = torch.zeros(150 * 1024 * 1024, dtype=torch.float32, device=dev)
    for _ in range(5):
        launch()
    cold = []
    for _ in range(20):
        flush.max()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); launch(); e1.record()
        cold.append((e0, e1))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        launch()
    b.record(); torch.cuda.synchronize()
    print("%-11s cold %.1f us   warm back to back %.1f us" % (sys.argv[2], sum(x.elapsed_time(y) for x, y in cold) / len(cold) * 1e3, a.elapsed_time(b) / 50 * 1e3))
    sys.exit(0)
for tag, _ in VARIANTS:
    subprocess.call([sys.executable, os.path.abspath(__file__), "--one", tag])
