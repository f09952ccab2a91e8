"""Replay ONE bundle adjustment 40 x 2 iterations back to back, for `rocprofv3 --kernel-trace --stats`: per-kernel averages of the BA
kernels on a given window.
    BA_DUMP=<file> python tools/ba_window_replay.py     a window dumped from a real run (PVO_BENCH_DUMP_BA=<file> python bench.py
                                                        --sequence-only --sequence-plain: the frontend's BA at graph update #600,
                                                        poses 89..109, 342 edges of which 48 active, 23 depth frames, 30 x 101)
    python tools/ba_window_replay.py                    the S-B window of the tests (8 keyframes, 36 edges, 48 x 64)
    LIB=<libpvo_hip variant>                            another build of the library (tools/variant.py)
profiles/r05_schur_sweep.txt was made with it (the chunk / slice variants were compile-time macros, removed after the sweep)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pvo_amd import _lib
if os.environ.get("LIB"):
    _lib.LIB_PATH = os.environ["LIB"]
from pvo_amd import droid_backends as db
dev = torch.device("cuda:0")
if os.environ.get("BA_DUMP"):
    z = torch.load(os.environ["BA_DUMP"])
    eta = (0.2 * z["damping"][z["rows"]] + 1e-7).contiguous()
    d = dict(poses=z["poses"].to(dev), disps=z["disps"].to(dev), intr=z["intr"].to(dev), target=z["target"].to(dev).contiguous(), weight=z["weight"].to(dev).contiguous(),
             eta=eta.to(dev), ii=z["ii"].to(dev), jj=z["jj"].to(dev))
    t0, t1 = int(z["t0"]), int(z["t1"])
else:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_geom_ba_gpu import _scene
    nf = int(os.environ.get("NF", "8"))            # NF / HT / WD / RAD: a synthetic window (NF=26 RAD=8 HT=30 WD=101: 16 neighbours per frame)
    sc = _scene(0, nf, int(os.environ.get("HT", "48")), int(os.environ.get("WD", "64")), int(os.environ.get("RAD", "3")), 1)
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}
    t0, t1 = 1, nf
for _ in range(40):
    db.ba(d["poses"].clone(), d["disps"].clone(), d["intr"], d["target"], d["weight"], d["eta"], d["ii"], d["jj"], t0, t1, 2, 1e-4, 0.1, False)
torch.cuda.synchronize()
