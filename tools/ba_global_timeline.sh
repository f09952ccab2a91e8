# usage (GPU box): bash tools/ba_global_timeline.sh [NF] [HT] [WD]   kernel sequence of ONE pvo_ba call (2 Gauss-Newton steps) at S-20 size
# under rocprofv3 --kernel-trace: start, duration, gap to the previous kernel.  TOOL_BA_SOLVER=pipe|twin selects the pose solve (pvo_debug_config).
NF=${1:-64}; HT=${2:-48}; WD=${3:-64}
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/bgt
cat > /tmp/bgt_run.py <<PY
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import torch
from pvo_amd import droid_backends as db
from test_geom_ba_gpu import _scene
db.debug_config("ba_solver", os.environ.get("TOOL_BA_SOLVER"))
s = _scene(7, $NF, $HT, $WD, 3, 1)
d = lambda t: t.cuda()
args = [d(s[k]) for k in ("intr", "target", "weight", "eta", "ii", "jj")]
for _ in range(8):
    p, q = d(s["poses"].clone()), d(s["disps"].clone())
    db.ba(p, q, *args, s["t0"], s["t1"], 2, 1e-4, 0.1, False)
    torch.cuda.synchronize()
print("partition", db.ba_last_partition(args[4].shape[0], s["t1"] - s["t0"], $NF, $HT * $WD, "cuda"))
PY
rocprofv3 --kernel-trace -f csv -d /tmp/bgt -- python /tmp/bgt_run.py 2>&1 | grep partition
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/bgt/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ba = [r for r in rows if "ba_" in r["Kernel_Name"]]
last = max(i for i, r in enumerate(ba) if "ba_plan" in r["Kernel_Name"])
seq = ba[last:]
t0 = int(seq[0]["Start_Timestamp"]); prev = t0
print("one pvo_ba call ($NF keyframes, ${HT}x$WD, 2 steps): %d kernels, %.1f us" % (len(seq), (int(seq[-1]["End_Timestamp"]) - t0) / 1e3))
for r in seq:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("  %8.1f  %7.1f us  gap %5.1f  %s  [%s wg]" % ((a - t0) / 1e3, (b - a) / 1e3, (a - prev) / 1e3, name[:50], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) * int(r["Grid_Size_Y"])))
    prev = b
PY
