cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pbs
rocprofv3 --kernel-trace -f csv -d /tmp/pbs -- python $GRAFT_REPO_ROOT/tools/ba_sizes.py | tail -4
python - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("/tmp/pbs/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if "ba_" in r["Kernel_Name"]]
solves = [d for n, d in seq if "ba_solve" in n]
for k in range(4):
    blk = solves[k * 20 + 5:(k + 1) * 20]
    print("solve, size group %d: avg %.1f us" % (k, sum(blk) / len(blk) / 1e3))
for name in ("ba_assemble", "ba_schur", "ba_plan", "ba_backsub", "ba_depth"):
    ds = [d for n, d in seq if name in n]
    for k in range(4):
        blk = ds[k * 20 + 5:(k + 1) * 20]
        print("%s group %d: %.1f us" % (name, k, sum(blk) / max(len(blk), 1) / 1e3))
PY
