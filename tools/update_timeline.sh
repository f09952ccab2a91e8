# usage (on the GPU box): bash tools/update_timeline.sh <tag>   -> gpurun_out/timeline_<tag>.txt
# rocprofv3 --kernel-trace of a short bench run; prints the kernel schedule (start, duration, stream/queue) of ONE graph
# update from the middle of the timed steps: what runs beside what, and where the launch stream waits.
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline_$TAG.txt
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ut
rocprofv3 --kernel-trace -f csv -d /tmp/ut -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > /tmp/ut_bench.log 2>&1
python - > $OUT <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/ut/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:52]
# a graph update starts with reproject_kernel, the lookup among the next three dispatches; take the 30th update from the end
starts = [i for i, r in enumerate(rows[:-4]) if "reproject_kernel" in r["Kernel_Name"]
          and any("corr_lookup" in rows[i + k]["Kernel_Name"] for k in (1, 2, 3))]
i0, i1 = starts[-30], starts[-29]
t0 = int(rows[i0]["Start_Timestamp"])
print("one graph update: %d dispatches, %.1f us from first start to next update's first start" % (i1 - i0, (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
print("%9s %8s %8s  %-6s %s" % ("start us", "dur us", "gap us", "queue", "kernel  [grid x block]"))
last_end = {}
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print("%9.1f %8.1f %8.1f  %-6s %s  [%s x %s]" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, short(r["Kernel_Name"]),
          r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
PY
tail -3 /tmp/ut_bench.log | cut -c1-200 >> $OUT
