# usage (on the GPU box): bash tools/update_timeline.sh <tag>   -> gpurun_out/timeline_<tag>.txt
# rocprofv3 --kernel-trace of a short bench run; prints the kernel schedule (start, duration, stream/queue) of ONE graph
# update from the middle of the timed steps: what runs beside what, and where the launch stream waits.
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline_$TAG.txt
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ut
# (second argument: an experiment build of the library, tools/variant.py build <variant> ...)
if [ -n "$2" ]; then BENCH="$GRAFT_REPO_ROOT/tools/variant.py bench $2"; else BENCH="$GRAFT_REPO_ROOT/bench.py"; fi
rocprofv3 --kernel-trace -f csv -d /tmp/ut -- python $BENCH --steps 10 --warmup 3 --steps-only > /tmp/ut_bench.log 2>&1
python - > $OUT <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/ut/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:52]
# a graph update starts with reproject_motion_kernel (reproject_kernel before round 3), the lookup among the next three dispatches; take the 30th update from the end
starts = [i for i, r in enumerate(rows[:-4]) if ("reproject_motion_kernel" in r["Kernel_Name"] or "reproject_kernel" in r["Kernel_Name"])
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
# what runs BETWEEN the graph updates of one step (edge rebuild, frame distances, the benchmark's own state restore):
# the six updates of a keyframe are consecutive entries of starts; take the step that contains update -30
import collections
ends = []
for i in starts:
    j = i
    while j + 1 < len(rows) and not (("reproject_motion_kernel" in rows[j + 1]["Kernel_Name"] or "reproject_kernel" in rows[j + 1]["Kernel_Name"]) and j + 1 in set(starts)):
        if "ba_backsub_kernel" in rows[j]["Kernel_Name"] and j > i + 10 and "ba_backsub" not in rows[j + 1]["Kernel_Name"] and "ba_assemble" not in rows[j + 1]["Kernel_Name"]:
            break
        j += 1
    ends.append(j)
k0 = len(starts) - 36
gap_rows = collections.OrderedDict()
span = 0
for a in range(k0, k0 + 6):
    lo, hi = ends[a] + 1, starts[a + 1]
    if hi > lo:
        span += int(rows[hi]["Start_Timestamp"]) - int(rows[lo - 1]["End_Timestamp"])
    for r in rows[lo:hi]:
        e = gap_rows.setdefault(short(r["Kernel_Name"]), [0, 0])
        e[0] += 1; e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
# ... and the schedule of the LONGEST of those six gaps (the keyframe boundary)
best = None
for a in range(k0, k0 + 6):
    lo, hi = ends[a] + 1, starts[a + 1]
    if hi > lo:
        d = int(rows[hi]["Start_Timestamp"]) - int(rows[lo - 1]["End_Timestamp"])
        if best is None or d > best[0]:
            best = (d, lo, hi)
if best:
    d, lo, hi = best
    tb = int(rows[lo - 1]["End_Timestamp"])
    print()
    print("the keyframe boundary: %.1f us between the last BA kernel of one update and the first kernel of the next; dispatches in it:" % (d / 1e3))
    for r in rows[lo:hi]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%9.1f %8.1f  q%-3s %s  [%s x %s]" % ((s - tb) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?")[-1:], short(r["Kernel_Name"]), r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")))
print()
print("between the updates of six consecutive graph updates (one keyframe step): %.1f us of wall time; kernels there:" % (span / 1e3))
for n, (c, t) in sorted(gap_rows.items(), key=lambda kv: -kv[1][1])[:25]:
    print("  %-52s x%-3d %8.1f us" % (n, c, t / 1e3))
PY
tail -3 /tmp/ut_bench.log | cut -c1-200 >> $OUT
