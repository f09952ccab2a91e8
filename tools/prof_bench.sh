# usage (on the GPU box): bash tools/prof_bench.sh <tag>   -> gpurun_out/prof_<tag>/
# rocprofv3 --kernel-trace --stats of the default bench command; the summary is restricted to
# dispatches AFTER MIOpen's find phase (its naive_conv_* candidates run only while solvers are
# being searched during warm-up) so that per-kernel averages describe the timed steps.
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pb
rocprofv3 --kernel-trace --stats -f csv -d /tmp/pb -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline ${PROF_ARGS---steps-only} > /tmp/pb_bench.log 2>&1
mkdir -p $OUT
cp /tmp/pb/*/*kernel_stats.csv $OUT/kernel_stats_full_run.csv
grep "^{\"metric\"" /tmp/pb_bench.log | tail -1 > $OUT/bench_line.json
python - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("/tmp/pb/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last_naive = max([i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("naive_conv")] + [-1])
rows = rows[last_naive + 1:]
agg = collections.OrderedDict()
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(r["Kernel_Name"], [0, 0, 1 << 62, 0]); a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
with open("$OUT/kernel_stats_after_find.csv", "w") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([n, a[0], a[1], "%.1f" % (a[1] / a[0]), a[2], a[3], "%.3f" % (100.0 * a[1] / tot)])
print("dispatches after the find phase: %d, kernel time %.1f ms, wall span %.1f ms" % (len(rows), tot / 1e6, span / 1e6))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print("%-86s calls=%5d avg=%8.1fus total=%7.2fms %5.1f%%" % (n[:86], a[0], a[1] / a[0] / 1e3, a[1] / 1e6, 100.0 * a[1] / tot))
PY
