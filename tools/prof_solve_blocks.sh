cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/rpb; SIZES=85 rocprofv3 --kernel-trace --stats -f csv -d /tmp/rpb -- python $GRAFT_REPO_ROOT/tools/solve_check.py blocks > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/rpb/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    print("  %-60s x%-5s avg %8.1f us  min %8.1f  max %8.1f  total %8.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"])/1e3))
PY
