cd /tmp; export TMPDIR=/tmp
for cfg in "NF=26 RAD=8 HT=30 WD=101" "NF=8 RAD=3 HT=48 WD=64" "NF=22 RAD=8 HT=30 WD=101"; do
  rm -rf /tmp/rp; env $cfg rocprofv3 --kernel-trace --stats -f csv -d /tmp/rp -- python $GRAFT_REPO_ROOT/tools/ba_window_replay.py > /dev/null 2>&1
  echo "== $cfg" ; python - <<'PY'
import csv, glob
f = glob.glob("/tmp/rp/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    print("  %-70s x%-5s avg %8.1f us  min %8.1f  max %8.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
