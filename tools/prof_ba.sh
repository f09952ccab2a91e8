cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/p0 -- python $GRAFT_REPO_ROOT/tools/microbench.py ba | tail -5
python - <<PY
import csv,glob
f=glob.glob("/tmp/p0/*/*kernel_stats.csv")[0]
print(" ".join("%s=%.1fus"%(r["Name"].split("::")[1].split("(")[0].split("<")[0], float(r["AverageNs"])/1e3) for r in csv.DictReader(open(f)) if "ba_" in r["Name"]))
PY
