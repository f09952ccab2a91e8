# usage: python tools/ba_ablate.py --build (here), then on the GPU box: bash tools/ba_ablate.sh  -> per-kernel times of the BA at S-B
# for the product build and two ablations (atomics replaced by plain stores; wave reductions skipped): where the 14.6 + 23 us of
# assembly + Schur go.  rocprofv3 --kernel-trace --stats per library.
cd /tmp; export TMPDIR=/tmp
for L in ${ABL:-base noatomic noreduce ppt1 pix256 ppt1pix256}; do
  rm -rf /tmp/abl_$L
  rocprofv3 --kernel-trace --stats -f csv -d /tmp/abl_$L -- python $GRAFT_REPO_ROOT/tools/ba_probe.py $GRAFT_REPO_ROOT/tools/_probe/libpvo_hip_abl_$L.so > /tmp/abl_$L.log 2>&1
  echo "== $L: $(tail -1 /tmp/abl_$L.log)"
  python - <<PY
import csv, glob
for r in csv.DictReader(open(glob.glob("/tmp/abl_$L/*/*kernel_stats.csv")[0])):
    if "ba_" in r["Name"]:
        print("   %-28s calls %5s  avg %8.1f us" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
