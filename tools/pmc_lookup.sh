TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$C
  rocprofv3 --pmc $C --kernel-trace -f csv -d /tmp/pm_$C -- python $GRAFT_REPO_ROOT/tools/pmc_lookup.py > /tmp/pm_$C.log 2>&1
  cp /tmp/pm_$C/*/*counter_collection.csv $OUT/${C}_counter_collection.csv 2>/dev/null || ls -R /tmp/pm_$C | head
done
python - <<PY
import csv, json, collections
res = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open("$OUT/%s_counter_collection.csv" % C)))
    per = collections.defaultdict(list)
    for r in rows:
        if r.get("Counter_Name") == C:
            per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        if "corr_lookup" in k:
            name = "lookup_enc" if ("true, true>" in k or "enc_kernel" in k) else ("lookup_tiled" if "true, false>" in k else "lookup")
            res.setdefault(name, {})[C] = v
        if "copy" in k.lower() and max(v) > 1e5: res.setdefault("copy", {})[C] = v
print(json.dumps(res)[:2000])
json.dump(res, open("$OUT/raw.json", "w"))
PY
