# usage (GPU box): bash tools/prof_global_update.sh   -> kernel timeline of ONE global update of the edge-sharded leg of bench.py (S-20: 64 keyframes,
# 372 edges, resident volumes) under rocprofv3 --kernel-trace
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pgu
rocprofv3 --kernel-trace -f csv -d /tmp/pgu -- python $GRAFT_REPO_ROOT/tools/prof_lowmem.py > /tmp/pgu.log 2>&1
tail -2 /tmp/pgu.log | cut -c1-600
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/pgu/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# ONE global update = from a reproject_motion launch of the 372-edge graph to the next one (the BA follows graph_post)
starts = [i for i, r in enumerate(rows) if "reproject_motion" in r["Kernel_Name"] and int(r["Grid_Size_Y"]) > 100]
a, b = starts[-2], starts[-1]
seq = rows[a:b]
t0 = int(seq[0]["Start_Timestamp"])
print("one global update (S-20: 372 edges, 64 keyframes, resident volumes) under rocprofv3: %d dispatches, %.2f ms to the next update's first kernel" % (len(seq), (int(rows[b]["Start_Timestamp"]) - t0) / 1e6))
print(" start us    dur us  queue  kernel  [grid threads / workgroup]")
for r in seq:
    s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:64]
    print("  %8.1f %8.1f  q%s  %s  [%s x %s x %s / %s]" % ((s_ - t0) / 1e3, (e_ - s_) / 1e3, r["Queue_Id"][-1:], k, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"]))
conv = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seq if "conv3x3_big" in r["Kernel_Name"] and r["Queue_Id"] == seq[0]["Queue_Id"])
ba = [r for r in seq if "ba_" in r["Kernel_Name"]]
print("launch-stream conv3x3_big kernels: %.2f ms; BA (two Gauss-Newton steps, first kernel to last): %.2f ms" % (conv / 1e6, (int(ba[-1]["End_Timestamp"]) - int(ba[0]["Start_Timestamp"])) / 1e6))
PY
