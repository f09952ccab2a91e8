# usage (on the GPU box): bash tools/sequence_timeline.sh <tag> [frames]   -> gpurun_out/sequence_timeline_<tag>.txt
# rocprofv3 --kernel-trace of the full-sequence leg (bench.py --sequence-only --sequence-plain: warm-up / calibration pass, then the
# plain pass).  The summary is about the LAST pass (passes are separated by the longest idle stretches of the trace: a new Droid is
# built in between): how busy the device is while frames are tracked, which kernels the time goes to, and how many dispatches a
# frame costs - i.e. what the per-frame loop of evaluation_scripts/test_vo.py:99-108 pays beside the graph updates.
TAG=${1:-r05}
FRAMES=${2:-120}
OUT=$GRAFT_REPO_ROOT/gpurun_out/sequence_timeline_$TAG.txt
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/st
rocprofv3 --kernel-trace -f csv -d /tmp/st -- python $GRAFT_REPO_ROOT/bench.py --sequence-only --sequence-plain --sequence-reps 1 --sequence-order ${ORDER:-plain} --sequence-frames $FRAMES > /tmp/st_bench.log 2>&1
python - > $OUT <<PY
import csv, glob, json, collections
rows = list(csv.DictReader(open(glob.glob("/tmp/st/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
S = [int(r["Start_Timestamp"]) for r in rows]; E = [int(r["End_Timestamp"]) for r in rows]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:64]
# the two longest idle stretches separate [warm-up pass | plain pass: tracking | ...]; take everything behind the longest gap that
# lies in the first two thirds of the trace (the plain pass is the longer one)
run, g_all = 0, []
for i in range(len(rows) - 1):
    run = max(run, E[i])
    g_all.append((S[i + 1] - run, i))
gaps = sorted(g_all, reverse=True)[:6]
print("trace: %d dispatches over %.2f s; longest idle stretches (ms at dispatch index): %s" %
      (len(rows), (E[-1] - S[0]) / 1e9, ", ".join("%.0f@%d" % (g / 1e6, i) for g, i in gaps)))
cut = max((i for g, i in gaps[:3] if i < 0.66 * len(rows)), default=0)
seg = rows[cut + 1:]
t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
# union of busy intervals
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
busy, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print("last pass: %d dispatches, %.3f s from first to last kernel, device busy (union of kernel intervals) %.3f s = %.1f %%" %
      (len(seg), (t1 - t0) / 1e9, busy / 1e9, 100.0 * busy / (t1 - t0)))
# the tracking part alone: everything in front of the backend's first launch (its Schur kernel runs 1024-pixel chunks, the frontend's 256)
cut_b = next((i for i, r in enumerate(seg) if "ba_schur_mfma_kernel<false, 1024>" in r["Kernel_Name"] or "dense_panel_kernel" in r["Kernel_Name"]), None)
if cut_b:
    # (back up over the backend's own preparation: the last frontend kernel in front of it is a ba_backsub_kernel or a frame's read-back)
    tr = seg[:cut_b]
    last_fe = max(i for i, r in enumerate(tr) if "ba_backsub_kernel" in r["Kernel_Name"] and i < cut_b - 1)
    # the backend's first update runs lookup + operator before its BA: cut at the last frontend-sized Schur launch instead
    last_fe = max(i for i, r in enumerate(tr) if "ba_schur_mfma_kernel<false, 256>" in r["Kernel_Name"])
    tr = seg[:last_fe + 8]
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in tr)
    b2, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce: b2 += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    b2 += ce - cs
    span = max(e for _, e in iv) - iv[0][0]
    print("tracking part (to the last frontend BA): %d dispatches, %.3f s, device busy %.3f s = %.1f %% UNDER THE PROFILER (every launch costs the host "
          "more here; the same kernels against the unprofiled wall time of bench.py's sequence leg give the share of an ordinary run)" %
          (len(tr), span / 1e9, b2 / 1e9, 100.0 * b2 / span))
if cut_b:
    te = seg[last_fe + 8:]
    ag2 = collections.OrderedDict()
    for r in te:
        a = ag2.setdefault(short(r["Kernel_Name"]), [0, 0])
        a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot2 = sum(v[1] for v in ag2.values())
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in te)
    b3, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce: b3 += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    b3 += ce - cs
    print("terminate part (backend x 2, trajectory filler): %d dispatches, %.3f s, device busy %.3f s; kernel time by name (sum %.3f s):" %
          (len(te), (max(e for _, e in iv) - iv[0][0]) / 1e9, b3 / 1e9, tot2 / 1e9))
    for n, (c, t) in sorted(ag2.items(), key=lambda kv: -kv[1][1])[:24]:
        print("  %-64s x%-6d %9.2f ms  %5.1f %%  avg %7.1f us" % (n, c, t / 1e6, 100.0 * t / tot2, t / 1e3 / c))
agg = collections.OrderedDict()
for r in seg:
    a = agg.setdefault(short(r["Kernel_Name"]), [0, 0])
    a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print("kernel time by name (sum of durations %.3f s):" % (tot / 1e9))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("  %-64s x%-6d %9.2f ms  %5.1f %%  avg %7.1f us" % (n, c, t / 1e6, 100.0 * t / tot, t / 1e3 / c))
# one graph update of the frontend from the second half of the tracking phase (kernel schedule: start, duration, queue)
ups = [i for i, r in enumerate(seg[:-30]) if "reproject_motion_kernel" in r["Kernel_Name"]]
if ups:
    # the frontend's updates come before terminate's (backend / filler): take one around 60 % of them
    i0 = ups[int(0.6 * len(ups))]
    i1 = next((j for j in ups if j > i0), i0 + 40)
    tb = int(seg[i0]["Start_Timestamp"])
    print()
    print("one graph update of the frontend (%d dispatches, %.1f us to the next update's first kernel):" % (i1 - i0, (int(seg[i1]["Start_Timestamp"]) - tb) / 1e3))
    for r in seg[i0:i1 + 1]:
        s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%9.1f %8.1f  q%-3s %s  [%s x %s]" % ((s_ - tb) / 1e3, (e_ - s_) / 1e3, r.get("Queue_Id", "?")[-1:], short(r["Kernel_Name"]), r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")))
# one frame of the motion filter (the 1-edge lookup in the reference's layout marks it): from the frame's first kernel behind
# the previous frame's last to the scalar read-back
mf = [i for i, r in enumerate(seg) if "corr_lookup_r3_kernel" in r["Kernel_Name"] and "enc" not in r["Kernel_Name"]]
if len(mf) > 20:
    c = mf[int(0.6 * len(mf))]
    prev = mf[int(0.6 * len(mf)) - 1]
    lo = max(prev + 25, c - 140)
    tb = int(seg[lo]["Start_Timestamp"])
    print()
    print("one frame of the motion filter (dispatches %d .. %d of the pass; encoder graph, 1-edge volume, lookup, operator, test):" % (lo, c + 40))
    for r in seg[lo:c + 40]:
        s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%9.1f %8.1f  q%-3s %s  [%s x %s]" % ((s_ - tb) / 1e3, (e_ - s_) / 1e3, r.get("Queue_Id", "?")[-1:], short(r["Kernel_Name"]), r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")))
# everything between two keyframes' graph updates: from the last BA kernel of one keyframe step to the first reprojection of the next
# step's first update (tracked frames of the motion filter, the keyframe's context encoder, edge bookkeeping, proximity, volume build)
if len(ups) > 40:
    gaps_u = [(int(seg[ups[j + 1]]["Start_Timestamp"]) - int(seg[ups[j]]["Start_Timestamp"]), j) for j in range(int(0.5 * len(ups)), int(0.7 * len(ups)))]
    big = sorted(g for g in gaps_u)[len(gaps_u) * 9 // 10][0]              # a typical LONG gap = a keyframe boundary
    j = next(j for g, j in gaps_u if g >= big)
    a = next(i for i in range(ups[j + 1], ups[j], -1) if "ba_backsub_kernel" in seg[i]["Kernel_Name"])
    tb = int(seg[a]["End_Timestamp"])
    print()
    print("between two keyframe steps (%d dispatches, %.1f us from the last BA kernel to the next step's first reprojection):" % (ups[j + 1] - a - 1, (int(seg[ups[j + 1]]["Start_Timestamp"]) - tb) / 1e3))
    prev_end = tb
    for r in seg[a + 1:ups[j + 1] + 1]:
        s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%9.1f %8.1f  idle %7.1f  q%-3s %s  [%s x %s]" % ((s_ - tb) / 1e3, (e_ - s_) / 1e3, max(0.0, (s_ - prev_end) / 1e3), r.get("Queue_Id", "?")[-1:], short(r["Kernel_Name"])[:70], r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")))
        prev_end = max(prev_end, e_)
# where the device waits for the host: idle stretches of the pass (no kernel running on any queue), grouped by the kernels on either side
idle = collections.OrderedDict()
run_end, prev = int(seg[0]["End_Timestamp"]), seg[0]
for r in seg[1:]:
    s_ = int(r["Start_Timestamp"])
    if s_ > run_end:
        a = idle.setdefault((short(prev["Kernel_Name"])[:44], short(r["Kernel_Name"])[:44]), [0, 0])
        a[0] += 1; a[1] += s_ - run_end
    if int(r["End_Timestamp"]) > run_end:
        run_end, prev = int(r["End_Timestamp"]), r
tot_idle = sum(v[1] for v in idle.values())
print()
print("device idle in the pass: %.3f s in %d stretches; by (last kernel before -> first kernel after), top 25:" % (tot_idle / 1e9, sum(v[0] for v in idle.values())))
for (a, b), (c, t) in sorted(idle.items(), key=lambda kv: -kv[1][1])[:25]:
    print("  %9.2f ms  x%-5d avg %7.1f us   %s -> %s" % (t / 1e6, c, t / 1e3 / c, a, b))
own = sum(t for n, (c, t) in agg.items() if not (n.startswith("at::") or "rocclr" in n or n.startswith("miopen") or "Cijk" in n or n.startswith("ck::") or "MIOpen" in n or "igemm" in n or "naive" in n))
print("share of kernel time in libpvo_hip kernels: %.1f %%; PyTorch / MIOpen / blit kernels: %.1f %%" % (100.0 * own / tot, 100.0 * (tot - own) / tot))
PY
tail -c 3000 /tmp/st_bench.log >> $OUT
