# usage (GPU box): bash tools/pmc_kernels.sh <tag>  -> gpurun_out/pmc_<tag>/kernel_counters.json
# Counter passes (one rocprofv3 run per set: SQ has 8 slots, GRBM 2) over tools/pmc_workload.py, kernel trace only.
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pk_$i
  timeout 300 rocprofv3 --pmc $SET --kernel-trace -f csv -d /tmp/pk_$i -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py > /tmp/pk_$i.log 2>&1
  cp /tmp/pk_$i/*/*counter_collection.csv $OUT/set${i}_counter_collection.csv 2>/dev/null || tail -5 /tmp/pk_$i.log
done
python - <<PY
import csv, json, collections, glob
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/set*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("conv3x3_big_kernel<pvo_half, false>", "conv3x3_big_kernel<pvo_half, true>", "corr_lookup_r3_enc_kernel",
                    "ba_assemble_kernel", "ba_schur_mfma_kernel<true, 256>", "ba_schur_mfma_kernel<false, 256>", "ba_depth_kernel", "ba_schur_reduce_kernel", "ba_solve_dense_kernel<4>",
                    "ba_solve_dense_kernel<14>", "ba_solve_kernel", "ba_backsub_kernel"):
            if key in k.replace("(anonymous namespace)::", ""):
                res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                for extra in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count"):
                    if extra in r: res[key]["_" + extra] = [float(r[extra])]
out = {k: {c: sorted(v)[len(v) // 2] for c, v in d.items()} for k, d in res.items()}
json.dump(out, open("$OUT/kernel_counters.json", "w"), indent=1)
print(json.dumps(out)[:3000])
PY
