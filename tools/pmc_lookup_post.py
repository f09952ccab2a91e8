"""gpurun_out/pmc_<tag>/raw.json (tools/pmc_lookup.sh) -> profiles/<tag>_lookup_pmc.json: HBM bytes per launch of the lookup
kernels, read side corrected by the factor the 512 MiB calibration copy of the SAME run shows (MI355X_MICROARCH.md: FETCH_SIZE
reports half the bytes of 16 B/lane reads on gfx950), stamped with the sha256 of the kernel source it was measured on
(bench.py reports `roofline.traffic` only while that source is unchanged).   python tools/pmc_lookup_post.py <tag> [out]"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
raw = json.load(open(os.path.join(ROOT, "gpurun_out", "pmc_" + tag, "raw.json")))
med = lambda v: sorted(v)[len(v) // 2]
copy_f, copy_w = med(raw["copy"]["FETCH_SIZE"]), med(raw["copy"]["WRITE_SIZE"])
fcorr, wcorr = 512 * 1024 / copy_f, 512 * 1024 / copy_w
E, HW = 36, 48 * 64
alg = {"lookup_enc": E * HW * (4 * 64 * 2 + 8 + 128 * 2), "lookup_tiled": E * HW * (4 * 64 * 2 + 8 + 196 * 2), "lookup": E * HW * (4 * 64 * 2 + 8 + 196 * 2)}
names = {"lookup_enc": ("fused_encoder", "corr_lookup_r3_enc_kernel<pvo_half> (lookup + Conv2d(196,128,1) + ReLU on the 8x8-tiled pool; what pvo_graph_update launches)"),
         "lookup_tiled": ("tiled", "corr_lookup_r3_kernel<pvo_half, true, false> (8x8-tiled pool, 196-channel output)"),
         "lookup": ("row_major", "corr_lookup_r3_kernel<pvo_half, false, false> (drop-in layout [N,h1,w1,h2,w2])")}
src = os.path.join(ROOT, "pvo_amd", "csrc", "corr_lookup.hip")
out = {"command": "bash tools/pmc_lookup.sh %s; python tools/pmc_lookup_post.py %s  (two passes: rocprofv3 --pmc FETCH_SIZE --kernel-trace, --pmc WRITE_SIZE --kernel-trace; "
                  "workload tools/pmc_lookup.py: S-B, E=36, 48x64, fp16; Infinity Cache flushed between launches)" % (tag, tag),
       "kernel_source": "pvo_amd/csrc/corr_lookup.hip", "kernel_source_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(),
       "git_head": subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], stdout=subprocess.PIPE, text=True).stdout.strip(),
       "raw_counter_units": "KB (1024 B) per dispatch, median over dispatches",
       "calibration_copy_512MiB": {"FETCH_SIZE_raw": copy_f, "WRITE_SIZE_raw": copy_w, "fetch_correction": fcorr, "write_correction": wcorr}}
for key, (label, kernel) in names.items():
    if key not in raw:
        continue
    f, w = med(raw[key]["FETCH_SIZE"]), med(raw[key]["WRITE_SIZE"])
    rd, wr = f * 1024 * fcorr, w * 1024 * wcorr
    out[label] = {"kernel": kernel, "FETCH_SIZE_raw_KB": f, "WRITE_SIZE_raw_KB": w, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                  "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg[key], "traffic_over_algorithmic": (rd + wr) / alg[key]}
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "%s_lookup_pmc.json" % tag)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out.get("fused_encoder", out), indent=1))
