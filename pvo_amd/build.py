"""Build libpvo_hip.so (HIP, gfx950 only) and the CPU oracle in-tree.

    python -m pvo_amd.build [--force]

hipcc cross-compiles without a GPU. The shared library has no torch/pybind
dependency: the drop-in boundary is the C ABI of include/pvo_hip.h.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pvo_amd", "csrc")
LIB = os.path.join(ROOT, "pvo_amd", "libpvo_hip.so")
PROBE_LIB = os.path.join(ROOT, "pvo_amd", "libpvo_probe.so")      # measurement kernels (bench.py, tools/): csrc/probe_tools.hip
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libpvo_oracle.so")

HIP_SOURCES = [
    "capi_misc.hip",
    "corr_lookup.hip",
    "corr_build.hip",
    "altcorr.hip",
    "geom.hip",
    "gru_fused.hip",
    "graph_glue.hip",
    "conv_small.hip",
    "operator_small.hip",
    "update_exec.hip",
    "se3_ops.hip",
    "encoder_ops.hip",
    "ba.hip",
]
# -target-feature -packed-fp32-ops: NO v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32 in the device code.  On MI355X
# (ROCm 7.0.2) a wave that executes packed-FP32 VALU instructions gets wrong results in single registers of single lanes while
# a kernel of ANOTHER stream that issues MFMA instructions is resident on the same compute unit - the cause of the bundle
# adjustment's run-to-run differences beside the side stream's convolutions (rounds 2-3).  Found in round 4
# (profiles/r04_coresidency.md, tools/sched_bisect.py): arrangements that differed in 200 of 200 runs are bit-identical in 2100 of
# 2100 with this flag; the step time does not change (bench: 229.8 vs 227.5 keyframe updates/s).  The host half of the
# compilation prints "not a recognized feature for this target" for it: expected, the flag is for the gfx950 half.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
               "-Wall", "-Wno-unused-function", "-Wno-unused-result"] + NO_PACKED_FP32


# per-file flags.  geom.hip (projmap / iproj / depth_filter / frame_distance / reproject: a few microseconds each, latency-bound)
# is compiled WITHOUT multiply-add contraction: its kernels then perform exactly the roundings the reference's text states, and
# projmap / iproj / the depth filter's integer counts equal the kernel-text fixtures bit for bit (tests/test_kernel_text_goldens.py;
# until round 5 the counts differed in ~0.5 % of the pixels, where |1/dj - 1/d| sat within a contraction's rounding of the threshold)
EXTRA_FLAGS = {"geom.hip": ["-ffp-contract=off"]}


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _digest(paths, extra=()):
    """sha256 over file contents + the command line: what an object file was built FROM.  Modification times say nothing on a
    box that received the tree as a snapshot (every file has the copy's time) or after a checkout that restored an old file."""
    import hashlib
    h = hashlib.sha256()
    for x in extra:
        h.update(x.encode() + b"\0")
    for p in sorted(paths):
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(target, deps, extra=()):
    """target missing, or built from other inputs than the present ones (stamp file `<target>.sha256` beside it)"""
    stamp = target + ".sha256"
    if not os.path.exists(target) or not os.path.exists(stamp):
        return True
    with open(stamp) as f:
        return f.read().strip() != _digest(deps, extra)


def _stamp(target, deps, extra=()):
    with open(target + ".sha256", "w") as f:
        f.write(_digest(deps, extra) + "\n")


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + "\n")
        raise RuntimeError("build failed: " + " ".join(cmd[:3]))
    return r.stdout


def build_hip(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "pvo_hip.h"))
    objs = []
    procs = []
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        flags = HIPCC_FLAGS + EXTRA_FLAGS.get(src, [])
        if force or _stale(o, [s] + headers, flags):
            cmd = [hipcc] + flags + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), (o, [s] + headers, flags)))
    for cmd, p, (o, deps, flags) in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(" ".join(cmd) + "\n" + out + "\n")
            raise RuntimeError("hipcc failed on " + cmd[-3])
        _stamp(o, deps, flags)
        if verbose and out.strip():
            print(out)
    if force or procs or _stale(LIB, objs):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        _stamp(LIB, objs)
    return LIB


def build_probe(force=False):
    """libpvo_probe.so: the shader-clock and memory-request probes, kept out of the product library"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    src = os.path.join(CSRC, "probe_tools.hip")
    deps = [src, os.path.join(CSRC, "common.h"), os.path.join(ROOT, "include", "pvo_probe.h"), os.path.join(ROOT, "include", "pvo_hip.h")]
    if force or _stale(PROBE_LIB, deps, HIPCC_FLAGS):
        _run([hipcc] + HIPCC_FLAGS + ["-shared", "-o", PROBE_LIB, src])
        _stamp(PROBE_LIB, deps, HIPCC_FLAGS)
    return PROBE_LIB


def build_oracle(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in sorted(os.listdir(ORACLE_DIR)) if f.endswith(".c")]
    hdrs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith(".h")]
    if not srcs:
        return None
    if force or _stale(ORACLE_LIB, srcs + hdrs):
        # -ffp-contract=off: the oracle states every rounding explicitly
        _run(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
              "-Wall", "-o", ORACLE_LIB] + srcs + ["-lm"])
        _stamp(ORACLE_LIB, srcs + hdrs)
    return ORACLE_LIB


def build_all(force=False, verbose=False):
    lib = build_hip(force, verbose)
    build_probe(force)
    return lib, build_oracle(force)


if __name__ == "__main__":
    f = "--force" in sys.argv
    print(build_all(force=f, verbose=True))
