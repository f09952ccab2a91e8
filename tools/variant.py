"""Build and run experiment variants of the library.
    python tools/variant.py build <tag> <source.hip> [-DNAME=VALUE ...]      (here)  -> tools/_probe/libpvo_hip_<tag>.so
    python tools/variant.py bench <tag> [bench.py arguments]                 (GPU)   -> bench.py's JSON line with that library
One translation unit is recompiled with the extra definitions, the rest of the objects are the product's."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBE_DIR = os.path.join(ROOT, "tools", "_probe")


def lib_path(tag):
    return os.path.join(PROBE_DIR, "libpvo_hip_%s.so" % tag)


if sys.argv[1] == "build":
    from pvo_amd import build
    tag, src, defs = sys.argv[2], sys.argv[3], sys.argv[4:]
    build.build_hip()
    os.makedirs(PROBE_DIR, exist_ok=True)
    obj = os.path.join(PROBE_DIR, "%s_%s" % (tag, src.replace(".hip", ".o")))
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + build.HIPCC_FLAGS + defs + ["-c", os.path.join(build.CSRC, src), "-o", obj])
    objs = [obj if s == src else os.path.join(build.CSRC, s.replace(".hip", ".o")) for s in build.HIP_SOURCES]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path(tag)] + objs)
    print(lib_path(tag))
elif sys.argv[1] == "bench":
    from pvo_amd import _lib
    _lib.LIB_PATH = lib_path(sys.argv[2])
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[3:]
    import bench
    bench.main()
