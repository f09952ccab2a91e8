"""build the ablation libraries of tools/ba_ablate.sh:  python tools/ba_ablate.py --build"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pvo_amd import build
build.build_hip()
out = os.path.join(ROOT, "tools", "_probe"); os.makedirs(out, exist_ok=True)
VARIANTS = (("base", []), ("noatomic", ["-DPVO_ABL_NOATOMIC"]), ("noreduce", ["-DPVO_ABL_NOREDUCE"]), ("ppt1", ["-DPVO_ASM_PPT=1"]),
            ("pix256", ["-DPVO_SCHUR_PIX=256"]), ("ppt1pix256", ["-DPVO_ASM_PPT=1", "-DPVO_SCHUR_PIX=256"]))
for tag, flags in VARIANTS:
    obj = os.path.join(out, "abl_%s_ba.o" % tag)
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + build.HIPCC_FLAGS + flags + ["-c", os.path.join(build.CSRC, "ba.hip"), "-o", obj])
    objs = [obj if s == "ba.hip" else os.path.join(build.CSRC, s.replace(".hip", ".o")) for s in build.HIP_SOURCES]
    lib = os.path.join(out, "libpvo_hip_abl_%s.so" % tag)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)
