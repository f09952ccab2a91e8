"""What this part's memory system delivers for scattered cache lines (on the GPU box):  python tools/mem_probe.py
pvo_mem_probe (capi_misc.hip): consecutive 128-byte lines, random 128-byte lines, random 64-byte half lines of a 1 GiB buffer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvo_amd import droid_backends as db
g = db.mem_probe_gbps("cuda:0")
print({k: round(v) for k, v in g.items()}, "GB/s fetched;  requests/s: %.1f G (128-byte), %.1f G (64-byte)" % (g["random_128B"] / 128, g["random_64B"] / 64))
