"""BA at several window sizes (run under rocprofv3 --kernel-trace --stats to see how the solve scales with 6P)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from pvo_amd import droid_backends as db
from test_geom_ba_gpu import _scene
dev = torch.device("cuda:0")
for nf in (3, 8, 16, 24):
    s = _scene(0, nf, 24, 32, 3, 1)
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in s.items()}
    for _ in range(20):
        poses, disps = d["poses"].clone(), d["disps"].clone()
        db.ba(poses, disps, d["intr"], d["target"], d["weight"], d["eta"], d["ii"], d["jj"], 1, nf, 1, 1e-4, 0.1, False)
    torch.cuda.synchronize()
    print("P", nf - 1, "E", d["ii"].shape[0])
