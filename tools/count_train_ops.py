"""Counts the aten operators DroidNet.forward dispatches in a training step, by region (SE3 ops / projective_transform / the PyTorch BA / rest):\nwhere the ~87 000 kernel launches of an S-T step come from (profiles/r04_train_step_stats.txt).  GPU box: python tools/count_train_ops.py"""
import sys, collections, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
from torch.utils._python_dispatch import TorchDispatchMode
from pvo_amd.geom import se3 as S, ba as BA, projective_ops as P
from pvo_amd import droid_net as DN
region = ["other"]; counts = collections.Counter()
class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        counts[region[-1]] += 1
        return func(*args, **(kwargs or {}))
def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        region.append(tag if region[-1] in ("other", "BA (geom/ba.py)", "projective_transform") else region[-1])
        try: return f(*a, **k)
        finally: region.pop()
    setattr(obj, name, g)
for n in ("inv", "mul", "act", "adjT", "retr", "exp", "log", "__mul__"):
    if hasattr(S.SE3, n): wrap(S.SE3, n, "SE3 ops")
wrap(P, "projective_transform", "projective_transform")
wrap(BA, "BA", "BA (geom/ba.py)")
DN.BA = BA.BA if hasattr(DN, "BA") else None
import train as T
from pvo_amd.geom import losses as L
from pvo_amd.geom.graph_utils import build_frame_graph
from pvo_amd.synthetic import TrainClips
torch.manual_seed(0)
dev = torch.device("cuda:0"); net = DN.DroidNet().train().to(dev)
clips = TrainClips(6, (200, 400), length=1)
images, poses, disps, intr, gt_masks, gt_vals, segments = [x[None].to(dev) for x in clips[0]]
graph = build_frame_graph(poses, disps, intr, num=20, need_inv=False)
Ps = S.SE3(poses); Gs = S.SE3.IdentityLike(Ps); Gs.data[:, 0] = Ps.data[:, 0]; Gs.data[:, 1:] = Ps.data[:, [1]]
with Count():
    out = net(Gs, images, torch.ones_like(disps[:, :, 3::8, 3::8]), intr / 8.0, graph, num_steps=15, fixedp=2, ret_flow=True, downsample=True, segments=segments)
tot = sum(counts.values())
print("forward aten ops per region (S-T: 6 frames, 20 edges, 15 updates):", {k: (v, round(100*v/tot)) for k, v in counts.items()}, "total", tot)
