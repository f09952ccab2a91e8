"""How long do the motion filter's captured graphs take to replay - back to back (the queue never empty) and launched into an idle
device right after a host synchronisation (what a tracked frame does: the keyframe decision is read back first)?
python tools/graph_gap_probe.py   (GPU box)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvo_amd.droid import Droid, default_args
from pvo_amd.synthetic import drifting_texture_stream

dev = torch.device("cuda:0")
torch.manual_seed(0)
droid = Droid(default_args(device=str(dev), image_size=[240, 808], buffer=64, segm_filter=True, thresh=0.8, filter_thresh=0.0, keyframe_thresh=0.0))
frames = list(drifting_texture_stream(10, seed=0))
for t, image, intr, segm in frames:
    droid.track(t, image, intrinsics=intr, segments=segm)
mf = droid.filterx
img = mf._upload(frames[-1][1])
torch.cuda.synchronize()


def timed(label, call, n=60):
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        call()
    b.record(); torch.cuda.synchronize()
    back_to_back = 1e3 * a.elapsed_time(b) / n
    t_host, t_all = 0.0, 0.0
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        call()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t_all += time.perf_counter() - t0; t_host += t1 - t0
    print("%-34s back to back %7.1f us | into an idle device: host call %7.1f us, call + completion %7.1f us" %
          (label, back_to_back, 1e6 * t_host / n, 1e6 * t_all / n), flush=True)


for g in (mf._context_g, mf._features_g, mf._frame_g):
    print(g.name, "replays", g.replays, "disabled", g.disabled, getattr(g, "error", None), "grad", torch.is_grad_enabled())
torch.set_grad_enabled(False)
from pvo_amd.modules.extractor import BasicEncoder
from pvo_amd.graphs import GraphedCall
for flag in (True, False):
    BasicEncoder.deterministic = flag
    gc_ = GraphedCall(mf._context_dev, name="cnet, deterministic=%s" % flag)
    gf_ = GraphedCall(mf._features_dev, name="fnet, deterministic=%s" % flag)
    for _ in range(4):
        gc_(img); gf_(img)
    timed("cnet graph, vendor deterministic=%s" % flag, lambda: gc_(img))
    timed("fnet graph, vendor deterministic=%s" % flag, lambda: gf_(img))
BasicEncoder.deterministic = True
timed("cnet graph", lambda: mf._context_g(img))
timed("fnet graph", lambda: mf._features_g(img))
args = (img, mf.fmap, mf.net, mf.inp) + tuple(mf._static or ())
timed("frame graph (fnet + test)", lambda: mf._frame_g(*args))
from pvo_amd import config
with torch.no_grad():
    timed("cnet eager", lambda: mf._context_dev(img), n=20)
    timed("fnet eager", lambda: mf._features_dev(img), n=20)
for g in (mf._context_g, mf._features_g, mf._frame_g):
    print(g.name, "replays", g.replays, "disabled", g.disabled, getattr(g, "error", None))
