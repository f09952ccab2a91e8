"""Host time of every piece of a keyframe step of the tracked sequence (bench.py `sequence`: 240 x 808 stream, panoptic segments), no
synchronisation added: a call's time is what the HOST spends in it - launches and Python, plus any wait for the device the call
contains (read-backs).  python tools/host_phase_probe.py [frames] [pipelined]   (GPU box)"""
import os, sys, time, collections
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from pvo_amd.droid import Droid, default_args
from pvo_amd.synthetic import drifting_texture_stream
import pvo_amd.factor_graph as fgm
import pvo_amd.depth_video as dvm
import pvo_amd.motion_filter as mfm
import pvo_amd.frontend as fem

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 120
pipelined = "pipelined" in sys.argv[2:]
with_terminate = "terminate" in sys.argv[2:]
T = collections.defaultdict(list)
stack = []


def wrap(cls, name, label=None):
    f = getattr(cls, name)
    label = label or "%s.%s" % (cls.__name__, name)

    def g(*a, **kw):
        t0 = time.perf_counter()
        stack.append(0.0)
        try:
            return f(*a, **kw)
        finally:
            dt = time.perf_counter() - t0
            inner = stack.pop()
            T[label].append((dt, dt - inner))           # (inclusive, exclusive of wrapped callees)
            if stack:
                stack[-1] += dt
    setattr(cls, name, g)


for cls, names in ((fgm.FactorGraph, ("rm_factors", "add_proximity_factors", "add_factors", "rm_keyframe", "update", "_sync_edge_index")),
                   (dvm.DepthVideo, ("distance", "append", "remember_features", "_dense_segments")),
                   (mfm.MotionFilter, ("begin", "finish", "_upload", "_new_reference", "_append")),
                   (fem.DroidFrontend, ("_update_begin", "_update_finish", "_initialize"))):
    for n in names:
        if hasattr(cls, n):
            wrap(cls, n)
import pvo_amd.backend as bem, pvo_amd.trajectory_filler as tfm, pvo_amd.modules.corr as cm
for cls, names in ((bem.DroidBackend, ("__call__", "_connect_all", "_volumes_fit", "_graph")), (fgm.FactorGraph, ("update_lowmem", "_update_fused", "clear_edges", "_filter_repeated")),
                   (tfm.PoseTrajectoryFiller, ("__call__", "_fill")), (dvm.DepthVideo, ("normalize",)),
                   (cm.CorrVolumePool, ("add", "reserve", "put", "keep", "__init__"))):
    for n in names:
        if hasattr(cls, n):
            wrap(cls, n)
wrap(torch.Tensor, "item", "Tensor.item (wait)")
wrap(torch.Tensor, "cpu", "Tensor.cpu (wait)")
wrap(torch.cuda.Event, "synchronize", "Event.synchronize (wait)")
if os.environ.get("PROBE_TENSOR_OPS"):
    for n in ("__setitem__", "copy_", "to", "clone", "index_select", "mean", "zero_"):
        wrap(torch.Tensor, n, "Tensor.%s" % n)
from pvo_amd.graphs import GraphedCall
wrap(GraphedCall, "__call__", "GraphedCall")

dev = torch.device("cuda:0")
for rep in range(int(os.environ.get("PROBE_REPS", "3"))):
    torch.manual_seed(0)
    droid = Droid(default_args(device=str(dev), image_size=[240, 808], buffer=n_frames + 40, segm_filter=True, thresh=0.8, filter_thresh=0.2026, keyframe_thresh=0.0, pipelined=pipelined))
    import random
    rng = random.Random(1234)
    sched = [rng.random() < bench.REMOVAL_RATE for _ in range(4 * n_frames + 64)]
    droid.frontend.keyframe_decision = lambda k, dist: sched[k]
    frames = list(drifting_texture_stream(n_frames, seed=0))
    for k in list(T):
        T[k].clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        for t, image, intr, segm in frames:
            droid.track(t, image, intrinsics=intr, segments=segm)
        droid.flush()
    torch.cuda.synchronize(); total = time.perf_counter() - t0
    if with_terminate:
        track_T = {k: list(v) for k, v in T.items()}
        for k in list(T):
            T[k].clear()
        t0 = time.perf_counter()
        droid.terminate(iter(frames), need_inv=True)
        torch.cuda.synchronize(); total_term = time.perf_counter() - t0
    del droid
if with_terminate:
    print("terminate: %.3f s; host time by call" % total_term)
    for k, v in sorted(T.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
        if v:
            print("%-44s %6d %10.1f %10.1f %12.1f" % (k, len(v), 1e3 * sum(x[0] for x in v), 1e3 * sum(x[1] for x in v), 1e6 * sum(x[1] for x in v) / len(v)))
    T = track_T
print("pipelined" if pipelined else "reference order"); print("tracking %d frames: %.3f s; host time by call (sum over the pass; exclusive = without the wrapped calls inside)" % (n_frames, total))
print("%-44s %6s %10s %10s %12s" % ("call", "n", "incl ms", "excl ms", "excl us/call"))
for k, v in sorted(T.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    if v:
        print("%-44s %6d %10.1f %10.1f %12.1f" % (k, len(v), 1e3 * sum(x[0] for x in v), 1e3 * sum(x[1] for x in v), 1e6 * sum(x[1] for x in v) / len(v)))
