"""Is the tracked sequence reproducible bit for bit from one run to the next in one process?  Runs bench.py's 240 x 808 stream twice
with the same seeds and reports the first frame / quantity whose bytes differ.  python tools/determinism_probe.py [frames]  (GPU box)"""
import os, sys, hashlib, random
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from pvo_amd.droid import Droid, default_args
from pvo_amd.synthetic import drifting_texture_stream

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")


def h(t):
    return hashlib.md5(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:10] if t is not None and torch.is_tensor(t) else None


def run(tag):
    torch.manual_seed(0)
    droid = Droid(default_args(device=str(dev), image_size=[240, 808], buffer=n_frames + 40, segm_filter=True, thresh=0.8, filter_thresh=0.2026, keyframe_thresh=0.0))
    rng = random.Random(1234)
    sched = [rng.random() < bench.REMOVAL_RATE for _ in range(4 * n_frames + 64)]
    fe, mf = droid.frontend, droid.filterx
    fe.keyframe_decision = lambda k, dist: sched[k]
    log = []
    g = fe.graph
    gupd = g.update
    state = {"frame": -1, "upd": 0}

    def _gupd(*a, **kw):
        r = gupd(*a, **kw)
        state["upd"] += 1
        n = droid.video.counter
        log.append(("frame %d update %d" % (state["frame"], state["upd"]),
                    {"poses": h(droid.video.poses[:n]), "disps": h(droid.video.disps[:n]), "net": h(g.net), "target": h(g.target_cam), "weight": h(g.weight),
                     "raw_mask": h(g.raw_mask), "delta_dy": h(g.delta_dy), "damping": h(g.damping[:n]), "ii": h(g.ii), "jj": h(g.jj),
                     "ii_inac": h(g.ii_inac), "target_inac": h(g.target_cam_inac), "weight_inac": h(g.weight_inac)}))
        return r
    g.update = _gupd
    frames = list(drifting_texture_stream(n_frames, seed=0))
    with torch.no_grad():
        for t, image, intr, segm in frames:
            state["frame"] = t; state["upd"] = 0
            droid.track(t, image, intrinsics=intr, segments=segm)
            n = droid.video.counter
            v = droid.video
            log.append(("frame %d tracked" % t, {"counter": n, "fmap_ref": h(mf.fmap), "net_ref": h(mf.net), "inp_ref": h(mf.inp),
                                                 "fmaps": h(v.fmaps[:n]), "nets": h(v.nets[:n]), "inps": h(v.inps[:n]), "segms": h(v.segms[:n]),
                                                 "poses": h(v.poses[:n + 1]), "disps": h(v.disps[:n + 1]), "intr": h(v.intrinsics[:n]),
                                                 "edges": (tuple(g._ii_h), tuple(g._jj_h))}))
    del droid
    return log


run("warm-up: MIOpen chooses its convolution kernels on the first calls of a process")
a, b = run("a"), run("b")
bad = 0
for (la, da), (lb, db_) in zip(a, b):
    diff = [k for k in da if da[k] != db_.get(k)]
    if la != lb or diff:
        print("first difference at:", la, "|", lb, "->", diff)
        bad = 1
        break
print("entries compared: %d; %s" % (min(len(a), len(b)), "DIFFERENT" if bad else "identical"))
