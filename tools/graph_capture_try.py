"""Experiment: one pvo_graph_update captured into a HIP graph (torch.cuda.CUDAGraph) and replayed - does it remove the event
bubbles / the host issue time, and are the results bit-identical?   (GPU box)  python tools/graph_capture_try.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
video, graph = bench.make_window(dev, seed=0)
names = ("net", "target_cam", "weight", "raw_mask", "delta_dy")
for _ in range(3):
    graph.update(None, None, use_inactive=True)
torch.cuda.synchronize()
state = {n: getattr(graph, n).clone() for n in names}
p0, d0, dm0 = video.poses.clone(), video.disps.clone(), graph.damping.clone()


def restore():
    for n in names:
        getattr(graph, n).copy_(state[n])
    video.poses.copy_(p0); video.disps.copy_(d0); graph.damping.copy_(dm0)
    torch.cuda.synchronize()


def snap():
    return [t.clone() for t in (video.poses, video.disps, graph.net, graph.target_cam, graph.damping)]


def timed(fn, n=6, reps=20):
    host, gpu = [], []
    for _ in range(reps):
        restore()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        host.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        gpu.append(e0.elapsed_time(e1))
    host.sort(); gpu.sort()
    return 1e3 * host[len(host) // 2] / n, gpu[len(gpu) // 2] / n * 1e3


eager = lambda: graph.update(None, None, use_inactive=True)
restore()
for _ in range(6):
    eager()
torch.cuda.synchronize()
ref = snap()
h, g = timed(eager)
print("eager   : host issue %.0f us per update, GPU %.1f us per update" % (h * 1e3, g))
restore()
cg = torch.cuda.CUDAGraph()
t0 = time.perf_counter()
with torch.cuda.graph(cg):
    graph.update(None, None, use_inactive=True)
torch.cuda.synchronize()
print("capture + instantiate: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
restore()
for _ in range(6):
    cg.replay()
torch.cuda.synchronize()
got = snap()
print("replayed 6 updates bit-identical to eager:", all(torch.equal(a, b) for a, b in zip(ref, got)))
h, g = timed(cg.replay)
print("replay  : host issue %.0f us per update, GPU %.1f us per update" % (h * 1e3, g))
