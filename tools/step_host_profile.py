"""where the host time of one bench step goes (edge bookkeeping vs updates), and how much of the step the GPU idles"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
video, graph = bench.make_window(dev)
snap = bench.Snapshot(video, graph)
snap.edge_list = list(zip(graph._ii_h, graph._jj_h))
for _ in range(12):
    bench.keyframe_update(video, graph, snap)
torch.cuda.synchronize()
# phase timing with syncs (serialised: host + GPU per phase)
import collections
acc = collections.defaultdict(float)
N = 10
for _ in range(N):
    def ph(name, fn):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); t1 = time.perf_counter(); torch.cuda.synchronize()
        acc[name + " host"] += t1 - t; acc[name + " host+gpu"] += time.perf_counter() - t
        return r
    newest = bench.NKF - 1
    ph("restore", snap.restore)
    pairs = [(i, j) for i, j in zip(graph._ii_h, graph._jj_h) if i == newest or j == newest]
    ph("rm_factors", lambda: graph.rm_factors([(i == newest or j == newest) for i, j in zip(graph._ii_h, graph._jj_h)]))
    ph("add_factors", lambda: graph.add_factors([p[0] for p in pairs], [p[1] for p in pairs]))
    ph("snap_fix", lambda: bench.snap_edges_fix(graph, snap))
    ph("distance", lambda: video.distance(beta=0.3, bidirectional=True))
    for k in range(6):
        ph("update%d" % k, lambda: graph.update(None, None, use_inactive=True))
for k, v in acc.items():
    print("%-22s %7.3f ms" % (k, v / N * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    snap.restore()
    newest = bench.NKF - 1
    pairs = [(i, j) for i, j in zip(graph._ii_h, graph._jj_h) if i == newest or j == newest]
    graph.rm_factors([(i == newest or j == newest) for i, j in zip(graph._ii_h, graph._jj_h)])
    graph.add_factors([p[0] for p in pairs], [p[1] for p in pairs])
    bench.snap_edges_fix(graph, snap)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
