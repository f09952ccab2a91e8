"""host-side (launch) time vs GPU time of one graph update at S-B"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
video, graph = bench.make_window(dev)
for _ in range(5):
    graph.update(None, None, use_inactive=True)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N):
    graph.update(None, None, use_inactive=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host launch time per update {1e3*(t1-t0)/N:.3f} ms ; wall per update {1e3*(t2-t0)/N:.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    graph.update(None, None, use_inactive=True)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(30)
