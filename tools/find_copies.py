"""attribute the non-compute GPU kernels (copies, fills, casts, adds) of one graph update to Python call sites"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
video, graph = bench.make_window(dev)
for _ in range(6):
    graph.update(None, None, use_inactive=True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(4):
        graph.update(None, None, use_inactive=True)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
want = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::mul", "aten::index", "aten::index_put_", "aten::cat",
        "aten::clamp", "aten::softplus", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::empty", "aten::zeros")
for ev in prof.events():
    if ev.name in want and ev.device_time_total > 0 or ev.name in ("aten::copy_", "aten::fill_"):
        st = [f for f in ev.stack if "/root/repo" in f or "pvo_amd" in f or "bench.py" in f][:2]
        key = (ev.name, str(ev.input_shapes)[:60], " <- ".join(s.split("/")[-1][:60] for s in st))
        agg[key][0] += 1; agg[key][1] += ev.device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
for (name, shp, st), (n, us) in rows[:45]:
    print(f"{us/4:8.1f} us/upd  x{n/4:4.1f}  {name:18s} {shp:60s} {st}")
