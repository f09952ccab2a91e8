"""Where one S-T training step goes (BASELINE.json configs[4] at N = 1: tools/train.py's step, 6 frames 200x400, 20 edges, 15
unrolled updates, bf16 volume, fp32 operator / BA, Adam).   On the GPU box:

    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -f csv -d /tmp/ts -- python $GRAFT_REPO_ROOT/tools/prof_train_step.py
    python $GRAFT_REPO_ROOT/tools/prof_train_step.py --summarise /tmp/ts  > gpurun_out/r04_train_step_stats.txt

Without rocprofv3 it prints the step's wall time and the host-side time of its phases (forward / loss / backward / optimizer),
each closed by a device synchronisation."""
import csv, glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))

if "--summarise" in sys.argv:
    d = sys.argv[sys.argv.index("--summarise") + 1]
    rows = list(csv.DictReader(open(glob.glob(os.path.join(d, "*", "*kernel_stats.csv"))[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    groups = {}
    def group(n):
        if "corr_lookup" in n or "corr_build" in n or "altcorr" in n: return "pvo: correlation (lookup fwd / bwd, volume build)"
        if "se3_" in n: return "pvo: SE3 kernels (forward)"
        if "(anonymous namespace)" in n and "at::" not in n: return "pvo: other HIP kernels"
        if "Cijk_" in n or "gemm" in n.lower() or "ck::" in n or "igemm" in n or "MIOpen" in n or "miopen" in n or "conv" in n.lower(): return "PyTorch: convolutions / GEMMs (MIOpen, hipBLASLt, CK)"
        if "elementwise" in n or "vectorized" in n or "reduce" in n or "index" in n or "gather" in n or "scatter" in n or "cat" in n.lower() or "copy" in n.lower() or "fill" in n.lower(): return "PyTorch: element-wise / reductions / indexing / copies"
        return "other"
    for r in rows:
        g = groups.setdefault(group(r["Name"]), [0, 0.0]); g[0] += int(r["Calls"]); g[1] += float(r["TotalDurationNs"])
    print("kernel time of the profiled steps: %.1f ms in %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
    for k, (c, t) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        print("  %-70s %8d launches %9.1f ms  %5.1f %%" % (k, c, t / 1e6, 100 * t / tot))
    print("\ntop 25 kernels:")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:25]:
        print("  %8s x %9.1f us = %8.1f ms  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Name"][:150]))
    sys.exit(0)

import torch
import train as T
from pvo_amd.droid_net import DroidNet
from pvo_amd.geom import losses as L
from pvo_amd.geom.graph_utils import build_frame_graph
from pvo_amd.geom.se3 import SE3
from pvo_amd.synthetic import TrainClips

device = torch.device("cuda:0")
args = T.parse_args(["--device", "cuda"])
torch.manual_seed(0)
net = DroidNet().to(device).train()
opt = torch.optim.Adam(net.parameters(), lr=args.lr, weight_decay=1e-5)
ssim = L.SSIM().to(device)
reps = int(os.environ.get("PVO_TRAIN_REPS", "3"))
clips = TrainClips(6, (200, 400), length=reps + 1)
ph = {"forward": [], "loss": [], "backward": [], "optimizer": [], "step": []}
def lap(t0):
    torch.cuda.synchronize(); return time.perf_counter() - t0
for k in range(reps + 1):
    images, poses, disps, intr, gt_masks, gt_vals, segments = [x[None].to(device) for x in clips[k]]
    graph = build_frame_graph(poses, disps, intr, num=20, need_inv=False)
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    opt.zero_grad()
    Ps = SE3(poses); Gs = SE3.IdentityLike(Ps)
    Gs.data[:, 0] = Ps.data[:, 0]; Gs.data[:, 1:] = Ps.data[:, [1]]
    t0 = time.perf_counter()
    out = net(Gs, images, torch.ones_like(disps[:, :, 3::8, 3::8]), intr / 8.0, graph, num_steps=15, fixedp=2, ret_flow=True,
              downsample=True, segments=segments, corr_dtype=torch.bfloat16)
    f = lap(t0); t0 = time.perf_counter()
    loss, _ = T.objective(args, L, out, (images, Ps, disps, intr, gt_masks, gt_vals), graph, ssim, 0)
    l = lap(t0); t0 = time.perf_counter()
    loss.backward()
    b = lap(t0); t0 = time.perf_counter()
    torch.nn.utils.clip_grad_norm_(net.parameters(), args.clip); opt.step()
    o = lap(t0)
    if k > 0:
        for name, v in (("forward", f), ("loss", l), ("backward", b), ("optimizer", o), ("step", lap(t_all) if False else f + l + b + o)):
            ph[name].append(v)
import statistics
print("S-T training step (median of %d after one warm-up): " % reps + ", ".join("%s %.1f ms" % (k, statistics.median(v) * 1e3) for k, v in ph.items()))
