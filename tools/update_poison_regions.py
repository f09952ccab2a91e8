"""Which stage of pvo_graph_update picks up stale LDS / register contents?  (on the GPU box)
One update from a fixed state, clean and after filling LDS + registers with a pattern; the update's workspace (every
intermediate tensor, layout of carve_up / carve_op in update_exec.hip) is compared region by region, in dataflow order."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                          # noqa: E402
exec(open(os.path.join(ROOT, "tools", "ba_poison_check.py")).read().split("bad = 0")[0])     # builds libpoison, defines poison()

video, graph = bench.make_window(dev, seed=0)
names = ("net", "target_cam", "weight", "raw_mask", "delta_dy")
state = {n: getattr(graph, n).clone() for n in names}
p0, d0, dm0 = video.poses.clone(), video.disps.clone(), graph.damping.clone()
graph.update(None, None, use_inactive=True)
st = graph._cache["fused"]
ws = st["ws"]
E, H, W = len(graph._ii_h), graph.ht, graph.wd
K, R = st["seg"][2], st["R"]
al = lambda x: (x + 255) & ~255
px, kpx = E * H * W, K * H * W
chunks = (H * W + 255) // 256
up = [("coords", px * 8), ("valid", px * 4), ("eta", R * H * W * 4), ("motion", px * 16), ("heads", px * 16), ("upmask", K * H * W * 576 * 2),
      ("vote_tot", 4), ("vote_dyn", 4)]
op = [("c1 lookup+enc0", px * 256), ("f1 conv7x7", px * 256), ("CF cenc2|fenc2", px * 384), ("Z gates", px * 256), ("RN gates", px * 256),
      ("h1 heads z", px * 288), ("a1 agg conv1", px * 256), ("am segment mean", kpx * 256), ("a2 agg conv2", kpx * 256), ("P_zr", px * 512), ("P_q", px * 256),
      ("part glo", E * chunks * 128 * 4), ("g gate context", E * 384 * 4)]
base = (ws.data_ptr() + 255) & ~255
off = base - ws.data_ptr()
regions = []
for n, b in up:
    regions.append((n, off, b)); off += al(b)
off = ((ws.data_ptr() + off + 255) & ~255) - ws.data_ptr()
for n, b in op:
    regions.append((n, off, b)); off += al(b)
assert off <= ws.numel(), (off, ws.numel())
order = ["coords", "motion", "c1 lookup+enc0", "f1 conv7x7", "CF cenc2|fenc2", "part glo", "g gate context", "Z gates", "RN gates", "NET (candidate)",
         "h1 heads z", "heads", "a1 agg conv1", "am segment mean", "a2 agg conv2", "eta", "upmask"]


def run(before=None):
    for n in names:
        getattr(graph, n).copy_(state[n])
    video.poses.copy_(p0); video.disps.copy_(d0); graph.damping.copy_(dm0)
    torch.cuda.synchronize()
    if before is not None:
        poison(before)
    graph.update(None, None, use_inactive=True)
    torch.cuda.synchronize()
    out = {n: ws[o:o + b].clone() for n, o, b in regions}
    out["NET (candidate)"] = graph.net[0].permute(0, 2, 3, 1).contiguous().view(torch.uint8).reshape(-1)
    return out


ref = run()
assert all(torch.equal(run()[k], ref[k]) for k in order), "clean runs differ"
for trial in range(6):
    pat = (0xFFFFFFFF, 0x9E3779B9)[trial & 1]
    out = run(before=pat)
    diff = [k for k in order if not torch.equal(out[k], ref[k])]
    print("poison %08x: %s" % (pat, "all regions equal" if not diff else "first differing stage: %s   (all: %s)" % (diff[0], ", ".join(
        "%s [%d bytes]" % (k, int((out[k] != ref[k]).sum())) for k in diff))), flush=True)
