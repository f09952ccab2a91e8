"""Does any kernel of a graph update read LDS / registers it has not written?  (on the GPU box)  python tools/update_poison_check.py
Same idea as tools/ba_poison_check.py for the whole of pvo_graph_update: one update from a fixed state, clean, then with every
compute unit's LDS and registers filled with a bit pattern just before it, then with short scribbling workgroups running beside
it on a third stream; the hidden state, the heads' outputs, the damping and the poses are compared bit for bit."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                          # noqa: E402
sys.argv = [sys.argv[0]]
exec(open(os.path.join(ROOT, "tools", "ba_poison_check.py")).read().split("bad = 0")[0])     # builds libpoison, defines poison()

video, graph = bench.make_window(dev, seed=0)
names = ("net", "target_cam", "weight", "raw_mask", "delta_dy")
state = {n: getattr(graph, n).clone() for n in names}
p0, d0, dm0 = video.poses.clone(), video.disps.clone(), graph.damping.clone()


def run(before=None, beside=None, n_updates=int(os.environ.get("PVO_CHECK_UPDATES", "2"))):
    for n in names:
        getattr(graph, n).copy_(state[n])
    video.poses.copy_(p0); video.disps.copy_(d0); graph.damping.copy_(dm0)
    torch.cuda.synchronize()
    if before is not None:
        poison(before)
    for _ in range(n_updates):
        if beside is not None:
            side.wait_stream(torch.cuda.current_stream(dev))
            for _ in range(40):
                poison(beside, stream=side, lds=24 * 1024, blocks=2048, spin=3)
        graph.update(None, None, use_inactive=True)
    torch.cuda.synchronize()
    return dict(net=graph.net.clone(), target=graph.target_cam.clone(), weight=graph.weight.clone(), raw_mask=graph.raw_mask.clone(),
                damping=graph.damping.clone(), poses=video.poses.clone(), disps=video.disps.clone())


run(); ref = run()
rows = [("clean, repeated", run())]
for pat in (0xFFFFFFFF, 0x7F800000, 0x9E3779B9):
    rows.append(("LDS + registers = %08x before" % pat, run(before=pat)))
    for k in range(3):
        rows.append(("scribbling workgroups (%08x) beside, run %d" % (pat, k), run(beside=pat)))
bad = 0
for what, out in rows:
    diff = [k for k in ref if not torch.equal(out[k].view(torch.int16 if out[k].dtype == torch.float16 else torch.int32),
                                              ref[k].view(torch.int16 if ref[k].dtype == torch.float16 else torch.int32))]
    bad += 1 if diff else 0
    print("%-52s %s" % (what, "bitwise equal" if not diff else "DIFFERS in " + ", ".join(
        "%s (max %.3g)" % (k, (out[k].float() - ref[k].float()).abs().max().item()) for k in diff)), flush=True)
print("result:", "no dependence on stale LDS / register contents" if bad == 0 else "%d runs differ" % bad)
