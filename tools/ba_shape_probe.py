"""BA time against the map shape and the window length (events; 2 Gauss-Newton steps per call): does a map whose pixel count is not a
multiple of four (30 x 101, the reference driver's own shape) cost the Schur kernel its 16-byte loads?   (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pvo_amd import droid_backends as db
from pvo_amd.geom.se3 import SE3
dev = torch.device("cuda:0")


def _scene(seed, P, ht, wd, radius, t0):
    """a window in the S-B recipe: targets = the reprojection of the true geometry + noise (HIP reprojection)"""
    g = torch.Generator().manual_seed(seed)
    intr = torch.tensor([wd * 0.625, wd * 0.625, wd / 2.0, ht / 2.0])
    xi = torch.tensor([0.05, 0.0, 0.02, 0.0, 0.01, 0.0])
    poses_gt = torch.stack([SE3.exp(k * xi).data for k in range(P)], 0)
    low = torch.rand(1, 1, 6, 8, generator=g) * 0.8 + 0.2
    disps_gt = torch.nn.functional.interpolate(low, size=(ht, wd), mode="bilinear", align_corners=True)[0, 0][None].repeat(P, 1, 1)
    e = [(i, j) for i in range(P) for j in range(P) if i != j and abs(i - j) <= radius]
    ii, jj = torch.tensor([a for a, _ in e]), torch.tensor([b for _, b in e])
    c, _ = db.reproject(poses_gt.to(dev), disps_gt.to(dev).contiguous(), intr[None].repeat(P, 1).to(dev), ii.to(dev), jj.to(dev))
    target = c.cpu() + 0.1 * torch.randn(len(e), ht, wd, 2, generator=g)
    weight = torch.rand(len(e), ht, wd, 2, generator=g)
    return dict(intr=intr, poses=torch.stack([poses_gt[max(k - 1, 0)] for k in range(P)], 0), disps=torch.ones(P, ht, wd),
                target=target.permute(0, 3, 1, 2).contiguous(), weight=weight.permute(0, 3, 1, 2).contiguous(),
                eta=torch.full((P, ht, wd), 1e-4) + 0.01 * torch.rand(P, ht, wd, generator=g), ii=ii, jj=jj)


for nf, rad in ((10, 3), (26, 2)):
    for ht, wd in ((30, 101), (30, 102), (30, 100), (48, 64)):
        s = _scene(0, nf, ht, wd, rad, 1)
        d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in s.items()}
        buf = 1024
        poses = torch.zeros(buf, 7, device=dev); poses[:, 6] = 1; poses[:nf] = d["poses"]
        disps = torch.ones(buf, ht, wd, device=dev); disps[:nf] = d["disps"]
        def run():
            p, z = poses.clone(), disps.clone()
            db.ba(p, z, d["intr"], d["target"], d["weight"], d["eta"], d["ii"], d["jj"], 1, nf, 2, 1e-4, 0.1, False)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        base0, base1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        base0.record()
        for _ in range(20):
            p, z = poses.clone(), disps.clone()
        base1.record()
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        print("keyframes %2d  edges %3d  map %2dx%-3d (HW %% 4 = %d): %7.1f us per BA call (2 steps), clones %.1f" % (
            nf, d["ii"].shape[0], ht, wd, (ht * wd) % 4, (e0.elapsed_time(e1) - base0.elapsed_time(base1)) * 50, base0.elapsed_time(base1) * 50))
