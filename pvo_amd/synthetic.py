"""Synthetic closed-loop VO scenes (no dataset or checkpoint is available for the reference:
/root/reference/checkpoints and datasets hold only READMEs).

A static world made of planes is viewed along a smooth trajectory; depth maps are exact ray/plane
intersections, so ground-truth correspondences between any two keyframes follow from geometry.  An
`OracleFlowOperator` stands in for the learned update operator: it answers every graph update with
the ground-truth correspondence field (+ fixed noise) and a confidence, which turns the frontend +
dense BA into a closed loop whose trajectory error can be measured (ATE-RMSE after Sim(3) alignment,
as test_vo.py:162-163 does with evo) and compared between the HIP path and the CPU oracle path.
"""
import math

import torch
import torch.utils.data

from .geom.se3 import SE3


class PlaneScene:
    def __init__(self, ht=48, wd=64, n_frames=24, seed=0, step=0.12, pattern=None):
        """pattern (optional): per-frame multipliers of `step`, repeated - e.g. (1, 1, 0.15) makes every third frame barely move, which
        the frontend's keyframe test (droid_frontend.py:54-58) then removes again: the rm_keyframe branch in a closed loop"""
        g = torch.Generator().manual_seed(seed)
        self.ht, self.wd, self.n = ht, wd, n_frames
        self.intr = torch.tensor([wd * 0.625, wd * 0.625, wd / 2.0, ht / 2.0])
        # planes n.X = d (world): a tilted far wall, a ground plane, a side wall
        self.planes = [(torch.tensor([0.15, 0.05, 1.0]), 4.0), (torch.tensor([0.0, 1.0, 0.12]), 1.3),
                       (torch.tensor([1.0, 0.0, 0.35]), 3.2)]
        xi = []
        a_run = 0.0
        for k in range(n_frames):
            a = k * step if pattern is None else a_run
            a_run += step * (1.0 if pattern is None else pattern[k % len(pattern)])
            xi.append(torch.tensor([-a, 0.03 * math.sin(1.3 * a), -0.25 * a, 0.02 * math.sin(a), -0.06 * a, 0.01 * a]))
        self.poses = torch.stack([SE3.exp(x).data for x in xi])         # world-to-camera, frame 0 = identity
        self.disps = torch.stack([self.render_disp(self.poses[k]) for k in range(n_frames)])
        self.noise = [torch.randn(ht, wd, 2, generator=g) for _ in range(8)]

    def render_disp(self, pose):
        ht, wd = self.ht, self.wd
        fx, fy, cx, cy = self.intr.tolist()
        y, x = torch.meshgrid(torch.arange(ht).float(), torch.arange(wd).float(), indexing="ij")
        rays_c = torch.stack([(x - cx) / fx, (y - cy) / fy, torch.ones_like(x)], -1)       # z_cam = 1
        G = SE3(pose).inv()                                                                # camera-to-world
        centre = G.data[:3]
        rays_w = G.act(torch.cat([rays_c, torch.zeros(ht, wd, 1)], -1))[..., :3]           # rotate only (w = 0)
        lam = torch.full((ht, wd), float("inf"))
        for n, d in self.planes:
            denom = (rays_w * n).sum(-1)
            l = (d - (centre * n).sum()) / denom
            l = torch.where((l > 0.05) & (denom.abs() > 1e-6), l, torch.full_like(l, float("inf")))
            lam = torch.minimum(lam, l)
        return 1.0 / lam                                                                   # disparity = 1 / depth


class OracleFlowOperator:
    """stand-in for DynamicUpdateModule: returns the ground-truth correspondences as its flow revision.
    reproject_fn(poses, disps, intrinsics[F,4], ii, jj) -> coords [E,h,w,2] (HIP kernel or CPU oracle)."""

    def __init__(self, scene, video, reproject_fn, noise=0.05, conf=4.0, eta=1e-3):
        self.scene, self.video, self.reproject_fn = scene, video, reproject_fn
        self.noise, self.conf, self.eta = noise, conf, eta
        dev = video.poses.device
        F = video.poses.shape[0]
        self.gt_poses = torch.zeros(F, 7); self.gt_poses[:, 6] = 1
        self.gt_disps = torch.ones(F, scene.ht, scene.wd)
        self.frame_of = {}            # video slot -> scene frame (slots are renumbered when keyframes are dropped)
        self.dev = dev

    def parameters(self):
        return iter(())

    def bind(self, slot, frame):
        self.frame_of[slot] = frame

    def _gt_tables(self, nslots):
        for s in range(nslots):
            f = self.frame_of.get(s, s)
            self.gt_poses[s] = self.scene.poses[f]; self.gt_disps[s] = self.scene.disps[f]
        return self.gt_poses.to(self.dev), self.gt_disps.to(self.dev)

    def __call__(self, net, inp, corr, motn, ii, jj, flag=False, **kw):
        v = self.video
        E = ii.shape[0]
        gp, gd = self._gt_tables(v.counter)
        gt = self.reproject_fn(gp, gd, v.intrinsics, ii, jj)
        cur = self.reproject_fn(v.poses, v.disps, v.intrinsics, ii, jj)
        ii_l, jj_l = ii.tolist(), jj.tolist()
        nz = torch.stack([self.scene.noise[(3 * i + j) % 8] for i, j in zip(ii_l, jj_l)]).to(self.dev)
        delta = torch.zeros(1, E, self.scene.ht, self.scene.wd, 4, device=self.dev)
        delta[0, ..., 0:2] = gt + self.noise * nz - cur
        weight = torch.full((1, E, self.scene.ht, self.scene.wd, 2), self.conf, device=self.dev)
        K = len(set(ii_l))
        eta = torch.full((1, K, self.scene.ht, self.scene.wd), self.eta, device=self.dev)
        delta_m = torch.zeros(1, E, self.scene.ht, self.scene.wd, 2, device=self.dev)
        return net, delta, weight, eta, {}, delta_m


def run_sequence(scene, video, frontend, operator, n_frames=None, backend=None, backend_steps=(7, 12)):
    """feed the scene's frames as keyframes (the motion filter is outside this path) and run the frontend; with `backend` (a
    DroidBackend over the same video / operator) the two global bundle adjustments of Droid.terminate (droid.py:84-90) follow.
    Returns (poses of the kept keyframes, the scene frame each of them is) - with a backend: (poses before, poses after, frames)."""
    n_frames = n_frames or scene.n
    dev = video.poses.device
    h, w = scene.ht, scene.wd
    g = torch.Generator().manual_seed(1)
    for k in range(n_frames):
        slot = video.counter
        operator.bind(slot, k)
        video.append(float(k), None if k else scene.poses[0].to(dev), None, scene.intr.to(dev),
                     torch.randn(h, w, 128, generator=g).half().to(dev),
                     torch.zeros(128, h, w, dtype=torch.half, device=dev), torch.zeros(128, h, w, dtype=torch.half, device=dev))
        frontend()
        # slots above a dropped keyframe move down by one (rm_keyframe)
        if video.counter <= slot:
            operator.frame_of.pop(slot, None)
            operator.bind(video.counter - 1, k)
    frames = [operator.frame_of.get(s, s) for s in range(video.counter)]
    before = video.poses[:video.counter].detach().cpu().clone()
    if backend is None:
        return before, frames
    for steps in backend_steps:
        backend(steps)
    return before, video.poses[:video.counter].detach().cpu().clone(), frames


class TrainClips(torch.utils.data.Dataset):
    """Synthetic training clips in the item layout of the reference's VKITTI2 reader as train.py consumes it
    (train.py:113-116, mode 'semisup'): images [N,3,H,W] (0..255), poses [N,7] world-to-camera, disps [N,H,W], intrinsics
    [N,4] at image resolution, gt_masks [N,H,W,1] (1 static), gt_vals [N,H,W,1], segments [N,H,W].  The world is
    PlaneScene's planes with a texture that is a function of the 3-D surface point, so photometric and geometric losses
    are consistent; a rectangle that carries its own texture phase per frame plays the dynamic object.  No dataset can be
    read in this environment (datasets/ holds a README only); `crop_size` defaults to train.py's [200, 400] (S-T)."""

    def __init__(self, n_frames=6, crop_size=(200, 400), length=64, seed=0, step=0.05):
        self.n, (self.ht, self.wd), self.length, self.seed, self.step = n_frames, crop_size, length, seed, step

    def __len__(self):
        return self.length

    def _texture(self, pts, phase):
        k = torch.tensor([[2.1, 0.7, 1.3], [0.9, 2.6, 0.4], [1.7, 1.1, 2.3]])
        s = torch.sin(pts @ k.T * 3.0 + phase) * torch.cos(pts @ k.flip(0).T * 1.7)
        return 127.5 + 110.0 * s                                                         # [H,W,3]

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 100003 + idx)
        scene = PlaneScene(self.ht, self.wd, self.n, seed=idx, step=self.step * (0.6 + 0.8 * torch.rand(1, generator=g).item()))
        ht, wd = self.ht, self.wd
        fx, fy, cx, cy = scene.intr.tolist()
        y, x = torch.meshgrid(torch.arange(ht).float(), torch.arange(wd).float(), indexing="ij")
        rays = torch.stack([(x - cx) / fx, (y - cy) / fy, torch.ones_like(x), torch.zeros_like(x)], -1)
        images, masks = [], []
        y0, x0 = int(ht * 0.55), int(wd * (0.2 + 0.5 * torch.rand(1, generator=g).item()))
        for k in range(self.n):
            Gi = SE3(scene.poses[k]).inv()
            pts_c = rays.clone()
            pts_c[..., :3] = rays[..., :3] / scene.disps[k][..., None]                    # camera-frame points
            pts_c[..., 3] = 1.0
            pts_w = Gi.act(pts_c)[..., :3]
            img = self._texture(pts_w, torch.tensor([0.0, 1.0, 2.0]))
            m = torch.ones(ht, wd, 1)
            ys, xs = slice(y0, y0 + ht // 6), slice(x0 + 3 * k, x0 + 3 * k + wd // 8)     # the moving patch
            img[ys, xs] = self._texture(pts_w[ys, xs] * 2.0, torch.tensor([0.5 * k, 1.0, 3.0]))
            m[ys, xs] = 0.0
            images.append(img.permute(2, 0, 1).clamp(0, 255))
            masks.append(m)
        segments = torch.zeros(self.n, ht, wd, dtype=torch.int32)
        return (torch.stack(images), scene.poses.clone(), scene.disps.clone(), scene.intr[None].repeat(self.n, 1),
                torch.stack(masks), torch.ones(self.n, ht, wd, 1), segments)


def drifting_texture_stream(n_frames, ht=240, wd=808, seed=0, segments=True, fast=9, slow=1, period=4):
    """A seeded image stream in the item layout of evaluation_scripts/test_vo.py:19-56 - (t, image [3,H,W] int BGR 0..255,
    intrinsics [4], segm [1,1,H/8,W/8] int or None) - at the reference driver's own input size (240 x 808).  A smoothed random
    texture drifts under the window, `fast` pixels per frame except every `period`-th frame (`slow` pixels: frames a motion
    filter would drop); the panoptic labels are eight rectangles of which two move (the S-3 pattern of SURVEY.md 8d).  The
    content carries no geometry - no checkpoint exists in this environment, so the frames only have to exercise every stage
    of the pipeline (bench.py `sequence`)."""
    g = torch.Generator().manual_seed(seed)
    span = fast * n_frames + wd + 64
    big = torch.randint(0, 256, (3, ht + 32, span), generator=g).float()
    big = torch.nn.functional.avg_pool2d(big[None], 5, stride=1, padding=2)[0]
    big = ((big - big.mean()) * 3.0 + 127.5).clamp(0, 255)
    intr = torch.tensor([wd * 0.9, wd * 0.9, wd / 2.0, ht / 2.0])
    h8, w8 = ht // 8, wd // 8
    x = 0
    for t in range(n_frames):
        x += slow if (t % period == period - 1) else fast
        image = big[:, 8:8 + ht, x:x + wd].round().int().contiguous()              # int32, as test_vo.py:41 hands it over
        segm = None
        if segments:
            seg = torch.zeros(h8, w8, dtype=torch.int32)
            bh, bw = h8 // 2, w8 // 4
            for n in range(8):
                r, c = divmod(n, 4)
                shift = (t % 8) if n + 1 in (3, 6) else 0
                y0, x0 = r * bh + 2, c * bw + 2 + shift
                seg[y0:y0 + bh - 4, max(x0, 0):min(x0 + bw - 4, w8)] = 1000 * (n + 1) + 7      # raw ids: category * 1000 + instance
            segm = seg[None, None]
        yield t, image, intr.clone(), segm
