"""MotionFilter - per-frame feature extraction and the "enough motion?" keyframe test.

Counterpart of the reference's MotionFilter (VO_Module/droid_slam/motion_filter.py:12-109): `track`
adds the first frame unconditionally and afterwards only frames whose one-step flow estimate against
the last keyframe exceeds `thresh` pixels (mean norm); `track_vo` adds every frame.  The one-step
estimate is a 1-edge correlation volume (HIP build), a lookup at the identity grid (HIP lookup) and one
pass of the update operator; one scalar is read back per frame.
"""
import torch

from .geom.projective_ops import coords_grid
from .modules.corr import CorrBlock


class MotionFilter:
    def __init__(self, net, video, thresh=2.5, device="cuda:0"):
        self.cnet, self.fnet, self.update = net.cnet, net.fnet, net.update
        self.video, self.thresh, self.device = video, thresh, torch.device(device)
        self.count = 0
        self.MEAN = torch.as_tensor([0.485, 0.456, 0.406], device=self.device)[:, None, None]
        self.STDV = torch.as_tensor([0.229, 0.224, 0.225], device=self.device)[:, None, None]
        self.net = self.inp = self.fmap = None

    def _autocast(self):
        return torch.autocast("cuda", dtype=torch.float16, enabled=self.device.type == "cuda")

    def _context(self, x):
        net, inp = self.cnet(x).split([128, 128], dim=2)
        return net.tanh().squeeze(0), inp.relu().squeeze(0)

    def _normalise(self, image):
        x = image[None, None, [2, 1, 0]].to(self.device).float() / 255.0
        return (x - self.MEAN) / self.STDV

    def _append(self, tstamp, image, pose, disp, intrinsics, gmap, net, inp, segments):
        kw = {}
        if tuple(gmap.shape[-2:]) == (128, 128):      # a [128,128,128] map: the layout cannot be read off the shape
            kw["channels_last"] = False
        self.video.append(tstamp, pose, disp, intrinsics / 8.0, gmap[0], net[0], inp[0], segm=segments, image=image, **kw)

    @torch.no_grad()
    def track(self, tstamp, image, depth=None, intrinsics=None, segments=None):
        """run on every incoming frame (motion_filter.py:46-87); image [3,H,W] BGR 0..255"""
        ident = torch.as_tensor([0, 0, 0, 0, 0, 0, 1.0], device=self.device)
        ht, wd = image.shape[-2] // 8, image.shape[-1] // 8
        with self._autocast():
            x = self._normalise(image)
            gmap = self.fnet(x).squeeze(0)                                     # [1,128,h,w]
            if self.video.counter == 0:
                net, inp = self._context(x)
                self.net, self.inp, self.fmap = net, inp, gmap
                self._append(tstamp, image, ident, 1.0, intrinsics.to(self.device), gmap, net, inp, segments)
                return True
            coords0 = coords_grid(ht, wd, device=self.device)[None, None]
            half = lambda t: t if t.dtype in (torch.float16, torch.bfloat16) or self.device.type != "cuda" else t.half()
            corr = CorrBlock(half(self.fmap[None]), half(gmap[None]))(coords0)
            _, delta, _, _ = self.update(self.net[None], self.inp[None], corr)
            if delta[..., 0:2].float().norm(dim=-1).mean().item() > self.thresh:
                self.count = 0
                net, inp = self._context(x)
                self.net, self.inp, self.fmap = net, inp, gmap
                self._append(tstamp, image, None, None, intrinsics.to(self.device), gmap, net, inp, segments)
                return True
            self.count += 1
            return False

    @torch.no_grad()
    def track_vo(self, tstamp, image, depth=None, intrinsics=None, segments=None):
        """every frame becomes a keyframe (motion_filter.py:89-109)"""
        ident = torch.as_tensor([0, 0, 0, 0, 0, 0, 1.0], device=self.device)
        with self._autocast():
            x = self._normalise(image)
            gmap = self.fnet(x).squeeze(0)
            net, inp = self._context(x)
        first = self.video.counter == 0
        self._append(tstamp, image, ident if first else None, 1.0 if first else None, intrinsics.to(self.device),
                     gmap, net, inp, segments)
        return True
