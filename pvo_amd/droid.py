"""Droid - the VO system object: motion filter -> frontend (local BA) -> backend (global BA) -> trajectory filler.

Counterpart of the reference's Droid (VO_Module/droid_slam/droid.py:19-124): same constructor argument object
(the fields of evaluation_scripts/test_vo.py:58-83), `track`, `terminate`, `get_traj`, `get_depth`, `get_flow`.
There is no visualiser process and no shared-memory video: one process owns one GPU.  With `args.weights = None`
the network keeps its seeded default initialisation (this image has no checkpoints).
"""
from argparse import Namespace
from collections import OrderedDict

import torch

from .backend import DroidBackend
from .depth_video import DepthVideo
from .droid_net import DroidNet, upsample_inter
from .frontend import DroidFrontend
from .geom.se3 import SE3
from .motion_filter import MotionFilter
from .trajectory_filler import PoseTrajectoryFiller


def default_args(**over):
    """the defaults of evaluation_scripts/test_vo.py:58-83"""
    a = dict(device="cuda:0", weights=None, buffer=1024, image_size=[240, 808], disable_vis=True, use_aff_bri=False,
             beta=0.6, filter_thresh=1.75, warmup=12, keyframe_thresh=2.25, frontend_thresh=12.0, frontend_window=25,
             frontend_radius=2, frontend_nms=1, backend_thresh=15.0, backend_radius=2, backend_nms=3,
             segm_filter=False, thresh=0.8, half_update=True, pipelined=False)
    a.update(over)
    return Namespace(**a)


class Droid:
    def __init__(self, args):
        self.args = args
        self.load_weights(args.weights, args.use_aff_bri)
        self.video = DepthVideo(args.image_size, args.buffer, args.device, args.segm_filter, args.thresh)
        self.filterx = MotionFilter(self.net, self.video, thresh=args.filter_thresh, device=args.device)
        self.filterx.overlap_upload = bool(getattr(args, "pipelined", False))
        self.frontend = DroidFrontend(self.net.update, self.video, args.device, warmup=args.warmup, beta=args.beta,
                                      frontend_nms=args.frontend_nms, keyframe_thresh=args.keyframe_thresh,
                                      frontend_window=args.frontend_window, frontend_thresh=args.frontend_thresh,
                                      frontend_radius=args.frontend_radius)
        self.filterx.before_context = self.frontend.keyframe_ahead
        self.backend = DroidBackend(self.net, self.video, args)
        self.traj_filler = PoseTrajectoryFiller(self.net, self.video, args.device)

    def load_weights(self, weights, use_aff_bri=False):
        """droid.py:55-62; DataParallel's "module." prefix is stripped"""
        self.net = DroidNet(use_aff_bri)
        if weights is not None:
            sd = torch.load(weights, map_location=self.args.device)
            self.net.load_state_dict(OrderedDict((k.replace("module.", ""), v) for k, v in sd.items()))
        self.net.to(self.args.device).eval()
        if getattr(self.args, "half_update", True) and torch.device(self.args.device).type == "cuda":
            self.net.update.half()          # the fused 16-bit operator path
            if getattr(self.args, "half_encoders", True):
                # the encoders run under fp16 autocast (motion_filter.py:50): every convolution, norm and add is an fp16 operation
                # either way, but fp32 parameters are cast again in EVERY forward - 33 tiny kernels of ~120 per frame.  Cast once.
                self.net.fnet.half(); self.net.cnet.half()

    def track(self, tstamp, image, depth=None, intrinsics=None, segments=None):
        """one frame (droid.py:64-75).  args.pipelined (default False: the reference's order, the video is final for this frame when
        the call returns): the frame's graph is launched FIRST, then the second half of the previous keyframe's frontend update (its
        keyframe test was left in flight when the previous call returned), then the motion test is read and this frame's frontend
        work is issued up to ITS keyframe test.  Same operations on the same data in the same dependency order - poses, depths and the
        trajectory are bit-identical (tests/test_vo_system.py) - but the device has work queued while the host waits for a scalar
        or books edges.  The video then lags by half a keyframe update between calls: `flush()` (called by terminate / get_*)
        completes it."""
        with torch.no_grad():
            if not getattr(self.args, "pipelined", False):
                self.filterx.track(tstamp, image, depth, intrinsics, segments)
                self.frontend()
                return
            self.filterx.begin(tstamp, image, depth, intrinsics, segments)
            self.frontend.finish()
            self.filterx.finish()
            self.frontend.begin()

    def flush(self):
        """complete a keyframe update a pipelined track() left half done"""
        fe = getattr(self, "frontend", None)
        if fe is not None:
            with torch.no_grad():
                fe.finish()

    def terminate(self, stream=None, need_inv=True):
        """two global BA passes, then fill in every frame's pose; returns [num_frames, 7] (t, q) (droid.py:77-98)"""
        self.flush()
        self.filterx.before_context = None          # (a bound method of the frontend: it would keep the frontend's volumes alive)
        del self.frontend
        self._release_cached_memory()
        self.backend(7)
        self._release_cached_memory()
        self.backend(12)
        traj = self.traj_filler(stream)
        return (traj.inv() if need_inv else traj).data.cpu().numpy()

    def _release_cached_memory(self):
        """droid.py:84,88 empty the allocator's cache before each global BA - on the 11-24 GB GPUs the reference runs on the
        frontend's volumes have to go first.  On MI355X that is 50 ms of hipFree per call (and the backend then allocates again)
        for nothing while HBM is mostly free: only done when less than a quarter of the device memory is available."""
        if torch.device(self.args.device).type != "cuda":
            return
        free, total = torch.cuda.mem_get_info(torch.device(self.args.device))
        if free < 0.25 * total:
            torch.cuda.empty_cache()

    def get_traj(self):
        self.flush()
        return SE3(self.video.poses[:self.video.counter]).data.cpu().numpy()

    def get_depth(self):
        self.flush()
        d = self.video.disps[:self.video.counter]
        return upsample_inter(d[None, ..., None]).squeeze(4).squeeze(0)

    def get_flow(self):
        self.flush()
        return upsample_inter(self.video.full_flow[:self.video.counter][None] * 8)
