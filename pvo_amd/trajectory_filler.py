"""PoseTrajectoryFiller - poses of the non-keyframe images (VO_Module/droid_slam/trajectory_filler.py:12-112).

Frames are processed in chunks of 16: each gets a pose interpolated on se(3) between its neighbouring keyframes,
is written temporarily behind the keyframes in the video buffers, connected to those two keyframes, and refined
by six motion-only updates (update operator + HIP BA with `motion_only=True`).
"""
import torch

from .factor_graph import FactorGraph
from .geom import se3 as lie
from .geom.se3 import SE3
from .graphs import GraphedCall


class PoseTrajectoryFiller:
    def __init__(self, net, video, device="cuda:0"):
        self.cnet, self.fnet, self.update = net.cnet, net.fnet, net.update
        self.count, self.video, self.device = 0, video, torch.device(device)
        self.MEAN = torch.as_tensor([0.485, 0.456, 0.406], device=self.device)[:, None, None]
        self.STDV = torch.as_tensor([0.229, 0.224, 0.225], device=self.device)[:, None, None]
        self.reuse_features = True
        self._ts_host = None
        from .motion_filter import _weights_guard
        self._one = GraphedCall(self._features_one, name="fnet (filler)", guard=_weights_guard(self.fnet))

    def _features_one(self, image_dev):
        """one frame [3,H,W] on the device -> [1,128,h,w]"""
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.device.type == "cuda"):
            enc = getattr(self.fnet, "forward_inference", None)
            if enc is not None and self.device.type == "cuda" and image_dev.dim() == 3 and image_dev.is_contiguous() \
                    and image_dev.dtype in (torch.int32, torch.uint8, torch.float32):
                from .droid_backends import frame_normalise
                from .motion_filter import MotionFilter
                return enc(frame_normalise(image_dev, MotionFilter._MEAN3, MotionFilter._STD3, torch.float16)[None]).squeeze(0)
            x = image_dev.flip(0)[None, None].float() / 255.0
            enc = enc if enc is not None else self.fnet
            return enc((x - self.MEAN) / self.STDV).squeeze(0)

    def _features(self, images):
        """feature maps of a chunk of frames [M,3,H,W].  On the GPU frame by frame through ONE captured graph: the encoder is
        instance-normalised (per image), so the result is the batched call's, MIOpen has no tuned kernel for the 16-image
        shapes (the batched call fell to its naive convolution: 3.1 ms per layer, bench.py `sequence`), and the one-image
        graph is the launch the motion filter's shapes already warmed."""
        if self.device.type != "cuda":
            images = torch.stack(list(images), 0) if not isinstance(images, torch.Tensor) else images
            with torch.autocast("cuda", dtype=torch.float16, enabled=False):
                x = images[None, :, [2, 1, 0]].to(self.device).float() / 255.0
                return self.fnet((x - self.MEAN) / self.STDV).squeeze(0)
        # (each frame goes up as it is and is never touched on the host: a torch.stack of 16 frames took 19 ms on a 128-core host)
        from .motion_filter import upload_frame
        return torch.cat([self._one(upload_frame(im, self.device)).clone() for im in images], 0)

    def _fill(self, tstamps, images, intrinsics):
        v = self.video
        N, M = v.counter, len(tstamps)
        on_gpu = self.device.type == "cuda"
        if not on_gpu or getattr(self.video, "images", True) is not None:
            images = torch.stack(images, 0)                # (on the GPU only a video that stores its frames needs them as one tensor)
        if on_gpu:
            # Nothing in a chunk reads the device back: the keyframes' time stamps come to the host ONCE per call (`_ts_host`), each
            # frame is bracketed there, and time stamps / indices / intrinsics go up through the pinned staging ring.  A `.tolist()` of
            # device indices (or a blocking upload) per chunk made the host wait for the previous chunk's six updates before it began
            # to prepare the next: the device idled through every chunk's bookkeeping (13 ms per chunk for 6 ms of device work).
            import numpy as np
            from .droid_backends import to_device_async
            ts_h = self._ts_host if self._ts_host is not None and len(self._ts_host) == N else v.tstamp[:N].cpu().numpy()
            tt_h = np.asarray([float(t) for t in tstamps], dtype=np.float32)
            t0_h = np.clip((ts_h[None, :] <= tt_h[:, None]).sum(axis=1) - 1, 0, None)
            t1_h = np.where(t0_h < N - 1, t0_h + 1, t0_h)
            tt = to_device_async(torch.from_numpy(tt_h), torch.float, self.device)
            both = to_device_async(torch.from_numpy(np.concatenate([t0_h, t1_h]).astype(np.int64)), torch.long, self.device)
            t0, t1 = both[:M], both[M:]
            t0_e, t1_e = t0_h.tolist(), t1_h.tolist()                            # the edges' endpoints as host lists: add_factors reads nothing back
            intrinsics = to_device_async(torch.stack(intrinsics, 0), torch.float, self.device)
        else:
            tt = torch.as_tensor(tstamps, device=self.device, dtype=torch.float)
            intrinsics = torch.stack(intrinsics, 0).to(self.device)
        ts = v.tstamp[:N]
        Ps = SE3(v.poses[:N])

        # bracket each time stamp by keyframes t0 <= t < t1 and interpolate with constant twist
        if not on_gpu:
            t0 = ((ts[None, :] <= tt[:, None]).sum(dim=1) - 1).clamp(min=0)
            t1 = torch.where(t0 < N - 1, t0 + 1, t0)
            t0_e, t1_e = t0, t1
        dt = ts[t1] - ts[t0] + 1e-3
        vel = (Ps[t1] * Ps[t0].inv()).log() / dt.unsqueeze(-1)
        Gs = SE3.exp(vel * (tt - ts[t0]).unsqueeze(-1)) * Ps[t0]

        # the motion filter encoded every one of these frames when it was tracked and the video kept the maps (same time stamp =
        # same frame: test_vo.py hands terminate() the stream it tracked; `reuse_features = False` re-encodes as the reference does)
        # ... checked per frame by a fingerprint of the image (DepthVideo.recall_features): a stream that reuses time stamps with
        # other images is re-encoded; entries are released as they are consumed
        recall = getattr(v, "recall_features", None) if self.reuse_features and getattr(v, "frame_fmaps", None) else None
        cached = [recall(t, im) for t, im in zip(tstamps, images)] if recall else [None]
        if all(c is not None for c in cached):
            fmap = torch.cat(cached, 0)
        elif self.device.type == "cuda" and any(c is not None for c in cached):
            fmap = torch.cat([c if c is not None else self._features([im]) for c, im in zip(cached, images)], 0)
        else:
            fmap = self._features(images)
        v.counter += M
        v[N:N + M] = (tt, images, Gs.data, 1.0, intrinsics / 8.0, fmap)

        graph = FactorGraph(v, self.update, self.device)
        new = list(range(N, N + M)) if on_gpu else torch.arange(N, N + M, device=self.device)
        graph.add_factors(t0_e, new)
        graph.add_factors(t1_e, new)
        for _ in range(6):
            graph.update(N, N + M, motion_only=True)
        Gs = SE3(v.poses[N:N + M].clone())
        v.counter -= M
        return [Gs]

    @torch.no_grad()
    def __call__(self, image_stream):
        """image_stream yields (tstamp, image, intrinsics, segments); returns SE3 [num_frames]"""
        pose_list, tstamps, images, intrinsics = [], [], [], []
        # (the keyframes' time stamps on the host, once: the one read-back of the call besides the result)
        self._ts_host = self.video.tstamp[:self.video.counter].cpu().numpy() if self.device.type == "cuda" else None
        for (tstamp, image, intrinsic, _segments) in image_stream:
            tstamps.append(tstamp); images.append(image); intrinsics.append(intrinsic)
            if len(tstamps) == 16:
                pose_list += self._fill(tstamps, images, intrinsics)
                tstamps, images, intrinsics = [], [], []
        if tstamps:
            pose_list += self._fill(tstamps, images, intrinsics)
        if hasattr(self.video, "forget_features"):
            self.video.forget_features()          # (frames tracked but not in this stream: nothing will ask for them any more)
        return lie.cat(pose_list, 0)
