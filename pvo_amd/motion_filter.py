"""MotionFilter - per-frame feature extraction and the "enough motion?" keyframe test.

Counterpart of the reference's MotionFilter (VO_Module/droid_slam/motion_filter.py:12-109): `track`
adds the first frame unconditionally and afterwards only frames whose one-step flow estimate against
the last keyframe exceeds `thresh` pixels (mean norm); `track_vo` adds every frame.  The one-step
estimate is a 1-edge correlation volume (HIP build), a lookup at the identity grid (HIP lookup) and one
pass of the update operator; one scalar is read back per frame.

Per-frame host cost (round 5, bench.py `sequence`): the frame goes up as it arrives, the BGR flip / scaling happen on the device,
and each encoder is ONE HIP-graph launch after its first two calls
(pvo_amd/graphs.py: the eager ~70-launch forward cost 4.6 ms of host time per network for ~0.3 ms of device work).
"""
import torch

from .geom.projective_ops import coords_grid
from .graphs import GraphedCall
from .modules.corr import CorrBlock


def _weights_guard(module):
    """what a captured graph of `module` stays valid for: the storage and dtype of its parameters (.half() / .to() re-allocate them;
    load_state_dict copies in place, which a replay sees)"""
    def guard():
        p = next(iter(module.parameters()), None) if hasattr(module, "parameters") else None
        return None if p is None else (p.data_ptr(), p.dtype)
    return guard


def _operator_guard(module):
    """what a captured graph that runs the UPDATE OPERATOR stays valid for: the operator's re-arranged weights
    (DynamicUpdateModule.packed_weights) are rebuilt - and the old pack freed - whenever a parameter's version counter moves, an
    in-place load_state_dict included, and a capture has the old pack's addresses baked in: every parameter's version is part of the key"""
    def guard():
        ps = module.__dict__.get("_param_list") if hasattr(module, "__dict__") else None
        if ps is None:
            ps = list(module.parameters()) if hasattr(module, "parameters") else []
        return None if not ps else (ps[0].data_ptr(), ps[0].dtype, tuple(p._version for p in ps))
    return guard


def upload_frame(image, device):
    """pageable host tensor -> device, a plain blocking copy on the current stream.
    Measured on an MI355X box (2.3 MB int32 frame, 1.7 ms of kernels queued on the stream in front of it): this form 1.78 ms per
    frame all in, non_blocking=True from pageable memory 10.1 ms (the runtime stages the copy in pieces that each wait behind the
    queued kernels), a stream of its own 1.70 ms.  In the tracker itself the three forms are within the run-to-run spread or worse
    (side stream: 90-92 frames/s against 102-110; blocking against non_blocking: 105-111 both) - the tracked sequence is device-bound
    where the frame goes up - so the form that cannot hit the 10 ms case stays."""
    return image.to(device) if isinstance(image, torch.Tensor) else image


class MotionFilter:
    def __init__(self, net, video, thresh=2.5, device="cuda:0"):
        self.cnet, self.fnet, self.update = net.cnet, net.fnet, net.update
        self.video, self.thresh, self.device = video, thresh, torch.device(device)
        self.count = 0
        self.MEAN = torch.as_tensor([0.485, 0.456, 0.406], device=self.device)[:, None, None]
        self.STDV = torch.as_tensor([0.229, 0.224, 0.225], device=self.device)[:, None, None]
        self.net = self.inp = self.fmap = None
        self.fused_encoders = True         # BasicEncoder.forward_inference: bias / instance norm / ReLU / residual add as one kernel per layer
        self.keep_features = True          # every frame's feature map stays resident for the trajectory filler (DepthVideo.remember_features)
        self._coords0 = None
        self._coords0_by_shape = {}        # identity grids, one per map size, never released: a captured frame graph has its grid's address baked in
        self._features_g = GraphedCall(self._features_dev, name="fnet", guard=_weights_guard(self.fnet))
        self._context_g = GraphedCall(self._context_dev, name="cnet", guard=_weights_guard(self.cnet))
        # a whole tracked frame - encoder, 1-edge volume, lookup, one operator pass, the mean flow norm - as ONE captured launch:
        # issued eagerly the ~25 launches behind the encoder left the device idle for 0.4 of a frame's 1.4 ms (bench.py `sequence`)
        self._frame_g = GraphedCall(self._frame_dev, name="motion filter frame",
                                    guard=lambda: (_weights_guard(self.fnet)(), _operator_guard(self.update)()),
                                    frozen=lambda i: i >= 1)       # the reference keyframe's maps / static terms: written once in _new_reference
                                                                   # (fresh tensors per keyframe), read-only until the next one replaces them
        self._static = None                # conv(W[:, inp], inp) of the reference keyframe's context: constant until the next keyframe
        self._pending = None               # a frame between begin() and finish()
        self.before_context = None         # called when a frame is known to become a keyframe, before anything is queued for it
        self.overlap_upload = False        # asynchronous upload through pinned staging (Droid sets it in pipelined mode)
        self._up_stream, self._stage, self._stage_k, self._stage_used = None, [None, None], 0, None
        self._mag_host = self._mag_ready = None

    def _upload(self, image):
        """host frame -> device, as it is (the reference's stream hands over int32, test_vo.py:41): NO tensor operation on the host
        (on the 128-core hosts of the MI355X boxes every CPU tensor op on a frame - a dtype cast, the [2, 1, 0] channel gather of
        motion_filter.py:52, torch.stack in the filler - costs 2-20 ms: an OpenMP team is woken for 0.6 M elements); the copy itself: `upload_frame`."""
        if self.overlap_upload and self.device.type == "cuda" and isinstance(image, torch.Tensor) and not image.is_cuda:
            # pipelined tracker: the launch stream still holds the previous keyframe's graph updates (and the library's second stream their
            # side chains); a blocking copy from pageable memory waits for them - measured 2.8 ms per frame, on either stream - and this
            # frame's graph could only be launched into an idle device afterwards.  The frame is staged in pinned memory (one host
            # memmove, no tensor operation: see above) and goes up asynchronously on an upload stream; the launch stream waits on the device.
            import ctypes
            if self._up_stream is None:
                from .droid_backends import upload_stream
                self._up_stream = upload_stream(self.device)               # (one per process: see there)
            k = self._stage_k = self._stage_k ^ 1
            st = self._stage[k]
            if st is None or st[0].shape != image.shape or st[0].dtype != image.dtype:
                # two (pinned host buffer, device buffer) pairs, used alternately and kept: no allocation per frame, and no tensor that one
                # stream allocates and another reads (the caching allocator would have to track it per use)
                st = self._stage[k] = (torch.empty(image.shape, dtype=image.dtype).pin_memory(),
                                       torch.empty(image.shape, dtype=image.dtype, device=self.device), torch.cuda.Event(), torch.cuda.Event())
                st[3].record(torch.cuda.current_stream(self.device))
            else:
                st[2].synchronize()                                        # (the copy out of this host buffer two frames ago)
            if image.is_contiguous():
                ctypes.memmove(st[0].data_ptr(), image.data_ptr(), image.numel() * image.element_size())
            else:
                st[0].copy_(image)
            cur = torch.cuda.current_stream(self.device)
            self._up_stream.wait_event(st[3])                              # (the launch stream's last read of this device buffer, two frames ago)
            with torch.cuda.stream(self._up_stream):
                st[1].copy_(st[0], non_blocking=True)
                st[2].record(self._up_stream)
            cur.wait_stream(self._up_stream)
            self._stage_used = st
            return st[1]
        return upload_frame(image, self.device)

    _MEAN3, _STD3 = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)

    def _normalise_dev(self, image_dev):
        if self.device.type == "cuda" and self.fused_encoders and image_dev.dim() == 3 and image_dev.shape[0] == 3 and image_dev.is_contiguous() \
                and image_dev.dtype in (torch.int32, torch.uint8, torch.float32):
            # one launch (pvo_frame_normalise): the same fp32 operations in the same order, rounded to fp16 as the fused encoder's first
            # cast does - bit-identical to the six element-wise kernels below, which ran in front of EACH encoder
            from .droid_backends import frame_normalise
            return frame_normalise(image_dev, self._MEAN3, self._STD3, torch.float16)[None]
        # (flip(0) = the reference's channel gather [2, 1, 0] without an index tensor: a list index is uploaded from the host on
        # every call, which cannot be captured into a graph)
        x = image_dev.flip(0)[None, None].float() / 255.0
        return (x - self.MEAN) / self.STDV

    def _features_dev(self, image_dev):
        """frame on the device -> feature map [1,128,h,w]"""
        with self._autocast():
            return self._encode(self.fnet, self._normalise_dev(image_dev)).squeeze(0)

    def _context_dev(self, image_dev):
        with self._autocast():
            return self._context(self._normalise_dev(image_dev))

    def _frame_dev(self, image_dev, fmap_ref, net_ref, inp_ref, *static):
        """one tracked frame against the reference keyframe (motion_filter.py:52-72): its feature map and the mean norm of the
        one-step flow estimate [1]"""
        gmap = self._features_dev(image_dev)
        with self._autocast():
            half = lambda t: t if t.dtype in (torch.float16, torch.bfloat16) or self.device.type != "cuda" else t.half()
            corr = CorrBlock(half(fmap_ref[None]), half(gmap[None]))(self._coords0)
            kw = {"static_terms": tuple(static)} if static else {}
            if self.device.type == "cuda" and hasattr(self.update, "packed_weights"):
                # (this runs inside a captured graph: the operator's side chains stay on the launch stream - one edge leaves nothing to
                # overlap, and a captured fork is a parallel branch on a stream the runtime picks per graph instance: trackers of one
                # process then ran 100 or 138 frames/s depending on the hardware queue that stream got)
                kw["single_stream"] = True
            _, delta, _, _ = self.update(net_ref[None], inp_ref[None], corr, **kw)
            return gmap, delta[..., 0:2].float().norm(dim=-1).mean().reshape(1)

    def _new_reference(self, gmap, net, inp):
        """the frame just stored becomes the reference of the motion test (graph outputs are static buffers: copies)"""
        src = [net, inp, gmap]
        dst = [torch.empty_like(x) for x in src]
        if self.device.type == "cuda":
            torch._foreach_copy_(dst, src)           # (one launch: three device-to-device blits are ~48 us of host time each)
        else:
            for d, x in zip(dst, src):
                d.copy_(x)
        self.net, self.inp, self.fmap = dst
        self._static = None
        if self.device.type == "cuda" and hasattr(self.update, "static_terms") and getattr(self.update, "fused_gru", False) \
                and next(self.update.parameters()).dtype in (torch.float16, torch.bfloat16):
            self._static = tuple(self.update.static_terms(self.inp))      # (16-bit operator: the ConvGRU's static-input terms, once per keyframe)

    def _autocast(self):
        return torch.autocast("cuda", dtype=torch.float16, enabled=self.device.type == "cuda")

    def _encode(self, net, x):
        """an encoder on the GPU inference path: its fused per-layer form (BasicEncoder.forward_inference) where it has one"""
        fused = getattr(net, "forward_inference", None) if self.device.type == "cuda" and self.fused_encoders else None
        return fused(x) if fused is not None else net(x)

    def _context(self, x):
        net, inp = self._encode(self.cnet, x).split([128, 128], dim=2)
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
        self.begin(tstamp, image, depth, intrinsics, segments)
        return self.finish()

    @torch.no_grad()
    def begin(self, tstamp, image, depth=None, intrinsics=None, segments=None):
        """first half of track(): the frame goes up and its graph (encoder, 1-edge volume, lookup, one operator pass, mean flow norm) is
        LAUNCHED; the scalar is copied to a pinned host buffer behind an event.  Nothing here reads or writes the video."""
        ht, wd = image.shape[-2] // 8, image.shape[-1] // 8
        img = self._upload(image)
        self._pending = (tstamp, image, img, intrinsics, segments, None, None)
        if self.video.counter == 0:
            return
        if self._coords0 is None or self._coords0.shape[-3:-1] != (ht, wd):
            if (ht, wd) not in self._coords0_by_shape:
                with self._autocast():
                    self._coords0_by_shape[(ht, wd)] = coords_grid(ht, wd, device=self.device)[None, None]
            self._coords0 = self._coords0_by_shape[(ht, wd)]
        gmap, mag = self._frame_g(img, self.fmap, self.net, self.inp, *(self._static or ()))
        if mag.is_cuda:
            if self._mag_host is None:
                self._mag_host = torch.empty(1, dtype=mag.dtype).pin_memory()
                self._mag_ready = torch.cuda.Event()
            self._mag_host.copy_(mag.reshape(-1)[:1], non_blocking=True)
            self._mag_ready.record()
            mag = None
        self._pending = (tstamp, image, img, intrinsics, segments, gmap, mag)

    @torch.no_grad()
    def finish(self):
        """second half: read the motion test's scalar; a frame that moved enough gets its context features and joins the video"""
        if self._pending is None:
            raise RuntimeError("MotionFilter.finish() without a begin(): no frame is in flight")
        tstamp, image, img, intrinsics, segments, gmap, mag = self._pending
        self._pending = None
        try:
            return self._finish(tstamp, image, img, intrinsics, segments, gmap, mag)
        finally:
            self._release_stage()

    def _release_stage(self):
        """the launch stream has issued its last read of the frame's device buffer (asynchronous upload): the next upload into it waits for this"""
        if self._stage_used is not None:
            self._stage_used[3].record(torch.cuda.current_stream(self.device))
            self._stage_used = None

    def _finish(self, tstamp, image, img, intrinsics, segments, gmap, mag):
        if gmap is None:                                                       # the first frame: always a keyframe
            gmap = self._features_g(img)                                       # [1,128,h,w]
            self._remember(tstamp, gmap, image)
            ident = torch.as_tensor([0, 0, 0, 0, 0, 0, 1.0], device=self.device)
            net, inp = self._context_g(img)
            self._new_reference(gmap, net, inp)
            self._append(tstamp, image, ident, 1.0, self._small_to_device(intrinsics), gmap, net, inp, segments)
            return True
        self._remember(tstamp, gmap, image)
        if mag is None:
            self._mag_ready.synchronize()
            moved = float(self._mag_host[0])
        else:
            moved = mag.item()
        if moved > self.thresh:
            self.count = 0
            if self.before_context is not None:
                self.before_context()                                          # (Droid: DroidFrontend.keyframe_ahead)
            net, inp = self._context_g(img)
            self._new_reference(gmap, net, inp)
            self._append(tstamp, image, None, None, self._small_to_device(intrinsics), gmap, net, inp, segments)
            return True
        self.count += 1
        return False

    def _small_to_device(self, t):
        """a few host floats (the frame's intrinsics) -> device through the pinned staging ring: `.to(device)` from pageable memory is a
        BLOCKING copy queued behind everything on the stream - in the pipelined tracker 1 ms per frame spent waiting for the previous
        keyframe's updates"""
        if self.device.type == "cuda" and isinstance(t, torch.Tensor) and not t.is_cuda:
            from .droid_backends import to_device_async
            return to_device_async(t, t.dtype, self.device)
        return t.to(self.device)

    def _remember(self, tstamp, gmap, image):
        if self.keep_features and hasattr(self.video, "remember_features"):
            self.video.remember_features(tstamp, gmap, image)

    @torch.no_grad()
    def track_vo(self, tstamp, image, depth=None, intrinsics=None, segments=None):
        """every frame becomes a keyframe (motion_filter.py:89-109)"""
        ident = torch.as_tensor([0, 0, 0, 0, 0, 0, 1.0], device=self.device)
        img = self._upload(image)
        gmap = self._features_g(img)
        net, inp = self._context_g(img)
        first = self.video.counter == 0
        self._append(tstamp, image, ident if first else None, 1.0 if first else None, self._small_to_device(intrinsics),
                     gmap, net, inp, segments)
        self._release_stage()
        return True
