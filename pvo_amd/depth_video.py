"""DepthVideo — keyframe state buffers and the native calls made on them.

Counterpart of the reference's DepthVideo (VO_Module/droid_slam/depth_video.py:13-214):
same attribute names and shapes (poses [buf,7], disps [buf,H/8,W/8], intrinsics [buf,4],
fmaps/nets/inps fp16, segms int32), same `reproject` / `distance` / `ba` methods.
Differences: one process per GPU, so the counter is a plain int and there is no lock;
feature maps are kept channels-last ([buf,H/8,W/8,128]) because that is the layout both
the matrix-core volume build and AltCorr read; `reproject` is one HIP kernel instead of the
lietorch broadcast chain.
"""
import torch

from . import droid_backends as db


class DepthVideo:
    def __init__(self, image_size=(480, 640), buffer=1024, device="cuda:0", segm_filter=False, thresh=0.8,
                 store_images=False):
        self.counter = 0
        self.ht, self.wd = ht, wd = int(image_size[0]), int(image_size[1])
        self.device = torch.device(device)
        h8, w8 = ht // 8, wd // 8
        kw = dict(device=self.device)
        self.tstamp = torch.zeros(buffer, dtype=torch.float, **kw)
        self.images = torch.zeros(buffer, 3, ht, wd, dtype=torch.uint8, **kw) if store_images else None
        self.dirty = torch.zeros(buffer, dtype=torch.bool, **kw)
        self.poses = torch.zeros(buffer, 7, dtype=torch.float, **kw)
        self.poses[:, 6] = 1.0                                   # identity (depth_video.py:49-50)
        self.disps = torch.ones(buffer, h8, w8, dtype=torch.float, **kw)
        self.disps_up = None
        self.intrinsics = torch.zeros(buffer, 4, dtype=torch.float, **kw)
        self.fmaps = torch.zeros(buffer, h8, w8, 128, dtype=torch.half, **kw)      # channels-last
        # [buffer,128,h,w] as the reference has them, stored channels-last: an edge's rows are gathered straight into the
        # layout the update operator reads (factor_graph.py add_factors)
        self.nets = torch.zeros(buffer, h8, w8, 128, dtype=torch.half, **kw).permute(0, 3, 1, 2)
        self.inps = torch.zeros(buffer, h8, w8, 128, dtype=torch.half, **kw).permute(0, 3, 1, 2)
        self.segms = torch.zeros(buffer, 1, h8, w8, dtype=torch.int, **kw)
        self.full_flow = torch.ones(buffer, h8, w8, 2, dtype=torch.float, **kw)
        self.segm_filter, self.thresh = segm_filter, thresh
        self.max_segments = 1024
        self._segments_seen = 1            # the largest number of dense labels any stored frame has had (segments_bound)
        # feature maps of EVERY tracked frame by time stamp (keyframe or not): the motion filter computes them anyway, the
        # trajectory filler needs them again at the end (trajectory_filler.py:32-38 re-encodes every image).  0.78 MB per
        # 240 x 808 frame: a 10 000-frame sequence is 7.8 GB of the 288 GB - kept resident instead of recomputed, up to this budget
        # Every entry carries a fingerprint of the frame it was computed from (`frame_fingerprint`), which the filler checks: a
        # terminate() stream that reuses time stamps with other images (another stride, resize, sequence) is re-encoded, as the
        # reference always does.  The budget is a tenth of the device memory (at most 32 GiB); entries are released as the filler
        # consumes them and the rest when it is done (`forget_features`).
        self.frame_fmaps = {}
        self.frame_fmaps_budget = 32 << 30
        if torch.device(device).type == "cuda" and torch.cuda.is_available():
            self.frame_fmaps_budget = min(self.frame_fmaps_budget, torch.cuda.mem_get_info(torch.device(device))[1] // 10)
        self._frame_fmaps_bytes = 0

    # ------------------------------------------------------------------ bookkeeping
    def _fmap_cl(self, f, channels_last):
        """feature map -> the stored channels-last layout.  The layout is taken from the shape wherever that is
        unambiguous ([..,128,h,w] vs [..,h,w,128] with h,w of THIS video); for maps that are 128 wide or high pass
        channels_last explicitly (MotionFilter / the trajectory filler produce the reference's [128,h,w])."""
        h8, w8 = self.ht // 8, self.wd // 8
        if channels_last is None:
            is_cl = tuple(f.shape[-3:]) == (h8, w8, 128)
            is_cf = tuple(f.shape[-3:]) == (128, h8, w8)
            if is_cl == is_cf:
                if is_cl:
                    raise ValueError("feature map layout is ambiguous for a %dx%d map: pass channels_last=True/False" % (h8, w8))
                raise ValueError("feature map of shape %s fits neither [128,%d,%d] nor [%d,%d,128]" % (tuple(f.shape), h8, w8, h8, w8))
            channels_last = is_cl
        return f if channels_last else f.movedim(-3, -1)

    def _dense_segments(self, segm):
        """panoptic ids -> dense per-frame labels in [0, max_segments) with 0 kept as 'no segment'.  The reference keys the
        vote by lay * 1e6 + id (factor_graph.py:259), i.e. by the raw id; raw ids (R + 256 G + 65536 B, category * 1000 +
        instance, ...) do not fit a histogram, and only the grouping inside one frame matters to the vote."""
        if isinstance(segm, torch.Tensor) and not segm.is_cuda and self.device.type == "cuda" and segm.numel() <= (1 << 16):
            # a frame's ids arrive on the host (test_vo.py hands over numpy / CPU tensors) and are a few thousand integers: relabelled
            # THERE and sent up through the pinned staging ring.  On the device torch.unique is ~15 launches and reads its result's size
            # back - with the blocking upload in front of it 0.9 ms per keyframe of waiting for whatever the stream still held (1.3 ms
            # in the pipelined tracker: the previous keyframe's graph updates).
            import numpy as np
            seg_h = segm.numpy().astype(np.int64, copy=False)
            u_h, inv_h = np.unique(seg_h, return_inverse=True)
            shift = 1 if u_h.size and int(u_h[0]) != 0 else 0
            n = int(u_h.size) + shift
            if n > self.max_segments:
                raise ValueError("frame has %d panoptic segments, more than max_segments = %d" % (n, self.max_segments))
            self._segments_seen = max(self._segments_seen, n)
            lab = torch.from_numpy((inv_h.reshape(seg_h.shape) + shift).astype(np.int32))
            return db.to_device_async(lab, torch.int32, self.device)
        seg = torch.as_tensor(segm, device=self.device).to(torch.int64)
        u, inv = torch.unique(seg, return_inverse=True)
        if u.numel() and int(u[0]) != 0:
            inv = inv + 1
        n = int(u.numel()) + (1 if u.numel() and int(u[0]) != 0 else 0)
        if n > self.max_segments:
            raise ValueError("frame has %d panoptic segments, more than max_segments = %d" % (n, self.max_segments))
        self._segments_seen = max(self._segments_seen, n)
        return inv.to(torch.int32).reshape(seg.shape)

    @staticmethod
    def frame_fingerprint(image):
        """what identifies a frame for the feature cache: its shape, dtype and the sum of a ~100-pixel lattice of its values, taken
        where the frame lies (a few hundred elements: no OpenMP team is woken on the host, no kernel worth naming on the device)"""
        if not isinstance(image, torch.Tensor):
            return None
        h, w = image.shape[-2], image.shape[-1]
        sample = image[..., h // 11::max(1, h // 7), w // 13::max(1, w // 11)]
        return (tuple(image.shape), str(image.dtype), float(sample.double().sum()))

    def remember_features(self, tstamp, fmap, image=None):
        """keep a tracked frame's feature map [..,128,h,w] (a copy) for the trajectory filler; `image`: the frame it came from"""
        n = fmap.numel() * fmap.element_size()
        if self.frame_fmaps_budget <= 0 or self._frame_fmaps_bytes + n > self.frame_fmaps_budget:
            return
        old = self.frame_fmaps.pop(float(tstamp), None)
        if old is not None:
            self._frame_fmaps_bytes -= old[0].numel() * old[0].element_size()
        self.frame_fmaps[float(tstamp)] = (fmap.detach().clone(), self.frame_fingerprint(image) if image is not None else None)
        self._frame_fmaps_bytes += n

    def recall_features(self, tstamp, image=None):
        """the kept feature map of the frame tracked under `tstamp` if `image` is that frame (same fingerprint), else None.  The
        entry is released either way: a frame's pose is filled once."""
        hit = self.frame_fmaps.pop(float(tstamp), None)
        if hit is None:
            return None
        fmap, fp = hit
        self._frame_fmaps_bytes -= fmap.numel() * fmap.element_size()
        if fp is None or image is None or fp != self.frame_fingerprint(image):
            return None
        return fmap

    def forget_features(self):
        self.frame_fmaps.clear()
        self._frame_fmaps_bytes = 0

    def segments_bound(self):
        """the histogram width the panoptic vote needs: a power of two above every dense label stored so far (at least 16, at most
        max_segments) - a frame has tens of segments, and a 1024-wide table per edge is filled and cleared in every update"""
        b = 16
        while b < self._segments_seen:
            b *= 2
        return max(1, min(self.max_segments, b))

    def append(self, tstamp, pose, disp, intrinsics, fmap, net, inp, segm=None, image=None, channels_last=None):
        """store one keyframe; fmap may be [128,h,w] (reference layout) or [h,w,128] (see _fmap_cl)"""
        k = self.counter
        # (a Python number goes in with fill_ on a slice - a kernel argument.  `buf[k] = number` builds a host tensor and copies it with a
        # BLOCKING transfer queued behind everything on the stream: measured 1.3 ms per keyframe in the pipelined tracker)
        if isinstance(tstamp, torch.Tensor):
            self.tstamp[k] = tstamp
        else:
            self.tstamp[k:k + 1].fill_(float(tstamp))
        if pose is not None:
            self.poses[k] = pose
        if disp is not None:
            if isinstance(disp, torch.Tensor):
                self.disps[k] = disp
            else:
                self.disps[k:k + 1].fill_(float(disp))
        self.intrinsics[k] = intrinsics
        self.fmaps[k] = self._fmap_cl(fmap, channels_last)
        self.nets[k] = net
        self.inps[k] = inp
        if segm is not None:
            self.segms[k] = self._dense_segments(segm).reshape(self.segms[k].shape)
        if image is not None and self.images is not None:
            self.images[k] = image
        self.counter = k + 1

    def __setitem__(self, index, item):
        """video[index] = (tstamp, image, pose, disp, intrinsics[, fmap[, net[, inp[, segm]]]]) with None = keep
        (depth_video.py:64-101).  The counter grows to cover an int index; fmap may be NCHW or channels-last."""
        if isinstance(index, int) and index >= self.counter:
            self.counter = index + 1
        self.tstamp[index] = torch.as_tensor(item[0], dtype=torch.float, device=self.device)
        if item[1] is not None and self.images is not None:
            self.images[index] = item[1].to(self.images.dtype)
        for buf, val in ((self.poses, item[2]), (self.disps, item[3]), (self.intrinsics, item[4])):
            if val is not None:
                buf[index] = val
        if len(item) > 5 and item[5] is not None:
            self.fmaps[index] = self._fmap_cl(item[5], None)
        if len(item) > 6:
            self.nets[index] = item[6]
        if len(item) > 7:
            self.inps[index] = item[7]
        if self.segm_filter and len(item) > 8 and item[8] is not None:
            seg = torch.as_tensor(item[8], device=self.device)
            if isinstance(index, int) or seg.dim() <= 3:
                self.segms[index] = self._dense_segments(seg).reshape(self.segms[index].shape)
            else:                                           # several frames at once: labels are per frame
                self.segms[index] = torch.stack([self._dense_segments(x) for x in seg]).reshape(self.segms[index].shape)

    def __getitem__(self, index):
        """(pose, disp, intrinsics, fmap, net, inp) of a keyframe; negative ints count from the end (:103-121)"""
        if isinstance(index, int) and index < 0:
            index = self.counter + index
        return (self.poses[index], self.disps[index], self.intrinsics[index], self.fmaps[index], self.nets[index],
                self.inps[index])

    def normalize(self):
        """rescale so the mean inverse depth of the stored keyframes is 1 (depth_video.py:145-152)"""
        n = self.counter
        s = self.disps[:n].mean()
        self.disps[:n] /= s
        self.poses[:n, :3] *= s
        self.dirty[:n] = True

    def upsample(self, ix, mask):
        """convex 8x upsampling of the inverse depth of keyframes ix (depth_video.py:139-143)"""
        from .droid_net import cvx_upsample
        if self.disps_up is None:
            self.disps_up = torch.zeros(self.disps.shape[0], self.ht, self.wd, dtype=torch.float, device=self.device)
        self.disps_up[ix] = cvx_upsample(self.disps[ix].unsqueeze(-1), mask).squeeze(-1)

    @staticmethod
    def format_indicies(ii, jj, device):
        if not isinstance(ii, torch.Tensor):
            ii = torch.as_tensor(ii)
        if not isinstance(jj, torch.Tensor):
            jj = torch.as_tensor(jj)
        return (ii.to(device=device, dtype=torch.long).reshape(-1).contiguous(),
                jj.to(device=device, dtype=torch.long).reshape(-1).contiguous())

    # ------------------------------------------------------------------ native calls
    def reproject(self, ii, jj):
        """project points ii -> jj (depth_video.py:154-163): coords [1,E,h,w,2], valid [1,E,h,w,1]"""
        ii, jj = self.format_indicies(ii, jj, self.device)
        coords, valid = db.reproject(self.poses, self.disps, self.intrinsics, ii, jj)
        return coords[None], valid[None]

    def reproject_into(self, ii, jj, coords_out):
        """reproject(ii, jj)[0][0] written into coords_out [E,h,w,2] (device tensors only)"""
        ii, jj = self.format_indicies(ii, jj, self.device)
        db.reproject(self.poses, self.disps, self.intrinsics, ii, jj, out=coords_out)

    def distance(self, ii=None, jj=None, beta=0.3, bidirectional=True):
        """frame distance metric (depth_video.py:165-195)"""
        return_matrix = ii is None
        if return_matrix:
            N = self.counter
            grid = self.__dict__.get("_grid")
            if grid is None or grid[0] != N:                       # (the N x N index grid: uploaded once per window size)
                gi, gj = torch.meshgrid(torch.arange(N), torch.arange(N), indexing="ij")
                both = db.to_device_async(torch.cat([gi.reshape(-1), gj.reshape(-1)]), torch.long, self.device)
                grid = self.__dict__["_grid"] = (N, both[:N * N], both[N * N:])
            ii, jj = grid[1], grid[2]
        if not (isinstance(ii, torch.Tensor) and ii.is_cuda):      # host index lists: staged through the pinned ring, no stream drain
            ii_h = torch.as_tensor(ii, dtype=torch.long).reshape(-1)
            jj_h = torch.as_tensor(jj, dtype=torch.long).reshape(-1)
            both = db.to_device_async(torch.cat([ii_h, jj_h]), torch.long, self.device)
            ii, jj = both[:ii_h.numel()], both[ii_h.numel():]
        ii, jj = self.format_indicies(ii, jj, self.device)
        if bidirectional and self.poses.is_cuda:
            # both directions and their mean in one launch, bit-identical to the two calls below (which remain the host /
            # test formulation: the reference's call sequence is pinned on them)
            d = db.frame_distance_bidirectional(self.poses, self.disps, self.intrinsics[0], ii, jj, beta)
        elif bidirectional:
            poses = self.poses[:self.counter]              # (the reference clones them, depth_video.py:183; the kernel only reads)
            d1 = db.frame_distance(poses, self.disps, self.intrinsics[0], ii, jj, beta)
            d2 = db.frame_distance(poses, self.disps, self.intrinsics[0], jj, ii, beta)
            d = 0.5 * (d1 + d2)
        else:
            d = db.frame_distance(self.poses, self.disps, self.intrinsics[0], ii, jj, beta)
        return d.reshape(N, N) if return_matrix else d

    def ba(self, target, weight, eta, ii, jj, t0=1, t1=None, itrs=2, lm=1e-4, ep=0.1, motion_only=False,
           t1_hint=None):
        """dense bundle adjustment (depth_video.py:197-214); in place on poses / disps"""
        if t1 is None:
            t1 = t1_hint if t1_hint is not None else int(max(ii.max().item(), jj.max().item())) + 1
        if eta is None and not motion_only:
            k = torch.unique(torch.cat([ii, jj], 0)).shape[0]
            eta = 1e-7 * torch.ones([k, self.ht // 8, self.wd // 8], device=self.device)
        db.ba(self.poses, self.disps, self.intrinsics[0], target, weight, eta, ii, jj, t0, t1, itrs, lm, ep,
              motion_only)
        self.disps.clamp_(min=0.001)
