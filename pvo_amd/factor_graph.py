"""FactorGraph — edge bookkeeping and the per-iteration hot loop `update()`.

Counterpart of the reference's FactorGraph (VO_Module/droid_slam/factor_graph.py:12-307):
same attributes (ii, jj, age, target_cam, weight, raw_mask, delta_dy, full_flow, *_inac,
damping, corr, net, inp, segm) and methods (add_factors, rm_factors, rm_keyframe,
add_neighborhood_factors, update).  One `update()` = reproject -> correlation lookup ->
update operator -> dynamic-mask / weight glue -> dense BA, as factor_graph.py:227-307.

What is different, because it is what costs time once the kernels are fast:
  * edge lists are mirrored on the host (`_ii_h`, `_jj_h`, `_age_h`): duplicate filtering,
    the default t0/t1 and the age-based eviction never read device memory back
    (the reference does `.item()` per edge, factor_graph.py:69-74, :247, depth_video.py:204);
  * the panoptic segment vote (:256-276) runs on the device (two bincounts keyed by
    (edge, segment id)) instead of np.unique on the host;
  * reproject, the 4-level lookup and the whole BA are single calls into libpvo_hip.
"""
import contextlib

import itertools
import torch

from .modules.corr import CorrBlock, CorrVolumePool


def coords_grid(ht, wd, device):
    y, x = torch.meshgrid(torch.arange(ht, device=device).float(), torch.arange(wd, device=device).float(),
                          indexing="ij")
    return torch.stack([x, y], dim=-1)


class _EdgeRows:
    """One per-edge state tensor (net, target_cam, weight, raw_mask, delta_dy, segm) as the first E rows of a
    preallocated [capacity, ...] buffer.

    The reference re-indexes every state tensor when edges are dropped and torch.cat's onto it when edges are added
    (factor_graph.py:135-161,177-200): per keyframe that moved all of `net` (28 MB for 36 edges) three times.  Here
      * appending writes the new rows behind the live ones (nothing old is touched);
      * dropping a SUFFIX of the edges (the edges of the newest keyframe, the common case in the frontend) moves nothing;
      * any other drop is one row gather into the second buffer of the pair, which then becomes the live one.
    The graph's attributes stay plain tensors (views of the live buffer).  A caller may assign something else to them
    (the PyTorch formulation of update() does): a tensor that is not the live view is simply copied in at the next
    edge change."""

    def __init__(self, capacity):
        self.capacity = capacity
        self.buf = [None, None]
        self.cur = 0

    def _fits(self, rows):
        b = self.buf[self.cur]
        return b is not None and b.dtype == rows.dtype and b.device == rows.device and tuple(b.shape[1:]) == tuple(rows.shape[1:])

    def owns(self, rows):
        b = self.buf[self.cur]
        return self._fits(rows) and rows.data_ptr() == b.data_ptr() and rows.is_contiguous() and rows.shape[0] <= b.shape[0]

    def _buffer(self, k, like, need):
        b = self.buf[k]
        if b is None or b.dtype != like.dtype or b.device != like.device or tuple(b.shape[1:]) != tuple(like.shape[1:]) \
                or b.shape[0] < need:
            self.capacity = max(self.capacity, need + 16 if need > self.capacity else need)
            b = self.buf[k] = torch.empty((self.capacity,) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
        return b

    def keep(self, rows, keep_l, keep_t):
        """rows[keep_l] (keep_l ascending, keep_t() its device tensor, built only when a gather is needed)"""
        n = len(keep_l)
        if self.owns(rows) and (n == 0 or keep_l[-1] == n - 1):
            return self.buf[self.cur][:n]
        other = self.cur ^ 1
        b = self.buf[other]
        if b is not None and rows.numel() and rows.untyped_storage().data_ptr() == b.untyped_storage().data_ptr():
            rows = rows.clone()                                    # (a stale view of the spare buffer: do not gather onto itself)
        dst = self._buffer(other, rows, max(n, self.capacity))
        if n:
            torch.index_select(rows, 0, keep_t(), out=dst[:n])
        self.cur = other
        return dst[:n]

    def append(self, rows, n, fill):
        """rows followed by n new rows; fill(dst) writes them (dst = the [n, ...] tail of the live buffer)"""
        E = rows.shape[0]
        if not self.owns(rows) or E + n > self.buf[self.cur].shape[0]:
            other = self.cur ^ 1
            b = self.buf[other]
            if b is not None and rows.numel() and rows.untyped_storage().data_ptr() == b.untyped_storage().data_ptr():
                rows = rows.clone()
            dst = self._buffer(other, rows, max(E + n, self.capacity))
            if E:
                dst[:E].copy_(rows)
            self.cur = other
        b = self.buf[self.cur]
        fill(b[E:E + n])
        return b[:E + n]


_CACHE_SERIAL = itertools.count(1)


class FactorGraph:
    _index_dirty = False             # host edge lists changed since (ii, jj, age) were last uploaded

    def __init__(self, video, update_op, device="cuda:0", corr_impl="volume", max_factors=-1):
        self.video, self.update_op = video, update_op
        self.device = torch.device(device)
        self.max_factors, self.corr_impl = max_factors, corr_impl
        self.ht, self.wd = ht, wd = video.ht // 8, video.wd // 8
        self.dy_thresh, self.mask_num = 0.5, 2
        self.coords0 = coords_grid(ht, wd, self.device)
        lng = dict(dtype=torch.long, device=self.device)
        self._age_lag = 0
        self.ii, self.jj, self.age = (torch.zeros(0, **lng) for _ in range(3))
        self._ii_h, self._jj_h, self._age_h = [], [], []
        self._inp = None
        self._corr_ready = None     # event: the side-stream build of the most recently added edges (see add_factors)
        self.build_on_side_stream = True
        self.native_select = True    # proximity-edge selection in the library's host code (False: the numpy form below, which tests compare it with)
        self._rows = {}             # per-edge state tensors as rows of fixed-capacity buffers (_EdgeRows)
        self._segm = None
        self.corr = self.net = self.inp = self.segm = None
        self.damping = 1e-6 * torch.ones_like(video.disps)
        z = lambda c: torch.zeros(1, 0, ht, wd, c, device=self.device, dtype=torch.float)
        self.target_cam, self.weight, self.delta_dy, self.full_flow = z(2), z(2), z(2), z(2)
        self.raw_mask = z(self.mask_num)
        self.ii_inac, self.jj_inac = torch.zeros(0, **lng), torch.zeros(0, **lng)
        self._ii_inac_h, self._jj_inac_h = [], []
        self.ii_bad, self.jj_bad = torch.zeros(0, **lng), torch.zeros(0, **lng)
        self.target_cam_inac, self.weight_inac, self.delta_dy_inac, self.full_flow_inac = z(2), z(2), z(2), z(2)
        self.raw_mask_inac = z(self.mask_num)
        self.fused_glue = True      # run update() as ONE call into libpvo_hip when the operator and the pool support it
        self._cache = {}            # device index tensors derived from the host edge lists; cleared on any edge change
        self._version = 0           # bumped on every edge change
        self.P_zr = self.P_q = None  # per-edge static-input terms of the ConvGRU (computed once when an edge is added)
        self._static_by_slot = False  # ... stored [E, ...] in edge order, or by slot in the volume pool ([capacity, ...])
        self.want_upmask = True     # compute GraphAgg's upsampling mask although update() discards it, as the reference does
        try:
            self._autocast = next(update_op.parameters()).dtype == torch.float32
        except (StopIteration, AttributeError, TypeError):
            self._autocast = True

    # ------------------------------------------------------------------ edges
    def _cl5(self, t):
        """[1,E,C,H,W] stored channels-last ([E,H,W,C] in memory): the layout the NHWC convolutions and the fused
        element-wise kernels read, fixed when edges change instead of on every update"""
        if t is None or self.device.type != "cuda" or t.shape[1] == 0:
            return t
        return t[0].contiguous(memory_format=torch.channels_last)[None]

    def _cat_cl(self, old, new):
        """append edges to a channels-last [1,E,C,H,W] state tensor.  The concatenation runs on the physical [E,H,W,C]
        views (a plain row append); torch.cat on the strided 5-D tensors falls into a generic element-wise copy kernel
        that needs 49 us for 28 MB (four of them per keyframe in the profile)."""
        new = self._cl5(new)
        if old is None:
            return new
        if self.device.type == "cuda":
            a, b = old[0].permute(0, 2, 3, 1), new[0].permute(0, 2, 3, 1)
            if a.is_contiguous() and b.is_contiguous():
                return torch.cat([a, b], 0).permute(0, 3, 1, 2)[None]
        return self._cl5(torch.cat([old, new], 1))

    def _take_cl(self, t, idx):
        """t[:, idx] for a channels-last [1,E,C,H,W] state tensor, as a row gather on the physical layout"""
        if self.device.type == "cuda" and t.shape[1] > 0:
            a = t[0].permute(0, 2, 3, 1)
            if a.is_contiguous():
                return a.index_select(0, idx).permute(0, 3, 1, 2)[None]
        return self._cl5(t[:, idx])

    def _cached(self, key, make):
        v = self._cache.get(key)
        if v is None:
            v = self._cache[key] = make()
        return v

    def _filter_repeated(self, ii, jj):
        """drop requested edges that already exist as active or inactive edges (factor_graph.py:65-77).  As in the
        reference, a pair that appears twice inside one request is NOT de-duplicated."""
        have = set(zip(self._ii_h, self._jj_h)) | set(zip(self._ii_inac_h, self._jj_inac_h))
        keep = [k for k, e in enumerate(zip(ii, jj)) if e not in have]
        return [ii[k] for k in keep], [jj[k] for k in keep]

    @staticmethod
    def _to_list(x):
        return [int(v) for v in (x.tolist() if isinstance(x, torch.Tensor) else x)]

    def add_factors(self, ii, jj, remove=False):
        """add edges (factor_graph.py:106-161)"""
        ii_l, jj_l = self._filter_repeated(self._to_list(ii), self._to_list(jj))
        if not ii_l:
            return
        if self.max_factors > 0 and len(self._ii_h) + len(ii_l) > self.max_factors and self.corr is not None and remove:
            order = sorted(range(len(self._age_h)), key=lambda k: self._age_h[k])      # argsort(age), stable
            rank = [0] * len(order)
            for pos, k in enumerate(order):
                rank[k] = pos
            # the reference masks by POSITION in the age-sorted index list (factor_graph.py:128-129)
            ix = [order[p] for p in range(len(order))]
            mask_l = [ix[p] >= self.max_factors - len(ii_l) for p in range(len(order))]
            self.rm_factors(mask_l, store=True)
        self._cache.clear(); self._version += 1
        n, E0 = len(ii_l), len(self._ii_h)
        pool_path = self.corr_impl == "volume" and self.device.type == "cuda" and \
            self.video.fmaps.dtype in (torch.float16, torch.bfloat16)
        if pool_path and self.corr is None:        # resident slot pool: edge changes never move a volume
            base = self.max_factors if 0 < self.max_factors <= 4096 else 96          # (the backend's budget is 'unlimited')
            self.corr = CorrVolumePool(max(base + 32, n + 32), self.ht, self.wd, self.device, self.video.fmaps.dtype)
        if pool_path and isinstance(self.corr, CorrVolumePool):
            # the new edges' endpoints and the slots their volumes will own: one staged upload
            from .droid_backends import to_device_packed
            if n > len(self.corr.free):
                # reserve() is about to grow the pool: its copies run on THIS stream and drop the old tensors, which a pending
                # side-stream build (two add_factors calls without an update between them) may still be writing
                self._corr_sync()
            ii, jj, new_slots = to_device_packed([(ii_l, torch.long), (jj_l, torch.long), (self.corr.reserve(n), torch.int32)],
                                                 self.device)
        else:
            both = self._idx(ii_l + jj_l)
            ii, jj, new_slots = both[:n], both[n:], None
        by_slot = isinstance(self.corr, CorrVolumePool) and self._static_ok()
        # The new edges' volumes and static GRU terms are first read by the next update's lookup / gates: with everything
        # by slot (nothing of them is concatenated onto the live state) they are built on a second stream, beside the row
        # appends, the frame distances and the index uploads of this keyframe; update() waits for them (_corr_sync).
        side = self._build_stream() if by_slot and self.build_on_side_stream else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(self.device))
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            if self.corr_impl == "volume":
                if pool_path:
                    self.corr.add(self.video.fmaps[ii], self.video.fmaps[jj], new_slots)
                else:
                    corr = CorrBlock(self.video.fmaps[ii][None], self.video.fmaps[jj][None], channels_last=True)
                    self.corr = corr if self.corr is None else self.corr.cat(corr)
                inp = self._frame_rows(self.video.inps, ii)[None]
                if not by_slot:
                    self.inp = self._cat_cl(self.inp, inp)   # stored channels-last once, so no update re-lays it out
                if self._static_ok():
                    # conv(W[:, inp], inp) of the ConvGRU's gate / candidate convolutions, once per edge (inp never changes)
                    pz, pq = self.update_op.static_terms(self._cl5(inp)[0], self._op_dtype())
                    if by_slot:
                        # ... kept in the slot the edge's volume owns: like the volume, never moved when edges come and go
                        # (as [E, ...] tensors they were 85 of the 141 MB gathered and concatenated per keyframe; `inp`
                        # itself is video.inps[ii], materialised only if the PyTorch formulation of the operator asks)
                        self.P_zr, self.P_q = self.corr.put("P_zr", pz), self.corr.put("P_q", pq)
                        self._static_by_slot = True
                    else:
                        self.P_zr = self._cat_cl(self.P_zr, pz[None])
                        self.P_q = self._cat_cl(self.P_q, pq[None])
        if side is not None:
            for t in (ii, jj, new_slots):
                t.record_stream(side)
            self._corr_ready = torch.cuda.Event()
            self._corr_ready.record(side)
        self._ii_h += ii_l; self._jj_h += jj_l; self._age_h += [0] * n
        self._index_dirty = True
        # per-edge state: the new rows are written behind the live ones (see _EdgeRows)
        nets = self.video.nets
        cl = self._net_cl()
        if self.net is None:
            self.net = self._net_view(torch.empty((0,) + ((self.ht, self.wd, 128) if cl else (128, self.ht, self.wd)),
                                                  dtype=nets.dtype, device=self.device))
        if cl and nets.permute(0, 2, 3, 1).is_contiguous() and self.net.dtype == nets.dtype:
            fill_net = lambda dst: torch.index_select(nets.permute(0, 2, 3, 1), 0, ii, out=dst)      # no temporary
        elif cl:
            fill_net = lambda dst: dst.copy_(nets[ii].permute(0, 2, 3, 1))
        else:
            fill_net = lambda dst: dst.copy_(nets[ii])
        self.net = self._net_view(self._edge_rows("net").append(self._net_rows(), n, fill_net))

        def fill_target(dst):
            if hasattr(self.video, "reproject_into"):
                self.video.reproject_into(ii, jj, dst)               # (no temporary, no copy)
            else:
                target, _ = self.video.reproject(ii, jj)
                dst.copy_(target[0])
        zero = lambda dst: dst.zero_()
        self.target_cam = self._edge_rows("target_cam").append(self.target_cam[0], n, fill_target)[None]
        self.weight = self._edge_rows("weight").append(self.weight[0], n, zero)[None]
        self.raw_mask = self._edge_rows("raw_mask").append(self.raw_mask[0], n, zero)[None]
        self.delta_dy = self._edge_rows("delta_dy").append(self.delta_dy[0], n, zero)[None]
        if getattr(self.video, "segm_filter", True):     # (without the panoptic filter nothing reads them: `segm` gathers on demand)
            segms = self.video.segms
            if self._segm is None:
                self._segm = torch.empty((1, 0) + tuple(segms.shape[1:]), dtype=segms.dtype, device=self.device)
            same = self._segm.dtype == segms.dtype                 # (a caller may have assigned labels of another integer type)
            self._segm = self._edge_rows("segm").append(
                self._segm[0], n, (lambda dst: torch.index_select(segms, 0, ii, out=dst)) if same else (lambda dst: dst.copy_(segms[ii])))[None]
        else:
            self._segm = None

    def _build_stream(self):
        """the library's own second stream (one per device, the one the update's side chains use): a stream of our own
        per graph changed which hardware queue the later streams of the process landed on - S-A in bench.py then ran
        with its two update streams serialised (lookup 66 -> 139 us)"""
        from .droid_backends import side_stream
        return side_stream(self.device)

    def _corr_sync(self):
        """order the current stream behind the side-stream build of the newest edges (no host wait)"""
        if self._corr_ready is not None:
            torch.cuda.current_stream(self.device).wait_event(self._corr_ready)
            self._corr_ready = None

    # -- per-edge state rows (see _EdgeRows)
    def _edge_rows(self, name):
        r = self._rows.get(name)
        if r is None:
            base = self.max_factors if 0 < self.max_factors <= 4096 else 96
            r = self._rows[name] = _EdgeRows(base + 32)
        return r

    def _net_cl(self):
        """`net` is kept channels-last ([E,H,W,128] rows) on the device, as the NHWC kernels read it"""
        return self.device.type == "cuda"

    def _net_rows(self):
        """the physical rows of `net` in the layout its _EdgeRows buffer has (a copy only if a caller assigned another layout)"""
        t = self.net[0]
        if self._net_cl():
            t = t.permute(0, 2, 3, 1)
        return t if t.is_contiguous() or t.shape[0] == 0 else t.contiguous()

    def _net_view(self, rows):
        return (rows.permute(0, 3, 1, 2) if self._net_cl() else rows)[None]

    def _frame_rows(self, frames, idx):
        """frames[idx] for a per-frame [N,C,H,W] buffer; a buffer stored channels-last (DepthVideo's nets / inps) comes
        back channels-last without a layout pass"""
        phys = frames.permute(0, 2, 3, 1)
        if self.device.type == "cuda" and phys.is_contiguous():
            return phys.index_select(0, idx).permute(0, 3, 1, 2)
        return frames[idx]

    def _sync_edge_index(self):
        """device copies of (ii, jj, age) from the host mirrors: ONE staged upload instead of three gathers or three
        concatenations per edge-set change (the lists are the source of truth for every decision anyway).  Deferred
        until something reads them: a drop followed by an add uploads once."""
        E = len(self._ii_h)
        packed = self._idx(self._ii_h + self._jj_h + self._age_h) if E else torch.zeros(0, dtype=torch.long, device=self.device)
        self._ii_dev, self._jj_dev, self._age_dev = packed[:E], packed[E:2 * E], packed[2 * E:3 * E]
        self._age_lag = 0
        self._index_dirty = False

    @property
    def ii(self):
        if self._index_dirty:
            self._sync_edge_index()
        return self._ii_dev

    @ii.setter
    def ii(self, t):
        if self._index_dirty:
            self._sync_edge_index()
        self._ii_dev = t

    @property
    def jj(self):
        if self._index_dirty:
            self._sync_edge_index()
        return self._jj_dev

    @jj.setter
    def jj(self, t):
        if self._index_dirty:
            self._sync_edge_index()
        self._jj_dev = t

    def _idx(self, values):
        """host list -> device int64 tensor without draining the stream (persistent pinned staging ring, asynchronous
        copy); `torch.tensor(list, device=...)` is a synchronous copy, and boolean-mask indexing synchronises again to
        size its result - a dozen pipeline drains per keyframe in the reference's bookkeeping"""
        from .droid_backends import to_device_async
        return to_device_async(values, torch.long, self.device)

    def rm_factors(self, mask, store=False):
        """drop edges (factor_graph.py:163-200); mask: bool tensor or list over the active edges"""
        mask_l = [bool(v) for v in (mask.tolist() if isinstance(mask, torch.Tensor) else mask)]
        self._cache.clear(); self._version += 1
        keep_l = [k for k, m in enumerate(mask_l) if not m]
        made = []

        def keep_t():                                # the device copy of keep_l, uploaded only if some tensor has to be gathered
            if not made:
                made.append(self._idx(keep_l))
            return made[0]
        if store:
            rm = self._idx([k for k, m in enumerate(mask_l) if m])
            self.ii_inac = torch.cat([self.ii_inac, self.ii[rm]])
            self.jj_inac = torch.cat([self.jj_inac, self.jj[rm]])
            self._ii_inac_h += [i for i, m in zip(self._ii_h, mask_l) if m]
            self._jj_inac_h += [j for j, m in zip(self._jj_h, mask_l) if m]
            self.target_cam_inac = torch.cat([self.target_cam_inac, self.target_cam[:, rm]], 1)
            self.weight_inac = torch.cat([self.weight_inac, self.weight[:, rm]], 1)
            self.raw_mask_inac = torch.cat([self.raw_mask_inac, self.raw_mask[:, rm]], 1)
            self.delta_dy_inac = torch.cat([self.delta_dy_inac, self.delta_dy[:, rm]], 1)
        self._ii_h = [self._ii_h[k] for k in keep_l]
        self._jj_h = [self._jj_h[k] for k in keep_l]
        self._age_h = [self._age_h[k] for k in keep_l]
        self._index_dirty = True
        if self.corr_impl == "volume" and self.corr is not None:
            if isinstance(self.corr, CorrVolumePool):
                self.corr.keep([not m for m in mask_l])
            else:
                self.corr = self.corr[keep_t()]
        if self.net is not None:
            self.net = self._net_view(self._edge_rows("net").keep(self._net_rows(), keep_l, keep_t))
        if self._inp is not None:
            self._inp = self._take_cl(self._inp, keep_t())
        if self.P_zr is not None and not self._static_by_slot:
            self.P_zr, self.P_q = self._take_cl(self.P_zr, keep_t()), self._take_cl(self.P_q, keep_t())
        if self._segm is not None:
            self._segm = self._edge_rows("segm").keep(self._segm[0], keep_l, keep_t)[None]
        for name in ("target_cam", "weight", "raw_mask", "delta_dy"):
            t = getattr(self, name)
            rows = t[0] if t.shape[0] == 1 else None
            if rows is None or (rows.shape[0] and not rows.is_contiguous()):
                setattr(self, name, t[:, keep_t()])            # (not the layout this class keeps: plain indexing)
            else:
                setattr(self, name, self._edge_rows(name).keep(rows, keep_l, keep_t)[None])

    def clear_edges(self):
        self.rm_factors([True] * len(self._ii_h))
        self.net = self.inp = self.P_zr = self.P_q = None
        self._static_by_slot = False

    def rm_keyframe(self, ix):
        """drop keyframe ix and every edge touching it (factor_graph.py:202-225)"""
        self._corr_sync()                  # (the per-frame buffers below are read by a pending side-stream build)
        v = self.video
        for buf in (v.poses, v.disps, v.intrinsics, v.nets, v.inps, v.fmaps) + ((v.segms,) if v.segm_filter else ()):
            buf[ix] = buf[ix + 1].clone()
        m = [(i == ix) or (j == ix) for i, j in zip(self._ii_h, self._jj_h)]
        for t in (self.ii_inac, self.jj_inac):                       # (masked in-place updates would synchronise)
            t -= (t >= ix).long()
        self._index_dirty = True                                      # ii / jj follow the host lists below
        self._cache.clear(); self._version += 1
        dec = lambda l: [a - 1 if a >= ix else a for a in l]
        self._ii_h, self._jj_h = dec(self._ii_h), dec(self._jj_h)
        self._ii_inac_h, self._jj_inac_h = dec(self._ii_inac_h), dec(self._jj_inac_h)
        self.rm_factors(m, store=False)

    @torch.no_grad()
    def update_lowmem(self, t0=None, t1=None, itrs=2, use_inactive=False, EP=1e-7, steps=8, sharded=None):
        """global-BA update without correlation volumes (factor_graph.py:309-360): features are correlated on the fly
        by the alt-corr kernel, the update operator runs over source-frame chunks of 8, then ONE dense BA over all
        keyframes [1, t) with lm=1e-5, ep=1e-2.  As in the reference the motion features are built from
        `target_cam - coords0` (not the fresh reprojection) and the damping is the raw eta (no 0.2 factor).

        sharded (a `pvo_amd.parallel.ShardedBA`): this graph holds only the edges whose SOURCE frame this rank owns
        (`parallel.partition_by_source`).  Lookup, update operator, GraphAgg's per-source mean and the depth update are
        then rank-local; the BA's reduced pose system is all-reduced once per Gauss-Newton step, every rank takes the
        identical pose step, and depth maps of frames a rank does not own stay untouched on that rank
        (`ShardedBA.sync_disps` merges them when something needs all of them)."""
        self._corr_sync()
        from .modules.corr import AltCorrBlock
        t = self.video.counter
        ht, wd = self.ht, self.wd
        if self.corr_impl == "volume" and self._fused_ok() and self.P_zr is not None:
            # MI355X: 288 GB of HBM hold the correlation volumes of every edge of a global graph (25 MB each), which the
            # 11-24 GB GPUs the reference targets cannot - that is the only reason it switches to alt-corr and 8-frame
            # operator chunks here.  Chunking by source frame does not change a value (GraphAgg averages per source frame),
            # so with resident volumes this update is update() with its own damping rule and solver constants: the whole
            # graph in ONE native call per step.  (Correlation values then carry the volume's fp16 rounding, as in update().)
            for _ in range(steps):
                self._update_fused(1, t, itrs, False, EP, False, eta_scale=1.0, lm=1e-5, ep=1e-2, sharded=sharded, segm_vote=False)
                self.video.dirty[:t] = True
            return
        corr_op = AltCorrBlock(self.video.fmaps[None, :t], channels_last=True)
        jmax = max(self._jj_h + self._ii_h) if sharded is not None else max(self._jj_h)
        chunks = []
        for i in range(0, jmax + 1, 8):
            sel = [k for k, a in enumerate(self._ii_h) if i <= a < i + 8]
            if sel:
                chunks.append((torch.tensor(sel, device=self.device), sorted({self._ii_h[k] for k in sel})))
        src_l = sorted(set(self._ii_h))
        src = torch.tensor(src_l, device=self.device)
        if sharded is not None:
            # the BA optimises every depth map of the window; eta needs a row for each of them, in the order of
            # unique([1, t) U local sources); frames without a local edge get a neutral row (their update is 0)
            rows_l = sorted(set(range(1, t)) | set(src_l))
            row_of_src = torch.tensor([rows_l.index(f) for f in src_l], device=self.device)
        for _ in range(steps):
            coords1, _ = self.video.reproject(self.ii, self.jj)
            cam = self.target_cam - self.coords0
            motn = torch.cat([cam, cam + self.delta_dy, self.target_cam - coords1, self.raw_mask], dim=-1)
            motn = motn.permute(0, 1, 4, 2, 3).clamp(-64.0, 64.0)
            for v, srcs in chunks:
                iis, jjs = self.ii[v], self.jj[v]
                corr1 = corr_op(coords1[:, v], iis, jjs)
                with torch.autocast("cuda", dtype=torch.float16, enabled=self._autocast and self.device.type == "cuda"):
                    net, delta, weight, damping, _, delta_m = self.update_op(
                        self.net[:, v], self.video.inps[iis][None], corr1, motn[:, v], iis, jjs, False)
                self.net[:, v] = net.to(self.net.dtype)
                self.target_cam[:, v] = coords1[:, v] + delta[..., 0:2].float()
                self.damping[torch.tensor(srcs, device=self.device)] = damping[0].float()
                raw = self.raw_mask[:, v] + delta_m.float()
                self.raw_mask[:, v] = raw
                bin_mask = (torch.sigmoid(raw) >= self.dy_thresh).float()
                self.delta_dy[:, v] = delta[..., 2:4].float() * (1 - bin_mask)
                self.weight[:, v] = torch.sigmoid(weight.float() + (1 - bin_mask) * 10)
            eta = self.damping[src].contiguous() + EP
            target = self.target_cam.view(-1, ht, wd, 2).permute(0, 3, 1, 2).contiguous()
            weight = self.weight.view(-1, ht, wd, 2).permute(0, 3, 1, 2).contiguous()
            if sharded is None:
                self.video.ba(target, weight, eta, self.ii, self.jj, 1, t, itrs=itrs, lm=1e-5, ep=1e-2, motion_only=False)
            else:
                eta_rows = torch.ones(len(rows_l), ht, wd, dtype=eta.dtype, device=self.device)
                eta_rows[row_of_src] = eta
                sharded.ba(self.video.poses, self.video.disps, self.video.intrinsics[0], target, weight, eta_rows,
                           self.ii.contiguous(), self.jj.contiguous(), 1, t, itrs=itrs, lm=1e-5, ep=1e-2,
                           plan_key=(id(self), self._version, t))
                self.video.disps.clamp_(min=0.001)                   # DepthVideo.ba's clamp (depth_video.py:214)
            self.video.dirty[:t] = True

    def add_neighborhood_factors(self, t0, t1, r=3):
        """edges between frames within temporal radius r (factor_graph.py:362-370)"""
        ii, jj = [], []
        for i in range(t0, t1):
            for j in range(t0, t1):
                if i != j and abs(i - j) <= r:
                    ii.append(i); jj.append(j)
        self.add_factors(ii, jj)

    def _agg_segments_host(self):
        """CSR of the active edges grouped by source frame, groups in sorted(unique(ii)) order (what
        torch.unique(ii, return_inverse=True) yields on the device, droid_net.py:83) — built from the host mirror"""
        frames = sorted(set(self._ii_h))
        pos = {f: k for k, f in enumerate(frames)}
        buckets = [[] for _ in frames]
        for e, i in enumerate(self._ii_h):
            buckets[pos[i]].append(e)
        ptr, idx = [0], []
        for b in buckets:
            idx += b
            ptr.append(len(idx))
        return ptr, idx, len(frames)

    def _agg_segments(self):
        from .droid_backends import to_device_async
        ptr, idx, n = self._agg_segments_host()
        both = to_device_async(ptr + idx, torch.int32, self.device)
        return both[:len(ptr)], both[len(ptr):], n

    def prefetch_proximity(self, t0, t1, t, beta):
        """launch the distance matrix add_proximity_factors(t0, t1, beta=beta) will read once the video holds t frames, and its copy to a
        pinned host buffer.  The frontend calls this as soon as the motion filter knows the frame becomes a keyframe - BEFORE the context
        encoder and the keyframe's bookkeeping are queued: the distances only read poses, depths and intrinsics[0], which the previous
        update left final (the new frame's pose / depth seed included), so the same kernel on the same inputs gives the same bits - but
        the host's wait for them ends in front of ~0.6 ms of device work, which then runs while the host selects and adds the edges."""
        if self.device.type != "cuda" or t - t0 <= 0 or t - t1 <= 0:
            return
        import numpy as np
        nj = t - t1
        ii, jj = np.repeat(np.arange(t0, t), nj), np.tile(np.arange(t1, t), t - t0)
        d = self.video.distance(ii, jj, beta=beta).float().reshape(-1)
        st = self.__dict__.get("_prox_stage")
        if st is None or st[0].numel() < d.numel():
            st = self.__dict__["_prox_stage"] = (torch.empty(max(4096, d.numel()), dtype=torch.float32).pin_memory(), torch.cuda.Event())
        st[0][:d.numel()].copy_(d, non_blocking=True)
        st[1].record()
        self._prox_prefetch = {"key": (t0, t1, t, float(beta)), "host": st[0], "ready": st[1]}

    def add_proximity_factors(self, t0=0, t1=0, rad=2, nms=2, beta=0.25, thresh=16.0, remove=False):
        """add edges chosen by frame distance with non-maximum suppression (factor_graph.py:372-429).
        The distance matrix comes back from the device ONCE (the reference reads it element by element with
        .item() inside the loop); the greedy selection itself is the reference's, on host lists."""
        t = self.video.counter
        ix, jx = list(range(t0, t)), list(range(t1, t))
        if not ix or not jx:
            return
        import numpy as np
        ni, nj = len(ix), len(jx)
        ii, jj = np.repeat(np.arange(t0, t), nj), np.tile(np.arange(t1, t), ni)      # (arrays, not lists: a list of 14 000 ints costs 1 ms to turn into a tensor)
        # the selection below is the reference's greedy loop on a [ni, nj] array: the suppression of everything near an existing
        # edge is one array operation per window offset (it was a Python loop per edge and offset: 0.57 s of a 1.75 s sequence
        # with 119 keyframes, bench.py `sequence`), the greedy pass visits only the candidates under the threshold
        pf = self.__dict__.pop("_prox_prefetch", None)
        if pf is not None and pf["key"] == (t0, t1, t, float(beta)):
            pf["ready"].synchronize()                              # (launched by prefetch_proximity, in front of the keyframe's context encoder)
            D32 = pf["host"][:ni * nj].numpy().reshape(ni, nj)
        else:
            D32 = self.video.distance(ii, jj, beta=beta).float().cpu().numpy().reshape(ni, nj)
        if self.native_select:
            # the selection as one call into the library's host code (pvo_proximity_select): the device has nothing queued while it runs
            from . import droid_backends as db
            have_i = np.array(list(self._ii_h) + self.ii_bad.tolist() + list(self._ii_inac_h), dtype=np.int64)
            have_j = np.array(list(self._jj_h) + self.jj_bad.tolist() + list(self._jj_inac_h), dtype=np.int64)
            ei, ej = db.proximity_select(D32, t0, t1, rad, nms, thresh, have_i, have_j)
            if ei:
                self.add_factors(ei, ej, remove)
            return
        D = D32.astype(np.float64)
        I = np.arange(t0, t)[:, None]
        J = np.arange(t1, t)[None, :]
        D = np.where((I - rad < J) | ~(D <= 100), np.inf, D)         # (~(D <= 100): values above 100 and NaN drop out)
        offsets = [(di, dj) for di in range(-nms, nms + 1) for dj in range(-nms, nms + 1)]

        def suppress_many(hi, hj):
            """D = inf inside the diamond |di| + |dj| <= max(min(|i - j| - 2, nms), 0) around every (i, j) of (hi, hj)"""
            r = np.maximum(np.minimum(np.abs(hi - hj) - 2, nms), 0)
            for di, dj in offsets:
                m = (abs(di) + abs(dj) <= r)
                a, b = hi[m] + di - t0, hj[m] + dj - t1
                ok = (a >= 0) & (a < ni) & (b >= 0) & (b < nj)
                D[a[ok], b[ok]] = np.inf

        diamonds = [np.add.outer(np.abs(np.arange(-r, r + 1)), np.abs(np.arange(-r, r + 1))) <= r for r in range(nms + 1)]

        def suppress_one(i, j):
            """the same for ONE accepted edge: a masked assignment on the (2r + 1)^2 window, clipped to the array"""
            r = max(min(abs(i - j) - 2, nms), 0)
            a0, b0 = i - r - t0, j - r - t1
            a1, b1 = a0 + 2 * r + 1, b0 + 2 * r + 1
            ca0, cb0, ca1, cb1 = max(a0, 0), max(b0, 0), min(a1, ni), min(b1, nj)
            if ca0 < ca1 and cb0 < cb1:
                D[ca0:ca1, cb0:cb1][diamonds[r][ca0 - a0:ca1 - a0, cb0 - b0:cb1 - b0]] = np.inf

        have_i = np.array(list(self._ii_h) + self.ii_bad.tolist() + list(self._ii_inac_h), dtype=np.int64)
        have_j = np.array(list(self._jj_h) + self.jj_bad.tolist() + list(self._jj_inac_h), dtype=np.int64)
        far = np.abs(have_i - have_j) > 2
        if far.any():
            suppress_many(have_i[far], have_j[far])
        es = []
        for i in range(t0, t):
            for j in range(i + 1, min(i + rad + 1, t)):
                es += [(i, j), (j, i)]
        flat = D.ravel()                                              # (a view: suppress_many writes through)
        order = np.argsort(flat, kind="stable")
        order = order[:int(np.count_nonzero(flat <= thresh))]        # sorted: everything behind is above the threshold already
        for k in order.tolist():
            if not flat[k] <= thresh:                                 # suppressed since the sort
                continue
            i, j = int(ii[k]), int(jj[k])
            es += [(i, j), (j, i)]                                    # bidirectional
            suppress_one(i, j)
        if es:
            self.add_factors([e[0] for e in es], [e[1] for e in es], remove)

    # ------------------------------------------------------------------ hot loop
    def _segment_vote(self, bin_mask):
        """factor_graph.py:256-276 on the device: a segment whose dynamic-pixel fraction exceeds
        video.thresh is forced dynamic on that edge.  Segment id 0 is 'no segment' (label % 1e6 == 0)."""
        E = bin_mask.shape[1]
        S = self.video.max_segments
        seg = self.segm[0, :, 0].long().clamp_(0, S - 1)                       # [E,h,w]
        dyn = ((bin_mask[0, ..., 0] == 0) | (bin_mask[0, ..., 1] == 0))        # [E,h,w]
        key = (torch.arange(E, device=seg.device).view(E, 1, 1) * S + seg).reshape(-1)
        tot = torch.bincount(key, minlength=E * S).float()
        dcn = torch.bincount(key, weights=dyn.reshape(-1).float(), minlength=E * S)
        forced = (dcn / tot.clamp(min=1) > self.video.thresh) & (torch.arange(E * S, device=seg.device) % S != 0)
        keep = ~forced[key].view(1, E, *seg.shape[1:])
        return bin_mask & keep.unsqueeze(-1)

    def _op_dtype(self):
        """the 16-bit dtype the update operator runs in: the module's own (.half() / .bfloat16()) or fp16 autocast"""
        dt = next(self.update_op.parameters()).dtype
        return dt if dt in (torch.float16, torch.bfloat16) else torch.float16

    def _fused_ok(self):
        """update() as one native call: HIP device, resident tiled volume pool, an update operator with packed weights"""
        return self.fused_glue and self._static_ok()

    def _static_ok(self):
        if self.device.type != "cuda" or self.corr_impl != "volume":
            return False
        if self.corr is not None and not getattr(self.corr, "tiled", False):
            return False
        if self.corr is None and not (self.video.fmaps.dtype in (torch.float16, torch.bfloat16)):
            return False
        op = self.update_op
        if not hasattr(op, "packed_weights") or getattr(op, "training", True) or not getattr(op, "fused_gru", False) \
                or getattr(op, "use_aff_bri", False):
            return False
        if self.corr is None:
            from . import droid_backends as db
            return db.tiled_supported(self.ht, self.wd, self.video.fmaps.dtype)
        return True

    def _ba_plan(self, ii, jj, t0, t1, motion_only, n_in, R):
        """the BA's plan (unique depth frames, per-frame edge lists - the reference rebuilds these on the host in every
        iteration, droid_kernels.cu:1314-1322) depends only on the edge set and the window: built once per edge set"""
        from . import droid_backends as db
        v = self.video
        F, ht, wd = v.disps.shape
        P = t1 - t0
        key = (self._version, t0, t1, n_in, bool(motion_only), int(ii.shape[0]), int(R))
        st = self.__dict__.get("_ba_state")
        if st is None or st["key"] != key:
            need = db.ba_workspace_bytes(int(ii.shape[0]), P, F, ht * wd)
            ws = st["ws"] if st is not None and st["ws"].numel() >= need else \
                torch.empty(need + (need >> 2), dtype=torch.uint8, device=self.device)
            n6 = 6 * P
            sysb = st["sys"] if st is not None and st["sys"].numel() >= n6 * n6 + n6 else \
                torch.zeros(max(n6 * n6 + n6, 1), dtype=torch.int64, device=self.device)   # zero on entry, left zero by every solve
            db.ba_plan(ii, jj, F, ht * wd, -1 if motion_only else int(R), t0, t1, ws)
            st = self.__dict__["_ba_state"] = {"key": key, "ws": ws, "sys": sysb, "ii": ii, "jj": jj}
        return st

    @torch.no_grad()
    def _update_fused(self, t0, t1, itrs, use_inactive, EP, motion_only, eta_scale=0.2, lm=1e-4, ep=0.1, sharded=None,
                      segm_vote=True):
        """factor_graph.py:227-307 as ONE call into libpvo_hip (pvo_graph_update): reproject, motion features, lookup +
        update operator, (panoptic vote), mask / weight glue, damping, BA.  Everything that depends only on the edge set
        (index tensors, the BA plan, the inactive edges' BA rows, output buffers) is prepared once per edge set; per
        update the host fills one argument struct.  State tensors (net, target_cam, delta_dy, raw_mask, weight,
        full_flow) are updated IN PLACE.  segm_vote=False: no panoptic vote even when the video filters by segments
        (update_lowmem: the reference's global update never votes, factor_graph.py:309-360)."""
        from . import droid_backends as db
        from ._lib import GraphUpdateArgs
        from .droid_backends import to_device_packed
        v = self.video
        ht, wd = self.ht, self.wd
        E = len(self._ii_h)
        dt = self._op_dtype()
        if t0 is None:
            t0 = max(1, min(self._ii_h) + 1)
        if t1 is None:
            t1 = max(max(self._ii_h), max(self._jj_h)) + 1
        vote = bool(segm_vote and v.segm_filter)
        S = (v.segments_bound() if hasattr(v, "segments_bound") else v.max_segments) if vote else 0
        key = (self._version, t0, t1, bool(use_inactive), bool(motion_only), E, float(eta_scale), float(lm), float(ep), sharded is not None,
               vote, S)
        st = self._cache.get("fused")
        if st is None or st["key"] != key:
            src = sorted(set(self._ii_h))
            m_l = [(i >= t0 - 3) and (j >= t0 - 3) for i, j in zip(self._ii_inac_h, self._jj_inac_h)] if use_inactive else []
            n_in = sum(m_l)
            # one eta row per depth map the BA optimises, in the order of unique([t0, t1) U ii) (droid_kernels.cu:1314-1322);
            # frames without an active local edge keep their stored damping (pos = -1)
            rows = sorted(set(src) | {i for i, k in zip(self._ii_inac_h, m_l) if k} | set(range(t0, t1)))
            where = {f: k for k, f in enumerate(src)}
            seg_ptr, seg_idx, nseg = self._agg_segments_host()
            # every index table of this edge set in ONE staged upload (they were eight separate ~4 us copies)
            parts = [(rows, torch.long), ([where.get(f, -1) for f in rows], torch.int32), (seg_ptr, torch.int32),
                     (seg_idx, torch.int32), (self.corr.slots, torch.int32), ([k for k, f in enumerate(m_l) if f], torch.long)]
            if self._index_dirty:
                parts += [(self._ii_h, torch.long), (self._jj_h, torch.long), (self._age_h, torch.long)]
            up = to_device_packed(parts, self.device)
            frames_t, pos_t, seg, slots_t, m = up[0], up[1], (up[2], up[3], nseg), up[4], up[5]
            if self._index_dirty:
                self._ii_dev, self._jj_dev, self._age_dev = up[6], up[7], up[8]
                self._age_lag, self._index_dirty = 0, False
            self._cache["agg"] = seg
            self.corr._slots_t = slots_t
            target_ba = torch.empty(n_in + E, 2, ht, wd, device=self.device)
            weight_ba = torch.empty(n_in + E, 2, ht, wd, device=self.device)
            if n_in:
                # (integer indices from the host mirror: a boolean mask would synchronise to size its result)
                ii_ba, jj_ba = torch.cat([self.ii_inac[m], self.ii]), torch.cat([self.jj_inac[m], self.jj])
                target_ba[:n_in] = self.target_cam_inac[0, m].permute(0, 3, 1, 2)      # inactive edges do not change
                weight_ba[:n_in] = self.weight_inac[0, m].permute(0, 3, 1, 2)
            else:
                ii_ba, jj_ba = self.ii.contiguous(), self.jj.contiguous()
            if sharded is None:
                ba = self._ba_plan(ii_ba, jj_ba, t0, t1, motion_only, n_in, len(rows))
            else:                                                  # the sharded BA plans for itself (pvo_amd/parallel.py)
                ba = {"ii": ii_ba.contiguous(), "jj": jj_ba.contiguous(), "sys": None, "ws": None}
            st = self._cache["fused"] = dict(
                key=key, n_in=n_in, target_ba=target_ba, weight_ba=weight_ba, ii_ba=ba["ii"], jj_ba=ba["jj"], frames=frames_t,
                pos=pos_t, seg=seg, ba=ba, R=len(rows), S=S, ii=self.ii.contiguous(), jj=self.jj.contiguous(),
                slots=self.corr.slots_tensor(),
                # (labels of another integer type - add_factors keeps a caller's dtype - are converted: the kernel reads int32)
                segm=self.segm[0, :, 0].to(torch.int32).contiguous() if vote else None,
                full_flow=torch.empty(1, E, ht, wd, 2, device=self.device),
                eta=torch.empty(len(rows), ht, wd, device=self.device) if sharded is not None else None,
                ws=db.graph_update_workspace(E, seg[2], len(rows), ht, wd, S, self.device), args=GraphUpdateArgs())
            a = st["args"]
            db._fill_operator_args(a.op, E, ht, wd, self.corr.levels, st["slots"], self.corr.capacity, None, None, None, None, None,
                                   None, None, None, seg, None, (frames_t, pos_t, self.damping, EP, eta_scale), st["eta"], None)
            a.nframes = v.disps.shape[0]
            a.poses, a.disps, a.intrinsics = v.poses.data_ptr(), v.disps.data_ptr(), v.intrinsics.data_ptr()
            a.ii, a.jj = st["ii"].data_ptr(), st["jj"].data_ptr()
            a.segm = st["segm"].data_ptr() if st["segm"] is not None else None
            a.max_segments, a.vote_thresh, a.dy_thresh = S, float(v.thresh), float(self.dy_thresh)
            a.n_in, a.target_ba, a.weight_ba = n_in, target_ba.data_ptr(), weight_ba.data_ptr()
            a.ii_ba, a.jj_ba = st["ii_ba"].data_ptr(), st["jj_ba"].data_ptr()
            a.t0, a.t1, a.motion_only, a.lm, a.ep = t0, t1, 1 if motion_only else 0, float(lm), float(ep)
            if sharded is None:
                a.sys, a.ba_ws, a.ba_ws_bytes = ba["sys"].data_ptr(), ba["ws"].data_ptr(), ba["ws"].numel()
            else:                                                  # itrs = 0 below: the native call stops in front of the BA
                dummy = st["dummy"] = torch.zeros(64, dtype=torch.int64, device=self.device)
                a.sys, a.ba_ws, a.ba_ws_bytes = dummy.data_ptr(), dummy.data_ptr(), dummy.numel() * 8
            a.clamp_frames, a.disp_min = v.disps.shape[0], 0.001
            a.want_upmask = 1 if self.want_upmask else 0
        a = st["args"]
        # per update: the state tensors (a caller may have re-assigned them) and the scalar arguments
        for n in ("target_cam", "delta_dy", "raw_mask", "weight"):
            t = getattr(self, n)
            if not t.is_contiguous() or t.dtype != torch.float32:
                setattr(self, n, t.float().contiguous())
        net = self.net[0]
        if net.dtype != dt or not net.is_contiguous(memory_format=torch.channels_last):
            net = net.to(dt).contiguous(memory_format=torch.channels_last)
            self.net = net[None]
        a.op.net = a.op.net_out = net.data_ptr()
        a.op.P_zr, a.op.P_q = self.P_zr.data_ptr(), self.P_q.data_ptr()
        a.op.static_by_slot = 1 if self._static_by_slot else 0
        a.op.EP = float(EP)
        a.target, a.delta_dy, a.raw_mask = self.target_cam.data_ptr(), self.delta_dy.data_ptr(), self.raw_mask.data_ptr()
        if tuple(self.weight.shape) != (1, E, ht, wd, 2):           # (a caller's own weights of another shape are only an input)
            self.weight = torch.empty(1, E, ht, wd, 2, device=self.device)
        self.full_flow = st["full_flow"]
        a.weight, a.full_flow = self.weight.data_ptr(), self.full_flow.data_ptr()
        a.itrs = int(itrs) if sharded is None else 0
        a.clamp_frames = v.disps.shape[0] if sharded is None else 0
        # The ConvGRU's gate context depends on the hidden state only: the library computes the NEXT update's inside this
        # update's pose solves (context_ahead) and uses it if we can promise that nobody wrote `net` in between
        # (context_ready): same tensor object state (torch's version counter: the native update writes through the raw
        # pointer and does not move it) for the same edge-set cache entry.
        weights = self.update_op.packed_weights(dt)
        token = (net.data_ptr(), net._version, st.setdefault("serial", next(_CACHE_SERIAL)), weights.serial)
        a.context_ahead = 1
        a.context_ready = 1 if getattr(self, "_ctx_token", None) == token else 0
        db.graph_update(weights, a, st["ws"])
        self._ctx_token = token
        if sharded is not None:
            # edge sharding: assembly + Schur on this rank's edges, ONE integer all-reduce of the reduced pose system per
            # Gauss-Newton step, identical solve on every rank (pvo_amd/parallel.py)
            sharded.ba(v.poses, v.disps, v.intrinsics[0], st["target_ba"], st["weight_ba"], st["eta"], st["ii_ba"], st["jj_ba"],
                       t0, t1, itrs=itrs, lm=lm, ep=ep, motion_only=motion_only, plan_key=(id(self), self._version, t0, t1))
            v.disps.clamp_(min=0.001)
        self._age_lag += 1                 # (the device copy of `age` is brought up to date when it is next read)
        self._age_h = [x + 1 for x in self._age_h]

    @property
    def inp(self):
        """per-edge context features (factor_graph.py:33).  With the static terms kept by slot nothing on the native path
        reads them after an edge is created: they are video.inps[ii], gathered on demand for the PyTorch formulation."""
        if self._inp is None and self._static_by_slot and self._ii_h:
            return self._cached("inp", lambda: self._cl5(self.video.inps[self.ii][None]))
        return self._inp

    @inp.setter
    def inp(self, t):
        self._inp = t

    @property
    def segm(self):
        """per-edge panoptic labels of the source frames (factor_graph.py:34): kept as rows while the video filters by
        segments; otherwise video.segms[ii], gathered only if somebody asks"""
        if self._segm is None and self._ii_h and not getattr(self.video, "segm_filter", True):
            return self._cached("segm", lambda: self.video.segms[self.ii][None])
        return self._segm

    @segm.setter
    def segm(self, t):
        self._segm = t

    @property
    def age(self):
        """per-edge update count (factor_graph.py:35); decisions use the host mirror `_age_h`, so the native update
        path only counts and the device tensor catches up here, on access"""
        if self._index_dirty:
            self._sync_edge_index()
        if self._age_lag:
            self._age_dev += self._age_lag
            self._age_lag = 0
        return self._age_dev

    @age.setter
    def age(self, t):
        if self._index_dirty:
            self._sync_edge_index()
        self._age_dev, self._age_lag = t, 0

    @torch.no_grad()
    def update(self, t0=None, t1=None, itrs=2, use_inactive=False, EP=1e-7, motion_only=False):
        """one update of the factor graph (factor_graph.py:227-307)"""
        self._corr_sync()
        if self._fused_ok() and self.P_zr is not None:
            return self._update_fused(t0, t1, itrs, use_inactive, EP, motion_only)
        ht, wd = self.ht, self.wd
        coords1, _ = self.video.reproject(self.ii, self.jj)
        motn = torch.cat([self.target_cam - self.coords0, self.target_cam - self.coords0 + self.delta_dy,
                          self.target_cam - coords1, self.raw_mask], dim=-1)
        motn = motn.permute(0, 1, 4, 2, 3).clamp(-64.0, 64.0)
        if self.corr is None:            # corr_impl != "volume": the update operator gets no correlation features
            corr = None
        elif getattr(self.corr, "supports_channels_last", False):
            corr = self.corr(coords1, channels_last=True)
        else:
            corr = self.corr(coords1)
        with torch.autocast("cuda", dtype=torch.float16, enabled=self._autocast and self.device.type == "cuda"):
            kw = {}
            if getattr(self.update_op, "agg", None) is not None and self.device.type == "cuda":
                kw["agg_segments"] = self._cached("agg", self._agg_segments)
            self.net, delta, weight, damping, upmask, delta_m = \
                self.update_op(self.net, self.inp, corr, motn, self.ii, self.jj, False, **kw)
        if t0 is None:
            t0 = max(1, min(self._ii_h) + 1)
        if t1 is None:
            t1 = max(max(self._ii_h), max(self._jj_h)) + 1
        self.target_cam = coords1 + delta[..., 0:2].float()
        self.raw_mask = self.raw_mask + delta_m.float()
        bin_mask = torch.sigmoid(self.raw_mask) >= self.dy_thresh
        if self.video.segm_filter:
            bin_mask = self._segment_vote(bin_mask)
        bin_mask = bin_mask.float()
        self.delta_dy = delta[..., 2:4].float() * (1 - bin_mask)
        self.weight = torch.sigmoid(weight.float() + (1 - bin_mask) * 10)
        src = sorted(set(self._ii_h))                                  # torch.unique(self.ii), host side
        src_t = self._cached("src", lambda: torch.tensor(src, device=self.device))
        self.damping[src_t] = damping[0].float()
        m_l = [(i >= t0 - 3) and (j >= t0 - 3) for i, j in zip(self._ii_inac_h, self._jj_inac_h)] if use_inactive else []
        if any(m_l):
            m = self._cached(("inac", t0), lambda: torch.tensor(m_l, dtype=torch.bool, device=self.device))
            ii, jj = torch.cat([self.ii_inac[m], self.ii]), torch.cat([self.jj_inac[m], self.jj])
            target_cam = torch.cat([self.target_cam_inac[:, m], self.target_cam], 1)
            weight = torch.cat([self.weight_inac[:, m], self.weight], 1)
            src2 = sorted(set(src) | {i for i, k in zip(self._ii_inac_h, m_l) if k})
            src_t = self._cached(("src2", t0), lambda: torch.tensor(src2, device=self.device))
        else:
            ii, jj, target_cam, weight = self.ii, self.jj, self.target_cam, self.weight
        eta = 0.2 * self.damping[src_t] + EP
        target_cam = target_cam.view(-1, ht, wd, 2).permute(0, 3, 1, 2).contiguous()
        weight = weight.view(-1, ht, wd, 2).permute(0, 3, 1, 2).contiguous()
        self.video.ba(target_cam, weight, eta, ii, jj, t0, t1, itrs=itrs, lm=1e-4, ep=0.1, motion_only=motion_only)
        self.full_flow = coords1 + self.delta_dy - self.coords0
        self.age += 1
        self._age_h = [a + 1 for a in self._age_h]
