"""CorrBlock — all-pairs correlation volume pyramid + radius-r lookup on the HIP kernels.

API of the reference's CorrBlock / CorrSampler (VO_Module/droid_slam/modules/corr.py:6-71):
`CorrBlock(fmap1, fmap2, num_levels, radius)(coords)`, `.cat`, `[index]`.  The volume and all
pyramid levels come from ONE launch of pvo_corr_build, the lookup of all levels from ONE launch
of pvo_corr_pyramid_lookup (the reference: matmul + 3 pools, then 4 lookups + torch.cat).
When gradients are required (training) the reference's differentiable formulation is used:
torch.matmul/avg_pool2d for the volume and CorrSampler (HIP forward + HIP backward) per level.
"""
import torch
import torch.nn.functional as F

from .. import droid_backends as db


class CorrSampler(torch.autograd.Function):
    @staticmethod
    def forward(ctx, volume, coords, radius):
        ctx.save_for_backward(volume, coords)
        ctx.radius = radius
        corr, = db.corr_index_forward(volume, coords, radius)
        return corr

    @staticmethod
    def backward(ctx, grad_output):
        volume, coords = ctx.saved_tensors
        grad_volume, = db.corr_index_backward(volume, coords, grad_output.contiguous(), ctx.radius)
        return grad_volume, None, None


class CorrBlock:
    supports_channels_last = True      # __call__(coords, channels_last=True) returns NHWC-stored features

    def __init__(self, fmap1, fmap2, num_levels=4, radius=3, channels_last=False):
        """fmap1, fmap2: [B,N,C,H,W] (reference layout) or [B,N,H,W,C] with channels_last=True."""
        self.num_levels, self.radius = num_levels, radius
        needs_grad = torch.is_grad_enabled() and (fmap1.requires_grad or fmap2.requires_grad)
        if needs_grad:
            if channels_last:
                fmap1, fmap2 = fmap1.permute(0, 1, 4, 2, 3), fmap2.permute(0, 1, 4, 2, 3)
            self.corr_pyramid = self._build_differentiable(fmap1, fmap2, num_levels)
            return
        if channels_last:
            B, N, H, W, C = fmap1.shape
            f1, f2 = fmap1.reshape(B * N, H, W, C), fmap2.reshape(B * N, H, W, C)
        else:
            B, N, C, H, W = fmap1.shape
            f1, f2 = fmap1.reshape(B * N, C, H, W), fmap2.reshape(B * N, C, H, W)
            if fmap1.dtype in (torch.float16, torch.bfloat16):   # matrix-core path wants channels-last rows
                f1, f2 = f1.permute(0, 2, 3, 1), f2.permute(0, 2, 3, 1)
                channels_last = True
        self.corr_pyramid = db.corr_build(f1.contiguous(), f2.contiguous(), num_levels, channels_last=channels_last)

    @staticmethod
    def _build_differentiable(fmap1, fmap2, num_levels):
        corr = CorrBlock.corr(fmap1, fmap2)
        batch, num, h1, w1, h2, w2 = corr.shape
        corr = corr.reshape(batch * num * h1 * w1, 1, h2, w2)
        pyramid = []
        for i in range(num_levels):
            pyramid.append(corr.view(batch * num, h1, w1, h2 // 2 ** i, w2 // 2 ** i))
            if i + 1 < num_levels:
                corr = F.avg_pool2d(corr, 2, stride=2)
        return pyramid

    @staticmethod
    def corr(fmap1, fmap2):
        """all-pairs correlation, reference formulation (corr.py:63-71)"""
        batch, num, dim, ht, wd = fmap1.shape
        a = fmap1.reshape(batch * num, dim, ht * wd) / 4.0
        b = fmap2.reshape(batch * num, dim, ht * wd) / 4.0
        return torch.matmul(a.transpose(1, 2), b).view(batch, num, ht, wd, ht, wd)

    def __call__(self, coords, channels_last=False):
        """[B,N,196,H,W]; channels_last=True returns the same tensor stored [B,N,H,W,196]"""
        batch, num, ht, wd, _ = coords.shape
        coords = coords.reshape(batch * num, ht, wd, 2)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.corr_pyramid):
            c = coords.permute(0, 3, 1, 2).contiguous().float()
            out = [CorrSampler.apply(self.corr_pyramid[i], c / 2 ** i, self.radius).view(batch, num, -1, ht, wd)
                   for i in range(self.num_levels)]
            return torch.cat(out, dim=2)
        out = db.corr_pyramid_lookup([p.contiguous() for p in self.corr_pyramid], coords.float().contiguous(),
                                     self.radius, channels_last=channels_last)
        return out.unflatten(0, (batch, num))

    def cat(self, other):
        self.corr_pyramid = [torch.cat([a, b], 0) for a, b in zip(self.corr_pyramid, other.corr_pyramid)]
        return self

    def __getitem__(self, index):
        self.corr_pyramid = [p[index] for p in self.corr_pyramid]
        return self


class CorrLayer(torch.autograd.Function):
    """altcorr forward/backward pair (modules/corr.py:74-88)"""

    @staticmethod
    def forward(ctx, fmap1, fmap2, coords, r):
        ctx.r = r
        ctx.save_for_backward(fmap1, fmap2, coords)
        corr, = db.altcorr_forward(fmap1, fmap2, coords, r)
        return corr

    @staticmethod
    def backward(ctx, grad_corr):
        fmap1, fmap2, coords = ctx.saved_tensors
        g1, g2, gc = db.altcorr_backward(fmap1, fmap2, coords, grad_corr.contiguous(), ctx.r)
        return g1, g2, gc, None


class AltCorrBlock:
    """volume-free correlation (modules/corr.py:91-139): features pyramid in channels-last,
    dot products on the fly.  `fmaps` [B,N,C,H,W] (reference layout) or [B,N,H,W,C]."""

    def __init__(self, fmaps, num_levels=4, radius=3, channels_last=False):
        self.num_levels, self.radius = num_levels, radius
        if channels_last:
            fmaps = fmaps.permute(0, 1, 4, 2, 3)
        B, N, C, H, W = fmaps.shape
        f = fmaps.reshape(B * N, C, H, W) / 4.0
        self.pyramid = []
        for i in range(num_levels):
            self.pyramid.append(f.permute(0, 2, 3, 1).contiguous().view(B, N, H // 2 ** i, W // 2 ** i, C))
            if i + 1 < num_levels:
                f = F.avg_pool2d(f, 2, stride=2)

    def corr_fn(self, coords, ii, jj):
        B, N, H, W, S, _ = coords.shape
        coords = coords.permute(0, 1, 4, 2, 3, 5)
        out = []
        for i in range(self.num_levels):
            f1 = self.pyramid[0][:, ii]
            f2 = self.pyramid[i][:, jj]
            c = (coords / 2 ** i).reshape(B * N, S, H, W, 2).contiguous().float()
            f1 = f1.reshape((B * N,) + f1.shape[2:]).float().contiguous()
            f2 = f2.reshape((B * N,) + f2.shape[2:]).float().contiguous()
            corr = CorrLayer.apply(f1, f2, c, self.radius)
            out.append(corr.view(B, N, S, -1, H, W).permute(0, 1, 3, 4, 5, 2))
        return torch.cat(out, dim=2)

    def __call__(self, coords, ii, jj):
        squeeze = coords.dim() == 5
        if squeeze:
            coords = coords.unsqueeze(-2)
        corr = self.corr_fn(coords, ii, jj)
        return (corr.squeeze(-1) if squeeze else corr).contiguous()


class CorrVolumePool:
    """Resident pool of correlation-volume pyramids for a factor graph whose edge set changes every keyframe.

    The reference keeps one dense pyramid tensor and, whenever edges are dropped or added, re-indexes it and
    torch.cat's the new edges onto it (factor_graph.py:135,177, modules/corr.py:52-60): at 25 MB per edge that is a
    ~1.5 GB copy per keyframe for a 36-edge window.  Here every edge owns a SLOT of preallocated level tensors
    (capacity x 25 MB: 1.6 GB for 64 slots, irrelevant against 288 GB of HBM); the build kernel writes new edges
    straight into free slots and the lookup kernel follows a per-edge slot index.  No volume is ever moved."""
    supports_channels_last = True

    def __init__(self, capacity, ht, wd, device, dtype=torch.float16, num_levels=4, radius=3):
        self.num_levels, self.radius, self.capacity = num_levels, radius, capacity
        # 8x8-tiled planes when the shape allows it: an 8x8 tap window then spans <= 4 cache lines instead of 8
        self.tiled = num_levels == 4 and radius == 3 and db.tiled_supported(ht, wd, dtype)
        if self.tiled:
            self.levels = [torch.empty((capacity, ht, wd) + db.tiled_level_shape(ht, wd, l), dtype=dtype, device=device)
                           for l in range(num_levels)]
        else:
            self.levels = [torch.empty(capacity, ht, wd, ht >> l, wd >> l, dtype=dtype, device=device) for l in range(num_levels)]
        self.free = list(range(capacity - 1, -1, -1))
        self.slots = []                       # slot of each active edge, in edge order
        self._slots_t = None
        self.device = device
        self.extra = {}                       # other per-edge data that never changes over an edge's life, by slot (`put`)
        self._last = None                     # slots handed out by the most recent add(), as a device tensor
        self._reserved = None

    def __len__(self):
        return len(self.slots)

    def reserve(self, n):
        """take n free slots for edges about to be added (host list); pass their device copy to add()"""
        if n > len(self.free):
            self._grow(len(self.slots) + n)
        self._reserved = [self.free.pop() for _ in range(n)]
        return self._reserved

    def add(self, fmap1, fmap2, slots_t=None):
        """fmap1, fmap2: [n,H,W,C] channels-last features of the new edges (appended in order).  slots_t: device copy of
        the slots reserve(n) returned (a caller that uploads other index tables anyway packs them into the same copy)"""
        n = fmap1.shape[0]
        if slots_t is None:
            new = self.reserve(n)
            st = db.to_device_async(new, torch.int32, self.device)
        else:
            new, st = self._reserved, slots_t
            if len(new) != n or st.numel() != n or st.dtype != torch.int32:
                raise ValueError("CorrVolumePool.add: slots_t does not match the slots of the last reserve()")
        self._reserved = None
        if self.tiled:
            db.corr_build_tiled(fmap1.contiguous(), fmap2.contiguous(), self.levels, st)
        else:
            db.corr_build(fmap1.contiguous(), fmap2.contiguous(), self.num_levels, channels_last=True, out=self.levels, out_slots=st)
        self.slots += new
        self._slots_t = None
        self._last = st

    def put(self, name, rows):
        """store `rows` [n, ...] (one row per edge of the most recent add(), same order) in the slots those edges own.
        The factor graph keeps the ConvGRU's static-input terms here: written once per edge, read by slot
        (pvo_gru_conv_gates' p_slots), never gathered or concatenated when the edge set changes."""
        t = self.extra.get(name)
        if t is None:
            t = self.extra[name] = torch.empty((self.capacity,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device,
                                               memory_format=torch.channels_last if rows.dim() == 4 and rows.is_contiguous(memory_format=torch.channels_last) else torch.contiguous_format)
        t.index_copy_(0, self._last.long(), rows)
        return t

    def _grow(self, need):
        """more edges than slots (the reference's pyramid simply grows, e.g. a long --warmup initialisation): move to
        larger level tensors; live slots keep their numbers.  Rare, so the one-off copy of the pool is acceptable."""
        cap = max(need + 16, self.capacity + self.capacity // 2)
        levels = []
        for lv in self.levels:
            new = torch.empty((cap,) + tuple(lv.shape[1:]), dtype=lv.dtype, device=lv.device)
            new[:self.capacity].copy_(lv)
            levels.append(new)
        self.levels = levels
        for name, t in list(self.extra.items()):
            cl = t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)
            new = torch.empty((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device,
                              memory_format=torch.channels_last if cl else torch.contiguous_format)
            new[:self.capacity].copy_(t)
            self.extra[name] = new
        self.free = list(range(cap - 1, self.capacity - 1, -1)) + self.free
        self.capacity = cap

    def keep(self, mask):
        """drop the edges whose mask entry is False"""
        for s, m in zip(self.slots, mask):
            if not m:
                self.free.append(s)
        self.slots = [s for s, m in zip(self.slots, mask) if m]
        self._slots_t = None

    def slots_tensor(self):
        if self._slots_t is None:
            self._slots_t = db.to_device_async(self.slots, torch.int32, self.device)
        return self._slots_t

    def encoded(self, coords, enc_weight, enc_bias):
        """relu(W lookup(coords) + b): the lookup fused with the update operator's first correlation-encoder layer
        (tiled pools only); [batch*num, 128, ht, wd] channels-last.  (The native update runs the same kernel itself.)"""
        batch, num, ht, wd, _ = coords.shape
        return db.corr_lookup_encode_tiled(self.levels, coords.reshape(batch * num, ht, wd, 2).float().contiguous(),
                                           enc_weight, enc_bias, slots=self.slots_tensor())

    def at(self, coords):
        """the pool sampled at `coords` [batch,num,ht,wd,2], as an argument of the fused update operator: the lookup then
        runs inside libpvo_hip, fused with corr_encoder[0] (tiled pools only)"""
        from .update import PoolLookup
        batch, num, ht, wd, _ = coords.shape
        return PoolLookup(self.levels, self.slots_tensor(), self.capacity, coords.reshape(batch * num, ht, wd, 2).float().contiguous())

    def __call__(self, coords, channels_last=False):
        batch, num, ht, wd, _ = coords.shape
        if self._slots_t is None:
            self._slots_t = db.to_device_async(self.slots, torch.int32, self.device)
        c = coords.reshape(batch * num, ht, wd, 2).float().contiguous()
        if self.tiled:
            out = db.corr_pyramid_lookup_tiled(self.levels, c, channels_last=channels_last, slots=self._slots_t)
        else:
            out = db.corr_pyramid_lookup(self.levels, c, self.radius, channels_last=channels_last, slots=self._slots_t)
        return out.unflatten(0, (batch, num))
