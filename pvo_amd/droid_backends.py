"""`droid_backends` — the reference's native op surface, backed by libpvo_hip.so.

Mirrors the pybind module of the reference (VO_Module/src/droid.cpp:234-247):
same function names, positional argument order, in-place semantics and error
behaviour (RuntimeError "<name> must be contiguous", droid.cpp:83-84).  Each
function forwards raw device pointers + sizes + torch's CURRENT HIP stream to the
C ABI of include/pvo_hip.h.  There is no CPU / PyTorch fallback: CPU tensors, a
missing shared library or a failing launch raise.

A maintainer of the reference replaces `import droid_backends` with
`from pvo_amd import droid_backends`.
"""
import ctypes
import itertools

import torch

from . import _lib
from ._lib import PvoHipError, check

_DT = {torch.float32: _lib.PVO_F32, torch.float16: _lib.PVO_F16,
       torch.bfloat16: _lib.PVO_BF16, torch.float64: _lib.PVO_F64}

_ws_cache = {}


def _contig(t, name):
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)  # droid.cpp:83 CHECK_CONTIGUOUS


def _dev(*ts):
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise PvoHipError("droid_backends (MI355X build) needs device tensors; got a CPU tensor "
                              "(there is no CPU fallback)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise PvoHipError("all tensors must live on the same device")
    return dev


class _PinnedRing:
    """One persistent page-locked staging buffer per process.  `tensor.pin_memory()` per call is a fresh page-locked
    allocation whenever the caching host allocator cannot recycle a block (its blocks are only recycled once the copy
    that used them has completed, and the host runs ahead of the GPU), i.e. a hipHostMalloc - milliseconds, and a
    device-wide lock - in the middle of the update loop.  Small index lists are copied into this ring instead; a quarter
    of the ring is reused only after the copies issued from it have completed (an event per quarter)."""

    def __init__(self, nbytes=1 << 20, pin=True):
        buf = torch.empty(nbytes, dtype=torch.uint8)
        self.buf = buf.pin_memory() if pin else buf
        self.nbytes, self.quarter = nbytes, nbytes // 4
        self.off = 0                  # next free byte, always inside quarter `self.q`
        self.q = 0                    # the quarter allocations currently come from (tracked, not derived from `off`:
        self.events = [None] * 4      # an allocation that ends exactly on a boundary leaves `off` in the next quarter's range)

    def _record(self, device):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        return ev

    def take(self, n, device):
        n = (n + 15) & ~15
        if n > self.quarter:
            return None
        if self.off + n > (self.q + 1) * self.quarter:        # does not fit in what is left of the current quarter: move on
            self.events[self.q] = self._record(device)        # copies issued from the quarter being left
            self.q = (self.q + 1) % 4
            self.off = self.q * self.quarter
            if self.events[self.q] is not None:               # copies issued from this quarter one lap ago
                self.events[self.q].synchronize()
                self.events[self.q] = None
        seg = self.buf[self.off:self.off + n]
        self.off += n
        return seg


_ring = None


def to_device_async(values, dtype, device):
    """host sequence -> device tensor through the pinned staging ring and an asynchronous copy: unlike
    torch.tensor(values, device=...) this neither drains the stream nor allocates page-locked memory"""
    global _ring
    t = values.to(dtype) if isinstance(values, torch.Tensor) else torch.tensor(values, dtype=dtype)
    device = torch.device(device)
    if device.type != "cuda" or t.numel() == 0:
        return t.to(device)
    if _ring is None:
        _ring = _PinnedRing()
    nbytes = t.numel() * t.element_size()
    seg = _ring.take(nbytes, device)
    if seg is None:                                           # larger than a quarter of the ring: rare, one-off
        return t.pin_memory().to(device, non_blocking=True)
    stage = seg[:nbytes].view(dtype).view(t.shape)
    stage.copy_(t)
    out = torch.empty(t.shape, dtype=dtype, device=device)
    out.copy_(stage, non_blocking=True)
    return out


def to_device_packed(parts, device):
    """several host index lists -> device tensors in ONE staged upload: parts = [(values, dtype), ...]; returns one
    tensor per part, each a 16-byte aligned view of a single device buffer.  An edge-set change needs about ten small
    index tables (edge lists, slots, segment CSR, damping rows ...); as separate copies each is a ~4 us blit on the
    stream, which at 200 keyframes/s is worth counting."""
    device = torch.device(device)
    ts = [(v.to(dt) if isinstance(v, torch.Tensor) else torch.tensor(v, dtype=dt)).reshape(-1) for v, dt in parts]
    if device.type != "cuda":
        return [t.to(device) for t in ts]
    offs, total = [], 0
    for t in ts:
        offs.append(total)
        total += (t.numel() * t.element_size() + 15) & ~15
    if total == 0:
        return [t.to(device) for t in ts]
    host = torch.empty(total, dtype=torch.uint8)
    for t, o in zip(ts, offs):
        if t.numel():
            host[o:o + t.numel() * t.element_size()].view(t.dtype).copy_(t)
    dev = to_device_async(host, torch.uint8, device)
    return [dev[o:o + t.numel() * t.element_size()].view(t.dtype) for t, o in zip(ts, offs)]


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else ctypes.c_void_p(0)


def _dtype_code(t, what):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise PvoHipError("%s: unsupported dtype %s" % (what, t.dtype))


def _long(t, name):
    if t.dtype != torch.int64:
        raise PvoHipError("%s must be int64 (the reference reads it through packed_accessor32<long>)" % name)


def _f32(t, name):
    if t.dtype != torch.float32:
        raise PvoHipError("%s must be float32" % name)


def _workspace(dev, nbytes):
    key = (dev.index if dev.index is not None else torch.cuda.current_device())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=dev)
        _ws_cache[key] = buf
    return buf


# --------------------------------------------------------------------------- correlation
def corr_index_forward(volume, coords, radius):
    """droid.cpp:167-175. volume [N,h1,w1,h2,w2], coords [N,2,h1,w1] f32 -> [corr [N,2r+1,2r+1,h1,w1]]."""
    _contig(volume, "volume"); _contig(coords, "coords")
    dev = _dev(volume, coords)
    _f32(coords, "coords")
    N, h1, w1, h2, w2 = volume.shape
    rd = 2 * radius + 1
    corr = torch.empty((N, rd, rd, h1, w1), dtype=volume.dtype, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        check(lib.pvo_corr_index_forward(_ptr(volume), _ptr(coords), _ptr(corr), N, h1, w1, h2, w2,
                                         radius, _dtype_code(volume, "volume"), _stream(dev)),
              "corr_index_forward")
    return [corr]


def corr_index_backward(volume, coords, corr_grad, radius):
    """droid.cpp:177-188 -> [volume_grad] (same shape/dtype as volume)."""
    _contig(volume, "volume"); _contig(coords, "coords"); _contig(corr_grad, "corr_grad")
    dev = _dev(volume, coords, corr_grad)
    _f32(coords, "coords")
    if corr_grad.dtype != volume.dtype:
        raise PvoHipError("corr_grad dtype must match volume dtype")
    N, h1, w1, h2, w2 = volume.shape
    volume_grad = torch.empty_like(volume)
    lib = _lib.load()
    with torch.cuda.device(dev):
        check(lib.pvo_corr_index_backward(_ptr(coords), _ptr(corr_grad), _ptr(volume_grad),
                                          N, h1, w1, h2, w2, radius,
                                          _dtype_code(volume, "volume"), _stream(dev)),
              "corr_index_backward")
    return [volume_grad]


def corr_pyramid_lookup(pyramid, coords, radius, channels_last=False, slots=None):
    """All levels of CorrBlock.__call__ (modules/corr.py:40-50) in one launch.

    pyramid: list of level tensors [N,h1,w1,h2>>l,w2>>l]; coords [N,h1,w1,2] f32
    -> [N, L*(2r+1)^2, h1, w1]; with channels_last=True the same tensor is returned with
    channels-last strides (stored [N,h1,w1,L*(2r+1)^2]), ready for NHWC convolutions.
    Not part of the reference surface: it is what the Python loop + torch.cat computes, fused."""
    dev = _dev(coords, *pyramid)
    _contig(coords, "coords"); _f32(coords, "coords")
    for lv in pyramid:
        _contig(lv, "volume")
    NV, h1, w1, h2, w2 = pyramid[0].shape
    L = len(pyramid)
    for l, lv in enumerate(pyramid):
        if tuple(lv.shape) != (NV, h1, w1, h2 >> l, w2 >> l) or lv.dtype != pyramid[0].dtype:
            raise PvoHipError("pyramid level %d has shape %s, expected %s"
                              % (l, tuple(lv.shape), (NV, h1, w1, h2 >> l, w2 >> l)))
    N = coords.shape[0]
    if slots is None:
        if N != NV:
            raise PvoHipError("coords has %d edges but the pyramid holds %d volumes" % (N, NV))
    elif slots.dtype != torch.int32 or slots.numel() != N or not slots.is_cuda:
        raise PvoHipError("slots must be a device int32 tensor with one entry per edge")
    rd = 2 * radius + 1
    if channels_last:
        out = torch.empty((N, h1, w1, L * rd * rd), dtype=pyramid[0].dtype, device=dev).permute(0, 3, 1, 2)
    else:
        out = torch.empty((N, L * rd * rd, h1, w1), dtype=pyramid[0].dtype, device=dev)
    ptrs = (ctypes.c_void_p * L)(*[lv.data_ptr() if lv.numel() else 0 for lv in pyramid])
    lib = _lib.load()
    with torch.cuda.device(dev):
        check(lib.pvo_corr_pyramid_lookup(ptrs, _ptr(coords), _ptr(out), N, h1, w1, h2, w2, L, radius,
                                          _dtype_code(pyramid[0], "volume"), 1 if channels_last else 0,
                                          _ptr(slots), NV, _stream(dev)),
              "corr_pyramid_lookup")
    return out


def tiled_level_shape(ht, wd, level):
    """trailing dims (th, tw, 8, 8) of level `level` of an 8x8-tiled pyramid over an ht x wd target image"""
    return (((ht >> level) + 7) // 8, ((wd >> level) + 7) // 8, 8, 8)


def tiled_supported(ht, wd, dtype):
    return dtype in (torch.float16, torch.bfloat16) and wd >= 8 and ht >= 8


def corr_build_tiled(fmap1, fmap2, out, out_slots):
    """pvo_corr_build_tiled: edge n -> slot out_slots[n] of the tiled level tensors `out`
    ([slots, H, W, th, tw, 8, 8] per level); fmap1/fmap2 [N,H,W,C] channels-last 16-bit."""
    _contig(fmap1, "fmap1"); _contig(fmap2, "fmap2")
    dev = _dev(fmap1, fmap2, out_slots, *out)
    N, H, W, C = fmap1.shape
    if fmap1.shape != fmap2.shape or fmap1.dtype != fmap2.dtype or len(out) != 4:
        raise PvoHipError("corr_build_tiled: fmap mismatch or not 4 levels")
    for l, lv in enumerate(out):
        if tuple(lv.shape[1:]) != (H, W) + tiled_level_shape(H, W, l) or lv.dtype != fmap1.dtype or not lv.is_contiguous():
            raise PvoHipError("corr_build_tiled: level %d has shape %s" % (l, tuple(lv.shape)))
    if out_slots.dtype != torch.int32 or out_slots.numel() != N:
        raise PvoHipError("corr_build_tiled: out_slots must be device int32, one per edge")
    ptrs = (ctypes.c_void_p * 4)(*[lv.data_ptr() for lv in out])
    with torch.cuda.device(dev):
        check(_lib.load().pvo_corr_build_tiled(_ptr(fmap1), _ptr(fmap2), ptrs, N, C, H, W, _dtype_code(fmap1, "fmap"),
                                               _ptr(out_slots), _stream(dev)), "corr_build_tiled")
    return out


def corr_pyramid_lookup_tiled(pyramid, coords, channels_last=False, slots=None):
    """radius-3 lookup of all levels of an 8x8-tiled pyramid (see corr_build_tiled); same result as
    corr_pyramid_lookup on the row-major pyramid."""
    dev = _dev(coords, *pyramid)
    _contig(coords, "coords"); _f32(coords, "coords")
    NV, h1, w1 = pyramid[0].shape[:3]
    th0, tw0 = pyramid[0].shape[3:5]
    L = len(pyramid)
    N = coords.shape[0]
    if slots is None:
        if N != NV:
            raise PvoHipError("coords has %d edges but the pyramid holds %d volumes" % (N, NV))
    elif slots.dtype != torch.int32 or slots.numel() != N or not slots.is_cuda:
        raise PvoHipError("slots must be a device int32 tensor with one entry per edge")
    for l, lv in enumerate(pyramid):
        if tuple(lv.shape) != (NV, h1, w1) + tiled_level_shape(h1, w1, l) or not lv.is_contiguous():
            raise PvoHipError("tiled pyramid level %d has shape %s" % (l, tuple(lv.shape)))
    if channels_last:
        out = torch.empty((N, h1, w1, L * 49), dtype=pyramid[0].dtype, device=dev).permute(0, 3, 1, 2)
    else:
        out = torch.empty((N, L * 49, h1, w1), dtype=pyramid[0].dtype, device=dev)
    ptrs = (ctypes.c_void_p * L)(*[lv.data_ptr() for lv in pyramid])
    with torch.cuda.device(dev):
        check(_lib.load().pvo_corr_pyramid_lookup_tiled(ptrs, _ptr(coords), _ptr(out), N, h1, w1, h1, w1, L,
                                                        _dtype_code(pyramid[0], "volume"), 1 if channels_last else 0,
                                                        _ptr(slots), NV, _stream(dev)), "corr_pyramid_lookup_tiled")
    return out


def corr_encoder_weights(weight, dtype):
    """Conv2d(196,128,1) filter [128,196,1,1] -> the zero-padded [128,224] matrix pvo_corr_lookup_encode_tiled reads"""
    if tuple(weight.shape[:2]) != (128, 196):
        raise PvoHipError("corr_encoder_weights: filter must be [128,196,1,1]")
    w = torch.zeros(128, 224, dtype=dtype, device=weight.device)
    w[:, :196] = weight.detach().reshape(128, 196).to(dtype)
    return w


def corr_lookup_encode_tiled(pyramid, coords, enc_weight, enc_bias, slots=None):
    """relu(W corr + b) with corr = the radius-3 lookup of the 8x8-tiled pyramid, in one kernel
    -> [N,128,h1,w1] stored channels-last (the input of corr_encoder[2])."""
    dev = _dev(coords, enc_weight, enc_bias, *pyramid)
    _contig(coords, "coords"); _f32(coords, "coords")
    NV, h1, w1 = pyramid[0].shape[:3]
    N = coords.shape[0]
    if len(pyramid) != 4:
        raise PvoHipError("corr_lookup_encode_tiled: 4 levels required")
    if slots is None:
        if N != NV:
            raise PvoHipError("coords has %d edges but the pyramid holds %d volumes" % (N, NV))
    elif slots.dtype != torch.int32 or slots.numel() != N or not slots.is_cuda:
        raise PvoHipError("slots must be a device int32 tensor with one entry per edge")
    for l, lv in enumerate(pyramid):
        if tuple(lv.shape) != (NV, h1, w1) + tiled_level_shape(h1, w1, l) or not lv.is_contiguous():
            raise PvoHipError("tiled pyramid level %d has shape %s" % (l, tuple(lv.shape)))
    if tuple(enc_weight.shape) != (128, 224) or enc_weight.dtype != pyramid[0].dtype or not enc_weight.is_contiguous():
        raise PvoHipError("enc_weight must be the [128,224] tensor of corr_encoder_weights in the volume dtype")
    out = torch.empty((N, h1, w1, 128), dtype=pyramid[0].dtype, device=dev).permute(0, 3, 1, 2)
    ptrs = (ctypes.c_void_p * 4)(*[lv.data_ptr() for lv in pyramid])
    with torch.cuda.device(dev):
        check(_lib.load().pvo_corr_lookup_encode_tiled(ptrs, _ptr(coords), _ptr(enc_weight), _bias(enc_bias, 128, "enc_bias"),
                                                       _ptr(out), N, h1, w1, _dtype_code(pyramid[0], "volume"),
                                                       _ptr(slots), NV, _stream(dev)), "corr_lookup_encode_tiled")
    return out


def altcorr_forward(fmap1, fmap2, coords, radius):
    """droid.cpp:190-200. fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C], coords [B,S,H1,W1,2] -> [corr [B,S,(2r+1)^2,H1,W1]]"""
    _contig(fmap1, "fmap1"); _contig(fmap2, "fmap2"); _contig(coords, "coords")
    dev = _dev(fmap1, fmap2, coords)
    _f32(coords, "coords")
    if fmap1.dtype != torch.float32 or fmap2.dtype != torch.float32:
        raise PvoHipError("altcorr_forward: float32 features only (AltCorrBlock casts with .float(), corr.py:120)")
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    S = coords.shape[1]
    rd = 2 * radius + 1
    corr = torch.empty(B, S, rd * rd, H1, W1, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_altcorr_forward(_ptr(fmap1), _ptr(fmap2), _ptr(coords), _ptr(corr), B, S, H1, W1, H2, W2, C,
                                              radius, _lib.PVO_F32, _stream(dev)), "altcorr_forward")
    return [corr]


def altcorr_backward(fmap1, fmap2, coords, corr_grad, radius):
    """droid.cpp:202-214 -> [fmap1_grad, fmap2_grad, coords_grad (zeros, as altcorr_kernel.cu:340)]"""
    for t, n in ((fmap1, "fmap1"), (fmap2, "fmap2"), (coords, "coords"), (corr_grad, "corr_grad")):
        _contig(t, n)
    dev = _dev(fmap1, fmap2, coords, corr_grad)
    if any(t.dtype != torch.float32 for t in (fmap1, fmap2, coords, corr_grad)):
        raise PvoHipError("altcorr_backward: float32 only (altcorr_kernel.cu:345)")
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    S = coords.shape[1]
    g1, g2 = torch.empty_like(fmap1), torch.empty_like(fmap2)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_altcorr_backward(_ptr(fmap1), _ptr(fmap2), _ptr(coords), _ptr(corr_grad), _ptr(g1), _ptr(g2),
                                               B, S, H1, W1, H2, W2, C, radius, _lib.PVO_F32, _stream(dev)),
              "altcorr_backward")
    return [g1, g2, torch.zeros(B, S, H1, W1, 2, dtype=torch.float32, device=dev)]


def corr_build(fmap1, fmap2, num_levels=4, channels_last=False, out=None, out_slots=None):
    """CorrBlock.corr + the avg-pool pyramid (modules/corr.py:24-38,63-71) in one launch.

    fmap1, fmap2: [N,C,H,W] (channels_last=False) or [N,H,W,C] (channels_last=True).
    Returns the pyramid: level l is [N,H,W,H>>l,W>>l] in the feature dtype.
    out / out_slots: write edge n into slot out_slots[n] (device int32) of the existing level tensors `out`."""
    _contig(fmap1, "fmap1"); _contig(fmap2, "fmap2")
    dev = _dev(fmap1, fmap2)
    if fmap1.shape != fmap2.shape or fmap1.dtype != fmap2.dtype:
        raise PvoHipError("corr_build: fmap1/fmap2 shape or dtype mismatch")
    if channels_last:
        N, H, W, C = fmap1.shape
    else:
        N, C, H, W = fmap1.shape
    if out is not None:
        levels = out
        if out_slots is None or out_slots.dtype != torch.int32 or out_slots.numel() != N:
            raise PvoHipError("corr_build: out needs out_slots (device int32, one per edge)")
    else:
        levels = [torch.empty((N, H, W, H >> l, W >> l), dtype=fmap1.dtype, device=dev) for l in range(num_levels)]
    ptrs = (ctypes.c_void_p * num_levels)(*[lv.data_ptr() if lv.numel() else 0 for lv in levels])
    with torch.cuda.device(dev):
        check(_lib.load().pvo_corr_build(_ptr(fmap1), _ptr(fmap2), ptrs, N, C, H, W, num_levels,
                                         _dtype_code(fmap1, "fmap"), 1 if channels_last else 0, _ptr(out_slots), _stream(dev)),
              "corr_build")
    return levels


# --------------------------------------------------------------------------- reprojection family
def frame_distance(poses, disps, intrinsics, ii, jj, beta):
    """droid.cpp:117-133 -> dist [M]."""
    for t, n in ((poses, "poses"), (disps, "disps"), (intrinsics, "intrinsics"), (ii, "ii"), (jj, "jj")):
        _contig(t, n)
    dev = _dev(poses, disps, intrinsics, ii, jj)
    _f32(poses, "poses"); _f32(disps, "disps"); _f32(intrinsics, "intrinsics"); _long(ii, "ii"); _long(jj, "jj")
    M = ii.shape[0]
    dist = torch.empty(M, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_frame_distance(_ptr(poses), _ptr(disps), _ptr(intrinsics), _ptr(ii), _ptr(jj),
                                             _ptr(dist), M, disps.shape[1], disps.shape[2], float(beta),
                                             _stream(dev)), "frame_distance")
    return dist


def frame_distance_bidirectional(poses, disps, intrinsics, ii, jj, beta):
    """0.5 * (frame_distance(ii, jj) + frame_distance(jj, ii)) - DepthVideo.distance's bidirectional metric
    (depth_video.py:183-193) - in one launch, bit-identical to the two-call formulation -> dist [M]."""
    for t, n in ((poses, "poses"), (disps, "disps"), (intrinsics, "intrinsics"), (ii, "ii"), (jj, "jj")):
        _contig(t, n)
    dev = _dev(poses, disps, intrinsics, ii, jj)
    _f32(poses, "poses"); _f32(disps, "disps"); _f32(intrinsics, "intrinsics"); _long(ii, "ii"); _long(jj, "jj")
    M = ii.shape[0]
    dist = torch.empty(M, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_frame_distance_bidirectional(_ptr(poses), _ptr(disps), _ptr(intrinsics), _ptr(ii), _ptr(jj),
                                                           _ptr(dist), M, disps.shape[1], disps.shape[2], float(beta),
                                                           _stream(dev)), "frame_distance_bidirectional")
    return dist


def projmap(poses, disps, intrinsics, ii, jj):
    """droid.cpp:136-151 -> [coords [E,ht,wd,3], valid [E,ht,wd,1]]."""
    for t, n in ((poses, "poses"), (disps, "disps"), (intrinsics, "intrinsics"), (ii, "ii"), (jj, "jj")):
        _contig(t, n)
    dev = _dev(poses, disps, intrinsics, ii, jj)
    _long(ii, "ii"); _long(jj, "jj")
    E, ht, wd = ii.shape[0], disps.shape[1], disps.shape[2]
    coords = torch.empty(E, ht, wd, 3, dtype=torch.float32, device=dev)
    valid = torch.empty(E, ht, wd, 1, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_projmap(_ptr(poses), _ptr(disps), _ptr(intrinsics), _ptr(ii), _ptr(jj),
                                      _ptr(coords), _ptr(valid), E, ht, wd, _stream(dev)), "projmap")
    return [coords, valid]


def iproj(poses, disps, intrinsics):
    """droid.cpp:154-163 -> points [N,ht,wd,3]."""
    for t, n in ((poses, "poses"), (disps, "disps"), (intrinsics, "intrinsics")):
        _contig(t, n)
    dev = _dev(poses, disps, intrinsics)
    N, ht, wd = disps.shape
    pts = torch.empty(N, ht, wd, 3, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_iproj(_ptr(poses), _ptr(disps), _ptr(intrinsics), _ptr(pts), N, ht, wd,
                                    _stream(dev)), "iproj")
    return pts


def depth_filter(poses, disps, intrinsics, ix, thresh):
    """droid.cpp:217-231 -> counter [N,ht,wd]."""
    for t, n in ((poses, "poses"), (disps, "disps"), (intrinsics, "intrinsics"), (ix, "ix"), (thresh, "thresh")):
        _contig(t, n)
    dev = _dev(poses, disps, intrinsics, ix, thresh)
    _long(ix, "ix"); _f32(thresh, "thresh")
    N, (nf, ht, wd) = ix.shape[0], disps.shape
    counter = torch.empty(N, ht, wd, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_depth_filter(_ptr(poses), _ptr(disps), _ptr(intrinsics), _ptr(ix), _ptr(thresh),
                                           _ptr(counter), N, nf, ht, wd, _stream(dev)), "depth_filter")
    return counter


def reproject(poses, disps, intrinsics, ii, jj, out=None):
    """DepthVideo.reproject (depth_video.py:154-163): poses [F,7], disps [F,ht,wd],
    intrinsics [F,4] -> coords [E,ht,wd,2], valid [E,ht,wd,1].  out: a contiguous fp32 [E,ht,wd,2] tensor to write the
    coordinates into (e.g. the rows of a state buffer)."""
    for t, n in ((poses, "poses"), (disps, "disps"), (intrinsics, "intrinsics"), (ii, "ii"), (jj, "jj")):
        _contig(t, n)
    dev = _dev(poses, disps, intrinsics, ii, jj)
    _long(ii, "ii"); _long(jj, "jj")
    E, ht, wd = ii.shape[0], disps.shape[1], disps.shape[2]
    if out is not None:
        if tuple(out.shape) != (E, ht, wd, 2) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev:
            raise ValueError("reproject: out must be a contiguous float32 [E,ht,wd,2] tensor on the inputs' device")
        coords = out
    else:
        coords = torch.empty(E, ht, wd, 2, dtype=torch.float32, device=dev)
    valid = torch.empty(E, ht, wd, 1, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_reproject(_ptr(poses), _ptr(disps), _ptr(intrinsics), _ptr(ii), _ptr(jj),
                                        _ptr(coords), _ptr(valid), E, ht, wd, _stream(dev)), "reproject")
    return coords, valid


# --------------------------------------------------------------------------- bundle adjustment
def ba(poses, disps, intrinsics, targets, weights, eta, ii, jj, t0, t1, iterations, lm, ep, motion_only,
       status=None):
    """droid.cpp:87-114 / ba_cuda droid_kernels.cu:1293-1410.

    poses [F,7] and disps [F,ht,wd] are updated IN PLACE; returns [dx [P,6], dz [K,ht*wd]]
    (dz is an empty tensor when motion_only, where the reference returns an undefined one).
    Fully asynchronous: no host synchronisation.  `status` (optional int32[4] device tensor)
    receives [non-SPD seen, K, eta-row mismatch, 0]."""
    for t, n in ((targets, "targets"), (weights, "weights"), (poses, "poses"), (disps, "disps"),
                 (intrinsics, "intrinsics"), (ii, "ii"), (jj, "jj")):
        _contig(t, n)      # droid.cpp:103-109 checks exactly these
    dev = _dev(poses, disps, intrinsics, targets, weights, eta, ii, jj)
    for t, n in ((poses, "poses"), (disps, "disps"), (intrinsics, "intrinsics"), (targets, "targets"),
                 (weights, "weights")):
        _f32(t, n)
    _long(ii, "ii"); _long(jj, "jj")
    F, ht, wd = disps.shape
    HW = ht * wd
    E = ii.shape[0]
    t0, t1 = int(t0), int(t1)
    P = t1 - t0
    if poses.shape[0] < t1 or F < t1:
        raise PvoHipError("ba: pose window [%d,%d) exceeds the buffers (%d poses, %d depth maps)"
                          % (t0, t1, poses.shape[0], F))
    if eta is not None and not motion_only:
        _f32(eta, "eta")
        eta = eta.contiguous().view(-1, HW)     # droid_kernels.cu:1376 eta.view({-1, ht*wd})
        K_eta = eta.shape[0]
    else:
        K_eta = 1
        if not motion_only:
            raise PvoHipError("ba: eta is required unless motion_only")
    # K = |unique([t0,t1) U ii)| is only known on the device.  eta carries one row per depth
    # map (the reference's C + eta broadcast requires it), so K == K_eta unless eta is a single
    # broadcast row; only then is K read back (one sync) to size dz.
    if motion_only:
        K = 0
    elif K_eta > 1 or E + P == 0:
        K = K_eta if E + P > 0 else 0
    else:
        K = int(torch.unique(torch.cat([torch.arange(t0, t1, device=dev), ii])).numel())
    dx = torch.zeros(max(P, 0), 6, dtype=torch.float32, device=dev)
    dz = torch.zeros(K, HW, dtype=torch.float32, device=dev)
    lib = _lib.load()
    nbytes = lib.pvo_ba_workspace_bytes(E, P, F, HW)
    ws = _workspace(dev, nbytes)
    with torch.cuda.device(dev):
        check(lib.pvo_ba(_ptr(poses), _ptr(disps), _ptr(intrinsics), _ptr(targets), _ptr(weights),
                         _ptr(eta) if eta is not None else ctypes.c_void_p(0), _ptr(ii), _ptr(jj),
                         E, F, ht, wd, K_eta, t0, t1, int(iterations), float(lm), float(ep),
                         1 if motion_only else 0, _ptr(dx), _ptr(dz), K,
                         _ptr(status) if status is not None else ctypes.c_void_p(0),
                         ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream(dev)), "ba")
    return [dx, dz]


# --------------------------------------------------------------------------- edge-sharded BA pieces
def ba_workspace_bytes(E, P, nframes, HW):
    return int(_lib.load().pvo_ba_workspace_bytes(int(E), int(P), int(nframes), int(HW)))


def ba_workspace(E, P, nframes, HW, device):
    """a private workspace tensor for the split BA entry points (one per rank / graph)"""
    n = _lib.load().pvo_ba_workspace_bytes(int(E), int(P), int(nframes), int(HW))
    return torch.empty(int(n), dtype=torch.uint8, device=device)


def ba_last_partition(E, P, nframes, HW, device, workspace=None):
    """diagnostic: (m, s) of the partitioned pose solve the last BA of this size ran with on `device` - poses [0, m) and
    [s, P) were eliminated by two workgroups at once, the separator [m, s) last; (0, 0) = one chain.  `workspace`: the
    private workspace of the split entry points, default the one `ba` uses.  Synchronises."""
    dev = torch.device(device)
    lib = _lib.load()
    ws = workspace if workspace is not None else _workspace(dev, lib.pvo_ba_workspace_bytes(int(E), int(P), int(nframes), int(HW)))
    out = (ctypes.c_int * 2)()
    with torch.cuda.device(dev):
        check(lib.pvo_ba_last_partition(ctypes.c_void_p(ws.data_ptr()), ws.numel(), int(E), int(P), int(nframes), int(HW), out, _stream(dev)),
              "ba_last_partition")
    return int(out[0]), int(out[1])


def ba_plan(ii, jj, nframes, HW, K_eta, t0, t1, workspace):
    dev = _dev(ii, jj, workspace)
    _long(ii, "ii"); _long(jj, "jj")
    with torch.cuda.device(dev):
        check(_lib.load().pvo_ba_plan(_ptr(ii), _ptr(jj), ii.shape[0], int(nframes), int(HW), int(K_eta), int(t0), int(t1),
                                      ctypes.c_void_p(workspace.data_ptr()), workspace.numel(), _stream(dev)), "ba_plan")


BA_SYS_DTYPE = torch.int64      # the reduced pose system is 64-bit fixed point (units of 2^-28): all-reduce it as integers


def ba_local(poses, disps, intrinsics, targets, weights, eta, ii, jj, t0, t1, motion_only, sys, workspace, sys_is_zero=False):
    """assemble + eliminate this rank's edges; `sys` (int64 [(6P)^2+6P], fixed point) receives the local reduced system.
    sys_is_zero: the buffer was left zeroed by ba_finish (skips the clear)."""
    dev = _dev(poses, disps, intrinsics, targets, weights, eta, ii, jj, sys, workspace)
    F, ht, wd = disps.shape
    if sys.dtype != torch.int64 or sys.numel() < (6 * (t1 - t0)) ** 2 + 6 * (t1 - t0):
        raise PvoHipError("ba_local: sys must be int64 with (6P)^2 + 6P elements")
    K_eta = 1
    if eta is not None:
        eta = eta.contiguous().view(-1, ht * wd)
        K_eta = eta.shape[0]
    with torch.cuda.device(dev):
        check(_lib.load().pvo_ba_local(_ptr(poses), _ptr(disps), _ptr(intrinsics), _ptr(targets), _ptr(weights),
                                       _ptr(eta) if eta is not None else ctypes.c_void_p(0), _ptr(ii), _ptr(jj),
                                       ii.shape[0], F, ht, wd, K_eta, int(t0), int(t1), (1 if motion_only else 0) | (2 if sys_is_zero else 0),
                                       _ptr(sys), ctypes.c_void_p(workspace.data_ptr()), workspace.numel(),
                                       _stream(dev)), "ba_local")


def ba_packed_elems(first):
    """int64 elements of the packed envelope message for the structural envelope `first` (a host list, one entry per free pose)"""
    arr = (ctypes.c_int * len(first))(*[int(f) for f in first])
    return int(_lib.load().pvo_ba_packed_elems(arr, len(first)))


def ba_pack(sys, first_dev, msg):
    """after ba_local: the structural envelope (first_dev: int32 [P] on the device) of `sys` -> msg (int64, ba_packed_elems long);
    `sys` is left zeroed.  msg is what an edge-sharded step all-reduces; ba_finish(..., packed=(msg, first_dev)) consumes it."""
    dev = _dev(sys, first_dev, msg)
    if sys.dtype != torch.int64 or msg.dtype != torch.int64 or first_dev.dtype != torch.int32:
        raise PvoHipError("ba_pack: sys / msg must be int64, first int32")
    with torch.cuda.device(dev):
        check(_lib.load().pvo_ba_pack(_ptr(sys), _ptr(first_dev), _ptr(msg), int(first_dev.shape[0]), _stream(dev)), "ba_pack")
    return msg


def ba_finish(poses, disps, sys, ii, jj, t0, t1, lm, ep, motion_only, workspace, dz_rows=0, status=None, outputs=True,
              clamp_frames=0, disp_min=0.001, rider=None, packed=None):
    """damp + solve the (all-reduced) system (left zeroed afterwards), retract poses, back-substitute this rank's depths -> [dx, dz]
    (outputs=False: poses / disps are updated in place and no dx / dz tensors are produced -> [None, None]).
    clamp_frames > 0: disps[:clamp_frames].clamp_(min=disp_min) in the same launch.
    rider = (x, w, bias): an independent 1x1 convolution of x [N,128,H,W] (conv1x1_c128 without ReLU) computed by additional
    workgroups of the pose solve's dispatch (pvo_ba_finish_conv1x1); its result is appended to the returned list.
    packed = (msg, first_dev): the all-reduced envelope message of ba_pack instead of the dense `sys` (which is not read)."""
    dev = _dev(poses, disps, sys, ii, jj, workspace)
    rx = rw = rb = ry = None
    rrows = rC = rdt = 0
    if rider is not None:
        rx, rw, rb = rider
        _cl(rx, "rider x", 128)
        if rw.dim() != 2 or rw.shape[1] != 128 or rw.dtype != rx.dtype or not rw.is_contiguous():
            raise PvoHipError("ba_finish: rider w must be a contiguous [Cout,128] tensor in x's dtype")
        rC = rw.shape[0]
        ry = _new_cl(rx.shape[0], rC, rx.shape[2], rx.shape[3], rx.dtype, dev)
        rrows, rdt = rx.shape[0] * rx.shape[2] * rx.shape[3], _dtype_code(rx, "rider x")
    F, ht, wd = disps.shape
    P = int(t1) - int(t0)
    dx = torch.zeros(max(P, 0), 6, dtype=torch.float32, device=dev) if outputs else None
    dz = torch.zeros(int(dz_rows), ht * wd, dtype=torch.float32, device=dev) if outputs else None
    if not outputs:
        dz_rows = 0
    if packed is not None:
        msg, first_dev = packed
        if rider is not None:
            raise PvoHipError("ba_finish: a rider and a packed message together are not supported")
        if msg.dtype != torch.int64 or first_dev.dtype != torch.int32 or first_dev.shape[0] != P:
            raise PvoHipError("ba_finish: packed = (int64 message, int32 [P] structural envelope)")
        with torch.cuda.device(dev):
            check(_lib.load().pvo_ba_finish_packed(
                _ptr(poses), _ptr(disps), _ptr(msg), _ptr(first_dev), _ptr(ii), _ptr(jj), ii.shape[0], F, ht, wd,
                int(t0), int(t1), float(lm), float(ep), 1 if motion_only else 0,
                int(clamp_frames), float(disp_min), _ptr(dx), _ptr(dz), int(dz_rows),
                _ptr(status) if status is not None else ctypes.c_void_p(0),
                ctypes.c_void_p(workspace.data_ptr()), workspace.numel(), _stream(dev)), "ba_finish_packed")
        return [dx, dz]
    with torch.cuda.device(dev):
        check(_lib.load().pvo_ba_finish_conv1x1(
            _ptr(poses), _ptr(disps), _ptr(sys), _ptr(ii), _ptr(jj), ii.shape[0], F, ht, wd,
            int(t0), int(t1), float(lm), float(ep), 1 if motion_only else 0,
            int(clamp_frames), float(disp_min), _ptr(dx), _ptr(dz), int(dz_rows),
            _ptr(status) if status is not None else ctypes.c_void_p(0),
            ctypes.c_void_p(workspace.data_ptr()), workspace.numel(),
            _ptr(rx), _ptr(rw), _bias(rb, rC, "rider bias") if rider is not None else ctypes.c_void_p(0), _ptr(ry),
            int(rrows), int(rC), int(rdt), _stream(dev)), "ba_finish")
    return [dx, dz] if rider is None else [dx, dz, ry]


# --------------------------------------------------------------------------- update operator layers
def _cl(t, name, C):
    """a [E,C,H,W] tensor stored channels-last (physically [E,H,W,C]), 16-bit"""
    if t.dim() != 4 or t.shape[1] != C or not t.is_contiguous(memory_format=torch.channels_last):
        raise PvoHipError("%s must be a channels-last [E,%d,H,W] tensor" % (name, C))
    if t.dtype not in (torch.float16, torch.bfloat16):
        raise PvoHipError("%s must be float16 or bfloat16" % name)


def _bias(b, C, what):
    if b is None:
        return ctypes.c_void_p(0)
    if b.dtype != torch.float32 or b.numel() != C or not b.is_contiguous() or not b.is_cuda:
        raise PvoHipError("%s must be a contiguous float32 device vector of %d elements" % (what, C))
    return _ptr(b)


def _new_cl(E, C, H, W, dtype, dev):
    return torch.empty(E, H, W, C, dtype=dtype, device=dev).permute(0, 3, 1, 2)


def gru_glo_fused(net, w_weight, w_bias=None):
    """global context with the 1x1 conv folded in, as partial means over 256-pixel chunks:
    returns [E, K, 128] f32 with mean_px sigmoid(w(net) + b) * net == result.sum(1).
    net [E,128,H,W] channels-last 16-bit; w_weight [128,128] (or [128,128,1,1]) in net's dtype."""
    _cl(net, "net", 128)
    dev = _dev(net, w_weight)
    E, C, H, W = net.shape
    w2d = w_weight.reshape(128, 128)
    if w2d.dtype != net.dtype or not w2d.is_contiguous():
        raise PvoHipError("gru_glo_fused: w_weight must be a contiguous [128,128] tensor in net's dtype")
    lib = _lib.load()
    part = torch.empty(E, lib.pvo_gru_glo_chunks(H * W), C, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.pvo_gru_glo_fused(_ptr(net), _ptr(w2d), _bias(w_bias, C, "w_bias"), _ptr(part), E, H * W,
                                    _dtype_code(net, "net"), _stream(dev)), "gru_glo_fused")
    return part


def gate_context(glo_part, wg_t, g_bias):
    """g [E,384] f32 = (sum over chunks of glo_part [E,K,128]) @ wg_t [128,384] + g_bias [384]: the ConvGRU's three
    global-context 1x1 convolutions (gru.py:13-15) with every per-channel bias of the gates folded in"""
    dev = _dev(glo_part, wg_t, g_bias)
    E, K, C = glo_part.shape
    for t, n in ((glo_part, "glo_part"), (wg_t, "wg_t"), (g_bias, "g_bias")):
        _f32(t, n); _contig(t, n)
    if C != 128 or tuple(wg_t.shape) != (128, 384) or g_bias.numel() != 384:
        raise PvoHipError("gate_context: glo_part [E,K,128], wg_t [128,384], g_bias [384]")
    g = torch.empty(E, 384, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_gate_context(_ptr(glo_part), _ptr(wg_t), _ptr(g_bias), _ptr(g), E, K, _stream(dev)), "gate_context")
    return g


def conv7x7_c8_weights(weight, dtype):
    """[128,8,7,7] conv filter -> the [52,128,8] tap-major layout pvo_conv7x7_c8 reads (3 zero taps of padding)"""
    co, ci, kh, kw = weight.shape
    if (co, ci, kh, kw) != (128, 8, 7, 7):
        raise PvoHipError("conv7x7_c8: filter must be [128,8,7,7]")
    w = torch.zeros(52, 128, 8, dtype=dtype, device=weight.device)
    w[:49] = weight.detach().permute(2, 3, 0, 1).reshape(49, 128, 8).to(dtype)
    return w.contiguous()


def conv7x7_c8(x, w_taps, bias):
    """relu(conv7x7(x) + bias): x [E,8,H,W] channels-last 16-bit -> [E,128,H,W] channels-last (flow_encoder[0:2])"""
    _cl(x, "x", 8)
    dev = _dev(x, w_taps, bias)
    E, _, H, W = x.shape
    if w_taps.dtype != x.dtype or tuple(w_taps.shape) != (52, 128, 8) or not w_taps.is_contiguous():
        raise PvoHipError("conv7x7_c8: w_taps must be the [52,128,8] tensor of conv7x7_c8_weights in x's dtype")
    y = _new_cl(E, 128, H, W, x.dtype, dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_conv7x7_c8(_ptr(x), _ptr(w_taps), _bias(bias, 128, "bias"), _ptr(y), E, H, W,
                                         _dtype_code(x, "x"), _stream(dev)), "conv7x7_c8")
    return y


def flow_encoder(x, w7_taps, bias7, w3_taps, bias3, out=None, out_offset=0):
    """relu(conv3x3(relu(conv7x7(x) + bias7)) + bias3) in one kernel (pvo_flow_encoder): x [E,8,H,W] channels-last 16-bit ->
    [E,64,H,W] channels-last, or a channel slice of `out`"""
    _cl(x, "x", 8)
    dev = _dev(x, w7_taps, bias7, w3_taps, out)
    E, _, H, W = x.shape
    if w7_taps.dtype != x.dtype or tuple(w7_taps.shape) != (52, 128, 8) or not w7_taps.is_contiguous():
        raise PvoHipError("flow_encoder: w7_taps must be the [52,128,8] tensor of conv7x7_c8_weights in x's dtype")
    if tuple(w3_taps.shape) != (9, 64, 128) or w3_taps.dtype != x.dtype or not w3_taps.is_contiguous():
        raise PvoHipError("flow_encoder: w3_taps must be the [9,64,128] tensor of conv3x3_c128_weights in x's dtype")
    ret, y, ys, yo = _out_slice(out, out_offset, E, 64, H, W, x.dtype, dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_flow_encoder(_ptr(x), _ptr(w7_taps), _bias(bias7, 128, "bias7"), _ptr(w3_taps), _bias(bias3, 64, "bias3"),
                                           _ptr(y), E, H, W, ys, yo, _dtype_code(x, "x"), _stream(dev)), "flow_encoder")
    return ret


def conv3x3_c128_weights(weight, dtype):
    """[Cout,128,3,3] conv filter -> the [9,Cout,128] tap-major layout pvo_conv3x3_c128 reads"""
    co, ci, kh, kw = weight.shape
    if (kh, kw) != (3, 3):
        raise PvoHipError("conv3x3: filter must be [Cout,Cin,3,3]")
    return weight.detach().permute(2, 3, 0, 1).reshape(9, co, ci).to(dtype).contiguous()


def conv3x3_weights(weight, dtype):
    """[Cout,Cin,3,3] conv filter (Cin % 32 == 0, Cout % 128 == 0) -> the layout pvo_conv3x3 / pvo_gru_conv_* read, as a
    [9,Cout,Cin]-shaped tensor holding MFMA-fragment order [Cout/128][Cin/32][9][2][2][2][64][8] (include/pvo_hip.h)"""
    co, ci, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or ci % 32 or co % 128:
        raise PvoHipError("conv3x3: filter must be [Cout,Cin,3,3] with Cin % 32 == 0 and Cout % 128 == 0")
    taps = weight.detach().permute(2, 3, 0, 1).reshape(9, co, ci).to(dtype)
    # (t, cg, wn, nt, li, cc, ks, kg, j) -> (cg, cc, t, wn, nt, ks, kg, li, j)
    f = taps.reshape(9, co // 128, 2, 2, 32, ci // 32, 2, 2, 8).permute(1, 5, 0, 2, 3, 6, 7, 4, 8)
    return f.contiguous().view(9, co, ci)


def _out_slice(out, out_offset, E, Cout, H, W, dtype, dev):
    """(tensor to return, pointer tensor, ystride, yoff): a fresh dense output, or a channel slice of `out`"""
    if out is None:
        y = _new_cl(E, Cout, H, W, dtype, dev)
        return y, y, 0, 0
    if out.dim() != 4 or out.shape[0] != E or tuple(out.shape[2:]) != (H, W) or out.dtype != dtype or \
            not out.is_contiguous(memory_format=torch.channels_last) or out_offset < 0 or out_offset + Cout > out.shape[1]:
        raise PvoHipError("out must be a channels-last [E,C>=offset+Cout,H,W] tensor of the input dtype")
    return out[:, out_offset:out_offset + Cout], out, out.shape[1], out_offset


def conv3x3(x, w_taps, bias=None, relu=False, out=None, out_offset=0):
    """act(conv3x3(x) + bias): x [E,Cin,H,W] channels-last 16-bit -> [E,Cout,H,W] channels-last (wide layers: Cin % 32
    == 0, Cout % 128 == 0).  out / out_offset: write into channels [out_offset, out_offset + Cout) of `out`."""
    dev = _dev(x, w_taps, out)
    E, Cin, H, W = x.shape
    _cl(x, "x", Cin)
    if w_taps.dim() != 3 or w_taps.shape[0] != 9 or w_taps.shape[2] != Cin or w_taps.dtype != x.dtype or not w_taps.is_contiguous():
        raise PvoHipError("conv3x3: w_taps must be the [9,Cout,Cin] tensor of conv3x3_weights in x's dtype")
    Cout = w_taps.shape[1]
    ret, y, ys, yo = _out_slice(out, out_offset, E, Cout, H, W, x.dtype, dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_conv3x3(_ptr(x), _ptr(w_taps), _bias(bias, Cout, "bias"), _ptr(y), E, H, W, Cin, Cout,
                                      1 if relu else 0, ys, yo, _dtype_code(x, "x"), _stream(dev)), "conv3x3")
    return ret


def conv3x3_c128(x, w_taps, bias=None, relu=False, out=None, out_offset=0):
    """act(conv3x3(x) + bias): x [E,128,H,W] channels-last 16-bit -> [E,Cout,H,W] channels-last, Cout in {64,128,256,512}"""
    _cl(x, "x", 128)
    dev = _dev(x, w_taps, out)
    E, _, H, W = x.shape
    if w_taps.dim() != 3 or w_taps.shape[0] != 9 or w_taps.shape[2] != 128 or w_taps.dtype != x.dtype or not w_taps.is_contiguous():
        raise PvoHipError("conv3x3_c128: w_taps must be the [9,Cout,128] tensor of conv3x3_weights in x's dtype")
    Cout = w_taps.shape[1]
    ret, y, ys, yo = _out_slice(out, out_offset, E, Cout, H, W, x.dtype, dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_conv3x3_c128(_ptr(x), _ptr(w_taps), _bias(bias, Cout, "bias"), _ptr(y), E, H, W, Cout,
                                           1 if relu else 0, ys, yo, _dtype_code(x, "x"), _stream(dev)), "conv3x3_c128")
    return ret


def gru_conv_gates(net, cf, w_taps, g, P_zr, p_slots=None):
    """gate convolution over [net | cf] + sigmoid gates in one kernel -> (Z, RN), each [E,128,H,W] channels-last.
    cf [E,C,H,W] (C % 32 == 0), w_taps [9,256,128+C], g [E,384] f32, P_zr [E,256,H,W]."""
    E, _, H, W = net.shape
    C = cf.shape[1]
    _cl(net, "net", 128); _cl(cf, "cf", C); _cl(P_zr, "P_zr", 256)
    dev = _dev(net, cf, w_taps, g, P_zr, p_slots)
    if p_slots is not None and (p_slots.dtype != torch.int32 or p_slots.numel() != E or not p_slots.is_contiguous()):
        raise PvoHipError("gru_conv_gates: p_slots must be a contiguous int32 [E] tensor")
    _f32(g, "g"); _contig(g, "g")
    if tuple(w_taps.shape) != (9, 256, 128 + C) or w_taps.dtype != net.dtype or not w_taps.is_contiguous():
        raise PvoHipError("gru_conv_gates: w_taps must be [9,256,128+C] in net's dtype")
    Z, RN = _new_cl(E, 128, H, W, net.dtype, dev), _new_cl(E, 128, H, W, net.dtype, dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_gru_conv_gates(_ptr(net), _ptr(cf), C, _ptr(w_taps), _ptr(g), _ptr(P_zr), _vp(p_slots), _ptr(Z), _ptr(RN),
                                             E, H, W, _dtype_code(net, "net"), _stream(dev)), "gru_conv_gates")
    return Z, RN


def gru_conv_candidate(RN, cf, w_taps, g, P_q, Z, net, p_slots=None):
    """candidate convolution over [RN | cf] + the GRU state update in one kernel -> new hidden state"""
    E, _, H, W = net.shape
    C = cf.shape[1]
    _cl(RN, "RN", 128); _cl(cf, "cf", C); _cl(P_q, "P_q", 128); _cl(Z, "Z", 128); _cl(net, "net", 128)
    dev = _dev(RN, cf, w_taps, g, P_q, Z, net)
    _f32(g, "g"); _contig(g, "g")
    if tuple(w_taps.shape) != (9, 128, 128 + C) or w_taps.dtype != net.dtype or not w_taps.is_contiguous():
        raise PvoHipError("gru_conv_candidate: w_taps must be [9,128,128+C] in net's dtype")
    out = _new_cl(E, 128, H, W, net.dtype, dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_gru_conv_candidate(_ptr(RN), _ptr(cf), C, _ptr(w_taps), _ptr(g), _ptr(P_q), _vp(p_slots), _ptr(Z), _ptr(net),
                                                 _ptr(out), E, H, W, _dtype_code(net, "net"), _stream(dev)), "gru_conv_candidate")
    return out


def corr_encode(corr, enc_weight, enc_bias):
    """relu(W corr + b): corr [E,196,H,W] channels-last 16-bit -> [E,128,H,W] channels-last (corr_encoder[0:2])"""
    _cl(corr, "corr", 196)
    dev = _dev(corr, enc_weight, enc_bias)
    E, _, H, W = corr.shape
    if tuple(enc_weight.shape) != (128, 224) or enc_weight.dtype != corr.dtype or not enc_weight.is_contiguous():
        raise PvoHipError("enc_weight must be the [128,224] tensor of corr_encoder_weights in the feature dtype")
    y = _new_cl(E, 128, H, W, corr.dtype, dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_corr_encode(_ptr(corr), _ptr(enc_weight), _bias(enc_bias, 128, "enc_bias"), _ptr(y), E * H * W,
                                          _dtype_code(corr, "corr"), _stream(dev)), "corr_encode")
    return y


def segment_mean(x, seg_ptr, seg_idx, K, in_bias=None):
    """out[k] = mean of x[seg_idx[e]] for e in [seg_ptr[k], seg_ptr[k+1]); x channels-last [E,C,H,W] -> [K,C,H,W].
    in_bias (f32 [C]): average relu(x + in_bias) instead (x = a bias-free convolution output)."""
    if x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last) or x.dtype not in (torch.float16, torch.bfloat16):
        raise PvoHipError("segment_mean: x must be a channels-last 16-bit [E,C,H,W] tensor")
    dev = _dev(x, seg_ptr, seg_idx)
    if seg_ptr.dtype != torch.int32 or seg_idx.dtype != torch.int32:
        raise PvoHipError("segment_mean: seg_ptr / seg_idx must be int32")
    E, C, H, W = x.shape
    out = _new_cl(K, C, H, W, x.dtype, dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_segment_mean(_ptr(x), _ptr(seg_ptr), _ptr(seg_idx), _bias(in_bias, C, "in_bias"), _ptr(out), K, H * W, C,
                                           _dtype_code(x, "x"), _stream(dev)), "segment_mean")
    return out


def heads_out(h1, bias1, w2, bias2):
    """y [E,8,H,W] (channels-last) = the four heads' second 3x3 convolutions of relu(h1 + bias1);
    h1 [E,512,H,W] channels-last 16-bit, w2 [4,2,9,128] same dtype, bias1 [512] / bias2 [8] float32"""
    _cl(h1, "h1", 512)
    dev = _dev(h1, bias1, w2, bias2)
    if tuple(w2.shape) != (4, 2, 9, 128) or w2.dtype != h1.dtype or not w2.is_contiguous():
        raise PvoHipError("heads_out: w2 must be a contiguous [4,2,9,128] tensor of the feature dtype")
    E, _, H, W = h1.shape
    y = _new_cl(E, 8, H, W, h1.dtype, dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_heads_out(_ptr(h1), _bias(bias1, 512, "bias1"), _ptr(w2), _bias(bias2, 8, "bias2"), _ptr(y),
                                        E, H, W, _dtype_code(h1, "h1"), _stream(dev)), "heads_out")
    return y


def heads2_fragments(w2, dtype):
    """second-stage filter of the four heads, [4,2,9,128] ([head][out][tap][channel]), as the MFMA B fragments
    pvo_conv3x3_heads multiplies the hidden activations with: [4][8 k-steps][64 lanes][8], column n = 2 tap + out"""
    w = w2.detach().to(dtype)
    m = torch.zeros(4, 32, 128, dtype=dtype, device=w.device)                 # [head][n][k]
    m[:, :18] = w.permute(0, 2, 1, 3).reshape(4, 18, 128)                       # n = tap * 2 + out
    # element (h, ks, lane, j) = m[h][lane & 31][ks*16 + (lane >> 5)*8 + j]
    f = m.reshape(4, 32, 8, 2, 8).permute(0, 2, 3, 1, 4)                        # [h][ks][kg][n][j]
    return f.reshape(4, 8, 64, 8).contiguous()


def heads_fused(x, w1_taps, bias1, w2_frags, bias2):
    """the four output heads from the hidden state in two launches and without the [E,512,H,W] intermediate:
    x [E,128,H,W] channels-last 16-bit, w1_taps = conv3x3_weights of the concatenated first-stage filters [512,128,3,3],
    w2_frags = heads2_fragments(...).  Returns y [E,8,H,W] channels-last as heads_out does."""
    _cl(x, "x", 128)
    dev = _dev(x, w1_taps, bias1, w2_frags, bias2)
    E, _, H, W = x.shape
    if tuple(w1_taps.shape) != (9, 512, 128) or w1_taps.dtype != x.dtype or not w1_taps.is_contiguous() or \
            tuple(w2_frags.shape) != (4, 8, 64, 8) or w2_frags.dtype != x.dtype or not w2_frags.is_contiguous():
        raise PvoHipError("heads_fused: w1_taps must be [9,512,128] and w2_frags [4,8,64,8] in x's dtype")
    z = torch.empty(E, H, W, 4, 18, dtype=torch.float32, device=dev)
    y = _new_cl(E, 8, H, W, x.dtype, dev)
    with torch.cuda.device(dev):
        lib = _lib.load()
        check(lib.pvo_conv3x3_heads(_ptr(x), _ptr(w1_taps), _bias(bias1, 512, "bias1"), _ptr(w2_frags), _ptr(z), E, H, W,
                                    _dtype_code(x, "x"), _stream(dev)), "conv3x3_heads")
        check(lib.pvo_heads_gather(_ptr(z), _bias(bias2, 8, "bias2"), _ptr(y), E, H, W, _dtype_code(x, "x"), _stream(dev)), "heads_gather")
    return y


def eta_head(x, w_taps, bias, frame=None, pos=None, damping=None, EP=0.0, eta_scale=0.2):
    """GraphAgg's eta head: x [K,128,H,W] channels-last 16-bit, w_taps [9,128], bias f32 [1].
    frame None -> 0.01 * softplus(conv(x) + bias) [K,H,W] f32 (what GraphAgg returns).
    frame int64 [R], pos int32 [R], damping f32 [buffer,H,W]: also FactorGraph's damping bookkeeping; returns the BA's
    eta [R,H,W] = eta_scale * damping[frame] + EP (see pvo_eta_head)."""
    _cl(x, "x", 128)
    dev = _dev(x, w_taps, bias, frame, pos, damping)
    K, _, H, W = x.shape
    if tuple(w_taps.shape) != (9, 128) or w_taps.dtype != x.dtype or not w_taps.is_contiguous():
        raise PvoHipError("eta_head: w_taps must be [9,128] in x's dtype")
    R = K
    if frame is not None:
        R = frame.shape[0]
        if frame.dtype != torch.int64 or pos.dtype != torch.int32 or pos.shape[0] != R or damping.dtype != torch.float32 \
                or tuple(damping.shape[1:]) != (H, W) or not damping.is_contiguous():
            raise PvoHipError("eta_head: frame int64 [R], pos int32 [R], damping contiguous f32 [*,H,W]")
    eta = torch.empty(R, H, W, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_eta_head(_ptr(x), _ptr(w_taps), _bias(bias, 1, "bias"), _ptr(frame), _ptr(pos), _ptr(damping),
                                       _ptr(eta), R, H, W, float(EP), float(eta_scale), _dtype_code(x, "x"), _stream(dev)), "eta_head")
    return eta


def conv1x1_c128(x, w, bias=None, relu=False):
    """act(conv1x1(x) + bias): x [N,128,H,W] channels-last 16-bit, w [Cout,128] (Cout % 192 == 0) -> [N,Cout,H,W] channels-last"""
    _cl(x, "x", 128)
    dev = _dev(x, w, bias)
    N, _, H, W = x.shape
    if w.dim() != 2 or w.shape[1] != 128 or w.dtype != x.dtype or not w.is_contiguous():
        raise PvoHipError("conv1x1_c128: w must be a contiguous [Cout,128] tensor in x's dtype")
    Cout = w.shape[0]
    y = _new_cl(N, Cout, H, W, x.dtype, dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_conv1x1_c128(_ptr(x), _ptr(w), _bias(bias, Cout, "bias"), _ptr(y), N * H * W, Cout,
                                           1 if relu else 0, _dtype_code(x, "x"), _stream(dev)), "conv1x1_c128")
    return y


# --------------------------------------------------------------------------- FactorGraph.update glue
def graph_motion(target, coords1, delta_dy, raw_mask, dtype):
    """motion features (factor_graph.py:233-237): [1,E,H,W,2] f32 x4 -> [1,E,8,H,W] in `dtype`, stored channels-last"""
    for t, n in ((target, "target"), (coords1, "coords1"), (delta_dy, "delta_dy"), (raw_mask, "raw_mask")):
        _contig(t, n); _f32(t, n)
    dev = _dev(target, coords1, delta_dy, raw_mask)
    _, E, H, W, _ = target.shape
    motn = torch.empty(E, H, W, 8, dtype=dtype, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_graph_motion(_ptr(target), _ptr(coords1), _ptr(delta_dy), _ptr(raw_mask), _ptr(motn), E, H, W,
                                           _DT[dtype], _stream(dev)), "graph_motion")
    return motn.permute(0, 3, 1, 2)[None]


def reproject_motion(poses, disps, intrinsics, ii, jj, target, delta_dy, raw_mask, dtype):
    """reproject + graph_motion in one pass (pvo_reproject_motion) -> coords [E,ht,wd,2], valid [E,ht,wd,1], motion [1,E,8,H,W]"""
    for t, n in ((poses, "poses"), (disps, "disps"), (intrinsics, "intrinsics"), (ii, "ii"), (jj, "jj"), (target, "target"),
                 (delta_dy, "delta_dy"), (raw_mask, "raw_mask")):
        _contig(t, n)
    for t, n in ((target, "target"), (delta_dy, "delta_dy"), (raw_mask, "raw_mask")):
        _f32(t, n)
    dev = _dev(poses, disps, intrinsics, ii, jj, target, delta_dy, raw_mask)
    _long(ii, "ii"); _long(jj, "jj")
    E, ht, wd = ii.shape[0], disps.shape[1], disps.shape[2]
    coords = torch.empty(E, ht, wd, 2, dtype=torch.float32, device=dev)
    valid = torch.empty(E, ht, wd, 1, dtype=torch.float32, device=dev)
    motn = torch.empty(E, ht, wd, 8, dtype=dtype, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_reproject_motion(_ptr(poses), _ptr(disps), _ptr(intrinsics), _ptr(ii), _ptr(jj), _ptr(coords), _ptr(valid),
                                               _ptr(target), _ptr(delta_dy), _ptr(raw_mask), _ptr(motn), E, ht, wd, _DT[dtype],
                                               _stream(dev)), "reproject_motion")
    return coords, valid, motn.permute(0, 3, 1, 2)[None]


def segment_hist(segm, raw_mask, heads, max_segments, dy_thresh=0.5):
    """counting half of the panoptic vote (factor_graph.py:256-261): segm int32 [E,H,W] dense labels in [0, max_segments),
    raw_mask [1,E,H,W,2] f32 (BEFORE the update), heads [E,8,H,W] channels-last -> (tot, dyn) int32 [E,max_segments]"""
    _cl(heads, "heads", 8)
    dev = _dev(segm, raw_mask, heads)
    _contig(segm, "segm"); _contig(raw_mask, "raw_mask")
    E, H, W = segm.shape
    if segm.dtype != torch.int32:
        raise PvoHipError("segment_hist: segm must be int32")
    tot = torch.empty(E, max_segments, dtype=torch.int32, device=dev)
    dyn = torch.empty(E, max_segments, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_segment_hist(_ptr(segm), _ptr(raw_mask), _ptr(heads), _ptr(tot), _ptr(dyn), E, H * W, int(max_segments),
                                           float(dy_thresh), _dtype_code(heads, "heads"), _stream(dev)), "segment_hist")
    return tot, dyn


def graph_post(coords1, heads, raw_mask, target_ba, weight_ba, dy_thresh=0.5, vote=None):
    """factor_graph.py:249-306 after the update operator.  heads [E,8,H,W] channels-last 16-bit (delta | delta_dy |
    weight | delta_mask); raw_mask [1,E,H,W,2] is updated IN PLACE; target_ba / weight_ba [E,2,H,W] f32 are filled.
    vote = (segm int32 [E,H,W], tot, dyn int32 [E,S], thresh): the panoptic vote of segment_hist is applied.
    Returns (target_cam, delta_dy, weight, full_flow), each [1,E,H,W,2] f32."""
    _cl(heads, "heads", 8)
    _contig(coords1, "coords1"); _contig(raw_mask, "raw_mask"); _contig(target_ba, "target_ba"); _contig(weight_ba, "weight_ba")
    dev = _dev(coords1, heads, raw_mask, target_ba, weight_ba)
    _, E, H, W, _ = coords1.shape
    new = lambda: torch.empty(1, E, H, W, 2, dtype=torch.float32, device=dev)
    target, delta_dy, weight, full_flow = new(), new(), new(), new()
    segm = tot = dyn = None
    S, vth = 0, 0.0
    if vote is not None:
        segm, tot, dyn, vth = vote
        S = tot.shape[1]
        if segm.dtype != torch.int32 or tuple(segm.shape) != (E, H, W) or not segm.is_contiguous() or tot.dtype != torch.int32 \
                or tuple(tot.shape) != (E, S) or tuple(dyn.shape) != (E, S) or dyn.dtype != torch.int32:
            raise PvoHipError("graph_post: vote = (segm int32 [E,H,W], tot int32 [E,S], dyn int32 [E,S], thresh)")
    with torch.cuda.device(dev):
        check(_lib.load().pvo_graph_post(_ptr(coords1), _ptr(heads), _ptr(raw_mask), _ptr(target), _ptr(delta_dy), _ptr(weight),
                                         _ptr(target_ba), _ptr(weight_ba), _ptr(full_flow), E, H, W, float(dy_thresh),
                                         _ptr(segm), _ptr(tot), _ptr(dyn), int(S), float(vth),
                                         _dtype_code(heads, "heads"), _stream(dev)), "graph_post")
    return target, delta_dy, weight, full_flow


# --------------------------------------------------------------------------- the operator / the graph update as one call
class PackedWeights:
    """pvo_update_weights + the tensors it points to (kept alive here)"""

    FIELDS = [n for n, _ in _lib.UpdateWeights._fields_[2:]]

    _serial = itertools.count(1)

    def __init__(self, dtype, tensors, flags=0):
        self.dtype, self.tensors, self.flags = dtype, dict(tensors), flags
        self.serial = next(PackedWeights._serial)      # (identity that survives address reuse: FactorGraph's context token)
        st = _lib.UpdateWeights()
        st.dtype, st.flags = _DT[dtype], flags
        dev = None
        for n in self.FIELDS:
            t = self.tensors[n]
            if not t.is_cuda or not t.is_contiguous():
                raise PvoHipError("packed weight %s must be a contiguous device tensor" % n)
            want = torch.float32 if (n.endswith("_b") or n in ("gate_wt",)) else dtype
            if t.dtype != want:
                raise PvoHipError("packed weight %s must be %s" % (n, want))
            dev = t.device if dev is None else dev
            setattr(st, n, t.data_ptr())
        self.struct, self.device = st, dev

    def on_one_stream(self):
        """the same weights with the operator's side chains (aggregation branch, flow encoder) kept on the launch stream: for callers that
        CAPTURE the operator into a HIP graph - a captured fork becomes a parallel branch the runtime runs on a stream of its own choosing"""
        one = self.__dict__.get("_one_stream")
        if one is None:
            flags = (self.flags | _lib.PVO_OP_SINGLE_STREAM) & ~_lib.PVO_OP_ENC_SIDE_STREAM
            one = self if flags == self.flags else PackedWeights(self.dtype, self.tensors, flags)
            self.__dict__["_one_stream"] = one
        return one

    # a derived cache entry: copies / pickles of the owning module rebuild it (ctypes structs with pointers cannot be copied)
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (_none, ())


def _none():
    return None


def _vp(t):
    return t.data_ptr() if t is not None and t.numel() > 0 else None


def _fill_operator_args(a, E, H, W, pool_levels, slots, num_slots, coords, corr, motion, net, net_out, inp, P_zr, P_q,
                        agg, heads, eta_rows, eta, upmask):
    a.E, a.H, a.W = E, H, W
    for l in range(4):
        a.levels[l] = pool_levels[l].data_ptr() if pool_levels is not None else None
    a.slots, a.num_slots = _vp(slots), int(num_slots)
    a.coords, a.corr, a.motion = _vp(coords), _vp(corr), _vp(motion)
    a.net, a.net_out, a.inp, a.P_zr, a.P_q = _vp(net), _vp(net_out), _vp(inp), _vp(P_zr), _vp(P_q)
    a.static_by_slot = 0
    if agg is not None:
        a.seg_ptr, a.seg_idx, a.K = agg[0].data_ptr(), agg[1].data_ptr(), int(agg[2])
    else:
        a.seg_ptr, a.seg_idx, a.K = None, None, 0
    a.heads = _vp(heads)
    if eta_rows is not None:
        frame, pos, damping, EP = eta_rows[:4]
        a.eta_frame, a.eta_pos, a.R, a.damping, a.EP = frame.data_ptr(), pos.data_ptr(), int(frame.shape[0]), damping.data_ptr(), float(EP)
        a.eta_scale = float(eta_rows[4]) if len(eta_rows) > 4 else 0.2
    else:
        a.eta_frame, a.eta_pos, a.R, a.damping, a.EP, a.eta_scale = None, None, 0, None, 0.0, 0.2
    a.eta, a.upmask = _vp(eta), _vp(upmask)


_op_ws = {}


def _op_workspace(key, dev, nbytes):
    buf = _op_ws.get(key)
    if buf is None or buf.numel() < nbytes or buf.device != dev:
        buf = _op_ws[key] = torch.empty(int(nbytes) + (int(nbytes) >> 3), dtype=torch.uint8, device=dev)
    return buf


def update_operator(weights, net, motion, inp=None, P=None, pool=None, coords=None, corr=None, agg=None,
                    eta_rows=None, want_eta=True, want_upmask=True, net_out=None):
    """DynamicUpdateModule.forward (droid_net.py:256-314) on the 16-bit inference path as ONE call into libpvo_hip.

    weights: PackedWeights.  net, inp [E,128,H,W], motion [E,8,H,W], corr [E,196,H,W]: channels-last 16-bit.
    Correlation features come either from `pool` = (tiled level tensors, slots int32 [E], num_slots) + coords [E,H,W,2]
    f32 (the lookup runs fused with the first encoder layer) or from the sampled tensor `corr`.
    P = (P_zr, P_q): cached static-input terms; None -> computed from `inp` for this call.
    agg = (seg_ptr, seg_idx, K) runs GraphAgg; eta_rows = (frame, pos, damping, EP) selects pvo_eta_head's bookkeeping form.
    Returns (net_out, heads [E,8,H,W], eta f32 or None, upmask [K,576,H,W] or None)."""
    _cl(net, "net", 128); _cl(motion, "motion", 8)
    dev = _dev(net, motion, inp, corr, coords)
    if weights.dtype != net.dtype or weights.device != dev:
        raise PvoHipError("update_operator: packed weights are %s on %s, net is %s on %s" % (weights.dtype, weights.device, net.dtype, dev))
    E, _, H, W = net.shape
    levels = slots = None
    num_slots = 0
    if pool is not None:
        levels, slots, num_slots = pool
        if coords is None or tuple(coords.shape) != (E, H, W, 2) or coords.dtype != torch.float32 or not coords.is_contiguous():
            raise PvoHipError("update_operator: coords must be a contiguous f32 [E,H,W,2] tensor")
        if len(levels) != 4 or levels[0].dtype != net.dtype or slots.dtype != torch.int32 or slots.numel() != E:
            raise PvoHipError("update_operator: pool = (4 tiled level tensors in net's dtype, slots int32 [E], num_slots)")
        for l, lv in enumerate(levels):
            if tuple(lv.shape[1:]) != (H, W) + tiled_level_shape(H, W, l) or not lv.is_contiguous():
                raise PvoHipError("update_operator: tiled pyramid level %d has shape %s" % (l, tuple(lv.shape)))
    elif corr is not None:
        _cl(corr, "corr", 196)
    else:
        raise PvoHipError("update_operator: pass pool + coords, or corr")
    P_zr = P_q = None
    if P is not None:
        P_zr, P_q = P
        _cl(P_zr, "P_zr", 256); _cl(P_q, "P_q", 128)
    elif inp is None:
        raise PvoHipError("update_operator: pass inp or P")
    else:
        _cl(inp, "inp", 128)
    if net_out is None:
        net_out = _new_cl(E, 128, H, W, net.dtype, dev)
    heads = _new_cl(E, 8, H, W, net.dtype, dev)
    K = int(agg[2]) if agg is not None else 0
    eta = upmask = None
    if K > 0:
        if agg[0].dtype != torch.int32 or agg[1].dtype != torch.int32 or agg[1].numel() != E or agg[0].numel() != K + 1:
            raise PvoHipError("update_operator: agg = (seg_ptr int32 [K+1], seg_idx int32 [E], K)")
        if want_eta:
            eta = torch.empty(eta_rows[0].shape[0] if eta_rows is not None else K, H, W, dtype=torch.float32, device=dev)
        if want_upmask:
            upmask = _new_cl(K, 576, H, W, net.dtype, dev)
    a = _lib.OperatorArgs()
    _fill_operator_args(a, E, H, W, levels, slots, num_slots, coords, corr, motion, net, net_out, inp, P_zr, P_q, agg, heads,
                        eta_rows, eta, upmask)
    lib = _lib.load()
    nbytes = lib.pvo_operator_workspace_bytes(E, K, H, W)
    if torch.cuda.is_current_stream_capturing():
        # a call being captured into a HIP graph (pvo_amd/graphs.py) gets a workspace of its own, allocated from the capture's private
        # memory pool: the shared scratch below is re-allocated whenever a later eager call needs more, and a replay would then
        # write into freed memory
        ws = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=dev)
    else:
        ws = _op_workspace(("op", dev.index), dev, nbytes)
    with torch.cuda.device(dev):
        check(lib.pvo_update_operator(ctypes.byref(weights.struct), ctypes.byref(a), ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                      _stream(dev)), "update_operator")
    return net_out, heads, eta, upmask


def bias_norm_act(x, bias=None, residual=None, norm=False, eps=1e-5, relu_inner=False, relu_outer=False, out=None, split=None):
    """what sits between two convolutions of an encoder layer, as one kernel (pvo_bias_norm_act): x [N,C,H,W] contiguous 16-bit ->
    relu_outer(residual + relu_inner(instance_norm(x + bias[c]))), every step rounded to x's dtype.  `out` may be x itself.
    split: None = large planes in slices over two launches (pvo_bias_norm_act_split), False = always the one-workgroup-per-plane kernel."""
    dev = _dev(x, bias, residual)
    _contig(x, "x")
    if x.dim() != 4 or x.dtype not in (torch.float16, torch.bfloat16):
        raise PvoHipError("bias_norm_act: x must be a contiguous 16-bit [N,C,H,W] tensor")
    N, C, H, W = x.shape
    for t, n in ((bias, "bias"), (residual, "residual")):
        if t is not None:
            _contig(t, n)
            if t.dtype != x.dtype:
                raise PvoHipError("bias_norm_act: %s must have x's dtype" % n)
    if bias is not None and bias.numel() != C or residual is not None and residual.shape != x.shape:
        raise PvoHipError("bias_norm_act: bias [C], residual of x's shape")
    y = torch.empty_like(x) if out is None else out
    lib = _lib.load()
    S = lib.pvo_bias_norm_act_slices(H * W) if split is None else (lib.pvo_bias_norm_act_slices(H * W) if split else 0)
    with torch.cuda.device(dev):
        if S > 0 and N * C <= 65535:
            # large planes: S slices per plane over two launches instead of one workgroup per plane (pvo_bias_norm_act_split)
            ws = torch.empty(2 * N * C * S, dtype=torch.float32, device=dev) if norm else None
            check(lib.pvo_bias_norm_act_split(_ptr(x), _vp(bias), _vp(residual), _ptr(y), N * C, C, H * W, 1 if norm else 0, float(eps),
                                              1 if relu_inner else 0, 1 if relu_outer else 0, _dtype_code(x, "x"), _vp(ws), ws.numel() if norm else 0,
                                              _stream(dev)), "bias_norm_act_split")
        else:
            check(lib.pvo_bias_norm_act(_ptr(x), _vp(bias), _vp(residual), _ptr(y), N * C, C, H * W, 1 if norm else 0, float(eps),
                                        1 if relu_inner else 0, 1 if relu_outer else 0, _dtype_code(x, "x"), _stream(dev)), "bias_norm_act")
    return y


def proximity_select(dist, t0, t1, rad, nms, thresh, have_i, have_j):
    """the greedy proximity-edge selection (pvo_proximity_select, a HOST function of the library): dist = float32 numpy array [ni, nj] of
    frame distances (frames t0.., t1..), have_i / have_j = int64 numpy arrays of the existing edges -> (ii, jj) Python lists"""
    import numpy as np
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    ni, nj = dist.shape
    have_i = np.ascontiguousarray(have_i, dtype=np.int64); have_j = np.ascontiguousarray(have_j, dtype=np.int64)
    cap = 2 * (ni * rad + ni * nj) + 2
    out = np.empty((2, cap), dtype=np.int64)
    n = ctypes.c_int(0)
    check(_lib.load().pvo_proximity_select(dist.ctypes.data, ni, nj, int(t0), int(t1), int(rad), int(nms), float(thresh),
                                           have_i.ctypes.data if have_i.size else None, have_j.ctypes.data if have_j.size else None, int(have_i.size),
                                           out[0].ctypes.data, out[1].ctypes.data, cap, ctypes.byref(n)), "proximity_select")
    return out[0, :n.value].tolist(), out[1, :n.value].tolist()


def conv1x1_planes_supported(cin, cout):
    return cin % 32 == 0 and cout % 64 == 0


def conv1x1_planes(x, w, bias=None, stride=1):
    """Conv2d(Cin, Cout, 1, stride) on a contiguous 16-bit NCHW tensor, deterministic (pvo_conv1x1_planes): x [N,Cin,H,W], w [Cout,Cin] or
    [Cout,Cin,1,1], bias [Cout] or None, all of x's dtype -> [N,Cout,(H-1)//stride+1,(W-1)//stride+1]"""
    dev = _dev(x, w, bias)
    _contig(x, "x"); _contig(w, "w")
    if x.dim() != 4 or x.dtype not in (torch.float16, torch.bfloat16) or w.dtype != x.dtype or (bias is not None and bias.dtype != x.dtype):
        raise PvoHipError("conv1x1_planes: x contiguous 16-bit [N,Cin,H,W], w and bias of the same dtype")
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    if w.numel() != Cout * Cin or (bias is not None and (bias.numel() != Cout or not bias.is_contiguous())):
        raise PvoHipError("conv1x1_planes: w [Cout,Cin], bias [Cout]")
    stride = int(stride)
    y = torch.empty(N, Cout, (H - 1) // stride + 1, (W - 1) // stride + 1, dtype=x.dtype, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_conv1x1_planes(_ptr(x), _ptr(w), _vp(bias), _ptr(y), N, Cin, Cout, H, W, stride, _dtype_code(x, "x"), _stream(dev)), "conv1x1_planes")
    return y


_FRAME_KINDS = {torch.int32: 0, torch.uint8: 1, torch.float32: 2}


def frame_normalise(image, mean, std, dtype=torch.float16):
    """[3,H,W] BGR frame (int32 / uint8 / float32, 0..255) on the device -> [1,3,H,W] RGB `dtype`, ((v / 255) - mean) / std in fp32 then
    rounded (pvo_frame_normalise); mean / std: sequences of three Python floats"""
    dev = _dev(image)
    _contig(image, "image")
    if image.dim() != 3 or image.shape[0] != 3 or image.dtype not in _FRAME_KINDS:
        raise PvoHipError("frame_normalise: a contiguous [3,H,W] int32 / uint8 / float32 frame")
    _, H, W = image.shape
    out = torch.empty(1, 3, H, W, dtype=dtype, device=dev)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean]); s = (ctypes.c_float * 3)(*[float(v) for v in std])
    with torch.cuda.device(dev):
        check(_lib.load().pvo_frame_normalise(_ptr(image), _ptr(out), H, W, m, s, _FRAME_KINDS[image.dtype], _DT[dtype], _stream(dev)), "frame_normalise")
    return out


def debug_config(knob, value):
    """the library's test hook (include/pvo_hip.h pvo_debug_config): "ba_solver" = None | "blocked" | "wave" | "pipe" | "twin" | "dense" | "blocks",
    "heads_gather_flat" = bool, "no_riders" = bool, "post_separate" = bool.  Process-wide; tests that compare bit-identical forms call it in a process of
    their own.  (These were environment variables read inside the library until round 5.)"""
    knobs = {"ba_solver": _lib.PVO_KNOB_BA_SOLVER, "heads_gather_flat": _lib.PVO_KNOB_HEADS_GATHER_FLAT, "no_riders": _lib.PVO_KNOB_NO_RIDERS,
             "post_separate": _lib.PVO_KNOB_POST_SEPARATE}
    if knob == "ba_solver":
        value = {None: 0, "": 0, "blocked": 1, "wave": 2, "pipe": 3, "twin": 4, "dense": 5, "blocks": 6}[value]
    check(_lib.load().pvo_debug_config(knobs[knob], int(value)), "pvo_debug_config")


def graph_update_workspace(E, K, R, H, W, max_segments, device):
    n = _lib.load().pvo_graph_update_workspace_bytes(int(E), int(K), int(R), int(H), int(W), int(max_segments))
    return _op_workspace(("up", torch.device(device).index), torch.device(device), n)


def graph_update(weights, args, workspace):
    """FactorGraph.update (factor_graph.py:227-307) as ONE call: `args` is a filled _lib.GraphUpdateArgs (the caller -
    pvo_amd.factor_graph - keeps every tensor it points to alive); see include/pvo_hip.h pvo_graph_update."""
    dev = workspace.device
    with torch.cuda.device(dev):
        check(_lib.load().pvo_graph_update(ctypes.byref(weights.struct), ctypes.byref(args), ctypes.c_void_p(workspace.data_ptr()),
                                           workspace.numel(), _stream(dev)), "graph_update")


# --------------------------------------------------------------------------- SE3 element-wise operations
SE3_OPS = {"exp": 0, "log": 1, "inv": 2, "mul": 3, "act4": 4, "act3": 5, "adj": 6, "adjT": 7}
_SE3_OUT = {"exp": 7, "log": 6, "inv": 7, "mul": 7, "act4": 4, "act3": 3, "adj": 6, "adjT": 6}


def se3_unary(op, x):
    """exp [n,6] -> [n,7], log [n,7] -> [n,6], inv [n,7] -> [n,7] (lietorch SE3, forward); x contiguous f32 / f64"""
    dev = _dev(x)
    _contig(x, "x")
    if x.dtype not in (torch.float32, torch.float64):
        raise PvoHipError("se3: float32 / float64 only")
    n = x.numel() // x.shape[-1]
    y = torch.empty(x.shape[:-1] + (_SE3_OUT[op],), dtype=x.dtype, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_se3_unary(SE3_OPS[op], _ptr(x), _ptr(y), n, _DT[x.dtype], _stream(dev)), "se3_" + op)
    return y


def se3_binary(op, a, rep_a, b, rep_b, out_batch):
    """mul / act4 / act3 / adj / adjT with index broadcasting: output element i reads a[i // rep_a], b[i // rep_b]"""
    dev = _dev(a, b)
    _contig(a, "a"); _contig(b, "b")
    if a.dtype != b.dtype or a.dtype not in (torch.float32, torch.float64):
        raise PvoHipError("se3: both operands float32 or both float64")
    y = torch.empty(tuple(out_batch) + (_SE3_OUT[op],), dtype=a.dtype, device=dev)
    n = y.numel() // _SE3_OUT[op]
    with torch.cuda.device(dev):
        check(_lib.load().pvo_se3_binary(SE3_OPS[op], _ptr(a), int(rep_a), _ptr(b), int(rep_b), _ptr(y), n, _DT[a.dtype],
                                         _stream(dev)), "se3_" + op)
    return y


def se3_unary_vjp(op, x, gy):
    """gx of se3_unary(op, x) for the output gradient gy (ambient coordinates)"""
    dev = _dev(x, gy)
    _contig(x, "x"); _contig(gy, "gy")
    n = x.numel() // x.shape[-1]
    gx = torch.empty_like(x)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_se3_unary_vjp(SE3_OPS[op], _ptr(x), _ptr(gy), _ptr(gx), n, _DT[x.dtype], _stream(dev)), "se3_%s_vjp" % op)
    return gx


def se3_binary_vjp(op, a, rep_a, b, rep_b, gy, need_a=True, need_b=True):
    """(ga, gb) of se3_binary for the output gradient gy, summed over the repeats of a broadcast operand"""
    dev = _dev(a, b, gy)
    _contig(a, "a"); _contig(b, "b"); _contig(gy, "gy")
    nb = _SE3_OUT[op]
    n = gy.numel() // nb
    ga = torch.empty((n, 7), dtype=a.dtype, device=dev) if need_a else None
    gb = torch.empty((n, nb), dtype=a.dtype, device=dev) if need_b else None
    with torch.cuda.device(dev):
        check(_lib.load().pvo_se3_binary_vjp(SE3_OPS[op], _ptr(a), int(rep_a), _ptr(b), int(rep_b), _ptr(gy), _vp(ga), _vp(gb), n,
                                             _DT[a.dtype], _stream(dev)), "se3_%s_vjp" % op)
    if ga is not None:
        ga = (ga.view(-1, int(rep_a), 7).sum(1) if rep_a > 1 else ga).view(a.shape)
    if gb is not None:
        gb = (gb.view(-1, int(rep_b), nb).sum(1) if rep_b > 1 else gb).view(b.shape)
    return ga, gb


def proj_transform(poses, depths, intrinsics, ii, jj, jacobian=False, return_depth=False):
    """projective_ops.projective_transform as ONE kernel (pvo_proj_transform): poses [B,P,7], depths [B,P,H,W], intrinsics [B,P,4]
    (fp32 or fp64, one device), ii / jj int64 [N] on that device -> coords [B,N,H,W,2|3], valid [B,N,H,W,1] and, with
    `jacobian`, (Ji [B,N,H,W,2,6], Jj [B,N,H,W,2,6], Jz [B,N,H,W,2,1])"""
    dev = _dev(poses, depths, intrinsics, ii, jj)
    for t, n in ((poses, "poses"), (depths, "depths"), (intrinsics, "intrinsics"), (ii, "ii"), (jj, "jj")):
        _contig(t, n)
    _long(ii, "ii"); _long(jj, "jj")
    B, P, ht, wd = depths.shape
    N, nx = ii.shape[0], 3 if return_depth else 2
    if poses.shape != (B, P, 7) or intrinsics.shape != (B, P, 4) or poses.dtype != depths.dtype or intrinsics.dtype != depths.dtype:
        raise PvoHipError("proj_transform: poses [B,P,7], depths [B,P,H,W], intrinsics [B,P,4] of one dtype")
    new = lambda *sh: torch.empty(sh, dtype=depths.dtype, device=dev)
    x1, valid = new(B, N, ht, wd, nx), new(B, N, ht, wd, 1)
    Ji, Jj, Jz = (new(B, N, ht, wd, 2, 6), new(B, N, ht, wd, 2, 6), new(B, N, ht, wd, 2, 1)) if jacobian else (None, None, None)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_proj_transform(_ptr(poses), _ptr(depths), _ptr(intrinsics), _ptr(ii), _ptr(jj), B, P, N, ht, wd, nx,
                                             _ptr(x1), _ptr(valid), _vp(Ji), _vp(Jj), _vp(Jz), _DT[depths.dtype], _stream(dev)), "proj_transform")
    return (x1, valid, (Ji, Jj, Jz)) if jacobian else (x1, valid)


def proj_transform_vjp(poses, depths, intrinsics, ii, jj, g_x1, g_Ji, g_Jj, g_Jz):
    """gradients of proj_transform's outputs (any may be None) -> (grad poses [B,P,7], grad depths [B,P,H,W]), ambient coordinates"""
    dev = _dev(poses, depths, intrinsics, ii, jj)
    B, P, ht, wd = depths.shape
    gs = [None if g is None else g.contiguous() for g in (g_x1, g_Ji, g_Jj, g_Jz)]
    nx = gs[0].shape[-1] if gs[0] is not None else 2
    gp, gd = torch.zeros_like(poses), torch.zeros_like(depths)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_proj_transform_vjp(_ptr(poses), _ptr(depths), _ptr(intrinsics), _ptr(ii), _ptr(jj), B, P, ii.shape[0], ht, wd, nx,
                                                 _vp(gs[0]), _vp(gs[1]), _vp(gs[2]), _vp(gs[3]), _ptr(gp), _ptr(gd), _DT[depths.dtype], _stream(dev)),
              "proj_transform_vjp")
    return gp, gd


STAGES = {"lookup": 0, "gates": 1, "candidate": 2, "ba": 3, "update": 4, "empty": 5}


_side_streams = {}


def side_stream(device):
    """the library's own second stream on `device` (pvo_side_stream) as a torch stream: work that is only needed by the
    next update can be queued beside the launch stream without creating another HIP stream"""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _side_streams.get(idx)
    if st is None:
        out = ctypes.c_void_p()
        with torch.cuda.device(idx):
            check(_lib.load().pvo_side_stream(ctypes.byref(out)), "side_stream")
        st = _side_streams[idx] = torch.cuda.ExternalStream(out.value, device=torch.device("cuda", idx))
    return st


_upload_streams = {}


def upload_stream(device):
    """ONE stream per device for asynchronous host-to-device frame uploads (the pipelined tracker), created once per process behind the
    library's second stream.  The runtime hands a new stream the hardware queue with the fewest streams on it at that moment: a stream
    created per tracker landed on a different queue in every tracker of a process - the first pipelined pass of bench.py's sequence leg
    ran at 147 frames/s, the next two at 103 and 106 with their upload stream on a queue the operator's kernels were using."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _upload_streams.get(idx)
    if st is None:
        side_stream(device)
        st = _upload_streams[idx] = torch.cuda.Stream(torch.device("cuda", idx))
    return st


def clock_probe(stream, iters=20000):
    """launch the shader-clock probe on `stream` (a torch.cuda.Stream); returns the device tensor it fills - after a
    synchronize, `clock_ghz(t)` is the clock the chip held while the probe ran"""
    dev = torch.device("cuda", stream.device_index if hasattr(stream, "device_index") else torch.cuda.current_device())
    with torch.cuda.device(dev), torch.cuda.stream(stream):
        # (allocated ON `stream`: a block the caching allocator recycles for the current stream may still be in use by
        # kernels queued there, which is only safe for work ordered behind them)
        out = torch.zeros(3, dtype=torch.int64, device=dev)
        check(_lib.load_probe().pvo_clock_probe(_ptr(out), int(iters), ctypes.c_void_p(stream.cuda_stream)), "clock_probe")
    return out


def mem_probe_gbps(device, nbytes=1 << 30, iters=16, blocks=8192, reps=5):
    """GB/s of fetched lines for consecutive 128-byte lines, random 128-byte lines and random 64-byte half lines
    (pvo_mem_probe; HIP events on the current stream): what this part's memory system sustains for scattered lines"""
    dev = torch.device(device)
    buf = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 255)
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    lib, out = _lib.load_probe(), {}
    with torch.cuda.device(dev):
        for name, mode in (("streaming_128B", 0), ("random_128B", 1), ("random_64B", 2)):
            run = lambda: lib.pvo_mem_probe(_ptr(buf), nbytes, mode, iters, blocks, _ptr(sink), _stream(dev))
            for _ in range(2):
                fetched = run()
            if fetched < 0:
                raise PvoHipError("mem_probe failed")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record(); torch.cuda.synchronize(dev)
            out[name] = fetched / (e0.elapsed_time(e1) / reps) / 1e6
    return out


def clock_ghz(t):
    c, r, _ = t.tolist()
    return c / (r * 10.0) if r else float("nan")


def probe_arm(stage, capacity, every=1):
    """record HIP events around `stage` ("lookup" | "gates" | "candidate" | "ba" | "update") of the next native updates,
    one occurrence in `every`"""
    check(_lib.load().pvo_probe_arm_every(STAGES[stage], int(capacity), int(every)), "probe_arm")


def probe_read(capacity):
    """elapsed milliseconds of the recorded stage occurrences (waits for them); disarms the probe"""
    buf = (ctypes.c_float * int(capacity))()
    n = _lib.load().pvo_probe_read(buf, int(capacity))
    if n < 0:
        raise PvoHipError("probe_read failed")
    return [float(buf[i]) for i in range(n)]
