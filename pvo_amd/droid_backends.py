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


def to_device_async(values, dtype, device):
    """host sequence -> device tensor through a pinned staging buffer and an asynchronous copy: unlike
    torch.tensor(values, device=...) this does not drain the stream"""
    t = torch.tensor(values, dtype=dtype)
    if torch.device(device).type != "cuda":
        return t
    return t.pin_memory().to(device, non_blocking=True)


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
    return dtype in (torch.float16, torch.bfloat16) and wd % 64 == 0 and ht % 8 == 0


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


def reproject(poses, disps, intrinsics, ii, jj):
    """DepthVideo.reproject (depth_video.py:154-163): poses [F,7], disps [F,ht,wd],
    intrinsics [F,4] -> coords [E,ht,wd,2], valid [E,ht,wd,1]."""
    for t, n in ((poses, "poses"), (disps, "disps"), (intrinsics, "intrinsics"), (ii, "ii"), (jj, "jj")):
        _contig(t, n)
    dev = _dev(poses, disps, intrinsics, ii, jj)
    _long(ii, "ii"); _long(jj, "jj")
    E, ht, wd = ii.shape[0], disps.shape[1], disps.shape[2]
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


def ba_plan(ii, jj, nframes, HW, K_eta, t0, t1, workspace):
    dev = _dev(ii, jj, workspace)
    _long(ii, "ii"); _long(jj, "jj")
    with torch.cuda.device(dev):
        check(_lib.load().pvo_ba_plan(_ptr(ii), _ptr(jj), ii.shape[0], int(nframes), int(HW), int(K_eta), int(t0), int(t1),
                                      ctypes.c_void_p(workspace.data_ptr()), workspace.numel(), _stream(dev)), "ba_plan")


def ba_local(poses, disps, intrinsics, targets, weights, eta, ii, jj, t0, t1, motion_only, sys, workspace):
    """assemble + eliminate this rank's edges; `sys` (fp64 [(6P)^2+6P]) receives the local reduced system"""
    dev = _dev(poses, disps, intrinsics, targets, weights, eta, ii, jj, sys, workspace)
    F, ht, wd = disps.shape
    if sys.dtype != torch.float64 or sys.numel() < (6 * (t1 - t0)) ** 2 + 6 * (t1 - t0):
        raise PvoHipError("ba_local: sys must be float64 with (6P)^2 + 6P elements")
    K_eta = 1
    if eta is not None:
        eta = eta.contiguous().view(-1, ht * wd)
        K_eta = eta.shape[0]
    with torch.cuda.device(dev):
        check(_lib.load().pvo_ba_local(_ptr(poses), _ptr(disps), _ptr(intrinsics), _ptr(targets), _ptr(weights),
                                       _ptr(eta) if eta is not None else ctypes.c_void_p(0), _ptr(ii), _ptr(jj),
                                       ii.shape[0], F, ht, wd, K_eta, int(t0), int(t1), 1 if motion_only else 0,
                                       _ptr(sys), ctypes.c_void_p(workspace.data_ptr()), workspace.numel(),
                                       _stream(dev)), "ba_local")


def ba_finish(poses, disps, sys, ii, jj, t0, t1, lm, ep, motion_only, workspace, dz_rows=0, status=None, outputs=True):
    """damp + solve the (all-reduced) system, retract poses, back-substitute this rank's depths -> [dx, dz]
    (outputs=False: poses / disps are updated in place and no dx / dz tensors are produced -> [None, None])"""
    dev = _dev(poses, disps, sys, ii, jj, workspace)
    F, ht, wd = disps.shape
    P = int(t1) - int(t0)
    dx = torch.zeros(max(P, 0), 6, dtype=torch.float32, device=dev) if outputs else None
    dz = torch.zeros(int(dz_rows), ht * wd, dtype=torch.float32, device=dev) if outputs else None
    if not outputs:
        dz_rows = 0
    with torch.cuda.device(dev):
        check(_lib.load().pvo_ba_finish(_ptr(poses), _ptr(disps), _ptr(sys), _ptr(ii), _ptr(jj), ii.shape[0], F, ht, wd,
                                        int(t0), int(t1), float(lm), float(ep), 1 if motion_only else 0,
                                        _ptr(dx), _ptr(dz), int(dz_rows),
                                        _ptr(status) if status is not None else ctypes.c_void_p(0),
                                        ctypes.c_void_p(workspace.data_ptr()), workspace.numel(), _stream(dev)),
              "ba_finish")
    return [dx, dz]


# --------------------------------------------------------------------------- fused ConvGRU element-wise ops
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


def gru_glo(wn, net, w_bias=None):
    """mean over pixels of sigmoid(wn + w_bias)*net (gru.py:23-24) -> [E,128] float32"""
    _cl(wn, "wn", 128); _cl(net, "net", 128)
    dev = _dev(wn, net)
    E, C, H, W = net.shape
    glo = torch.empty(E, C, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_gru_glo(_ptr(wn), _ptr(net), _bias(w_bias, 128, "w_bias"), _ptr(glo), E, H * W, C, _dtype_code(net, "net"), _stream(dev)), "gru_glo")
    return glo


def gru_assemble(net, inp, corr_feat, flow_feat, X, corr_bias=None, flow_bias=None):
    """X [E,448,H,W] (channels-last) <- [net | inp | relu(corr_feat + corr_bias) | relu(flow_feat + flow_bias)];
    with inp=None, X is [E,320,H,W] without the inp block (its convolution is precomputed, see gru_gate)."""
    _cl(net, "net", 128); _cl(corr_feat, "corr_feat", 128); _cl(flow_feat, "flow_feat", 64)
    if inp is not None:
        _cl(inp, "inp", 128)
    _cl(X, "X", 448 if inp is not None else 320)
    dev = _dev(net, inp, corr_feat, flow_feat, X)
    E, _, H, W = net.shape
    with torch.cuda.device(dev):
        check(_lib.load().pvo_gru_assemble(_ptr(net), _ptr(inp), _ptr(corr_feat), _ptr(flow_feat),
                                           _bias(corr_bias, 128, "corr_bias"), _bias(flow_bias, 64, "flow_bias"), _ptr(X), E * H * W,
                                           1 if inp is not None else 0,
                                           _dtype_code(net, "net"), _stream(dev)), "gru_assemble")


def gru_gate(zr, g, net, Z, X, P_zr=None):
    """Z <- sigmoid(zr[:, :128] + P_zr[:, :128] + g_z);  X[:, :128] <- sigmoid(zr[:, 128:] + P_zr[:, 128:] + g_r) * net
    (gru.py:26-28); X has 448 or 320 channels"""
    _cl(zr, "zr", 256); _cl(net, "net", 128); _cl(Z, "Z", 128); _cl(X, "X", X.shape[1])
    if P_zr is not None:
        _cl(P_zr, "P_zr", 256)
    dev = _dev(zr, g, net, Z, X, P_zr)
    _f32(g, "g"); _contig(g, "g")
    E, _, H, W = net.shape
    with torch.cuda.device(dev):
        check(_lib.load().pvo_gru_gate(_ptr(zr), _ptr(g), _ptr(net), _ptr(Z), _ptr(X), _ptr(P_zr), X.shape[1], E, H * W,
                                       _dtype_code(net, "net"), _stream(dev)), "gru_gate")


def gru_out(q, g, Z, net, P_q=None):
    """(1-Z)*net + Z*tanh(q + P_q + g_q)   (gru.py:28-31) -> new hidden state, channels-last [E,128,H,W]"""
    _cl(q, "q", 128); _cl(Z, "Z", 128); _cl(net, "net", 128)
    if P_q is not None:
        _cl(P_q, "P_q", 128)
    dev = _dev(q, g, Z, net, P_q)
    _f32(g, "g"); _contig(g, "g")
    E, _, H, W = net.shape
    out = torch.empty_like(net, memory_format=torch.channels_last)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_gru_out(_ptr(q), _ptr(g), _ptr(Z), _ptr(net), _ptr(out), _ptr(P_q), E, H * W,
                                      _dtype_code(net, "net"), _stream(dev)), "gru_out")
    return out


def bias_act_(x, bias, relu=True):
    """in place x <- act(x + bias[c]) on a channels-last 16-bit [N,C,H,W] tensor; returns x"""
    if x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last) or x.dtype not in (torch.float16, torch.bfloat16):
        raise PvoHipError("bias_act_: x must be a channels-last 16-bit [N,C,H,W] tensor")
    dev = _dev(x)
    N, C, H, W = x.shape
    with torch.cuda.device(dev):
        check(_lib.load().pvo_bias_act(_ptr(x), _bias(bias, C, "bias"), N * H * W, C, 1 if relu else 0,
                                       _dtype_code(x, "x"), _stream(dev)), "bias_act")
    return x


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
    y = torch.empty(E, H, W, 128, dtype=x.dtype, device=dev).permute(0, 3, 1, 2)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_conv7x7_c8(_ptr(x), _ptr(w_taps), _bias(bias, 128, "bias"), _ptr(y), E, H, W,
                                         _dtype_code(x, "x"), _stream(dev)), "conv7x7_c8")
    return y


def conv3x3_weights(weight, dtype):
    """[Cout,Cin,3,3] conv filter -> the [9,Cout,Cin] tap-major layout pvo_conv3x3 reads"""
    co, ci, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or ci % 32 or co % 128:
        raise PvoHipError("conv3x3: filter must be [Cout,Cin,3,3] with Cin % 32 == 0 and Cout % 128 == 0")
    return weight.detach().permute(2, 3, 0, 1).reshape(9, co, ci).to(dtype).contiguous()


def conv3x3(x, w_taps, bias=None, relu=False):
    """act(conv3x3(x) + bias): x [E,Cin,H,W] channels-last 16-bit -> [E,Cout,H,W] channels-last (wide layers)"""
    dev = _dev(x, w_taps)
    E, Cin, H, W = x.shape
    _cl(x, "x", Cin)
    if w_taps.dim() != 3 or w_taps.shape[0] != 9 or w_taps.shape[2] != Cin or w_taps.dtype != x.dtype or not w_taps.is_contiguous():
        raise PvoHipError("conv3x3: w_taps must be the [9,Cout,Cin] tensor of conv3x3_weights in x's dtype")
    Cout = w_taps.shape[1]
    y = torch.empty(E, H, W, Cout, dtype=x.dtype, device=dev).permute(0, 3, 1, 2)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_conv3x3(_ptr(x), _ptr(w_taps), _bias(bias, Cout, "bias"), _ptr(y), E, H, W, Cin, Cout,
                                      1 if relu else 0, _dtype_code(x, "x"), _stream(dev)), "conv3x3")
    return y


def gru_conv_gates(X, w_taps, g, P_zr, net):
    """gate convolution + sigmoid gates in one kernel -> (Z, RN), each [E,128,H,W] channels-last.
    X [E,Cin,H,W] = [net | corr | flow], w_taps [9,256,Cin], g [E,384] f32, P_zr [E,256,H,W], net [E,128,H,W]."""
    E, Cin, H, W = X.shape
    _cl(X, "X", Cin); _cl(P_zr, "P_zr", 256); _cl(net, "net", 128)
    dev = _dev(X, w_taps, g, P_zr, net)
    _f32(g, "g"); _contig(g, "g")
    if tuple(w_taps.shape) != (9, 256, Cin) or w_taps.dtype != X.dtype or not w_taps.is_contiguous():
        raise PvoHipError("gru_conv_gates: w_taps must be [9,256,Cin] in X's dtype")
    Z = torch.empty(E, H, W, 128, dtype=X.dtype, device=dev).permute(0, 3, 1, 2)
    RN = torch.empty(E, H, W, 128, dtype=X.dtype, device=dev).permute(0, 3, 1, 2)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_gru_conv_gates(_ptr(X), _ptr(w_taps), _ptr(g), _ptr(P_zr), _ptr(net), _ptr(Z), _ptr(RN),
                                             E, H, W, Cin, _dtype_code(X, "X"), _stream(dev)), "gru_conv_gates")
    return Z, RN


def gru_conv_candidate(X, RN, w_taps, g, P_q, Z, net):
    """candidate convolution over [RN | X[:, 128:]] + the GRU state update in one kernel -> new hidden state"""
    E, Cin, H, W = X.shape
    _cl(X, "X", Cin); _cl(RN, "RN", 128); _cl(P_q, "P_q", 128); _cl(Z, "Z", 128); _cl(net, "net", 128)
    dev = _dev(X, RN, w_taps, g, P_q, Z, net)
    _f32(g, "g"); _contig(g, "g")
    if tuple(w_taps.shape) != (9, 128, Cin) or w_taps.dtype != X.dtype or not w_taps.is_contiguous():
        raise PvoHipError("gru_conv_candidate: w_taps must be [9,128,Cin] in X's dtype")
    out = torch.empty(E, H, W, 128, dtype=X.dtype, device=dev).permute(0, 3, 1, 2)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_gru_conv_candidate(_ptr(X), _ptr(RN), _ptr(w_taps), _ptr(g), _ptr(P_q), _ptr(Z), _ptr(net),
                                                 _ptr(out), E, H, W, Cin, _dtype_code(X, "X"), _stream(dev)), "gru_conv_candidate")
    return out


def gru_gates(net, cf, ff, cf_bias, ff_bias, w_taps, g, P_zr):
    """gate convolution over [net | relu(cf + b) | relu(ff + b)] read from the three tensors + sigmoid gates -> (Z, RN)"""
    _cl(net, "net", 128); _cl(cf, "cf", 128); _cl(ff, "ff", 64); _cl(P_zr, "P_zr", 256)
    dev = _dev(net, cf, ff, w_taps, g, P_zr)
    _f32(g, "g"); _contig(g, "g")
    E, _, H, W = net.shape
    if tuple(w_taps.shape) != (9, 256, 320) or w_taps.dtype != net.dtype or not w_taps.is_contiguous():
        raise PvoHipError("gru_gates: w_taps must be [9,256,320] in net's dtype")
    Z = torch.empty(E, H, W, 128, dtype=net.dtype, device=dev).permute(0, 3, 1, 2)
    RN = torch.empty(E, H, W, 128, dtype=net.dtype, device=dev).permute(0, 3, 1, 2)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_gru_gates(_ptr(net), _ptr(cf), _ptr(ff), _bias(cf_bias, 128, "cf_bias"), _bias(ff_bias, 64, "ff_bias"),
                                        _ptr(w_taps), _ptr(g), _ptr(P_zr), _ptr(Z), _ptr(RN), E, H, W,
                                        _dtype_code(net, "net"), _stream(dev)), "gru_gates")
    return Z, RN


def gru_candidate(RN, cf, ff, cf_bias, ff_bias, w_taps, g, P_q, Z, net):
    """candidate convolution over [RN | relu(cf + b) | relu(ff + b)] + the GRU state update -> new hidden state"""
    _cl(RN, "RN", 128); _cl(cf, "cf", 128); _cl(ff, "ff", 64); _cl(P_q, "P_q", 128); _cl(Z, "Z", 128); _cl(net, "net", 128)
    dev = _dev(RN, cf, ff, w_taps, g, P_q, Z, net)
    _f32(g, "g"); _contig(g, "g")
    E, _, H, W = net.shape
    if tuple(w_taps.shape) != (9, 128, 320) or w_taps.dtype != net.dtype or not w_taps.is_contiguous():
        raise PvoHipError("gru_candidate: w_taps must be [9,128,320] in net's dtype")
    out = torch.empty(E, H, W, 128, dtype=net.dtype, device=dev).permute(0, 3, 1, 2)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_gru_candidate(_ptr(RN), _ptr(cf), _ptr(ff), _bias(cf_bias, 128, "cf_bias"), _bias(ff_bias, 64, "ff_bias"),
                                            _ptr(w_taps), _ptr(g), _ptr(P_q), _ptr(Z), _ptr(net), _ptr(out), E, H, W,
                                            _dtype_code(net, "net"), _stream(dev)), "gru_candidate")
    return out


def conv3x3_c128_weights(weight, dtype):
    """[Cout,128,3,3] conv filter -> the [9,Cout,128] tap-major layout pvo_conv3x3_c128 reads"""
    co, ci, kh, kw = weight.shape
    if (ci, kh, kw) != (128, 3, 3) or (co != 64 and co % 128):
        raise PvoHipError("conv3x3_c128: filter must be [Cout,128,3,3] with Cout in {64,128,256,512}")
    return weight.detach().permute(2, 3, 0, 1).reshape(9, co, 128).to(dtype).contiguous()


def conv3x3_c128(x, w_taps, bias=None, relu=False):
    """act(conv3x3(x) + bias): x [E,128,H,W] channels-last 16-bit -> [E,Cout,H,W] channels-last"""
    _cl(x, "x", 128)
    dev = _dev(x, w_taps)
    E, _, H, W = x.shape
    if w_taps.dim() != 3 or w_taps.shape[0] != 9 or w_taps.shape[2] != 128 or w_taps.dtype != x.dtype or not w_taps.is_contiguous():
        raise PvoHipError("conv3x3_c128: w_taps must be the [9,Cout,128] tensor of conv3x3_c128_weights in x's dtype")
    Cout = w_taps.shape[1]
    y = torch.empty(E, H, W, Cout, dtype=x.dtype, device=dev).permute(0, 3, 1, 2)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_conv3x3_c128(_ptr(x), _ptr(w_taps), _bias(bias, Cout, "bias"), _ptr(y), E, H, W, Cout,
                                           1 if relu else 0, _dtype_code(x, "x"), _stream(dev)), "conv3x3_c128")
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
    out = torch.empty(K, H, W, C, dtype=x.dtype, device=dev).permute(0, 3, 1, 2)
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
    y = torch.empty(E, H, W, 8, dtype=h1.dtype, device=dev).permute(0, 3, 1, 2)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_heads_out(_ptr(h1), _bias(bias1, 512, "bias1"), _ptr(w2), _bias(bias2, 8, "bias2"), _ptr(y),
                                        E, H, W, _dtype_code(h1, "h1"), _stream(dev)), "heads_out")
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


def eta_finish(raw, bias, frame, pos, damping, EP):
    """eta head + damping bookkeeping: raw [K,1,H,W] 16-bit (bias-free eta convolution), bias f32 [1], frame int64 [R],
    pos int32 [R] (row of raw, or -1), damping f32 [buffer,H,W] (updated in place) -> eta f32 [R,H,W] for the BA"""
    dev = _dev(raw, bias, frame, pos, damping)
    K, _, H, W = raw.shape
    R = frame.shape[0]
    if raw.dtype not in (torch.float16, torch.bfloat16) or frame.dtype != torch.int64 or pos.dtype != torch.int32 \
            or pos.shape[0] != R or damping.dtype != torch.float32 or tuple(damping.shape[1:]) != (H, W):
        raise PvoHipError("eta_finish: raw 16-bit [K,1,H,W], frame int64 [R], pos int32 [R], damping f32 [*,H,W]")
    _contig(damping, "damping")
    eta = torch.empty(R, H, W, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().pvo_eta_finish(_ptr(raw.contiguous()), _bias(bias, 1, "bias"), _ptr(frame), _ptr(pos), _ptr(damping),
                                         _ptr(eta), R, H * W, float(EP), _dtype_code(raw, "raw"), _stream(dev)), "eta_finish")
    return eta


def graph_post(coords1, heads, raw_mask, target_ba, weight_ba, dy_thresh=0.5, force_dyn=None):
    """factor_graph.py:249-306 after the update operator.  heads [E,8,H,W] channels-last 16-bit (delta | delta_dy |
    weight | delta_mask); raw_mask [1,E,H,W,2] is updated IN PLACE; target_ba / weight_ba [E,2,H,W] f32 are filled.
    Returns (target_cam, delta_dy, weight, full_flow), each [1,E,H,W,2] f32."""
    _cl(heads, "heads", 8)
    _contig(coords1, "coords1"); _contig(raw_mask, "raw_mask"); _contig(target_ba, "target_ba"); _contig(weight_ba, "weight_ba")
    dev = _dev(coords1, heads, raw_mask, target_ba, weight_ba)
    _, E, H, W, _ = coords1.shape
    new = lambda: torch.empty(1, E, H, W, 2, dtype=torch.float32, device=dev)
    target, delta_dy, weight, full_flow = new(), new(), new(), new()
    with torch.cuda.device(dev):
        if force_dyn is not None and (force_dyn.dtype != torch.uint8 or tuple(force_dyn.shape) != (E, H, W) or not force_dyn.is_contiguous()):
            raise PvoHipError("graph_post: force_dyn must be a contiguous uint8 [E,H,W] tensor")
        check(_lib.load().pvo_graph_post(_ptr(coords1), _ptr(heads), _ptr(raw_mask), _ptr(target), _ptr(delta_dy), _ptr(weight),
                                         _ptr(target_ba), _ptr(weight_ba), _ptr(full_flow), E, H, W, float(dy_thresh), _ptr(force_dyn),
                                         _dtype_code(heads, "heads"), _stream(dev)), "graph_post")
    return target, delta_dy, weight, full_flow
