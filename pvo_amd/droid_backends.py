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


def corr_pyramid_lookup(pyramid, coords, radius):
    """All levels of CorrBlock.__call__ (modules/corr.py:40-50) in one launch.

    pyramid: list of level tensors [N,h1,w1,h2>>l,w2>>l]; coords [N,h1,w1,2] f32
    -> [N, L*(2r+1)^2, h1, w1].  Not part of the reference surface: it is what the
    Python loop + torch.cat computes, fused."""
    dev = _dev(coords, *pyramid)
    _contig(coords, "coords"); _f32(coords, "coords")
    for lv in pyramid:
        _contig(lv, "volume")
    N, h1, w1, h2, w2 = pyramid[0].shape
    L = len(pyramid)
    for l, lv in enumerate(pyramid):
        if tuple(lv.shape) != (N, h1, w1, h2 >> l, w2 >> l) or lv.dtype != pyramid[0].dtype:
            raise PvoHipError("pyramid level %d has shape %s, expected %s"
                              % (l, tuple(lv.shape), (N, h1, w1, h2 >> l, w2 >> l)))
    rd = 2 * radius + 1
    out = torch.empty((N, L * rd * rd, h1, w1), dtype=pyramid[0].dtype, device=dev)
    ptrs = (ctypes.c_void_p * L)(*[lv.data_ptr() if lv.numel() else 0 for lv in pyramid])
    lib = _lib.load()
    with torch.cuda.device(dev):
        check(lib.pvo_corr_pyramid_lookup(ptrs, _ptr(coords), _ptr(out), N, h1, w1, h2, w2, L, radius,
                                          _dtype_code(pyramid[0], "volume"), _stream(dev)),
              "corr_pyramid_lookup")
    return out
