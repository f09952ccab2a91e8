"""ctypes binding of libpvo_hip.so (the C ABI declared in include/pvo_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  If it is
missing, or a call returns a non-zero status, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpvo_hip.so")

PVO_F32, PVO_F16, PVO_BF16, PVO_F64 = 0, 1, 2, 3

_c = ctypes
_vp, _i, _f, _sz = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t

# name -> (restype, argtypes); mirrors include/pvo_hip.h one to one
SIGNATURES = {
    "pvo_strerror": (_c.c_char_p, [_i]),
    "pvo_version": (_i, []),
    "pvo_corr_index_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pvo_corr_index_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pvo_corr_pyramid_lookup": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "pvo_altcorr_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pvo_altcorr_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pvo_corr_build": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "pvo_eta_finish": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, ctypes.c_float, _i, _vp]),
    "pvo_gru_glo_chunks": (_i, [_i]),
    "pvo_gru_glo_fused": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "pvo_gru_gates": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_gru_candidate": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_gru_conv_gates": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "pvo_gru_conv_candidate": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "pvo_conv3x3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pvo_conv3x3_c128": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "pvo_conv7x7_c8": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_corr_build_tiled": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "pvo_corr_lookup_encode_tiled": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "pvo_corr_pyramid_lookup_tiled": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "pvo_gru_glo": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_gru_assemble": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _c.c_longlong, _i, _i, _vp]),
    "pvo_heads_out": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_bias_act": (_i, [_vp, _vp, _c.c_longlong, _i, _i, _i, _vp]),
    "pvo_segment_mean": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_gru_gate": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_gru_out": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "pvo_graph_motion": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_graph_post": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _i, _vp]),
    "pvo_frame_distance": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "pvo_projmap": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "pvo_iproj": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "pvo_depth_filter": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_reproject": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "pvo_ba_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "pvo_ba": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i,
                    _f, _f, _i, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "pvo_ba_plan": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "pvo_ba_local": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i,
                          _i, _vp, _vp, _sz, _vp]),
    "pvo_ba_finish": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _i,
                           _vp, _vp, _i, _vp, _vp, _sz, _vp]),
}

_lib = None


class PvoHipError(RuntimeError):
    pass


def load():
    """Load libpvo_hip.so, binding every symbol of the header. Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PvoHipError(
            "libpvo_hip.so not found at %s - build it with `python -m pvo_amd.build` "
            "(there is no CPU fallback)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().pvo_strerror(status).decode()
        raise PvoHipError("%s failed: %s (status %d)" % (what, msg, status))
