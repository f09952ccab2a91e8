"""ctypes binding of libpvo_hip.so (the C ABI declared in include/pvo_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  If it is
missing, or a call returns a non-zero status, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpvo_hip.so")

PVO_F32, PVO_F16, PVO_BF16, PVO_F64 = 0, 1, 2, 3
PVO_ABI_VERSION = 103          # include/pvo_hip.h

_c = ctypes
_vp, _i, _f, _sz = _c.c_void_p, _c.c_int, _c.c_float, _c.c_size_t

# name -> (restype, argtypes); mirrors include/pvo_hip.h one to one
SIGNATURES = {
    "pvo_strerror": (_c.c_char_p, [_i]),
    "pvo_version": (_i, []),
    "pvo_graph_update_args_size": (_sz, []),
    "pvo_last_hip_error": (_c.c_char_p, []),
    "pvo_debug_config": (_i, [_i, _i]),
    "pvo_knob": (_i, [_i]),
    "pvo_corr_index_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pvo_corr_index_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pvo_corr_pyramid_lookup": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "pvo_altcorr_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pvo_altcorr_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pvo_corr_build": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "pvo_gru_glo_chunks": (_i, [_i]),
    "pvo_gru_glo_fused": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "pvo_gate_context": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "pvo_gru_conv_gates": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_gru_conv_candidate": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_conv3x3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pvo_conv3x3_c128": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pvo_conv7x7_c8": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_flow_encoder": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "pvo_corr_encode": (_i, [_vp, _vp, _vp, _vp, _c.c_longlong, _i, _vp]),
    "pvo_eta_head": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp]),
    "pvo_conv1x1_c128": (_i, [_vp, _vp, _vp, _vp, _c.c_longlong, _i, _i, _i, _vp]),
    "pvo_corr_build_tiled": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "pvo_corr_lookup_encode_tiled": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "pvo_corr_pyramid_lookup_tiled": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "pvo_heads_out": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_conv3x3_heads": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_heads_gather": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_segment_mean": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_graph_motion": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_reproject_motion": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_bias_norm_act": (_i, [_vp, _vp, _vp, _vp, _c.c_longlong, _i, _i, _i, _f, _i, _i, _i, _vp]),
    "pvo_bias_norm_act_slices": (_i, [_i]),
    "pvo_bias_norm_act_split": (_i, [_vp, _vp, _vp, _vp, _c.c_longlong, _i, _i, _i, _f, _i, _i, _i, _vp, _sz, _vp]),
    "pvo_conv1x1_planes": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pvo_frame_normalise": (_i, [_vp, _vp, _i, _i, _c.POINTER(_f), _c.POINTER(_f), _i, _i, _vp]),
    "pvo_segment_hist": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "pvo_graph_post": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _i, _f, _i, _vp]),
    "pvo_operator_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "pvo_update_operator": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "pvo_graph_update_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "pvo_graph_update": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "pvo_se3_unary": (_i, [_i, _vp, _vp, _c.c_longlong, _i, _vp]),
    "pvo_se3_binary": (_i, [_i, _vp, _c.c_longlong, _vp, _c.c_longlong, _vp, _c.c_longlong, _i, _vp]),
    "pvo_se3_unary_vjp": (_i, [_i, _vp, _vp, _vp, _c.c_longlong, _i, _vp]),
    "pvo_se3_binary_vjp": (_i, [_i, _vp, _c.c_longlong, _vp, _c.c_longlong, _vp, _vp, _vp, _c.c_longlong, _i, _vp]),
    "pvo_proj_transform": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "pvo_proj_transform_vjp": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "pvo_side_stream": (_i, [_c.POINTER(_vp)]),
    "pvo_proximity_select": (_i, [_vp, _i, _i, _i, _i, _i, _i, _c.c_double, _vp, _vp, _i, _vp, _vp, _i, _c.POINTER(_i)]),
    "pvo_probe_arm": (_i, [_i, _i]),
    "pvo_probe_arm_every": (_i, [_i, _i, _i]),
    "pvo_probe_read": (_i, [_vp, _i]),
    "pvo_frame_distance": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "pvo_frame_distance_bidirectional": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "pvo_projmap": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "pvo_iproj": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "pvo_depth_filter": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pvo_reproject": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "pvo_ba_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "pvo_ba": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i,
                    _f, _f, _i, _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "pvo_ba_plan": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "pvo_ba_last_partition": (_i, [_vp, _sz, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_int), _vp]),
    "pvo_ba_packed_elems": (_sz, [ctypes.POINTER(ctypes.c_int), _i]),
    "pvo_ba_pack": (_i, [_vp, _vp, _vp, _i, _vp]),
    "pvo_ba_finish_packed": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _i, _i, _f,
                                  _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "pvo_ba_local": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i,
                          _i, _vp, _vp, _sz, _vp]),
    "pvo_ba_finish": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _i, _i, _f,
                           _vp, _vp, _i, _vp, _vp, _sz, _vp]),
    "pvo_ba_finish_riders": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _i, _i, _f,
                                  _vp, _vp, _i, _vp, _vp, _sz, _vp, _vp]),
    "pvo_ba_finish_conv1x1": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _i, _i, _f,
                                   _vp, _vp, _i, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _c.c_longlong, _i, _i, _vp]),
}



class UpdateWeights(_c.Structure):
    """pvo_update_weights (include/pvo_hip.h)"""
    _fields_ = [("dtype", _i), ("flags", _i)] + [(n, _vp) for n in (
        "enc0_w", "enc0_b", "cenc2_w", "cenc2_b", "fenc0_w", "fenc0_b", "fenc2_w", "fenc2_b", "glo_w", "glo_b",
        "gate_wt", "gate_b", "zr_w", "q_w", "zr_inp_w", "q_inp_w", "heads1_w", "heads1_b", "heads2_w", "heads2_b",
        "agg1_w", "agg1_b", "agg2_w", "agg2_b", "eta_w", "eta_b", "up_w", "up_b")]


class OperatorArgs(_c.Structure):
    """pvo_operator_args"""
    _fields_ = [("E", _i), ("H", _i), ("W", _i), ("levels", _vp * 4), ("slots", _vp), ("num_slots", _i),
                ("coords", _vp), ("corr", _vp), ("motion", _vp), ("net", _vp), ("net_out", _vp), ("inp", _vp),
                ("P_zr", _vp), ("P_q", _vp), ("static_by_slot", _i), ("seg_ptr", _vp), ("seg_idx", _vp), ("K", _i), ("heads", _vp),
                ("eta_frame", _vp), ("eta_pos", _vp), ("R", _i), ("damping", _vp), ("EP", _f), ("eta_scale", _f), ("eta", _vp),
                ("upmask", _vp)]


class GraphUpdateArgs(_c.Structure):
    """pvo_graph_update_args"""
    _fields_ = [("op", OperatorArgs), ("nframes", _i), ("poses", _vp), ("disps", _vp), ("intrinsics", _vp),
                ("ii", _vp), ("jj", _vp), ("target", _vp), ("delta_dy", _vp), ("raw_mask", _vp), ("weight", _vp),
                ("full_flow", _vp), ("segm", _vp), ("max_segments", _i), ("vote_thresh", _f), ("dy_thresh", _f),
                ("n_in", _i), ("target_ba", _vp), ("weight_ba", _vp), ("ii_ba", _vp), ("jj_ba", _vp),
                ("t0", _i), ("t1", _i), ("itrs", _i), ("motion_only", _i), ("lm", _f), ("ep", _f),
                ("sys", _vp), ("ba_ws", _vp), ("ba_ws_bytes", _sz), ("clamp_frames", _i), ("disp_min", _f),
                ("want_upmask", _i), ("context_ahead", _i), ("context_ready", _i)]


PVO_OP_CONV128_WIDE, PVO_OP_SINGLE_STREAM, PVO_OP_ENC_SIDE_STREAM = 1, 2, 4
PVO_KNOB_BA_SOLVER, PVO_KNOB_HEADS_GATHER_FLAT, PVO_KNOB_NO_RIDERS, PVO_KNOB_POST_SEPARATE = 0, 1, 2, 3

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
    # PyTorch first: its wheel carries its own libamdhip64, and the streams / pointers this module passes into the library
    # are that runtime's.  Loaded before torch, libpvo_hip.so would bind the system ROCm's copy instead - two HIP runtimes
    # in one process, and every launch on a torch stream fails (seen as "HIP launch error" on the first call).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.pvo_version() != PVO_ABI_VERSION or lib.pvo_graph_update_args_size() != ctypes.sizeof(GraphUpdateArgs):
        raise PvoHipError("libpvo_hip.so is ABI version %d with a %d-byte pvo_graph_update_args; this binding is written for version %d / %d bytes - "
                          "rebuild with `python -m pvo_amd.build`" % (lib.pvo_version(), lib.pvo_graph_update_args_size(), PVO_ABI_VERSION,
                                                                     ctypes.sizeof(GraphUpdateArgs)))
    _lib = lib
    return lib


PROBE_LIB_PATH = os.path.join(_HERE, "libpvo_probe.so")
PROBE_SIGNATURES = {                       # include/pvo_probe.h
    "pvo_clock_probe": (_i, [_vp, _i, _vp]),
    "pvo_mem_probe": (_c.c_longlong, [_vp, _sz, _i, _i, _i, _vp, _vp]),
}
_probe_lib = None


def load_probe():
    """libpvo_probe.so: the measurement kernels of bench.py / tools (not part of the product library)"""
    global _probe_lib
    if _probe_lib is None:
        load()                                            # (torch's HIP runtime first, as above)
        if not os.path.exists(PROBE_LIB_PATH):
            raise PvoHipError("libpvo_probe.so not found at %s - build it with `python -m pvo_amd.build`" % PROBE_LIB_PATH)
        lib = ctypes.CDLL(PROBE_LIB_PATH)
        for name, (res, args) in PROBE_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _probe_lib = lib
    return _probe_lib


def check(status, what):
    if status != 0:
        msg = load().pvo_strerror(status).decode()
        if status == 2:                                   # PVO_ELAUNCH: say which HIP error
            msg += " - " + load().pvo_last_hip_error().decode()
        raise PvoHipError("%s failed: %s (status %d)" % (what, msg, status))
