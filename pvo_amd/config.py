"""Host-layer test / A-B knobs, set by CODE - never read from the environment (two ranks with different environments would
otherwise run different schedules; the library's own knobs went the same way in round 5: `droid_backends.debug_config`).

    from pvo_amd import config
    config.debug_config("hip_graphs", False)     # no HIP-graph capture of the per-frame work (pvo_amd/graphs.py)
    config.debug_config("se3_torch", True)       # the PyTorch formulation of the SE3 operations / projective_transform everywhere
    config.debug_config("conv128_wide", False)   # corr_encoder[2] / GraphAgg.conv1 on the 128-input kernel
    config.debug_config("agg_side_stream", False), config.debug_config("enc_side_stream", False)   # operator on one stream
    config.debug_config("graph_check_skipped", True)   # GraphedCall verifies every argument copy it skips (slow: a device sync)

Process-wide; a product caller never touches them.  Knobs of the operator's schedule take effect at the next
`DynamicUpdateModule.packed_weights()` (they are part of its cache key)."""

KNOBS = {
    "hip_graphs": True,
    "se3_torch": False,
    "conv128_wide": True,
    "agg_side_stream": True,
    "enc_side_stream": True,
    "graph_check_skipped": False,
}


def debug_config(knob, value):
    if knob not in KNOBS:
        raise KeyError("unknown knob %r (have: %s)" % (knob, ", ".join(sorted(KNOBS))))
    KNOBS[knob] = bool(value)
    if knob == "se3_torch":
        from .geom import se3
        se3.FORCE_TORCH = bool(value)


def get(knob):
    return KNOBS[knob]
