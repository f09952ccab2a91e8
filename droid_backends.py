"""Top-level `droid_backends`: the name the reference's callers import (`import droid_backends`,
VO_Module/droid_slam/modules/corr.py:4, depth_video.py:8; PYBIND11_MODULE(droid_backends ...) VO_Module/src/droid.cpp:234).
With this repository's root on sys.path those import lines work unmodified and reach libpvo_hip.so
through pvo_amd.droid_backends (there is no CPU fallback behind it)."""
from pvo_amd.droid_backends import *                    # noqa: F401,F403
from pvo_amd.droid_backends import (ba, frame_distance, projmap, iproj, depth_filter, corr_index_forward,      # noqa: F401
                                    corr_index_backward, altcorr_forward, altcorr_backward)
