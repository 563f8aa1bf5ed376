"""slam_amd - MI355X-native RGB-D frame-pair hot path (geometry, warp, loss,
pose encoder) behind the reference project's Python entry points.

The compute lives in slam_amd/_lib/libslamhip.so (hand-written HIP for gfx950,
C ABI in include/slamhip.h).  There is no CPU fallback: importing
slam_amd.capi without the built library raises.
"""
__version__ = "0.1.0"
