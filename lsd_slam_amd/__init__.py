"""lsd_slam_amd — MI355X-native (gfx950, hand-written HIP) implementation of LSD-SLAM's dense per-pixel hot path:
SE3Tracker::trackFrame and DepthMap::updateKeyframe / createKeyFrame, behind a C ABI (include/lsdhip.h).

This package holds only what that path needs: csrc/ (HIP kernels + the C ABI), capi.py (ctypes binding), slam.py
(host-side mirror of the reference's class interface), synth.py (deterministic synthetic sequences), build.py.
"""
from .capi import HYP_DTYPE, LsdHipError  # noqa: F401
from .slam import IDENTITY, Context, DepthMap, Frame, SE3Tracker, Sim3Tracker, SlamLoop, TrackingReference  # noqa: F401
