"""Shared helpers for the parity tests: deterministic synthetic inputs (lsd_slam_amd.synth) at sizes the CPU oracle
finishes in seconds, and small comparison utilities."""
import functools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from lsd_slam_amd import synth  # noqa: E402

ODOMETRY_ITS = [5, 20, 50, 100, 0]  # SlamSystem forces level 4 to 0 iterations (C/SlamSystem.cpp:80-81)


@functools.lru_cache(maxsize=8)
def sequence(w, h, n, seq_index=0, kind="S1"):
    return synth.make_sequence(w, h, n, seq_index, kind)


def bits(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        return a.view(np.uint32)
    return a


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    ne = bits(a) != bits(b)
    # NaN payloads: treat NaN == NaN
    if a.dtype == np.float32:
        ne &= ~(np.isnan(a) & np.isnan(b))
    n = int(ne.sum())
    assert n == 0, "%s: %d of %d elements differ (first at %s: %r vs %r)" % (
        what, n, a.size, np.argwhere(ne)[0] if n else None, a[ne][0] if n else None, b[ne][0] if n else None)


def pose_distance(pa, pb, po):
    """|log(Ta^-1 Tb)| split into (translation, rotation) norms; po = oracle.pyoracle (SE3 helpers)."""
    d = po.se3_log(po.se3_mul(po.se3_inv(pa), pb))
    return float(np.linalg.norm(d[:3])), float(np.linalg.norm(d[3:]))
