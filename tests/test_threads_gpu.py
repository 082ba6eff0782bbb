"""The reference drives the hot path from two threads that share frames (tracking and mapping, C/SlamSystem.h:124-131); both
use ONE context here (include/lsd_slam_hip.hpp Context::get).  liblsdhip keeps host-side state per context (arena free list,
deferred-result slot ring, profiling events): every C-ABI entry takes the context mutex (lsdhip_internal.hpp LSD_CTX_LOCK), and
trackFrame releases it while it waits for the device.  Stress: a mapping thread (updateKeyframe + frame churn) against a
tracking thread on the same context; every tracking result must equal the single-threaded one bit for bit."""
import threading

import numpy as np
import pytest

from common import ODOMETRY_ITS, sequence

pytestmark = pytest.mark.gpu


def test_tracking_and_mapping_threads_share_a_context():
    import lsd_slam_amd as la
    w, h = 320, 240
    frames, depth0, K, gt = sequence(w, h, 8)
    ctx = la.Context(w, h, K)
    kf = la.Frame(ctx, 0, frames[0])
    kf.setDepthFromGroundTruth(depth0)
    ref = la.TrackingReference()
    ref.importFrame(kf)
    tr = la.SE3Tracker(ctx)
    tr.set_maxItsPerLvl(ODOMETRY_ITS)
    want = [np.array(tr.trackFrame(ref, la.Frame(ctx, i, frames[i]), la.IDENTITY)) for i in range(1, 6)]

    # the mapping thread works on its OWN keyframe / depth map (so that tracking inputs do not change under the tracker) but
    # on the same context: it shares the arena free list, the slot ring and the stream with the tracking thread
    kf2 = la.Frame(ctx, 100, frames[0])
    kf2.setDepthFromGroundTruth(depth0)
    dm = la.DepthMap(ctx)
    dm.initializeFromGTDepth(kf2)
    errors, stop = [], threading.Event()

    def mapper():
        try:
            k = 0
            while not stop.is_set():
                f = la.Frame(ctx, 200 + k, frames[1 + k % 7])
                f.setPose(np.concatenate([gt[1 + k % 7], [1.0]]), kf2, 0.5)
                dm.updateKeyframe([f])
                kf2.stats()                   # resolves deferred results (slot ring)
                junk = [la.Frame(ctx, 1000 + k * 4 + j, frames[j]) for j in range(4)]   # arena churn
                del junk, f
                k += 1
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    th = threading.Thread(target=mapper)
    th.start()
    try:
        for rep in range(40):
            for i in range(1, 6):
                got = np.array(tr.trackFrame(ref, la.Frame(ctx, i, frames[i]), la.IDENTITY))
                assert np.array_equal(got, want[i - 1]), (rep, i)
    finally:
        stop.set()
        th.join(60)
    assert not errors, errors
    assert not th.is_alive()
