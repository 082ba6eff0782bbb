"""The two execution models of the sequence loop (tests/seq_loops.py::run_loop), on the ORACLE alone: what `lag = 1` — the reference's
tracking thread beside its mapping thread with the mapper one frame behind (C/SlamSystem.cpp:1026-1040, :907-920, :559-566) — changes
against blockUntilMapped, before any GPU is involved.  The GPU tests (test_sequence_gpu.py, test_dataset_gpu.py) run the HIP side
through the same loop and compare with these records."""
import numpy as np

import seq_loops as sl
from common import sequence


def test_lag1_schedule_of_the_oracle_loop(oracle):
    w, h, n = 176, 144, 26
    frames, depth0, K, gt = sequence(w, h, n + 1)
    a = sl.run_oracle(oracle, frames, depth0, K, n, lag=0)
    b = sl.run_oracle(oracle, frames, depth0, K, n, lag=1)
    # the mapper promotes a keyframe every 10 frames in both models ...
    assert a.kf_frames == [10, 20] and b.kf_frames == [10, 20]
    # ... blockUntilMapped maps every tracked frame; with the mapper one frame behind the frame that FOLLOWS a keyframe change has been
    # tracked on the replaced keyframe and is dropped by updateKeyframe (:559-566)
    assert a.dropped == [] and b.dropped == [11, 21]
    # tracked_on[i - 1]: the keyframe frame i was tracked on (its frame number; -1: a keyframe the mapper has meanwhile replaced)
    assert b.tracked_on[9] == 0 and b.tracked_on[10] == 0 and b.tracked_on[11] == 10          # frames 10, 11 on keyframe 0; 12 on keyframe 10
    assert b.tracked_on[19] == 10 and b.tracked_on[20] == -1 and b.tracked_on[21] == 20       # frame 21 still on the replaced keyframe 10
    assert a.tracked_on[9] == 0 and a.tracked_on[10] == 10 and a.tracked_on[20] == 20         # blockUntilMapped: the very next frame
    assert not any(a.diverged) and not any(b.diverged) and all(a.good) and all(b.good)
    # same scene, same frames: the camera centres agree to a fraction of the inter-frame motion (the lag costs one frame of depth refinement)
    step = np.linalg.norm(np.diff(a.trajectory(), axis=0), axis=1).mean()
    assert sl.rmse(a.trajectory(), b.trajectory()) < 0.5 * step


def test_lag0_loop_is_the_plain_track_then_map_loop(oracle):
    """run_loop(lag = 0) against the loop written out by hand with the oracle's classes: the refactoring into one loop for both models
    and both sides changed nothing for the model rounds 1-3 tested."""
    w, h, n = 176, 144, 12
    frames, depth0, K, gt = sequence(w, h, n + 1)
    rec = sl.run_oracle(oracle, frames, depth0, K, n, lag=0)
    po = oracle
    kf = po.Frame(0, frames[0], K)
    kf.set_depth_gt(depth0)
    dm = po.DepthMap(w, h, K)
    dm.init_gt(kf)
    ref = po.TrackingReference()
    ref.import_frame(kf)
    tr = po.SE3Tracker(w, h, K, mode=po.SSE)
    tr.set_max_its([5, 20, 50, 100, 0])
    pose = np.array([1.0, 0, 0, 0, 0, 0, 0])
    since = 0
    for i in range(1, n + 1):
        f = po.Frame(i, frames[i], K)
        r = tr.track(ref, f, pose)
        assert np.array_equal(np.array(r.frameToRef), rec.frameToKF[i - 1]), i
        since += 1
        if since >= 10:
            dm.finalize()
            dm.create_keyframe(f)
            kf = f
            ref.import_frame(kf)
            pose = np.array([1.0, 0, 0, 0, 0, 0, 0])
            since = 0
        else:
            dm.update([f])
            ref.import_frame(kf)
            pose = np.array(r.frameToRef)
