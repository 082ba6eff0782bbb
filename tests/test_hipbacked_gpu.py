"""INTEGRATION.md path B on the GPU: oracle/_ref/liblsd_ref_hipbacked.so is the reference's own Frame / TrackingReference code and its
UNMODIFIED SE3Tracker.h / DepthMap.h, with the two classes' members defined by integration/hip_backed/*.cpp over the C ABI.  It sits
behind the same orc_* entry points as the compiled reference (oracle/ref/ref_capi.cpp), so the calls below ARE
lsd_slam::SE3Tracker::trackFrame(TrackingReference*, Frame*, const SE3&) and lsd_slam::DepthMap::updateKeyframe / createKeyFrame /
finalizeKeyFrame as SlamSystem makes them (SlamSystem.cpp:932, :571, :473, :400) — and they land in k_track_step / k_observe / k_reg_fused.
Held against (1) the same operations through the ctypes binding of the C ABI: bit-identical (same library, same inputs), which also proves
the host-side mirroring (Frame::setDepth of the reference on the downloaded hypotheses == the device's planes); (2) the reference itself
(liblsd_ref_sse.so) at the tolerance of tests/test_hip_vs_ref_gpu.py."""
import numpy as np
import pytest

from common import ODOMETRY_ITS, assert_bit_equal, pose_distance, sequence

pytestmark = pytest.mark.gpu
IDENT7 = np.array([1.0, 0, 0, 0, 0, 0, 0])
HYP_INT = ("isValid", "blacklisted", "validity_counter")
HYP_F = ("nextStereoFrameMinID", "idepth", "idepth_var", "idepth_smoothed", "idepth_var_smoothed")


@pytest.fixture(scope="module")
def HB(oracle):
    import os
    if not os.path.exists(os.path.join(oracle.REF_DIR, "liblsd_ref_hipbacked.so")):
        pytest.skip("oracle/_ref/liblsd_ref_hipbacked.so not in this snapshot (make -C oracle ref, needs /root/reference)")
    import ctypes
    L = oracle.lib(ref="hipbacked")
    L.orc_ref_build_info.restype = ctypes.c_char_p
    assert b"integration/hip_backed" in L.orc_ref_build_info()
    return L


@pytest.fixture(scope="module")
def hip():
    import lsd_slam_amd as la
    return la


def maps_equal(a, b, what):
    for k in HYP_INT:
        assert np.array_equal(a[k], b[k]), "%s: %s differs at %d pixels" % (what, k, int((a[k] != b[k]).sum()))
    v = a["isValid"] > 0
    for k in HYP_F:
        assert_bit_equal(a[k][v], b[k][v], "%s: %s" % (what, k))


@pytest.mark.parametrize("w,h", [(320, 240), (640, 480)])
def test_reference_class_api_runs_on_the_device(oracle, HB, hip, w, h):
    frames, depth0, K, gt = sequence(w, h, 14)
    # ---- through the reference's classes (hip-backed) --------------------------------------------------------------------------------
    kf_b = oracle.Frame(0, frames[0], K, L=HB)
    kf_b.set_depth_gt(depth0)
    dm_b = oracle.DepthMap(w, h, K, L=HB)
    dm_b.init_gt(kf_b)
    ref_b = oracle.TrackingReference(L=HB)
    ref_b.import_frame(kf_b)
    tr_b = oracle.SE3Tracker(w, h, K, L=HB)
    tr_b.set_max_its(ODOMETRY_ITS)
    # ---- the same operations through the ctypes binding of the C ABI --------------------------------------------------------------------
    ctx = hip.Context(w, h, K)
    kf_g = hip.Frame(ctx, 0, frames[0])
    kf_g.setDepthFromGroundTruth(depth0)
    dm_g = hip.DepthMap(ctx)
    dm_g.initializeFromGTDepth(kf_g)
    ref_g = hip.TrackingReference()
    ref_g.importFrame(kf_g)
    tr_g = hip.SE3Tracker(ctx)
    tr_g.set_maxItsPerLvl(ODOMETRY_ITS)
    # ---- and the reference itself --------------------------------------------------------------------------------------------------------
    R = oracle.lib(ref="sse") if oracle.have_ref() else None
    if R is not None:
        kf_r = oracle.Frame(0, frames[0], K, L=R)
        kf_r.set_depth_gt(depth0)
        ref_r = oracle.TrackingReference(L=R)
        ref_r.import_frame(kf_r)
        tr_r = oracle.SE3Tracker(w, h, K, L=R)
        tr_r.set_max_its(ODOMETRY_ITS)
    maps_equal(dm_b.get(), dm_g.currentDepthMap(), "initializeFromGTDepth")
    init = IDENT7.copy()
    kf_b_cur, kf_g_cur = kf_b, kf_g
    for i in range(1, 13):
        f_b, f_g = oracle.Frame(i, frames[i], K, L=HB), hip.Frame(ctx, i, frames[i])
        if kf_b_cur.stats()["depthHasBeenUpdatedFlag"]:                      # SlamSystem::trackFrame's import-and-clear (SlamSystem.cpp:907-912)
            ref_b.import_frame(kf_b_cur)
            st = kf_b_cur.stats()
            kf_b_cur.set_counters(int(st["numFramesTrackedOnThis"]), int(st["numMappedOnThis"]), int(st["numMappedOnThisTotal"]), 0)
            ref_g.importFrame(kf_g_cur)
            kf_g_cur.clearDepthHasBeenUpdatedFlag()
        r_b = tr_b.track(ref_b, f_b, init)
        p_g = tr_g.trackFrame(ref_g, f_g, init)
        # lsd_slam::SE3Tracker::trackFrame through the reference's header == lsdhip_tracker_track through ctypes: the same job on the same
        # inputs — the returned pose has passed through Sophus::SE3d, whose constructor re-normalises the quaternion (1 ulp)
        assert np.allclose(np.array(r_b.frameToRef), p_g, rtol=0, atol=1e-15), (i, np.array(r_b.frameToRef), p_g)
        for a, b in (("pointUsage", "pointUsage"), ("lastGoodCount", "lastGoodCount"), ("lastBadCount", "lastBadCount"), ("lastMeanRes", "lastMeanRes"),
                     ("lastResidual", "lastResidual"), ("affine_a", "affineEstimation_a"), ("affine_b", "affineEstimation_b")):
            assert getattr(r_b, a) == getattr(tr_g.last, b), (i, a)
        assert bool(r_b.diverged) == tr_g.diverged and bool(r_b.trackingWasGood) == tr_g.trackingWasGood and tr_g.trackingWasGood
        # the side effects on the HOST frame of the reference (SE3Tracker.cpp:479-485)
        assert np.array_equal(f_b.wasgood(), f_g.refPixelWasGoodNoCreate()), "refPixelWasGood of frame %d" % i
        assert np.allclose(f_b.pose(), f_g.thisToParent_raw(), rtol=0, atol=1e-15) and f_b.stats()["initialTrackedResidual"] == f_g.stats()["initialTrackedResidual"]
        assert int(kf_b_cur.stats()["numFramesTrackedOnThis"]) == int(kf_g_cur.stats()["numFramesTrackedOnThis"])
        if R is not None and i == 1:
            f_r = oracle.Frame(i, frames[i], K, L=R)
            r_r = tr_r.track(ref_r, f_r, init)
            dt, dr = pose_distance(np.array(r_b.frameToRef), np.array(r_r.frameToRef), oracle)
            assert dt < 5e-4 and dr < 5e-4, (dt, dr)                                   # the reference's own arithmetic (SSE path with _mm_rcp_ps)
            assert float((f_b.wasgood() != f_r.wasgood()).mean()) < 2e-3
        init = p_g
        if i % 6 == 0:
            # keyframe change as SlamSystem does it: finishCurrentKeyframe + createNewCurrentKeyframe (SlamSystem.cpp:400, :473)
            dm_b.finalize(); dm_g.finalizeKeyFrame()
            maps_equal(dm_b.get(), dm_g.currentDepthMap(), "finalizeKeyFrame at frame %d" % i)
            s_b = dm_b.create_keyframe(f_b)
            s_g = dm_g.createKeyFrame(f_g)
            assert s_b == pytest.approx(s_g, rel=0, abs=0)
            maps_equal(dm_b.get(), dm_g.currentDepthMap(), "createKeyFrame at frame %d" % i)
            assert np.allclose(f_b.pose(), f_g.thisToParent_raw(), rtol=0, atol=1e-15)
            kf_b_cur, kf_g_cur = f_b, f_g
            ref_b.import_frame(kf_b_cur); ref_g.importFrame(kf_g_cur)
            init = IDENT7.copy()
        else:
            dm_b.update([f_b]); dm_g.updateKeyframe([f_g])
            maps_equal(dm_b.get(), dm_g.currentDepthMap(), "updateKeyframe with frame %d" % i)
            f_b.clear_wasgood(); f_g.clear_refPixelWasGood()
        # the host keyframe of the reference holds the depth the device computed: Frame::setDepth (the reference's code, on the downloaded
        # hypotheses) against the device's planes, every pyramid level
        for lvl in range(5):
            assert_bit_equal(kf_b_cur.plane("idepth", lvl), kf_g_cur.idepth(lvl), "host idepth L%d after frame %d" % (lvl, i))
            assert_bit_equal(kf_b_cur.plane("idepthVar", lvl), kf_g_cur.idepthVar(lvl), "host idepthVar L%d after frame %d" % (lvl, i))
        sb, sg = kf_b_cur.stats(), kf_g_cur.stats()
        assert int(sb["numMappedOnThis"]) == int(sg["numMappedOnThis"]) and int(sb["numMappedOnThisTotal"]) == int(sg["numMappedOnThisTotal"])
        # (Frame::setDepth of the reference adds sequentially in float32, the device in a float64 tree: the mean of ~10^5 values differs by up to 6e-5 relative)
        assert sb["meanIdepth"] == pytest.approx(sg["meanIdepth"], rel=2e-4) and int(sb["numPoints"]) == int(sg["numPoints"])
