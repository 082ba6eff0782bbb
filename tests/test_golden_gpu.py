"""HIP path against the committed golden fixture (no oracle involved at run time): bit-exact digests for everything the
contract makes bit-exact (pyramids, masks at fixed pose, the whole depth update), tolerances for reductions / pose."""
import hashlib
import os

import numpy as np
import pytest

from common import ROOT

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(ROOT, "tests", "golden", "s1_160x128.npz")
W, H = 160, 128


def digest(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


@pytest.fixture(scope="module")
def env():
    import lsd_slam_amd as la
    g = np.load(GOLDEN)
    ctx = la.Context(W, H, g["K"])
    kf = la.Frame(ctx, 0, g["frames"][0])
    kf.setDepthFromGroundTruth(g["depth0"])
    return la, g, ctx, kf


def test_pyramids_match_golden_digests(env):
    la, g, ctx, kf = env
    for lvl in range(5):
        assert np.array_equal(digest(kf.image(lvl)), g["image_L%d" % lvl])
        assert np.array_equal(digest(kf.gradients(lvl)), g["gradients_L%d" % lvl])
        assert np.array_equal(digest(kf.idepth(lvl)), g["idepth_L%d" % lvl])
        assert np.array_equal(digest(kf.idepthVar(lvl)), g["idepthVar_L%d" % lvl])
    assert np.array_equal(digest(kf.maxGradients(0)), g["maxGradients_L0"])
    ref = la.TrackingReference()
    ref.importFrame(kf)
    assert [len(ref.makePointCloud(l)[0]) for l in range(5)] == list(g["pointcloud_counts"])


def test_residual_kernel_matches_golden_records(env):
    la, g, ctx, kf = env
    ref = la.TrackingReference()
    ref.importFrame(kf)
    tr = la.SE3Tracker(ctx)
    for row, lvl in zip(g["residual_records"], (4, 3, 2, 1)):
        f = la.Frame(ctx, 3, g["frames"][3])
        r = tr.evaluate(ref, f, g["fixed_pose"], lvl, 1.0, 0.0)
        assert (r.warped_size, r.goodCount, r.badCount, r.num_constraints) == tuple(row[:4]), lvl   # exact
        assert r.pointUsage == pytest.approx(row[4], rel=2e-5)
        assert r.weightedError == pytest.approx(row[5], rel=5e-5)
        A, b = np.array(row[6:42]), np.array(row[42:48])
        assert np.allclose(np.array(r.A), A, rtol=3e-4, atol=1e-4 * np.abs(A).max())
        assert np.allclose(np.array(r.b), b, rtol=3e-4, atol=1e-4 * np.abs(b).max())
        if lvl == 1:
            m = f.refPixelWasGoodNoCreate()
            assert np.array_equal(np.packbits(m == 1), g["wasgood_fixed_pose"])     # mask: bit-exact
            assert np.array_equal(np.packbits(m == 255), g["wasgood_unset"])


def test_trackframe_and_update_match_golden(env):
    la, g, ctx, kf0 = env
    kf = la.Frame(ctx, 0, g["frames"][0])
    kf.setDepthFromGroundTruth(g["depth0"])
    ref = la.TrackingReference()
    ref.importFrame(kf)
    tr = la.SE3Tracker(ctx)
    tr.set_maxItsPerLvl([5, 20, 50, 100, 0])
    f2 = la.Frame(ctx, 2, g["frames"][2])
    est = tr.trackFrame(ref, f2, la.IDENTITY)
    gp = g["track_pose"]
    # pose: the oracle's SSE path uses _mm_rcp_ps, the device exact reciprocals; tolerance = DESIGN.md §4
    dq = min(np.linalg.norm(est[:4] - gp[:4]), np.linalg.norm(est[:4] + gp[:4]))
    assert dq < 5e-4 and np.linalg.norm(est[4:] - gp[4:]) < 5e-4
    s = g["track_scalars"]
    assert tr.pointUsage == pytest.approx(s[0], rel=1e-3) and tr.lastResidual == pytest.approx(s[3], rel=5e-2)
    assert int(tr.diverged) == int(s[4]) and int(tr.trackingWasGood) == int(s[5])
    assert (f2.refPixelWasGoodNoCreate() != g["track_wasgood"]).mean() < 2e-3
    # depth update with exactly the golden pose and mask: bit-exact everything
    f2.setPose(np.concatenate([gp, [1.0]]), kf, float(g["track_initialTrackedResidual"][0]))
    f2.set_refPixelWasGood(g["track_wasgood"])
    dm = la.DepthMap(ctx)
    dm.initializeFromGTDepth(kf)
    dm.updateKeyframe([f2])
    hyp = dm.currentDepthMap()
    v = hyp["isValid"] > 0
    assert np.array_equal(np.packbits(v), g["update_valid"])
    assert int(v.sum()) == int(g["update_num_valid"][0])
    assert np.array_equal(digest(hyp["blacklisted"]), g["update_blacklisted"])
    assert np.array_equal(digest(hyp["validity_counter"][v]), g["update_validity_counter"])
    for k in ("idepth", "idepth_var", "idepth_smoothed", "idepth_var_smoothed"):
        assert np.array_equal(digest(hyp[k][v]), g["update_" + k]), k
    assert np.array_equal(digest(kf.idepth(1)), g["update_kf_idepth_L1"])


def test_sim3_matches_golden_records(env):
    la, g, ctx, kf = env
    fb = la.Frame(ctx, 3, g["frames"][3])
    fb.setDepthFromGroundTruth((g["depth0"] / 1.1).astype(np.float32))
    s3 = la.Sim3Tracker(ctx)
    for row, lvl in zip(g["sim3_records"], (3, 2, 1)):
        r = s3.evaluate(kf, fb, g["sim3_fixed_pose"], lvl, 1.0, 0.0)
        assert (r.warped_size, r.numTermsD, r.numTermsP, r.num_constraints) == tuple(row[:4]), lvl   # exact
        assert r.pointUsage == pytest.approx(row[4], rel=2e-5)
        assert r.sumResD == pytest.approx(row[5], rel=5e-5) and r.sumResP == pytest.approx(row[6], rel=5e-5)
        A, b = np.array(row[7:56]), np.array(row[56:63])
        assert np.abs(np.array(r.A) - A).max() <= 5e-5 * np.abs(A).max()
        assert np.abs(np.array(r.b) - b).max() <= 5e-5 * np.abs(b).max()
    est, rs = s3.trackFrameSim3(kf, fb, np.array([1.0, 0, 0, 0, 0, 0, 0, 1.0]), 3, 1)
    gp, sc = g["sim3_track_pose"], g["sim3_track_scalars"]
    assert int(rs.diverged) == int(sc[4])
    dq = min(np.linalg.norm(est[:4] - gp[:4]), np.linalg.norm(est[:4] + gp[:4]))
    assert dq < 1e-3 and np.linalg.norm(est[4:7] - gp[4:7]) < 2e-3 and est[7] == pytest.approx(gp[7], rel=2e-3)
    assert rs.lastResidual == pytest.approx(sc[0], rel=3e-2) and rs.pointUsage == pytest.approx(sc[3], rel=1e-3)
