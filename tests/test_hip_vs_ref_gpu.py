"""The HIP library against THE REFERENCE ITSELF (oracle/_ref/liblsd_ref_sse.so: the reference's own SE3Tracker.cpp /
DepthMap.cpp / Frame.cpp compiled against stand-in dependency headers, see oracle/ref/ — prebuilt in the container that has
/root/reference, shipped to the GPU box with the snapshot).  No restatement in between: pyramids, point clouds, the
refPixelWasGood mask at a fixed pose, every depth stage, updateKeyframe and createKeyFrame bit-exact; trackFrame within the
tolerance of DESIGN.md §4 (the reference's SSE path uses _mm_rcp_ps, the device exact reciprocals)."""
import numpy as np
import pytest

from common import ODOMETRY_ITS, assert_bit_equal, pose_distance, sequence
from test_gpu_parity import assert_hyp_equal

pytestmark = pytest.mark.gpu
IDENT7 = np.array([1.0, 0, 0, 0, 0, 0, 0])


@pytest.fixture(scope="module")
def R(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not in this snapshot (built where /root/reference exists: make -C oracle ref)")
    return oracle.lib(ref="sse")


@pytest.fixture(scope="module")
def hip():
    import lsd_slam_amd as la
    return la


@pytest.mark.parametrize("w,h", [(176, 144), (640, 480)])
def test_pyramids_pointcloud_and_mask(oracle, R, hip, w, h):
    frames, depth0, K, gt = sequence(w, h, 4)
    ctx = hip.Context(w, h, K)
    fr, fg = oracle.Frame(0, frames[0], K, L=R), hip.Frame(ctx, 0, frames[0])
    for lvl in range(5):
        assert np.array_equal(ctx.intrinsics(lvl), fr.intrinsics(lvl))
        assert_bit_equal(fg.image(lvl), fr.plane("image", lvl), "image L%d" % lvl)
        assert_bit_equal(fg.gradients(lvl), fr.plane("gradients", lvl), "gradients L%d" % lvl)
    assert_bit_equal(fg.maxGradients(0), fr.plane("maxGradients", 0), "maxGradients")
    fr.set_depth_gt(depth0)
    fg.setDepthFromGroundTruth(depth0)
    for lvl in range(5):
        assert_bit_equal(fg.idepth(lvl), fr.plane("idepth", lvl), "idepth L%d" % lvl)
        assert_bit_equal(fg.idepthVar(lvl), fr.plane("idepthVar", lvl), "idepthVar L%d" % lvl)
    rr, rg = oracle.TrackingReference(L=R), hip.TrackingReference()
    rr.import_frame(fr)
    rg.importFrame(fg)
    for lvl in (4, 3, 2, 1):
        a, b = rr.pointcloud(lvl), rg.makePointCloud(lvl)
        assert_bit_equal(b[0], a[0], "posData L%d" % lvl)
        assert_bit_equal(b[1], a[1], "colorAndVarData L%d" % lvl)
        assert np.array_equal(b[3], a[3])
    trr, trg = oracle.SE3Tracker(w, h, K, L=R), hip.SE3Tracker(ctx)
    T = oracle.se3_exp(np.array([0.05, -0.03, 0.02, 0.01, -0.02, 0.03])).astype(np.float32)
    for lvl in (4, 3, 2, 1):
        f_r, f_g = oracle.Frame(3, frames[3], K, L=R), hip.Frame(ctx, 3, frames[3])
        a, b = trr.evaluate(rr, f_r, T, lvl), trg.evaluate(rg, f_g, T, lvl)
        assert (b.warped_size, b.goodCount, b.badCount, b.num_constraints) == (a.warped_size, a.goodCount, a.badCount, a.num_constraints)
        assert b.pointUsage == pytest.approx(a.pointUsage, rel=2e-5)
        # the reference's weights go through _mm_rcp_ps (relative error up to 3.7e-4): tolerance, not bits
        assert np.allclose(np.array(b.A), np.array(a.A), rtol=1e-3, atol=5e-4 * np.abs(np.array(a.A)).max())
        if lvl == 1:
            assert_bit_equal(f_g.refPixelWasGoodNoCreate(), f_r.wasgood(), "refPixelWasGood")


def test_trackframe(oracle, R, hip):
    w, h = 640, 480
    frames, depth0, K, gt = sequence(w, h, 6)
    ctx = hip.Context(w, h, K)
    kfr, kfg = oracle.Frame(0, frames[0], K, L=R), hip.Frame(ctx, 0, frames[0])
    kfr.set_depth_gt(depth0)
    kfg.setDepthFromGroundTruth(depth0)
    rr, rg = oracle.TrackingReference(L=R), hip.TrackingReference()
    rr.import_frame(kfr)
    rg.importFrame(kfg)
    trr, trg = oracle.SE3Tracker(w, h, K, L=R), hip.SE3Tracker(ctx)
    trr.set_max_its(ODOMETRY_ITS)
    trg.set_maxItsPerLvl(ODOMETRY_ITS)
    init = IDENT7.copy()
    for i in range(1, 6):
        f_r, f_g = oracle.Frame(i, frames[i], K, L=R), hip.Frame(ctx, i, frames[i])
        a = trr.track(rr, f_r, init)
        est = trg.trackFrame(rg, f_g, init)
        dt, dr = pose_distance(est, np.array(a.frameToRef), oracle)
        assert max(dt, dr) <= 5e-4, (i, dt, dr)
        assert trg.diverged == bool(a.diverged) and trg.trackingWasGood == bool(a.trackingWasGood)
        assert trg.pointUsage == pytest.approx(a.pointUsage, rel=1e-3)
        assert (f_g.refPixelWasGoodNoCreate() != f_r.wasgood()).mean() < 2e-3
        init = est


def _maps(oracle, R, hip, ctx, frames, depth0, K, w, h, op, seed=1):
    kfr, kfg = oracle.Frame(0, frames[0], K, L=R), hip.Frame(ctx, 0, frames[0])
    kfr.set_depth_gt(depth0)
    kfg.setDepthFromGroundTruth(depth0)
    dmr, dmg = oracle.DepthMap(w, h, K, params=op, L=R), hip.DepthMap(ctx)
    dmr.init_gt(kfr)
    dmg.initializeFromGTDepth(kfg)
    hyp = dmr.get()
    rng = np.random.default_rng(seed)
    v = hyp["isValid"] > 0
    noise = rng.normal(0, 0.1, hyp.shape).astype(np.float32)
    for k in ("idepth", "idepth_smoothed"):
        hyp[k][v] += noise[v]
    for k in ("idepth_var", "idepth_var_smoothed"):
        hyp[k][v] = 0.01
    hyp["isValid"][rng.uniform(size=hyp.shape) < 0.15] = 0
    hyp["blacklisted"][rng.uniform(size=hyp.shape) < 0.03] = -2
    hyp["validity_counter"] = rng.integers(0, 60, hyp.shape).astype(np.int32)
    hyp["nextStereoFrameMinID"][rng.uniform(size=hyp.shape) < 0.2] = 4.0
    dmr.set(kfr, hyp)
    dmg.setCurrentDepthMap(kfg, hyp)
    return kfr, kfg, dmr, dmg


def _frames(oracle, R, hip, ctx, frames, K, gt, kfr, kfg, ids, seed=5):
    rng = np.random.default_rng(seed)
    frs, fgs = [], []
    for i in ids:
        f_r, f_g = oracle.Frame(i, frames[i], K, L=R), hip.Frame(ctx, i, frames[i])
        twist = np.array([0.0, 0.0, 0.003, 0.0008, -0.0006, 0.0004]) * (1 + 0.1 * i)
        sim3 = np.concatenate([oracle.se3_mul(gt[i], oracle.se3_exp(twist)), [1.0]])
        f_r.set_pose(sim3, kfr, 0.3 + 0.1 * i)
        f_g.setPose(sim3, kfg, 0.3 + 0.1 * i)
        m = (rng.uniform(size=(f_r.h >> 1, f_r.w >> 1)) < 0.9).astype(np.uint8)
        f_r.set_wasgood(m)
        f_g.set_refPixelWasGood(m)
        frs.append(f_r)
        fgs.append(f_g)
    return frs, fgs


@pytest.mark.parametrize("params", [pytest.param({}, id="defaults"),
                                    pytest.param({"allowNegativeIdepths": 0, "useSubpixelStereo": 0}, id="ros-all0")])
def test_depth_stages_update_and_create_keyframe(oracle, R, hip, params):
    w, h = 640, 480
    frames, depth0, K, gt = sequence(w, h, 10)
    ctx = hip.Context(w, h, K, params=params)
    op = oracle.default_params(R)
    for k, v in params.items():
        setattr(op, k, v)
    kfr, kfg, dmr, dmg = _maps(oracle, R, hip, ctx, frames, depth0, K, w, h, op)
    kfr.set_counters(7, 3, 3, 0)
    kfg.setCounters(7, 3, 3, 0)
    frs, fgs = _frames(oracle, R, hip, ctx, frames, K, gt, kfr, kfg, [3, 4, 6, 7])
    for st, args in (("observe", (frs, fgs)), ("fillholes", None), ("regularize", None), ("regularize_occ", None)):
        if args:
            dmr.stage(st, args[0])
            dmg.stage(st, args[1])
        else:
            dmr.stage(st)
            dmg.stage(st)
        assert_hyp_equal(dmg.currentDepthMap(), dmr.get(), "HIP vs reference: " + st)
    for i in (5, 8):
        a, b = _frames(oracle, R, hip, ctx, frames, K, gt, kfr, kfg, [i], seed=i)
        dmr.update(a)
        dmg.updateKeyframe(b)
        assert_hyp_equal(dmg.currentDepthMap(), dmr.get(), "HIP vs reference: updateKeyframe %d" % i)
    for lvl in range(5):
        assert_bit_equal(kfg.idepth(lvl), kfr.plane("idepth", lvl), "kf idepth L%d" % lvl)
    a, b = _frames(oracle, R, hip, ctx, frames, K, gt, kfr, kfg, [9], seed=9)
    s_r, s_g = dmr.create_keyframe(a[0]), dmg.createKeyFrame(b[0])
    assert s_g == pytest.approx(s_r, rel=2e-5)      # float32 sequential sum (reference) vs float64 tree sum (device)
    assert_hyp_equal(dmg.currentDepthMap(), dmr.get(), "HIP vs reference: createKeyFrame", float_exact=False, rtol=1e-4)
