"""Sim3Tracker on the GPU against the CPU oracle (oracle/orc_sim3.cpp, SSE semantics with IEEE
reciprocals).  Evaluation level: integer quantities exact, float32 sums within 2e-5 (relative to the largest entry) of the
oracle's — the per-point terms follow the same operation order, only the summation order differs.  Function level: same
divergence verdicts, poses / scale within the oracle's own scalar-vs-SSE spread."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import lsd_slam_amd as la
    return la


def pair(oracle, hip, w, h, k, scale, holes=False):
    from lsd_slam_amd import synth
    sc = synth.Scene(0)
    K = synth.intrinsics(w, h)
    imgA, depthA = sc.render(0, w, h)
    imgB, depthB = sc.render(k, w, h)
    depthB = (depthB / scale).astype(np.float32)
    if holes:
        rng = np.random.default_rng(5)
        depthA = depthA.copy(); depthB = depthB.copy()
        depthA[rng.uniform(size=depthA.shape) < 0.5] = 0
        depthB[rng.uniform(size=depthB.shape) < 0.5] = 0
    fa, fb = oracle.Frame(0, imgA, K), oracle.Frame(k, imgB, K)
    fa.set_depth_gt(depthA); fb.set_depth_gt(depthB)
    ra = oracle.TrackingReference(); ra.import_frame(fa)
    ctx = hip.Context(w, h, K)
    ga, gb = hip.Frame(ctx, 0, imgA), hip.Frame(ctx, k, imgB)
    ga.setDepthFromGroundTruth(depthA); gb.setDepthFromGroundTruth(depthB)
    R, t = sc.frame_to_ref(k, 0)
    exp = np.concatenate([synth.rot_to_quat(R), t, [scale]])
    return dict(K=K, ctx=ctx, ra=ra, fa=fa, fb=fb, ga=ga, gb=gb, exp=exp, keep=(fa, fb))


def close(a, b, rel, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    tol = rel * max(np.abs(b).max(), 1e-30)
    assert np.abs(a - b).max() <= tol, "%s: max diff %g > %g" % (what, np.abs(a - b).max(), tol)


@pytest.mark.parametrize("w,h,level,holes", [(320, 240, 1, False), (320, 240, 3, False), (640, 480, 2, True), (160, 128, 0, True)])
def test_sim3_evaluation_matches_oracle(oracle, hip, w, h, level, holes):
    P = pair(oracle, hip, w, h, 2, 1.1, holes)
    T = oracle.sim3_inv(P["exp"])
    T[4:7] += [0.004, -0.003, 0.002]            # off the optimum: non-trivial residuals
    T[7] *= 1.03
    a, b = 0.97, 1.5
    ro = oracle.Sim3Tracker(w, h, P["K"], mode=oracle.SSE_EXACT_RCP).evaluate(P["ra"], P["fb"], T, level, a, b)
    rg = hip.Sim3Tracker(P["ctx"]).evaluate(P["ga"], P["gb"], T, level, a, b)
    assert rg.warped_size == ro.warped_size and rg.warped_size > 50
    assert rg.numTermsD == ro.numTermsD and rg.numTermsP == ro.numTermsP == (ro.warped_size // 4) * 4
    assert rg.num_constraints == ro.num_constraints == 10 * (ro.warped_size // 4)
    for name in ("pointUsage", "sumResD", "sumResP", "meanD", "meanP", "mean"):
        assert getattr(rg, name) == pytest.approx(getattr(ro, name), rel=2e-5), name
    # the affine fit subtracts float32 sums of ~1e8 magnitude (syy - sy^2/sw, then sy - a sx); the reference accumulates them
    # sequentially in float32, so a is only defined to ~1e-4 relative and b to a*128 grey levels of that
    assert rg.affine_a_lastIt == pytest.approx(ro.affine_a_lastIt, rel=3e-4)
    assert rg.affine_b_lastIt == pytest.approx(ro.affine_b_lastIt, abs=5e-2)
    close(rg.A, ro.A, 2e-5, "A")
    close(rg.b, ro.b, 2e-5, "b")


def test_sim3_tail_drop_is_the_last_points_in_column_order(oracle, hip):
    """M % 4 != 0 on purpose: the dropped points are the last ones in x-outer / y-inner order, as in the SSE loops"""
    w, h = 320, 240
    seen = set()
    for k in (1, 2, 3, 4, 5):
        P = pair(oracle, hip, w, h, k, 1.0, holes=True)
        T = oracle.sim3_inv(P["exp"])
        ro = oracle.Sim3Tracker(w, h, P["K"], mode=oracle.SSE_EXACT_RCP).evaluate(P["ra"], P["fb"], T, 2)
        rg = hip.Sim3Tracker(P["ctx"]).evaluate(P["ga"], P["gb"], T, 2)
        assert rg.warped_size == ro.warped_size
        seen.add(ro.warped_size % 4)
        close(rg.A, ro.A, 2e-5, "A k=%d" % k)
        assert rg.sumResP == pytest.approx(ro.sumResP, rel=2e-5) and rg.numTermsD == ro.numTermsD
    assert seen - {0}, "no case with a tail"


@pytest.mark.parametrize("scale", [1.0, 1.25, 0.8])
def test_sim3_track_recovers_pose_and_scale_like_the_oracle(oracle, hip, scale):
    w, h = 320, 240
    P = pair(oracle, hip, w, h, 3, scale)
    init = P["exp"].copy()
    init[7] = 1.0
    to = oracle.Sim3Tracker(w, h, P["K"], mode=oracle.SSE_EXACT_RCP)
    ro = to.track(P["ra"], P["fb"], init, 3, 1)
    tg = hip.Sim3Tracker(P["ctx"])
    got, rg = tg.trackFrameSim3(P["ga"], P["gb"], init, 3, 1)
    want = np.array(ro.frameToRef)
    assert not ro.diverged and not tg.diverged
    assert got[7] == pytest.approx(scale, rel=2e-2) and got[7] == pytest.approx(want[7], rel=2e-3)
    assert np.linalg.norm(got[4:7] - want[4:7]) < 2e-3
    assert min(np.linalg.norm(got[:4] - want[:4]), np.linalg.norm(got[:4] + want[:4])) < 1e-3
    assert rg.lastResidual == pytest.approx(ro.lastResidual, rel=2e-2)
    assert rg.lastDepthResidual == pytest.approx(ro.lastDepthResidual, rel=5e-2)
    assert rg.lastPhotometricResidual == pytest.approx(ro.lastPhotometricResidual, rel=2e-2)
    assert abs(rg.numEvaluations - ro.numEvaluations) <= max(3, ro.numEvaluations // 5)
    Hg, Ho = np.array(rg.lastSim3Hessian).reshape(7, 7), np.array(ro.hessian).reshape(7, 7)
    assert np.allclose(Hg, Hg.T)
    close(Hg, Ho, 3e-2, "Hessian")


def test_sim3_first_iteration_is_identical(oracle, hip):
    """maxIts = 1 on one level: same evaluations, so the pose after one accepted LM step agrees to float32 noise"""
    w, h = 320, 240
    P = pair(oracle, hip, w, h, 2, 1.1)
    init = P["exp"].copy()
    init[7] = 1.0
    its = [0, 0, 1, 0, 0]
    to = oracle.Sim3Tracker(w, h, P["K"], mode=oracle.SSE_EXACT_RCP); to.set_max_its(its)
    tg = hip.Sim3Tracker(P["ctx"]); tg.setMaxItsPerLvl(its)
    ro = to.track(P["ra"], P["fb"], init, 3, 1)
    got, rg = tg.trackFrameSim3(P["ga"], P["gb"], init, 3, 1)
    assert rg.numEvaluations == ro.numEvaluations
    assert np.allclose(got, np.array(ro.frameToRef), atol=2e-6)
    close(rg.lastSim3Hessian, ro.hessian, 5e-5, "Hessian")


def test_sim3_diverges_on_too_few_points_and_needs_depth(oracle, hip):
    w, h = 320, 240
    P = pair(oracle, hip, w, h, 2, 1.0)
    far = np.array([1.0, 0, 0, 0, 50.0, 0, 0, 1.0])          # frame far to the side: nothing projects into the image
    ro = oracle.Sim3Tracker(w, h, P["K"], mode=oracle.SSE_EXACT_RCP).track(P["ra"], P["fb"], far, 3, 1)
    tg = hip.Sim3Tracker(P["ctx"])
    got, rg = tg.trackFrameSim3(P["ga"], P["gb"], far, 3, 1)
    assert ro.diverged and tg.diverged
    assert np.array_equal(got, [1, 0, 0, 0, 0, 0, 0, 1]) and np.array_equal(np.array(ro.frameToRef), got)
    from lsd_slam_amd import synth
    bare = hip.Frame(P["ctx"], 9, synth.Scene(0).render(1, w, h)[0])
    with pytest.raises(hip.LsdHipError):
        tg.trackFrameSim3(P["ga"], bare, far, 3, 1)


def test_sim3_batch_equals_single_calls(hip):
    """lsdhip_sim3tracker_track_batch: jobs in lock step share launches; each job computes exactly what a single call computes
    (same kernels, one slot per job), including a job that returns early while the others go on."""
    from lsd_slam_amd import synth
    w, h = 320, 240
    scn = synth.Scene(0)
    ctx = hip.Context(w, h, synth.intrinsics(w, h))
    tr = hip.Sim3Tracker(ctx)
    kfs, frs, inits = [], [], []
    for (k, scale) in ((2, 1.0), (3, 1.25), (4, 0.8)):
        imgA, depthA = scn.render(0, w, h)
        imgB, depthB = scn.render(k, w, h)
        a, b = hip.Frame(ctx, 0, imgA), hip.Frame(ctx, k, imgB)
        a.setDepthFromGroundTruth(depthA)
        b.setDepthFromGroundTruth((depthB / scale).astype(np.float32))
        R, t = scn.frame_to_ref(k, 0)
        kfs.append(a); frs.append(b)
        inits.append(np.concatenate([synth.rot_to_quat(R), t, [1.0]]))
    kfs.append(kfs[0]); frs.append(frs[0])
    inits.append(np.array([1.0, 0, 0, 0, 50.0, 0, 0, 1.0]))            # a job that diverges on its first evaluation
    want = []
    for j in range(4):
        pose, r = tr.trackFrameSim3(kfs[j], frs[j], inits[j], 3, 1)
        want.append((pose, r.numEvaluations, r.lastResidual, r.diverged, np.array(r.lastSim3Hessian)))
    poses, recs = tr.trackFrameSim3Batch(kfs, frs, np.array(inits), 3, 1)
    for j in range(4):
        assert np.array_equal(poses[j], want[j][0]), j
        assert recs[j].numEvaluations == want[j][1] and recs[j].lastResidual == want[j][2] and recs[j].diverged == want[j][3]
        assert np.array_equal(np.array(recs[j].lastSim3Hessian), want[j][4])
    assert recs[3].diverged == 1 and recs[0].diverged == 0 and recs[1].numEvaluations > 5
    assert poses[1][7] == pytest.approx(1.25, rel=2e-2)


def test_sim3_launch_budget_top_up_changes_nothing(hip):
    """The device runs the LM loop inside a budget of launches sized by the previous call (evaluations + 3; 24 for a tracker's first
    call); a job that needs more gets further budgets of 8.  A long job behind a short one (budget far too small, several top-ups)
    returns bit for bit what it returns from a fresh tracker, and so does a short one behind a long one (idle launches behind its end)."""
    from lsd_slam_amd import synth
    w, h = 320, 240
    sc = synth.Scene(0)
    K = synth.intrinsics(w, h)
    imgA, depthA = sc.render(0, w, h)
    imgB, depthB = sc.render(3, w, h)
    ctx = hip.Context(w, h, K)
    a, b = hip.Frame(ctx, 0, imgA), hip.Frame(ctx, 3, imgB)
    a.setDepthFromGroundTruth(depthA)
    b.setDepthFromGroundTruth((depthB / 1.25).astype(np.float32))
    R, t = sc.frame_to_ref(3, 0)
    init = np.concatenate([synth.rot_to_quat(R), t, [1.0]])
    short_its, long_its = [0, 0, 0, 2, 0], [5, 20, 50, 100, 100]

    def run(tr, its, levels):
        tr.setMaxItsPerLvl(its)
        T, r = tr.trackFrameSim3(a, b, init, *levels)
        return np.array(T), r.numEvaluations, r.lastResidual, np.array(r.lastSim3Hessian)

    fresh_long = run(hip.Sim3Tracker(ctx), long_its, (3, 1))
    fresh_short = run(hip.Sim3Tracker(ctx), short_its, (3, 3))
    assert fresh_long[1] > fresh_short[1] + 11, (fresh_long[1], fresh_short[1])    # at least one top-up below
    tr = hip.Sim3Tracker(ctx)
    seq = [run(tr, short_its, (3, 3)), run(tr, long_its, (3, 1)), run(tr, short_its, (3, 3)), run(tr, long_its, (3, 1))]
    for got, want in zip(seq, (fresh_short, fresh_long, fresh_short, fresh_long)):
        assert np.array_equal(got[0], want[0]) and got[1] == want[1] and got[2] == want[2] and np.array_equal(got[3], want[3])
