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


def _pose_delta(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return (float(np.linalg.norm(a[4:7] - b[4:7])), float(min(np.linalg.norm(a[:4] - b[:4]), np.linalg.norm(a[:4] + b[:4]))), float(abs(a[7] - b[7]) / abs(b[7])))


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-30)


@pytest.mark.parametrize("scale", [1.0, 1.25, 0.8])
def test_sim3_track_recovers_pose_and_scale_like_the_oracle(oracle, hip, scale):
    """Whole trackFrameSim3 calls AND prefixes of the Levenberg-Marquardt loop (the first n iterations of the coarsest level, then the
    finer levels added) against the oracle's three arithmetic modes.  Since round 6 the device solves the damped 7x7 system with diagonal
    pivoting like the reference's LDL^T, and the two loops stay in step: the NUMBER OF EVALUATIONS is equal at every prefix and for the
    whole call.  The bounds are set from the reference's own spread, measured here in the same call (its SSE path with _mm_rcp_ps and its
    scalar path against its SSE path with IEEE reciprocals, which is the arithmetic the device follows; tools/sim3_parity_probe.py,
    profiles/r06_notes.md section 3): pose within 2e-5 / 5e-6 / 5e-6 (translation / quaternion / relative scale; round 5: 2e-3 / 1e-3 /
    2e-3), residuals within 3x that spread with a floor of 5e-4 (round 5: 2e-2 / 5e-2), Hessian within 3x the spread, floor 1e-4 (3e-2)."""
    w, h = 320, 240
    P = pair(oracle, hip, w, h, 3, scale)
    init = P["exp"].copy()
    init[7] = 1.0
    tg = hip.Sim3Tracker(P["ctx"])
    full = [5, 20, 50, 100, 100]
    schedules = [("level 3, %d iterations" % n, [0, 0, 0, n, 0], 3, 3) for n in (1, 2, 3, 4, 100)] + [("levels 3-2", full, 3, 2), ("levels 3-1", full, 3, 1)]
    for name, its, first, last in schedules:
        runs = {}
        for mname, mode in (("exact", oracle.SSE_EXACT_RCP), ("sse", oracle.SSE), ("scalar", oracle.SCALAR)):
            to = oracle.Sim3Tracker(w, h, P["K"], mode=mode)
            to.set_max_its(its)
            runs[mname] = to.track(P["ra"], P["fb"], init, first, last)
        tg.setMaxItsPerLvl(its)
        got, rg = tg.trackFrameSim3(P["ga"], P["gb"], init, first, last)
        ro = runs["exact"]
        want = np.array(ro.frameToRef)
        assert not ro.diverged and not tg.diverged, name
        assert rg.numEvaluations == ro.numEvaluations, (name, rg.numEvaluations, ro.numEvaluations)
        dt, dq, ds = _pose_delta(got, want)
        assert dt < 2e-5 and dq < 5e-6 and ds < 5e-6, (name, dt, dq, ds)

        def spread(f):
            return max(_rel(f(runs["sse"]), f(ro)), _rel(f(runs["scalar"]), f(ro)))
        for fname, fg, fo in (("lastResidual", rg.lastResidual, lambda r: r.lastResidual), ("lastDepthResidual", rg.lastDepthResidual, lambda r: r.lastDepthResidual),
                              ("lastPhotometricResidual", rg.lastPhotometricResidual, lambda r: r.lastPhotometricResidual)):
            assert _rel(fg, fo(ro)) <= max(3 * spread(fo), 5e-4), (name, fname, fg, fo(ro), spread(fo))
        Hg, Ho, Hs = np.array(rg.lastSim3Hessian).reshape(7, 7), np.array(ro.hessian).reshape(7, 7), np.array(runs["scalar"].hessian).reshape(7, 7)
        assert np.allclose(Hg, Hg.T)
        close(Hg, Ho, max(3 * np.abs(Hs - Ho).max() / np.abs(Ho).max(), 1e-4), "Hessian (%s)" % name)
    assert got[7] == pytest.approx(scale, rel=2e-2)


@pytest.mark.parametrize("case", ["no depth terms", "a 6x6 patch of depth"])
def test_sim3_weak_depth_constraints(oracle, hip, case):
    """After an accepted step LM_lambda falls back to 0 (Sim3Tracker.cpp:336-337), so the 7x7 system is only positive SEMI-definite when
    the depth residual constrains it weakly: with no depth term at all, row / column 6 (the scale) is exactly zero.  The reference's
    A.ldlt().solve(b) pivots on the largest diagonal and treats a zero pivot as 'no step in that unknown' — a finite increment; the
    unpivoted Gauss-Jordan of round 5 divided by that zero and the job came back as diverged.  The device now pivots the same
    way: same verdict, same number of evaluations, pose and scale as the oracle's."""
    from lsd_slam_amd import synth
    w, h = 320, 240
    sc = synth.Scene(0)
    K = synth.intrinsics(w, h)
    imgA, depthA = sc.render(0, w, h)
    imgB, depthB = sc.render(2, w, h)
    sparse = np.zeros_like(depthB)
    if case == "no depth terms":
        sparse[0, 0] = depthB[0, 0]            # the frame "has depth", but no warped point ever lands on that pixel
    else:
        sparse[117:123, 157:163] = depthB[117:123, 157:163]
    fa, fb = oracle.Frame(0, imgA, K), oracle.Frame(2, imgB, K)
    fa.set_depth_gt(depthA); fb.set_depth_gt(sparse)
    ra = oracle.TrackingReference(); ra.import_frame(fa)
    ctx = hip.Context(w, h, K)
    ga, gb = hip.Frame(ctx, 0, imgA), hip.Frame(ctx, 2, imgB)
    ga.setDepthFromGroundTruth(depthA); gb.setDepthFromGroundTruth(sparse)
    R, t = sc.frame_to_ref(2, 0)
    init = np.concatenate([synth.rot_to_quat(R), t + np.array([0.003, -0.002, 0.001]), [1.0]])
    ro = oracle.Sim3Tracker(w, h, K, mode=oracle.SSE_EXACT_RCP).track(ra, fb, init, 3, 1)
    tg = hip.Sim3Tracker(ctx)
    got, rg = tg.trackFrameSim3(ga, gb, init, 3, 1)
    want = np.array(ro.frameToRef)
    assert bool(tg.diverged) == bool(ro.diverged), (case, tg.diverged, ro.diverged)
    assert np.all(np.isfinite(got))
    if not ro.diverged:
        assert rg.numEvaluations == ro.numEvaluations, (case, rg.numEvaluations, ro.numEvaluations)
        dt, dq, ds = _pose_delta(got, want)
        assert dt < 1e-4 and dq < 2e-5 and ds < 1e-4, (case, dt, dq, ds)
        print("%s: %d evaluations, pose delta (t, q, scale) %.1e %.1e %.1e, scale %.6f" % (case, rg.numEvaluations, dt, dq, ds, got[7]))


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
