"""CPU tests of the oracle itself (no GPU): self-checks that catch restatement errors, the Sophus group-property
tests the reference vendors (thirdparty/Sophus/sophus/test_se3.cpp:42-86, tests.hpp) applied to the minimal SE3, and
agreement of the oracle's three tracker arithmetic paths."""
import numpy as np
import pytest

from common import ODOMETRY_ITS, assert_bit_equal, pose_distance, sequence


def test_se3_exp_log_roundtrip_sophus_elements(oracle):
    # the tangent / group element list of thirdparty/Sophus/sophus/test_se3.cpp:42-86
    tangents = [np.array(v, float) for v in [
        [0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0], [0, 1, 0, 1, 0, 0], [0, -5, 10, 0, 0, 0], [-1, 1, 0, 0, 0, 1],
        [20, -1, 0, -1, 1, 0], [30, 5, -1, 20, -1, 0]]]
    omegas = [[0.2, 0.5, 0.0], [0.2, 0.5, -1.0], [0, 0, 0], [0, 0, 0.00001], [np.pi, 0, 0], [0.2, 0.5, 0.0]]
    trans = [[0, 0, 0], [10, 0, 0], [0, 100, 5], [0, 0, 0], [4, -5, 0], [1e-6, 2e-6, 3e-6]]
    elems = []
    for om, t in zip(omegas, trans):
        T = oracle.se3_exp(np.array([0, 0, 0] + om, float))
        T[4:] = t
        elems.append(T)
    for T in elems:  # exp(log(T)) == T   (tests.hpp: expLogTest)
        T2 = oracle.se3_exp(oracle.se3_log(T))
        dt, dr = pose_distance(T, T2, oracle)
        assert dt < 1e-8 and dr < 1e-8
    for a in tangents:  # log(exp(a)) == a for |omega| < pi (tests.hpp: expMapTest)
        if np.linalg.norm(a[3:]) < np.pi:
            a2 = oracle.se3_log(oracle.se3_exp(a))
            assert np.allclose(a, a2, atol=1e-9)
    for A in elems:  # group law: (A*B)*B^-1 == A
        for B in elems:
            AB = oracle.se3_mul(A, B)
            A2 = oracle.se3_mul(AB, oracle.se3_inv(B))
            dt, dr = pose_distance(A, A2, oracle)
            assert dt < 1e-6 and dr < 1e-9


def test_ldlt6_solves_spd_system(oracle):
    rng = np.random.default_rng(3)
    L = oracle.lib()
    for _ in range(20):
        M = rng.standard_normal((6, 6)).astype(np.float32)
        A = (M @ M.T + 6 * np.eye(6)).astype(np.float32)
        b = rng.standard_normal(6).astype(np.float32)
        x = np.zeros(6, np.float32)
        L.orc_ldlt6_solve(np.ascontiguousarray(A.ravel()), b, x)
        assert np.allclose(A.astype(np.float64) @ x, b, atol=2e-5)


def test_image_pyramid_sse_and_scalar_association_agree(oracle):
    # for uint8-sourced images both associations are exact => identical bits on every level
    frames, depth0, K, _ = sequence(160, 128, 2)
    fa = oracle.Frame(0, frames[0], K)
    fb = oracle.Frame(0, frames[0], K)
    fb.L.orc_frame_set_sse_pyramid(fb.h_, 0)
    for lvl in range(5):
        assert_bit_equal(fa.plane("image", lvl), fb.plane("image", lvl), "image level %d" % lvl)
        assert_bit_equal(fa.plane("gradients", lvl), fb.plane("gradients", lvl), "gradients level %d" % lvl)


def test_gradients_border_convention(oracle):
    frames, _, K, _ = sequence(160, 128, 1)
    f = oracle.Frame(0, frames[0], K)
    g = f.plane("gradients", 0)
    img = f.plane("image", 0)
    assert np.all(g[0] == 0) and np.all(g[-1] == 0)          # rows 0 / h-1 never written (defined as 0)
    assert np.all(g[1:-1, 1:-1, 2] == img[1:-1, 1:-1])
    assert np.all(g[1:-1, 1:-1, 0] == 0.5 * (img[1:-1, 2:] - img[1:-1, :-2]))
    assert np.all(g[1:-1, 1:-1, 1] == 0.5 * (img[2:, 1:-1] - img[:-2, 1:-1]))
    # column 0 wraps across the row boundary in the linear walk (Frame.cpp:658-677)
    assert g[5, 0, 0] == 0.5 * (img[5, 1] - img[4, -1])


def test_jacobian_matches_finite_differences(oracle):
    """-b of the normal equations is the gradient of the weighted cost: check against finite differences of the
    weighted error along the 6 twist directions (SURVEY.md §8(c) self-check)."""
    frames, depth0, K, gt = sequence(160, 128, 3)
    kf = oracle.Frame(0, frames[0], K)
    kf.set_depth_gt(depth0)
    fr = oracle.Frame(1, frames[2], K)
    ref = oracle.TrackingReference()
    ref.import_frame(kf)
    tr = oracle.SE3Tracker(160, 128, K, mode=oracle.SCALAR)
    T0 = np.array([1, 0, 0, 0, 0, 0, 0], np.float32)
    r0 = tr.evaluate(ref, fr, T0, 1)
    A = np.array(r0.A).reshape(6, 6)
    b = np.array(r0.b)
    assert np.allclose(A, A.T)
    assert np.all(np.linalg.eigvalsh(A.astype(np.float64)) > 0)
    # Gauss-Newton step must reduce the weighted error
    inc = np.linalg.solve(A.astype(np.float64), -b.astype(np.float64))
    T1 = oracle.se3_exp(inc)
    r1 = tr.evaluate(ref, fr, T1.astype(np.float32), 1)
    assert r1.weightedError < r0.weightedError


def test_tracker_recovers_rendered_pose_and_paths_agree(oracle):
    w, h = 320, 240
    frames, depth0, K, gt = sequence(w, h, 3)
    kf = oracle.Frame(0, frames[0], K)
    kf.set_depth_gt(depth0)
    ref = oracle.TrackingReference()
    ref.import_frame(kf)
    est = {}
    for mode in (oracle.SCALAR, oracle.SSE, oracle.SSE_EXACT_RCP):
        tr = oracle.SE3Tracker(w, h, K, mode=mode)
        tr.set_max_its(ODOMETRY_ITS)
        fr = oracle.Frame(2, frames[2], K)
        r = tr.track(ref, fr, np.array([1, 0, 0, 0, 0, 0, 0], float))
        assert not r.diverged and r.trackingWasGood
        est[mode] = np.array(r.frameToRef)
        dt, dr = pose_distance(est[mode], gt[2], oracle)
        assert dt < 1e-2 and dr < 5e-3, (dt, dr)  # small baseline: translation/rotation ambiguity of direct alignment
    # the reference's own scalar-vs-SSE spread defines the function-level tolerance
    dt, dr = pose_distance(est[oracle.SCALAR], est[oracle.SSE], oracle)
    assert dt < 5e-4 and dr < 5e-4
    dt, dr = pose_distance(est[oracle.SSE_EXACT_RCP], est[oracle.SSE], oracle)
    assert dt < 5e-4 and dr < 5e-4


def test_sse_tail_drop_and_constraint_count(oracle):
    frames, depth0, K, _ = sequence(160, 128, 2)
    kf = oracle.Frame(0, frames[0], K)
    kf.set_depth_gt(depth0)
    fr = oracle.Frame(1, frames[1], K)
    ref = oracle.TrackingReference()
    ref.import_frame(kf)
    tr = oracle.SE3Tracker(160, 128, K, mode=oracle.SSE)
    r = tr.evaluate(ref, fr, np.array([1, 0, 0, 0, 0, 0, 0], np.float32), 2)
    assert r.num_constraints == 6 * (r.warped_size // 4)      # LGSX.h:385 quirk
    w = tr.buffer("weight_p")
    assert len(w) == r.warped_size


def test_stereo_recovers_plane_depth(oracle):
    """observe + regularise on a noisy initial map must pull inverse depths towards ground truth."""
    w, h = 320, 240
    frames, depth0, K, gt = sequence(w, h, 8)
    kf = oracle.Frame(0, frames[0], K)
    kf.set_depth_gt(depth0)
    dm = oracle.DepthMap(w, h, K)
    dm.init_gt(kf)
    hyp = dm.get()
    rng = np.random.default_rng(1)
    v = hyp["isValid"] > 0
    noise = rng.normal(0, 0.1, hyp.shape).astype(np.float32)
    for k in ("idepth", "idepth_smoothed"):
        hyp[k][v] += noise[v]
    for k in ("idepth_var", "idepth_var_smoothed"):
        hyp[k][v] = 0.1 ** 2
    dm.set(kf, hyp)
    err0 = np.abs(hyp["idepth_smoothed"][v] - 1.0 / depth0[v]).mean()
    for i in range(3, 8):
        fr = oracle.Frame(i, frames[i], K)
        fr.set_pose(np.concatenate([gt[i], [1.0]]), kf, 0.5)
        dm.update([fr])
    out = dm.get()
    v2 = (out["isValid"] > 0) & v
    err1 = np.abs(out["idepth_smoothed"][v2] - 1.0 / depth0[v2]).mean()
    assert v2.sum() > 0.5 * v.sum()
    assert err1 < 0.5 * err0, (err0, err1)


def test_regulariser_idempotent_on_constant_map(oracle):
    w, h = 160, 128
    frames, depth0, K, _ = sequence(w, h, 1)
    kf = oracle.Frame(0, frames[0], K)
    kf.set_depth_planes(np.full((h, w), 0.5, np.float32), np.full((h, w), 0.01, np.float32))
    dm = oracle.DepthMap(w, h, K)
    dm.init_gt(kf)
    dm.stage("regularize")
    a = dm.get()
    inner = a[2:-2, 2:-2]
    assert np.all(inner["isValid"] == 1)
    assert np.allclose(inner["idepth_smoothed"], 0.5, rtol=1e-6)
    dm.stage("regularize")
    b = dm.get()
    assert_bit_equal(a["idepth_smoothed"], b["idepth_smoothed"], "idempotent")


# ---- Sim3 tracker -------------------------------------------------------------------------------
def _sim3_pair(oracle, w, h, k, scale):
    """Keyframe A = frame 0 with its GT depth; keyframe B = frame k with its own GT depth, all depths of B divided by
    `scale` (B's map lives in a world `scale` times smaller).  Returns (refA, frameB, expected B->A Sim3)."""
    from lsd_slam_amd import synth
    sc = synth.Scene(0)
    K = synth.intrinsics(w, h)
    imgA, depthA = sc.render(0, w, h)
    imgB, depthB = sc.render(k, w, h)
    fa = oracle.Frame(0, imgA, K)
    fa.set_depth_gt(depthA)
    fb = oracle.Frame(k, imgB, K)
    fb.set_depth_gt((depthB / scale).astype(np.float32))
    ra = oracle.TrackingReference()
    ra.import_frame(fa)
    R, t = sc.frame_to_ref(k, 0)              # p_A = R p_B + t (metric)
    # points of B's (shrunk) map: p_Bs = p_B / scale  =>  p_A = scale R p_Bs + t
    exp = np.concatenate([synth.rot_to_quat(R), t, [scale]])
    return ra, fb, fa, exp


def test_sim3_exp_reduces_to_se3_and_scales(oracle):
    a = np.array([0.1, -0.2, 0.05, 0.02, -0.01, 0.03, 0.0])
    T = oracle.sim3_exp(a)
    assert T[7] == pytest.approx(1.0) and np.allclose(T[:7], oracle.se3_exp(a[:6]), atol=1e-12)
    T2 = oracle.sim3_exp(np.array([0, 0, 0, 0, 0, 0, np.log(2.0)]))
    assert T2[7] == pytest.approx(2.0) and np.allclose(T2[:7], [1, 0, 0, 0, 0, 0, 0])


def test_sim3_exp_on_the_sophus_test_tangents(oracle):
    """The tangent vectors of thirdparty/Sophus/sophus/test_sim3.cpp:72-85 through the oracle's Sim3::exp, against an
    independent float64 evaluation of the closed form (rotation = SO3 exp, scale = e^sigma, translation = W(omega, sigma) upsilon,
    sim3.hpp:417-428, :608-650) and the group properties tests.hpp checks for them: exp(a) exp(-a) = 1, (A B) p = A (B p)."""
    def hat(o):
        return np.array([[0, -o[2], o[1]], [o[2], 0, -o[0]], [-o[1], o[0], 0]])

    def closed_form(a):
        ups, om, sig = a[:3], a[3:6], a[6]
        th = np.linalg.norm(om)
        Om = hat(om)
        R = np.eye(3) if th < 1e-12 else np.eye(3) + np.sin(th) / th * Om + (1 - np.cos(th)) / th ** 2 * Om @ Om
        s = np.exp(sig)
        # W = integral_0^1 exp(t sigma) exp(t Omega) dt, evaluated numerically (Gauss-Legendre, 40 nodes): no formula shared
        x, wq = np.polynomial.legendre.leggauss(40)
        W = np.zeros((3, 3))
        for xi, wi in zip(0.5 * (x + 1), 0.5 * wq):
            tt = th * xi
            Rt = np.eye(3) if th < 1e-12 else np.eye(3) + np.sin(tt) / th * Om + (1 - np.cos(tt)) / th ** 2 * Om @ Om
            W += wi * np.exp(sig * xi) * Rt
        return R, W @ ups, s

    def act(T, p):
        return T[7] * (oracle.quat_to_rot(T[:4]) @ p) + T[4:7]

    def mul(A, B):
        Ra, Rb = oracle.quat_to_rot(A[:4]), oracle.quat_to_rot(B[:4])
        return Ra @ Rb, A[7] * (Ra @ B[4:7]) + A[4:7], A[7] * B[7]

    tangents = [[0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0], [0, 1, 0, 1, 0, 0, 0.1], [0, 0, 1, 0, 1, 0, 0.1],
                [-1, 1, 0, 0, 0, 1, -0.1], [20, -1, 0, -1, 1, 0, -0.1], [30, 5, -1, 20, -1, 0, 1.5]]
    p = np.array([1.0, 2.0, 4.0])                      # test_sim3.cpp:89
    Ts = []
    for a in tangents:
        a = np.array(a, np.float64)
        T = oracle.sim3_exp(a)
        R, t, s = closed_form(a)
        # rotations by more than pi wrap: compare the matrices, not the quaternions
        assert np.allclose(oracle.quat_to_rot(T[:4]), R, atol=1e-9), a
        assert T[7] == pytest.approx(s, rel=1e-12)
        assert np.allclose(T[4:7], t, rtol=1e-8, atol=1e-8), a
        Rm, tm, sm = mul(T, oracle.sim3_exp(-a))
        assert np.allclose(Rm, np.eye(3), atol=1e-9) and np.allclose(tm, 0, atol=1e-7) and sm == pytest.approx(1.0, rel=1e-12)
        assert np.allclose(act(oracle.sim3_inv(T), act(T, p)), p, atol=1e-8)
        Ts.append(T)
    for A in Ts[1:4]:
        for B in Ts[3:]:
            Rm, tm, sm = mul(A, B)
            assert np.allclose(sm * (Rm @ p) + tm, act(A, act(B, p)), atol=1e-7)


@pytest.mark.parametrize("mode", ["SCALAR", "SSE", "SSE_EXACT_RCP"])
def test_sim3_tracker_recovers_pose_and_scale(oracle, mode):
    w, h = 320, 240
    ra, fb, fa, exp = _sim3_pair(oracle, w, h, 3, 1.25)
    from lsd_slam_amd import synth
    tr = oracle.Sim3Tracker(w, h, synth.intrinsics(w, h), mode=getattr(oracle, mode))
    init = exp.copy()
    init[7] = 1.0                                   # start from the right pose but the wrong scale
    r = tr.track(ra, fb, init, 3, 1)
    got = np.array(r.frameToRef)
    assert not r.diverged and r.numEvaluations > 5
    assert got[7] == pytest.approx(1.25, rel=2e-2)                  # scale recovered from the depth residuals
    assert np.linalg.norm(got[4:7] - exp[4:7]) < 5e-3
    assert min(np.linalg.norm(got[:4] - exp[:4]), np.linalg.norm(got[:4] + exp[:4])) < 2e-3
    H = np.array(r.hessian).reshape(7, 7)
    assert np.allclose(H, H.T) and np.all(np.diag(H) > 0)


def test_sim3_sse_drops_tail_and_counts_constraints(oracle):
    from lsd_slam_amd import synth
    w, h = 320, 240
    ra, fb, fa, exp = _sim3_pair(oracle, w, h, 2, 1.0)
    K = synth.intrinsics(w, h)
    T = oracle.sim3_inv(exp)
    rs = oracle.Sim3Tracker(w, h, K, mode=oracle.SSE_EXACT_RCP).evaluate(ra, fb, T, 2)
    rc = oracle.Sim3Tracker(w, h, K, mode=oracle.SCALAR).evaluate(ra, fb, T, 2)
    assert rs.warped_size == rc.warped_size
    assert rs.numTermsP == (rs.warped_size // 4) * 4 and rc.numTermsP == rc.warped_size
    assert rs.num_constraints == 10 * (rs.warped_size // 4)          # LGS6 counts 6, LGS4 counts 4 per group of four
    assert rc.num_constraints == 2 * rc.warped_size
    # raw (undivided) systems agree up to the <= 3 dropped points of ~1000 and the reassociated sums
    assert np.allclose(np.array(rs.A), np.array(rc.A), rtol=2e-2, atol=5e-3 * np.abs(np.array(rc.A)).max())


def _hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], float)


# tangent vectors of the reference's own Sophus tests (thirdparty/Sophus/sophus/test_se3.cpp:69-81, test_sim3.cpp:73-85): (v, w[, sigma])
SOPHUS_SE3_TANGENTS = [(0, 0, 0, 0, 0, 0), (1, 0, 0, 0, 0, 0), (0, 1, 0, 1, 0, 0), (0, -5, 10, 0, 0, 0), (-1, 1, 0, 0, 0, 1), (20, -1, 0, -1, 1, 0),
                       (30, 5, -1, 20, -1, 0)]
SOPHUS_SIM3_TANGENTS = [(0, 0, 0, 0, 0, 0, 0), (1, 0, 0, 0, 0, 0, 0), (0, 1, 0, 1, 0, 0, 0.1), (0, 0, 1, 0, 1, 0, 0.1), (-1, 1, 0, 0, 0, 1, -0.1),
                        (20, -1, 0, -1, 1, 0, -0.1), (30, 5, -1, 20, -1, 0, 1.5)]


def _se3_matrix(oracle, p):
    M = np.eye(4); M[:3, :3] = oracle.quat_to_rot(p[:4]); M[:3, 3] = p[4:7]
    return M


def _sim3_matrix(oracle, T):
    M = np.eye(4); M[:3, :3] = T[7] * oracle.quat_to_rot(T[:4]); M[:3, 3] = T[4:7]
    return M


def _twist_matrix(a):
    M = np.zeros((4, 4)); M[:3, :3] = _hat(a[3:6]) + (a[6] if len(a) > 6 else 0.0) * np.eye(3); M[:3, 3] = a[:3]
    return M


def test_exp_maps_pass_the_references_sophus_tests(oracle):
    """The oracle and oracle/_ref share the builder's stand-ins for Sophus (SE3::exp / log, Sim3::exp) — the one part of the pin that is
    not the reference's own source (Sophus is vendored under thirdparty/, Eigen is not in this image, so it cannot be compiled).  The
    reference ships Sophus' own tests: expMapTest (thirdparty/Sophus/sophus/tests.hpp:90-110) demands ||exp(x).matrix() - expm(hat(x))||_F
    <= 10 epsilon (epsilon = 1e-10 for double) on the tangent vectors of test_se3.cpp / test_sim3.cpp, expLogTest (:70-88)
    ||T - exp(log(T))|| <= epsilon.  The stand-ins are held to exactly those vectors and thresholds, with scipy.linalg.expm as Eigen's
    MatrixFunctions."""
    from scipy.linalg import expm
    eps = 1e-10
    for a in SOPHUS_SE3_TANGENTS:
        a = np.array(a, float)
        T = _se3_matrix(oracle, oracle.se3_exp(a))
        assert np.linalg.norm(T - expm(_twist_matrix(a))) <= 10 * eps, a
        # (beyond the reference's list: the exponentials of its tangent vectors as group elements — (30, 5, -1, 20, -1, 0) turns by 20 rad,
        # its quaternion has w < 0 and log's theta is negative: the stand-in tested `theta < epsilon` where se3.hpp:566 tests |theta|,
        # found by this test and fixed)
        T2 = _se3_matrix(oracle, oracle.se3_exp(oracle.se3_log(oracle.se3_exp(a))))
        assert np.linalg.norm(T - T2) <= 10 * eps, a
    # expLogTest on the group elements of test_se3.cpp:40-66 (rotation vector, translation; the two products spelled out)
    rot = lambda w: oracle.se3_exp(np.concatenate([[0, 0, 0], w]))
    def elem(w, t):
        p = rot(np.array(w, float)); p[4:7] = t
        return p
    pi = np.pi
    group = [elem((0.2, 0.5, 0.0), (0, 0, 0)), elem((0.2, 0.5, -1.0), (10, 0, 0)), elem((0, 0, 0), (0, 100, 5)), elem((0, 0, 0.00001), (0, 0, 0)),
             elem((0, 0, 0.00001), (0, -0.00000001, 0.0000000001)), elem((0, 0, 0.00001), (0.01, 0, 0)), elem((pi, 0, 0), (4, -5, 0)),
             oracle.se3_mul(oracle.se3_mul(elem((0.2, 0.5, 0.0), (0, 0, 0)), elem((pi, 0, 0), (0, 0, 0))), elem((-0.2, -0.5, -0.0), (0, 0, 0))),
             oracle.se3_mul(oracle.se3_mul(elem((0.3, 0.5, 0.1), (2, 0, -7)), elem((pi, 0, 0), (0, 0, 0))), elem((-0.3, -0.5, -0.1), (0, 6, 0)))]
    for g in group:
        T1 = _se3_matrix(oracle, g)
        T2 = _se3_matrix(oracle, oracle.se3_exp(oracle.se3_log(g)))
        assert np.linalg.norm(T1 - T2) <= eps, g
    for a in SOPHUS_SIM3_TANGENTS:
        a = np.array(a, float)
        assert np.linalg.norm(_sim3_matrix(oracle, oracle.sim3_exp(a)) - expm(_twist_matrix(a))) <= 10 * eps, a


def test_exp_maps_against_the_matrix_exponential(oracle):
    """Beyond the reference's seven vectors per group: random twists over nine decades and the branch switches of the closed forms
    (rotation / scale below Sophus' epsilon, one of them only), against the matrix exponential in float64.  Bounds: relative 1e-11,
    plus what the published closed forms themselves lose to cancellation — (1 - cos theta) / theta^2 and (e^sigma - 1) / sigma carry
    eps_double / theta resp. eps_double / sigma of relative error into the translation (the reference's arithmetic, restated as is:
    3e-11 |v| at theta = 2e-7, 1e-7 |v| at sigma = 1e-9), and V = R below theta = 1e-10 is off by theta |v| / 2."""
    from scipy.linalg import expm
    rng = np.random.default_rng(11)
    twists = [rng.standard_normal(7) * sc for sc in (1e-9, 1e-5, 1e-2, 0.3, 1.5) for _ in range(6)]
    twists.append(np.array([0.1, -0.2, 0.3, 0, 0, 0, 0.0]))              # pure translation
    twists.append(np.array([0, 0, 0, 0, 0, 0, 0.4]))                     # pure scale
    twists.append(np.array([0.3, 0.1, -0.2, 1e-7, -2e-7, 1e-7, 0.2]))    # small rotation, ordinary scale
    twists.append(np.array([0.3, 0.1, -0.2, 1e-11, -2e-11, 1e-11, 0.2]))  # rotation below the small-angle switch
    twists.append(np.array([0.3, 0.1, -0.2, 0.5, -0.4, 0.2, 1e-9]))      # small scale, ordinary rotation
    twists.append(np.array([0.3, 0.1, -0.2, 0.5, -0.4, 0.2, 1e-11]))     # scale below its switch
    tiny = 4e-16
    for a in twists:
        v, theta, sigma = np.linalg.norm(a[:3]), np.linalg.norm(a[3:6]), abs(a[6])
        lost_rot = tiny / theta if theta >= 1e-10 else 0.5 * theta
        lost_scale = tiny / sigma if sigma >= 1e-10 else 0.5 * sigma
        p = oracle.se3_exp(a[:6])
        E = expm(_twist_matrix(a[:6]))
        assert np.allclose(oracle.quat_to_rot(p[:4]), E[:3, :3], rtol=0, atol=1e-12), a
        assert np.allclose(p[4:7], E[:3, 3], rtol=1e-11, atol=1e-13 + v * lost_rot), a
        assert np.allclose(oracle.se3_log(p), a[:6], rtol=1e-9, atol=1e-12 + 10 * v * lost_rot), a
        T = oracle.sim3_exp(a)
        E = expm(_twist_matrix(a))
        assert np.allclose(T[7] * oracle.quat_to_rot(T[:4]), E[:3, :3], rtol=1e-11, atol=1e-12), a
        assert np.allclose(T[4:7], E[:3, 3], rtol=1e-11, atol=1e-13 + v * (lost_rot + lost_scale)), a


def test_ldlt_stand_ins_against_numpy(oracle):
    """Eigen's A.ldlt().solve(b) (LGSX.h:411-443 call sites) is restated by the builder for oracle and oracle/_ref alike: on symmetric
    positive definite systems of the LM loop's kind — J^T W J (1 + lambda on the diagonal) over four decades of conditioning — the 6x6
    solve agrees with LAPACK (numpy.linalg.solve in float64 on the same float32 matrix) to float32 solve accuracy."""
    rng = np.random.default_rng(5)
    L = oracle.lib()
    for cond in (1e1, 1e2, 1e3, 1e4):
        for _ in range(10):
            Q, _r = np.linalg.qr(rng.standard_normal((6, 6)))
            ev = np.geomspace(1.0, cond, 6)
            A = ((Q * ev) @ Q.T).astype(np.float32)
            A = ((A + A.T) * 0.5).astype(np.float32)
            b = rng.standard_normal(6).astype(np.float32)
            x = np.zeros(6, np.float32)
            L.orc_ldlt6_solve(np.ascontiguousarray(A.ravel()), b, x)
            want = np.linalg.solve(A.astype(np.float64), b.astype(np.float64))
            assert np.linalg.norm(x - want) <= 4e-7 * cond * np.linalg.norm(want) + 1e-7, (cond, x, want)
