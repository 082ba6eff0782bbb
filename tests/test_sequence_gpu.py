"""Sequence-level parity (SURVEY.md Appendix C level 5): 50 frames of scene S1 at 640x480 through trackFrame ->
updateKeyframe -> (every 10th frame) finalizeKeyFrame + createKeyFrame, the HIP library and the oracle each feeding
their own outputs forward (tests/seq_loops.py; the loop is C/SlamSystem.cpp:890-1040 + :542-614 with doSlam=false,
blockUntilMapped=true).  Errors compound over the sequence (pose -> mask -> depth map -> next reference), so the bar is
expressed in units of the reference's own arithmetic spread: the oracle's scalar path vs its SSE path on the same
sequence."""
import numpy as np
import pytest

import seq_loops as sl
from common import sequence

pytestmark = pytest.mark.gpu

N_FRAMES = 50
ROS_ALL0 = {"useAffineLightningEstimation": 0, "allowNegativeIdepths": 0, "useSubpixelStereo": 0}
# (parameters, sequence index = seed of the synthetic scene — BASELINE.json configs[3] runs indices 0..7 —, frames handed to
#  updateKeyframe per mapping iteration: 1 = blockUntilMapped, 4 = the reference's live operation, C/SlamSystem.cpp:559-571)
#  lag: 0 = blockUntilMapped; 1 = tracking beside mapping with the mapper one frame behind (lsdhip_ctx_set_pipeline, the reference's two
#  threads with blockUntilMapped == false: tests/seq_loops.py::run_loop); flag: SlamSystem::trackFrame's import-and-clear of
#  depthHasBeenUpdatedFlag, i.e. Frame::setDepth in EVERY updateKeyframe — what lsd_slam_hip::SlamLoop and bench.py run)
CONFIGS = [pytest.param({}, 0, 1, 0, False, id="defaults"),
           pytest.param(ROS_ALL0, 0, 1, 0, False, id="ros-all0"),
           pytest.param({}, 3, 1, 0, False, id="defaults-seq3"),
           pytest.param({}, 7, 1, 0, False, id="defaults-seq7"),
           pytest.param({}, 0, 4, 0, False, id="defaults-livequeue4"),
           pytest.param(ROS_ALL0, 3, 4, 0, False, id="ros-all0-seq3-livequeue4"),
           pytest.param({}, 0, 1, 0, True, id="defaults-setdepth-every-update"),
           pytest.param({}, 0, 1, 1, True, id="lag1"),
           pytest.param({}, 3, 1, 1, True, id="lag1-seq3"),
           pytest.param(ROS_ALL0, 7, 4, 1, True, id="lag1-ros-all0-seq7-livequeue4")]


@pytest.mark.parametrize("params,seq_index,live_queue,lag,flag", CONFIGS)
def test_sequence_50_frames_hip_vs_oracle(oracle, params, seq_index, live_queue, lag, flag):
    import lsd_slam_amd as la
    w, h = 640, 480
    frames, depth0, K, gt = sequence(w, h, N_FRAMES, seq_index)
    op = oracle.default_params()
    for k, v in params.items():
        setattr(op, k, v)
    run_o = lambda mode, perturb=None: sl.run_oracle(oracle, frames, depth0, K, N_FRAMES, mode=mode, params=op, live_queue=live_queue, lag=lag,
                                                     clear_flag=flag, init_perturb=perturb)
    o_sse = run_o(oracle.SSE)
    o_sc = run_o(oracle.SCALAR)
    ensemble = [o_sse]
    if flag:
        # With Frame::setDepth in every updateKeyframe the tracker sees the map change every frame, and the loop becomes sensitive to
        # single decisions: the reference's OWN trajectory moves by ~3e-4 (a keyframe's pose lands on the other side of one LM stopping
        # test) when the first frame's initial estimate is shifted by 1e-6 or the SSE path is swapped for the scalar one — measured with
        # exactly this ensemble.  The device arithmetic is a third such variation, so it is compared with the member of the reference's
        # own ensemble it agrees with best, at the usual tight bound; the yardstick is that member's nearest neighbour.
        ensemble += [run_o(oracle.SSE, p) for p in ([1e-6, 0, 0], [0, -1e-6, 1e-6], [0, 1e-6, 0], [-1e-6, 0, -1e-6])]

    def hip_run():
        ctx = la.Context(w, h, K, params=params)
        if lag:
            ctx.set_pipeline(True)       # mapping stream beside tracking stream; DepthMap calls only queue their work
            ctx.set_async(True)
        return sl.run_hip(la, ctx, frames, depth0, N_FRAMES, live_queue=live_queue, lag=lag, clear_flag=flag)

    g = hip_run()
    o_ref = min(ensemble, key=lambda m: sl.rmse(g.trajectory(), m.trajectory()))
    o_alt = min([m for m in ensemble + [o_sc] if m is not o_ref], key=lambda m: sl.rmse(o_ref.trajectory(), m.trajectory()))
    if flag:
        print("reference ensemble: trajectory RMSE of each member to the SSE run %s; HIP to each member %s"
              % (["%.1e" % sl.rmse(o_sse.trajectory(), m.trajectory()) for m in ensemble + [o_sc]],
                 ["%.1e" % sl.rmse(g.trajectory(), m.trajectory()) for m in ensemble + [o_sc]]))
    try:
        assert g.dropped == o_ref.dropped == ([11, 21, 31, 41] if lag else [])
        if flag:
            # ... and, so that best-of-five is not the only bar: against the UNPERTURBED SSE run the device must sit inside the cloud the
            # reference's own ensemble spans — no farther from it than 1.5 x the distance of the two members that are farthest apart
            members = ensemble + [o_sc]
            diam = max(sl.rmse(a.trajectory(), b.trajectory()) for i, a in enumerate(members) for b in members[i + 1:])
            d_plain = sl.rmse(g.trajectory(), o_sse.trajectory())
            print("closest ensemble member: #%d of %d (0 = the unperturbed SSE run); HIP vs the unperturbed SSE run %.2e, ensemble diameter %.2e"
                  % ([m is o_ref for m in ensemble].index(True), len(ensemble), d_plain, diam))
            assert d_plain <= max(1.5 * diam, 1e-4), (d_plain, diam)      # (the device is a sixth member: it may sit at the cloud's rim)
        _compare(g, o_ref, o_alt, gt, affine_on=params.get("useAffineLightningEstimation", 1) != 0)
    except AssertionError as e:
        # diagnosis: a second HIP run on a fresh context tells a nondeterministic device path from a real disagreement with the oracle
        g2 = hip_run()
        same = len(g2.frameToKF) == len(g.frameToKF) and all(np.array_equal(a, b) for a, b in zip(g.frameToKF, g2.frameToKF))
        raise AssertionError("%s | a second HIP run gives %s poses" % (e, "IDENTICAL" if same else "DIFFERENT")) from e


def _compare(g, o_sse, o_sc, gt, affine_on=True, decision_edges=None, residual_ensemble=None, residual_abs_floor=0.0):
    """decision_edges (a list, or None): frames on which the device's LM loop ran a different number of evaluations than the oracle's — a
    stopping test that fell the other way under a different summation order; the pose bounds hold there as everywhere, but `lastResidual`
    is then the error of a different iterate — possibly of a different pyramid level: trackFrame reports the error of the last ACCEPTED
    step, and a level whose first retry is already rejected leaves the coarser level's value (SE3Tracker.cpp:399-401) — so on such frames it
    is not compared; they are appended to the list and the caller bounds how many there may be.  None: every frame at the tight bound,
    whatever the evaluation counts.
    residual_ensemble (records of the oracle's own perturbed runs, or None): where the caller compares with the closest member of such an
    ensemble, a frame's `lastResidual` bound is also at least 3x the largest deviation any member shows from that member on that frame —
    late in a sequence the members' maps have drifted apart, and with them the residual they report for the same image.
    residual_abs_floor: absolute difference of `lastResidual` that is accepted whatever the relative one — for SMALL residuals with affine
    lighting on: the device's estimate of the offset b differs from the reference's by up to 0.07 grey levels (the reference's sequential
    float32 sums being the less accurate side, test_affine_lighting_estimate_against_float64_truth), which is ~2e-3 of weighted residual
    however small that residual is; both of the oracle's paths add in the same order and therefore do not show this in their spread."""
    assert len(g.frameToKF) == len(o_sse.frameToKF) == N_FRAMES
    # identical verdicts, frame for frame
    assert g.diverged == o_sse.diverged and not any(g.diverged)
    assert g.good == o_sse.good
    assert g.kf_frames == o_sse.kf_frames == [10, 20, 30, 40, 50]

    # trajectory: camera centres in the world frame (keyframe chain with the createKeyFrame rescale factors)
    spread = sl.rmse(o_sse.trajectory(), o_sc.trajectory())
    err = sl.rmse(g.trajectory(), o_sse.trajectory())
    gtc = np.array([gt[i % N_FRAMES][4:7] for i in range(1, N_FRAMES + 1)])
    err_gt_g, err_gt_o = sl.rmse(g.trajectory(), gtc), sl.rmse(o_sse.trajectory(), gtc)
    print("trajectory RMSE: HIP vs oracle-SSE %.3e, oracle scalar vs SSE %.3e, vs GT: HIP %.3e oracle %.3e"
          % (err, spread, err_gt_g, err_gt_o))
    assert err <= max(10.0 * spread, 1e-4), (err, spread)
    assert err_gt_g <= 1.5 * err_gt_o + 1e-4, (err_gt_g, err_gt_o)
    # per-frame relative poses
    for i, (a, b) in enumerate(zip(g.frameToKF, o_sse.frameToKF)):
        assert np.linalg.norm(a[4:7] - b[4:7]) < 5e-4 and min(np.linalg.norm(a[:4] - b[:4]), np.linalg.norm(a[:4] + b[:4])) < 5e-4, i
    du = max(abs(a - b) / abs(b) for a, b in zip(g.usage, o_sse.usage))
    du_ref = max(abs(a - b) / abs(b) for a, b in zip(o_sc.usage, o_sse.usage))
    print("pointUsage: max relative difference HIP vs oracle-SSE %.2e (oracle scalar vs SSE %.2e)" % (du, du_ref))
    for i, (a, b) in enumerate(zip(g.usage, o_sse.usage)):
        assert a == pytest.approx(b, rel=5e-3), ("pointUsage of frame %d" % (i + 1), a, b)

    # lastResidual, frame by frame: within 10x the reference's own scalar-vs-SSE spread on that frame; floor 2e-3, or 2e-2 with
    # affine lighting on (its float32 sums are order-sensitive: tests/test_gpu_parity.py::test_trackframe_parity)
    floor = 2e-2 if affine_on else 2e-3
    worst = (0.0, 0, 0.0)
    for i, (a, b, c) in enumerate(zip(g.residual, o_sse.residual, o_sc.residual)):
        d, sp = abs(a - b) / abs(b), abs(c - b) / abs(b)
        if residual_ensemble:
            sp = max(sp, 0.3 * max(abs(m.residual[i] - b) / abs(b) for m in residual_ensemble if len(m.residual) > i))
        if decision_edges is not None and g.evals[i] != o_sse.evals[i]:
            decision_edges.append((i + 1, g.evals[i], o_sse.evals[i], a, b))
            continue
        if d / max(10 * sp, floor) > worst[0]:
            worst = (d / max(10 * sp, floor), i + 1, d)
        assert d <= max(10 * sp, floor) or abs(a - b) <= residual_abs_floor, ("lastResidual of frame %d" % (i + 1), a, b, c)
    print("lastResidual: tightest frame %d at %.0f %% of its bound (relative difference %.2e)" % (worst[1], 100 * worst[0], worst[2]))

    # keyframe changes: rescale factors and semi-dense counts (within 1 %)
    dr = max(abs(a - b) / abs(b) for a, b in zip(g.rescale, o_sse.rescale))
    dr_ref = max(abs(a - b) / abs(b) for a, b in zip(o_sc.rescale, o_sse.rescale))
    print("keyframe rescale factors: max relative difference HIP vs oracle-SSE %.2e (oracle scalar vs SSE %.2e)" % (dr, dr_ref))
    for i, (a, b) in enumerate(zip(g.rescale, o_sse.rescale)):
        assert a == pytest.approx(b, rel=1e-3), ("rescale factor of keyframe %d" % (i + 1), a, b)
    for a, b in zip(g.semidense + [g.final_semidense], o_sse.semidense + [o_sse.final_semidense]):
        assert abs(a - b) <= 0.01 * b, (a, b)
    # the final map itself: validity masks nearly identical, inverse depths close where both are valid
    ham = float((g.final_valid != o_sse.final_valid).mean())
    ham_ref = float((o_sc.final_valid != o_sse.final_valid).mean())
    print("final map: %d valid (oracle %d), mask Hamming %.2e (oracle scalar vs SSE %.2e)"
          % (g.final_semidense, o_sse.final_semidense, ham, ham_ref))
    assert ham <= max(10 * ham_ref, 2e-3)
    both = g.final_valid & o_sse.final_valid
    d = np.abs(g.final_map["idepth"][both] - o_sse.final_map["idepth"][both])
    both_ref = o_sc.final_valid & o_sse.final_valid
    d_ref = np.abs(o_sc.final_map["idepth"][both_ref] - o_sse.final_map["idepth"][both_ref])
    print("final idepth |diff|: median %.2e p99 %.2e (oracle scalar vs SSE: %.2e, %.2e)"
          % (np.median(d), np.percentile(d, 99), np.median(d_ref), np.percentile(d_ref, 99)))
    # without sub-pixel stereo the matched position is quantised, so a flipped match moves a pixel by a whole step:
    # the tail is bounded relative to the reference's own scalar-vs-SSE tail
    assert np.median(d) <= max(3 * np.median(d_ref), 1e-4)
    assert np.percentile(d, 99) <= max(3 * np.percentile(d_ref, 99), 5e-3)
