"""Several sequences sharing one GPU (BASELINE.json configs[3] with more sequences than GPUs): the batched entries
lsdhip_frame_create_batch and lsdhip_depth_update_batch put the work of n sequences into the same launches (blockIdx.z = sequence).
Per sequence nothing changes — same arithmetic, same launch geometry inside its slice — so every plane must be BIT-identical to what
the single calls leave, which are the ones checked against the oracle in test_gpu_parity.py."""
import numpy as np
import pytest

from common import ODOMETRY_ITS, assert_bit_equal, sequence

pytestmark = pytest.mark.gpu

IDENT7 = np.array([1.0, 0, 0, 0, 0, 0, 0])
HYP_FIELDS = ("isValid", "blacklisted", "nextStereoFrameMinID", "validity_counter", "idepth", "idepth_var", "idepth_smoothed", "idepth_var_smoothed")


@pytest.fixture(scope="module")
def hip():
    import lsd_slam_amd as la
    return la


def assert_maps_equal(a, b, what):
    for k in HYP_FIELDS:
        if k in ("isValid", "blacklisted", "nextStereoFrameMinID", "validity_counter"):
            assert np.array_equal(a[k], b[k]), "%s: %s differs at %d pixels" % (what, k, int((a[k] != b[k]).sum()))
        else:
            v = a["isValid"] > 0
            assert_bit_equal(a[k][v], b[k][v], "%s: %s" % (what, k))


@pytest.mark.parametrize("on_device", [False, True])
def test_frame_create_batch_equals_single_creations(hip, on_device):
    w, h = 320, 240
    frames, depth0, K, gt = sequence(w, h, 6)
    ctx = hip.Context(w, h, K)
    single = [hip.Frame(ctx, 10 + j, frames[j]) for j in range(6)]
    if on_device:
        import torch
        d = torch.from_numpy(np.ascontiguousarray(frames[:6])).cuda()
        torch.cuda.synchronize()
        batch = hip.Frame.createBatch(ctx, [10 + j for j in range(6)], device_ptrs=[d[j].data_ptr() for j in range(6)])
    else:
        batch = hip.Frame.createBatch(ctx, [10 + j for j in range(6)], images=[frames[j] for j in range(6)])
    for j in range(6):
        assert batch[j].id() == 10 + j
        for lvl in range(5):
            assert_bit_equal(batch[j].image(lvl), single[j].image(lvl), "frame %d image level %d" % (j, lvl))
            assert_bit_equal(batch[j].gradients(lvl), single[j].gradients(lvl), "frame %d gradients level %d" % (j, lvl))
        assert_bit_equal(batch[j].maxGradients(0), single[j].maxGradients(0), "frame %d maxGradients" % j)
        assert batch[j].refPixelWasGoodNoCreate() is None            # the mask starts "never written", as after a single creation


def test_depth_update_batch_equals_single_calls(hip):
    """S sequences, each with its own keyframe / depth map, twice: set A goes through lsdhip_depth_update one map at a time, set B
    through lsdhip_depth_update_batch.  Four rounds, with a keyframe whose depthHasBeenUpdatedFlag is still set in some of them (that
    map skips Frame::setDepth inside the shared launch).  Depth maps and keyframe depth pyramids equal bit for bit after every round."""
    w, h = 320, 240
    S = 5
    seqs = [sequence(w, h, 6, seq_index=s) for s in range(S)]
    ctx = hip.Context(w, h, seqs[0][2])
    tr = hip.SE3Tracker(ctx)
    tr.set_maxItsPerLvl(ODOMETRY_ITS)
    sets = []
    for copy in range(2):
        cur = []
        for s, (frames, depth0, K, gt) in enumerate(seqs):
            kf = hip.Frame(ctx, 1000 * s, frames[0])
            kf.setDepthFromGroundTruth(depth0)
            dm = hip.DepthMap(ctx)
            dm.initializeFromGTDepth(kf)
            ref = hip.TrackingReference()
            ref.importFrame(kf)
            kf.clearDepthHasBeenUpdatedFlag()
            cur.append({"kf": kf, "map": dm, "ref": ref, "pose": IDENT7.copy()})
        sets.append(cur)
    for rnd in range(1, 5):
        tracked = [[], []]
        for copy in range(2):
            for s in range(S):
                st = sets[copy][s]
                fr = hip.Frame(ctx, 1000 * s + rnd, seqs[s][0][rnd])
                st["pose"] = tr.trackFrame(st["ref"], fr, st["pose"])
                tracked[copy].append(fr)
        for s in range(S):
            sets[0][s]["map"].updateKeyframe([tracked[0][s]])
        hip.DepthMap.updateKeyframeBatch([sets[1][s]["map"] for s in range(S)], tracked[1])
        for s in range(S):
            a, b = sets[0][s], sets[1][s]
            assert_maps_equal(a["map"].currentDepthMap(), b["map"].currentDepthMap(), "round %d sequence %d" % (rnd, s))
            assert a["kf"].depthHasBeenUpdatedFlag() == b["kf"].depthHasBeenUpdatedFlag()
            for lvl in range(5):
                assert_bit_equal(a["kf"].idepth(lvl), b["kf"].idepth(lvl), "round %d sequence %d idepth level %d" % (rnd, s, lvl))
                assert_bit_equal(a["kf"].idepthVar(lvl), b["kf"].idepthVar(lvl), "round %d sequence %d idepthVar level %d" % (rnd, s, lvl))
            sa, sb = a["kf"].stats(), b["kf"].stats()
            assert sa["meanIdepth"] == sb["meanIdepth"] and sa["numPoints"] == sb["numPoints"] and sa["numMappedOnThis"] == sb["numMappedOnThis"]
            # the tracking thread's import: every sequence but the odd ones in round 2 (their keyframes keep the flag: no setDepth next round)
            if not (rnd == 2 and s % 2 == 1):
                for st in (a, b):
                    st["ref"].importFrame(st["kf"])
                    st["kf"].clearDepthHasBeenUpdatedFlag()


def test_depth_update_batch_rejects_what_the_single_call_rejects(hip):
    w, h = 320, 240
    frames, depth0, K, gt = sequence(w, h, 3)
    ctx = hip.Context(w, h, K)
    kf = hip.Frame(ctx, 0, frames[0])
    kf.setDepthFromGroundTruth(depth0)
    dm = hip.DepthMap(ctx)
    dm.initializeFromGTDepth(kf)
    fr = hip.Frame(ctx, 1, frames[1])                       # never tracked: no tracking parent
    with pytest.raises(Exception):
        hip.DepthMap.updateKeyframeBatch([dm], [fr])
    empty = hip.DepthMap(ctx)                                # no active keyframe
    with pytest.raises(Exception):
        hip.DepthMap.updateKeyframeBatch([empty], [fr])
    with pytest.raises(Exception):
        hip.DepthMap.updateKeyframeBatch([dm, dm], [fr, fr])  # the same map twice


@pytest.mark.parametrize("S,pipelined", [(3, False), (9, False), (3, True), (9, True)])
def test_batch_loop_follows_the_single_sequence_loops(hip, S, pipelined):
    """lsd_slam_hip::SlamLoopBatch against S separate lsd_slam_hip::SlamLoop runs over the same images, in both execution models
    (blockUntilMapped; pipelined = tracking beside mapping, mapper one frame behind): same keyframes, same dropped frames, every frame
    tracked, the poses within the tolerance a different summation order of the tracker's batch form allows (S = 3: the chain form with
    fewer, fatter workgroups per job; S = 9: throughput mode) — and a sequence replicated inside the batch comes out bit-identical to
    its twin (same inputs through different slices of the same launches)."""
    from lsd_slam_amd.driver import DriverLoop, DriverLoopBatch
    import oracle.pyoracle as po
    from common import pose_distance
    w, h, n = 320, 240, 25
    seqs = [sequence(w, h, 26, seq_index=s % 2) for s in range(S)]          # sequences 0, 1, 0, 1, ...: twins two apart
    imgs = [[np.ascontiguousarray(f) for f in q[0]] for q in seqs]
    K = seqs[0][2]
    single = []
    for s in range(2):
        drv = DriverLoop(w, h, K, imgs[s][0].ctypes.data, seqs[s][1], kf_every=10, images_on_device=False)
        drv.set_pipeline(pipelined)
        done, poses = drv.run([imgs[s][t].ctypes.data for t in range(1, n + 1)], want_poses=True)
        st = drv.stats()
        drv.close()
        assert done == n and int(st.tracked_good) == n and int(st.keyframes) == 2 and int(st.dropped) == (2 if pipelined else 0)
        single.append(np.asarray(poses))
    bl = DriverLoopBatch(w, h, K, [imgs[s][0].ctypes.data for s in range(S)], [seqs[s][1] for s in range(S)], kf_every=10, images_on_device=False)
    bl.set_pipeline(pipelined)
    done, poses = bl.run([[imgs[s][t].ctypes.data for s in range(S)] for t in range(1, n + 1)], want_poses=True)
    st = bl.stats()
    dropped = bl.dropped()
    bl.close()
    assert done == n
    for s in range(S):
        nd = 2 if pipelined else 0
        assert st[s]["frames"] == n and st[s]["tracked_good"] == n and st[s]["keyframes"] == 2 and st[s]["updates"] == n - 2 - nd and st[s]["lost"] == 0
        assert dropped[s] == nd
        if s >= 2:
            assert np.array_equal(poses[:, s], poses[:, s - 2]), "sequence %d differs from its twin %d" % (s, s - 2)
        worst = max(max(pose_distance(poses[t, s], single[s % 2][t], po)) for t in range(n))
        assert worst < 1e-3, (s, worst)


def test_change_keyframe_batch_equals_finalize_plus_create(hip):
    """finalizeKeyFrame + createKeyFrame of S maps through lsdhip_depth_change_keyframe_batch (six shared launches: the finalize pass
    with takeReActivationData and the propagation candidates fused in, propagation merge, the two regulariser passes, rescale +
    Frame::setDepth, both keyframes' pyramids) against the two single calls per map: the new maps, both keyframes' depth pyramids and
    statistics, the rescale factors, the new keyframes' poses and the old keyframes' re-activation data, bit for bit — twice in a row
    (the second change starts from the propagation scratch the first one left), with updates in between."""
    w, h = 320, 240
    S = 4
    seqs = [sequence(w, h, 9, seq_index=s) for s in range(S)]
    ctx = hip.Context(w, h, seqs[0][2])
    tr = hip.SE3Tracker(ctx)
    tr.set_maxItsPerLvl(ODOMETRY_ITS)
    sets = []
    for copy in range(2):
        cur = []
        for s, (frames, depth0, K, gt) in enumerate(seqs):
            kf = hip.Frame(ctx, 1000 * s, frames[0])
            kf.setDepthFromGroundTruth(depth0)
            dm = hip.DepthMap(ctx)
            dm.initializeFromGTDepth(kf)
            ref = hip.TrackingReference()
            ref.importFrame(kf)
            kf.clearDepthHasBeenUpdatedFlag()
            cur.append({"kf": kf, "map": dm, "ref": ref, "pose": IDENT7.copy(), "old": []})
        sets.append(cur)

    def track_all(t):
        out = [[], []]
        for copy in range(2):
            for s in range(S):
                st = sets[copy][s]
                fr = hip.Frame(ctx, 1000 * s + t, seqs[s][0][t])
                st["pose"] = tr.trackFrame(st["ref"], fr, st["pose"])
                out[copy].append(fr)
        return out

    def update_all(frames):
        for copy in range(2):
            for s in range(S):
                st = sets[copy][s]
                st["map"].updateKeyframe([frames[copy][s]])
                st["ref"].importFrame(st["kf"])
                st["kf"].clearDepthHasBeenUpdatedFlag()

    def compare(tag):
        for s in range(S):
            a, b = sets[0][s], sets[1][s]
            assert_maps_equal(a["map"].currentDepthMap(), b["map"].currentDepthMap(), "%s sequence %d" % (tag, s))
            for fa, fb, name in [(a["kf"], b["kf"], "keyframe")] + [(x, y, "old keyframe %d" % i) for i, (x, y) in enumerate(zip(a["old"], b["old"]))]:
                for lvl in range(5):
                    assert_bit_equal(fa.idepth(lvl), fb.idepth(lvl), "%s sequence %d %s idepth level %d" % (tag, s, name, lvl))
                    assert_bit_equal(fa.idepthVar(lvl), fb.idepthVar(lvl), "%s sequence %d %s idepthVar level %d" % (tag, s, name, lvl))
                sa, sb = fa.stats(), fb.stats()
                assert sa["meanIdepth"] == sb["meanIdepth"] and sa["numPoints"] == sb["numPoints"], (tag, s, name, sa, sb)
                assert fa.depthHasBeenUpdatedFlag() == fb.depthHasBeenUpdatedFlag()
                assert np.array_equal(np.asarray(fa.thisToParent_raw()), np.asarray(fb.thisToParent_raw())), (tag, s, name)

    t = 1
    for change in range(2):
        for _ in range(2):
            update_all(track_all(t))
            t += 1
        frames = track_all(t)
        t += 1
        scalesA = []
        for s in range(S):
            st = sets[0][s]
            st["map"].finalizeKeyFrame()
            scalesA.append(st["map"].createKeyFrame(frames[0][s]))
        scalesB = hip.DepthMap.changeKeyframeBatch([sets[1][s]["map"] for s in range(S)], frames[1])
        assert [np.float32(v) for v in scalesA] == [np.float32(v) for v in scalesB], (scalesA, scalesB)
        for copy in range(2):
            for s in range(S):
                st = sets[copy][s]
                st["old"].append(st["kf"])
                st["kf"] = frames[copy][s]
                st["ref"].importFrame(st["kf"])
                st["kf"].clearDepthHasBeenUpdatedFlag()
                st["pose"] = IDENT7.copy()
        compare("change %d" % change)
        # Frame::takeReActivationData of the finalised keyframes: what setFromExistingKF rebuilds from it
        for s in range(S):
            ma, mb = hip.DepthMap(ctx), hip.DepthMap(ctx)
            ma.setFromExistingKF(sets[0][s]["old"][-1])
            mb.setFromExistingKF(sets[1][s]["old"][-1])
            assert_maps_equal(ma.currentDepthMap(), mb.currentDepthMap(), "change %d sequence %d re-activation" % (change, s))
    update_all(track_all(t))
    compare("after the second change")


@pytest.mark.parametrize("pipelined", [False, True])
def test_keyframe_chains_in_a_lane_region_of_a_synchronous_context(hip, pipelined):
    """finalizeKeyFrame + createKeyFrame of several maps dealt to the side streams of a lane region, on a context that is NOT asynchronous
    (the Context default): every call resolves its deferred results — Frame::setDepth's meanIdepth / numPoints, the rescale factor —
    before it returns, i.e. while the region is open and the kernels that write them sit on a lane, not on the mapping stream.  Same
    maps, pyramids, statistics, rescale factors and poses as the same calls without lanes, on one-stream and on pipelined contexts
    (the pipelined branch of the resolve used to wait for the mapping stream only: advisor, round 4)."""
    w, h = 320, 240
    S = 3
    seqs = [sequence(w, h, 6, seq_index=s) for s in range(S)]
    results = []
    for lanes in (0, 2):
        ctx = hip.Context(w, h, seqs[0][2])
        ctx.set_pipeline(pipelined)
        tr = hip.SE3Tracker(ctx)
        tr.set_maxItsPerLvl(ODOMETRY_ITS)
        cur = []
        for s, (frames, depth0, K, gt) in enumerate(seqs):
            kf = hip.Frame(ctx, 1000 * s, frames[0])
            kf.setDepthFromGroundTruth(depth0)
            dm = hip.DepthMap(ctx)
            dm.initializeFromGTDepth(kf)
            ref = hip.TrackingReference()
            ref.importFrame(kf)
            kf.clearDepthHasBeenUpdatedFlag()
            cur.append({"kf": kf, "map": dm, "ref": ref, "pose": IDENT7.copy()})
        out = []
        for t in range(1, 5):
            frs = []
            for s in range(S):
                st = cur[s]
                fr = hip.Frame(ctx, 1000 * s + t, seqs[s][0][t])
                st["pose"] = tr.trackFrame(st["ref"], fr, st["pose"])
                frs.append(fr)
            if t % 2 == 1:
                for s in range(S):
                    cur[s]["map"].updateKeyframe([frs[s]])
                    cur[s]["ref"].importFrame(cur[s]["kf"])
                    cur[s]["kf"].clearDepthHasBeenUpdatedFlag()
                continue
            if lanes:
                ctx.lanes_begin(lanes)
            scales, olds = [], []
            for s in range(S):
                if lanes:
                    ctx.lane_select(s % lanes)
                cur[s]["map"].finalizeKeyFrame()
                olds.append(cur[s]["kf"].stats())                     # resolved inside the region
                scales.append(cur[s]["map"].createKeyFrame(frs[s]))
            if lanes:
                ctx.lanes_end()
            for s in range(S):
                st = cur[s]
                old = st["kf"]
                st["kf"] = frs[s]
                st["ref"].importFrame(st["kf"])
                st["kf"].clearDepthHasBeenUpdatedFlag()
                st["pose"] = IDENT7.copy()
                out.append((np.float32(scales[s]), olds[s]["meanIdepth"], olds[s]["numPoints"], st["kf"].stats()["meanIdepth"], st["kf"].stats()["numPoints"],
                            np.asarray(st["kf"].thisToParent_raw()).copy(), st["map"].currentDepthMap(), [st["kf"].idepth(l).copy() for l in range(5)],
                            [old.idepth(l).copy() for l in range(5)]))
        results.append(out)
    assert len(results[0]) == len(results[1]) == 2 * S
    for k, (a, b) in enumerate(zip(*results)):
        assert a[0] == b[0] and a[1:5] == b[1:5], (k, a[:5], b[:5])
        assert np.isfinite(a[0]) and a[4] > 100
        assert np.array_equal(a[5], b[5]), k
        assert_maps_equal(a[6], b[6], "change %d" % k)
        for l in range(5):
            assert_bit_equal(a[7][l], b[7][l], "change %d new keyframe idepth level %d" % (k, l))
            assert_bit_equal(a[8][l], b[8][l], "change %d old keyframe idepth level %d" % (k, l))


def test_change_keyframe_batch_rejects_bad_arguments(hip):
    w, h = 320, 240
    frames, depth0, K, gt = sequence(w, h, 3)
    ctx = hip.Context(w, h, K)
    kf = hip.Frame(ctx, 0, frames[0])
    kf.setDepthFromGroundTruth(depth0)
    dm = hip.DepthMap(ctx)
    dm.initializeFromGTDepth(kf)
    fr = hip.Frame(ctx, 1, frames[1])                       # never tracked: no tracking parent
    with pytest.raises(Exception):
        hip.DepthMap.changeKeyframeBatch([dm], [fr])
    with pytest.raises(Exception):
        hip.DepthMap.changeKeyframeBatch([hip.DepthMap(ctx)], [fr])   # no active keyframe
    with pytest.raises(Exception):
        hip.DepthMap.changeKeyframeBatch([dm], [kf])                   # the new keyframe is the current one


# ---- BASELINE.json configs[3] at its own size against the ORACLE (VERDICT r05 next #1b) ----------------------------------------------
_S8 = {}


def _s8_inputs():
    """8 sequences of scene S1 at 640x480 ("Config 4 uses 8 sequence indices (8 seeds)"), rendered once per session"""
    if "seqs" not in _S8:
        from concurrent.futures import ThreadPoolExecutor
        from lsd_slam_amd import synth
        with ThreadPoolExecutor(8) as ex:
            _S8["seqs"] = list(ex.map(lambda s: synth.make_sequence(640, 480, 50, s, "S1"), range(8)))
    return _S8["seqs"]


@pytest.mark.parametrize("pipelined,coarse", [(False, None), (True, None), (True, 1)])
def test_batch_loop_8_seeds_640x480_against_the_oracle(hip, pipelined, coarse):
    """lsd_slam_hip::SlamLoopBatch at the BASELINE configuration — 640x480, S = 8 sequences with 8 DISTINCT seeds, throughput mode of the
    tracker on (>= 8 jobs), shared updateKeyframe / keyframe-change launches, both execution models — with every sequence held against the
    ORACLE's run of the single-sequence loop (tests/seq_loops.py: C/SlamSystem.cpp:890-1040 + :542-614, Frame::setDepth in every
    updateKeyframe as SlamLoop runs it; lag = 1: the mapper one frame behind) under the bounds of tests/test_sequence_gpu.py: verdicts and
    keyframes equal, trajectory within 10x the oracle's scalar-vs-SSE spread, per-frame poses 5e-4, residuals, usage, rescale factors,
    semi-dense counts, final masks and inverse depths.  One allowance on top of those bounds, made visible: a frame on which the device's
    LM loop ran a different number of evaluations than the oracle's (at most 3 of a sequence's 50; its pose is inside the 5e-4 bound like
    every other) has a `lastResidual` that belongs to a different iterate — trackFrame reports the error of the last ACCEPTED step, which
    can even be a coarser level's — and is listed instead of compared.  Seed 3, frame 2 is one: both the C++ batch loop and the
    Python-driven single loop take the device's path there (tools/s8_debug.py).
    Where a sequence sits on one of the loop's decision edges (the oracle itself moves
    by ~3e-4 under a 1e-6 perturbation of the first estimate there), it is compared with the closest member of the oracle's own ensemble,
    exactly as test_sequence_50_frames_hip_vs_oracle does for this loop.
    coarse = 1: the same with levels 4 and 3 of every tracking batch walked by one workgroup per sequence (k_track_solo: the default from
    24 sequences per step, lsdhip_tracker_set_batch_coarse_min_jobs), held to the same bounds."""
    from concurrent.futures import ThreadPoolExecutor
    import oracle.pyoracle as po
    import seq_loops as sl
    from test_sequence_gpu import _compare
    from lsd_slam_amd.driver import DriverLoopBatch
    w, h, n, S, lag = 640, 480, 50, 8, 1 if pipelined else 0
    seqs = _s8_inputs()
    K = seqs[0][2]
    imgs = [[np.ascontiguousarray(f) for f in q[0]] for q in seqs]
    run_o = lambda s, mode, perturb=None: sl.run_oracle(po, seqs[s][0], seqs[s][1], K, n, mode=mode, lag=lag, clear_flag=True, init_perturb=perturb)
    with ThreadPoolExecutor(8) as ex:
        fut = {(s, m): ex.submit(run_o, s, mode) for s in range(S) for m, mode in (("sse", po.SSE), ("scalar", po.SCALAR))}
        # ---- the batch loop, one step per call so that every frame's result can be read ------------------------------------------------
        bl = DriverLoopBatch(w, h, K, [imgs[s][0].ctypes.data for s in range(S)], [seqs[s][1] for s in range(S)], kf_every=10, images_on_device=False)
        bl.keep_keyframes(True)
        bl.set_pipeline(pipelined)
        if coarse is not None:
            bl.set_coarse_min_jobs(coarse)
        recs = [sl.Record() for _ in range(S)]
        kfs = [0] * S
        for t in range(1, n + 1):
            done, poses = bl.run([[imgs[s][t % n].ctypes.data for s in range(S)]], want_poses=True)
            assert done == 1
            st = bl.stats()
            for s in range(S):
                r, g = bl.last_result(s), recs[s]
                g.frameToKF.append(poses[0, s].copy())
                g.diverged.append(bool(r.diverged)); g.good.append(bool(r.trackingWasGood)); g.usage.append(r.pointUsage); g.residual.append(r.lastResidual)
                g.evals.append(int(r.numEvaluations))
                if st[s]["keyframes"] > kfs[s]:
                    kfs[s] = st[s]["keyframes"]
                    g.kf_frames.append(t)
                    g.semidense.append(int((bl.download_map(s, w, h)["isValid"] > 0).sum()))
        st, dropped = bl.stats(), bl.dropped()
        for s in range(S):
            g = recs[s]
            g.rescale = [float(v) for v in bl.keyframe_log(s)[0]]
            g.final_map = bl.download_map(s, w, h)
            g.final_valid = g.final_map["isValid"] > 0
            g.final_semidense = int(g.final_valid.sum())
            g.centres = list(sl.replay_centres(g.frameToKF, g.rescale, 10, lag))
        bl.close()
        oracle_runs = {k: f.result() for k, f in fut.items()}
        # the replay of the world chain is the loop's own bookkeeping: check it on the oracle's records
        o0 = oracle_runs[(0, "sse")]
        assert np.allclose(sl.replay_centres(o0.frameToKF, o0.rescale, 10, lag), o0.trajectory(), rtol=0, atol=1e-12)
        worst = []
        for s in range(S):
            g, o_sse, o_sc = recs[s], oracle_runs[(s, "sse")], oracle_runs[(s, "scalar")]
            assert st[s]["frames"] == n and st[s]["lost"] == 0 and dropped[s] == (4 if pipelined else 0), (s, st[s], dropped[s])
            assert o_sse.dropped == ([11, 21, 31, 41] if pipelined else [])
            edges = []
            try:
                _compare(g, o_sse, o_sc, seqs[s][3], decision_edges=edges, residual_abs_floor=2.5e-3)
                member = 0
            except AssertionError as first:
                print("  seed %d against the unperturbed oracle run: %s" % (s, str(first).splitlines()[0][:300]))
                edges = []
                # a decision edge of the loop: the oracle's own ensemble (first estimate moved by 1e-6), closest member at the same bounds
                ens = [o_sse] + list(ex.map(lambda p: run_o(s, po.SSE, p), ([1e-6, 0, 0], [0, -1e-6, 1e-6], [0, 1e-6, 0], [-1e-6, 0, -1e-6])))
                o_ref = min(ens, key=lambda m: sl.rmse(g.trajectory(), m.trajectory()))
                o_alt = min([m for m in ens + [o_sc] if m is not o_ref], key=lambda m: sl.rmse(o_ref.trajectory(), m.trajectory()))
                members = ens + [o_sc]
                diam = max(sl.rmse(a.trajectory(), b.trajectory()) for i, a in enumerate(members) for b in members[i + 1:])
                d_plain = sl.rmse(g.trajectory(), o_sse.trajectory())
                assert d_plain <= max(1.5 * diam, 1e-4), (s, d_plain, diam)
                _compare(g, o_ref, o_alt, seqs[s][3], decision_edges=edges, residual_ensemble=members, residual_abs_floor=2.5e-3)
                member = [m is o_ref for m in ens].index(True)
            # frames whose LM loop ran a different number of evaluations than the oracle's (a stopping test on its edge): few, and listed
            assert len(edges) <= 3, (s, edges)
            if edges:
                print("  seed %d: frames with LM loops of different length (frame, evaluations HIP / oracle, lastResidual HIP / oracle): %s"
                      % (s, ["(%d, %d/%d, %.5f/%.5f)" % e for e in edges]))
            worst.append((s, member, sl.rmse(g.trajectory(), o_sse.trajectory()), sl.rmse(o_sc.trajectory(), o_sse.trajectory())))
        print("S = 8, 640x480, %s: per sequence (seed, oracle ensemble member used, trajectory RMSE vs oracle-SSE, oracle scalar-vs-SSE): %s"
              % ("pipelined" if pipelined else "blockUntilMapped", ["(%d, %d, %.1e, %.1e)" % w_ for w_ in worst]))
