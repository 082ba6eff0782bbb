#!/usr/bin/env python
"""bench.py — LSD-SLAM dense hot path on MI355X: tracked frames/sec (+ depth-map Mpixels/sec) at 640x480.

One "step" = one frame of the full track+map loop of BASELINE.json configs[1]: device-resident uint8 frame ->
image/gradient pyramids -> SE3Tracker::trackFrame against the current keyframe (5-level pyramid, LM over levels
4..1) -> DepthMap::updateKeyframe([frame]) (observe/doLineStereo, fill holes, regularise, setDepth); every
KF_EVERY-th frame finalizeKeyFrame + createKeyFrame (propagate, 2x regularise, fill holes, rescale) instead.
Inputs (the rendered uint8 frames) are resident in HBM before the timed region starts.

Execution model (default): the reference's own — tracking beside mapping (SlamSystem's two threads with blockUntilMapped == false,
C/SlamSystem.cpp:1026-1040) on two HIP streams of one context, the mapper exactly one frame behind the tracker (lsd_slam_hip::SlamLoop,
pipelined): frame t + 1 is tracked against the depth map(t - 1) left while map(t) runs beside it; the frame that follows a keyframe
change is tracked on the old keyframe and dropped by the mapper (C/SlamSystem.cpp:559-566).  --block-until-mapped runs the one-stream
loop of rounds 1-3 (every frame mapped before the next is tracked); its frames/s is reported beside `value` either way.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), every rank runs an independent sequence
(weak scaling; BASELINE.json configs[3]).  Finished keyframes' inverse-depth planes leave the C++ loop through a device
ring and are collected on rank 0 with one RCCL gather over xGMI per batch of frames — the only exchange step of the path.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

KF_EVERY = 10
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md chip table
HBM_ACHIEVABLE_GBS = 6290.0  # what a copy kernel achieves on this part (same table): every roofline object also carries frac_of_achievable


def pmc_traffic(kernel="k_track_step"):
    """HBM bytes per launch of a kernel (default: the residual kernel) from the committed PMC passes (profiles/rNN_pmc_traffic.json, made by
    tools/gpu_profile.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this same command, KiB units,
    FETCH_SIZE doubled on gfx950).  Counters cannot be collected from inside an un-profiled run, so the latest committed
    summary is reported; None if there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        for name, t in d["kernels"].items():
            if kernel in name and "batch" not in name and t.get("hbm_bytes_per_launch") is not None:
                return float(t["hbm_bytes_per_launch"]), os.path.basename(files[-1])
    except Exception:
        pass
    return None, None


def add_achievable(obj):
    """every roofline object (achieved / peak / frac in GB/s) also gets its fraction of the bandwidth a copy kernel reaches on this part
    (report the fraction of both 8.0 TB/s and ~6.29 TB/s)"""
    if isinstance(obj, dict):
        if obj.get("unit") == "GB/s" and "achieved" in obj and "peak" in obj and obj.get("achieved") is not None:
            obj["frac_of_achievable"] = obj["achieved"] / (HBM_ACHIEVABLE_GBS * obj["peak"] / HBM_PEAK_GBS)
        for v in obj.values():
            add_achievable(v)
    elif isinstance(obj, list):
        for v in obj:
            add_achievable(v)


def host_cpu():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return {"nproc": os.cpu_count(), "cpu_model": model}


def _cpu_loop(po, L, frames, depth0, K, n_frames, w, h, mode):
    """the same track + map loop (new keyframe every KF_EVERY frames) on a CPU library behind the oracle's entry points"""
    kf = po.Frame(0, frames[0], K, L=L)
    kf.set_depth_gt(depth0)
    dm = po.DepthMap(w, h, K, L=L, threads=4)
    dm.init_gt(kf)
    ref = po.TrackingReference(L=L)
    ref.import_frame(kf)
    tr = po.SE3Tracker(w, h, K, mode=mode, L=L)
    tr.set_max_its([5, 20, 50, 100, 0])
    last = np.array([1.0, 0, 0, 0, 0, 0, 0])
    t_track = t_map = 0.0
    n_upd = 0
    since = 0
    good = 0
    t_all = time.perf_counter()
    for i in range(1, n_frames + 1):
        t0 = time.perf_counter()
        f = po.Frame(i, frames[i % len(frames)], K, L=L)
        ref.import_frame(kf)
        r = tr.track(ref, f, last)
        good += int(r.trackingWasGood)
        t1 = time.perf_counter()
        t_track += t1 - t0
        since += 1
        if since >= KF_EVERY:
            dm.finalize()
            dm.create_keyframe(f)
            kf = f
            last = np.array([1.0, 0, 0, 0, 0, 0, 0])
            since = 0
        else:
            dm.update([f])
            f.clear_wasgood()
            last = np.array(r.frameToRef)
            n_upd += 1
        t_map += time.perf_counter() - t1
    total = time.perf_counter() - t_all
    return {"value": n_frames / total, "unit": "frames/s", "track_fps": n_frames / t_track,
            "depth_mpix_per_s": (w * h * n_upd) / t_map / 1e6 if t_map > 0 else None, "seconds": total, "tracked_good": good,
            "updates_per_frame": n_upd / n_frames, "execution": "blockUntilMapped: track, then map, one thread at a time (+ the mapping pool's 4 workers)"}


def _quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx])


def _relative_pose(kfToParent8, frameToParent7):
    """se3FromSim3(newKeyframe.camToWorld^-1 * frame.camToWorld) for two frames tracked on the same parent (C/SlamSystem.cpp:918-920)"""
    qk, tk, sk = np.asarray(kfToParent8[:4]), np.asarray(kfToParent8[4:7]), float(kfToParent8[7])
    qf, tf = np.asarray(frameToParent7[:4]), np.asarray(frameToParent7[4:7])
    qi = qk * np.array([1.0, -1, -1, -1])
    q = _quat_mul(qi, qf)
    q = q / np.linalg.norm(q)
    return np.concatenate([q, quat_to_rot(qi) @ (tf - tk) / sk])


def _cpu_loop_pipelined(po, L, frames, depth0, K, n_frames, w, h, mode):
    """The reference's two-thread model on the CPU library (C/SlamSystem.cpp:1026-1040 with blockUntilMapped == false, :542-614, :907-920), in
    the deterministic form the GPU loop's pipelined mode has: the mapper exactly one frame behind the tracker.  Step t: the tracking
    thread imports the keyframe the mapper left one step ago and builds its point clouds (TrackingReference::makePointCloud, which reads
    the keyframe's depth planes), then trackFrame(t) runs on this thread WHILE the mapping iteration of frame t - 1 (updateKeyframe, or
    finalizeKeyFrame + createKeyFrame) runs on a second thread (its IndexThreadReduce pool of 4 workers beside it); a frame tracked on a
    keyframe the mapper has meanwhile replaced is dropped unmapped (:559-566)."""
    import threading
    kf = po.Frame(0, frames[0], K, L=L)
    kf.set_depth_gt(depth0)
    dm = po.DepthMap(w, h, K, L=L, threads=4)
    dm.init_gt(kf)
    ref = po.TrackingReference(L=L)
    tr = po.SE3Tracker(w, h, K, mode=mode, L=L)
    tr.set_max_its([5, 20, 50, 100, 0])
    track_kf, pending_kf = kf, None
    last = np.array([1.0, 0, 0, 0, 0, 0, 0])
    deferred = None                 # mapping work of the previous step: ("upd", frame) | ("kf", frame)
    since = n_upd = n_kf = n_drop = good = 0
    t_all = time.perf_counter()

    def map_work(work):
        kind, fr = work
        if kind == "upd":
            dm.update([fr])
            fr.clear_wasgood()
        else:
            dm.finalize()
            dm.create_keyframe(fr)

    for i in range(1, n_frames + 1):
        f = po.Frame(i, frames[i % len(frames)], K, L=L)
        ref.import_frame(track_kf)
        for lvl in (4, 3, 2, 1):
            L.orc_ref_pointcloud(ref.h_, lvl, None, None, None, None)      # makePointCloud(lvl): nothing is copied out
        th = None
        if deferred is not None:
            th = threading.Thread(target=map_work, args=(deferred,))
            th.start()
        tracked_on = track_kf
        r = tr.track(ref, f, last)
        good += int(r.trackingWasGood)
        if th is not None:
            th.join()
        if deferred is not None and deferred[0] == "kf":
            n_kf += 1
        elif deferred is not None:
            n_upd += 1
        deferred = None
        if pending_kf is not None:
            # the mapper promoted pending_kf while this frame was tracked on the old keyframe
            last = _relative_pose(pending_kf.pose(), np.array(r.frameToRef))
            track_kf, pending_kf = pending_kf, None
        else:
            last = np.array(r.frameToRef)
        since += 1
        if tracked_on is not kf:
            f.clear_wasgood()
            n_drop += 1
            continue
        if since >= KF_EVERY:
            deferred = ("kf", f)
            kf = f
            pending_kf = f
            since = 0
        else:
            deferred = ("upd", f)
    if deferred is not None:
        map_work(deferred)
        n_kf += deferred[0] == "kf"
        n_upd += deferred[0] == "upd"
    total = time.perf_counter() - t_all
    return {"value": n_frames / total, "unit": "frames/s", "seconds": total, "tracked_good": good, "frames": n_frames,
            "updates_per_frame": n_upd / n_frames, "keyframes_per_frame": n_kf / n_frames, "dropped_frames_per_frame": n_drop / n_frames,
            "cores": 5, "threads": "tracking thread (incl. frame creation and makePointCloud) beside a mapping thread with the reference's pool of 4 workers"}


def cpu_baseline(frames, depth0, K, n_frames, w, h):
    """CPU baseline on this box's host cores, bounded sample of the same workload, two candidates:
      * kind "reference": the REFERENCE's own hot-path sources (SE3Tracker.cpp / DepthMap.cpp / Frame.cpp ... compiled where they lie,
        against the stand-in dependency headers of oracle/ref) in their TIMING build — -O3 -DENABLE_SSE -DNDEBUG, contraction at the
        compiler default, as lsd_slam_core/CMakeLists.txt:11,36-43 builds it, at the highest portable -march level this host's cpuid
        supports (x86-64-v4 / -v3 instead of -march=native: /root/reference does not travel, the library is prebuilt).  Tracking on 1
        thread, mapping on the reference's own IndexThreadReduce pool of 4 workers (settings.h:94);
      * kind "port": the oracle's restatement built on this host with -O3 -march=native (persistent worker pool).
    `cpu_baseline` is the FASTER of the two; the other one is returned second."""
    from oracle import pyoracle as po
    po.build()
    info = host_cpu()
    native = po.build_native()
    Lp = po.lib(native=True) if native else po.lib(fast=True)
    port = _cpu_loop(po, Lp, frames, depth0, K, n_frames, w, h, po.SSE)
    port.update({"cores": 4, "kind": "port", "build": "-O3 -march=native (built on this host)" if native else "-O3 -march=x86-64-v3 (prebuilt)",
                 "sample": "%d frames of the same %dx%d track+map loop (oracle: SSE tracker on 1 thread, depth map on a persistent "
                           "pool of 4 workers x 10-row chunks as the reference)" % (n_frames, w, h)})
    port.update(info)
    key, flags = po.ref_timing_variant()
    if key is None:
        return port, None
    refb = _cpu_loop(po, po.lib(ref=key), frames, depth0, K, n_frames, w, h, po.SSE)
    refb.update({"cores": 4, "kind": "reference",
                 "build": "reference sources, " + flags + "; stand-in Eigen/Sophus/boost/OpenCV headers; prebuilt where /root/reference exists",
                 "sample": "%d frames of the same %dx%d track+map loop through the reference's own SE3Tracker / DepthMap / Frame code "
                           "(tracking 1 thread, mapping: IndexThreadReduce pool of 4 workers)" % (n_frames, w, h)})
    refb.update(info)
    # the same library in the reference's two-thread model, mapper one frame behind (what `value` of a pipelined run is measured in)
    try:
        refb["pipelined"] = _cpu_loop_pipelined(po, po.lib(ref=key), frames, depth0, K, max(60, (n_frames * 3) // 5), w, h, po.SSE)
    except Exception as e:
        refb["pipelined"] = {"error": str(e)}
    # the reference's scalar path ("both scalar-path and SSE-path numbers"): the same sources and timing flags without
    # -DENABLE_SSE, a shorter sample
    skey = key.replace("sse_", "scalar_")
    try:
        n_sc = max(50, n_frames // 2)
        sc = _cpu_loop(po, po.lib(ref=skey), frames, depth0, K, n_sc, w, h, po.SCALAR)
        refb["scalar_path"] = {"value": sc["value"], "unit": "frames/s", "track_fps": sc["track_fps"], "depth_mpix_per_s": sc["depth_mpix_per_s"],
                               "build": "reference sources, " + flags.replace(" -DENABLE_SSE", "") + " (no ENABLE_SSE)", "sample": "%d frames" % n_sc}
    except Exception as e:
        refb["scalar_path"] = {"error": str(e)}
    port["scalar_path"] = refb["scalar_path"]        # (reported with whichever candidate ends up as cpu_baseline)
    return (refb, port) if refb["value"] >= port["value"] else (port, refb)


def quat_to_rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def trajectory_rmse(poses, scales, gt_of_frame, first_scored, lag=0):
    """Camera centres in the frame of image 0 from the loop's outputs — frame->keyframe poses chained through every keyframe change
    with the Sim3 scale createKeyFrame assigned (C/DepthEstimation/DepthMap.cpp:1305) — against the synthetic ground truth.
    lag = 1 (pipelined loop): the frame after a keyframe change is still expressed in the OLD keyframe (the tracker adopts the new one
    a frame later) and is not mapped.  Returns (rmse over frames >= first_scored, number of keyframe changes)."""
    s, R, t = 1.0, np.eye(3), np.zeros(3)
    pending = None
    since = k = 0
    err = []
    for i, est in enumerate(poses):
        c = s * (R @ est[4:7]) + t
        if i >= first_scored:
            err.append(float(np.sum((c - gt_of_frame(i)[4:7]) ** 2)))
        since += 1
        if pending is not None:        # this frame was tracked on the replaced keyframe: the tracker adopts the new one now
            s, R, t = pending
            pending = None
            continue
        if since >= KF_EVERY:
            if k >= len(scales):
                break
            new = (s * float(scales[k]), R @ quat_to_rot(est[:4]), s * (R @ est[4:7]) + t)
            if lag:
                pending = new
            else:
                s, R, t = new
            k += 1
            since = 0
    return (float(np.sqrt(np.mean(err))) if err else None), k


def other_execution_model(DriverLoop, capi, w, h, K, ptr, depth0, device, pipelined, steps, warmup, trials, trial_cap, regions=3):
    """the same loop in the OTHER execution model (one stream / pipelined), a short leg after the timed region: median of `regions`
    regions of `steps` frames; reported beside `value`"""
    loop = DriverLoop(w, h, K, ptr(0), depth0, kf_every=KF_EVERY, images_on_device=True, device=device)
    loop.set_pipeline(pipelined)      # (the context is shared with the timed loop, which is not used again: switched explicitly)
    if trials > 0:
        loop.set_speculation(trials, trial_cap)
    L = capi.lib()
    ctx_h = loop.ctx_handle()
    fi = 0
    loop.run([ptr(fi + 1 + k) for k in range(warmup)])
    fi += warmup
    capi.check(L.lsdhip_ctx_synchronize(ctx_h))
    dts = []
    for _ in range(regions):
        t0 = time.perf_counter()
        loop.run([ptr(fi + 1 + k) for k in range(steps)])
        capi.check(L.lsdhip_ctx_synchronize(ctx_h))
        dts.append(time.perf_counter() - t0)
        fi += steps
    st = loop.stats()
    loop.close()
    dt = float(np.median(dts))
    return {"value": steps / dt, "unit": "frames/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "regions": regions,
            "tracked_good": int(st.tracked_good), "frames": int(st.frames), "updates": int(st.updates), "dropped": int(st.dropped)}


def throughput_mode(la, w, h, K, d_frames, depth0, device, jobs=64, rounds=3):
    """The same residual evaluation with enough independent work per launch: `jobs` trackFrame jobs (frames 1..5 of the
    sequence against the keyframe with ground-truth depth) share every launch (lsdhip_tracker_track_batch, throughput mode: the
    coarse levels in one workgroup per job — k_track_solo, from 24 jobs on —, then one fused launch per lock-step round).  Reports the algorithmic bandwidth of the step kernels over ALL launches of the
    jobs, HIP events around each batch; the level-1 evaluation launches alone run ~2.3x higher (profiles/r01_sizes.md).
    An extra, not the bench metric: the single-sequence loop above is what `value` and `roofline` describe."""
    ctx = la.Context(w, h, K, device=device)
    refs, keep = [], []
    for j in range(jobs):
        kf = la.Frame(ctx, 100000 + j, device_ptr=d_frames[0].data_ptr())
        kf.setDepthFromGroundTruth(depth0)
        r = la.TrackingReference()
        r.importFrame(kf)
        refs.append(r)
        keep.append(kf)
    tr = la.SE3Tracker(ctx)
    tr.set_maxItsPerLvl([5, 20, 50, 100, 0])
    inits = np.tile(la.IDENTITY, (jobs, 1))

    def one_round(k):
        frs = [la.Frame(ctx, 200000 + jobs * k + j, device_ptr=d_frames[1 + (j + k) % 5].data_ptr()) for j in range(jobs)]
        return tr.trackFrameBatch(refs, frs, inits)

    one_round(0)
    ctx.prof_reset()
    ctx.prof_enable(True)
    ctx.synchronize()
    for k in range(rounds):
        one_round(k + 1)
    ctx.synchronize()
    ctx.prof_enable(False)
    ms, n_eval, nbytes = ctx.prof_read()
    achieved = nbytes / (ms * 1e-3) / 1e9
    out = {"bound": "hbm", "jobs_per_launch": jobs, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS, "evaluations": int(n_eval), "stream_ms": ms,
           "kernel": "k_track_solo (levels 4, 3: one workgroup per job) + k_track_step<256, true, TS_FUSED> (one launch per lock-step round), all launches of %d batches of %d jobs (the small pyramid levels carry almost no bytes)" % (rounds, jobs)}
    # the residual evaluation launch alone, per pyramid level, at fixed poses (lsdhip_tracker_eval_throughput: 20 identical launches
    # between two HIP events): level 1 is where the bytes are
    T = np.zeros((jobs, 7), np.float32)
    T[:, 0] = 1.0
    frs = [la.Frame(ctx, 300000 + j, device_ptr=d_frames[1 + j % 5].data_ptr()) for j in range(jobs)]
    per_level = {}
    for lvl in (3, 2, 1):
        ms_l, nb_l = tr.evalThroughput(refs, frs, T, lvl, 20)
        ach = nb_l / (ms_l * 1e-3) / 1e9
        per_level["L%d" % lvl] = {"us_per_launch": ms_l * 1e3, "algorithmic_bytes_per_launch": nb_l, "achieved": ach, "frac": ach / HBM_PEAK_GBS}
    out["evaluation_launch_by_level"] = per_level
    out["level1_evaluation"] = dict(per_level["L1"], bound="hbm", peak=HBM_PEAK_GBS, unit="GB/s", kernel="k_track_step<256, true, TS_EVAL>, %d jobs per launch" % jobs)
    return out


def bands_distributed(torch, dist, rank, world, local_rank, passes=20):
    """BASELINE.json configs[4] across the ranks of this run: the 3840x2160 regulariser (fill holes + regularise fused per pass, scene S3)
    cut into one row band per rank, over BOTH transports of the C++ band loop (lsd_slam_amd/driver/slam_loop.cpp lsdband_run):
      rccl — halo rows packed and exchanged with grouped ncclSend / ncclRecv on the loop's stream;
      ipc  — rows packed straight into the neighbour's IPC-mapped mailbox (peer access over xGMI), ready / consumed flags as one-lane
             kernels on the stream: no collective library in the pass at all.  A band's pass is ~17-19 us of compute at 8 GPUs, so the
             latency of the exchange, not its 2 MB, decides which one the 8-GPU run should use.
    Each with the exchange under the interior rows of the pass (overlap 1) and after the pass (overlap 0).  Every rank calls this; rank 0
    returns the record.  With one rank (LSD_FORCE_DIST=1) the one band is the full frame and nothing is exchanged: the keys are there,
    the exchange is not exercised."""
    from lsd_slam_amd.bands import NativeBandRegularizer, synth_s3
    from lsd_slam_amd.driver import DriverLoop
    W, H = 3840, 2160
    hyp, maxgrad = synth_s3(W, H)
    rec = {"workload": "3840x2160 depth-map regularisation, one row band per rank (BASELINE.json configs[4]), %d passes, 64 algorithmic B/px per pass" % passes,
           "bands": world}

    def measure(br):
        out = {}
        for overlap in (1, 0):
            br.set_overlap(overlap)
            br.load(hyp, maxgrad)
            br.run(2)
            br.synchronize()
            dts = []
            for _ in range(3):
                br.load(hyp, maxgrad)
                br.synchronize()
                torch.cuda.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                br.run(passes)
                br.synchronize()
                dist.barrier()
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dts.append(float(t.item()))
            dt = float(np.median(dts))
            ach = W * H * passes * 64.0 / dt / 1e9
            out["overlap_%d" % overlap] = {"ms_per_pass": dt / passes * 1e3, "mpix_per_s": W * H * passes / dt / 1e6, "achieved": ach,
                                           "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": ach / (HBM_PEAK_GBS * world)}
        return out

    for transport in ("rccl", "ipc"):
        ok = torch.tensor([1], device="cuda")
        tr = {"transport": transport if world > 1 else "none (one rank: the band is the full frame)"}
        br = None
        try:
            br = NativeBandRegularizer(W, H, world, [rank], device=local_rank)
            if transport == "rccl":
                uid = [DriverLoop.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                br.comm_init(uid[0], world, rank, list(range(world)))
            elif world > 1:
                handle = br.ipc_init(world, rank, list(range(world)))
                handles = [None] * world
                dist.all_gather_object(handles, handle)
                br.ipc_connect(handles)
            tr["halo_bytes_per_pass"] = int(br.halo_bytes_per_pass())
        except Exception as e:        # (a transport this node cannot set up must not take the other one, or the line, down)
            tr["error"] = str(e)
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)                 # all ranks or none: nobody may wait in a barrier the others skip
        if int(ok.item()) == 1:
            tr.update(measure(br))
            if transport == "ipc":
                tr["flag_waits_failed"] = br.ipc_failed()
        elif "error" not in tr:
            tr["error"] = "another rank could not set this transport up"
        if br is not None:
            br.close()
        rec[transport] = tr
    # (round-5 readers: the RCCL record's keys at the top level as well)
    for k in ("overlap_1", "overlap_0", "halo_bytes_per_pass"):
        if k in rec["rccl"]:
            rec[k] = rec["rccl"][k]
    rec["transport"] = rec["rccl"]["transport"]
    return rec if rank == 0 else None


def multi_seq(torch, w, h, K, seeds, device, single_value, sizes=(8, 32, 64, 128), steps=30, warmup=10, regions=3):
    """BASELINE.json configs[3] with more sequences than GPUs: S independent sequences share this GPU (lsd_slam_hip::SlamLoopBatch — frame
    creation, tracking jobs, updateKeyframe calls and keyframe changes of all sequences in shared launches).  Two execution models per S,
    as for the single sequence: pipelined (`frames_s`: the tracking batch of step t + 1 beside the mapping work of step t, per sequence
    the mapper one frame behind) and blockUntilMapped (`frames_s_block_until_mapped`).
    `seeds`: 8 (device frames, ground-truth depth of frame 0) pairs — scene S1 rendered with the 8 seeds of SURVEY.md section 8(d)
    ("Config 4 uses 8 sequence indices").  Sequence s plays seed s % 8 with camera motion (s // 8) % 4 (the closed camera loop forwards /
    backwards at one and two frames per step); S = 8 is exactly the 8 seeds of configs[3], the configuration
    tests/test_multiseq_gpu.py::test_batch_loop_8_seeds_640x480_against_the_oracle holds against the oracle sequence by sequence.  From
    S = 64 on every (seed, motion) pair exists more than once: such replicas must come out bit-identical (same inputs through different
    slices of the same launches) — checked here.  S = 64 / 128 say where frames/s saturates."""
    from lsd_slam_amd.driver import DriverLoopBatch
    from lsd_slam_amd import capi
    n = seeds[0][0].shape[0]
    NS = len(seeds)
    motions = [(1, 1), (-1, 1), (1, 2), (-1, 2)]
    out = {"workload": "S sequences x 640x480 track+map loop on one GPU (BASELINE.json configs[3], sequences > GPUs): scene S1 with %d seeds x 4 camera "
                       "motions (sequence s: seed s %% %d, motion (s // %d) %% 4; S = 8: the 8 seeds, camera forwards), new keyframe every %d frames (phases 0 / 2 / 5 / 7 "
                       "by s %% 4), %d timed steps (= frames per sequence)" % (NS, NS, NS, KF_EVERY, steps)}
    L = capi.lib()
    for S in sizes:
        mo = lambda s: motions[(s // NS) % 4]
        idx = lambda s, t: (mo(s)[0] * mo(s)[1] * t) % n
        ptrs = lambda t: [seeds[s % NS][0][idx(s, t)].data_ptr() for s in range(S)]
        rec = {"sequences": S}
        try:
            for pipelined in (True, False):
                loop = DriverLoopBatch(w, h, K, ptrs(0), [seeds[s % NS][1] for s in range(S)], kf_every=KF_EVERY, images_on_device=True, device=device)
                loop.set_keyframe_phases([((s % 4) * KF_EVERY) // 4 for s in range(S)])   # independent cameras do not change keyframe in the same step
                loop.set_pipeline(pipelined)
                if os.environ.get("LSD_COARSE_MIN_JOBS"):      # developer A/B (tools/bench_multiseq.py): sequences per step from which the coarse levels run in one workgroup each
                    loop.set_coarse_min_jobs(int(os.environ["LSD_COARSE_MIN_JOBS"]))
                ctx_h = loop.ctx_handle()
                t = 1
                loop.run([ptrs(t + k) for k in range(warmup)])
                t += warmup
                capi.check(L.lsdhip_ctx_synchronize(ctx_h))
                dts, poses = [], None
                for _ in range(regions):
                    batch = [ptrs(t + k) for k in range(steps)]
                    t0 = time.perf_counter()
                    done, poses = loop.run(batch, want_poses=True)
                    capi.check(L.lsdhip_ctx_synchronize(ctx_h))
                    dts.append(time.perf_counter() - t0)
                    t += steps
                st = loop.stats()
                dropped = sum(loop.dropped())
                roof = None
                if pipelined and S <= 32:
                    try:
                        roof = multi_seq_roofline(loop, L, ctx_h, ptrs, t, S, w, h, steps)
                    except Exception as e:
                        roof = {"error": str(e)}
                loop.close()
                dt = float(np.median(dts))
                fps = S * steps / dt
                if not pipelined:
                    rec["frames_s_block_until_mapped"] = fps
                    rec["ms_per_step_block_until_mapped"] = dt / steps * 1e3
                    continue
                # replicas: the same (seed, motion, keyframe phase) pair again, 4 NS sequences further on
                period = 4 * NS
                identical = all(np.array_equal(poses[:, s], poses[:, s % period]) for s in range(S)) if S > period else None
                rec.update({"frames_s": fps, "ms_per_step": dt / steps * 1e3, "us_per_frame": dt / steps / S * 1e6,
                            "vs_one_sequence": (fps / single_value) if single_value else None,
                            "tracked_good": sum(q["tracked_good"] for q in st), "frames": sum(q["frames"] for q in st),
                            "updates": sum(q["updates"] for q in st), "keyframes": sum(q["keyframes"] for q in st), "dropped": dropped,
                            "lost": sum(q["lost"] for q in st),
                            "lm_evaluations_per_frame": sum(q["evaluations"] for q in st) / max(1, sum(q["frames"] for q in st)),
                            "distinct_sequences": min(S, period), "replicas_bit_identical": identical, "roofline": roof})
        except Exception as e:     # (a size the box cannot hold must not take the smaller ones down)
            rec["error"] = str(e)
        out["S%d" % S] = rec
    return out


# algorithmic bytes per map pixel of the shared launches of the S-sequence loop (unique bytes read + written):
#   frame pyramids : k_image_pyramid_batch (1 B in, five image levels 4 x 1.332 B) + k_gradients_max_batch (image levels 1-4 in 1.33 B,
#                    texels of levels 1-4 out 5.31 B)                                                                            = 13.0 B
#                    (until round 6 also the level-0 texels 16 B, |grad| 4 + 4 B and maxGradients 4 B of EVERY frame = 41.0 B; these are
#                    keyframe planes now, built when a frame becomes a DepthMap's keyframe — 32 B per pixel of a new keyframe, queued ahead
#                    of the keyframe change's bracket and inside keyframe_ms)
#   regularise     : K5 ~34 + K6 30 (+ K8 17 when the pass carries Frame::setDepth; counted without)                           = 64.0 B
#   idepth pyramids: level-0 (idepth, var) in 8 B, levels 1-4 out 2.66 B                                                         = 10.7 B
#   keyframe change: finalize pass 64 + setDepth 17 + re-activation data 9 + candidates 16 | merge 20 + 29 | regularise(occ) 30 | fill +
#                    regularise 64 | rescale + setDepth 41 | two idepth pyramids 21.3                                            = 311 B
#   observe        : 13 B per map pixel (the cheap rejections) + 57 B per searched pixel + 4 B per reference-image sample (steps + 4)
MS_BYTES_PER_PX = {"frame_pyramids": 13.0, "regularise": 64.0, "idepth_pyramids": 10.7, "keyframe_change": 311.0}


def multi_seq_roofline(loop, L, ctx_h, ptrs, t0, S, w, h, steps):
    """Where the shared launches of the S-sequence loop sit against the HBM roofline: a profiled leg after the timed regions (HIP events
    around every third call of each batched entry and around the tracking batches' launch budgets; the events cost time, so this leg is
    not what frames_s is measured on).  Algorithmic bytes per launch / event time per launch, per kind."""
    import ctypes as C
    from lsd_slam_amd import capi
    capi.check(L.lsdhip_prof_reset(ctx_h))
    capi.check(L.lsdhip_prof_enable(ctx_h, 1))
    loop.run([ptrs(t0 + k) for k in range(steps)])
    capi.check(L.lsdhip_ctx_synchronize(ctx_h))
    capi.check(L.lsdhip_prof_enable(ctx_h, 0))
    ms_, n_, b_ = C.c_double(), C.c_longlong(), C.c_double()
    capi.check(L.lsdhip_prof_read(ctx_h, C.byref(ms_), C.byref(n_), C.byref(b_)))
    ms = (C.c_double * 5)()
    calls = (C.c_longlong * 5)()
    units = (C.c_double * 5)()
    obs = (C.c_double * 3)()
    capi.check(L.lsdhip_ctx_batch_prof_read(ctx_h, ms, calls, units, obs))

    def rec(kernel, ms_tot, n, bytes_tot, **kw):
        if n <= 0 or ms_tot <= 0:
            return None
        ach = bytes_tot / (ms_tot * 1e-3) / 1e9
        d = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
             "avg_launch_us": ms_tot / n * 1e3, "algorithmic_bytes_per_launch": bytes_tot / n, "launches_sampled": int(n)}
        d.update(kw)
        return d

    out = {"note": "profiled leg of %d steps after the timed regions; bytes are algorithmic (SURVEY.md 8(d)), times HIP events on the launches' stream" % steps}
    out["track_batch"] = rec("k_track_step<TS_LM> + k_track_step<TS_EVAL>: all rounds of a tracking batch (one bracket per batch)", ms_.value,
                             max(1, steps), b_.value, evaluations=int(n_.value), note="per batch of %d jobs, not per launch" % S)
    names = [("frame_pyramids", 0, "k_image_pyramid_batch + k_gradients_max_batch"), ("regularise", 2, "k_reg_fused_batch (fill holes + regularise [+ setDepth])"),
             ("idepth_pyramids", 3, "k_idepth_pyramid_batch"), ("keyframe_change", 4, "the six launches of lsdhip_depth_change_keyframe_batch")]
    for key, k, kern in names:
        out[key] = rec(kern, ms[k], calls[k], MS_BYTES_PER_PX[key] * units[k], maps_per_launch=(units[k] / calls[k] / (w * h)) if calls[k] else None)
    if calls[1] > 0 and obs[0] > 0:
        srch, stp = obs[1] / obs[0], obs[2] / obs[0]
        px = units[1] / calls[1]
        act = 13.0 * px + 57.0 * srch + 4.0 * (stp + 4.0 * srch)
        out["observe"] = rec("k_observe_select_batch + k_observe_walk_batch", ms[1], calls[1], act * calls[1], maps_per_launch=px / (w * h),
                             searched_pixels_per_launch=srch, searched_fraction=srch / px, walk_steps_per_launch=stp,
                             upper_bound_bytes_per_launch=78.0 * px, stereo_steps_per_s=stp / (ms[1] / calls[1] * 1e-3))
    return out


def extra_configs(la, torch, device):
    """Two short legs after the timed region (not the bench metric): BASELINE.json configs[2] — the full track+map loop at 1280x1024 on
    the edge-texture scene S2 — and configs[4] — the 3840x2160 depth regularisation (fill holes + regularise fused), full frame and as
    8 row bands with halo refreshes on this one GPU through the C++ band loop."""
    import ctypes as C
    from lsd_slam_amd import synth, capi
    from lsd_slam_amd.driver import DriverLoop
    out = {}
    # ---- configs[2]: 1280x1024, scene S2 ---------------------------------------------------------------------------------------
    w, h, n = 1280, 1024, 12
    frames, depth0, K, gt = synth.make_sequence(w, h, n, kind="S2")
    d_frames = torch.from_numpy(frames).cuda(device)
    order = list(range(n)) + list(range(n - 2, 0, -1))
    ptr = lambda i: d_frames[order[i % len(order)]].data_ptr()
    loop = DriverLoop(w, h, K, ptr(0), depth0, kf_every=KF_EVERY, images_on_device=True, device=device)
    L = capi.lib()
    ctx_h = loop.ctx_handle()
    loop.run([ptr(1 + k) for k in range(12)])
    loop.reset_stats()
    capi.check(L.lsdhip_prof_reset(ctx_h))
    capi.check(L.lsdhip_prof_enable(ctx_h, 1))
    capi.check(L.lsdhip_ctx_synchronize(ctx_h))
    steps = 66
    t0 = time.perf_counter()
    loop.run([ptr(13 + k) for k in range(steps)])
    capi.check(L.lsdhip_ctx_synchronize(ctx_h))
    dt = time.perf_counter() - t0
    capi.check(L.lsdhip_prof_enable(ctx_h, 0))
    st = loop.stats()
    ms_, n_, b_ = C.c_double(), C.c_longlong(), C.c_double()
    capi.check(L.lsdhip_prof_read(ctx_h, C.byref(ms_), C.byref(n_), C.byref(b_)))
    roof = None
    if n_.value > 0 and ms_.value > 0:
        ach = (b_.value / n_.value) / (ms_.value / n_.value * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "k_track_step", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "avg_launch_us": ms_.value / n_.value * 1e3, "algorithmic_bytes_per_launch": b_.value / n_.value, "launches": int(n_.value)}
    out["s2_1280x1024"] = {"workload": "1280x1024 track+map loop, edge-texture scene S2 (BASELINE.json configs[2]), %d timed frames" % steps,
                           "frames_s": steps / dt, "ms_per_step": dt / steps * 1e3, "tracked_good": int(st.tracked_good), "frames": int(st.frames),
                           "lm_evaluations_per_frame": st.evaluations / max(1, st.frames),
                           "track_launches_per_frame": st.track_launches / max(1, st.frames),
                           "depth_mpix_per_s": (w * h * st.updates) / st.seconds_map / 1e6 if st.seconds_map > 0 else None,
                           "roofline": roof}
    loop.close()
    del d_frames
    # ---- configs[4]: 3840x2160 regularisation, full frame and 8 row bands on this one GPU -------------------------------------------
    from lsd_slam_amd.bands import NativeBandRegularizer, synth_s3
    W, H, passes = 3840, 2160, 20
    hyp, maxgrad = synth_s3(W, H)
    reg = {"workload": "3840x2160 depth-map regularisation (fill holes + regularise fused per pass), scene S3 (BASELINE.json configs[4]), "
                       "%d passes; 64 algorithmic B/px (SURVEY.md 8(d): K5 ~34 + K6 30)" % passes}
    for bands in (1, 8):
        br = NativeBandRegularizer(W, H, bands, list(range(bands)), device=device)
        br.load(hyp, maxgrad)
        br.run(2)
        br.synchronize()
        br.load(hyp, maxgrad)
        br.synchronize()
        dts = []
        for _ in range(3):          # median of three regions of `passes` passes (the first one after a load can carry one-time costs)
            t0 = time.perf_counter()
            br.run(passes)
            br.synchronize()
            dts.append(time.perf_counter() - t0)
        dt = float(np.median(dts))
        ach = W * H * passes * 64.0 / dt / 1e9
        key = "full_frame" if bands == 1 else "bands_%d_one_gpu" % bands
        reg[key] = {"ms_per_pass": dt / passes * 1e3, "mpix_per_s": W * H * passes / dt / 1e6, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS}
        if bands > 1:
            reg[key]["halo_bytes_per_pass"] = br.plan.halo_bytes_per_pass(W)
        br.close() if hasattr(br, "close") else None
    reg["bands_vs_full_frame"] = reg["bands_8_one_gpu"]["ms_per_pass"] / reg["full_frame"]["ms_per_pass"]
    out["reg_3840x2160"] = reg
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=250)
    ap.add_argument("--no-roofline-events", action="store_true")
    ap.add_argument("--no-throughput-mode", action="store_true", help="skip the extra 64-jobs-per-launch measurement")
    ap.add_argument("--scene", type=str, default="S1", choices=["S1", "S2"],
                    help="synthetic scene (SURVEY.md §8(d)): S1 smooth texture (configs[1]); S2 Voronoi edge texture (configs[2], ~23 %% semi-dense)")
    ap.add_argument("--live-queue", type=int, default=1,
                    help="frames handed to updateKeyframe per mapping iteration (1 = blockUntilMapped as BASELINE configs[1]; K > 1 "
                         "restates live operation, C/SlamSystem.cpp:559-571)")
    ap.add_argument("--trials", type=int, default=-1, help="LM retries evaluated per k_track_step launch (lsdhip_tracker_set_speculation); -1 = library default")
    ap.add_argument("--trial-cap", type=int, default=0, help="workgroups per trial at the finest level (0 = library default)")
    ap.add_argument("--no-pcie-leg", action="store_true", help="skip the extra host-image (PCIe-inclusive) measurement")
    ap.add_argument("--block-until-mapped", action="store_true",
                    help="one stream: every frame's mapping iteration runs behind its tracking and the next frame waits for it "
                         "(SlamSystem's blockUntilMapped, C/SlamSystem.cpp:1026-1040).  Default: pipelined — tracking stream beside "
                         "mapping stream, the mapper one frame behind (the reference's two threads; lsd_slam_hip::SlamLoop)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / process-group plumbing only (no GPU work, gloo when there is no GPU): prints the JSON skeleton")
    ap.add_argument("--repeats", type=int, default=0,
                    help="the timed region of exactly --steps steps is run this many times back to back (each between barriers); `value` and "
                         "`ms_per_step` are those of the MEDIAN region, so one 5 ms window does not decide the line.  0 = automatic: at least 5 "
                         "regions and at least ~1000 frames in total (the loop shows ~15 %% slower phases lasting tens of milliseconds on some "
                         "boxes: profiles/r03_notes.md)")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the two short extra legs (BASELINE.json configs[2] and configs[4])")
    ap.add_argument("--seq-frames", type=int, default=50,
                    help="synthetic frames to render (one camera revolution = 50); fewer are played back and forth")
    args = ap.parse_args()

    # --gpus N without a launcher: start the N ranks ourselves (one process per GPU, torch.distributed.run on this node)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    # ONE JSON line on stdout: everything the libraries print meanwhile (RCCL's version banner goes to stdout) is sent to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(line)
        sys.stdout.flush()
        os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LSD_FORCE_DIST=1: run the RCCL path (process group, keyframe gather, barriers) even with one rank — a one-GPU check of the N > 1 code
    distributed = world > 1 or os.environ.get("LSD_FORCE_DIST") == "1"
    if args.dry_run:
        # plumbing check of the N-rank path: rendezvous, barrier, max-over-ranks reduction, one JSON line from rank 0
        if distributed:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="nccl" if torch.cuda.is_available() else "gloo")
            t = torch.tensor([float(rank + 1)], dtype=torch.float64, device="cuda" if torch.cuda.is_available() else "cpu")
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            assert int(t.item()) == world
        if rank == 0:
            emit(json.dumps({"metric": "tracked frames/sec (full track+map loop) at %dx%d, frames resident in HBM (8-bit upload excluded: value_pcie_inclusive has it)" % (args.width, args.height), "value": None,
                             "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                             "scaling": "weak",
                             # what a real N-rank run adds to the line: every rank's own frames/s (configs[3]: a straggler shows) and the
                             # row-band regulariser over the ranks (configs[4], RCCL halo exchange inside the C++ loop)
                             "per_rank_frames_s": [None] * world if distributed else None,
                             "extra_configs": {"reg_3840x2160_bands": {"rccl": None, "ipc": None}} if distributed else
                                              {"multi_seq": {"S32": {"frames_s": None, "roofline": {k: None for k in ("track_batch", "frame_pyramids", "observe", "regularise", "idepth_pyramids", "keyframe_change")}}}},
                             # like-for-like CPU comparison (one-GPU runs): the reference's sources in both execution models
                             "cpu_baseline": None if distributed else {"value": None, "updates_per_frame": None, "pipelined": {"value": None, "updates_per_frame": None}},
                             "speedup_vs_cpu_baseline": None if distributed else {"block_until_mapped": None, "pipelined": None, "updates_per_frame": {"gpu": None, "cpu": None}}}))
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: there is no CPU fallback for the hot path")
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:   # LSD_FORCE_DIST=1 without a launcher
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import lsd_slam_amd as la
    from lsd_slam_amd import synth

    w, h = args.width, args.height
    n_seq = args.seq_frames
    frames, depth0, K, gt = synth.make_sequence(w, h, n_seq, seq_index=rank, kind=args.scene)
    d_frames = torch.from_numpy(frames).cuda(local_rank)           # inputs resident in HBM
    torch.cuda.synchronize()
    if n_seq >= 50:
        order = list(range(n_seq))                                  # a closed camera loop
    else:
        order = list(range(n_seq)) + list(range(n_seq - 2, 0, -1))  # back and forth: no jump at the wrap-around
    ptr = lambda i: d_frames[order[i % len(order)]].data_ptr()

    # the frame loop itself runs in C++ (liblsdhip_driver.so, include/lsd_slam_hip.hpp): Python only hands over
    # batches of device pointers and, for N > 1, gathers the batch's finished keyframes afterwards
    from lsd_slam_amd.driver import DriverLoop
    from lsd_slam_amd import capi
    loop = DriverLoop(w, h, K, ptr(0), depth0, kf_every=KF_EVERY, images_on_device=True, device=local_rank)
    pipelined = not args.block_until_mapped
    if pipelined:
        loop.set_pipeline(True)
    if args.trials > 0:
        loop.set_speculation(args.trials, args.trial_cap)
    if args.live_queue > 1:
        loop.set_live_queue(args.live_queue)
    loop.keep_keyframes(True)          # validation: the rescale factor of every keyframe, read back after the timed region
    L = capi.lib()
    ctx_h = loop.ctx_handle()
    # frame-memory pool: the validation keeps every keyframe of the run alive, i.e. the loop takes a fresh arena per keyframe — allocated
    # ahead (lsdhip_ctx_reserve_frames), not with one hipMalloc (0.5 ms) per keyframe inside the timed region
    repeats = args.repeats if args.repeats > 0 else max(5, -(-1000 // max(1, args.steps)))
    capi.check(L.lsdhip_ctx_reserve_frames(ctx_h, (args.warmup + args.steps * repeats) // KF_EVERY + 24))

    # Finished keyframes leave the loop through a device ring (smoothed idepth + variance planes, copied on the loop's
    # stream); after every batch of frames the ring's new entries are collected on rank 0 with ONE gather (RCCL over xGMI) —
    # the only exchange step of the path.  With one rank the export still runs, so per-GPU work is the same for every N.
    from lsd_slam_amd.multigpu import gather_keyframe_ring
    RING = 32
    ring = torch.empty((RING, 2, h, w), dtype=torch.float32, device=torch.device("cuda", local_rank))
    loop.set_keyframe_ring(ring.data_ptr(), RING)
    # The gather itself is issued by the C++ loop: RCCL send/recv on the loop's own stream, no host synchronisation
    # (lsdloop_gather_keyframes).  torch.distributed only carries the rendezvous (the 128-byte ncclUniqueId), the barriers and the
    # max-over-ranks of the timing.  LSD_GATHER=torch (or a failed communicator) falls back to a torch.distributed gather.
    gather_impl = "none"
    recv = None
    recv_all = None
    if distributed:
        gather_impl = "torch.distributed"
        if os.environ.get("LSD_GATHER", "rccl") != "torch":
            try:
                uid = [DriverLoop.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                loop.comm_init(uid[0], rank, world)
                gather_impl = "rccl (C++ loop, ncclSend/ncclRecv on the loop's stream)"
            except Exception as e:   # keep the run alive on the fallback
                sys.stderr.write("RCCL gather in the C++ loop unavailable (%s): falling back to torch.distributed\n" % e)
                gather_impl = "torch.distributed"
        ok = torch.tensor([1 if gather_impl.startswith("rccl") else 0], device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)     # all ranks or none
        if int(ok.item()) == 0:
            gather_impl = "torch.distributed"
        if gather_impl.startswith("rccl"):
            recv_all = torch.empty((world,) + tuple(ring.shape), dtype=torch.float32, device=ring.device) if rank == 0 else None
        else:
            recv = [torch.empty_like(ring) for _ in range(world)] if rank == 0 else None
    state = {"fi": 0, "exported": 0, "gathered_bytes": 0, "poses": []}

    def run_frames(n):
        """n frames through the C++ loop in batches that produce at most RING keyframes; one gather per batch"""
        left = n
        while left > 0:
            m = min(left, RING * KF_EVERY)
            ptrs = [ptr(state["fi"] + 1 + k) for k in range(m)]
            loop.set_keyframe_ring(ring.data_ptr(), RING)       # the batch's keyframes land in ring[0:new]
            done, poses = loop.run(ptrs, want_poses=True)       # returns with the stream drained: the ring is complete
            state["poses"].append(poses)
            state["fi"] += done
            left -= done
            new = loop.keyframes_exported()
            state["exported"] += new
            if distributed and gather_impl.startswith("rccl"):
                loop.gather_keyframes(new, 0, recv_all.data_ptr() if rank == 0 else None, ring.numel())
                state["gathered_bytes"] += new * ring[0].numel() * 4
            elif distributed:
                state["gathered_bytes"] += gather_keyframe_ring(ring, new, recv)

    run_frames(args.warmup)

    # timed region: exactly K steps between barrier + synchronize
    loop.reset_stats()
    if not args.no_roofline_events:
        capi.check(L.lsdhip_prof_reset(ctx_h))
        capi.check(L.lsdhip_prof_enable(ctx_h, 1))
    capi.check(L.lsdhip_ctx_synchronize(ctx_h))
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    clock = time.perf_counter
    elapsed_all = []
    own_all = []
    region_wall = []
    for rep_i in range(repeats):
        region_wall.append(time.time())
        # every region: barrier + synchronize, exactly K steps, synchronize + barrier; max over ranks
        t0 = clock()
        run_frames(args.steps)
        capi.check(L.lsdhip_ctx_synchronize(ctx_h))
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        e = clock() - t0
        own_all.append(e)
    if distributed:
        # every region is bracketed by barriers on all ranks; its time is the MAX over ranks — reduced once, after the last region, so
        # that a region of K = 20 frames (3.5 ms) is not stretched by a second collective and a host read-back
        t = torch.tensor(own_all, dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_all = [float(x) for x in t.tolist()]
    else:
        elapsed_all = list(own_all)
    capi.check(L.lsdhip_prof_enable(ctx_h, 0))
    elapsed = float(np.median(elapsed_all))
    # configs[3]: every rank's own rate (its own median region, not the max over ranks that `value` is made of)
    per_rank = None
    bands_rec = None
    if distributed:
        mine = torch.tensor([args.steps / float(np.median(own_all))], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [float(x.item()) for x in allr]
        try:
            bands_rec = bands_distributed(torch, dist, rank, world, local_rank)
        except Exception as e:   # an extra: never let it take the bench line down
            bands_rec = {"error": str(e)}

    st = loop.stats()
    obs_ms, obs_calls = loop.observe_time()
    obs_counted, obs_searched, obs_steps = loop.observe_work()
    # ---- self-validation of the run that was just timed (nothing below is inside the timed region) ----------------------------------------
    poses_all = np.concatenate(state["poses"]) if state["poses"] else np.zeros((0, 7))
    kf_scales, kf_points = loop.keyframe_log()
    rmse, n_kf = trajectory_rmse(poses_all, kf_scales, lambda i: gt[order[(i + 1) % len(order)]], args.warmup, lag=1 if pipelined else 0)
    import ctypes as C
    ms_, n_, b_ = C.c_double(), C.c_longlong(), C.c_double()
    capi.check(L.lsdhip_prof_read(ctx_h, C.byref(ms_), C.byref(n_), C.byref(b_)))
    res_ms, res_launches, res_bytes = ms_.value, n_.value, b_.value
    out = None
    if rank == 0:
        value = world * args.steps / elapsed
        roofline = None
        if res_launches > 0 and res_ms > 0:
            achieved = (res_bytes / res_launches) / (res_ms / res_launches * 1e-3) / 1e9
            # the committed PMC pass was taken on the default workload; other sizes / scenes report no counter traffic
            traffic, traffic_src = pmc_traffic() if ((w, h) == (640, 480) and args.scene == "S1") else (None, None)
            roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                        "kernel": "k_track_step (LM step + fused K0+K1+K2+K3 residual evaluation)", "launches": int(res_launches), "launches_note": "HIP events bracket every 8th trackFrame call of the timed region",
                        "avg_launch_us": res_ms / res_launches * 1e3,
                        "algorithmic_bytes_per_launch": res_bytes / res_launches}
        out = {
            "metric": "tracked frames/sec (full track+map loop) at %dx%d, frames resident in HBM (8-bit upload excluded: value_pcie_inclusive has it)" % (w, h),
            "value": value, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "repeats": repeats, "ms_per_step_all_regions": [e / args.steps * 1e3 for e in elapsed_all],
            "region_start_unix_s": region_wall,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%dx%d 5-level-pyramid SE3 tracking + semi-dense depth update, single sequence per GPU "
                                   "(BASELINE.json configs[%d]); synthetic scene %s, GT-depth init, new keyframe every %d frames"
                                   % (w, h, 1 if (w, h) == (640, 480) else 2, args.scene, KF_EVERY),
                       "width": w, "height": h, "parallelism": "1 sequence per GPU; finished keyframes' depth planes exported to a device ring and gathered to rank 0 once per batch",
                       "gather": gather_impl},
            "track_fps": st.frames / st.seconds_track if st.seconds_track > 0 else None,
            "depth_mpix_per_s": (w * h * st.updates) / st.seconds_map / 1e6 if st.seconds_map > 0 else None,
            "keyframe_ms": st.seconds_keyframe / st.keyframes * 1e3 if st.keyframes > 0 else None,
            "lm_evaluations_per_frame": st.evaluations / max(1, st.frames),
            "track_launches_per_frame": st.track_launches / max(1, st.frames),
            "lm_evaluations_per_frame_by_level": {"L%d" % l: st.level_evaluations[l] / max(1, st.frames) for l in (4, 3, 2, 1)},
            # self-validation: a loop that diverged or drifted shows here, not only in frames/s
            "validation": {"frames": int(st.frames), "tracked_good": int(st.tracked_good), "keyframes": int(st.keyframes),
                           "rmse_vs_gt": rmse, "rmse_unit": "scene units (camera circle radius 0.08; depth ~2)",
                           "semidense_px": float(np.mean(kf_points)) if len(kf_points) else None,
                           "semidense_px_last": int(kf_points[-1]) if len(kf_points) else None,
                           "ok": bool(st.tracked_good == st.frames and rmse is not None and rmse < 5e-3 and len(kf_points) > 0 and np.min(kf_points) > 0.05 * w * h)},
            "live_queue": args.live_queue,
            "execution": {"model": "pipelined" if pipelined else "blockUntilMapped",
                          "description": ("tracking stream beside mapping stream, the mapper one frame behind the tracker (the reference's two threads, "
                                          "blockUntilMapped == false, C/SlamSystem.cpp:1026-1040, :559-566, :907-920)") if pipelined else
                                         "one stream: every frame mapped before the next one is tracked (C/SlamSystem.cpp:1026-1040 with the wait)",
                          "updates_per_frame": st.updates / max(1, st.frames), "keyframes_per_frame": st.keyframes / max(1, st.frames),
                          "dropped_frames_per_frame": st.dropped / max(1, st.frames),
                          "dropped_note": "frames tracked on the keyframe the mapper had just replaced: tracked, not mapped (pipelined only)"},
            "roofline": roofline,
        }
        if per_rank is not None:
            out["per_rank_frames_s"] = per_rank
        if bands_rec is not None:
            out.setdefault("extra_configs", {})["reg_3840x2160_bands"] = bands_rec
        if world == 1 and (w, h) == (640, 480):
            try:
                key = "value_block_until_mapped" if pipelined else "value_pipelined"
                out[key] = other_execution_model(DriverLoop, capi, w, h, K, ptr, depth0, local_rank, not pipelined, min(args.steps, 200), args.warmup,
                                                 args.trials, args.trial_cap)
            except Exception as e:
                out["value_block_until_mapped" if pipelined else "value_pipelined"] = {"error": str(e)}
        if obs_calls > 0 and obs_ms > 0:
            # K4 algorithmic bytes: hypothesis 29 read + <= 29 written, maxGradients 4, keyframe image 4, keyframe gradients 8,
            # reference image 4 per map pixel = 78 B/px upper bound (x live_queue reference images is not counted)
            obs_us = obs_ms / obs_calls * 1e3
            ach = 78.0 * w * h / (obs_us * 1e-6) / 1e9
            out["roofline_depth"] = {"bound": "hbm", "kernel": "k_observe (makeAndCheckEPL + doLineStereo + observeDepthCreate/Update, one launch per updateKeyframe)",
                                     "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "avg_launch_us": obs_us,
                                     "launches": int(obs_calls), "algorithmic_bytes_per_launch": 78.0 * w * h,
                                     "note": "upper-bound bytes (every pixel assumed to carry a hypothesis); HIP events around every 7th launch"}
            obs_traffic, obs_traffic_src = pmc_traffic("k_observe") if ((w, h) == (640, 480) and args.scene == "S1") else (None, None)
            if obs_counted > 0:
                # the same launches with the bytes of the pixels that were actually searched (counted in-kernel: searches = doLineStereo
                # calls, steps = sum of their loopCounter): every pixel 13 B (isValid, maxGradients, blacklisted, nextStereoFrameMinID: the
                # cheap rejections); a searched pixel + 16 B (rest of its hypothesis) + <= 29 B written + 4 B keyframe image + 8 B keyframe
                # gradients + 4 B reference image per sample (steps + 4 per search)
                srch, stp = obs_searched / obs_counted, obs_steps / obs_counted
                act_bytes = 13.0 * w * h + 57.0 * srch + 4.0 * (stp + 4.0 * srch)
                ach_a = act_bytes / (obs_us * 1e-6) / 1e9
                out["roofline_depth"]["active"] = {"algorithmic_bytes_per_launch": act_bytes, "achieved": ach_a, "frac": ach_a / HBM_PEAK_GBS,
                                                   "searched_pixels_per_launch": srch, "searched_fraction": srch / (w * h),
                                                   "walk_steps_per_launch": stp, "walk_steps_per_search": stp / max(1.0, srch),
                                                   "launches_counted": int(obs_counted)}
                out["stereo_steps_per_s"] = stp / (obs_us * 1e-6)
                # the object's own achieved / frac are those of the ACTIVE bytes (what the searched pixels need); the every-pixel-carries-a-
                # hypothesis bound moves under `upper_bound`; `traffic` = HBM bytes per launch of the committed PMC passes, beside its ratio
                rd = out["roofline_depth"]
                rd["upper_bound"] = {"algorithmic_bytes_per_launch": rd["algorithmic_bytes_per_launch"], "achieved": rd["achieved"], "frac": rd["frac"], "unit": "GB/s",
                                     "note": rd.pop("note")}
                rd.update({"achieved": ach_a, "frac": ach_a / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": act_bytes, "traffic": obs_traffic,
                           "traffic_source": obs_traffic_src, "traffic_vs_algorithmic": (obs_traffic / act_bytes) if obs_traffic else None,
                           "note": "algorithmic bytes of the pixels that were actually searched (counted in-kernel on the sampled launches); HIP events around every 7th launch"})
        if world == 1 and (w, h) == (640, 480) and not args.no_throughput_mode:
            try:
                out["roofline_throughput_mode"] = throughput_mode(la, w, h, K, d_frames, depth0, local_rank)
            except Exception as e:   # an extra: never let it take the bench line down
                out["roofline_throughput_mode"] = {"error": str(e)}
        if world == 1 and (w, h) == (640, 480) and not args.no_extra_configs:
            # (before the dense 4K leg: a burst like that leaves a box in its slow state for seconds, profiles/r03_notes.md section 1)
            try:
                # the 8 seeds of configs[3]: this rank's sequence + 7 more, rendered here (outside every timed region)
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(7) as ex:
                    more = list(ex.map(lambda q: synth.make_sequence(w, h, n_seq, seq_index=q, kind=args.scene), range(1, 8)))
                seeds = [(d_frames, depth0)] + [(torch.from_numpy(fr).cuda(local_rank), d0) for fr, d0, _, _ in more]
                out.setdefault("extra_configs", {})["multi_seq"] = multi_seq(torch, w, h, K, seeds, local_rank, out["value"])
                del seeds, more
            except Exception as e:
                out.setdefault("extra_configs", {})["multi_seq"] = {"error": str(e)}
            try:
                out.setdefault("extra_configs", {}).update(extra_configs(la, torch, local_rank))
            except Exception as e:   # extras: never let them take the bench line down
                out.setdefault("extra_configs", {})["error"] = str(e)
        if world == 1 and not args.no_pcie_leg:
            # SURVEY.md §8(d) counts the 8-bit upload inside tracked frames/s: the same loop fed from (pinned) host memory, every frame
            # crossing PCIe inside lsdhip_frame_create.  Reported beside `value`, never as `value`.
            try:
                h_frames = torch.from_numpy(frames).pin_memory()
                hptr = lambda i: h_frames[order[i % len(order)]].data_ptr()
                loop2 = DriverLoop(w, h, K, hptr(0), depth0, kf_every=KF_EVERY, images_on_device=False, device=local_rank)
                if pipelined:
                    loop2.set_pipeline(True)     # the next frame's upload is queued on the mapping stream, beside the tracking job
                if args.trials > 0:
                    loop2.set_speculation(args.trials, args.trial_cap)
                n2 = min(args.steps, 200)
                loop2.run([hptr(1 + k) for k in range(args.warmup)])
                capi.check(L.lsdhip_ctx_synchronize(ctx_h))
                dts2, fi2 = [], 1 + args.warmup
                for _ in range(5):
                    t1 = clock()
                    loop2.run([hptr(fi2 + k) for k in range(n2)])
                    capi.check(L.lsdhip_ctx_synchronize(ctx_h))
                    dts2.append(clock() - t1)
                    fi2 += n2
                dt2 = float(np.median(dts2))
                out["value_pcie_inclusive"] = {"value": n2 / dt2, "unit": "frames/s", "steps": n2, "regions": 5, "vs_value": (n2 / dt2) / out["value"],
                                               "note": "frames start in pinned host memory; upload (%d KB per frame) inside the timed loop%s"
                                                       % (w * h // 1024, ", queued asynchronously on the mapping stream while the previous frame is tracked "
                                                          "(lsdhip_frame_create_async)" if pipelined else "")}
                loop2.close()
            except Exception as e:
                out["value_pcie_inclusive"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], other = cpu_baseline(frames, depth0, K, args.cpu_frames, w, h)
            if other is not None:
                out["cpu_baseline_other"] = other
            # like for like: pipelined / pipelined and blockUntilMapped / blockUntilMapped, the mapping iterations per frame of both sides beside it
            cb = out["cpu_baseline"]
            pip = out["value"] if pipelined else (out.get("value_pipelined") or {}).get("value")
            blk = (out.get("value_block_until_mapped") or {}).get("value") if pipelined else out["value"]
            gpu_upd = {"pipelined": out["execution"]["updates_per_frame"] if pipelined else None,
                       "block_until_mapped": None if pipelined else out["execution"]["updates_per_frame"]}
            for k, rec in (("pipelined", out.get("value_pipelined")), ("block_until_mapped", out.get("value_block_until_mapped"))):
                if isinstance(rec, dict) and rec.get("frames"):
                    gpu_upd[k] = rec["updates"] / rec["frames"]
            cpu_pip = (cb.get("pipelined") or {}).get("value")
            out["speedup_vs_cpu_baseline"] = {
                "block_until_mapped": (blk / cb["value"]) if blk else None,
                "pipelined": (pip / cpu_pip) if (pip and cpu_pip) else None,
                "updates_per_frame": {"gpu": gpu_upd, "cpu": {"block_until_mapped": cb.get("updates_per_frame"),
                                                              "pipelined": (cb.get("pipelined") or {}).get("updates_per_frame")}},
                "note": "same execution model on both sides (cpu_baseline.value: sequential track-then-map; cpu_baseline.pipelined: tracking thread beside mapping thread, mapper one frame behind)"}
        else:
            out["cpu_baseline"] = None
        # compact copies of the figures the other objects of this line hold, INSIDE `roofline` (a reader that keeps only `roofline` /
        # `cpu_baseline` / `config` of the line still gets them): fractions are of the 8 TB/s HBM peak, algorithmic bytes / measured time
        if out.get("roofline") is not None:
            def dig(d, *keys):
                for k in keys:
                    if not isinstance(d, dict) or d.get(k) is None:
                        return None
                    d = d[k]
                return d
            ms = dig(out, "extra_configs", "multi_seq") or {}
            others = {
                "throughput_mode.level1_evaluation.frac": dig(out, "roofline_throughput_mode", "level1_evaluation", "frac"),
                "throughput_mode.level1_evaluation.us_per_launch": dig(out, "roofline_throughput_mode", "level1_evaluation", "us_per_launch"),
                "throughput_mode.whole_batches.frac": dig(out, "roofline_throughput_mode", "frac"),
                "depth.k_observe.active_bytes.frac": dig(out, "roofline_depth", "frac"),
                "reg_3840x2160.full_frame.frac": dig(out, "extra_configs", "reg_3840x2160", "full_frame", "frac"),
                "reg_3840x2160.full_frame.ms_per_pass": dig(out, "extra_configs", "reg_3840x2160", "full_frame", "ms_per_pass"),
                "reg_3840x2160.bands_vs_full_frame": dig(out, "extra_configs", "reg_3840x2160", "bands_vs_full_frame"),
                "s2_1280x1024.frames_s": dig(out, "extra_configs", "s2_1280x1024", "frames_s"),
                "value_block_until_mapped": dig(out, "value_block_until_mapped", "value"),
                "value_pcie_inclusive": dig(out, "value_pcie_inclusive", "value"),
                "keyframe_ms": out.get("keyframe_ms"), "depth_mpix_per_s": out.get("depth_mpix_per_s"),
            }
            for key in ("S8", "S32", "S64", "S128"):
                others["multi_seq.%s.frames_s" % key] = dig(ms, key, "frames_s")
                others["multi_seq.%s.frames_s_block_until_mapped" % key] = dig(ms, key, "frames_s_block_until_mapped")
            for key in ("S8", "S32"):
                for stage in ("track_batch", "observe", "regularise", "keyframe_change"):
                    others["multi_seq.%s.%s.frac" % (key, stage)] = dig(ms, key, "roofline", stage, "frac")
                    others["multi_seq.%s.%s.us" % (key, stage)] = dig(ms, key, "roofline", stage, "avg_launch_us")
            out["roofline"]["others"] = others
        add_achievable(out)
        emit(json.dumps(out))
    if distributed:
        if gather_impl.startswith("rccl"):
            loop.comm_destroy()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
