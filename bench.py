#!/usr/bin/env python
"""bench.py — LSD-SLAM dense hot path on MI355X: tracked frames/sec (+ depth-map Mpixels/sec) at 640x480.

One "step" = one frame of the full track+map loop of BASELINE.json configs[1]: device-resident uint8 frame ->
image/gradient pyramids -> SE3Tracker::trackFrame against the current keyframe (5-level pyramid, LM over levels
4..1) -> DepthMap::updateKeyframe([frame]) (observe/doLineStereo, fill holes, regularise, setDepth); every
KF_EVERY-th frame finalizeKeyFrame + createKeyFrame (propagate, 2x regularise, fill holes, rescale) instead.
Inputs (the rendered uint8 frames) are resident in HBM before the timed region starts.

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


def pmc_traffic():
    """HBM bytes per launch of the residual kernel from the committed PMC passes (profiles/rNN_pmc_traffic.json, made by
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
            if "k_track_step" in name and t.get("hbm_bytes_per_launch") is not None:
                return float(t["hbm_bytes_per_launch"]), os.path.basename(files[-1])
    except Exception:
        pass
    return None, None


def cpu_baseline(frames, depth0, K, n_frames, w, h):
    """The oracle's timing build (-O3, SSE tracker path, 4 mapping threads) on the same loop, bounded sample."""
    from oracle import pyoracle as po
    po.build()
    L = po.lib(fast=True)
    kf = po.Frame(0, frames[0], K, L=L)
    kf.set_depth_gt(depth0)
    dm = po.DepthMap(w, h, K, L=L, threads=4)
    dm.init_gt(kf)
    ref = po.TrackingReference(L=L)
    ref.import_frame(kf)
    tr = po.SE3Tracker(w, h, K, mode=po.SSE, L=L)
    tr.set_max_its([5, 20, 50, 100, 0])
    last = np.array([1.0, 0, 0, 0, 0, 0, 0])
    t_track = t_map = 0.0
    n_upd = 0
    since = 0
    keep = [kf]
    t_all = time.perf_counter()
    for i in range(1, n_frames + 1):
        t0 = time.perf_counter()
        f = po.Frame(i, frames[i % len(frames)], K, L=L)
        ref.import_frame(kf)
        s = kf.stats()
        kf.set_counters(int(s["numFramesTrackedOnThis"]), int(s["numMappedOnThis"]), int(s["numMappedOnThisTotal"]), 0)
        r = tr.track(ref, f, last)
        t1 = time.perf_counter()
        t_track += t1 - t0
        since += 1
        if since >= KF_EVERY:
            dm.finalize()
            dm.create_keyframe(f)
            kf = f
            keep = [kf]
            last = np.array([1.0, 0, 0, 0, 0, 0, 0])
            since = 0
        else:
            dm.update([f])
            f.clear_wasgood()
            last = np.array(r.frameToRef)
            n_upd += 1
        t_map += time.perf_counter() - t1
    total = time.perf_counter() - t_all
    return {
        "value": n_frames / total, "unit": "frames/s", "cores": 4, "kind": "port",
        "sample": "%d frames of the same %dx%d track+map loop (oracle -O3 build: SSE tracker on 1 thread, depth map on 4 "
                  "threads x 10-row chunks as the reference)" % (n_frames, w, h),
        "track_fps": n_frames / t_track, "depth_mpix_per_s": (w * h * n_upd) / t_map / 1e6 if t_map > 0 else None,
        "seconds": total,
    }


def throughput_mode(la, w, h, K, d_frames, depth0, device, jobs=64, rounds=3):
    """The same residual evaluation with enough independent work per launch: `jobs` trackFrame jobs (frames 1..5 of the
    sequence against the keyframe with ground-truth depth) share every launch (lsdhip_tracker_track_batch, throughput mode:
    LM launch + evaluation launch per step).  Reports the algorithmic bandwidth of the step kernels over ALL launches of the
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
    return {"bound": "hbm", "jobs_per_launch": jobs, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "evaluations": int(n_eval), "stream_ms": ms,
            "kernel": "k_track_step<TS_LM> + k_track_step<TS_EVAL>, all launches of %d rounds of %d jobs" % (rounds, jobs)}


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
    ap.add_argument("--seq-frames", type=int, default=50,
                    help="synthetic frames to render (one camera revolution = 50); fewer are played back and forth")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LSD_FORCE_DIST=1: run the RCCL path (process group, keyframe gather, barriers) even with one rank — a one-GPU check of the N > 1 code
    distributed = world > 1 or os.environ.get("LSD_FORCE_DIST") == "1"
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: there is no CPU fallback for the hot path")
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
    L = capi.lib()
    ctx_h = loop.ctx_handle()

    # Finished keyframes leave the loop through a device ring (smoothed idepth + variance planes, copied on the loop's
    # stream); after every batch of frames the ring's new entries are collected on rank 0 with ONE gather (RCCL over xGMI) —
    # the only exchange step of the path.  With one rank the export still runs, so per-GPU work is the same for every N.
    from lsd_slam_amd.multigpu import gather_keyframe_ring
    RING = 32
    ring = torch.empty((RING, 2, h, w), dtype=torch.float32, device=torch.device("cuda", local_rank))
    loop.set_keyframe_ring(ring.data_ptr(), RING)
    recv = [torch.empty_like(ring) for _ in range(world)] if (distributed and rank == 0) else None
    state = {"fi": 0, "exported": 0, "gathered_bytes": 0}

    def run_frames(n):
        """n frames through the C++ loop in batches that produce at most RING keyframes; one gather per batch"""
        left = n
        while left > 0:
            m = min(left, RING * KF_EVERY)
            ptrs = [ptr(state["fi"] + 1 + k) for k in range(m)]
            loop.set_keyframe_ring(ring.data_ptr(), RING)       # the batch's keyframes land in ring[0:new]
            done, _ = loop.run(ptrs)                            # returns with the stream drained: the ring is complete
            state["fi"] += done
            left -= done
            new = loop.keyframes_exported()
            state["exported"] += new
            if distributed:
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
    t0 = clock()
    run_frames(args.steps)
    capi.check(L.lsdhip_ctx_synchronize(ctx_h))
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = clock() - t0
    capi.check(L.lsdhip_prof_enable(ctx_h, 0))
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    st = loop.stats()
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
            "metric": "tracked frames/sec (full track+map loop) at %dx%d" % (w, h), "value": value, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%dx%d 5-level-pyramid SE3 tracking + semi-dense depth update, single sequence per GPU "
                                   "(BASELINE.json configs[%d]); synthetic scene %s, GT-depth init, new keyframe every %d frames"
                                   % (w, h, 1 if (w, h) == (640, 480) else 2, args.scene, KF_EVERY),
                       "width": w, "height": h, "parallelism": "1 sequence per GPU; finished keyframes' depth planes exported to a device ring and gathered to rank 0 (RCCL) once per batch"},
            "track_fps": st.frames / st.seconds_track if st.seconds_track > 0 else None,
            "depth_mpix_per_s": (w * h * st.updates) / st.seconds_map / 1e6 if st.seconds_map > 0 else None,
            "keyframe_ms": st.seconds_keyframe / st.keyframes * 1e3 if st.keyframes > 0 else None,
            "lm_evaluations_per_frame": st.evaluations / max(1, st.frames),
            "roofline": roofline,
        }
        if world == 1 and (w, h) == (640, 480) and not args.no_throughput_mode:
            try:
                out["roofline_throughput_mode"] = throughput_mode(la, w, h, K, d_frames, depth0, local_rank)
            except Exception as e:   # an extra: never let it take the bench line down
                out["roofline_throughput_mode"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames, depth0, K, args.cpu_frames, w, h)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
        sys.stdout.flush()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
