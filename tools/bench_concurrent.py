#!/usr/bin/env python
"""K independent single-sequence loops (lsd_slam_hip::SlamLoop, pipelined) in K PROCESSES on one GPU, started together: the other way of
putting several sequences on one GPU — the alternative to the shared launches of SlamLoopBatch (bench.multi_seq).  Every process owns its
context, its two streams and its hardware queues; nothing is shared but the chip.  Prints one JSON line.
    python tools/bench_concurrent.py --K 1 2 4 8 [--frames 1500] [--block]"""
import argparse, json, os, sys, time
import multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, K, frames_n, pipelined, barrier, q, own_context=False):
    import numpy as np
    import torch
    from lsd_slam_amd import synth, capi
    from lsd_slam_amd.driver import DriverLoop
    w, h = 640, 480
    frames, depth0, Kc, _ = synth.make_sequence(w, h, 50, seq_index=rank % 8, kind="S1")
    if own_context:
        # threads of one process: lsd_slam_hip::Context::get shares one context (one mutex, one pair of streams) per (w, h, K) — a camera
        # matrix that differs in the sixth digit gives every loop its own
        Kc = np.array(Kc, np.float32).copy()
        Kc[0] *= 1.0 + 1e-6 * (rank + 1)
    d_frames = torch.from_numpy(frames).cuda(0)
    n = d_frames.shape[0]
    order = list(range(n)) + list(range(n - 2, 0, -1))
    ptr = lambda i: d_frames[order[i % len(order)]].data_ptr()
    loop = DriverLoop(w, h, Kc, ptr(0), depth0, kf_every=10, images_on_device=True, device=0)
    loop.set_pipeline(pipelined)
    L = capi.lib()
    ctx_h = loop.ctx_handle()
    capi.check(L.lsdhip_ctx_reserve_frames(ctx_h, (frames_n + 40) // 10 + 24))
    loop.run([ptr(1 + k) for k in range(30)])
    capi.check(L.lsdhip_ctx_synchronize(ctx_h))
    barrier.wait()
    t0 = time.time()
    loop.run([ptr(31 + k) for k in range(frames_n)])
    capi.check(L.lsdhip_ctx_synchronize(ctx_h))
    t1 = time.time()
    st = loop.stats()
    q.put({"rank": rank, "t0": t0, "t1": t1, "frames": frames_n, "tracked_good": int(st.tracked_good), "lost": int(st.frames - st.tracked_good)})
    loop.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--K", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--frames", type=int, default=1500)
    ap.add_argument("--block", action="store_true", help="blockUntilMapped loops instead of pipelined ones")
    ap.add_argument("--threads", action="store_true", help="K threads of ONE process, each loop on its own context (own streams), instead of K processes")
    args = ap.parse_args()
    mp.set_start_method("spawn")
    out = {}
    if args.threads:
        import threading, queue
        for K in args.K:
            barrier = threading.Barrier(K)
            q = queue.Queue()
            ts = [threading.Thread(target=worker, args=(r, K, args.frames, not args.block, barrier, q, True)) for r in range(K)]
            for t in ts:
                t.start()
            recs = [q.get(timeout=600) for _ in range(K)]
            for t in ts:
                t.join(timeout=60)
            span = max(r["t1"] for r in recs) - min(r["t0"] for r in recs)
            own = [r["frames"] / (r["t1"] - r["t0"]) for r in recs]
            out["K%d" % K] = {"frames_s": K * args.frames / span, "per_thread_frames_s": [round(v, 1) for v in sorted(own)], "span_s": span,
                              "lost": sum(r["lost"] for r in recs)}
        print(json.dumps({"model": "blockUntilMapped" if args.block else "pipelined", "threads": True, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), **out}))
        sys.exit(0)
    for K in args.K:
        barrier = mp.Barrier(K)
        q = mp.Queue()
        ps = [mp.Process(target=worker, args=(r, K, args.frames, not args.block, barrier, q)) for r in range(K)]
        for p in ps:
            p.start()
        recs = [q.get(timeout=600) for _ in range(K)]
        for p in ps:
            p.join(timeout=60)
        span = max(r["t1"] for r in recs) - min(r["t0"] for r in recs)
        own = [r["frames"] / (r["t1"] - r["t0"]) for r in recs]
        out["K%d" % K] = {"frames_s": K * args.frames / span, "per_process_frames_s": [round(v, 1) for v in sorted(own)], "span_s": span,
                          "lost": sum(r["lost"] for r in recs), "start_skew_ms": (max(r["t0"] for r in recs) - min(r["t0"] for r in recs)) * 1e3}
    print(json.dumps({"model": "blockUntilMapped" if args.block else "pipelined", **out}))
