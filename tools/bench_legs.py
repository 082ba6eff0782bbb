#!/usr/bin/env python
"""Developer tool: the 640x480 loop of bench.py as short legs that differ in ONE thing each (execution model, keyframes kept alive,
keyframe ring export, roofline events), alternated inside one process so that the box's state is the same for all of them.
  python tools/bench_legs.py [--rounds 3] [--steps 200]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--legs", default="pipe,pipe+keep,pipe+ring,pipe+keep+ring,pipe+prof,block,block+keep+ring")
    args = ap.parse_args()
    import torch
    from lsd_slam_amd import synth, capi
    from lsd_slam_amd.driver import DriverLoop
    w, h = 640, 480
    frames, depth0, K, gt = synth.make_sequence(w, h, 100, seq_index=0, kind="s1")
    d_frames = torch.from_numpy(frames).cuda(0)
    torch.cuda.synchronize()
    ptr = lambda i: d_frames[i % 100].data_ptr()
    L = capi.lib()
    RING = 32
    ring = torch.empty((RING, 2, h, w), dtype=torch.float32, device="cuda")
    for rnd in range(args.rounds):
        for leg in args.legs.split(","):
            opts = set(leg.split("+"))
            loop = DriverLoop(w, h, K, ptr(0), depth0, kf_every=10, images_on_device=True, device=0)
            loop.set_pipeline("pipe" in opts)
            if "keep" in opts:
                loop.keep_keyframes(True)
            ctx_h = loop.ctx_handle()
            if "reserve" in opts:
                capi.check(L.lsdhip_ctx_reserve_frames(ctx_h, 100))
            fi = 0

            def run(n):
                nonlocal fi
                left = n
                while left > 0:
                    m = min(left, RING * 10)
                    if "ring" in opts:
                        loop.set_keyframe_ring(ring.data_ptr(), RING)
                    done, _ = loop.run([ptr(fi + 1 + k) for k in range(m)], want_poses=True)
                    fi += done
                    left -= done

            run(20)
            if "prof" in opts:
                capi.check(L.lsdhip_prof_reset(ctx_h))
                capi.check(L.lsdhip_prof_enable(ctx_h, 1))
            capi.check(L.lsdhip_ctx_synchronize(ctx_h))
            dts = []
            for _ in range(3):
                t0 = time.perf_counter()
                run(args.steps)
                capi.check(L.lsdhip_ctx_synchronize(ctx_h))
                dts.append((time.perf_counter() - t0) / args.steps * 1e3)
            capi.check(L.lsdhip_prof_enable(ctx_h, 0))
            loop.close()
            print("round %d %-22s ms/frame %s" % (rnd, leg, " ".join("%.4f" % d for d in dts)), flush=True)


if __name__ == "__main__":
    main()
