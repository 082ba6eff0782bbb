#!/usr/bin/env python
"""BASELINE.json configs[4]: 3840x2160 depth-map regularisation (fill holes + regularise per pass) tiled in row bands.
Not the driver's bench line (that is bench.py); prints one JSON line with Mpixel/s and algorithmic GB/s.

  python tools/bench_bands.py --bands 8                 # 8 windows on one GPU (LocalComm), halo rows copied on-device
  python -m torch.distributed.run --nproc-per-node 8 tools/bench_bands.py     # one band per GPU, RCCL send/recv halos
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=3840)
ap.add_argument("--height", type=int, default=2160)
ap.add_argument("--bands", type=int, default=1)
ap.add_argument("--passes", type=int, default=20)
ap.add_argument("--native", action="store_true", help="the C++ loop (liblsdhip_driver.so lsdband_*): nothing synchronises the host between passes")
args = ap.parse_args()

import torch
from lsd_slam_amd.bands import BandPlan, BandRegularizer, DistComm, HipBandEngine, LocalComm, synth_s3

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
w, H = args.width, args.height
hyp, maxgrad = synth_s3(w, H)
if args.native:
    from lsd_slam_amd.bands import NativeBandRegularizer
    from lsd_slam_amd import driver
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")            # rendezvous + barriers only: the halo rows go through RCCL inside the C++ loop
        uid = [driver.DriverLoop.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        br = NativeBandRegularizer(w, H, world, [rank], device=local_rank)
        br.comm_init(uid[0], world, rank, list(range(world)))
    else:
        br = NativeBandRegularizer(w, H, args.bands, list(range(args.bands)), device=local_rank)
    plan = br.plan
    br.load(hyp, maxgrad)
    br.run(2)
    br.synchronize()
    br.load(hyp, maxgrad)
    br.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    br.run(args.passes)
    br.synchronize()
    if world > 1:
        dist.barrier()
else:
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        plan = BandPlan(H, world)
        br = BandRegularizer(plan, [HipBandEngine(w, plan.window_rows, device=local_rank)], DistComm(), [rank])
    else:
        plan = BandPlan(H, args.bands)
        br = BandRegularizer(plan, [HipBandEngine(w, plan.window_rows) for _ in range(args.bands)], LocalComm(), list(range(args.bands)))
    br.load(hyp, maxgrad)
    br.run(2)          # warm-up
    br.load(hyp, maxgrad)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    br.run(args.passes)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
dt = time.perf_counter() - t0
if rank == 0:
    px = w * H * args.passes
    # algorithmic bytes per pixel per pass: K5 ~34 + K6 30
    print(json.dumps({"metric": "depth regularisation Mpixel/s (%dx%d, fill holes + regularise per pass)" % (w, H),
                      "value": px / dt / 1e6, "unit": "Mpixel/s", "n_gpus": world, "bands": plan.world, "passes": args.passes,
                      "ms_per_pass": dt / args.passes * 1e3, "algorithmic_GBps": px * 64.0 / dt / 1e9,
                      "halo_bytes_per_pass": plan.halo_bytes_per_pass(w), "window_rows": plan.window_rows}))
if world > 1:
    dist.destroy_process_group()
