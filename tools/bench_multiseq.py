#!/usr/bin/env python
"""The S-sequence leg of bench.py alone (bench.multi_seq: lsd_slam_hip::SlamLoopBatch, pipelined and blockUntilMapped), for A/B runs
under environment switches (LSD_COARSE_MIN_JOBS=n: lsdloopbatch_set_coarse_min_jobs of every loop).  Prints one JSON line.
    python tools/bench_multiseq.py [--S 32] [--steps 30] [--tag name]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from lsd_slam_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--S", type=int, nargs="+", default=[32])
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--regions", type=int, default=3)
ap.add_argument("--tag", type=str, default="")
args = ap.parse_args()
w, h = 640, 480
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(8) as ex:
    seqs = list(ex.map(lambda q: synth.make_sequence(w, h, 50, seq_index=q, kind="S1"), range(8)))
K = seqs[0][2]
seeds = [(torch.from_numpy(q[0]).cuda(0), q[1]) for q in seqs]
torch.cuda.synchronize()
out = bench.multi_seq(torch, w, h, K, seeds, 0, None, sizes=tuple(args.S), steps=args.steps, warmup=10, regions=args.regions)
out.pop("workload", None)
print(json.dumps({"tag": args.tag, **out}))
