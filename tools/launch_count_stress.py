#!/usr/bin/env python
"""Developer tool (round 6): root cause of the launch-count off-by-one (profiles/r05_notes.md section 7, VERDICT r05 "next" #1a).

Runs the SAME trackFrame job many times on trackers with different launch budgets (normal, LSDHIP_BUDGET_FIXED = 1 / 2 / 3 / 12) and compares
poses, evaluation counts and the device-side count of evaluating launches (TrackSummary::numLaunches).  With the LSD_DEVTOOLS library
(lsd_slam_amd/build.py build_variant("devtools", ["LSD_DEVTOOLS"]); LSDHIP_LIB=... LSDHIP_LAUNCH_LOG=1) every launch leaves a 16-int record
(which exit it took, the state it loaded and the state it published); on a mismatch both logs are printed side by side.

    python tools/launch_count_stress.py [--reps 200] [--spec 0|1] [--size 640x480]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--size", default="640x480")
ap.add_argument("--spec", type=int, default=1, help="1: default speculation, 0: one trial per launch")
ap.add_argument("--budgets", default="0,1,2,3,7")
ap.add_argument("--frames", type=int, default=4)
args = ap.parse_args()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lsd_slam_amd as hip
from lsd_slam_amd import synth, capi

w, h = [int(v) for v in args.size.split("x")]
frames, depth0, K, gt = synth.make_sequence(w, h, args.frames + 1)
ctx = hip.Context(w, h, K, device=0)
kf = hip.Frame(ctx, 0, frames[0])
kf.setDepthFromGroundTruth(depth0)
ref = hip.TrackingReference()
ref.importFrame(kf)
L = capi.lib()
have_log = hasattr(L, "lsdhip_tracker_debug_log") and os.environ.get("LSDHIP_LAUNCH_LOG")
if have_log:
    L.lsdhip_tracker_debug_log.restype = C.c_int
    L.lsdhip_tracker_debug_log.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ITS = [5, 20, 50, 100, 100]
budgets = [int(b) for b in args.budgets.split(",")]
trackers = []
for b in budgets:
    if b > 0:
        os.environ["LSDHIP_BUDGET_FIXED"] = str(b)
    else:
        os.environ.pop("LSDHIP_BUDGET_FIXED", None)
    t = hip.SE3Tracker(ctx)
    t.set_maxItsPerLvl(ITS)
    if not args.spec:
        t.set_speculation(1)
    trackers.append(t)
os.environ.pop("LSDHIP_BUDGET_FIXED", None)


def get_log(t):
    if not have_log:
        return None
    buf = np.zeros(4096 * 16, np.int32)
    n = L.lsdhip_tracker_debug_log(t.h_, buf.ctypes.data, buf.size)
    return buf[:n].reshape(-1, 16).copy()


def show(log, tag):
    names = "seq exit par|first|last lvlIn pendIn ncandIn pc lvlOut nEval nLaunch done phase incTry ncandOut pendOut nLaunchIn".split()
    print("  log of %s (%s)" % (tag, " ".join(names)))
    for r in log:
        if r[0] == 0 and r[1] == 0:
            continue
        print("   ord %3d exit %d flags %d | in: lvl %d pend %d ncand %d nL %d | pc %2d | out: lvl %d nEval %3d nL %3d done %d phase %d incTry %d ncand %d pend %d" % (
            r[0] & 0xFFF, r[1], r[2], r[3], r[4], r[5], r[15], r[6], r[7], r[8], r[9], r[10], r[11], r[12], r[13], r[14]))


IDENT7 = hip.IDENTITY
mism = 0
seen = {}
for rep in range(args.reps):
    fi = 1 + rep % args.frames
    res = []
    for b, t in zip(budgets, trackers):
        f = hip.Frame(ctx, 100 + rep, frames[fi])
        p = t.trackFrame(ref, f, IDENT7)
        res.append((np.asarray(p).tobytes(), int(t.last.numEvaluations), int(t.launch_stats()[0]), get_log(t)))
        f.close() if hasattr(f, "close") else None
    base = res[0]
    for b, r in zip(budgets[1:], res[1:]):
        if r[:3] != base[:3]:
            mism += 1
            print("rep %d frame %d: budget %d: evaluations %d launches %d pose %s | normal: evaluations %d launches %d" % (
                rep, fi, b, r[1], r[2], "equal" if r[0] == base[0] else "DIFFERS", base[1], base[2]))
            if have_log and mism <= 6:
                show(base[3], "normal budget")
                show(r[3], "budget %d" % b)
    key = (fi, base[1], base[2])
    seen[key] = seen.get(key, 0) + 1
if hasattr(trackers[0], "summary_stats"):
    for b, t in zip(budgets, trackers):
        print("  budget %d: summary stats (polled, late, max late ns, stale words, first / last stale word) = %s" % (b, t.summary_stats()))
print("launch_count_stress: %d reps x %d budgets %s, spec %d: %d mismatches; (frame, evaluations, launches) of the normal tracker: %s" % (
    args.reps, len(budgets), budgets, args.spec, mism, sorted(seen.items())))
