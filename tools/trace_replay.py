#!/usr/bin/env python
"""LSDHIP_TRACK_REPLAY=1 LSDHIP_TRACE_SUMS=<file>: per frame, did the job that ran beside the mapping stream (kinds 24 / 25) give what
its replay on a quiet GPU gave (kinds 20 / 21)?  With LSDHIP_TRACE_INPUTS=1 also: were the job's inputs the same at both starts
(kinds 11-18, 51-58 vs 111-118, 151-158)?
(These switches live in the developer build only: python -c "from lsd_slam_amd import build; build.build_variant('devtools', ['LSD_DEVTOOLS'])", then LSDHIP_LIB=lsd_slam_amd/liblsdhip_devtools.so LD_PRELOAD=$LSDHIP_LIB.)"""
import sys
from collections import defaultdict
rows = defaultdict(dict)
for line in open(sys.argv[1]):
    k, i, v = line.split()
    rows[int(i)].setdefault(int(k), v)
bad = 0
for fid in sorted(rows):
    r = rows[fid]
    if 24 not in r:
        continue
    same = r.get(24) == r.get(20) and r.get(25) == r.get(21)
    inputs = [k for k in list(range(11, 19)) + list(range(51, 59)) if k in r and (100 + k) in r and r[k] != r[100 + k]]
    if not same or inputs:
        bad += 1
        print("frame %d: result %s (evals*1000+lastCand %s vs replay %s), inputs differing at kinds %s" % (fid, "same" if same else "DIFFERENT", r.get(25), r.get(21), inputs))
print("%d frames, %d with a difference" % (sum(1 for f in rows if 24 in rows[f]), bad))
