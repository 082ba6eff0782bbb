#!/usr/bin/env python
"""Compare two LSDHIP_TRACE_SUMS files (developer tool): entries are (kind, frame id, checksum); the streams interleave differently from
run to run, so entries are matched by (kind, id, occurrence).  Prints the earliest frames at which each kind differs.
kinds: 2 frame pyramid arena | 10-15 tracking inputs at job start (kf idepth L1, var L1, image L1, frame grad L1, kf idepth L3, frame grad L3)
| 20 pose, 21 evaluations*1000 + lastCand | 30 frame mask as the depth update saw it, 31-36 hypothesis planes after the update, 37-39 keyframe depth planes written
(These switches live in the developer build only: python -c "from lsd_slam_amd import build; build.build_variant('devtools', ['LSD_DEVTOOLS'])", then LSDHIP_LIB=lsd_slam_amd/liblsdhip_devtools.so LD_PRELOAD=$LSDHIP_LIB.)"""
import sys
from collections import defaultdict
def load(p):
    d, occ = {}, defaultdict(int)
    for line in open(p):
        k, i, v = line.split()
        key = (int(k), int(i))
        d[key + (occ[key],)] = v
        occ[key] += 1
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
first = {}
for key in sorted(set(a) & set(b), key=lambda k: (k[1], k[0], k[2])):
    if a[key] != b[key] and key[0] not in first:
        first[key[0]] = key[1]
print("entries: %d vs %d, common %d" % (len(a), len(b), len(set(a) & set(b))))
for kind, fid in sorted(first.items(), key=lambda kv: (kv[1], kv[0])):
    print("kind %2d first differs at frame id %d" % (kind, fid))
if not first:
    print("identical")
