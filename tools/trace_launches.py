#!/usr/bin/env python
"""For the frames whose job differed from its replay (tools/trace_replay.py): which launch of the job was the first to finish the previous
evaluation with other totals?  Needs the LSD_ORDER_CHECK build: LSDHIP_ORDER_DUMP=<hashes> next to LSDHIP_TRACE_SUMS=<trace>.
(These switches live in the developer build only: python -c "from lsd_slam_amd import build; build.build_variant('devtools', ['LSD_DEVTOOLS'])", then LSDHIP_LIB=lsd_slam_amd/liblsdhip_devtools.so LD_PRELOAD=$LSDHIP_LIB.)"""
import sys
from collections import defaultdict
rows = defaultdict(dict)
for line in open(sys.argv[1]):
    k, i, v = line.split()
    rows[int(i)].setdefault(int(k), v)
hashes = {}
for line in open(sys.argv[2]):
    i, lo, hi = line.split()
    hashes[int(i)] = (lo, hi)
for fid in sorted(rows):
    r = rows[fid]
    if 24 not in r or (r.get(24) == r.get(20) and r.get(25) == r.get(21)):
        continue
    a0, b0 = int(r[26], 16), int(r[27], 16)
    seq_a = [hashes.get(a0 + k) for k in range(0, 30)]
    seq_b = [hashes.get(b0 + k) for k in range(0, 30)]
    first = next((k for k in range(30) if seq_a[k] != seq_b[k]), None)
    print("frame %d: launches %d.. vs replay %d..: first launch whose finishing totals differ: #%s  (run: %s, replay: %s)" % (fid, a0, b0, first, seq_a[first] if first is not None else None, seq_b[first] if first is not None else None))
    print("    run   :", [h[0][-6:] if h else None for h in seq_a[:16]])
    print("    replay:", [h[0][-6:] if h else None for h in seq_b[:16]])
