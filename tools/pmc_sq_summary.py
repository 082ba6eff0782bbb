#!/usr/bin/env python
"""Per-kernel means of the SQ counters of rocprofv3 --pmc passes (counter_collection.csv files under the given directories) and the
ratios that say what a kernel waits for: python tools/pmc_sq_summary.py <dir> [<dir> ...] [--min-dispatches N] > summary.json"""
import csv, glob, json, sys
from collections import defaultdict
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
agg = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].strip()
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for name, cs in agg.items():
    rec = {k: sum(v) / len(v) for k, v in cs.items()}
    rec["dispatches"] = max(len(v) for v in cs.values())
    wc = rec.get("SQ_WAVE_CYCLES")
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
            if k in rec:
                rec[k + "_per_WAVE_CYCLE"] = rec[k] / wc
    out[name] = rec
json.dump(out, sys.stdout, indent=1, sort_keys=True)
