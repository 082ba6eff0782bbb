#!/bin/bash
# SQ counters of the tracking batch kernels (k_track_solo, the fused rounds) over pure 32-job batches: two rocprofv3 --pmc passes with
# --kernel-trace only, summarised by tools/pmc_sq_summary.py.   tools/gpu.sh run pmc_solo bash tools/gpu_pmc_solo.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_solo
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_batch.py --batches 32 --rounds 6"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1; echo "pass 1 rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p2 -- $CMD > $OUT/p2.log 2>&1; echo "pass 2 rc=$?"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/p3 -- $CMD > $OUT/p3.log 2>&1; echo "pass 3 rc=$?"
cd $R
python tools/pmc_sq_summary.py $OUT/p1 $OUT/p2 $OUT/p3 > $OUT/summary.json
python - <<PY
import json
d = json.load(open("$OUT/summary.json"))
for k, v in d.items():
    if "k_track" in k:
        print(k[:60], {a: (round(b, 4) if b < 10 else round(b)) for a, b in v.items()})
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
