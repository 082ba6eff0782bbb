#!/bin/bash
# round 5, third GPU session: reject-chain speculation of throughput-mode batches (LSDHIP_BATCH_SPEC=1: off)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_ab3
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_multiseq_gpu.py -m gpu -q -k "batch or multiseq or loop" > $O/pytest_batch.log 2>&1; echo "pytest batch rc=$?"; tail -12 $O/pytest_batch.log
for sp in 4 1; do
  echo "--- batches, LSDHIP_BATCH_SPEC=$sp"; LSDHIP_BATCH_SPEC=$sp timeout 200 python tools/bench_batch.py --batches 8,32,64 --rounds 10 2> $O/batch_spec$sp.err | tee $O/batch_spec$sp.json
done
ms() { python tools/bench_multiseq.py --S $2 --tag "$1" 2>> $O/multiseq.err | tee -a $O/multiseq.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=[x for x in d if x.startswith('S')][0]; print(d['tag'], k, round(d[k]['frames_s']), round(d[k]['frames_s_block_until_mapped']), d[k]['replicas_bit_identical'], d[k]['tracked_good'], d[k]['lm_evaluations_per_frame'])"; }
for rep in 1 2; do
  ms spec4 32
  LSDHIP_BATCH_SPEC=1 ms spec1 32
  LSDHIP_BATCH_SPEC=2 ms spec2 32
done
ms spec4 8
LSDHIP_BATCH_SPEC=1 ms spec1 8
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie-leg --no-extra-configs > $O/bench_tm.json 2> $O/bench_tm.err; python -c "
import json; d=json.load(open('$O/bench_tm.json')); print('value', d['value'], 'keyframe_ms', d.get('keyframe_ms')); print(json.dumps(d.get('roofline_throughput_mode'))[:900])"
timeout 600 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/$O/counters.txt 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/$O/pmc_sq -- python $R/tools/bench_eval.py --levels 1 --repeats 5 > $R/$O/pmc_sq.log 2>&1; echo "pmc rc=$?"
cd $R
python - <<PY
import csv,glob
fs=glob.glob("$O/pmc_sq/**/*counter_collection.csv",recursive=True)
if fs:
    agg={}
    for r in csv.DictReader(open(fs[0])):
        if "Li2" in r["Kernel_Name"] and r["Grid_Size"]!="":
            agg.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(k, len(v), sum(v)/len(v))
else: print("no pmc csv"); print(open("$O/pmc_sq.log").read()[-800:])
PY
rm -rf $O/pmc_sq
