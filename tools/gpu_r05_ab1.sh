#!/bin/bash
# round 5, first GPU session: the suite, then A/B of the granule evaluation (LSDHIP_BATCH_GRAN) and the two-launch observe (LSDHIP_OBS_SPLIT)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_ab1
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for g in 1 0; do
  LSDHIP_BATCH_GRAN=$g timeout 200 python tools/bench_eval.py --levels 3,2,1 > $O/eval_gran$g.json 2> $O/eval_gran$g.err; echo "eval gran=$g rc=$?"; cat $O/eval_gran$g.json
  LSDHIP_BATCH_GRAN=$g timeout 200 python tools/bench_batch.py --batches 32,64 --rounds 10 > $O/batch_gran$g.json 2> $O/batch_gran$g.err; echo "batch gran=$g rc=$?"; cat $O/batch_gran$g.json
done
for cfg in "1 1" "0 0" "1 0" "0 1" "1 1" "0 0"; do
  set -- $cfg
  LSDHIP_BATCH_GRAN=$1 LSDHIP_OBS_SPLIT=$2 timeout 300 python tools/bench_multiseq.py --S 32 --tag "gran$1_split$2" >> $O/multiseq.json 2>> $O/multiseq.err; echo "multiseq $cfg rc=$?"
done
cat $O/multiseq.json
