#!/bin/bash
# round 5, fourth GPU session: run-to-run determinism of the single-sequence loop (round-4 tree beside this one), bands through one launch, speculation budget
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_ab4
mkdir -p $O
for p in "" "--pipelined"; do
  timeout 300 python tools/determinism_chunks.py --root $R/gpurun_in/r04tree $p 2>&1 | tail -12
  timeout 300 python tools/determinism_chunks.py $p 2>&1 | tail -12
done | tee $O/determinism.txt
timeout 300 python -m pytest tests/test_bands_gpu.py -q 2>&1 | tail -3
for b in 1 8 1 8; do timeout 200 python tools/bench_bands.py --native --bands $b 2>&1 | tail -1; done | tee $O/bands.txt
ms() { python tools/bench_multiseq.py --S $2 --tag "$1" 2>> $O/multiseq.err | tee -a $O/multiseq.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=[x for x in d if x.startswith('S')][0]; print(d['tag'], k, round(d[k]['frames_s']), round(d[k]['frames_s_block_until_mapped']), d[k]['replicas_bit_identical'], d[k]['tracked_good'])"; }
for rep in 1 2; do
  ms spec_budget1M 32
  LSDHIP_BATCH_SPEC=1 ms spec_off 32
done
ms spec_budget1M 8
