#!/bin/bash
# round 5: the artefacts of the final build — suite, bench (default and driver flags), rocprofv3 kernel statistics and PMC traffic of the
# bench and of the 32-sequence loop, Sim3 timing, smoke.  One gpurun call; summaries are copied to profiles/r05_* afterwards.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_final
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/gpu_profile.sh r05 > $O/gpu_profile.log 2>&1; tail -3 $O/gpu_profile.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "bench driver flags rc=$?"
bash tools/gpu_multiseq_prof.sh r05 > $O/multiseq_prof.log 2>&1; tail -25 $O/multiseq_prof.log | head -40
bash tools/gpu_pmc_multiseq.sh r05 > $O/multiseq_pmc.log 2>&1; cat $O/multiseq_pmc.log | tail -20
timeout 200 python tools/bench_sim3.py 2>&1 | tail -2 | tee $O/sim3.json
