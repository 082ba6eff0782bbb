#!/bin/bash
# One gpurun call that produces everything under profiles/ for a round: the bench line, the rocprofv3 kernel-trace stats
# of the same command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE, each alone with --kernel-trace only, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Usage: tools/gpu_profile.sh <round-tag>   (e.g. r01)
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-throughput-mode"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_under_trace.json 2> $OUT/trace.err; echo "trace rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH --no-roofline-events > $OUT/bench_under_pmc_fetch.json 2> $OUT/pmc_fetch.err; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH --no-roofline-events > $OUT/bench_under_pmc_write.json 2> $OUT/pmc_write.err; echo "pmc write rc=$?"
# the throughput-mode evaluation launch alone (64 jobs, level 1): rocprofv3 per-launch durations + algorithmic bytes from the tool
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/eval_trace -- python $GRAFT_REPO_ROOT/tools/bench_eval.py --levels 1 --repeats 20 > $OUT/eval_l1.json 2> $OUT/eval_trace.err; echo "eval trace rc=$?"
cd $GRAFT_REPO_ROOT
timeout 300 python tools/bench_eval.py > $OUT/eval_levels.json 2> /dev/null
find $OUT -name "*.csv" | head -20
# keep the merged payload small: the per-dispatch traces are summarised here, only the summaries travel back
python tools/profile_summary.py $OUT $TAG
rm -rf $OUT/eval_trace $OUT/trace/*/*kernel_trace.csv $OUT/pmc_fetch/*/*kernel_trace.csv $OUT/pmc_write/*/*kernel_trace.csv 2>/dev/null
ls -la $OUT
