#!/bin/bash
# bench.py at the larger frame sizes of BASELINE.json (not the driver's bench line): 1280x1024 (configs[2] size) and
# 3840x2160 (the HBM-stress size of configs[4]).  Usage: tools/bench_sizes.sh <tag>
TAG=${1:-sizes}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python bench.py --width 1280 --height 1024 --steps 60 --warmup 10 --seq-frames 16 --cpu-frames 20 > $OUT/bench_1280x1024.json 2> $OUT/bench_1280.err; echo rc=$?
cat $OUT/bench_1280x1024.json
timeout 1200 python bench.py --width 3840 --height 2160 --steps 24 --warmup 6 --seq-frames 6 --no-cpu-baseline > $OUT/bench_3840x2160.json 2> $OUT/bench_4k.err; echo rc=$?
cat $OUT/bench_3840x2160.json
tail -3 $OUT/bench_4k.err
