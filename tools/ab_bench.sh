#!/bin/bash
# A/B of one environment switch inside a single gpurun call (same box, alternating runs).  Usage: tools/ab_bench.sh VAR
VAR=$1
for i in 1 2 3; do
  A=$(python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline-events --no-throughput-mode 2>/dev/null | python -c "import json,sys; print('%.0f' % json.load(sys.stdin)['value'])")
  B=$(env $VAR=1 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline-events --no-throughput-mode 2>/dev/null | python -c "import json,sys; print('%.0f' % json.load(sys.stdin)['value'])")
  echo "default: $A fps    $VAR=1: $B fps"
done
