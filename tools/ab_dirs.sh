#!/bin/bash
# A/B of two checkouts inside a single gpurun call (same box, alternating runs).  Usage: tools/ab_dirs.sh DIR_A DIR_B [rounds]
A=$1; B=$2; N=${3:-3}
run() { (cd $1 && python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-roofline-events --no-throughput-mode 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.0f fps (track %.0f, depth %.0f Mpx/s)' % (d['value'], d['track_fps'], d['depth_mpix_per_s']))"); }
for i in $(seq $N); do
  echo "$A: $(run $A)    $B: $(run $B)"
done
