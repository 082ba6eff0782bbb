#!/bin/bash
# same-box A/B of two builds of liblsdhip.so on the bench loop: tools/gpu_ab_lib.sh <baseline .so> [bench flags]
# (the baseline is preloaded so that liblsdhip_driver.so binds to it as well)
cd "$(dirname "$0")/.."
BASE=$(readlink -f "$1"); shift
for rep in 1 2; do
  for which in base new; do
    if [ $which = base ]; then export LD_PRELOAD=$BASE LSDHIP_LIB=$BASE; else unset LD_PRELOAD LSDHIP_LIB; fi
    timeout 300 python bench.py --no-cpu-baseline --no-throughput-mode --no-pcie-leg --no-extra-configs "$@" > gpurun_out/abl_${which}_$rep.json 2> gpurun_out/abl_${which}_$rep.err
    unset LD_PRELOAD LSDHIP_LIB
    python - <<PY
import json
try:
    d = json.load(open("gpurun_out/abl_${which}_$rep.json"))
    print("${which} #$rep: %.0f frames/s, %.4f ms/step, launches/frame %.2f, evals %.2f, us per chain launch %.2f, depth %.0f Mpx/s, observe %.1f us, ok %s" % (
        d["value"], d["ms_per_step"], d["track_launches_per_frame"], d["lm_evaluations_per_frame"], d["roofline"]["avg_launch_us"], d["depth_mpix_per_s"], d["roofline_depth"]["avg_launch_us"], d["validation"]["ok"]))
except Exception as e:
    print("${which} #$rep: FAILED", e); print(open("gpurun_out/abl_${which}_$rep.err").read()[-1500:])
PY
  done
done
