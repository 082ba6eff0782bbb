#!/bin/bash
# same-box A/B of the two execution models of the loop: blockUntilMapped (one stream) vs pipelined (tracking beside mapping)
# tools/gpu_ab_pipe.sh [bench flags]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
  for which in block pipe; do
    if [ $which = block ]; then extra="--block-until-mapped"; else extra=""; fi
    timeout 300 python bench.py --no-cpu-baseline --no-throughput-mode --no-pcie-leg --no-extra-configs $extra "$@" > gpurun_out/abp_${which}_$rep.json 2> gpurun_out/abp_${which}_$rep.err
    python - <<PY
import json
try:
    d = json.load(open("gpurun_out/abp_${which}_$rep.json"))
    print("${which} #$rep: %.0f frames/s, %.4f ms/step, launches/frame %.2f, evals %.2f, us per chain launch %.2f, depth %.0f Mpx/s, observe %.1f us, ok %s rmse %.2e" % (
        d["value"], d["ms_per_step"], d["track_launches_per_frame"], d["lm_evaluations_per_frame"], d["roofline"]["avg_launch_us"], d["depth_mpix_per_s"] or 0, (d.get("roofline_depth") or {}).get("avg_launch_us", 0), d["validation"]["ok"], d["validation"]["rmse_vs_gt"]))
except Exception as e:
    print("${which} #$rep: FAILED", e); print(open("gpurun_out/abp_${which}_$rep.err").read()[-2500:])
PY
  done
done
