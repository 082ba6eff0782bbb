#!/bin/bash
# same-box A/B of bench.py variants: tools/gpu_ab.sh "<label>|<bench flags>" ...   (results: gpurun_out/ab_<label>.json)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for spec in "$@"; do
  label="${spec%%|*}"; flags="${spec#*|}"
  for rep in 1 2; do
    timeout 300 python bench.py --no-cpu-baseline --no-throughput-mode --no-pcie-leg $flags > gpurun_out/ab_${label}_$rep.json 2> gpurun_out/ab_${label}_$rep.err
    python - <<PY
import json
try:
    d = json.load(open("gpurun_out/ab_${label}_$rep.json"))
    print("${label} #$rep: %.0f frames/s, %.4f ms/step, launches/frame %.2f, evals %.2f, ok %s" % (
        d["value"], d["ms_per_step"], d["track_launches_per_frame"], d["lm_evaluations_per_frame"], d["validation"]["ok"]))
except Exception as e:
    print("${label} #$rep: FAILED", e)
    print(open("gpurun_out/ab_${label}_$rep.err").read()[-2000:])
PY
  done
done
