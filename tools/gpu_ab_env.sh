#!/bin/bash
# same-box A/B of environment settings on the bench loop: tools/gpu_ab_env.sh "<label>|<VAR=value ...>" ...  (three alternating rounds)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2 3; do
  for spec in "$@"; do
    label="${spec%%|*}"; envs="${spec#*|}"
    env $envs timeout 300 python bench.py --no-cpu-baseline --no-throughput-mode --no-pcie-leg --no-extra-configs --no-roofline-events $BENCH_FLAGS > gpurun_out/abe_${label}_$rep.json 2> gpurun_out/abe_${label}_$rep.err
    python - <<PY
import json
try:
    d = json.load(open("gpurun_out/abe_${label}_$rep.json"))
    print("${label} #$rep: %.0f frames/s, %.4f ms/step, regions %s, launches/frame %.2f, ok %s" % (
        d["value"], d["ms_per_step"], " ".join("%.0f" % (1e3 * x) for x in d["ms_per_step_all_regions"]), d["track_launches_per_frame"], d["validation"]["ok"]))
except Exception as e:
    print("${label} #$rep: FAILED", e)
    print(open("gpurun_out/abe_${label}_$rep.err").read()[-2000:])
PY
  done
done
