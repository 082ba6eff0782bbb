#!/bin/bash
# does the GPU's performance level explain the two speeds the bench loop shows on one box?
cd "$(dirname "$0")/.."
rocm-smi --showperflevel --showclocks 2>&1 | grep -v "^=\|^$" | head -20
run() { timeout 300 python bench.py --no-cpu-baseline --no-throughput-mode --no-pcie-leg --no-extra-configs --repeats 8 > gpurun_out/pl_$1.json 2> gpurun_out/pl_$1.err; python -c "
import json; d=json.load(open('gpurun_out/pl_$1.json')); print('$1', '%.0f fps' % d['value'], ['%.3f' % x for x in d['ms_per_step_all_regions']])"; }
run auto1; run auto2
rocm-smi --setperflevel high 2>&1 | grep -v "^=\|^$" | head -5
rocm-smi --showperflevel --showclocks 2>&1 | grep -v "^=\|^$" | head -12
run high1; run high2
rocm-smi --setperflevel auto 2>&1 | grep -v "^=\|^$" | head -3
