#!/bin/bash
cd "$(dirname "$0")/.."
run() { timeout 300 python bench.py --no-cpu-baseline --no-throughput-mode --no-pcie-leg --no-extra-configs --repeats $2 --steps $3 > gpurun_out/rg_$1.json 2> gpurun_out/rg_$1.err; python -c "
import json; d=json.load(open('gpurun_out/rg_$1.json')); t0=d['region_start_unix_s'][0]
print('$1', '%.0f fps' % d['value'], ' '.join('%.0f@%.1f' % (x*1000, t % 1000) for x, t in zip(d['ms_per_step_all_regions'], d['region_start_unix_s'])))"; }
date +%s.%N
run a 30 100; run b 30 100; run c 12 200
