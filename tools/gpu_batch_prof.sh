#!/bin/bash
# kernel durations and the launch sequence of one throughput-mode batch round (64 jobs; LM launch + evaluation launch per step)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in 0; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/chain_$v -- python $R/bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-pcie-leg --no-extra-configs > /dev/null 2>&1
  python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/chain_$v/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if ("true, 1>" in r["Kernel_Name"] or "true, 2>" in r["Kernel_Name"]) and r["Grid_Size_Y"]!="1"]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the three measured rounds of 64 jobs: the last 3 x budget launches before the eval_throughput hook; print durations of one round
names={}
for r in rows:
    n=r["Kernel_Name"].split("(")[0][-28:]; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    names.setdefault((n,r["Grid_Size_X"],r["Grid_Size_Y"]),[]).append(d)
for k,v in names.items():
    v2=sorted(v); print(k, "n=%d mean %.1f median %.1f max %.1f sum %.0f us" % (len(v), sum(v)/len(v), v2[len(v2)//2], v2[-1], sum(v)))
# sequence of one batch round: consecutive launches with grid y = 64 jobs
seq=[(r["Kernel_Name"].split("(")[0][-14:], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, (int(r["Start_Timestamp"]))/1e3) for r in rows if int(r["Grid_Size_Y"])//1 in (64,)]
if seq:
    t0=seq[0][2]
    print(" ".join("%s:%.0f@%.0f" % (a[-6:],b,c-t0) for a,b,c in seq[:60]))
PY
  rm -rf $R/gpurun_out/chain_$v
done
