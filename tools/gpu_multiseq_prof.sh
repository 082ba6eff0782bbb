#!/bin/bash
# rocprofv3 kernel statistics of the S-sequence loop (tools/multiseq_check.py --S 32): tools/gpu_multiseq_prof.sh <tag>
TAG=${1:-ms}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/msprof_$TAG -- python $R/tools/multiseq_check.py --S 32 --steps 40 > $R/gpurun_out/msprof_$TAG.txt 2>&1
cd $R
f=$(find gpurun_out/msprof_$TAG -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/msprof_${TAG}_kernel_stats.csv
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/msprof_$TAG/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# one steady step: between two k_image_pyramid_batch launches in the middle of the second run
st=[i for i,r in enumerate(rows) if r["Kernel_Name"].startswith("k_image_pyramid_batch")]
for pick in (len(st)-8, len(st)-5):
    a,b=st[pick],st[pick+1]
    t0=int(rows[a]["Start_Timestamp"]); prev=None
    agg={}
    for r in rows[a:b]:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        name=r["Kernel_Name"].split("(")[0][:46]
        k=agg.setdefault(name,[0,0.0]); k[0]+=1; k[1]+=(e-s)/1e3
    tot=(int(rows[b]["Start_Timestamp"])-t0)/1e3
    busy=sum(v[1] for v in agg.values())
    print("step of %.1f us, kernels busy %.1f us, %d launches" % (tot,busy,b-a))
    for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1]): print("   %-48s n=%3d  %8.1f us" % (k,v[0],v[1]))
PY
rm -rf gpurun_out/msprof_$TAG
head -30 gpurun_out/msprof_${TAG}_kernel_stats.csv | cut -c1-160
