#!/bin/bash
# SQ counters (two passes) of the level-1 evaluation launch, the 4K regulariser pass and the S = 32 loop
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU"
run() { # tag, counters, command...
  tag=$1; shift; pmc=$1; shift
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $O/$tag -- "$@" > $O/$tag.log 2>&1; echo "$tag rc=$?"
}
run eval_p1 "$P1" python $R/tools/bench_eval.py --levels 1 --repeats 5
run eval_p2 "$P2" python $R/tools/bench_eval.py --levels 1 --repeats 5
run reg_p1 "$P1" python $R/tools/bench_bands.py --native --bands 1 --passes 10
run reg_p2 "$P2" python $R/tools/bench_bands.py --native --bands 1 --passes 10
run ms_p1 "$P1" python $R/tools/bench_multiseq.py --S 32 --steps 12 --regions 1
run ms_p2 "$P2" python $R/tools/bench_multiseq.py --S 32 --steps 12 --regions 1
cd $R
python tools/pmc_sq_summary.py $O/eval_p1 $O/eval_p2 > $O/eval_sq.json
python tools/pmc_sq_summary.py $O/reg_p1 $O/reg_p2 > $O/reg_sq.json
python tools/pmc_sq_summary.py $O/ms_p1 $O/ms_p2 > $O/multiseq_sq.json
rm -rf $O/eval_p1 $O/eval_p2 $O/reg_p1 $O/reg_p2 $O/ms_p1 $O/ms_p2
python - <<PY
import json
for f in ("eval_sq","reg_sq","multiseq_sq"):
    d=json.load(open("$O/%s.json" % f))
    print("==", f)
    for k,v in sorted(d.items(), key=lambda kv:-kv[1].get("SQ_BUSY_CYCLES",0)*kv[1].get("dispatches",0))[:9]:
        if not k.startswith("k_"): continue
        print(k[:44], v.get("dispatches"), {kk.replace("SQ_","").replace("_per_WAVE_CYCLE","/wc"): round(vv,3) for kk,vv in v.items() if kk.endswith("_per_WAVE_CYCLE")},
              "valu", round(v.get("SQ_INSTS_VALU",0)), "lds", round(v.get("SQ_INSTS_LDS",0)), "vmem_rd", round(v.get("SQ_INSTS_VMEM_RD",0)), "bankconf", round(v.get("SQ_LDS_BANK_CONFLICT",0)), "waves", round(v.get("SQ_WAVES",0)))
PY
