#!/bin/bash
# round 5, fifth GPU session: the suite with the new evidence tests, the bench line with the new keys
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_s5
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log; grep -n "affine lighting vs float64\|closest ensemble member" $O/pytest.log | cut -c1-400
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_driver_flags.json"))
print("value", d["value"], "ms", d["ms_per_step"], "kf_ms", d.get("keyframe_ms"), "valid", d["validation"]["ok"])
print("block", (d.get("value_block_until_mapped") or {}).get("value"))
print("speedup", json.dumps(d.get("speedup_vs_cpu_baseline")))
cb=d.get("cpu_baseline") or {}
print("cpu", cb.get("value"), "pipelined", json.dumps(cb.get("pipelined")))
print("roofline_depth", json.dumps(d.get("roofline_depth"))[:700])
tm=d.get("roofline_throughput_mode") or {}
print("throughput frac", tm.get("frac"), "L1", (tm.get("level1_evaluation") or {}).get("frac"))
ms=(d.get("extra_configs") or {}).get("multi_seq") or {}
for k in ("S8","S32"):
    r=ms.get(k) or {}
    print(k, r.get("frames_s"), r.get("frames_s_block_until_mapped"), r.get("replicas_bit_identical"))
    for kk,v in (r.get("roofline") or {}).items():
        if isinstance(v, dict): print("   ", kk, "us", round(v.get("avg_launch_us",0),1), "frac", round(v.get("frac",0),4), "MB", round(v.get("algorithmic_bytes_per_launch",0)/1e6,1))
        else: print("   ", kk, v)
ec=d.get("extra_configs") or {}
print("s2", (ec.get("s2_1280x1024") or {}).get("frames_s"), "reg", json.dumps((ec.get("reg_3840x2160") or {}).get("bands_vs_full_frame")), (ec.get("reg_3840x2160") or {}).get("full_frame"))
PY
