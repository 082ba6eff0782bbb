#!/bin/bash
# Every GPU-side step of a round, one parametrised script: gpurun -- 'bash tools/gpu.sh <step> [args]'; output under gpurun_out/r06_<step>/
# steps: suite | pytest <args> | bench [args] | run <name> <cmd...> | ab <tag> "ENV=.."... | ablib <tag> <variant>... | ablibtrack <tag> <variant>... | batchprof <tag> [S] |
#        profile <tag> (tools/gpu_profile.sh: bench line + rocprofv3 kernel stats + PMC traffic passes) | pmc <tag> (HBM traffic of the S = 32 loop: tools/gpu_pmc_multiseq.sh) |
#        launchcount | launchcount2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
step=$1; shift
out=gpurun_out/r06_$step
mkdir -p $out
case $step in
  launchcount)
    # VERDICT r05 next #1a: the launch-count off-by-one
    timeout 600 python tools/launch_count_stress.py --reps 300 --spec 1 2>&1 | tail -n 40 | cut -c1-400 | tee $out/default_spec1.txt
    timeout 600 python tools/launch_count_stress.py --reps 150 --spec 0 2>&1 | tail -n 40 | cut -c1-400 | tee $out/default_spec0.txt
    LSDHIP_LIB=$R/lsd_slam_amd/liblsdhip_devtools.so LSDHIP_LAUNCH_LOG=1 timeout 900 python tools/launch_count_stress.py --reps 300 --spec 1 2>&1 | tail -n 400 | cut -c1-400 | tee $out/devtools_spec1.txt
    timeout 300 python tools/determinism_frames.py --reps 6 2>&1 | tail -n 8 | cut -c1-300 | tee $out/frames.txt
    ;;
  launchcount2)
    timeout 900 python tools/launch_count_stress.py --reps 4000 --spec 1 2>&1 | tail -n 40 | cut -c1-400 | tee $out/spec1.txt
    timeout 900 python tools/launch_count_stress.py --reps 2000 --spec 1 --budgets 0,1,1,1,1 2>&1 | tail -n 40 | cut -c1-400 | tee $out/spec1_b1.txt
    timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "budget or speculative or trackframe" 2>&1 | tail -n 5 | tee $out/pytest.txt
    ;;
  suite)
    # the whole GPU suite (what the driver runs at round end) + smoke
    timeout 1500 python -m pytest tests -x -q -m gpu "$@" 2>&1 | tail -n 15 | tee $out/pytest.txt
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 3 | tee $out/smoke.txt
    ;;
  pytest)
    # selected tests: tools/gpu.sh pytest <pytest args>
    timeout 1500 python -m pytest -x -q -m gpu "$@" > $out/pytest_full.txt 2>&1
    grep -v "^  *[a-z_]* = \|^    " $out/pytest_full.txt | tail -n 60 | cut -c1-1500 | tee $out/pytest.txt
    ;;
  run)
    # any developer tool: tools/gpu.sh run <name> <command ...>   (output to gpurun_out/r06_run/<name>.txt)
    name=$1; shift
    timeout 1500 "$@" 2>&1 | tail -n 200 | cut -c1-700 | tee $out/$name.txt
    ;;
  bench)
    # the driver's own command line (+ extra args); the JSON line goes to $out/bench.json
    timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 "$@" > $out/bench.json 2> $out/bench.err
    tail -n 5 $out/bench.err | cut -c1-300
    python - <<PY
import json
d = json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "validation", d["validation"]["ok"])
print("roofline", {k: v for k, v in d["roofline"].items() if k != "others"})
for k, v in (d["roofline"].get("others") or {}).items():
    print("   ", k, v)
cb = d.get("cpu_baseline") or {}
print("cpu_baseline", cb.get("value"), cb.get("kind"), "speedup", d.get("speedup_vs_cpu_baseline", {}).get("pipelined"), d.get("speedup_vs_cpu_baseline", {}).get("block_until_mapped"))
PY
    ;;
  batchprof)
    # rocprofv3 kernel trace of the S-sequence loop: timeline of the tracking batches + kernel statistics.  tools/gpu.sh batchprof <tag> [S]
    tag=${1:-base}; S=${2:-32}
    (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_$tag -- python $R/tools/bench_multiseq.py --S $S --steps 30 --regions 2 > $R/$out/${tag}_bench.json 2> $R/$out/${tag}_err.txt)
    f=$(find $out/prof_$tag -name "*kernel_trace.csv" | head -1)
    g=$(find $out/prof_$tag -name "*kernel_stats.csv" | head -1)
    python tools/batch_timeline.py $f $S | cut -c1-1500 | tee $out/${tag}_timeline.txt
    head -16 $g | cut -c1-170 | tee $out/${tag}_kernel_stats.csv
    python -c "
import json; d=json.loads(open('$out/${tag}_bench.json').read().strip().splitlines()[-1]); k='S$S'
print(k, 'frames_s', d[k].get('frames_s'), 'block', d[k].get('frames_s_block_until_mapped'), 'track_batch us', ((d[k].get('roofline') or {}).get('track_batch') or {}).get('avg_launch_us'))"
    rm -rf $out/prof_$tag
    ;;
  ab)
    # A/B of environment switches on the pure tracking batches and the S-sequence loop, alternating: tools/gpu.sh ab <tag> "ENV=1 ..." ["ENV2=..."]
    # (ABB="8,32,64" batch sizes of the pure batches, ABS="8 32" sequence counts of the loop)
    tag=$1; shift
    for rep in 1 2; do
      i=0
      for envs in "" "$@"; do
        echo "== rep $rep arm $i: [$envs]" | tee -a $out/$tag.txt
        env $envs timeout 300 python tools/bench_batch.py --batches ${ABB:-8,32,64} --rounds 20 2>/dev/null | cut -c1-200 | tee -a $out/$tag.txt
        env $envs timeout 300 python tools/bench_multiseq.py --S ${ABS:-8 32} --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in [k for k in d if k[:1] == 'S' and k[1:].isdigit()]:
    r=(d[k].get('roofline') or {}).get('track_batch') or {}
    print(k, 'frames_s %.0f block %.0f track_batch %.1f us frac %.4f' % (d[k]['frames_s'], d[k]['frames_s_block_until_mapped'], r.get('avg_launch_us') or 0, r.get('frac') or 0))" | tee -a $out/$tag.txt
        i=$((i+1))
      done
    done
    ;;
  ablib)
    # same-box A/B of builds of liblsdhip.so (lsd_slam_amd/liblsdhip_<name>.so, build.build_variant), alternating with the default library:
    # tools/gpu.sh ablib <tag> <name> [<name> ...]   — the 4K regulariser (full frame, 8 bands), the S = 32 loop and the bench loop's keyframe time
    tag=$1; shift
    for rep in 1 2; do
      for name in default "$@"; do
        if [ $name = default ]; then unset LD_PRELOAD LSDHIP_LIB; else export LD_PRELOAD=$R/lsd_slam_amd/liblsdhip_$name.so LSDHIP_LIB=$R/lsd_slam_amd/liblsdhip_$name.so; fi
        echo "== rep $rep lib $name" | tee -a $out/$tag.txt
        for b in 1 8; do timeout 300 python tools/bench_bands.py --native --bands $b --passes 20 2>/dev/null | tail -n 1 | cut -c1-220 | tee -a $out/$tag.txt; done
        timeout 300 python tools/bench_multiseq.py --S 32 --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in ('S32',):
    r=d[k].get('roofline') or {}
    print(k, 'frames_s %.0f block %.0f' % (d[k]['frames_s'], d[k]['frames_s_block_until_mapped']), ' '.join('%s %.1f us' % (n, (r.get(n) or {}).get('avg_launch_us') or 0) for n in ('track_batch','observe','regularise','keyframe_change','idepth_pyramids','frame_pyramids')))" | tee -a $out/$tag.txt
        timeout 300 python bench.py --no-cpu-baseline --no-throughput-mode --no-pcie-leg --no-extra-configs --steps 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench value %.0f keyframe_ms %.4f depth_mpix_per_s %.0f' % (d['value'], d['keyframe_ms'], d['depth_mpix_per_s']))" | tee -a $out/$tag.txt
        unset LD_PRELOAD LSDHIP_LIB
      done
    done
    ;;
  ablibtrack)
    # same-box A/B of builds of liblsdhip.so on the tracking batches and the S-sequence loop: tools/gpu.sh ablibtrack <tag> <name> [<name> ...]
    tag=$1; shift
    for rep in 1 2; do
      for name in default "$@"; do
        if [ $name = default ]; then unset LD_PRELOAD LSDHIP_LIB; else export LD_PRELOAD=$R/lsd_slam_amd/liblsdhip_$name.so LSDHIP_LIB=$R/lsd_slam_amd/liblsdhip_$name.so; fi
        echo "== rep $rep lib $name" | tee -a $out/$tag.txt
        timeout 300 python tools/bench_batch.py --batches 8,32,64 --rounds 20 2>/dev/null | cut -c1-200 | tee -a $out/$tag.txt
        timeout 300 python tools/bench_multiseq.py --S 8 32 64 --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in ('S8','S32','S64'):
    r=(d[k].get('roofline') or {}).get('track_batch') or {}
    print(k, 'frames_s %.0f block %.0f track_batch %.1f us frac %.4f' % (d[k]['frames_s'], d[k]['frames_s_block_until_mapped'], r.get('avg_launch_us') or 0, r.get('frac') or 0))" | tee -a $out/$tag.txt
        unset LD_PRELOAD LSDHIP_LIB
      done
    done
    ;;
  profile)
    # the round's profile set (bench line, rocprofv3 --kernel-trace --stats of the same command, FETCH_SIZE / WRITE_SIZE passes): tools/gpu_profile.sh
    GRAFT_REPO_ROOT=$R bash tools/gpu_profile.sh ${1:-r06} 2>&1 | tail -n 40 | cut -c1-300
    ;;
  pmc)
    # HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of the shared launches of the S = 32 loop against their durations
    GRAFT_REPO_ROOT=$R bash tools/gpu_pmc_multiseq.sh ${1:-r06} 2>&1 | tail -n 60 | cut -c1-400
    ;;
  repeat)
    # the same tests N times in a row (fresh interpreter each time): tools/gpu.sh repeat <N> <pytest args>
    n=$1; shift
    ok=0
    for i in $(seq 1 $n); do
      if timeout 900 python -m pytest -x -q -m gpu "$@" > $out/run_$i.txt 2>&1; then ok=$((ok+1)); else echo "run $i FAILED"; tail -n 30 $out/run_$i.txt | cut -c1-300; fi
      tail -n 1 $out/run_$i.txt
    done
    echo "repeat: $ok of $n runs green" | tee $out/summary.txt
    ;;
  *) echo "unknown step $step"; exit 2;;
esac
