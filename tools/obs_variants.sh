for V in 0 1 2 3; do
  export LSDHIP_OBS_VARIANT=$V
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$V -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline-events > /tmp/b_$V.json 2>/dev/null
  cd $GRAFT_REPO_ROOT
  python - <<PY
import csv, glob, json
f = glob.glob("/tmp/prof_$V/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_observe" in r["Name"]:
        print("variant $V: %s avg=%.1f us" % (r["Name"][:40], float(r["AverageNs"]) / 1e3), "fps(profiled)=%.0f" % json.load(open("/tmp/b_$V.json"))["value"])
PY
done
