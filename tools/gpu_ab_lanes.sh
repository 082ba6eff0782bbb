cd $GRAFT_REPO_ROOT
python -m pytest tests/test_multiseq_gpu.py -x -q 2>&1 | tail -3
for v in 8 1 16 8 1; do
  echo "LSDHIP_KF_LANES=$v"
  LSDHIP_KF_LANES=$v python - <<'PY' 2>/dev/null
import sys, json, numpy as np
sys.path.insert(0, '.')
import torch, bench
from lsd_slam_amd import synth
w, h = 640, 480
frames, depth0, K, gt = synth.make_sequence(w, h, 60, seq_index=0, kind="s1")
d = torch.from_numpy(frames).cuda(0)
r = bench.multi_seq(torch, w, h, K, d, depth0, 0, None, sizes=(8, 32))
print({k: (round(v["frames_s"]), round(v["ms_per_step"], 4), v["tracked_good"], v["replicas_bit_identical"]) for k, v in r.items() if k.startswith("S")})
PY
done
