#!/bin/bash
# same-box A/B of two builds of liblsdhip.so on the regulariser: 4K full frame + bands (tools/bench_bands.py) and the S-sequence loop
# tools/gpu_ab_reg.sh <baseline .so>
cd "$(dirname "$0")/.."
BASE=$(readlink -f "$1")
python -m pytest tests/test_gpu_parity.py tests/test_bands_gpu.py tests/test_multiseq_gpu.py -x -q -k "reg or fill or band or depth or keyframe or update or multiseq" 2>&1 | tail -3
for rep in 1 2; do
  for which in base new; do
    if [ $which = base ]; then export LD_PRELOAD=$BASE LSDHIP_LIB=$BASE; else unset LD_PRELOAD LSDHIP_LIB; fi
    python - <<'PY' 2>/dev/null
import sys, time, os, numpy as np
sys.path.insert(0, '.')
import torch
from lsd_slam_amd.bands import NativeBandRegularizer, synth_s3
W, H, passes = 3840, 2160, 20
hyp, mg = synth_s3(W, H)
br = NativeBandRegularizer(W, H, 1, [0], device=0)
br.load(hyp, mg); br.run(2); br.synchronize(); br.load(hyp, mg); br.synchronize()
dts = []
for _ in range(5):
    t0 = time.perf_counter(); br.run(passes); br.synchronize(); dts.append((time.perf_counter() - t0) / passes * 1e3)
import bench
from lsd_slam_amd import synth
w, h = 640, 480
frames, depth0, K, gt = synth.make_sequence(w, h, 60, seq_index=0, kind="s1")
d = torch.from_numpy(frames).cuda(0)
r = bench.multi_seq(torch, w, h, K, d, depth0, 0, None, sizes=(32,))
print(("base" if os.environ.get("LSDHIP_LIB") else "new "), "4K reg ms/pass", " ".join("%.4f" % x for x in dts), "-> %.1f %% of 8 TB/s" % (W * H * 64.0 / (np.median(dts) * 1e-3) / 8e12 * 100), "| S32", round(r["S32"]["frames_s"]), r["S32"]["replicas_bit_identical"])
PY
    unset LD_PRELOAD LSDHIP_LIB
  done
done
