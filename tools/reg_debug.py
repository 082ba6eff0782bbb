import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lsd_slam_amd.bands import NativeBandRegularizer, synth_s3
W, H, passes = 3840, 2160, 20
hyp, maxgrad = synth_s3(W, H)
for bands in (1, 8, 1, 1):
    br = NativeBandRegularizer(W, H, bands, list(range(bands)), device=0)
    br.load(hyp, maxgrad); br.run(2); br.synchronize(); br.load(hyp, maxgrad); br.synchronize()
    for rep in range(3):
        t0 = time.perf_counter(); br.run(passes); br.synchronize(); dt = time.perf_counter() - t0
        print("bands %d rep %d: %.3f ms per pass" % (bands, rep, dt / passes * 1e3))
    br.close()
