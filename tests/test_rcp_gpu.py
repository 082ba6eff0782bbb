"""lsd_rcp_exact (lsd_slam_amd/csrc/rcp_exact.hpp) stands in for `1.0f / x` in kernels whose outputs are held to bit equality with the
reference (the regulariser's neighbour weights, propagateDepth's 1 / z, the idepth pyramid's pooling, the point generation of the
trackers).  The claim — the same bits for every input — is checked on the device itself, exhaustively: all 2^32 bit patterns against the
division the compiler emits (tools/rcp_exhaustive.hip, which includes the product's header; built by lsd_slam_amd/build.py)."""
import json
import os
import subprocess

import pytest

from common import ROOT

pytestmark = pytest.mark.gpu


def test_exact_reciprocal_equals_ieee_division_for_every_float():
    exe = os.path.join(ROOT, "lsd_slam_amd", "rcp_exhaustive.bin")
    if not os.path.exists(exe):
        from lsd_slam_amd import build
        build.build_rcp_check()
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["inputs_checked"] == 2 ** 32
    assert d["mismatch_lsd_rcp_exact"] == 0, d
    # the bare hardware estimate is NOT the quotient (1 ulp), and the correction alone is not enough outside the normal range: the check
    # would notice if it compared a function with itself
    assert d["normal_recip_normal"]["mismatch_rcp_hw"] > 10 ** 8 and d["normal_recip_normal"]["mismatch_rcp_plus_one_correction"] == 0
    assert d["normal_recip_denormal"]["mismatch_rcp_plus_one_correction"] > 10 ** 7
    for cls in ("zero", "denormal", "normal_recip_normal", "normal_recip_denormal", "inf", "nan"):
        assert d[cls]["mismatch_lsd_rcp_exact"] == 0, (cls, d[cls])
