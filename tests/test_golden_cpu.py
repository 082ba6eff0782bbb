"""The committed golden fixture (tests/golden/s1_160x128.npz, made by tests/golden/make_golden.py FROM THE REFERENCE's own
sources, oracle/_ref) pins the oracle: it must reproduce every entry bit for bit; any change of the oracle's arithmetic or of
the synthetic generator shows up here, on CPU, also where /root/reference is absent."""
import importlib.util
import os

import numpy as np

from common import ROOT

GOLDEN = os.path.join(ROOT, "tests", "golden", "s1_160x128.npz")


def _maker():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_synthetic_generator_reproduces_fixture_inputs():
    from lsd_slam_amd import synth
    g = np.load(GOLDEN)
    frames, depth0, K, gt = synth.make_sequence(160, 128, 4)
    assert np.array_equal(frames, g["frames"])
    assert np.array_equal(depth0.astype(np.float32), g["depth0"])
    assert np.array_equal(np.asarray(K, np.float32), g["K"])


def test_oracle_reproduces_golden_outputs(oracle):
    m = _maker()
    g = np.load(GOLDEN)
    assert "oracle/_ref" in str(g["generated_by"])
    now = m.compute(g["frames"], g["depth0"], g["K"])
    for k, v in now.items():
        assert np.array_equal(np.asarray(v), g[k]), "oracle output %s drifted from the committed fixture" % k
