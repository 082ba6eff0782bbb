"""The committed measurement artefacts keep the bench contract: profiles/rNN_bench.json is one bench.py line with the roofline and
cpu_baseline objects, and the rocprofv3 / PMC summaries it refers to are next to it."""
import glob
import json
import os

from common import ROOT


def _latest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    assert files, pattern
    return files[-1]


def test_latest_bench_line_has_the_contract_fields():
    d = json.load(open(_latest("r*_bench.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert str(base.get("metric", ""))[:20].lower().split()[0] in d["metric"].lower() or "frames" in d["metric"].lower()
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1
    assert d["validation"]["ok"] is True and d["validation"]["tracked_good"] == d["validation"]["frames"]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) < 1e-6          # frames/s x s/frame


def test_profile_summaries_of_the_round_are_committed():
    tag = os.path.basename(_latest("r*_bench.json")).split("_")[0]
    for name in ("kernel_stats.csv", "pmc_traffic.json", "notes.md", "track_step_durations.txt"):
        assert os.path.exists(os.path.join(ROOT, "profiles", "%s_%s" % (tag, name))), name
    rows = open(os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv")).read()
    assert "k_track_step" in rows and "k_observe" in rows and "k_reg_fused" in rows
    pmc = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic.json")))
    assert any("k_track_step" in k for k in pmc["kernels"])
