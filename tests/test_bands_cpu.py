"""Row-band decomposition of the regulariser (lsd_slam_amd/bands.py) checked on CPU: the plan
arithmetic, and — with the oracle as the per-window engine — that banded passes with halo exchange reproduce the
full-frame result exactly, in one process and across two gloo ranks."""
import os
import subprocess
import sys

import numpy as np
import pytest

from common import ROOT
from lsd_slam_amd.bands import HALO_BOTTOM, HALO_TOP, BandPlan, BandRegularizer, LocalComm, synth_s3


@pytest.mark.parametrize("H,world", [(2160 + 16 - 2160 % 16, 8), (480, 4), (192, 3), (64, 2), (48, 1)])
def test_plan_covers_rows_and_windows_are_sufficient(H, world):
    p = BandPlan(H, world)
    rows = []
    for (y0, y1), (a, b) in zip(p.owned, p.window):
        rows += list(range(y0, y1))
        assert (b - a) == p.window_rows and p.window_rows % 16 == 0 and 0 <= a and b <= H
        assert a == 0 or a <= y0 - HALO_TOP
        assert b == H or b >= y1 + HALO_BOTTOM
    assert rows == list(range(H))
    for r in range(world):
        a, b = p.window[r]
        got = sorted(list(range(*p.owned[r])) + [y for (_, lo, n) in p.recv_list(r) for y in range(lo, lo + n)])
        assert got == list(range(a, b))                       # every window row is owned or received exactly once
        for (d, lo, n) in p.send_list(r):
            assert (r, lo, n) in p.recv_list(d) and p.owned[r][0] <= lo and lo + n <= p.owned[r][1]
    assert sum(len(p.send_list(r)) for r in range(world)) == sum(len(p.recv_list(r)) for r in range(world))


def _full_frame(oracle, hyp, maxgrad, passes):
    from band_engines import OracleBandEngine
    h, w = hyp.shape
    e = OracleBandEngine(w, h)
    e.load(hyp, maxgrad)
    for _ in range(passes):
        e.run_pass()
    return e.get()


def _assert_rows_equal(got, ref, what):
    for k in ("isValid", "blacklisted", "validity_counter"):
        assert np.array_equal(got[k], ref[k]), (what, k)
    v = ref["isValid"] > 0
    for k in ("idepth", "idepth_var", "idepth_smoothed", "idepth_var_smoothed"):
        assert np.array_equal(got[k][v].view(np.uint32), ref[k][v].view(np.uint32)), (what, k)


@pytest.mark.parametrize("world", [2, 3, 5])
def test_banded_passes_equal_full_frame_oracle(oracle, world):
    from band_engines import OracleBandEngine
    w, H, passes = 160, 192, 3
    hyp, maxgrad = synth_s3(w, H)
    ref = _full_frame(oracle, hyp, maxgrad, passes)
    assert (ref["isValid"] != hyp["isValid"]).sum() > 100      # the passes do create and delete hypotheses
    plan = BandPlan(H, world)
    engines = [OracleBandEngine(w, plan.window_rows) for _ in range(world)]
    br = BandRegularizer(plan, engines, LocalComm(), list(range(world)))
    br.load(hyp, maxgrad)
    br.run(passes)
    for r, rows in br.owned_rows().items():
        y0, y1 = plan.owned[r]
        _assert_rows_equal(rows, ref[y0:y1], "band %d of %d" % (r, world))


WORKER = r"""
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from lsd_slam_amd.bands import BandPlan, BandRegularizer, DistComm, synth_s3
from band_engines import OracleBandEngine
world = int(sys.argv[2])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=world)
w, H, passes = 160, int(sys.argv[3]), 3
hyp, maxgrad = synth_s3(w, H)
plan = BandPlan(H, world)
br = BandRegularizer(plan, [OracleBandEngine(w, plan.window_rows)], DistComm(), [dist.get_rank()])
br.load(hyp, maxgrad)
br.run(passes)
rows = br.owned_rows()[dist.get_rank()]
np.save(os.path.join({out!r}, "band%d.npy" % dist.get_rank()), rows)
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world,H", [(2, 96), (8, 256)])
def test_banded_passes_over_gloo_ranks(oracle, tmp_path, world, H):
    """one band per process, halo rows through torch.distributed point-to-point (gloo here, RCCL on GPUs): 2 ranks, and the 8 ranks
    BASELINE.json configs[4] names (every interior band exchanges with two neighbours; the middle bands' windows overlap on both sides)"""
    port = 29500 + (os.getpid() + 7 * world) % 2000
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, port=port, out=str(tmp_path)))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(H)]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    w, passes = 160, 3
    hyp, maxgrad = synth_s3(w, H)
    ref = _full_frame(oracle, hyp, maxgrad, passes)
    plan = BandPlan(H, world)
    for r in range(world):
        y0, y1 = plan.owned[r]
        _assert_rows_equal(np.load(tmp_path / ("band%d.npy" % r)), ref[y0:y1], "rank %d" % r)


def test_native_band_plan_equals_python_plan():
    """The index arithmetic of the C++ band loop (liblsdhip_driver.so lsdband_plan: owned rows, windows, halo segments) against
    BandPlan for a sweep of map heights and band counts — no GPU involved (what the multi-GPU exchange sends where cannot be run
    on a one-GPU box, so at least its plan is pinned here)."""
    import ctypes as C
    from lsd_slam_amd import driver
    from lsd_slam_amd.bands import BandPlan
    L = driver.lib()
    for H in (64, 256, 384, 480, 1024, 2160 - 2160 % 16, 2160):
        if H % 16:
            continue
        for world in (1, 2, 3, 5, 8, 13):
            if world > H // 16:
                continue
            plan = BandPlan(H, world)
            wr = C.c_int()
            lay = (C.c_int * (4 * world))()
            cap = 8 * world
            seg = (C.c_int * (4 * cap))()
            n = L.lsdband_plan(H, world, C.byref(wr), lay, seg, cap)
            assert n >= 0 and wr.value == plan.window_rows, (H, world)
            for r in range(world):
                assert (lay[4 * r], lay[4 * r + 1]) == plan.owned[r] and (lay[4 * r + 2], lay[4 * r + 3]) == plan.window[r], (H, world, r)
            want = [(r, s, lo, cnt) for r in range(world) for (s, lo, cnt) in plan.recv_list(r)]
            got = [(seg[4 * k], seg[4 * k + 1], seg[4 * k + 2], seg[4 * k + 3]) for k in range(n)]
            assert got == want, (H, world)
            # what a band sends is what the others receive from it
            for r in range(world):
                sends = sorted((d, lo, cnt) for (d, s, lo, cnt) in got if s == r)
                assert sends == sorted(plan.send_list(r)), (H, world, r)


def test_native_band_tile_runs_separate_edge_from_interior():
    """The overlapped pass of the C++ band loop issues a window's tile rows (8 map rows) in two parts: `edge` runs before the halo
    exchange forks, `interior` runs beside it.  Race-freedom is index arithmetic, checked here without a GPU for a sweep of heights
    and band counts, from BandPlan's lists alone: the runs cover exactly the tile rows that hold owned rows, once; an interior tile
    row contains no row any other band receives (send_list) and its tiles' read footprint (4 rows above the tile row, 4 below: the
    fused pass's tile halo) contains no row this band receives (recv_list) — so the exchange may write those while it runs."""
    import ctypes as C
    from lsd_slam_amd import driver
    from lsd_slam_amd.bands import BandPlan
    L = driver.lib()
    seen_interior = 0
    for H in (64, 256, 384, 512, 1024, 2160):
        if H % 16:
            continue
        for world in (1, 2, 3, 5, 8, 13):
            if world > H // 16:
                continue
            plan = BandPlan(H, world)
            for r in range(world):
                buf = (C.c_int * (3 * 64))()
                n = L.lsdband_tile_runs(H, world, r, buf, 64)
                assert 0 < n <= 64, (H, world, r, n)
                runs = [(buf[3 * k], buf[3 * k + 1], buf[3 * k + 2]) for k in range(n)]
                a, _ = plan.window[r]
                y0, y1 = plan.owned[r]
                o0, o1 = y0 - a, y1 - a
                tiles = [t for (t0, cnt, _) in runs for t in range(t0, t0 + cnt)]
                assert tiles == list(range(o0 // 8, (o1 + 7) // 8)), (H, world, r)          # every owned row, once, in order
                assert all(runs[k][2] != runs[k + 1][2] for k in range(n - 1)), (H, world, r)  # runs are maximal
                sent = set()
                for (_, lo, cnt) in plan.send_list(r):
                    sent.update(range(lo - a, lo - a + cnt))
                received = set()
                for (_, lo, cnt) in plan.recv_list(r):
                    received.update(range(lo - a, lo - a + cnt))
                assert received == set(range(plan.window_rows)) - set(range(o0, o1)), (H, world, r)   # all non-owned rows arrive
                for (t0, cnt, edge) in runs:
                    for t in range(t0, t0 + cnt):
                        produces = set(range(8 * t, 8 * t + 8))
                        reads = set(range(8 * t - 4, 8 * t + 12))
                        clean = not (produces & sent) and not (reads & received)
                        assert clean == (edge == 0), (H, world, r, t, edge)
                        seen_interior += (edge == 0)
    assert seen_interior > 100
