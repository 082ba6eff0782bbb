"""N>1 path on CPU: world_size-2 gloo run of the keyframe gather (the path's only exchange step)."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from lsd_slam_amd.multigpu import KeyframeGather
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
kg = KeyframeGather((2, 12, 16), torch.device("cpu"))
for k in range(3):     # three keyframes per rank; record = f(rank, k)
    kg.submit(lambda buf, k=k: buf.copy_(torch.full((2, 12, 16), float(100 * rank + k))), keep=True)
kg.wait(keep=True)
if rank == 0:
    assert len(kg.collected) == 3
    for k, rec in enumerate(kg.collected):
        assert rec.shape == (world, 2, 12, 16)
        for r in range(world):   # result of the N-rank run == concatenation of N single-rank runs, bitwise
            assert torch.equal(rec[r], torch.full((2, 12, 16), float(100 * r + k)))
    print("GATHER_OK")
# the bench's exchange step: one gather per batch of the first `new` ring slots
from lsd_slam_amd.multigpu import gather_keyframe_ring
ring = torch.zeros((5, 2, 6, 8))
for batch, new in enumerate((3, 0, 5)):
    for k in range(new):
        ring[k].fill_(1000.0 * batch + 100 * rank + k)
    recv = [torch.zeros_like(ring) for _ in range(world)] if rank == 0 else None
    nbytes = gather_keyframe_ring(ring, new, recv)
    assert nbytes == new * 2 * 6 * 8 * 4
    if rank == 0:
        for r in range(world):
            for k in range(new):
                assert torch.equal(recv[r][k], torch.full((2, 6, 8), 1000.0 * batch + 100 * r + k))
            assert torch.count_nonzero(recv[r][new:]) == 0
if rank == 0:
    print("RING_OK")
dist.barrier()
dist.destroy_process_group()
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_keyframe_gather_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GATHER_OK" in out.stdout and "RING_OK" in out.stdout


def test_single_rank_is_a_noop_gather():
    import torch
    from lsd_slam_amd.multigpu import KeyframeGather
    kg = KeyframeGather((2, 4, 4), torch.device("cpu"))
    kg.submit(lambda b: b.fill_(3.0), keep=True)
    kg.wait()
    assert kg.world == 1 and torch.equal(kg.collected[0][0], torch.full((2, 4, 4), 3.0))


import pytest


@pytest.mark.parametrize("gpus", [2, 8])
def test_bench_gpus_flag_starts_the_ranks_itself(gpus):
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run (one rank per GPU); the
    --dry-run leg runs the same rendezvous / barrier / max-over-ranks plumbing on gloo and prints n_gpus == N — at 2 ranks and at
    the 8 ranks the two sharded configurations of BASELINE.json name (configs[3], configs[4])."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--dry-run", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["steps"] == 3 and d["warmup"] == 1 and d["dry_run"] is True
    # what an N-rank run adds to the line: one rate per rank (BASELINE.json configs[3]) and the row-band regulariser across the ranks
    # over BOTH transports (configs[4]: lsdband_run's RCCL halo exchange, and the IPC mailboxes) — bench.py::bands_distributed
    assert d["per_rank_frames_s"] == [None] * gpus
    assert set(d["extra_configs"]["reg_3840x2160_bands"]) >= {"rccl", "ipc"}


def test_bench_line_skeleton_of_a_one_gpu_run():
    """What a one-GPU bench line carries beside `value` (printed by --dry-run without a GPU): the CPU baseline in BOTH execution models with
    the like-for-like speed-ups and the mapping iterations per frame of both sides, and the roofline objects of the S-sequence leg."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert set(d["speedup_vs_cpu_baseline"]) == {"block_until_mapped", "pipelined", "updates_per_frame"}
    assert "pipelined" in d["cpu_baseline"] and "updates_per_frame" in d["cpu_baseline"]["pipelined"]
    assert set(d["extra_configs"]["multi_seq"]["S32"]["roofline"]) >= {"track_batch", "frame_pyramids", "observe", "regularise", "idepth_pyramids", "keyframe_change"}


def test_cpu_baseline_two_thread_model():
    """bench.py's cpu_baseline.pipelined: the reference's two-thread model on the CPU library with the mapper one frame behind — every frame
    tracked well, the frame after a keyframe change dropped unmapped, the rest mapped (the GPU loop's pipelined mode, frame for frame)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from oracle import pyoracle as po
    from lsd_slam_amd import synth
    w, h, n = 160, 128, 45
    frames, depth0, K, gt = synth.make_sequence(w, h, 30)
    po.build()
    rec = bench._cpu_loop_pipelined(po, po.lib(fast=True), frames, depth0, K, n, w, h, po.SSE)
    assert rec["tracked_good"] == n and rec["frames"] == n
    # keyframes every 10 mapped-or-dropped frames; each change is followed by exactly one dropped frame
    assert abs(rec["keyframes_per_frame"] - rec["dropped_frames_per_frame"]) <= 1.0 / n
    assert abs(rec["updates_per_frame"] + rec["keyframes_per_frame"] + rec["dropped_frames_per_frame"] - 1.0) < 1e-9
    seq = bench._cpu_loop(po, po.lib(fast=True), frames, depth0, K, n, w, h, po.SSE)
    assert seq["tracked_good"] == n and seq["updates_per_frame"] > rec["updates_per_frame"]
