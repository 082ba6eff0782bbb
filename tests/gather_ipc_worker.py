"""Worker of tests/test_dataset_gpu.py::test_cpp_loop_keyframe_gather_two_processes_over_ipc: one of two processes on ONE GPU, each
running its own sequence through the C++ loop (different seeds AND different keyframe intervals, so the ranks finish different numbers of
keyframes); the finished keyframes' planes are gathered on rank 0 through the loop's IPC transport (lsdloop_ipc_*), counts in-band.
torch.distributed (gloo) carries the root's 64-byte handle and, for the check, rank 1's own ring.
Usage: python -m torch.distributed.run --nproc-per-node 2 tests/gather_ipc_worker.py <out.json>"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from common import sequence  # noqa: E402
from lsd_slam_amd.driver import DriverLoop  # noqa: E402

out_path = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
w, h, SLOTS = 320, 240, 4
frames, depth0, K, gt = sequence(w, h, 26, seq_index=rank)
imgs = [np.ascontiguousarray(f) for f in frames]
drv = DriverLoop(w, h, K, imgs[0].ctypes.data, depth0, kf_every=10 if rank == 0 else 8, images_on_device=False, device=0)
ring = torch.zeros((SLOTS, 2, h, w), dtype=torch.float32, device="cuda:0")
drv.set_keyframe_ring(ring.data_ptr(), SLOTS)
handle = drv.ipc_init(rank, world, 0)
handles = [None] * world
dist.all_gather_object(handles, handle)
drv.ipc_connect(handles[0])
result = {"ok": True, "detail": []}
for batch in range(2):                      # two gathers: the consumed gate of the second one is exercised
    drv.set_keyframe_ring(ring.data_ptr(), SLOTS)
    done, _ = drv.run([imgs[1 + (batch * 25 + i) % 25].ctypes.data for i in range(25)])
    new = drv.keyframes_exported()
    drv.gather_keyframes(new, 0, None, 0)
    fail, counts, ptr = drv.ipc_result()
    torch.cuda.synchronize()
    mine = ring[:new].cpu().numpy()
    rings = [None] * world
    dist.all_gather_object(rings, (new, mine.tobytes()))
    fails = [None] * world
    dist.all_gather_object(fails, fail)
    if rank == 0:
        import ctypes
        plane2 = 2 * h * w
        buf = np.empty(world * SLOTS * plane2, np.float32)
        from lsd_slam_amd import capi
        capi.check(capi.lib().lsdhip_ctx_read_dev(drv.ctx_handle(), buf.ctypes.data, ctypes.c_void_p(ptr), buf.nbytes))
        got = buf.reshape(world, SLOTS, 2, h, w)
        want_counts = [r[0] for r in rings]
        if counts != want_counts or any(f != 0 for f in fails):
            result["ok"] = False
            result["detail"].append("batch %d: counts %s (want %s), failed waits %s" % (batch, counts, want_counts, fails))
        for r, (n, raw) in enumerate(rings):
            ref = np.frombuffer(raw, np.float32).reshape(n, 2, h, w)
            if n == 0 or not np.array_equal(got[r, :n].view(np.uint32), ref.view(np.uint32)) or float(np.abs(ref).sum()) == 0:
                result["ok"] = False
                result["detail"].append("batch %d: planes of rank %d differ (n = %d)" % (batch, r, n))
        result.setdefault("counts", []).append(counts)
    dist.barrier()
if rank == 0:
    json.dump(result, open(out_path, "w"))
drv.close()
dist.barrier()
dist.destroy_process_group()
