"""Worker of tests/test_bands_gpu.py::test_native_band_loop_two_processes_over_ipc: one of N processes that share ONE GPU, each
holding a contiguous range of row bands of the regulariser; the halo rows travel through the C++ band loop's IPC transport
(lsdband_ipc_*: pack into the peer's mapped mailbox, ready / consumed flags on the stream).  torch.distributed (gloo) only carries
the 64-byte handles and collects the owned rows on rank 0, which compares them with the full-frame result, bit for bit.
Usage: python -m torch.distributed.run --nproc-per-node N tests/band_ipc_worker.py <w> <H> <bands> <passes> <out.json> [overlap]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist  # noqa: E402
from lsd_slam_amd.bands import NativeBandRegularizer, synth_s3  # noqa: E402

w, H, bands, passes, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
overlap = int(sys.argv[6]) if len(sys.argv) > 6 else 1
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
per = bands // world
mine = list(range(rank * per, (rank + 1) * per))
proc_of = [r // per for r in range(bands)]
hyp, maxgrad = synth_s3(w, H)
br = NativeBandRegularizer(w, H, bands, mine, device=0)          # every process on the same GPU
handle = br.ipc_init(world, rank, proc_of)
handles = [None] * world
dist.all_gather_object(handles, handle)
br.ipc_connect(handles)
br.set_overlap(overlap)
br.load(hyp, maxgrad)
br.synchronize()
dist.barrier()
# two calls: the exchange ordinals (buffer parities, consumed gates) continue across calls
# (split after the FIRST pass, where the regulariser is far from its fixed point: a missing refresh between the calls shows)
br.run(1)
br.run(passes)                                                    # starts with the halo refresh that belongs behind the first call's pass
br.synchronize()
failed = br.ipc_failed()
rows = br.owned_rows()
gathered = [None] * world
dist.all_gather_object(gathered, {r: np.ascontiguousarray(v).tobytes() for r, v in rows.items()})
fails = [None] * world
dist.all_gather_object(fails, failed)
if rank == 0:
    full = NativeBandRegularizer(w, H, 1, [0], device=0)
    full.load(hyp, maxgrad)
    full.run(passes + 1)
    full.synchronize()
    ref = full.owned_rows()[0]
    ok, detail = all(f == 0 for f in fails), []
    for part in gathered:
        for r, raw in part.items():
            y0, y1 = br.plan.owned[r]
            rows_r = np.frombuffer(raw, dtype=ref.dtype).reshape(y1 - y0, w)
            for k in ("isValid", "blacklisted", "validity_counter", "idepth", "idepth_var", "idepth_smoothed", "idepth_var_smoothed"):
                got = rows_r[k]
                want = ref[k][y0:y1]
                if k in ("isValid", "blacklisted", "validity_counter"):
                    same = np.array_equal(got, want)
                else:
                    v = ref["isValid"][y0:y1] > 0
                    same = np.array_equal(got[v].view(np.uint32), want[v].view(np.uint32))
                if not same:
                    ok = False
                    detail.append("band %d plane %s differs" % (r, k))
    json.dump({"ok": bool(ok), "flag_waits_failed": fails, "detail": detail[:10], "valid": int((ref["isValid"] > 0).sum())}, open(out_path, "w"))
dist.barrier()
dist.destroy_process_group()
