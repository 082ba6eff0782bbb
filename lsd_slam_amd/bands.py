"""Row-band decomposition of the depth-map regulariser across GPUs (BASELINE.json configs[4]:
3840x2160 maps, regularisation tiled across 8 GPUs).

One pass = regularizeDepthMapFillHoles + regularizeDepthMap(false, VAL_SUM_MIN_FOR_KEEP) (C/DepthEstimation/DepthMap.cpp:
656-720, :758-880) — what updateKeyframe runs after the observation step.  Rank r owns rows [y0_r, y1_r) of the full
H-row map and keeps a *window* of rows [a_r, b_r) around them in an ordinary depth map of that height:

    K6 at an owned row needs the post-K5 map at rows +-2, K5 there needs the pre-pass map at rows +-2 more, and the
    kernels skip the first 3 / last 2 rows of whatever map they are given (the reference's own border rule).  So a
    window that reaches 5 rows above and 4 rows below the owned rows (or ends at the border of the full map, where the
    window's border rule *is* the full map's) reproduces the full-frame result on the owned rows exactly, with plain
    whole-map kernels and no global scan (the validity integral of the reference is only ever used as 5x5 window sums).

After every pass the non-owned rows of each window are refreshed from their owners: one exchange step per pass
(point-to-point sends between neighbouring ranks, 29 B per pixel of halo), then the next pass.  `BandPlan` is the pure
index arithmetic; an *engine* does the per-window work (HipBandEngine: liblsdhip.so on one GPU; the tests plug the CPU
oracle in to check the arithmetic on CPU); a *comm* moves halo rows (LocalComm: all bands in one process; DistComm:
torch.distributed, RCCL on GPUs / gloo in tests).
"""
import numpy as np

HALO_TOP, HALO_BOTTOM = 5, 4
PACK_BYTES_PER_PX = 29


class BandPlan:
    """Owned rows, windows and exchange lists for `world` bands over an H-row map (window heights are multiples of 16,
    the granularity the frame pyramids require)."""

    def __init__(self, H, world, window_multiple=16):
        if H % window_multiple:
            raise ValueError("H must be a multiple of %d" % window_multiple)
        self.H, self.world = H, world
        base, extra = divmod(H, world)
        self.owned = []
        y = 0
        for r in range(world):
            n = base + (1 if r < extra else 0)
            self.owned.append((y, y + n))
            y += n
        need = max(b - a for a, b in self.owned) + HALO_TOP + HALO_BOTTOM
        self.window_rows = min(H, -(-need // window_multiple) * window_multiple)
        if any(b - a < 1 for a, b in self.owned):
            raise ValueError("more bands than rows")
        self.window = []
        for (y0, y1) in self.owned:
            a = min(max(y0 - HALO_TOP, 0), H - self.window_rows)
            b = a + self.window_rows
            assert a <= max(y0 - HALO_TOP, 0) and b >= min(y1 + HALO_BOTTOM, H)
            self.window.append((a, b))

    def recv_list(self, r):
        """[(src rank, global row0, nrows)]: the non-owned rows of r's window, grouped by owner."""
        a, b = self.window[r]
        out = []
        for s in range(self.world):
            if s == r:
                continue
            lo, hi = max(a, self.owned[s][0]), min(b, self.owned[s][1])
            if hi > lo:
                out.append((s, lo, hi - lo))
        return out

    def send_list(self, r):
        """[(dst rank, global row0, nrows)]: owned rows of r that lie in other ranks' windows."""
        out = []
        for d in range(self.world):
            if d == r:
                continue
            for (s, lo, n) in self.recv_list(d):
                if s == r:
                    out.append((d, lo, n))
        return out

    def halo_bytes_per_pass(self, w):
        return sum(n for r in range(self.world) for (_, _, n) in self.recv_list(r)) * w * PACK_BYTES_PER_PX


class HipBandEngine:
    """One window on one GPU through the C ABI: an ordinary (w x window_rows) depth map."""

    def __init__(self, w, window_rows, device=0):
        import lsd_slam_amd as la
        self.la = la
        self.w, self.h = w, window_rows
        K = np.array([0.5 * w, 0.5 * w, 0.5 * w, 0.5 * window_rows], np.float32)   # unused by the regulariser
        self.device = device
        self.ctx = la.Context(w, window_rows, K, device=device)
        self.kf = la.Frame(self.ctx, 0, np.zeros((window_rows, w), np.uint8))
        self.map = la.DepthMap(self.ctx)

    def load(self, hyp_window, maxgrad_window):
        self.kf.setMaxGradients(maxgrad_window)
        self.map.setCurrentDepthMap(self.kf, hyp_window)

    def run_pass(self):
        self.map.stage("fill_regularize")

    def get(self):
        return self.map.currentDepthMap()

    # halo rows as one packed device buffer (torch uint8 tensor on this GPU)
    def new_buffer(self, nrows):
        import torch
        return torch.empty(nrows * self.w * PACK_BYTES_PER_PX, dtype=torch.uint8, device="cuda:%d" % self.device)

    def pack_rows(self, local_row0, nrows, buf):
        self.map.copyRows(local_row0, nrows, buf.data_ptr(), False)

    def unpack_rows(self, local_row0, nrows, buf):
        self.map.copyRows(local_row0, nrows, buf.data_ptr(), True)


class LocalComm:
    """All bands live in this process (one GPU or the CPU tests): halo rows are handed over directly."""

    def exchange(self, plan, engines):
        staged = []
        for r, eng in enumerate(engines):
            a = plan.window[r][0]
            for (s, lo, n) in plan.recv_list(r):
                buf = engines[s].new_buffer(n)
                engines[s].pack_rows(lo - plan.window[s][0], n, buf)
                staged.append((eng, lo - a, n, buf))
        for (eng, row, n, buf) in staged:   # unpack after every pack: a pass's exchange reads the pre-exchange state
            eng.unpack_rows(row, n, buf)


class DistComm:
    """One band per rank; torch.distributed point-to-point (backend nccl = RCCL over xGMI on GPUs, gloo in CPU tests)."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def exchange(self, plan, engines):
        (eng,) = engines
        r, dist = self.rank, self.dist
        a = plan.window[r][0]
        ops, recvs = [], []
        for (d, lo, n) in plan.send_list(r):
            buf = eng.new_buffer(n)
            eng.pack_rows(lo - a, n, buf)
            ops.append(dist.P2POp(dist.isend, buf, d))
        for (s, lo, n) in plan.recv_list(r):
            buf = eng.new_buffer(n)
            ops.append(dist.P2POp(dist.irecv, buf, s))
            recvs.append((lo - a, n, buf))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for (row, n, buf) in recvs:
            eng.unpack_rows(row, n, buf)


class BandRegularizer:
    """Runs `passes` regularisation passes on a full (H x w) hypothesis map split into bands.
    engines: the engines of the bands this process holds (all of them with LocalComm, one with DistComm)."""

    def __init__(self, plan, engines, comm, ranks):
        self.plan, self.engines, self.comm, self.ranks = plan, engines, comm, ranks

    def load(self, hyp_full, maxgrad_full):
        for eng, r in zip(self.engines, self.ranks):
            a, b = self.plan.window[r]
            eng.load(np.ascontiguousarray(hyp_full[a:b]), np.ascontiguousarray(maxgrad_full[a:b]))

    def run(self, passes):
        for p in range(passes):
            for eng in self.engines:
                eng.run_pass()
            if p + 1 < passes:
                self.comm.exchange(self.plan, self.engines)

    def owned_rows(self):
        """{rank: hypothesis rows of the band it owns}"""
        out = {}
        for eng, r in zip(self.engines, self.ranks):
            a = self.plan.window[r][0]
            y0, y1 = self.plan.owned[r]
            out[r] = eng.get()[y0 - a:y1 - a]
        return out


class NativeBandRegularizer:
    """The same decomposition run by the C++ loop (liblsdhip_driver.so, `lsdband_*`): every window's pass and every halo
    refresh is queued on the context's stream and nothing synchronises the host between passes.  Windows of this process
    refresh each other with one map -> map copy launch per pass; rows owned by other processes travel packed through RCCL
    (grouped ncclSend / ncclRecv on the same stream).  `ranks`: the consecutive bands this process holds."""

    def __init__(self, w, H, world, ranks, device=0):
        import ctypes as C
        from . import driver
        self.C, self.L = C, driver.lib()
        self.w, self.H, self.world, self.ranks = w, H, world, list(ranks)
        assert self.ranks == list(range(self.ranks[0], self.ranks[0] + len(self.ranks)))
        h = C.c_void_p()
        driver._check(self.L.lsdband_create(device, w, H, world, self.ranks[0], len(self.ranks), C.byref(h)))
        self.h_ = h
        self.window_rows = self.L.lsdband_window_rows(h)
        self.plan = BandPlan(H, world)
        assert self.plan.window_rows == self.window_rows
        for r in range(world):      # the C++ index arithmetic is the Python plan's
            out = (C.c_int * 4)()
            driver._check(self.L.lsdband_layout(h, r, out))
            assert (out[0], out[1]) == self.plan.owned[r] and (out[2], out[3]) == self.plan.window[r], (r, list(out))
        self._check = driver._check

    def comm_init(self, unique_id, nprocs, proc, proc_of_band):
        C = self.C
        arr = (C.c_int * self.world)(*proc_of_band)
        uid = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        self._check(self.L.lsdband_comm_init(self.h_, uid, nprocs, proc, arr))

    def ipc_init(self, nprocs, proc, proc_of_band):
        """second transport (processes of one node, IPC-mapped mailboxes, no RCCL): returns this process's 64-byte handle"""
        C = self.C
        arr = (C.c_int * self.world)(*proc_of_band)
        out = (C.c_ubyte * 64)()
        self._check(self.L.lsdband_ipc_init(self.h_, int(nprocs), int(proc), arr, out))
        return bytes(out)

    def ipc_connect(self, handles):
        """handles: the 64-byte handles of all processes in process order"""
        C = self.C
        blob = b"".join(handles)
        buf = (C.c_ubyte * len(blob)).from_buffer_copy(blob)
        self._check(self.L.lsdband_ipc_connect(self.h_, buf))

    def ipc_failed(self):
        return int(self.L.lsdband_ipc_failed(self.h_))

    def load(self, hyp_full, maxgrad_full):
        from .capi import HYP_DTYPE
        for i, r in enumerate(self.ranks):
            a, b = self.plan.window[r]
            hyp = np.ascontiguousarray(hyp_full[a:b], dtype=HYP_DTYPE)
            mg = np.ascontiguousarray(maxgrad_full[a:b], dtype=np.float32)
            self._check(self.L.lsdband_load(self.h_, i, hyp.ctypes.data, mg.ctypes.data))

    def run(self, passes):
        self._check(self.L.lsdband_run(self.h_, int(passes)))

    def synchronize(self):
        self._check(self.L.lsdband_synchronize(self.h_))

    def set_packed_exchange(self, on):
        """test hook: local windows exchange through pack -> copy -> unpack (the multi-GPU wire format) instead of map -> map"""
        self._check(self.L.lsdband_set_packed_exchange(self.h_, int(on)))

    def set_overlap(self, on):
        """exchange with other processes under the interior rows of a pass (default) / after the pass"""
        self._check(self.L.lsdband_set_overlap(self.h_, int(on)))

    def halo_bytes_per_pass(self):
        return int(self.L.lsdband_halo_bytes_per_pass(self.h_))

    def owned_rows(self):
        from .capi import HYP_DTYPE
        out = {}
        for i, r in enumerate(self.ranks):
            buf = np.zeros((self.window_rows, self.w), HYP_DTYPE)
            self._check(self.L.lsdband_get(self.h_, i, buf.ctypes.data))
            a = self.plan.window[r][0]
            y0, y1 = self.plan.owned[r]
            out[r] = buf[y0 - a:y1 - a]
        return out

    def close(self):
        if self.h_:
            self.L.lsdband_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def synth_s3(w, h, seed=0xC5):
    """Scene S3: hypothesis planes generated directly — validity Bernoulli(0.3) clustered along
    edges of a blocky pattern, idepth = 1/Z + N(0, 0.02^2), idepth_var in U[1e-4, 0.05], validity_counter in {0..50},
    blacklisted in {0,-1,-2} on 5 % of the pixels; maxGradients above minUseGrad on the edge band.  numpy Generator
    (PCG64) with a fixed seed: deterministic for a given numpy major version; tests regenerate rather than store it."""
    from .capi import HYP_DTYPE
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[0:h, 0:w]
    cell = 24
    edge = ((xx % cell) < 5) | ((yy % cell) < 5)            # an edge lattice, ~37 % of the pixels
    valid = edge & (rng.random((h, w)) < 0.8)                # ~30 % valid
    Z = 2.0 + 0.25 * np.sin(2 * np.pi * xx / 640.0) * np.cos(2 * np.pi * yy / 480.0)
    hyp = np.zeros((h, w), HYP_DTYPE)
    hyp["isValid"] = valid
    hyp["idepth"] = (1.0 / Z + rng.normal(0, 0.02, (h, w))).astype(np.float32)
    hyp["idepth_var"] = rng.uniform(1e-4, 0.05, (h, w)).astype(np.float32)
    hyp["idepth_smoothed"] = hyp["idepth"]
    hyp["idepth_var_smoothed"] = hyp["idepth_var"]
    hyp["validity_counter"] = rng.integers(0, 51, (h, w)).astype(np.int32)
    bl = np.zeros((h, w), np.int32)
    m = rng.random((h, w)) < 0.05
    bl[m] = -rng.integers(1, 3, int(m.sum()))
    hyp["blacklisted"] = bl
    hyp["nextStereoFrameMinID"] = 0
    maxgrad = np.where(edge, 20.0, 1.0).astype(np.float32)
    return hyp, maxgrad
