"""ctypes binding of liblsdhip_driver.so (include/lsdhip_driver.h): the C++ sequence loop, so that a batch of frames runs
without the Python interpreter between frames.  No fallback: a missing library raises."""
import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
DRIVER_PATH = os.path.join(_HERE, "liblsdhip_driver.so")
_lib = None


class LoopStats(C.Structure):
    _fields_ = [("seconds_track", C.c_double), ("seconds_map", C.c_double), ("seconds_keyframe", C.c_double), ("frames", C.c_longlong),
                ("updates", C.c_longlong), ("keyframes", C.c_longlong), ("evaluations", C.c_longlong), ("tracked_good", C.c_longlong),
                ("level_evaluations", C.c_longlong * 5), ("track_launches", C.c_longlong), ("dropped", C.c_longlong)]


EXPORTED_SYMBOLS = ["lsdloop_create", "lsdloop_destroy", "lsdloop_run", "lsdloop_get_stats", "lsdloop_reset_stats",
                    "lsdloop_copy_keyframe_planes", "lsdloop_set_keyframe_ring", "lsdloop_keyframes_exported", "lsdloop_ctx",
                    "lsdloop_last_error", "lsdloop_keep_keyframes", "lsdloop_keyframe_log", "lsdloop_set_live_queue", "lsdloop_set_pipeline", "lsdloop_set_speculation", "lsdloop_observe_time", "lsdloop_comm_unique_id", "lsdloop_comm_init", "lsdloop_comm_destroy",
                    "lsdloop_gather_keyframes", "lsdband_create", "lsdband_destroy", "lsdband_window_rows", "lsdband_layout", "lsdband_load",
                    "lsdband_get", "lsdband_comm_init", "lsdband_run", "lsdband_synchronize", "lsdband_halo_bytes_per_pass", "lsdband_set_packed_exchange", "lsdband_set_overlap", "lsdband_tile_runs", "lsdband_plan", "lsdband_ipc_init", "lsdband_ipc_connect", "lsdband_ipc_failed", "lsdloop_gather_counts", "lsdloop_ipc_init", "lsdloop_ipc_connect", "lsdloop_ipc_result", "lsdloop_observe_work",
                    "lsdloopbatch_create", "lsdloopbatch_destroy", "lsdloopbatch_run", "lsdloopbatch_get_stats", "lsdloopbatch_ctx", "lsdloopbatch_set_keyframe_phases", "lsdloopbatch_set_pipeline", "lsdloopbatch_set_coarse_min_jobs", "lsdloopbatch_dropped",
                    "lsdloopbatch_keep_keyframes", "lsdloopbatch_keyframe_log", "lsdloopbatch_last_result", "lsdloopbatch_download_map"]


def lib():
    global _lib
    if _lib is None:
        capi.lib()  # liblsdhip.so first (same file the driver links against)
        if not os.path.exists(DRIVER_PATH):
            raise RuntimeError("liblsdhip_driver.so is missing (%s): run __graft_entry__.build()" % DRIVER_PATH)
        L = C.CDLL(DRIVER_PATH)
        vp, i = C.c_void_p, C.c_int
        L.lsdloop_create.restype = i
        L.lsdloop_create.argtypes = [i, i, i, vp, vp, i, vp, i, C.POINTER(vp)]
        L.lsdloop_destroy.restype = None
        L.lsdloop_destroy.argtypes = [vp]
        L.lsdloop_run.restype = i
        L.lsdloop_run.argtypes = [vp, vp, i, i, vp]
        L.lsdloop_get_stats.restype = i
        L.lsdloop_get_stats.argtypes = [vp, C.POINTER(LoopStats)]
        L.lsdloop_reset_stats.restype = i
        L.lsdloop_reset_stats.argtypes = [vp]
        L.lsdloop_copy_keyframe_planes.restype = i
        L.lsdloop_copy_keyframe_planes.argtypes = [vp, vp, vp]
        L.lsdloop_set_keyframe_ring.restype = i
        L.lsdloop_set_keyframe_ring.argtypes = [vp, vp, i]
        L.lsdloop_keyframes_exported.restype = C.c_longlong
        L.lsdloop_keyframes_exported.argtypes = [vp]
        L.lsdloop_keep_keyframes.restype = i
        L.lsdloop_keep_keyframes.argtypes = [vp, i]
        L.lsdloop_keyframe_log.restype = i
        L.lsdloop_keyframe_log.argtypes = [vp, vp, vp, i]
        L.lsdloop_set_live_queue.restype = i
        L.lsdloop_set_live_queue.argtypes = [vp, i]
        L.lsdloop_set_pipeline.restype = i
        L.lsdloop_set_pipeline.argtypes = [vp, i]
        L.lsdloop_set_speculation.restype = i
        L.lsdloop_set_speculation.argtypes = [vp, i, i]
        L.lsdloop_observe_time.restype = i
        L.lsdloop_observe_time.argtypes = [vp, vp, vp]
        L.lsdloop_observe_work.restype = i
        L.lsdloop_observe_work.argtypes = [vp, vp]
        L.lsdloop_comm_unique_id.restype = i
        L.lsdloop_comm_unique_id.argtypes = [vp]
        for name, args in (("lsdband_create", [i, i, i, i, i, i, C.POINTER(vp)]), ("lsdband_window_rows", [vp]), ("lsdband_layout", [vp, i, vp]),
                           ("lsdband_load", [vp, i, vp, vp]), ("lsdband_get", [vp, i, vp]), ("lsdband_comm_init", [vp, vp, i, i, vp]),
                           ("lsdband_run", [vp, i]), ("lsdband_synchronize", [vp]), ("lsdband_set_packed_exchange", [vp, i]), ("lsdband_set_overlap", [vp, i]), ("lsdband_tile_runs", [i, i, i, vp, i]), ("lsdband_plan", [i, i, vp, vp, vp, i]),
                           ("lsdband_ipc_init", [vp, i, i, vp, vp]), ("lsdband_ipc_connect", [vp, vp]), ("lsdband_ipc_failed", [vp])):
            getattr(L, name).restype = i
            getattr(L, name).argtypes = args
        L.lsdband_destroy.restype = None
        L.lsdband_destroy.argtypes = [vp]
        L.lsdband_halo_bytes_per_pass.restype = C.c_longlong
        L.lsdband_halo_bytes_per_pass.argtypes = [vp]
        L.lsdloop_comm_init.restype = i
        L.lsdloop_comm_init.argtypes = [vp, vp, i, i]
        L.lsdloop_comm_destroy.restype = i
        L.lsdloop_comm_destroy.argtypes = [vp]
        L.lsdloop_gather_keyframes.restype = i
        L.lsdloop_gather_keyframes.argtypes = [vp, i, i, vp, C.c_longlong]
        for name, args in (("lsdloop_gather_counts", [vp, vp, i]), ("lsdloop_ipc_init", [vp, i, i, i, vp]), ("lsdloop_ipc_connect", [vp, vp]),
                           ("lsdloop_ipc_result", [vp, vp, vp])):
            getattr(L, name).restype = i
            getattr(L, name).argtypes = args
        L.lsdloop_ctx.restype = vp
        L.lsdloop_ctx.argtypes = [vp]
        L.lsdloopbatch_create.restype = i
        L.lsdloopbatch_create.argtypes = [i, i, i, vp, i, vp, i, vp, i, C.POINTER(vp)]
        L.lsdloopbatch_destroy.restype = None
        L.lsdloopbatch_destroy.argtypes = [vp]
        L.lsdloopbatch_run.restype = i
        L.lsdloopbatch_run.argtypes = [vp, vp, i, vp]
        L.lsdloopbatch_get_stats.restype = i
        L.lsdloopbatch_get_stats.argtypes = [vp, vp]
        L.lsdloopbatch_ctx.restype = vp
        L.lsdloopbatch_ctx.argtypes = [vp]
        L.lsdloopbatch_set_keyframe_phases.restype = i
        L.lsdloopbatch_set_keyframe_phases.argtypes = [vp, vp]
        L.lsdloopbatch_set_pipeline.restype = i
        L.lsdloopbatch_set_pipeline.argtypes = [vp, i]
        L.lsdloopbatch_set_coarse_min_jobs.restype = i
        L.lsdloopbatch_set_coarse_min_jobs.argtypes = [vp, i]
        L.lsdloopbatch_dropped.restype = C.c_longlong
        L.lsdloopbatch_dropped.argtypes = [vp, i]
        L.lsdloopbatch_keep_keyframes.restype = i
        L.lsdloopbatch_keep_keyframes.argtypes = [vp, i]
        L.lsdloopbatch_keyframe_log.restype = i
        L.lsdloopbatch_keyframe_log.argtypes = [vp, i, vp, vp, i]
        L.lsdloopbatch_last_result.restype = i
        L.lsdloopbatch_last_result.argtypes = [vp, i, vp]
        L.lsdloopbatch_download_map.restype = i
        L.lsdloopbatch_download_map.argtypes = [vp, i, vp]
        L.lsdloop_last_error.restype = C.c_char_p
        L.lsdloop_last_error.argtypes = []
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise capi.LsdHipError("liblsdhip_driver call failed (%d): %s" % (rc, (lib().lsdloop_last_error() or b"").decode()))
    return rc


class DriverLoopBatch:
    """lsd_slam_hip::SlamLoopBatch: S sequences sharing one GPU, one frame of each per step.  Images are raw pointers (device pointers when
    images_on_device, else host pointers)."""

    def __init__(self, w, h, K, first_image_ptrs, depth0s, kf_every=10, images_on_device=True, device=0):
        self.L = lib()
        self.S = len(first_image_ptrs)
        K4 = np.ascontiguousarray(K, dtype=np.float32)
        self._d0 = [np.ascontiguousarray(d, dtype=np.float32) for d in depth0s]
        imgs = (C.c_void_p * self.S)(*first_image_ptrs)
        d0s = (C.c_void_p * self.S)(*[d.ctypes.data for d in self._d0])
        h_ = C.c_void_p()
        _check(self.L.lsdloopbatch_create(device, w, h, K4.ctypes.data, self.S, imgs, int(images_on_device), d0s, kf_every, C.byref(h_)))
        self.h_ = h_

    def close(self):
        if getattr(self, "h_", None):
            self.L.lsdloopbatch_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        self.close()

    def run(self, image_ptrs, want_poses=False):
        """image_ptrs: n steps x S pointers (list of lists) -> (steps run, poses n x S x 7 or None)"""
        n = len(image_ptrs)
        flat = [p for step in image_ptrs for p in step]
        assert len(flat) == n * self.S
        arr = (C.c_void_p * (n * self.S))(*flat)
        poses = np.zeros((n, self.S, 7), np.float64) if want_poses else None
        done = _check(self.L.lsdloopbatch_run(self.h_, arr, n, poses.ctypes.data if want_poses else None))
        return done, poses

    def stats(self):
        """per sequence: dict(frames, tracked_good, updates, keyframes, evaluations, lost)"""
        out = np.zeros((self.S, 6), np.int64)
        _check(self.L.lsdloopbatch_get_stats(self.h_, out.ctypes.data))
        keys = ("frames", "tracked_good", "updates", "keyframes", "evaluations", "lost")
        return [dict(zip(keys, (int(v) for v in row))) for row in out]

    def ctx_handle(self):
        return C.c_void_p(self.L.lsdloopbatch_ctx(self.h_))

    def set_coarse_min_jobs(self, min_jobs):
        """sequences per step from which a tracking batch walks its coarse levels in one workgroup per sequence (0: never)"""
        _check(self.L.lsdloopbatch_set_coarse_min_jobs(self.h_, int(min_jobs)))

    def set_pipeline(self, on=True):
        """tracking beside mapping for all sequences, the mapper one frame behind (before the first run)"""
        _check(self.L.lsdloopbatch_set_pipeline(self.h_, int(bool(on))))

    def dropped(self):
        """per sequence: frames tracked on a keyframe the mapper had already replaced (pipelined loops)"""
        return [int(self.L.lsdloopbatch_dropped(self.h_, s)) for s in range(self.S)]

    def keep_keyframes(self, on=True):
        _check(self.L.lsdloopbatch_keep_keyframes(self.h_, int(on)))

    def keyframe_log(self, s, max_entries=1024):
        """(rescale factors, point counts) of the keyframes sequence s promoted since keep_keyframes(True); synchronises"""
        sc = np.zeros(max_entries, np.float64)
        pts = np.zeros(max_entries, np.int64)
        n = min(_check(self.L.lsdloopbatch_keyframe_log(self.h_, int(s), sc.ctypes.data, pts.ctypes.data, max_entries)), max_entries)
        return sc[:n], pts[:n]

    def last_result(self, s):
        r = capi.TrackResult()
        _check(self.L.lsdloopbatch_last_result(self.h_, int(s), C.byref(r)))
        return r

    def download_map(self, s, w, h):
        out = np.zeros(w * h, capi.HYP_DTYPE)
        _check(self.L.lsdloopbatch_download_map(self.h_, int(s), out.ctypes.data))
        return out.reshape(h, w)

    def set_keyframe_phases(self, phases):
        """phase[s] in [0, kf_every): how old sequence s's first keyframe already is (unsynchronised keyframe changes)"""
        a = (C.c_int * self.S)(*[int(p) for p in phases])
        _check(self.L.lsdloopbatch_set_keyframe_phases(self.h_, a))


class DriverLoop:
    """lsd_slam_hip::SlamLoop (include/lsd_slam_hip.hpp).  Images are raw pointers: device pointers when
    images_on_device, else host pointers (numpy .ctypes.data)."""

    def __init__(self, w, h, K, first_image_ptr, depth0, kf_every=10, images_on_device=True, device=0):
        self.L = lib()
        K4 = np.ascontiguousarray(K, dtype=np.float32)
        d0 = np.ascontiguousarray(depth0, dtype=np.float32)
        h_ = C.c_void_p()
        _check(self.L.lsdloop_create(device, w, h, K4.ctypes.data, C.c_void_p(first_image_ptr), int(images_on_device),
                                     d0.ctypes.data, kf_every, C.byref(h_)))
        self.h_ = h_
        self.w, self.h = w, h

    def close(self):
        if getattr(self, "h_", None):
            self.L.lsdloop_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        self.close()

    def run(self, image_ptrs, stop_at_keyframe=False, want_poses=False):
        """Runs the frames; returns (frames consumed, poses or None)."""
        n = len(image_ptrs)
        arr = (C.c_void_p * n)(*image_ptrs)
        poses = np.zeros((n, 7), np.float64) if want_poses else None
        done = _check(self.L.lsdloop_run(self.h_, arr, n, int(stop_at_keyframe), poses.ctypes.data if want_poses else None))
        return done, (poses[:done] if want_poses else None)

    def stats(self):
        s = LoopStats()
        _check(self.L.lsdloop_get_stats(self.h_, C.byref(s)))
        return s

    def reset_stats(self):
        _check(self.L.lsdloop_reset_stats(self.h_))

    def copy_keyframe_planes(self, idepth_ptr, var_ptr):
        _check(self.L.lsdloop_copy_keyframe_planes(self.h_, C.c_void_p(idepth_ptr), C.c_void_p(var_ptr)))

    def set_keyframe_ring(self, ring_ptr, slots):
        """finished keyframes' (idepth, idepthVar) planes go to slot (count % slots) of the device buffer; None switches it off"""
        _check(self.L.lsdloop_set_keyframe_ring(self.h_, C.c_void_p(ring_ptr) if ring_ptr else None, slots))

    def gather_counts(self, world):
        out = (C.c_int * world)()
        n = _check(self.L.lsdloop_gather_counts(self.h_, out, world))
        return list(out)[:n]

    def ipc_init(self, rank, world, root=0):
        """second transport of the gather (processes of one node, no RCCL): returns this rank's 64-byte handle (zeros unless root)"""
        out = (C.c_ubyte * 64)()
        _check(self.L.lsdloop_ipc_init(self.h_, rank, world, root, out))
        self._ipc_world = world
        return bytes(out)

    def ipc_connect(self, root_handle):
        buf = (C.c_ubyte * 64).from_buffer_copy(root_handle)
        _check(self.L.lsdloop_ipc_connect(self.h_, buf))

    def ipc_result(self):
        """(failed waits, counts per rank, device pointer of the gathered planes) — counts / pointer only on the root"""
        counts = (C.c_int * self._ipc_world)()
        ptr = C.c_void_p()
        fail = self.L.lsdloop_ipc_result(self.h_, counts, C.byref(ptr))
        if fail < 0:
            _check(fail)
        return fail, list(counts), ptr.value

    def keep_keyframes(self, on=True):
        _check(self.L.lsdloop_keep_keyframes(self.h_, int(on)))

    def keyframe_log(self, max_entries=4096):
        """(scales, numPoints) of the keyframes kept since keep_keyframes(True); synchronises"""
        sc = np.zeros(max_entries, np.float64)
        pts = np.zeros(max_entries, np.int64)
        n = _check(self.L.lsdloop_keyframe_log(self.h_, sc.ctypes.data, pts.ctypes.data, max_entries))
        n = min(n, max_entries)
        return sc[:n], pts[:n]

    def set_live_queue(self, frames):
        _check(self.L.lsdloop_set_live_queue(self.h_, int(frames)))

    def set_speculation(self, trials, finest_level_workgroups=0):
        _check(self.L.lsdloop_set_speculation(self.h_, int(trials), int(finest_level_workgroups)))

    def set_pipeline(self, on=True):
        """tracking beside mapping, the mapper one frame behind (lsdloop_set_pipeline); before the first run()"""
        _check(self.L.lsdloop_set_pipeline(self.h_, int(bool(on))))

    def observe_time(self):
        """(ms, calls) of the observe kernel alone, sampled while profiling is on; synchronises"""
        ms, n = C.c_double(), C.c_longlong()
        _check(self.L.lsdloop_observe_time(self.h_, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def observe_work(self):
        """(launches counted, searched pixels, walk steps) of the sampled k_observe launches; synchronises"""
        out = np.zeros(3, np.float64)
        _check(self.L.lsdloop_observe_work(self.h_, out.ctypes.data))
        return float(out[0]), float(out[1]), float(out[2])

    @staticmethod
    def comm_unique_id():
        """128 bytes from ncclGetUniqueId (call on one rank, hand to the others)"""
        buf = (C.c_ubyte * 128)()
        _check(lib().lsdloop_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        _check(self.L.lsdloop_comm_init(self.h_, buf, rank, world))

    def comm_destroy(self):
        _check(self.L.lsdloop_comm_destroy(self.h_))

    def gather_keyframes(self, count, root, recv_ptr, stride_floats):
        """RCCL send/recv of the first `count` ring slots to `root`, enqueued on the loop's stream (no host synchronisation)"""
        _check(self.L.lsdloop_gather_keyframes(self.h_, int(count), int(root), C.c_void_p(recv_ptr) if recv_ptr else None, int(stride_floats)))

    def keyframes_exported(self):
        return int(self.L.lsdloop_keyframes_exported(self.h_))

    def ctx_handle(self):
        return self.L.lsdloop_ctx(self.h_)
