"""Test-side engine for lsd_slam_amd.bands: the CPU oracle doing a window's regularisation pass, so that the band
decomposition (windows, halo exchange, ownership) can be checked on CPU.  Test infrastructure only."""
import numpy as np


class OracleBandEngine:
    def __init__(self, w, window_rows):
        from oracle import pyoracle as po
        self.po = po
        self.w, self.h = w, window_rows
        self.K = np.array([0.5 * w, 0.5 * w, 0.5 * w, 0.5 * window_rows], np.float32)
        self.kf = po.Frame(0, np.zeros((window_rows, w), np.uint8), self.K)
        self.map = po.DepthMap(w, window_rows, self.K)

    def load(self, hyp_window, maxgrad_window):
        self.kf.set_maxgrad(maxgrad_window)
        self.map.set(self.kf, hyp_window)

    def run_pass(self):
        self.map.stage("fillholes")
        self.map.stage("regularize")

    def get(self):
        return self.map.get()

    def new_buffer(self, nrows):
        import torch
        return torch.empty(nrows * self.w * 32, dtype=torch.uint8)

    def pack_rows(self, local_row0, nrows, buf):
        import torch
        rows = np.ascontiguousarray(self.map.get()[local_row0:local_row0 + nrows])
        buf.copy_(torch.from_numpy(rows.view(np.uint8).reshape(-1)))

    def unpack_rows(self, local_row0, nrows, buf):
        hyp = self.map.get().copy()
        hyp[local_row0:local_row0 + nrows] = buf.numpy().view(self.po.HYP_DTYPE).reshape(nrows, self.w)
        self.map.set(self.kf, hyp)
