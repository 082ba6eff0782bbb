"""Deterministic synthetic sequences for parity tests and bench.py.

Scene S1: a partly textured, gently undulating surface Z(X,Y) = 2 + 0.1 sin(2πX/1.7) cos(2πY/1.3) seen by a camera moving on a small circle
parallel to the image plane (the motion the reference README recommends for initialisation,
/root/reference/README.md:238), rendered by analytic ray/surface intersection.  Intrinsics follow
/root/reference/lsd_slam_core/calib/pinhole_example_calib.cfg:1 with the reference's convention
(C/util/Undistorter.cpp:340-344): fx = 0.527334 w, fy = 0.827306 h, cx = 0.473568 w - 0.5, cy = 0.499436 h - 0.5.

Scene S2: same geometry with a piecewise-constant (Voronoi) texture => sharp edges, semi-dense pixel
count high (config #3 of BASELINE.json).

Everything here is numpy on the host: it produces *inputs* (uint8 frames, GT depth of frame 0, GT poses);
it is not part of the timed hot path.
"""
import math

import numpy as np


class PCG32:
    """Minimal PCG-XSH-RR 64/32 so that scene parameters do not depend on numpy's generator version."""

    def __init__(self, seed, seq=54):
        self.state = 0
        self.inc = ((seq << 1) | 1) & 0xFFFFFFFFFFFFFFFF
        self.next_u32()
        self.state = (self.state + seed) & 0xFFFFFFFFFFFFFFFF
        self.next_u32()

    def next_u32(self):
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & 0xFFFFFFFFFFFFFFFF
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xFFFFFFFF

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * (self.next_u32() / 4294967296.0)


def intrinsics(w, h):
    """(fx, fy, cx, cy) as float32, reference convention."""
    return np.array([0.527334 * w, 0.827306 * h, 0.473568 * w - 0.5, 0.499436 * h - 0.5], dtype=np.float32)


def _surface(X, Y):
    return 2.0 + 0.1 * np.sin(2 * np.pi * X / 1.7) * np.cos(2 * np.pi * Y / 1.3)


def _surface_grad(X, Y):
    a = 2 * np.pi / 1.7
    b = 2 * np.pi / 1.3
    return (0.1 * a * np.cos(a * X) * np.cos(b * Y), -0.1 * b * np.sin(a * X) * np.sin(b * Y))


class Scene:
    def __init__(self, seq_index=0, kind="S1", n_frames=50, radius=0.08):
        self.kind = kind
        self.n_frames = n_frames
        self.radius = radius
        rng = PCG32(0x5D1A + seq_index)
        if kind == "S1":
            K = 24
            self.fxk = np.array([rng.uniform(0.5, 12.0) * (1 if rng.uniform() < 0.5 else -1) for _ in range(K)])
            self.fyk = np.array([rng.uniform(0.5, 12.0) for _ in range(K)])
            self.phk = np.array([rng.uniform(0, 2 * math.pi) for _ in range(K)])
            a = 1.0 / np.sqrt(np.arange(1, K + 1))
            self.ak = a * (40.0 / math.sqrt(0.5 * float(np.sum(a * a))))
        else:
            rng = PCG32(0xB16 + seq_index)
            # Voronoi sites on a jittered grid over the visible patch [-3,3]^2, ~0.12 world units apart
            g = np.arange(-3.0, 3.0, 0.12)
            gx, gy = np.meshgrid(g, g)
            n = gx.size
            jit = np.array([[rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05)] for _ in range(n)])
            self.sites = np.stack([gx.ravel() + jit[:, 0], gy.ravel() + jit[:, 1]], axis=1)
            self.grey = np.array([rng.uniform(40, 215) for _ in range(n)])
            self.grid0 = -3.0
            self.grid_step = 0.12
            self.grid_n = g.size

    # ---- camera -----------------------------------------------------------------------------------
    def cam_to_world(self, i):
        """Rotation (3x3) and camera centre (3,) of frame i; frame 0 is the world origin with R = I."""
        th = 2 * math.pi * i / 50.0
        C = np.array([self.radius * (math.cos(th) - 1.0), self.radius * math.sin(th), 0.0])
        roll = math.radians(0.5) * math.sin(th)
        c, s = math.cos(roll), math.sin(roll)
        R = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
        return R, C

    def frame_to_ref(self, i, ref=0):
        """GT pose of frame i expressed in frame `ref`: (R, t) with p_ref = R p_i + t."""
        Ri, Ci = self.cam_to_world(i)
        Rr, Cr = self.cam_to_world(ref)
        R = Rr.T @ Ri
        t = Rr.T @ (Ci - Cr)
        return R, t

    # ---- texture ----------------------------------------------------------------------------------
    def texture(self, X, Y):
        if self.kind == "S1":
            T = np.zeros(X.shape)
            for k in range(len(self.ak)):
                T += self.ak[k] * np.sin(2 * np.pi * (self.fxk[k] * X + self.fyk[k] * Y) + self.phk[k])
            # low-frequency envelope: texture-free regions make the map semi-dense (≈100k px at 640x480,
            # the figure /root/reference/README.md:314 quotes) instead of dense
            s = np.sin(2 * np.pi * 0.6 * X + 0.7) * np.sin(2 * np.pi * 0.8 * Y + 0.3)
            E = np.clip((s + 0.2) / 0.4, 0.0, 1.0)
            return 128.0 + E * T
        # S2: nearest jittered-grid site among the 3x3 neighbouring cells
        gi = np.clip(np.rint((X - self.grid0) / self.grid_step).astype(np.int64), 1, self.grid_n - 2)
        gj = np.clip(np.rint((Y - self.grid0) / self.grid_step).astype(np.int64), 1, self.grid_n - 2)
        best = np.full(X.shape, np.inf)
        out = np.zeros(X.shape)
        for dj in (-1, 0, 1):
            for di in (-1, 0, 1):
                idx = (gj + dj) * self.grid_n + (gi + di)
                d = (X - self.sites[idx, 0]) ** 2 + (Y - self.sites[idx, 1]) ** 2
                m = d < best
                best = np.where(m, d, best)
                out = np.where(m, self.grey[idx], out)
        return out

    # ---- rendering --------------------------------------------------------------------------------
    def render(self, i, w, h):
        """uint8 image (h,w) and per-pixel depth (float32, camera z) of frame i."""
        fx, fy, cx, cy = [float(v) for v in intrinsics(w, h)]
        R, C = self.cam_to_world(i)
        u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
        d = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=-1)
        dw = d @ R.T
        lam = np.full(u.shape, 2.0)
        for _ in range(6):  # fixed 6 Newton steps
            X = C[0] + lam * dw[..., 0]
            Y = C[1] + lam * dw[..., 1]
            f = C[2] + lam * dw[..., 2] - _surface(X, Y)
            gx, gy = _surface_grad(X, Y)
            df = dw[..., 2] - gx * dw[..., 0] - gy * dw[..., 1]
            lam = lam - f / df
        X = C[0] + lam * dw[..., 0]
        Y = C[1] + lam * dw[..., 1]
        T = self.texture(X, Y)
        if self.kind != "S1":
            # 1-px box blur (3x3 mean) to get finite gradients at the cell edges
            P = np.pad(T, 1, mode="edge")
            T = sum(P[a:a + h, b:b + w] for a in range(3) for b in range(3)) / 9.0
        img = np.clip(np.rint(T), 0, 255).astype(np.uint8)
        return img, lam.astype(np.float32)


def rot_to_quat(R):
    """(w,x,y,z) of a rotation matrix (double)."""
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s]
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s]
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s]
    q = np.array(q, dtype=np.float64)
    return q / np.linalg.norm(q)


def pose7(R, t):
    """(qw,qx,qy,qz,tx,ty,tz) double — the pose layout of the C ABI."""
    return np.concatenate([rot_to_quat(np.asarray(R, dtype=np.float64)), np.asarray(t, dtype=np.float64)])


def make_sequence(w, h, n_frames, seq_index=0, kind="S1"):
    """Returns (frames uint8 [n,h,w], depth0 float32 [h,w], K float32[4], gt_poses double [n,7] frame->frame0)."""
    sc = Scene(seq_index, kind, n_frames)
    frames = np.zeros((n_frames, h, w), dtype=np.uint8)
    depth0 = None
    poses = np.zeros((n_frames, 7), dtype=np.float64)
    for i in range(n_frames):
        img, depth = sc.render(i, w, h)
        frames[i] = img
        if i == 0:
            depth0 = depth
        R, t = sc.frame_to_ref(i, 0)
        poses[i] = pose7(R, t)
    return frames, depth0, intrinsics(w, h), poses
