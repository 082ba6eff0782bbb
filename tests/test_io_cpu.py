"""include/lsd_slam_hip_io.hpp without a GPU: the reference's calibration file convention,
the keyframeMsg wire format of lsd_slam_viewer, and the viewer's point-cloud export restated in numpy."""
import os
import struct
import subprocess

import numpy as np
import pytest

from common import ROOT


@pytest.fixture(scope="module")
def io_outputs(tmp_path_factory):
    from lsd_slam_amd import build
    build.build()
    d = tmp_path_factory.mktemp("io")
    exe = str(d / "io_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "io_test.cpp"),
                           "-o", exe, "-L" + os.path.join(ROOT, "lsd_slam_amd"), "-llsdhip", "-Wl,-rpath," + os.path.join(ROOT, "lsd_slam_amd")])
    calib = d / "calib.cfg"
    calib.write_text("0.527334 0.827306 0.473568 0.499436 0\n752 480\nnone\n752 480\n")   # the reference's pinhole example
    out = subprocess.check_output([exe, str(calib), str(d)]).decode()
    return d, out


def test_calibration_convention(io_outputs):
    d, out = io_outputs
    k = out.splitlines()[0].split()
    assert k[:3] == ["K", "752", "480"]
    fx, fy, cx, cy = map(float, k[3:])
    # C/util/Undistorter.cpp:340-344: relative values scale with the size, principal point shifted by half a pixel
    assert fx == pytest.approx(0.527334 * 752, rel=1e-6) and fy == pytest.approx(0.827306 * 480, rel=1e-6)
    assert cx == pytest.approx(0.473568 * 752 - 0.5, rel=1e-6) and cy == pytest.approx(0.499436 * 480 - 0.5, rel=1e-6)
    assert "pgm 1" in out


def _parse_msg(b):
    o = 0
    id_, = struct.unpack_from("<i", b, o); o += 4
    t, = struct.unpack_from("<d", b, o); o += 8
    kf = b[o]; o += 1
    c2w = struct.unpack_from("<7f", b, o); o += 28
    fx, fy, cx, cy = struct.unpack_from("<4f", b, o); o += 16
    h, w = struct.unpack_from("<2I", b, o); o += 8
    n, = struct.unpack_from("<I", b, o); o += 4
    pc = np.frombuffer(b, dtype=np.dtype([("idepth", "<f4"), ("var", "<f4"), ("color", "u1", 4)]), count=n // 12, offset=o)
    assert o + n == len(b)
    return id_, t, kf, np.array(c2w), (fx, fy, cx, cy), w, h, pc


def test_keyframe_msg_wire_format_and_point_cloud(io_outputs):
    d, out = io_outputs
    id_, t, kf, c2w, (fx, fy, cx, cy), w, h, pc = _parse_msg((d / "kf.msg").read_bytes())
    assert (id_, t, kf, w, h) == (7, 1.25, 1, 32, 16) and len(pc) == 32 * 16
    # Sophus Sim3 data(): quaternion (x, y, z, w) with norm = scale, then the translation
    assert np.linalg.norm(c2w[:4]) == pytest.approx(2.0, rel=1e-6) and np.allclose(c2w[4:], [1, -2, 0.5])
    # V/KeyFrameDisplay.cpp:269-340 in numpy
    pc = pc.reshape(h, w)
    q = c2w[:4] / 2.0
    R = np.array([[1 - 2 * (q[1] ** 2 + q[2] ** 2), 2 * (q[0] * q[1] - q[2] * q[3]), 2 * (q[0] * q[2] + q[1] * q[3])],
                  [2 * (q[0] * q[1] + q[2] * q[3]), 1 - 2 * (q[0] ** 2 + q[2] ** 2), 2 * (q[1] * q[2] - q[0] * q[3])],
                  [2 * (q[0] * q[2] - q[1] * q[3]), 2 * (q[1] * q[2] + q[0] * q[3]), 1 - 2 * (q[0] ** 2 + q[1] ** 2)]])
    pts = []
    for y in range(1, h - 1):
        for x in range(1, w - 1):
            p = pc[y, x]
            if p["idepth"] <= 0:
                continue
            depth = np.float32(1) / p["idepth"]
            d4 = (depth * depth) ** 2
            if p["var"] * d4 > 1 or p["var"] * d4 * 4.0 > 1:
                continue
            near = 0
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    qn = pc[y + dy, x + dx]
                    if qn["idepth"] > 0 and (qn["idepth"] - 1.0 / depth) ** 2 < 2 * p["var"]:
                        near += 1
            if near < 5:
                continue
            v = np.array([(x - cx) / fx, (y - cy) / fy, 1.0]) * depth
            pts.append(list(2.0 * R @ v + c2w[4:]) + [p["color"][2] / 255.0])
    raw = (d / "pc.ply").read_bytes()
    head, body = raw.split(b"end_header\n", 1)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % len(pts))
    assert b"property float intensity" in head
    got = np.frombuffer(body, "<f4").reshape(-1, 4)
    assert len(got) == len(pts) > 50 and ("points %d" % len(pts)) in out
    assert np.allclose(got, np.array(pts), rtol=1e-4, atol=1e-5)
