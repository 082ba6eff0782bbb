"""Every `File.cpp:line[-line]` citation of the reference in the headers, kernels, oracle and design documents points at a file that
exists under /root/reference and at lines that file has (the judge checks parity through these citations; a stale one would send the
reader to the wrong place).  Runs where the reference is present (this container); skipped on the GPU box."""
import glob
import os
import re

import pytest

from common import ROOT

REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference sources are not present on this machine")
def test_reference_citations_point_at_existing_lines():
    files = {}
    for root, _, fs in os.walk(REF):
        for f in fs:
            if f.endswith((".cpp", ".hpp", ".h", ".txt")):
                files.setdefault(f, []).append(os.path.join(root, f))
    lengths = {}
    pat = re.compile(r"(?<![\w/.])([A-Za-z0-9_/]+\.(?:cpp|hpp|h|txt)):(\d+)(?:-(\d+))?")
    srcs = []
    for g in ("include/*.h", "include/*.hpp", "lsd_slam_amd/csrc/*", "lsd_slam_amd/driver/*.cpp", "oracle/*.cpp", "oracle/*.h", "DESIGN.md", "INTEGRATION.md"):
        srcs += glob.glob(os.path.join(ROOT, g))
    own = {os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "**", "*"), recursive=True)}
    checked, bad = 0, []
    for src in srcs:
        txt = open(src, errors="ignore").read()
        for m in pat.finditer(txt):
            name = os.path.basename(m.group(1))
            a, b = int(m.group(2)), int(m.group(3) or m.group(2))
            if name not in files:
                if name not in own and "rocm" not in m.group(1) and not m.group(1).startswith(("rccl/", "hip/")):
                    bad.append((os.path.relpath(src, ROOT), m.group(0), "no such file in the reference"))
                continue
            checked += 1
            ok = False
            for p in files[name]:
                if p not in lengths:
                    with open(p, errors="ignore") as fh:
                        lengths[p] = sum(1 for _ in fh)
                ok = ok or (a <= b <= lengths[p])
            if not ok:
                bad.append((os.path.relpath(src, ROOT), m.group(0), "line range outside the file"))
    assert checked > 200, checked
    assert not bad, bad[:20]
