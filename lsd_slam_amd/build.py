"""Build liblsdhip.so (hipcc, gfx950) in-tree.  Used by __graft_entry__.build() and by developers.

Flags that matter for parity: -ffp-contract=off (no FMA contraction: the oracle is built the same way), default
IEEE-correct f32 division / sqrt (hipcc's -fhip-fp32-correctly-rounded-divide-sqrt default), no fast-math.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "liblsdhip.so")
SOURCES = ["host_math.cpp", "frame.hip", "tracker.hip", "depthmap.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
         "-Wno-unused-but-set-variable"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(SRC, s) for s in SOURCES] + [os.path.join(SRC, "lsdhip_internal.hpp"), os.path.join(SRC, "pose_math.hpp"),
                                                       os.path.join(HERE, "..", "include", "lsdhip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_trace(verbose=False):
    """Developer build with per-phase timestamps inside k_track_step (tools/phase_trace.py); never loaded by default."""
    out = os.path.join(HERE, "liblsdhip_trace.so")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-DLSD_PHASE_TRACE", "-x", "hip"] + [os.path.join(SRC, s) for s in SOURCES] + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-x", "hip"] + [os.path.join(SRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    if "--trace" in sys.argv:
        build_trace(verbose=True)
    else:
        build(force="--force" in sys.argv, verbose=True)
