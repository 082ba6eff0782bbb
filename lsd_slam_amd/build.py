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
# kernels allowed to spill a few loop-invariant registers (long, memory-bound batch kernels where occupancy matters more;
# the latency-critical single-job kernels must stay at zero)
SCRATCH_OK = ()
SOURCES = ["host_math.cpp", "frame.hip", "tracker.hip", "depthmap.hip", "sim3.hip"]
# -disable-promote-alloca-to-lds: a private array the optimiser cannot split must show up as scratch (and fail the
# guard below) instead of silently moving to LDS, where indexing it by thread costs a read of the AQL dispatch packet
# in host memory (~20 us per launch, measured — profiles/r01_notes.md).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-mllvm", "-disable-promote-alloca-to-lds=1", "-Rpass-analysis=kernel-resource-usage",
         "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
         "-Wno-unused-but-set-variable"]


def _compile_checked(cmd):
    """Run hipcc; fail the build if any kernel needs scratch memory (every hot kernel must stay in registers)."""
    import re
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    diag = [l for l in r.stderr.splitlines() if "kernel-resource-usage" not in l]
    if diag:
        sys.stderr.write("\n".join(diag) + "\n")
    if r.returncode != 0:
        raise subprocess.CalledProcessError(r.returncode, cmd)
    name = None
    bad = []
    for l in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", l)
        if m and int(m.group(1)) > 0 and not any(k in (name or "") for k in SCRATCH_OK):
            bad.append((name, int(m.group(1))))
    if bad:
        raise RuntimeError("kernels using scratch memory: %r" % bad)


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(SRC, s) for s in SOURCES] + [os.path.join(SRC, "lsdhip_internal.hpp"), os.path.join(SRC, "pose_math.hpp"),
                                                       os.path.join(SRC, "track_device.hpp"), os.path.join(SRC, "rcp_exact.hpp"),
                                                       os.path.join(HERE, "..", "include", "lsdhip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_trace(verbose=False):
    """Developer build with per-phase timestamps inside k_track_step (tools/phase_trace.py); never loaded by default."""
    out = os.path.join(HERE, "liblsdhip_trace.so")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-DLSD_PHASE_TRACE", "-x", "hip"] + [os.path.join(SRC, s) for s in SOURCES] + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    _compile_checked(cmd)
    return out


def build_variant(name, defines, verbose=False):
    """Developer build of liblsdhip with extra -D flags (liblsdhip_<name>.so); loaded through LSDHIP_LIB + LD_PRELOAD by the A/B tools."""
    out = os.path.join(HERE, "liblsdhip_%s.so" % name)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-D" + d for d in defines] + ["-x", "hip"] + [os.path.join(SRC, s) for s in SOURCES] + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    _compile_checked(cmd)
    return out


def build_variant_if_stale(name, defines):
    out = os.path.join(HERE, "liblsdhip_%s.so" % name)
    if os.path.exists(out) and os.path.exists(OUT) and os.path.getmtime(out) >= os.path.getmtime(OUT):
        return out
    return build_variant(name, defines)


DRIVER_OUT = os.path.join(HERE, "liblsdhip_driver.so")
DRIVER_SRC = os.path.join(HERE, "driver", "slam_loop.cpp")


def build_driver(force=False, verbose=False):
    """liblsdhip_driver.so: the C++ host loop (include/lsd_slam_hip.hpp) — plain g++, links liblsdhip.so."""
    deps = [DRIVER_SRC, os.path.join(HERE, "driver", "dataset_slam.cpp"), os.path.join(HERE, "..", "include", "lsd_slam_hip_io.hpp"),
            os.path.join(HERE, "..", "include", "lsd_slam_hip.hpp"), os.path.join(HERE, "..", "include", "lsdhip.h"),
            os.path.join(HERE, "..", "include", "lsdhip_driver.h"), OUT]
    if not force and os.path.exists(DRIVER_OUT) and all(os.path.getmtime(d) <= os.path.getmtime(DRIVER_OUT) for d in deps):
        return DRIVER_OUT
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", DRIVER_SRC, "-o", DRIVER_OUT,
           "-L" + HERE, "-llsdhip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    # dataset_slam: the ROS-free image-folder driver (lsd_slam_amd/driver/dataset_slam.cpp)
    cmd2 = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(HERE, "..", "include"),
            os.path.join(HERE, "driver", "dataset_slam.cpp"), "-o", os.path.join(HERE, "dataset_slam"), "-L" + HERE, "-llsdhip",
            "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd2))
    subprocess.check_call(cmd2)
    return DRIVER_OUT


RCP_CHECK_SRC = os.path.join(HERE, "..", "tools", "rcp_exhaustive.hip")
RCP_CHECK_OUT = os.path.join(HERE, "rcp_exhaustive.bin")


def build_rcp_check(force=False, verbose=False):
    """The exhaustive check of lsd_rcp_exact (csrc/rcp_exact.hpp) against `1.0f / x` over all 2^32 inputs: a small device program run by
    tests/test_rcp_gpu.py.  Built with the library so that it travels to the GPU box with it."""
    deps = [RCP_CHECK_SRC, os.path.join(SRC, "rcp_exact.hpp")]
    if not force and os.path.exists(RCP_CHECK_OUT) and all(os.path.getmtime(d) <= os.path.getmtime(RCP_CHECK_OUT) for d in deps):
        return RCP_CHECK_OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-Wno-unused-value", RCP_CHECK_SRC, "-o", RCP_CHECK_OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return RCP_CHECK_OUT


def build(force=False, verbose=False):
    build_rcp_check(force, verbose)
    if not force and not needs_build():
        build_driver(False, verbose)
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-x", "hip"] + [os.path.join(SRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    _compile_checked(cmd)
    build_driver(True, verbose)
    return OUT


if __name__ == "__main__":
    if "--trace" in sys.argv:
        build_trace(verbose=True)
    else:
        build(force="--force" in sys.argv, verbose=True)
