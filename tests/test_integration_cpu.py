"""Link-level proof of the drop-in (VERDICT r05 missing #3, INTEGRATION.md path B): the reference's OWN class declarations
(lsd_slam_core/src/Tracking/SE3Tracker.h, DepthEstimation/DepthMap.h — unmodified) get their member definitions from
integration/hip_backed/*.cpp, which call the C ABI of liblsdhip.so; integration/lsd_slam_core.patch is the change to the reference's
build that swaps the two translation units.  Here (CPU): the patch applies to the reference's CMakeLists.txt, the replaced library builds
from the reference's sources where they lie, exports the reference's symbols and imports the C ABI.  tests/test_hipbacked_gpu.py runs it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/lsd_slam_core"
HB = os.path.join(ROOT, "oracle", "_ref", "liblsd_ref_hipbacked.so")


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout exists only in the build container")
def test_patch_applies_to_the_reference_build(tmp_path):
    dst = tmp_path / "lsd_slam_core"
    dst.mkdir()
    shutil.copy(os.path.join(REF, "CMakeLists.txt"), dst / "CMakeLists.txt")
    r = subprocess.run(["patch", "-p1", "-d", str(tmp_path), "-i", os.path.join(ROOT, "integration", "lsd_slam_core.patch")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    text = (dst / "CMakeLists.txt").read_text()
    assert "integration/hip_backed/SE3Tracker_hip.cpp" in text and "integration/hip_backed/DepthMap_hip.cpp" in text
    assert "src/Tracking/SE3Tracker.cpp" not in text and "src/DepthEstimation/DepthMap.cpp" not in text
    assert "target_link_libraries(lsdslam lsdhip " in text
    # everything else of the build is untouched: the other 20 translation units of lsd_SOURCE_FILES still come from the reference
    for keep in ("src/SlamSystem.cpp", "src/Tracking/TrackingReference.cpp", "src/Tracking/Sim3Tracker.cpp", "src/DataStructures/Frame.cpp",
                 "src/GlobalMapping/KeyFrameGraph.cpp", "src/Tracking/Relocalizer.cpp"):
        assert keep in text, keep


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout exists only in the build container")
def test_hip_backed_library_builds_from_the_reference_headers():
    """make -C oracle _ref/liblsd_ref_hipbacked.so: the reference's unmodified headers + its Frame / FrameMemory / FramePoseStruct /
    TrackingReference / Sim3Tracker / settings sources compiled where they lie, SE3Tracker.cpp and DepthMap.cpp replaced by
    integration/hip_backed/.  (The dependencies the image lacks are the stand-ins of oracle/ref/shim, as for oracle/_ref.)"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "_ref/liblsd_ref_hipbacked.so"])
    assert os.path.exists(HB)


@pytest.mark.skipif(not os.path.exists(HB), reason="oracle/_ref/liblsd_ref_hipbacked.so is built where /root/reference exists")
def test_hip_backed_library_defines_the_reference_classes_over_the_c_abi():
    out = subprocess.check_output(["nm", "-DC", HB], text=True)
    defined = [l.split(" ", 2)[2] for l in out.splitlines() if " T " in l]
    undefined = [l.split()[-1] for l in out.splitlines() if " U " in l]
    for sym in ("lsd_slam::SE3Tracker::trackFrame(lsd_slam::TrackingReference*, lsd_slam::Frame*, Sophus::SE3Group<double, 0> const&)",
                "lsd_slam::SE3Tracker::trackFrameOnPermaref(lsd_slam::Frame*, lsd_slam::Frame*, Sophus::SE3Group<double, 0>)",
                "lsd_slam::SE3Tracker::checkPermaRefOverlap(lsd_slam::Frame*, Sophus::SE3Group<double, 0>)",
                "lsd_slam::DepthMap::updateKeyframe(std::deque<std::shared_ptr<lsd_slam::Frame>, std::allocator<std::shared_ptr<lsd_slam::Frame> > >)",
                "lsd_slam::DepthMap::createKeyFrame(lsd_slam::Frame*)", "lsd_slam::DepthMap::finalizeKeyFrame()",
                "lsd_slam::DepthMap::initializeFromGTDepth(lsd_slam::Frame*)", "lsd_slam::DepthMap::setFromExistingKF(lsd_slam::Frame*)"):
        assert sym in defined, sym
    for sym in ("lsdhip_tracker_track", "lsdhip_tracker_track_permaref", "lsdhip_depth_update", "lsdhip_depth_create_keyframe", "lsdhip_depth_finalize",
                "lsdhip_frame_create", "lsdhip_frame_set_depth_planes"):
        assert sym in undefined, sym
    # none of the CPU implementation's stages is in there: the only definitions of these classes are the bindings
    for gone in ("SE3Tracker::calcResidualAndBuffers", "SE3Tracker::calcWeightsAndResidual", "SE3Tracker::calculateWarpUpdate", "DepthMap::doLineStereo",
                 "DepthMap::observeDepthRow", "DepthMap::regularizeDepthMap", "DepthMap::propagateDepth"):
        assert gone not in out, gone
    # and the reference's own Frame / TrackingReference / Sim3Tracker are: the rest of lsd_slam_core links against what it always did
    for kept in ("lsd_slam::Frame::setDepth(lsd_slam::DepthMapPixelHypothesis const*)", "lsd_slam::TrackingReference::makePointCloud(int)",
                 "lsd_slam::Sim3Tracker::trackFrameSim3"):
        assert any(kept in d for d in defined), kept


@pytest.mark.skipif(not os.path.exists(HB), reason="oracle/_ref/liblsd_ref_hipbacked.so is built where /root/reference exists")
def test_hip_backed_library_loads_without_a_gpu():
    import ctypes
    L = ctypes.CDLL(HB)
    L.orc_ref_build_info.restype = ctypes.c_char_p
    assert b"integration/hip_backed" in L.orc_ref_build_info()
