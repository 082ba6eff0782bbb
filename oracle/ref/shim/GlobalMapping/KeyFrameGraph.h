// ORACLE — TEST INFRASTRUCTURE ONLY.  STAND-IN for C/GlobalMapping/KeyFrameGraph.h (g2o pose graph — out of scope and
// not buildable here).  DepthMap.cpp / TrackingReference.cpp include it but use nothing from it.
#pragma once
namespace lsd_slam { class KeyFrameGraph; }
