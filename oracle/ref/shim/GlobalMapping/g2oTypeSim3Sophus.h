// ORACLE — TEST INFRASTRUCTURE ONLY.  STAND-IN for C/GlobalMapping/g2oTypeSim3Sophus.h (needs g2o).  FramePoseStruct.h
// only stores a VertexSim3* that the hot path never dereferences.
#pragma once
#include "util/SophusUtil.h"
namespace lsd_slam { class VertexSim3; }
