// ORACLE — TEST INFRASTRUCTURE ONLY.  Sim3Tracker restatement (C/Tracking/Sim3Tracker.{h,cpp}); see orc_sim3.cpp.
#pragma once
#include "lsd_oracle.hpp"
#include "orc_sim3_exp.hpp"

namespace orc {

struct Sim3ResidualStruct {   // Sim3Tracker.h:36-57
  float sumResD, sumResP;
  int numTermsD, numTermsP;
  float meanD, meanP, mean;
};
struct LGS7s {
  float A[49], b[7];
  size_t num_constraints;
};
struct Sim3EvalRecord {
  int warped_size;
  float pointUsage, affine_a_lastIt, affine_b_lastIt;
  Sim3ResidualStruct res;
  float A[49], b[7];        // LGS7 after initializeFrom (not divided by num_constraints)
  double num_constraints;
};
void sim3_ldlt7_solve(const float A[49], const float b[7], float x[7]);

class Sim3Tracker {
 public:
  Sim3Tracker(int w, int h, const float K[4], const Params& p);
  ~Sim3Tracker();
  Sim3d trackFrameSim3(TrackingReference* reference, Frame* frame, const Sim3d& frameToReference_initialEstimate, int startLevel,
                       int finalLevel);
  // buffers + weights + LGS once at a fixed transformation (kernel-level parity hook)
  void evaluate(TrackingReference* reference, Frame* frame, const Sim3d& referenceToFrame, int level, float a, float b,
                Sim3EvalRecord* out);

  DenseDepthTrackerSettings settings;
  TrackerMode mode = TRACKER_SSE;
  float lastResidual = 0, lastDepthResidual = 0, lastPhotometricResidual = 0, pointUsage = 0;
  float affineEstimation_a = 1, affineEstimation_b = 0;
  bool diverged = false;
  float lastSim3Hessian[49] = {};
  int numEvaluations = 0;

  float *buf_warped_residual, *buf_warped_dx, *buf_warped_dy, *buf_warped_x, *buf_warped_y, *buf_warped_z, *buf_d, *buf_residual_d,
      *buf_idepthVar, *buf_warped_idepthVar, *buf_weight_p, *buf_weight_d;
  int buf_warped_size = 0;

 private:
  void calcSim3Buffers(TrackingReference* reference, Frame* frame, const Sim3d& referenceToFrame, int level);
  Sim3ResidualStruct calcSim3WeightsAndResidual(const Sim3d& referenceToFrame);
  Sim3ResidualStruct calcSim3WeightsAndResidualSSE(const Sim3d& referenceToFrame, bool exactRcp);
  void calcSim3LGS(LGS7s& ls7);
  void calcSim3LGSSSE(LGS7s& ls7, bool exactRcp);
  Sim3ResidualStruct callWeights(const Sim3d& T);
  void callLGS(LGS7s& ls7);
  int width, height;
  Params params;
  float affineEstimation_a_lastIt = 1, affineEstimation_b_lastIt = 0;
};

}  // namespace orc
