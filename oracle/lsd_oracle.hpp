// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// A dependency-free CPU restatement of the hot path, function by function, each citing the reference
// file:line it follows.  `C/` = /root/reference/lsd_slam_core/src/.
//
// PINNING: tum-vision/lsd_slam ships no tests, golden vectors or fixtures for SE3Tracker / DepthMap, and
// lsd_slam_core as a whole cannot be built here (ROS rosbuild, Eigen, boost, OpenCV, g2o absent).  The
// hot-path translation units themselves CAN be compiled, unchanged and where they lie, against stand-in
// headers for those dependencies: oracle/_ref (recipe: oracle/Makefile target `ref`, stand-ins under
// oracle/ref/shim).  tests/test_ref_pin_cpu.py runs this restatement against that build — pyramids, point
// clouds, K1/K2/K3 buffers and normal equations (SSE and scalar paths), whole trackFrame calls, every
// depth-map stage, doLineStereo pixel by pixel with its status codes, updateKeyframe / createKeyFrame /
// finalizeKeyFrame, a 30-frame sequence fed forward, and Sim3Tracker (buffers / weights / LGS7 at fixed transformations and
// whole trackFrameSim3 calls) — all BIT-EXACT.  What is NOT the reference's own
// code in that build, and therefore still rests on restatement: the Eigen fixed-size arithmetic
// (3x3 products / inverse, 6x6 LDL^T) and the Sophus SE3 / Sim3 group operations, which oracle/_ref
// shares with this oracle (orc_math.hpp); SE3 exp / log and Sim3::exp pass Sophus' own expMapTest / expLogTest on the vectors of the
// reference's thirdparty/Sophus/sophus/test_se3.cpp / test_sim3.cpp at Sophus' thresholds, and the 6x6 LDL^T agrees with LAPACK
// (tests/test_oracle_cpu.py; Sophus itself is vendored there but needs Eigen, which this image lacks); the 7x7 LDL^T and
// Quaternionf::setFromTwoVectors are shared stand-in algebra in the same sense.
//
// Conventions: unwritten pool memory is defined as 0 (the reference recycles FrameMemory buffers,
// C/DataStructures/FrameMemory.cpp:80-86); float op order is the *written* order of the reference and
// the parity build uses -ffp-contract=off.
#pragma once
#include <cstdint>
#include <deque>
#include <memory>
#include <vector>
#include "orc_math.hpp"

namespace orc {

// ---- C/util/settings.h:50-174 (compile-time constants that are part of the contract) -------------
#define ORC_PYRAMID_LEVELS 5
#define ORC_SE3TRACKING_MIN_LEVEL 1
#define ORC_SE3TRACKING_MAX_LEVEL 5
#define ORC_QUICK_KF_CHECK_LVL 4
#define ORC_MAPPING_THREADS 4

// C/util/settings.cpp:77-88 — mutable globals of the reference, gathered in one parameter block.
struct Params {
  float minUseGrad = 5;
  float cameraPixelNoise2 = 4 * 4;
  float depthSmoothingFactor = 1;
  bool allowNegativeIdepths = true;
  bool useSubpixelStereo = true;
  bool multiThreading = true;
  bool useAffineLightningEstimation = true;
  float KFDistWeight = 4;
  float KFUsageWeight = 3;
};

// C/util/settings.h:355-402
struct DenseDepthTrackerSettings {
  float lambdaSuccessFac, lambdaFailFac;
  float lambdaInitial[ORC_PYRAMID_LEVELS], stepSizeMin[ORC_PYRAMID_LEVELS], convergenceEps[ORC_PYRAMID_LEVELS];
  int maxItsPerLvl[ORC_PYRAMID_LEVELS];
  float lambdaInitialTestTrack, stepSizeMinTestTrack, convergenceEpsTestTrack, maxItsTestTrack;
  float huber_d, var_weight;
  DenseDepthTrackerSettings();
};

// C/DepthEstimation/DepthMapPixelHypothesis.h:37-94 (32-byte AoS layout)
struct DepthMapPixelHypothesis {
  bool isValid;
  int blacklisted;
  float nextStereoFrameMinID;
  int validity_counter;
  float idepth, idepth_var, idepth_smoothed, idepth_var_smoothed;
  DepthMapPixelHypothesis() : isValid(false), blacklisted(0) {}
  DepthMapPixelHypothesis(float id, float id_s, float var, float var_s, int val)
      : isValid(true), blacklisted(0), nextStereoFrameMinID(0), validity_counter(val), idepth(id), idepth_var(var),
        idepth_smoothed(id_s), idepth_var_smoothed(var_s) {}
  DepthMapPixelHypothesis(float id, float var, int val)
      : isValid(true), blacklisted(0), nextStereoFrameMinID(0), validity_counter(val), idepth(id), idepth_var(var),
        idepth_smoothed(-1), idepth_var_smoothed(-1) {}
};
static_assert(sizeof(DepthMapPixelHypothesis) == 32, "hypothesis must be 32 bytes like the reference");

// C/DataStructures/Frame.{h,cpp} — only what the hot path touches.
class Frame {
 public:
  Frame(int id, int width, int height, const float K[4] /*fx,fy,cx,cy*/, const unsigned char* image);
  int id() const { return id_; }
  int width(int l = 0) const { return w_[l]; }
  int height(int l = 0) const { return h_[l]; }

  const float* image(int l = 0);
  const float* gradients(int l = 0);     // Vector4f plane (gx, gy, I, 0), 4 floats per pixel
  const float* maxGradients(int l = 0);
  void overrideMaxGradients(const float* plane);   // test hook, level 0
  const float* idepth(int l = 0);
  const float* idepthVar(int l = 0);
  bool hasIDepthBeenSet() const { return hasIDepth_; }

  void setDepth(const DepthMapPixelHypothesis* map);               // Frame.cpp:199-243
  void setDepthFromGroundTruth(const float* depth, float cov_scale, float minUseGrad);  // Frame.cpp:245-293
  void setDepthPlanes(const float* idepth, const float* idepthVar);  // test helper: raw level-0 planes
  void prepareForStereoWith(Frame* other, const Sim3d& thisToOther, const M3f& K, int level);  // Frame.cpp:295-317
  void takeReActivationData(const DepthMapPixelHypothesis* map);   // Frame.cpp:107-145

  uint8_t* refPixelWasGood();           // Frame.h:421-437 (level-1 mask, 0xFF on creation)
  uint8_t* refPixelWasGoodNoCreate() { return wasGood_.empty() ? nullptr : wasGood_.data(); }
  void clear_refPixelWasGood() { wasGood_.clear(); }

  // intrinsics per level (Frame.cpp:397-459)
  float fx[ORC_PYRAMID_LEVELS], fy[ORC_PYRAMID_LEVELS], cx[ORC_PYRAMID_LEVELS], cy[ORC_PYRAMID_LEVELS];
  float fxInv[ORC_PYRAMID_LEVELS], fyInv[ORC_PYRAMID_LEVELS], cxInv[ORC_PYRAMID_LEVELS], cyInv[ORC_PYRAMID_LEVELS];
  M3f K[ORC_PYRAMID_LEVELS], KInv[ORC_PYRAMID_LEVELS];

  // pose tree node (C/DataStructures/FramePoseStruct.h) reduced to what the path writes/reads
  Sim3d thisToParent_raw;
  Frame* trackingParent = nullptr;
  float initialTrackedResidual = 0;
  int numFramesTrackedOnThis = 0, numMappedOnThis = 0, numMappedOnThisTotal = 0;
  float meanIdepth = 1;
  int numPoints = 0;
  float numMappablePixels = -1;
  bool depthHasBeenUpdatedFlag = false;

  // stereo pre-computes (Frame.h:184-192)
  M3f K_otherToThis_R;
  V3f K_otherToThis_t, otherToThis_t, K_thisToOther_t, thisToOther_t;
  M3f thisToOther_R;
  V3f otherToThis_R_row0, otherToThis_R_row1, otherToThis_R_row2;
  float distSquared = 0;
  int referenceID = -1;

  // re-activation data (Frame.cpp:107-145)
  std::vector<float> idepth_reAct, idepthVar_reAct;
  std::vector<unsigned char> validity_reAct;

  bool sseImagePyramid = true;  // association of buildImage (SSE branch Frame.cpp:516-553 vs scalar :614-627)

 private:
  void buildImage(int l);
  void buildGradients(int l);
  void buildMaxGradients(int l);
  void buildIDepthAndIDepthVar(int l);
  void releaseIDepthPyramid();

  int id_;
  int w_[ORC_PYRAMID_LEVELS], h_[ORC_PYRAMID_LEVELS];
  std::vector<float> image_[ORC_PYRAMID_LEVELS], grad_[ORC_PYRAMID_LEVELS], maxGrad_[ORC_PYRAMID_LEVELS];
  std::vector<float> idepth_[ORC_PYRAMID_LEVELS], idepthVar_[ORC_PYRAMID_LEVELS];
  bool imageValid_[ORC_PYRAMID_LEVELS], gradValid_[ORC_PYRAMID_LEVELS], maxGradValid_[ORC_PYRAMID_LEVELS],
      idepthValid_[ORC_PYRAMID_LEVELS];
  bool hasIDepth_ = false;
  std::vector<uint8_t> wasGood_;
  float minUseGradForMappable_ = 5;
};

// C/Tracking/TrackingReference.{h,cpp}
class TrackingReference {
 public:
  void importFrame(Frame* kf);                 // TrackingReference.cpp:71-87
  void makePointCloud(int level);              // TrackingReference.cpp:96-147
  Frame* keyframe = nullptr;
  std::vector<float> posData[ORC_PYRAMID_LEVELS];          // 3 floats per point
  std::vector<float> gradData[ORC_PYRAMID_LEVELS];         // 2 floats per point
  std::vector<float> colorAndVarData[ORC_PYRAMID_LEVELS];  // 2 floats per point
  std::vector<int> pointPosInXYGrid[ORC_PYRAMID_LEVELS];
  int numData[ORC_PYRAMID_LEVELS] = {0, 0, 0, 0, 0};
};

// C/Tracking/LGSX.h:184-402
struct LGS6 {
  float A[36];
  float b[6];
  float error;
  size_t num_constraints;
  alignas(16) float SSEData[4 * 28];
  void initialize();
  void update(const float J[6], float res, float weight);
  void finishNoDivide();
  void finish();
};

enum TrackerMode {
  TRACKER_SCALAR = 0,        // the non-SIMD member functions (SE3Tracker.cpp:749-790, :1258-1299)
  TRACKER_SSE = 1,           // ENABLE_SSE path with real _mm_rcp_ps (SE3Tracker.cpp:492-575, :1033-1130)
  TRACKER_SSE_EXACT_RCP = 2  // SSE op order, _mm_rcp_ps replaced by an IEEE 1/x (what a GPU can match)
};

// per-evaluation record exposed for kernel-level tests
struct ResidualRecord {
  int warped_size;       // buf_warped_size
  float goodCount, badCount, pointUsage, meanRes, retval;  // retval = sumResUnweighted/goodCount
  float affine_a_lastIt, affine_b_lastIt;
  float weightedError;   // calcWeightsAndResidual return
  float A[36], b[6], lsError;
  double num_constraints;
};

// C/Tracking/SE3Tracker.{h,cpp}
class SE3Tracker {
 public:
  SE3Tracker(int w, int h, const float K[4], const Params& p);
  SE3d trackFrame(TrackingReference* reference, Frame* frame, const SE3d& frameToReference_initialEstimate);
  SE3d trackFrameOnPermaref(const float* permaRef_pos, const float* permaRef_colVar, int permaRefNumPts, Frame* frame,
                            const SE3d& referenceToFrame);
  float checkPermaRefOverlap(const float* permaRef_pos, int permaRefNumPts, Frame* reference,
                             const SE3d& referenceToFrame);
  // one fused evaluation (K1 + K2 + K3) at a fixed pose for kernel-level parity tests
  void evaluate(TrackingReference* reference, Frame* frame, const SE3f& referenceToFrame, int level, float aff_a,
                float aff_b, ResidualRecord* out);

  DenseDepthTrackerSettings settings;
  TrackerMode mode = TRACKER_SSE;
  float pointUsage = 0, lastGoodCount = 0, lastMeanRes = 0, lastBadCount = 0, lastResidual = 0;
  float affineEstimation_a = 1, affineEstimation_b = 0;
  bool diverged = false, trackingWasGood = false;
  int numEvaluations = 0, numWarpUpdates = 0;  // instrumentation (not in the reference)
  long long levelEvaluations[ORC_PYRAMID_LEVELS] = {}, levelPoints[ORC_PYRAMID_LEVELS] = {}, levelWarped[ORC_PYRAMID_LEVELS] = {};   // per pyramid level, cumulative

  // SoA scratch (SE3Tracker.cpp:67-78)
  float *buf_warped_residual, *buf_warped_dx, *buf_warped_dy, *buf_warped_x, *buf_warped_y, *buf_warped_z, *buf_d,
      *buf_idepthVar, *buf_weight_p;
  int buf_warped_size = 0;
  ~SE3Tracker();

 private:
  float calcResidualAndBuffers(const float* refPoint, const float* refColVar, const int* idxBuf, int refNum,
                               Frame* frame, const SE3f& referenceToFrame, int level);
  float calcWeightsAndResidual(const SE3f& referenceToFrame);
  float calcWeightsAndResidualSSE(const SE3f& referenceToFrame, bool exactRcp);
  void calculateWarpUpdate(LGS6& ls);
  void calculateWarpUpdateSSE(LGS6& ls, bool exactRcp);
  float callWeights(const SE3f& T);
  void callWarpUpdate(LGS6& ls);

  int width, height;
  Params params;
  float affineEstimation_a_lastIt = 1, affineEstimation_b_lastIt = 0;
};

// C/DepthEstimation/DepthMap.{h,cpp}
class DepthMap {
 public:
  DepthMap(int w, int h, const float K[4], const Params& p);
  ~DepthMap();
  void reset();
  void updateKeyframe(std::deque<std::shared_ptr<Frame>> referenceFrames);  // DepthMap.cpp:1072-1213
  void createKeyFrame(Frame* new_keyframe);                                 // DepthMap.cpp:1222-1327
  void finalizeKeyFrame();                                                  // DepthMap.cpp:1363-1395
  void invalidate() { activeKeyFrame = nullptr; }
  bool isValid() const { return activeKeyFrame != nullptr; }
  void initializeFromGTDepth(Frame* new_frame);                             // DepthMap.cpp:965-1018
  void initializeRandomly(Frame* new_frame);                                // DepthMap.cpp:883-916
  void setFromExistingKF(Frame* kf);                                        // DepthMap.cpp:920-962

  // individual stages, public for kernel-level parity tests
  void observeDepth();
  void regularizeDepthMapFillHoles();
  void regularizeDepthMap(bool removeOcclusions, int validityTH);
  void propagateDepth(Frame* new_keyframe);
  void setReferenceFrames(std::deque<std::shared_ptr<Frame>>& referenceFrames);  // the preamble of updateKeyframe
  void refreshActiveKeyFrameImage() { if (activeKeyFrame) activeKeyFrameImageData = activeKeyFrame->image(0); }

  DepthMapPixelHypothesis* currentDepthMap;
  DepthMapPixelHypothesis* otherDepthMap;
  Frame* activeKeyFrame = nullptr;
  bool activeKeyFrameIsReactivated = false;
  float lastRescaleFactor = 1;   // rescaleFactor of the last createKeyFrame (DepthMap.cpp:1294)
  Params params;
  int numThreads = ORC_MAPPING_THREADS;
  struct Pool;                       // persistent worker pool (IndexThreadReduce)
  std::shared_ptr<Pool> pool;

 private:
  void observeDepthRow(int yMin, int yMax);
  bool observeDepthCreate(int x, int y, int idx);
  bool observeDepthUpdate(int x, int y, int idx, const float* keyFrameMaxGradBuf);
 public:   // per-pixel parity hook (orc_depth_line_stereo)
  bool makeAndCheckEPL(int x, int y, const Frame* ref, float* pepx, float* pepy);
  float doLineStereo(float u, float v, float epxn, float epyn, float min_idepth, float prior_idepth, float max_idepth,
                     const Frame* referenceFrame, const float* referenceFrameImage, float& result_idepth,
                     float& result_var, float& result_eplLength);
 private:
  void regularizeDepthMapFillHolesRow(int yMin, int yMax);
  void buildRegIntegralBuffer();
  void buildRegIntegralBufferRow1(int yMin, int yMax);
  template <bool removeOcclusions> void regularizeDepthMapRow(int validityTH, int yMin, int yMax);
  template <typename F> void reduce(F f, int first, int end, int stepSize);  // IndexThreadReduce.h:68-123

  int width, height;
  M3f K, KInv;
  float fx, fy, cx, cy, fxi, fyi, cxi, cyi;
  const float* activeKeyFrameImageData = nullptr;
  Frame* oldest_referenceFrame = nullptr;
  Frame* newest_referenceFrame = nullptr;
  std::vector<Frame*> referenceFrameByID;
  int referenceFrameByID_offset = 0;
  int* validityIntegralBuffer;
};

float getInterpolatedElement(const float* mat, float x, float y, int width);  // C/util/globalFuncs.h:43-61

}  // namespace orc
