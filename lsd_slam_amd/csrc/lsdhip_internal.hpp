// Internal structures of liblsdhip.so (MI355X / gfx950).  Host-side bookkeeping mirrors what the reference keeps in
// Frame / FramePoseStruct / SE3Tracker / DepthMap members; all per-pixel data lives in HBM as SoA planes.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/lsdhip.h"

#define LSD_LEVELS LSDHIP_PYRAMID_LEVELS
#define LSD_TRACK_MIN_LEVEL 1   // SE3TRACKING_MIN_LEVEL, C/util/settings.h:98
#define LSD_TRACK_MAX_LEVEL 5   // SE3TRACKING_MAX_LEVEL
#define LSD_QUICK_KF_CHECK_LVL 4

void lsd_set_error(const char* fmt, ...);
#define HIPCHK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      lsd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return LSDHIP_E_HIP;                                                                 \
    }                                                                                      \
  } while (0)

// ---- host math (double/float quaternion poses; independent of oracle/) ---------------------------------
namespace lsdm {
struct Quatd { double w, x, y, z; };
struct Quatf { float w, x, y, z; };
struct SE3dH { Quatd q; double t[3]; };
struct SE3fH { Quatf q; float t[3]; };
struct Sim3dH { Quatd q; double t[3]; double s; };

SE3dH se3d_from7(const double p[7]);
void se3d_to7(const SE3dH& T, double p[7]);
SE3dH se3d_inverse(const SE3dH& T);
SE3fH se3f_from_d(const SE3dH& T);
SE3dH se3d_from_f(const SE3fH& T);
SE3fH se3f_inverse(const SE3fH& T);
SE3fH se3f_mul(const SE3fH& a, const SE3fH& b);  // Sophus operator*: fastMultiply + normalize
SE3fH se3f_exp(const float a[6]);                 // Sophus SE3Group<float>::exp
void quatf_to_rot(const Quatf& q, float R[9]);    // row-major
void quatd_to_rot(const Quatd& q, double R[9]);
Sim3dH sim3_inverse(const Sim3dH& S);
void ldlt6_solve(const float A[36], const float b[6], float x[6]);  // pivoted LDL^T like Eigen's A.ldlt().solve(b)
void inverse3_eigen(const float K[9], float Kinv[9]);              // Eigen compute_inverse<3>
}  // namespace lsdm

// ---- device-visible parameter blocks --------------------------------------------------------------------
struct LevelIntr { float fx, fy, cx, cy, fxi, fyi, cxi, cyi; };

// raw sums one residual-kernel evaluation produces (before the tail-drop correction and normalisation)
#define RS_NUM 48
enum {
  RS_M = 0,        // in-image point count (buf_warped_size)
  RS_GOOD, RS_BAD, RS_SUMRES2, RS_SUMSIGNED, RS_SXX, RS_SYY, RS_SX, RS_SY, RS_SW, RS_USAGE,
  RS_WERR,         // sum of wh*w_p*r^2 (calcWeightsAndResidual)
  RS_A0,           // 21 upper-triangular entries of J J^T w
  RS_B0 = RS_A0 + 21,  // 6 entries of J r w
  RS_ERR = RS_B0 + 6,  // sum w r^2 (LGS error)
  RS_NREF,         // number of valid reference points (numData[level])
  RS_END
};
static_assert(RS_END <= RS_NUM, "record too small");

struct ResidualArgs {
  // keyframe (reference) planes at this level
  const float* kf_idepth;
  const float* kf_idepthVar;
  const float* kf_image;
  // tracked frame texels (gx, gy, I, 0) at this level
  const float4* fr_grad;
  uint8_t* wasGood;      // level-1 mask or nullptr
  // explicit point list instead of keyframe planes (permaref path); npts < 0 => dense keyframe grid
  const float* pts_pos;     // 3 floats per point
  const float* pts_colvar;  // 2 floats per point
  int npts;
  int w, h;
  float fx, fy, cx, cy, fxi, fyi, cxi, cyi;
  float R[9];
  float t[3];
  float aff_a, aff_b;
  float cameraPixelNoise2, var_weight, huber_half;
  float* partials;       // [gridDim.x][RS_NUM]
  float* out_record;     // RS_NUM floats, host-visible
  int nblocks;
};

struct lsdhip_ctx {
  int device = 0;
  int w = 0, h = 0;
  int wl[LSD_LEVELS], hl[LSD_LEVELS];
  LevelIntr intr[LSD_LEVELS];
  float K0[9], K0inv[9];
  lsdhip_params params;
  hipStream_t stream = nullptr;
  // profiling of the residual kernel (bench.py roofline leg)
  bool prof_on = false;
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  double prof_ms = 0, prof_bytes = 0;
  long long prof_launches = 0;
};

struct lsdhip_frame {
  lsdhip_ctx* ctx = nullptr;
  int id = 0;
  uint8_t* d_gray = nullptr;            // level-0 source (uint8)
  float* d_image[LSD_LEVELS] = {};      // float planes
  float4* d_grad[LSD_LEVELS] = {};      // (gx, gy, I, 0)
  float* d_absgrad = nullptr;           // level-0 |grad| (temp of buildMaxGradients)
  float* d_maxgrad = nullptr;           // level-0 maxGradients
  float* d_idepth[LSD_LEVELS] = {};
  float* d_idepthVar[LSD_LEVELS] = {};
  bool hasIDepth = false;
  uint8_t* d_wasGood = nullptr;         // level-1 mask (lazily created, 0xFF)
  bool wasGoodValid = false;
  // pose-tree node
  lsdm::Sim3dH thisToParent_raw;
  lsdhip_frame* trackingParent = nullptr;
  int trackingParentID = -1;
  float initialTrackedResidual = 0;
  int numFramesTrackedOnThis = 0, numMappedOnThis = 0, numMappedOnThisTotal = 0;
  float meanIdepth = 1;
  int numPoints = 0;
  bool depthHasBeenUpdatedFlag = false;
  // re-activation data (Frame::takeReActivationData)
  float* d_idepth_reAct = nullptr;
  float* d_idepthVar_reAct = nullptr;
  uint8_t* d_validity_reAct = nullptr;
  bool reActValid = false;
};

struct lsdhip_tracker {
  lsdhip_ctx* ctx = nullptr;
  // DenseDepthTrackerSettings (C/util/settings.h:355-402)
  float lambdaSuccessFac = 0.5f, lambdaFailFac = 2.0f;
  float lambdaInitial[LSD_LEVELS], stepSizeMin[LSD_LEVELS], convergenceEps[LSD_LEVELS];
  int maxItsPerLvl[LSD_LEVELS];
  float lambdaInitialTestTrack = 0, stepSizeMinTestTrack = 1e-3, convergenceEpsTestTrack = 0.98, maxItsTestTrack = 5;
  float huber_d = 3, var_weight = 1.0;
  // results (public members of SE3Tracker)
  float pointUsage = 0, lastGoodCount = 0, lastMeanRes = 0, lastBadCount = 0, lastResidual = 0;
  float affineEstimation_a = 1, affineEstimation_b = 0, affineEstimation_a_lastIt = 1, affineEstimation_b_lastIt = 0;
  bool diverged = false, trackingWasGood = false;
  int numEvaluations = 0, numWarpUpdates = 0;
  // device scratch
  float* d_partials = nullptr;   // [max_blocks][RS_NUM]
  int max_blocks = 0;
  float* h_record = nullptr;     // pinned, device-mapped: final RS_NUM sums
  float* d_record = nullptr;     // device alias of h_record
  float* d_pts = nullptr;        // permaref point upload
  int pts_capacity = 0;
};

// SoA hypothesis planes in HBM (29 B/px + one spare validity plane for snapshot semantics)
struct HypPlanes {
  uint8_t* valid;       // isValid
  int32_t* blacklisted;
  float* nextID;        // nextStereoFrameMinID
  int32_t* validity;    // validity_counter
  float* idepth;
  float* var;
  float* idepth_s;
  float* var_s;
};

struct StereoRef {      // one reference (tracked) frame as K4 sees it
  const float* image;   // level-0 image plane
  const uint8_t* wasGood;  // level-1 mask or nullptr (only consulted when parentIsKF)
  int parentIsKF;
  int id;
  float initialTrackedResidual;
  float K_otherToThis_R[9];
  float K_otherToThis_t[3];
  float otherToThis_t[3];
  float thisToOther_t[3];
  float row0[3], row1[3], row2[3];
};

struct lsdhip_depthmap {
  lsdhip_ctx* ctx = nullptr;
  HypPlanes cur;         // currentDepthMap
  HypPlanes oth;         // otherDepthMap (propagation target; swapped)
  uint8_t* d_validSnap = nullptr;  // spare validity plane (ping-pong target of K5/K6)
  void* bases[3] = {nullptr, nullptr, nullptr};  // arena base pointers (for hipFree)
  lsdhip_frame* activeKeyFrame = nullptr;
  bool activeKeyFrameIsReactivated = false;
  StereoRef* d_refs = nullptr;
  int* d_refByID = nullptr;
  int refs_capacity = 0, byid_capacity = 0;
  // K7 scratch
  int* d_slotCount = nullptr;      // per target
  int* d_slots = nullptr;          // per target x capacity source indices
  float4* d_cand = nullptr;        // per source candidate (new_idepth, new_var, validity as float bits, target)
  int* d_flags = nullptr;          // overflow flag etc.
  double* d_red = nullptr;         // reduction scratch (sum, count)
  double* h_red = nullptr;         // pinned
  float msUpdate = 0, msCreate = 0, msFinalize = 0, msObserve = 0, msRegularize = 0, msPropagate = 0, msFillHoles = 0,
        msSetDepth = 0;
};

// kernels / launchers implemented in the .hip files
int lsd_frame_build_pyramids(lsdhip_frame* f);
int lsd_frame_build_idepth_pyramid(lsdhip_frame* f);
int lsd_frame_ensure_depth_planes(lsdhip_frame* f);
int lsd_frame_ensure_wasgood(lsdhip_frame* f);
