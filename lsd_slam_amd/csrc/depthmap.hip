// DepthMap on the device: semi-dense inverse-depth estimation kernels K4 (observe / doLineStereo / EKF update),
// K5 (fill holes), K6 (regularise), K7 (propagate to a new keyframe), K8 (setDepth + rescale).  gfx950 only.
//
// Reference behaviour restated (C/ = lsd_slam_core/src/):
//   DepthMap::observeDepthRow / observeDepthCreate / observeDepthUpdate   C/DepthEstimation/DepthMap.cpp:111-146, :237-473
//   DepthMap::makeAndCheckEPL                                             :184-234
//   DepthMap::doLineStereo                                                :1442-1972
//   DepthMap::regularizeDepthMapFillHolesRow + buildRegIntegralBuffer     :656-754
//   DepthMap::regularizeDepthMapRow<removeOcclusions>                     :758-848
//   DepthMap::propagateDepth                                              :475-653
//   DepthMap::updateKeyframe / createKeyFrame / finalizeKeyFrame          :1072-1213, :1222-1327, :1363-1395
//   DepthMap::initializeFromGTDepth / initializeRandomly / setFromExistingKF  :883-1018
//   Frame::setDepth / takeReActivationData / prepareForStereoWith         C/DataStructures/Frame.cpp:199-243, :107-145, :295-317
//
// HBM layout: the 32-byte AoS hypothesis of the reference becomes eight SoA planes (29 B/px).  The reference's
// "memcpy current -> other, read other, write current" snapshot (64 B/px of copy traffic per pass) is replaced by a
// ping-pong of the 1-byte validity plane only: K5 and K6 read every other field in place because neither of them
// modifies a field that a neighbouring pixel reads.  The validity integral image of K5 is replaced by a direct 5x5
// window sum (int32, exact — the integral is only ever used as that window difference).
#include <chrono>
#include <cstdlib>
#include "lsdhip_internal.hpp"

// ---- constants, C/util/settings.h:34-35, :50-174 ------------------------------------------------------------------
#define DIVISION_EPS 1e-10f
#define VALIDITY_COUNTER_MAX (5.0f)
#define VALIDITY_COUNTER_MAX_VARIABLE (250.0f)
#define VALIDITY_COUNTER_INC 5
#define VALIDITY_COUNTER_DEC 5
#define VALIDITY_COUNTER_INITIAL_OBSERVE 5
#define VAL_SUM_MIN_FOR_CREATE (30)
#define VAL_SUM_MIN_FOR_KEEP (24)
#define VAL_SUM_MIN_FOR_UNBLACKLIST (100)
#define MIN_BLACKLIST -1
#define SUCC_VAR_INC_FAC (1.01f)
#define FAIL_VAR_INC_FAC 1.1f
#define MAX_VAR (0.5f * 0.5f)
#define VAR_GT_INIT_INITIAL 0.01f * 0.01f
#define VAR_RANDOM_INIT_INITIAL (0.5f * MAX_VAR)
#define MIN_DEPTH 0.05f
#define MAX_EPL_LENGTH_CROP 30.0f
#define MIN_EPL_LENGTH_CROP (3.0f)
#define GRADIENT_SAMPLE_DIST 1.0f
#define SAMPLE_POINT_TO_BORDER 7
#define MAX_ERROR_STEREO (1300.0f)
#define MIN_DISTANCE_ERROR_STEREO (1.5f)
#define STEREO_EPL_VAR_FAC 2.0f
#define DIFF_FAC_SMOOTHING (1.0f * 1.0f)
#define DIFF_FAC_OBSERVE (1.0f * 1.0f)
#define DIFF_FAC_PROP_MERGE (1.0f * 1.0f)
#define MIN_EPL_GRAD_SQUARED (2.0f * 2.0f)
#define MIN_EPL_LENGTH_SQUARED (1.0f * 1.0f)
#define MIN_EPL_ANGLE_SQUARED (0.3f * 0.3f)
#define MAX_DIFF_CONSTANT (40.0f * 40.0f)
#define MAX_DIFF_GRAD_MULT (0.5f * 0.5f)
#define PROP_SLOT_CAP 8

#if defined(LSD_PHASE_TRACE) && defined(__HIP_DEVICE_COMPILE__)
#define OBS_MARK(k) do { if (a.trace && threadIdx.x == 0) a.trace[(blockIdx.x + blockIdx.y * gridDim.x) * 16 + (k)] = clock64(); } while (0)
#define OBS_VAL(k, v) do { if (a.trace && threadIdx.x == 0) a.trace[(blockIdx.x + blockIdx.y * gridDim.x) * 16 + (k)] = (unsigned long long)(v); } while (0)
#else
#define OBS_MARK(k) do { } while (0)
#define OBS_VAL(k, v) do { } while (0)
#endif
// per-pixel arithmetic is host+device so that tests/ can shadow-execute it on a CPU when chasing a parity bit
#define LSD_HD __host__ __device__ __forceinline__
#define LSD_HD_NOINLINE __host__ __device__

// UNZERO (settings.h:35) evaluates in double in the reference; the float form below returns the same float for every
// float input (|val| is compared against 1e-10, the result is rounded to float either way).
LSD_HD float unzero(float val) {
  return (float)(val < 0 ? (val > -1e-10 ? -1e-10 : (double)val) : (val < 1e-10 ? 1e-10 : (double)val));
}

// getInterpolatedElement, C/util/globalFuncs.h:43-61
LSD_HD float interp1(const float* __restrict__ mat, float x, float y, int width) {
  int ix = (int)x;
  int iy = (int)y;
  float dx = x - ix;
  float dy = y - iy;
  float dxdy = dx * dy;
  const float* bp = mat + ix + iy * width;
  return dxdy * bp[1 + width] + (dy - dxdy) * bp[width] + (dx - dxdy) * bp[1] + (1 - dx - dy + dxdy) * bp[0];
}

struct ObserveArgs {
  HypPlanes m;
  LSD_G const float* kfImage;
  LSD_G const float4* kfGrad;
  LSD_G const float* kfMaxGrad;
  LSD_G const StereoRef* refs;
  LSD_G const int* refByID;
  int nByID, byIDOffset, nRefs;
  int reactivated;
  int w, h;
  float fx, fy, cx, cy, fxi, fyi, cxi, cyi;
  float minUseGrad, cameraPixelNoise2;
  int allowNegativeIdepths, useSubpixelStereo;
  int kfNumFramesTrackedOnThis, kfNumMappedOnThis;
#ifdef LSD_PHASE_TRACE
  LSD_G unsigned long long* trace;   // developer build: per-stage timestamps of one traced pixel per workgroup
#endif
  LSD_G unsigned long long* counters;   // null, or (sampled launches while profiling) two words per wave: pixels that entered the search, walk steps
  LSD_G float4* queue;                  // batches, two-launch form: this map's search queue (pixel index bits, epx, epy, -) and its fill count
  LSD_G int* qcount;
  LSD_G const uint16_t* gradCand;       // the keyframe's gradient candidates (lsdhip_frame::d_gradCand; k_observe_select_cand_batch)
  StereoRef one;   // nRefs == 1 (the blockUntilMapped case): the reference travels in the kernel arguments, no staging copy
};

// DepthMap::doLineStereo (DepthMap.cpp:1442-1972) in three parts.  Status codes as in the reference: >= 0 matching error,
// -1 out of bounds, -2 ambiguous / negative, -3 error too large, -4 arithmetic.
//   stereo_setup  (:1449-1611): epipolar segment in the reference image, the five keyframe samples; 0 or a status code;
//   stereo_walk_* (:1622-1744): the walk along the segment -> best / second-best match and their neighbours;
//   stereo_finish (:1746-1972): ambiguity tests, sub-pixel refinement, inverse depth and variance.
// The walk is capped at 1000 steps (the reference loop is unbounded; no finite input reaches the cap).
struct StereoSetup {
  float KinvP0, KinvP1;              // KinvP2 == 1
  float rescaleFactor;
  float realVal_m2, realVal_m1, realVal, realVal_p1, realVal_p2;
  float pFar0, pFar1, pClose0, pClose1, incx, incy, eplLength;
  float gI0, gI1;                    // keyframe gradient at (u, v) (getInterpolatedElement42) for the geometric error term
};
struct WalkResult {
  float best_match_err, second_best_match_err, best_match_errPre, best_match_errPost, best_match_DiffErrPre, best_match_DiffErrPost;
  float best_match_x, best_match_y;
  int loopCBest, loopCSecond, loopCounter;
};

LSD_HD float stereo_setup(const ObserveArgs& a, const StereoRef& rf, const float u, const float v, const float epxn, const float epyn,
                          const float min_idepth, const float prior_idepth, float max_idepth, StereoSetup& S) {
  const int width = a.w, height = a.h;
  const float* __restrict__ kfImg = a.kfImage;
  // the keyframe gradient taps of the geometric-error term (used after the search) depend on (u, v) only: issue their loads
  // now so that they do not add a memory round trip after the walk
  const float4* const gbp = a.kfGrad + (int)u + (int)v * width;
  const float4 gt00 = gbp[0], gt10 = gbp[1], gt01 = gbp[width], gt11 = gbp[1 + width];
  float KinvP0 = a.fxi * u + a.cxi, KinvP1 = a.fyi * v + a.cyi, KinvP2 = 1.0f;
  S.KinvP0 = KinvP0; S.KinvP1 = KinvP1;
  float pInf0 = (rf.K_otherToThis_R[0] * KinvP0 + rf.K_otherToThis_R[1] * KinvP1) + rf.K_otherToThis_R[2] * KinvP2;
  float pInf1 = (rf.K_otherToThis_R[3] * KinvP0 + rf.K_otherToThis_R[4] * KinvP1) + rf.K_otherToThis_R[5] * KinvP2;
  float pInf2 = (rf.K_otherToThis_R[6] * KinvP0 + rf.K_otherToThis_R[7] * KinvP1) + rf.K_otherToThis_R[8] * KinvP2;
  float pReal2 = pInf2 / prior_idepth + rf.K_otherToThis_t[2];
  float rescaleFactor = pReal2 * prior_idepth;
  S.rescaleFactor = rescaleFactor;

  float firstX = u - 2 * epxn * rescaleFactor;
  float firstY = v - 2 * epyn * rescaleFactor;
  float lastX = u + 2 * epxn * rescaleFactor;
  float lastY = v + 2 * epyn * rescaleFactor;
  if (firstX <= 0 || firstX >= width - 2 || firstY <= 0 || firstY >= height - 2 || lastX <= 0 || lastX >= width - 2 ||
      lastY <= 0 || lastY >= height - 2)
    return -1;
  if (!(rescaleFactor > 0.7f && rescaleFactor < 1.4f)) return -1;

  OBS_MARK(2);
  S.realVal_p1 = interp1(kfImg, u + epxn * rescaleFactor, v + epyn * rescaleFactor, width);
  S.realVal_m1 = interp1(kfImg, u - epxn * rescaleFactor, v - epyn * rescaleFactor, width);
  S.realVal = interp1(kfImg, u, v, width);
  S.realVal_m2 = interp1(kfImg, u - 2 * epxn * rescaleFactor, v - 2 * epyn * rescaleFactor, width);
  S.realVal_p2 = interp1(kfImg, u + 2 * epxn * rescaleFactor, v + 2 * epyn * rescaleFactor, width);

  const float Kt0 = rf.K_otherToThis_t[0], Kt1 = rf.K_otherToThis_t[1], Kt2 = rf.K_otherToThis_t[2];
  float pClose0 = pInf0 + Kt0 * max_idepth, pClose1 = pInf1 + Kt1 * max_idepth, pClose2 = pInf2 + Kt2 * max_idepth;
  if (pClose2 < 0.001f) {
    max_idepth = (0.001f - pInf2) / Kt2;
    pClose0 = pInf0 + Kt0 * max_idepth; pClose1 = pInf1 + Kt1 * max_idepth; pClose2 = pInf2 + Kt2 * max_idepth;
  }
  pClose0 = pClose0 / pClose2; pClose1 = pClose1 / pClose2;

  float pFar0 = pInf0 + Kt0 * min_idepth, pFar1 = pInf1 + Kt1 * min_idepth, pFar2 = pInf2 + Kt2 * min_idepth;
  if (pFar2 < 0.001f || max_idepth < min_idepth) return -1;
  pFar0 = pFar0 / pFar2; pFar1 = pFar1 / pFar2;

  if (__builtin_isnan((float)(pFar0 + pClose0))) return -4;

  float incx = pClose0 - pFar0;
  float incy = pClose1 - pFar1;
  float eplLength = sqrtf(incx * incx + incy * incy);
  if (eplLength == 0.0f || __builtin_isinf(eplLength)) return -4;  // `!eplLength > 0 || isinf` of the reference (:1518)

  if (eplLength > MAX_EPL_LENGTH_CROP) {
    pClose0 = pFar0 + incx * MAX_EPL_LENGTH_CROP / eplLength;
    pClose1 = pFar1 + incy * MAX_EPL_LENGTH_CROP / eplLength;
  }
  incx *= GRADIENT_SAMPLE_DIST / eplLength;
  incy *= GRADIENT_SAMPLE_DIST / eplLength;

  pFar0 -= incx; pFar1 -= incy;
  pClose0 += incx; pClose1 += incy;

  if (eplLength < MIN_EPL_LENGTH_CROP) {
    float pad = (MIN_EPL_LENGTH_CROP - (eplLength)) / 2.0f;
    pFar0 -= incx * pad; pFar1 -= incy * pad;
    pClose0 += incx * pad; pClose1 += incy * pad;
  }

  if (pFar0 <= SAMPLE_POINT_TO_BORDER || pFar0 >= width - SAMPLE_POINT_TO_BORDER || pFar1 <= SAMPLE_POINT_TO_BORDER ||
      pFar1 >= height - SAMPLE_POINT_TO_BORDER)
    return -1;

  if (pClose0 <= SAMPLE_POINT_TO_BORDER || pClose0 >= width - SAMPLE_POINT_TO_BORDER || pClose1 <= SAMPLE_POINT_TO_BORDER ||
      pClose1 >= height - SAMPLE_POINT_TO_BORDER) {
    if (pClose0 <= SAMPLE_POINT_TO_BORDER) {
      float toAdd = (SAMPLE_POINT_TO_BORDER - pClose0) / incx;
      pClose0 += toAdd * incx; pClose1 += toAdd * incy;
    } else if (pClose0 >= width - SAMPLE_POINT_TO_BORDER) {
      float toAdd = (width - SAMPLE_POINT_TO_BORDER - pClose0) / incx;
      pClose0 += toAdd * incx; pClose1 += toAdd * incy;
    }
    if (pClose1 <= SAMPLE_POINT_TO_BORDER) {
      float toAdd = (SAMPLE_POINT_TO_BORDER - pClose1) / incy;
      pClose0 += toAdd * incx; pClose1 += toAdd * incy;
    } else if (pClose1 >= height - SAMPLE_POINT_TO_BORDER) {
      float toAdd = (height - SAMPLE_POINT_TO_BORDER - pClose1) / incy;
      pClose0 += toAdd * incx; pClose1 += toAdd * incy;
    }
    float fincx = pClose0 - pFar0;
    float fincy = pClose1 - pFar1;
    float newEplLength = sqrtf(fincx * fincx + fincy * fincy);
    if (pClose0 <= SAMPLE_POINT_TO_BORDER || pClose0 >= width - SAMPLE_POINT_TO_BORDER || pClose1 <= SAMPLE_POINT_TO_BORDER ||
        pClose1 >= height - SAMPLE_POINT_TO_BORDER || newEplLength < 8.0f)
      return -1;
  }
  S.pFar0 = pFar0; S.pFar1 = pFar1; S.pClose0 = pClose0; S.pClose1 = pClose1; S.incx = incx; S.incy = incy; S.eplLength = eplLength;
  {
    // getInterpolatedElement42 on the keyframe gradients (globalFuncs.h:95-109): interpolated here, where the taps have long
    // arrived, so that two values instead of eight stay live across the walk
    const int ix = (int)u, iy = (int)v;
    const float dx = u - ix, dy = v - iy;
    const float dxdy = dx * dy;
    S.gI0 = dxdy * gt11.x + (dy - dxdy) * gt01.x + (dx - dxdy) * gt10.x + (1 - dx - dy + dxdy) * gt00.x;
    S.gI1 = dxdy * gt11.y + (dy - dxdy) * gt01.y + (dx - dxdy) * gt10.y + (1 - dx - dy + dxdy) * gt00.y;
  }
  OBS_MARK(3);
  return 0;
}

// the walk, one lane per search (the reference's loop; two steps of samples kept in flight)
LSD_HD void stereo_walk_serial(const float* __restrict__ refImg, const int width, const StereoSetup& S, WalkResult& W) {
  const float incx = S.incx, incy = S.incy, pClose0 = S.pClose0, pClose1 = S.pClose1;
  const float realVal_p2 = S.realVal_p2, realVal_p1 = S.realVal_p1, realVal = S.realVal, realVal_m1 = S.realVal_m1, realVal_m2 = S.realVal_m2;
  float cpx = S.pFar0;
  float cpy = S.pFar1;
  float val_cp_m2 = interp1(refImg, cpx - 2.0f * incx, cpy - 2.0f * incy, width);
  float val_cp_m1 = interp1(refImg, cpx - incx, cpy - incy, width);
  float val_cp = interp1(refImg, cpx, cpy, width);
  float val_cp_p1 = interp1(refImg, cpx + incx, cpy + incy, width);
  float val_cp_p2;


  int loopCounter = 0;
  float best_match_x = -1;
  float best_match_y = -1;
  float best_match_err = INFINITY;          // `1e50` assigned to a float (:1658)
  float second_best_match_err = INFINITY;
  float best_match_errPre = NAN, best_match_errPost = NAN, best_match_DiffErrPre = NAN, best_match_DiffErrPost = NAN;
  bool bestWasLastLoop = false;
  float eeLast = -1;
  float e1A = NAN, e1B = NAN, e2A = NAN, e2B = NAN, e3A = NAN, e3B = NAN, e4A = NAN, e4B = NAN, e5A = NAN, e5B = NAN;
  int loopCBest = -1, loopCSecond = -1;
  // The sample positions of the walk do not depend on the sampled values, so the loads of the next two steps are kept in
  // flight while a step is evaluated (same positions, formed by the same additions as the loop's own cpx += incx; they stay
  // inside the image: at most 4 samples beyond pClose, which is SAMPLE_POINT_TO_BORDER = 7 pixels away from the border).
  float w1x = cpx + incx, w1y = cpy + incy;
  float w2x = w1x + incx, w2y = w1y + incy;
  float p2_0 = interp1(refImg, cpx + 2 * incx, cpy + 2 * incy, width);
  float p2_1 = interp1(refImg, w1x + 2 * incx, w1y + 2 * incy, width);
  while ((((incx < 0) == (cpx > pClose0) && (incy < 0) == (cpy > pClose1)) || loopCounter == 0) && loopCounter < 1000) {
    const float p2_2 = interp1(refImg, w2x + 2 * incx, w2y + 2 * incy, width);
    val_cp_p2 = p2_0;
    p2_0 = p2_1;
    p2_1 = p2_2;
    w2x += incx;
    w2y += incy;
    float ee = 0;
    if (loopCounter % 2 == 0) {
      e1A = val_cp_p2 - realVal_p2; ee += e1A * e1A;
      e2A = val_cp_p1 - realVal_p1; ee += e2A * e2A;
      e3A = val_cp - realVal;       ee += e3A * e3A;
      e4A = val_cp_m1 - realVal_m1; ee += e4A * e4A;
      e5A = val_cp_m2 - realVal_m2; ee += e5A * e5A;
    } else {
      e1B = val_cp_p2 - realVal_p2; ee += e1B * e1B;
      e2B = val_cp_p1 - realVal_p1; ee += e2B * e2B;
      e3B = val_cp - realVal;       ee += e3B * e3B;
      e4B = val_cp_m1 - realVal_m1; ee += e4B * e4B;
      e5B = val_cp_m2 - realVal_m2; ee += e5B * e5B;
    }
    if (ee < best_match_err) {
      second_best_match_err = best_match_err;
      loopCSecond = loopCBest;
      best_match_err = ee;
      loopCBest = loopCounter;
      best_match_errPre = eeLast;
      best_match_DiffErrPre = e1A * e1B + e2A * e2B + e3A * e3B + e4A * e4B + e5A * e5B;
      best_match_errPost = -1;
      best_match_DiffErrPost = -1;
      best_match_x = cpx;
      best_match_y = cpy;
      bestWasLastLoop = true;
    } else {
      if (bestWasLastLoop) {
        best_match_errPost = ee;
        best_match_DiffErrPost = e1A * e1B + e2A * e2B + e3A * e3B + e4A * e4B + e5A * e5B;
        bestWasLastLoop = false;
      }
      if (ee < second_best_match_err) {
        second_best_match_err = ee;
        loopCSecond = loopCounter;
      }
    }
    eeLast = ee;
    val_cp_m2 = val_cp_m1; val_cp_m1 = val_cp; val_cp = val_cp_p1; val_cp_p1 = val_cp_p2;
    cpx += incx;
    cpy += incy;
    loopCounter++;
  }


  W.best_match_err = best_match_err; W.second_best_match_err = second_best_match_err;
  W.best_match_errPre = best_match_errPre; W.best_match_errPost = best_match_errPost;
  W.best_match_DiffErrPre = best_match_DiffErrPre; W.best_match_DiffErrPost = best_match_DiffErrPost;
  W.best_match_x = best_match_x; W.best_match_y = best_match_y;
  W.loopCBest = loopCBest; W.loopCSecond = loopCSecond; W.loopCounter = loopCounter;
}

LSD_HD float stereo_finish(const ObserveArgs& a, const StereoRef& rf, const float u, const float v, const float epxn, const float epyn,
                           const StereoSetup& S, const WalkResult& W, float& result_idepth, float& result_var, float& result_eplLength) {
  float best_match_err = W.best_match_err, best_match_x = W.best_match_x, best_match_y = W.best_match_y;
  const float second_best_match_err = W.second_best_match_err;
  const float best_match_errPre = W.best_match_errPre, best_match_errPost = W.best_match_errPost;
  const float best_match_DiffErrPre = W.best_match_DiffErrPre, best_match_DiffErrPost = W.best_match_DiffErrPost;
  const int loopCBest = W.loopCBest, loopCSecond = W.loopCSecond;
  const float incx = S.incx, incy = S.incy, rescaleFactor = S.rescaleFactor, eplLength = S.eplLength;
  const float realVal_p2 = S.realVal_p2, realVal_p1 = S.realVal_p1, realVal = S.realVal, realVal_m1 = S.realVal_m1, realVal_m2 = S.realVal_m2;
  const float KinvP0 = S.KinvP0, KinvP1 = S.KinvP1, KinvP2 = 1.0f;
  if (best_match_err > 4.0f * (float)MAX_ERROR_STEREO) return -3;
  if (abs(loopCBest - loopCSecond) > 1.0f && MIN_DISTANCE_ERROR_STEREO * best_match_err > second_best_match_err) return -2;
  bool didSubpixel = false;
  if (a.useSubpixelStereo) {
    float gradPre_pre = -(best_match_errPre - best_match_DiffErrPre);
    float gradPre_this = +(best_match_err - best_match_DiffErrPre);
    float gradPost_this = -(best_match_err - best_match_DiffErrPost);
    float gradPost_post = +(best_match_errPost - best_match_DiffErrPost);
    bool interpPost = false;
    bool interpPre = false;
    if ((gradPost_this < 0) ^ (gradPre_this < 0)) {
    } else if ((gradPre_pre < 0) ^ (gradPre_this < 0)) {
      if ((gradPost_post < 0) ^ (gradPost_this < 0)) {
      } else
        interpPre = true;
    } else if ((gradPost_post < 0) ^ (gradPost_this < 0)) {
      interpPost = true;
    }
    if (interpPre) {
      float d = gradPre_this / (gradPre_this - gradPre_pre);
      best_match_x -= d * incx;
      best_match_y -= d * incy;
      best_match_err = best_match_err - 2 * d * gradPre_this - (gradPre_pre - gradPre_this) * d * d;
      didSubpixel = true;
    } else if (interpPost) {
      float d = gradPost_this / (gradPost_this - gradPost_post);
      best_match_x += d * incx;
      best_match_y += d * incy;
      best_match_err = best_match_err + 2 * d * gradPost_this + (gradPost_post - gradPost_this) * d * d;
      didSubpixel = true;
    }
  }

  float sampleDist = GRADIENT_SAMPLE_DIST * rescaleFactor;
  float gradAlongLine = 0;
  float tmp = realVal_p2 - realVal_p1; gradAlongLine += tmp * tmp;
  tmp = realVal_p1 - realVal;          gradAlongLine += tmp * tmp;
  tmp = realVal - realVal_m1;          gradAlongLine += tmp * tmp;
  tmp = realVal_m1 - realVal_m2;       gradAlongLine += tmp * tmp;
  gradAlongLine /= sampleDist * sampleDist;

  if (best_match_err > (float)MAX_ERROR_STEREO + sqrtf(gradAlongLine) * 20) return -3;

  float idnew_best_match;
  float alpha;
  if (incx * incx > incy * incy) {
    float oldX = a.fxi * best_match_x + a.cxi;
    float nominator = (oldX * rf.otherToThis_t[2] - rf.otherToThis_t[0]);
    // Vector3f::dot — Eigen redux order x0 + (x1 + x2)
    float dot0 = KinvP0 * rf.row0[0] + (KinvP1 * rf.row0[1] + KinvP2 * rf.row0[2]);
    float dot2 = KinvP0 * rf.row2[0] + (KinvP1 * rf.row2[1] + KinvP2 * rf.row2[2]);
    idnew_best_match = (dot0 - oldX * dot2) / nominator;
    alpha = incx * a.fxi * (dot0 * rf.otherToThis_t[2] - dot2 * rf.otherToThis_t[0]) / (nominator * nominator);
  } else {
    float oldY = a.fyi * best_match_y + a.cyi;
    float nominator = (oldY * rf.otherToThis_t[2] - rf.otherToThis_t[1]);
    float dot1 = KinvP0 * rf.row1[0] + (KinvP1 * rf.row1[1] + KinvP2 * rf.row1[2]);
    float dot2 = KinvP0 * rf.row2[0] + (KinvP1 * rf.row2[1] + KinvP2 * rf.row2[2]);
    idnew_best_match = (dot1 - oldY * dot2) / nominator;
    alpha = incy * a.fyi * (dot1 * rf.otherToThis_t[2] - dot2 * rf.otherToThis_t[1]) / (nominator * nominator);
  }

  if (idnew_best_match < 0) {
    if (!a.allowNegativeIdepths) return -2;
  }

  float photoDispError = 4.0f * a.cameraPixelNoise2 / (gradAlongLine + DIVISION_EPS);
  float trackingErrorFac = 0.25f * (1.0f + rf.initialTrackedResidual);
  const float gI0 = S.gI0, gI1 = S.gI1;
  float geoDispError = (gI0 * epxn + gI1 * epyn) + DIVISION_EPS;
  geoDispError = trackingErrorFac * trackingErrorFac * (gI0 * gI0 + gI1 * gI1) / (geoDispError * geoDispError);
  OBS_MARK(6);
  result_var = alpha * alpha * ((didSubpixel ? 0.05f : 0.5f) * sampleDist * sampleDist + geoDispError + photoDispError);
  result_idepth = idnew_best_match;
  result_eplLength = eplLength;
  return best_match_err;
}


LSD_HD_NOINLINE float do_line_stereo(const ObserveArgs& a, const StereoRef& rf, const float u, const float v, const float epxn,
                                const float epyn, const float min_idepth, const float prior_idepth, float max_idepth,
                                float& result_idepth, float& result_var, float& result_eplLength, int& steps) {
  StereoSetup S;
  steps = 0;
  const float st = stereo_setup(a, rf, u, v, epxn, epyn, min_idepth, prior_idepth, max_idepth, S);
  if (st < 0) return st;
  WalkResult W;
  stereo_walk_serial(rf.image, a.w, S, W);
  steps = W.loopCounter;
  return stereo_finish(a, rf, u, v, epxn, epyn, S, W, result_idepth, result_var, result_eplLength);
}

// DepthMap::makeAndCheckEPL (DepthMap.cpp:184-234)
// (the four keyframe-image taps of the gradient travel as arguments: the caller loads them together with the other
// per-pixel inputs, one memory round trip instead of one per check)
LSD_HD bool make_and_check_epl(const ObserveArgs& a, const StereoRef& rf, int x, int y, float kR, float kL, float kD, float kU, float* pepx,
                               float* pepy) {
  float epx = -a.fx * rf.thisToOther_t[0] + rf.thisToOther_t[2] * (x - a.cx);
  float epy = -a.fy * rf.thisToOther_t[1] + rf.thisToOther_t[2] * (y - a.cy);
  if (__builtin_isnan(epx + epy)) return false;
  float eplLengthSquared = epx * epx + epy * epy;
  if (eplLengthSquared < MIN_EPL_LENGTH_SQUARED) return false;
  float gx = kR - kL;
  float gy = kD - kU;
  float eplGradSquared = gx * epx + gy * epy;
  eplGradSquared = eplGradSquared * eplGradSquared / eplLengthSquared;
  if (eplGradSquared < MIN_EPL_GRAD_SQUARED) return false;
  if (eplGradSquared / (gx * gx + gy * gy) < MIN_EPL_ANGLE_SQUARED) return false;
  // the reference calls the C library's double sqrt() here (unqualified sqrt in DepthMap.cpp:229), so the quotient is
  // formed in double and rounded to float once; f64 sqrt / divide are IEEE-exact on gfx950
  float fac = (float)((double)GRADIENT_SAMPLE_DIST / sqrt((double)eplLengthSquared));
  *pepx = epx * fac;
  *pepy = epy * fac;
  return true;
}

// observeDepthRow body for one pixel (DepthMap.cpp:117-143), in two halves:
//   observe_front: everything up to and including makeAndCheckEPL — the cheap rejections every pixel goes through; returns
//                  true for the ~1 % of pixels that go on to the epipolar search, with the reference frame and EPL direction;
//   observe_back : doLineStereo and observeDepthCreate / observeDepthUpdate's bookkeeping for such a pixel.
// k_observe runs both in the same lane.  (A select kernel queueing the survivors + a walk kernel over the dense queue, and a form that
// loads all samples of short walks up front, were built and measured in round 2: 37-52 us and 28 us against 21 us — removed.)
template <bool ONE>
LSD_HD bool observe_front(const ObserveArgs& a, const int x, const int y, int& refIdxOut, float& epxOut, float& epyOut) {
  if (x < 3 || x >= a.w - 3 || y < 3 || y >= a.h - 3) return false;
  const int idx = x + y * a.w;
  // The per-pixel inputs are loaded in two batches, each issued before any of its values is tested, so that the checks
  // below cost two memory round trips instead of one each (every plane was last written by a kernel on another XCD):
  // batch 1 = what the cheap rejections need (13 B per pixel, every pixel), batch 2 = everything else of this pixel.
  const bool hasHypothesis = a.m.valid[idx] != 0;
  const float maxGrad = a.kfMaxGrad[idx];
  const int blacklisted0 = a.m.blacklisted[idx];
  const float nextID = a.m.nextID[idx];
  if (hasHypothesis && maxGrad < a.minUseGrad) { a.m.valid[idx] = 0; return false; }
  if (maxGrad < a.minUseGrad || blacklisted0 < MIN_BLACKLIST) return false;

  int refIdx = 0;
  if (!hasHypothesis) {
    refIdx = a.reactivated ? a.nRefs - 1 : 0;
  } else if (!a.reactivated) {
    int rel = (int)nextID - a.byIDOffset;
    if (rel >= a.nByID) return false;
    refIdx = (ONE || rel < 0) ? 0 : a.refByID[rel];
  } else {
    refIdx = a.nRefs - 1;
  }
  const StereoRef& rf = ONE ? a.one : a.refs[refIdx];
  const bool useMask = rf.parentIsKF && rf.wasGood != nullptr;
  const uint8_t wasGoodHere = useMask ? rf.wasGood[(x >> 1) + (a.w >> 1) * (y >> 1)] : (uint8_t)1;
  const float kR = a.kfImage[idx + 1], kL = a.kfImage[idx - 1], kD = a.kfImage[idx + a.w], kU = a.kfImage[idx - a.w];
  if (!wasGoodHere) return false;       // observeDepthCreate :246-252 / observeDepthUpdate :323-329
  if (!make_and_check_epl(a, rf, x, y, kR, kL, kD, kU, &epxOut, &epyOut)) return false;
  refIdxOut = refIdx;
  return true;
}

struct ObsPixel {        // what the back half needs of a pixel's hypothesis
  bool hasHypothesis;
  float maxGrad;
  int blacklisted0, validity0;
  float idepth_smoothed, var_smoothed, var0, idepth0;
  float min_idepth, prior_idepth, max_idepth;    // search interval handed to doLineStereo
};
LSD_HD void observe_back_load(const ObserveArgs& a, const int idx, ObsPixel& P) {
  P.hasHypothesis = a.m.valid[idx] != 0;
  P.maxGrad = a.kfMaxGrad[idx];
  P.blacklisted0 = a.m.blacklisted[idx];
  P.idepth_smoothed = a.m.idepth_s[idx];
  P.var_smoothed = a.m.var_s[idx];
  P.var0 = a.m.var[idx];
  P.idepth0 = a.m.idepth[idx];
  P.validity0 = a.m.validity[idx];
  if (!P.hasHypothesis) {          // observeDepthCreate (DepthMap.cpp:269-270)
    P.min_idepth = 0.0f; P.prior_idepth = 1.0f; P.max_idepth = 1.0f / MIN_DEPTH;
  } else {                         // observeDepthUpdate (:334-341)
    float sv = sqrtf(P.var_smoothed);
    float min_idepth = P.idepth_smoothed - sv * STEREO_EPL_VAR_FAC;
    float max_idepth = P.idepth_smoothed + sv * STEREO_EPL_VAR_FAC;
    if (min_idepth < 0) min_idepth = 0;
    if (max_idepth > 1 / MIN_DEPTH) max_idepth = 1 / MIN_DEPTH;
    P.min_idepth = min_idepth; P.prior_idepth = P.idepth_smoothed; P.max_idepth = max_idepth;
  }
}
// what observeDepthCreate (:272-291) / observeDepthUpdate (:346-471) do with doLineStereo's verdict
LSD_HD void observe_back_store(const ObserveArgs& a, const StereoRef& rf, const int idx, const ObsPixel& P, const float error,
                               float result_idepth, const float result_var, const float result_eplLength) {
  const int blacklisted0 = P.blacklisted0, validity0 = P.validity0;
  const float var0 = P.var0, idepth0 = P.idepth0, idepth_smoothed = P.idepth_smoothed, var_smoothed = P.var_smoothed, maxGrad = P.maxGrad;
  if (!P.hasHypothesis) {
    if (error == -3 || error == -2) a.m.blacklisted[idx] = blacklisted0 - 1;
    if (error < 0 || result_var > MAX_VAR) return;
    result_idepth = unzero(result_idepth);
    a.m.valid[idx] = 1;
    a.m.blacklisted[idx] = 0;
    a.m.nextID[idx] = 0;
    a.m.validity[idx] = VALIDITY_COUNTER_INITIAL_OBSERVE;
    a.m.idepth[idx] = result_idepth;
    a.m.var[idx] = result_var;
    a.m.idepth_s[idx] = -1;
    a.m.var_s[idx] = -1;
    return;
  }
  float diff = result_idepth - idepth_smoothed;
  if (error == -1) return;
  if (error == -2) {
    int vc = validity0 - VALIDITY_COUNTER_DEC;
    if (vc < 0) vc = 0;
    a.m.validity[idx] = vc;
    a.m.nextID[idx] = 0;
    float nv = var0 * FAIL_VAR_INC_FAC;
    a.m.var[idx] = nv;
    if (nv > MAX_VAR) { a.m.valid[idx] = 0; a.m.blacklisted[idx] = blacklisted0 - 1; }
    return;
  }
  if (error == -3 || error == -4) return;
  if (DIFF_FAC_OBSERVE * diff * diff > result_var + var_smoothed) {
    float nv = var0 * FAIL_VAR_INC_FAC;
    a.m.var[idx] = nv;
    if (nv > MAX_VAR) a.m.valid[idx] = 0;
    return;
  }
  {
    const float old_var = var0;
    float id_var = old_var * SUCC_VAR_INC_FAC;
    float w = result_var / (result_var + id_var);
    float new_idepth = (1 - w) * result_idepth + w * idepth0;
    a.m.idepth[idx] = unzero(new_idepth);
    id_var = id_var * w;
    if (id_var < old_var) a.m.var[idx] = id_var;
    int vc = validity0 + VALIDITY_COUNTER_INC;
    float absGrad = maxGrad;
    if (vc > VALIDITY_COUNTER_MAX + absGrad * (VALIDITY_COUNTER_MAX_VARIABLE) / 255.0f)
      vc = VALIDITY_COUNTER_MAX + absGrad * (VALIDITY_COUNTER_MAX_VARIABLE) / 255.0f;
    a.m.validity[idx] = vc;
    if (result_eplLength < MIN_EPL_LENGTH_CROP) {
      float inc = a.kfNumFramesTrackedOnThis / (float)(a.kfNumMappedOnThis + 5);
      if (inc < 3) inc = 3;
      inc += ((int)(result_eplLength * 10000) % 2);
      if (result_eplLength < 0.5 * MIN_EPL_LENGTH_CROP) inc *= 3;
      a.m.nextID[idx] = rf.id + inc;
    }
  }
}
template <bool ONE>
LSD_HD_NOINLINE int observe_back(const ObserveArgs& a, const int idx, const int refIdx, const float epx, const float epy) {
  const int x = idx % a.w, y = idx / a.w;
  const StereoRef& rf = ONE ? a.one : a.refs[refIdx];
  ObsPixel P;
  observe_back_load(a, idx, P);
  float result_idepth = 0, result_var = 0, result_eplLength = 0;   // uninitialised in the reference (:344)
  int steps;
  const float error = do_line_stereo(a, rf, x, y, epx, epy, P.min_idepth, P.prior_idepth, P.max_idepth, result_idepth, result_var, result_eplLength, steps);
  observe_back_store(a, rf, idx, P, error, result_idepth, result_var, result_eplLength);
  return steps;
}

// returns -1 for a pixel the cheap tests reject, else the number of walk steps of its epipolar search
template <bool ONE>
LSD_HD int observe_pixel(const ObserveArgs& a, const int x, const int y) {
  int refIdx = 0;
  float epx = 0, epy = 0;
  if (observe_front<ONE>(a, x, y, refIdx, epx, epy)) return observe_back<ONE>(a, x + y * a.w, refIdx, epx, epy);
  return -1;
}
// measurement (sampled launches): the wave's number of searches and walk steps -> its slot of a.counters
__device__ __forceinline__ void observe_count(const ObserveArgs& a, const int steps) {
  if (a.counters == nullptr) return;
  const unsigned long long act = __popcll(__ballot(steps >= 0));
  int s = steps > 0 ? steps : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) {
    const size_t wv = ((size_t)blockIdx.x + (size_t)blockIdx.y * gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    a.counters[2 * wv] = act;
    a.counters[2 * wv + 1] = (unsigned long long)s;
  }
}
__global__ __launch_bounds__(256) void k_obs_count_sum(const unsigned long long* __restrict__ counters, int nwaves, unsigned long long* __restrict__ acc) {
  __shared__ unsigned long long s_a[256], s_b[256];
  unsigned long long a0 = 0, b0 = 0;
  for (int i = threadIdx.x; i < nwaves; i += 256) { a0 += counters[2 * i]; b0 += counters[2 * i + 1]; }
  s_a[threadIdx.x] = a0; s_b[threadIdx.x] = b0;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) { s_a[threadIdx.x] += s_a[threadIdx.x + off]; s_b[threadIdx.x] += s_b[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { acc[0] += s_a[0]; acc[1] += s_b[0]; acc[2] += 1; }
}

// K4: observeDepthRow over the whole map, one pixel per lane.
// K4: observeDepthRow over the whole map, one pixel per lane.  (Compacting the ~20 % of pixels that reach the epipolar
// search into dense waves, and batching the loads of four walk steps, were both measured and gave nothing: the kernel
// is bound by chains of dependent first-touch loads, not by issue slots — profiles/r01_notes.md.)
// ROWS = rows of the 32-pixel-wide tile a workgroup owns (workgroup size 32 * ROWS).  The kernel has no barriers and no LDS,
// so the workgroup is only a scheduling unit: small workgroups free their wave slots as soon as their own lanes are done.
template <bool ONE, int ROWS>
__global__ __launch_bounds__(32 * ROWS) void k_observe(ObserveArgs a) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31);
  const int y = blockIdx.y * ROWS + (threadIdx.x >> 5);
  OBS_MARK(0);
  const int steps = observe_pixel<ONE>(a, x, y);
  observe_count(a, steps);
  OBS_MARK(7);
}

// the same over the maps of several sequences (blockIdx.z = map; lsdhip_depth_update_batch): one reference frame per map, its
// arguments read from a device array (uniform per workgroup: scalar loads)
template <int ROWS>
__global__ __launch_bounds__(32 * ROWS) void k_observe_batch(const ObserveArgs* __restrict__ items) {
  const ObserveArgs& a = items[blockIdx.z];
  const int x = blockIdx.x * 32 + (threadIdx.x & 31);
  const int y = blockIdx.y * ROWS + (threadIdx.x >> 5);
  (void)observe_pixel<true>(a, x, y);
}

// ---- K7: propagateDepth (DepthMap.cpp:475-653) ---------------------------------------------------------------------
struct PropArgs {
  HypPlanes src;   // currentDepthMap (old keyframe)
  HypPlanes dst;   // otherDepthMap (new keyframe)
  LSD_G const float* oldKFImage;
  LSD_G const float* newKFImage;
  LSD_G const float* newKFMaxGrad;
  LSD_G const uint8_t* trackingWasGood;  // level-1 mask of the new keyframe or nullptr
  LSD_G float4* cand;
  LSD_G int* slotCount;
  LSD_G int* slots;
  LSD_G int* flags;      // [0] error (chain storage exhausted), [1] chain entries handed out
  LSD_G int* ovfHead;    // per target: newest chain entry or -1
  LSD_G int2* ovf;       // (source index, next entry)
  int ovfCap;
  int w, h;
  float fx, fy, cx, cy, fxi, fyi, cxi, cyi;
  float R[9], t[3];
  float minUseGrad;
};
// phase A for one valid source hypothesis (its smoothed inverse depth, variance and validity counter): target pixel and candidate,
// registered in the target's slot list.  The slot lists, chain heads and flags are clean at rest: k_prop_resolve leaves slotCount = 0 /
// ovfHead = -1 behind, the kernel that reads the flags zeroes them.
__device__ __forceinline__ void prop_candidate(const PropArgs& a, const int x, const int y, const int idx, const float ids, const float srcVar,
                                               const int srcValidity) {
  float r0 = x * a.fxi + a.cxi, r1 = y * a.fyi + a.cyi, r2 = 1.0f;
  float Rr0 = (a.R[0] * r0 + a.R[1] * r1) + a.R[2] * r2;
  float Rr1 = (a.R[3] * r0 + a.R[4] * r1) + a.R[5] * r2;
  float Rr2 = (a.R[6] * r0 + a.R[7] * r1) + a.R[8] * r2;
  float pn0 = Rr0 / ids + a.t[0], pn1 = Rr1 / ids + a.t[1], pn2 = Rr2 / ids + a.t[2];
  float new_idepth = lsd_rcp_exact(pn2);
  float u_new = pn0 * new_idepth * a.fx + a.cx;
  float v_new = pn1 * new_idepth * a.fy + a.cy;
  if (!(u_new > 2.1f && v_new > 2.1f && u_new < a.w - 3.1f && v_new < a.h - 3.1f)) return;
  int newIDX = (int)(u_new + 0.5f) + ((int)(v_new + 0.5f)) * a.w;
  float destAbsGrad = a.newKFMaxGrad[newIDX];
  if (a.trackingWasGood != nullptr) {
    if (!a.trackingWasGood[(x >> 1) + (a.w >> 1) * (y >> 1)] || destAbsGrad < a.minUseGrad) return;
  } else {
    float sourceColor = a.oldKFImage[idx];
    float destColor = interp1(a.newKFImage, u_new, v_new, a.w);
    float residual = destColor - sourceColor;
    if (residual * residual / (MAX_DIFF_CONSTANT + MAX_DIFF_GRAD_MULT * destAbsGrad * destAbsGrad) > 1.0f || destAbsGrad < a.minUseGrad) return;
  }
  float idepth_ratio_4 = new_idepth / ids;
  idepth_ratio_4 *= idepth_ratio_4;
  idepth_ratio_4 *= idepth_ratio_4;
  float new_var = idepth_ratio_4 * srcVar;
  a.cand[idx] = make_float4(new_idepth, new_var, __int_as_float(srcValidity), 0.f);
  int pos = atomicAdd(&a.slotCount[newIDX], 1);
  if (pos < PROP_SLOT_CAP) a.slots[(size_t)newIDX * PROP_SLOT_CAP + pos] = idx;
  else {
    // more sources than slots for this target (zoom-out / backward motion): the rest go on a per-target chain
    const int o = atomicAdd(&a.flags[1], 1);
    if (o < a.ovfCap) a.ovf[o] = make_int2(idx, atomicExch(&a.ovfHead[newIDX], o));
    else atomicExch(&a.flags[0], 1);
  }
}

// The same for many maps in TWO launches (lsdhip_depth_update_batch from LSD_OBS_SPLIT_MIN_MAPS maps on).  In k_observe_batch a wave
// that holds a single searching pixel walks the whole dependent chain of doLineStereo at 126 registers per lane with one lane busy, and
// three waves in four hold one (2-7 % of the pixels search).  With enough maps in a launch to fill the chip that is what bounds it, not
// the planes it reads.  So: k_observe_select_batch runs observe_front — the cheap rejections, a streaming pass at few registers — over
// every pixel (four per lane, their loads in flight together) and appends the survivors of a 64x16 tile, compacted, to their map's queue
// (one atomic per tile); k_observe_walk_batch
// runs observe_back over the queues with every lane busy: chunks of 64 consecutive entries (neighbouring tiles: shared lines), dealt to a
// fixed number of one-wave workgroups.  A pixel's update reads and writes nothing but its own hypothesis and constant planes, so the
// order of the queue does not matter: every plane is bit-identical to the one-launch form (tests/test_multiseq_gpu.py).
// (For ONE map the split was measured in round 2 and lost — two latency-bound launches instead of one; it stays one launch there.)
#define LSD_OBS_WALK_MAX_MAPS 256
// observe_front<true> for the four pixels (x, y0 + 4 k) of a lane, staged: the 13 bytes every pixel pays for the cheap rejections of all
// four are requested before the first is looked at, likewise the second batch (mask byte, four keyframe-image taps) of the survivors —
// the pass is a chain of two dependent memory round trips per pixel, and a lane that keeps eight of them in flight needs a quarter of
// the residency rounds (the one-pixel form ran at 1.7 TB/s: 0.84 of its wave cycles waiting, profiles/r05_notes.md).
// Same tests in the same order as observe_front (incl. its one store: a hypothesis on a pixel below the gradient threshold is dropped).
// CAND: the four pixels come from the keyframe's gradient candidates (k_observe_select_cand_batch): the border and gradient tests are
// already answered (and no hypothesis sits below the threshold: lsdhip_depthmap::lowGradHypPossible), maxGradients is not read.
template <bool CAND>
__device__ __forceinline__ void observe_front4(const ObserveArgs& a, const int (&xs)[4], const int (&ys)[4], const bool (&in)[4], bool (&hit)[4],
                                               float (&epx)[4], float (&epy)[4]) {
  const StereoRef& rf = a.one;
  const bool useMask = rf.parentIsKF && rf.wasGood != nullptr;
  bool hyp[4];
  int idx[4], bl[4];
  float mg[4], nid[4];
  uint8_t val[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    idx[k] = in[k] ? xs[k] + ys[k] * a.w : 0;
    if (CAND) {
      // (a quarter of the slots of a group hold a candidate on a typical keyframe: only those request anything)
      val[k] = 0; bl[k] = 0; nid[k] = 0.f; mg[k] = a.minUseGrad;
      if (in[k]) { val[k] = a.m.valid[idx[k]]; bl[k] = a.m.blacklisted[idx[k]]; nid[k] = a.m.nextID[idx[k]]; }
    } else {
      val[k] = a.m.valid[idx[k]];
      mg[k] = a.kfMaxGrad[idx[k]];
      bl[k] = a.m.blacklisted[idx[k]];
      nid[k] = a.m.nextID[idx[k]];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    hyp[k] = val[k] != 0;
    bool pass = in[k];
    if (!CAND && pass && hyp[k] && mg[k] < a.minUseGrad) { a.m.valid[idx[k]] = 0; pass = false; }
    if (pass && ((!CAND && mg[k] < a.minUseGrad) || bl[k] < MIN_BLACKLIST)) pass = false;
    if (pass && hyp[k] && !a.reactivated) {
      const int rel = (int)nid[k] - a.byIDOffset;
      if (rel >= a.nByID) pass = false;
    }
    hit[k] = pass;
  }
  uint8_t good[4];
  float kR[4], kL[4], kD[4], kU[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int x = xs[k], y = ys[k];
    const int j = hit[k] ? idx[k] : a.w + 1;                          // (a pixel with all four neighbours inside the plane)
    if (CAND) {
      good[k] = 1; kR[k] = kL[k] = kD[k] = kU[k] = 0.f;
      if (hit[k]) {
        if (useMask) good[k] = rf.wasGood[(x >> 1) + (a.w >> 1) * (y >> 1)];
        kR[k] = a.kfImage[j + 1]; kL[k] = a.kfImage[j - 1]; kD[k] = a.kfImage[j + a.w]; kU[k] = a.kfImage[j - a.w];
      }
    } else {
      good[k] = (useMask && hit[k]) ? rf.wasGood[(x >> 1) + (a.w >> 1) * (y >> 1)] : (uint8_t)1;
      kR[k] = a.kfImage[j + 1]; kL[k] = a.kfImage[j - 1]; kD[k] = a.kfImage[j + a.w]; kU[k] = a.kfImage[j - a.w];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    epx[k] = 0; epy[k] = 0;
    if (hit[k]) {
      if (!good[k]) hit[k] = false;
      else hit[k] = make_and_check_epl(a, rf, xs[k], ys[k], kR[k], kL[k], kD[k], kU[k], &epx[k], &epy[k]);
    }
  }
}
// the survivors of a workgroup's pixels, compacted, go to their map's queue (one atomic per workgroup and call)
__device__ __forceinline__ void observe_queue_append(const ObserveArgs& a, const int (&xs)[4], const int (&ys)[4], const bool (&hit)[4],
                                                     const float (&epx)[4], const float (&epy)[4], int* s_wcount, int* s_base) {
  const int tid = threadIdx.x, wave = tid >> 6;
  int rank[4], wc = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const unsigned long long bal = __ballot(hit[k]);
    rank[k] = wc + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
    wc += __popcll(bal);
  }
  if ((tid & 63) == 0) s_wcount[wave] = wc;
  __syncthreads();
  if (tid == 0) {
    const int tot = (s_wcount[0] + s_wcount[1]) + (s_wcount[2] + s_wcount[3]);
    *s_base = tot > 0 ? atomicAdd(a.qcount, tot) : 0;
  }
  __syncthreads();
  int off = *s_base;
#pragma unroll
  for (int k = 0; k < 4; k++) off += k < wave ? s_wcount[k] : 0;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (hit[k]) a.queue[off + rank[k]] = make_float4(__int_as_float(xs[k] + ys[k] * a.w), epx[k], epy[k], 0.f);
}
__global__ __launch_bounds__(256) void k_observe_select_batch(const ObserveArgs* __restrict__ items) {
  const ObserveArgs& a = items[blockIdx.z];
  __shared__ int s_wcount[4];
  __shared__ int s_base;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int x = blockIdx.x * 64 + lane;
  const int y0 = blockIdx.y * 16 + wave;                              // the lane's pixels: rows y0, y0 + 4, y0 + 8, y0 + 12 of the 64x16 tile
  int xs[4], ys[4];
  bool in[4], hit[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    xs[k] = x; ys[k] = y0 + 4 * k;
    in[k] = !(x < 3 || x >= a.w - 3 || ys[k] < 3 || ys[k] >= a.h - 3);
  }
  float epx[4], epy[4];
  observe_front4<false>(a, xs, ys, in, hit, epx, epy);
  observe_queue_append(a, xs, ys, hit, epx, epy, s_wcount, &s_base);
}
// The select pass over the keyframe's gradient candidates (round 6): the two tests that depend on the keyframe alone — the 3-pixel border
// and maxGradients >= minUseGrad, which three pixels in four fail on a typical keyframe — were answered when the frame became a keyframe
// (k_grad_candidates, frame.hip).  One wave per group of 1024 consecutive pixels walks the group's list, 256 candidates per trip, four per
// lane with their loads in flight together (the entries of a trip are requested before the count is known: beyond it they are ignored):
// a quarter of the waves of the every-pixel pass, each with one more (short) round trip, and only candidates request hypothesis planes.
// Used when no map of the call can hold a hypothesis below the threshold (lsdhip_depthmap::lowGradHypPossible); same tests in the same
// order on the same pixels otherwise, and the queue's order is immaterial: every plane bit-identical (tests/test_multiseq_gpu.py).
__global__ __launch_bounds__(256) void k_observe_select_cand_batch(const ObserveArgs* __restrict__ items) {
  const ObserveArgs& a = items[blockIdx.z];
  __shared__ int s_wcount[4];
  __shared__ int s_base;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n0 = a.w * a.h, ngroups = (n0 + 1023) >> 10;
  const int g = blockIdx.x * 4 + wave;
  const bool live = g < ngroups;
  LSD_G const uint16_t* list = a.gradCand + (size_t)(live ? g : 0) * 1024;
  const int count = live ? (int)a.gradCand[(size_t)ngroups * 1024 + g] : 0;
  int maxCount = count;                                                // (the workgroup's waves take the same number of trips: the appends hold barriers)
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int gq = blockIdx.x * 4 + q;
    const int cq = gq < ngroups ? (int)a.gradCand[(size_t)ngroups * 1024 + gq] : 0;
    maxCount = cq > maxCount ? cq : maxCount;
  }
  const float inv_w = 1.0f / (float)a.w;
  for (int e0 = 0; e0 < maxCount; e0 += 256) {
    int xs[4], ys[4];
    bool in[4], hit[4];
    unsigned off[4];
#pragma unroll
    for (int k = 0; k < 4; k++) off[k] = list[e0 + lane + 64 * k];      // (within the group's 1024 slots)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      in[k] = e0 + lane + 64 * k < count;
      const int i = (g << 10) + (int)off[k];
      int y = (int)((float)i * inv_w);
      int x = i - y * a.w;
      if (x < 0) { y--; x += a.w; }
      if (x >= a.w) { y++; x -= a.w; }
      xs[k] = in[k] ? x : 3; ys[k] = in[k] ? y : 3;
    }
    float epx[4], epy[4];
    observe_front4<true>(a, xs, ys, in, hit, epx, epy);
    observe_queue_append(a, xs, ys, hit, epx, epy, s_wcount, &s_base);
    __syncthreads();                                                   // (s_wcount / s_base are reused by the next trip)
  }
}
// (144 registers, three waves per SIMD; capped at 128 — four waves, 32 bytes of scratch per lane — the 32-sequence loop ran the same:
// profiles/r05_notes.md)
__global__ __launch_bounds__(64) void k_observe_walk_batch(const ObserveArgs* __restrict__ items, const int* __restrict__ counts, int n,
                                                           unsigned long long* __restrict__ acc) {
  __shared__ int s_incl[LSD_OBS_WALK_MAX_MAPS];     // chunks of 64 entries of maps 0 .. m (inclusive)
  __shared__ int s_cnt[LSD_OBS_WALK_MAX_MAPS];
  const int lane = threadIdx.x;
  int TC = 0, total = 0;
#pragma unroll
  for (int c = 0; c < LSD_OBS_WALK_MAX_MAPS / 64; c++) {
    const int m = c * 64 + lane;
    const int cnt = m < n ? counts[m < n ? m : 0] : 0;
    total += cnt;
    int v = (cnt + 63) >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(v, off); if (lane >= off) v += u; }
    s_incl[m] = TC + v;
    s_cnt[m] = cnt;
    TC += __builtin_amdgcn_readlane(v, 63);
  }
  __syncthreads();
  if (acc != nullptr && blockIdx.x == 0) {     // sampled launch (profiling): its searches
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
    if (lane == 0) { atomicAdd(&acc[0], (unsigned long long)total); atomicAdd(&acc[65], 1ull); }
  }
  for (int ck = (int)blockIdx.x; ck < TC; ck += (int)gridDim.x) {
    int m = 0;
#pragma unroll
    for (int c = 0; c < LSD_OBS_WALK_MAX_MAPS / 64; c++) m += __popcll(__ballot(s_incl[c * 64 + lane] <= ck));
    m = __builtin_amdgcn_readfirstlane(m);
    const int e = ((ck - (m > 0 ? s_incl[m - 1] : 0)) << 6) + lane;
    int steps = 0;
    if (e < s_cnt[m]) {
      const ObserveArgs& a = items[m];
      const float4 q = a.queue[e];
      [[clang::always_inline]] steps = observe_back<true>(a, __float_as_int(q.x), 0, q.y, q.z);   // (as a call the function is compiled without the kernel's register budget)
    }
    if (acc != nullptr) {                         // ... and walk steps
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) steps += __shfl_xor(steps, off);
      if (lane == 0) atomicAdd(&acc[1 + (ck & 63)], (unsigned long long)steps);
    }
  }
}

struct RegArgs {
  HypPlanes m;
  LSD_G const uint8_t* validIn;   // snapshot plane (read)
  LSD_G uint8_t* validOut;        // new validity plane (written for every pixel)
  LSD_G const float* kfMaxGrad;
  int w, h;
  float minUseGrad, regDistVar;
  int validityTH;
  int tileRow0;             // k_reg_fused on a range of tile rows (8 rows each): first tile row of the launch (0: whole map)
  int tileRows;             // > 0: the launch's grid may hold more tile rows than this part has (several parts per launch): the rest leave
};

// K5: regularizeDepthMapFillHolesRow (DepthMap.cpp:656-703); 5x5 validity sum replaces the integral image.
__global__ __launch_bounds__(256) void k_fill_holes(RegArgs a) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31);
  const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= a.w || y >= a.h) return;
  const int idx = x + y * a.w;
  const uint8_t wasValid = a.validIn[idx];
  uint8_t nowValid = wasValid;
  if (x >= 3 && x < a.w - 2 && y >= 3 && y < a.h - 2 && !wasValid && !(a.kfMaxGrad[idx] < a.minUseGrad)) {
    int val = 0;
    for (int yy = y - 2; yy <= y + 2; yy++)
      for (int xx = x - 2; xx <= x + 2; xx++) {
        int j = xx + yy * a.w;
        if (a.validIn[j]) val += a.m.validity[j];
      }
    if ((a.m.blacklisted[idx] >= MIN_BLACKLIST && val > VAL_SUM_MIN_FOR_CREATE) || val > VAL_SUM_MIN_FOR_UNBLACKLIST) {
      float sumIdepthObs = 0, sumIVarObs = 0;
      for (int yy = y - 2; yy <= y + 2; yy++)
        for (int xx = x - 2; xx <= x + 2; xx++) {
          int j = xx + yy * a.w;
          if (!a.validIn[j]) continue;
          float sv = a.m.var[j];
          sumIdepthObs += a.m.idepth[j] / sv;
          sumIVarObs += 1.0f / sv;
        }
      float idepthObs = sumIdepthObs / sumIVarObs;
      idepthObs = unzero(idepthObs);
      nowValid = 1;
      a.m.blacklisted[idx] = 0;
      a.m.nextID[idx] = 0;
      a.m.validity[idx] = 0;
      a.m.idepth[idx] = idepthObs;
      a.m.var[idx] = VAR_RANDOM_INIT_INITIAL;
      a.m.idepth_s[idx] = -1;
      a.m.var_s[idx] = -1;
    }
  }
  a.validOut[idx] = nowValid;
}

// K6: regularizeDepthMapRow<removeOcclusions> (DepthMap.cpp:758-848)
template <bool removeOcclusions>
__global__ __launch_bounds__(256) void k_regularize(RegArgs a) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31);
  const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= a.w || y >= a.h) return;
  const int idx = x + y * a.w;
  const uint8_t wasValid = a.validIn[idx];
  uint8_t nowValid = wasValid;
  if (x >= 2 && x < a.w - 2 && y >= 2 && y < a.h - 2 && wasValid) {
    const float c_id = a.m.idepth[idx];
    const float c_var = a.m.var[idx];
    float sum = 0, val_sum = 0, sumIvar = 0;
    int numOccluding = 0, numNotOccluding = 0;
    for (int dx = -2; dx <= 2; dx++)
      for (int dy = -2; dy <= 2; dy++) {
        int j = idx + dx + dy * a.w;
        if (!a.validIn[j]) continue;
        float s_id = a.m.idepth[j];
        float s_var = a.m.var[j];
        float diff = s_id - c_id;
        if (DIFF_FAC_SMOOTHING * diff * diff > s_var + c_var) {
          if (removeOcclusions) { if (s_id > c_id) numOccluding++; }
          continue;
        }
        val_sum += a.m.validity[j];
        if (removeOcclusions) numNotOccluding++;
        float distFac = (float)(dx * dx + dy * dy) * a.regDistVar;
        float ivar = lsd_rcp_exact(s_var + distFac);
        sum += s_id * ivar;
        sumIvar += ivar;
      }
    if (val_sum < a.validityTH) {
      nowValid = 0;
      a.m.blacklisted[idx] = a.m.blacklisted[idx] - 1;
    } else if (removeOcclusions && numOccluding > numNotOccluding) {
      nowValid = 0;
    } else {
      sum = sum / sumIvar;
      sum = unzero(sum);
      a.m.idepth_s[idx] = sum;
      a.m.var_s[idx] = lsd_rcp_exact(sumIvar);
    }
  }
  a.validOut[idx] = nowValid;
}

// K5 + K6 (+ K8) fused over one LDS tile: regularizeDepthMapFillHoles -> regularizeDepthMap<removeOcclusions> ->
// Frame::setDepth, in one launch.  A workgroup owns 32x8 pixels; K6 on them needs the post-K5 map on a 2-pixel halo, and
// K5 on that 36x12 region needs the pre-K5 snapshot on a 4-pixel halo, so the tile is 40x16 (36x12 without K5).  The
// halo's K5 results are recomputed locally (identical arithmetic => identical values in every workgroup that needs
// them) and only the owner writes them to HBM.  Every neighbour access is an LDS read issued in a fully unrolled batch;
// the per-pixel arithmetic and its order are those of k_fill_holes / k_regularize above (and of the reference).
//   id0 / var0 / partials: outputs of K8 (SET only): level-0 idepth planes of the keyframe and per-workgroup (sum, count).
// KF (keyframe change, lsdhip_depth_change_keyframe_batch): the pass is finalizeKeyFrame's, and the owner of a pixel goes on with what
//   the reference does next with that pixel's finalised hypothesis — Frame::takeReActivationData (kx->react*) and phase A of
//   propagateDepth into the new keyframe (prop_candidate on kx->prop) — while the values are in its registers.
// SUMV: per-workgroup (sum of idepth_smoothed, count) over the valid pixels after the pass -> partials (createKeyFrame's rescale sums).
struct KfExtra {
  LSD_G float* reactId;        // old keyframe's re-activation planes
  LSD_G float* reactVar;
  LSD_G uint8_t* reactVal;
  PropArgs prop;         // (src unused)
};
// PY: pixels per lane = units of 8 map rows a workgroup owns (owned tile 32 x 8 PY).  The taller tile loads (32 + 2 HALO) x (8 PY + 2 HALO)
// entries for 256 PY owned pixels — 2.5x (PY = 1), 1.88x (2), 1.56x (4) with K5 — and recomputes K5 on 1.69x / 1.41x / 1.27x of them; the
// per-pixel arithmetic does not know the tile it runs in, so every shape gives the same planes bit for bit (profiles/r06_notes.md section 5).
template <bool FILL, bool OCC, bool SET, bool KF, bool SUMV, int PY>
__device__ __forceinline__ void reg_fused_tile(const RegArgs& a, float* __restrict__ id0, float* __restrict__ var0,
                                               double* __restrict__ partials, const KfExtra* __restrict__ kx = nullptr) {
  constexpr int HALO = FILL ? 4 : 2;
  constexpr int OH = 8 * PY;
  constexpr int TW = 32 + 2 * HALO, TH = OH + 2 * HALO, TN = TW * TH;
  // Per tile entry ONE 16-byte word (idepth, idepth_var, (float)pk) with pk = validity_counter of a valid pixel (never negative), -1 for an
  // invalid one: a neighbour costs K6 one LDS round trip (round 4: two dependent ones — the validity word, then, behind the branch on
  // it, the hypothesis; rounds 1-3: four arrays).  s_pk0 is the pre-K5 snapshot of pk that K5 reads.
  __shared__ int s_pk0[TN];
  __shared__ uint8_t s_created[TN];
  typedef float rege __attribute__((ext_vector_type(4)));
  typedef float rege3 __attribute__((ext_vector_type(3)));   // (what K6 reads of an entry: a 12-byte load leaves no dead register behind)
  __shared__ rege s_e[TN];
  __shared__ double s_sum[4];
  __shared__ int s_cnt[4];
  const int tid = threadIdx.x;
  // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2, and every tile re-reads a 4-pixel halo of its
  // neighbours: give XCD x the contiguous tile range [x nt/8, (x+1) nt/8) (whole bands of tile rows), so that a halo line is
  // fetched into one L2 instead of into the L2s of up to four XCDs.
  const int nt = gridDim.x * gridDim.y;
  const int lin = blockIdx.x + blockIdx.y * gridDim.x;
  const int tIdx = (nt & 7) == 0 ? (lin & 7) * (nt >> 3) + (lin >> 3) : lin;
  // tile rows are counted in units of 8 map rows (the C ABI's unit, lsdhip_depth_stage_rows): this workgroup owns PY of them
  const int unit0 = (tIdx / (int)gridDim.x) * PY;
  if (a.tileRows > 0 && unit0 >= a.tileRows) return;
  const int tbx = tIdx % gridDim.x;
  const int x0 = tbx * 32 - HALO, y0 = (unit0 + a.tileRow0) * 8 - HALO;
  const int ownedRows = a.tileRows > 0 ? (a.tileRows - unit0 < PY ? (a.tileRows - unit0) * 8 : OH) : OH;   // a part may end inside the tile
  const int w = a.w, h = a.h;

  // ---- tile load (unconditional, clamped addresses; out-of-image entries are invalid) ----
#pragma unroll
  for (int it = 0; it < (TN + 255) / 256; it++) {
    const int e = tid + it * 256;
    if (e < TN) {
      const int gx = x0 + e % TW, gy = y0 + e / TW;
      const bool inb = gx >= 0 && gx < w && gy >= 0 && gy < h;
      const int g = inb ? gx + gy * w : 0;
      const uint8_t v = a.validIn[g];
      const int vc = a.m.validity[g];
      const float id = a.m.idepth[g];
      const float var = a.m.var[g];
      const int pk = (inb && v) ? vc : -1;
      s_pk0[e] = pk;
      s_created[e] = 0;
      s_e[e] = rege{id, var, (float)pk, 0.f};             // (the counter as a float: K6 adds it to a float sum once per neighbour)
    }
  }
  // K5's gradient test reads the keyframe's maxGradients at the pixels of its region: issued with the tile loads (one memory round
  // trip for the workgroup instead of two; the lines are needed by nearly every wave of a semi-dense map anyway)
  constexpr int FW_ = TW - 4, FN_ = FW_ * (TH - 4);
  float mg[FILL ? (FN_ + 255) / 256 : 1];
  if (FILL) {
#pragma unroll
    for (int it = 0; it < (FN_ + 255) / 256; it++) {
      const int f = tid + it * 256;
      const int x = x0 + 2 + f % FW_, y = y0 + 2 + f / FW_;
      const bool inb = f < FN_ && x >= 0 && x < w && y >= 0 && y < h;
      mg[it] = a.kfMaxGrad[inb ? x + y * w : 0];
    }
  }
  __syncthreads();

  if (FILL) {
    // ---- K5 on the inner (TW-4) x (TH-4) region ----
    constexpr int FW = TW - 4, FH = TH - 4, FN = FW * FH;
#pragma unroll
    for (int it = 0; it < (FN + 255) / 256; it++) {
      const int f = tid + it * 256;
      if (f < FN) {
        const int lx = 2 + f % FW, ly = 2 + f / FW;
        const int e = lx + ly * TW;
        const int x = x0 + lx, y = y0 + ly;
        if (x >= 3 && x < w - 2 && y >= 3 && y < h - 2 && s_pk0[e] < 0) {
          const int idx = x + y * w;
          if (!(mg[it] < a.minUseGrad)) {
            int val = 0;
#pragma unroll
            for (int dy = -2; dy <= 2; dy++)
#pragma unroll
              for (int dx = -2; dx <= 2; dx++) {
                const int j = e + dx + dy * TW;
                const int pk0 = s_pk0[j];
                val += pk0 > 0 ? pk0 : 0;
              }
            if (val > VAL_SUM_MIN_FOR_UNBLACKLIST || (val > VAL_SUM_MIN_FOR_CREATE && a.m.blacklisted[idx] >= MIN_BLACKLIST)) {
              float sumIdepthObs = 0, sumIVarObs = 0;
#pragma unroll
              for (int dy = -2; dy <= 2; dy++)
#pragma unroll
                for (int dx = -2; dx <= 2; dx++) {
                  const int j = e + dx + dy * TW;
                  if (s_pk0[j] >= 0) {
                    const rege iv = s_e[j];
                    const float sv = iv.y;
                    sumIdepthObs += iv.x / sv;
                    sumIVarObs += lsd_rcp_exact(sv);
                  }
                }
              float idepthObs = sumIdepthObs / sumIVarObs;
              idepthObs = unzero(idepthObs);
              s_created[e] = 1;
              // nobody reads these entries during K5 (their snapshot validity is 0)
              s_e[e] = rege{idepthObs, VAR_RANDOM_INIT_INITIAL, 0.f, 0.f};
            }
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- K6 on the owned 32 x (8 PY) pixels: lane (tid & 31, tid >> 5) takes rows tid >> 5, + 8, ... ----
  // the six distance terms (dx^2 + dy^2) * regDistVar of the 5x5 window and the reciprocal's class mask, once, in scalar registers: as
  // literals inside the loop each costs a move (+ a multiply) per neighbour in a loop of ~17 vector instructions per neighbour
  auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
  const float df0 = uni(0.0f * a.regDistVar), df1 = uni(1.0f * a.regDistVar), df2 = uni(2.0f * a.regDistVar), df4 = uni(4.0f * a.regDistVar),
              df5 = uni(5.0f * a.regDistVar), df8 = uni(8.0f * a.regDistVar);
  const int nmask = lsd_rcp_mask();
  double accS = 0;          // (sum, count) of this lane's pixels: createKeyFrame's rescale sums (SUMV) or Frame::setDepth's statistics (SET)
  int accC = 0;
#pragma unroll
  for (int p = 0; p < PY; p++) {
  const int lx = HALO + (tid & 31), ly = HALO + (tid >> 5) + 8 * p;
  const int e = lx + ly * TW;
  const int x = x0 + lx, y = y0 + ly;
  const bool inImage = x < w && y < h && ly - HALO < ownedRows;
  const int idx = inImage ? x + y * w : 0;
  const bool created = FILL && s_created[e] != 0;
  const rege c_e = s_e[e];
  const int c_pk = (int)c_e.z;
  const uint8_t wasValid = c_pk >= 0 ? 1 : 0;
  uint8_t nowValid = wasValid;
  bool smoothed = false, blacklistDec = false;
  float out_ids = 0.f, out_vars = 0.f;
  // (the 25-neighbour loop on compacted centres — ballots, a list in LDS, the outcome handed back to the pixel's owner — measured the
  // same as this form at 30 % and at 50 % valid pixels: profiles/r05_notes.md section 5)
  const bool doK6 = inImage && x >= 2 && x < w - 2 && y >= 2 && y < h - 2 && wasValid;
  if (doK6) {
    const float c_id = c_e.x;
    const float c_var = c_e.y;
    float sum = 0, val_sum = 0, sumIvar = 0;
    int numOccluding = 0, numNotOccluding = 0;
    // (the branches stay: on semi-dense maps whole waves skip most neighbours — the predicated form of this loop, every neighbour's
    // arithmetic for every valid centre, ran the 32-map pass in 214 us against 166, profiles/r05_notes.md section 5.)
    // The next neighbour's entry is requested before the current one is worked on (volatile: the read stays where it is written instead
    // of sinking below the branches to its first use), so an LDS round trip overlaps the arithmetic instead of preceding it.
    auto entry = [&](int k) { const int dx = k / 5 - 2, dy = k % 5 - 2; return *(const volatile __attribute__((address_space(3))) rege3*)&s_e[e + dx + dy * TW]; };
    rege3 ej = entry(0);
#pragma unroll
    for (int k = 0; k < 25; k++) {
      const int dx = k / 5 - 2, dy = k % 5 - 2;     // dx outer, dy inner: the reference's order of the sums
      const rege3 en = entry(k < 24 ? k + 1 : 24);
      const float vcj = ej.z;                       // validity counter (small integer, exact) or -1
      const float s_idj = ej.x;
      const float s_varj = ej.y;
      ej = en;
      if (vcj < 0) continue;
      const float diff = s_idj - c_id;
      if (DIFF_FAC_SMOOTHING * diff * diff > s_varj + c_var) {
        if (OCC) { if (s_idj > c_id) numOccluding++; }
        continue;
      }
      val_sum += vcj;
      if (OCC) numNotOccluding++;
      const int d2 = dx * dx + dy * dy;             // (float)(d2) * regDistVar, as the reference forms it
      const float distFac = d2 == 0 ? df0 : (d2 == 1 ? df1 : (d2 == 2 ? df2 : (d2 == 4 ? df4 : (d2 == 5 ? df5 : df8))));
      const float ivar = lsd_rcp_exact(s_varj + distFac, nmask);
      sum += s_idj * ivar;
      sumIvar += ivar;
    }
    if (val_sum < a.validityTH) {
      nowValid = 0;
      blacklistDec = true;
    } else if (OCC && numOccluding > numNotOccluding) {
      nowValid = 0;
    } else {
      sum = sum / sumIvar;
      out_ids = unzero(sum);
      out_vars = lsd_rcp_exact(sumIvar);
      smoothed = true;
    }
  }
  int blFinal = 0;           // the pixel's blacklisted counter after the pass, where this lane came to know it
  bool blKnown = false;
  if (inImage) {
    a.validOut[idx] = nowValid;
    if (created) {
      a.m.nextID[idx] = 0;
      a.m.validity[idx] = 0;
      a.m.idepth[idx] = c_e.x;
      a.m.var[idx] = c_e.y;
      a.m.blacklisted[idx] = blacklistDec ? -1 : 0;
      blFinal = blacklistDec ? -1 : 0; blKnown = true;
      if (!smoothed) { a.m.idepth_s[idx] = -1; a.m.var_s[idx] = -1; }
    } else if (blacklistDec) {
      const int b0 = a.m.blacklisted[idx];
      a.m.blacklisted[idx] = b0 - 1;
      blFinal = b0 - 1; blKnown = true;
    }
    if (smoothed) { a.m.idepth_s[idx] = out_ids; a.m.var_s[idx] = out_vars; }
  }
  // the pixel's smoothed pair after the pass (SET / SUMV / KF)
  float ids = -1.f, vars = -1.f;
  if ((SET || SUMV || KF) && inImage) {
    if (smoothed) { ids = out_ids; vars = out_vars; }
    else if (created || !nowValid) { ids = -1.f; vars = -1.f; }    // (not valid: the value is not looked at)
    else { ids = a.m.idepth_s[idx]; vars = a.m.var_s[idx]; }      // valid, not smoothed: the 2-pixel border only
  }
  if (KF && inImage) {
    // Frame::takeReActivationData (Frame.cpp:107-145) of the keyframe being finalised, then this hypothesis' step into the new keyframe
    if (nowValid) {
      kx->reactId[idx] = c_e.x;
      kx->reactVar[idx] = c_e.y;
      kx->reactVal[idx] = (uint8_t)c_pk;
      prop_candidate(kx->prop, x, y, idx, ids, c_e.y, c_pk);
    } else {
      const int bl = blKnown ? blFinal : a.m.blacklisted[idx];
      kx->reactVar[idx] = bl < MIN_BLACKLIST ? -2.f : -1.f;
    }
  }
  if (SUMV) {
    // createKeyFrame's rescale sums (DepthMap.cpp:1286-1294): idepth_smoothed over the valid pixels
    if (inImage && nowValid) { accS += ids; accC += 1; }
  }
  if (SET) {
    // ---- K8: Frame::setDepth on the owned pixel ----
    if (inImage) {
      if (nowValid && (double)ids >= -0.05) {
        id0[idx] = ids;
        var0[idx] = vars;
        accS += ids;
        accC += 1;
      } else {
        id0[idx] = -1.f;
        var0[idx] = -1.f;
      }
    }
  }
  }   // pixels of this lane
  if (SUMV || SET) {
    // (sum, count) of the workgroup: butterfly inside each wave, then four partials through LDS.  The sum of at most 256 PY floats
    // of one sign and a dynamic range below 2^20 is exact in double, so its value does not depend on the order.
    double s = accS;
    int c = accC;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      s += __shfl_xor(s, off);
      c += __shfl_xor(c, off);
    }
    if ((tid & 63) == 0) { s_sum[tid >> 6] = s; s_cnt[tid >> 6] = c; }
    __syncthreads();
    if (tid == 0) {
      const int b = tIdx;
      partials[2 * b] = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
      partials[2 * b + 1] = (double)(((s_cnt[0] + s_cnt[1]) + s_cnt[2]) + s_cnt[3]);
    }
  }
}
template <bool FILL, bool OCC, bool SET, int PY>
__global__ __launch_bounds__(256) void k_reg_fused(RegArgs a, float* __restrict__ id0, float* __restrict__ var0,
                                                    double* __restrict__ partials) {
  reg_fused_tile<FILL, OCC, SET, false, false, PY>(a, id0, var0, partials);
}
// the update-time pass (fill holes + regularise [+ setDepth]) over the maps of several sequences, blockIdx.z = map
struct RegBatchItem {
  RegArgs a;
  LSD_G float* id0;          // null: this map's keyframe is not due for Frame::setDepth (depthHasBeenUpdatedFlag still set)
  LSD_G float* var0;
  LSD_G double* partials;
};
template <bool SET, int PY>
__global__ __launch_bounds__(256) void k_reg_fused_batch(const RegBatchItem* __restrict__ items) {
  const RegBatchItem& it = items[blockIdx.z];
  if (SET) {
    if (it.id0 == nullptr) return;
    reg_fused_tile<true, false, true, false, false, PY>(it.a, it.id0, it.var0, it.partials);
  } else {
    if (it.id0 != nullptr) return;
    reg_fused_tile<true, false, false, false, false, PY>(it.a, nullptr, nullptr, nullptr);
  }
}

// ---- keyframe change of n maps in shared launches (lsdhip_depth_change_keyframe_batch), blockIdx.z = map -------------------------------
//   k_kf_finalize_prop     finalizeKeyFrame's pass (fill holes + regularise + Frame::setDepth of the old keyframe) + takeReActivationData
//                          + phase A of propagateDepth, per owner pixel
//   k_prop_resolve_batch   phase B into the new map
//   k_kf_reg<false, true>  regularizeDepthMap(true, ...)        (createKeyFrame, DepthMap.cpp:1271)
//   k_kf_reg<true, false>  fill holes + regularizeDepthMap(false, ...) + the rescale sums per tile   (:1278-1294)
//   k_kf_rescale_setdepth  mean inverse depth to one (:1296-1304) + Frame::setDepth of the new keyframe
//   k_idepth_pyramid_batch both keyframes' pyramids and their (sum, count) records
struct KfItem {
  RegArgs a;
  LSD_G float* id0;
  LSD_G float* var0;
  LSD_G double* partials;
  KfExtra x;
};
template <int PY>
__global__ __launch_bounds__(256) void k_kf_finalize_prop(const KfItem* __restrict__ items) {
  const KfItem& it = items[blockIdx.z];
  reg_fused_tile<true, false, true, true, false, PY>(it.a, it.id0, it.var0, it.partials, &it.x);
}
template <bool FILL, bool OCC, int PY>
__global__ __launch_bounds__(256) void k_kf_reg(const RegBatchItem* __restrict__ items) {
  const RegBatchItem& it = items[blockIdx.z];
  reg_fused_tile<FILL, OCC, false, false, FILL, PY>(it.a, nullptr, nullptr, it.partials, nullptr);   // (the rescale sums ride on the second pass)
}
struct RescaleItem {
  HypPlanes m;
  LSD_G const double* sumPartials;   // per-tile (sum, count) of the pass before
  int nSumPartials;
  LSD_G int* flags;                  // propagation flags: [0] -> slot, then both zeroed
  LSD_G double* slot;                // pinned DeferredSlot of the rescale factor: (sum, count, flag)
  LSD_G float* id0;                  // new keyframe's level-0 planes
  LSD_G float* var0;
  LSD_G double* statPartials;        // (sum, count) per workgroup of this launch
  int n;
};
// Tile height of a regulariser launch: workgroups that own 32 x 16 pixels (PY = 2) move a quarter less through LDS per owned pixel and
// recompute a sixth less of K5 than 32 x 8 ones — where the launch has enough of them to fill the chip (several maps, large frames:
// -19 % for the 32-map pass, -6 % at 3840x2160); one 640x480 map is 1200 short workgroups or 600 long ones on 256 CUs, and the short ones
// win there (keyframe change 0.085 against 0.097 ms).  32 x 32 (PY = 4, 33.6 KB of LDS) lost everywhere.  profiles/r06_notes.md section 5.
static inline int lsd_reg_py(long long workgroups_at_py1) {
  static const int forced = getenv("LSDHIP_REG_PY") ? atoi(getenv("LSDHIP_REG_PY")) : 0;     // developer A/B
  if (forced == 1 || forced == 2) return forced;
  return workgroups_at_py1 >= 2048 ? 2 : 1;
}
// grid rows of a regulariser launch over `units` tile rows of 8 map rows each
static inline int lsd_reg_grid_rows(int units, int py) { return (units + py - 1) / py; }
#define LSD_REG_DISPATCH(py, ...) do { if ((py) == 2) { constexpr int PYV = 2; __VA_ARGS__; } else { constexpr int PYV = 1; __VA_ARGS__; } } while (0)
#define LSD_RESCALE_PX 2048    // pixels per workgroup of k_kf_rescale_setdepth
__global__ __launch_bounds__(256) void k_kf_rescale_setdepth(const RescaleItem* __restrict__ items) {
  const RescaleItem& it = items[blockIdx.y];
  __shared__ double s_a[256], s_b[256];
  const int tid = threadIdx.x;
  // every workgroup adds the tiles' partials in the same fixed order: the same factor everywhere, no second launch, no atomics
  double sa = 0, sb = 0;
  for (int i = tid; i < it.nSumPartials; i += 256) { sa += it.sumPartials[2 * i]; sb += it.sumPartials[2 * i + 1]; }
  s_a[tid] = sa; s_b[tid] = sb;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { s_a[tid] += s_a[tid + off]; s_b[tid] += s_b[tid + off]; }
    __syncthreads();
  }
  const double sumD = s_a[0], cntD = s_b[0];
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0) {
    it.slot[0] = sumD; it.slot[1] = cntD; it.slot[2] = (double)it.flags[0];
    it.flags[0] = 0; it.flags[1] = 0;
  }
  // rescaleFactor = numIdepth / sumIdepth with both cast to float first (DepthMap.cpp:1286-1294)
  const float rescaleFactor = (float)cntD / (float)sumD;
  const float rescaleFactor2 = rescaleFactor * rescaleFactor;
  double s = 0;
  int c = 0;
#pragma unroll
  for (int k = 0; k < LSD_RESCALE_PX / 256; k++) {
    const int i = blockIdx.x * LSD_RESCALE_PX + k * 256 + tid;
    if (i < it.n) {
      float ids = -1.f, vars = -1.f;
      const bool valid = it.m.valid[i] != 0;
      if (valid) {
        it.m.idepth[i] *= rescaleFactor;
        ids = it.m.idepth_s[i] * rescaleFactor;
        it.m.idepth_s[i] = ids;
        it.m.var[i] *= rescaleFactor2;
        vars = it.m.var_s[i] * rescaleFactor2;
        it.m.var_s[i] = vars;
      }
      // Frame::setDepth (Frame.cpp:199-243)
      if (valid && (double)ids >= -0.05) {
        it.id0[i] = ids;
        it.var0[i] = vars;
        s += ids;
        c += 1;
      } else {
        it.id0[i] = -1.f;
        it.var0[i] = -1.f;
      }
    }
  }
  s_a[tid] = s; s_b[tid] = (double)c;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { s_a[tid] += s_a[tid + off]; s_b[tid] += s_b[tid + off]; }
    __syncthreads();
  }
  if (tid == 0) { it.statPartials[2 * blockIdx.x] = s_a[0]; it.statPartials[2 * blockIdx.x + 1] = s_b[0]; }
}

// parts of the fused fill-holes + regularise pass of several maps in one launch (row-band decomposition: the windows of one process),
// blockIdx.z = part
template <int PY>
__global__ __launch_bounds__(256) void k_reg_rows_batch(const RegBatchItem* __restrict__ items) {
  const RegBatchItem& it = items[blockIdx.z];
  reg_fused_tile<true, false, false, false, false, PY>(it.a, nullptr, nullptr, nullptr);
}

// K8: Frame::setDepth (Frame.cpp:199-243) + per-workgroup (sum, count) partials for meanIdepth / numPoints
__global__ __launch_bounds__(256) void k_set_depth(HypPlanes m, float* __restrict__ id0, float* __restrict__ var0, int n, double* __restrict__ partials) {
  __shared__ double s_sum[256];
  __shared__ int s_cnt[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double s = 0;
  int c = 0;
  if (i < n) {
    float ids = m.idepth_s[i];
    if (m.valid[i] && (double)ids >= -0.05) {
      id0[i] = ids;
      var0[i] = m.var_s[i];
      s = ids;
      c = 1;
    } else {
      id0[i] = -1.f;
      var0[i] = -1.f;
    }
  }
  s_sum[threadIdx.x] = s;
  s_cnt[threadIdx.x] = c;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) { s_sum[threadIdx.x] += s_sum[threadIdx.x + off]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = s_sum[0]; partials[2 * blockIdx.x + 1] = (double)s_cnt[0]; }
}
// sum of idepth_smoothed over valid pixels (createKeyFrame rescale, DepthMap.cpp:1286-1294)
__global__ __launch_bounds__(256) void k_sum_valid_idepth(HypPlanes m, int n, double* __restrict__ partials) {
  __shared__ double s_sum[256];
  __shared__ int s_cnt[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double s = 0;
  int c = 0;
  if (i < n && m.valid[i]) { s = m.idepth_s[i]; c = 1; }
  s_sum[threadIdx.x] = s;
  s_cnt[threadIdx.x] = c;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) { s_sum[threadIdx.x] += s_sum[threadIdx.x + off]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = s_sum[0]; partials[2 * blockIdx.x + 1] = (double)s_cnt[0]; }
}
// out = pinned DeferredSlot (sum, count, flag); dev_out = the same pair in HBM for a kernel that follows (k_rescale)
__global__ __launch_bounds__(256) void k_reduce_pairs(const double* __restrict__ partials, int nblocks, double* __restrict__ out,
                                                       double* __restrict__ dev_out, const int* __restrict__ flag) {
  __shared__ double s_a[256], s_b[256];
  double a = 0, b = 0;
  for (int i = threadIdx.x; i < nblocks; i += 256) { a += partials[2 * i]; b += partials[2 * i + 1]; }
  s_a[threadIdx.x] = a;
  s_b[threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) { s_a[threadIdx.x] += s_a[threadIdx.x + off]; s_b[threadIdx.x] += s_b[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = s_a[0]; out[1] = s_b[0]; out[2] = flag ? (double)flag[0] : 0.0;
    if (dev_out) { dev_out[0] = s_a[0]; dev_out[1] = s_b[0]; }
  }
}
// rescale (DepthMap.cpp:1296-1304)
__global__ __launch_bounds__(256) void k_rescale(HypPlanes m, int n, const double* __restrict__ sumCount) {
  // rescaleFactor = numIdepth / sumIdepth with both cast to float first (DepthMap.cpp:1286-1294)
  const float sumIdepth = (float)sumCount[0], numIdepth = (float)sumCount[1];
  const float rescaleFactor = numIdepth / sumIdepth;
  const float rescaleFactor2 = rescaleFactor * rescaleFactor;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || !m.valid[i]) return;
  m.idepth[i] *= rescaleFactor;
  m.idepth_s[i] *= rescaleFactor;
  m.var[i] *= rescaleFactor2;
  m.var_s[i] *= rescaleFactor2;
}

// phase A: every valid source hypothesis computes its target pixel and candidate, and registers in the target's slot list
__global__ __launch_bounds__(256) void k_prop_candidates(PropArgs a) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31);
  const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= a.w || y >= a.h) return;
  const int idx = x + y * a.w;
  if (!a.src.valid[idx]) return;
  prop_candidate(a, x, y, idx, a.src.idepth_s[idx], a.src.var[idx], a.src.validity[idx]);
}
// one step of the reference's per-target merge (DepthMap.cpp:587-640), sources replayed in the order its double loop visits them
__device__ __forceinline__ void prop_merge(const float4 c, bool& tValid, float& t_idepth, float& t_var, int& t_validity) {
  float new_idepth = c.x, new_var = c.y;
  int src_validity = __float_as_int(c.z);
  if (tValid) {
    float diff = t_idepth - new_idepth;
    if (DIFF_FAC_PROP_MERGE * diff * diff > new_var + t_var) {
      if (new_idepth < t_idepth) return;
      else tValid = false;
    }
  }
  if (!tValid) {
    tValid = true;
    t_idepth = new_idepth; t_var = new_var; t_validity = src_validity;
  } else {
    float w = new_var / (t_var + new_var);
    float merged_new_idepth = w * t_idepth + (1.0f - w) * new_idepth;
    int merged_validity = src_validity + t_validity;
    if (merged_validity > VALIDITY_COUNTER_MAX + (VALIDITY_COUNTER_MAX_VARIABLE))
      merged_validity = VALIDITY_COUNTER_MAX + (VALIDITY_COUNTER_MAX_VARIABLE);
    float mv = 1.0f / (1.0f / t_var + 1.0f / new_var);
    t_idepth = merged_new_idepth; t_var = mv; t_validity = merged_validity;
  }
}
// phase B: per target, replay its candidates in source order (row-major, as the reference's double loop visits them)
__device__ __forceinline__ void prop_resolve_pixel(const PropArgs& a, const int i) {
  const int cnt = a.slotCount[i];
  int n = cnt;
  if (n > PROP_SLOT_CAP) n = PROP_SLOT_CAP;
  int srcs[PROP_SLOT_CAP];
  for (int k = 0; k < PROP_SLOT_CAP; k++) srcs[k] = (k < n) ? a.slots[(size_t)i * PROP_SLOT_CAP + k] : 0x7fffffff;
  for (int p = 1; p < PROP_SLOT_CAP; p++) {  // insertion sort, ascending source index
    int key = srcs[p];
    int q = p - 1;
    while (q >= 0 && srcs[q] > key) { srcs[q + 1] = srcs[q]; q--; }
    srcs[q + 1] = key;
  }
  bool tValid = false;
  float t_idepth = 0, t_var = 0;
  int t_validity = 0;
  if (cnt <= PROP_SLOT_CAP) {
    for (int k = 0; k < n; k++) prop_merge(a.cand[srcs[k]], tValid, t_idepth, t_var, t_validity);
  } else {
    // rare: slots + chain, replayed in ascending source order by repeated selection (no per-thread list needed)
    int last = -1;
    while (true) {
      int best = 0x7fffffff;
      for (int k = 0; k < PROP_SLOT_CAP; k++) if (srcs[k] > last && srcs[k] < best) best = srcs[k];
      for (int o = a.ovfHead[i]; o >= 0;) {
        const int2 e = a.ovf[o];
        if (e.x > last && e.x < best) best = e.x;
        o = e.y;
      }
      if (best == 0x7fffffff) break;
      prop_merge(a.cand[best], tValid, t_idepth, t_var, t_validity);
      last = best;
    }
  }
  a.dst.valid[i] = tValid ? 1 : 0;
  a.dst.blacklisted[i] = 0;
  if (tValid) {
    a.dst.nextID[i] = 0;
    a.dst.validity[i] = t_validity;
    a.dst.idepth[i] = t_idepth;
    a.dst.var[i] = t_var;
    a.dst.idepth_s[i] = -1;
    a.dst.var_s[i] = -1;
  }
  // clean at rest: the next propagation into this scratch starts without clears
  if (cnt != 0) a.slotCount[i] = 0;
  if (cnt > PROP_SLOT_CAP) a.ovfHead[i] = -1;      // (a chain exists only where the slots overflowed)
}
__global__ __launch_bounds__(256) void k_prop_resolve(PropArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.w * a.h) return;
  prop_resolve_pixel(a, i);
}
__global__ __launch_bounds__(256) void k_prop_resolve_batch(const KfItem* __restrict__ items) {
  const PropArgs& a = items[blockIdx.y].x.prop;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.w * a.h) return;
  prop_resolve_pixel(a, i);
}

// ---- initialisation / re-activation kernels ------------------------------------------------------------------------
// initializeFromGTDepth (DepthMap.cpp:979-1014)
__global__ __launch_bounds__(256) void k_init_gt(HypPlanes m, const float* __restrict__ idepth0, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = idepth0[i];
  if (!isnan(v) && v > 0) {
    m.valid[i] = 1; m.blacklisted[i] = 0; m.nextID[i] = 0; m.validity[i] = 20;
    m.idepth[i] = v; m.idepth_s[i] = v; m.var[i] = VAR_GT_INIT_INITIAL; m.var_s[i] = VAR_GT_INIT_INITIAL;
  } else {
    m.valid[i] = 0; m.blacklisted[i] = 0;
  }
}
// Frame::takeReActivationData (Frame.cpp:107-145)
__global__ __launch_bounds__(256) void k_take_react(HypPlanes m, float* __restrict__ id, float* __restrict__ var, uint8_t* __restrict__ val, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (m.valid[i]) { id[i] = m.idepth[i]; var[i] = m.var[i]; val[i] = (uint8_t)m.validity[i]; }
  else if (m.blacklisted[i] < MIN_BLACKLIST) var[i] = -2;
  else var[i] = -1;
}
// setFromExistingKF (DepthMap.cpp:937-959)
__global__ __launch_bounds__(256) void k_from_react(HypPlanes m, const float* __restrict__ id, const float* __restrict__ var, const uint8_t* __restrict__ val, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = var[i];
  if (v > 0) {
    m.valid[i] = 1; m.blacklisted[i] = 0; m.nextID[i] = 0; m.validity[i] = val[i];
    m.idepth[i] = id[i]; m.var[i] = v; m.idepth_s[i] = -1; m.var_s[i] = -1;
  } else {
    m.valid[i] = 0;
    m.blacklisted[i] = (v == -2) ? MIN_BLACKLIST - 1 : 0;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------
static int alloc_planes(HypPlanes& p, size_t n, void** base_out) {
  char* base = nullptr;
  size_t bytes = n * 29 + 8 * 256;
  HIPCHK(hipMalloc((void**)&base, bytes));
  HIPCHK(hipMemset(base, 0, bytes));
  HIPCHK(hipStreamSynchronize(nullptr));   // (hipMemset runs on the null stream, which this library's non-blocking streams do not wait for)
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  size_t off = 0;
  p.blacklisted = lsd_g((int32_t*)(base + off)); off = up(off + n * 4);
  p.nextID = lsd_g((float*)(base + off)); off = up(off + n * 4);
  p.validity = lsd_g((int32_t*)(base + off)); off = up(off + n * 4);
  p.idepth = lsd_g((float*)(base + off)); off = up(off + n * 4);
  p.var = lsd_g((float*)(base + off)); off = up(off + n * 4);
  p.idepth_s = lsd_g((float*)(base + off)); off = up(off + n * 4);
  p.var_s = lsd_g((float*)(base + off)); off = up(off + n * 4);
  p.valid = lsd_g((uint8_t*)(base + off));
  *base_out = base;
  return LSDHIP_OK;
}

static int depth_create_impl(lsdhip_ctx* c, lsdhip_depthmap* dm) {
  size_t n = (size_t)c->w * c->h;
  int rc = alloc_planes(dm->cur, n, &dm->bases[0]);
  if (rc) return rc;
  rc = alloc_planes(dm->oth, n, &dm->bases[1]);
  if (rc) return rc;
  HIPCHK(hipMalloc((void**)&dm->d_validSnap, n));
  dm->bases[2] = dm->d_validSnap;
  HIPCHK(hipMemset(dm->d_validSnap, 0, n));
  HIPCHK(hipStreamSynchronize(nullptr));
  HIPCHK(hipMalloc((void**)&dm->d_slotCount, n * 4));
  HIPCHK(hipMalloc((void**)&dm->d_slots, n * 4 * PROP_SLOT_CAP));
  HIPCHK(hipMalloc((void**)&dm->d_ovfHead, n * 4));
  HIPCHK(hipMalloc((void**)&dm->d_ovf, n * sizeof(int2)));
  HIPCHK(hipMalloc((void**)&dm->d_cand, n * 16));
  HIPCHK(hipMalloc((void**)&dm->d_flags, 64));
  HIPCHK(hipMemsetAsync(dm->d_flags, 0, 64, lsd_map_stream(c)));
  int nb = (int)((n + 255) / 256);
  const int ntiles = ((c->w + 31) / 32) * ((c->h + 7) / 8);   // partials of the fused regulariser: one pair per 32x8 tile
  // (sum, count) partials: [16, ...) the setDepth / rescale passes of the single calls; the keyframe change in shared launches keeps
  // three sets alive at once (old keyframe's setDepth | rescale sums | new keyframe's setDepth)
  dm->redStride = (size_t)2 * (nb > ntiles ? nb : ntiles);
  HIPCHK(hipMalloc((void**)&dm->d_red, (3 * dm->redStride + 16) * sizeof(double)));
  // K7 scratch is clean at rest (k_prop_resolve / the reader of the flags leave it so)
  HIPCHK(hipMemsetAsync(dm->d_slotCount, 0, n * 4, lsd_map_stream(c)));
  HIPCHK(hipMemsetAsync(dm->d_ovfHead, 0xFF, n * 4, lsd_map_stream(c)));
  dm->propClean = true;
  HIPCHK(hipHostMalloc((void**)&dm->h_red, 16 * sizeof(double), hipHostMallocMapped));
  return LSDHIP_OK;
}
extern "C" int lsdhip_depth_create(lsdhip_ctx* c, lsdhip_depthmap** out) {
  if (!c || !out) return LSDHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  LSD_CTX_LOCK(c);
  lsdhip_depthmap* dm = new lsdhip_depthmap();
  dm->ctx = c;
  c->depthmaps.push_back(dm);
  const int rc = depth_create_impl(c, dm);
  if (rc) { lsdhip_depth_destroy(dm); return rc; }   // frees whatever was allocated (null pointers are fine)
  *out = dm;
  return LSDHIP_OK;
}
// a frame is going away: no depth map may keep it as its active keyframe (a later frame can get the same heap address)
void lsd_depthmaps_forget_frame(lsdhip_ctx* c, lsdhip_frame* f) {
  for (lsdhip_depthmap* dm : c->depthmaps)
    if (dm->activeKeyFrame == f) { dm->activeKeyFrame = nullptr; dm->activeKeyFrameIsReactivated = false; }
}
extern "C" void lsdhip_depth_destroy(lsdhip_depthmap* dm) {
  if (!dm) return;
  {
    LSD_CTX_LOCK(dm->ctx);
    auto& v = dm->ctx->depthmaps;
    for (size_t i = 0; i < v.size(); i++) if (v[i] == dm) { v.erase(v.begin() + i); break; }
  }
  (void)hipSetDevice(dm->ctx->device);
  (void)hipStreamSynchronize(lsd_map_stream(dm->ctx));
  for (int i = 0; i < 3; i++) (void)hipFree(dm->bases[i]);  // plane pointers get swapped around; free the arenas
  (void)hipFree(dm->d_slotCount);
  (void)hipFree(dm->d_slots);
  (void)hipFree(dm->d_ovfHead);
  (void)hipFree(dm->d_ovf);
  (void)hipFree(dm->d_cand);
  (void)hipFree(dm->d_flags);
  (void)hipFree(dm->d_red);
  (void)hipHostFree(dm->h_red);
#ifdef LSD_PHASE_TRACE
  if (const char* path = getenv("LSDHIP_OBS_TRACE_FILE")) {
    if (dm->d_obs_trace) {
      std::vector<unsigned long long> h(dm->obs_trace_words);
      if (hipMemcpy(h.data(), dm->d_obs_trace, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
        if (FILE* f = fopen(path, "w")) {
          for (size_t i = 0; i + 16 <= h.size(); i += 16) { for (int k = 0; k < 16; k++) fprintf(f, "%llu ", h[i + k]); fprintf(f, "\n"); }
          fclose(f);
        }
      }
    }
  }
#endif
  for (int i = 0; i < 8; i++) if (dm->ev[i][0]) { (void)hipEventDestroy(dm->ev[i][0]); (void)hipEventDestroy(dm->ev[i][1]); }
  if (dm->d_stage) (void)hipFree(dm->d_stage);
  if (dm->h_stage) (void)hipHostFree(dm->h_stage);
  if (dm->d_obsCounters) (void)hipFree(dm->d_obsCounters);
  if (dm->d_obsQueue) (void)hipFree(dm->d_obsQueue);
  delete dm;
}
extern "C" int lsdhip_depth_is_valid(lsdhip_depthmap* dm) { return dm && dm->activeKeyFrame != nullptr; }
extern "C" int lsdhip_depth_invalidate(lsdhip_depthmap* dm) { if (!dm) return LSDHIP_E_ARG; dm->activeKeyFrame = nullptr; return LSDHIP_OK; }
extern "C" int lsdhip_depth_reset(lsdhip_depthmap* dm) {
  if (!dm) return LSDHIP_E_ARG;
  size_t n = (size_t)dm->ctx->w * dm->ctx->h;
  HIPCHK(hipMemsetAsync(dm->cur.valid, 0, n, lsd_map_stream(dm->ctx)));
  HIPCHK(hipMemsetAsync(dm->oth.valid, 0, n, lsd_map_stream(dm->ctx)));
  return LSDHIP_OK;
}

static int set_depth_finish(lsdhip_depthmap* dm, int nPartials);
// Frame::setDepth on the active keyframe: level-0 planes, idepth pyramid, meanIdepth / numPoints
static int set_depth(lsdhip_depthmap* dm) {
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  lsdhip_frame* kf = dm->activeKeyFrame;
  int n = c->w * c->h;
  int nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_set_depth, dim3(nb), dim3(256), 0, lsd_map_stream(c), dm->cur, lsd_depth_w(kf)[0], lsd_depthvar_w(kf)[0], n, dm->d_red + 16);
  return set_depth_finish(dm, nb);
}
// second half of Frame::setDepth: mean inverse depth / point count from the (sum, count) partials, idepth pyramid
static int set_depth_finish(lsdhip_depthmap* dm, int nPartials) {
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  lsdhip_frame* kf = dm->activeKeyFrame;
  const int slot = lsd_ctx_take_slot(c);
  if (slot < 0) return slot;
  if (kf->pendStats >= 0) c->slot_stats_owner[kf->pendStats] = nullptr;   // superseded by this setDepth
  kf->pendStats = slot;
  c->slot_stats_owner[slot] = kf;
  // the (sum, count) reduction rides on the pyramid launch (one extra workgroup): both depend only on the level-0 planes
  int rc = lsd_frame_build_idepth_pyramid(kf, dm->d_red + 16, nPartials, (double*)&c->h_slots[slot]);
  if (rc) return rc;
  kf->depthHasBeenUpdatedFlag = true;
  if (!c->async) return lsd_frame_resolve(kf);
  return LSDHIP_OK;
}

static RegArgs reg_args(lsdhip_depthmap* dm, int validityTH) {
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  RegArgs a;
  a.m = dm->cur;
  a.validIn = dm->cur.valid;
  a.validOut = lsd_g(dm->d_validSnap);
  a.kfMaxGrad = lsd_g(dm->activeKeyFrame->d_maxgrad);
  a.w = c->w; a.h = c->h;
  a.minUseGrad = c->params.minUseGrad;
  a.regDistVar = 0.075f * 0.075f * c->params.depthSmoothingFactor * c->params.depthSmoothingFactor;  // REG_DIST_VAR
  a.validityTH = validityTH;
  a.tileRow0 = 0;
  a.tileRows = 0;
  return a;
}
static void swap_valid(lsdhip_depthmap* dm) { uint8_t* v = (uint8_t*)dm->cur.valid; dm->cur.valid = lsd_g(dm->d_validSnap); dm->d_validSnap = v; }

static int fill_holes(lsdhip_depthmap* dm) {
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  RegArgs a = reg_args(dm, 0);
  hipLaunchKernelGGL(k_fill_holes, dim3((c->w + 31) / 32, (c->h + 7) / 8), dim3(256), 0, lsd_map_stream(c), a);
  HIPCHK(hipGetLastError());
  swap_valid(dm);
  return LSDHIP_OK;
}
static int regularize(lsdhip_depthmap* dm, bool removeOcclusions, int validityTH) {
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  RegArgs a = reg_args(dm, validityTH);
  const int tx = (c->w + 31) / 32, units = (c->h + 7) / 8, py = lsd_reg_py((long long)tx * units);
  dim3 grid(tx, lsd_reg_grid_rows(units, py));
  LSD_REG_DISPATCH(py,
    if (removeOcclusions) hipLaunchKernelGGL((k_reg_fused<false, true, false, PYV>), grid, dim3(256), 0, lsd_map_stream(c), a, nullptr, nullptr, nullptr);
    else hipLaunchKernelGGL((k_reg_fused<false, false, false, PYV>), grid, dim3(256), 0, lsd_map_stream(c), a, nullptr, nullptr, nullptr));
  HIPCHK(hipGetLastError());
  swap_valid(dm);
  return LSDHIP_OK;
}
// regularizeDepthMapFillHoles + regularizeDepthMap(false, validityTH) [+ Frame::setDepth] in one launch
static int fill_regularize(lsdhip_depthmap* dm, int validityTH, bool setDepth) {
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  lsdhip_frame* kf = dm->activeKeyFrame;
  RegArgs a = reg_args(dm, validityTH);
  const int tx = (c->w + 31) / 32, units = (c->h + 7) / 8, py = lsd_reg_py((long long)tx * units);
  dim3 grid(tx, lsd_reg_grid_rows(units, py));
  LSD_REG_DISPATCH(py,
    if (setDepth)
      hipLaunchKernelGGL((k_reg_fused<true, false, true, PYV>), grid, dim3(256), 0, lsd_map_stream(c), a, lsd_depth_w(kf)[0], lsd_depthvar_w(kf)[0], dm->d_red + 16);
    else
      hipLaunchKernelGGL((k_reg_fused<true, false, false, PYV>), grid, dim3(256), 0, lsd_map_stream(c), a, nullptr, nullptr, nullptr));
  HIPCHK(hipGetLastError());
  swap_valid(dm);
  if (setDepth) return set_depth_finish(dm, (int)(grid.x * grid.y));
  return LSDHIP_OK;
}

// The fused pass on tile rows [tileRow0, tileRow0 + nTileRows) only (a tile row = 8 map rows).  The parts of one pass may be issued
// in any order and on different streams: every part reads the snapshot validity plane and the pre-pass hypotheses (a pixel K5 creates
// was invalid in the snapshot, so nobody reads the fields its owner rewrites) and writes the other validity plane; `last` swaps the
// two once all parts are queued.  Rows outside the parts keep whatever the second validity plane held: the caller refreshes them
// (row-band decomposition: rows another band owns arrive with the halo exchange).
static int fill_regularize_rows(lsdhip_depthmap* dm, int validityTH, int tileRow0, int nTileRows, bool last) {
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  const int T = (c->h + 7) / 8;
  if (tileRow0 < 0 || nTileRows < 0 || tileRow0 + nTileRows > T) { lsd_set_error("depth stage rows: tile rows [%d, %d) outside [0, %d)", tileRow0, tileRow0 + nTileRows, T); return LSDHIP_E_ARG; }
  if (nTileRows > 0) {
    RegArgs a = reg_args(dm, validityTH);
    a.tileRow0 = tileRow0;
    a.tileRows = nTileRows;
    const int tx = (c->w + 31) / 32, py = lsd_reg_py((long long)tx * nTileRows);
    dim3 grid(tx, lsd_reg_grid_rows(nTileRows, py));
    LSD_REG_DISPATCH(py, hipLaunchKernelGGL((k_reg_fused<true, false, false, PYV>), grid, dim3(256), 0, lsd_map_stream(c), a, nullptr, nullptr, nullptr));
    HIPCHK(hipGetLastError());
  }
  if (last) swap_valid(dm);
  return LSDHIP_OK;
}

// Frame::prepareForStereoWith (Frame.cpp:295-317): Sim3 algebra in double, results cast to float
static void prepare_stereo(lsdhip_depthmap* dm, lsdhip_frame* fr, StereoRef& s) {
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  const lsdm::Sim3dH& thisToOther = fr->thisToParent_raw;
  lsdm::Sim3dH otherToThis = lsdm::sim3_inverse(thisToOther);
  double Rd[9];
  lsdm::quatd_to_rot(otherToThis.q, Rd);
  float Rf[9];
  for (int i = 0; i < 9; i++) Rf[i] = (float)Rd[i];
  const float* K = c->K0;
  float sc = (float)otherToThis.s;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      float acc = K[i * 3 + 0] * Rf[0 * 3 + j];
      acc += K[i * 3 + 1] * Rf[1 * 3 + j];
      acc += K[i * 3 + 2] * Rf[2 * 3 + j];
      s.K_otherToThis_R[i * 3 + j] = acc * sc;
    }
  for (int i = 0; i < 3; i++) s.otherToThis_t[i] = (float)otherToThis.t[i];
  for (int i = 0; i < 3; i++) {
    float acc = K[i * 3 + 0] * s.otherToThis_t[0];
    acc += K[i * 3 + 1] * s.otherToThis_t[1];
    acc += K[i * 3 + 2] * s.otherToThis_t[2];
    s.K_otherToThis_t[i] = acc;
  }
  for (int i = 0; i < 3; i++) s.thisToOther_t[i] = (float)thisToOther.t[i];
  double R2[9];
  lsdm::quatd_to_rot(thisToOther.q, R2);
  float R2f[9];
  for (int i = 0; i < 9; i++) R2f[i] = (float)R2[i] * (float)thisToOther.s;
  for (int i = 0; i < 3; i++) { s.row0[i] = R2f[i * 3 + 0]; s.row1[i] = R2f[i * 3 + 1]; s.row2[i] = R2f[i * 3 + 2]; }
  s.image = lsd_g(fr->d_image[0]);
  // (pointer AND id: a destroyed frame's heap address can be handed out again)
  s.parentIsKF = (fr->trackingParent == dm->activeKeyFrame && fr->trackingParentID == dm->activeKeyFrame->id) ? 1 : 0;
  s.wasGood = lsd_g(fr->wasGoodValid ? fr->d_wasGood : nullptr);
  s.id = fr->id;
  s.initialTrackedResidual = fr->initialTrackedResidual;
}

// the part of k_observe's arguments that does not depend on the reference frames
static void observe_args_common(lsdhip_depthmap* dm, ObserveArgs& a) {
  lsdhip_ctx* c = dm->ctx;
  lsdhip_frame* kf = dm->activeKeyFrame;
  a.m = dm->cur;
  a.kfImage = lsd_g(kf->d_image[0]);
  a.kfGrad = lsd_g(kf->d_grad[0]);
  a.kfMaxGrad = lsd_g(kf->d_maxgrad);
  a.reactivated = dm->activeKeyFrameIsReactivated ? 1 : 0;
  a.w = c->w; a.h = c->h;
  const LevelIntr& in = c->intr[0];
  a.fx = in.fx; a.fy = in.fy; a.cx = in.cx; a.cy = in.cy; a.fxi = in.fxi; a.fyi = in.fyi; a.cxi = in.cxi; a.cyi = in.cyi;
  a.minUseGrad = c->params.minUseGrad;
  a.cameraPixelNoise2 = c->params.cameraPixelNoise2;
  a.allowNegativeIdepths = c->params.allowNegativeIdepths;
  a.useSubpixelStereo = c->params.useSubpixelStereo;
  a.kfNumFramesTrackedOnThis = kf->numFramesTrackedOnThis;
  a.kfNumMappedOnThis = kf->numMappedOnThis;
  a.counters = nullptr;
  a.queue = nullptr;
  a.qcount = nullptr;
}

static int observe(lsdhip_depthmap* dm, lsdhip_frame** refs, int n) {
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  lsdhip_frame* kf = dm->activeKeyFrame;
  if (n <= 0) { lsd_set_error("updateKeyframe: empty reference deque"); return LSDHIP_E_ARG; }
  for (int i = 0; i < n; i++) {
    if (refs[i]->trackingParentID != kf->id) {
      // DepthMap.cpp:1099 needs getScaledCamToWorld() from the pose graph, which stays on the host side of the boundary
      lsd_set_error("updateKeyframe: frame %d was tracked on keyframe %d, active keyframe is %d (pose-graph path is out of scope)",
                    refs[i]->id, refs[i]->trackingParentID, kf->id);
      return LSDHIP_E_STATE;
    }
  }
  size_t nByID = 0;
  const int offset = refs[0]->id;
  ObserveArgs a;
  if (n == 1) {
    // one reference frame (blockUntilMapped): it travels in the kernel arguments
    prepare_stereo(dm, refs[0], a.one);
    nByID = 1;
    dm->d_refs = nullptr;
    dm->d_refByID = nullptr;
  } else {
    // host staging lives in one pinned block (refs | refByID); it is reused by the next call, which must not start
    // before this copy has been consumed
    size_t byIDCount = 0;
    {
      int last = offset - 1;
      for (int i = 0; i < n; i++) if (refs[i]->id > last) last = refs[i]->id;
      byIDCount = (size_t)(last - offset + 1);
    }
    const size_t refsBytes = (sizeof(StereoRef) * (size_t)n + 255) / 256 * 256;
    const size_t need = refsBytes + sizeof(int) * byIDCount;
    HIPCHK(hipStreamSynchronize(lsd_map_stream(c)));
    if (need > dm->stage_bytes) {
      if (dm->h_stage) { (void)hipHostFree(dm->h_stage); dm->h_stage = nullptr; }   // never leave a freed pointer behind
      if (dm->d_stage) { (void)hipFree(dm->d_stage); dm->d_stage = nullptr; }
      dm->stage_bytes = 0;
      const size_t want = need > 65536 ? need * 2 : 65536;
      HIPCHK(hipHostMalloc((void**)&dm->h_stage, want, hipHostMallocDefault));
      HIPCHK(hipMalloc((void**)&dm->d_stage, want));
      dm->stage_bytes = want;
    }
    StereoRef* hrefs = (StereoRef*)dm->h_stage;
    int* byID = (int*)(dm->h_stage + refsBytes);
    for (int i = 0; i < n; i++) {
      prepare_stereo(dm, refs[i], hrefs[i]);
      while ((int)nByID + offset <= refs[i]->id) byID[nByID++] = i;
    }
    HIPCHK(hipMemcpyAsync(dm->d_stage, dm->h_stage, refsBytes + sizeof(int) * nByID, hipMemcpyHostToDevice, lsd_map_stream(c)));
    dm->d_refs = (StereoRef*)dm->d_stage;
    dm->d_refByID = (int*)(dm->d_stage + refsBytes);
    memset(&a.one, 0, sizeof(a.one));
  }

  a.refs = lsd_g(dm->d_refs);
  a.refByID = lsd_g(dm->d_refByID);
  a.nByID = (int)nByID;
  a.byIDOffset = offset;
  a.nRefs = n;
  observe_args_common(dm, a);
#ifdef LSD_PHASE_TRACE
  {
    const size_t words = (size_t)((c->w + 31) / 32) * ((c->h + 7) / 8) * 16;
    if (!dm->d_obs_trace) HIPCHK(hipMalloc((void**)&dm->d_obs_trace, words * 8));
    HIPCHK(hipMemsetAsync(dm->d_obs_trace, 0, words * 8, lsd_map_stream(c)));
    a.trace = (decltype(a.trace))dm->d_obs_trace;
    dm->obs_trace_words = words;
  }
#endif
  {
#ifdef LSD_PHASE_TRACE
    constexpr int rows = 8;   // the trace buffer is laid out for 32x8 tiles
#else
    static const int rows = [] { const char* e = getenv("LSDHIP_OBS_ROWS"); int r = e ? atoi(e) : 2; return (r == 8 || r == 4) ? r : 2; }();   // 64-thread workgroups measured best (+2.5 % Mpixel/s over 256)
#endif
    const dim3 grid((c->w + 31) / 32, (c->h + rows - 1) / rows);
    const int nwaves = (int)(grid.x * grid.y) * (rows * 32 / 64);
    if (dm->countNext) {
      // sampled launch while profiling: searches and walk steps, per wave, summed behind the launch (bench.py: stereo_steps_per_s,
      // roofline_depth on the bytes of the pixels that were actually searched)
      if (!dm->d_obsCounters) {
        HIPCHK(hipMalloc((void**)&dm->d_obsCounters, (size_t)nwaves * 16 + 32));
        dm->d_obsAcc = dm->d_obsCounters + (size_t)nwaves * 2;
        HIPCHK(hipMemsetAsync(dm->d_obsAcc, 0, 32, lsd_map_stream(c)));
        dm->obsCounterWaves = nwaves;
      }
      if (dm->obsCounterWaves == nwaves) a.counters = lsd_g(dm->d_obsCounters);
    }
    if (rows == 8) {
      if (n == 1) hipLaunchKernelGGL((k_observe<true, 8>), grid, dim3(256), 0, lsd_map_stream(c), a);
      else hipLaunchKernelGGL((k_observe<false, 8>), grid, dim3(256), 0, lsd_map_stream(c), a);
    } else if (rows == 4) {
      if (n == 1) hipLaunchKernelGGL((k_observe<true, 4>), grid, dim3(128), 0, lsd_map_stream(c), a);
      else hipLaunchKernelGGL((k_observe<false, 4>), grid, dim3(128), 0, lsd_map_stream(c), a);
    } else {
      if (n == 1) hipLaunchKernelGGL((k_observe<true, 2>), grid, dim3(64), 0, lsd_map_stream(c), a);
      else hipLaunchKernelGGL((k_observe<false, 2>), grid, dim3(64), 0, lsd_map_stream(c), a);
    }
    dm->lowGradHypPossible = false;   // (the pass drops a hypothesis on a pixel below the gradient threshold: observe_front)
  }
  if (a.counters) hipLaunchKernelGGL(k_obs_count_sum, dim3(1), dim3(256), 0, lsd_map_stream(c), (const unsigned long long*)dm->d_obsCounters, dm->obsCounterWaves, dm->d_obsAcc);
  HIPCHK(hipGetLastError());
  return LSDHIP_OK;
}

// propagateDepth's arguments for the map's current keyframe -> nk (source = currentDepthMap, target = otherDepthMap)
static void prop_args(lsdhip_depthmap* dm, lsdhip_frame* nk, PropArgs& a) {
  lsdhip_ctx* c = dm->ctx;
  lsdhip_frame* old = dm->activeKeyFrame;
  size_t n = (size_t)c->w * c->h;
  a.src = dm->cur;
  a.dst = dm->oth;
  a.oldKFImage = lsd_g(old->d_image[0]);
  a.newKFImage = lsd_g(nk->d_image[0]);
  a.newKFMaxGrad = lsd_g(nk->d_maxgrad);
  a.trackingWasGood = lsd_g((nk->trackingParent == old && nk->trackingParentID == old->id && nk->wasGoodValid) ? nk->d_wasGood : nullptr);
  a.cand = lsd_g(dm->d_cand);
  a.slotCount = lsd_g(dm->d_slotCount);
  a.slots = lsd_g(dm->d_slots);
  a.flags = lsd_g(dm->d_flags);
  a.ovfHead = lsd_g(dm->d_ovfHead);
  a.ovf = lsd_g(dm->d_ovf);
  a.ovfCap = (int)n;
  a.w = c->w; a.h = c->h;
  const LevelIntr& in = c->intr[0];
  a.fx = in.fx; a.fy = in.fy; a.cx = in.cx; a.cy = in.cy; a.fxi = in.fxi; a.fyi = in.fyi; a.cxi = in.cxi; a.cyi = in.cyi;
  a.minUseGrad = c->params.minUseGrad;
  // oldToNew_SE3 = se3FromSim3(new_keyframe->pose->thisToParent_raw).inverse() (double), cast to float
  lsdm::SE3dH newToOld;
  // se3FromSim3 = SE3(sim3.quaternion(), translation): the SCALED quaternion, normalised by the SO3 constructor (so3.hpp:630-633)
  newToOld.q = nk->thisToParent_raw.q;
  { const double s_ = nk->thisToParent_raw.s; newToOld.q.w *= s_; newToOld.q.x *= s_; newToOld.q.y *= s_; newToOld.q.z *= s_; }
  lsdm::q_normalize(newToOld.q);
  for (int i = 0; i < 3; i++) newToOld.t[i] = nk->thisToParent_raw.t[i];
  lsdm::SE3dH oldToNew = lsdm::se3d_inverse(newToOld);
  double Rd[9];
  lsdm::quatd_to_rot(oldToNew.q, Rd);
  for (int i = 0; i < 9; i++) a.R[i] = (float)Rd[i];
  for (int i = 0; i < 3; i++) a.t[i] = (float)oldToNew.t[i];
}
static int propagate(lsdhip_depthmap* dm, lsdhip_frame* nk, bool checkOverflowNow) {
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  size_t n = (size_t)c->w * c->h;
  PropArgs a;
  prop_args(dm, nk, a);

  HIPCHK(hipMemsetAsync(dm->d_slotCount, 0, n * 4, lsd_map_stream(c)));
  HIPCHK(hipMemsetAsync(dm->d_ovfHead, 0xFF, n * 4, lsd_map_stream(c)));
  HIPCHK(hipMemsetAsync(dm->d_flags, 0, 64, lsd_map_stream(c)));
  dm->propClean = false;      // (the flags stay set until somebody has read them: lsdhip_depth_change_keyframe_batch then clears first)
  hipLaunchKernelGGL(k_prop_candidates, dim3((c->w + 31) / 32, (c->h + 7) / 8), dim3(256), 0, lsd_map_stream(c), a);
  hipLaunchKernelGGL(k_prop_resolve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lsd_map_stream(c), a);
  if (checkOverflowNow) {
    int flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, dm->d_flags, 4, hipMemcpyDeviceToHost, lsd_map_stream(c)));
    HIPCHK(hipStreamSynchronize(lsd_map_stream(c)));
    if (flag) {
      lsd_set_error("propagateDepth: overflow chains exhausted (more than w*h sources beyond %d per target)", PROP_SLOT_CAP);
      return LSDHIP_E_CAPACITY;
    }
  }   // otherwise the flag travels with the rescale sums of createKeyFrame (one synchronisation for the whole call)
  std::swap(dm->cur, dm->oth);
  return LSDHIP_OK;
}

struct StageTimer {
  hipEvent_t a, b;
  lsdhip_ctx* c;
  bool ok;
  StageTimer(lsdhip_ctx* c_) : c(c_) { ok = hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess; }
  ~StageTimer() { if (ok) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); } }
};
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// GPU-side duration of a mapping call: an event pair on the stream, read back lazily (kind 0 update, 1 create, 2 finalize)
static void timing_collect(lsdhip_depthmap* dm, bool all) {
  for (int i = 0; i < 8; i++) {
    if (!dm->ev_pending[i]) continue;
    if (!all && hipEventQuery(dm->ev[i][1]) != hipSuccess) continue;
    float ms = 0;
    if (hipEventElapsedTime(&ms, dm->ev[i][0], dm->ev[i][1]) == hipSuccess) {
      dm->gpu_ms[dm->ev_kind[i]] += ms;
      dm->gpu_calls[dm->ev_kind[i]]++;
    }
    dm->ev_pending[i] = false;
  }
}
// GPU time of the DepthMap calls, SAMPLED: an event record is a barrier packet with a completion signal, and the kernel behind it
// starts ~10 us late (rocprofv3 timeline, profiles/r03_notes.md §2b) — two of them around every updateKeyframe cost 9 % of the
// track + map loop.  Every 7th updateKeyframe and every 2nd createKeyFrame / finalizeKeyFrame is bracketed (kind 3, k_observe
// alone, is sampled by its caller); gpu_ms / gpu_calls describe the bracketed calls only.  Returns EV_SKIP when the call is not sampled.
// (7, not 8: a loop with a keyframe every 10 frames runs exactly 8 (pipelined) or 9 updateKeyframe calls per keyframe, and the first
// update on a new keyframe searches 30x more pixels than the eighth — a period of 8 sampled one phase of that cycle only.)
constexpr int EV_SKIP = -1000;
static int timing_begin(lsdhip_depthmap* dm, int kind) {
  if (kind < 3) {
    static const bool every = getenv("LSDHIP_DEPTH_EVENTS_ALL") != nullptr;   // developer switch: A/B of the sampling itself
    const unsigned period = every ? 1u : (kind == 0 ? 7u : 2u);
    if ((dm->ev_tick[kind]++ % period) != 0) return EV_SKIP;
  }
  const int i = dm->ev_next;
  dm->ev_next = (dm->ev_next + 1) % 8;
  if (dm->ev_pending[i]) {
    HIPCHK(hipEventSynchronize(dm->ev[i][1]));
    timing_collect(dm, false);
  }
  if (!dm->ev[i][0]) { HIPCHK(hipEventCreate(&dm->ev[i][0])); HIPCHK(hipEventCreate(&dm->ev[i][1])); }
  dm->ev_kind[i] = kind;
  HIPCHK(hipEventRecord(dm->ev[i][0], lsd_map_stream(dm->ctx)));
  return i;
}
static int timing_end(lsdhip_depthmap* dm, int i) {
  if (i == EV_SKIP) return LSDHIP_OK;
  HIPCHK(hipEventRecord(dm->ev[i][1], lsd_map_stream(dm->ctx)));
  dm->ev_pending[i] = true;
  return LSDHIP_OK;
}

extern "C" int lsdhip_depth_init_gt(lsdhip_depthmap* dm, lsdhip_frame* kf) {
  if (!dm || !kf) return LSDHIP_E_ARG;
  if (!kf->hasIDepth) { lsd_set_error("initializeFromGTDepth: frame has no depth"); return LSDHIP_E_STATE; }
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  if (int rcl = lsd_frame_require_level0(kf)) return rcl;   // a DepthMap's keyframe is asked for gradients(0) / maxGradients(0)
  dm->lowGradHypPossible = true;
  dm->activeKeyFrame = kf;
  dm->activeKeyFrameIsReactivated = false;
  int n = c->w * c->h;
  if (int rcb = lsd_m_begin(c)) return rcb;
  hipLaunchKernelGGL(k_init_gt, dim3((n + 255) / 256), dim3(256), 0, lsd_map_stream(c), dm->cur, kf->d_idepth[0], n);
  int rc = set_depth(dm);
  if (rc == LSDHIP_OK && lsd_m_record(c) < 0) rc = LSDHIP_E_HIP;
  return rc;
}

extern "C" int lsdhip_depth_init_random(lsdhip_depthmap* dm, lsdhip_frame* kf) {
  if (!dm || !kf) return LSDHIP_E_ARG;
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  if (int rcl = lsd_frame_require_level0(kf)) return rcl;
  dm->lowGradHypPossible = true;
  dm->activeKeyFrame = kf;
  dm->activeKeyFrameIsReactivated = false;
  // initializeRandomly (DepthMap.cpp:883-916) draws from the C library's rand() in pixel order: a host-side init path.
  int w = c->w, h = c->h;
  size_t n = (size_t)w * h;
  std::vector<float> mg(n);
  if (int rcb = lsd_m_begin(c)) return rcb;
  HIPCHK(hipMemcpyAsync(mg.data(), kf->d_maxgrad, n * 4, hipMemcpyDeviceToHost, lsd_map_stream(c)));
  std::vector<lsdhip_hypothesis> hyp(n);
  HIPCHK(hipStreamSynchronize(lsd_map_stream(c)));
  int rc = lsdhip_depth_download(dm, hyp.data());
  if (rc) return rc;
  for (int y = 1; y < h - 1; y++)
    for (int x = 1; x < w - 1; x++) {
      lsdhip_hypothesis& p = hyp[x + y * w];
      if (mg[x + y * w] > c->params.minUseGrad) {
        float idepth = 0.5f + 1.0f * ((rand() % 100001) / 100000.0f);
        p.isValid = 1; p.blacklisted = 0; p.nextStereoFrameMinID = 0; p.validity_counter = 20;
        p.idepth = idepth; p.idepth_smoothed = idepth; p.idepth_var = VAR_RANDOM_INIT_INITIAL; p.idepth_var_smoothed = VAR_RANDOM_INIT_INITIAL;
      } else {
        p.isValid = 0; p.blacklisted = 0;
      }
    }
  rc = lsdhip_depth_upload(dm, kf, hyp.data(), 0);
  if (rc) return rc;
  rc = set_depth(dm);
  if (rc == LSDHIP_OK && lsd_m_record(c) < 0) rc = LSDHIP_E_HIP;
  return rc;
}

extern "C" int lsdhip_depth_set_from_existing(lsdhip_depthmap* dm, lsdhip_frame* kf) {
  if (!dm || !kf) return LSDHIP_E_ARG;
  if (!kf->reActValid) { lsd_set_error("setFromExistingKF: frame %d has no re-activation data", kf->id); return LSDHIP_E_STATE; }
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  if (int rcl = lsd_frame_require_level0(kf)) return rcl;
  dm->lowGradHypPossible = true;
  dm->activeKeyFrame = kf;
  kf->numMappedOnThis = 0;
  kf->numFramesTrackedOnThis = 0;
  dm->activeKeyFrameIsReactivated = true;
  int n = c->w * c->h;
  if (int rcb = lsd_m_begin(c)) return rcb;
  hipLaunchKernelGGL(k_from_react, dim3((n + 255) / 256), dim3(256), 0, lsd_map_stream(c), dm->cur, kf->d_idepth_reAct, kf->d_idepthVar_reAct,
                     kf->d_validity_reAct, n);
  int rc = regularize(dm, false, VAL_SUM_MIN_FOR_KEEP);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(lsd_map_stream(c)));
  return LSDHIP_OK;
}

static float ema(float old, double sample) { return (float)(0.9 * old + 0.1 * sample); }

extern "C" int lsdhip_depth_update(lsdhip_depthmap* dm, lsdhip_frame** refs, int n) {
  if (!dm || !refs || n <= 0) return LSDHIP_E_ARG;
  if (!dm->activeKeyFrame) { lsd_set_error("updateKeyframe: depth map has no active keyframe"); return LSDHIP_E_STATE; }
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  double t0 = now_ms();
  if (int rcb = lsd_m_begin(c)) return rcb;
  if (int rcg = lsd_gate_wait(c)) return rcg;
  const int ev = timing_begin(dm, 0);
  if (ev < 0 && ev != EV_SKIP) return ev;
  // while profiling, every 7th call brackets the observe kernel alone with a second event pair (bench.py roofline_depth)
  const bool sampleObs = c->prof_on && ((dm->obs_tick++ % 7) == 0);
  int evo = -1;
  if (sampleObs) { evo = timing_begin(dm, 3); if (evo < 0) return evo; }
  lsdhip_host_mark(10);
  dm->countNext = sampleObs;
  int rc = observe(dm, refs, n);
  dm->countNext = false;
  if (rc) return rc;
  lsdhip_host_mark(11);
  if (sampleObs) { rc = timing_end(dm, evo); if (rc) return rc; }
  lsdhip_frame* kf = dm->activeKeyFrame;
  const bool setDepth = !kf->depthHasBeenUpdatedFlag;
  rc = fill_regularize(dm, VAL_SUM_MIN_FOR_KEEP, setDepth);
  if (rc) return rc;
  lsdhip_host_mark(12);
  rc = timing_end(dm, ev);
  if (rc) return rc;
#ifdef LSD_DEVTOOLS
  {
    const size_t npx = (size_t)c->w * c->h;
    const int fid = refs[n - 1]->id;
    lsd_trace_sum(c, lsd_map_stream(c), 30, fid, refs[n - 1]->d_wasGood, (size_t)c->wl[1] * c->hl[1]);
    lsd_trace_sum(c, lsd_map_stream(c), 31, fid, dm->cur.valid, npx);
    lsd_trace_sum(c, lsd_map_stream(c), 32, fid, dm->cur.idepth, npx * 4);
    lsd_trace_sum(c, lsd_map_stream(c), 33, fid, dm->cur.var, npx * 4);
    lsd_trace_sum(c, lsd_map_stream(c), 34, fid, dm->cur.idepth_s, npx * 4);
    lsd_trace_sum(c, lsd_map_stream(c), 35, fid, dm->cur.validity, npx * 4);
    lsd_trace_sum(c, lsd_map_stream(c), 36, fid, dm->cur.blacklisted, npx * 4);
    lsd_trace_sum(c, lsd_map_stream(c), 37, fid, lsd_depth_latest(kf)[0], npx * 4);
    lsd_trace_sum(c, lsd_map_stream(c), 38, fid, lsd_depth_latest(kf)[1], npx);
    lsd_trace_sum(c, lsd_map_stream(c), 39, fid, lsd_depthvar_latest(kf)[1], npx);
  }
#endif
  if (lsd_m_record(c) < 0) return LSDHIP_E_HIP;
  if (!c->async) HIPCHK(hipStreamSynchronize(lsd_map_stream(c)));
  dm->msUpdate = ema(dm->msUpdate, now_ms() - t0);   // host wall time of the call (enqueue time only in async mode)
  kf->numMappedOnThis++;
  kf->numMappedOnThisTotal++;
  return LSDHIP_OK;
}

// DepthMap::updateKeyframe for the maps of n sequences, one tracked frame each (the blockUntilMapped deque), in three launches for
// all of them: k_observe_batch, k_reg_fused_batch (+ its non-setDepth twin if some keyframe is not due), k_idepth_pyramid_batch.
// Per map the arithmetic, the launch geometry inside its z-slice and therefore every plane are those of lsdhip_depth_update.
extern "C" int lsdhip_depth_update_batch(int n, lsdhip_depthmap** maps, lsdhip_frame** refs) {
  if (n <= 0 || !maps || !refs) return LSDHIP_E_ARG;
  for (int j = 0; j < n; j++) if (!maps[j] || !refs[j]) return LSDHIP_E_ARG;
  lsdhip_ctx* c = maps[0]->ctx;
  LSD_CTX_LOCK(c);
  for (int j = 0; j < n; j++) {
    lsdhip_depthmap* dm = maps[j];
    if (dm->ctx != c || refs[j]->ctx != c) { lsd_set_error("updateKeyframe batch: maps and frames of one batch live on one context"); return LSDHIP_E_ARG; }
    if (!dm->activeKeyFrame) { lsd_set_error("updateKeyframe batch: depth map %d has no active keyframe", j); return LSDHIP_E_STATE; }
    for (int i = 0; i < j; i++) if (maps[i] == dm) { lsd_set_error("updateKeyframe batch: depth map %d appears twice", j); return LSDHIP_E_ARG; }
    if (refs[j]->trackingParentID != dm->activeKeyFrame->id) {
      lsd_set_error("updateKeyframe: frame %d was tracked on keyframe %d, active keyframe is %d (pose-graph path is out of scope)",
                    refs[j]->id, refs[j]->trackingParentID, dm->activeKeyFrame->id);
      return LSDHIP_E_STATE;
    }
  }
  HIPCHK(hipSetDevice(c->device));
  double t0 = now_ms();
  if (int rcb = lsd_m_begin(c)) return rcb;
  const hipStream_t ms = lsd_map_stream(c);
  // ---- observe ---------------------------------------------------------------------------------------------------------------------
  const size_t obsBytes = (sizeof(ObserveArgs) * (size_t)n + 255) / 256 * 256;
  const size_t regBytes = (sizeof(RegBatchItem) * (size_t)n + 255) / 256 * 256;
  static const bool splitOff = getenv("LSDHIP_OBS_SPLIT") && getenv("LSDHIP_OBS_SPLIT")[0] == '0';   // A/B of round 5
  const bool split = !splitOff && n >= LSD_OBS_SPLIT_MIN_MAPS && n <= LSD_OBS_WALK_MAX_MAPS;
  if (split)
    for (int j = 0; j < n; j++)
      if (!maps[j]->d_obsQueue) HIPCHK(hipMalloc((void**)&maps[j]->d_obsQueue, (size_t)c->w * c->h * sizeof(float4)));
  void* hostBlob = nullptr;
  void* dev = nullptr;
  int rc = lsd_args_begin(c, obsBytes + regBytes + sizeof(int) * (size_t)n, &hostBlob, &dev);
  if (rc) return rc;
  ObserveArgs* oa = (ObserveArgs*)hostBlob;
  RegBatchItem* ra = (RegBatchItem*)((uint8_t*)hostBlob + obsBytes);
  int* qcountHost = (int*)((uint8_t*)hostBlob + obsBytes + regBytes);       // the queues' fill counts start from the zeros copied here
  int* qcountDev = (int*)((uint8_t*)dev + obsBytes + regBytes);
  int nSet = 0;
  static const bool candOff = getenv("LSDHIP_OBS_CAND") && getenv("LSDHIP_OBS_CAND")[0] == '0';   // developer A/B (round 6)
  bool candidates = !candOff;     // the select pass walks the keyframes' gradient candidates: every map of the call must qualify
  for (int j = 0; j < n; j++) {
    lsdhip_depthmap* dm = maps[j];
    lsdhip_frame* kf = dm->activeKeyFrame;
    ObserveArgs& a = oa[j];
    memset((void*)&a, 0, sizeof(a));
    prepare_stereo(dm, refs[j], a.one);
    a.refs = nullptr; a.refByID = nullptr; a.nByID = 1; a.byIDOffset = refs[j]->id; a.nRefs = 1;
    observe_args_common(dm, a);
#ifdef LSD_PHASE_TRACE
    a.trace = nullptr;
#endif
    a.queue = lsd_g(split ? dm->d_obsQueue : nullptr);
    a.qcount = lsd_g(qcountDev + j);
    a.gradCand = lsd_g((const uint16_t*)kf->d_gradCand);
    if (dm->lowGradHypPossible || kf->gradCandTh != a.minUseGrad) candidates = false;
    qcountHost[j] = 0;
    dm->d_refs = nullptr;
    dm->d_refByID = nullptr;
    // the regulariser pass that follows: reads cur.valid, writes the spare plane (swapped below, as fill_regularize does)
    RegBatchItem& r = ra[j];
    r.a = reg_args(dm, VAL_SUM_MIN_FOR_KEEP);
    const bool setDepth = !kf->depthHasBeenUpdatedFlag;
    r.id0 = lsd_g(setDepth ? lsd_depth_w(kf)[0] : nullptr);
    r.var0 = lsd_g(setDepth ? lsd_depthvar_w(kf)[0] : nullptr);
    r.partials = lsd_g(dm->d_red + 16);
    if (setDepth) nSet++;
  }
  rc = lsd_args_commit(c, ms);
  if (rc) return rc;
  const double mapPixels = (double)n * c->w * c->h;
  int bp = lsd_bprof_begin(c, 1, ms);
  if (bp < -1) return bp;
  if (split) {
    static const int walkWaves = getenv("LSDHIP_OBS_WALK_WAVES") ? atoi(getenv("LSDHIP_OBS_WALK_WAVES")) : LSD_OBS_WALK_WAVES;   // developer sweep
    unsigned long long* acc = nullptr;
    if (bp >= 0) {     // a sampled call also counts its searches and walk steps
      if (!c->d_obsBatchAcc) { HIPCHK(hipMalloc((void**)&c->d_obsBatchAcc, 66 * 8)); HIPCHK(hipMemsetAsync(c->d_obsBatchAcc, 0, 66 * 8, ms)); }
      acc = c->d_obsBatchAcc;
    }
    if (candidates) hipLaunchKernelGGL(k_observe_select_cand_batch, dim3((lsd_gradcand_groups(c->w * c->h) + 3) / 4, 1, n), dim3(256), 0, ms, (const ObserveArgs*)dev);
    else hipLaunchKernelGGL(k_observe_select_batch, dim3((c->w + 63) / 64, (c->h + 15) / 16, n), dim3(256), 0, ms, (const ObserveArgs*)dev);
    hipLaunchKernelGGL(k_observe_walk_batch, dim3(walkWaves), dim3(64), 0, ms, (const ObserveArgs*)dev, (const int*)qcountDev, n, acc);
  } else {
    hipLaunchKernelGGL((k_observe_batch<2>), dim3((c->w + 31) / 32, (c->h + 1) / 2, n), dim3(64), 0, ms, (const ObserveArgs*)dev);
  }
  rc = lsd_bprof_end(c, bp, ms, mapPixels);
  if (rc) return rc;
  const RegBatchItem* dra = (const RegBatchItem*)((const uint8_t*)dev + obsBytes);
  const int rpy = lsd_reg_py((long long)((c->w + 31) / 32) * ((c->h + 7) / 8) * n);
  const dim3 rgrid((c->w + 31) / 32, lsd_reg_grid_rows((c->h + 7) / 8, rpy), n);
  bp = lsd_bprof_begin(c, 2, ms);
  if (bp < -1) return bp;
  LSD_REG_DISPATCH(rpy,
    if (nSet > 0) hipLaunchKernelGGL((k_reg_fused_batch<true, PYV>), rgrid, dim3(256), 0, ms, dra);
    if (nSet < n) hipLaunchKernelGGL((k_reg_fused_batch<false, PYV>), rgrid, dim3(256), 0, ms, dra));
  rc = lsd_bprof_end(c, bp, ms, mapPixels);
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  rc = lsd_args_release(c, dev, ms);
  if (rc) return rc;
  for (int j = 0; j < n; j++) { swap_valid(maps[j]); maps[j]->lowGradHypPossible = false; }   // (every form of the observe pass drops a hypothesis below the threshold)
  // ---- Frame::setDepth's second half for the keyframes that were due -------------------------------------------------------------------
  if (nSet > 0) {
    std::vector<lsdhip_frame*> kfs;
    std::vector<const double*> parts;
    std::vector<double*> outs;
    for (int j = 0; j < n; j++) {
      lsdhip_frame* kf = maps[j]->activeKeyFrame;
      if (kf->depthHasBeenUpdatedFlag) continue;
      const int slot = lsd_ctx_take_slot(c);
      if (slot < 0) return slot;
      if (kf->pendStats >= 0) c->slot_stats_owner[kf->pendStats] = nullptr;
      kf->pendStats = slot;
      c->slot_stats_owner[slot] = kf;
      kfs.push_back(kf);
      parts.push_back(maps[j]->d_red + 16);
      outs.push_back((double*)&c->h_slots[slot]);
    }
    bp = lsd_bprof_begin(c, 3, ms);
    if (bp < -1) return bp;
    rc = lsd_frame_build_idepth_pyramid_batch(kfs.data(), (int)kfs.size(), parts.data(), (int)(rgrid.x * rgrid.y), outs.data());
    if (rc) return rc;
    rc = lsd_bprof_end(c, bp, ms, (double)kfs.size() * c->w * c->h);
    if (rc) return rc;
    for (lsdhip_frame* kf : kfs) kf->depthHasBeenUpdatedFlag = true;
  }
  if (lsd_m_record(c) < 0) return LSDHIP_E_HIP;
  if (!c->async) {
    HIPCHK(hipStreamSynchronize(ms));
    for (int j = 0; j < n; j++) { rc = lsd_frame_resolve(maps[j]->activeKeyFrame); if (rc) return rc; }
  }
  const double dt = now_ms() - t0;
  for (int j = 0; j < n; j++) {
    lsdhip_depthmap* dm = maps[j];
    dm->msUpdate = ema(dm->msUpdate, dt / n);
    dm->activeKeyFrame->numMappedOnThis++;
    dm->activeKeyFrame->numMappedOnThisTotal++;
  }
  return LSDHIP_OK;
}

extern "C" int lsdhip_depth_create_keyframe(lsdhip_depthmap* dm, lsdhip_frame* nk, float* rescale_out) {
  if (!dm || !nk) return LSDHIP_E_ARG;
  if (!dm->activeKeyFrame) { lsd_set_error("createKeyFrame: depth map has no active keyframe"); return LSDHIP_E_STATE; }
  if (!nk->trackingParent) { lsd_set_error("createKeyFrame: new keyframe has no tracking parent"); return LSDHIP_E_STATE; }
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  double t0 = now_ms();
  lsdm::SE3dH newToOld;
  // se3FromSim3 = SE3(sim3.quaternion(), translation): the SCALED quaternion, normalised by the SO3 constructor (so3.hpp:630-633)
  newToOld.q = nk->thisToParent_raw.q;
  { const double s_ = nk->thisToParent_raw.s; newToOld.q.w *= s_; newToOld.q.x *= s_; newToOld.q.y *= s_; newToOld.q.z *= s_; }
  lsdm::q_normalize(newToOld.q);
  for (int i = 0; i < 3; i++) newToOld.t[i] = nk->thisToParent_raw.t[i];
  lsdm::SE3dH oldToNew = lsdm::se3d_inverse(newToOld);

  if (int rcb = lsd_m_begin(c)) return rcb;
  const int ev = timing_begin(dm, 1);
  if (ev < 0 && ev != EV_SKIP) return ev;
  if (int rcl = lsd_frame_require_level0(nk)) return rcl;   // the new keyframe's maxGradients(0): propagateDepth reads it (DepthMap.cpp:758)
  int rc = propagate(dm, nk, false);
  if (rc) return rc;
  dm->msPropagate = ema(dm->msPropagate, now_ms() - t0);
  dm->activeKeyFrame = nk;
  dm->activeKeyFrameIsReactivated = false;
  rc = regularize(dm, true, VAL_SUM_MIN_FOR_KEEP);
  if (rc) return rc;
  rc = fill_regularize(dm, VAL_SUM_MIN_FOR_KEEP, false);
  if (rc) return rc;

  // make mean inverse depth be one (DepthMap.cpp:1285-1304): the factor is formed on the device (k_rescale), the host
  // copy of (sum, count) and the propagation overflow flag are a deferred result of the new keyframe
  int n = c->w * c->h;
  int nb = (n + 255) / 256;
  const int slot = lsd_ctx_take_slot(c);
  if (slot < 0) return slot;
  hipLaunchKernelGGL(k_sum_valid_idepth, dim3(nb), dim3(256), 0, lsd_map_stream(c), dm->cur, n, dm->d_red + 16);
  hipLaunchKernelGGL(k_reduce_pairs, dim3(1), dim3(256), 0, lsd_map_stream(c), dm->d_red + 16, nb, (double*)&c->h_slots[slot], dm->d_red,
                     (const int*)dm->d_flags);
  hipLaunchKernelGGL(k_rescale, dim3(nb), dim3(256), 0, lsd_map_stream(c), dm->cur, n, dm->d_red);
  lsdm::SE3dH back = lsdm::se3d_inverse(oldToNew);
  nk->thisToParent_raw.q = back.q;
  lsdm::q_normalize(nk->thisToParent_raw.q);   // sim3FromSE3 -> Sim3::setScale normalises the quaternion (rxso3.hpp:332-335)
  for (int i = 0; i < 3; i++) nk->thisToParent_raw.t[i] = back.t[i];
  if (nk->pendRescale >= 0) c->slot_rescale_owner[nk->pendRescale] = nullptr;
  nk->pendRescale = slot;
  c->slot_rescale_owner[slot] = nk;
  rc = set_depth(dm);
  if (rc) return rc;
  rc = timing_end(dm, ev);
  if (rc) return rc;
  if (lsd_m_record(c) < 0) return LSDHIP_E_HIP;
  if (!c->async || rescale_out) {
    rc = lsd_frame_resolve(nk);
    if (rc) return rc;
    if (rescale_out) *rescale_out = (float)nk->thisToParent_raw.s;
  }
  dm->msCreate = ema(dm->msCreate, now_ms() - t0);
  return LSDHIP_OK;
}

// DepthMap::finalizeKeyFrame (DepthMap.cpp:1363-1395) on the current keyframes of n maps followed by DepthMap::createKeyFrame(new_keyframes[j])
// (:1222-1327), in six launches shared by all maps (kernels above).  Per map every plane, the re-activation data, both keyframes' idepth
// pyramids and the deferred results (mean inverse depth / point counts, rescale factor, overflow flag) are those of
// lsdhip_depth_finalize + lsdhip_depth_create_keyframe (tests/test_multiseq_gpu.py).
extern "C" int lsdhip_depth_change_keyframe_batch(int n, lsdhip_depthmap** maps, lsdhip_frame** new_keyframes, float* rescale_out) {
  if (n <= 0 || !maps || !new_keyframes) return LSDHIP_E_ARG;
  for (int j = 0; j < n; j++) if (!maps[j] || !new_keyframes[j]) return LSDHIP_E_ARG;
  lsdhip_ctx* c = maps[0]->ctx;
  LSD_CTX_LOCK(c);
  for (int j = 0; j < n; j++) {
    lsdhip_depthmap* dm = maps[j];
    if (dm->ctx != c || new_keyframes[j]->ctx != c) { lsd_set_error("keyframe change batch: maps and frames of one batch live on one context"); return LSDHIP_E_ARG; }
    if (!dm->activeKeyFrame) { lsd_set_error("keyframe change batch: depth map %d has no active keyframe", j); return LSDHIP_E_STATE; }
    if (!new_keyframes[j]->trackingParent) { lsd_set_error("createKeyFrame: new keyframe has no tracking parent"); return LSDHIP_E_STATE; }
    if (new_keyframes[j] == dm->activeKeyFrame) { lsd_set_error("keyframe change batch: map %d: the new keyframe is the current one", j); return LSDHIP_E_ARG; }
    for (int i = 0; i < j; i++)
      if (maps[i] == dm || new_keyframes[i] == new_keyframes[j]) { lsd_set_error("keyframe change batch: entry %d appears twice", j); return LSDHIP_E_ARG; }
  }
  HIPCHK(hipSetDevice(c->device));
  const double t0 = now_ms();
  if (int rcb = lsd_m_begin(c)) return rcb;
  const hipStream_t ms = lsd_map_stream(c);
  const int npx = c->w * c->h;
  const int tpy = lsd_reg_py((long long)((c->w + 31) / 32) * ((c->h + 7) / 8) * n);
  const dim3 tgrid((c->w + 31) / 32, lsd_reg_grid_rows((c->h + 7) / 8, tpy), n);
  const int ntiles = (int)(tgrid.x * tgrid.y);
  const int nwg5 = (npx + LSD_RESCALE_PX - 1) / LSD_RESCALE_PX;
  // Everything below edits host-side state of the maps, of both keyframes and of the context's slot ring while it builds the launches'
  // arguments, and several steps can fail (a slot, the argument ring, a launch).  The edits are undone unless the call gets as far as its
  // last launch: an error leaves every map on its old keyframe with its planes where they were, not half-way.
  struct Txn {
    lsdhip_ctx* c; int n; lsdhip_depthmap** maps; lsdhip_frame** nks;
    struct M { HypPlanes cur, oth; uint8_t* snap; lsdhip_frame* kf; bool react; bool propClean; int oldStats; int nkRescale, nkStats; lsdm::Sim3dH nkPose; };
    std::vector<M> m;
    std::vector<int> slots;      // slots taken by this call
    int slotNext; int ev = EV_SKIP; bool timing = false; bool done = false;
    Txn(lsdhip_ctx* c_, int n_, lsdhip_depthmap** maps_, lsdhip_frame** nks_) : c(c_), n(n_), maps(maps_), nks(nks_), m((size_t)n_), slotNext(c_->slot_next) {
      for (int j = 0; j < n; j++) {
        lsdhip_depthmap* dm = maps[j];
        m[j] = M{dm->cur, dm->oth, dm->d_validSnap, dm->activeKeyFrame, dm->activeKeyFrameIsReactivated, dm->propClean, dm->activeKeyFrame->pendStats,
                 nks[j]->pendRescale, nks[j]->pendStats, nks[j]->thisToParent_raw};
      }
    }
    ~Txn() {
      if (done) return;
      for (int s : slots) { c->slot_stats_owner[s] = nullptr; c->slot_rescale_owner[s] = nullptr; }
      c->slot_next = slotNext;
      for (int j = 0; j < n; j++) {
        lsdhip_depthmap* dm = maps[j];
        dm->cur = m[j].cur; dm->oth = m[j].oth; dm->d_validSnap = m[j].snap; dm->activeKeyFrame = m[j].kf; dm->activeKeyFrameIsReactivated = m[j].react;
        dm->propClean = false;                       // (the K7 scratch may have been written: the next call clears it)
        m[j].kf->pendStats = m[j].oldStats;
        if (m[j].oldStats >= 0) c->slot_stats_owner[m[j].oldStats] = m[j].kf;
        nks[j]->pendRescale = m[j].nkRescale; nks[j]->pendStats = m[j].nkStats;
        if (m[j].nkRescale >= 0) c->slot_rescale_owner[m[j].nkRescale] = nks[j];
        if (m[j].nkStats >= 0) c->slot_stats_owner[m[j].nkStats] = nks[j];
        nks[j]->thisToParent_raw = m[j].nkPose;
      }
      if (timing) (void)timing_end(maps[0], ev);     // (keeps the event pair of the timing sample balanced)
    }
  } txn(c, n, maps, new_keyframes);
  for (int j = 0; j < n; j++) {
    lsdhip_depthmap* dm = maps[j];
    if (!dm->propClean) {     // (a single-call propagation, or a failed call, left the K7 scratch in use)
      HIPCHK(hipMemsetAsync(dm->d_slotCount, 0, (size_t)npx * 4, ms));
      HIPCHK(hipMemsetAsync(dm->d_ovfHead, 0xFF, (size_t)npx * 4, ms));
      HIPCHK(hipMemsetAsync(dm->d_flags, 0, 64, ms));
    }
    dm->propClean = false;
  }
  const int ev = timing_begin(maps[0], 1);
  if (ev < 0 && ev != EV_SKIP) return ev;
  txn.ev = ev; txn.timing = true;
  if (int rcl = lsd_frames_require_level0(new_keyframes, n)) return rcl;   // the new keyframes' gradients(0) / maxGradients(0), one launch pair
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t kfBytes = up(sizeof(KfItem) * (size_t)n), regBytes = up(sizeof(RegBatchItem) * (size_t)n), resBytes = up(sizeof(RescaleItem) * (size_t)n);
  void* hostBlob = nullptr;
  void* dev = nullptr;
  int rc = lsd_args_begin(c, kfBytes + 2 * regBytes + resBytes, &hostBlob, &dev);
  if (rc) return rc;
  KfItem* kf = (KfItem*)hostBlob;
  RegBatchItem* regOcc = (RegBatchItem*)((uint8_t*)hostBlob + kfBytes);
  RegBatchItem* regFill = (RegBatchItem*)((uint8_t*)hostBlob + kfBytes + regBytes);
  RescaleItem* res = (RescaleItem*)((uint8_t*)hostBlob + kfBytes + 2 * regBytes);
  std::vector<lsdhip_frame*> pyrFrames((size_t)2 * n);
  std::vector<const double*> pyrParts((size_t)2 * n);
  std::vector<double*> pyrOuts((size_t)2 * n);
  std::vector<int> pyrN((size_t)2 * n);
  for (int j = 0; j < n; j++) {
    lsdhip_depthmap* dm = maps[j];
    lsdhip_frame* old = dm->activeKeyFrame;
    lsdhip_frame* nk = new_keyframes[j];
    double* setA = dm->d_red + 16;
    double* setB = setA + dm->redStride;
    double* setC = setB + dm->redStride;
    // finalizeKeyFrame: the pass on the old keyframe's map, its Frame::setDepth (level 0 + partials), re-activation data; phase A of
    // propagateDepth into otherDepthMap
    KfItem& k = kf[j];
    memset((void*)&k, 0, sizeof(k));
    k.a = reg_args(dm, VAL_SUM_MIN_FOR_KEEP);
    k.id0 = lsd_g(lsd_depth_w(old)[0]); k.var0 = lsd_g(lsd_depthvar_w(old)[0]); k.partials = lsd_g(setA);
    k.x.reactId = lsd_g(old->d_idepth_reAct); k.x.reactVar = lsd_g(old->d_idepthVar_reAct); k.x.reactVal = lsd_g(old->d_validity_reAct);
    prop_args(dm, nk, k.x.prop);
    swap_valid(dm);
    {
      const int slot = lsd_ctx_take_slot(c);
      if (slot < 0) return slot;
      txn.slots.push_back(slot);
      if (old->pendStats >= 0) c->slot_stats_owner[old->pendStats] = nullptr;
      old->pendStats = slot;
      c->slot_stats_owner[slot] = old;
      pyrFrames[j] = old; pyrParts[j] = setA; pyrN[j] = ntiles; pyrOuts[j] = (double*)&c->h_slots[slot];
    }
    // createKeyFrame: the new keyframe's map becomes the current one
    std::swap(dm->cur, dm->oth);
    dm->activeKeyFrame = nk;
    dm->activeKeyFrameIsReactivated = false;
    memset((void*)&regOcc[j], 0, sizeof(RegBatchItem));
    regOcc[j].a = reg_args(dm, VAL_SUM_MIN_FOR_KEEP);
    swap_valid(dm);
    memset((void*)&regFill[j], 0, sizeof(RegBatchItem));
    regFill[j].a = reg_args(dm, VAL_SUM_MIN_FOR_KEEP);
    regFill[j].partials = lsd_g(setB);
    swap_valid(dm);
    RescaleItem& r = res[j];
    memset((void*)&r, 0, sizeof(r));
    r.m = dm->cur;
    r.sumPartials = lsd_g(setB); r.nSumPartials = ntiles;
    r.flags = lsd_g(dm->d_flags);
    {
      const int slot = lsd_ctx_take_slot(c);
      if (slot < 0) return slot;
      txn.slots.push_back(slot);
      if (nk->pendRescale >= 0) c->slot_rescale_owner[nk->pendRescale] = nullptr;
      nk->pendRescale = slot;
      c->slot_rescale_owner[slot] = nk;
      r.slot = lsd_g((double*)&c->h_slots[slot]);
    }
    r.id0 = lsd_g(lsd_depth_w(nk)[0]); r.var0 = lsd_g(lsd_depthvar_w(nk)[0]);
    r.statPartials = lsd_g(setC);
    r.n = npx;
    {
      const int slot = lsd_ctx_take_slot(c);
      if (slot < 0) return slot;
      txn.slots.push_back(slot);
      if (nk->pendStats >= 0) c->slot_stats_owner[nk->pendStats] = nullptr;
      nk->pendStats = slot;
      c->slot_stats_owner[slot] = nk;
      pyrFrames[n + j] = nk; pyrParts[n + j] = setC; pyrN[n + j] = nwg5; pyrOuts[n + j] = (double*)&c->h_slots[slot];
    }
    // the new keyframe's pose: sim3FromSE3(se3FromSim3(thisToParent_raw)) — the scale follows with the rescale factor (DepthMap.cpp:1305)
    {
      lsdm::SE3dH newToOld;
      newToOld.q = nk->thisToParent_raw.q;
      { const double s_ = nk->thisToParent_raw.s; newToOld.q.w *= s_; newToOld.q.x *= s_; newToOld.q.y *= s_; newToOld.q.z *= s_; }
      lsdm::q_normalize(newToOld.q);
      for (int i = 0; i < 3; i++) newToOld.t[i] = nk->thisToParent_raw.t[i];
      const lsdm::SE3dH back = lsdm::se3d_inverse(lsdm::se3d_inverse(newToOld));
      nk->thisToParent_raw.q = back.q;
      lsdm::q_normalize(nk->thisToParent_raw.q);
      for (int i = 0; i < 3; i++) nk->thisToParent_raw.t[i] = back.t[i];
    }
  }
  rc = lsd_args_commit(c, ms);
  if (rc) return rc;
  const int bp = lsd_bprof_begin(c, 4, ms);
  if (bp < -1) return bp;
  const KfItem* dkf = (const KfItem*)dev;
  const RegBatchItem* dOcc = (const RegBatchItem*)((const uint8_t*)dev + kfBytes);
  const RegBatchItem* dFill = (const RegBatchItem*)((const uint8_t*)dev + kfBytes + regBytes);
  const RescaleItem* dRes = (const RescaleItem*)((const uint8_t*)dev + kfBytes + 2 * regBytes);
  LSD_REG_DISPATCH(tpy, hipLaunchKernelGGL(k_kf_finalize_prop<PYV>, tgrid, dim3(256), 0, ms, dkf));
  hipLaunchKernelGGL(k_prop_resolve_batch, dim3((unsigned)((npx + 255) / 256), n), dim3(256), 0, ms, dkf);
  LSD_REG_DISPATCH(tpy, hipLaunchKernelGGL((k_kf_reg<false, true, PYV>), tgrid, dim3(256), 0, ms, dOcc));
  LSD_REG_DISPATCH(tpy, hipLaunchKernelGGL((k_kf_reg<true, false, PYV>), tgrid, dim3(256), 0, ms, dFill));
  hipLaunchKernelGGL(k_kf_rescale_setdepth, dim3(nwg5, n), dim3(256), 0, ms, dRes);
  HIPCHK(hipGetLastError());
  rc = lsd_args_release(c, dev, ms);
  if (rc) return rc;
  rc = lsd_frame_build_idepth_pyramid_batch(pyrFrames.data(), 2 * n, pyrParts.data(), 0, pyrOuts.data(), pyrN.data());
  if (rc) return rc;
  rc = lsd_bprof_end(c, bp, ms, (double)n * npx);
  if (rc) return rc;
  txn.done = true;            // every launch of the change is queued: the new state stands
  for (int j = 0; j < n; j++) {
    maps[j]->propClean = true;
    pyrFrames[j]->depthHasBeenUpdatedFlag = true;
    pyrFrames[j]->reActValid = true;
    pyrFrames[n + j]->depthHasBeenUpdatedFlag = true;
  }
  rc = timing_end(maps[0], ev);
  if (rc) return rc;
  if (lsd_m_record(c) < 0) return LSDHIP_E_HIP;
  if (!c->async || rescale_out) {
    for (int j = 0; j < n; j++) {
      if (!c->async) { rc = lsd_frame_resolve(pyrFrames[j]); if (rc) return rc; }
      rc = lsd_frame_resolve(pyrFrames[n + j]);
      if (rc) return rc;
      if (rescale_out) rescale_out[j] = (float)pyrFrames[n + j]->thisToParent_raw.s;
    }
  }
  const double dt = now_ms() - t0;
  for (int j = 0; j < n; j++) maps[j]->msCreate = ema(maps[j]->msCreate, dt / n);
  return LSDHIP_OK;
}

extern "C" int lsdhip_depth_finalize(lsdhip_depthmap* dm) {
  if (!dm) return LSDHIP_E_ARG;
  if (!dm->activeKeyFrame) { lsd_set_error("finalizeKeyFrame: depth map has no active keyframe"); return LSDHIP_E_STATE; }
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  double t0 = now_ms();
  if (int rcb = lsd_m_begin(c)) return rcb;
  const int ev = timing_begin(dm, 2);
  if (ev < 0 && ev != EV_SKIP) return ev;
  int rc = fill_regularize(dm, VAL_SUM_MIN_FOR_KEEP, true);
  if (rc) return rc;
  lsdhip_frame* kf = dm->activeKeyFrame;
  size_t n = (size_t)c->w * c->h;
  hipLaunchKernelGGL(k_take_react, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lsd_map_stream(c), dm->cur, kf->d_idepth_reAct,
                     kf->d_idepthVar_reAct, kf->d_validity_reAct, (int)n);
  rc = timing_end(dm, ev);
  if (rc) return rc;
  if (lsd_m_record(c) < 0) return LSDHIP_E_HIP;
  if (!c->async) HIPCHK(hipStreamSynchronize(lsd_map_stream(c)));
  kf->reActValid = true;
  dm->msFinalize = ema(dm->msFinalize, now_ms() - t0);
  return LSDHIP_OK;
}

// GPU time (ms, summed) and call counts of updateKeyframe / createKeyFrame / finalizeKeyFrame since creation
extern "C" int lsdhip_depth_observe_time(lsdhip_depthmap* dm, double* ms_out, long long* calls_out) {
  if (!dm || !ms_out || !calls_out) return LSDHIP_E_ARG;
  HIPCHK(hipSetDevice(dm->ctx->device));
  HIPCHK(hipStreamSynchronize(lsd_map_stream(dm->ctx)));
  timing_collect(dm, true);
  *ms_out = dm->gpu_ms[3];
  *calls_out = dm->gpu_calls[3];
  return LSDHIP_OK;
}
// the sampled k_observe launches' work: out[0] = launches counted, out[1] = pixels that entered the epipolar search, out[2] = walk steps
// (sum of doLineStereo's loopCounter); synchronises
extern "C" int lsdhip_depth_observe_work(lsdhip_depthmap* dm, double out[3]) {
  if (!dm || !out) return LSDHIP_E_ARG;
  out[0] = out[1] = out[2] = 0;
  if (!dm->d_obsAcc) return LSDHIP_OK;
  HIPCHK(hipSetDevice(dm->ctx->device));
  HIPCHK(hipStreamSynchronize(lsd_map_stream(dm->ctx)));
  unsigned long long acc[3];
  HIPCHK(hipMemcpy(acc, dm->d_obsAcc, sizeof(acc), hipMemcpyDeviceToHost));
  out[0] = (double)acc[2]; out[1] = (double)acc[0]; out[2] = (double)acc[1];
  return LSDHIP_OK;
}
extern "C" int lsdhip_depth_gpu_times(lsdhip_depthmap* dm, double ms_out[3], long long calls_out[3]) {
  if (!dm || !ms_out || !calls_out) return LSDHIP_E_ARG;
  HIPCHK(hipSetDevice(dm->ctx->device));
  HIPCHK(hipStreamSynchronize(lsd_map_stream(dm->ctx)));
  timing_collect(dm, true);
  for (int i = 0; i < 3; i++) { ms_out[i] = dm->gpu_ms[i]; calls_out[i] = dm->gpu_calls[i]; }
  return LSDHIP_OK;
}

extern "C" int lsdhip_depth_download(lsdhip_depthmap* dm, lsdhip_hypothesis* out) {
  if (!dm || !out) return LSDHIP_E_ARG;
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  size_t n = (size_t)c->w * c->h;
  std::vector<uint8_t> v(n);
  std::vector<int32_t> bl(n), vc(n);
  std::vector<float> nid(n), id(n), var(n), ids(n), vars(n);
  if (int rcb = lsd_m_begin(c)) return rcb;
  HIPCHK(hipMemcpyAsync(v.data(), dm->cur.valid, n, hipMemcpyDeviceToHost, lsd_map_stream(c)));
  HIPCHK(hipMemcpyAsync(bl.data(), dm->cur.blacklisted, n * 4, hipMemcpyDeviceToHost, lsd_map_stream(c)));
  HIPCHK(hipMemcpyAsync(vc.data(), dm->cur.validity, n * 4, hipMemcpyDeviceToHost, lsd_map_stream(c)));
  HIPCHK(hipMemcpyAsync(nid.data(), dm->cur.nextID, n * 4, hipMemcpyDeviceToHost, lsd_map_stream(c)));
  HIPCHK(hipMemcpyAsync(id.data(), dm->cur.idepth, n * 4, hipMemcpyDeviceToHost, lsd_map_stream(c)));
  HIPCHK(hipMemcpyAsync(var.data(), dm->cur.var, n * 4, hipMemcpyDeviceToHost, lsd_map_stream(c)));
  HIPCHK(hipMemcpyAsync(ids.data(), dm->cur.idepth_s, n * 4, hipMemcpyDeviceToHost, lsd_map_stream(c)));
  HIPCHK(hipMemcpyAsync(vars.data(), dm->cur.var_s, n * 4, hipMemcpyDeviceToHost, lsd_map_stream(c)));
  HIPCHK(hipStreamSynchronize(lsd_map_stream(c)));
  for (size_t i = 0; i < n; i++) {
    lsdhip_hypothesis& h = out[i];
    h.isValid = v[i]; h.pad_[0] = h.pad_[1] = h.pad_[2] = 0;
    h.blacklisted = bl[i]; h.nextStereoFrameMinID = nid[i]; h.validity_counter = vc[i];
    h.idepth = id[i]; h.idepth_var = var[i]; h.idepth_smoothed = ids[i]; h.idepth_var_smoothed = vars[i];
  }
  return LSDHIP_OK;
}
extern "C" int lsdhip_depth_upload(lsdhip_depthmap* dm, lsdhip_frame* kf, const lsdhip_hypothesis* in, int reactivated) {
  if (!dm || !kf || !in) return LSDHIP_E_ARG;
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  size_t n = (size_t)c->w * c->h;
  std::vector<uint8_t> v(n);
  std::vector<int32_t> bl(n), vc(n);
  std::vector<float> nid(n), id(n), var(n), ids(n), vars(n);
  for (size_t i = 0; i < n; i++) {
    const lsdhip_hypothesis& h = in[i];
    // (the fused regulariser packs "invalid" as a negative counter in its LDS tile; the reference never produces a negative one)
    if (h.isValid && h.validity_counter < 0) { lsd_set_error("lsdhip_depth_upload: pixel %zu is valid with validity_counter %d < 0", i, h.validity_counter); return LSDHIP_E_ARG; }
    v[i] = h.isValid ? 1 : 0; bl[i] = h.blacklisted; nid[i] = h.nextStereoFrameMinID; vc[i] = h.validity_counter;
    id[i] = h.idepth; var[i] = h.idepth_var; ids[i] = h.idepth_smoothed; vars[i] = h.idepth_var_smoothed;
  }
  if (int rcb = lsd_m_begin(c)) return rcb;
  HIPCHK(hipMemcpyAsync(dm->cur.valid, v.data(), n, hipMemcpyHostToDevice, lsd_map_stream(c)));
  HIPCHK(hipMemcpyAsync(dm->cur.blacklisted, bl.data(), n * 4, hipMemcpyHostToDevice, lsd_map_stream(c)));
  HIPCHK(hipMemcpyAsync(dm->cur.validity, vc.data(), n * 4, hipMemcpyHostToDevice, lsd_map_stream(c)));
  HIPCHK(hipMemcpyAsync(dm->cur.nextID, nid.data(), n * 4, hipMemcpyHostToDevice, lsd_map_stream(c)));
  HIPCHK(hipMemcpyAsync(dm->cur.idepth, id.data(), n * 4, hipMemcpyHostToDevice, lsd_map_stream(c)));
  HIPCHK(hipMemcpyAsync(dm->cur.var, var.data(), n * 4, hipMemcpyHostToDevice, lsd_map_stream(c)));
  HIPCHK(hipMemcpyAsync(dm->cur.idepth_s, ids.data(), n * 4, hipMemcpyHostToDevice, lsd_map_stream(c)));
  HIPCHK(hipMemcpyAsync(dm->cur.var_s, vars.data(), n * 4, hipMemcpyHostToDevice, lsd_map_stream(c)));
  HIPCHK(hipStreamSynchronize(lsd_map_stream(c)));
  if (int rcl = lsd_frame_require_level0(kf)) return rcl;
  dm->lowGradHypPossible = true;
  dm->activeKeyFrame = kf;
  dm->activeKeyFrameIsReactivated = reactivated != 0;
  return LSDHIP_OK;
}

extern "C" int lsdhip_depth_stage(lsdhip_depthmap* dm, int stage, lsdhip_frame** refs, int n) {
  if (!dm) return LSDHIP_E_ARG;
  if (!dm->activeKeyFrame) { lsd_set_error("depth stage: no active keyframe"); return LSDHIP_E_STATE; }
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  int rc = LSDHIP_E_ARG;
  if (int rcb = lsd_m_begin(c)) return rcb;
  switch (stage) {
    case 0: rc = observe(dm, refs, n); break;
    case 1: rc = fill_holes(dm); break;
    case 2: rc = regularize(dm, false, VAL_SUM_MIN_FOR_KEEP); break;
    case 3: rc = regularize(dm, true, VAL_SUM_MIN_FOR_KEEP); break;
    case 4:
      if (!refs || n < 1) return LSDHIP_E_ARG;
      rc = lsd_frame_require_level0(refs[0]);
      if (rc == LSDHIP_OK) rc = propagate(dm, refs[0], true);
      if (rc == LSDHIP_OK) { dm->activeKeyFrame = refs[0]; dm->activeKeyFrameIsReactivated = false; }
      break;
    case 5: rc = fill_regularize(dm, VAL_SUM_MIN_FOR_KEEP, false); break;
    default: return LSDHIP_E_ARG;
  }
  if (rc) return rc;
  if (lsd_m_record(c) < 0) return LSDHIP_E_HIP;
  if (!c->async) HIPCHK(hipStreamSynchronize(lsd_map_stream(c)));   // asynchronous contexts: ordered on the stream
  return LSDHIP_OK;
}

extern "C" int lsdhip_depth_stage_rows(lsdhip_depthmap* dm, int stage, int tile_row0, int n_tile_rows, int last) {
  if (!dm || stage != 5) return LSDHIP_E_ARG;
  if (!dm->activeKeyFrame) { lsd_set_error("depth stage: no active keyframe"); return LSDHIP_E_STATE; }
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  int rc = fill_regularize_rows(dm, VAL_SUM_MIN_FOR_KEEP, tile_row0, n_tile_rows, last != 0);
  if (rc) return rc;
  if (!c->async) HIPCHK(hipStreamSynchronize(lsd_map_stream(c)));
  return LSDHIP_OK;
}

extern "C" int lsdhip_depth_stage_rows_batch(lsdhip_ctx* c, int n, lsdhip_depthmap** maps, const int* tile_row0, const int* n_tile_rows) {
  if (!c || n < 0 || (n > 0 && (!maps || !tile_row0 || !n_tile_rows))) return LSDHIP_E_ARG;
  if (n == 0) return LSDHIP_OK;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  const int T = (c->h + 7) / 8;
  int maxRows = 0;
  for (int k = 0; k < n; k++) {
    if (!maps[k] || maps[k]->ctx != c) return LSDHIP_E_ARG;
    if (!maps[k]->activeKeyFrame) { lsd_set_error("depth stage: no active keyframe"); return LSDHIP_E_STATE; }
    if (tile_row0[k] < 0 || n_tile_rows[k] <= 0 || tile_row0[k] + n_tile_rows[k] > T) { lsd_set_error("depth stage rows: tile rows [%d, %d) outside [0, %d)", tile_row0[k], tile_row0[k] + n_tile_rows[k], T); return LSDHIP_E_ARG; }
    if (n_tile_rows[k] > maxRows) maxRows = n_tile_rows[k];
  }
  void* host = nullptr;
  void* dev = nullptr;
  int rc = lsd_args_begin(c, sizeof(RegBatchItem) * (size_t)n, &host, &dev);
  if (rc) return rc;
  RegBatchItem* items = (RegBatchItem*)host;
  for (int k = 0; k < n; k++) {
    memset((void*)&items[k], 0, sizeof(RegBatchItem));
    items[k].a = reg_args(maps[k], VAL_SUM_MIN_FOR_KEEP);
    items[k].a.tileRow0 = tile_row0[k];
    items[k].a.tileRows = n_tile_rows[k];
  }
  const hipStream_t ms = lsd_map_stream(c);
  rc = lsd_args_commit(c, ms);
  if (rc) return rc;
  long long wg1 = 0;
  for (int k = 0; k < n; k++) wg1 += (long long)((c->w + 31) / 32) * n_tile_rows[k];
  const int bpy = lsd_reg_py(wg1);
  LSD_REG_DISPATCH(bpy, hipLaunchKernelGGL(k_reg_rows_batch<PYV>, dim3((c->w + 31) / 32, lsd_reg_grid_rows(maxRows, bpy), n), dim3(256), 0, ms, (const RegBatchItem*)dev));
  HIPCHK(hipGetLastError());
  rc = lsd_args_release(c, dev, ms);
  if (rc) return rc;
  if (!c->async) HIPCHK(hipStreamSynchronize(ms));
  return LSDHIP_OK;
}

extern "C" int lsdhip_depth_copy_planes_dev(lsdhip_depthmap* dm, float* idepth_dev, float* var_dev) {
  if (!dm || !idepth_dev || !var_dev) return LSDHIP_E_ARG;
  if (!dm->activeKeyFrame || !(dm->activeKeyFrame->hasIDepth || dm->activeKeyFrame->depthPending)) return LSDHIP_E_STATE;
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  size_t n = (size_t)c->w * c->h;
  // what the last Frame::setDepth of the mapping side left (on a pipelined context possibly not yet published to the tracker)
  if (int rcb = lsd_m_begin(c)) return rcb;
  HIPCHK(hipMemcpyAsync(idepth_dev, lsd_depth_latest(dm->activeKeyFrame)[0], n * 4, hipMemcpyDeviceToDevice, lsd_map_stream(c)));
  HIPCHK(hipMemcpyAsync(var_dev, lsd_depthvar_latest(dm->activeKeyFrame)[0], n * 4, hipMemcpyDeviceToDevice, lsd_map_stream(c)));
  if (lsd_m_record(c) < 0) return LSDHIP_E_HIP;
  if (!c->async) HIPCHK(hipStreamSynchronize(lsd_map_stream(c)));   // asynchronous contexts: ordered on the stream, see lsdhip_ctx_synchronize
  return LSDHIP_OK;
}

// Rows [row0, row0 + nrows) of the eight hypothesis planes <-> one packed device buffer (plane after plane: isValid u8,
// blacklisted i32, nextStereoFrameMinID f32, validity_counter i32, idepth, idepth_var, idepth_smoothed,
// idepth_var_smoothed = 29 bytes per pixel): the halo payload of the row-band decomposition.
extern "C" int lsdhip_depth_copy_rows_dev(lsdhip_depthmap* dm, int row0, int nrows, void* packed_dev, int to_map) {
  if (!dm || !packed_dev || row0 < 0 || nrows <= 0) return LSDHIP_E_ARG;
  lsdhip_ctx* c = dm->ctx;
  LSD_CTX_LOCK(c);
  if (row0 + nrows > c->h) return LSDHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  const size_t px = (size_t)nrows * c->w, off = (size_t)row0 * c->w;
  char* p = (char*)packed_dev;
  struct { void* plane; size_t elt; } planes[8] = {
      {dm->cur.valid, 1}, {dm->cur.blacklisted, 4}, {dm->cur.nextID, 4}, {dm->cur.validity, 4},
      {dm->cur.idepth, 4}, {dm->cur.var, 4}, {dm->cur.idepth_s, 4}, {dm->cur.var_s, 4}};
  for (int k = 0; k < 8; k++) {
    char* mp = (char*)planes[k].plane + off * planes[k].elt;
    const size_t bytes = px * planes[k].elt;
    if (to_map) HIPCHK(hipMemcpyAsync(mp, p, bytes, hipMemcpyDeviceToDevice, lsd_map_stream(c)));
    else HIPCHK(hipMemcpyAsync(p, mp, bytes, hipMemcpyDeviceToDevice, lsd_map_stream(c)));
    p += bytes;
  }
  HIPCHK(hipStreamSynchronize(lsd_map_stream(c)));
  return LSDHIP_OK;
}

// Several row ranges in ONE launch: map rows -> map rows of another window (halo refresh between windows that live on the same
// GPU), map rows -> packed buffer (before an ncclSend), packed buffer -> map rows (after an ncclRecv).  Everything is ordered
// on the context's stream; nothing synchronises the host.
struct CopyPlanes { char* p[8]; };
struct CopyItem { CopyPlanes src, dst; int npx; };
#define LSD_COPY_ITEMS 16
struct CopyBatch { CopyItem it[LSD_COPY_ITEMS]; };
__global__ __launch_bounds__(256) void k_copy_rows(CopyBatch b, int vec) {
  // the item is picked by blockIdx.y: its 16 plane pointers and its size are read from the kernel-argument segment directly
  // (indexing the by-value struct with a runtime index would copy it to scratch)
  typedef const unsigned long long __attribute__((address_space(4))) cu64;
  cu64* q = (cu64*)__builtin_amdgcn_kernarg_segment_ptr() + (size_t)blockIdx.y * (sizeof(CopyItem) / 8);
  const char* s0 = (const char*)q[0]; const char* s1 = (const char*)q[1]; const char* s2 = (const char*)q[2]; const char* s3 = (const char*)q[3];
  const char* s4 = (const char*)q[4]; const char* s5 = (const char*)q[5]; const char* s6 = (const char*)q[6]; const char* s7 = (const char*)q[7];
  char* d0 = (char*)q[8]; char* d1 = (char*)q[9]; char* d2 = (char*)q[10]; char* d3 = (char*)q[11];
  char* d4 = (char*)q[12]; char* d5 = (char*)q[13]; char* d6 = (char*)q[14]; char* d7 = (char*)q[15];
  const int npx = (int)(unsigned)q[16];
  if (vec) {   // 4 pixels per lane: one 4-byte word of the validity plane, one 16-byte vector of each 4-byte plane
    const int n4 = npx >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
      const unsigned v0 = ((const unsigned*)s0)[i];
      const uint4 v1 = ((const uint4*)s1)[i], v2 = ((const uint4*)s2)[i], v3 = ((const uint4*)s3)[i], v4 = ((const uint4*)s4)[i];
      const uint4 v5 = ((const uint4*)s5)[i], v6 = ((const uint4*)s6)[i], v7 = ((const uint4*)s7)[i];
      ((unsigned*)d0)[i] = v0;
      ((uint4*)d1)[i] = v1; ((uint4*)d2)[i] = v2; ((uint4*)d3)[i] = v3; ((uint4*)d4)[i] = v4;
      ((uint4*)d5)[i] = v5; ((uint4*)d6)[i] = v6; ((uint4*)d7)[i] = v7;
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npx; i += gridDim.x * 256) {
      d0[i] = s0[i];
      ((unsigned*)d1)[i] = ((const unsigned*)s1)[i]; ((unsigned*)d2)[i] = ((const unsigned*)s2)[i]; ((unsigned*)d3)[i] = ((const unsigned*)s3)[i];
      ((unsigned*)d4)[i] = ((const unsigned*)s4)[i]; ((unsigned*)d5)[i] = ((const unsigned*)s5)[i]; ((unsigned*)d6)[i] = ((const unsigned*)s6)[i];
      ((unsigned*)d7)[i] = ((const unsigned*)s7)[i];
    }
  }
}
static_assert(sizeof(CopyItem) == 17 * 8, "k_copy_rows reads a CopyItem as 17 quadwords");
static bool copy_planes_of(lsdhip_ctx* c, lsdhip_depthmap* dm, int row0, void* packed, int nrows, CopyPlanes& out) {
  const size_t px = (size_t)nrows * c->w;
  if (dm) {
    if (dm->ctx != c || row0 < 0 || row0 + nrows > c->h) return false;
    const size_t off = (size_t)row0 * c->w;
    void* planes[8] = {dm->cur.valid, dm->cur.blacklisted, dm->cur.nextID, dm->cur.validity, dm->cur.idepth, dm->cur.var, dm->cur.idepth_s, dm->cur.var_s};
    for (int k = 0; k < 8; k++) out.p[k] = (char*)planes[k] + off * (k == 0 ? 1 : 4);
  } else {
    if (!packed) return false;
    char* p = (char*)packed;
    for (int k = 0; k < 8; k++) { out.p[k] = p; p += px * (k == 0 ? 1 : 4); }
  }
  return true;
}
extern "C" int lsdhip_depth_copy_rows_batch(lsdhip_ctx* c, int n, const lsdhip_row_copy* items) {
  if (!c || n < 0 || (n > 0 && !items)) return LSDHIP_E_ARG;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  for (int base = 0; base < n; base += LSD_COPY_ITEMS) {
    CopyBatch b;
    memset(&b, 0, sizeof(b));
    const int m = n - base < LSD_COPY_ITEMS ? n - base : LSD_COPY_ITEMS;
    int maxpx = 0, vec = 1;
    for (int i = 0; i < m; i++) {
      const lsdhip_row_copy& r = items[base + i];
      if (r.nrows <= 0 || !copy_planes_of(c, r.src_map, r.src_row0, r.src_packed, r.nrows, b.it[i].src) ||
          !copy_planes_of(c, r.dst_map, r.dst_row0, r.dst_packed, r.nrows, b.it[i].dst)) {
        lsd_set_error("lsdhip_depth_copy_rows_batch: item %d is not a row range of a depth map of this context / a packed buffer", base + i);
        return LSDHIP_E_ARG;
      }
      b.it[i].npx = r.nrows * c->w;
      if (b.it[i].npx > maxpx) maxpx = b.it[i].npx;
      if (b.it[i].npx & 3) vec = 0;
      for (int k = 0; k < 8; k++)
        if (((uintptr_t)b.it[i].src.p[k] | (uintptr_t)b.it[i].dst.p[k]) & (k == 0 ? 3 : 15)) vec = 0;
    }
    const int per = vec ? maxpx / 4 : maxpx;
    int gx = (per + 255) / 256;
    if (gx > 512) gx = 512;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_copy_rows, dim3(gx, m), dim3(256), 0, lsd_transport_stream(c), b, vec);
  }
  HIPCHK(hipGetLastError());
  if (!c->async) HIPCHK(hipStreamSynchronize(lsd_transport_stream(c)));
  return LSDHIP_OK;
}

extern "C" int lsdhip_depth_timings(lsdhip_depthmap* dm, float out[8]) {
  if (!dm || !out) return LSDHIP_E_ARG;
  out[0] = dm->msUpdate; out[1] = dm->msCreate; out[2] = dm->msFinalize; out[3] = dm->msObserve; out[4] = dm->msRegularize;
  out[5] = dm->msPropagate; out[6] = dm->msFillHoles; out[7] = dm->msSetDepth;
  return LSDHIP_OK;
}
