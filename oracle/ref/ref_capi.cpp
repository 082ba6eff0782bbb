// ORACLE — TEST INFRASTRUCTURE ONLY.  The reference itself behind the oracle's flat C entry points.
//
// oracle/_ref/liblsd_ref_{sse,scalar}.so = the reference's OWN hot-path translation units
//   C/util/settings.cpp, util/SophusUtil.cpp, DataStructures/{Frame,FramePoseStruct,FrameMemory}.cpp,
//   DepthEstimation/{DepthMap,DepthMapPixelHypothesis}.cpp, Tracking/{SE3Tracker,Sim3Tracker,TrackingReference}.cpp
// compiled UNCHANGED from where they lie under /root/reference/lsd_slam_core/src (C/), plus this file, which exports
// the same orc_* C symbols as oracle/orc_capi.cpp so that oracle/pyoracle.py can drive either library with the same
// Python classes and tests/test_ref_pin_cpu.py can compare them call for call.  The external dependencies the
// reference needs and this machine lacks (Eigen, Sophus-on-Eigen, boost, OpenCV, g2o) are replaced by the stand-in
// headers under oracle/ref/shim/ (each says what it stands in for).  Nothing here is copied from the reference: this
// file only calls its classes.  Private members are reached with the usual test-harness `#define private public`.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <vector>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <condition_variable>
#include <functional>
#include <unordered_map>
#include <unordered_set>
#include <list>
#include <string>
#include <iostream>

#include <Eigen/Core>
#include "sophus/sim3.hpp"
#include <opencv2/core/core.hpp>
#include "boost/thread.hpp"

#define private public
#define protected public
#include "DataStructures/Frame.h"
#include "DataStructures/FrameMemory.h"
#include "DepthEstimation/DepthMap.h"
#include "DepthEstimation/DepthMapPixelHypothesis.h"
#include "Tracking/SE3Tracker.h"
#include "Tracking/Sim3Tracker.h"
#include "Tracking/TrackingReference.h"
#undef private
#undef protected
#include "IOWrapper/ImageDisplay.h"
#include "util/globalFuncs.h"
#include "util/settings.h"

using namespace lsd_slam;

// ---- symbols the compiled reference files reference but whose translation units are out of scope ----------------
namespace lsd_slam {
namespace Util {
void displayImage(const char*, const cv::Mat&, bool) {}       // C/IOWrapper/OpenCV/ImageDisplay_OpenCV.cpp (GUI)
int waitKey(int) { return 0; }
int waitKeyNoConsume(int) { return 0; }
void closeAllWindows() {}
}  // namespace Util
}  // namespace lsd_slam
extern "C" {

struct orc_params {
  float minUseGrad, cameraPixelNoise2, depthSmoothingFactor;
  int allowNegativeIdepths, useSubpixelStereo, multiThreading, useAffineLightningEstimation;
  float KFDistWeight, KFUsageWeight;
};
static orc_params default_params_() {
  orc_params p;
  p.minUseGrad = 5; p.cameraPixelNoise2 = 4 * 4; p.depthSmoothingFactor = 1;     // C/util/settings.cpp:77-88
  p.allowNegativeIdepths = 1; p.useSubpixelStereo = 1; p.multiThreading = 1; p.useAffineLightningEstimation = 1;
  p.KFDistWeight = 4; p.KFUsageWeight = 3;
  return p;
}
// the reference keeps these as process-wide globals: every entry point installs its object's block first
static void fresh();
static void apply(const orc_params& p) {
  fresh();
  lsd_slam::minUseGrad = p.minUseGrad; lsd_slam::cameraPixelNoise2 = p.cameraPixelNoise2;
  lsd_slam::depthSmoothingFactor = p.depthSmoothingFactor;
  lsd_slam::allowNegativeIdepths = p.allowNegativeIdepths != 0; lsd_slam::useSubpixelStereo = p.useSubpixelStereo != 0;
  lsd_slam::multiThreading = p.multiThreading != 0;
  lsd_slam::useAffineLightningEstimation = p.useAffineLightningEstimation != 0;
  lsd_slam::KFDistWeight = p.KFDistWeight; lsd_slam::KFUsageWeight = p.KFUsageWeight;
}
void orc_default_params(orc_params* p) { *p = default_params_(); }

// The reference reads memory it never wrote: Frame::buildMaxGradients (C/DataStructures/Frame.cpp:722-733) takes the
// 3-row maximum of rows 0 and h-1 of a plane whose first and last rows are never written, and its buffers come
// uninitialised from the FrameMemory pool (C/DataStructures/FrameMemory.cpp:80-127) — the real binary's results there
// depend on what the recycled buffer held.  The oracle defines unwritten pool memory as 0.  To give the reference the
// same, deterministic, convention the stand-in aligned_malloc returns zeroed memory and every entry point below starts
// with an empty free list, so that no buffer is handed out twice across calls.
static void fresh() { FrameMemory::getInstance().releaseBuffes(); }

static Eigen::Matrix3f Kmat(const float K[4]) {
  Eigen::Matrix3f m;
  m << K[0], 0.f, K[2], 0.f, K[1], K[3], 0.f, 0.f, 1.f;
  return m;
}

// ---- Frame -------------------------------------------------------------------------------------
typedef std::shared_ptr<Frame> FramePtr;
struct FrameBox {
  FramePtr f;
  FramePtr parent_keep;     // the tracking parent must outlive its children (pose->trackingParent is a raw pointer)
};
static Frame* F(void* f) { return ((FrameBox*)f)->f.get(); }
void* orc_frame_create(int id, int w, int h, const float K[4], const unsigned char* img) {
  FrameBox* b = new FrameBox();
  b->f.reset(new Frame(id, w, h, Kmat(K), 0.0, img));
  return b;
}
void orc_frame_destroy(void* f) {
  delete (FrameBox*)f;
  // the pool hands recycled buffers out uninitialised; the oracle defines unwritten pool memory as 0, and so does the
  // stand-in aligned_malloc — dropping the free list keeps that true across frames
  FrameMemory::getInstance().releaseBuffes();
}
int orc_frame_get(void* f, int what, int level, float* out) {
  fresh();
  Frame* fr = F(f);
  size_t n = (size_t)fr->width(level) * fr->height(level);
  const float* src = nullptr;
  switch (what) {
    case 0: src = fr->image(level); break;
    case 1: src = (const float*)fr->gradients(level); n *= 4; break;
    case 2: src = fr->maxGradients(level); break;
    case 3: if (!fr->hasIDepthBeenSet()) return -1; src = fr->idepth(level); break;
    case 4: if (!fr->hasIDepthBeenSet()) return -1; src = fr->idepthVar(level); break;
    default: return -2;
  }
  memcpy(out, src, n * sizeof(float));
  return 0;
}
void orc_frame_intrinsics(void* f, int level, float out[8]) {
  Frame* fr = F(f);
  out[0] = fr->fx(level); out[1] = fr->fy(level); out[2] = fr->cx(level); out[3] = fr->cy(level);
  out[4] = fr->fxInv(level); out[5] = fr->fyInv(level); out[6] = fr->cxInv(level); out[7] = fr->cyInv(level);
}
void orc_frame_set_sse_pyramid(void*, int) {}    // compile-time in the reference (ENABLE_SSE)
void orc_frame_set_depth_gt(void* f, const float* depth, float cov_scale, float minUseGrad_) {
  fresh();
  float saved = lsd_slam::minUseGrad;
  lsd_slam::minUseGrad = minUseGrad_;
  F(f)->setDepthFromGroundTruth(depth, cov_scale);
  lsd_slam::minUseGrad = saved;
}
// test hook: raw level-0 (idepth, idepthVar) planes, bookkeeping as Frame::setDepthFromGroundTruth (C/DataStructures/Frame.cpp:246-292)
void orc_frame_set_depth_planes(void* f, const float* id, const float* var) {
  fresh();
  Frame* fr = F(f);
  size_t n = (size_t)fr->width(0) * fr->height(0);
  if (fr->data.idepth[0] == 0) fr->data.idepth[0] = FrameMemory::getInstance().getFloatBuffer(n);
  if (fr->data.idepthVar[0] == 0) fr->data.idepthVar[0] = FrameMemory::getInstance().getFloatBuffer(n);
  memcpy(fr->data.idepth[0], id, n * sizeof(float));
  memcpy(fr->data.idepthVar[0], var, n * sizeof(float));
  fr->data.idepthValid[0] = true;
  fr->data.idepthVarValid[0] = true;
  fr->release(Frame::IDEPTH | Frame::IDEPTH_VAR, true, true);
  fr->data.hasIDepthBeenSet = true;
}
int orc_frame_get_wasgood(void* f, unsigned char* out) {
  Frame* fr = F(f);
  bool* p = fr->refPixelWasGoodNoCreate();
  if (!p) return 0;
  memcpy(out, p, (size_t)fr->width(1) * fr->height(1));
  return 1;
}
void orc_frame_set_wasgood(void* f, const unsigned char* in) {
  Frame* fr = F(f);
  memcpy(fr->refPixelWasGood(), in, (size_t)fr->width(1) * fr->height(1));
}
void orc_frame_clear_wasgood(void* f) { F(f)->clear_refPixelWasGood(); }
void orc_frame_set_maxgrad(void* f, const float* in) {
  fresh();
  Frame* fr = F(f);
  fr->maxGradients(0);
  memcpy(fr->data.maxGradients[0], in, sizeof(float) * (size_t)fr->width(0) * fr->height(0));
}
void orc_frame_set_pose(void* f, const double sim3[8], void* parent, float initialTrackedResidual) {
  FrameBox* b = (FrameBox*)f;
  Frame* fr = b->f.get();
  Sim3 T(Eigen::Quaterniond(sim3[0], sim3[1], sim3[2], sim3[3]), Eigen::Vector3d(sim3[4], sim3[5], sim3[6]));
  T.setScale(sim3[7]);
  fr->pose->thisToParent_raw = T;
  if (parent) {
    b->parent_keep = ((FrameBox*)parent)->f;
    fr->pose->trackingParent = F(parent)->pose;
  } else {
    fr->pose->trackingParent = nullptr;
  }
  fr->initialTrackedResidual = initialTrackedResidual;
}
void orc_frame_get_pose(void* f, double sim3[8]) {
  const Sim3& T = F(f)->pose->thisToParent_raw;
  sim3[0] = T.q.w; sim3[1] = T.q.x; sim3[2] = T.q.y; sim3[3] = T.q.z;
  sim3[4] = T.translation()[0]; sim3[5] = T.translation()[1]; sim3[6] = T.translation()[2];
  sim3[7] = T.scale();
}
void orc_frame_stats(void* f, float out[8]) {
  Frame* fr = F(f);
  out[0] = fr->initialTrackedResidual; out[1] = fr->meanIdepth; out[2] = (float)fr->numPoints;
  out[3] = (float)fr->numFramesTrackedOnThis; out[4] = (float)fr->numMappedOnThis; out[5] = (float)fr->numMappedOnThisTotal;
  out[6] = fr->depthHasBeenUpdatedFlag ? 1.f : 0.f; out[7] = (float)fr->numMappablePixels;
}
void orc_frame_set_counters(void* f, int numFramesTrackedOnThis, int numMappedOnThis, int numMappedOnThisTotal, int depthUpdatedFlag) {
  Frame* fr = F(f);
  fr->numFramesTrackedOnThis = numFramesTrackedOnThis; fr->numMappedOnThis = numMappedOnThis;
  fr->numMappedOnThisTotal = numMappedOnThisTotal; fr->depthHasBeenUpdatedFlag = depthUpdatedFlag != 0;
}
void orc_frame_stereo_precomp(void* f, float out[27]) {
  Frame* fr = F(f);
  int k = 0;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) out[k++] = fr->K_otherToThis_R(i, j);
  for (int i = 0; i < 3; i++) out[k++] = fr->K_otherToThis_t[i];
  for (int i = 0; i < 3; i++) out[k++] = fr->otherToThis_t[i];
  for (int i = 0; i < 3; i++) out[k++] = fr->thisToOther_t[i];
  for (int i = 0; i < 3; i++) out[k++] = fr->otherToThis_R_row0[i];
  for (int i = 0; i < 3; i++) out[k++] = fr->otherToThis_R_row1[i];
  for (int i = 0; i < 3; i++) out[k++] = fr->otherToThis_R_row2[i];
}

// ---- TrackingReference ---------------------------------------------------------------------------
struct RefBox {
  TrackingReference ref;
  FramePtr keep;
  ~RefBox() { ref.invalidate(); }
};
void* orc_ref_create() { return new RefBox(); }
void orc_ref_destroy(void* r) { delete (RefBox*)r; }
void orc_ref_import(void* r, void* f) {
  RefBox* b = (RefBox*)r;
  b->ref.invalidate();       // drop the shared lock on the previous keyframe (C/SlamSystem.cpp does this through importFrame's reassignment)
  b->keep = ((FrameBox*)f)->f;
  b->ref.importFrame(F(f));
}
int orc_ref_pointcloud(void* r, int level, float* pos, float* colvar, float* grad, int* idx) {
  fresh();
  TrackingReference* ref = &((RefBox*)r)->ref;
  ref->makePointCloud(level);
  int n = ref->numData[level];
  if (pos) memcpy(pos, ref->posData[level], sizeof(float) * 3 * n);
  if (colvar) memcpy(colvar, ref->colorAndVarData[level], sizeof(float) * 2 * n);
  if (grad) memcpy(grad, ref->gradData[level], sizeof(float) * 2 * n);
  if (idx) memcpy(idx, ref->pointPosInXYGrid[level], sizeof(int) * n);
  return n;
}

// ---- SE3Tracker -----------------------------------------------------------------------------------
struct orc_track_result {
  double frameToRef[7];
  float pointUsage, lastGoodCount, lastBadCount, lastMeanRes, lastResidual, affine_a, affine_b;
  int diverged, trackingWasGood, numEvaluations, numWarpUpdates;
};
struct ResidualRecord {
  int warped_size;
  float goodCount, badCount, pointUsage, meanRes, retval;
  float affine_a_lastIt, affine_b_lastIt;
  float weightedError;
  float A[36], b[6], lsError;
  double num_constraints;
};
struct TrackerBox {
  SE3Tracker tr;
  orc_params p;
  TrackerBox(int w, int h, const float K[4], const orc_params* pp) : tr(w, h, Kmat(K)), p(pp ? *pp : default_params_()) {}
};
void* orc_tracker_create(int w, int h, const float K[4], const orc_params* p) { return new TrackerBox(w, h, K, p); }
void orc_tracker_destroy(void* t) { delete (TrackerBox*)t; }
void orc_tracker_set_mode(void*, int) {}     // compile-time in the reference: liblsd_ref_sse.so / liblsd_ref_scalar.so
void orc_tracker_set_max_its(void* t, const int its[5]) {
  for (int i = 0; i < 5; i++) ((TrackerBox*)t)->tr.settings.maxItsPerLvl[i] = its[i];
}
static SE3 pose_in(const double p[7]) {
  return SE3(Eigen::Quaterniond(p[0], p[1], p[2], p[3]), Eigen::Vector3d(p[4], p[5], p[6]));
}
static void pose_out(const SE3& T, double p[7]) {
  p[0] = T.g.q.w; p[1] = T.g.q.x; p[2] = T.g.q.y; p[3] = T.g.q.z; p[4] = T.g.t[0]; p[5] = T.g.t[1]; p[6] = T.g.t[2];
}
static void fill_result(SE3Tracker* tr, const SE3& T, orc_track_result* out) {
  pose_out(T, out->frameToRef);
  out->pointUsage = tr->pointUsage; out->lastGoodCount = tr->lastGoodCount; out->lastBadCount = tr->lastBadCount;
  out->lastMeanRes = tr->lastMeanRes; out->lastResidual = tr->lastResidual;
  out->affine_a = tr->affineEstimation_a; out->affine_b = tr->affineEstimation_b;
  out->diverged = tr->diverged; out->trackingWasGood = tr->trackingWasGood;
  out->numEvaluations = -1; out->numWarpUpdates = -1;   // locals of trackFrame in the reference (SE3Tracker.cpp:310-311): not observable
}
void orc_tracker_track(void* t, void* ref, void* frame, const double init_frameToRef[7], orc_track_result* out) {
  TrackerBox* b = (TrackerBox*)t;
  apply(b->p);
  SE3 T = b->tr.trackFrame(&((RefBox*)ref)->ref, F(frame), pose_in(init_frameToRef));
  ((FrameBox*)frame)->parent_keep = ((RefBox*)ref)->keep;
  fill_result(&b->tr, T, out);
}
#ifndef LSD_REF_HIP_BACKED
#if defined(ENABLE_SSE)
#define REF_CALL(function, arguments) function##SSE arguments
#else
#define REF_CALL(function, arguments) function arguments
#endif
// K1 + K2 + K3 once at a fixed pose: the three private members in the order trackFrame calls them (C/Tracking/SE3Tracker.cpp:323-345)
void orc_tracker_evaluate(void* t, void* ref, void* frame, const float refToFrame[7], int level, float a, float b,
                          ResidualRecord* out) {
  TrackerBox* bx = (TrackerBox*)t;
  apply(bx->p);
  SE3Tracker* tr = &bx->tr;
  TrackingReference* reference = &((RefBox*)ref)->ref;
  Sophus::SE3f T(Eigen::Quaternionf(refToFrame[0], refToFrame[1], refToFrame[2], refToFrame[3]),
                 Eigen::Vector3f(refToFrame[4], refToFrame[5], refToFrame[6]));
  // the oracle's hook takes the quaternion as given (no re-normalisation): keep the bits
  T.g.q.w = refToFrame[0]; T.g.q.x = refToFrame[1]; T.g.q.y = refToFrame[2]; T.g.q.z = refToFrame[3];
  reference->makePointCloud(level);
  tr->affineEstimation_a = a;
  tr->affineEstimation_b = b;
  float rv = tr->REF_CALL(calcResidualAndBuffers, (reference->posData[level], reference->colorAndVarData[level],
                                                   SE3TRACKING_MIN_LEVEL == level ? reference->pointPosInXYGrid[level] : 0,
                                                   reference->numData[level], F(frame), T, level, false));
  out->warped_size = tr->buf_warped_size;
  out->goodCount = tr->lastGoodCount; out->badCount = tr->lastBadCount; out->pointUsage = tr->pointUsage;
  out->meanRes = tr->lastMeanRes; out->retval = rv;
  out->affine_a_lastIt = tr->affineEstimation_a_lastIt; out->affine_b_lastIt = tr->affineEstimation_b_lastIt;
  out->weightedError = tr->REF_CALL(calcWeightsAndResidual, (T));
  LGS6 ls;
  tr->REF_CALL(calculateWarpUpdate, (ls));
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) out->A[i * 6 + j] = ls.A(i, j);
  for (int i = 0; i < 6; i++) out->b[i] = ls.b[i];
  out->lsError = ls.error;
  out->num_constraints = (double)ls.num_constraints;
}
int orc_tracker_buffer(void* t, int which, float* out) {
  SE3Tracker* tr = &((TrackerBox*)t)->tr;
  float* bufs[9] = {tr->buf_warped_x, tr->buf_warped_y, tr->buf_warped_z, tr->buf_warped_dx, tr->buf_warped_dy,
                    tr->buf_warped_residual, tr->buf_d, tr->buf_idepthVar, tr->buf_weight_p};
  if (out) memcpy(out, bufs[which], sizeof(float) * tr->buf_warped_size);
  return tr->buf_warped_size;
}
#else
// (hip-backed build, integration/hip_backed: SE3Tracker's members are defined over the C ABI — the private CPU stages do not exist)
void orc_tracker_evaluate(void*, void*, void*, const float*, int, float, float, ResidualRecord*) { fprintf(stderr, "orc_tracker_evaluate: not part of the hip-backed build\n"); abort(); }
int orc_tracker_buffer(void*, int, float*) { return -1; }
#endif
// the reference takes the permanent reference from the keyframe object (Frame::permaRef_*); the oracle's hook takes the
// cloud: install it on a scratch frame of the right size
static void install_permaref(Frame* fr, const float* pos, const float* colvar, int n) {
  if (fr->permaRef_colorAndVarData) delete[] fr->permaRef_colorAndVarData;
  if (fr->permaRef_posData) delete[] fr->permaRef_posData;
  fr->permaRefNumPts = n;
  fr->permaRef_colorAndVarData = new Eigen::Vector2f[n];
  fr->permaRef_posData = new Eigen::Vector3f[n];
  memcpy(fr->permaRef_colorAndVarData, colvar, sizeof(float) * 2 * n);
  memcpy(fr->permaRef_posData, pos, sizeof(float) * 3 * n);
}
void orc_tracker_track_permaref(void* t, const float* pos, const float* colvar, int n, void* frame, const double refToFrame[7],
                                orc_track_result* out) {
  TrackerBox* b = (TrackerBox*)t;
  apply(b->p);
  Frame* fr = F(frame);
  std::vector<unsigned char> blank((size_t)fr->width(0) * fr->height(0), 0);
  Frame holder(-1, fr->width(0), fr->height(0), fr->K(0), 0.0, blank.data());
  install_permaref(&holder, pos, colvar, n);
  SE3 T = b->tr.trackFrameOnPermaref(&holder, fr, pose_in(refToFrame));
  fill_result(&b->tr, T, out);
  // ~Frame uses scalar delete on these arrays (C/DataStructures/Frame.cpp:91-94); hand them back here instead
  delete[] holder.permaRef_colorAndVarData; holder.permaRef_colorAndVarData = 0;
  delete[] holder.permaRef_posData; holder.permaRef_posData = 0;
}
float orc_tracker_check_overlap(void* t, const float* pos, int n, void* refFrame, const double refToFrame[7]) {
  TrackerBox* b = (TrackerBox*)t;
  apply(b->p);
  Frame* fr = F(refFrame);
  std::vector<unsigned char> blank((size_t)fr->width(0) * fr->height(0), 0);
  Frame holder(-1, fr->width(0), fr->height(0), fr->K(0), 0.0, blank.data());
  std::vector<float> cv((size_t)2 * n, 0.f);
  install_permaref(&holder, pos, cv.data(), n);
  float r = b->tr.checkPermaRefOverlap(&holder, pose_in(refToFrame));
  delete[] holder.permaRef_colorAndVarData; holder.permaRef_colorAndVarData = 0;
  delete[] holder.permaRef_posData; holder.permaRef_posData = 0;
  return r;
}

// ---- DepthMap ---------------------------------------------------------------------------------------
struct DepthBox {
  DepthMap dm;
  orc_params p;
  FramePtr kf_keep;
  std::vector<FramePtr> keep;
  float lastRescale = 1;
  DepthBox(int w, int h, const float K[4], const orc_params* pp) : dm(w, h, Kmat(K)), p(pp ? *pp : default_params_()) {}
};
static DepthMap* D(void* d) { return &((DepthBox*)d)->dm; }
void* orc_depth_create(int w, int h, const float K[4], const orc_params* p) { return new DepthBox(w, h, K, p); }
void orc_depth_destroy(void* d) {
  DepthBox* b = (DepthBox*)d;
  b->dm.invalidate();
  delete b;
}
void orc_depth_set_threads(void* d, int n) { ((DepthBox*)d)->p.multiThreading = n > 1; }   // the pool itself is fixed at MAPPING_THREADS = 4
void orc_depth_init_gt(void* d, void* f) {
  DepthBox* b = (DepthBox*)d; apply(b->p);
  b->kf_keep = ((FrameBox*)f)->f;
  b->dm.initializeFromGTDepth(F(f));
}
void orc_depth_init_random(void* d, void* f) {
  DepthBox* b = (DepthBox*)d; apply(b->p);
  b->kf_keep = ((FrameBox*)f)->f;
  b->dm.initializeRandomly(F(f));
}
void orc_depth_get(void* d, void* out32) {
  DepthMap* dm = D(d);
  static_assert(sizeof(DepthMapPixelHypothesis) == 32, "hypothesis must be 32 bytes");
  memcpy(out32, (void*)dm->currentDepthMap, 32 * (size_t)dm->width * dm->height);
}
#ifndef LSD_REF_HIP_BACKED
void orc_depth_set(void* d, void* kf, const void* in32, int reactivated) {
  DepthBox* b = (DepthBox*)d;
  DepthMap* dm = &b->dm;
  Frame* f = F(kf);
  if (dm->activeKeyFrame != f) {
    if (dm->activeKeyFrame != 0) dm->activeKeyFramelock.unlock();
    dm->activeKeyFramelock = f->getActiveLock();
    dm->activeKeyFrame = f;
    b->kf_keep = ((FrameBox*)kf)->f;
  }
  dm->activeKeyFrameImageData = f->image(0);
  dm->activeKeyFrameIsReactivated = reactivated != 0;
  memcpy((void*)dm->currentDepthMap, in32, 32 * (size_t)f->width(0) * f->height(0));
}
#else
void orc_depth_set(void*, void*, const void*, int) { fprintf(stderr, "orc_depth_set: not part of the hip-backed build\n"); abort(); }
#endif
static std::deque<FramePtr> frames_in(void** frames, int n) {
  std::deque<FramePtr> q;
  for (int i = 0; i < n; i++) q.push_back(((FrameBox*)frames[i])->f);
  return q;
}
void orc_depth_update(void* d, void** frames, int n) {
  DepthBox* b = (DepthBox*)d; apply(b->p);
  b->dm.updateKeyframe(frames_in(frames, n));
}
void orc_depth_create_keyframe(void* d, void* f) {
  DepthBox* b = (DepthBox*)d; apply(b->p);
  FramePtr old = b->kf_keep;            // keep the old keyframe alive until the switch is done
  b->dm.createKeyFrame(F(f));
  b->kf_keep = ((FrameBox*)f)->f;
  b->lastRescale = (float)F(f)->pose->thisToParent_raw.scale();   // sim3FromSE3(oldToNew^-1, rescaleFactor), DepthMap.cpp:1305
}
float orc_depth_last_rescale(void* d) { return ((DepthBox*)d)->lastRescale; }
void orc_depth_finalize(void* d) { DepthBox* b = (DepthBox*)d; apply(b->p); b->dm.finalizeKeyFrame(); }
void orc_depth_set_from_existing(void* d, void* f) {
  DepthBox* b = (DepthBox*)d; apply(b->p);
  FramePtr old = b->kf_keep;
  b->dm.setFromExistingKF(F(f));
  b->kf_keep = ((FrameBox*)f)->f;
}
void orc_frame_take_reactivation(void* f, void* d) { fresh(); F(f)->takeReActivationData(D(d)->currentDepthMap); }
void orc_frame_set_depth_from_map(void* f, void* d) { fresh(); F(f)->setDepth(D(d)->currentDepthMap); }

#ifndef LSD_REF_HIP_BACKED
// stage: 0 observe (needs frames), 1 fillHoles, 2 regularize(false,24), 3 regularize(true,24), 4 propagate(frames[0] = new KF)
void orc_depth_stage(void* d, int stage, void** frames, int n) {
  fresh();
  DepthBox* b = (DepthBox*)d; apply(b->p);
  DepthMap* dm = &b->dm;
  dm->activeKeyFrameImageData = dm->activeKeyFrame->image(0);
  if (stage == 0) {
    // the frame bookkeeping of DepthMap::updateKeyframe up to observeDepth() (C/DepthEstimation/DepthMap.cpp:1079-1126)
    std::deque<FramePtr> q = frames_in(frames, n);
    b->keep.assign(q.begin(), q.end());
    dm->oldest_referenceFrame = q.front().get();
    dm->newest_referenceFrame = q.back().get();
    dm->referenceFrameByID.clear();
    dm->referenceFrameByID_offset = dm->oldest_referenceFrame->id();
    for (FramePtr frame : q) {
      Sim3 refToKf;
      if (frame->pose->trackingParent->frameID == dm->activeKeyFrame->id())
        refToKf = frame->pose->thisToParent_raw;
      else
        refToKf = dm->activeKeyFrame->getScaledCamToWorld().inverse() * frame->getScaledCamToWorld();
      frame->prepareForStereoWith(dm->activeKeyFrame, refToKf, dm->K, 0);
      while ((int)dm->referenceFrameByID.size() + dm->referenceFrameByID_offset <= frame->id())
        dm->referenceFrameByID.push_back(frame.get());
    }
    dm->resetCounters();
    dm->observeDepth();
  } else if (stage == 1) dm->regularizeDepthMapFillHoles();
  else if (stage == 2) dm->regularizeDepthMap(false, VAL_SUM_MIN_FOR_KEEP);
  else if (stage == 3) dm->regularizeDepthMap(true, VAL_SUM_MIN_FOR_KEEP);
  else if (stage == 4) {
    FramePtr old = b->kf_keep;
    Frame* nk = F(frames[0]);
    dm->propagateDepth(nk);
    dm->activeKeyFramelock.unlock();
    dm->activeKeyFramelock = nk->getActiveLock();
    dm->activeKeyFrame = nk;
    dm->activeKeyFrameImageData = nk->image(0);
    dm->activeKeyFrameIsReactivated = false;
    b->kf_keep = ((FrameBox*)frames[0])->f;
  }
}

#else
void orc_depth_stage(void*, int, void**, int) { fprintf(stderr, "orc_depth_stage: not part of the hip-backed build\n"); abort(); }
#endif
double orc_now_seconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
const char* orc_ref_build_info() {
#if defined(LSD_REF_HIP_BACKED)
  return "reference headers + Frame / TrackingReference / Sim3Tracker sources, with SE3Tracker and DepthMap defined over liblsdhip.so (integration/hip_backed)";
#elif defined(ENABLE_SSE)
  return "reference sources (lsd_slam_core/src) compiled with -DENABLE_SSE against oracle/ref/shim stand-ins";
#else
  return "reference sources (lsd_slam_core/src) compiled without ENABLE_SSE against oracle/ref/shim stand-ins";
#endif
}

}  // extern "C"

#ifndef LSD_REF_HIP_BACKED
// Per-pixel stereo hook: makeAndCheckEPL + doLineStereo for pixel (x, y) against frame `ref` with the search interval the
// caller gives, nothing written to the map.  out: isGood, epx, epy, error, result_idepth, result_var, result_eplLength.
// (doLineStereo is `inline` in DepthMap.cpp; the build keeps an out-of-line copy with -fkeep-inline-functions.)
extern "C" void orc_depth_line_stereo(void* d, void* ref, int x, int y, float min_idepth, float prior_idepth, float max_idepth, float out[7]) {
  DepthBox* b = (DepthBox*)d; apply(b->p);
  DepthMap* dm = &b->dm;
  dm->activeKeyFrameImageData = dm->activeKeyFrame->image(0);
  Frame* rf = F(ref);
  RunningStats st;
  float epx = 0, epy = 0;
  bool good = dm->makeAndCheckEPL(x, y, rf, &epx, &epy, &st);
  out[0] = good; out[1] = epx; out[2] = epy; out[3] = out[4] = out[5] = out[6] = 0;
  if (!good) return;
  float ri = 0, rv = 0, rl = 0;
  out[3] = dm->doLineStereo((float)x, (float)y, epx, epy, min_idepth, prior_idepth, max_idepth, rf, rf->image(0), ri, rv, rl, &st);
  out[4] = ri; out[5] = rv; out[6] = rl;
}

#else
extern "C" void orc_depth_line_stereo(void*, void*, int, int, float, float, float, float*) { fprintf(stderr, "orc_depth_line_stereo: not part of the hip-backed build\n"); abort(); }
#endif
// ---- Sim3Tracker (C/Tracking/Sim3Tracker.{h,cpp}, compiled unchanged) --------------------------------------------------------
// Sim3 as double[8] = (qw, qx, qy, qz, tx, ty, tz, scale); the stand-in Sophus::Sim3d keeps (unit quaternion, scale, translation).
extern "C" {
struct orc_sim3_result {
  double frameToRef[8];
  float lastResidual, lastDepthResidual, lastPhotometricResidual, pointUsage, affine_a, affine_b;
  int diverged, numEvaluations;
  float hessian[49];
};
struct Sim3ResidualRec { float sumResD, sumResP; int numTermsD, numTermsP; float meanD, meanP, mean; };
struct Sim3EvalRecord {
  int warped_size;
  float pointUsage, affine_a_lastIt, affine_b_lastIt;
  Sim3ResidualRec res;
  float A[49], b[7];
  double num_constraints;
};
struct Sim3Box {
  Sim3Tracker tr;
  orc_params p;
  Sim3Box(int w, int h, const float K[4], const orc_params* pp) : tr(w, h, Kmat(K)), p(pp ? *pp : default_params_()) {}
};
static Sim3 sim3_in(const double p[8]) {
  Sim3 T;
  T.q.w = p[0]; T.q.x = p[1]; T.q.y = p[2]; T.q.z = p[3];     // bits as given (the oracle's hook does not re-normalise either)
  T.translation() = Eigen::Vector3d(p[4], p[5], p[6]);
  T.s = p[7];
  return T;
}
void* orc_sim3tracker_create(int w, int h, const float K[4], const orc_params* p) { return new Sim3Box(w, h, K, p); }
void orc_sim3tracker_destroy(void* t) { delete (Sim3Box*)t; }
void orc_sim3tracker_set_mode(void*, int) {}   // compile-time in the reference (ENABLE_SSE)
void orc_sim3tracker_set_max_its(void* t, const int its[5]) { for (int i = 0; i < 5; i++) ((Sim3Box*)t)->tr.settings.maxItsPerLvl[i] = its[i]; }
void orc_sim3tracker_track(void* t, void* ref, void* frame, const double init_frameToRef[8], int startLevel, int finalLevel, orc_sim3_result* out) {
  Sim3Box* b = (Sim3Box*)t;
  apply(b->p);
  Sim3Tracker* tr = &b->tr;
  Sim3 T = tr->trackFrameSim3(&((RefBox*)ref)->ref, F(frame), sim3_in(init_frameToRef), startLevel, finalLevel);
  out->frameToRef[0] = T.q.w; out->frameToRef[1] = T.q.x; out->frameToRef[2] = T.q.y; out->frameToRef[3] = T.q.z;
  out->frameToRef[4] = T.translation()[0]; out->frameToRef[5] = T.translation()[1]; out->frameToRef[6] = T.translation()[2];
  out->frameToRef[7] = T.s;
  out->lastResidual = tr->lastResidual; out->lastDepthResidual = tr->lastDepthResidual; out->lastPhotometricResidual = tr->lastPhotometricResidual;
  out->pointUsage = tr->pointUsage; out->affine_a = tr->affineEstimation_a; out->affine_b = tr->affineEstimation_b;
  out->diverged = tr->diverged; out->numEvaluations = -1;   // the reference does not count them
  for (int i = 0; i < 7; i++) for (int j = 0; j < 7; j++) out->hessian[i * 7 + j] = tr->lastSim3Hessian(i, j);
}
#if defined(ENABLE_SSE)
#define REF_CALL3(function, arguments) function##SSE arguments
#else
#define REF_CALL3(function, arguments) function arguments
#endif
// buffers + weights + LGS once at a fixed transformation: the three private members in the order trackFrameSim3 calls them
void orc_sim3tracker_evaluate(void* t, void* ref, void* frame, const double refToFrame[8], int level, float a, float b, Sim3EvalRecord* out) {
  Sim3Box* bx = (Sim3Box*)t;
  apply(bx->p);
  Sim3Tracker* tr = &bx->tr;
  TrackingReference* reference = &((RefBox*)ref)->ref;
  Sim3 T = sim3_in(refToFrame);
  tr->affineEstimation_a = a; tr->affineEstimation_b = b;
  reference->makePointCloud(level);
  tr->REF_CALL3(calcSim3Buffers, (reference, F(frame), T, level));
  Sim3ResidualStruct r = tr->REF_CALL3(calcSim3WeightsAndResidual, (T));
  LGS7 ls7;
  tr->REF_CALL3(calcSim3LGS, (ls7));
  out->warped_size = tr->buf_warped_size;
  out->pointUsage = tr->pointUsage;
  out->affine_a_lastIt = tr->affineEstimation_a_lastIt; out->affine_b_lastIt = tr->affineEstimation_b_lastIt;
  out->res.sumResD = r.sumResD; out->res.sumResP = r.sumResP; out->res.numTermsD = r.numTermsD; out->res.numTermsP = r.numTermsP;
  out->res.meanD = r.meanD; out->res.meanP = r.meanP; out->res.mean = r.mean;
  for (int i = 0; i < 7; i++) for (int j = 0; j < 7; j++) out->A[i * 7 + j] = ls7.A(i, j);
  for (int i = 0; i < 7; i++) out->b[i] = ls7.b[i];
  out->num_constraints = (double)ls7.num_constraints;
}
void orc_sim3_exp(const double a[7], double out[8]) {
  orc::Sim3d T = orc::sim3_exp(a);
  out[0] = T.q.w; out[1] = T.q.x; out[2] = T.q.y; out[3] = T.q.z; out[4] = T.t[0]; out[5] = T.t[1]; out[6] = T.t[2]; out[7] = T.s;
}
}  // extern "C"
