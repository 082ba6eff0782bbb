// ORACLE — TEST INFRASTRUCTURE ONLY (see lsd_oracle.hpp for what is pinned against oracle/_ref and what is not).
// Flat C entry points over the oracle classes so that tests/ and bench.py's cpu_baseline leg can drive it
// through ctypes.  Poses are double[7] = (qw,qx,qy,qz,tx,ty,tz); Sim3 adds the scale as an 8th element.
#include <chrono>
#include <cstring>
#include "lsd_oracle.hpp"
#include "orc_sim3.hpp"

using namespace orc;

extern "C" {

struct orc_params {
  float minUseGrad, cameraPixelNoise2, depthSmoothingFactor;
  int allowNegativeIdepths, useSubpixelStereo, multiThreading, useAffineLightningEstimation;
  float KFDistWeight, KFUsageWeight;
};

static Params to_params(const orc_params* p) {
  Params r;
  if (!p) return r;
  r.minUseGrad = p->minUseGrad; r.cameraPixelNoise2 = p->cameraPixelNoise2; r.depthSmoothingFactor = p->depthSmoothingFactor;
  r.allowNegativeIdepths = p->allowNegativeIdepths; r.useSubpixelStereo = p->useSubpixelStereo;
  r.multiThreading = p->multiThreading; r.useAffineLightningEstimation = p->useAffineLightningEstimation;
  r.KFDistWeight = p->KFDistWeight; r.KFUsageWeight = p->KFUsageWeight;
  return r;
}
void orc_default_params(orc_params* p) {
  Params d;
  p->minUseGrad = d.minUseGrad; p->cameraPixelNoise2 = d.cameraPixelNoise2; p->depthSmoothingFactor = d.depthSmoothingFactor;
  p->allowNegativeIdepths = d.allowNegativeIdepths; p->useSubpixelStereo = d.useSubpixelStereo;
  p->multiThreading = d.multiThreading; p->useAffineLightningEstimation = d.useAffineLightningEstimation;
  p->KFDistWeight = d.KFDistWeight; p->KFUsageWeight = d.KFUsageWeight;
}

// ---- Frame -------------------------------------------------------------------------------------
typedef std::shared_ptr<Frame> FramePtr;
void* orc_frame_create(int id, int w, int h, const float K[4], const unsigned char* img) {
  return new FramePtr(new Frame(id, w, h, K, img));
}
void orc_frame_destroy(void* f) { delete (FramePtr*)f; }
static Frame* F(void* f) { return ((FramePtr*)f)->get(); }

// what: 0 image, 1 gradients (4 floats/px), 2 maxGradients, 3 idepth, 4 idepthVar
int orc_frame_get(void* f, int what, int level, float* out) {
  Frame* fr = F(f);
  size_t n = (size_t)fr->width(level) * fr->height(level);
  const float* src = nullptr;
  switch (what) {
    case 0: src = fr->image(level); break;
    case 1: src = fr->gradients(level); n *= 4; break;
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
  out[0] = fr->fx[level]; out[1] = fr->fy[level]; out[2] = fr->cx[level]; out[3] = fr->cy[level];
  out[4] = fr->fxInv[level]; out[5] = fr->fyInv[level]; out[6] = fr->cxInv[level]; out[7] = fr->cyInv[level];
}
void orc_frame_set_sse_pyramid(void* f, int sse) { F(f)->sseImagePyramid = sse != 0; }
void orc_frame_set_depth_gt(void* f, const float* depth, float cov_scale, float minUseGrad) {
  F(f)->setDepthFromGroundTruth(depth, cov_scale, minUseGrad);
}
void orc_frame_set_depth_planes(void* f, const float* id, const float* var) { F(f)->setDepthPlanes(id, var); }
int orc_frame_get_wasgood(void* f, unsigned char* out) {
  Frame* fr = F(f);
  uint8_t* p = fr->refPixelWasGoodNoCreate();
  if (!p) return 0;
  memcpy(out, p, (size_t)fr->width(1) * fr->height(1));
  return 1;
}
void orc_frame_set_wasgood(void* f, const unsigned char* in) {
  Frame* fr = F(f);
  memcpy(fr->refPixelWasGood(), in, (size_t)fr->width(1) * fr->height(1));
}
void orc_frame_clear_wasgood(void* f) { F(f)->clear_refPixelWasGood(); }
// test hook (scene S3 generates gradient masks directly): overwrite the level-0 maxGradients plane
void orc_frame_set_maxgrad(void* f, const float* in) { F(f)->overrideMaxGradients(in); }
void orc_frame_set_pose(void* f, const double sim3[8], void* parent, float initialTrackedResidual) {
  Frame* fr = F(f);
  fr->thisToParent_raw.q.w = sim3[0]; fr->thisToParent_raw.q.x = sim3[1]; fr->thisToParent_raw.q.y = sim3[2];
  fr->thisToParent_raw.q.z = sim3[3];
  fr->thisToParent_raw.t = mk3<double>(sim3[4], sim3[5], sim3[6]);
  fr->thisToParent_raw.s = sim3[7];
  fr->trackingParent = parent ? F(parent) : nullptr;
  fr->initialTrackedResidual = initialTrackedResidual;
}
void orc_frame_get_pose(void* f, double sim3[8]) {
  Frame* fr = F(f);
  sim3[0] = fr->thisToParent_raw.q.w; sim3[1] = fr->thisToParent_raw.q.x; sim3[2] = fr->thisToParent_raw.q.y;
  sim3[3] = fr->thisToParent_raw.q.z;
  sim3[4] = fr->thisToParent_raw.t[0]; sim3[5] = fr->thisToParent_raw.t[1]; sim3[6] = fr->thisToParent_raw.t[2];
  sim3[7] = fr->thisToParent_raw.s;
}
// out: initialTrackedResidual, meanIdepth, numPoints, numFramesTrackedOnThis, numMappedOnThis, numMappedOnThisTotal,
//      depthHasBeenUpdatedFlag, numMappablePixels
void orc_frame_stats(void* f, float out[8]) {
  Frame* fr = F(f);
  out[0] = fr->initialTrackedResidual; out[1] = fr->meanIdepth; out[2] = (float)fr->numPoints;
  out[3] = (float)fr->numFramesTrackedOnThis; out[4] = (float)fr->numMappedOnThis; out[5] = (float)fr->numMappedOnThisTotal;
  out[6] = fr->depthHasBeenUpdatedFlag ? 1.f : 0.f; out[7] = fr->numMappablePixels;
}
void orc_frame_set_counters(void* f, int numFramesTrackedOnThis, int numMappedOnThis, int numMappedOnThisTotal, int depthUpdatedFlag) {
  Frame* fr = F(f);
  fr->numFramesTrackedOnThis = numFramesTrackedOnThis; fr->numMappedOnThis = numMappedOnThis;
  fr->numMappedOnThisTotal = numMappedOnThisTotal; fr->depthHasBeenUpdatedFlag = depthUpdatedFlag != 0;
}
// 27 stereo pre-computes after prepareForStereoWith: K_otherToThis_R[9], K_otherToThis_t[3], otherToThis_t[3],
// thisToOther_t[3], otherToThis_R_row0/1/2 [9]
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
void* orc_ref_create() { return new TrackingReference(); }
void orc_ref_destroy(void* r) { delete (TrackingReference*)r; }
void orc_ref_import(void* r, void* f) { ((TrackingReference*)r)->importFrame(F(f)); }
int orc_ref_pointcloud(void* r, int level, float* pos, float* colvar, float* grad, int* idx) {
  TrackingReference* ref = (TrackingReference*)r;
  ref->makePointCloud(level);
  int n = ref->numData[level];
  if (pos) memcpy(pos, ref->posData[level].data(), sizeof(float) * 3 * n);
  if (colvar) memcpy(colvar, ref->colorAndVarData[level].data(), sizeof(float) * 2 * n);
  if (grad) memcpy(grad, ref->gradData[level].data(), sizeof(float) * 2 * n);
  if (idx) memcpy(idx, ref->pointPosInXYGrid[level].data(), sizeof(int) * n);
  return n;
}

// ---- SE3Tracker -----------------------------------------------------------------------------------
struct orc_track_result {
  double frameToRef[7];
  float pointUsage, lastGoodCount, lastBadCount, lastMeanRes, lastResidual, affine_a, affine_b;
  int diverged, trackingWasGood, numEvaluations, numWarpUpdates;
};
void* orc_tracker_create(int w, int h, const float K[4], const orc_params* p) { return new SE3Tracker(w, h, K, to_params(p)); }
void orc_tracker_destroy(void* t) { delete (SE3Tracker*)t; }
// instrumentation: cumulative evaluations / reference points / in-image points per pyramid level
void orc_tracker_level_stats(void* t, long long out[15]) {
  SE3Tracker* tr = (SE3Tracker*)t;
  for (int l = 0; l < 5; l++) { out[l] = tr->levelEvaluations[l]; out[5 + l] = tr->levelPoints[l]; out[10 + l] = tr->levelWarped[l]; }
}
void orc_tracker_set_huber(void* t, float huber_d) { ((SE3Tracker*)t)->settings.huber_d = huber_d; }
void orc_tracker_set_mode(void* t, int mode) { ((SE3Tracker*)t)->mode = (TrackerMode)mode; }
void orc_tracker_set_max_its(void* t, const int its[5]) {
  for (int i = 0; i < 5; i++) ((SE3Tracker*)t)->settings.maxItsPerLvl[i] = its[i];
}
static SE3d pose_in(const double p[7]) {
  SE3d T;
  T.q.w = p[0]; T.q.x = p[1]; T.q.y = p[2]; T.q.z = p[3];
  T.t = mk3<double>(p[4], p[5], p[6]);
  return T;
}
static void pose_out(const SE3d& T, double p[7]) {
  p[0] = T.q.w; p[1] = T.q.x; p[2] = T.q.y; p[3] = T.q.z; p[4] = T.t[0]; p[5] = T.t[1]; p[6] = T.t[2];
}
static void fill_result(SE3Tracker* tr, const SE3d& T, orc_track_result* out) {
  pose_out(T, out->frameToRef);
  out->pointUsage = tr->pointUsage; out->lastGoodCount = tr->lastGoodCount; out->lastBadCount = tr->lastBadCount;
  out->lastMeanRes = tr->lastMeanRes; out->lastResidual = tr->lastResidual;
  out->affine_a = tr->affineEstimation_a; out->affine_b = tr->affineEstimation_b;
  out->diverged = tr->diverged; out->trackingWasGood = tr->trackingWasGood;
  out->numEvaluations = tr->numEvaluations; out->numWarpUpdates = tr->numWarpUpdates;
}
void orc_tracker_track(void* t, void* ref, void* frame, const double init_frameToRef[7], orc_track_result* out) {
  SE3Tracker* tr = (SE3Tracker*)t;
  tr->numEvaluations = 0; tr->numWarpUpdates = 0;
  SE3d T = tr->trackFrame((TrackingReference*)ref, F(frame), pose_in(init_frameToRef));
  fill_result(tr, T, out);
}
// referenceToFrame as float[7]; evaluates K1+K2+K3 once at that pose
void orc_tracker_evaluate(void* t, void* ref, void* frame, const float refToFrame[7], int level, float a, float b,
                          ResidualRecord* out) {
  SE3f T;
  T.q.w = refToFrame[0]; T.q.x = refToFrame[1]; T.q.y = refToFrame[2]; T.q.z = refToFrame[3];
  T.t = mk3<float>(refToFrame[4], refToFrame[5], refToFrame[6]);
  ((SE3Tracker*)t)->evaluate((TrackingReference*)ref, F(frame), T, level, a, b, out);
}
// which: 0 x 1 y 2 z 3 dx 4 dy 5 residual 6 d 7 idepthVar 8 weight_p ; returns buf_warped_size
int orc_tracker_buffer(void* t, int which, float* out) {
  SE3Tracker* tr = (SE3Tracker*)t;
  float* bufs[9] = {tr->buf_warped_x, tr->buf_warped_y, tr->buf_warped_z, tr->buf_warped_dx, tr->buf_warped_dy,
                    tr->buf_warped_residual, tr->buf_d, tr->buf_idepthVar, tr->buf_weight_p};
  if (out) memcpy(out, bufs[which], sizeof(float) * tr->buf_warped_size);
  return tr->buf_warped_size;
}
void orc_tracker_track_permaref(void* t, const float* pos, const float* colvar, int n, void* frame, const double refToFrame[7],
                                orc_track_result* out) {
  SE3Tracker* tr = (SE3Tracker*)t;
  tr->numEvaluations = 0; tr->numWarpUpdates = 0;
  SE3d T = tr->trackFrameOnPermaref(pos, colvar, n, F(frame), pose_in(refToFrame));
  fill_result(tr, T, out);
}
float orc_tracker_check_overlap(void* t, const float* pos, int n, void* refFrame, const double refToFrame[7]) {
  return ((SE3Tracker*)t)->checkPermaRefOverlap(pos, n, F(refFrame), pose_in(refToFrame));
}

// SE3 helpers for tests (Sophus property tests, pose distances)
void orc_se3_exp_f(const float a[6], float out[7]) {
  SE3f T = se3_exp<float>(a);
  out[0] = T.q.w; out[1] = T.q.x; out[2] = T.q.y; out[3] = T.q.z; out[4] = T.t[0]; out[5] = T.t[1]; out[6] = T.t[2];
}
void orc_se3_exp_d(const double a[6], double out[7]) { pose_out(se3_exp<double>(a), out); }
void orc_se3_log_d(const double p[7], double out[6]) { se3_log<double>(pose_in(p), out); }
void orc_se3_mul_d(const double a[7], const double b[7], double out[7]) { pose_out(pose_in(a) * pose_in(b), out); }
void orc_se3_inv_d(const double a[7], double out[7]) { pose_out(pose_in(a).inverse(), out); }
void orc_ldlt6_solve(const float A[36], const float b[6], float x[6]) { ldlt6_solve(A, b, x); }

// ---- DepthMap ---------------------------------------------------------------------------------------
void* orc_depth_create(int w, int h, const float K[4], const orc_params* p) { return new DepthMap(w, h, K, to_params(p)); }
void orc_depth_destroy(void* d) { delete (DepthMap*)d; }
void orc_depth_set_threads(void* d, int n) { ((DepthMap*)d)->numThreads = n; }
void orc_depth_init_gt(void* d, void* f) { ((DepthMap*)d)->initializeFromGTDepth(F(f)); }
void orc_depth_init_random(void* d, void* f) { ((DepthMap*)d)->initializeRandomly(F(f)); }
void orc_depth_get(void* d, void* out32) {
  DepthMap* dm = (DepthMap*)d;
  Frame* kf = dm->activeKeyFrame;
  memcpy(out32, (void*)dm->currentDepthMap, 32 * (size_t)kf->width(0) * kf->height(0));
}
// raw overwrite of the current map + active keyframe (kernel-level tests start from arbitrary states)
void orc_depth_set(void* d, void* kf, const void* in32, int reactivated) {
  DepthMap* dm = (DepthMap*)d;
  Frame* f = F(kf);
  dm->activeKeyFrame = f;
  dm->activeKeyFrameIsReactivated = reactivated != 0;
  memcpy((void*)dm->currentDepthMap, in32, 32 * (size_t)f->width(0) * f->height(0));
  // private activeKeyFrameImageData is refreshed by the stage entry points below
}
static std::deque<FramePtr> frames_in(void** frames, int n) {
  std::deque<FramePtr> q;
  for (int i = 0; i < n; i++) q.push_back(*(FramePtr*)frames[i]);
  return q;
}
void orc_depth_update(void* d, void** frames, int n) { ((DepthMap*)d)->updateKeyframe(frames_in(frames, n)); }
void orc_depth_create_keyframe(void* d, void* f) { ((DepthMap*)d)->createKeyFrame(F(f)); }
float orc_depth_last_rescale(void* d) { return ((DepthMap*)d)->lastRescaleFactor; }
void orc_depth_finalize(void* d) { ((DepthMap*)d)->finalizeKeyFrame(); }
void orc_depth_set_from_existing(void* d, void* f) { ((DepthMap*)d)->setFromExistingKF(F(f)); }
void orc_frame_take_reactivation(void* f, void* d) { F(f)->takeReActivationData(((DepthMap*)d)->currentDepthMap); }
void orc_frame_set_depth_from_map(void* f, void* d) { F(f)->setDepth(((DepthMap*)d)->currentDepthMap); }

}  // extern "C"

extern "C" {
// stage: 0 observe (needs frames), 1 fillHoles, 2 regularize(false,24), 3 regularize(true,24), 4 propagate(frames[0] = new KF)
void orc_depth_stage(void* d, int stage, void** frames, int n) {
  DepthMap* dm = (DepthMap*)d;
  // refresh the cached keyframe image pointer the stages rely on
  dm->refreshActiveKeyFrameImage();
  if (stage == 0) {
    std::deque<FramePtr> q = frames_in(frames, n);
    dm->setReferenceFrames(q);
    dm->observeDepth();
  } else if (stage == 1) dm->regularizeDepthMapFillHoles();
  else if (stage == 2) dm->regularizeDepthMap(false, 24);
  else if (stage == 3) dm->regularizeDepthMap(true, 24);
  else if (stage == 4) {
    Frame* nk = F(frames[0]);
    dm->propagateDepth(nk);
    dm->activeKeyFrame = nk;
    dm->refreshActiveKeyFrameImage();
  }
}

// ---- Sim3Tracker ----------------------------------------------------------------------------------
// Sim3 as double[8] = (qw, qx, qy, qz, tx, ty, tz, scale)
struct orc_sim3_result {
  double frameToRef[8];
  float lastResidual, lastDepthResidual, lastPhotometricResidual, pointUsage, affine_a, affine_b;
  int diverged, numEvaluations;
  float hessian[49];
};
static Sim3d sim3_in(const double p[8]) {
  Sim3d T;
  T.q.w = p[0]; T.q.x = p[1]; T.q.y = p[2]; T.q.z = p[3];
  T.t = mk3<double>(p[4], p[5], p[6]);
  T.s = p[7];
  return T;
}
void* orc_sim3tracker_create(int w, int h, const float K[4], const orc_params* p) { return new Sim3Tracker(w, h, K, to_params(p)); }
void orc_sim3tracker_destroy(void* t) { delete (Sim3Tracker*)t; }
void orc_sim3tracker_set_mode(void* t, int mode) { ((Sim3Tracker*)t)->mode = (TrackerMode)mode; }
void orc_sim3tracker_set_max_its(void* t, const int its[5]) { for (int i = 0; i < 5; i++) ((Sim3Tracker*)t)->settings.maxItsPerLvl[i] = its[i]; }
void orc_sim3tracker_track(void* t, void* ref, void* frame, const double init_frameToRef[8], int startLevel, int finalLevel, orc_sim3_result* out) {
  Sim3Tracker* tr = (Sim3Tracker*)t;
  Sim3d T = tr->trackFrameSim3((TrackingReference*)ref, F(frame), sim3_in(init_frameToRef), startLevel, finalLevel);
  out->frameToRef[0] = T.q.w; out->frameToRef[1] = T.q.x; out->frameToRef[2] = T.q.y; out->frameToRef[3] = T.q.z;
  out->frameToRef[4] = T.t[0]; out->frameToRef[5] = T.t[1]; out->frameToRef[6] = T.t[2]; out->frameToRef[7] = T.s;
  out->lastResidual = tr->lastResidual; out->lastDepthResidual = tr->lastDepthResidual; out->lastPhotometricResidual = tr->lastPhotometricResidual;
  out->pointUsage = tr->pointUsage; out->affine_a = tr->affineEstimation_a; out->affine_b = tr->affineEstimation_b;
  out->diverged = tr->diverged; out->numEvaluations = tr->numEvaluations;
  memcpy(out->hessian, tr->lastSim3Hessian, sizeof(out->hessian));
}
void orc_sim3tracker_evaluate(void* t, void* ref, void* frame, const double refToFrame[8], int level, float a, float b, Sim3EvalRecord* out) {
  ((Sim3Tracker*)t)->evaluate((TrackingReference*)ref, F(frame), sim3_in(refToFrame), level, a, b, out);
}
void orc_sim3_exp(const double a[7], double out[8]) {
  Sim3d T = sim3_exp(a);
  out[0] = T.q.w; out[1] = T.q.x; out[2] = T.q.y; out[3] = T.q.z; out[4] = T.t[0]; out[5] = T.t[1]; out[6] = T.t[2]; out[7] = T.s;
}

double orc_now_seconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // extern "C"

// Per-pixel stereo hook (same contract as oracle/ref/ref_capi.cpp orc_depth_line_stereo)
extern "C" void orc_depth_line_stereo(void* d, void* ref, int x, int y, float min_idepth, float prior_idepth, float max_idepth, float out[7]) {
  orc::DepthMap* dm = (orc::DepthMap*)d;
  dm->refreshActiveKeyFrameImage();
  orc::Frame* rf = F(ref);
  float epx = 0, epy = 0;
  bool good = dm->makeAndCheckEPL(x, y, rf, &epx, &epy);
  out[0] = good; out[1] = epx; out[2] = epy; out[3] = out[4] = out[5] = out[6] = 0;
  if (!good) return;
  float ri = 0, rv = 0, rl = 0;
  out[3] = dm->doLineStereo((float)x, (float)y, epx, epy, min_idepth, prior_idepth, max_idepth, rf, rf->image(0), ri, rv, rl);
  out[4] = ri; out[5] = rv; out[6] = rl;
}

namespace orc { extern long long orc_walk_hist[64]; }
// instrumentation: histogram of epipolar-walk lengths (steps) since the last call with reset != 0
extern "C" void orc_walk_histogram(long long out[64], int reset) {
  for (int i = 0; i < 64; i++) { out[i] = orc::orc_walk_hist[i]; if (reset) orc::orc_walk_hist[i] = 0; }
}
