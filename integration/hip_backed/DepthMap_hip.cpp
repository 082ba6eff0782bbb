// lsd_slam::DepthMap (lsd_slam_core/src/DepthEstimation/DepthMap.h:47-98, header unmodified) with its members defined over the C ABI of
// liblsdhip.so.  Replaces lsd_slam_core/src/DepthEstimation/DepthMap.cpp in the build (integration/lsd_slam_core.patch); see
// lsd_hip_binding.hpp.  The hypothesis map lives on the device; after every call the 32-byte AoS copy `currentDepthMap` on the host is
// refreshed (lsdhip_depth_download), because the host Frame keeps its own depth planes in this binding: the reference's side effects
//   activeKeyFrame->setDepth(currentDepthMap)            DepthMap.cpp:1148-1154, :1311, :1385
//   numMappedOnThis++ / numMappedOnThisTotal++           :1165-1166
//   prepareForStereoWith on every reference frame        :1101
//   new keyframe pose = sim3FromSE3(oldToNew^-1, rescaleFactor) + invalidateCache()   :1305-1306
//   calculateMeanInformation / takeReActivationData      :1386-1387
//   the shared lock on the active keyframe               :885, :924, :969, :1256
// are carried out on the HOST frames with the reference's own Frame methods, fed from what the device computed.
#include <map>
#include <mutex>

#include "lsd_hip_binding.hpp"

#include "DepthEstimation/DepthMap.h"
#include "DepthEstimation/DepthMapPixelHypothesis.h"

namespace lsd_slam {
namespace {

struct MapState { lsdhip_ctx* ctx = nullptr; lsdhip_depthmap* h = nullptr; };
std::mutex g_mu;
std::map<const DepthMap*, MapState> g_state;
MapState& state_of(const DepthMap* d) {
  std::lock_guard<std::mutex> lock(g_mu);
  return g_state[d];
}
static_assert(sizeof(DepthMapPixelHypothesis) == sizeof(lsdhip_hypothesis), "the exchange format is the reference's 32-byte hypothesis");

// the host keyframe's counters and flag to its mirror before a call, and back afterwards: SlamSystem edits them between calls
// (depthHasBeenUpdatedFlag = false at SlamSystem.cpp:910)
void counters_to_device(Frame* kf, lsdhip_frame* kfh) {
  lsd_slam_hipbind::check(lsdhip_frame_set_counters(kfh, kf->numFramesTrackedOnThis, kf->numMappedOnThis, kf->numMappedOnThisTotal,
                                                    kf->depthHasBeenUpdatedFlag ? 1 : 0), "lsdhip_frame_set_counters");
}

}  // namespace

DepthMap::DepthMap(int w, int h, const Eigen::Matrix3f& K) {
  width = w; height = h;
  this->K = K;
  fx = K(0, 0); fy = K(1, 1); cx = K(0, 2); cy = K(1, 2);
  KInv = K.inverse();
  fxi = KInv(0, 0); fyi = KInv(1, 1); cxi = KInv(0, 2); cyi = KInv(1, 2);
  activeKeyFrame = 0;
  activeKeyFrameIsReactivated = false;
  activeKeyFrameImageData = 0;
  oldest_referenceFrame = newest_referenceFrame = 0;
  referenceFrameByID_offset = 0;
  // host copy of the map in the reference's layout (readers: Frame::setDepth, Frame::takeReActivationData, debug plots); the working
  // buffers of the CPU implementation (otherDepthMap, the validity integral) have no use here
  currentDepthMap = new DepthMapPixelHypothesis[(size_t)w * h];
  otherDepthMap = 0;
  validityIntegralBuffer = 0;
  msUpdate = msCreate = msFinalize = 0; msObserve = msRegularize = msPropagate = msFillHoles = msSetDepth = 0;
  nUpdate = nCreate = nFinalize = 0; nObserve = nRegularize = nPropagate = nFillHoles = nSetDepth = 0;
  nAvgUpdate = nAvgCreate = nAvgFinalize = 0; nAvgObserve = nAvgRegularize = nAvgPropagate = nAvgFillHoles = nAvgSetDepth = 0;
  gettimeofday(&lastHzUpdate, NULL);
  MapState& S = state_of(this);
  S.ctx = lsd_slam_hipbind::context_for(w, h, K);
  lsd_slam_hipbind::check(lsdhip_depth_create(S.ctx, &S.h), "lsdhip_depth_create");
  reset();
}

DepthMap::~DepthMap() {
  if (activeKeyFrame != 0) activeKeyFramelock.unlock();
  delete[] currentDepthMap;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_state.find(this);
  if (it != g_state.end()) { lsdhip_depth_destroy(it->second.h); g_state.erase(it); }
}

void DepthMap::reset() {
  MapState& S = state_of(this);
  lsd_slam_hipbind::check(lsdhip_depth_reset(S.h), "lsdhip_depth_reset");
  lsd_slam_hipbind::check(lsdhip_depth_download(S.h, (lsdhip_hypothesis*)currentDepthMap), "lsdhip_depth_download");
}

void DepthMap::invalidate() {
  if (activeKeyFrame == 0) return;
  lsd_slam_hipbind::check(lsdhip_depth_invalidate(state_of(this).h), "lsdhip_depth_invalidate");
  activeKeyFrame = 0;
  activeKeyFramelock.unlock();
}

void DepthMap::initializeFromGTDepth(Frame* new_frame) {
  MapState& S = state_of(this);
  activeKeyFramelock = new_frame->getActiveLock();
  activeKeyFrame = new_frame;
  activeKeyFrameImageData = activeKeyFrame->image(0);
  activeKeyFrameIsReactivated = false;
  lsd_slam_hipbind::sync_depth(new_frame, S.ctx);            // the ground-truth idepth / idepthVar planes Frame::setDepthFromGroundTruth left
  lsdhip_frame* kfh = lsd_slam_hipbind::device_frame(new_frame, S.ctx);
  lsd_slam_hipbind::check(lsdhip_depth_init_gt(S.h, kfh), "lsdhip_depth_init_gt");
  lsd_slam_hipbind::check(lsdhip_depth_download(S.h, (lsdhip_hypothesis*)currentDepthMap), "lsdhip_depth_download");
  activeKeyFrame->setDepth(currentDepthMap);
  lsd_slam_hipbind::depth_in_sync(activeKeyFrame);
}

void DepthMap::initializeRandomly(Frame* new_frame) {
  // (the reference draws from rand(): only the distribution is reproducible, DepthMap.cpp:883-916)
  MapState& S = state_of(this);
  activeKeyFramelock = new_frame->getActiveLock();
  activeKeyFrame = new_frame;
  activeKeyFrameImageData = activeKeyFrame->image(0);
  activeKeyFrameIsReactivated = false;
  lsdhip_frame* kfh = lsd_slam_hipbind::device_frame(new_frame, S.ctx);
  lsd_slam_hipbind::check(lsdhip_depth_init_random(S.h, kfh), "lsdhip_depth_init_random");
  lsd_slam_hipbind::check(lsdhip_depth_download(S.h, (lsdhip_hypothesis*)currentDepthMap), "lsdhip_depth_download");
  activeKeyFrame->setDepth(currentDepthMap);
  lsd_slam_hipbind::depth_in_sync(activeKeyFrame);
}

void DepthMap::setFromExistingKF(Frame* kf) {
  MapState& S = state_of(this);
  activeKeyFramelock = kf->getActiveLock();
  activeKeyFrame = kf;
  activeKeyFrameImageData = activeKeyFrame->image(0);
  activeKeyFrameIsReactivated = true;
  // the re-activation data stayed with the keyframe's mirror when it was finalised on the device (lsdhip_depth_finalize)
  lsdhip_frame* kfh = lsd_slam_hipbind::device_frame(kf, S.ctx);
  lsd_slam_hipbind::check(lsdhip_depth_set_from_existing(S.h, kfh), "lsdhip_depth_set_from_existing");
  lsd_slam_hipbind::check(lsdhip_depth_download(S.h, (lsdhip_hypothesis*)currentDepthMap), "lsdhip_depth_download");
  activeKeyFrame->setDepth(currentDepthMap);
  lsd_slam_hipbind::depth_in_sync(activeKeyFrame);
}

void DepthMap::updateKeyframe(std::deque<std::shared_ptr<Frame>> referenceFrames) {
  MapState& S = state_of(this);
  struct timeval t0, t1;
  gettimeofday(&t0, NULL);
  oldest_referenceFrame = referenceFrames.front().get();
  newest_referenceFrame = referenceFrames.back().get();
  lsdhip_frame* kfh = lsd_slam_hipbind::device_frame(activeKeyFrame, S.ctx);
  std::vector<lsdhip_frame*> refs;
  for (const std::shared_ptr<Frame>& frame : referenceFrames) {
    // frame -> keyframe similarity: the frame's own tracking result, or through the pose graph when it was tracked elsewhere
    Sim3 refToKf;
    if (frame->pose->trackingParent->frameID == activeKeyFrame->id()) refToKf = frame->pose->thisToParent_raw;
    else refToKf = activeKeyFrame->getScaledCamToWorld().inverse() * frame->getScaledCamToWorld();
    frame->prepareForStereoWith(activeKeyFrame, refToKf, K, 0);        // (host-side pre-computes: other host readers may rely on them)
    lsdhip_frame* fh = lsd_slam_hipbind::device_frame(frame.get(), S.ctx);
    double p8[8];
    lsd_slam_hipbind::sim3_to8(refToKf, p8);
    lsd_slam_hipbind::push_pose(frame.get(), fh, p8, activeKeyFrame, kfh, frame->initialTrackedResidual);
    lsd_slam_hipbind::push_mask(frame.get(), fh);
    refs.push_back(fh);
  }
  counters_to_device(activeKeyFrame, kfh);
  lsd_slam_hipbind::check(lsdhip_depth_update(S.h, refs.data(), (int)refs.size()), "lsdhip_depth_update");
  lsd_slam_hipbind::check(lsdhip_depth_download(S.h, (lsdhip_hypothesis*)currentDepthMap), "lsdhip_depth_download");
  if (!activeKeyFrame->depthHasBeenUpdatedFlag) {
    activeKeyFrame->setDepth(currentDepthMap);
    lsd_slam_hipbind::depth_in_sync(activeKeyFrame);
    nSetDepth++;
  }
  activeKeyFrame->numMappedOnThis++;
  activeKeyFrame->numMappedOnThisTotal++;
  gettimeofday(&t1, NULL);
  msUpdate = 0.9f * msUpdate + 0.1f * ((t1.tv_sec - t0.tv_sec) * 1000.0f + (t1.tv_usec - t0.tv_usec) / 1000.0f);
  nUpdate++; nObserve++; nRegularize++; nFillHoles++;
}

void DepthMap::createKeyFrame(Frame* new_keyframe) {
  MapState& S = state_of(this);
  struct timeval t0, t1;
  gettimeofday(&t0, NULL);
  boost::shared_lock<boost::shared_mutex> lock2 = new_keyframe->getActiveLock();
  lsdhip_frame* oldh = lsd_slam_hipbind::device_frame(activeKeyFrame, S.ctx);
  lsdhip_frame* nkh = lsd_slam_hipbind::device_frame(new_keyframe, S.ctx);
  double p8[8];
  lsd_slam_hipbind::sim3_to8(new_keyframe->pose->thisToParent_raw, p8);
  lsd_slam_hipbind::push_pose(new_keyframe, nkh, p8, activeKeyFrame, oldh, new_keyframe->initialTrackedResidual);
  lsd_slam_hipbind::push_mask(new_keyframe, nkh);          // propagateDepth consults the new keyframe's refPixelWasGood (DepthMap.cpp:540-547)
  float rescaleFactor = 1;
  lsd_slam_hipbind::check(lsdhip_depth_create_keyframe(S.h, nkh, &rescaleFactor), "lsdhip_depth_create_keyframe");
  lsd_slam_hipbind::check(lsdhip_depth_download(S.h, (lsdhip_hypothesis*)currentDepthMap), "lsdhip_depth_download");
  activeKeyFrame = new_keyframe;
  activeKeyFramelock = activeKeyFrame->getActiveLock();
  activeKeyFrameImageData = new_keyframe->image(0);
  activeKeyFrameIsReactivated = false;
  // the new keyframe's pose relative to the old one, carrying the rescale factor as its Sim3 scale: as the device frame now has it
  lsd_slam_hipbind::check(lsdhip_frame_get_pose(nkh, p8), "lsdhip_frame_get_pose");
  activeKeyFrame->pose->thisToParent_raw = lsd_slam_hipbind::sim3_from8(p8);
  activeKeyFrame->pose->invalidateCache();
  lsd_slam_hipbind::sim3_to8(activeKeyFrame->pose->thisToParent_raw, p8);
  lsd_slam_hipbind::note_pose(activeKeyFrame, p8, 0, activeKeyFrame->initialTrackedResidual);
  activeKeyFrame->setDepth(currentDepthMap);
  lsd_slam_hipbind::depth_in_sync(activeKeyFrame);
  gettimeofday(&t1, NULL);
  msCreate = 0.9f * msCreate + 0.1f * ((t1.tv_sec - t0.tv_sec) * 1000.0f + (t1.tv_usec - t0.tv_usec) / 1000.0f);
  nCreate++; nPropagate++; nSetDepth++;
}

void DepthMap::finalizeKeyFrame() {
  MapState& S = state_of(this);
  struct timeval t0, t1;
  gettimeofday(&t0, NULL);
  lsdhip_frame* kfh = lsd_slam_hipbind::device_frame(activeKeyFrame, S.ctx);
  counters_to_device(activeKeyFrame, kfh);
  lsd_slam_hipbind::check(lsdhip_depth_finalize(S.h), "lsdhip_depth_finalize");
  lsd_slam_hipbind::check(lsdhip_depth_download(S.h, (lsdhip_hypothesis*)currentDepthMap), "lsdhip_depth_download");
  activeKeyFrame->setDepth(currentDepthMap);
  activeKeyFrame->calculateMeanInformation();
  activeKeyFrame->takeReActivationData(currentDepthMap);
  lsd_slam_hipbind::depth_in_sync(activeKeyFrame);
  gettimeofday(&t1, NULL);
  msFinalize = 0.9f * msFinalize + 0.1f * ((t1.tv_sec - t0.tv_sec) * 1000.0f + (t1.tv_usec - t0.tv_usec) / 1000.0f);
  nFinalize++; nSetDepth++;
}

int DepthMap::debugPlotDepthMap() { return 0; }        // the debug images of the CPU implementation are not produced
void DepthMap::addTimingSample() {
  struct timeval now;
  gettimeofday(&now, NULL);
  const float dt = (now.tv_sec - lastHzUpdate.tv_sec) + (now.tv_usec - lastHzUpdate.tv_usec) / 1000000.0f;
  if (dt > 2) {
    nAvgUpdate = 0.8f * nAvgUpdate + 0.2f * (nUpdate / dt); nUpdate = 0;
    nAvgCreate = 0.8f * nAvgCreate + 0.2f * (nCreate / dt); nCreate = 0;
    nAvgFinalize = 0.8f * nAvgFinalize + 0.2f * (nFinalize / dt); nFinalize = 0;
    nAvgObserve = 0.8f * nAvgObserve + 0.2f * (nObserve / dt); nObserve = 0;
    nAvgRegularize = 0.8f * nAvgRegularize + 0.2f * (nRegularize / dt); nRegularize = 0;
    nAvgPropagate = 0.8f * nAvgPropagate + 0.2f * (nPropagate / dt); nPropagate = 0;
    nAvgFillHoles = 0.8f * nAvgFillHoles + 0.2f * (nFillHoles / dt); nFillHoles = 0;
    nAvgSetDepth = 0.8f * nAvgSetDepth + 0.2f * (nSetDepth / dt); nSetDepth = 0;
    lastHzUpdate = now;
  }
}

}  // namespace lsd_slam
