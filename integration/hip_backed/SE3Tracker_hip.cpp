// lsd_slam::SE3Tracker (lsd_slam_core/src/Tracking/SE3Tracker.h:41-93, header unmodified) with its members defined over the C ABI of
// liblsdhip.so.  Replaces lsd_slam_core/src/Tracking/SE3Tracker.cpp in the build (integration/lsd_slam_core.patch); see
// lsd_hip_binding.hpp.  Public behaviour per SURVEY.md section 8(b): the signatures, the public result fields, and the side effects
//   frame->refPixelWasGood() (level 1), frame->initialTrackedResidual, frame->pose->thisToParent_raw / trackingParent,
//   reference->keyframe->numFramesTrackedOnThis++ when trackingWasGood, identity + diverged on divergence (SE3Tracker.cpp:324-329, :479-485).
#include <map>
#include <mutex>

#include "lsd_hip_binding.hpp"

#include "Tracking/SE3Tracker.h"
#include "Tracking/TrackingReference.h"

namespace lsd_slam {
namespace {

// the device objects of a tracker: the header has no room for them
struct TrackerState { lsdhip_ctx* ctx = nullptr; lsdhip_tracker* h = nullptr; };
std::mutex g_mu;
std::map<const SE3Tracker*, TrackerState> g_state;

TrackerState& state_of(const SE3Tracker* t) {
  std::lock_guard<std::mutex> lock(g_mu);
  return g_state[t];
}
// `settings` is a public plain-data member that callers edit (SlamSystem.cpp:80-81): handed to the device before every job
void push_settings(const SE3Tracker* t, lsdhip_tracker* h) {
  lsdhip_tracker_settings st;
  const DenseDepthTrackerSettings& s = t->settings;
  st.lambdaSuccessFac = s.lambdaSuccessFac; st.lambdaFailFac = s.lambdaFailFac;
  for (int l = 0; l < LSDHIP_PYRAMID_LEVELS; l++) {
    st.lambdaInitial[l] = s.lambdaInitial[l]; st.stepSizeMin[l] = s.stepSizeMin[l];
    st.convergenceEps[l] = s.convergenceEps[l]; st.maxItsPerLvl[l] = s.maxItsPerLvl[l];
  }
  st.lambdaInitialTestTrack = s.lambdaInitialTestTrack; st.stepSizeMinTestTrack = s.stepSizeMinTestTrack;
  st.convergenceEpsTestTrack = s.convergenceEpsTestTrack; st.maxItsTestTrack = s.maxItsTestTrack;
  st.huber_d = s.huber_d; st.var_weight = s.var_weight;
  lsd_slam_hipbind::check(lsdhip_tracker_set_settings(h, &st), "lsdhip_tracker_set_settings");
}
void publish(SE3Tracker* t, const lsdhip_track_result& r) {
  t->pointUsage = r.pointUsage; t->lastGoodCount = r.lastGoodCount; t->lastBadCount = r.lastBadCount; t->lastMeanRes = r.lastMeanRes;
  t->lastResidual = r.lastResidual; t->affineEstimation_a = r.affineEstimation_a; t->affineEstimation_b = r.affineEstimation_b;
  t->diverged = r.diverged != 0; t->trackingWasGood = r.trackingWasGood != 0;
}
SE3 pose_of(const double p[7]) { return lsd_slam_hip::fromHip<SE3>(lsd_slam_hip::SE3::from7(p)); }
void pose_to(const SE3& T, double p[7]) { lsd_slam_hip::toHip(T).to7(p); }

}  // namespace

SE3Tracker::SE3Tracker(int w, int h, Eigen::Matrix3f K) {
  width = w; height = h;
  this->K = K;
  fx = K(0, 0); fy = K(1, 1); cx = K(0, 2); cy = K(1, 2);
  settings = DenseDepthTrackerSettings();
  KInv = K.inverse();
  fxi = KInv(0, 0); fyi = KInv(1, 1); cxi = KInv(0, 2); cyi = KInv(1, 2);
  // the CPU scratch buffers of the reference implementation do not exist here
  buf_warped_residual = buf_warped_dx = buf_warped_dy = buf_warped_x = buf_warped_y = buf_warped_z = nullptr;
  buf_d = buf_idepthVar = buf_weight_p = nullptr;
  buf_warped_size = 0;
  lastResidual = 0; iterationNumber = 0; pointUsage = 0; lastGoodCount = lastBadCount = 0; lastMeanRes = 0;
  affineEstimation_a = 1; affineEstimation_b = 0; affineEstimation_a_lastIt = 1; affineEstimation_b_lastIt = 0;
  diverged = false; trackingWasGood = false;
  TrackerState& S = state_of(this);
  S.ctx = lsd_slam_hipbind::context_for(w, h, K);
  lsd_slam_hipbind::check(lsdhip_tracker_create(S.ctx, &S.h), "lsdhip_tracker_create");
}

SE3Tracker::~SE3Tracker() {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_state.find(this);
  if (it != g_state.end()) { lsdhip_tracker_destroy(it->second.h); g_state.erase(it); }
}

SE3 SE3Tracker::trackFrame(TrackingReference* reference, Frame* frame, const SE3& frameToReference_initialEstimate) {
  TrackerState& S = state_of(this);
  push_settings(this, S.h);
  Frame* kf = reference->keyframe;
  lsd_slam_hipbind::sync_depth(kf, S.ctx);                                  // the keyframe's depth as the host frame holds it now
  lsdhip_frame* kfh = lsd_slam_hipbind::device_frame(kf, S.ctx);
  lsdhip_frame* fh = lsd_slam_hipbind::device_frame(frame, S.ctx);
  lsd_slam_hipbind::check(lsdhip_frame_set_counters(kfh, kf->numFramesTrackedOnThis, kf->numMappedOnThis, kf->numMappedOnThisTotal,
                                                    kf->depthHasBeenUpdatedFlag ? 1 : 0), "lsdhip_frame_set_counters");
  double init[7];
  pose_to(frameToReference_initialEstimate, init);
  lsdhip_track_result r;
  lsd_slam_hipbind::check(lsdhip_tracker_track(S.h, kfh, fh, init, &r), "lsdhip_tracker_track");
  publish(this, r);
  if (r.diverged) return SE3();                                             // SE3Tracker.cpp:324-329 / :369-374
  // the side effects of SE3Tracker.cpp:479-485, mirrored onto the host objects from what the device frame now holds
  const size_t n1 = (size_t)frame->width(SE3TRACKING_MIN_LEVEL) * frame->height(SE3TRACKING_MIN_LEVEL);
  std::vector<uint8_t> mask(n1);
  if (lsdhip_frame_get_wasgood(fh, mask.data()) == 1) std::memcpy(frame->refPixelWasGood(), mask.data(), n1);
  if (trackingWasGood) kf->numFramesTrackedOnThis++;
  float st[8];
  lsd_slam_hipbind::check(lsdhip_frame_stats(fh, st), "lsdhip_frame_stats");
  frame->initialTrackedResidual = st[0];
  double p8[8];
  lsd_slam_hipbind::check(lsdhip_frame_get_pose(fh, p8), "lsdhip_frame_get_pose");
  frame->pose->thisToParent_raw = lsd_slam_hipbind::sim3_from8(p8);
  frame->pose->trackingParent = kf->pose;
  double host8[8];
  lsd_slam_hipbind::sim3_to8(frame->pose->thisToParent_raw, host8);
  lsd_slam_hipbind::note_pose(frame, host8, kf, frame->initialTrackedResidual);
  return pose_of(r.frameToReference);
}

SE3 SE3Tracker::trackFrameOnPermaref(Frame* reference, Frame* frame, SE3 referenceToFrame) {
  TrackerState& S = state_of(this);
  push_settings(this, S.h);
  lsdhip_frame* fh = lsd_slam_hipbind::device_frame(frame, S.ctx);
  double init[7];
  pose_to(referenceToFrame, init);
  lsdhip_track_result r;
  boost::shared_lock<boost::shared_mutex> lock = frame->getActiveLock();
  boost::unique_lock<boost::mutex> lock2(reference->permaRef_mutex);
  lsd_slam_hipbind::check(lsdhip_tracker_track_permaref(S.h, (const float*)reference->permaRef_posData, (const float*)reference->permaRef_colorAndVarData,
                                                        reference->permaRefNumPts, fh, init, &r), "lsdhip_tracker_track_permaref");
  publish(this, r);
  return pose_of(r.frameToReference);
}

float SE3Tracker::checkPermaRefOverlap(Frame* reference, SE3 referenceToFrame) {
  TrackerState& S = state_of(this);
  double init[7];
  pose_to(referenceToFrame, init);
  float usage = 0;
  boost::unique_lock<boost::mutex> lock2(reference->permaRef_mutex);
  lsd_slam_hipbind::check(lsdhip_tracker_check_overlap(S.h, (const float*)reference->permaRef_posData, reference->permaRefNumPts, init, &usage),
                          "lsdhip_tracker_check_overlap");
  pointUsage = usage;
  return usage;
}

}  // namespace lsd_slam
