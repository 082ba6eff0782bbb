// lsd_slam_hip.hpp — header-only C++ host side above the C ABI (include/lsdhip.h).
//
// The reference's boundary for this path is a C++ class API; these classes put the reference's
// signatures back on top of liblsdhip.so so that SlamSystem-style callers read the same:
//
//   reference (lsd_slam_core/src/…)                         here (namespace lsd_slam_hip)
//   Frame(int id,int w,int h,const Eigen::Matrix3f& K,       Frame(int id,int w,int h,const Mat3f& K,double ts,
//         double timestamp,const unsigned char* image)            const unsigned char* image)
//                              DataStructures/Frame.h:43
//   TrackingReference::importFrame(Frame*)                  TrackingReference::importFrame(Frame*)
//                              Tracking/TrackingReference.h:49
//   SE3Tracker(int w,int h,Eigen::Matrix3f K)               SE3Tracker(int w,int h,const Mat3f& K)
//   SE3 trackFrame(TrackingReference*,Frame*,const SE3&)    SE3 trackFrame(TrackingReference*,Frame*,const SE3&)
//   SE3 trackFrameOnPermaref(Frame*,Frame*,SE3)             SE3 trackFrameOnPermaref(Frame*,Frame*,SE3)
//   float checkPermaRefOverlap(Frame*,SE3)                  float checkPermaRefOverlap(Frame*,SE3)
//                              Tracking/SE3Tracker.h:41-93
//   DepthMap(int w,int h,const Eigen::Matrix3f& K)          DepthMap(int w,int h,const Mat3f& K)
//   updateKeyframe(std::deque<std::shared_ptr<Frame>>)      updateKeyframe(std::deque<std::shared_ptr<Frame>>)
//   createKeyFrame(Frame*) / finalizeKeyFrame() / …         same names
//                              DepthEstimation/DepthMap.h:47-98
//
// Only the value types differ: Eigen / Sophus are not part of this repository, so Mat3f and SE3 / Sim3 are plain
// structs with the same content (row-major 3x3; unit quaternion + translation [+ scale], what Sophus::SE3d / Sim3d
// store).  INTEGRATION.md shows the two-line conversions a lsd_slam_core build adds.
//
// Error behaviour: the reference asserts on misuse and reports divergence through flags.  Here divergence is reported
// the same way (diverged = true, identity returned); usage / runtime errors of the C ABI (negative status) throw
// lsd_slam_hip::Error carrying lsdhip_last_error().  Nothing is computed on the host: without liblsdhip.so and a GPU
// every call fails loudly.
#ifndef LSD_SLAM_HIP_HPP
#define LSD_SLAM_HIP_HPP

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <exception>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "lsdhip.h"

namespace lsd_slam_hip {

struct Error : std::runtime_error {
  int status;
  Error(int st, const std::string& what) : std::runtime_error(what), status(st) {}
};
inline int check(int rc, const char* where) {
  if (rc < 0) throw Error(rc, std::string(where) + ": " + lsdhip_last_error());
  return rc;
}

struct Mat3f {
  float m[9];  // row-major
  float fx() const { return m[0]; }
  float fy() const { return m[4]; }
  float cx() const { return m[2]; }
  float cy() const { return m[5]; }
  static Mat3f intrinsics(float fx, float fy, float cx, float cy) { return Mat3f{{fx, 0, cx, 0, fy, cy, 0, 0, 1}}; }
};

// Sophus::SE3d content: unit quaternion (w,x,y,z) + translation
struct SE3 {
  double q[4] = {1, 0, 0, 0};
  double t[3] = {0, 0, 0};
  void to7(double p[7]) const { for (int i = 0; i < 4; i++) p[i] = q[i]; for (int i = 0; i < 3; i++) p[4 + i] = t[i]; }
  static SE3 from7(const double p[7]) { SE3 r; for (int i = 0; i < 4; i++) r.q[i] = p[i]; for (int i = 0; i < 3; i++) r.t[i] = p[4 + i]; return r; }
};
// Sophus::Sim3d content
struct Sim3 {
  double q[4] = {1, 0, 0, 0};
  double t[3] = {0, 0, 0};
  double s = 1;
};

// One device context per (device, w, h, K): what the (w, h, K) constructor arguments of Frame / SE3Tracker / DepthMap
// share in the reference.  Objects created with the same arguments use the same context and therefore the same stream.
class Context {
 public:
  static std::shared_ptr<Context> get(int w, int h, const Mat3f& K, int device = defaultDevice()) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, float, float, float, float>, std::weak_ptr<Context>> registry;
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_tuple(device, w, h, K.fx(), K.fy(), K.cx(), K.cy());
    if (auto sp = registry[key].lock()) return sp;
    std::shared_ptr<Context> sp(new Context(device, w, h, K));
    registry[key] = sp;
    return sp;
  }
  static int& defaultDevice() { static int d = 0; return d; }
  ~Context() { lsdhip_ctx_destroy(h_); }
  lsdhip_ctx* handle() const { return h_; }
  int width() const { return w_; }
  int height() const { return h_px_; }
  const Mat3f& K() const { return K_; }
  lsdhip_params params;  // the hot-path globals of util/settings.cpp:77-88 this context was created with
  void synchronize() { check(lsdhip_ctx_synchronize(h_), "lsdhip_ctx_synchronize"); }
  // mapping calls return after enqueueing; host-side results are picked up lazily (lsdhip.h: lsdhip_ctx_set_async)
  void setAsync(bool on) { check(lsdhip_ctx_set_async(h_, on ? 1 : 0), "lsdhip_ctx_set_async"); }
  // tracking stream beside mapping stream (lsdhip.h: lsdhip_ctx_set_pipeline): the reference's two threads, blockUntilMapped == false
  void setPipeline(bool on) { check(lsdhip_ctx_set_pipeline(h_, on ? 1 : 0), "lsdhip_ctx_set_pipeline"); }
  // lanes: DepthMap call chains of different maps side by side (lsdhip.h: lsdhip_ctx_lanes_begin)
  void lanesBegin(int n) { check(lsdhip_ctx_lanes_begin(h_, n), "lsdhip_ctx_lanes_begin"); }
  void laneSelect(int lane) { check(lsdhip_ctx_lane_select(h_, lane), "lsdhip_ctx_lane_select"); }
  void lanesEnd() { check(lsdhip_ctx_lanes_end(h_), "lsdhip_ctx_lanes_end"); }
  bool pipeline() const { return lsdhip_ctx_pipeline(h_) == 1; }

 private:
  Context(int device, int w, int h, const Mat3f& K) : w_(w), h_px_(h), K_(K) {
    lsdhip_default_params(&params);
    const float k4[4] = {K.fx(), K.fy(), K.cx(), K.cy()};
    check(lsdhip_ctx_create(device, w, h, k4, &params, &h_), "lsdhip_ctx_create");
  }
  lsdhip_ctx* h_ = nullptr;
  int w_, h_px_;
  Mat3f K_;
};

class TrackingReference;

// DataStructures/Frame.h — a frame whose pyramids live on the device
class Frame {
 public:
  Frame(int id, int width, int height, const Mat3f& K, double timestamp, const unsigned char* image)
      : ctx_(Context::get(width, height, K)), id_(id), timestamp_(timestamp) {
    check(lsdhip_frame_create(ctx_->handle(), id, image, &h_), "lsdhip_frame_create");
  }
  // image already resident on the context's GPU (uint8, w*h): no PCIe transfer
  struct DeviceImage { const unsigned char* ptr; };
  Frame(int id, int width, int height, const Mat3f& K, double timestamp, DeviceImage image)
      : ctx_(Context::get(width, height, K)), id_(id), timestamp_(timestamp) {
    check(lsdhip_frame_create_from_device(ctx_->handle(), id, image.ptr, &h_), "lsdhip_frame_create_from_device");
  }
  // host image whose bytes stay unchanged until the frame has been tracked (or the context synchronised): upload and pyramids are
  // queued, nothing waits (lsdhip_frame_create_async)
  struct HostImageAsync { const unsigned char* ptr; };
  Frame(int id, int width, int height, const Mat3f& K, double timestamp, HostImageAsync image)
      : ctx_(Context::get(width, height, K)), id_(id), timestamp_(timestamp) {
    check(lsdhip_frame_create_async(ctx_->handle(), id, image.ptr, &h_), "lsdhip_frame_create_async");
  }
  Frame(const Frame&) = delete;
  Frame& operator=(const Frame&) = delete;
  ~Frame() { lsdhip_frame_destroy(h_); }
  // the new frames of n sequences (one context) in two launches instead of two per frame (lsdhip_frame_create_batch)
  static std::vector<std::shared_ptr<Frame>> createBatch(int width, int height, const Mat3f& K, const std::vector<int>& ids,
                                                         const unsigned char* const* images, bool imagesOnDevice) {
    std::shared_ptr<Context> ctx = Context::get(width, height, K);
    const int n = (int)ids.size();
    std::vector<lsdhip_frame*> hs((size_t)n, nullptr);
    check(lsdhip_frame_create_batch(ctx->handle(), n, ids.data(), images, imagesOnDevice ? 1 : 0, hs.data()), "lsdhip_frame_create_batch");
    std::vector<std::shared_ptr<Frame>> out;
    out.reserve((size_t)n);
    for (int j = 0; j < n; j++) out.push_back(std::shared_ptr<Frame>(new Frame(ctx, ids[j], hs[j])));
    return out;
  }

  int id() const { return id_; }
  int width(int level = 0) const { return ctx_->width() >> level; }
  int height(int level = 0) const { return ctx_->height() >> level; }
  double timestamp() const { return timestamp_; }
  lsdhip_frame* handle() const { return h_; }
  const std::shared_ptr<Context>& context() const { return ctx_; }

  // Frame::image / gradients / maxGradients / idepth / idepthVar (Frame.h:357-418), copied to the host
  std::vector<float> image(int level = 0) const { return plane(0, level, 1); }
  std::vector<float> gradients(int level = 0) const { return plane(1, level, 4); }
  std::vector<float> maxGradients(int level = 0) const { return plane(2, level, 1); }
  std::vector<float> idepth(int level = 0) const { return plane(3, level, 1); }
  std::vector<float> idepthVar(int level = 0) const { return plane(4, level, 1); }

  void setDepthFromGroundTruth(const float* depth, float cov_scale = 1.0f) {
    check(lsdhip_frame_set_depth_gt(h_, depth, cov_scale), "lsdhip_frame_set_depth_gt");
  }
  // refPixelWasGood() (Frame.h:421-437): empty vector when the mask does not exist
  std::vector<uint8_t> refPixelWasGoodNoCreate() const {
    std::vector<uint8_t> m((size_t)width(1) * height(1));
    if (check(lsdhip_frame_get_wasgood(h_, m.data()), "lsdhip_frame_get_wasgood") != 1) m.clear();
    return m;
  }
  void clear_refPixelWasGood() { check(lsdhip_frame_clear_wasgood(h_), "lsdhip_frame_clear_wasgood"); }

  // FramePoseStruct::thisToParent_raw / trackingParent (written by trackFrame; settable for callers that track elsewhere)
  Sim3 thisToParent_raw() const {
    double p[8];
    check(lsdhip_frame_get_pose(h_, p), "lsdhip_frame_get_pose");
    Sim3 s;
    for (int i = 0; i < 4; i++) s.q[i] = p[i];
    for (int i = 0; i < 3; i++) s.t[i] = p[4 + i];
    s.s = p[7];
    return s;
  }
  void setPose(const Sim3& thisToParent, Frame* trackingParent, float initialTrackedResidual) {
    const double p[8] = {thisToParent.q[0], thisToParent.q[1], thisToParent.q[2], thisToParent.q[3],
                         thisToParent.t[0], thisToParent.t[1], thisToParent.t[2], thisToParent.s};
    check(lsdhip_frame_set_pose(h_, p, trackingParent ? trackingParent->h_ : nullptr, initialTrackedResidual), "lsdhip_frame_set_pose");
  }

  struct Stats {
    float initialTrackedResidual, meanIdepth;
    int numPoints, numFramesTrackedOnThis, numMappedOnThis, numMappedOnThisTotal;
    bool depthHasBeenUpdatedFlag;
  };
  Stats stats() const {
    float o[8];
    check(lsdhip_frame_stats(h_, o), "lsdhip_frame_stats");
    return Stats{o[0], o[1], (int)o[2], (int)o[3], (int)o[4], (int)o[5], o[6] != 0.0f};
  }
  float initialTrackedResidual() const { return stats().initialTrackedResidual; }
  float meanIdepth() const { return stats().meanIdepth; }
  int numPoints() const { return stats().numPoints; }
  bool depthHasBeenUpdatedFlag() const { return check(lsdhip_frame_depth_updated(h_), "lsdhip_frame_depth_updated") != 0; }
  // currentKeyFrame->depthHasBeenUpdatedFlag = false (SlamSystem.cpp:910)
  void clearDepthHasBeenUpdatedFlag() { check(lsdhip_frame_clear_depth_updated(h_), "lsdhip_frame_clear_depth_updated"); }

  // Frame::setPermaRef (Frame.cpp:149-174): keeps the level-QUICK_KF_CHECK_LVL point cloud of `reference` on the host
  inline void setPermaRef(TrackingReference* reference);
  std::vector<float> permaRef_posData;       // 3 floats per point
  std::vector<float> permaRef_colorAndVarData;  // 2 floats per point
  int permaRefNumPts = 0;

 private:
  std::vector<float> plane(int what, int level, int channels) const {
    std::vector<float> v((size_t)width(level) * height(level) * channels);
    check(lsdhip_frame_download(h_, what, level, v.data()), "lsdhip_frame_download");
    return v;
  }
  Frame(std::shared_ptr<Context> ctx, int id, lsdhip_frame* adopted) : ctx_(std::move(ctx)), h_(adopted), id_(id), timestamp_(0.0) {}
  std::shared_ptr<Context> ctx_;
  lsdhip_frame* h_ = nullptr;
  int id_;
  double timestamp_;
};

// Tracking/TrackingReference.h — on the device the point cloud is generated inside the residual kernel, so this object
// only remembers which keyframe it refers to; makePointCloud() exports the compacted arrays in the reference's order.
class TrackingReference {
 public:
  Frame* keyframe = nullptr;
  int frameID = -1;
  // TrackingReference.cpp:71-87.  The reference's object is a snapshot of the keyframe's depth taken here (lazily, at the next
  // trackFrame); on a pipelined context this is the moment the mapping side's newest Frame::setDepth result is handed to the tracker
  // (lsdhip_frame_publish_depth) — until then SE3Tracker jobs keep reading the planes of the previous import.
  void importFrame(Frame* source) {
    keyframe = source;
    frameID = source ? source->id() : -1;
    if (source) check(lsdhip_frame_publish_depth(source->handle()), "lsdhip_frame_publish_depth");
  }
  void invalidate() { keyframe = nullptr; frameID = -1; }
  // returns the number of points; any output may be null (TrackingReference.cpp:96-147)
  int makePointCloud(int level, std::vector<float>* posData, std::vector<float>* colorAndVarData,
                     std::vector<float>* gradData = nullptr, std::vector<int>* pointPosInXYGrid = nullptr) const {
    if (!keyframe) throw Error(LSDHIP_E_STATE, "TrackingReference::makePointCloud: no keyframe");
    const size_t cap = (size_t)keyframe->width(level) * keyframe->height(level);
    std::vector<float> pos(cap * 3), cv(cap * 2), gr(cap * 2);
    std::vector<int> idx(cap);
    const int n = check(lsdhip_ref_pointcloud(keyframe->handle(), level, pos.data(), cv.data(), gr.data(), idx.data()), "lsdhip_ref_pointcloud");
    if (posData) posData->assign(pos.begin(), pos.begin() + (size_t)n * 3);
    if (colorAndVarData) colorAndVarData->assign(cv.begin(), cv.begin() + (size_t)n * 2);
    if (gradData) gradData->assign(gr.begin(), gr.begin() + (size_t)n * 2);
    if (pointPosInXYGrid) pointPosInXYGrid->assign(idx.begin(), idx.begin() + n);
    return n;
  }
};

inline void Frame::setPermaRef(TrackingReference* reference) {
  permaRefNumPts = reference->makePointCloud(4 /* QUICK_KF_CHECK_LVL */, &permaRef_posData, &permaRef_colorAndVarData);
}

// util/settings.h:355-402, every field with the reference's defaults (settings.h:360-400)
struct DenseDepthTrackerSettings {
  float lambdaSuccessFac = 0.5f, lambdaFailFac = 2.0f;
  float lambdaInitial[LSDHIP_PYRAMID_LEVELS] = {0, 0, 0, 0, 0};
  float stepSizeMin[LSDHIP_PYRAMID_LEVELS] = {1e-8f, 1e-8f, 1e-8f, 1e-8f, 1e-8f};
  float convergenceEps[LSDHIP_PYRAMID_LEVELS] = {0.999f, 0.999f, 0.999f, 0.999f, 0.999f};
  int maxItsPerLvl[LSDHIP_PYRAMID_LEVELS] = {5, 20, 50, 100, 100};
  float lambdaInitialTestTrack = 0, stepSizeMinTestTrack = 1e-3f, convergenceEpsTestTrack = 0.98f, maxItsTestTrack = 5;
  float huber_d = 3, var_weight = 1.0f;
};

// Tracking/SE3Tracker.h:41-93
class SE3Tracker {
 public:
  int width, height;
  Mat3f K;
  DenseDepthTrackerSettings settings;

  SE3Tracker(int w, int h, const Mat3f& K_) : width(w), height(h), K(K_), ctx_(Context::get(w, h, K_)) {
    check(lsdhip_tracker_create(ctx_->handle(), &h_), "lsdhip_tracker_create");
  }
  SE3Tracker(const SE3Tracker&) = delete;
  SE3Tracker& operator=(const SE3Tracker&) = delete;
  ~SE3Tracker() { lsdhip_tracker_destroy(h_); }

  SE3 trackFrame(TrackingReference* reference, Frame* frame, const SE3& frameToReference_initialEstimate) {
    pushSettings();
    double init[7];
    frameToReference_initialEstimate.to7(init);
    lsdhip_track_result r;
    hookError_ = nullptr;
    if (!reference || !reference->keyframe || !frame) throw Error(LSDHIP_E_STATE, "SE3Tracker::trackFrame: the tracking reference holds no keyframe");
    check(lsdhip_tracker_track(h_, reference->keyframe->handle(), frame->handle(), init, &r), "lsdhip_tracker_track");
    publish(r);
    if (hookError_) std::rethrow_exception(hookError_);
    return SE3::from7(r.frameToReference);
  }
  // trackFrame for n independent (reference, frame) pairs in the same launches (lsdhip_tracker_track_batch): several sequences sharing
  // one GPU.  results[j] carries what the members above carry after a single call.  A diverged job does not throw: look at
  // results[j].diverged.
  std::vector<SE3> trackFrameBatch(const std::vector<TrackingReference*>& references, const std::vector<Frame*>& frames,
                                   const std::vector<SE3>& inits, std::vector<lsdhip_track_result>& results) {
    pushSettings();
    const int n = (int)frames.size();
    std::vector<lsdhip_frame*> kfs((size_t)n), frs((size_t)n);
    std::vector<double> init((size_t)n * 7);
    for (int j = 0; j < n; j++) {
      if (!references[j] || !references[j]->keyframe || !frames[j]) throw Error(LSDHIP_E_STATE, "SE3Tracker::trackFrameBatch: a tracking reference holds no keyframe");
      kfs[j] = references[j]->keyframe->handle();
      frs[j] = frames[j]->handle();
      inits[j].to7(&init[(size_t)j * 7]);
    }
    results.assign((size_t)n, lsdhip_track_result());
    hookError_ = nullptr;
    const int rc = lsdhip_tracker_track_batch(h_, n, kfs.data(), frs.data(), init.data(), results.data());
    if (rc != LSDHIP_OK && rc != LSDHIP_DIVERGED) check(rc, "lsdhip_tracker_track_batch");
    if (hookError_) std::rethrow_exception(hookError_);
    std::vector<SE3> out((size_t)n);
    for (int j = 0; j < n; j++) out[j] = SE3::from7(results[j].frameToReference);
    return out;
  }
  SE3 trackFrameOnPermaref(Frame* reference, Frame* frame, SE3 referenceToFrame) {
    pushSettings();
    double init[7];
    referenceToFrame.to7(init);
    lsdhip_track_result r;
    check(lsdhip_tracker_track_permaref(h_, reference->permaRef_posData.data(), reference->permaRef_colorAndVarData.data(),
                                        reference->permaRefNumPts, frame->handle(), init, &r), "lsdhip_tracker_track_permaref");
    publish(r);
    return SE3::from7(r.frameToReference);
  }
  // Runs on the calling thread inside trackFrame once the job is queued on the device and before the host waits for it
  // (lsdhip_tracker_set_enqueue_hook): queue independent work here, e.g. the next frame's upload and pyramids.
  // Exceptions thrown by the hook surface from trackFrame after the tracking result has been collected.
  void setEnqueueHook(std::function<void()> fn) {
    hook_ = std::move(fn);
    check(lsdhip_tracker_set_enqueue_hook(h_, hook_ ? &SE3Tracker::hookTrampoline : nullptr, this), "lsdhip_tracker_set_enqueue_hook");
  }
  float checkPermaRefOverlap(Frame* reference, SE3 referenceToFrame) {
    double init[7];
    referenceToFrame.to7(init);
    float usage = 0;
    check(lsdhip_tracker_check_overlap(h_, reference->permaRef_posData.data(), reference->permaRefNumPts, init, &usage), "lsdhip_tracker_check_overlap");
    pointUsage = usage;
    return usage;
  }

  float pointUsage = 0, lastGoodCount = 0, lastMeanRes = 0, lastBadCount = 0, lastResidual = 0;
  float affineEstimation_a = 1, affineEstimation_b = 0;
  bool diverged = false, trackingWasGood = false;
  int numEvaluations = 0, numWarpUpdates = 0;  // instrumentation (residual-kernel launches, LM outer iterations)
  int levelEvaluations[LSDHIP_PYRAMID_LEVELS] = {0, 0, 0, 0, 0};   // evaluations of the last trackFrame per pyramid level
  int numLaunches = 0;                         // evaluating launches of the last job (< numEvaluations: retries share launches)
  void setSpeculation(int trials, int finestLevelWorkgroups = 0) { check(lsdhip_tracker_set_speculation(h_, trials, finestLevelWorkgroups), "lsdhip_tracker_set_speculation"); }
  void setBatchCoarseMinJobs(int minJobs) { check(lsdhip_tracker_set_batch_coarse_min_jobs(h_, minJobs), "lsdhip_tracker_set_batch_coarse_min_jobs"); }

 private:
  void pushSettings() {   // the public `settings` member is plain data in the reference: hand it over before every job
    lsdhip_tracker_settings st;
    st.lambdaSuccessFac = settings.lambdaSuccessFac; st.lambdaFailFac = settings.lambdaFailFac;
    for (int l = 0; l < LSDHIP_PYRAMID_LEVELS; l++) {
      st.lambdaInitial[l] = settings.lambdaInitial[l]; st.stepSizeMin[l] = settings.stepSizeMin[l];
      st.convergenceEps[l] = settings.convergenceEps[l]; st.maxItsPerLvl[l] = settings.maxItsPerLvl[l];
    }
    st.lambdaInitialTestTrack = settings.lambdaInitialTestTrack; st.stepSizeMinTestTrack = settings.stepSizeMinTestTrack;
    st.convergenceEpsTestTrack = settings.convergenceEpsTestTrack; st.maxItsTestTrack = settings.maxItsTestTrack;
    st.huber_d = settings.huber_d; st.var_weight = settings.var_weight;
    check(lsdhip_tracker_set_settings(h_, &st), "lsdhip_tracker_set_settings");
  }
  void publish(const lsdhip_track_result& r) {
    pointUsage = r.pointUsage; lastGoodCount = r.lastGoodCount; lastMeanRes = r.lastMeanRes; lastBadCount = r.lastBadCount;
    lastResidual = r.lastResidual; affineEstimation_a = r.affineEstimation_a; affineEstimation_b = r.affineEstimation_b;
    diverged = r.diverged != 0; trackingWasGood = r.trackingWasGood != 0;
    numEvaluations = r.numEvaluations; numWarpUpdates = r.numWarpUpdates;
    int st[8];
    if (lsdhip_tracker_exec_stats(h_, st) == 0) for (int l = 0; l < LSDHIP_PYRAMID_LEVELS; l++) levelEvaluations[l] = st[3 + l];
    int ls[2];
    if (lsdhip_tracker_launch_stats(h_, ls) == 0) numLaunches = ls[0];
  }
  static void hookTrampoline(void* self) {
    SE3Tracker* t = static_cast<SE3Tracker*>(self);
    try { if (t->hook_) t->hook_(); } catch (...) { t->hookError_ = std::current_exception(); }
  }
  std::function<void()> hook_;
  std::exception_ptr hookError_;
  std::shared_ptr<Context> ctx_;
  lsdhip_tracker* h_ = nullptr;
};

// Tracking/Sim3Tracker.h:71-187
class Sim3Tracker {
 public:
  int width, height;
  Mat3f K;
  DenseDepthTrackerSettings settings;

  Sim3Tracker(int w, int h, const Mat3f& K_) : width(w), height(h), K(K_), ctx_(Context::get(w, h, K_)) {
    check(lsdhip_sim3tracker_create(ctx_->handle(), &h_), "lsdhip_sim3tracker_create");
  }
  Sim3Tracker(const Sim3Tracker&) = delete;
  Sim3Tracker& operator=(const Sim3Tracker&) = delete;
  ~Sim3Tracker() { lsdhip_sim3tracker_destroy(h_); }

  Sim3 trackFrameSim3(TrackingReference* reference, Frame* frame, const Sim3& frameToReference_initialEstimate, int startLevel,
                      int finalLevel) {
    check(lsdhip_sim3tracker_set_max_its(h_, settings.maxItsPerLvl), "lsdhip_sim3tracker_set_max_its");
    double init[8] = {frameToReference_initialEstimate.q[0], frameToReference_initialEstimate.q[1], frameToReference_initialEstimate.q[2],
                      frameToReference_initialEstimate.q[3], frameToReference_initialEstimate.t[0], frameToReference_initialEstimate.t[1],
                      frameToReference_initialEstimate.t[2], frameToReference_initialEstimate.s};
    lsdhip_sim3_result r;
    if (!reference || !reference->keyframe || !frame) throw Error(LSDHIP_E_STATE, "Sim3Tracker::trackFrameSim3: the tracking reference holds no keyframe");
    check(lsdhip_sim3tracker_track(h_, reference->keyframe->handle(), frame->handle(), init, startLevel, finalLevel, &r),
          "lsdhip_sim3tracker_track");
    pointUsage = r.pointUsage; lastResidual = r.lastResidual; lastDepthResidual = r.lastDepthResidual;
    lastPhotometricResidual = r.lastPhotometricResidual; affineEstimation_a = r.affineEstimation_a; affineEstimation_b = r.affineEstimation_b;
    diverged = r.diverged != 0; numEvaluations = r.numEvaluations;
    for (int i = 0; i < 49; i++) lastSim3Hessian[i] = r.lastSim3Hessian[i];
    Sim3 out;
    for (int i = 0; i < 4; i++) out.q[i] = r.frameToReference[i];
    for (int i = 0; i < 3; i++) out.t[i] = r.frameToReference[4 + i];
    out.s = r.frameToReference[7];
    return out;
  }

  // n independent jobs in lock step (their evaluations share launches); results[j] as a single call would leave them
  std::vector<Sim3> trackFrameSim3Batch(const std::vector<TrackingReference*>& references, const std::vector<Frame*>& frames,
                                        const std::vector<Sim3>& inits, int startLevel, int finalLevel,
                                        std::vector<lsdhip_sim3_result>* results = nullptr) {
    check(lsdhip_sim3tracker_set_max_its(h_, settings.maxItsPerLvl), "lsdhip_sim3tracker_set_max_its");
    const size_t n = frames.size();
    std::vector<lsdhip_frame*> kfs(n), frs(n);
    std::vector<double> init(8 * n);
    for (size_t j = 0; j < n; j++) {
      kfs[j] = references[j]->keyframe->handle();
      frs[j] = frames[j]->handle();
      for (int i = 0; i < 4; i++) init[8 * j + i] = inits[j].q[i];
      for (int i = 0; i < 3; i++) init[8 * j + 4 + i] = inits[j].t[i];
      init[8 * j + 7] = inits[j].s;
    }
    std::vector<lsdhip_sim3_result> res(n);
    check(lsdhip_sim3tracker_track_batch(h_, (int)n, kfs.data(), frs.data(), init.data(), startLevel, finalLevel, res.data()),
          "lsdhip_sim3tracker_track_batch");
    std::vector<Sim3> out(n);
    for (size_t j = 0; j < n; j++) {
      for (int i = 0; i < 4; i++) out[j].q[i] = res[j].frameToReference[i];
      for (int i = 0; i < 3; i++) out[j].t[i] = res[j].frameToReference[4 + i];
      out[j].s = res[j].frameToReference[7];
    }
    if (results) *results = res;
    return out;
  }

  float pointUsage = 0, lastResidual = 0, lastDepthResidual = 0, lastPhotometricResidual = 0;
  float affineEstimation_a = 1, affineEstimation_b = 0;
  float lastSim3Hessian[49] = {};   // row-major 7x7
  bool diverged = false;
  int numEvaluations = 0;

 private:
  std::shared_ptr<Context> ctx_;
  lsdhip_sim3tracker* h_ = nullptr;
};

// DepthEstimation/DepthMap.h:47-98
class DepthMap {
 public:
  DepthMap(int w, int h, const Mat3f& K) : ctx_(Context::get(w, h, K)) { check(lsdhip_depth_create(ctx_->handle(), &h_), "lsdhip_depth_create"); }
  DepthMap(const DepthMap&) = delete;
  DepthMap& operator=(const DepthMap&) = delete;
  ~DepthMap() { lsdhip_depth_destroy(h_); }

  void reset() { check(lsdhip_depth_reset(h_), "lsdhip_depth_reset"); }
  void updateKeyframe(std::deque<std::shared_ptr<Frame>> referenceFrames) {
    std::vector<lsdhip_frame*> refs;
    refs.reserve(referenceFrames.size());
    for (auto& f : referenceFrames) refs.push_back(f->handle());
    check(lsdhip_depth_update(h_, refs.data(), (int)refs.size()), "lsdhip_depth_update");
    timings();
  }
  // updateKeyframe of several sequences' maps (one context), one tracked frame each, in shared launches (lsdhip_depth_update_batch)
  static void updateKeyframeBatch(const std::vector<DepthMap*>& maps, const std::vector<Frame*>& frames) {
    if (maps.empty()) return;
    std::vector<lsdhip_depthmap*> ms;
    std::vector<lsdhip_frame*> fs;
    for (DepthMap* m : maps) ms.push_back(m->h_);
    for (Frame* f : frames) fs.push_back(f->handle());
    check(lsdhip_depth_update_batch((int)ms.size(), ms.data(), fs.data()), "lsdhip_depth_update_batch");
  }
  void createKeyFrame(Frame* new_keyframe) {
    check(lsdhip_depth_create_keyframe(h_, new_keyframe->handle(), nullptr), "lsdhip_depth_create_keyframe");
    timings();
  }
  // finalizeKeyFrame() + createKeyFrame(newKeyframes[j]) of several sequences' maps (one context) in six shared launches
  // (lsdhip_depth_change_keyframe_batch); one map: the same six launches instead of the two calls' fifteen
  static void changeKeyframeBatch(const std::vector<DepthMap*>& maps, const std::vector<Frame*>& newKeyframes) {
    if (maps.empty()) return;
    std::vector<lsdhip_depthmap*> ms;
    std::vector<lsdhip_frame*> fs;
    for (DepthMap* m : maps) ms.push_back(m->h_);
    for (Frame* f : newKeyframes) fs.push_back(f->handle());
    check(lsdhip_depth_change_keyframe_batch((int)ms.size(), ms.data(), fs.data(), nullptr), "lsdhip_depth_change_keyframe_batch");
  }
  void changeKeyframe(Frame* new_keyframe) { changeKeyframeBatch({this}, {new_keyframe}); timings(); }
  // GPU time (ms, summed) and call counts of updateKeyframe [0] / createKeyFrame [1] / finalizeKeyFrame [2]; synchronises
  void gpuTimes(double ms[3], long long calls[3]) { check(lsdhip_depth_gpu_times(h_, ms, calls), "lsdhip_depth_gpu_times"); }
  void finalizeKeyFrame() { check(lsdhip_depth_finalize(h_), "lsdhip_depth_finalize"); timings(); }
  void invalidate() { check(lsdhip_depth_invalidate(h_), "lsdhip_depth_invalidate"); }
  bool isValid() { return lsdhip_depth_is_valid(h_) != 0; }
  void initializeFromGTDepth(Frame* new_frame) { check(lsdhip_depth_init_gt(h_, new_frame->handle()), "lsdhip_depth_init_gt"); }
  void initializeRandomly(Frame* new_frame) { check(lsdhip_depth_init_random(h_, new_frame->handle()), "lsdhip_depth_init_random"); }
  void setFromExistingKF(Frame* kf) { check(lsdhip_depth_set_from_existing(h_, kf->handle()), "lsdhip_depth_set_from_existing"); }

  // currentDepthMap in the reference's 32-byte AoS layout (DepthMapPixelHypothesis.h:43-60)
  std::vector<lsdhip_hypothesis> currentDepthMap() const {
    std::vector<lsdhip_hypothesis> v((size_t)ctx_->width() * ctx_->height());
    check(lsdhip_depth_download(h_, v.data()), "lsdhip_depth_download");
    return v;
  }
  // smoothed idepth / variance planes of the active keyframe, device to device (payload of the multi-GPU gather)
  void copyPlanesToDevice(float* idepth_dev, float* idepthVar_dev) { check(lsdhip_depth_copy_planes_dev(h_, idepth_dev, idepthVar_dev), "lsdhip_depth_copy_planes_dev"); }

  float msUpdate = 0, msCreate = 0, msFinalize = 0, msObserve = 0, msRegularize = 0, msPropagate = 0, msFillHoles = 0, msSetDepth = 0;
  lsdhip_depthmap* handle() const { return h_; }

 private:
  void timings() {
    float t[8];
    if (lsdhip_depth_timings(h_, t) == 0) { msUpdate = t[0]; msCreate = t[1]; msFinalize = t[2]; msObserve = t[3]; msRegularize = t[4]; msPropagate = t[5]; msFillHoles = t[6]; msSetDepth = t[7]; }
  }
  std::shared_ptr<Context> ctx_;
  lsdhip_depthmap* h_ = nullptr;
};

// The part of SlamSystem either side of the hot path with doSlam = false (SlamSystem.cpp:890-1040 trackFrame, :739-828
// doMappingIteration, :542-614 updateKeyframe, :458-490 createNewCurrentKeyframe) and a deterministic keyframe policy (a new keyframe
// every `kfEvery` frames) in place of the distance / usage score.  Orchestration only: every per-pixel operation is a liblsdhip.so call.
//
// Two execution models, both the reference's own:
//   * blockUntilMapped = true (default): track frame t, then its mapping iteration, on one stream; frame t + 1 is tracked against what
//     map(t) left (SlamSystem.cpp:1026-1040 with the wait);
//   * pipelined (setPipelined(true); blockUntilMapped = false with the mapper exactly one frame behind): the mapping iteration of frame t
//     is queued on the context's mapping stream when trackFrame(t) ends and runs BESIDE trackFrame(t + 1), which tracks against the
//     keyframe and depth version map(t - 1) left — what the reference's tracking thread does when the mapping thread is still busy
//     (:907-912: it re-imports the tracking reference only when it finds depthHasBeenUpdatedFlag set / the keyframe replaced).  The
//     frame that follows a keyframe change is therefore still tracked on the old keyframe; the mapper drops it (updateKeyframe pops
//     frames whose tracking parent is not the current keyframe, :559-566) and the first frame tracked on the new keyframe starts from
//     se3FromSim3(newKeyframe^-1 * lastTrackedFrame) (:918-920).  The lag is fixed at one frame, so results do not depend on timing.
class SlamLoop {
 public:
  // gtDepth0 != null: SlamSystem::gtDepthInit (SlamSystem.cpp:831-854); null: SlamSystem::randomInit (:857-881).
  // kfEvery > 0: a new keyframe every kfEvery frames (deterministic; what bench.py and the parity tests use);
  // kfEvery == 0: the reference's distance / usage score (SlamSystem.cpp:997-1015 with doSlam = false, see keyframeScore).
  SlamLoop(int w, int h, const Mat3f& K, const unsigned char* firstImage, bool imagesOnDevice, const float* gtDepth0, int kfEvery)
      : tracker(w, h, K), map(w, h, K), w_(w), h_(h), K_(K), onDevice_(imagesOnDevice), kfEvery_(kfEvery) {
    const int its[LSDHIP_PYRAMID_LEVELS] = {5, 20, 50, 100, 0};  // SlamSystem.cpp:80-81 (no level-4 iterations are run by the tracker)
    std::memcpy(tracker.settings.maxItsPerLvl, its, sizeof(its));
    keyframe = makeFrame(0, firstImage);
    if (gtDepth0) {
      keyframe->setDepthFromGroundTruth(gtDepth0);
      map.initializeFromGTDepth(keyframe.get());
    } else {
      map.initializeRandomly(keyframe.get());
    }
    trackKF_ = keyframe;
    reference.importFrame(keyframe.get());
    keyframe->clearDepthHasBeenUpdatedFlag();
    tracker.setEnqueueHook([this]() {
      // runs once the tracking job's launches are queued: the place for everything the device can do beside it
      flushDeferredMapping();
      if (!pendingNext_) return;
      prefetched_ = makeFrame(frameId_ + 1, pendingNext_, true);
      prefetchedSrc_ = pendingNext_;
    });
  }
  SlamLoop(const SlamLoop&) = delete;
  SlamLoop& operator=(const SlamLoop&) = delete;
  // Tracking beside mapping with the mapper one frame behind (see above).  Switch before the first step() call.
  void setPipelined(bool on) {
    if (frameId_ != 0) throw Error(LSDHIP_E_STATE, "SlamLoop::setPipelined: switch before the first frame");
    Context::get(w_, h_, K_)->setPipeline(on);
    pipelined_ = on;
  }
  bool pipelined() const { return pipelined_; }
  // Pipelined loops queue a frame's updateKeyframe only once the NEXT frame's tracking launches are queued (from the tracker's enqueue
  // hook: between two tracking jobs the host queues nothing but the next job).  After the last step() of a batch this queues what is
  // still waiting.
  void flushDeferredMapping() {
    if (!deferredMap_) return;
    std::shared_ptr<Frame> frame = std::move(deferredMap_);
    deferredMap_.reset();
    runUpdate(frame);
  }
  // TrackableKeyFrameSearch::getRefFrameScore (GlobalMapping/TrackableKeyFrameSearch.h:75-79) with the default weights
  // KFDistWeight = 4, KFUsageWeight = 3 (util/settings.cpp:77-78)
  static float keyframeScore(float distanceSquared, float usage) {
    return distanceSquared * 4.0f * 4.0f + (1 - usage) * (1 - usage) * 3.0f * 3.0f;
  }
  // called with the keyframe that has just been finalised (before the next one replaces it): output hook (PLY, messages)
  std::function<void(Frame&, DepthMap&)> onKeyframeFinished;
  // track one frame, then one mapping iteration; returns frameToKeyframe (of the keyframe the frame was tracked on).  Throws when
  // tracking diverges.
  SE3 step(const unsigned char* image) { return step(image, [](double) {}); }
  // same, reporting the wall-clock instant (seconds, steady clock) at which tracking ended and mapping began.
  // nextImage (optional): the image of the following step() call, if it is already available — its upload and pyramids
  // are queued while the host waits for this frame's tracking result, so they fill the device's idle time of that round
  // trip.  The bytes behind nextImage must stay unchanged until that following call (same pointer) has returned.
  template <typename F>
  SE3 step(const unsigned char* image, F&& onTrackEnd, const unsigned char* nextImage = nullptr) {
    // after a tracking loss the keyframe and the map are gone (SlamSystem::trackingIsGood == false until the relocaliser — out of
    // scope — finds a pose again): every further step fails the same way instead of touching the invalidated reference
    if (trackingLost) throw Error(LSDHIP_E_STATE, "SlamLoop: tracking lost (no relocaliser here): create a new loop");
    lsdhip_host_mark(0);
    frameId_++;
    std::shared_ptr<Frame> frame;
    if (prefetched_ && prefetchedSrc_ == image) frame = std::move(prefetched_);
    else frame = makeFrame(frameId_, image);
    prefetched_.reset();
    pendingNext_ = nextImage;
    if (!pipelined_ && keyframe->depthHasBeenUpdatedFlag()) {
      reference.importFrame(keyframe.get());
      keyframe->clearDepthHasBeenUpdatedFlag();
    }
    lsdhip_host_mark(1);
    const std::shared_ptr<Frame> trackedOn = trackKF_;     // == keyframe unless the mapper is promoting a new one right now (pipelined)
    SE3 est = tracker.trackFrame(&reference, frame.get(), lastFrameToKF_);
    lsdhip_host_mark(8);
    pendingNext_ = nullptr;
    evaluations += tracker.numEvaluations;
    launches += tracker.numLaunches;
    for (int l = 0; l < LSDHIP_PYRAMID_LEVELS; l++) levelEvaluations[l] += tracker.levelEvaluations[l];
    if (tracker.trackingWasGood) numTrackedGood++;
    numTracked++;
    lastTrackEnd = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    onTrackEnd(lastTrackEnd);
    // SlamSystem::trackFrame (SlamSystem.cpp:946-966): tracking is lost when the tracker diverged, or — once the initialisation
    // phase is over (more than INITIALIZATION_PHASE_COUNT = 5 keyframes in the graph) — when trackingWasGood is false.  The
    // frame is then neither mapped nor promoted; doMappingIteration's lost branch (:809-817) finalises the current keyframe if
    // it has been mapped on at least MIN_NUM_MAPPED = 5 times, discards it otherwise, and invalidates the map.  What follows in
    // the reference is the relocaliser (out of scope): the loop reports the loss.
    if (tracker.diverged || (numKeyframesFinished_ > 5 /* INITIALIZATION_PHASE_COUNT */ && !tracker.trackingWasGood)) {
      trackingLost = true;
      reference.invalidate();
      if (map.isValid()) {
        if (mappedOnKF_ >= 5 /* MIN_NUM_MAPPED */) {
          map.finalizeKeyFrame();
          numKeyframesFinished_++;
          if (onKeyframeFinished) onKeyframeFinished(*keyframe, map);
        }   // else discardCurrentKeyframe (:428-456): nothing of it is kept
        map.invalidate();
      }
      throw Error(LSDHIP_DIVERGED, std::string("SlamLoop: tracking lost at frame ") + std::to_string(frameId_) +
                                       (tracker.diverged ? " (diverged)" : " (trackingWasGood false)"));
    }
    if (pipelined_) {
      // What the tracking thread finds when it comes back for the next frame: the mapping iteration queued one frame ago has finished.
      if (pendingKF_) {
        // the mapper promoted pendingKF_ (createNewCurrentKeyframe) while this frame was being tracked on the old keyframe: the next
        // frame is tracked on the new one, from se3FromSim3(newKeyframe^-1 * thisFrame) (SlamSystem.cpp:907-920)
        double p[7];
        check(lsdhip_frame_relative_pose(pendingKF_->handle(), frame->handle(), p), "lsdhip_frame_relative_pose");
        lastFrameToKF_ = SE3::from7(p);
        trackKF_ = pendingKF_;
        pendingKF_.reset();
        reference.importFrame(trackKF_.get());
        trackKF_->clearDepthHasBeenUpdatedFlag();
      } else {
        lastFrameToKF_ = est;
        if (trackKF_->depthHasBeenUpdatedFlag()) {
          reference.importFrame(trackKF_.get());
          trackKF_->clearDepthHasBeenUpdatedFlag();
        }
      }
    }
    // ---- this frame's mapping iteration (pipelined: queued on the mapping stream, it runs beside the next frame's tracking) ----------
    ++sinceKF_;
    newKeyframe = false;
    flushDeferredMapping();      // (normally empty: the tracking job's hook has queued it)
    if (trackedOn != keyframe) {
      // tracked on the keyframe the mapper has just replaced: SlamSystem::updateKeyframe pops such frames unmapped (:559-566)
      frame->clear_refPixelWasGood();
      numDropped++;
      return est;
    }
    bool createNewKeyFrame = kfEvery_ > 0 && sinceKF_ >= kfEvery_;
    // numMappedOnThisTotal of the current keyframe = updateKeyframe calls since it was created: counted here, so that the
    // asynchronous pipeline is not drained every frame by Frame::stats()
    if (kfEvery_ == 0 && mappedOnKF_ > 5 /* MIN_NUM_MAPPED */) {
      // SlamSystem.cpp:997-1015: dist = translation * meanIdepth; keyframesAll is empty without the pose graph, so
      // minVal = 0.2 * 0.7 (SURVEY.md §8(d) driver notes)
      const float mi = keyframe->meanIdepth();
      const float d0 = (float)est.t[0] * mi, d1 = (float)est.t[1] * mi, d2 = (float)est.t[2] * mi;
      createNewKeyFrame = keyframeScore(d0 * d0 + d1 * d1 + d2 * d2, tracker.pointUsage) > 0.2f * 0.7f;
    }
    if (createNewKeyFrame) {
      numKeyframesFinished_++;
      mappedOnKF_ = 0;
      if (onKeyframeFinished || !sharedKeyframeChange) {
        // an output hook reads the finalised map (PLY, keyframe messages, the multi-GPU gather): the two calls, the hook between them
        map.finalizeKeyFrame();
        if (onKeyframeFinished) onKeyframeFinished(*keyframe, map);
        map.createKeyFrame(frame.get());
      } else {
        map.changeKeyframe(frame.get());
      }
      keyframe = frame;
      if (keepKeyframes) keyframeLog.push_back(frame);
      liveQueue_.clear();
      if (pipelined_) {
        pendingKF_ = frame;        // the tracker keeps the old keyframe for one more frame
      } else {
        trackKF_ = frame;
        reference.importFrame(keyframe.get());
        keyframe->clearDepthHasBeenUpdatedFlag();
        lastFrameToKF_ = SE3();
      }
      sinceKF_ = 0;
      newKeyframe = true;
    } else if (pipelined_ && deferMapping) {
      deferredMap_ = frame;       // queued from the next tracking job's enqueue hook (or flushDeferredMapping)
    } else {
      runUpdate(frame);
      if (!pipelined_) lastFrameToKF_ = est;
    }
    return est;
  }
  SE3Tracker tracker;
  DepthMap map;
  TrackingReference reference;
  std::shared_ptr<Frame> keyframe;                   // the mapper's current keyframe (DepthMap::activeKeyFrame)
  bool newKeyframe = false;
  bool trackingLost = false;                         // set before step() throws: SlamSystem::trackingIsGood == false
  long evaluations = 0, launches = 0, numTracked = 0, numUpdates = 0, numTrackedGood = 0;
  long numDropped = 0;                               // frames tracked on a keyframe the mapper had already replaced (pipelined): not mapped
  long levelEvaluations[LSDHIP_PYRAMID_LEVELS] = {0, 0, 0, 0, 0};
  double lastTrackEnd = 0;
  int liveQueueLength = 1;                           // frames handed to updateKeyframe per mapping iteration
  bool deferMapping = true;                          // pipelined loops: see flushDeferredMapping
  bool sharedKeyframeChange = true;                  // finalizeKeyFrame + createKeyFrame as DepthMap::changeKeyframe (six launches); false: the two calls
  bool keepKeyframes = false;                        // keep every keyframe alive in keyframeLog (validation: rescale factors)
  std::vector<std::shared_ptr<Frame>> keyframeLog;

 private:
  // DepthMap::updateKeyframe for one tracked frame (SlamSystem::updateKeyframe, SlamSystem.cpp:542-614)
  void runUpdate(const std::shared_ptr<Frame>& frame) {
    // blockUntilMapped = true: the unmapped queue holds exactly this frame (SlamSystem.cpp:559-571, :1030-1039).
    // liveQueueLength > 1 restates live operation, where the mapper finds several tracked frames waiting and passes the
    // whole deque (oldest first, at most liveQueueLength of them) to updateKeyframe.
    liveQueue_.push_back(frame);
    while ((int)liveQueue_.size() > (liveQueueLength > 1 ? liveQueueLength : 1)) liveQueue_.pop_front();
    std::deque<std::shared_ptr<Frame>> q(liveQueue_.begin(), liveQueue_.end());
    lsdhip_host_mark(9);
    map.updateKeyframe(q);
    lsdhip_host_mark(13);
    mappedOnKF_++;
    frame->clear_refPixelWasGood();
    numUpdates++;
    lsdhip_host_mark(14);
  }
  std::shared_ptr<Frame> deferredMap_;
  std::shared_ptr<Frame> makeFrame(int id, const unsigned char* img, bool mayDefer = false) {
    if (onDevice_) return std::make_shared<Frame>(id, w_, h_, K_, 0.0, Frame::DeviceImage{img});
    // a prefetched host image on a pipelined context: the copy is queued on the mapping stream and nothing waits for it (the caller
    // keeps the bytes unchanged until the frame has been tracked: step()'s nextImage contract)
    if (mayDefer && pipelined_) return std::make_shared<Frame>(id, w_, h_, K_, 0.0, Frame::HostImageAsync{img});
    return std::make_shared<Frame>(id, w_, h_, K_, 0.0, img);
  }
  int w_, h_;
  Mat3f K_;
  bool onDevice_;
  bool pipelined_ = false;
  int kfEvery_, sinceKF_ = 0, frameId_ = 0;
  int mappedOnKF_ = 0;            // Frame::numMappedOnThisTotal of the current keyframe
  int numKeyframesFinished_ = 0;  // keyFrameGraph->keyframesAll.size()
  SE3 lastFrameToKF_;
  std::shared_ptr<Frame> trackKF_;     // the keyframe the tracker's reference points at
  std::shared_ptr<Frame> pendingKF_;   // pipelined: promoted by the mapping iteration queued last, not yet adopted by the tracker
  std::shared_ptr<Frame> prefetched_;
  const unsigned char* prefetchedSrc_ = nullptr;
  const unsigned char* pendingNext_ = nullptr;
  std::deque<std::shared_ptr<Frame>> liveQueue_;
};

// S independent sequences through ONE GPU, one frame of each per step() (BASELINE.json configs[3] with more sequences than GPUs).
// Per sequence this is SlamLoop with blockUntilMapped = true: track the frame, then its mapping iteration, the next frame tracked against
// what that left.  Across sequences every stage shares its launches: the new frames (Frame::createBatch), the tracking jobs
// (SE3Tracker::trackFrameBatch: throughput mode from 8 jobs on), the updateKeyframe calls (DepthMap::updateKeyframeBatch).  Keyframe
// changes (finalize + createKeyFrame, once every kfEvery frames per sequence) stay per-sequence calls.  A sequence whose tracking is
// lost stops taking part (lost(s) == true, SlamLoop's rules); the others go on.
class SlamLoopBatch {
 public:
  struct Sequence {
    Sequence(int w, int h, const Mat3f& K) : map(w, h, K) {}
    DepthMap map;
    TrackingReference reference;
    std::shared_ptr<Frame> keyframe;       // the mapper's current keyframe
    std::shared_ptr<Frame> trackKF;        // the keyframe the tracking reference points at (== keyframe unless a promotion is pending)
    std::shared_ptr<Frame> pendingKF;      // pipelined: promoted by the mapping iteration queued last, not yet adopted by the tracker
    SE3 lastFrameToKF;
    int sinceKF = 0, mappedOnKF = 0, numKeyframesFinished = 0;
    long numTracked = 0, numTrackedGood = 0, numUpdates = 0, evaluations = 0, numDropped = 0;
    bool trackingLost = false, newKeyframe = false;
    lsdhip_track_result last = lsdhip_track_result();
    std::vector<std::shared_ptr<Frame>> keyframeLog;   // every keyframe this sequence promoted (SlamLoopBatch::keepKeyframes)
  };
  // firstImages / gtDepth0: S pointers each (gtDepth0 == null or gtDepth0[s] == null: random initialisation of that sequence)
  SlamLoopBatch(int w, int h, const Mat3f& K, int S, const unsigned char* const* firstImages, bool imagesOnDevice,
                const float* const* gtDepth0, int kfEvery)
      : tracker(w, h, K), w_(w), h_(h), K_(K), onDevice_(imagesOnDevice), kfEvery_(kfEvery) {
    if (S <= 0 || kfEvery <= 0) throw Error(LSDHIP_E_ARG, "SlamLoopBatch: S > 0 sequences, a fixed keyframe interval > 0");
    const int its[LSDHIP_PYRAMID_LEVELS] = {5, 20, 50, 100, 0};
    std::memcpy(tracker.settings.maxItsPerLvl, its, sizeof(its));
    tracker.setBatchCoarseMinJobs(coarseMinJobs(false));
    std::vector<int> ids((size_t)S);
    for (int s = 0; s < S; s++) ids[s] = idOf(s, 0);
    std::vector<std::shared_ptr<Frame>> first = Frame::createBatch(w, h, K, ids, firstImages, imagesOnDevice);
    for (int s = 0; s < S; s++) {
      seqs_.emplace_back(new Sequence(w, h, K));
      Sequence& q = *seqs_.back();
      q.keyframe = first[s];
      if (gtDepth0 && gtDepth0[s]) {
        q.keyframe->setDepthFromGroundTruth(gtDepth0[s]);
        q.map.initializeFromGTDepth(q.keyframe.get());
      } else {
        q.map.initializeRandomly(q.keyframe.get());
      }
      q.reference.importFrame(q.keyframe.get());
      q.keyframe->clearDepthHasBeenUpdatedFlag();
      q.trackKF = q.keyframe;
    }
  }
  // Sequences started together change keyframe in the same step; real cameras do not.  phase[s] in [0, kfEvery): sequence s behaves as
  // if its current keyframe were already phase[s] frames old (its first keyframe change comes kfEvery - phase[s] frames in).  Call before
  // the first step().
  void setKeyframePhases(const std::vector<int>& phase) {
    if (frameId_ != 0 || (int)phase.size() != size()) throw Error(LSDHIP_E_STATE, "SlamLoopBatch::setKeyframePhases: one phase per sequence, before the first step");
    for (int s = 0; s < size(); s++) seqs_[s]->sinceKF = phase[s] < 0 ? 0 : phase[s] % kfEvery_;
  }
  SlamLoopBatch(const SlamLoopBatch&) = delete;
  SlamLoopBatch& operator=(const SlamLoopBatch&) = delete;
  int size() const { return (int)seqs_.size(); }
  Sequence& sequence(int s) { return *seqs_[s]; }
  bool lost(int s) const { return seqs_[s]->trackingLost; }
  // frame ids count per sequence, as in S separate SlamLoops (an id is only compared with ids of the same sequence: tracking parent,
  // nextStereoFrameMinID — which is a float, DepthMapPixelHypothesis.h:49: ids beyond 2^24 would round)
  static int idOf(int /*s*/, int t) { return t; }
  // one frame of every sequence that is still tracking (images[s] is ignored for lost ones); returns frameToKeyframe per sequence
  // Tracking beside mapping for ALL sequences (the context becomes a pipelined one): per sequence exactly SlamLoop::setPipelined — the
  // mapper one frame behind the tracker, frames tracked on a keyframe the mapper has meanwhile replaced are dropped
  // (SlamSystem.cpp:559-566), the first frame on a new keyframe starts from se3FromSim3(newKeyframe^-1 * lastTrackedFrame) (:913-920).
  // The tracking batch of step t + 1 — a chain of small latency-bound launches — runs beside the shared updateKeyframe launches and the
  // keyframe changes of step t.  Call before the first step().
  // One workgroup per sequence for the coarse levels of a tracking batch (lsdhip_tracker_set_batch_coarse_min_jobs) frees the chip for the
  // mapping launches beside it: worth it from the library's default (24 sequences) where tracking runs beside mapping, from twice as
  // many where nothing runs beside the batch (measured: profiles/r06_notes.md section 21).  tracker.setBatchCoarseMinJobs overrides.
  static int coarseMinJobs(bool pipelined) {
    lsdhip_build_defaults_t d;
    lsdhip_build_defaults(&d);
    return pipelined ? d.batch_coarse_min_jobs : 2 * d.batch_coarse_min_jobs;
  }
  void setPipelined(bool on) {
    if (frameId_ != 0) throw Error(LSDHIP_E_STATE, "SlamLoopBatch::setPipelined: switch before the first step");
    Context::get(w_, h_, K_)->setPipeline(on);
    pipelined_ = on;
    tracker.setBatchCoarseMinJobs(coarseMinJobs(on));
    for (auto& q : seqs_) { q->trackKF = q->keyframe; q->pendingKF.reset(); }
    if (on) tracker.setEnqueueHook([this]() {
      // the tracking batch's launches are queued: now the mapping work of the previous step, then the next step's frames
      flush();
      if (!pendingNext_) return;
      std::vector<int> all;
      for (int s = 0; s < size(); s++) all.push_back(s);
      prefetched_ = makeFrames(all, pendingNext_, frameId_ + 1);
      prefetchedId_ = frameId_ + 1;
    });
    else tracker.setEnqueueHook(nullptr);
  }
  bool pipelined() const { return pipelined_; }
  // nextImages (optional, pipelined loops): the images of the following step() call — their upload / pyramids are queued on the mapping
  // stream AHEAD of this step's mapping work, so that the next tracking batch does not wait behind it
  std::vector<SE3> step(const unsigned char* const* images, const unsigned char* const* nextImages = nullptr) {
    frameId_++;
    const int S = size();
    std::vector<SE3> out((size_t)S);
    std::vector<int> all;
    for (int s = 0; s < S; s++) all.push_back(s);
    std::vector<std::shared_ptr<Frame>> groupFrames;
    if (prefetchedId_ == frameId_ && (int)prefetched_.size() == S) groupFrames = std::move(prefetched_);
    else groupFrames = makeFrames(all, images, frameId_);
    prefetched_.clear();
    prefetchedId_ = -1;
    if (!pipelined_) {
      MapWork w = trackGroup(all, groupFrames, out);
      mapGroup(w);
      return out;
    }
    // ---- pipelined: track, adopt what the mapper did one step ago, then queue [next frames, this step's mapping] ----------------------
    std::vector<int> alive;
    std::vector<std::shared_ptr<Frame>> frames, trackedOn;
    std::vector<TrackingReference*> refs;
    std::vector<Frame*> frs;
    std::vector<SE3> inits;
    for (int s = 0; s < S; s++) {
      Sequence& q = *seqs_[s];
      if (q.trackingLost || !groupFrames[s]) continue;
      alive.push_back(s); frames.push_back(groupFrames[s]); trackedOn.push_back(q.trackKF);
      refs.push_back(&q.reference); frs.push_back(groupFrames[s].get()); inits.push_back(q.lastFrameToKF);
    }
    MapWork work;
    pendingNext_ = nextImages;
    if (alive.empty()) { flush(); pendingNext_ = nullptr; }
    if (!alive.empty()) {
      std::vector<lsdhip_track_result> res;
      std::vector<SE3> est = tracker.trackFrameBatch(refs, frs, inits, res);    // (the enqueue hook queues the previous step's mapping and the next frames)
      pendingNext_ = nullptr;
      for (size_t k = 0; k < alive.size(); k++) {
        Sequence& q = *seqs_[alive[k]];
        q.last = res[k];
        q.numTracked++;
        q.evaluations += res[k].numEvaluations;
        if (res[k].trackingWasGood) q.numTrackedGood++;
        out[alive[k]] = est[k];
        q.newKeyframe = false;
        if (res[k].diverged || (q.numKeyframesFinished > 5 /* INITIALIZATION_PHASE_COUNT */ && !res[k].trackingWasGood)) {
          q.trackingLost = true;
          q.reference.invalidate();
          if (q.map.isValid()) {
            if (q.mappedOnKF >= 5 /* MIN_NUM_MAPPED */) { q.map.finalizeKeyFrame(); q.numKeyframesFinished++; }
            q.map.invalidate();
          }
          continue;
        }
        if (q.pendingKF) {
          // the mapper promoted pendingKF while this frame was being tracked on the old keyframe (SlamSystem.cpp:907-920)
          double p[7];
          check(lsdhip_frame_relative_pose(q.pendingKF->handle(), frames[k]->handle(), p), "lsdhip_frame_relative_pose");
          q.lastFrameToKF = SE3::from7(p);
          q.trackKF = q.pendingKF;
          q.pendingKF.reset();
          q.reference.importFrame(q.trackKF.get());
          q.trackKF->clearDepthHasBeenUpdatedFlag();
        } else {
          q.lastFrameToKF = est[k];
          if (q.trackKF->depthHasBeenUpdatedFlag()) {
            q.reference.importFrame(q.trackKF.get());
            q.trackKF->clearDepthHasBeenUpdatedFlag();
          }
        }
        ++q.sinceKF;
        if (trackedOn[k] != q.keyframe) {        // tracked on the keyframe the mapper has just replaced: popped unmapped (:559-566)
          frames[k]->clear_refPixelWasGood();
          q.numDropped++;
          continue;
        }
        if (q.sinceKF >= kfEvery_) { work.kfChange.emplace_back(alive[k], frames[k]); q.newKeyframe = true; }
        else { work.updMaps.push_back(&q.map); work.updFrames.push_back(frames[k]); work.updSeq.push_back(alive[k]); }
      }
    }
    // queued from the NEXT tracking batch's enqueue hook (or flush()): the host's enqueueing of ~10^2 launches then lies under that
    // batch, and so does the device work
    deferred_ = std::move(work);
    return out;
  }
  // queues the mapping work the last step left (a pipelined loop keeps it back for the next tracking batch's enqueue hook); call after the
  // last step() before reading maps or statistics
  void flush() { MapWork w = std::move(deferred_); deferred_ = MapWork(); mapGroup(w); }
  SE3Tracker tracker;
  bool keepKeyframes = false;         // keep every promoted keyframe alive in its sequence's keyframeLog (validation: rescale factors, point counts)
  bool sharedKeyframeChange = true;   // the keyframe changes of a step in shared launches (DepthMap::changeKeyframeBatch); false: per-sequence
                                      // finalizeKeyFrame + createKeyFrame call chains, dealt to keyframeLanes streams (rounds 2-4)
  int keyframeLanes = 8;      // (per-sequence chains) streams the keyframe changes of one step are dealt to (1: all on the context's stream)

 private:
  struct MapWork {            // what a tracking batch leaves for the mapping side
    std::vector<DepthMap*> updMaps;
    std::vector<std::shared_ptr<Frame>> updFrames;
    std::vector<int> updSeq;
    std::vector<std::pair<int, std::shared_ptr<Frame>>> kfChange;   // (sequence, its new keyframe)
  };
  // new frames of the sequences in `group` that are still tracking (null entries for lost ones)
  std::vector<std::shared_ptr<Frame>> makeFrames(const std::vector<int>& group, const unsigned char* const* images, int frameId) {
    std::vector<int> ids;
    std::vector<const unsigned char*> imgs;
    for (int s : group) if (!seqs_[s]->trackingLost) { ids.push_back(idOf(s, frameId)); imgs.push_back(images[s]); }
    std::vector<std::shared_ptr<Frame>> out(group.size());
    if (ids.empty()) return out;
    std::vector<std::shared_ptr<Frame>> made = Frame::createBatch(w_, h_, K_, ids, imgs.data(), onDevice_);
    size_t k = 0;
    for (size_t i = 0; i < group.size(); i++) if (!seqs_[group[i]]->trackingLost) out[i] = made[k++];
    return out;
  }
  // SlamSystem::trackFrame for every sequence of the group in shared launches + the decisions of doMappingIteration; the device work
  // of the mapping iterations is returned, not queued
  MapWork trackGroup(const std::vector<int>& group, const std::vector<std::shared_ptr<Frame>>& groupFrames, std::vector<SE3>& out) {
    MapWork work;
    std::vector<int> alive;
    std::vector<std::shared_ptr<Frame>> frames;
    for (size_t i = 0; i < group.size(); i++) if (!seqs_[group[i]]->trackingLost && groupFrames[i]) { alive.push_back(group[i]); frames.push_back(groupFrames[i]); }
    if (alive.empty()) return work;
    std::vector<TrackingReference*> refs;
    std::vector<Frame*> frs;
    std::vector<SE3> inits;
    for (size_t k = 0; k < alive.size(); k++) {
      Sequence& q = *seqs_[alive[k]];
      if (q.keyframe->depthHasBeenUpdatedFlag()) {        // the tracking thread's import (SlamSystem.cpp:907-912)
        q.reference.importFrame(q.keyframe.get());
        q.keyframe->clearDepthHasBeenUpdatedFlag();
      }
      refs.push_back(&q.reference);
      frs.push_back(frames[k].get());
      inits.push_back(q.lastFrameToKF);
    }
    std::vector<lsdhip_track_result> res;
    std::vector<SE3> est = tracker.trackFrameBatch(refs, frs, inits, res);
    for (size_t k = 0; k < alive.size(); k++) {
      Sequence& q = *seqs_[alive[k]];
      q.last = res[k];
      q.numTracked++;
      q.evaluations += res[k].numEvaluations;
      if (res[k].trackingWasGood) q.numTrackedGood++;
      out[alive[k]] = est[k];
      q.newKeyframe = false;
      if (res[k].diverged || (q.numKeyframesFinished > 5 /* INITIALIZATION_PHASE_COUNT */ && !res[k].trackingWasGood)) {
        // SlamSystem::trackFrame's lost branch (:946-966) and doMappingIteration's (:809-817), as in SlamLoop::step
        q.trackingLost = true;
        q.reference.invalidate();
        if (q.map.isValid()) {
          if (q.mappedOnKF >= 5 /* MIN_NUM_MAPPED */) { q.map.finalizeKeyFrame(); q.numKeyframesFinished++; }
          q.map.invalidate();
        }
        continue;
      }
      ++q.sinceKF;
      if (q.sinceKF >= kfEvery_) {
        work.kfChange.emplace_back(alive[k], frames[k]);
        q.newKeyframe = true;
      } else {
        work.updMaps.push_back(&q.map);
        work.updFrames.push_back(frames[k]);
        work.updSeq.push_back(alive[k]);
        q.lastFrameToKF = est[k];
      }
    }
    return work;
  }
  // the mapping iterations a tracking batch left: updateKeyframe of all its sequences in shared launches, then the keyframe changes —
  // per-sequence call chains (finalizeKeyFrame + createKeyFrame, ~18 small dependent launches each), independent of each other: on a
  // one-stream context dealt to `keyframeLanes` streams, and queued AFTER the shared launches so that the host's enqueueing lies under them
  void mapGroup(MapWork& work) {
    if (work.updMaps.empty() && work.kfChange.empty()) return;
    std::shared_ptr<Context> ctx = Context::get(w_, h_, K_);
    const int nkf = (int)work.kfChange.size();
    const int lanes = !sharedKeyframeChange && keyframeLanes > 1 && nkf > 0 ? (keyframeLanes < nkf ? keyframeLanes : nkf) : 0;
    // (closed by the destructor too: an exception from one of the calls below must not leave the context inside a lane region)
    struct LaneRegion {
      Context* c; bool open;
      LaneRegion(Context* c_, int n) : c(c_), open(n > 0) { if (open) c->lanesBegin(n); }
      void end() { if (open) { open = false; c->lanesEnd(); } }
      ~LaneRegion() { if (open) { try { c->lanesEnd(); } catch (...) {} } }
    } region(ctx.get(), lanes);
    std::vector<Frame*> updFrames;
    for (auto& f : work.updFrames) updFrames.push_back(f.get());
    DepthMap::updateKeyframeBatch(work.updMaps, updFrames);
    if (sharedKeyframeChange && nkf > 0) {
      // all keyframe changes of the step in six launches
      std::vector<DepthMap*> kfMaps;
      std::vector<Frame*> kfFrames;
      for (int i = 0; i < nkf; i++) { kfMaps.push_back(&seqs_[work.kfChange[i].first]->map); kfFrames.push_back(work.kfChange[i].second.get()); }
      DepthMap::changeKeyframeBatch(kfMaps, kfFrames);
    }
    for (int i = 0; i < nkf; i++) {
      Sequence& q = *seqs_[work.kfChange[i].first];
      const std::shared_ptr<Frame>& frame = work.kfChange[i].second;
      if (lanes) ctx->laneSelect(i % lanes);
      if (!sharedKeyframeChange) q.map.finalizeKeyFrame();
      q.numKeyframesFinished++;
      q.mappedOnKF = 0;
      if (!sharedKeyframeChange) q.map.createKeyFrame(frame.get());
      q.keyframe = frame;
      if (keepKeyframes) q.keyframeLog.push_back(frame);
      q.sinceKF = 0;
      if (pipelined_) {
        q.pendingKF = frame;          // the tracker keeps the old keyframe for one more frame
      } else {
        q.trackKF = frame;
        q.reference.importFrame(q.keyframe.get());
        q.keyframe->clearDepthHasBeenUpdatedFlag();
        q.lastFrameToKF = SE3();
      }
    }
    region.end();
    for (size_t k = 0; k < work.updSeq.size(); k++) {
      Sequence& q = *seqs_[work.updSeq[k]];
      q.mappedOnKF++;
      q.numUpdates++;
      work.updFrames[k]->clear_refPixelWasGood();
    }
  }
  bool pipelined_ = false;
  std::vector<std::shared_ptr<Frame>> prefetched_;
  int prefetchedId_ = -1;
  const unsigned char* const* pendingNext_ = nullptr;
  MapWork deferred_;
  int w_, h_;
  Mat3f K_;
  bool onDevice_;
  int kfEvery_, frameId_ = 0;
  std::vector<std::unique_ptr<Sequence>> seqs_;
};

}  // namespace lsd_slam_hip
#endif  // LSD_SLAM_HIP_HPP
