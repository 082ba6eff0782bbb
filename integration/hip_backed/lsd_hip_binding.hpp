// integration/hip_backed — INTEGRATION.md path B as code.
//
// These translation units define the member functions of the REFERENCE's OWN classes lsd_slam::SE3Tracker
// (lsd_slam_core/src/Tracking/SE3Tracker.h:41-93) and lsd_slam::DepthMap (lsd_slam_core/src/DepthEstimation/DepthMap.h:47-98) on top
// of the C ABI of liblsdhip.so (include/lsdhip.h).  The headers are used UNMODIFIED: a lsd_slam_core whose build lists
// integration/hip_backed/SE3Tracker_hip.cpp and DepthMap_hip.cpp instead of src/Tracking/SE3Tracker.cpp and
// src/DepthEstimation/DepthMap.cpp (integration/lsd_slam_core.patch: two lines of CMakeLists.txt + the link line) runs SlamSystem,
// TrackableKeyFrameSearch, Relocalizer ... unchanged, with trackFrame / updateKeyframe / createKeyFrame / finalizeKeyFrame executing on
// the MI355X.  Nothing here is copied from the files it replaces: the reference's members are declared by its headers, and what these
// definitions do is call the C ABI and mirror, on the host objects, the side effects the reference documents.
//
// The reference's Frame stays what it is (host pyramids, FrameMemory, pose graph links).  Because its header cannot grow a member here,
// the device handle of a Frame lives in a side table keyed by the Frame's address and validated by (id, image pointer); a maintainer
// would add `lsdhip_frame* hip` to Frame and drop the table.  Keyframe depth reaches the device as the keyframe's
// level-0 idepth / idepthVar planes (what Frame::setDepth leaves); the device rebuilds the pyramid (bit-identical to
// Frame::buildIDepthAndIDepthVar, tests/test_hip_vs_ref_gpu.py).
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "lsdhip.h"
#include "lsd_slam_hip_sophus.hpp"      // toHip / fromHip / toHipSim3 / fromHipSim3 (include/)

#include "DataStructures/Frame.h"
#include "DataStructures/FramePoseStruct.h"
#include "util/settings.h"
#include "util/SophusUtil.h"

namespace lsd_slam_hipbind {

inline void check(int rc, const char* what) {
  if (rc < 0) throw std::runtime_error(std::string(what) + ": " + lsdhip_last_error());
}

// One device context per (w, h, K, the mutable globals of util/settings.cpp:77-88 the path reads): what the reference's objects share
// through their constructor arguments and through those globals.
inline lsdhip_ctx* context_for(int w, int h, const Eigen::Matrix3f& K) {
  static std::mutex mu;
  typedef std::tuple<int, int, float, float, float, float, float, float, float, int, int, int> Key;
  static std::map<Key, lsdhip_ctx*> registry;
  std::lock_guard<std::mutex> lock(mu);
  const Key key(w, h, K(0, 0), K(1, 1), K(0, 2), K(1, 2), lsd_slam::minUseGrad, lsd_slam::cameraPixelNoise2, lsd_slam::depthSmoothingFactor,
                (int)lsd_slam::allowNegativeIdepths, (int)lsd_slam::useSubpixelStereo, (int)lsd_slam::useAffineLightningEstimation);
  auto it = registry.find(key);
  if (it != registry.end()) return it->second;
  lsdhip_params p;
  p.minUseGrad = lsd_slam::minUseGrad; p.cameraPixelNoise2 = lsd_slam::cameraPixelNoise2; p.depthSmoothingFactor = lsd_slam::depthSmoothingFactor;
  p.allowNegativeIdepths = lsd_slam::allowNegativeIdepths; p.useSubpixelStereo = lsd_slam::useSubpixelStereo;
  p.useAffineLightningEstimation = lsd_slam::useAffineLightningEstimation;
  const float K4[4] = {K(0, 0), K(1, 1), K(0, 2), K(1, 2)};
  lsdhip_ctx* c = nullptr;
  check(lsdhip_ctx_create(0, w, h, K4, &p, &c), "lsdhip_ctx_create");
  registry[key] = c;
  return c;
}

// ---- device mirrors of lsd_slam::Frame ---------------------------------------------------------------------------------
struct FrameMirror {
  lsdhip_frame* h = nullptr;
  lsdhip_ctx* ctx = nullptr;
  int id = -1;
  const float* image = nullptr;       // the host frame's level-0 image buffer the mirror was made from
  uint64_t depthHash = 0;             // of the level-0 (idepth, idepthVar) planes last handed to the device (0: none)
  uint64_t lastUse = 0;
  // the tracking result the device left on the mirror, as it was written to the host frame (through Sophus, whose constructors
  // re-normalise the quaternion): while the host frame still holds exactly this, the mirror's own pose is the original and is kept
  bool poseMirrored = false;
  double hostPose8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const lsd_slam::Frame* poseParent = nullptr;
  float hostResidual = 0;
};
inline std::map<const lsd_slam::Frame*, FrameMirror>& mirrors() { static std::map<const lsd_slam::Frame*, FrameMirror> m; return m; }
inline std::mutex& mirror_mutex() { static std::mutex m; return m; }

inline uint64_t hash_planes(const float* a, const float* b, size_t n) {
  uint64_t h = 1469598103934665603ull;
  const uint32_t* pa = (const uint32_t*)a;
  const uint32_t* pb = (const uint32_t*)b;
  for (size_t i = 0; i < n; i++) { h = (h ^ pa[i]) * 1099511628211ull; h = (h ^ pb[i]) * 1099511628211ull; }
  return h ? h : 1;
}

// the device frame of `f` on `ctx` (created on first use: image upload + pyramids, lsdhip_frame_create)
inline lsdhip_frame* device_frame(lsd_slam::Frame* f, lsdhip_ctx* ctx) {
  static uint64_t tick = 0;
  std::lock_guard<std::mutex> lock(mirror_mutex());
  auto& tab = mirrors();
  const float* img = f->image(0);
  FrameMirror& m = tab[f];
  if (m.h && (m.id != f->id() || m.image != img || m.ctx != ctx)) { lsdhip_frame_destroy(m.h); m = FrameMirror(); }   // the address was recycled
  if (!m.h) {
    const size_t n = (size_t)f->width(0) * f->height(0);
    std::vector<uint8_t> gray(n);
    for (size_t i = 0; i < n; i++) {
      const float v = img[i];
      if (!(v >= 0.0f && v <= 255.0f) || v != (float)(int)v) throw std::runtime_error("hip-backed binding: frames must come from 8-bit images (Frame.cpp:35-48)");
      gray[i] = (uint8_t)(int)v;
    }
    check(lsdhip_frame_create(ctx, f->id(), gray.data(), &m.h), "lsdhip_frame_create");
    m.ctx = ctx; m.id = f->id(); m.image = img; m.depthHash = 0;
    // keep the table bounded: frames that were destroyed on the host leave their mirror behind (no hook in the unmodified header)
    if (tab.size() > 96) {
      const lsd_slam::Frame* oldest = nullptr;
      uint64_t t = ~0ull;
      for (auto& kv : tab) if (kv.first != f && kv.second.lastUse < t) { t = kv.second.lastUse; oldest = kv.first; }
      if (oldest) { lsdhip_frame_destroy(tab[oldest].h); tab.erase(oldest); }
    }
  }
  tab[f].lastUse = ++tick;
  return tab[f].h;
}
// hands the host frame's level-0 depth planes to its mirror when they changed since the last hand-over
inline void sync_depth(lsd_slam::Frame* f, lsdhip_ctx* ctx) {
  lsdhip_frame* h = device_frame(f, ctx);
  if (!f->hasIDepthBeenSet()) return;
  const size_t n = (size_t)f->width(0) * f->height(0);
  const float* id = f->idepth(0);
  const float* var = f->idepthVar(0);
  const uint64_t hv = hash_planes(id, var, n);
  std::lock_guard<std::mutex> lock(mirror_mutex());
  FrameMirror& m = mirrors()[f];
  if (m.depthHash == hv) return;
  check(lsdhip_frame_set_depth_planes(h, id, var), "lsdhip_frame_set_depth_planes");
  m.depthHash = hv;
}
// after a DepthMap call whose device side ran Frame::setDepth on the mirror AND whose host side ran it on the host frame with the same
// hypotheses: both hold the same planes
inline void depth_in_sync(lsd_slam::Frame* f) {
  if (!f->hasIDepthBeenSet()) return;
  const uint64_t hv = hash_planes(f->idepth(0), f->idepthVar(0), (size_t)f->width(0) * f->height(0));
  std::lock_guard<std::mutex> lock(mirror_mutex());
  auto it = mirrors().find(f);
  if (it != mirrors().end()) it->second.depthHash = hv;
}
// remembers what trackFrame / createKeyFrame wrote into the host frame's pose from the device frame's
inline void note_pose(lsd_slam::Frame* f, const double hostPose8[8], const lsd_slam::Frame* parent, float residual) {
  std::lock_guard<std::mutex> lock(mirror_mutex());
  FrameMirror& m = mirrors()[f];
  m.poseMirrored = true;
  std::memcpy(m.hostPose8, hostPose8, sizeof(m.hostPose8));
  m.poseParent = parent;
  m.hostResidual = residual;
}
// the host frame's (pose relative to `parent`, tracking residual) to its mirror — unless they are what the device produced itself
inline void push_pose(lsd_slam::Frame* f, lsdhip_frame* h, const double p8[8], const lsd_slam::Frame* parent, lsdhip_frame* parentHandle, float residual) {
  {
    std::lock_guard<std::mutex> lock(mirror_mutex());
    const FrameMirror& m = mirrors()[f];
    if (m.poseMirrored && m.poseParent == parent && m.hostResidual == residual && std::memcmp(m.hostPose8, p8, sizeof(m.hostPose8)) == 0) return;
  }
  check(lsdhip_frame_set_pose(h, p8, parentHandle, residual), "lsdhip_frame_set_pose");
  std::lock_guard<std::mutex> lock(mirror_mutex());
  mirrors()[f].poseMirrored = false;
}
inline void push_mask(lsd_slam::Frame* f, lsdhip_frame* h) {
  const bool* m = f->refPixelWasGoodNoCreate();
  if (m) check(lsdhip_frame_set_wasgood(h, (const uint8_t*)m), "lsdhip_frame_set_wasgood");
  else check(lsdhip_frame_clear_wasgood(h), "lsdhip_frame_clear_wasgood");
}
inline void sim3_to8(const Sim3& S, double p[8]) {
  const lsd_slam_hip::Sim3 s = lsd_slam_hip::toHipSim3(S);
  for (int i = 0; i < 4; i++) p[i] = s.q[i];
  for (int i = 0; i < 3; i++) p[4 + i] = s.t[i];
  p[7] = s.s;
}
inline Sim3 sim3_from8(const double p[8]) {
  lsd_slam_hip::Sim3 s;
  for (int i = 0; i < 4; i++) s.q[i] = p[i];
  for (int i = 0; i < 3; i++) s.t[i] = p[4 + i];
  s.s = p[7];
  return lsd_slam_hip::fromHipSim3<Sim3>(s);
}

}  // namespace lsd_slam_hipbind
