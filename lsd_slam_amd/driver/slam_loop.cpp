// liblsdhip_driver.so — the C++ host side of the sequence loop (include/lsd_slam_hip.hpp) behind a C interface.
// Plain host C++ (g++): everything that touches pixels is a liblsdhip.so call.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <string>

#include "../../include/lsd_slam_hip.hpp"
#include "../../include/lsdhip_driver.h"

using namespace lsd_slam_hip;

static thread_local std::string g_err;
typedef void* NcclComm;   // ncclComm_t (rccl.h), opaque

struct lsdloop {
  std::unique_ptr<SlamLoop> loop;
  std::shared_ptr<Context> ctx;
  lsdloop_stats st{};
  double wall = 0;                       // wall time inside lsdloop_run since the last reset
  double gpu0[3] = {0, 0, 0};            // DepthMap GPU times at the last reset
  float* ring = nullptr;                 // lsdloop_set_keyframe_ring
  int ring_slots = 0;
  long long ring_count = 0;
  NcclComm comm = nullptr;               // lsdloop_comm_init
  int comm_rank = 0, comm_world = 1;
};

// ---- RCCL, bound by name (prototypes: /opt/rocm/include/rccl/rccl.h:43,187,220,260,339 and the ncclSend / ncclRecv /
// ncclGroupStart / ncclGroupEnd declarations there; ncclFloat = 7, ncclSuccess = 0) -------------------------------------------
namespace {
struct NcclId { char internal[128]; };
struct Rccl {
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, NcclComm, void*) = nullptr;
  int (*Recv)(void*, size_t, int, int, NcclComm, void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
  std::string why;
};
Rccl& rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r;
  tried = true;
  void* h = RTLD_DEFAULT;                       // the RCCL torch.distributed already loaded, if any
  if (!dlsym(h, "ncclCommInitRank")) {
    h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { r.why = "no RCCL in the process and librccl.so not found"; return r; }
  }
  auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) r.why = std::string("RCCL symbol missing: ") + n; return p; };
  r.GetUniqueId = (int (*)(NcclId*))sym("ncclGetUniqueId");
  r.CommInitRank = (int (*)(NcclComm*, int, NcclId, int))sym("ncclCommInitRank");
  r.CommDestroy = (int (*)(NcclComm))sym("ncclCommDestroy");
  r.GroupStart = (int (*)())sym("ncclGroupStart");
  r.GroupEnd = (int (*)())sym("ncclGroupEnd");
  r.Send = (int (*)(const void*, size_t, int, int, NcclComm, void*))sym("ncclSend");
  r.Recv = (int (*)(void*, size_t, int, int, NcclComm, void*))sym("ncclRecv");
  r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv;
  return r;
}
}  // namespace

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

extern "C" const char* lsdloop_last_error(void) { return g_err.c_str(); }

extern "C" int lsdloop_create(int device, int w, int h, const float K4[4], const uint8_t* first_image, int images_on_device,
                              const float* gt_depth0_host, int kf_every, lsdloop** out) {
  if (!K4 || !first_image || !gt_depth0_host || !out || kf_every < 1) return LSDHIP_E_ARG;
  try {
    Context::defaultDevice() = device;
    Mat3f K = Mat3f::intrinsics(K4[0], K4[1], K4[2], K4[3]);
    std::unique_ptr<lsdloop> l(new lsdloop());
    l->ctx = Context::get(w, h, K, device);
    l->loop.reset(new SlamLoop(w, h, K, first_image, images_on_device != 0, gt_depth0_host, kf_every));
    l->ctx->setAsync(true);   // mapping kernels are enqueued behind the tracker's; the host never waits for them
    *out = l.release();
    return LSDHIP_OK;
  } catch (const Error& e) {
    g_err = e.what();
    return e.status;
  } catch (const std::exception& e) {
    g_err = e.what();
    return LSDHIP_E_STATE;
  }
}
extern "C" void lsdloop_destroy(lsdloop* l) {
  if (l && l->comm) { rccl().CommDestroy(l->comm); l->comm = nullptr; }
  delete l;
}

extern "C" int lsdloop_run(lsdloop* l, const uint8_t* const* images, int n, int stop_at_keyframe, double* out7) {
  if (!l || !images || n < 0) return LSDHIP_E_ARG;
  try {
    int done = 0;
    static const bool prefetch = !std::getenv("LSDHIP_NO_PREFETCH");   // developer switch for A/B timing
    const double t0 = now_s();
    for (int i = 0; i < n; i++) {
      SlamLoop& L = *l->loop;
      const long upd0 = L.numUpdates;
      const long ev0 = L.evaluations;
      const long la0 = L.launches;
      const long good0 = L.numTrackedGood;
      long lev0[5];
      for (int k = 0; k < 5; k++) lev0[k] = L.levelEvaluations[k];
      SE3 est = L.step(images[i], [](double) {}, (prefetch && i + 1 < n) ? images[i + 1] : nullptr);
      l->st.frames++;
      l->st.updates += L.numUpdates - upd0;
      l->st.evaluations += L.evaluations - ev0;
      l->st.track_launches += L.launches - la0;
      l->st.tracked_good += L.numTrackedGood - good0;
      for (int k = 0; k < 5; k++) l->st.level_evaluations[k] += L.levelEvaluations[k] - lev0[k];
      if (L.newKeyframe) l->st.keyframes++;
      if (out7) est.to7(out7 + 7 * (size_t)i);
      done++;
      if (stop_at_keyframe && L.newKeyframe) break;
    }
    l->ctx->synchronize();   // one synchronisation per batch: the enqueued mapping work belongs to this batch's time
    l->wall += now_s() - t0;
    return done;
  } catch (const Error& e) {
    g_err = e.what();
    return e.status > 0 ? -100 - e.status : e.status;
  } catch (const std::exception& e) {
    g_err = e.what();
    return LSDHIP_E_STATE;
  }
}
// Mapping runs asynchronously behind tracking on one stream, so the split is made from GPU event times: map / keyframe
// = HIP-event time of the DepthMap calls; track = wall time of the batches minus those.
extern "C" int lsdloop_get_stats(lsdloop* l, lsdloop_stats* out) {
  if (!l || !out) return LSDHIP_E_ARG;
  try {
    double ms[3];
    long long calls[3];
    l->loop->map.gpuTimes(ms, calls);
    l->st.seconds_map = (ms[0] - l->gpu0[0]) * 1e-3;
    l->st.seconds_keyframe = ((ms[1] - l->gpu0[1]) + (ms[2] - l->gpu0[2])) * 1e-3;
    l->st.seconds_track = l->wall - l->st.seconds_map - l->st.seconds_keyframe;
    *out = l->st;
    return LSDHIP_OK;
  } catch (const Error& e) { g_err = e.what(); return e.status; }
}
extern "C" int lsdloop_reset_stats(lsdloop* l) {
  if (!l) return LSDHIP_E_ARG;
  try {
    long long calls[3];
    l->loop->map.gpuTimes(l->gpu0, calls);
    l->st = lsdloop_stats{};
    l->wall = 0;
    return LSDHIP_OK;
  } catch (const Error& e) { g_err = e.what(); return e.status; }
}
extern "C" int lsdloop_copy_keyframe_planes(lsdloop* l, float* idepth_dev, float* var_dev) {
  if (!l) return LSDHIP_E_ARG;
  try { l->loop->map.copyPlanesToDevice(idepth_dev, var_dev); return LSDHIP_OK; }
  catch (const Error& e) { g_err = e.what(); return e.status; }
}
extern "C" int lsdloop_set_keyframe_ring(lsdloop* l, float* ring_dev, int slots) {
  if (!l || (ring_dev && slots < 1)) return LSDHIP_E_ARG;
  l->ring = ring_dev;
  l->ring_slots = slots;
  l->ring_count = 0;
  if (!ring_dev) { l->loop->onKeyframeFinished = nullptr; return LSDHIP_OK; }
  const size_t plane = (size_t)l->ctx->width() * l->ctx->height();
  l->loop->onKeyframeFinished = [l, plane](Frame&, DepthMap& map) {
    float* slot = l->ring + (size_t)(l->ring_count % l->ring_slots) * 2 * plane;
    map.copyPlanesToDevice(slot, slot + plane);
    l->ring_count++;
  };
  return LSDHIP_OK;
}
extern "C" int lsdloop_keep_keyframes(lsdloop* l, int on) {
  if (!l) return LSDHIP_E_ARG;
  l->loop->keepKeyframes = on != 0;
  if (!on) l->loop->keyframeLog.clear();
  return LSDHIP_OK;
}
extern "C" int lsdloop_keyframe_log(lsdloop* l, double* scales_out, long long* points_out, int max) {
  if (!l) return LSDHIP_E_ARG;
  try {
    const auto& log = l->loop->keyframeLog;
    for (int i = 0; i < (int)log.size() && i < max; i++) {
      if (scales_out) scales_out[i] = log[i]->thisToParent_raw().s;
      if (points_out) points_out[i] = log[i]->stats().numPoints;
    }
    return (int)log.size();
  } catch (const Error& e) { g_err = e.what(); return e.status; }
}
extern "C" int lsdloop_set_live_queue(lsdloop* l, int frames) {
  if (!l || frames < 1) return LSDHIP_E_ARG;
  l->loop->liveQueueLength = frames;
  return LSDHIP_OK;
}
extern "C" int lsdloop_set_persistent(lsdloop* l, int max_strips) {
  if (!l) return LSDHIP_E_ARG;
  try { l->loop->tracker.setPersistent(max_strips); return LSDHIP_OK; }
  catch (const Error& e) { g_err = e.what(); return e.status; }
}
extern "C" int lsdloop_set_speculation(lsdloop* l, int trials, int finest_level_workgroups) {
  if (!l) return LSDHIP_E_ARG;
  try { l->loop->tracker.setSpeculation(trials, finest_level_workgroups); return LSDHIP_OK; }
  catch (const Error& e) { g_err = e.what(); return e.status; }
}
extern "C" int lsdloop_observe_time(lsdloop* l, double* ms_out, long long* calls_out) {
  if (!l) return LSDHIP_E_ARG;
  return lsdhip_depth_observe_time(l->loop->map.handle(), ms_out, calls_out);
}
extern "C" int lsdloop_comm_unique_id(unsigned char out128[128]) {
  Rccl& R = rccl();
  if (!R.ok) { g_err = "RCCL unavailable: " + R.why; return LSDHIP_E_STATE; }
  NcclId id;
  const int rc = R.GetUniqueId(&id);
  if (rc != 0) { g_err = std::string("ncclGetUniqueId: ") + (R.GetErrorString ? R.GetErrorString(rc) : "?"); return LSDHIP_E_HIP; }
  std::memcpy(out128, id.internal, 128);
  return LSDHIP_OK;
}
extern "C" int lsdloop_comm_init(lsdloop* l, const unsigned char id128[128], int rank, int world) {
  if (!l || !id128 || rank < 0 || rank >= world) return LSDHIP_E_ARG;
  Rccl& R = rccl();
  if (!R.ok) { g_err = "RCCL unavailable: " + R.why; return LSDHIP_E_STATE; }
  if (l->comm) { R.CommDestroy(l->comm); l->comm = nullptr; }
  NcclId id;
  std::memcpy(id.internal, id128, 128);
  const int rc = R.CommInitRank(&l->comm, world, id, rank);
  if (rc != 0) { l->comm = nullptr; g_err = std::string("ncclCommInitRank: ") + (R.GetErrorString ? R.GetErrorString(rc) : "?"); return LSDHIP_E_HIP; }
  l->comm_rank = rank;
  l->comm_world = world;
  return LSDHIP_OK;
}
extern "C" int lsdloop_comm_destroy(lsdloop* l) {
  if (!l) return LSDHIP_E_ARG;
  if (l->comm) { rccl().CommDestroy(l->comm); l->comm = nullptr; }
  return LSDHIP_OK;
}
extern "C" int lsdloop_gather_keyframes(lsdloop* l, int count, int root, float* recv_dev, long long stride_floats) {
  if (!l || count < 0 || !l->ring) return LSDHIP_E_ARG;
  if (count == 0) return LSDHIP_OK;
  if (count > l->ring_slots) { g_err = "gather: more keyframes than ring slots"; return LSDHIP_E_ARG; }
  const size_t n = (size_t)count * 2 * (size_t)l->ctx->width() * l->ctx->height();   // floats per rank
  void* stream = lsdhip_ctx_stream(l->ctx->handle());
  try {
    if (l->comm_world == 1 || !l->comm) {
      if (l->comm_world != 1) { g_err = "gather: no communicator (lsdloop_comm_init)"; return LSDHIP_E_STATE; }
      if (recv_dev) check(lsdhip_ctx_copy_dev(l->ctx->handle(), recv_dev, l->ring, n * sizeof(float)), "lsdhip_ctx_copy_dev");
      return LSDHIP_OK;
    }
    Rccl& R = rccl();
    auto nc = [&](int rc, const char* what) { if (rc != 0) throw Error(LSDHIP_E_HIP, std::string(what) + ": " + (R.GetErrorString ? R.GetErrorString(rc) : "?")); };
    if (l->comm_rank == root) {
      if (!recv_dev) return LSDHIP_E_ARG;
      check(lsdhip_ctx_copy_dev(l->ctx->handle(), recv_dev + (size_t)root * stride_floats, l->ring, n * sizeof(float)), "lsdhip_ctx_copy_dev");
      nc(R.GroupStart(), "ncclGroupStart");
      for (int r = 0; r < l->comm_world; r++)
        if (r != root) nc(R.Recv(recv_dev + (size_t)r * stride_floats, n, 7 /* ncclFloat */, r, l->comm, stream), "ncclRecv");
      nc(R.GroupEnd(), "ncclGroupEnd");
    } else {
      nc(R.GroupStart(), "ncclGroupStart");
      nc(R.Send(l->ring, n, 7 /* ncclFloat */, root, l->comm, stream), "ncclSend");
      nc(R.GroupEnd(), "ncclGroupEnd");
    }
    return LSDHIP_OK;
  } catch (const Error& e) { g_err = e.what(); return e.status; }
}
extern "C" long long lsdloop_keyframes_exported(lsdloop* l) { return l ? l->ring_count : -1; }
extern "C" void* lsdloop_ctx(lsdloop* l) { return l ? (void*)l->ctx->handle() : nullptr; }
