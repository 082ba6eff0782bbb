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
  long long calls0[3] = {0, 0, 0};       // ... and the number of calls they were measured on (the calls are sampled)
  float* ring = nullptr;                 // lsdloop_set_keyframe_ring
  int ring_slots = 0;
  long long ring_count = 0;
  NcclComm comm = nullptr;               // lsdloop_comm_init
  int comm_rank = 0, comm_world = 1;
  std::vector<int> last_counts;          // every rank's count of the last RCCL gather (lsdloop_gather_counts)
  int* counts_dev = nullptr;             // [world + 1]: every rank's keyframe count of the gather being issued (+ this rank's own, last)
  // second transport for the gather (processes of one node, no RCCL): the ROOT owns an IPC mailbox — flag block (fail word, per rank:
  // ready, consumed, count) followed by world x ring_slots x 2 x w x h floats — and the other ranks map it
  bool ipc = false;
  int ipc_root = 0;
  char* mailbox = nullptr;               // the root's allocation (own or mapped)
  int* ipc_fail = nullptr;               // non-root: a local fail word for its bounded waits
  long long gathers = 0;
  float* ipc_data() const { return (float*)(mailbox + 4096); }
  int* ipc_flag(int rank, int which) const { return (int*)(mailbox + 64 + (size_t)rank * 16 + (size_t)which * 4); }   // 0 ready, 1 consumed, 2 count
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
  int (*AllGather)(const void*, void*, size_t, int, NcclComm, void*) = nullptr;
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
  r.AllGather = (int (*)(const void*, void*, size_t, int, NcclComm, void*))sym("ncclAllGather");
  r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.AllGather;
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
    if (const char* e = std::getenv("LSDHIP_KF_SHARED")) l->loop->sharedKeyframeChange = e[0] != '0';   // developer A/B of round 5
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
  if (!l) return;
  if (l->comm) { rccl().CommDestroy(l->comm); l->comm = nullptr; }
  if (l->ctx) {
    lsdhip_ctx* c = l->ctx->handle();
    (void)lsdhip_ctx_synchronize(c);
    if (l->mailbox) { if (l->ipc && l->comm_rank != l->ipc_root) (void)lsdhip_ctx_ipc_close(c, l->mailbox); else (void)lsdhip_ctx_free_dev(c, l->mailbox); }
    if (l->ipc_fail) (void)lsdhip_ctx_free_dev(c, l->ipc_fail);
    if (l->counts_dev) (void)lsdhip_ctx_free_dev(c, l->counts_dev);
  }
  delete l;
}

// ---- S sequences sharing one GPU ------------------------------------------------------------------------------------------------------
struct lsdloopbatch {
  std::shared_ptr<Context> ctx;
  std::unique_ptr<lsd_slam_hip::SlamLoopBatch> loop;
  std::vector<long long> keyframes;
};
extern "C" int lsdloopbatch_create(int device, int w, int h, const float K4[4], int S, const uint8_t* const* first_images, int images_on_device,
                                   const float* const* gt_depth0_host, int kf_every, lsdloopbatch** out) {
  if (!K4 || !first_images || !out || kf_every < 1 || S < 1) return LSDHIP_E_ARG;
  try {
    Context::defaultDevice() = device;
    Mat3f K = Mat3f::intrinsics(K4[0], K4[1], K4[2], K4[3]);
    std::unique_ptr<lsdloopbatch> l(new lsdloopbatch());
    l->ctx = Context::get(w, h, K, device);
    l->ctx->setPipeline(false);
    l->loop.reset(new lsd_slam_hip::SlamLoopBatch(w, h, K, S, first_images, images_on_device != 0, gt_depth0_host, kf_every));
    if (const char* e = std::getenv("LSDHIP_KF_LANES")) l->loop->keyframeLanes = std::atoi(e);   // developer A/B (1: keyframe changes on one stream)
    if (const char* e = std::getenv("LSDHIP_KF_SHARED")) l->loop->sharedKeyframeChange = e[0] != '0';   // developer A/B of round 5
    l->ctx->setAsync(true);
    l->keyframes.assign((size_t)S, 0);
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
extern "C" void lsdloopbatch_destroy(lsdloopbatch* l) {
  if (!l) return;
  if (l->ctx) (void)lsdhip_ctx_synchronize(l->ctx->handle());
  std::shared_ptr<Context> ctx = l->ctx;
  delete l;
  if (ctx) { try { ctx->setPipeline(false); } catch (...) {} }   // the context is shared by (w, h, K): leave it as a one-stream context
}
extern "C" int lsdloopbatch_run(lsdloopbatch* l, const uint8_t* const* images, int n, double* out7) {
  if (!l || !images || n < 0) return LSDHIP_E_ARG;
  try {
    const int S = l->loop->size();
    for (int t = 0; t < n; t++) {
      std::vector<SE3> est = l->loop->step(images + (size_t)t * S, t + 1 < n ? images + (size_t)(t + 1) * S : nullptr);
      for (int s = 0; s < S; s++) {
        if (l->loop->sequence(s).newKeyframe) l->keyframes[s]++;
        if (out7) est[s].to7(out7 + 7 * ((size_t)t * S + s));
      }
    }
    l->loop->flush();     // (a pipelined loop keeps the last step's mapping work back for the next tracking batch)
    return n;
  } catch (const Error& e) {
    g_err = e.what();
    return e.status < 0 ? e.status : LSDHIP_E_STATE;
  } catch (const std::exception& e) {
    g_err = e.what();
    return LSDHIP_E_STATE;
  }
}
extern "C" int lsdloopbatch_get_stats(lsdloopbatch* l, long long* out) {
  if (!l || !out) return LSDHIP_E_ARG;
  for (int s = 0; s < l->loop->size(); s++) {
    const lsd_slam_hip::SlamLoopBatch::Sequence& q = l->loop->sequence(s);
    long long* o = out + 6 * (size_t)s;
    o[0] = q.numTracked; o[1] = q.numTrackedGood; o[2] = q.numUpdates; o[3] = l->keyframes[s]; o[4] = q.evaluations; o[5] = q.trackingLost ? 1 : 0;
  }
  return LSDHIP_OK;
}
extern "C" void* lsdloopbatch_ctx(lsdloopbatch* l) { return l ? (void*)l->ctx->handle() : nullptr; }
extern "C" int lsdloopbatch_set_coarse_min_jobs(lsdloopbatch* l, int min_jobs) {
  if (!l) return LSDHIP_E_ARG;
  try { l->loop->tracker.setBatchCoarseMinJobs(min_jobs); return LSDHIP_OK; }
  catch (const Error& e) { g_err = e.what(); return e.status < 0 ? e.status : LSDHIP_E_STATE; }
}
extern "C" int lsdloopbatch_set_pipeline(lsdloopbatch* l, int on) {
  if (!l) return LSDHIP_E_ARG;
  try {
    l->loop->setPipelined(on != 0);
    return LSDHIP_OK;
  } catch (const Error& e) {
    g_err = e.what();
    return e.status < 0 ? e.status : LSDHIP_E_STATE;
  }
}
extern "C" long long lsdloopbatch_dropped(lsdloopbatch* l, int s) {
  if (!l || s < 0 || s >= l->loop->size()) return -1;
  return l->loop->sequence(s).numDropped;
}
extern "C" int lsdloopbatch_keep_keyframes(lsdloopbatch* l, int on) {
  if (!l) return LSDHIP_E_ARG;
  l->loop->keepKeyframes = on != 0;
  if (!on) for (int s = 0; s < l->loop->size(); s++) l->loop->sequence(s).keyframeLog.clear();
  return LSDHIP_OK;
}
extern "C" int lsdloopbatch_keyframe_log(lsdloopbatch* l, int s, double* scales_out, long long* points_out, int max) {
  if (!l || s < 0 || s >= l->loop->size()) return LSDHIP_E_ARG;
  try {
    const auto& log = l->loop->sequence(s).keyframeLog;
    for (int i = 0; i < (int)log.size() && i < max; i++) {
      if (scales_out) scales_out[i] = log[i]->thisToParent_raw().s;
      if (points_out) points_out[i] = log[i]->stats().numPoints;
    }
    return (int)log.size();
  } catch (const Error& e) { g_err = e.what(); return e.status < 0 ? e.status : LSDHIP_E_STATE; }
}
extern "C" int lsdloopbatch_last_result(lsdloopbatch* l, int s, lsdhip_track_result* out) {
  if (!l || !out || s < 0 || s >= l->loop->size()) return LSDHIP_E_ARG;
  *out = l->loop->sequence(s).last;
  return LSDHIP_OK;
}
extern "C" int lsdloopbatch_download_map(lsdloopbatch* l, int s, lsdhip_hypothesis* out) {
  if (!l || !out || s < 0 || s >= l->loop->size()) return LSDHIP_E_ARG;
  try {
    const std::vector<lsdhip_hypothesis> m = l->loop->sequence(s).map.currentDepthMap();
    std::memcpy(out, m.data(), m.size() * sizeof(lsdhip_hypothesis));
    return LSDHIP_OK;
  } catch (const Error& e) { g_err = e.what(); return e.status < 0 ? e.status : LSDHIP_E_STATE; }
}
extern "C" int lsdloopbatch_set_keyframe_phases(lsdloopbatch* l, const int* phase) {
  if (!l || !phase) return LSDHIP_E_ARG;
  try {
    l->loop->setKeyframePhases(std::vector<int>(phase, phase + l->loop->size()));
    return LSDHIP_OK;
  } catch (const Error& e) {
    g_err = e.what();
    return e.status < 0 ? e.status : LSDHIP_E_STATE;
  }
}

extern "C" int lsdloop_run(lsdloop* l, const uint8_t* const* images, int n, int stop_at_keyframe, double* out7) {
  if (!l || !images || n < 0) return LSDHIP_E_ARG;
  try {
    int done = 0;
    long updSeen = l->loop->numUpdates;
    static const bool prefetch = !std::getenv("LSDHIP_NO_PREFETCH");   // developer switch for A/B timing
    const double t0 = now_s();
    for (int i = 0; i < n; i++) {
      SlamLoop& L = *l->loop;
      const long ev0 = L.evaluations;
      const long la0 = L.launches;
      const long good0 = L.numTrackedGood;
      const long drop0 = L.numDropped;
      long lev0[5];
      for (int k = 0; k < 5; k++) lev0[k] = L.levelEvaluations[k];
      SE3 est = L.step(images[i], [](double) {}, (prefetch && i + 1 < n) ? images[i + 1] : nullptr);
      l->st.frames++;
      l->st.updates += L.numUpdates - updSeen;     // (a pipelined loop counts a frame's update when it is queued: one step later)
      updSeen = L.numUpdates;
      l->st.evaluations += L.evaluations - ev0;
      l->st.track_launches += L.launches - la0;
      l->st.tracked_good += L.numTrackedGood - good0;
      l->st.dropped += L.numDropped - drop0;
      for (int k = 0; k < 5; k++) l->st.level_evaluations[k] += L.levelEvaluations[k] - lev0[k];
      if (L.newKeyframe) l->st.keyframes++;
      if (out7) est.to7(out7 + 7 * (size_t)i);
      done++;
      if (stop_at_keyframe && L.newKeyframe) break;
    }
    l->loop->flushDeferredMapping();
    {
      const long upd = l->loop->numUpdates - updSeen;
      l->st.updates += upd;
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
    // the DepthMap calls are timed by sampling (every 8th updateKeyframe, every 2nd createKeyFrame / finalizeKeyFrame): mean of
    // the bracketed calls since the last reset x the calls the loop made
    auto scaled = [&](int k, long n) {
      const long long sampled = calls[k] - l->calls0[k];
      if (sampled > 0) return (ms[k] - l->gpu0[k]) / (double)sampled * (double)n * 1e-3;
      return calls[k] > 0 ? ms[k] / (double)calls[k] * (double)n * 1e-3 : 0.0;   // nothing bracketed since the reset: all-time mean
    };
    l->st.seconds_map = scaled(0, l->st.updates);
    l->st.seconds_keyframe = scaled(1, l->st.keyframes) + scaled(2, l->st.keyframes);
    // one stream: tracking = the rest of the wall time; pipelined: the mapping stream runs beside the tracking stream, which is the
    // critical path, so tracking is the whole wall time
    l->st.seconds_track = l->loop->pipelined() ? l->wall : l->wall - l->st.seconds_map - l->st.seconds_keyframe;
    *out = l->st;
    return LSDHIP_OK;
  } catch (const Error& e) { g_err = e.what(); return e.status; }
}
extern "C" int lsdloop_reset_stats(lsdloop* l) {
  if (!l) return LSDHIP_E_ARG;
  try {
    l->loop->map.gpuTimes(l->gpu0, l->calls0);
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
extern "C" int lsdloop_set_pipeline(lsdloop* l, int on) {
  if (!l) return LSDHIP_E_ARG;
  try { l->loop->setPipelined(on != 0); return LSDHIP_OK; }
  catch (const Error& e) { g_err = e.what(); return e.status; }
}
extern "C" int lsdloop_set_live_queue(lsdloop* l, int frames) {
  if (!l || frames < 1) return LSDHIP_E_ARG;
  l->loop->liveQueueLength = frames;
  return LSDHIP_OK;
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
extern "C" int lsdloop_observe_work(lsdloop* l, double out3[3]) {
  if (!l) return LSDHIP_E_ARG;
  return lsdhip_depth_observe_work(l->loop->map.handle(), out3);
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
// IPC transport of the gather: rank r copies its ring slots into slot r of the root's mailbox, then publishes its count and raises
// `ready`; the root waits for every rank's `ready`.  `consumed` (raised by the root when the NEXT gather starts) gates the overwrite.
static int gather_ipc(lsdloop* l, int count) {
  const size_t plane2 = (size_t)2 * l->ctx->width() * l->ctx->height();
  const size_t slot_floats = (size_t)l->ring_slots * plane2;
  const long long g = l->gathers;
  try {
    lsdhip_ctx* c = l->ctx->handle();
    if (l->comm_rank == l->ipc_root) {
      for (int r = 0; r < l->comm_world; r++) if (r != l->ipc_root) check(lsdhip_ctx_flag_set(c, l->ipc_flag(r, 1), (int)g), "lsdhip_ctx_flag_set");   // gather g - 1 may be overwritten
      if (count > 0) check(lsdhip_ctx_copy_dev(c, l->ipc_data() + (size_t)l->ipc_root * slot_floats, l->ring, (size_t)count * plane2 * sizeof(float)), "lsdhip_ctx_copy_dev");
      check(lsdhip_ctx_flag_set(c, l->ipc_flag(l->ipc_root, 2), count), "lsdhip_ctx_flag_set");
      for (int r = 0; r < l->comm_world; r++) if (r != l->ipc_root) check(lsdhip_ctx_flag_wait(c, l->ipc_flag(r, 0), (int)(g + 1), (int*)l->mailbox), "lsdhip_ctx_flag_wait");
    } else {
      if (g >= 1) check(lsdhip_ctx_flag_wait(c, l->ipc_flag(l->comm_rank, 1), (int)g, l->ipc_fail), "lsdhip_ctx_flag_wait");
      if (count > 0) check(lsdhip_ctx_copy_dev(c, l->ipc_data() + (size_t)l->comm_rank * slot_floats, l->ring, (size_t)count * plane2 * sizeof(float)), "lsdhip_ctx_copy_dev");
      check(lsdhip_ctx_flag_set(c, l->ipc_flag(l->comm_rank, 2), count), "lsdhip_ctx_flag_set");
      check(lsdhip_ctx_flag_set(c, l->ipc_flag(l->comm_rank, 0), (int)(g + 1)), "lsdhip_ctx_flag_set");
    }
    l->gathers++;
    return LSDHIP_OK;
  } catch (const Error& e) { g_err = e.what(); return e.status; }
}
// IPC transport, step 1 (after lsdloop_set_keyframe_ring): the root allocates and exports the mailbox (handle64_out), the others get zeros
extern "C" int lsdloop_ipc_init(lsdloop* l, int rank, int world, int root, unsigned char handle64_out[64]) {
  if (!l || !handle64_out || world < 2 || rank < 0 || rank >= world || root < 0 || root >= world || !l->ring || world > 250) return LSDHIP_E_ARG;
  l->comm_rank = rank; l->comm_world = world; l->ipc_root = root;
  std::memset(handle64_out, 0, 64);
  try {
    lsdhip_ctx* c = l->ctx->handle();
    if (rank == root) {
      const size_t bytes = 4096 + (size_t)world * l->ring_slots * 2 * l->ctx->width() * l->ctx->height() * sizeof(float);
      check(lsdhip_ctx_alloc_dev(c, bytes, (void**)&l->mailbox), "lsdhip_ctx_alloc_dev");
      check(lsdhip_ctx_memset_dev(c, l->mailbox, 0, 4096), "lsdhip_ctx_memset_dev");
      check(lsdhip_ctx_synchronize(c), "lsdhip_ctx_synchronize");
      check(lsdhip_ctx_ipc_export(c, l->mailbox, handle64_out), "lsdhip_ctx_ipc_export");
    } else {
      check(lsdhip_ctx_alloc_dev(c, 256, (void**)&l->ipc_fail), "lsdhip_ctx_alloc_dev");
      check(lsdhip_ctx_memset_dev(c, l->ipc_fail, 0, 256), "lsdhip_ctx_memset_dev");
    }
  } catch (const Error& e) { g_err = e.what(); return e.status; }
  return LSDHIP_OK;
}
// step 2: the root's handle (ignored on the root)
extern "C" int lsdloop_ipc_connect(lsdloop* l, const unsigned char root_handle64[64]) {
  if (!l || !root_handle64) return LSDHIP_E_ARG;
  try {
    if (l->comm_rank != l->ipc_root) {
      void* p = nullptr;
      check(lsdhip_ctx_ipc_open(l->ctx->handle(), root_handle64, &p), "lsdhip_ctx_ipc_open");
      l->mailbox = (char*)p;
    }
  } catch (const Error& e) { g_err = e.what(); return e.status; }
  l->ipc = true;
  l->gathers = 0;
  return LSDHIP_OK;
}
// root, after lsdloop_gather_keyframes: synchronises, writes every rank's count of the last gather to counts_out[world], returns the
// device pointer of the gathered planes ([world][ring_slots][2][h][w] floats) through data_out, and the number of failed waits
extern "C" int lsdloop_ipc_result(lsdloop* l, int* counts_out, float** data_out) {
  if (!l || !l->ipc || !l->mailbox) return LSDHIP_E_ARG;
  try {
    lsdhip_ctx* c = l->ctx->handle();
    int fail = 0;
    if (l->comm_rank == l->ipc_root) {
      std::vector<int> block(64 / 4 + (size_t)l->comm_world * 4);
      check(lsdhip_ctx_read_dev(c, block.data(), l->mailbox, block.size() * sizeof(int)), "lsdhip_ctx_read_dev");
      fail = block[0];
      if (counts_out) for (int r = 0; r < l->comm_world; r++) counts_out[r] = block[16 + 4 * r + 2];
      if (data_out) *data_out = l->ipc_data();
    } else {
      check(lsdhip_ctx_read_dev(c, &fail, l->ipc_fail, sizeof(int)), "lsdhip_ctx_read_dev");
    }
    return fail;
  } catch (const Error& e) { g_err = e.what(); return e.status; }
}
extern "C" int lsdloop_gather_keyframes(lsdloop* l, int count, int root, float* recv_dev, long long stride_floats) {
  if (!l || count < 0 || !l->ring) return LSDHIP_E_ARG;
  if (count > l->ring_slots) { g_err = "gather: more keyframes than ring slots"; return LSDHIP_E_ARG; }
  if (l->ipc) return gather_ipc(l, count);
  if (count == 0 && (l->comm_world == 1 || !l->comm)) return LSDHIP_OK;
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
    // The ranks run independent sequences, so their keyframe counts need not agree (distance-based keyframe selection): every
    // rank's count travels first (one int per rank, ncclAllGather; the host reads them — one stream synchronisation per gather,
    // i.e. per batch of frames) and sizes the root's receives.  A rank with nothing to send skips its send.
    if (!l->counts_dev) check(lsdhip_ctx_alloc_dev(l->ctx->handle(), (size_t)(l->comm_world + 1) * sizeof(int), (void**)&l->counts_dev), "lsdhip_ctx_alloc_dev");
    check(lsdhip_ctx_flag_set(l->ctx->handle(), l->counts_dev + l->comm_world, count), "lsdhip_ctx_flag_set");
    nc(R.AllGather(l->counts_dev + l->comm_world, l->counts_dev, 1, 2 /* ncclInt32 */, l->comm, stream), "ncclAllGather");
    std::vector<int> counts((size_t)l->comm_world, 0);
    check(lsdhip_ctx_read_dev(l->ctx->handle(), counts.data(), l->counts_dev, counts.size() * sizeof(int)), "lsdhip_ctx_read_dev");
    l->last_counts = counts;
    if (l->comm_rank == root) {
      if (!recv_dev) return LSDHIP_E_ARG;
      check(lsdhip_ctx_copy_dev(l->ctx->handle(), recv_dev + (size_t)root * stride_floats, l->ring, n * sizeof(float)), "lsdhip_ctx_copy_dev");
      nc(R.GroupStart(), "ncclGroupStart");
      for (int r = 0; r < l->comm_world; r++) {
        if (r == root || counts[r] <= 0) continue;
        if (counts[r] > l->ring_slots) throw Error(LSDHIP_E_ARG, "gather: a rank reports more keyframes than ring slots");
        nc(R.Recv(recv_dev + (size_t)r * stride_floats, (size_t)counts[r] * 2 * (size_t)l->ctx->width() * l->ctx->height(), 7 /* ncclFloat */, r, l->comm, stream), "ncclRecv");
      }
      nc(R.GroupEnd(), "ncclGroupEnd");
    } else if (count > 0) {
      nc(R.GroupStart(), "ncclGroupStart");
      nc(R.Send(l->ring, n, 7 /* ncclFloat */, root, l->comm, stream), "ncclSend");
      nc(R.GroupEnd(), "ncclGroupEnd");
    }
    return LSDHIP_OK;
  } catch (const Error& e) { g_err = e.what(); return e.status; }
}
// every rank's keyframe count of the last gather through RCCL (what sized the root's receives); returns the number written
extern "C" int lsdloop_gather_counts(lsdloop* l, int* counts_out, int cap) {
  if (!l || !counts_out) return LSDHIP_E_ARG;
  int n = 0;
  for (; n < (int)l->last_counts.size() && n < cap; n++) counts_out[n] = l->last_counts[n];
  return n;
}
extern "C" long long lsdloop_keyframes_exported(lsdloop* l) { return l ? l->ring_count : -1; }
extern "C" void* lsdloop_ctx(lsdloop* l) { return l ? (void*)l->ctx->handle() : nullptr; }

// =====================================================================================================================
// Row-band decomposition of the depth-map regulariser (BASELINE.json configs[4]: 3840x2160 maps tiled
// across 8 GPUs).  The index arithmetic is lsd_slam_amd/bands.py's BandPlan (the CPU tests check both against each other);
// this is the native loop: every window's pass and every halo refresh is queued on the context's stream, nothing
// synchronises the host between passes.
//   * windows held by this process refresh each other's halo rows with ONE map -> map copy launch per pass;
//   * rows owned by another process travel packed (29 B per pixel): one pack launch, ncclGroupStart / ncclSend / ncclRecv /
//     ncclGroupEnd on the same stream, one unpack launch;
//   * with other processes to talk to, a pass is issued in two parts: first the tile rows that produce rows another band needs or
//     that read rows another band sends ("edge": a few tile rows at the top and bottom of the owned range), then the rest
//     ("interior").  The exchange (pack, send / receive or mailbox + flags, unpack) runs on the context's transport stream behind
//     the edge part and beside the interior part; the next pass's edge part waits for it.  Interior tile rows neither read received
//     rows nor produce sent ones, and the parts of a pass are order-independent (lsdhip_depth_stage_rows), so the result is the
//     full frame's, bit for bit, either way.  Tile rows without an owned row are not computed at all (their rows arrive).
// One pass = regularizeDepthMapFillHoles + regularizeDepthMap(false, VAL_SUM_MIN_FOR_KEEP) (C/DepthEstimation/DepthMap.cpp:
// 656-720, :758-880), the fused launch behind lsdhip_depth_stage(dm, 5).
namespace {
const int BAND_HALO_TOP = 5, BAND_HALO_BOTTOM = 4, BAND_PACK_BYTES = 29;
struct BandSeg { int peer, row0, nrows; };   // global rows [row0, row0 + nrows) exchanged with band `peer`
}
struct lsdband {
  int device = 0, w = 0, H = 0, world = 0, first = 0, nlocal = 0, window_rows = 0;
  std::vector<std::pair<int, int>> owned, window;   // per band: [y0, y1) and [a, b)
  lsdhip_ctx* ctx = nullptr;
  std::vector<lsdhip_frame*> kf;
  std::vector<lsdhip_depthmap*> dm;
  // communicator over PROCESSES (one per GPU); band r lives in process proc_of[r]
  NcclComm comm = nullptr;
  int nprocs = 1, proc = 0;
  std::vector<int> proc_of;
  std::vector<char*> sendbuf, recvbuf;   // one packed device buffer per remote segment
  long long passes_run = 0;
  bool packedLocal = false;              // test hook: windows of this process exchange through pack -> copy -> unpack as well
  bool overlap = true;                   // exchange with other processes under the interior rows of the pass (lsdband_set_overlap)
  // Second transport between PROCESSES of one node (lsdband_ipc_*): every process exports one "mailbox" allocation — flags and,
  // twice (two exchange parities), the packed receive buffer of each of its incoming remote segments — and maps the others'.
  // A sender packs its rows STRAIGHT INTO the receiver's buffer and raises the segment's `ready` flag; the receiver unpacks
  // and raises `consumed`, which gates the next reuse of that parity.  Same pack -> transfer -> unpack schedule as the RCCL path.
  bool ipc = false;
  char* mailbox = nullptr;
  std::vector<char*> peerMailbox;        // by process; own entry = mailbox
  long long exchanges = 0;               // exchange ordinal since lsdband_ipc_connect
  struct IpcSeg { int r, peer, row0, nrows; size_t bytes, off[2]; int flag; };   // flag: index of (ready, consumed) in the flag block
  // incoming remote segments of process q in canonical order (both ends compute the same layout)
  std::vector<IpcSeg> ipc_layout(int q, size_t* total = nullptr) const {
    std::vector<IpcSeg> out;
    size_t off = 4096;                   // flag block: fail word + (ready, consumed) per segment
    int k = 0;
    for (int r = 0; r < world; r++) {
      if (proc_of[r] != q) continue;
      for (const BandSeg& sg : recv_list(r)) {
        if (proc_of[sg.peer] == q) continue;
        IpcSeg e{r, sg.peer, sg.row0, sg.nrows, (size_t)sg.nrows * w * BAND_PACK_BYTES, {0, 0}, k++};
        e.bytes = (e.bytes + 255) & ~(size_t)255;
        e.off[0] = off; off += e.bytes;
        e.off[1] = off; off += e.bytes;
        out.push_back(e);
      }
    }
    if (total) *total = off;
    return out;
  }
  std::vector<BandSeg> recv_list(int r) const {
    std::vector<BandSeg> out;
    const int a = window[r].first, b = window[r].second;
    for (int s = 0; s < world; s++) {
      if (s == r) continue;
      const int lo = std::max(a, owned[s].first), hi = std::min(b, owned[s].second);
      if (hi > lo) out.push_back({s, lo, hi - lo});
    }
    return out;
  }
  bool local(int r) const { return r >= first && r < first + nlocal; }
};

// owned rows and windows of every band (lsd_slam_amd/bands.py BandPlan.__init__): pure index arithmetic, no device
static bool band_fill_layout(lsdband* b) {
  const int H = b->H, world = b->world;
  const int base = H / world, extra = H % world;
  int y = 0, need = 0;
  b->owned.clear(); b->window.clear();
  for (int r = 0; r < world; r++) {
    const int n = base + (r < extra ? 1 : 0);
    if (n < 1) return false;
    b->owned.push_back({y, y + n});
    y += n;
    need = std::max(need, n + BAND_HALO_TOP + BAND_HALO_BOTTOM);
  }
  b->window_rows = std::min(H, (need + 15) / 16 * 16);
  for (int r = 0; r < world; r++) {
    const int a = std::min(std::max(b->owned[r].first - BAND_HALO_TOP, 0), H - b->window_rows);
    b->window.push_back({a, a + b->window_rows});
  }
  b->proc_of.assign(world, 0);
  return true;
}
// Tile rows (8 map rows each) of band r's window that hold at least one owned row, as runs of equal kind.  `edge`: the tile row
// produces a row some other band's window holds, or its tiles read (the fused pass's tile halo: 4 rows above its first row, 4 below
// its last) a row this band does not own — such tile rows wait for the previous exchange and are issued before the next one;
// the others ("interior") neither read received rows nor produce sent ones.
struct BandRun { int t0, n; bool edge; };
static std::vector<BandRun> band_tile_runs(const lsdband& b, int r) {
  std::vector<BandRun> runs;
  const int wf = b.window[r].first, WR = b.window_rows;
  const int o0 = b.owned[r].first - wf, o1 = b.owned[r].second - wf;
  std::vector<char> sent(WR, 0);
  for (int q = 0; q < b.world; q++) {
    if (q == r) continue;
    const int lo = std::max(b.window[q].first, b.owned[r].first), hi = std::min(b.window[q].second, b.owned[r].second);
    for (int y = lo; y < hi; y++) sent[y - wf] = 1;
  }
  for (int t = o0 / 8; t < (o1 + 7) / 8; t++) {
    bool edge = false;
    for (int y = 8 * t; y < 8 * t + 8 && y < WR; y++) edge = edge || sent[y];
    for (int y = 8 * t - 4; y < 8 * t + 12; y++) edge = edge || (y >= 0 && y < WR && (y < o0 || y >= o1));
    if (!runs.empty() && runs.back().edge == edge) runs.back().n++;
    else runs.push_back({t, 1, edge});
  }
  return runs;
}
// the runs of band `band` of an H-row map cut into `world` bands: (first tile row, tile rows, edge) triples; returns their number
extern "C" int lsdband_tile_runs(int H, int world, int band, int* runs3, int cap) {
  if (H <= 0 || world < 1 || (H % 16) != 0 || band < 0 || band >= world) return LSDHIP_E_ARG;
  lsdband b;
  b.H = H; b.world = world;
  if (!band_fill_layout(&b)) return LSDHIP_E_ARG;
  const std::vector<BandRun> runs = band_tile_runs(b, band);
  for (int k = 0; k < (int)runs.size() && k < cap && runs3; k++) { runs3[3 * k] = runs[k].t0; runs3[3 * k + 1] = runs[k].n; runs3[3 * k + 2] = runs[k].edge ? 1 : 0; }
  return (int)runs.size();
}
// The plan alone, no GPU needed (CPU tests compare it with BandPlan): layout4[4 r .. 4 r + 3] = owned [y0, y1), window [a, b) of
// band r; segments: up to cap triples (receiving band, sending band, first global row, rows) -> 4 ints each; returns their number.
extern "C" int lsdband_plan(int H, int world, int* window_rows_out, int* layout4, int* segments4, int cap) {
  if (H <= 0 || world < 1 || (H % 16) != 0) return LSDHIP_E_ARG;
  lsdband b;
  b.H = H; b.world = world;
  if (!band_fill_layout(&b)) return LSDHIP_E_ARG;
  if (window_rows_out) *window_rows_out = b.window_rows;
  if (layout4)
    for (int r = 0; r < world; r++) {
      layout4[4 * r] = b.owned[r].first; layout4[4 * r + 1] = b.owned[r].second;
      layout4[4 * r + 2] = b.window[r].first; layout4[4 * r + 3] = b.window[r].second;
    }
  int n = 0;
  for (int r = 0; r < world; r++)
    for (const BandSeg& s : b.recv_list(r)) {
      if (segments4 && n < cap) { segments4[4 * n] = r; segments4[4 * n + 1] = s.peer; segments4[4 * n + 2] = s.row0; segments4[4 * n + 3] = s.nrows; }
      n++;
    }
  return n;
}
extern "C" int lsdband_create(int device, int w, int H, int world, int first_band, int n_local, lsdband** out) {
  if (!out || w <= 0 || H <= 0 || world < 1 || first_band < 0 || n_local < 1 || first_band + n_local > world || (H % 16) != 0) return LSDHIP_E_ARG;
  lsdband* b = new lsdband();
  b->device = device; b->w = w; b->H = H; b->world = world; b->first = first_band; b->nlocal = n_local;
  if (!band_fill_layout(b)) { delete b; g_err = "lsdband_create: more bands than rows"; return LSDHIP_E_ARG; }
  const float K4[4] = {0.5f * w, 0.5f * w, 0.5f * w, 0.5f * b->window_rows};   // unused by the regulariser
  int rc = lsdhip_ctx_create(device, w, b->window_rows, K4, nullptr, &b->ctx);
  if (rc == LSDHIP_OK) rc = lsdhip_ctx_set_async(b->ctx, 1);
  std::vector<uint8_t> zeros((size_t)w * b->window_rows, 0);
  for (int i = 0; i < n_local && rc == LSDHIP_OK; i++) {
    lsdhip_frame* f = nullptr;
    lsdhip_depthmap* d = nullptr;
    rc = lsdhip_frame_create(b->ctx, first_band + i, zeros.data(), &f);
    if (rc == LSDHIP_OK) { b->kf.push_back(f); rc = lsdhip_depth_create(b->ctx, &d); }
    if (rc == LSDHIP_OK) b->dm.push_back(d);
  }
  if (rc != LSDHIP_OK) { g_err = lsdhip_last_error(); lsdband_destroy(b); return rc; }
  *out = b;
  return LSDHIP_OK;
}
extern "C" void lsdband_destroy(lsdband* b) {
  if (!b) return;
  if (b->ctx) (void)lsdhip_ctx_synchronize(b->ctx);
  if (b->comm && rccl().ok) (void)rccl().CommDestroy(b->comm);
  for (int q = 0; q < (int)b->peerMailbox.size(); q++)
    if (b->peerMailbox[q] && b->peerMailbox[q] != b->mailbox) (void)lsdhip_ctx_ipc_close(b->ctx, b->peerMailbox[q]);
  if (b->mailbox) (void)lsdhip_ctx_free_dev(b->ctx, b->mailbox);
  for (char* p : b->sendbuf) if (p) (void)lsdhip_ctx_free_dev(b->ctx, p);
  for (char* p : b->recvbuf) if (p) (void)lsdhip_ctx_free_dev(b->ctx, p);
  for (auto* d : b->dm) lsdhip_depth_destroy(d);
  for (auto* f : b->kf) lsdhip_frame_destroy(f);
  if (b->ctx) lsdhip_ctx_destroy(b->ctx);
  delete b;
}
extern "C" int lsdband_window_rows(const lsdband* b) { return b ? b->window_rows : -1; }
extern "C" int lsdband_layout(const lsdband* b, int band, int out4[4]) {
  if (!b || !out4 || band < 0 || band >= b->world) return LSDHIP_E_ARG;
  out4[0] = b->owned[band].first; out4[1] = b->owned[band].second; out4[2] = b->window[band].first; out4[3] = b->window[band].second;
  return LSDHIP_OK;
}
// rows [a, b) of the full map for local window `local`: hypotheses in the reference's 32-byte AoS layout + maxGradients
extern "C" int lsdband_load(lsdband* b, int local, const void* hyp_window, const float* maxgrad_window) {
  if (!b || local < 0 || local >= b->nlocal || !hyp_window || !maxgrad_window) return LSDHIP_E_ARG;
  b->passes_run = 0;   // fresh data in the window, halo rows included: no refresh is owed before the next call's first pass
  int rc = lsdhip_frame_set_maxgrad(b->kf[local], maxgrad_window);
  if (rc == LSDHIP_OK) rc = lsdhip_depth_upload(b->dm[local], b->kf[local], (const lsdhip_hypothesis*)hyp_window, 0);
  if (rc != LSDHIP_OK) g_err = lsdhip_last_error();
  return rc;
}
static int band_check_fail(lsdband* b);
extern "C" int lsdband_get(lsdband* b, int local, void* hyp_window_out) {
  if (!b || local < 0 || local >= b->nlocal || !hyp_window_out) return LSDHIP_E_ARG;
  int rc = lsdhip_ctx_synchronize(b->ctx);
  if (rc == LSDHIP_OK && (rc = band_check_fail(b)) != LSDHIP_OK) return rc;
  if (rc == LSDHIP_OK) rc = lsdhip_depth_download(b->dm[local], (lsdhip_hypothesis*)hyp_window_out);
  if (rc != LSDHIP_OK) g_err = lsdhip_last_error();
  return rc;
}
// Waits for everything queued.  On the IPC transport a flag wait that gave up (a wedged or dead peer: k_flag_wait's bounded spin)
// leaves the value it waited for in the mailbox's fail word and the stream carries on with stale rows: the maps are then INVALID and
// this returns LSDHIP_E_STATE (lsdband_last error text names the flag value); lsdband_get refuses likewise.
static int band_check_fail(lsdband* b) {
  if (!b->ipc || !b->mailbox) return LSDHIP_OK;
  int v = 0;
  if (lsdhip_ctx_read_dev(b->ctx, &v, b->mailbox, sizeof(int)) != LSDHIP_OK) { g_err = lsdhip_last_error(); return LSDHIP_E_HIP; }
  if (v != 0) { g_err = "lsdband: a halo flag wait timed out at value " + std::to_string(v) + " (peer process wedged or gone): results are invalid"; return LSDHIP_E_STATE; }
  return LSDHIP_OK;
}
extern "C" int lsdband_synchronize(lsdband* b) {
  if (!b) return LSDHIP_E_ARG;
  const int rc = lsdhip_ctx_synchronize(b->ctx);
  if (rc != LSDHIP_OK) { g_err = lsdhip_last_error(); return rc; }
  return band_check_fail(b);
}
// one process per GPU: process p of nprocs holds the bands with proc_of_band[r] == p (must match its lsdband_create range)
extern "C" int lsdband_comm_init(lsdband* b, const void* unique_id128, int nprocs, int proc, const int* proc_of_band) {
  if (!b || !unique_id128 || nprocs < 1 || proc < 0 || proc >= nprocs || !proc_of_band) return LSDHIP_E_ARG;
  for (int r = 0; r < b->world; r++) {
    b->proc_of[r] = proc_of_band[r];
    if ((proc_of_band[r] == proc) != b->local(r)) { g_err = "lsdband_comm_init: band ownership does not match lsdband_create"; return LSDHIP_E_ARG; }
  }
  b->nprocs = nprocs; b->proc = proc;
  if (nprocs == 1) return LSDHIP_OK;
  Rccl& R = rccl();
  if (!R.ok) { g_err = "RCCL not available: " + R.why; return LSDHIP_E_STATE; }
  NcclId id;
  std::memcpy(&id, unique_id128, sizeof(id));
  int rc = R.CommInitRank(&b->comm, nprocs, id, proc);
  if (rc != 0) { g_err = std::string("ncclCommInitRank: ") + (R.GetErrorString ? R.GetErrorString(rc) : "?"); return LSDHIP_E_HIP; }
  return LSDHIP_OK;
}
// IPC transport, step 1: band ownership as in lsdband_comm_init; allocates this process's mailbox and exports it (64 bytes)
extern "C" int lsdband_ipc_init(lsdband* b, int nprocs, int proc, const int* proc_of_band, unsigned char handle64_out[64]) {
  if (!b || nprocs < 2 || proc < 0 || proc >= nprocs || !proc_of_band || !handle64_out) return LSDHIP_E_ARG;
  for (int r = 0; r < b->world; r++) {
    b->proc_of[r] = proc_of_band[r];
    if ((proc_of_band[r] == proc) != b->local(r)) { g_err = "lsdband_ipc_init: band ownership does not match lsdband_create"; return LSDHIP_E_ARG; }
  }
  b->nprocs = nprocs; b->proc = proc;
  size_t total = 0;
  (void)b->ipc_layout(proc, &total);
  try {
    check(lsdhip_ctx_alloc_dev(b->ctx, total, (void**)&b->mailbox), "lsdhip_ctx_alloc_dev");
    check(lsdhip_ctx_memset_dev(b->ctx, b->mailbox, 0, total), "lsdhip_ctx_memset_dev");
    check(lsdhip_ctx_synchronize(b->ctx), "lsdhip_ctx_synchronize");
    check(lsdhip_ctx_ipc_export(b->ctx, b->mailbox, handle64_out), "lsdhip_ctx_ipc_export");
  } catch (const Error& e) { g_err = e.what(); return e.status; }
  return LSDHIP_OK;
}
// step 2: the mailboxes of all processes (nprocs x 64 bytes, in process order; the own entry is ignored)
extern "C" int lsdband_ipc_connect(lsdband* b, const unsigned char* handles) {
  if (!b || !handles || !b->mailbox) return LSDHIP_E_ARG;
  try {
    b->peerMailbox.assign(b->nprocs, nullptr);
    for (int q = 0; q < b->nprocs; q++) {
      if (q == b->proc) { b->peerMailbox[q] = b->mailbox; continue; }
      void* p = nullptr;
      check(lsdhip_ctx_ipc_open(b->ctx, handles + 64 * (size_t)q, &p), "lsdhip_ctx_ipc_open");
      b->peerMailbox[q] = (char*)p;
    }
  } catch (const Error& e) { g_err = e.what(); return e.status; }
  b->ipc = true;
  b->exchanges = 0;
  return LSDHIP_OK;
}
// after lsdband_synchronize: 0 if every flag wait of this process was satisfied, else the flag value a wait gave up on
extern "C" int lsdband_ipc_failed(lsdband* b) {
  if (!b || !b->mailbox) return LSDHIP_E_ARG;
  int v = 0;
  if (lsdhip_ctx_read_dev(b->ctx, &v, b->mailbox, sizeof(int)) != LSDHIP_OK) { g_err = lsdhip_last_error(); return LSDHIP_E_HIP; }
  return v;
}
extern "C" int lsdband_set_packed_exchange(lsdband* b, int on) {
  if (!b) return LSDHIP_E_ARG;
  b->packedLocal = on != 0;
  return LSDHIP_OK;
}
extern "C" long long lsdband_halo_bytes_per_pass(const lsdband* b) {
  if (!b) return -1;
  long long rows = 0;
  for (int r = 0; r < b->world; r++) for (const BandSeg& s : b->recv_list(r)) rows += s.nrows;
  return rows * b->w * BAND_PACK_BYTES;
}
// `passes` passes over every local window, halo refresh between passes.  Returns after everything is QUEUED.
extern "C" int lsdband_run(lsdband* b, int passes) {
  if (!b || passes < 0) return LSDHIP_E_ARG;
  try {
    Rccl& R = rccl();
    auto nc = [&](int rc, const char* what) { if (rc != 0) throw Error(LSDHIP_E_HIP, std::string(what) + ": " + (R.GetErrorString ? R.GetErrorString(rc) : "?")); };
    void* stream = lsdhip_ctx_stream(b->ctx);
    // exchange plan of this process (fixed for the object's lifetime): local copies, remote sends, remote receives
    std::vector<lsdhip_row_copy> localCopies, packs, unpacks;
    struct Wire { int peerProc; size_t bytes; char* buf; char* buf2; };
    std::vector<Wire> sends, recvs, loop;
    size_t si = 0, ri = 0;
    for (int r = 0; r < b->world; r++) {
      for (const BandSeg& s : b->recv_list(r)) {          // band r receives rows owned by band s.peer
        const bool dstLocal = b->local(r), srcLocal = b->local(s.peer);
        const size_t bytes = (size_t)s.nrows * b->w * BAND_PACK_BYTES;
        if (dstLocal && srcLocal && b->packedLocal) {
          // the wire path without the wire: pack into a send buffer, device copy in place of ncclSend / ncclRecv, unpack
          if (b->sendbuf.size() <= si) b->sendbuf.push_back(nullptr);
          if (!b->sendbuf[si]) check(lsdhip_ctx_alloc_dev(b->ctx, bytes, (void**)&b->sendbuf[si]), "lsdhip_ctx_alloc_dev");
          if (b->recvbuf.size() <= ri) b->recvbuf.push_back(nullptr);
          if (!b->recvbuf[ri]) check(lsdhip_ctx_alloc_dev(b->ctx, bytes, (void**)&b->recvbuf[ri]), "lsdhip_ctx_alloc_dev");
          lsdhip_row_copy c{};
          c.src_map = b->dm[s.peer - b->first]; c.src_row0 = s.row0 - b->window[s.peer].first;
          c.dst_packed = b->sendbuf[si]; c.nrows = s.nrows;
          packs.push_back(c);
          lsdhip_row_copy u{};
          u.src_packed = b->recvbuf[ri];
          u.dst_map = b->dm[r - b->first]; u.dst_row0 = s.row0 - b->window[r].first; u.nrows = s.nrows;
          unpacks.push_back(u);
          loop.push_back({0, bytes, b->sendbuf[si], b->recvbuf[ri]});
          si++; ri++;
        } else if (dstLocal && srcLocal) {
          lsdhip_row_copy c{};
          c.src_map = b->dm[s.peer - b->first]; c.src_row0 = s.row0 - b->window[s.peer].first;
          c.dst_map = b->dm[r - b->first]; c.dst_row0 = s.row0 - b->window[r].first;
          c.nrows = s.nrows;
          localCopies.push_back(c);
        } else if (b->ipc) {
          // remote segment on the IPC transport: planned below from the mailbox layouts
        } else if (srcLocal) {                             // we own the rows: pack + send to r's process
          if (b->sendbuf.size() <= si) b->sendbuf.push_back(nullptr);
          if (!b->sendbuf[si]) check(lsdhip_ctx_alloc_dev(b->ctx, bytes, (void**)&b->sendbuf[si]), "lsdhip_ctx_alloc_dev");
          lsdhip_row_copy c{};
          c.src_map = b->dm[s.peer - b->first]; c.src_row0 = s.row0 - b->window[s.peer].first;
          c.dst_packed = b->sendbuf[si]; c.nrows = s.nrows;
          packs.push_back(c);
          sends.push_back({b->proc_of[r], bytes, b->sendbuf[si], nullptr});
          si++;
        } else if (dstLocal) {                             // we need the rows: receive from the owner's process + unpack
          if (b->recvbuf.size() <= ri) b->recvbuf.push_back(nullptr);
          if (!b->recvbuf[ri]) check(lsdhip_ctx_alloc_dev(b->ctx, bytes, (void**)&b->recvbuf[ri]), "lsdhip_ctx_alloc_dev");
          lsdhip_row_copy c{};
          c.src_packed = b->recvbuf[ri];
          c.dst_map = b->dm[r - b->first]; c.dst_row0 = s.row0 - b->window[r].first; c.nrows = s.nrows;
          unpacks.push_back(c);
          recvs.push_back({b->proc_of[s.peer], bytes, b->recvbuf[ri], nullptr});
          ri++;
        }
      }
    }
    if ((!sends.empty() || !recvs.empty()) && !b->comm && !b->ipc)
      throw Error(LSDHIP_E_STATE, "lsdband_run: bands of other processes but no communicator (lsdband_comm_init / lsdband_ipc_connect)");
    // IPC transport: the remote segments of this process, outgoing (rows we own, needed by a band of process q: a slot of q's
    // mailbox) and incoming (our own mailbox)
    struct IpcOut { int q; lsdband::IpcSeg seg; };
    std::vector<IpcOut> ipcOut;
    std::vector<lsdband::IpcSeg> ipcIn;
    if (b->ipc) {
      ipcIn = b->ipc_layout(b->proc);
      for (int q = 0; q < b->nprocs; q++) {
        if (q == b->proc) continue;
        for (const lsdband::IpcSeg& e : b->ipc_layout(q))
          if (b->local(e.peer)) ipcOut.push_back({q, e});
      }
    }
    auto flagPtr = [](char* box, int k, int which) { return (int*)(box + 64 + (size_t)k * 8 + (size_t)which * 4); };   // (ready, consumed) pairs behind the fail word
    std::vector<std::vector<BandRun>> runs(b->nlocal);
    for (int i = 0; i < b->nlocal; i++) runs[i] = band_tile_runs(*b, b->first + i);
    const bool remote = b->ipc || !sends.empty() || !recvs.empty();
    const bool overlap = b->overlap && remote;
    // one halo exchange: every window's non-owned rows are refreshed from their owners.  onAux: between lsdhip_ctx_aux_begin / _end
    // (the transport primitives go to the transport stream, and so do the RCCL calls); otherwise everything on the main stream
    auto exchange = [&](bool onAux) {
      void* xstream = onAux ? lsdhip_ctx_aux_stream(b->ctx) : stream;
      // the halo rows a window receives are never rows it owns, and sources are always owned rows: the copies of one
      // exchange cannot overwrite each other's inputs, so one launch serves all of them
      if (!packs.empty()) check(lsdhip_depth_copy_rows_batch(b->ctx, (int)packs.size(), packs.data()), "lsdhip_depth_copy_rows_batch");
      if (!localCopies.empty()) check(lsdhip_depth_copy_rows_batch(b->ctx, (int)localCopies.size(), localCopies.data()), "lsdhip_depth_copy_rows_batch");
      for (const Wire& l : loop) check(lsdhip_ctx_copy_dev(b->ctx, l.buf2, l.buf, l.bytes), "lsdhip_ctx_copy_dev");
      if (!loop.empty() && sends.empty() && recvs.empty())
        check(lsdhip_depth_copy_rows_batch(b->ctx, (int)unpacks.size(), unpacks.data()), "lsdhip_depth_copy_rows_batch");
      if (b->ipc) {
        const long long x = b->exchanges;
        const int par = (int)(x & 1);
        std::vector<lsdhip_row_copy> ipack, iunpack;
        for (const IpcOut& o : ipcOut) {
          char* box = b->peerMailbox[o.q];
          if (x >= 2) check(lsdhip_ctx_flag_wait(b->ctx, flagPtr(box, o.seg.flag, 1), (int)(x - 1), (int*)b->mailbox), "lsdhip_ctx_flag_wait");
          lsdhip_row_copy c{};
          c.src_map = b->dm[o.seg.peer - b->first]; c.src_row0 = o.seg.row0 - b->window[o.seg.peer].first;
          c.dst_packed = box + o.seg.off[par]; c.nrows = o.seg.nrows;
          ipack.push_back(c);
        }
        if (!ipack.empty()) check(lsdhip_depth_copy_rows_batch(b->ctx, (int)ipack.size(), ipack.data()), "lsdhip_depth_copy_rows_batch");
        for (const IpcOut& o : ipcOut) check(lsdhip_ctx_flag_set(b->ctx, flagPtr(b->peerMailbox[o.q], o.seg.flag, 0), (int)(x + 1)), "lsdhip_ctx_flag_set");
        for (const lsdband::IpcSeg& e : ipcIn) {
          check(lsdhip_ctx_flag_wait(b->ctx, flagPtr(b->mailbox, e.flag, 0), (int)(x + 1), (int*)b->mailbox), "lsdhip_ctx_flag_wait");
          lsdhip_row_copy u{};
          u.src_packed = b->mailbox + e.off[par];
          u.dst_map = b->dm[e.r - b->first]; u.dst_row0 = e.row0 - b->window[e.r].first; u.nrows = e.nrows;
          iunpack.push_back(u);
        }
        if (!iunpack.empty()) check(lsdhip_depth_copy_rows_batch(b->ctx, (int)iunpack.size(), iunpack.data()), "lsdhip_depth_copy_rows_batch");
        for (const lsdband::IpcSeg& e : ipcIn) check(lsdhip_ctx_flag_set(b->ctx, flagPtr(b->mailbox, e.flag, 1), (int)(x + 1)), "lsdhip_ctx_flag_set");
        b->exchanges++;
      } else if (!sends.empty() || !recvs.empty()) {
        nc(R.GroupStart(), "ncclGroupStart");
        for (const Wire& s : sends) nc(R.Send(s.buf, s.bytes, 0 /* ncclInt8 */, s.peerProc, b->comm, xstream), "ncclSend");
        for (const Wire& r : recvs) nc(R.Recv(r.buf, r.bytes, 0 /* ncclInt8 */, r.peerProc, b->comm, xstream), "ncclRecv");
        nc(R.GroupEnd(), "ncclGroupEnd");
        check(lsdhip_depth_copy_rows_batch(b->ctx, (int)unpacks.size(), unpacks.data()), "lsdhip_depth_copy_rows_batch");
      }
    };
    // A call ends with its last pass, not with an exchange (the caller may only want the owned rows): when passes of an earlier call
    // have run, the halo rows are one pass old (and tile rows without an owned row are never computed locally), so the refresh that
    // belongs between that pass and this call's first one is queued here, on the main stream.
    if (passes > 0 && b->passes_run > 0) exchange(false);
    // the parts of a pass — (window, run of tile rows) pairs — share launches (blockIdx.z = part): all edge parts of this process's
    // windows in one, all interior parts in one; without an exchange to hide, every window's whole range in one
    std::vector<lsdhip_depthmap*> edgeMaps, innerMaps, allMaps;
    std::vector<int> edgeT0, edgeN, innerT0, innerN, allT0, allN;
    for (int i = 0; i < b->nlocal; i++) {
      for (const BandRun& u : runs[i]) {
        (u.edge ? edgeMaps : innerMaps).push_back(b->dm[i]);
        (u.edge ? edgeT0 : innerT0).push_back(u.t0);
        (u.edge ? edgeN : innerN).push_back(u.n);
      }
      allMaps.push_back(b->dm[i]);
      allT0.push_back(runs[i].front().t0);
      allN.push_back(runs[i].back().t0 + runs[i].back().n - runs[i].front().t0);
    }
    auto parts = [&](std::vector<lsdhip_depthmap*>& m, std::vector<int>& t0, std::vector<int>& n) {
      check(lsdhip_depth_stage_rows_batch(b->ctx, (int)m.size(), m.data(), t0.data(), n.data()), "lsdhip_depth_stage_rows_batch");
    };
    for (int p = 0; p < passes; p++) {
      if (overlap) {
        // edge parts first (they wait for the previous exchange), the exchange forks behind them, interior parts run beside it
        check(lsdhip_ctx_aux_join(b->ctx), "lsdhip_ctx_aux_join");
        parts(edgeMaps, edgeT0, edgeN);
        if (p + 1 < passes) check(lsdhip_ctx_aux_begin(b->ctx), "lsdhip_ctx_aux_begin");
        parts(innerMaps, innerT0, innerN);
      } else {
        parts(allMaps, allT0, allN);
      }
      for (int i = 0; i < b->nlocal; i++) check(lsdhip_depth_stage_rows(b->dm[i], 5, 0, 0, 1), "lsdhip_depth_stage_rows");   // validity planes swapped: the pass is queued
      b->passes_run++;
      if (p + 1 == passes) break;
      exchange(overlap);
      if (overlap) check(lsdhip_ctx_aux_end(b->ctx), "lsdhip_ctx_aux_end");
    }
    if (overlap) check(lsdhip_ctx_aux_join(b->ctx), "lsdhip_ctx_aux_join");
    return LSDHIP_OK;
  } catch (const Error& e) { (void)lsdhip_ctx_aux_end(b->ctx); g_err = e.what(); return e.status; }
}
// 1 (default): with other processes to exchange with, the exchange of a pass runs under its interior rows; 0: one launch per
// window and pass, then the exchange, all on one stream
extern "C" int lsdband_set_overlap(lsdband* b, int on) {
  if (!b) return LSDHIP_E_ARG;
  b->overlap = on != 0;
  return LSDHIP_OK;
}
