/* lsdhip.h — C ABI of the MI355X-native LSD-SLAM dense hot path (liblsdhip.so).
 *
 * The reference (tum-vision/lsd_slam) has no plugin/FFI layer: the boundary of this path is the C++ class API
 * consumed by SlamSystem.  Each entry point below names the reference interface it replaces
 * (`C/` = lsd_slam_core/src/).  Plain pointers and sizes only; no C++/torch types cross this boundary.
 * The header-only C++ adapters in include/lsd_slam_hip.hpp put the reference's class signatures
 * (lsd_slam::SE3Tracker::trackFrame, lsd_slam::DepthMap::updateKeyframe / createKeyFrame …) back on top of it.
 *
 * Conventions
 *   - poses: double[7] = (qw,qx,qy,qz,tx,ty,tz) (Sophus::SE3d content), Sim3: double[8] with scale last.
 *   - every call returns an int status: 0 ok, >0 algorithmic condition (1 = tracking diverged), <0 usage /
 *     runtime error (LSDHIP_E_*).  Nothing throws across the boundary.
 *   - a context owns one HIP stream; all calls on objects of one context are serialised on that stream and are
 *     synchronous from the caller's point of view unless documented otherwise (lsdhip_ctx_set_async).  A PIPELINED
 *     context (lsdhip_ctx_set_pipeline) owns two: SE3Tracker jobs run on the tracking stream, frame creation and
 *     DepthMap calls on the mapping stream, beside each other (C/SlamSystem.h:124-132: tracking and mapping threads).
 *     Host calls on one context are serialised by a per-context mutex; different contexts may be used concurrently
 *     from different host threads.
 *   - "host" pointers are ordinary host memory; "dev" pointers are device memory on the context's GPU.
 */
#ifndef LSDHIP_H
#define LSDHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSDHIP_PYRAMID_LEVELS 5 /* C/util/settings.h:98-106 */

#define LSDHIP_OK 0
#define LSDHIP_DIVERGED 1
#define LSDHIP_E_ARG (-1)
#define LSDHIP_E_HIP (-2)
#define LSDHIP_E_STATE (-3)
#define LSDHIP_E_CAPACITY (-4)

typedef struct lsdhip_ctx lsdhip_ctx;
typedef struct lsdhip_frame lsdhip_frame;
typedef struct lsdhip_tracker lsdhip_tracker;
typedef struct lsdhip_depthmap lsdhip_depthmap;

/* The mutable globals of C/util/settings.cpp:77-88 that the hot path reads. */
typedef struct lsdhip_params {
  float minUseGrad;              /* 5  */
  float cameraPixelNoise2;       /* 16 */
  float depthSmoothingFactor;    /* 1  */
  int allowNegativeIdepths;      /* 1  */
  int useSubpixelStereo;         /* 1  */
  int useAffineLightningEstimation; /* 1 (ROS cfg default 0, cfg/LSDParams.cfg:27) */
} lsdhip_params;
void lsdhip_default_params(lsdhip_params* p);
/* The execution defaults this header states in prose, as built into the library (no GPU needed): tests/test_abi_cpu.py compares the
 * two, so that the text cannot drift from the code. */
typedef struct lsdhip_build_defaults_t {
  int ctx_async, ctx_pipeline;                          /* lsdhip_ctx_set_async / lsdhip_ctx_set_pipeline without a call */
  int spec_trials_small, spec_small_pixels;             /* automatic speculation policy of lsdhip_tracker_set_speculation ... */
  int spec_trials_mid, spec_mid_pixels;
  int spec_workgroups, spec_workgroups_above_pixels;
  int spec_trials_max;                                  /* largest `trials` lsdhip_tracker_set_speculation accepts */
  int batch_throughput_min_jobs, batch_strip_workgroups;  /* lsdhip_tracker_track_batch: throughput mode */
  int batch_coarse_min_jobs, batch_coarse_max_pixels, batch_coarse_max_points;   /* ... its coarse levels: lsdhip_tracker_set_batch_coarse_min_jobs */
} lsdhip_build_defaults_t;
void lsdhip_build_defaults(lsdhip_build_defaults_t* out);

/* DepthMapPixelHypothesis, C/DepthEstimation/DepthMapPixelHypothesis.h:43-60 (32-byte AoS exchange format). */
typedef struct lsdhip_hypothesis {
  uint8_t isValid;
  uint8_t pad_[3];
  int32_t blacklisted;
  float nextStereoFrameMinID;
  int32_t validity_counter;
  float idepth, idepth_var, idepth_smoothed, idepth_var_smoothed;
} lsdhip_hypothesis;

/* ---- context ------------------------------------------------------------------------------------- */
/* Replaces the (w,h,K) constructor arguments shared by SE3Tracker (C/Tracking/SE3Tracker.cpp:46-94),
 * DepthMap (C/DepthEstimation/DepthMap.cpp:41-83) and Frame (C/DataStructures/Frame.cpp:397-484).
 * K = (fx, fy, cx, cy) of level 0; w and h must be multiples of 16 (C/SlamSystem.cpp:55-59). */
int lsdhip_ctx_create(int device, int w, int h, const float K[4], const lsdhip_params* params, lsdhip_ctx** out);
void lsdhip_ctx_destroy(lsdhip_ctx* ctx);
void* lsdhip_ctx_stream(lsdhip_ctx* ctx);        /* hipStream_t of the context */
/* on != 0: the DepthMap calls (update / createKeyFrame / finalizeKeyFrame) return once their kernels are enqueued; the
 * values the host only needs later (Frame::meanIdepth / numPoints, the Sim3 scale of a new keyframe, a propagation
 * overflow) are picked up by the first call that asks for them (lsdhip_frame_stats / _get_pose) or by
 * lsdhip_ctx_synchronize.  Default 0: every call is synchronous, like the reference's. */
int lsdhip_ctx_set_async(lsdhip_ctx* ctx, int on);
/* The reference's execution model — SE3Tracker::trackFrame on the tracking thread BESIDE DepthMap::updateKeyframe on the mapping thread,
 * blockUntilMapped == false (C/SlamSystem.h:124-132, C/SlamSystem.cpp:1026-1040) — on one GPU: on != 0 gives the context a second
 * stream.  Tracker calls keep running on lsdhip_ctx_stream; frame creation (upload + pyramids) and every DepthMap call run on
 * lsdhip_ctx_map_stream.  The streams are ordered by events only where data crosses: a tracker call waits for the point its frame's
 * pyramids and its keyframe's PUBLISHED depth planes were complete at.  In the other direction there is no event: a DepthMap call
 * consumes the pose and refPixelWasGood mask of tracker calls that have RETURNED (the host has seen them finish), which is how
 * SlamSystem's mapping thread uses them; call it after them.  Frame::setDepth on the mapping stream writes a second set of depth planes,
 * so a tracker call running beside it keeps reading an unchanged TrackingReference; lsdhip_frame_publish_depth hands the new planes
 * over.  Default 0: one stream, every call ordered behind the previous one (blockUntilMapped == true).  Usually combined with
 * lsdhip_ctx_set_async(ctx, 1); host-synchronous entries (downloads, uploads) drain both streams. */
int lsdhip_ctx_set_pipeline(lsdhip_ctx* ctx, int on);
int lsdhip_ctx_pipeline(lsdhip_ctx* ctx);          /* 1 / 0 */
void* lsdhip_ctx_map_stream(lsdhip_ctx* ctx);      /* hipStream_t of the mapping side (== lsdhip_ctx_stream on a non-pipelined context) */
int lsdhip_ctx_synchronize(lsdhip_ctx* ctx);
/* Lanes (no reference counterpart — S SlamSystems would each have their own mapping thread): the DepthMap call
 * chains of DIFFERENT depth maps are independent, so between lanes_begin(n) and lanes_end the caller may route each map's calls to
 * one of n extra streams (lane_select(lane), -1 = back to the context's stream) and the chains run side by side: finalizeKeyFrame +
 * createKeyFrame is ~18 small dependent launches per sequence, S sequences changing keyframe take S times that on one stream.  Work
 * queued before lanes_begin is visible on every lane; lanes_end orders the context's stream behind all lanes.  All calls on ONE map
 * between begin and end must use the same lane.  n <= 16. */
int lsdhip_ctx_lanes_begin(lsdhip_ctx* ctx, int n);
int lsdhip_ctx_lane_select(lsdhip_ctx* ctx, int lane);
int lsdhip_ctx_lanes_end(lsdhip_ctx* ctx);
const char* lsdhip_last_error(void);
/* per-level intrinsics fx,fy,cx,cy,fxi,fyi,cxi,cyi (C/DataStructures/Frame.cpp:445-459) */
int lsdhip_ctx_intrinsics(lsdhip_ctx* ctx, int level, float out[8]);

/* ---- Frame ---------------------------------------------------------------------------------------- */
/* Frame::Frame(id,w,h,K,timestamp,const unsigned char*) C/DataStructures/Frame.cpp:35-48 plus the lazy pyramid
 * builders buildImage/buildGradients/buildMaxGradients (Frame.cpp:491-767), executed eagerly on the device. */
int lsdhip_frame_create(lsdhip_ctx* ctx, int id, const uint8_t* gray_host, lsdhip_frame** out);
int lsdhip_frame_create_from_device(lsdhip_ctx* ctx, int id, const uint8_t* gray_dev, lsdhip_frame** out);
/* lsdhip_frame_create without the final wait: the upload and the pyramid kernels are queued (on a pipelined context: on the mapping
 * stream, i.e. beside the tracking job in flight — what the reference's image-loader thread does ahead of the tracking thread,
 * C/LiveSLAMWrapper.cpp:82-118).  gray_host must stay unchanged until a tracker call on the frame has returned or the context has
 * been synchronised; pinned memory makes the copy a true asynchronous DMA. */
int lsdhip_frame_create_async(lsdhip_ctx* ctx, int id, const uint8_t* gray_host, lsdhip_frame** out);
/* The new frames of n sequences at once: two launches for all of them instead of two per frame; the planes of every frame are
 * those lsdhip_frame_create_from_device (images_on_device != 0) / lsdhip_frame_create (== 0: returns after the uploads) produce.
 * gray: n image pointers (all device or all host), out: n handles.  Several sequences sharing one GPU (BASELINE.json configs[3]
 * with more sequences than GPUs) enter the device through this call; no reference counterpart (one Frame constructor per image). */
int lsdhip_frame_create_batch(lsdhip_ctx* ctx, int n, const int* ids, const uint8_t* const* gray, int images_on_device, lsdhip_frame** out);
/* Frame-memory pool (FrameMemory::getBuffer / returnBuffer keep returned buffers for reuse, C/DataStructures/FrameMemory.cpp:67-117):
 * make sure n frame arenas are allocated and waiting.  A loop that keeps its keyframes alive takes a fresh arena per keyframe; without
 * the pool that is a hipMalloc of ~20 MB (0.5 ms) every keyframe. */
int lsdhip_ctx_reserve_frames(lsdhip_ctx* ctx, int n);
void lsdhip_frame_destroy(lsdhip_frame* f);
int lsdhip_frame_id(lsdhip_frame* f);
/* Frame::image/gradients/maxGradients/idepth/idepthVar(level) accessors (Frame.h:357-418), copied to host.
 * what: 0 image, 1 gradients (4 floats per pixel: gx,gy,I,0), 2 maxGradients (level 0), 3 idepth, 4 idepthVar,
 * 5 (levels >= 1; not a member of the reference's Frame) the level's reference blocks as the tracker's throughput mode reads them: per 256
 *   consecutive pixels the in-block offsets of the pixels TrackingReference::makePointCloud would take (TrackingReference.cpp:120-131),
 *   compacted in pixel order (256 bytes per block; list slot s in byte (s mod 64) * 4 + s / 64), then one int32 count per block;
 *   out_host holds ceil(pixels / 256) * 260 bytes
 * 6 (level 0; not a member of the reference's Frame) the gradient candidates the batched DepthMap update walks: per 1024 consecutive
 *   pixels the in-group offsets (uint16, pixel order) of the pixels inside the 3-pixel border with maxGradients >= minUseGrad — the two
 *   tests of DepthMap::observeDepthRow that depend on the keyframe alone (DepthMap.cpp:111-131) — then one uint16 count per group;
 *   out_host holds ceil(pixels / 1024) * 1025 * 2 bytes */
int lsdhip_frame_download(lsdhip_frame* f, int what, int level, float* out_host);
/* Frame::setDepthFromGroundTruth (Frame.cpp:245-293) */
int lsdhip_frame_set_depth_gt(lsdhip_frame* f, const float* depth_host, float cov_scale);
/* raw level-0 idepth / idepthVar planes (what Frame::setDepth leaves behind, Frame.cpp:199-243) */
int lsdhip_frame_set_depth_planes(lsdhip_frame* f, const float* idepth_host, const float* idepthVar_host);
/* test / synthetic-benchmark hook: overwrite the level-0 maxGradients plane (Frame::maxGradients(0)); the frame's gradient candidates
 * follow.  Call it before the frame becomes a DepthMap's keyframe (a map that already uses it must be re-initialised / uploaded again). */
int lsdhip_frame_set_maxgrad(lsdhip_frame* f, const float* maxgrad_host);
/* Frame::refPixelWasGood() (Frame.h:421-437): level-1 mask, bytes 0xFF until the tracker writes 0/1.
 * returns 1 and fills out_host if the mask exists, 0 if it was never created / was cleared. */
int lsdhip_frame_get_wasgood(lsdhip_frame* f, uint8_t* out_host);
int lsdhip_frame_set_wasgood(lsdhip_frame* f, const uint8_t* in_host);
int lsdhip_frame_clear_wasgood(lsdhip_frame* f); /* Frame::clear_refPixelWasGood */
/* FramePoseStruct::thisToParent_raw / trackingParent and Frame::initialTrackedResidual
 * (written by trackFrame, SE3Tracker.cpp:482-484; read by DepthMap, DepthMap.cpp:1095-1101, :1918) */
int lsdhip_frame_set_pose(lsdhip_frame* f, const double thisToParent_sim3[8], lsdhip_frame* trackingParent,
                          float initialTrackedResidual);
int lsdhip_frame_get_pose(lsdhip_frame* f, double thisToParent_sim3[8]);
/* se3FromSim3(reference->getCamToWorld().inverse() * frame->getCamToWorld()), the initial estimate SlamSystem::trackFrame forms
 * (C/SlamSystem.cpp:918-920), for the pose-tree shapes of this path: `frame` tracked on `reference`, or both tracked on the same parent
 * (the frame tracked on the old keyframe while the mapper was promoting `reference`).  LSDHIP_E_STATE otherwise. */
int lsdhip_frame_relative_pose(lsdhip_frame* reference, lsdhip_frame* frame, double frameToReference[7]);
/* TrackingReference::importFrame (C/Tracking/TrackingReference.cpp:71-87) as the hand-over point between the mapping and the tracking
 * side of a pipelined context: the planes the keyframe's latest Frame::setDepth wrote become what tracker calls read (the reference's
 * tracking thread does this when it finds depthHasBeenUpdatedFlag set, C/SlamSystem.cpp:907-912).  No-op on other contexts. */
int lsdhip_frame_publish_depth(lsdhip_frame* f);
/* out: initialTrackedResidual, meanIdepth, numPoints, numFramesTrackedOnThis, numMappedOnThis,
 *      numMappedOnThisTotal, depthHasBeenUpdatedFlag, reserved */
int lsdhip_frame_stats(lsdhip_frame* f, float out[8]);
int lsdhip_frame_set_counters(lsdhip_frame* f, int numFramesTrackedOnThis, int numMappedOnThis,
                              int numMappedOnThisTotal, int depthHasBeenUpdatedFlag);
/* Frame::depthHasBeenUpdatedFlag alone (read / reset by SlamSystem::trackFrame, C/SlamSystem.cpp:907-911): host-side flag,
 * never synchronises */
int lsdhip_frame_depth_updated(lsdhip_frame* f);
int lsdhip_frame_clear_depth_updated(lsdhip_frame* f);
/* TrackingReference::makePointCloud(level) (C/Tracking/TrackingReference.cpp:96-147) for the keyframe `kf`,
 * in the reference's x-outer/y-inner order.  Any output pointer may be NULL.  Returns the number of points. */
int lsdhip_ref_pointcloud(lsdhip_frame* kf, int level, float* pos3_host, float* colorAndVar2_host,
                          float* grad2_host, int* idx_host);

/* ---- SE3Tracker ----------------------------------------------------------------------------------- */
typedef struct lsdhip_track_result {
  double frameToReference[7]; /* return value of trackFrame (SE3Tracker.cpp:485) */
  float pointUsage, lastGoodCount, lastBadCount, lastMeanRes, lastResidual;
  float affineEstimation_a, affineEstimation_b;
  int diverged, trackingWasGood;
  int numEvaluations, numWarpUpdates; /* instrumentation: residual-kernel launches, LM outer iterations */
} lsdhip_track_result;

/* one fused evaluation K1+K2+K3 (calcResidualAndBuffers + calcWeightsAndResidual + calculateWarpUpdate) */
typedef struct lsdhip_residual_record {
  int warped_size;                /* buf_warped_size */
  float goodCount, badCount, pointUsage, meanRes, retval;
  float affine_a_lastIt, affine_b_lastIt;
  float weightedError;            /* calcWeightsAndResidual[SSE] return value */
  float A[36], b[6], lsError;     /* LGS6 after finish() */
  double num_constraints;
} lsdhip_residual_record;

/* SE3Tracker::SE3Tracker (SE3Tracker.cpp:46-94); settings = DenseDepthTrackerSettings (settings.h:355-402) */
int lsdhip_tracker_create(lsdhip_ctx* ctx, lsdhip_tracker** out);
void lsdhip_tracker_destroy(lsdhip_tracker* t);
int lsdhip_tracker_set_max_its(lsdhip_tracker* t, const int maxItsPerLvl[LSDHIP_PYRAMID_LEVELS]);
/* DenseDepthTrackerSettings, every field (C/util/settings.h:355-402; defaults C/util/settings.h:360-400): the reference exposes
 * `settings` as a public member of SE3Tracker (C/Tracking/SE3Tracker.h:49) and SlamSystem edits it (SlamSystem.cpp:80-81). */
typedef struct lsdhip_tracker_settings {
  float lambdaSuccessFac, lambdaFailFac;
  float lambdaInitial[LSDHIP_PYRAMID_LEVELS], stepSizeMin[LSDHIP_PYRAMID_LEVELS], convergenceEps[LSDHIP_PYRAMID_LEVELS];
  int maxItsPerLvl[LSDHIP_PYRAMID_LEVELS];
  float lambdaInitialTestTrack, stepSizeMinTestTrack, convergenceEpsTestTrack, maxItsTestTrack;   /* trackFrameOnPermaref */
  float huber_d, var_weight;
} lsdhip_tracker_settings;
int lsdhip_tracker_get_settings(const lsdhip_tracker* t, lsdhip_tracker_settings* out);
int lsdhip_tracker_set_settings(lsdhip_tracker* t, const lsdhip_tracker_settings* in);
/* Diagnostics (no reference counterpart): out[0..2] = 0 (reserved: the counters of round 3's coarse-level cluster kernel, removed in
 * round 4 after it measured slower than the launch chain, profiles/r03_notes.md section 3); residual evaluations of the last job per
 * pyramid level 0..4 in out[3..7]. */
int lsdhip_tracker_exec_stats(const lsdhip_tracker* t, int out[8]);
/* Execution strategy of the launch-per-step chain: the LM loop's "increase lambda and retry" sequence
 * (C/Tracking/SE3Tracker.cpp:341-447) depends only on A, b and lambda, so a step evaluates the next `trials` retries side by side and
 * the following step consumes them in the reference's order — same decisions, same evaluation counts, fewer dependent steps.
 * Default (no call): AUTOMATIC, by the size of the level — 6 trials on levels of up to 6 K pixels, 5 up to 88 K (on 80 workgroups per
 * trial above 24 K), one evaluation per step on larger (work-bound) levels: 5 / 5 / 6 at levels 1 / 2 / 3 of a 640x480 frame.
 * This call switches the automatic policy off: `trials` (1..6) at every level; trials = 1: one evaluation per step.
 * finestLevelWorkgroups: workgroups per trial at the job's finest level while trials > 1 — 0 keeps the current value (default 80),
 * other values are rounded down to a multiple of 8 (one band of tiles per XCD), values below 8 up to 8.  The per-level overrides of
 * the environment (LSDHIP_SPEC_LEVELS) are cleared by this call. */
int lsdhip_tracker_set_speculation(lsdhip_tracker* t, int trials, int finestLevelWorkgroups);
/* Throughput-mode batches (lsdhip_tracker_track_batch): from `minJobs` jobs on, the levels of at most 4800 pixels and 4608 valid
 * reference points that do not write refPixelWasGood (levels 4 and 3 of a 640x480 job) are walked by ONE workgroup per job — the tracked
 * frame's texel plane of the level staged in LDS (the tile of the bilinear taps), the level's reference points in registers, the whole LM
 * loop of the level in that workgroup — ahead of the lock-step rounds for the larger levels: same per-point arithmetic and LM decisions,
 * sums in that workgroup's order.  Default 24 jobs (below, the jobs have the chip to themselves
 * and the lock-step rounds are as fast); 0: never. */
int lsdhip_tracker_set_batch_coarse_min_jobs(lsdhip_tracker* t, int minJobs);
/* out[0] = k_track_step launches of the last job that evaluated (<= its numEvaluations), out[1] = most trials per step (the
 * per-level numbers follow the automatic policy above). */
int lsdhip_tracker_launch_stats(const lsdhip_tracker* t, int out[2]);
/* out[0] = k_track_step launches of the last job that evaluated = its dependent steps, out[1] = out[2] = 0 (reserved), out[3] = most
 * trials per step. */
int lsdhip_tracker_step_stats(const lsdhip_tracker* t, int out[4]);
/* The job summary a tracker polls for in pinned host memory carries a check word; the host takes the record only once its words add up
 * (they are separate posted writes: the words stored last have been seen to arrive after the `done` word).  out[0] = jobs polled,
 * out[1] = jobs whose record was incomplete when `done` arrived, out[2] = longest wait for the rest (ns), out[3] = words seen stale,
 * out[4] / out[5] = lowest / highest stale word index (-1: none). */
int lsdhip_tracker_summary_stats(const lsdhip_tracker* t, long long out[6]);
/* Measurement hook (no reference counterpart): the throughput-mode residual evaluation launch alone — n >= 8 jobs
 * (keyframes[j], frames[j]) at pyramid level `level` and poses refToFrame (n x 7 floats: q w x y z, t), `repeats` identical launches
 * between two HIP events.  Reports the mean launch time and the algorithmic bytes of one launch over all jobs (SURVEY.md 8(d)
 * formula): bytes / time is the kernel's position against the HBM roofline (profiles/r03_sizes.md, bench.py).
 * Side effect: at level 1 the launches write frames[j]'s refPixelWasGood plane like a trackFrame job does (those bytes are part of the
 * measured evaluation): do not run it on frames whose mask a later DepthMap::updateKeyframe is still to read. */
int lsdhip_tracker_eval_throughput(lsdhip_tracker* t, int n, lsdhip_frame** keyframes, lsdhip_frame** frames, const float* refToFrame,
                                   int level, int repeats, double* ms_per_launch, double* bytes_per_launch);
/* Host-side pipelining: `fn(user)` is called on the calling thread by lsdhip_tracker_track once the job's launches are
 * queued and before the host waits for the result — the place to queue independent work on the same context (the next
 * image's upload and pyramids, what the reference's image-loader thread does ahead of the tracking thread,
 * C/LiveSLAMWrapper.cpp:82-118).  fn = NULL removes the hook.  The hook must not call into this tracker.  lsdhip_tracker_track_batch
 * calls it too, once the batch's first budget of launches is queued. */
typedef void (*lsdhip_enqueue_hook)(void* user);
int lsdhip_tracker_set_enqueue_hook(lsdhip_tracker* t, lsdhip_enqueue_hook fn, void* user);
/* SE3Tracker::trackFrame(TrackingReference*, Frame*, const SE3& frameToReference_initialEstimate)
 * (SE3Tracker.cpp:280-486).  `keyframe` plays the role of reference->keyframe (its idepth planes must be set).
 * Side effects as in the reference: frame mask refPixelWasGood, frame pose / trackingParent /
 * initialTrackedResidual, keyframe numFramesTrackedOnThis++.  Returns LSDHIP_DIVERGED when diverged. */
int lsdhip_tracker_track(lsdhip_tracker* t, lsdhip_frame* keyframe, lsdhip_frame* frame,
                         const double frameToReference_initialEstimate[7], lsdhip_track_result* out);
/* trackFrame for n independent (keyframe, frame) pairs in the same kernel launches (one job per blockIdx.y): the decisions and
 * per-point arithmetic of n lsdhip_tracker_track calls, sums in another order (fewer, fatter workgroups per job); n evaluations share
 * every launch.  From 8 jobs on the batch runs in throughput mode: a lock-step round = one launch whose (trial, strip) workgroups x jobs
 * fill the chip's 768 workgroup slots, each repeating the LM decision in front of its evaluation.  This is how several sequences share
 * one GPU (BASELINE configs[3] with fewer GPUs than sequences) and how batches of keyframe candidates are checked
 * (SURVEY.md §8(f) N2).  inits: n x 7 doubles, results: n records.  Returns LSDHIP_DIVERGED if any job diverged. */
int lsdhip_tracker_track_batch(lsdhip_tracker* t, int n, lsdhip_frame** keyframes, lsdhip_frame** frames,
                               const double* frameToReference_initialEstimates, lsdhip_track_result* results);
/* K1+K2+K3 once, at a fixed referenceToFrame (float (qw,qx,qy,qz,tx,ty,tz)) — kernel-level parity hook. */
int lsdhip_tracker_evaluate(lsdhip_tracker* t, lsdhip_frame* keyframe, lsdhip_frame* frame,
                            const float referenceToFrame[7], int level, float affine_a, float affine_b,
                            lsdhip_residual_record* out);
/* SE3Tracker::trackFrameOnPermaref (SE3Tracker.cpp:162-272) / checkPermaRefOverlap (:121-157) on an explicit
 * level-4 point cloud (Frame::setPermaRef data, Frame.cpp:149-174). */
int lsdhip_tracker_track_permaref(lsdhip_tracker* t, const float* pos3_host, const float* colorAndVar2_host, int n,
                                  lsdhip_frame* frame, const double referenceToFrame[7], lsdhip_track_result* out);
/* trackFrameOnPermaref for n permanent references in the same launches (reference clouds concatenated, counts[j] points
 * each; each tracked against frames[j]) — many keyframe candidates against one new frame */
int lsdhip_tracker_track_permaref_batch(lsdhip_tracker* t, int n, const float* pos3_host, const float* colorAndVar2_host,
                                        const int* counts, lsdhip_frame** frames, const double* referenceToFrame,
                                        lsdhip_track_result* results);
int lsdhip_tracker_check_overlap(lsdhip_tracker* t, const float* pos3_host, int n, const double referenceToFrame[7],
                                 float* usage_out);

/* ---- Sim3Tracker ----------------------------------------------------------------- */
/* Sim3 as double[8] = (qw,qx,qy,qz,tx,ty,tz,scale): p' = scale * R(q) p + t.
 * Replaces Sim3Tracker (C/Tracking/Sim3Tracker.h:71-187): the keyframe-to-keyframe constraint tracker of
 * SlamSystem::tryTrackSim3 (C/SlamSystem.cpp:1153-1228). */
typedef struct lsdhip_sim3tracker lsdhip_sim3tracker;
typedef struct lsdhip_sim3_result {
  double frameToReference[8];          /* return value of trackFrameSim3; identity when diverged */
  float lastResidual, lastDepthResidual, lastPhotometricResidual; /* Sim3Tracker.h:89-91 */
  float pointUsage;                    /* Sim3Tracker.h:85 */
  float affineEstimation_a, affineEstimation_b;
  int diverged;
  int numEvaluations;
  float lastSim3Hessian[49];           /* Sim3Tracker.h:93, row-major */
} lsdhip_sim3_result;
/* one evaluation (calcSim3Buffers + calcSim3WeightsAndResidual + calcSim3LGS, Sim3Tracker.cpp:414-983) */
typedef struct lsdhip_sim3_eval_record {
  int warped_size;
  float pointUsage, affine_a_lastIt, affine_b_lastIt;
  float sumResD, sumResP;
  int numTermsD, numTermsP;
  float meanD, meanP, mean;
  float A[49], b[7];                   /* LGS7 after initializeFrom(ls6, ls4), before the division by num_constraints */
  double num_constraints;
} lsdhip_sim3_eval_record;
int lsdhip_sim3tracker_create(lsdhip_ctx* ctx, lsdhip_sim3tracker** out);     /* Sim3Tracker::Sim3Tracker, Sim3Tracker.cpp:43-103 */
void lsdhip_sim3tracker_destroy(lsdhip_sim3tracker* t);
int lsdhip_sim3tracker_set_max_its(lsdhip_sim3tracker* t, const int maxItsPerLvl[5]);
/* Sim3Tracker::trackFrameSim3(reference, frame, frameToReference_initialEstimate, startLevel, finalLevel)
 * (Sim3Tracker.cpp:149-378).  `keyframe` is reference->keyframe; both frames need inverse-depth planes (Frame::setDepth).
 * Returns LSDHIP_DIVERGED when the reference sets diverged or returns Sim3() early. */
int lsdhip_sim3tracker_track(lsdhip_sim3tracker* t, lsdhip_frame* keyframe, lsdhip_frame* frame, const double init_frameToReference[8],
                             int startLevel, int finalLevel, lsdhip_sim3_result* out);
/* n independent trackFrameSim3 jobs advanced in lock step, their evaluations sharing launches (the constraint search tests
 * every candidate in both directions, C/SlamSystem.cpp:1140-1187).  inits: n x 8 doubles, results: n records; each job
 * computes what a single call computes.  Returns LSDHIP_DIVERGED if any job returned early (see results[j].diverged). */
int lsdhip_sim3tracker_track_batch(lsdhip_sim3tracker* t, int n, lsdhip_frame** keyframes, lsdhip_frame** frames,
                                   const double* init_frameToReference, int startLevel, int finalLevel, lsdhip_sim3_result* results);
/* test hooks for the host-side arithmetic of the Sim3 LM step (pure CPU, usable without a GPU): out = exp(increment) *
 * referenceToFrame with Sophus semantics (sim3.hpp:417-428, :160-163); x = A.ldlt().solve(b) for the 7x7 system */
/* the same for the SE3 tracker's LM step (float, Sophus SE3f semantics, C/Tracking/SE3Tracker.cpp:356-363): the functions the
 * device kernel uses, compiled for the host */
int lsdhip_host_se3f_step(const float increment[6], const float referenceToFrame[7], float out[7]);
int lsdhip_host_ldlt6(const float A[36], const float b[6], float x[6]);
int lsdhip_host_sim3_step(const double increment[7], const double referenceToFrame[8], double out[8]);
int lsdhip_host_ldlt7(const float A[49], const float b[7], float x[7]);
/* test hook: one evaluation at referenceToFrame on `level` with affine (a, b) */
int lsdhip_sim3tracker_evaluate(lsdhip_sim3tracker* t, lsdhip_frame* keyframe, lsdhip_frame* frame, const double referenceToFrame[8],
                                int level, float aff_a, float aff_b, lsdhip_sim3_eval_record* out);

/* ---- DepthMap ------------------------------------------------------------------------------------- */
/* DepthMap::DepthMap (DepthMap.cpp:41-83) */
int lsdhip_depth_create(lsdhip_ctx* ctx, lsdhip_depthmap** out);
void lsdhip_depth_destroy(lsdhip_depthmap* dm);
int lsdhip_depth_is_valid(lsdhip_depthmap* dm);          /* DepthMap::isValid */
int lsdhip_depth_invalidate(lsdhip_depthmap* dm);        /* DepthMap::invalidate */
int lsdhip_depth_reset(lsdhip_depthmap* dm);             /* DepthMap::reset (DepthMap.cpp:102-108) */
int lsdhip_depth_init_gt(lsdhip_depthmap* dm, lsdhip_frame* kf);      /* initializeFromGTDepth :965-1018 */
int lsdhip_depth_init_random(lsdhip_depthmap* dm, lsdhip_frame* kf);  /* initializeRandomly :883-916 */
int lsdhip_depth_set_from_existing(lsdhip_depthmap* dm, lsdhip_frame* kf); /* setFromExistingKF :920-962 */
/* DepthMap::updateKeyframe(std::deque<std::shared_ptr<Frame>> referenceFrames) (DepthMap.cpp:1072-1213).
 * refs[0] is the oldest, refs[n-1] the newest frame of the deque. */
int lsdhip_depth_update(lsdhip_depthmap* dm, lsdhip_frame** refs, int n);
/* DepthMap::updateKeyframe for the depth maps of n sequences (one context), one tracked frame each (the blockUntilMapped deque of
 * SlamSystem.cpp:559-571), in three launches for all of them (blockIdx.z = map): observe, fill holes + regularise + Frame::setDepth,
 * idepth pyramid.  Every map ends bit-identical to lsdhip_depth_update(maps[j], &refs[j], 1).  Keyframes whose
 * depthHasBeenUpdatedFlag is still set skip Frame::setDepth, as in the single call. */
int lsdhip_depth_update_batch(int n, lsdhip_depthmap** maps, lsdhip_frame** refs);
/* DepthMap::createKeyFrame(Frame* new_keyframe) (DepthMap.cpp:1222-1327); rescale_out = rescaleFactor (:1294) */
int lsdhip_depth_create_keyframe(lsdhip_depthmap* dm, lsdhip_frame* new_keyframe, float* rescale_out);
/* DepthMap::finalizeKeyFrame (DepthMap.cpp:1363-1395), incl. Frame::takeReActivationData */
int lsdhip_depth_finalize(lsdhip_depthmap* dm);
/* DepthMap::finalizeKeyFrame() followed by DepthMap::createKeyFrame(new_keyframes[j]) (DepthMap.cpp:1363-1395, :1222-1327, propagateDepth
 * :475-653) for the depth maps of n sequences (one context; n = 1: one sequence) in six launches shared by all of them (blockIdx.z =
 * map): [finalize pass + Frame::setDepth + takeReActivationData + propagation candidates] -> [propagation merge] -> regularizeDepthMap(true)
 * -> [fill holes + regularizeDepthMap(false) + rescale sums] -> [rescale + Frame::setDepth] -> both keyframes' idepth pyramids.  Planes,
 * re-activation data, pyramids and deferred results as lsdhip_depth_finalize + lsdhip_depth_create_keyframe leave them.
 * rescale_out: NULL or n floats (asking for them waits for the device). */
int lsdhip_depth_change_keyframe_batch(int n, lsdhip_depthmap** maps, lsdhip_frame** new_keyframes, float* rescale_out);
/* currentDepthMap <-> host in the reference's 32-byte AoS layout (debug / parity / drop-in users that read it) */
int lsdhip_depth_download(lsdhip_depthmap* dm, lsdhip_hypothesis* out_host);
int lsdhip_depth_upload(lsdhip_depthmap* dm, lsdhip_frame* kf, const lsdhip_hypothesis* in_host, int reactivated);
/* single stages for kernel-level parity: 0 observeDepth (refs/n as in update), 1 regularizeDepthMapFillHoles,
 * 2 regularizeDepthMap(false,24), 3 regularizeDepthMap(true,24), 4 propagateDepth(refs[0] = new keyframe),
 * 5 regularizeDepthMapFillHoles + regularizeDepthMap(false,24) fused in one launch (what updateKeyframe runs) */
int lsdhip_depth_stage(lsdhip_depthmap* dm, int stage, lsdhip_frame** refs, int n);
/* stage 5 (the fused fill-holes + regularise pass, regularizeDepthMapFillHoles + regularizeDepthMap, DepthMap.cpp:656-720, :758-880) on
 * tile rows [tile_row0, tile_row0 + n_tile_rows) only; a tile row = 8 map rows.  The parts of one pass may be issued in any order;
 * last != 0 on the final part of the pass (it swaps the validity planes).  Rows no part covers must be refreshed by the caller before
 * the next pass reads them (row-band decomposition: they are another band's rows and arrive with the halo exchange). */
int lsdhip_depth_stage_rows(lsdhip_depthmap* dm, int stage, int tile_row0, int n_tile_rows, int last);
/* n such parts (stage 5), of the same or of different maps of one context, in ONE launch (blockIdx.z = part): the windows of a process in
 * the row-band decomposition (BASELINE.json configs[4]).  No validity planes are swapped: lsdhip_depth_stage_rows(dm, 5, 0, 0, 1) per map
 * once its parts are queued. */
int lsdhip_depth_stage_rows_batch(lsdhip_ctx* c, int n, lsdhip_depthmap** maps, const int* tile_row0, const int* n_tile_rows);
/* smoothed idepth / variance planes of the active keyframe (what setDepth produced), device to device —
 * the payload the multi-GPU gather collects per keyframe. */
int lsdhip_depth_copy_planes_dev(lsdhip_depthmap* dm, float* idepth_dev, float* idepthVar_dev);
/* device-to-device copy ordered on the context's stream (no host synchronisation): the root's own share of a keyframe gather */
int lsdhip_ctx_copy_dev(lsdhip_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);
/* device buffers on the context's device for callers that do not link the HIP runtime themselves (the C++ drivers: staging
 * of halo rows, keyframe rings); free waits for the context's stream */
int lsdhip_ctx_alloc_dev(lsdhip_ctx* ctx, size_t bytes, void** out_dev);
/* Inter-process exchange on one node without RCCL (the driver library's second transport; no reference counterpart): a device
 * allocation is exported as a 64-byte hipIpcMemHandle_t and mapped by the peer process; flags are ints inside such an allocation
 * whose values only grow.  _flag_set / _flag_wait are stream-ordered: set publishes what the stream did before it (system scope),
 * wait holds the stream until the flag has reached `value` (bounded spin, ~2 s: *fail_dev = value on a time-out). */
int lsdhip_ctx_ipc_export(lsdhip_ctx* ctx, void* dev, unsigned char handle64[64]);
int lsdhip_ctx_ipc_open(lsdhip_ctx* ctx, const unsigned char handle64[64], void** out_dev);
int lsdhip_ctx_ipc_close(lsdhip_ctx* ctx, void* dev);
int lsdhip_ctx_flag_set(lsdhip_ctx* ctx, int* flag_dev, int value);
int lsdhip_ctx_flag_wait(lsdhip_ctx* ctx, const int* flag_dev, int value, int* fail_dev);
int lsdhip_ctx_memset_dev(lsdhip_ctx* ctx, void* dev, int byte, size_t bytes);
int lsdhip_ctx_read_dev(lsdhip_ctx* ctx, void* host, const void* dev, size_t bytes);   /* synchronises the stream */
/* Transport stream (exchange under compute): between _aux_begin and _aux_end the transport primitives of this context
 * (lsdhip_depth_copy_rows_batch, lsdhip_ctx_copy_dev, lsdhip_ctx_flag_set / _wait; a caller's RCCL calls on lsdhip_ctx_aux_stream) are
 * queued on a second stream, ordered behind everything the main stream held at _aux_begin; _aux_join makes the main stream wait for
 * them.  lsdband_run sends a pass's boundary rows this way while the interior rows of the same pass are computed. */
int lsdhip_ctx_aux_begin(lsdhip_ctx* ctx);
int lsdhip_ctx_aux_end(lsdhip_ctx* ctx);
int lsdhip_ctx_aux_join(lsdhip_ctx* ctx);
void* lsdhip_ctx_aux_stream(lsdhip_ctx* ctx);   /* hipStream_t; null before the first lsdhip_ctx_aux_begin */
void lsdhip_host_mark(int id);   /* developer instrumentation (LSDHIP_HOST_TRACE=1): host time between consecutive marks, printed at context destruction */
int lsdhip_ctx_free_dev(lsdhip_ctx* ctx, void* dev);
/* (returns after the copies have finished; on an asynchronous context — lsdhip_ctx_set_async — after they are queued) */
/* rows [row0, row0+nrows) of the eight hypothesis planes <-> one packed device buffer (29 bytes per pixel, plane after
 * plane in the order of lsdhip_hypothesis): halo exchange of the row-band decomposition.
 * to_map != 0 copies the buffer into the map. */
int lsdhip_depth_copy_rows_dev(lsdhip_depthmap* dm, int row0, int nrows, void* packed_dev, int to_map);
/* The same for several row ranges in ONE launch, ordered on the context's stream (no host synchronisation in asynchronous
 * contexts): each item copies nrows rows from its source to its destination, where a side is either rows [row0, row0 + nrows)
 * of a depth map of this context (map != NULL) or a packed device buffer (map == NULL, layout as above).  Map -> map refreshes the
 * halo of a window from the window that owns those rows when both live on one GPU; map -> packed / packed -> map bracket the
 * ncclSend / ncclRecv of the multi-GPU exchange. */
typedef struct lsdhip_row_copy {
  lsdhip_depthmap* src_map; int src_row0; void* src_packed;
  lsdhip_depthmap* dst_map; int dst_row0; void* dst_packed;
  int nrows;
} lsdhip_row_copy;
int lsdhip_depth_copy_rows_batch(lsdhip_ctx* c, int n, const lsdhip_row_copy* items);
/* timing fields DepthMap keeps public (DepthMap.h:86-93): msUpdate, msCreate, msFinalize, msObserve, msRegularize,
 * msPropagate, msFillHoles, msSetDepth (exponential moving averages, ms) */
int lsdhip_depth_timings(lsdhip_depthmap* dm, float out[8]);
/* GPU time (HIP events on the context's stream, ms, summed since creation) of the TIMED calls of updateKeyframe [0],
 * createKeyFrame [1], finalizeKeyFrame [2] and how many calls were timed: every 7th updateKeyframe and every 2nd createKeyFrame /
 * finalizeKeyFrame is bracketed (an event record delays the kernel behind it by ~10 us); mean = ms / calls; synchronises the stream */
int lsdhip_depth_gpu_times(lsdhip_depthmap* dm, double ms_out[3], long long calls_out[3]);
/* GPU time (ms, summed) and count of the observe kernel alone (DepthMap::observeDepth, C/DepthEstimation/DepthMap.cpp:147-150),
 * sampled on every 7th updateKeyframe while lsdhip_prof_enable is on; synchronises the stream. */
/* Work of the k_observe launches sampled while profiling (every 7th updateKeyframe): out[0] = launches counted, out[1] = pixels that
 * entered the epipolar search (DepthMap::doLineStereo calls), out[2] = steps of the search loops (sum of loopCounter,
 * DepthMap.cpp:1622-1744).  bench.py: stereo_steps_per_s, roofline_depth on the bytes of searched pixels.  Synchronises. */
int lsdhip_depth_observe_work(lsdhip_depthmap* dm, double out[3]);
int lsdhip_depth_observe_time(lsdhip_depthmap* dm, double* ms_out, long long* calls_out);

/* ---- measurement hooks ---------------------------------------------------------------------------- */
/* Accumulated HIP-event time (ms) and launch count of the residual kernel on the context's stream since the last
 * reset, plus the algorithmic bytes those launches moved; bench.py's roofline leg. */
int lsdhip_prof_enable(lsdhip_ctx* ctx, int on);
int lsdhip_prof_read(lsdhip_ctx* ctx, double* residual_ms, long long* residual_launches, double* residual_bytes);
/* While profiling is on, every third call of the batched entries brackets its shared launches with HIP events.  ms / calls / units: 5
 * entries each — [0] lsdhip_frame_create_batch (image pyramids + gradients), [1] the observe launch(es), [2] fill holes + regularise
 * (+ setDepth) and [3] the idepth pyramids of lsdhip_depth_update_batch, [4] the launches of lsdhip_depth_change_keyframe_batch;
 * units = map pixels the bracketed launches processed.  obs (may be NULL): [0] walk launches counted, [1] their searches (doLineStereo
 * calls), [2] their walk steps.  Waits for the brackets still in flight; lsdhip_prof_reset clears. */
int lsdhip_ctx_batch_prof_read(lsdhip_ctx* ctx, double* ms, long long* calls, double* units, double obs[3]);
int lsdhip_prof_reset(lsdhip_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* LSDHIP_H */
