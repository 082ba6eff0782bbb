// Internal structures of liblsdhip.so (MI355X / gfx950).  Host-side bookkeeping mirrors what the reference keeps in
// Frame / FramePoseStruct / SE3Tracker / DepthMap members; all per-pixel data lives in HBM as SoA planes.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/lsdhip.h"

#define LSD_LEVELS LSDHIP_PYRAMID_LEVELS
#define LSD_TRACK_MIN_LEVEL 1   // SE3TRACKING_MIN_LEVEL, C/util/settings.h:98
#define LSD_TRACK_MAX_LEVEL 5   // SE3TRACKING_MAX_LEVEL
#define LSD_QUICK_KF_CHECK_LVL 4

void lsd_set_error(const char* fmt, ...);
#define HIPCHK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      lsd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return LSDHIP_E_HIP;                                                                 \
    }                                                                                      \
  } while (0)

#include "pose_math.hpp"  // lsdm:: pose algebra (host + device)

#include "rcp_exact.hpp"   // lsd_rcp_exact: 1.0f / x bit for bit in 4 instructions (its own header: tools/rcp_exhaustive.hip checks the same code)

// ---- device-visible parameter blocks --------------------------------------------------------------------
struct LevelIntr { float fx, fy, cx, cy, fxi, fyi, cxi, cyi; };

// raw sums one residual-kernel evaluation produces (before the tail-drop correction and normalisation)
#define RS_NUM 48
enum {
  RS_M = 0,        // in-image point count (buf_warped_size)
  RS_GOOD, RS_BAD, RS_SUMRES2, RS_SUMSIGNED, RS_SXX, RS_SYY, RS_SX, RS_SY, RS_SW, RS_USAGE,
  RS_WERR,         // sum of wh*w_p*r^2 (calcWeightsAndResidual)
  RS_A0,           // 21 upper-triangular entries of J J^T w
  RS_B0 = RS_A0 + 21,  // 6 entries of J r w
  RS_ERR = RS_B0 + 6,  // sum w r^2 (LGS error)
  RS_NREF,         // number of valid reference points (numData[level])
  RS_END,
  RS_TOP0 = RS_END // 3 slots (int bits): the workgroup's largest reference-order keys among in-image points
};
static_assert(RS_TOP0 + 3 <= RS_NUM, "record too small");
static_assert(RS_END <= RS_NUM, "record too small");

// One pyramid level of a tracking job, as the residual kernel sees it.
struct TrackLevel {
  const float* kf_idepth;    // keyframe (reference) planes at this level
  const float* kf_idepthVar;
  const float* kf_image;
  const uint8_t* kf_refBlk;  // reference blocks of the level (k_ref_blocks, frame.hip): per 256 consecutive pixels the in-block offsets of the valid
                             // reference pixels, compacted in pixel order (256 bytes), and behind all of them one count per block
  const float4* fr_grad;     // tracked-frame texels (gx, gy, I, 0)
  const float* pts_pos;      // explicit point list (permaref path) instead of keyframe planes; npts < 0 => dense grid
  const float* pts_colvar;
  int npts;
  int w, h;
  int nblocks;               // workgroups (tiles) that have work at this level
  int singlePass;            // nblocks * workgroup size >= points: every lane evaluates at most one point
  int tilePx;                // > 0: batch throughput mode, one strip of tilePx pixels per workgroup (compacted in LDS)
  float fx, fy, cx, cy, fxi, fyi, cxi, cyi;
  float lambdaInitial, stepSizeMin, convergenceEps;
  int maxIts;
  float minWarped;           // MIN_GOODPERALL_PIXEL_ABSMIN * (width>>lvl) * (height>>lvl)
  int writeMask;             // level == SE3TRACKING_MIN_LEVEL: write frame->refPixelWasGood
};

// Everything one trackFrame call needs, resident in HBM (uploaded once per call).
struct TrackJob {
  TrackLevel lv[LSD_LEVELS];
  uint8_t* wasGood;
  float cameraPixelNoise2, var_weight, huber_half;
  float lambdaSuccessFac, lambdaFailFac;
  int useAffine;
  int lastLevel;             // the LM loop runs levels topLevel .. lastLevel
  int topLevel;              // first (coarsest) level of the job
  lsdm::SE3fH T0;            // referenceToFrame the job starts from
  float aff_a0, aff_b0;      // affine-lighting parameters the job starts from
  int evalOnly;              // 1: a step only finalises the sums of one evaluation (kernel-level parity hook / host LM)
  int trackFrameSemantics;   // lastResidual bookkeeping of trackFrame (1) vs trackFrameOnPermaref (0)
};
// Reject-chain speculation (k_track_step, single jobs): a launch evaluates the next trials of the LM loop's "increase lambda
// and retry" chain (SE3Tracker.cpp:341-447) side by side — they depend only on A, b and lambda, not on each other's
// residuals — and the next launch consumes them in the reference's order.
struct TrackSpec {
  int specC;                 // <= 1: one evaluation per launch
  int specGrid;              // workgroups per trial (grid = specGrid x most trials of any level)
  int trials[LSD_LEVELS];    // trials per launch at each level
  uint8_t* wasGoodSide;      // refPixelWasGood planes of trials 1 .. (trial 0 writes the frame's own plane)
  unsigned maskStride;
  int seq;                   // progress tag this launch reports to the host (TrackSummary::seq); 0 = none
  int copyMask;              // 1: the finishing launch copies the side plane of the last executed trial into the frame's plane; 0 (pipelined
                             //    contexts): the host queues that merge on the mapping stream instead (the finishing launch must leave nothing
                             //    behind that a mapping kernel could still be waiting for once the host has seen `done`)
  unsigned long long dbgCum;         // developer build LSD_ORDER_CHECK: workgroups of all launches queued before this one
  unsigned long long* dbgCounters;   // ... [0] workgroups finished, [1] workgroups that started before all earlier ones had finished, [2] largest deficit seen
  int last;                  // 1: the last launch of the enqueued budget — if the job is not finished when it ends it says so (TrackSummary::exhausted)
#ifdef LSD_PHASE_TRACE
  int traceWg;               // developer build: the workgroup (blockIdx.x of job 0) of a batch launch that leaves the per-phase timestamps
#endif
#ifdef LSD_DEVTOOLS
  int* dbgLog;               // developer build: 16 ints per launch of the job (slot = launch ordinal), written by workgroup 0 — tools/launch_count_stress.py
#endif
};

// Levenberg-Marquardt state of a tracking job, resident in HBM, advanced by k_lm_step.
#define LSD_SPEC_MAX 6
struct TrackState {
  lsdm::SE3fH T;             // last accepted referenceToFrame
  lsdm::SE3fH Tn;            // pose being evaluated
  float R[9], t[3];          // rotation matrix / translation of Tn (what the residual kernel reads)
  float aff_a, aff_b, aff_a_lastIt, aff_b_lastIt;
  float lastErr, LM_lambda, last_residual;
  int level, iteration, incTry, phase;   // phase 0: first evaluation of a level, 1: trial evaluation
  int pending;               // partial sums of an evaluation are waiting for their LM step
  float A[36], b[6];         // normal equations of the last accepted evaluation (LGS6 after finish())
  float inc[6];              // increment of the trial under evaluation
  float bytes;               // algorithmic bytes moved by the evaluations so far (bench.py roofline leg)
  int done, diverged;
  int numEvaluations, numWarpUpdates;
  float pointUsage, goodCount, badCount, meanRes;
  int levelEvals[LSD_LEVELS];  // evaluations per pyramid level (diagnostics / bench line)
  int ncand;                 // trials the launch that produced the pending sums evaluated (>= 1)
  int lastCand;              // which of them was the last one the LM loop actually executed
  int numLaunches;           // k_track_step launches that did an evaluation so far
};

// What the host reads back (pinned, device-mapped): written by k_lm_step when the job finishes (or every step in
// evalOnly mode).
struct TrackSummary {
  int done;                  // raised last: the job's tag (TrackSpec::seq >> 12) for jobs the host polls for, 1 for jobs it synchronises the stream for
  int diverged, level, numEvaluations, numWarpUpdates, pad_[3];
  float q[4], t[3];
  float lastResidual, pointUsage, goodCount, badCount, meanRes, aff_a, aff_b, aff_a_lastIt, aff_b_lastIt;
  float sums[RS_NUM];        // raw sums of the last evaluation (tail-drop corrected)
  double bytes;              // algorithmic bytes of all evaluations of the job
  int levelEvals[LSD_LEVELS];
  int numLaunches;           // k_track_step launches that evaluated (< numEvaluations with reject-chain speculation)
  int lastCand;              // trial (within the launch that produced the final sums) the LM loop executed last: its mask plane is the frame's
  int seq;                   // (job tag << 12) | ordinal of the latest k_track_step launch of the chain that has started
  int exhausted;             // seq of the budget's last launch, written when that launch ends with the job unfinished: the host tops up
  unsigned check;            // position-weighted sum of the words [1, seq) + the `done` word (lsd_summary_check): the host accepts a summary only
                             // when it adds up — the words of one record are separate posted writes across PCIe and have been seen to land
                             // AFTER the `done` word that was stored behind a system-scope release fence (profiles/r06_notes.md section 1)
};
#define LSD_SUMMARY_CHECK_WORDS (offsetof(TrackSummary, seq) / 4)
// term of word i (32 bits) in TrackSummary::check; the sum of terms is order-independent, so the device adds them as it stores the fields
__host__ __device__ static inline unsigned lsd_summary_term(unsigned i, unsigned bits) { return bits * (2u * i + 1u); }

// A result the host needs eventually but not now (mean inverse depth / point count of a setDepth, the rescale factor
// of a createKeyFrame): written by the device into a pinned slot, read by the host at the first later point that
// synchronises anyway (or when somebody asks for the value).
struct DeferredSlot { double sum, count, flag, pad; };
#define LSD_NUM_SLOTS 256
// Defaults that include/lsdhip.h states in prose: single source here, exported by lsdhip_build_defaults, compared with the header's text
// by tests/test_abi_cpu.py::test_header_prose_states_the_built_defaults.
#define LSD_DEFAULT_ASYNC 0                 // lsdhip_ctx_set_async
#define LSD_DEFAULT_PIPELINE 0              // lsdhip_ctx_set_pipeline
#define LSD_SPEC_TRIALS_SMALL 6             // automatic speculation policy: trials per step on levels of up to LSD_SPEC_SMALL_PX pixels
#define LSD_SPEC_SMALL_PX 6144
#define LSD_SPEC_TRIALS_MID 5               // ... up to LSD_SPEC_MID_PX pixels; larger levels: one evaluation per step
#define LSD_SPEC_MID_PX 90112
#define LSD_SPEC_CAP_WORKGROUPS 80          // workgroups per trial on speculating levels above LSD_SPEC_CAP_ABOVE_PX pixels
#define LSD_SPEC_CAP_ABOVE_PX 24576
#ifndef LSD_OBS_SPLIT_MIN_MAPS
#define LSD_OBS_SPLIT_MIN_MAPS 4             // lsdhip_depth_update_batch: select + walk launches from this many maps on
#endif
#define LSD_OBS_WALK_WAVES 4096             // one-wave workgroups of the walk launch
#define LSD_BATCH_THROUGHPUT_MIN_JOBS 8     // lsdhip_tracker_track_batch: throughput mode from this many jobs on
#define LSD_BATCH_SPEC_MAX 4                // reject-chain speculation of batches in throughput mode: most trials per step (levels without a mask)
#define LSD_BATCH_SPEC_PIXELS 500000        // ... trials per step at a level = what keeps jobs x trials x pixels of the level within this (1 M until round 5: 64-job batches lost 12 % to it)
#define LSD_BATCH_STRIP_WORKGROUPS 768      // strips x jobs of a throughput-mode evaluation launch (3 workgroups per CU)
struct lsdhip_frame;

// Host-side state of a context (arena free list, deferred-result slot ring and its owner tables, profiling events, the
// staging blocks of the objects created on it) is shared by every Frame / SE3Tracker / DepthMap of that context, and the
// reference drives them from two threads (tracking and mapping, C/SlamSystem.h:124-131).  Every C-ABI entry that touches it
// holds the context's mutex (recursive: entries call each other, and the tracker's enqueue hook re-enters from the same
// thread); lsdhip_tracker_track releases it while it waits for the device, so the mapping thread can keep enqueueing.
#define LSD_CTX_LOCK(c) std::lock_guard<std::recursive_mutex> lsd_ctx_lock_((c)->mtx)
struct lsdhip_ctx {
  std::recursive_mutex mtx;
  int device = 0;
  bool async = LSD_DEFAULT_ASYNC != 0;                 // lsdhip_ctx_set_async: mapping calls return after enqueueing
  DeferredSlot* h_slots = nullptr;       // pinned, device-mapped ring
  lsdhip_frame* slot_stats_owner[LSD_NUM_SLOTS] = {};
  lsdhip_frame* slot_rescale_owner[LSD_NUM_SLOTS] = {};
  int slot_next = 0;
  // Which deferred results have landed without asking the stream: a tracking job whose completion the host has observed proves
  // that everything enqueued before it has completed.  enqEpoch counts the jobs enqueued, doneEpoch is the newest one seen done,
  // slot_epoch[i] the value of enqEpoch when the slot's kernel was enqueued: the slot is complete once doneEpoch > slot_epoch[i].
  long long enqEpoch = 0, doneEpoch = 0;
  long long slot_epoch[LSD_NUM_SLOTS] = {};
  int w = 0, h = 0;
  bool refBlocksWanted = false;          // set by the first throughput-mode tracking batch: from then on every idepth pyramid is followed by k_ref_blocks
  int wl[LSD_LEVELS], hl[LSD_LEVELS];
  LevelIntr intr[LSD_LEVELS];
  float K0[9], K0inv[9];
  lsdhip_params params;
  hipStream_t stream = nullptr;
  // Pipelined operation (lsdhip_ctx_set_pipeline; the reference's tracking thread beside its mapping thread, C/SlamSystem.h:124-132):
  // SE3Tracker jobs run on `stream`, frame creation (upload + pyramids) and every DepthMap call on `mstream`.  The two are ordered by
  // events where mapping products reach the tracker: a tracking job waits for the M-sequence point its frame's pyramids / its
  // keyframe's PUBLISHED depth planes were recorded at (mstream is in-order, so a later point implies every earlier one; the events live
  // in a ring indexed by sequence number modulo LSD_EVR).  The other direction needs no device-side ordering: a mapping operation that
  // consumes a tracking job's results (its pose — through the host — and its refPixelWasGood mask) is queued by the host after it has
  // SEEN that job finish, and everything the job wrote for others was written by launches that completed before the finishing one
  // started (the mask of a speculative trial is merged into the frame's plane by a kernel on mstream, see lsdhip_tracker_track).
#define LSD_EVR 64
  bool pipeline = LSD_DEFAULT_PIPELINE != 0;
  hipStream_t mstream = nullptr;
  hipEvent_t mEv[LSD_EVR] = {};
  long long mSeq = 0;                    // record points on mstream so far
  long long mDoneSeq = 0;                // newest M-sequence the host knows to be complete
  long long tWaitedM = 0;                // newest M-sequence `stream` has been ordered behind
  long long slot_mseq[LSD_NUM_SLOTS] = {};   // M-sequence whose completion implies the slot's value has landed (pipelined contexts)
  // Pipelined contexts: merges of a speculative trial's refPixelWasGood plane into its frame's plane (k_mask_merge) that have not been
  // queued yet: the tracking call only notes them (nothing may delay the next tracking job's launches), the next mapping-stream
  // operation queues them first (lsd_m_begin)
  struct PendingMerge { uint8_t* plane; const uint8_t* side; long long* doneSeq; };
  std::vector<PendingMerge> pendingMerges;
  unsigned long long* d_sums = nullptr;  // LSDHIP_TRACE_SUMS: checksum slots
  std::vector<int> sums_meta;            // (kind, id) per slot; kind < 0: host value stored in sums_host
  std::vector<unsigned long long> sums_host;
  int* d_gate = nullptr;                 // developer hook LSDHIP_PIPE_GATE: the next tracking job's first launch releases the mapping work queued before it
  int gateSeq = 0, gateWaited = 0;
  // Second stream for the transport primitives of the multi-process loops (row copies, flags, device copies, the caller's RCCL
  // calls): between lsdhip_ctx_aux_begin and lsdhip_ctx_aux_end they are queued there, ordered behind what the main stream held at
  // `begin`; lsdhip_ctx_aux_join makes the main stream wait for them.  Created on first use.
  // lanes: independent DepthMap call chains of different maps side by side (lsdhip_ctx_lanes_begin / _lane_select / _lanes_end)
  static constexpr int MAX_LANES = 16;
  hipStream_t lanes[MAX_LANES] = {};
  hipEvent_t lane_done[MAX_LANES] = {};
  hipEvent_t lane_fork = nullptr;
  int lanes_open = 0;      // > 0 between lanes_begin and lanes_end: number of lanes
  int lane_cur = -1;       // lane the mapping calls currently go to (-1: the context's own stream)
  bool lane_used[MAX_LANES] = {};
  bool lane_record_pending = false;   // pipelined: a DepthMap call inside the open region asked for a record point
  hipStream_t aux_stream = nullptr;
  unsigned* d_flagArrive = nullptr;      // arrival counter of k_flag_set's workgroups
  hipEvent_t aux_fork = nullptr, aux_done = nullptr;
  bool aux_active = false, aux_pending = false;
  // profiling of the residual kernel (bench.py roofline leg)
  bool prof_on = false;
  unsigned prof_tick = 0;                // trackFrame jobs seen while profiling (every 8th is timed)
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  bool prof_pending = false;             // ev_a / ev_b of the last launch batch not yet read
  double prof_ms = 0, prof_bytes = 0;
  long long prof_launches = 0;
  std::vector<hipEvent_t> prof_events;   // per-launch event pairs for the device-resident LM loop
  // sampled event brackets around the shared launches of the batched entries (bench.py: extra_configs.multi_seq.*.roofline), while
  // prof_on: every LSD_BPROF_PERIOD-th call of a kind.  kinds: 0 frame pyramids (lsdhip_frame_create_batch), 1 observe, 2 fill holes +
  // regularise (+ setDepth), 3 idepth pyramids of lsdhip_depth_update_batch, 4 the keyframe change chain; units = map pixels processed
#define LSD_BPROF_KINDS 5
#define LSD_BPROF_SLOTS 16
#define LSD_BPROF_PERIOD 3
  struct BProfSlot { hipEvent_t a = nullptr, b = nullptr; int kind = -1; double units = 0; bool pending = false; } bprof[LSD_BPROF_SLOTS];
  int bprof_next = 0;
  unsigned bprof_tick[LSD_BPROF_KINDS] = {};
  double bprof_ms[LSD_BPROF_KINDS] = {}, bprof_units[LSD_BPROF_KINDS] = {};
  long long bprof_calls[LSD_BPROF_KINDS] = {};
  unsigned long long* d_obsBatchAcc = nullptr;   // [0] searches, [1 .. 64] walk steps (by wave), [65] counted launches — of the sampled walk launches
  // recycled frame arenas (the FrameMemory idea, C/DataStructures/FrameMemory.cpp:67-127, for device buffers);
  // reuse is stream-ordered, so no synchronisation is needed when a frame dies
  std::vector<void*> free_arenas;
  size_t arena_keep = 16;                // arenas of destroyed frames kept for reuse (grows with the batch width of lsdhip_frame_create_batch)
  float* d_gtStage = nullptr;                        // w x h floats: staging of lsdhip_frame_set_depth_gt
  // kernel-argument arrays of the batched launches (several sequences per launch): pinned staging slots and their device twins,
  // reused round-robin; a slot is rewritten only after the launches that read it have completed (lsd_args_push / lsd_args_release)
  struct ArgRing {
    static constexpr int NS = 32;
    uint8_t* h = nullptr; uint8_t* d = nullptr; size_t slotBytes = 0; int next = 0; hipEvent_t ev[NS] = {}; bool used[NS] = {};
    int cur = -1; size_t curBytes = 0;   // the slot being filled / last committed
  } args;
  std::vector<struct lsdhip_depthmap*> depthmaps;   // alive on this context: a destroyed frame is unhooked from them
  size_t arena_bytes = 0;
};

struct lsdhip_frame {
  lsdhip_ctx* ctx = nullptr;
  int id = 0;
  uint8_t* d_gray = nullptr;            // level-0 source (uint8)
  float* d_image[LSD_LEVELS] = {};      // float planes
  float4* d_grad[LSD_LEVELS] = {};      // (gx, gy, I, 0)
  float* d_absgrad = nullptr;           // level-0 |grad| (temp of buildMaxGradients)
  float* d_maxgrad = nullptr;           // level-0 maxGradients
  float* d_idepth[LSD_LEVELS] = {};     // the depth planes the TRACKING side reads (TrackingReference's view of the keyframe)
  float* d_idepthVar[LSD_LEVELS] = {};
  bool hasIDepth = false;
  bool level0Ready = false;             // d_grad[0], d_absgrad, d_maxgrad are built (keyframe planes, on demand: lsd_frames_require_level0)
  // gradient candidates of a keyframe (built with the level-0 planes): per group of 1024 consecutive pixels the in-group offsets (uint16,
  // pixel order) of the pixels DepthMap::observeDepthRow can ever search — inside the 3-pixel border, maxGradients >= minUseGrad
  // (DepthMap.cpp:111-131) — then one uint16 count per group.  The select pass of a batched update walks these lists instead of all pixels.
  uint16_t* d_gradCand = nullptr;
  float gradCandTh = -1.0f;             // the threshold the lists were built for (< 0: not built)
  // Pipelined contexts: Frame::setDepth on the mapping stream writes the second plane set while a tracking job may still read the
  // first; lsdhip_frame_publish_depth (= TrackingReference::importFrame) swaps them.  Non-pipelined contexts write d_idepth directly.
  float* d_idepthW[LSD_LEVELS] = {};
  float* d_idepthVarW[LSD_LEVELS] = {};
  bool depthPending = false;            // the W set holds a setDepth result that has not been published yet
  long long depthPendingSeq = 0;        // ... M-sequence it is complete at
  long long depthSeq = 0;               // M-sequence the published planes are complete at (0: written synchronously)
  long long readySeq = 0;               // M-sequence the image pyramids are complete at (0: built on `stream` / synchronously)
  unsigned depthVersion = 0;            // incremented whenever the idepth / idepthVar pyramids are rewritten (setDepth)
  // reference blocks of levels >= 1, written behind every idepth pyramid (k_ref_blocks) and published with it: what the strips of a
  // throughput-mode tracking batch read instead of scanning the level's validity (TrackLevel::kf_refBlk)
  uint8_t* d_refBlk[LSD_LEVELS] = {};
  uint8_t* d_refBlkW[LSD_LEVELS] = {};
  bool refBlkValid = false, refBlkValidW = false;   // the set holds the blocks of its depth planes (contexts that have not run a throughput-mode
                                                    // batch yet do not build them: lsdhip_ctx::refBlocksWanted, lsd_frames_require_ref_blocks)
  uint8_t* d_wasGood = nullptr;         // level-1 mask (lazily created, 0xFF)
  bool wasGoodValid = false;
  bool wasGoodPristine = false;         // the mask still holds the 0xFF fill of frame creation
  // pose-tree node
  lsdm::Sim3dH thisToParent_raw;
  lsdhip_frame* trackingParent = nullptr;
  int trackingParentID = -1;
  float initialTrackedResidual = 0;
  int numFramesTrackedOnThis = 0, numMappedOnThis = 0, numMappedOnThisTotal = 0;
  float meanIdepth = 1;
  int numPoints = 0;
  int pendStats = -1;                   // slot index of a setDepth whose (sum, count) the host has not read yet
  int pendRescale = -1;                 // slot index of a createKeyFrame whose rescale factor the host has not read yet
  bool depthHasBeenUpdatedFlag = false;
  // re-activation data (Frame::takeReActivationData)
  float* d_idepth_reAct = nullptr;
  float* d_idepthVar_reAct = nullptr;
  uint8_t* d_validity_reAct = nullptr;
  bool reActValid = false;
};

struct lsdhip_tracker {
  lsdhip_ctx* ctx = nullptr;
  // DenseDepthTrackerSettings (C/util/settings.h:355-402)
  float lambdaSuccessFac = 0.5f, lambdaFailFac = 2.0f;
  float lambdaInitial[LSD_LEVELS], stepSizeMin[LSD_LEVELS], convergenceEps[LSD_LEVELS];
  int maxItsPerLvl[LSD_LEVELS];
  float lambdaInitialTestTrack = 0, stepSizeMinTestTrack = 1e-3, convergenceEpsTestTrack = 0.98, maxItsTestTrack = 5;
  float huber_d = 3, var_weight = 1.0;
  // results (public members of SE3Tracker)
  float pointUsage = 0, lastGoodCount = 0, lastMeanRes = 0, lastBadCount = 0, lastResidual = 0;
  float affineEstimation_a = 1, affineEstimation_b = 0, affineEstimation_a_lastIt = 1, affineEstimation_b_lastIt = 0;
  bool diverged = false, trackingWasGood = false;
  int numEvaluations = 0, numWarpUpdates = 0;
  void (*enqueueHook)(void*) = nullptr;   // lsdhip_tracker_set_enqueue_hook
  void* enqueueHookUser = nullptr;
  bool spinWait = true;           // poll the pinned summary instead of hipStreamSynchronize (LSDHIP_SPIN=0 disables)
  bool hostLM = false;            // debugging: run the LM control loop on the host, one evaluation per round trip
  // device scratch
  float* d_partials = nullptr;    // TrackScratch arena (sums | topkey | topval), see tracker.hip
  int max_blocks = 0;
  TrackState* d_state = nullptr;  // [2], double-buffered by launch parity
  int levelEvaluations[LSD_LEVELS] = {};   // evaluations of the last job per pyramid level
  int block = 256;                // workgroup size of k_track_step (LSDHIP_TRACK_BLOCK)
  int grid_cap = 304;             // most workgroups one evaluation uses (LSDHIP_TRACK_CAP); larger levels grid-stride
  int batch_jobs = 0;             // > 1 while the jobs of a batch are being described
  int cap_override = 0;           // batch tracking: per-job workgroup cap while the jobs of a batch are being described
  int recent[4] = {0, 0, 0, 0};   // evaluating launches of the last jobs: size the launch budget of the next one
  int specC = 6;                  // most trials per launch (LSDHIP_SPEC; 1 = no speculation)
  int soloMinJobs = -1;           // batches of at least this many jobs walk their coarse levels in one workgroup per job (k_track_solo); 0: never; -1: build default
  bool specAuto = true;           // trials per level from the level's size (see track_device); false after set_speculation
  int specCaps[LSD_LEVELS] = {0, 0, 0, 0, 0};    // per-level workgroups per trial (LSDHIP_SPEC_CAPS; 0 = automatic)
  int specLevel[LSD_LEVELS] = {0, 0, 0, 0, 0};   // per-level trials (LSDHIP_SPEC_LEVELS = "l0,l1,l2,l3,l4"; 0 = automatic / specC)
  int specCap = LSD_SPEC_CAP_WORKGROUPS;              // workgroups per trial on levels of 24 K - 88 K pixels when speculating (0 = grid_cap / 2)
  uint8_t* d_maskSide = nullptr;  // 2 sets (alternating by job) of (SPEC_MAX - 1) mask planes of (w >> 1) x (h >> 1) bytes
  int maskSet = 0;                // set the job being launched writes
  long long maskMergeSeq[2] = {0, 0};   // pipelined contexts: M-sequence behind which the merge that reads set s has completed
  size_t maskStride = 0;
  int numLaunches = 0;
  // polled summaries (lsdhip_tracker_summary_stats): jobs polled, jobs whose record did not add up when `done` arrived, longest wait for the
  // rest of the record (ns), words of the record that were still stale at the first look (all jobs)
  long long sumPolled = 0, sumLate = 0, sumLateMaxNs = 0, sumLateWords = 0;
  int sumLateFirstWord = -1, sumLateLastWord = -1;   // lowest / highest word index ever seen stale
  unsigned long long dbgCum = 0;
  std::vector<unsigned> dumpL0;   // developer dump (LSDHIP_DUMP_L0): what the job's first launch left in the scratch
  unsigned long long* d_dbg = nullptr;
#ifdef LSD_DEVTOOLS
  int* d_log = nullptr;      // launch log of the current job (TrackSpec::dbgLog), 4096 x 16 ints
  std::vector<int> lastLog;  // ... of the last finished job, read by lsdhip_tracker_debug_log
#endif
  TrackSpec spec = {};             // of the job being launched
  const lsdhip_frame* jobKf = nullptr;   // keyframe whose planes the job being run reads (trackFrame jobs), and their version at its start
  unsigned jobKfVersion = 0;
  int jobTag = 0, launchOrdinal = 0;   // progress tag of the launch chain (TrackSummary::seq)
  int budgetExtra = 2;                 // launches queued beyond the most the recent jobs needed (finishing step + margin; LSDHIP_BUDGET_EXTRA)
  int budgetFixed = 0;   // LSDHIP_BUDGET_FIXED (test hook): launches per budget, however many the job needs
  long long dbgJobs = 0, dbgEnqueued = 0, dbgMisses = 0, dbgWaitNs = 0, dbgLaunchNs = 0;   // LSDHIP_TRACK_DEBUG=1: printed at destroy
  TrackSummary* h_summary = nullptr;  // pinned, device-mapped
  TrackSummary* d_summary = nullptr;  // device alias of h_summary
  unsigned long long* d_trace = nullptr;  // LSD_PHASE_TRACE developer build only
  // batch tracking (lsdhip_tracker_track_batch): per-job descriptions, states, scratch and summaries
  int batch_capacity = 0;
  TrackJob* d_bjobs = nullptr;
  TrackJob* h_bjobs = nullptr;        // pinned staging
  TrackState* d_bstate = nullptr;     // [capacity][2]
  float* d_bscratch = nullptr;        // TrackScratch arena x capacity (sums | topkey | topval | recs, each [job][parity][trial][...])
  int batchRecent[4] = {0, 0, 0, 0};  // rounds the last batches needed: size the launch budget of the next one
  int batchTag = 1;                   // what `done` of a polled batch's summaries is raised to (>= 2; 1 = a batch the host synchronises for)
  TrackSummary* h_bsummary = nullptr; // pinned, device-mapped
  float* d_pts = nullptr;         // permaref point upload
  int pts_capacity = 0;
};

// SoA hypothesis planes in HBM (29 B/px + one spare validity plane for snapshot semantics)
// A pointer read from a job description in device memory is a generic pointer to the compiler, and accesses through it are FLAT
// instructions: 64-bit vector addresses only, and counted against the LDS counter as well as the memory counter, so every wait for an
// LDS read also drains the loads in flight.  The pointer fields of the descriptions that kernels read from memory (batches) are
// therefore typed as global-address-space pointers in device code (LSD_G; same layout on the host, where it is empty): whatever is
// reached through them is a global_load / global_store (also through the plain-pointer parameters of inlined functions: the compiler
// still knows where the pointer came from).
#ifdef __HIP_DEVICE_COMPILE__
#define LSD_G __attribute__((address_space(1)))
#else
#define LSD_G
#endif
template <typename T> __host__ __device__ inline LSD_G T* lsd_g(T* p) { return (LSD_G T*)p; }    // what the host code fills such a field with
struct HypPlanes {
  LSD_G uint8_t* valid;       // isValid
  LSD_G int32_t* blacklisted;
  LSD_G float* nextID;        // nextStereoFrameMinID
  LSD_G int32_t* validity;    // validity_counter
  LSD_G float* idepth;
  LSD_G float* var;
  LSD_G float* idepth_s;
  LSD_G float* var_s;
};

struct StereoRef {      // one reference (tracked) frame as K4 sees it
  LSD_G const float* image;   // level-0 image plane
  LSD_G const uint8_t* wasGood;  // level-1 mask or nullptr (only consulted when parentIsKF)
  int parentIsKF;
  int id;
  float initialTrackedResidual;
  float K_otherToThis_R[9];
  float K_otherToThis_t[3];
  float otherToThis_t[3];
  float thisToOther_t[3];
  float row0[3], row1[3], row2[3];
};

struct lsdhip_depthmap {
  lsdhip_ctx* ctx = nullptr;
  HypPlanes cur;         // currentDepthMap
  HypPlanes oth;         // otherDepthMap (propagation target; swapped)
  uint8_t* d_validSnap = nullptr;  // spare validity plane (ping-pong target of K5/K6)
  void* bases[3] = {nullptr, nullptr, nullptr};  // arena base pointers (for hipFree)
  lsdhip_frame* activeKeyFrame = nullptr;
  bool activeKeyFrameIsReactivated = false;
  StereoRef* d_refs = nullptr;     // views into d_stage
  int* d_refByID = nullptr;
  char* h_stage = nullptr;         // pinned staging block (refs | refByID) and its device twin
  char* d_stage = nullptr;
  size_t stage_bytes = 0;
  // K7 scratch
  int* d_slotCount = nullptr;      // per target
  int* d_slots = nullptr;          // per target x capacity source indices
  int* d_ovfHead = nullptr;        // per target: head of the chain of sources beyond the capacity (-1: none)
  int2* d_ovf = nullptr;           // chain entries (source index, next entry), w*h of them
  float4* d_cand = nullptr;        // per source candidate (new_idepth, new_var, validity as float bits, target)
  int* d_flags = nullptr;          // overflow flag etc.
  double* d_red = nullptr;         // reduction scratch (sum, count)
  size_t redStride = 0;            // doubles per set of partials (three sets behind the first 16 doubles)
  bool propClean = false;          // slot counts / chain heads / flags of the K7 scratch are in their rest state
  // A hypothesis on a pixel below the gradient threshold can only come from outside the update loop (ground-truth / random initialisation,
  // an upload, re-activation data): propagateDepth and the hole filling create none (DepthMap.cpp:551, :565, :668), and the first
  // observe pass drops them (:125).  While this is set the select pass of a batched update scans every pixel; afterwards the keyframe's
  // gradient candidates.
  bool lowGradHypPossible = true;
  double* h_red = nullptr;         // pinned
  // GPU-side timing of the mapping calls (events on the context's stream; read back lazily)
  hipEvent_t ev[8][2] = {};
  int ev_kind[8] = {};
  bool ev_pending[8] = {};
  int ev_next = 0;
  unsigned long long* d_obs_trace = nullptr;   // LSD_PHASE_TRACE developer build only
  size_t obs_trace_words = 0;
  double gpu_ms[4] = {0, 0, 0, 0};    // update, createKeyFrame, finalizeKeyFrame, k_observe alone (sampled while profiling)
  long long gpu_calls[4] = {0, 0, 0, 0};
  unsigned obs_tick = 0;
  bool countNext = false;                        // the next k_observe launch counts its searches / walk steps (sampled while profiling)
  unsigned long long* d_obsCounters = nullptr;   // per wave (searches, steps) of the last counted launch | d_obsAcc: totals (searches, steps, launches)
  unsigned long long* d_obsAcc = nullptr;
  int obsCounterWaves = 0;
  float4* d_obsQueue = nullptr;                  // w x h entries: search queue of the two-launch observe of batches (created on first use)
  unsigned ev_tick[3] = {0, 0, 0};   // calls per kind: every 8th updateKeyframe / 2nd createKeyFrame, finalizeKeyFrame is timed
  float msUpdate = 0, msCreate = 0, msFinalize = 0, msObserve = 0, msRegularize = 0, msPropagate = 0, msFillHoles = 0,
        msSetDepth = 0;
};

// kernels / launchers implemented in the .hip files
inline hipStream_t lsd_transport_stream(lsdhip_ctx* c) { return c->aux_active ? c->aux_stream : c->stream; }
// the stream frame creation and the DepthMap calls run on
inline hipStream_t lsd_map_stream(lsdhip_ctx* c) { return c->lane_cur >= 0 ? c->lanes[c->lane_cur] : (c->pipeline ? c->mstream : c->stream); }
int lsd_m_begin(lsdhip_ctx* c);                    // an mstream operation starts (developer switches only: see the note on tracking -> mapping ordering in frame.hip)
long long lsd_m_record(lsdhip_ctx* c);             // record point on mstream -> its M-sequence (0 when the context is not pipelined, < 0: error)
int lsd_t_wait_m(lsdhip_ctx* c, long long seq);    // order `stream` behind M-sequence `seq`
bool lsd_m_done(lsdhip_ctx* c, long long seq);     // has mstream passed M-sequence `seq`?  (never blocks)
int lsd_sync_all(lsdhip_ctx* c);                   // both streams drained
// Developer instrumentation of the pipeline bug hunt of round 4 (profiles/r04_notes.md §1a) — compiled only into the LSD_DEVTOOLS build
// (lsd_slam_amd/build.py build_variant("devtools", ["LSD_DEVTOOLS"]), loaded through LSDHIP_LIB); the default library carries none of it:
//   LSDHIP_PIPE_GATE=1      mapping stream holds until the next tracking job starts (lsd_gate_wait / lsd_gate_open)
//   LSDHIP_PIPE_DUMMY=<k>   unrelated kernels on the mapping stream beside a tracking job (lsd_pipe_dummy)
//   LSDHIP_TRACE_SUMS=<f>   order-independent checksums of device buffers, queued on a stream at chosen points of the loop and written
//                           out when the context is destroyed — two runs of the same loop are compared entry by entry (tools/trace_cmp.py)
//   LSDHIP_TRACK_REPLAY / LSDHIP_DUMP_L0 / LSDHIP_TRACE_INPUTS / LSDHIP_TRACK_DEBUG / LSDHIP_HOST_TRACE   (tracker.hip, frame.hip)
#ifdef LSD_DEVTOOLS
int lsd_gate_wait(lsdhip_ctx* c);
int lsd_gate_open(lsdhip_ctx* c);
int lsd_pipe_dummy(lsdhip_ctx* c);
void lsd_trace_sum(lsdhip_ctx* c, hipStream_t s, int kind, int id, const void* p, size_t bytes);
void lsd_trace_val(lsdhip_ctx* c, int kind, int id, unsigned long long v);
#else
inline int lsd_gate_wait(lsdhip_ctx*) { return LSDHIP_OK; }
inline int lsd_gate_open(lsdhip_ctx*) { return LSDHIP_OK; }
inline int lsd_pipe_dummy(lsdhip_ctx*) { return LSDHIP_OK; }
inline void lsd_trace_sum(lsdhip_ctx*, hipStream_t, int, int, const void*, size_t) {}
inline void lsd_trace_val(lsdhip_ctx*, int, int, unsigned long long) {}
#endif
// depth planes a Frame::setDepth writes / the most recently written ones (== d_idepth on non-pipelined contexts)
inline float** lsd_depth_w(lsdhip_frame* f) { return f->ctx->pipeline ? f->d_idepthW : f->d_idepth; }
inline float** lsd_depthvar_w(lsdhip_frame* f) { return f->ctx->pipeline ? f->d_idepthVarW : f->d_idepthVar; }
inline uint8_t** lsd_refblk_w(lsdhip_frame* f) { return f->ctx->pipeline ? f->d_refBlkW : f->d_refBlk; }
inline int lsd_gradcand_groups(int pixels) { return (pixels + 1023) >> 10; }
inline size_t lsd_gradcand_bytes(int pixels) { return ((size_t)lsd_gradcand_groups(pixels) * 1024 + (size_t)lsd_gradcand_groups(pixels)) * 2; }
inline int lsd_refblk_blocks(int pixels) { return (pixels + 255) >> 8; }
inline size_t lsd_refblk_bytes(int pixels) { return (size_t)lsd_refblk_blocks(pixels) * (256 + 4); }
inline float** lsd_depth_latest(lsdhip_frame* f) { return f->depthPending ? f->d_idepthW : f->d_idepth; }
inline float** lsd_depthvar_latest(lsdhip_frame* f) { return f->depthPending ? f->d_idepthVarW : f->d_idepthVar; }
int lsd_frame_publish_depth(lsdhip_frame* f);
int lsd_flush_merges(lsdhip_ctx* c);               // tracker.hip: queue the pending mask merges on the mapping stream
// A job on the tracking stream of a pipelined context: entered behind everything the mapping stream holds (lsdhip_tracker_track
// needs less and says so itself); left with the tracking stream DRAINED, because mapping-stream operations are ordered behind tracking
// work by the host having seen it complete (see frame.hip).  lsdhip_tracker_track, the one hot entry, does not drain: it leaves only
// launches behind that touch nothing but the tracker's own state.
struct LsdTrackJobScope {
  lsdhip_ctx* c;
  int rc = LSDHIP_OK;
  bool drain;
  LsdTrackJobScope(lsdhip_ctx* c_, bool waitAll) : c(c_), drain(waitAll) { if (c->pipeline && waitAll) rc = lsd_t_wait_m(c, c->mSeq); }
  ~LsdTrackJobScope() { if (c->pipeline && drain) (void)hipStreamSynchronize(c->stream); }
};
int lsd_frame_build_pyramids(lsdhip_frame* f, const uint8_t* src, hipStream_t stream);
int lsd_frames_require_ref_blocks(lsdhip_frame** kfs, int n, hipStream_t stream);   // the published planes' reference blocks, where missing (frame.hip)
int lsd_frames_require_level0(lsdhip_frame** fs, int n);    // Frame::gradients(0) / maxGradients(0) on demand, on lsd_map_stream (frame.hip)
int lsd_frame_require_level0(lsdhip_frame* f);
int lsd_frame_require_level0_for_tracking(lsdhip_frame* f);
// copies `bytes` of kernel-argument records to the device on `s` (stream-ordered) and returns the device address; lsd_args_release after
// the launches that read them (frame.hip).  begin / commit: the same in two steps, for records that hold their own device address
int lsd_args_push(lsdhip_ctx* c, const void* src, size_t bytes, hipStream_t s, void** dev_out);
int lsd_args_begin(lsdhip_ctx* c, size_t bytes, void** host_out, void** dev_out);
int lsd_args_commit(lsdhip_ctx* c, hipStream_t s);
int lsd_args_release(lsdhip_ctx* c, const void* dev, hipStream_t s);
// Frame::setDepth's second half for n keyframes in one launch (lsdhip_depth_update_batch)
int lsd_frame_build_idepth_pyramid_batch(lsdhip_frame** f, int n, const double* const* redPartials, int redN, double* const* redOut,
                                         const int* redNs = nullptr);   // redNs: partial counts per frame (else redN for all)
int lsd_frame_build_idepth_pyramid(lsdhip_frame* f, const double* redPartials = nullptr, int redN = 0, double* redOut = nullptr);   // on lsd_map_stream, into lsd_depth_w
int lsd_frame_ensure_depth_planes(lsdhip_frame* f);
int lsd_frame_ensure_wasgood(lsdhip_frame* f);
void lsd_depthmaps_forget_frame(lsdhip_ctx* c, lsdhip_frame* f);   // depthmap.hip: unhook a frame that is being destroyed
int lsd_prof_collect(lsdhip_ctx* c);
int lsd_bprof_begin(lsdhip_ctx* c, int kind, hipStream_t s);               // slot (>= 0) when this call is sampled, -1 otherwise, < -1: error
int lsd_bprof_end(lsdhip_ctx* c, int slot, hipStream_t s, double units);
int lsd_frame_resolve(lsdhip_frame* f);        // reads the frame's deferred results (synchronises the stream if any)
int lsd_ctx_take_slot(lsdhip_ctx* c);          // next slot of the ring (resolving whoever still waits on it)
