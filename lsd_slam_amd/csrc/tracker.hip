// SE3Tracker on the device.  gfx950 only.
//
// One evaluation = one launch of k_track_step on the context's stream:
//   (1) every workgroup finishes the *previous* evaluation: fixed-order sum of the partial rows, SSE tail-drop
//       correction, and the Levenberg-Marquardt decision of SE3Tracker::trackFrame (accept / reject, lambda schedule,
//       6x6 LDL^T solve, SE3 exp, level change) — redundantly and identically in every workgroup, so the launch needs no
//       atomics, fences or inter-workgroup waits and the pose never leaves HBM between evaluations;
//   (2) fused K0 (point generation from the keyframe planes) + K1 (warp / bilinear sample / mask) + K2 (weights) + K3
//       (normal equations) at the pose (1) produced; 41 sums reduced wave (DPP) -> workgroup (LDS) -> partials[tile].
// The host enqueues a budget of k_track_step launches back to back and synchronises once; steps launched
// after the job has finished return immediately.  A host-driven LM loop (one evaluation per round trip) is kept behind
// lsdhip_tracker.hostLM for debugging and as the kernel-level parity hook (lsdhip_tracker_evaluate).
//
// Reference behaviour restated:
//   TrackingReference::makePointCloud   C/Tracking/TrackingReference.cpp:128-138
//   SE3Tracker::calcResidualAndBuffers  C/Tracking/SE3Tracker.cpp:885-1029
//   SE3Tracker::calcWeightsAndResidualSSE  :492-575   (op order of the SSE path; _mm_rcp_ps -> IEEE 1/x)
//   SE3Tracker::calculateWarpUpdateSSE  :1033-1130 + LGS6::updateSSE / finish  C/Tracking/LGSX.h:205-386
//   SE3Tracker::trackFrame              :280-486
//   SE3Tracker::trackFrameOnPermaref    :162-272, checkPermaRefOverlap :121-157
//
// Quirks kept on purpose: the SSE loops ignore the last size%4 in-image points (in the reference's
// x-outer point order) for K2/K3 — emulated in k_lm_step; LGS6::updateSSE counts 6 constraints per group of 4;
// `LM_lambda <= 0.2` compares a float with a double.
//
// Data layout: keyframe planes idepth/idepthVar/image (3 x 4 B per pixel, row-major, coalesced per wave), tracked-frame
// texels float4 (gx, gy, I, 0) so that one bilinear tap is one 16-byte load.  Algorithmic bytes per evaluation at level l
// (SURVEY.md §8(d)): 20 N_l + [l==1] 5 N_l + 12 min(w_l h_l, 4 N_l).
#include <atomic>
#include <chrono>
#include "track_device.hpp"


#ifdef LSD_PHASE_TRACE
#define PHASE_MARK(k) do { if (trOn_ && threadIdx.x == 0) tr_[k] = clock64(); } while (0)
#define LSD_TRACE_WORDS 32
#else
#define PHASE_MARK(k) do { } while (0)
#endif

#ifdef LSD_DEVTOOLS
// one record per launch, written by workgroup 0: which exit the launch took and the state it saw / left
#define LAUNCH_LOG(exitCode, pendingIn, ncandIn, pcUsed) do { \
    if (!BATCH && MODE == TS_FUSED && blockIdx.x == 0 && threadIdx.x == 0 && spec.dbgLog && spec.seq != 0) { \
      int* lg_ = spec.dbgLog + ((spec.seq & 0xFFF) * 16); \
      lg_[0] = spec.seq; lg_[1] = (exitCode); lg_[2] = parity | (first << 1) | (spec.last << 2); lg_[3] = lvlIn_; lg_[4] = (pendingIn); lg_[5] = (ncandIn); \
      lg_[6] = (pcUsed); lg_[7] = S.level; lg_[8] = S.numEvaluations; lg_[9] = S.numLaunches; lg_[10] = S.done; lg_[11] = S.phase; lg_[12] = S.incTry; \
      lg_[13] = S.ncand; lg_[14] = S.pending; lg_[15] = numLaunchesIn_; } } while (0)
#else
#define LAUNCH_LOG(exitCode, pendingIn, ncandIn, pcUsed) do { } while (0)
#endif

// Fused tracking step, one launch per step of the LM loop:
//   (1) every workgroup finishes the *previous* launch's evaluation(s) from the tiles' partial rows: fixed-order column sums
//       (row-major rows, whole rows per wave load, all loads issued before the first add), SSE tail drop, LGS6::finish, then
//       the LM decision in wave 0 — identical inputs, identical code, hence identical state in every workgroup without any
//       inter-workgroup communication inside the launch (no atomics, no fences, run-to-run deterministic).  The waves fetch
//       (or, on multi-pass levels, re-evaluate) the at most 3 tail points of every pending trial while the sums travel;
//   (2) the residual evaluation (K0+K1+K2+K3) of the pose(s) that decision produced, grid-stride over the level's pixels,
//       41 sums reduced lane -> workgroup (LDS, transposed) -> sums[.][trial][tile].
// Reject-chain speculation (single jobs, TS_FUSED; TrackSpec): the LM loop's retries after a rejection depend only on A, b
// and lambda, so (2) evaluates the next `trials` of them side by side — workgroup = (trial, tile); each trial group derives its
// lambda by the closed-form recurrence and solves for its own increment — and (1) of the next launch picks, lane-parallel, the
// first pending trial at which the reference's loop stops (diverged / accepted / step below stepSizeMin), advances lambda,
// incTry and the counters past the plain rejections before it, and runs ONE LM step on that trial's totals.  Same decisions,
// same evaluation counts, same refPixelWasGood as one evaluation per launch, in about half the dependent launches.
// The first launch of a job (first = 1) builds the initial state from the job instead of loading it.
// BATCH: blockIdx.y selects one of several independent jobs (tracking a batch of frames / permanent references in the
// same launches): the job descriptions then live in HBM (`jobs`), and state / scratch / summary are arrays over jobs.
// MODE (batches in throughput mode split a step into two launches, so that the LM replay is paid once per job instead of
// once per workgroup): TS_FUSED = finish + evaluate as described above; TS_LM = finish the pending evaluation and publish the
// state (grid.x = 1); TS_EVAL = evaluate the published state (reads st2[parity], writes scratch[parity]; no state change).
enum { TS_FUSED = 0, TS_LM = 1, TS_EVAL = 2 };
// the state a job's first launch starts from (SE3Tracker.cpp:280-320: the initial estimate, affine parameters of the settings, the top level)
__device__ __forceinline__ void track_state_begin(TrackState& S, const TrackJob& job, const int tid) {
  if (tid == 0) {
    S.T = job.T0;
    set_eval_pose(S, job.T0);
    S.aff_a = job.aff_a0; S.aff_b = job.aff_b0; S.aff_a_lastIt = job.aff_a0; S.aff_b_lastIt = job.aff_b0;
    S.lastErr = 0; S.LM_lambda = 0; S.last_residual = 0;
    S.level = job.topLevel; S.iteration = 0; S.incTry = 0; S.phase = 0; S.pending = 0;
    S.done = 0; S.diverged = 0; S.numEvaluations = 0; S.numWarpUpdates = 0;
    S.pointUsage = 0; S.goodCount = 0; S.badCount = 0; S.meanRes = 0;
    S.bytes = 0;
    for (int l = 0; l < LSD_LEVELS; l++) S.levelEvals[l] = 0;
    S.ncand = 1; S.lastCand = 0; S.numLaunches = 0;
  }
  if (tid < 36) S.A[tid] = 0;
  if (tid < 6) { S.b[tid] = 0; S.inc[tid] = 0; }
}
// one point's K2 / K3 contributions as the 29-float row the tail drop subtracts: werr | the 21 upper-triangular J J^T w | the 6 J r w | r^2 w
__device__ __forceinline__ void tail_row(const PointOut& o, float* row) {
  row[0] = o.werr;
  int k = 1;
#pragma unroll
  for (int rr = 0; rr < 6; rr++) {
    const float Jw = o.J[rr] * o.w;
#pragma unroll
    for (int cc = rr; cc < 6; cc++) row[k++] = Jw * o.J[cc];
  }
  const float resw = o.res * o.w;
#pragma unroll
  for (int rr = 0; rr < 6; rr++) row[k++] = resw * o.J[rr];
  row[k++] = resw * o.res;
}

template <int BLOCK, bool BATCH, int MODE>
__device__ __forceinline__ void track_step_impl(const TrackJob& jobv, const TrackJob* __restrict__ jobs, TrackState* __restrict__ st2,
                                                TrackScratch sc, TrackSummary* __restrict__ out, int parity, int first, const TrackSpec& spec) {
  const TrackJob& job = BATCH ? jobs[blockIdx.y] : jobv;
  // The job description travels in the kernel arguments, which the host has just written: the first touch of each of its
  // cache lines misses down to HBM (~1 us), and the fields of lv[level] are addressed only once the level is known, i.e. in
  // the middle of the dependent chain.  Touch every line now, together with the loads of the prologue.
  unsigned warm = 0;
  if (!BATCH) {
    typedef const unsigned __attribute__((address_space(4))) cuint;
    cuint* ka = (cuint*)__builtin_amdgcn_kernarg_segment_ptr();
#pragma unroll
    for (int i = 0; i < (int)(sizeof(TrackJob) / 4); i += 32) warm ^= ka[i];
    asm volatile("" ::"s"(warm));   // the loads complete here, in the shadow of the prologue's own argument loads
  }
  if (BATCH) {
    const size_t j = blockIdx.y, rows = (size_t)sc.max_rows, cm = (size_t)sc.cmax;
    st2 += 2 * j;
    out += j;
    sc.sums += j * 2 * cm * RS_COLS * rows;
    sc.topkey += j * 2 * cm * rows;
    sc.topval += j * 2 * cm * rows * 96;
    if (sc.recs) sc.recs += j * 2 * cm * 32;
  }
  constexpr int WAVES = BLOCK / 64;
  constexpr int SUMW = WAVES - 1;                    // waves that sum partials (the last one does the tail work)
  constexpr int NSLICE = (SUMW * 64) / RS_END;       // row slices per column
  constexpr int QMAX = 20;                           // float4 loads per thread: rows per slice <= 80
  __shared__ TrackState S;
  __shared__ LmShared sh;
  __shared__ LmPar s_par;
  // the workgroup reduction handles the columns in NPASS batches (2 halves the LDS; measured slower — also for the throughput-mode
  // evaluation, where it would let four workgroups share a CU: profiles/r05_notes.md)
  constexpr int NPASS = 1;
  constexpr int CPP = (RS_END + NPASS - 1) / NPASS;      // columns per batch
  constexpr int RSLICE_ = BLOCK / CPP;
  __shared__ float s_sum[(NSLICE > RSLICE_ ? NSLICE : RSLICE_)][64];
  __shared__ __attribute__((aligned(16))) float s_red[CPP * (BLOCK + 1) + 8];
  __shared__ int s_wtop[WAVES][3];
  __shared__ int s_top[3];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int max_rows = sc.max_rows;
  // reject-chain speculation (single jobs): workgroup = (trial `cand`, tile `bx`) of the level of the pending evaluation —
  // fixed below, once the state says which level that is
  int cand = 0, bx = (int)blockIdx.x;
  const bool leader = blockIdx.x == 0;
  const int cmax = sc.cmax;
  TrackState* next = st2 + (1 - parity);
  const int outp = MODE == TS_EVAL ? parity : 1 - parity;
  const float* sums_in0 = sc.sums + (size_t)parity * cmax * RS_COLS * max_rows;
  float* sums_out = sc.sums + (size_t)outp * cmax * RS_COLS * max_rows;              // + the trial's slot, below
  const int4* topkey_in0 = sc.topkey + (size_t)parity * cmax * max_rows;
  int4* topkey_out = sc.topkey + (size_t)outp * cmax * max_rows;
  const float* topval_in0 = sc.topval + (size_t)parity * cmax * max_rows * 96;
  float* topval_out = sc.topval + (size_t)outp * cmax * max_rows * 96;
  constexpr int NSLOT = 16;                                  // row slots of the column sums: slot s adds rows s, s + 16, ...
  // trials a launch can finish: single jobs up to LSD_SPEC_MAX; batches in throughput mode up to LSD_BATCH_SPEC_MAX (the LM launch has one
  // workgroup per (trial, job): each proposes its own retry, as the trial groups of a single job's launch do); small batches one
  constexpr int TRIALS_MAX = BATCH ? (MODE == TS_EVAL ? 1 : LSD_BATCH_SPEC_MAX) : LSD_SPEC_MAX;
  constexpr bool SPEC_LM = MODE != TS_EVAL;
  // The finishing phase's per-trial tables live in the reduction buffer s_red: neither the strip list nor the reduction is live before the
  // barrier that ends the LM step, and nothing of these tables is read after it — so a launch that can finish four (batches) or six
  // (single jobs) trials needs no more LDS than one that finishes one (three workgroups per CU stay three).
  static_assert(TRIALS_MAX * (NSLOT * RS_COLS + 96 + 1) <= CPP * (BLOCK + 1) + 8, "the finishing phase's tables must fit the reduction buffer");
  float (*const s_sumT)[NSLOT][RS_COLS] = (float (*)[NSLOT][RS_COLS])s_red;                        // per pending trial: column sums by row slot
  float (*const s_subT)[3][32] = (float (*)[3][32])(s_red + TRIALS_MAX * NSLOT * RS_COLS);          // ... K2/K3 contributions of its (up to 3) tail points
  int* const s_nsubT = (int*)(s_red + TRIALS_MAX * (NSLOT * RS_COLS + 96));
  __shared__ float s_rec[TRIALS_MAX][32];                   // ... increment [0..5], pose [6..12], R [13..21], t [22..24] (trials > 0)
#ifdef LSD_PHASE_TRACE
  __shared__ unsigned long long* s_trp;
  unsigned long long* tr_ = sc.trace;
  // the traced workgroup: workgroup 0 (batches: workgroup LSDHIP_TRACE_WG of job 0; tools/phase_trace_batch.py)
  const bool trOn_ = sc.trace != nullptr && (int)blockIdx.x == (BATCH ? spec.traceWg : 0) && (!BATCH || blockIdx.y == 0);
  if (trOn_ && tid == 0) {
    unsigned long long n = sc.trace[0];
    sc.trace[0] = n + 1;
    tr_ = sc.trace + 1 + (n % 4096) * LSD_TRACE_WORDS;
    for (int k = 0; k < LSD_TRACE_WORDS; k++) tr_[k] = 0;
    tr_[8] = wall_clock64();
    tr_[24] = (unsigned long long)gridDim.x; tr_[25] = (unsigned long long)gridDim.y;
    s_trp = tr_;
  }
#endif
  PHASE_MARK(0);
  if (!BATCH && MODE == TS_FUSED && !first && !leader && st2[parity].done) return;   // launches queued behind the finishing one
  if (!BATCH && MODE == TS_FUSED && leader && tid == 0 && spec.seq != 0) out->seq = spec.seq;   // pinned host memory: fire and forget
  if (first) {
    track_state_begin(S, job, tid);
  } else {
    copy_words<sizeof(TrackState) / 4>(&S, st2 + parity, tid, BLOCK);
    if (SPEC_LM && cmax > 1 && tid < 32 * TRIALS_MAX) (&s_rec[0][0])[tid] = sc.recs[(size_t)parity * cmax * 32 + (tid < 32 * cmax ? tid : 0)];
  }
  __syncthreads();
  PHASE_MARK(1);
#ifdef LSD_DEVTOOLS
  const int lvlIn_ = S.level, numLaunchesIn_ = S.numLaunches, pendingIn_ = S.pending, ncandIn_ = S.ncand;
  int pcLog_ = -1;
#endif
  if (S.done) {
    LAUNCH_LOG(1, pendingIn_, ncandIn_, -1);
    if (MODE != TS_EVAL && leader) { copy_words<sizeof(TrackState) / 4>(next, &S, tid, BLOCK); }   // keep both buffers "done"
    return;
  }
  const int lvlPending = S.level;
  // what the finishing launch raises `done` to: the job's tag where the host polls pinned memory for it (single jobs), 1 where it synchronises
  // (batches whose host polls: spec.seq is the batch's tag itself)
  const int doneWord = (MODE == TS_FUSED && spec.seq != 0) ? (BATCH ? spec.seq : (spec.seq >> 12)) : 1;
  if (MODE == TS_FUSED) {
    // The launch is sized for the level with the most (tiles x trials).  This one either evaluates trials at S.level —
    // workgroup = (trial, tile) of that level — or, if the pending decision ends the level, the first evaluation of
    // S.level - 1 (workgroup = tile).  Workgroups that have no work either way leave before the finishing phase (whose loads
    // they would only add to everybody else's).
    const int nbl = job.lv[lvlPending].nblocks;
    const int tr = (spec.specC > 1 && spec.trials[lvlPending] > 1) ? spec.trials[lvlPending] : 1;
    cand = (int)blockIdx.x / nbl;
    bx = (int)blockIdx.x - cand * nbl;
    const bool needTrial = S.pending ? cand < tr : cand == 0;
    const bool needNext = S.pending && lvlPending > job.lastLevel && (int)blockIdx.x < job.lv[lvlPending - 1].nblocks;
    if (!leader && !needTrial && !needNext) return;
  }

  if (BATCH && MODE == TS_LM) {
    // one workgroup per (trial, job): workgroup `cand` proposes the cand-th retry down the reject chain; those without a trial leave
    cand = (int)blockIdx.x;
    bx = 0;
    const int tr = (spec.specC > 1 && spec.trials[lvlPending] > 1) ? spec.trials[lvlPending] : 1;
    if (!leader && !(S.pending && cand < tr)) return;
  }
  if (BATCH && MODE == TS_EVAL) {
    // workgroup = (trial, strip) of the job's level
    const int nbl = job.lv[lvlPending].nblocks;
    cand = (int)blockIdx.x / nbl;
    bx = (int)blockIdx.x - cand * nbl;
  }

  // Throughput-mode strips (batches): the strip's list is the concatenation of its reference blocks (k_ref_blocks, frame.hip).  The loads —
  // one block count per lane (at most 32 blocks), one word of four offsets per lane and step — depend on the level alone, not on the pose:
  // a (trial, strip) workgroup that is going to evaluate at the pending level (the usual case; a level ends three times per job) requests
  // them at the head of the finishing phase, so they travel with the previous round's partial sums instead of behind the LM step.
  constexpr int CHMAX = 8;                       // tilePx <= 8192 (fill_level): 32 blocks, 8 steps of 4 waves
  int cntv = 0;
  unsigned ow[CHMAX];
  auto strip_request = [&](const int lvl, const int tile_) {
    const TrackLevel& L = job.lv[lvl];
    const int px = L.w * L.h;
    const int base_ = tile_ * L.tilePx;
    const int b0_ = base_ >> 8;
    const int mblk_ = (min(base_ + L.tilePx, px) - base_ + 255) >> 8;
    const gbyte* offs = (const gbyte*)L.kf_refBlk;
    const __attribute__((address_space(1))) int* cnts = (const __attribute__((address_space(1))) int*)(offs + ((size_t)((px + 255) >> 8) << 8));
    cntv = lane < mblk_ ? cnts[b0_ + lane] : 0;
#pragma unroll
    for (int c = 0; c < CHMAX; c++) {
      const int blk = c * 4 + wave;
      ow[c] = blk < mblk_ ? *(const __attribute__((address_space(1))) unsigned*)(offs + ((size_t)(b0_ + blk) << 8) + (lane << 2)) : 0u;
    }
  };
  bool havePre = false;
  if (BATCH && MODE == TS_FUSED && S.pending && job.lv[lvlPending].tilePx > 0) {
    const int tr = (spec.specC > 1 && spec.trials[lvlPending] > 1) ? spec.trials[lvlPending] : 1;
    havePre = cand < tr;
    if (havePre) strip_request(lvlPending, xcd_tile(bx, job.lv[lvlPending].nblocks));
  }

  if (MODE != TS_EVAL && S.pending) {
    // Finish the trials of the previous launch, in the order the LM loop would have run them.  All their partial sums, order
    // keys and tail contributions are fetched TOGETHER (one memory round trip, whatever the number of trials), then wave 0 walks
    // through them without further barriers: totals -> LM decision -> (rejected, retry waiting) next trial, whose increment
    // and pose come from the record the workgroups that evaluated it left behind.
    const int ncandPending = S.ncand < 1 ? 1 : (S.ncand > TRIALS_MAX ? TRIALS_MAX : S.ncand);
    const int level = S.level;
    const int nb = job.lv[level].nblocks;
    const size_t trialStrideF4 = (size_t)RS_COLS * max_rows / 4;
      // the last (M % 4) in-image points in reference order = the largest keys (x * h + y, or list index) over all
      // tiles' top-3 lists left behind by the residual pass — per trial, one wave per trial (two when there are more than
      // four), the last wave first: it has no column sums to add
      // a wave takes up to two trials (ca, cb; cb < 0: one): both key loads travel together, the second merge runs while the
      // first trial's contributions travel, and on multi-pass levels the (up to 6) tail points are re-evaluated side by side
      auto tail_two = [&](const int ca, const int cb, auto krn) {
        constexpr int KROWS = decltype(krn)::value;   // rows per lane: KROWS * 64 >= nb
        const int cbb = cb < 0 ? ca : cb;
        const int4* tka = topkey_in0 + (size_t)ca * max_rows;
        const int4* tkb = topkey_in0 + (size_t)cbb * max_rows;
        int4 kva[KROWS], kvb[KROWS];
#pragma unroll
        for (int q = 0; q < KROWS; q++) {
          const int row = lane + 64 * q;
          const int4 ka = tka[row < max_rows ? row : 0];   // unconditional, issued together
          const int4 kb = tkb[row < max_rows ? row : 0];
          kva[q] = row < nb ? ka : make_int4(-1, -1, -1, -1);
          kvb[q] = row < nb ? kb : make_int4(-1, -1, -1, -1);
        }
        auto merge = [&](const int4 (&kv)[KROWS], int (&keys)[3], int (&src)[3]) {
          int k0 = -1, k1 = -1, k2 = -1, e0 = 0, e1 = 0, e2 = 0;
#pragma unroll
          for (int q = 0; q < KROWS; q++) {
            const int e = (lane + 64 * q) * 3;
            const int ks[3] = {kv[q].x, kv[q].y, kv[q].z};
#pragma unroll
            for (int rr = 0; rr < 3; rr++) {
              const int k = ks[rr], ek = e + rr;
              const bool g0 = k > k0, g1 = k > k1, g2 = k > k2;
              k2 = g1 ? k1 : (g2 ? k : k2); e2 = g1 ? e1 : (g2 ? ek : e2);
              k1 = g0 ? k0 : (g1 ? k : k1); e1 = g0 ? e0 : (g1 ? ek : e1);
              k0 = g0 ? k : k0; e0 = g0 ? ek : e0;
            }
          }
#pragma unroll
          for (int r = 0; r < 3; r++) {
            const int m = __builtin_amdgcn_readlane(wave_max_to_lane63(k0), 63);
            const unsigned long long own = __ballot(k0 == m && m >= 0);
            const int owner = own ? (int)__ffsll((long long)own) - 1 : 0;
            keys[r] = m;
            src[r] = __builtin_amdgcn_readlane(e0, owner);
            if (k0 == m && m >= 0) { k0 = k1; e0 = e1; k1 = k2; e1 = e2; k2 = -1; }
          }
        };
        // the owners of the keys left their K2/K3 contributions next to them: single-pass levels (one point per lane) and the strips of
        // a throughput-mode batch (the three largest keys of a strip are re-evaluated at the end of its launch)
        const bool single = job.lv[level].singlePass != 0 || job.lv[level].tilePx > 0;
        const int j = lane & 31, rj = lane >> 5;
        int keysA[3], srcA[3], keysB[3], srcB[3];
        float a0 = 0.f, a2 = 0.f, b0 = 0.f, b2 = 0.f;
        merge(kva, keysA, srcA);
        if (single && j < 29) {
          // the owners left their K2/K3 contributions next to the keys
          const float* tv = topval_in0 + (size_t)ca * max_rows * 96;
          if (keysA[rj] >= 0) a0 = tv[(size_t)srcA[rj] * 32 + j];
          if (rj == 0 && keysA[2] >= 0) a2 = tv[(size_t)srcA[2] * 32 + j];
        }
        if (cb >= 0) {
          merge(kvb, keysB, srcB);
          if (single && j < 29) {
            const float* tv = topval_in0 + (size_t)cb * max_rows * 96;
            if (keysB[rj] >= 0) b0 = tv[(size_t)srcB[rj] * 32 + j];
            if (rj == 0 && keysB[2] >= 0) b2 = tv[(size_t)srcB[2] * 32 + j];
          }
        }
        if (lane == 0) {
          s_nsubT[ca] = (keysA[0] >= 0) + (keysA[1] >= 0) + (keysA[2] >= 0);
          if (cb >= 0) s_nsubT[cb] = (keysB[0] >= 0) + (keysB[1] >= 0) + (keysB[2] >= 0);
        }
        if (single) {
          if (j < 29) {
            if (keysA[rj] >= 0) s_subT[ca][rj][j] = a0;
            if (rj == 0 && keysA[2] >= 0) s_subT[ca][2][j] = a2;
            if (cb >= 0) {
              if (keysB[rj] >= 0) s_subT[cb][rj][j] = b0;
              if (rj == 0 && keysB[2] >= 0) s_subT[cb][2][j] = b2;
            }
          }
        } else {
          // multi-pass level: re-evaluate the tail points at their trial's pose: lanes 0..2 trial ca, 3..5 trial cb
          const bool second = lane >= 3;
          const int r = second ? lane - 3 : lane;
          const int c = second ? cb : ca;
          int key = -1;
          if (lane < 6 && c >= 0) key = second ? (r == 0 ? keysB[0] : (r == 1 ? keysB[1] : keysB[2])) : (r == 0 ? keysA[0] : (r == 1 ? keysA[1] : keysA[2]));
          if (key >= 0) {
            EvalCtx a;
            make_ctx_dev(job, S, level, a);
            if (c > 0) {
              const float* rec = s_rec[c];
#pragma unroll
              for (int i = 0; i < 9; i++) a.R[i] = rec[13 + i];
#pragma unroll
              for (int i = 0; i < 3; i++) a.t[i] = rec[22 + i];
            }
            const int idx = (a.npts >= 0) ? key : ((key / a.h) + (key % a.h) * a.w);
            float px, py, pz, I_ref, var;
            int maskIdx;
            fetch_point(a, idx, px, py, pz, I_ref, var, maskIdx);
            PointOut o;
            eval_point(a, px, py, pz, I_ref, var, o);
            float* sub = s_subT[c][r];
            sub[0] = o.werr;
            int k = 1;
#pragma unroll
            for (int rr = 0; rr < 6; rr++) {
              float Jw = o.J[rr] * o.w;
#pragma unroll
              for (int cc = rr; cc < 6; cc++) sub[k++] = Jw * o.J[cc];
            }
            float resw = o.res * o.w;
#pragma unroll
            for (int rr = 0; rr < 6; rr++) sub[k++] = resw * o.J[rr];
            sub[k++] = resw * o.res;
          }
        }
      };
      // Strips of a throughput-mode batch, at most 24 per job: a strip's launch leaves one 32-float row per candidate — its (up to) three
      // largest keys' K2/K3 contributions [0..28] and the key itself [31] — so ALL rows of the job travel in the same round trip as the
      // partial sums (NQ 16-byte loads per lane), the three largest keys are picked from the rows in registers, and their rows go to the
      // tail table: no second, dependent round trip (the keys first, then the owners' contributions) in front of the LM step.
      auto tail_rows = [&](const int ca, const int cb, auto nqn) {
        constexpr int NQ = decltype(nqn)::value;   // NQ * 64 >= nb * 24 (16-byte words of the rows)
        const int cbb = cb < 0 ? ca : cb;
        const float4* ta = (const float4*)(topval_in0 + (size_t)ca * max_rows * 96);
        const float4* tb = (const float4*)(topval_in0 + (size_t)cbb * max_rows * 96);
        const int nf4 = nb * 24;
        float4 va[NQ], vb[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          const int f = q * 64 + lane;
          va[q] = ta[f < nf4 ? f : 0];   // unconditional, issued together
          vb[q] = tb[f < nf4 ? f : 0];
        }
        auto pick = [&](const float4 (&v)[NQ], const int c) {
          int k0 = -1, k1 = -1, k2 = -1, e0 = 0, e1 = 0, e2 = 0;
#pragma unroll
          for (int q = 0; q < NQ; q++) {
            const int f = q * 64 + lane;
            const int k = ((lane & 7) == 7 && f < nf4) ? __float_as_int(v[q].w) : -1, ek = f >> 3;   // the row's key sits in its last word
            const bool g0 = k > k0, g1 = k > k1, g2 = k > k2;
            k2 = g1 ? k1 : (g2 ? k : k2); e2 = g1 ? e1 : (g2 ? ek : e2);
            k1 = g0 ? k0 : (g1 ? k : k1); e1 = g0 ? e0 : (g1 ? ek : e1);
            k0 = g0 ? k : k0; e0 = g0 ? ek : e0;
          }
          int nsub = 0;
#pragma unroll
          for (int r = 0; r < 3; r++) {
            const int m = __builtin_amdgcn_readlane(wave_max_to_lane63(k0), 63);
            const unsigned long long own = __ballot(k0 == m && m >= 0);
            const int owner = own ? (int)__ffsll((long long)own) - 1 : 0;
            const int row = __builtin_amdgcn_readlane(e0, owner);
            if (k0 == m && m >= 0) { k0 = k1; e0 = e1; k1 = k2; e1 = e2; k2 = -1; }
            if (m >= 0) {
              nsub++;
              // the row's 16-byte words sit in v[row / 8] of lanes 8 (row % 8) ..: a scalar branch per q (row is uniform).  (The empty asm
              // keeps the compiler from folding the branches into an indexed read of v[] — scratch memory; a chain of 4 (NQ - 1) selects
              // per candidate was the first form.)
              const int qsel = row >> 3;
#pragma unroll
              for (int q = 0; q < NQ; q++) {
                if (q == qsel) {
                  float4 w4 = v[q];
                  asm volatile("" : "+v"(w4.x), "+v"(w4.y), "+v"(w4.z), "+v"(w4.w));
                  if ((lane >> 3) == (row & 7)) *(float4*)&s_subT[c][r][(lane & 7) * 4] = w4;
                }
              }
            }
          }
          if (lane == 0) s_nsubT[c] = nsub;
        };
        pick(va, ca);
        if (cb >= 0) pick(vb, cb);
      };
    // this wave's share of the tail work, run between the issue of the column-sum loads and their additions
    auto wave_tail = [&]() {
      // trial c -> wave WAVES-1 - (c mod WAVES): the last wave first (it has no column sums to add)
      const int ca = WAVES - 1 - wave;
      const int cb = ca + WAVES < ncandPending ? ca + WAVES : -1;
      if (ca < ncandPending) {
        if (BATCH && job.lv[level].tilePx > 0 && nb <= 24) {
          if (nb <= 5) tail_rows(ca, cb, std::integral_constant<int, 2>());
          else if (nb <= 13) tail_rows(ca, cb, std::integral_constant<int, 5>());
          else tail_rows(ca, cb, std::integral_constant<int, 9>());
        } else if (nb <= 128) tail_two(ca, cb, std::integral_constant<int, 2>());
        else tail_two(ca, cb, std::integral_constant<int, 5>());
      }
    };
    if (tid == SUMW * 64 - 1) stage_lm_par(job, level, s_par, (spec.specC > 1 && spec.trials[level] > 1) ? spec.trials[level] : 1);
    if (wave < SUMW) {
      // fixed-order column sums over the tiles' partial rows (row-major [tile][RS_COLS]: a wave's load covers whole rows, i.e.
      // contiguous memory): thread t < NSLOT * 11 takes columns 4 c4 .. 4 c4 + 3 of rows slot, slot + NSLOT, ... , for every
      // trial (the other lanes of these waves run along and store nothing: the tail work in the middle is wave-wide)
      const int t = tid;
      const int slot_ = t / (RS_COLS / 4);
      const bool sumLane = slot_ < NSLOT;
      const int slot = sumLane ? slot_ : 0, c4 = sumLane ? t - slot_ * (RS_COLS / 4) : 0;
      {
        const int K = (nb + NSLOT - 1) / NSLOT;                // rows per slot
        const float4* p = (const float4*)sums_in0 + c4;
        const size_t trialStrideF4 = (size_t)RS_COLS * max_rows / 4;
        // the loads are unconditional and issued together (trials past the last one re-read it, rows past the last one re-read
        // the slot's first row); coarse levels (few tiles) take the short forms
        auto colsum = [&](auto qn, auto ntn, const int base) {
          constexpr int Q = decltype(qn)::value, NT = decltype(ntn)::value;
          float4 v[NT][Q];
#pragma unroll
          for (int c = 0; c < NT; c++) {
            const int pc = base + c < ncandPending ? base + c : ncandPending - 1;
            const float4* pp = p + (size_t)pc * trialStrideF4;
#pragma unroll
            for (int q = 0; q < Q; q++) {
              const int r = slot + NSLOT * q;
              v[c][q] = pp[(size_t)(r < nb ? r : slot) * (RS_COLS / 4)];
            }
          }
#pragma unroll
          for (int c = 0; c < NT; c++) {
            float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < Q; q++) {
              const bool live = q < K && slot + NSLOT * q < nb;
              a4.x += live ? v[c][q].x : 0.f;
              a4.y += live ? v[c][q].y : 0.f;
              a4.z += live ? v[c][q].z : 0.f;
              a4.w += live ? v[c][q].w : 0.f;
            }
            if (sumLane && base + c < ncandPending) *(float4*)&s_sumT[base + c][slot][c4 * 4] = a4;
          }
        };
        typedef std::integral_constant<int, 1> I1;
        typedef std::integral_constant<int, 3> I3;
        typedef std::integral_constant<int, TRIALS_MAX> IM;
        if (ncandPending == 1) {
          if (K <= 2) colsum(std::integral_constant<int, 2>(), I1(), 0);
          else if (K <= 5) colsum(std::integral_constant<int, 5>(), I1(), 0);
          else if (K <= 10) colsum(std::integral_constant<int, 10>(), I1(), 0);
          else colsum(std::integral_constant<int, QMAX>(), I1(), 0);
        } else if (K <= 2) colsum(std::integral_constant<int, 2>(), IM(), 0);
        else if (K <= 5) colsum(std::integral_constant<int, 5>(), IM(), 0);
        else if (K <= 10) { for (int base = 0; base < ncandPending; base += 3) colsum(std::integral_constant<int, 10>(), I3(), base); }
        else { for (int base = 0; base < ncandPending; base++) colsum(std::integral_constant<int, QMAX>(), I1(), base); }
#ifdef LSD_PHASE_TRACE
        if (trOn_ && tid == 0) tr_[17] = clock64();
#endif
      }
    }
    wave_tail();
#ifdef LSD_PHASE_TRACE
    if (trOn_ && tid == SUMW * 64) s_trp[18] = clock64();
#endif
    __syncthreads();
    PHASE_MARK(2);
    if (wave == 0) {
      // Which pending trial does the LM loop stop at?  Trial c stops it when it diverges (M < minWarped), is accepted
      // (error < lastErr) or is rejected with a step below stepSizeMin — each test needs only that trial's own totals and
      // increment, so lane c answers for trial c and the first "yes" wins.  The trials before it were plain rejections: they
      // leave nothing behind but lambda, incTry and the counters, which are advanced in closed form; then ONE LM step runs,
      // on the totals of the trial that stopped the loop (or of the last pending one).
      int pc = 0;
      if (ncandPending > 1) {
        const int c = tid < ncandPending ? tid : 0;
        float Mf = s_sumT[c][0][RS_M], ws = s_sumT[c][0][RS_WERR];
#pragma unroll
        for (int k = 1; k < NSLOT; k++) { Mf += s_sumT[c][k][RS_M]; ws += s_sumT[c][k][RS_WERR]; }
        const int Mc = (int)Mf;
        int needc = Mc & 3;
        if (needc > s_nsubT[c]) needc = s_nsubT[c];
        if (needc > 0) ws -= s_subT[c][0][0];
        if (needc > 1) ws -= s_subT[c][1][0];
        if (needc > 2) ws -= s_subT[c][2][0];
        const float werrc = lm_werr(ws, Mc);
        const float* ic = c == 0 ? S.inc : s_rec[c];
        const float i0 = ic[0], i1 = ic[1], i2 = ic[2], i3 = ic[3], i4 = ic[4], i5 = ic[5];
        const float incdot = (i0 * i0 + (i1 * i1 + i2 * i2)) + (i3 * i3 + (i4 * i4 + i5 * i5));
        const bool stop = Mc < s_par.minWarped || werrc < S.lastErr || !(incdot > s_par.stepSizeMin);
        const unsigned long long sm = __ballot(stop && tid < ncandPending);
        pc = sm ? (int)__ffsll((long long)sm) - 1 : ncandPending - 1;
        if (pc > 0) {
          float lam = S.LM_lambda;
          const int it0 = S.incTry;
          for (int j = 0; j < pc; j++) lam = lm_lambda_fail(lam, it0 + j, s_par.lambdaFailFac);
          // algorithmic bytes of the skipped evaluations (as in lm_wave)
          float skipped;
          {
            float NR = s_sumT[0][0][RS_NREF];
#pragma unroll
            for (int k = 1; k < NSLOT; k++) NR += s_sumT[0][k][RS_NREF];
            const float wh = (float)s_par.w * (float)s_par.h;
            const float texels = 4.0f * NR < wh ? 4.0f * NR : wh;
            skipped = 20.0f * NR + (s_par.writeMask ? 5.0f * NR : 0.0f) + 12.0f * texels;
          }
          const float* rec = s_rec[pc];
          const float bytes0 = S.bytes;
          const int ne0 = S.numEvaluations, le0 = S.levelEvals[level];
          float bytes1 = bytes0;
          for (int j = 0; j < pc; j++) bytes1 = bytes1 + skipped;
          S.LM_lambda = lam;
          S.incTry = it0 + pc;
          S.numEvaluations = ne0 + pc;
          if (tid == 0) S.levelEvals[level] = le0 + pc;
          S.bytes = bytes1;
          if (tid < 6) S.inc[tid] = rec[tid];
          if (tid < 7) ((float*)&S.Tn)[tid] = rec[6 + tid];
          if (tid < 9) S.R[tid] = rec[13 + tid];
          if (tid < 3) S.t[tid] = rec[22 + tid];
        }
      }
      {
        // column totals and tail-drop correction, one column per lane; then the LM decision in the same wave
        float Mf = s_sumT[pc][0][RS_M];
#pragma unroll
        for (int k = 1; k < NSLOT; k++) Mf += s_sumT[pc][k][RS_M];
        float s = 0.f;
        if (tid < RS_END) {
          s = s_sumT[pc][0][tid];
#pragma unroll
          for (int k = 1; k < NSLOT; k++) s += s_sumT[pc][k][tid];
        }
        const int M = (int)Mf;
        int need = M & 3;
        if (need > s_nsubT[pc]) need = s_nsubT[pc];
        // RS_WERR -> 0, RS_A0.. -> 1..21, RS_B0.. -> 22..27, RS_ERR -> 28
        const int subIdx = (tid == RS_WERR) ? 0 : ((tid >= RS_A0 && tid < RS_B0) ? 1 + tid - RS_A0 : ((tid >= RS_B0 && tid < RS_ERR) ? 22 + tid - RS_B0 : (tid == RS_ERR ? 28 : -1)));
        const float sub0 = s_subT[pc][0][subIdx < 0 ? 0 : subIdx], sub1 = s_subT[pc][1][subIdx < 0 ? 0 : subIdx], sub2 = s_subT[pc][2][subIdx < 0 ? 0 : subIdx];
        if (subIdx >= 0) {
          if (need > 0) s -= sub0;
          if (need > 1) s -= sub1;
          if (need > 2) s -= sub2;
        }
        if (tid < RS_NUM) sh.tot[tid] = s;
#ifdef LSD_ORDER_CHECK
        if (!BATCH && spec.dbgCounters) {
          // do all workgroups of this launch finish the previous evaluation with the same totals?
          unsigned hv = tid < RS_END ? __float_as_uint(s) * (2654435761u * (unsigned)(tid + 1)) : 0u;
          hv += (unsigned)pc * 97u;
          for (int off = 32; off > 0; off >>= 1) hv += __shfl_xor(hv, off);
          if (tid == 0) {
            const unsigned long long slot = 8 + 2 * ((spec.dbgCum / 400ull) % 8192ull);
            __hip_atomic_fetch_min(&spec.dbgCounters[slot], (unsigned long long)hv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max(&spec.dbgCounters[slot + 1], (unsigned long long)hv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
#endif
        PHASE_MARK(3);
#ifdef LSD_PHASE_TRACE
        if (trOn_ && tid == 0) { tr_[19] = (unsigned long long)ncandPending; tr_[7] = (unsigned long long)pc; }
        lm_wave<SPEC_LM>(s_par, S, s, sh.tot, tid, leader ? out : nullptr, trOn_ ? tr_ : nullptr, pc, cand, doneWord);
#else
        lm_wave<SPEC_LM>(s_par, S, s, sh.tot, tid, leader ? out : nullptr, nullptr, pc, cand, doneWord);
#endif
      }
    }
    __syncthreads();
    PHASE_MARK(4);
    if (S.done) {
      // the frame's refPixelWasGood must be what the last trial the LM loop executed wrote: trials > 0 wrote side planes
      if (!BATCH && MODE == TS_FUSED && S.lastCand > 0 && spec.copyMask != 0 && job.lv[level].writeMask && cand == 0 && bx < nb) {
        const uint8_t* side = spec.wasGoodSide + (size_t)(S.lastCand - 1) * spec.maskStride;
        EvalCtx a;
        make_ctx_dev(job, S, level, a);
        const int work = a.npts >= 0 ? a.npts : a.w * a.h;
        for (int i = bx * BLOCK + tid; i < work; i += nb * BLOCK) {
          float px, py, pz, I_ref, var;
          int maskIdx;
          if (fetch_point(a, i, px, py, pz, I_ref, var, maskIdx) && maskIdx >= 0) job.wasGood[maskIdx] = side[maskIdx];
        }
      }
      if (leader) copy_words<sizeof(TrackState) / 4>(next, &S, tid, BLOCK);
      LAUNCH_LOG(2, pendingIn_, ncandIn_, S.lastCand);
      return;
    }
    // a workgroup that evaluates trial c > 0 leaves that trial's increment and pose for the launch that finishes it
    if (cand > 0 && bx == 0 && S.phase == 1 && cand < S.ncand && tid < 25) {
      float v;
      if (tid < 6) v = S.inc[tid];
      else if (tid < 13) v = ((const float*)&S.Tn)[tid - 6];
      else if (tid < 22) v = S.R[tid - 13];
      else v = S.t[tid - 22];
      sc.recs[((size_t)outp * cmax + cand) * 32 + tid] = v;
    }
  }

  if (MODE == TS_LM) {
    __syncthreads();              // (every wave has tested S.pending before it changes: see the note at the residual evaluation below)
    if (tid == 0) { S.pending = 1; S.numLaunches = S.numLaunches + 1; }   // (numLaunches of a batch job: its rounds)
    __syncthreads();
    if (leader) copy_words<sizeof(TrackState) / 4>(next, &S, tid, BLOCK);
    return;
  }

  // ---- residual evaluation at S.level / S.R, S.t ------------------------------------------------------------------
  const int level = S.level;
  // (S.pending = 1 / S.numLaunches++ for the state this launch publishes are set further down, behind the first barrier of the
  // reduction: in a job's first launch the finishing phase above is skipped, so no barrier separates a wave that is still about to test
  // `S.pending` from wave 0 arriving here — with the write here, a wave held up by LDS traffic of another stream's workgroups on
  // the same CU could read 1, enter the finishing phase alone and pair its barriers with the others' reduction barriers: one
  // tile's sums came out as LDS residue.  Seen only with a second stream active, profiles/r04_notes.md.)
  const int nb = job.lv[level].nblocks;
  if (MODE == TS_FUSED && level != lvlPending) { cand = 0; bx = (int)blockIdx.x; }   // first evaluation of the next level
  if (bx >= nb || cand >= S.ncand) return;   // workgroup 0 always has work: it publishes the state at the end
  sums_out += (size_t)cand * RS_COLS * max_rows;
  topkey_out += (size_t)cand * max_rows;
  topval_out += (size_t)cand * max_rows * 96;
  const int tile = xcd_tile(bx, nb);
  EvalCtx a;
  make_ctx_dev(job, S, level, a);   // (fetched beside the finishing phase for the pending level: measured, +-0 — the scalar loads are not what the step behind the LM waits for)
  if (BATCH && MODE == TS_EVAL && cand > 0) {
    // a retry further down the reject chain: its pose is in the record the LM workgroup that proposed it left
    const float* rec = sc.recs + ((size_t)parity * cmax + cand) * 32;
#pragma unroll
    for (int i = 0; i < 9; i++) a.R[i] = rec[13 + i];
#pragma unroll
    for (int i = 0; i < 3; i++) a.t[i] = rec[22 + i];
  }
  gbyte* wasGood = (gbyte*)(job.lv[level].writeMask ? (cand == 0 ? job.wasGood : spec.wasGoodSide + (size_t)(cand - 1) * spec.maskStride) : nullptr);
  const int work = a.npts >= 0 ? a.npts : a.w * a.h;
  float acc[RS_END];
#pragma unroll
  for (int k = 0; k < RS_END; k++) acc[k] = 0.f;
  int key0 = -1, key1 = -1, key2 = -1;   // reference-order keys of this lane's in-image points (descending)

  if (BATCH && job.lv[level].tilePx > 0) {
    // Throughput mode (batches): the workgroup owns a strip of tilePx consecutive pixels of the keyframe level.  It first
    // compacts the strip's valid reference pixels (semi-dense: ~30 %) into an LDS list — fixed order: chunk, then pixel
    // slot, then lane, so the result is run-to-run deterministic — and then evaluates the list with all lanes busy.
    unsigned* s_list = (unsigned*)s_red;   // (x | y << 16); s_red is not live before the reduction
    // The strip's list is the concatenation of its reference blocks (k_ref_blocks, frame.hip: per 256 consecutive pixels the offsets of the
    // valid ones, compacted in pixel order, and a count): wave v takes block 4 c + v of the strip in step c, lane l its slots l, l + 64, ...;
    // every wave derives the blocks' places in the list from the counts itself (at most 32 blocks: one lane each), so nothing is exchanged
    // and the only barrier is the one in front of the evaluation.  (Until round 6 the strip streamed the level's validity planes, 8 bytes per
    // pixel, and compacted them with four ballots per 1024 pixels in every evaluation: a quarter of a level-1 round at 64 jobs.)
    const int tilePx = job.lv[level].tilePx;
    const int base = tile * tilePx;
    const int stripEnd = min(base + tilePx, work);
    const int b0 = base >> 8;                                  // strips are multiples of 256 pixels
    const int mblk = (stripEnd - base + 255) >> 8;             // <= 32 (tilePx <= 8192, fill_level)
    const float inv_w = 1.0f / (float)a.w;
    if (!(havePre && level == lvlPending)) strip_request(level, tile);   // (else: requested next to the state, at the head of the finishing phase)
    // inclusive scan of the (at most 32) block counts over lanes 0..31 in six DPP additions, the blocks' entries through v_readlane (the
    // wave index is uniform: readfirstlane tells the compiler so)
    int incl = cntv + dpp_i0<0x111, 0xf, 0xf>(cntv);      // row_shr:1
    incl += dpp_i0<0x112, 0xf, 0xf>(cntv);                // row_shr:2
    incl += dpp_i0<0x113, 0xf, 0xf>(cntv);                // row_shr:3
    incl += dpp_i0<0x114, 0xf, 0xe>(incl);                // row_shr:4 bank_mask:0xe
    incl += dpp_i0<0x118, 0xf, 0xc>(incl);                // row_shr:8 bank_mask:0xc
    incl += dpp_i0<0x142, 0xa, 0xf>(incl);                // row_bcast:15 row_mask:0xa
    const int total = __builtin_amdgcn_readlane(incl, 31);
    const int waveU = __builtin_amdgcn_readfirstlane(wave);
    PHASE_MARK(20);
#pragma unroll
    for (int c = 0; c < CHMAX; c++) {
      const int blk = c * 4 + waveU;
      if (blk < mblk) {
        const int cb = __builtin_amdgcn_readlane(cntv, blk);
        const int pb = __builtin_amdgcn_readlane(incl, blk) - cb;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int sl = lane + 64 * k;                       // (the block's bytes are slot-interleaved: k_ref_blocks)
          if (sl < cb) {
            const int i = ((b0 + blk) << 8) + (int)((ow[c] >> (8 * k)) & 255u);
            int y = (int)((float)i * inv_w);
            int x = i - y * a.w;
            if (x < 0) { y--; x += a.w; }
            if (x >= a.w) { y++; x -= a.w; }
            s_list[pb + sl] = (unsigned)x | ((unsigned)y << 16);
          }
        }
      }
    }
    __syncthreads();
    PHASE_MARK(21);
#ifdef LSD_PHASE_TRACE
    if (trOn_ && tid == 0) tr_[23] = (unsigned long long)total;
#endif
    if (total > 0) {
      // three-stage software pipeline over the list entries tid, tid + BLOCK, ...: (A) list entry + keyframe planes,
      // (B) reference point, warp, texel fetch, (C) residual / weights / normal equations.  Stage A of entry r + 2 and
      // stage B of entry r + 1 are issued before stage C of entry r, so two dependent memory round trips overlap the
      // arithmetic.  Lanes past the end of the list run the loads on entry 0 and discard them.
      struct StA { unsigned xy; float var, id, img; bool live; };
      struct StB { unsigned xy; float pz, I_ref, var; PointWarp q; PointTexels t; bool live; };
      auto stageA = [&](int p, StA& A) {
        A.live = p < total;
        A.xy = s_list[A.live ? p : 0];
        const int i = __mul24((int)(A.xy >> 16), a.w) + (int)(A.xy & 0xffffu);
        A.var = a.kf_idepthVar[i];
        A.id = a.kf_idepth[i];
        A.img = a.kf_image[i];
      };
      auto stageB = [&](const StA& A, StB& B) {
        B.live = A.live;
        B.xy = A.xy;
        const int bx_ = (int)(A.xy & 0xffffu), by_ = (int)(A.xy >> 16);
        const float inv = lsd_rcp_exact(A.id);
        const float px = inv * (a.fxi * bx_ + a.cxi), py = inv * (a.fyi * by_ + a.cyi);
        B.pz = inv * 1.0f;
        B.I_ref = A.img; B.var = A.var;
        eval_warp(a, px, py, B.pz, B.q);
        eval_fetch(a, B.q, B.live && B.q.in_image, B.t);
      };
      const int rounds = (total + BLOCK - 1) / BLOCK;
      StA A1, A2;
      StB B0, B1;
      stageA(tid, A1);
      stageB(A1, B0);
      stageA(tid + BLOCK, A1);
      PHASE_MARK(22);
      auto stageC = [&](const StB& B) {
        if (B.live) {
          const int x_ = (int)(B.xy & 0xffffu), y_ = (int)(B.xy >> 16);
          const int i = __mul24(y_, a.w) + x_;
          acc[RS_NREF] += 1.f;
          if (!B.q.in_image) {
            if (wasGood) wasGood[i] = 0;
          } else {
            PointOut o;
            eval_finish(a, B.q, B.t, B.pz, B.I_ref, B.var, o);
            if (wasGood) wasGood[i] = o.good ? 1 : 0;
            top3_insert(__mul24(x_, a.h) + y_, key0, key1, key2);
            accumulate_point(o, acc);
          }
        }
      };
      // two rounds per trip with the roles of (A1, A2) and (B0, B1) swapped in the second half: a rotating pipeline written with
      // struct copies costs ~30 register moves per round
      for (int r = 0; r < rounds; r += 2) {
        stageB(A1, B1);
        stageA(tid + (r + 2) * BLOCK, A2);
        stageC(B0);
        stageB(A2, B0);
        stageA(tid + (r + 3) * BLOCK, A1);
        stageC(B1);                       // entry r + 1: not live past the end of the list
      }
    }
    __syncthreads();   // the list aliases s_red
  } else
  for (int i = tile * BLOCK + tid; i < work; i += nb * BLOCK) {
    float px, py, pz, I_ref, var;
    int maskIdx;
    if (fetch_point(a, i, px, py, pz, I_ref, var, maskIdx)) {
      acc[RS_NREF] += 1.f;
      PointOut o;
      eval_point(a, px, py, pz, I_ref, var, o);
      if (!o.in_image) {
        if (wasGood && maskIdx >= 0) wasGood[maskIdx] = 0;
      } else {
        if (wasGood && maskIdx >= 0) wasGood[maskIdx] = o.good ? 1 : 0;
        top3_insert(a.npts >= 0 ? i : (i % a.w) * a.h + (i / a.w), key0, key1, key2);
        accumulate_point(o, acc);
      }
    }
  }

  PHASE_MARK(5);
#ifdef LSD_PHASE_TRACE
  if (trOn_ && tid == 0) { tr_[10] = (unsigned long long)level; tr_[11] = (unsigned long long)nb; tr_[26] = (unsigned long long)lvlPending; }
#endif
  // workgroup reduction through LDS: every lane parks its 41 accumulators in column `tid` of s_red (row stride
  // BLOCK + 1: conflict-free both ways), then thread (slice, k) adds a contiguous run of lanes of row k in lane order
  // and 41 threads add the slices — ~130 instructions per wave instead of 41 x 7 DPP steps
  constexpr int RSLICE = BLOCK / CPP;                  // slices per row
  constexpr int RRUN = (BLOCK + RSLICE - 1) / RSLICE;  // lanes per slice
  // Strips of a throughput-mode batch: the (up to) three in-image points of the strip with the largest reference-order keys are
  // candidates for the job's tail (the last M % 4 in-image points, which the SSE loops of the reference leave out): three lanes evaluate
  // them once more, alone, and leave their K2/K3 contributions in the strip's candidate rows — the launch that finishes this evaluation
  // then needs no evaluation of its own in front of the LM step (it used to: keys, reference point, texels = three dependent round
  // trips).  The two round trips of this evaluation run under the barriers of the reduction.
  const bool stripRows = BATCH && job.lv[level].tilePx > 0;
  const bool candLane = tid >= BLOCK - 64 && tid < BLOCK - 61;   // lanes 0..2 of the LAST wave: wave 0 adds the slices and stores the sums meanwhile
  int cKey = -1, cX = 0, cY = 0;
  float cVar = 0.f, cId = 1.f, cImg = 0.f, cPz = 0.f;
  PointWarp cq;
  PointTexels ct;
  if (stripRows) {
    block_top3(key0, key1, key2, s_wtop, s_top);
    if (candLane) {
      cKey = s_top[tid - (BLOCK - 64)];
      if (cKey >= 0) {
        cX = cKey / a.h;
        cY = cKey - cX * a.h;
        const int ci = __mul24(cY, a.w) + cX;
        cVar = a.kf_idepthVar[ci];
        cId = a.kf_idepth[ci];
        cImg = a.kf_image[ci];
      }
    }
  }
#pragma unroll
  for (int hp = 0; hp < NPASS; hp++) {
    if (hp > 0) __syncthreads();
#pragma unroll
    for (int k = 0; k < CPP; k++)
      if (hp * CPP + k < RS_END) s_red[k * (BLOCK + 1) + tid] = acc[hp * CPP + k];
    __syncthreads();
    if (MODE == TS_FUSED && hp == 0 && tid == 0) { S.pending = 1; S.numLaunches = S.numLaunches + 1; }   // every wave is past its reads of S.pending
    if (stripRows && hp == 0 && candLane && cKey >= 0) {
      const float inv = lsd_rcp_exact(cId);                // (fetch_point's arithmetic)
      const float px = inv * (a.fxi * cX + a.cxi), py = inv * (a.fyi * cY + a.cyi);
      cPz = inv * 1.0f;
      eval_warp(a, px, py, cPz, cq);
      eval_fetch(a, cq, cq.in_image, ct);
    }
    {
      const int slice = tid / CPP, k = tid - slice * CPP;
      if (slice < RSLICE) {
        const float* row = s_red + k * (BLOCK + 1);
        const int j0 = slice * RRUN;
        float v[RRUN];
#pragma unroll
        for (int j = 0; j < RRUN; j++) v[j] = row[j0 + j];   // unconditional: the run may spill 2 words into the next row (allocated)
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < RRUN; j++) s += (j0 + j < BLOCK) ? v[j] : 0.f;
        s_sum[slice][k] = s;
      }
    }
    __syncthreads();
    if (stripRows && hp == 0 && candLane) {
      float* row = topval_out + (size_t)(tile * 3 + (tid - (BLOCK - 64))) * 32;
      if (cKey >= 0) {
        PointOut o;
        eval_finish(a, cq, ct, cPz, cImg, cVar, o);
        tail_row(o, row);
      }
      ((int*)row)[31] = cKey;
    }
    if (tid < CPP && hp * CPP + tid < RS_END) {
      float s = s_sum[0][tid];
#pragma unroll
      for (int sl = 1; sl < RSLICE; sl++) s += s_sum[sl][tid];
      sums_out[(size_t)tile * RS_COLS + (hp * CPP + tid)] = s;
    }
  }
  if (!stripRows) block_top3(key0, key1, key2, s_wtop, s_top);
  if (tid == 0) topkey_out[tile] = make_int4(s_top[0], s_top[1], s_top[2], -1);
  if (job.lv[level].singlePass && key0 >= 0) {
    // one point per lane: its accumulators are exactly its K2/K3 contributions (0 + x == x)
    const int r = key0 == s_top[0] ? 0 : (key0 == s_top[1] ? 1 : (key0 == s_top[2] ? 2 : -1));
    if (r >= 0) {
      float* dst = topval_out + (size_t)(tile * 3 + r) * 32;
      dst[0] = acc[RS_WERR];
#pragma unroll
      for (int k = 0; k < 21; k++) dst[1 + k] = acc[RS_A0 + k];
#pragma unroll
      for (int k = 0; k < 6; k++) dst[22 + k] = acc[RS_B0 + k];
      dst[28] = acc[RS_ERR];
    }
  }
  if (MODE == TS_FUSED && leader) copy_words<sizeof(TrackState) / 4>(next, &S, tid, BLOCK);   // S.pending was set before the barriers above
  // the budget's last launch ends with the job unfinished: tell the host (pinned memory), which polls this next to `done` and
  // appends launches — the stream order makes them follow; no hipStreamQuery in the wait loop (each one puts a marker packet into
  // the queue the chain runs through: ~3 us per frame)
 
  if (MODE == TS_FUSED && leader && tid == 0 && spec.last != 0) out->exhausted = spec.seq;
  LAUNCH_LOG(3, pendingIn_, ncandIn_, S.lastCand);
  PHASE_MARK(6);
#ifdef LSD_PHASE_TRACE
  if (trOn_ && tid == 0) tr_[9] = wall_clock64();
#endif
}

template <int BLOCK, bool BATCH, int MODE = TS_FUSED>
__global__ __launch_bounds__(BLOCK) void k_track_step(TrackJob jobv, const TrackJob* __restrict__ jobs, TrackState* __restrict__ st2,
                                                       TrackScratch sc, TrackSummary* __restrict__ out, int parity, int first, TrackSpec spec) {
#ifdef LSD_ORDER_CHECK
  // developer build: is every workgroup of every earlier launch of this stream finished when a workgroup of this launch starts?
  if (!BATCH && spec.dbgCounters && threadIdx.x == 0) {
    const unsigned long long fin = __hip_atomic_load(&spec.dbgCounters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (fin < spec.dbgCum) {
      __hip_atomic_fetch_add(&spec.dbgCounters[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_max(&spec.dbgCounters[2], spec.dbgCum - fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#endif
  track_step_impl<BLOCK, BATCH, MODE>(jobv, jobs, st2, sc, out, parity, first, spec);
#ifdef LSD_ORDER_CHECK
  __syncthreads();
  if (!BATCH && spec.dbgCounters && threadIdx.x == 0) __hip_atomic_fetch_add(&spec.dbgCounters[0], 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// ---- The coarse levels of a throughput-mode job inside ONE workgroup (round 6) --------------------------------------------------------
// A lock-step round of a batch costs its latency chain (state, partial sums of other workgroups, the LM step, list, pipeline fill,
// reduction: 11-13 us of workgroup time with 1.5-2.5 us of evaluation in it at levels 3 and 4, profiles/r06_notes.md section 9), every
// job pays the rounds of the slowest, and every (trial, strip) workgroup repeats the finishing phase.  A level small enough to be one
// strip and one LDS tile (at most LSD_SOLO_MAX_PX pixels: levels 3 and 4 of a 640x480 job) needs none of that: one workgroup per job runs
// the level's whole LM loop at its own pace and leaves the state at the first larger level for the lock-step rounds
// (k_track_step<.., true, TS_FUSED> with first = 0).  Everything an iteration reads is staged when the workgroup enters the level:
//   * the TRACKED FRAME'S TEXEL PLANE in LDS (gx, gy, I, 0: 16 bytes per pixel, 76.8 KB at 80x60) — the form BASELINE.json's north_star
//     names: LDS-staged image tiles for the bilinear taps (getInterpolatedElement43, C/util/globalFuncs.h:63-77); the tile is the level;
//   * the level's REFERENCE POINTS in registers: lane t owns points t, t + 512, ... of the level's list (the keyframe's reference blocks
//     in pixel order, as a strip of the lock-step rounds builds its own), at most LSD_SOLO_TRIPS of them, 2 registers each: pixel,
//     1 / idepth (= the point's z; x and y follow with the two multiply-adds of makePointCloud, TrackingReference.cpp:128-138); colour and
//     variance, read once per point and iteration, sit in LDS beside the tile — none of it depends on the pose.
// An LM iteration then touches no global memory: warp from registers, four 16-byte taps from LDS, finish, accumulate; the pose in scalar
// registers.  Four barriers per iteration: the upper half parks its sums and
// every wave its top-3 keys | the halves fold, wave 0 merges the top-3 and requests the three tail candidates' reference pixels (the only
// loads of the iteration, three lanes, under two barriers) | runs of columns | wave 0: totals, the candidates from the tile, tail drop,
// lm_wave on its own lanes (no barrier between them) | next iteration.  Same arithmetic per point, same LM step, the sums in this
// kernel's own fixed order; no speculation (a retry is one more iteration of a few microseconds, not a launch).
// Measured (profiles/r06_notes.md section 21): the tile neither gains nor costs against gathering the taps from L2 / HBM — the iteration
// is bound by what one CU can issue — and the launch takes about what the ten rounds it replaces took, on n CUs instead of the chip.
#define LSD_SOLO_MAX_PX 4800
#define LSD_SOLO_TRIPS 9          // x 512 lanes = 4608 points >= the (w - 2)(h - 2) interior of any level of at most 4800 pixels
#define LSD_SOLO_MAX_PTS (LSD_SOLO_TRIPS * 512)
#define LSD_SOLO_LDS_PTS 4544     // (w - 2)(h - 2) <= w h - 4 sqrt(w h) + 4 <= 4527 interior pixels of a level of at most 4800
// Worth it from this many jobs per batch: one workgroup per job walks its coarse levels in about the time the lock-step rounds take, on
// n CUs instead of the chip — a gain where other work (the mapping stream of the S-sequence loop) wants the other CUs, a small loss for
// a few jobs that have the chip to themselves (profiles/r06_notes.md section 21).
#define LSD_SOLO_MIN_JOBS 24
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_track_solo(const TrackJob* __restrict__ jobs, TrackState* __restrict__ st2, TrackSummary* __restrict__ outs,
                                                          const int parity, const int doneWord) {
  const TrackJob& job = jobs[blockIdx.x];
  st2 += 2 * (size_t)blockIdx.x;
  TrackSummary* out = outs + blockIdx.x;
  constexpr int WAVES = BLOCK / 64;
  constexpr int HALF = BLOCK / 2;
  constexpr int CPP = RS_END;
  constexpr int RSLICE = BLOCK / CPP;
  constexpr int RRUN = (HALF + RSLICE - 1) / RSLICE;
  constexpr int TRIPS = LSD_SOLO_TRIPS;
  static_assert(WAVES * 3 <= 64, "top-3 merge in one wave");
  static_assert((size_t)TRIPS * BLOCK * 4 <= sizeof(float) * (CPP * (HALF + 1) + RRUN), "the level's list borrows the reduction buffer");
  __shared__ TrackState S;
  __shared__ LmShared sh;
  __shared__ LmPar s_par;
  __shared__ __attribute__((aligned(16))) float s_red[CPP * (HALF + 1) + RRUN];
  __shared__ float s_sum[RSLICE][64];
  __shared__ float s_sub[3][32];
  __shared__ int s_wtop[WAVES][3];
  __shared__ v4f s_tex[LSD_SOLO_MAX_PX];
  __shared__ float s_var[LSD_SOLO_LDS_PTS], s_img[LSD_SOLO_LDS_PTS];     // the points' variances and colours, list order (the registers hold the rest of a point)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  track_state_begin(S, job, tid);
  __syncthreads();
  int listLevel = -1, total = 0;
  EvalCtx a;
  // this lane's reference points of the level (trip r: point tid + r BLOCK of the list)
  unsigned pXY[TRIPS];
  float pZ[TRIPS];
  // the four taps of getInterpolatedElement43 (C/util/globalFuncs.h:63-77) out of the tile; `fetch` = false reads texel 0 instead
  auto taps = [&](const PointWarp& q, const bool fetch, PointTexels& t) {
    const int ix = fetch ? (int)q.u_new : 0;
    const int iy = fetch ? (int)q.v_new : 0;
    const v4f* bp = s_tex + (ix + __mul24(iy, a.w));
    const v4f v00 = bp[0], v10 = bp[1], v01 = bp[a.w], v11 = bp[a.w + 1];
    t.t00 = Texel3{v00.x, v00.y, v00.z}; t.t10 = Texel3{v10.x, v10.y, v10.z};
    t.t01 = Texel3{v01.x, v01.y, v01.z}; t.t11 = Texel3{v11.x, v11.y, v11.z};
  };
  for (int guard = 0; guard < 4096; guard++) {
    const int level = S.level;
    if (S.done) break;
    const TrackLevel& L = job.lv[level];
    const int work = L.w * L.h;
    if (L.tilePx <= 0 || work > LSD_SOLO_MAX_PX || L.writeMask) break;     // the larger levels (and the one that writes refPixelWasGood): lock-step rounds of strips
    if (level != listLevel) {
      make_ctx_dev(job, S, level, a);
      if (tid == BLOCK - 1) stage_lm_par(job, level, s_par, 1);
      const int mblk = (work + 255) >> 8;                   // <= 19
      const gbyte* offs = (const gbyte*)L.kf_refBlk;
      const __attribute__((address_space(1))) int* cnts = (const __attribute__((address_space(1))) int*)(offs + ((size_t)mblk << 8));
      const int cntv = lane < mblk ? cnts[lane] : 0;
      int incl = cntv + dpp_i0<0x111, 0xf, 0xf>(cntv);
      incl += dpp_i0<0x112, 0xf, 0xf>(cntv);
      incl += dpp_i0<0x113, 0xf, 0xf>(cntv);
      incl += dpp_i0<0x114, 0xf, 0xe>(incl);
      incl += dpp_i0<0x118, 0xf, 0xc>(incl);
      incl += dpp_i0<0x142, 0xa, 0xf>(incl);
      total = __builtin_amdgcn_readlane(incl, 31);
      if (total > LSD_SOLO_LDS_PTS) break;                  // (every wave computes the same total) cannot happen below 4800 pixels; strips if it does
      // the tile: the level's texel plane, 16 bytes per lane and step
      {
        gv4f* src = (gv4f*)a.fr_grad;
        for (int i = tid; i < work; i += BLOCK) s_tex[i] = src[i];
      }
      // the level's list (pixel order, as a strip of the lock-step rounds builds its own) in the reduction buffer
      unsigned* s_list = (unsigned*)s_red;
      const int waveU = __builtin_amdgcn_readfirstlane(wave);
      const float inv_w = 1.0f / (float)a.w;
      for (int blk = waveU; blk < mblk; blk += WAVES) {
        const int cb = __builtin_amdgcn_readlane(cntv, blk);
        const int pb = __builtin_amdgcn_readlane(incl, blk) - cb;
        const unsigned ow = *(const __attribute__((address_space(1))) unsigned*)(offs + ((size_t)blk << 8) + (lane << 2));
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int sl = lane + 64 * k;
          if (sl < cb) {
            const int i = (blk << 8) + (int)((ow >> (8 * k)) & 255u);
            int y = (int)((float)i * inv_w);
            int x = i - y * a.w;
            if (x < 0) { y--; x += a.w; }
            if (x >= a.w) { y++; x -= a.w; }
            s_list[pb + sl] = (unsigned)x | ((unsigned)y << 16);
          }
        }
      }
      __syncthreads();
      // this lane's points: everything of the keyframe an evaluation reads, once per level
      {
        float var[TRIPS], id[TRIPS], img[TRIPS];
#pragma unroll
        for (int r = 0; r < TRIPS; r++) {
          const int p = tid + r * BLOCK;
          const unsigned xy = s_list[p < total ? p : 0];
          const int i = __mul24((int)(xy >> 16), a.w) + (int)(xy & 0xffffu);
          pXY[r] = xy;
          var[r] = a.kf_idepthVar[i];
          id[r] = a.kf_idepth[i];
          img[r] = a.kf_image[i];
        }
#pragma unroll
        for (int r = 0; r < TRIPS; r++) {
          const int p = tid + r * BLOCK;
          if (p < total) { s_var[p] = var[r]; s_img[p] = img[r]; }
          pZ[r] = (p < total) ? lsd_rcp_exact(id[r]) * 1.0f : 1.0f;
        }
      }
      listLevel = level;
      __syncthreads();                                      // the list's words become the reduction buffer again
    } else {
      // the pose under evaluation (the LM step of the previous iteration left it in S)
#pragma unroll
      for (int i = 0; i < 9; i++) a.R[i] = S.R[i];
#pragma unroll
      for (int i = 0; i < 3; i++) a.t[i] = S.t[i];
      a.aff_a = S.aff_a; a.aff_b = S.aff_b;
    }
    // (the pose is the same in every lane: scalar registers — the lane's points and the 41 running sums want the vector ones)
    auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
#pragma unroll
    for (int i = 0; i < 9; i++) a.R[i] = uni(a.R[i]);
#pragma unroll
    for (int i = 0; i < 3; i++) a.t[i] = uni(a.t[i]);
    a.aff_a = uni(a.aff_a); a.aff_b = uni(a.aff_b);
    float acc[RS_END];
#pragma unroll
    for (int k = 0; k < RS_END; k++) acc[k] = 0.f;
    int key0 = -1, key1 = -1, key2 = -1;
    {
      const int rounds = (total + BLOCK - 1) / BLOCK;
      struct StT { PointWarp q; PointTexels t; };
      StT T0;
      auto stageB = [&](const int r, StT& T) {
        const bool live = tid + r * BLOCK < total;
        const int bx_ = (int)(pXY[r] & 0xffffu), by_ = (int)(pXY[r] >> 16);
        const float px = pZ[r] * (a.fxi * bx_ + a.cxi), py = pZ[r] * (a.fyi * by_ + a.cyi);
        eval_warp(a, px, py, pZ[r], T.q);
        taps(T.q, live && T.q.in_image, T.t);
      };
      auto stageC = [&](const int r, const StT& T) {
        if (tid + r * BLOCK < total) {
          acc[RS_NREF] += 1.f;
          if (T.q.in_image) {
            PointOut o;
            eval_finish(a, T.q, T.t, pZ[r], s_img[tid + r * BLOCK], s_var[tid + r * BLOCK], o);
            top3_insert(__mul24((int)(pXY[r] & 0xffffu), a.h) + (int)(pXY[r] >> 16), key0, key1, key2);   // reference order (x h + y, TrackingReference.cpp:128-138)
            accumulate_point(o, acc);
          }
        }
      };
#pragma unroll
      for (int r = 0; r < TRIPS; r++) {
        // (uniform branch: every lane sees the same `rounds`.  Requesting the taps of trip r + 1 before trip r is finished — two points in
        // flight — pushed the kernel over its 256 registers; the LDS round trip hides behind the SIMD's other wave)
        if (r < rounds) { stageB(r, T0); stageC(r, T0); }
      }
    }
    // each wave's three largest keys; the upper half parks its sums
    {
      int c0 = key0, c1 = key1, c2 = key2;
#pragma unroll
      for (int r = 0; r < 3; r++) {
        const int m = __builtin_amdgcn_readlane(wave_max_to_lane63(c0), 63);
        if (lane == 0) s_wtop[wave][r] = m;
        if (c0 == m && m >= 0) { c0 = c1; c1 = c2; c2 = -1; }
      }
    }
    if (tid >= HALF) {
#pragma unroll
      for (int k = 0; k < CPP; k++) s_red[k * (HALF + 1) + tid - HALF] = acc[k];
    }
    __syncthreads();
    // wave 0: the level's three largest keys = the candidates for the tail (the last M % 4 in-image points in reference order); lanes
    // 0..2 evaluate them once more — their reference pixel travels under the next two barriers, the taps come out of the tile
    int cKey = -1, cX = 0, cY = 0;
    float cVar = 0.f, cId = 1.f, cImg = 0.f;
    if (wave == 0) {
      int v = lane < WAVES * 3 ? s_wtop[lane / 3][lane - 3 * (lane / 3)] : -1;
      int top[3];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        top[r] = __builtin_amdgcn_readlane(wave_max_to_lane63(v), 63);
        if (v == top[r]) v = -1;
      }
      cKey = lane == 0 ? top[0] : (lane == 1 ? top[1] : (lane == 2 ? top[2] : -1));
      if (cKey >= 0) {
        cX = cKey / a.h;
        cY = cKey - cX * a.h;
        const int ci = __mul24(cY, a.w) + cX;
        cVar = a.kf_idepthVar[ci];
        cId = a.kf_idepth[ci];
        cImg = a.kf_image[ci];
      }
    }
    if (tid < HALF) {
#pragma unroll
      for (int k = 0; k < CPP; k++) s_red[k * (HALF + 1) + tid] = acc[k] + s_red[k * (HALF + 1) + tid];
    }
    __syncthreads();
    {
      const int slice = tid / CPP, k = tid - slice * CPP;
      if (slice < RSLICE) {
        const float* row = s_red + k * (HALF + 1);
        const int j0 = slice * RRUN;
        float sacc = 0.f;
#pragma unroll
        for (int j = 0; j < RRUN; j++) { const float v = row[j0 + j]; sacc += (j0 + j < HALF) ? v : 0.f; }
        s_sum[slice][k] = sacc;
      }
    }
    __syncthreads();
    if (wave == 0) {
      float sv = 0.f;
      if (lane < CPP) {
        sv = s_sum[0][lane];
#pragma unroll
        for (int sl = 1; sl < RSLICE; sl++) sv += s_sum[sl][lane];
      }
      if (cKey >= 0) {
        const float inv = lsd_rcp_exact(cId);                // (fetch_point's arithmetic)
        const float px = inv * (a.fxi * cX + a.cxi), py = inv * (a.fyi * cY + a.cyi);
        const float cPz = inv * 1.0f;
        PointWarp cq;
        PointTexels ct;
        eval_warp(a, px, py, cPz, cq);
        taps(cq, cq.in_image, ct);
        PointOut o;
        eval_finish(a, cq, ct, cPz, cImg, cVar, o);
        float* row = s_sub[lane];
        tail_row(o, row);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int M = (int)rl(sv, RS_M);
      const int nsub = (__builtin_amdgcn_readlane(cKey, 0) >= 0) + (__builtin_amdgcn_readlane(cKey, 1) >= 0) + (__builtin_amdgcn_readlane(cKey, 2) >= 0);
      int need = M & 3;
      if (need > nsub) need = nsub;
      // RS_WERR -> 0, RS_A0.. -> 1..21, RS_B0.. -> 22..27, RS_ERR -> 28
      const int subIdx = (lane == RS_WERR) ? 0 : ((lane >= RS_A0 && lane < RS_B0) ? 1 + lane - RS_A0 : ((lane >= RS_B0 && lane < RS_ERR) ? 22 + lane - RS_B0 : (lane == RS_ERR ? 28 : -1)));
      if (subIdx >= 0) {
        if (need > 0) sv -= s_sub[0][subIdx];
        if (need > 1) sv -= s_sub[1][subIdx];
        if (need > 2) sv -= s_sub[2][subIdx];
      }
      if (lane < RS_NUM) sh.tot[lane] = sv;
      lm_wave<false>(s_par, S, sv, sh.tot, lane, out, nullptr, 0, 0, doneWord);
    }
    __syncthreads();
  }
  // what the lock-step rounds load (they read st2[parity of their launch]; a finished job keeps both buffers "done")
  copy_words<sizeof(TrackState) / 4>(st2 + (1 - parity), &S, tid, BLOCK);
  if (S.done) copy_words<sizeof(TrackState) / 4>(st2 + parity, &S, tid, BLOCK);
}

// Pipelined contexts: the frame's refPixelWasGood is what the LAST trial the LM loop executed wrote.  Trial 0 of every launch writes the
// frame's own plane, trials > 0 write side planes — at the same pixels (the valid reference points do not depend on the pose), so the
// pixels the job visited are those whose byte in the frame's plane is no longer the 0xFF of frame creation: there the side plane's byte
// replaces it.  Queued on the mapping stream by lsdhip_tracker_track (one-stream contexts: the finishing launch copies instead).
__global__ __launch_bounds__(256) void k_mask_merge(uint32_t* __restrict__ plane, const uint32_t* __restrict__ side, int nwords) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nwords) return;
  const uint32_t a = plane[i], b = side[i];
  // per byte: visited (a != 0xFF) -> b, else a
  uint32_t m = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) if (((a >> (8 * k)) & 0xFFu) != 0xFFu) m |= 0xFFu << (8 * k);
  plane[i] = (a & ~m) | (b & m);
}

int lsd_flush_merges(lsdhip_ctx* c) {
  const int nwords = (int)((((size_t)c->wl[LSD_TRACK_MIN_LEVEL] * c->hl[LSD_TRACK_MIN_LEVEL]) + 3) / 4);
  for (const lsdhip_ctx::PendingMerge& m : c->pendingMerges) {
    hipLaunchKernelGGL(k_mask_merge, dim3((nwords + 255) / 256), dim3(256), 0, c->mstream, (uint32_t*)m.plane, (const uint32_t*)m.side, nwords);
    *m.doneSeq = c->mSeq + 1;      // complete at the mapping stream's next record point
  }
  c->pendingMerges.clear();
  HIPCHK(hipGetLastError());
  return LSDHIP_OK;
}

// checkPermaRefOverlap (SE3Tracker.cpp:121-157): usage only, explicit point list
__global__ __launch_bounds__(256) void k_overlap(const float* __restrict__ pos, int n, EvalCtx a, float* __restrict__ out) {
  __shared__ float s_w[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    float px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
    float Wx = ((a.R[0] * px + a.R[1] * py) + a.R[2] * pz) + a.t[0];
    float Wy = ((a.R[3] * px + a.R[4] * py) + a.R[5] * pz) + a.t[1];
    float Wz = ((a.R[6] * px + a.R[7] * py) + a.R[8] * pz) + a.t[2];
    float u_new = (Wx / Wz) * a.fx + a.cx;
    float v_new = (Wy / Wz) * a.fy + a.cy;
    if (u_new > 0 && v_new > 0 && u_new < a.w - 1 && v_new < a.h - 1) {
      float depthChange = pz / Wz;
      acc += depthChange < 1 ? depthChange : 1;
    }
  }
  float s = wave_sum_to_lane63(acc);
  if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
}

// ---------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------
extern "C" int lsdhip_tracker_create(lsdhip_ctx* c, lsdhip_tracker** out) {
  if (!c || !out) return LSDHIP_E_ARG;
  HIPCHK(hipSetDevice(c->device));
  lsdhip_tracker* t = new lsdhip_tracker();
  t->ctx = c;
  const int maxIterations[6] = {5, 20, 50, 100, 100, 100};
  for (int l = 0; l < LSD_LEVELS; l++) {
    t->lambdaInitial[l] = 0;
    t->stepSizeMin[l] = 1e-8;
    t->convergenceEps[l] = 0.999f;
    t->maxItsPerLvl[l] = maxIterations[l];
  }
  if (const char* e = getenv("LSDHIP_TRACK_BLOCK")) t->block = atoi(e);
  if (const char* e = getenv("LSDHIP_TRACK_CAP")) t->grid_cap = atoi(e);
  if (t->block != 256) { lsd_set_error("LSDHIP_TRACK_BLOCK must be 256"); delete t; return LSDHIP_E_ARG; }
  if (t->grid_cap < 8) t->grid_cap = 8;
  t->grid_cap &= ~7;
  {
    // rows per column slice must fit the 20 float4 loads of the column-sum phase: 80 rows x NSLICE slices
    const int nslice = ((t->block / 64 - 1) * 64) / RS_END;
    if (t->grid_cap > 80 * nslice) t->grid_cap = (80 * nslice) & ~7;
  }
  t->max_blocks = t->grid_cap;
  const size_t rows = (size_t)t->max_blocks;
  const size_t scratch_bytes = (size_t)LSD_SPEC_MAX * (2 * RS_COLS * rows * 4 + 2 * rows * 16 + 2 * rows * 96 * 4 + 2 * 32 * 4);
  if (const char* e = getenv("LSDHIP_SPEC")) t->specC = atoi(e);
  if (t->specC < 1) t->specC = 1;
  if (t->specC > LSD_SPEC_MAX) t->specC = LSD_SPEC_MAX;
  if (const char* e = getenv("LSDHIP_SPEC_CAP")) t->specCap = atoi(e) & ~7;
  if (const char* e = getenv("LSDHIP_SPEC_CAPS")) {
    int v[LSD_LEVELS] = {0, 0, 0, 0, 0};
    sscanf(e, "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]);
    for (int l = 0; l < LSD_LEVELS; l++) t->specCaps[l] = v[l] < 0 ? 0 : (v[l] & ~7);
  }
  if (const char* e = getenv("LSDHIP_SPEC_LEVELS")) {
    int v[LSD_LEVELS] = {0, 0, 0, 0, 0};
    sscanf(e, "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]);
    for (int l = 0; l < LSD_LEVELS; l++) t->specLevel[l] = v[l] < 0 ? 0 : (v[l] > LSD_SPEC_MAX ? LSD_SPEC_MAX : v[l]);
  }
  t->maskStride = (((size_t)c->wl[LSD_TRACK_MIN_LEVEL] * c->hl[LSD_TRACK_MIN_LEVEL]) + 255) & ~(size_t)255;
  HIPCHK(hipMalloc((void**)&t->d_maskSide, 2 * t->maskStride * (LSD_SPEC_MAX - 1)));
  HIPCHK(hipMemsetAsync(t->d_maskSide, 0, 2 * t->maskStride * (LSD_SPEC_MAX - 1), c->stream));
  HIPCHK(hipMalloc((void**)&t->d_partials, scratch_bytes));
  HIPCHK(hipMemsetAsync(t->d_partials, 0, scratch_bytes, c->stream));
#ifdef LSD_ORDER_CHECK
  {
    const size_t nd = 8 + 2 * 8192;
    HIPCHK(hipMalloc((void**)&t->d_dbg, nd * 8));
    std::vector<unsigned long long> init(nd, 0);
    for (size_t i = 8; i < nd; i += 2) init[i] = ~0ull;     // (min, max) pairs of the per-launch totals hash
    HIPCHK(hipMemcpy(t->d_dbg, init.data(), nd * 8, hipMemcpyHostToDevice));
  }
#endif
#ifdef LSD_DEVTOOLS
  HIPCHK(hipMalloc((void**)&t->d_log, 4096 * 16 * 4));
  HIPCHK(hipMemset(t->d_log, 0, 4096 * 16 * 4));
#endif
  HIPCHK(hipMalloc((void**)&t->d_state, 2 * sizeof(TrackState)));
  HIPCHK(hipMemsetAsync(t->d_state, 0, 2 * sizeof(TrackState), c->stream));
#ifdef LSD_PHASE_TRACE
  HIPCHK(hipMalloc((void**)&t->d_trace, (1 + 4096 * LSD_TRACE_WORDS) * 8));
  HIPCHK(hipMemsetAsync(t->d_trace, 0, (1 + 4096 * LSD_TRACE_WORDS) * 8, c->stream));
#endif
  HIPCHK(hipHostMalloc((void**)&t->h_summary, sizeof(TrackSummary), hipHostMallocMapped));
  HIPCHK(hipHostGetDevicePointer((void**)&t->d_summary, t->h_summary, 0));
  memset(t->h_summary, 0, sizeof(TrackSummary));
  const char* env = getenv("LSDHIP_HOST_LM");
  t->hostLM = env && env[0] == '1';
  if (const char* e = getenv("LSDHIP_SPIN")) t->spinWait = e[0] != '0';
  if (const char* e = getenv("LSDHIP_BUDGET_EXTRA")) { t->budgetExtra = atoi(e); if (t->budgetExtra < 1) t->budgetExtra = 1; }
  if (const char* e = getenv("LSDHIP_BUDGET_FIXED")) { t->budgetFixed = atoi(e); if (t->budgetFixed < 1) t->budgetFixed = 0; }   // test hook: every job runs out of budget
  *out = t;
  return LSDHIP_OK;
}
extern "C" void lsdhip_tracker_destroy(lsdhip_tracker* t) {
  if (!t) return;
  {
    lsdhip_ctx* c = t->ctx;
    LSD_CTX_LOCK(c);
    for (size_t i = 0; i < c->pendingMerges.size();)     // merges out of this tracker's side planes die with it
      if (c->pendingMerges[i].doneSeq == &t->maskMergeSeq[0] || c->pendingMerges[i].doneSeq == &t->maskMergeSeq[1]) c->pendingMerges.erase(c->pendingMerges.begin() + i);
      else i++;
  }
#ifdef LSD_DEVTOOLS
  if (getenv("LSDHIP_TRACK_DEBUG") && t->dbgJobs > 0)
    fprintf(stderr, "TRACKDBG jobs %lld launches enqueued %.2f/job, budget misses %lld, host launch %.1f us/job, host wait %.1f us/job\n", t->dbgJobs,
            (double)t->dbgEnqueued / t->dbgJobs, t->dbgMisses, t->dbgLaunchNs / 1e3 / t->dbgJobs, t->dbgWaitNs / 1e3 / t->dbgJobs);
#endif
  (void)hipSetDevice(t->ctx->device);
  (void)hipStreamSynchronize(t->ctx->stream);
#ifdef LSD_ORDER_CHECK
  {
    std::vector<unsigned long long> h(8 + 2 * 8192, 0);
    (void)hipMemcpy(h.data(), t->d_dbg, h.size() * 8, hipMemcpyDeviceToHost);
    int used = 0, split = 0;
    for (size_t i = 8; i < h.size(); i += 2) if (h[i] != ~0ull) { used++; if (h[i] != h[i + 1]) split++; }
    if (const char* path = getenv("LSDHIP_ORDER_DUMP")) {
      if (FILE* f = fopen(path, "w")) {
        for (size_t i = 8; i < h.size(); i += 2) if (h[i] != ~0ull) fprintf(f, "%zu %llx %llx\n", (i - 8) / 2, h[i], h[i + 1]);
        fclose(f);
      }
    }
    fprintf(stderr, "ORDERCHECK: %llu workgroups finished (%llu launched), %llu started before every earlier workgroup had finished (largest deficit %llu); "
                    "%d launch slots hashed, %d in which the workgroups did NOT all see the same totals\n", h[0], t->dbgCum, h[1], h[2], used, split);
  }
#endif
  if (t->d_dbg) (void)hipFree(t->d_dbg);
#ifdef LSD_PHASE_TRACE
  if (const char* path = getenv("LSDHIP_TRACE_FILE")) {
    std::vector<unsigned long long> h(1 + 4096 * LSD_TRACE_WORDS);
    if (hipMemcpy(h.data(), t->d_trace, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
      if (FILE* f = fopen(path, "w")) {
        unsigned long long n = h[0] < 4096 ? h[0] : 4096;
        for (unsigned long long i = 0; i < n; i++) {
          for (int k = 0; k < LSD_TRACE_WORDS; k++) fprintf(f, "%llu ", h[1 + i * LSD_TRACE_WORDS + k]);
          fprintf(f, "\n");
        }
        fclose(f);
      }
    }
  }
  (void)hipFree(t->d_trace);
#endif
  if (t->d_bjobs) { (void)hipFree(t->d_bjobs); (void)hipFree(t->d_bstate); (void)hipFree(t->d_bscratch); (void)hipHostFree(t->h_bjobs); (void)hipHostFree(t->h_bsummary); }
  (void)hipFree(t->d_partials);
  (void)hipFree(t->d_maskSide);
  (void)hipFree(t->d_state);
#ifdef LSD_DEVTOOLS
  (void)hipFree(t->d_log);
#endif
  (void)hipHostFree(t->h_summary);
  if (t->d_pts) (void)hipFree(t->d_pts);
  delete t;
}
extern "C" int lsdhip_tracker_set_enqueue_hook(lsdhip_tracker* t, lsdhip_enqueue_hook fn, void* user) {
  if (!t) return LSDHIP_E_ARG;
  t->enqueueHook = fn;
  t->enqueueHookUser = user;
  return LSDHIP_OK;
}
extern "C" int lsdhip_tracker_get_settings(const lsdhip_tracker* t, lsdhip_tracker_settings* o) {
  if (!t || !o) return LSDHIP_E_ARG;
  o->lambdaSuccessFac = t->lambdaSuccessFac; o->lambdaFailFac = t->lambdaFailFac;
  for (int l = 0; l < LSD_LEVELS; l++) {
    o->lambdaInitial[l] = t->lambdaInitial[l]; o->stepSizeMin[l] = t->stepSizeMin[l]; o->convergenceEps[l] = t->convergenceEps[l];
    o->maxItsPerLvl[l] = t->maxItsPerLvl[l];
  }
  o->lambdaInitialTestTrack = t->lambdaInitialTestTrack; o->stepSizeMinTestTrack = t->stepSizeMinTestTrack;
  o->convergenceEpsTestTrack = t->convergenceEpsTestTrack; o->maxItsTestTrack = t->maxItsTestTrack;
  o->huber_d = t->huber_d; o->var_weight = t->var_weight;
  return LSDHIP_OK;
}
extern "C" int lsdhip_tracker_set_settings(lsdhip_tracker* t, const lsdhip_tracker_settings* in) {
  if (!t || !in) return LSDHIP_E_ARG;
  t->lambdaSuccessFac = in->lambdaSuccessFac; t->lambdaFailFac = in->lambdaFailFac;
  for (int l = 0; l < LSD_LEVELS; l++) {
    t->lambdaInitial[l] = in->lambdaInitial[l]; t->stepSizeMin[l] = in->stepSizeMin[l]; t->convergenceEps[l] = in->convergenceEps[l];
    t->maxItsPerLvl[l] = in->maxItsPerLvl[l];
  }
  t->lambdaInitialTestTrack = in->lambdaInitialTestTrack; t->stepSizeMinTestTrack = in->stepSizeMinTestTrack;
  t->convergenceEpsTestTrack = in->convergenceEpsTestTrack; t->maxItsTestTrack = in->maxItsTestTrack;
  t->huber_d = in->huber_d; t->var_weight = in->var_weight;
  return LSDHIP_OK;
}
extern "C" int lsdhip_tracker_exec_stats(const lsdhip_tracker* t, int out[8]) {
  if (!t || !out) return LSDHIP_E_ARG;
  out[0] = 0; out[1] = 0; out[2] = 0;   // (were: cluster-kernel job counts — the kernel was removed in round 4, profiles/r03_notes.md §3)
  for (int l = 0; l < LSD_LEVELS; l++) out[3 + l] = t->levelEvaluations[l];
  return LSDHIP_OK;
}
extern "C" int lsdhip_tracker_set_speculation(lsdhip_tracker* t, int trials, int finestLevelWorkgroups) {
  if (!t || trials < 1 || trials > LSD_SPEC_MAX || finestLevelWorkgroups < 0) { lsd_set_error("lsdhip_tracker_set_speculation: trials must be 1..%d", LSD_SPEC_MAX); return LSDHIP_E_ARG; }
  LSD_CTX_LOCK(t->ctx);
  t->specC = trials;
  t->specAuto = false;                                        // the same number of trials at every level
  if (finestLevelWorkgroups > 0) t->specCap = finestLevelWorkgroups < 8 ? 8 : (finestLevelWorkgroups & ~7);   // multiples of 8 (one tile band per XCD), at least 8
  for (int l = 0; l < LSD_LEVELS; l++) t->specLevel[l] = 0;
  for (int i = 0; i < 4; i++) t->recent[i] = 0;
  return LSDHIP_OK;
}
extern "C" int lsdhip_tracker_set_batch_coarse_min_jobs(lsdhip_tracker* t, int minJobs) {
  if (!t || minJobs < 0) { lsd_set_error("lsdhip_tracker_set_batch_coarse_min_jobs: minJobs must be >= 0"); return LSDHIP_E_ARG; }
  LSD_CTX_LOCK(t->ctx);
  t->soloMinJobs = minJobs;
  return LSDHIP_OK;
}
#ifdef LSD_DEVTOOLS
// developer build: the launch log of the last job (LSDHIP_LAUNCH_LOG=1): 16 ints per queued launch, slot = launch ordinal (1-based)
extern "C" int lsdhip_tracker_debug_log(const lsdhip_tracker* t, int* out, int maxInts) {
  if (!t || !out) return -1;
  const int n = (int)t->lastLog.size() < maxInts ? (int)t->lastLog.size() : maxInts;
  memcpy(out, t->lastLog.data(), (size_t)n * 4);
  return n;
}
#endif
extern "C" int lsdhip_tracker_launch_stats(const lsdhip_tracker* t, int out[2]) {
  if (!t || !out) return LSDHIP_E_ARG;
  out[0] = t->numLaunches; out[1] = t->specC;
  return LSDHIP_OK;
}
extern "C" int lsdhip_tracker_step_stats(const lsdhip_tracker* t, int out[4]) {
  if (!t || !out) return LSDHIP_E_ARG;
  out[0] = t->numLaunches; out[1] = 0; out[2] = 0; out[3] = t->specC;
  return LSDHIP_OK;
}
extern "C" int lsdhip_tracker_set_max_its(lsdhip_tracker* t, const int its[LSD_LEVELS]) {
  if (!t || !its) return LSDHIP_E_ARG;
  for (int l = 0; l < LSD_LEVELS; l++) t->maxItsPerLvl[l] = its[l];
  return LSDHIP_OK;
}

static const float MIN_GOODPERGOODBAD_PIXEL = 0.5f;
static const float MIN_GOODPERALL_PIXEL = 0.04f;
static const float MIN_GOODPERALL_PIXEL_ABSMIN = 0.01f;

static void fill_level(lsdhip_tracker* t, TrackJob& job, int level, lsdhip_frame* kf, lsdhip_frame* frame, const float* pts_pos,
                       const float* pts_colvar, int npts) {
  lsdhip_ctx* c = t->ctx;
  LSD_CTX_LOCK(c);
  TrackLevel& L = job.lv[level];
  const LevelIntr& in = c->intr[level];
  L.w = c->wl[level]; L.h = c->hl[level];
  L.fx = in.fx; L.fy = in.fy; L.cx = in.cx; L.cy = in.cy; L.fxi = in.fxi; L.fyi = in.fyi; L.cxi = in.cxi; L.cyi = in.cyi;
  L.fr_grad = frame->d_grad[level];
  if (level == 0) (void)lsd_frame_require_level0_for_tracking(frame);   // (levels >= 1 are what a tracked frame has: the level-0 texels on demand)
  if (npts >= 0) {
    L.pts_pos = pts_pos; L.pts_colvar = pts_colvar; L.npts = npts;
    L.kf_idepth = L.kf_idepthVar = L.kf_image = nullptr;
    L.kf_refBlk = nullptr;
  } else {
    L.kf_idepth = kf->d_idepth[level]; L.kf_idepthVar = kf->d_idepthVar[level]; L.kf_image = kf->d_image[level];
    L.kf_refBlk = kf->d_refBlk[level];
    L.pts_pos = L.pts_colvar = nullptr; L.npts = -1;
  }
  int work = npts >= 0 ? npts : L.w * L.h;
  L.nblocks = (work + t->block - 1) / t->block;
  if (L.nblocks >= 16) L.nblocks = (L.nblocks + 7) & ~7;   // multiples of 8: one contiguous band of tiles per XCD
  const int cap = t->cap_override > 0 ? t->cap_override : t->grid_cap;
  if (L.nblocks > cap) L.nblocks = cap;                     // larger levels grid-stride
  if (L.nblocks < 1) L.nblocks = 1;
  L.singlePass = (long long)L.nblocks * t->block >= work ? 1 : 0;
  L.tilePx = 0;
  if (t->batch_jobs >= LSD_BATCH_THROUGHPUT_MIN_JOBS && npts < 0 && L.kf_refBlk != nullptr) {   // (levels >= 1: level 0 has no reference blocks)
    // throughput mode: strips of tilePx pixels, compacted in the workgroup; enough strips over all jobs to fill the chip
    static const int wgTarget = getenv("LSDHIP_BATCH_WGS") ? atoi(getenv("LSDHIP_BATCH_WGS")) : LSD_BATCH_STRIP_WORKGROUPS;   // developer sweep
    // strips x jobs = the chip's 768 workgroup slots (3 per CU) where the level is large enough: one full round of equal strips;
    // a strip is a multiple of 256 pixels (the lanes take 4 consecutive pixels each)
    long long px = (((long long)work * t->batch_jobs + wgTarget - 1) / wgTarget + 255) & ~255LL;
    if (px < 1024) px = 1024;
    if (px > 8192) px = 8192;                                  // the strip's list lives in the reduction's LDS (10545 words)
    if ((work + px - 1) / px <= t->max_blocks) {               // (levels beyond 2.4 Mpixel keep the grid-stride form)
      L.tilePx = (int)px;
      L.nblocks = (int)((work + px - 1) / px);
      L.singlePass = 0;
    }
  }
  L.lambdaInitial = t->lambdaInitial[level]; L.stepSizeMin = t->stepSizeMin[level]; L.convergenceEps = t->convergenceEps[level];
  L.maxIts = t->maxItsPerLvl[level];
  L.minWarped = MIN_GOODPERALL_PIXEL_ABSMIN * (c->w >> level) * (c->h >> level);
  L.writeMask = 0;
}
static void fill_job_common(lsdhip_tracker* t, TrackJob& job) {
  lsdhip_ctx* c = t->ctx;
  LSD_CTX_LOCK(c);
  memset(&job, 0, sizeof(job));
  job.cameraPixelNoise2 = c->params.cameraPixelNoise2;
  job.var_weight = t->var_weight;
  job.huber_half = t->huber_d / 2;
  job.lambdaSuccessFac = t->lambdaSuccessFac;
  job.lambdaFailFac = t->lambdaFailFac;
  job.useAffine = c->params.useAffineLightningEstimation;
}

static TrackScratch scratch_of(lsdhip_tracker* t) {
  TrackScratch sc;
  const size_t rows = (size_t)t->max_blocks, C = LSD_SPEC_MAX;
  sc.sums = t->d_partials;
  sc.topkey = (int4*)(t->d_partials + C * 2 * RS_COLS * rows);
  sc.topval = t->d_partials + C * 2 * RS_COLS * rows + C * 2 * 4 * rows;
  sc.recs = t->d_partials + C * 2 * RS_COLS * rows + C * 2 * 4 * rows + C * 2 * 96 * rows;
  sc.max_rows = t->max_blocks;
  sc.cmax = LSD_SPEC_MAX;
#ifdef LSD_PHASE_TRACE
  sc.trace = t->d_trace;
#endif
  return sc;
}
static void launch_step(lsdhip_tracker* t, const TrackJob& job, int grid, int parity, int first) {
  lsdhip_ctx* c = t->ctx;
  LSD_CTX_LOCK(c);
  TrackScratch sc = scratch_of(t);
  t->launchOrdinal++;
  t->spec.seq = t->jobTag ? ((t->jobTag << 12) | (t->launchOrdinal & 0xFFF)) : 0;
#ifdef LSD_DEVTOOLS
  t->spec.dbgLog = t->d_log;
#endif
#ifdef LSD_ORDER_CHECK
  t->spec.dbgCum = t->dbgCum;
  t->spec.dbgCounters = t->d_dbg;
  t->dbgCum += (unsigned long long)grid;
#endif
  hipLaunchKernelGGL((k_track_step<256, false>), dim3(grid), dim3(256), 0, c->stream, job, (const TrackJob*)nullptr, t->d_state, sc,
                     t->d_summary, parity, first, t->spec);
}
// launch `steps` fused k_track_step kernels (alternating parity); grid = the largest level the job can still visit.
static int launch_steps(lsdhip_tracker* t, TrackJob& job, int steps, int* parity, int* first) {
  int grid = 1;
  for (int l = job.lastLevel; l <= job.topLevel; l++) if (job.lv[l].nblocks > grid) grid = job.lv[l].nblocks;
  t->spec.specGrid = 0;
  if (t->spec.specC > 1) {
    // workgroup = (trial, tile) of the level being evaluated: the launch needs the most tiles x trials of any level
    for (int l = job.lastLevel; l <= job.topLevel; l++) {
      const int g = job.lv[l].nblocks * (t->spec.trials[l] > 1 ? t->spec.trials[l] : 1);
      if (g > grid) grid = g;
    }
    t->spec.specGrid = grid;
  }
#ifdef LSD_DEVTOOLS
  const auto tl0 = std::chrono::steady_clock::now();
#endif
  for (int i = 0; i < steps; i++) {
    t->spec.last = (i + 1 == steps) ? 1 : 0;
    launch_step(t, job, grid, *parity, *first);
    lsdhip_host_mark(20);
    *first = 0;
    *parity ^= 1;
  }
#ifdef LSD_DEVTOOLS
  t->dbgEnqueued += steps;
  t->dbgLaunchNs += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tl0).count();
#endif
  HIPCHK(hipGetLastError());
  return LSDHIP_OK;
}

static int prof_collect(lsdhip_ctx* c);
struct EvalOut {       // what one evaluation leaves behind, in the reference's terms
  int warped_size;
  float retval;        // calcResidualAndBuffers return value
  float weightedError; // calcWeightsAndResidualSSE return value
  float A[36], b[6], lsError;
  double num_constraints;
};

// one evaluation with a host round trip (evalOnly job): kernel-level parity hook and host-LM debugging path
static int evaluate_pose(lsdhip_tracker* t, TrackJob& job, const lsdm::SE3fH& T, int level, EvalOut* eo) {
  lsdhip_ctx* c = t->ctx;
  LSD_CTX_LOCK(c);
  job.evalOnly = 1;
  job.lastLevel = level;
  job.topLevel = level;
  job.T0 = T;
  job.aff_a0 = t->affineEstimation_a; job.aff_b0 = t->affineEstimation_b;
  t->h_summary->done = 0;
  t->spec = TrackSpec{};                              // one evaluation, one trial
  t->jobTag = 0;
  if (int rcp = prof_collect(c)) return rcp;
  if (c->prof_on) HIPCHK(hipEventRecord(c->ev_a, c->stream));
  t->spec.last = 0;
  launch_step(t, job, job.lv[level].nblocks, 0, 1);   // residual evaluation
  if (c->prof_on) HIPCHK(hipEventRecord(c->ev_b, c->stream));
  launch_step(t, job, 1, 1, 0);                       // finalises the sums (evalOnly)
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  const TrackSummary* S = t->h_summary;
  const float* r = S->sums;
  if (c->prof_on) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev_a, c->ev_b));
    c->prof_ms += ms;
    c->prof_launches++;
    c->prof_bytes += S->bytes;
  }
  t->numEvaluations++;
  t->pointUsage = S->pointUsage; t->lastGoodCount = S->goodCount; t->lastBadCount = S->badCount; t->lastMeanRes = S->meanRes;
  t->affineEstimation_a_lastIt = S->aff_a_lastIt; t->affineEstimation_b_lastIt = S->aff_b_lastIt;
  int M = (int)r[RS_M];
  eo->warped_size = M;
  eo->retval = r[RS_SUMRES2] / r[RS_GOOD];
  eo->weightedError = r[RS_WERR] / ((M >> 2) << 2);
  size_t num_constraints = (size_t)6 * (size_t)(M >> 2);
  float n = (float)num_constraints;
  int k = RS_A0;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++, k++) {
      float v = (0.0f + r[k]) / n;
      eo->A[i * 6 + j] = v;
      eo->A[j * 6 + i] = v;
    }
  for (int i = 0; i < 6; i++) eo->b[i] = (0.0f - r[RS_B0 + i]) / n;
  eo->lsError = (0.0f + r[RS_ERR]) / n;
  eo->num_constraints = (double)num_constraints;
  return LSDHIP_OK;
}

// host-driven LM for one level (debugging path): SE3Tracker.cpp:323-449
static int lm_level_host(lsdhip_tracker* t, TrackJob& job, int lvl, lsdm::SE3fH& referenceToFrame, float* lastResidualOut) {
  lsdhip_ctx* c = t->ctx;
  LSD_CTX_LOCK(c);
  const TrackLevel& L = job.lv[lvl];
  EvalOut ev;
  int rc = evaluate_pose(t, job, referenceToFrame, lvl, &ev);
  if (rc) return rc;
  if (ev.warped_size < L.minWarped) return LSDHIP_DIVERGED;
  if (c->params.useAffineLightningEstimation) {
    t->affineEstimation_a = t->affineEstimation_a_lastIt;
    t->affineEstimation_b = t->affineEstimation_b_lastIt;
  }
  float lastErr = ev.weightedError;
  float LM_lambda = L.lambdaInitial;
  EvalOut cur = ev;
  for (int iteration = 0; iteration < L.maxIts; iteration++) {
    t->numWarpUpdates++;
    int incTry = 0;
    while (true) {
      float b[6], A[36], inc[6];
      for (int i = 0; i < 6; i++) b[i] = -cur.b[i];
      memcpy(A, cur.A, sizeof(A));
      for (int i = 0; i < 6; i++) A[i * 6 + i] *= 1 + LM_lambda;
      lsdm::ldlt6_solve(A, b, inc);
      incTry++;
      lsdm::SE3fH new_referenceToFrame = lsdm::se3f_mul(lsdm::se3f_exp(inc), referenceToFrame);
      EvalOut nev;
      rc = evaluate_pose(t, job, new_referenceToFrame, lvl, &nev);
      if (rc) return rc;
      if (nev.warped_size < L.minWarped) return LSDHIP_DIVERGED;
      float error = nev.weightedError;
      if (error < lastErr) {
        referenceToFrame = new_referenceToFrame;
        cur = nev;
        if (c->params.useAffineLightningEstimation) {
          t->affineEstimation_a = t->affineEstimation_a_lastIt;
          t->affineEstimation_b = t->affineEstimation_b_lastIt;
        }
        if (error / lastErr > L.convergenceEps) iteration = L.maxIts;
        lastErr = error;
        if (job.trackFrameSemantics) *lastResidualOut = error;
        if (LM_lambda <= 0.2) LM_lambda = 0;
        else LM_lambda *= t->lambdaSuccessFac;
        break;
      } else {
        float incdot = (inc[0] * inc[0] + (inc[1] * inc[1] + inc[2] * inc[2])) + (inc[3] * inc[3] + (inc[4] * inc[4] + inc[5] * inc[5]));
        if (!(incdot > L.stepSizeMin)) { iteration = L.maxIts; break; }
        if (LM_lambda == 0) LM_lambda = 0.2;
        else LM_lambda *= std::pow(t->lambdaFailFac, incTry);
      }
    }
  }
  if (!job.trackFrameSemantics) *lastResidualOut = lastErr;
  return LSDHIP_OK;
}

// elapsed time of the last profiled launch batch (events are read one call late so that nothing waits for them)
static int prof_collect(lsdhip_ctx* c) {
  if (!c->prof_pending) return LSDHIP_OK;
  HIPCHK(hipEventSynchronize(c->ev_b));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, c->ev_a, c->ev_b));
  c->prof_ms += ms;
  c->prof_pending = false;
  return LSDHIP_OK;
}
int lsd_prof_collect(lsdhip_ctx* c) { return prof_collect(c); }

// A polled summary is accepted when its words add up to its `check` word: `done` has arrived, but the record's other words are separate
// posted writes and — one job in a few thousand, measured (profiles/r06_notes.md section 1) — the ones stored last (numLaunches, lastCand,
// levelEvals) still held the previous job's values at that moment.  Spins until the record is whole; counts what it saw.
static int summary_wait_consistent(lsdhip_tracker* t, const int doneWord, const TrackSummary* rec = nullptr) {
  if (!rec) rec = t->h_summary;
  volatile const unsigned* w = (volatile const unsigned*)rec;
  t->sumPolled++;
  unsigned first[LSD_SUMMARY_CHECK_WORDS];
  std::chrono::steady_clock::time_point t0;
  for (unsigned spins = 0;; spins++) {
    unsigned cur[LSD_SUMMARY_CHECK_WORDS];
    unsigned chk = (unsigned)doneWord;
    for (unsigned i = 1; i < LSD_SUMMARY_CHECK_WORDS; i++) { cur[i] = w[i]; chk += lsd_summary_term(i, cur[i]); }
    const unsigned want = *(volatile const unsigned*)&rec->check;
    if (chk == want) {
      if (spins > 0) {
        const long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        if (ns > t->sumLateMaxNs) t->sumLateMaxNs = ns;
        for (unsigned i = 1; i < LSD_SUMMARY_CHECK_WORDS; i++)
          if (first[i] != cur[i]) {
            t->sumLateWords++;
            if (t->sumLateFirstWord < 0 || (int)i < t->sumLateFirstWord) t->sumLateFirstWord = (int)i;
            if ((int)i > t->sumLateLastWord) t->sumLateLastWord = (int)i;
          }
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      return LSDHIP_OK;
    }
    if (spins == 0) {
      t->sumLate++;
      t0 = std::chrono::steady_clock::now();
      memcpy(first, cur, sizeof(first));
    }
    if ((spins & 0xFFFFu) == 0xFFFFu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
      lsd_set_error("tracking summary in pinned memory never became consistent (check %08x, sum %08x)", want, chk);
      return LSDHIP_E_STATE;
    }
    __builtin_ia32_pause();
  }
}
extern "C" int lsdhip_tracker_summary_stats(const lsdhip_tracker* t, long long out[6]) {
  if (!t || !out) return LSDHIP_E_ARG;
  out[0] = t->sumPolled; out[1] = t->sumLate; out[2] = t->sumLateMaxNs; out[3] = t->sumLateWords; out[4] = t->sumLateFirstWord; out[5] = t->sumLateLastWord;
  return LSDHIP_OK;
}

// device-resident LM over levels topLevel..job.lastLevel; one host synchronisation per budget of launches.  The budget
// is the previous job's launch count (evaluations + the finalising step) plus a margin, so that few steps run empty.
// With profiling on, the whole budget is bracketed by one HIP event pair on the context's stream and charged to the
// launches that did work.
static int track_device(lsdhip_tracker* t, TrackJob& job, int topLevel, const lsdm::SE3fH& T0, lsdm::SE3fH* Tout) {
  lsdhip_ctx* c = t->ctx;
  // (the caller — an extern "C" entry — holds the context mutex once; it is released below while the host waits)
  job.evalOnly = 0;
  job.topLevel = topLevel;
  job.T0 = T0;
  job.aff_a0 = 1.0f; job.aff_b0 = 0.0f;
  t->spec.specC = t->specC;
  t->spec.specGrid = 0;
  // two sets of side planes, alternating by job: on a pipelined context the merge of job t's final mask runs on the mapping stream
  // while job t + 1 writes the other set
  t->maskSet ^= 1;
  t->spec.wasGoodSide = t->d_maskSide + (size_t)t->maskSet * t->maskStride * (LSD_SPEC_MAX - 1);
  t->spec.maskStride = (unsigned)t->maskStride;
  t->spec.copyMask = c->pipeline ? 0 : 1;
  if (c->pipeline && t->maskMergeSeq[t->maskSet] != 0) {
    // the merge that reads this set (noted two jobs ago) must have run before the set is overwritten (in a frame loop a mapping
    // operation has queued it long ago and the job's own frame was created behind it)
    if (t->maskMergeSeq[t->maskSet] < 0) { if (int rcf = lsd_flush_merges(c)) return rcf; }
    if (t->maskMergeSeq[t->maskSet] > c->mSeq && lsd_m_record(c) < 0) return LSDHIP_E_HIP;
    if (int rcw = lsd_t_wait_m(c, t->maskMergeSeq[t->maskSet])) return rcw;
    t->maskMergeSeq[t->maskSet] = 0;
  }
  // Trials per launch and workgroups per trial, per level.  Speculation pays where a level is latency-bound, i.e. small: the
  // automatic policy goes by the level's pixel (or point) count — <= 6 K: 6 trials, <= 24 K: 5, <= 88 K: 5 trials on specCap (80)
  // workgroups each (multi-pass; 4 x 104 measured 1.5 % slower), larger: one evaluation per launch on the full grid (such levels are work-bound: at
  // 1280x1024 speculating on level 1 cost 15 % of the frame rate).  An explicit lsdhip_tracker_set_speculation / the
  // LSDHIP_SPEC_LEVELS / _CAPS environment overrides it.
  for (int l = 0; l < LSD_LEVELS; l++) t->spec.trials[l] = 1;
  if (t->specC > 1) {
    for (int l = job.lastLevel; l <= topLevel; l++) {
      TrackLevel& L = job.lv[l];
      const long long work = L.npts >= 0 ? L.npts : (long long)L.w * L.h;
      int trials, cap = t->specCaps[l];
      if (t->specLevel[l] > 0) trials = t->specLevel[l];
      else if (!t->specAuto) trials = t->specC;
      else trials = work <= LSD_SPEC_SMALL_PX ? LSD_SPEC_TRIALS_SMALL : (work <= LSD_SPEC_MID_PX ? LSD_SPEC_TRIALS_MID : 1);
      if (trials > t->specC) trials = t->specC;
      if (cap <= 0 && trials > 1 && (t->specAuto ? work > LSD_SPEC_CAP_ABOVE_PX : l == job.lastLevel)) cap = t->specCap > 0 ? t->specCap : ((t->grid_cap / 2 + 7) & ~7);
      t->spec.trials[l] = trials;
      if (trials > 1 && cap > 0 && L.nblocks > cap && L.tilePx == 0) {
        L.nblocks = cap;
        L.singlePass = (long long)L.nblocks * t->block >= work ? 1 : 0;
      }
    }
  }
  t->h_summary->done = 0;
  const TrackSummary* S = t->h_summary;
  const long long myEpoch = ++c->enqEpoch;   // everything enqueued on the stream so far precedes this job
  // Launches of the k_track_step chain a job needs = its evaluating launches + the finalising step; budget = the most of the
  // recent jobs + 2 (launches queued behind the finishing one leave at once, ~4 us each).
  int budget = 12;
  if (t->recent[0] > 0) {
    budget = 0;
    for (int i = 0; i < 4; i++) if (t->recent[i] > budget) budget = t->recent[i];
    budget += t->budgetExtra;
  }
  if (t->budgetFixed > 0) budget = t->budgetFixed;
  // timing events on every 8th job only: two event packets and a host-side event query per job cost ~5 % of a frame
  const bool sample = c->prof_on && ((c->prof_tick++ & 7) == 0);
#ifdef LSD_DEVTOOLS
  t->dbgJobs++;
#endif
  t->jobTag = (t->jobTag % 0x7FFFF) + 1;
  t->launchOrdinal = 0;
  t->h_summary->seq = 0;
  t->h_summary->exhausted = 0;
  int guard = 0;
  int parity = 0, first = 1;
  if (int rc = prof_collect(c)) return rc;
  lsdhip_host_mark(2);
  while (true) {
    if (sample) HIPCHK(hipEventRecord(c->ev_a, c->stream));
    int rc = LSDHIP_OK;
    rc = launch_steps(t, job, budget, &parity, &first);
    if (rc) return rc;
    if (sample) { HIPCHK(hipEventRecord(c->ev_b, c->stream)); c->prof_pending = true; }
    lsdhip_host_mark(3);
    if (guard == 0 && t->enqueueHook && job.trackFrameSemantics) t->enqueueHook(t->enqueueHookUser);
    lsdhip_host_mark(4);
    // The finishing step writes the summary to pinned host memory and raises `done` last (system-scope fence in
    // between): poll it instead of sleeping in hipStreamSynchronize, whose wake-up costs more than two evaluations.
    // Steps of the budget still queued behind the finishing one exit immediately; later work is stream-ordered.
#ifdef LSD_DEVTOOLS
    const auto tw0 = std::chrono::steady_clock::now();
    struct WaitClock { lsdhip_tracker* t; std::chrono::steady_clock::time_point t0; ~WaitClock() { t->dbgWaitNs += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); } } waitClock_{t, tw0};
#endif
    c->mtx.unlock();    // nothing below touches context state until the result is in: let the mapping thread enqueue
    struct Relock { std::recursive_mutex& m; ~Relock() { m.lock(); } };
    {
    Relock relock_{c->mtx};
    if (t->spinWait) {
      volatile const int* done = &S->done;
      volatile const int* exhausted = &S->exhausted;
      const int lastSeq = t->spec.seq;                      // tag of the last launch enqueued
      const auto tStart = std::chrono::steady_clock::now();
      unsigned spins = 0;
      while (*done != t->jobTag) {
        if (*exhausted == lastSeq) break;                   // budget consumed, job unfinished (the last launch said so)
        if ((++spins & 0xFFFFFu) == 0) {                    // safety net only (a faulted launch never reports): every ~30 ms
          hipError_t q = hipStreamQuery(c->stream);
          if (q == hipSuccess) break;                       // budget consumed (done or not)
          if (q != hipErrorNotReady) { lsd_set_error("hipStreamQuery failed: %s", hipGetErrorString(q)); return LSDHIP_E_HIP; }
          if (std::chrono::steady_clock::now() - tStart > std::chrono::seconds(5)) { HIPCHK(hipStreamSynchronize(c->stream)); break; }
        }
        __builtin_ia32_pause();
      }
      std::atomic_thread_fence(std::memory_order_acquire);
    } else {
      HIPCHK(hipStreamSynchronize(c->stream));
    }
    }
    lsdhip_host_mark(5);
    if (*(volatile const int*)&S->done == t->jobTag) break;
#ifdef LSD_DEVTOOLS
    t->dbgMisses++;
#endif
    HIPCHK(hipStreamSynchronize(c->stream));   // out of budget: rare
#ifdef LSD_DEVTOOLS
    {
      static const bool dumpL0 = getenv("LSDHIP_DUMP_L0") != nullptr;
      if (dumpL0 && guard == 0 && t->budgetFixed == 1) {
        // the first launch (level `topLevel`, one trial) wrote parity 1, trial 0: rows | keys | tail contributions of its tiles
        const TrackScratch sc = scratch_of(t);
        const size_t rows = (size_t)t->max_blocks;
        const int nb = job.lv[topLevel].nblocks;
        t->dumpL0.assign((size_t)nb * (RS_COLS + 4 + 96) + sizeof(TrackState) / 4, 0u);
        unsigned* d = t->dumpL0.data();
        HIPCHK(hipMemcpy(d, sc.sums + (size_t)(1 * LSD_SPEC_MAX + 0) * RS_COLS * rows, (size_t)nb * RS_COLS * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(d + (size_t)nb * RS_COLS, sc.topkey + (size_t)(1 * LSD_SPEC_MAX + 0) * rows, (size_t)nb * 16, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(d + (size_t)nb * (RS_COLS + 4), sc.topval + (size_t)(1 * LSD_SPEC_MAX + 0) * rows * 96, (size_t)nb * 96 * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(d + (size_t)nb * (RS_COLS + 4 + 96), t->d_state + 1, sizeof(TrackState), hipMemcpyDeviceToHost));
      }
    }
#endif
    if (*(volatile const int*)&S->done == t->jobTag) break;
    if (int rc2 = prof_collect(c)) return rc2;
    if (t->jobKf && t->jobKf->depthVersion != t->jobKfVersion && guard < 8) {
      // While the host waited without the context lock, the mapping thread rewrote the keyframe's depth planes (setDepth): the
      // launches appended now would read the new planes in the middle of a job that started on the old ones.  The reference's
      // TrackingReference is a snapshot that cannot change under a job, so run the job again, whole, on the planes as they are now.
      t->jobKfVersion = t->jobKf->depthVersion;
      t->h_summary->done = 0;
      parity = 0; first = 1;
      budget = 12;
      ++guard;
      continue;
    }
    budget = t->budgetFixed > 0 ? t->budgetFixed : 6;
    if (++guard > 200) { lsd_set_error("tracking job did not terminate"); return LSDHIP_E_STATE; }
  }
  // `done` is in; the rest of the record is taken only once it adds up (summary_wait_consistent)
  if (int rcs = summary_wait_consistent(t, t->jobTag)) return rcs;
  if (sample) {
    c->prof_bytes += S->bytes;
    c->prof_launches += S->numLaunches;
  }
  if (myEpoch > c->doneEpoch) c->doneEpoch = myEpoch;
#ifdef LSD_DEVTOOLS
  {
    static const bool keepLog = getenv("LSDHIP_LAUNCH_LOG") != nullptr;
    if (keepLog) {
      // every launch of the job has been queued; wait for the ones behind the finishing launch too, then keep the log and clear it
      HIPCHK(hipStreamSynchronize(c->stream));
      const int n = t->launchOrdinal < 4095 ? t->launchOrdinal + 1 : 4096;
      t->lastLog.assign((size_t)n * 16, 0);
      HIPCHK(hipMemcpy(t->lastLog.data(), t->d_log, (size_t)n * 64, hipMemcpyDeviceToHost));
      HIPCHK(hipMemset(t->d_log, 0, (size_t)n * 64));
    }
  }
#endif
  t->numLaunches = S->numLaunches;
  t->recent[3] = t->recent[2]; t->recent[2] = t->recent[1]; t->recent[1] = t->recent[0]; t->recent[0] = t->numLaunches;
  for (int l = 0; l < LSD_LEVELS; l++) t->levelEvaluations[l] = S->levelEvals[l];
  t->numEvaluations = S->numEvaluations;
  t->numWarpUpdates = S->numWarpUpdates;
  t->pointUsage = S->pointUsage; t->lastGoodCount = S->goodCount; t->lastBadCount = S->badCount; t->lastMeanRes = S->meanRes;
  t->affineEstimation_a = S->aff_a; t->affineEstimation_b = S->aff_b;
  t->affineEstimation_a_lastIt = S->aff_a_lastIt; t->affineEstimation_b_lastIt = S->aff_b_lastIt;
  t->lastResidual = S->lastResidual;
  Tout->q = {S->q[0], S->q[1], S->q[2], S->q[3]};
  Tout->t[0] = S->t[0]; Tout->t[1] = S->t[1]; Tout->t[2] = S->t[2];
  return S->diverged ? LSDHIP_DIVERGED : LSDHIP_OK;
}

static void fill_result(lsdhip_tracker* t, const lsdm::SE3dH& T, lsdhip_track_result* out) {
  lsdm::se3d_to7(T, out->frameToReference);
  out->pointUsage = t->pointUsage; out->lastGoodCount = t->lastGoodCount; out->lastBadCount = t->lastBadCount;
  out->lastMeanRes = t->lastMeanRes; out->lastResidual = t->lastResidual;
  out->affineEstimation_a = t->affineEstimation_a; out->affineEstimation_b = t->affineEstimation_b;
  out->diverged = t->diverged; out->trackingWasGood = t->trackingWasGood;
  out->numEvaluations = t->numEvaluations; out->numWarpUpdates = t->numWarpUpdates;
}
static lsdm::SE3dH identity_d() { lsdm::SE3dH I; I.q = {1, 0, 0, 0}; I.t[0] = I.t[1] = I.t[2] = 0; return I; }

// trackFrame job description (SE3Tracker.cpp:280-322): levels SE3TRACKING_MAX_LEVEL-1 .. SE3TRACKING_MIN_LEVEL
static int fill_trackframe_job(lsdhip_tracker* t, TrackJob& job, lsdhip_frame* kf, lsdhip_frame* frame) {
  fill_job_common(t, job);
  for (int lvl = LSD_TRACK_MIN_LEVEL; lvl < LSD_TRACK_MAX_LEVEL; lvl++) fill_level(t, job, lvl, kf, frame, nullptr, nullptr, -1);
  job.lv[LSD_TRACK_MIN_LEVEL].writeMask = 1;
  int rc = lsd_frame_ensure_wasgood(frame);
  if (rc) return rc;
  job.wasGood = frame->d_wasGood;
  job.lastLevel = LSD_TRACK_MIN_LEVEL;
  job.trackFrameSemantics = 1;
  return LSDHIP_OK;
}
// epilogue of trackFrame (SE3Tracker.cpp:451-485) from the job's summary: flags, frame / keyframe side effects, result
static int finish_trackframe(lsdhip_tracker* t, const TrackSummary* S, lsdhip_frame* kf, lsdhip_frame* frame, lsdhip_track_result* out) {
  lsdhip_ctx* c = t->ctx;
  LSD_CTX_LOCK(c);
  t->numEvaluations = S->numEvaluations;
  t->numWarpUpdates = S->numWarpUpdates;
  t->pointUsage = S->pointUsage; t->lastGoodCount = S->goodCount; t->lastBadCount = S->badCount; t->lastMeanRes = S->meanRes;
  t->affineEstimation_a = S->aff_a; t->affineEstimation_b = S->aff_b;
  t->affineEstimation_a_lastIt = S->aff_a_lastIt; t->affineEstimation_b_lastIt = S->aff_b_lastIt;
  t->lastResidual = S->lastResidual;
  if (S->diverged) {
    t->diverged = true;
    t->trackingWasGood = false;
    fill_result(t, identity_d(), out);
    return LSDHIP_DIVERGED;
  }
  t->diverged = false;
  lsdm::SE3fH referenceToFrame;
  referenceToFrame.q = {S->q[0], S->q[1], S->q[2], S->q[3]};
  referenceToFrame.t[0] = S->t[0]; referenceToFrame.t[1] = S->t[1]; referenceToFrame.t[2] = S->t[2];
  t->trackingWasGood = t->lastGoodCount / (c->wl[LSD_TRACK_MIN_LEVEL] * c->hl[LSD_TRACK_MIN_LEVEL]) > MIN_GOODPERALL_PIXEL &&
                       t->lastGoodCount / (t->lastGoodCount + t->lastBadCount) > MIN_GOODPERGOODBAD_PIXEL;
  if (t->trackingWasGood) kf->numFramesTrackedOnThis++;
  frame->initialTrackedResidual = t->lastResidual / t->pointUsage;
  lsdm::SE3dH f2r = lsdm::se3d_from_f(lsdm::se3f_inverse(referenceToFrame));
  frame->thisToParent_raw.q = f2r.q;
  lsdm::q_normalize(frame->thisToParent_raw.q);   // sim3FromSE3 -> Sim3::setScale normalises the quaternion (rxso3.hpp:332-335)
  frame->thisToParent_raw.t[0] = f2r.t[0]; frame->thisToParent_raw.t[1] = f2r.t[1]; frame->thisToParent_raw.t[2] = f2r.t[2];
  frame->thisToParent_raw.s = 1;
  frame->trackingParent = kf;
  frame->trackingParentID = kf->id;
  fill_result(t, f2r, out);
  return LSDHIP_OK;
}

extern "C" int lsdhip_tracker_track(lsdhip_tracker* t, lsdhip_frame* kf, lsdhip_frame* frame, const double init[7],
                                    lsdhip_track_result* out) {
  if (!t || !kf || !frame || !init || !out) return LSDHIP_E_ARG;
  if (!kf->hasIDepth) { lsd_set_error("lsdhip_tracker_track: keyframe %d has no depth", kf->id); return LSDHIP_E_STATE; }
  lsdhip_ctx* c = t->ctx;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  t->diverged = false;
  t->trackingWasGood = true;
  t->affineEstimation_a = 1; t->affineEstimation_b = 0;
  t->numEvaluations = 0; t->numWarpUpdates = 0;
  TrackJob job;
  // pipelined contexts: the job reads the frame's pyramids and the keyframe's image pyramid + published depth planes, all products of
  // the mapping stream; it does NOT wait for mapping work queued behind those points (DepthMap::updateKeyframe of the previous frame)
  LsdTrackJobScope tjob_(c, false);
  if (c->pipeline) {
    long long need = frame->readySeq > kf->readySeq ? frame->readySeq : kf->readySeq;
    if (kf->depthSeq > need) need = kf->depthSeq;
    if (int rcw = lsd_t_wait_m(c, need)) return rcw;
  }
  int rc = fill_trackframe_job(t, job, kf, frame);
  if (rc) return rc;
  if (int rcg = lsd_gate_open(c)) return rcg;
  if (int rcd = lsd_pipe_dummy(c)) return rcd;
#ifdef LSD_DEVTOOLS
  static const bool traceInputs = getenv("LSDHIP_TRACE_INPUTS") != nullptr;
  auto trace_inputs = [&](int base) {
    for (int l = 1; l <= 4; l++) {
      const size_t nl = (size_t)c->wl[l] * c->hl[l];
      lsd_trace_sum(c, c->stream, base + 10 + l, frame->id, kf->d_idepth[l], nl * 4);
      lsd_trace_sum(c, c->stream, base + 14 + l, frame->id, kf->d_idepthVar[l], nl * 4);
      lsd_trace_sum(c, c->stream, base + 50 + l, frame->id, kf->d_image[l], nl * 4);
      lsd_trace_sum(c, c->stream, base + 54 + l, frame->id, frame->d_grad[l], nl * 16);
    }
  };
  if (traceInputs) trace_inputs(0);
  {
    unsigned long long pv = 0;
    for (int i = 0; i < 7; i++) { unsigned long long u; memcpy(&u, &init[i], 8); pv = pv * 1000003ull + u; }
    lsd_trace_val(c, 22, frame->id, pv);
    lsd_trace_val(c, 23, frame->id, (unsigned long long)kf->id);
  }
#endif
  lsdm::SE3fH referenceToFrame = lsdm::se3f_from_d(lsdm::se3d_inverse(lsdm::se3d_from7(init)));

  if (t->hostLM) {
    float last_residual = 0;
    rc = LSDHIP_OK;
    for (int lvl = LSD_TRACK_MAX_LEVEL - 1; lvl >= LSD_TRACK_MIN_LEVEL && rc == LSDHIP_OK; lvl--)
      rc = lm_level_host(t, job, lvl, referenceToFrame, &last_residual);
    t->lastResidual = last_residual;
    if (rc == LSDHIP_DIVERGED) {
      t->diverged = true;
      t->trackingWasGood = false;
      fill_result(t, identity_d(), out);
      return LSDHIP_DIVERGED;
    }
    if (rc) return rc;
    // host-LM debugging path: build the summary the common epilogue expects
    TrackSummary S = *t->h_summary;
    S.diverged = 0;
    S.q[0] = referenceToFrame.q.w; S.q[1] = referenceToFrame.q.x; S.q[2] = referenceToFrame.q.y; S.q[3] = referenceToFrame.q.z;
    S.t[0] = referenceToFrame.t[0]; S.t[1] = referenceToFrame.t[1]; S.t[2] = referenceToFrame.t[2];
    S.lastResidual = last_residual; S.numEvaluations = t->numEvaluations; S.numWarpUpdates = t->numWarpUpdates;
    S.pointUsage = t->pointUsage; S.goodCount = t->lastGoodCount; S.badCount = t->lastBadCount; S.meanRes = t->lastMeanRes;
    S.aff_a = t->affineEstimation_a; S.aff_b = t->affineEstimation_b;
    S.aff_a_lastIt = t->affineEstimation_a_lastIt; S.aff_b_lastIt = t->affineEstimation_b_lastIt;
    return finish_trackframe(t, &S, kf, frame, out);
  }
  t->jobKf = kf;
  t->jobKfVersion = kf->depthVersion;
#ifdef LSD_DEVTOOLS
  const lsdm::SE3fH referenceToFrame0 = referenceToFrame;
  lsd_trace_val(c, 26, frame->id, t->dbgCum / 400ull);
#endif
  rc = track_device(t, job, LSD_TRACK_MAX_LEVEL - 1, referenceToFrame, &referenceToFrame);
  if (rc != LSDHIP_OK && rc != LSDHIP_DIVERGED) { t->jobKf = nullptr; return rc; }
#ifdef LSD_DEVTOOLS
  static const bool replay = getenv("LSDHIP_TRACK_REPLAY") != nullptr;
  if (replay && c->pipeline) {
    // developer check: the same job once more with the mapping stream drained — identical inputs must give the identical result
    unsigned long long pv = 0;
    for (int i = 0; i < 4; i++) { unsigned u; memcpy(&u, &t->h_summary->q[i], 4); pv = pv * 1000003ull + u; }
    for (int i = 0; i < 3; i++) { unsigned u; memcpy(&u, &t->h_summary->t[i], 4); pv = pv * 1000003ull + u; }
    lsd_trace_val(c, 24, frame->id, pv);
    lsd_trace_val(c, 25, frame->id, (unsigned long long)t->h_summary->numEvaluations * 1000 + t->h_summary->lastCand);
    HIPCHK(hipStreamSynchronize(c->mstream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->mDoneSeq = c->mSeq;
    const std::vector<unsigned> dump1 = t->dumpL0;
    if (traceInputs) trace_inputs(100);
    lsd_trace_val(c, 27, frame->id, t->dbgCum / 400ull);
    referenceToFrame = referenceToFrame0;
    rc = track_device(t, job, LSD_TRACK_MAX_LEVEL - 1, referenceToFrame, &referenceToFrame);
    if (rc != LSDHIP_OK && rc != LSDHIP_DIVERGED) { t->jobKf = nullptr; return rc; }
    HIPCHK(hipStreamSynchronize(c->stream));
    if (!dump1.empty() && dump1.size() == t->dumpL0.size()) {
      const int nb = job.lv[LSD_TRACK_MAX_LEVEL - 1].nblocks;
      int shown = 0;
      for (size_t i = 0; i < dump1.size(); i++)
        if (dump1[i] != t->dumpL0[i] && shown++ < 12) {
          const size_t a = (size_t)nb * RS_COLS, b = a + (size_t)nb * 4, cst = b + (size_t)nb * 96;
          float f1, f2; memcpy(&f1, &dump1[i], 4); memcpy(&f2, &t->dumpL0[i], 4);
          if (i < a) fprintf(stderr, "L0DIFF frame %d: sums tile %zu column %zu: run %.9g (%08x) replay %.9g (%08x)\n", frame->id, i / RS_COLS, i % RS_COLS, f1, dump1[i], f2, t->dumpL0[i]);
          else if (i < b) fprintf(stderr, "L0DIFF frame %d: topkey tile %zu [%zu]: run %d replay %d\n", frame->id, (i - a) / 4, (i - a) % 4, (int)dump1[i], (int)t->dumpL0[i]);
          else if (i < cst) fprintf(stderr, "L0DIFF frame %d: topval tile %zu slot %zu entry %zu: run %.9g replay %.9g\n", frame->id, (i - b) / 96, ((i - b) % 96) / 32, (i - b) % 32, f1, f2);
          else fprintf(stderr, "L0DIFF frame %d: state word %zu: run %08x (%.9g) replay %08x (%.9g)\n", frame->id, i - cst, dump1[i], f1, t->dumpL0[i], f2);
        }
      if (shown) fprintf(stderr, "L0DIFF frame %d: %d words differ after the first launch\n", frame->id, shown);
    }
  }
#endif
  t->jobKf = nullptr;
  lsdhip_host_mark(6);
#ifdef LSD_DEVTOOLS
  {
    unsigned long long pv = 0;
    for (int i = 0; i < 4; i++) { unsigned u; memcpy(&u, &t->h_summary->q[i], 4); pv = pv * 1000003ull + u; }
    for (int i = 0; i < 3; i++) { unsigned u; memcpy(&u, &t->h_summary->t[i], 4); pv = pv * 1000003ull + u; }
    lsd_trace_val(c, 20, frame->id, pv);
    lsd_trace_val(c, 21, frame->id, (unsigned long long)t->h_summary->numEvaluations * 1000 + t->h_summary->lastCand);
  }
#endif
  if (c->pipeline && rc == LSDHIP_OK && t->h_summary->lastCand > 0 && t->h_summary->level == LSD_TRACK_MIN_LEVEL) {
    // the final mask sits in a side plane: to be merged into the frame's plane on the mapping stream, ahead of whatever reads the mask
    // next — noted here, queued by the next mapping-stream operation (nothing is launched between two tracking jobs)
    const uint8_t* side = t->spec.wasGoodSide + (size_t)(t->h_summary->lastCand - 1) * t->maskStride;
    t->maskMergeSeq[t->maskSet] = -1;               // pending
    c->pendingMerges.push_back({frame->d_wasGood, side, &t->maskMergeSeq[t->maskSet]});
  }
  rc = finish_trackframe(t, t->h_summary, kf, frame, out);
  lsdhip_host_mark(7);
  return rc;
}

// ---- batches: n independent jobs in the same launches (job = blockIdx.y) -------------------------------------------
static int batch_reserve(lsdhip_tracker* t, int n) {
  lsdhip_ctx* c = t->ctx;
  LSD_CTX_LOCK(c);
  if (n <= t->batch_capacity) return LSDHIP_OK;
  HIPCHK(hipStreamSynchronize(c->stream));
  if (t->d_bjobs) { (void)hipFree(t->d_bjobs); (void)hipFree(t->d_bstate); (void)hipFree(t->d_bscratch); (void)hipHostFree(t->h_bjobs); (void)hipHostFree(t->h_bsummary); }
  t->batch_capacity = n < 8 ? 8 : n;
  const size_t B = (size_t)t->batch_capacity, rows = (size_t)t->max_blocks;
  const size_t per_job = (size_t)LSD_BATCH_SPEC_MAX * (2 * RS_COLS * rows * 4 + 2 * rows * 16 + 2 * rows * 96 * 4 + 2 * 32 * 4);
  HIPCHK(hipMalloc((void**)&t->d_bjobs, B * sizeof(TrackJob)));
  HIPCHK(hipMalloc((void**)&t->d_bstate, B * 2 * sizeof(TrackState)));
  HIPCHK(hipMalloc((void**)&t->d_bscratch, B * per_job));
  HIPCHK(hipMemsetAsync(t->d_bscratch, 0, B * per_job, c->stream));
  HIPCHK(hipHostMalloc((void**)&t->h_bjobs, B * sizeof(TrackJob), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&t->h_bsummary, B * sizeof(TrackSummary), hipHostMallocMapped));
  return LSDHIP_OK;
}
// scratch of a batch: arrays over [job][parity][trial]; cmax = trial slots per parity (1: no speculation)
static TrackScratch batch_scratch(lsdhip_tracker* t, int cmax) {
  TrackScratch sc;
  const size_t B = (size_t)t->batch_capacity, rows = (size_t)t->max_blocks, C = (size_t)LSD_BATCH_SPEC_MAX;
  sc.sums = t->d_bscratch;
  sc.topkey = (int4*)(t->d_bscratch + B * C * 2 * RS_COLS * rows);
  sc.topval = t->d_bscratch + B * C * 2 * RS_COLS * rows + B * C * 2 * 4 * rows;
  sc.recs = cmax > 1 ? t->d_bscratch + B * C * 2 * RS_COLS * rows + B * C * 2 * 4 * rows + B * C * 2 * 96 * rows : nullptr;
  sc.max_rows = t->max_blocks;
  sc.cmax = cmax;
#ifdef LSD_PHASE_TRACE
  sc.trace = t->d_trace;
#endif
  return sc;
}
// With many jobs in flight the other jobs hide a job's latency, so each job gets fewer, fatter workgroups: the
// per-workgroup LM replay (the price of the launch needing no inter-workgroup communication) shrinks accordingly.
static void batch_begin(lsdhip_tracker* t, int n) {
  t->batch_jobs = n;
  t->cap_override = (t->grid_cap / n) & ~7;
  if (t->cap_override < 16) t->cap_override = 16;
  if (n == 1) t->cap_override = 0;
}
// runs the n jobs described in t->h_bjobs[0..n) to completion; summaries in t->h_bsummary
static int batch_run(lsdhip_tracker* t, int n, bool callHook = false) {
  lsdhip_ctx* c = t->ctx;
  LSD_CTX_LOCK(c);
  t->cap_override = 0;
  t->batch_jobs = 0;
  int grid = 1;
  bool split = false;
  for (int j = 0; j < n; j++) {
    const TrackJob& job = t->h_bjobs[j];
    for (int l = job.lastLevel; l <= job.topLevel; l++) {
      if (job.lv[l].nblocks > grid) grid = job.lv[l].nblocks;
      if (job.lv[l].tilePx > 0) split = true;
    }
    t->h_bsummary[j].done = 0;
  }
  HIPCHK(hipMemcpyAsync(t->d_bjobs, t->h_bjobs, (size_t)n * sizeof(TrackJob), hipMemcpyHostToDevice, c->stream));
  // Reject-chain speculation in throughput mode (as single jobs have it, SE3Tracker.cpp:341-447): a step evaluates the next `trials`
  // retries of the LM loop side by side, the next step consumes them in the reference's order — same decisions, same evaluation
  // counts, fewer dependent rounds.  Per level as many trials as keep jobs x trials x pixels of the level within LSD_BATCH_SPEC_PIXELS
  // (a round must not cost more than the rounds it saves); one at the level that writes refPixelWasGood (no side planes in batches).
  TrackSpec spec = TrackSpec{};
#ifdef LSD_PHASE_TRACE
  spec.traceWg = getenv("LSDHIP_TRACE_WG") ? atoi(getenv("LSDHIP_TRACE_WG")) : 0;
#endif
  int lmGrid = 1;
  static const int specMaxEnv = getenv("LSDHIP_BATCH_SPEC") ? atoi(getenv("LSDHIP_BATCH_SPEC")) : LSD_BATCH_SPEC_MAX;   // developer A/B (1: off)
  int specMax = specMaxEnv < 1 ? 1 : (specMaxEnv > LSD_BATCH_SPEC_MAX ? LSD_BATCH_SPEC_MAX : specMaxEnv);
  if (t->specC < specMax) specMax = t->specC;        // lsdhip_tracker_set_speculation(t, 1, 0): one evaluation per step, batches too
  if (getenv("LSDHIP_BATCH_FUSED") && atoi(getenv("LSDHIP_BATCH_FUSED")) == 1) specMax = 1;   // developer A/B: the fused form without speculation
  if (split && specMax > 1) {
    for (int l = 0; l < LSD_LEVELS; l++) spec.trials[l] = 1;
    const TrackJob& j0 = t->h_bjobs[0];
    for (int l = j0.lastLevel; l <= j0.topLevel; l++) {
      bool ok = true;
      for (int j = 0; j < n; j++) ok = ok && t->h_bjobs[j].lv[l].tilePx > 0 && !t->h_bjobs[j].lv[l].writeMask && t->h_bjobs[j].lastLevel <= l && t->h_bjobs[j].topLevel >= l;
      if (!ok) continue;
      static const long long specPixels = getenv("LSDHIP_BATCH_SPEC_PIXELS") ? atoll(getenv("LSDHIP_BATCH_SPEC_PIXELS")) : LSD_BATCH_SPEC_PIXELS;   // developer sweep
      long long tr = specPixels / ((long long)j0.lv[l].w * j0.lv[l].h * n);
      if (tr > specMax) tr = specMax;
      if (tr < 1) tr = 1;
      spec.trials[l] = (int)tr;
      if (tr > lmGrid) lmGrid = (int)tr;
    }
    spec.specC = lmGrid;
    // the evaluation launch holds (trial, strip) workgroups of the level with the most of them
    for (int j = 0; j < n; j++) {
      const TrackJob& job = t->h_bjobs[j];
      for (int l = job.lastLevel; l <= job.topLevel; l++) if (job.lv[l].nblocks * spec.trials[l] > grid) grid = job.lv[l].nblocks * spec.trials[l];
    }
  }
  const TrackScratch sc = batch_scratch(t, split && lmGrid > 1 ? LSD_BATCH_SPEC_MAX : 1);
  TrackSummary* d_sum = nullptr;
  HIPCHK(hipHostGetDevicePointer((void**)&d_sum, t->h_bsummary, 0));
  if (int rcp = prof_collect(c)) return rcp;
  // budget of rounds: what the recent batches needed (+ the finishing step and a margin); launches behind the last job's finishing step
  // cost ~3 us each
  int budget = 26;
  if (split && t->batchRecent[0] > 0) {
    budget = 0;
    for (int i = 0; i < 4; i++) if (t->batchRecent[i] > budget) budget = t->batchRecent[i];
    static const int marginEnv = getenv("LSDHIP_BATCH_MARGIN") ? atoi(getenv("LSDHIP_BATCH_MARGIN")) : 3;   // developer sweep
    budget += marginEnv;
  }
  int parity = 0, first = 1, guard = 0;
  static const int fusedEnv = getenv("LSDHIP_BATCH_FUSED") ? atoi(getenv("LSDHIP_BATCH_FUSED")) : 2;   // developer A/B (round 6): 0 = LM launch + evaluation launch per round
  // Fused rounds: the host polls the jobs' summaries in pinned memory (as lsdhip_tracker_track does for one job) instead of draining the
  // stream: the budget's launches behind the last job's finishing step (~3 us each, a handful per batch) then run while the host is
  // already reading the results and queueing what follows.  `done` carries the batch's tag, the record is taken once it adds up to its
  // check word; the budget's last launch reports a job it leaves unfinished (`exhausted`).
  static const bool pollEnv = !(getenv("LSDHIP_BATCH_POLL") && getenv("LSDHIP_BATCH_POLL")[0] == '0');   // developer A/B
  const bool polled = split && fusedEnv && t->spinWait && pollEnv;
  if (polled) {
    t->batchTag = t->batchTag >= 0x3FFFFFFF ? 2 : t->batchTag + 1;
    if (t->batchTag < 2) t->batchTag = 2;
    spec.seq = t->batchTag;
    for (int j = 0; j < n; j++) t->h_bsummary[j].exhausted = 0;
  }
  // the coarse levels of every job inside one workgroup (k_track_solo), then lock-step rounds for the rest
  static const int soloMinEnv = getenv("LSDHIP_BATCH_SOLO_MIN") ? atoi(getenv("LSDHIP_BATCH_SOLO_MIN")) : LSD_SOLO_MIN_JOBS;   // developer A/B (0: never)
  const int soloMin = t->soloMinJobs >= 0 ? t->soloMinJobs : soloMinEnv;     // lsdhip_tracker_set_batch_coarse_min_jobs
  bool soloDue = split && fusedEnv && soloMin > 0 && n >= soloMin;
  if (soloDue) {
    // (nothing to walk if no job's top level fits the tile — 1280x1024: level 4 is 80x64 = 5120 pixels —: the launch would only copy states)
    bool any = false;
    for (int j = 0; j < n && !any; j++) {
      const TrackLevel& L = t->h_bjobs[j].lv[t->h_bjobs[j].topLevel];
      any = L.tilePx > 0 && !L.writeMask && (long long)L.w * L.h <= LSD_SOLO_MAX_PX;
    }
    soloDue = any;
  }
  while (true) {
    if (c->prof_on) HIPCHK(hipEventRecord(c->ev_a, c->stream));
    if (soloDue) {
      soloDue = false;
      const int dw = polled ? t->batchTag : 1;
      hipLaunchKernelGGL(k_track_solo<512>, dim3(n), dim3(512), 0, c->stream, (const TrackJob*)t->d_bjobs, t->d_bstate, d_sum, parity, dw);
      first = 0;
      parity ^= 1;
    }
    for (int i = 0; i < budget; i++) {
      spec.last = (polled && i == budget - 1) ? 1 : 0;
      if (split && fusedEnv) {
        hipLaunchKernelGGL((k_track_step<256, true, TS_FUSED>), dim3(grid, n), dim3(256), 0, c->stream, t->h_bjobs[0], (const TrackJob*)t->d_bjobs,
                           t->d_bstate, sc, d_sum, parity, first, spec);
      } else if (split) {
        // throughput mode: one LM workgroup per (trial, job), then a pure evaluation launch over all jobs' (trial, strip) pairs
        hipLaunchKernelGGL((k_track_step<256, true, TS_LM>), dim3(lmGrid, n), dim3(256), 0, c->stream, t->h_bjobs[0], (const TrackJob*)t->d_bjobs,
                           t->d_bstate, sc, d_sum, parity, first, spec);
        hipLaunchKernelGGL((k_track_step<256, true, TS_EVAL>), dim3(grid, n), dim3(256), 0, c->stream, t->h_bjobs[0], (const TrackJob*)t->d_bjobs,
                           t->d_bstate, sc, d_sum, 1 - parity, 0, spec);
      } else {
        hipLaunchKernelGGL((k_track_step<256, true>), dim3(grid, n), dim3(256), 0, c->stream, t->h_bjobs[0], (const TrackJob*)t->d_bjobs,
                           t->d_bstate, sc, d_sum, parity, first, TrackSpec{});
      }
      first = 0;
      parity ^= 1;
    }
    HIPCHK(hipGetLastError());
    if (c->prof_on) { HIPCHK(hipEventRecord(c->ev_b, c->stream)); c->prof_pending = true; }
    // the batch's launches are queued: the place for everything the device can do beside them (lsdhip_tracker_set_enqueue_hook)
    if (callHook && guard == 0 && t->enqueueHook) t->enqueueHook(t->enqueueHookUser);
    bool all = true;
    if (polled) {
      const auto t0 = std::chrono::steady_clock::now();
      for (int j = 0; j < n && all; j++) {
        volatile const int* done = &t->h_bsummary[j].done;
        volatile const int* exhausted = &t->h_bsummary[j].exhausted;
        for (unsigned spins = 0;; spins++) {
          if (*done == t->batchTag) break;
          if (*exhausted == t->batchTag) { all = false; break; }
          if ((spins & 0xFFFFu) == 0xFFFFu) {
            // the stream may have stopped on an error, or (a budget cut short by a failed launch) nobody is left to report
            if (hipStreamQuery(c->stream) != hipErrorNotReady) { all = *done == t->batchTag; if (!all) break; }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) { lsd_set_error("tracking batch: no progress report from the device"); return LSDHIP_E_STATE; }
          }
          __builtin_ia32_pause();
        }
        if (all) { if (int rcs = summary_wait_consistent(t, t->batchTag, &t->h_bsummary[j])) return rcs; }
      }
      if (!all) {
        HIPCHK(hipStreamSynchronize(c->stream));   // out of budget: rare
        all = true;
        for (int j = 0; j < n; j++) { all = all && t->h_bsummary[j].done == t->batchTag; t->h_bsummary[j].exhausted = 0; }
      }
    } else {
      HIPCHK(hipStreamSynchronize(c->stream));
      for (int j = 0; j < n; j++) all = all && t->h_bsummary[j].done;
    }
    if (all) break;
    if (int rc2 = prof_collect(c)) return rc2;
    budget = 6;
    if (++guard > 200) { lsd_set_error("tracking batch did not terminate"); return LSDHIP_E_STATE; }
  }
  if (c->prof_on)
    for (int j = 0; j < n; j++) { c->prof_bytes += t->h_bsummary[j].bytes; c->prof_launches += t->h_bsummary[j].numEvaluations; }
  if (split) {
    int rounds = 0;
    for (int j = 0; j < n; j++) if (t->h_bsummary[j].numLaunches > rounds) rounds = t->h_bsummary[j].numLaunches;
    t->batchRecent[3] = t->batchRecent[2]; t->batchRecent[2] = t->batchRecent[1]; t->batchRecent[1] = t->batchRecent[0]; t->batchRecent[0] = rounds;
    t->numLaunches = rounds;
  }
  return LSDHIP_OK;
}

// SE3Tracker::trackFrame for n independent (keyframe, frame) pairs in the same launches.  Each job runs the arithmetic
// of lsdhip_tracker_track (same kernel; a batch tiles a level into fewer workgroups, which only changes summation
// order); the point is throughput — n evaluations share one launch and its latency chain.  inits: n x 7, results: n.
// Returns LSDHIP_OK, or LSDHIP_DIVERGED if any job diverged (see results[j].diverged).
extern "C" void lsdhip_build_defaults(lsdhip_build_defaults_t* out) {
  if (!out) return;
  out->ctx_async = LSD_DEFAULT_ASYNC; out->ctx_pipeline = LSD_DEFAULT_PIPELINE;
  out->spec_trials_small = LSD_SPEC_TRIALS_SMALL; out->spec_small_pixels = LSD_SPEC_SMALL_PX;
  out->spec_trials_mid = LSD_SPEC_TRIALS_MID; out->spec_mid_pixels = LSD_SPEC_MID_PX;
  out->spec_workgroups = LSD_SPEC_CAP_WORKGROUPS; out->spec_workgroups_above_pixels = LSD_SPEC_CAP_ABOVE_PX;
  out->spec_trials_max = LSD_SPEC_MAX;
  out->batch_throughput_min_jobs = LSD_BATCH_THROUGHPUT_MIN_JOBS; out->batch_strip_workgroups = LSD_BATCH_STRIP_WORKGROUPS;
  out->batch_coarse_min_jobs = LSD_SOLO_MIN_JOBS; out->batch_coarse_max_pixels = LSD_SOLO_MAX_PX; out->batch_coarse_max_points = LSD_SOLO_MAX_PTS;
}

extern "C" int lsdhip_tracker_track_batch(lsdhip_tracker* t, int n, lsdhip_frame** keyframes, lsdhip_frame** frames,
                                          const double* inits, lsdhip_track_result* results) {
  if (!t || n <= 0 || !keyframes || !frames || !inits || !results) return LSDHIP_E_ARG;
  lsdhip_ctx* c = t->ctx;
  LSD_CTX_LOCK(c);
  HIPCHK(hipSetDevice(c->device));
  for (int j = 0; j < n; j++) {
    if (!keyframes[j] || !frames[j]) return LSDHIP_E_ARG;
    if (!keyframes[j]->hasIDepth) { lsd_set_error("lsdhip_tracker_track_batch: keyframe %d has no depth", keyframes[j]->id); return LSDHIP_E_STATE; }
  }
  if (c->pipeline) {
    // as lsdhip_tracker_track: the tracking stream waits for the mapping-stream points its inputs were complete at (the frames'
    // pyramids, the keyframes' PUBLISHED depth) and for nothing queued behind them — the mapping iterations of OTHER sequences run
    // beside this batch (SlamLoopBatch::setOverlapped).  batch_run leaves the tracking stream drained.
    long long need = 0;
    for (int j = 0; j < n; j++) {
      need = std::max(need, std::max(frames[j]->readySeq, std::max(keyframes[j]->readySeq, keyframes[j]->depthSeq)));
    }
    if (int rcw = lsd_t_wait_m(c, need)) return rcw;
  }
  if (n >= LSD_BATCH_THROUGHPUT_MIN_JOBS) { if (int rcb = lsd_frames_require_ref_blocks(keyframes, n, c->stream)) return rcb; }   // the strips read them
  int rc = batch_reserve(t, n);
  if (rc) return rc;
  batch_begin(t, n);
  for (int j = 0; j < n; j++) {
    TrackJob& job = t->h_bjobs[j];
    rc = fill_trackframe_job(t, job, keyframes[j], frames[j]);
    if (rc) { t->cap_override = 0; t->batch_jobs = 0; return rc; }
    job.evalOnly = 0;
    job.topLevel = LSD_TRACK_MAX_LEVEL - 1;
    job.T0 = lsdm::se3f_from_d(lsdm::se3d_inverse(lsdm::se3d_from7(inits + 7 * (size_t)j)));
    job.aff_a0 = 1.0f; job.aff_b0 = 0.0f;
  }
  rc = batch_run(t, n, true);
  if (rc) return rc;
  int rcAll = LSDHIP_OK;
  for (int j = 0; j < n; j++) {
    rc = finish_trackframe(t, &t->h_bsummary[j], keyframes[j], frames[j], &results[j]);
    if (rc == LSDHIP_DIVERGED) rcAll = LSDHIP_DIVERGED;
    else if (rc) return rc;
  }
  return rcAll;
}

// Measurement hook (profiles/r03_sizes.md, bench.py's roofline_throughput_mode): the throughput-mode evaluation launch alone.
// n jobs (keyframes[j], frames[j]) are evaluated at pyramid level `level` at the poses refToFrame[j] (7 floats each: q w x y z, t):
// one LM launch builds the states, then `repeats` identical evaluation launches (k_track_step<.., TS_EVAL>: idempotent — it reads
// the published state and rewrites the same partial rows) are timed with one HIP event pair, and a closing LM launch finalises the
// sums so that the algorithmic bytes of ONE evaluation launch (sum over the jobs, SURVEY.md 8(d) formula as counted by lm_wave)
// can be reported.  Needs n >= 8 (throughput mode).
extern "C" int lsdhip_tracker_eval_throughput(lsdhip_tracker* t, int n, lsdhip_frame** keyframes, lsdhip_frame** frames, const float* refToFrame,
                                              int level, int repeats, double* ms_per_launch, double* bytes_per_launch) {
  if (!t || n < 8 || !keyframes || !frames || !refToFrame || level < 0 || level >= LSD_LEVELS || repeats < 1 || !ms_per_launch || !bytes_per_launch)
    return LSDHIP_E_ARG;
  lsdhip_ctx* c = t->ctx;
  LSD_CTX_LOCK(c);
  LsdTrackJobScope tjob_(c, true);
  if (tjob_.rc) return tjob_.rc;
  HIPCHK(hipSetDevice(c->device));
  for (int j = 0; j < n; j++) if (!keyframes[j] || !frames[j] || !keyframes[j]->hasIDepth) return LSDHIP_E_ARG;
  if (int rcb = lsd_frames_require_ref_blocks(keyframes, n, c->stream)) return rcb;
  int rc = batch_reserve(t, n);
  if (rc) return rc;
  batch_begin(t, n);
  int grid = 1;
  for (int j = 0; j < n; j++) {
    if (!keyframes[j] || !frames[j] || !keyframes[j]->hasIDepth) { t->cap_override = 0; t->batch_jobs = 0; return LSDHIP_E_ARG; }
    TrackJob& job = t->h_bjobs[j];
    fill_job_common(t, job);
    fill_level(t, job, level, keyframes[j], frames[j], nullptr, nullptr, -1);
    if (level == LSD_TRACK_MIN_LEVEL) {
      rc = lsd_frame_ensure_wasgood(frames[j]);
      if (rc) { t->cap_override = 0; t->batch_jobs = 0; return rc; }
      job.wasGood = frames[j]->d_wasGood;
      job.lv[level].writeMask = 1;
    }
    job.trackFrameSemantics = 1;
    job.evalOnly = 1;
    job.lastLevel = level;
    job.topLevel = level;
    const float* T7 = refToFrame + 7 * (size_t)j;
    job.T0.q = {T7[0], T7[1], T7[2], T7[3]};
    job.T0.t[0] = T7[4]; job.T0.t[1] = T7[5]; job.T0.t[2] = T7[6];
    job.aff_a0 = 1.0f; job.aff_b0 = 0.0f;
    if (job.lv[level].nblocks > grid) grid = job.lv[level].nblocks;
    if (job.lv[level].tilePx == 0) { t->cap_override = 0; t->batch_jobs = 0; lsd_set_error("lsdhip_tracker_eval_throughput: level %d is not in throughput mode", level); return LSDHIP_E_STATE; }
    t->h_bsummary[j].done = 0;
  }
  t->cap_override = 0;
  t->batch_jobs = 0;
  HIPCHK(hipMemcpyAsync(t->d_bjobs, t->h_bjobs, (size_t)n * sizeof(TrackJob), hipMemcpyHostToDevice, c->stream));
  const TrackScratch sc = batch_scratch(t, 1);
  TrackSummary* d_sum = nullptr;
  HIPCHK(hipHostGetDevicePointer((void**)&d_sum, t->h_bsummary, 0));
  struct EventPair {      // destroyed on every exit path
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } evp;
  HIPCHK(hipEventCreate(&evp.a));
  HIPCHK(hipEventCreate(&evp.b));
  const hipEvent_t e0 = evp.a, e1 = evp.b;
  hipLaunchKernelGGL((k_track_step<256, true, TS_LM>), dim3(1, n), dim3(256), 0, c->stream, t->h_bjobs[0], (const TrackJob*)t->d_bjobs, t->d_bstate, sc,
                     d_sum, 0, 1, TrackSpec{});
  auto eval_launch = [&]() -> int {
    hipLaunchKernelGGL((k_track_step<256, true, TS_EVAL>), dim3(grid, n), dim3(256), 0, c->stream, t->h_bjobs[0], (const TrackJob*)t->d_bjobs, t->d_bstate, sc,
                       d_sum, 1, 0, TrackSpec{});
    return LSDHIP_OK;
  };
  if (int rcw = eval_launch()) return rcw;     // warm-up
  HIPCHK(hipEventRecord(e0, c->stream));
  for (int r = 0; r < repeats; r++)
    if (int rce = eval_launch()) return rce;
  HIPCHK(hipEventRecord(e1, c->stream));
  hipLaunchKernelGGL((k_track_step<256, true, TS_LM>), dim3(1, n), dim3(256), 0, c->stream, t->h_bjobs[0], (const TrackJob*)t->d_bjobs, t->d_bstate, sc,
                     d_sum, 1, 0, TrackSpec{});
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  double bytes = 0;
  for (int j = 0; j < n; j++) {
    if (!t->h_bsummary[j].done) { lsd_set_error("lsdhip_tracker_eval_throughput: job %d did not finish", j); return LSDHIP_E_STATE; }
    bytes += t->h_bsummary[j].bytes;
  }
  *ms_per_launch = ms / repeats;
  *bytes_per_launch = bytes;
  return LSDHIP_OK;
}

extern "C" int lsdhip_tracker_evaluate(lsdhip_tracker* t, lsdhip_frame* kf, lsdhip_frame* frame, const float T7[7], int level,
                                       float aff_a, float aff_b, lsdhip_residual_record* out) {
  if (!t || !kf || !frame || !T7 || !out || level < 0 || level >= LSD_LEVELS) return LSDHIP_E_ARG;
  if (!kf->hasIDepth) { lsd_set_error("lsdhip_tracker_evaluate: keyframe has no depth"); return LSDHIP_E_STATE; }
  HIPCHK(hipSetDevice(t->ctx->device));
  LSD_CTX_LOCK(t->ctx);
  LsdTrackJobScope tjob_(t->ctx, true);
  if (tjob_.rc) return tjob_.rc;
  lsdm::SE3fH T;
  T.q = {T7[0], T7[1], T7[2], T7[3]};
  T.t[0] = T7[4]; T.t[1] = T7[5]; T.t[2] = T7[6];
  t->affineEstimation_a = aff_a; t->affineEstimation_b = aff_b;
  TrackJob job;
  fill_job_common(t, job);
  fill_level(t, job, level, kf, frame, nullptr, nullptr, -1);
  if (level == LSD_TRACK_MIN_LEVEL) {
    int rc = lsd_frame_ensure_wasgood(frame);
    if (rc) return rc;
    job.wasGood = frame->d_wasGood;
    job.lv[level].writeMask = 1;
  }
  job.trackFrameSemantics = 1;
  EvalOut ev;
  int rc = evaluate_pose(t, job, T, level, &ev);
  if (rc) return rc;
  out->warped_size = ev.warped_size;
  out->goodCount = t->lastGoodCount; out->badCount = t->lastBadCount; out->pointUsage = t->pointUsage;
  out->meanRes = t->lastMeanRes; out->retval = ev.retval;
  out->affine_a_lastIt = t->affineEstimation_a_lastIt; out->affine_b_lastIt = t->affineEstimation_b_lastIt;
  out->weightedError = ev.weightedError;
  memcpy(out->A, ev.A, sizeof(ev.A)); memcpy(out->b, ev.b, sizeof(ev.b));
  out->lsError = ev.lsError; out->num_constraints = ev.num_constraints;
  return LSDHIP_OK;
}

static int upload_points(lsdhip_tracker* t, const float* pos, const float* colvar, int n) {
  if (n > t->pts_capacity) {
    if (t->d_pts) HIPCHK(hipFree(t->d_pts));
    t->pts_capacity = n > 4096 ? n : 4096;
    HIPCHK(hipMalloc((void**)&t->d_pts, (size_t)t->pts_capacity * 5 * sizeof(float)));
  }
  HIPCHK(hipMemcpyAsync(t->d_pts, pos, (size_t)n * 12, hipMemcpyHostToDevice, t->ctx->stream));
  if (colvar) HIPCHK(hipMemcpyAsync(t->d_pts + (size_t)t->pts_capacity * 3, colvar, (size_t)n * 8, hipMemcpyHostToDevice, t->ctx->stream));
  return LSDHIP_OK;
}

extern "C" int lsdhip_tracker_track_permaref(lsdhip_tracker* t, const float* pos, const float* colvar, int n, lsdhip_frame* frame,
                                             const double refToFrame[7], lsdhip_track_result* out) {
  if (!t || !pos || !colvar || n <= 0 || !frame || !refToFrame || !out) return LSDHIP_E_ARG;
  lsdhip_ctx* c = t->ctx;
  LSD_CTX_LOCK(c);
  LsdTrackJobScope tjob_(c, true);
  if (tjob_.rc) return tjob_.rc;
  HIPCHK(hipSetDevice(c->device));
    int rc = upload_points(t, pos, colvar, n);
  if (rc) return rc;
  lsdm::SE3fH referenceToFrame = lsdm::se3f_from_d(lsdm::se3d_from7(refToFrame));
  t->affineEstimation_a = 1; t->affineEstimation_b = 0;
  t->diverged = false; t->trackingWasGood = true;
  t->numEvaluations = 0; t->numWarpUpdates = 0;
  const int L = LSD_QUICK_KF_CHECK_LVL;
  TrackJob job;
  fill_job_common(t, job);
  fill_level(t, job, L, nullptr, frame, t->d_pts, t->d_pts + (size_t)t->pts_capacity * 3, n);
  job.lv[L].lambdaInitial = t->lambdaInitialTestTrack; job.lv[L].stepSizeMin = t->stepSizeMinTestTrack;
  job.lv[L].convergenceEps = t->convergenceEpsTestTrack; job.lv[L].maxIts = (int)t->maxItsTestTrack;
  job.lastLevel = L;
  job.trackFrameSemantics = 0;
  if (t->hostLM) {
    float lastErr = 0;
    rc = lm_level_host(t, job, L, referenceToFrame, &lastErr);
    t->lastResidual = lastErr;
  } else {
    rc = track_device(t, job, L, referenceToFrame, &referenceToFrame);
  }
  if (rc == LSDHIP_DIVERGED) {
    t->diverged = true; t->trackingWasGood = false;
    fill_result(t, identity_d(), out);
    return LSDHIP_DIVERGED;
  }
  if (rc) return rc;
  t->trackingWasGood = !t->diverged && t->lastGoodCount / (c->wl[L] * c->hl[L]) > MIN_GOODPERALL_PIXEL &&
                       t->lastGoodCount / (t->lastGoodCount + t->lastBadCount) > MIN_GOODPERGOODBAD_PIXEL;
  fill_result(t, lsdm::se3d_from_f(referenceToFrame), out);
  return LSDHIP_OK;
}

// SE3Tracker::trackFrameOnPermaref (SE3Tracker.cpp:162-272) for n permanent references against frames[j] in the same
// launches — the shape of TrackableKeyFrameSearch::findRePositionCandidate and of the relocaliser, which test many
// keyframes against one new frame.  pos / colvar: the references' level-4 point clouds
// concatenated (counts[j] points each); refToFrame: n x 7; results: n.
extern "C" int lsdhip_tracker_track_permaref_batch(lsdhip_tracker* t, int n, const float* pos, const float* colvar, const int* counts,
                                                   lsdhip_frame** frames, const double* refToFrame, lsdhip_track_result* results) {
  if (!t || n <= 0 || !pos || !colvar || !counts || !frames || !refToFrame || !results) return LSDHIP_E_ARG;
  lsdhip_ctx* c = t->ctx;
  LSD_CTX_LOCK(c);
  LsdTrackJobScope tjob_(c, true);
  if (tjob_.rc) return tjob_.rc;
  HIPCHK(hipSetDevice(c->device));
  int total = 0;
  for (int j = 0; j < n; j++) { if (counts[j] <= 0 || !frames[j]) return LSDHIP_E_ARG; total += counts[j]; }
  int rc = upload_points(t, pos, colvar, total);
  if (rc) return rc;
  rc = batch_reserve(t, n);
  if (rc) return rc;
  batch_begin(t, n);
  const int L = LSD_QUICK_KF_CHECK_LVL;
  int off = 0;
  for (int j = 0; j < n; j++) {
    TrackJob& job = t->h_bjobs[j];
    fill_job_common(t, job);
    fill_level(t, job, L, nullptr, frames[j], t->d_pts + (size_t)off * 3, t->d_pts + (size_t)t->pts_capacity * 3 + (size_t)off * 2, counts[j]);
    job.lv[L].lambdaInitial = t->lambdaInitialTestTrack; job.lv[L].stepSizeMin = t->stepSizeMinTestTrack;
    job.lv[L].convergenceEps = t->convergenceEpsTestTrack; job.lv[L].maxIts = (int)t->maxItsTestTrack;
    job.lastLevel = L;
    job.topLevel = L;
    job.trackFrameSemantics = 0;
    job.evalOnly = 0;
    job.T0 = lsdm::se3f_from_d(lsdm::se3d_from7(refToFrame + 7 * (size_t)j));
    job.aff_a0 = 1.0f; job.aff_b0 = 0.0f;
    off += counts[j];
  }
  rc = batch_run(t, n);
  if (rc) return rc;
  int rcAll = LSDHIP_OK;
  for (int j = 0; j < n; j++) {
    const TrackSummary* S = &t->h_bsummary[j];
    t->numEvaluations = S->numEvaluations; t->numWarpUpdates = S->numWarpUpdates;
    t->pointUsage = S->pointUsage; t->lastGoodCount = S->goodCount; t->lastBadCount = S->badCount; t->lastMeanRes = S->meanRes;
    t->affineEstimation_a = S->aff_a; t->affineEstimation_b = S->aff_b;
    t->lastResidual = S->lastResidual;
    if (S->diverged) {
      t->diverged = true; t->trackingWasGood = false;
      fill_result(t, identity_d(), &results[j]);
      rcAll = LSDHIP_DIVERGED;
      continue;
    }
    t->diverged = false;
    t->trackingWasGood = t->lastGoodCount / (c->wl[L] * c->hl[L]) > MIN_GOODPERALL_PIXEL &&
                         t->lastGoodCount / (t->lastGoodCount + t->lastBadCount) > MIN_GOODPERGOODBAD_PIXEL;
    lsdm::SE3fH T;
    T.q = {S->q[0], S->q[1], S->q[2], S->q[3]};
    T.t[0] = S->t[0]; T.t[1] = S->t[1]; T.t[2] = S->t[2];
    fill_result(t, lsdm::se3d_from_f(T), &results[j]);
  }
  return rcAll;
}

extern "C" int lsdhip_tracker_check_overlap(lsdhip_tracker* t, const float* pos, int n, const double refToFrame[7], float* usage_out) {
  if (!t || !pos || n <= 0 || !refToFrame || !usage_out) return LSDHIP_E_ARG;
  lsdhip_ctx* c = t->ctx;
  LSD_CTX_LOCK(c);
  LsdTrackJobScope tjob_(c, true);
  if (tjob_.rc) return tjob_.rc;
  HIPCHK(hipSetDevice(c->device));
  int rc = upload_points(t, pos, nullptr, n);
  if (rc) return rc;
  lsdm::SE3fH T = lsdm::se3f_from_d(lsdm::se3d_from7(refToFrame));
  const int L = LSD_QUICK_KF_CHECK_LVL;
  EvalCtx a;
  memset(&a, 0, sizeof(a));
  const LevelIntr& in = c->intr[L];
  a.w = c->wl[L]; a.h = c->hl[L];
  a.fx = in.fx; a.fy = in.fy; a.cx = in.cx; a.cy = in.cy;
  lsdm::quatf_to_rot(T.q, a.R);
  a.t[0] = T.t[0]; a.t[1] = T.t[1]; a.t[2] = T.t[2];
  hipLaunchKernelGGL(k_overlap, dim3(1), dim3(256), 0, c->stream, t->d_pts, n, a, t->d_summary->sums);
  HIPCHK(hipStreamSynchronize(c->stream));
  t->pointUsage = t->h_summary->sums[0] / (float)n;
  *usage_out = t->pointUsage;
  return LSDHIP_OK;
}
