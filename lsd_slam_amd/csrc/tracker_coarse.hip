// Coarse-level cluster kernel of the SE3 tracker (opt-in execution form, lsdhip_tracker_set_coarse).  gfx950 only.
#include "track_device.hpp"
#include "tracker_coarse.hpp"

// =====================================================================================================================
// Coarse-level cluster kernel: ONE launch runs the LM loops of the coarsest pyramid levels of SE3Tracker::trackFrame
// (SE3Tracker.cpp:316-447), reject-chain speculation included.
//
// The launch-per-step chain above pays per step a kernel boundary and dispatch (~4.5 us), a re-read of every tile's partial
// row written by other XCDs (2-3.5 us) and the cold start of a few hundred workgroups — for levels of a few thousand pixels.
// Here the workgroups stay resident for the whole coarse part of a job:
//   * workgroup (c, g): trial c < nt of every speculative set, strip g of the level = its pixels [512 g, 512 g + 512), one pixel
//     per lane.  A lane loads its pixel of every cluster level ONCE per job (x, y, 1 / idepth, colour, variance: the
//     pose-independent part of TrackingReference::makePointCloud stays in registers), so an evaluation touches only the tracked
//     frame's texels.  (A single workgroup per trial over an LDS point list measured 10 us per level-3 evaluation: one point
//     evaluation is a ~3000-cycle dependent chain, and several per lane leave nothing to hide it behind.)
//   * one evaluation = warp + texel fetch + residual / weights / normal equations of the lane's point at the trial's pose,
//     workgroup top-3 order keys in the shadow of the texel loads, workgroup reduction of the 41 sums;
//   * ONE exchange per step: every active workgroup publishes a row of 8-byte {tag, value} granules — 41 partial sums, its 3
//     largest order keys and their K2 error terms, the increment / pose of its trial, and (read only on demand) the 3 x 29 K2/K3
//     contributions of those points for the SSE tail drop — with relaxed agent-scope stores: the datum is its own flag, no fences
//     (cdna_hip_programming.md G16 form R2).  Thread (trial, column) of every workgroup reads that column of all strips' rows
//     (all loads in flight together: one round trip) and adds them in strip order;
//   * every workgroup then takes the same decision from the same numbers: merge of the strips' order keys per trial, lane-parallel
//     scan for the trial that stops the reference's loop, lambda / incTry / counters advanced past the plain rejections before
//     it, tail contributions of that trial fetched, ONE LM step (lm_wave) — and workgroup (c, .) derives the pose of trial c of
//     the next set by the closed-form lambda recurrence: no second hop.
// Decisions, evaluation counts and poses are those of the same kernel run with one trial per step, bit for bit (same strips,
// same summation order).  Every spin is bounded: on a time-out the kernel reports done = 2 and the host reruns the job on the
// k_track_step chain.  The finest level of a job never runs here (it writes refPixelWasGood): the kernel hands the state to the
// chain through st_out.
// =====================================================================================================================
__device__ __forceinline__ void ct_store(ct_u64* p, unsigned tag, unsigned val) {
  __hip_atomic_store((ct_gu64*)p, ((ct_u64)tag << 32) | val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ ct_u64 ct_load(const ct_u64* p) {
  return __hip_atomic_load((ct_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the K2/K3 contributions of one point, in the order of the tail-drop tables (werr | 21 A | 6 b | err)
__device__ __forceinline__ void point_contrib(const PointOut& o, float* dst) {
  dst[0] = o.werr;
  int k = 1;
#pragma unroll
  for (int r = 0; r < 6; r++) {
    const float Jw = o.J[r] * o.w;
#pragma unroll
    for (int c = r; c < 6; c++) dst[k++] = Jw * o.J[c];
  }
  const float resw = o.res * o.w;
#pragma unroll
  for (int r = 0; r < 6; r++) dst[k++] = resw * o.J[r];
  dst[k] = resw * o.res;
}

#ifdef LSD_PHASE_TRACE
#define CT_MARK(k) do { if (b == 0 && tid == 0) ctr_[k] = clock64(); } while (0)
#else
#define CT_MARK(k) do { } while (0)
#endif
__global__ __launch_bounds__(CT_BLOCK) void k_track_coarse(TrackJob job, CoarsePlan plan, TrackState* __restrict__ st_out,
                                                           ct_u64* __restrict__ rows, unsigned salt, TrackSummary* __restrict__ out
#ifdef LSD_PHASE_TRACE
                                                           , unsigned long long* __restrict__ ctrace
#endif
                                                           ) {
  constexpr int BLOCK = CT_BLOCK, WAVES = BLOCK / 64, HALF = 256;
  constexpr int SW = sizeof(TrackState) / 4;
  __shared__ TrackState S;
  __shared__ LmShared sh;
  __shared__ LmPar s_par;
  __shared__ float s_red[RS_END * (HALF + 1) + 8];       // workgroup reduction (transposed, conflict-free both ways)
  __shared__ float s_sum[HALF / RS_END][64];
  __shared__ float s_tot[LSD_SPEC_MAX][48];              // per trial: fixed-order sums over its strips
  __shared__ int s_keys[LSD_SPEC_MAX][CT_GMAX][3];       // per (trial, strip): its three largest order keys
  __shared__ float s_kw[LSD_SPEC_MAX][CT_GMAX][3];       // ... and the K2 error terms of those points
  __shared__ float s_rec[LSD_SPEC_MAX][32];              // per trial: increment / pose record
  __shared__ int s_wtop[WAVES][3];
  __shared__ int s_top[3];
  __shared__ float s_contrib[3][32];
  __shared__ float s_sub[3][32];
  __shared__ int s_flag;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.x;
  const int nt = plan.nt;
  const int mycand = b / plan.gmax, g = b - mycand * plan.gmax;
  const unsigned tagbase = salt << 12;
#ifdef LSD_PHASE_TRACE
  // developer build: 16 words per step — 0 step start, 1 evaluated, 2 published, 3 gathered + summed, 4 decided (shader clock),
  // 5 level, 6 trials, 7 strips, 8 / 9 wall clock at kernel entry / step end, 10 points loaded, 11 step ordinal, 12 tail fetched
  unsigned long long* ctr_ = ctrace + 1;
  unsigned long long ct_t0 = 0, ct_tl = 0, ct_w0 = 0;
  if (b == 0 && tid == 0) { ct_t0 = clock64(); ct_w0 = wall_clock64(); }
#endif

  // ---- every workgroup starts from the same state ---------------------------------------------------------------------
  if (tid == 0) {
    S.T = job.T0;
    set_eval_pose(S, job.T0);
    S.aff_a = job.aff_a0; S.aff_b = job.aff_b0; S.aff_a_lastIt = job.aff_a0; S.aff_b_lastIt = job.aff_b0;
    S.lastErr = 0; S.LM_lambda = 0; S.last_residual = 0;
    S.level = job.topLevel; S.iteration = 0; S.incTry = 0; S.phase = 0; S.pending = 0;
    S.done = 0; S.diverged = 0; S.numEvaluations = 0; S.numWarpUpdates = 0;
    S.ncand = 1; S.lastCand = 0; S.numLaunches = 0; S.coarseSteps = 0; S.coarseBytes = 0;
    S.pointUsage = 0; S.goodCount = 0; S.badCount = 0; S.meanRes = 0;
    S.bytes = 0;
    for (int l = 0; l < LSD_LEVELS; l++) S.levelEvals[l] = 0;
    s_flag = 1;
  }
  if (tid < 36) S.A[tid] = 0;
  if (tid < 6) { S.b[tid] = 0; S.inc[tid] = 0; }

  // ---- this lane's reference pixel of every cluster level (slot s <-> level topLevel - s): loaded once, all levels' loads in
  // flight together ----------------------------------------------------------------------------------------------------------
  bool pvalid[CT_LEVELS];
  float pinv[CT_LEVELS], pI[CT_LEVELS], pvar[CT_LEVELS];
  int pxy[CT_LEVELS];
  {
    float vv[CT_LEVELS], dd[CT_LEVELS], ii[CT_LEVELS];
#pragma unroll
    for (int sl = 0; sl < CT_LEVELS; sl++) {
      const int l = job.topLevel - sl;
      vv[sl] = 0.f; dd[sl] = 0.f; ii[sl] = 0.f;
      if (l >= plan.low) {
        const TrackLevel& L = job.lv[l];
        const int i = g * BLOCK + tid;
        if (i < L.w * L.h) { vv[sl] = L.kf_idepthVar[i]; dd[sl] = L.kf_idepth[i]; ii[sl] = L.kf_image[i]; }
      }
    }
#pragma unroll
    for (int sl = 0; sl < CT_LEVELS; sl++) {
      const int l = job.topLevel - sl;
      pvalid[sl] = false; pinv[sl] = 0.f; pI[sl] = ii[sl]; pvar[sl] = vv[sl]; pxy[sl] = 0;
      if (l >= plan.low) {
        const TrackLevel& L = job.lv[l];
        const int w = L.w, h = L.h;
        const int i = g * BLOCK + tid;
        const int y = i / w, x = i - y * w;
        pvalid[sl] = i < w * h && !(x < 1 || x >= w - 1 || y < 1 || y >= h - 1) && !(vv[sl] <= 0 || dd[sl] == 0);
        pinv[sl] = 1.0f / dd[sl];
        pxy[sl] = x | (y << 16);
      }
    }
  }

  __syncthreads();
#ifdef LSD_PHASE_TRACE
  if (b == 0 && tid == 0) ct_tl = clock64();
#endif
  int step = 0;
  while (true) {
    // ---- leaving: job finished, or the next level belongs to the k_track_step chain (reads st_out[0], pending = 0) -------
    if (S.done || S.level < plan.low) {
      if (b == 0) {
        copy_words<SW>(st_out, &S, tid, BLOCK);
        copy_words<SW>(st_out + 1, &S, tid, BLOCK);
      }
      break;
    }
    const int level = S.level;
#ifdef LSD_PHASE_TRACE
    if (b == 0 && tid == 0) {
      const unsigned long long n = ctrace[0];
      ctrace[0] = n + 1;
      ctr_ = ctrace + 1 + (n % 4096) * 16;
      for (int k = 0; k < 16; k++) ctr_[k] = 0;
      ctr_[5] = (unsigned long long)level; ctr_[8] = ct_w0; ctr_[10] = step == 0 ? ct_tl - ct_t0 : 0; ctr_[11] = (unsigned long long)step;
    }
    CT_MARK(0);
#endif
    const bool trialPhase = S.phase == 1;
    const int ncand = trialPhase ? (S.ncand < 1 ? 1 : (S.ncand > nt ? nt : S.ncand)) : 1;
    const int lw = job.lv[level].w, lh = job.lv[level].h;
    const int G = (lw * lh + BLOCK - 1) / BLOCK;            // strips of this level (<= plan.gmax)
    const unsigned epoch = tagbase | ((unsigned)(step + 1) & 0xFFFu);
    ct_u64* const rowsE = rows + (size_t)(step & 1) * (LSD_SPEC_MAX * CT_GMAX) * CT_ROW;   // double-buffered by step parity
    if (tid == BLOCK - 1) stage_lm_par(job, level, s_par, plan.trials[level] > 1 ? plan.trials[level] : 1);

    if (mycand < ncand && g < G) {
      // ---- evaluate this lane's point of `level` at this workgroup's trial pose ------------------------------------------
      EvalCtx a;
      make_ctx_dev(job, S, level, a);
      const int sl = job.topLevel - level;
      const bool valid = sl == 0 ? pvalid[0] : (sl == 1 ? pvalid[1] : pvalid[2]);
      const float inv = sl == 0 ? pinv[0] : (sl == 1 ? pinv[1] : pinv[2]);
      const float I_ref = sl == 0 ? pI[0] : (sl == 1 ? pI[1] : pI[2]);
      const float var = sl == 0 ? pvar[0] : (sl == 1 ? pvar[1] : pvar[2]);
      const int xy = sl == 0 ? pxy[0] : (sl == 1 ? pxy[1] : pxy[2]);
      const int x = xy & 0xffff, y = xy >> 16;
      float acc[RS_END];
#pragma unroll
      for (int k = 0; k < RS_END; k++) acc[k] = 0.f;
      PointWarp q;
      PointTexels tx;
      eval_warp(a, inv * (a.fxi * x + a.cxi), inv * (a.fyi * y + a.cyi), inv * 1.0f, q);
      const bool inimg = valid && q.in_image;
      eval_fetch(a, q, inimg, tx);
      const int key = x * a.h + y;
      block_top3(inimg ? key : -1, -1, -1, s_wtop, s_top);       // two barriers, in the shadow of the texel loads
      const int top0 = s_top[0], top1 = s_top[1], top2 = s_top[2];
      if (valid) {
        acc[RS_NREF] = 1.f;
        if (q.in_image) {
          PointOut o;
          eval_finish(a, q, tx, inv * 1.0f, I_ref, var, o);
          accumulate_point(o, acc);
          const int r = key == top0 ? 0 : (key == top1 ? 1 : (key == top2 ? 2 : -1));
          if (r >= 0) point_contrib(o, s_contrib[r]);
        }
      }
      // ---- workgroup reduction: upper half onto lower half, then the transposed 256-lane form of k_track_step: thread
      // (slice, k) adds a run of 43 lanes of column k, 41 threads add the 6 slices ---------------------------------------
      if (tid >= HALF) {
#pragma unroll
        for (int k = 0; k < RS_END; k++) s_red[k * (HALF + 1) + (tid - HALF)] = acc[k];
      }
      __syncthreads();
      if (tid < HALF) {
#pragma unroll
        for (int k = 0; k < RS_END; k++) s_red[k * (HALF + 1) + tid] += acc[k];
      }
      __syncthreads();
      constexpr int RSLICE = HALF / RS_END;                  // 6
      constexpr int RRUN = (HALF + RSLICE - 1) / RSLICE;     // 43
      {
        const int slice = tid / RS_END, k = tid - slice * RS_END;
        if (slice < RSLICE) {
          const float* row = s_red + k * (HALF + 1);
          const int j0 = slice * RRUN;
          float v[RRUN];
#pragma unroll
          for (int j = 0; j < RRUN; j++) v[j] = row[j0 + j];   // the last run reads 2 words of the next row (allocated)
          float s = 0.f;
#pragma unroll
          for (int j = 0; j < RRUN; j++) s += (j0 + j < HALF) ? v[j] : 0.f;
          s_sum[slice][k] = s;
        }
      }
      __syncthreads();
      CT_MARK(1);
#ifdef LSD_PHASE_TRACE
      if (b == 0 && tid == 0) { ctr_[6] = (unsigned long long)ncand; ctr_[7] = (unsigned long long)G; }
#endif
      // ---- publish this workgroup's row ----------------------------------------------------------------------------------
      ct_u64* myrow = rowsE + (size_t)(mycand * CT_GMAX + g) * CT_ROW;
      if (tid < RS_END) {
        float s = s_sum[0][tid];
#pragma unroll
        for (int sl2 = 1; sl2 < RSLICE; sl2++) s += s_sum[sl2][tid];
        ct_store(myrow + tid, epoch, __float_as_uint(s));
      } else if (tid < RS_END + 3) ct_store(myrow + tid, epoch, (unsigned)s_top[tid - RS_END]);
      else if (tid < RS_END + 6) { const int r = tid - RS_END - 3; ct_store(myrow + tid, epoch, __float_as_uint(s_top[r] >= 0 ? s_contrib[r][0] : 0.f)); }
      else if (tid >= 64 && tid < 64 + 25) {
        if (g == 0 && trialPhase) {
          const int j = tid - 64;
          float v;
          if (j < 6) v = S.inc[j];
          else if (j < 13) v = ((const float*)&S.Tn)[j - 6];
          else if (j < 22) v = S.R[j - 13];
          else v = S.t[j - 22];
          ct_store(myrow + CT_REC0 + j, epoch, __float_as_uint(v));
        }
      } else if (tid >= 128 && tid < 128 + 87) {
        const int r = (tid - 128) / 29, j = (tid - 128) - r * 29;
        ct_store(myrow + CT_SUB0 + r * 29 + j, epoch, __float_as_uint(s_top[r] >= 0 ? s_contrib[r][j] : 0.f));
      }
    }
    CT_MARK(2);
    // ---- gather + sum: thread (trial c, head column k) reads column k of the rows of all strips of trial c — every load in
    // flight at once, re-polling only what has not arrived — and adds them in strip order; the trials' records likewise -------
    {
      bool bad = false;
      if (tid < ncand * CT_HEAD) {
        const int c = tid / CT_HEAD, k = tid - c * CT_HEAD;
        const ct_u64* p0 = rowsE + (size_t)(c * CT_GMAX) * CT_ROW + k;
        unsigned v[CT_GMAX];
        unsigned pend = 0;
#pragma unroll
        for (int gg = 0; gg < CT_GMAX; gg++) {
          v[gg] = 0;
          if (gg < G) {
            const ct_u64 x = ct_load(p0 + (size_t)gg * CT_ROW);
            v[gg] = (unsigned)x;
            if ((unsigned)(x >> 32) != epoch) pend |= 1u << gg;
          }
        }
        for (unsigned spins = 0; pend != 0; spins++) {
          if (spins > CT_SPIN_LIMIT) { bad = true; break; }
          __builtin_amdgcn_s_sleep(1);
#pragma unroll
          for (int gg = 0; gg < CT_GMAX; gg++)
            if (pend & (1u << gg)) {
              const ct_u64 x = ct_load(p0 + (size_t)gg * CT_ROW);
              v[gg] = (unsigned)x;
              if ((unsigned)(x >> 32) == epoch) pend &= ~(1u << gg);
            }
        }
        if (k < RS_END) {
          float s = __uint_as_float(v[0]);
#pragma unroll
          for (int gg = 1; gg < CT_GMAX; gg++) s += gg < G ? __uint_as_float(v[gg]) : 0.f;
          s_tot[c][k] = s;
        } else if (k < RS_END + 3) {
#pragma unroll
          for (int gg = 0; gg < CT_GMAX; gg++) s_keys[c][gg][k - RS_END] = gg < G ? (int)v[gg] : -1;
        } else {
#pragma unroll
          for (int gg = 0; gg < CT_GMAX; gg++) s_kw[c][gg][k - RS_END - 3] = __uint_as_float(v[gg]);
        }
      } else if (trialPhase && tid >= 320 && tid < 320 + ncand * 25) {
        const int c = (tid - 320) / 25, j = (tid - 320) - c * 25;
        const ct_u64* p = rowsE + (size_t)(c * CT_GMAX) * CT_ROW + CT_REC0 + j;
        ct_u64 x = ct_load(p);
        for (unsigned spins = 0; (unsigned)(x >> 32) != epoch; spins++) {
          if (spins > CT_SPIN_LIMIT) { bad = true; break; }
          __builtin_amdgcn_s_sleep(1);
          x = ct_load(p);
        }
        s_rec[c][j] = __uint_as_float((unsigned)x);
      }
      if (bad) s_flag = 3;
    }
    __syncthreads();
    if (s_flag == 3) {                     // give up: the host reruns the job on the launch-per-evaluation chain
      if (b == 0 && tid == 0) { out->diverged = 0; __threadfence_system(); out->done = 2; }
      return;
    }
    CT_MARK(3);
    // ---- the decision, identically in every workgroup ---------------------------------------------------------------------
    if (wave == 0) {
      // lane c < ncand: merge of trial c's strips' order keys (the last M mod 4 in-image points in reference order = the largest keys)
      const int cc = tid < ncand ? tid : 0;
      int k0 = -1, k1 = -1, k2 = -1, e0 = 0, e1 = 0, e2 = 0;
      for (int gg = 0; gg < G; gg++)
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
          const int k = s_keys[cc][gg][rr], ek = gg * 3 + rr;
          const bool g0 = k > k0, g1 = k > k1, g2 = k > k2;
          k2 = g1 ? k1 : (g2 ? k : k2); e2 = g1 ? e1 : (g2 ? ek : e2);
          k1 = g0 ? k0 : (g1 ? k : k1); e1 = g0 ? e0 : (g1 ? ek : e1);
          k0 = g0 ? k : k0; e0 = g0 ? ek : e0;
        }
      const int nkeys = (k0 >= 0) + (k1 >= 0) + (k2 >= 0);
      const int Mc = (int)s_tot[cc][RS_M];
      int needc = Mc & 3;
      if (needc > nkeys) needc = nkeys;
      int pc = 0;
      if (trialPhase) {
        // back to the common view of the state: this workgroup's own trial moved incTry / inc / Tn / R / t
        const int it0 = S.incTry - mycand;
        float ws = s_tot[cc][RS_WERR];
        if (needc > 0) ws -= s_kw[cc][e0 / 3][e0 % 3];
        if (needc > 1) ws -= s_kw[cc][e1 / 3][e1 % 3];
        if (needc > 2) ws -= s_kw[cc][e2 / 3][e2 % 3];
        const float werrc = lm_werr(ws, Mc);
        const float* ic = s_rec[cc];
        const float i0 = ic[0], i1 = ic[1], i2 = ic[2], i3 = ic[3], i4 = ic[4], i5 = ic[5];
        const float incdot = (i0 * i0 + (i1 * i1 + i2 * i2)) + (i3 * i3 + (i4 * i4 + i5 * i5));
        const bool stop = Mc < s_par.minWarped || werrc < S.lastErr || !(incdot > s_par.stepSizeMin);
        const unsigned long long sm = __ballot(stop && tid < ncand);
        pc = sm ? (int)__ffsll((long long)sm) - 1 : ncand - 1;
        // SSE tail drop of the trial the loop stops at: its contributions wait in the rows of the strips that own the points —
        // one more round trip (only when M mod 4 != 0), issued now and collected after the bookkeeping below
      }
      const int need = rli(needc, pc);
      const int src0 = rli(e0, pc), src1 = rli(e1, pc), src2 = rli(e2, pc);
      ct_u64 tx0 = 0, tx1 = 0;
      const ct_u64 *tp0 = nullptr, *tp1 = nullptr;
      if (tid < need * 29) {
        const int r = tid / 29, j = tid - r * 29;
        const int src = r == 0 ? src0 : (r == 1 ? src1 : src2);
        tp0 = rowsE + (size_t)(pc * CT_GMAX + src / 3) * CT_ROW + CT_SUB0 + (src % 3) * 29 + j;
        tx0 = ct_load(tp0);
      }
      if (tid + 64 < need * 29) {
        const int qd = tid + 64;
        const int r = qd / 29, j = qd - r * 29;
        const int src = r == 0 ? src0 : (r == 1 ? src1 : src2);
        tp1 = rowsE + (size_t)(pc * CT_GMAX + src / 3) * CT_ROW + CT_SUB0 + (src % 3) * 29 + j;
        tx1 = ct_load(tp1);
      }
      if (trialPhase) {
        const int it0 = S.incTry - mycand;
        float lam = S.LM_lambda;
        for (int j = 0; j < pc; j++) lam = lm_lambda_fail(lam, it0 + j, s_par.lambdaFailFac);
        float skipped;
        {
          const float NR = s_tot[0][RS_NREF];
          const float wh = (float)s_par.w * (float)s_par.h;
          const float texels = 4.0f * NR < wh ? 4.0f * NR : wh;
          skipped = 20.0f * NR + (s_par.writeMask ? 5.0f * NR : 0.0f) + 12.0f * texels;
        }
        const float* rec = s_rec[pc];
        const int ne0 = S.numEvaluations, le0 = S.levelEvals[level];
        float bytes1 = S.bytes;
        for (int j = 0; j < pc; j++) bytes1 = bytes1 + skipped;
        const float r0 = rec[tid < 25 ? tid : 0];
        S.LM_lambda = lam;
        S.incTry = it0 + pc;
        S.numEvaluations = ne0 + pc;
        if (tid == 0) S.levelEvals[level] = le0 + pc;
        S.bytes = bytes1;
        if (tid < 6) S.inc[tid] = r0;
        else if (tid < 13) ((float*)&S.Tn)[tid - 6] = r0;
        else if (tid < 22) S.R[tid - 13] = r0;
        else if (tid < 25) S.t[tid - 22] = r0;
      }
      float col = tid < RS_END ? s_tot[pc][tid] : 0.f;
      if (need > 0) {
        bool bad = false;
        if (tp0) {
          for (unsigned spins = 0; (unsigned)(tx0 >> 32) != epoch; spins++) { if (spins > CT_SPIN_LIMIT) { bad = true; break; } tx0 = ct_load(tp0); }
          s_sub[tid / 29][tid % 29] = __uint_as_float((unsigned)tx0);
        }
        if (tp1) {
          for (unsigned spins = 0; (unsigned)(tx1 >> 32) != epoch; spins++) { if (spins > CT_SPIN_LIMIT) { bad = true; break; } tx1 = ct_load(tp1); }
          s_sub[(tid + 64) / 29][(tid + 64) % 29] = __uint_as_float((unsigned)tx1);
        }
        if (__any(bad)) s_flag = 3;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int subIdx = (tid == RS_WERR) ? 0 : ((tid >= RS_A0 && tid < RS_B0) ? 1 + tid - RS_A0 : ((tid >= RS_B0 && tid < RS_ERR) ? 22 + tid - RS_B0 : (tid == RS_ERR ? 28 : -1)));
        const int si = subIdx < 0 ? 0 : subIdx;
        const float sub0 = s_sub[0][si], sub1 = s_sub[1][si], sub2 = s_sub[2][si];
        if (subIdx >= 0) {
          if (need > 0) col -= sub0;
          if (need > 1) col -= sub1;
          if (need > 2) col -= sub2;
        }
      }
#ifdef LSD_PHASE_TRACE
      CT_MARK(12);
#endif
      if (tid < RS_NUM) sh.tot[tid] = col;
      S.coarseSteps = step + 1;
      lm_wave<true>(s_par, S, col, sh.tot, tid, b == 0 ? out : nullptr, nullptr, pc, mycand);
      S.coarseBytes = S.bytes;
    }
    __syncthreads();
    if (s_flag == 3) {
      if (b == 0 && tid == 0) { out->diverged = 0; __threadfence_system(); out->done = 2; }
      return;
    }
    CT_MARK(4);
#ifdef LSD_PHASE_TRACE
    if (b == 0 && tid == 0) ctr_[9] = wall_clock64();
#endif
    step++;
  }
}

// host side: one launch of the cluster kernel on the context's stream (called from track_device in tracker.hip)
int lsd_track_coarse_launch(lsdhip_tracker* t, const TrackJob& job, const CoarsePlan& plan) {
  lsdhip_ctx* c = t->ctx;
  t->ctSalt = (t->ctSalt + 1) & 0xFFFFFu;
  if (t->ctSalt == 0) {   // tags wrap: clear the granules so that no stale tag can match
    HIPCHK(hipMemsetAsync(t->d_ctrows, 0, lsd_track_coarse_rows_bytes(), c->stream));
    t->ctSalt = 1;
  }
#ifdef LSD_PHASE_TRACE
  hipLaunchKernelGGL(k_track_coarse, dim3(plan.nt * plan.gmax), dim3(CT_BLOCK), 0, c->stream, job, plan, t->d_state, t->d_ctrows, t->ctSalt,
                     t->d_summary, t->d_ctrace);
#else
  hipLaunchKernelGGL(k_track_coarse, dim3(plan.nt * plan.gmax), dim3(CT_BLOCK), 0, c->stream, job, plan, t->d_state, t->d_ctrows, t->ctSalt,
                     t->d_summary);
#endif
  HIPCHK(hipGetLastError());
  return LSDHIP_OK;
}
