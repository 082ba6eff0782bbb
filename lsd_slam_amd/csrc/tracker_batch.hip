// Throughput mode of the SE3 tracker (batches of >= 8 jobs, lsdhip_tracker_track_batch): the residual evaluation launch.  gfx950 only.
//
// With enough independent jobs per launch the dependent-launch latency of a single sequence no longer matters and the evaluation
// is what BASELINE.json's north_star describes: coalesced loads of the keyframe planes, LDS-staged image tiles for the bilinear
// sampling, lane-local accumulation, one workgroup reduction per strip.  A step of a batch = k_track_step<.., TS_LM> (one workgroup
// per job: finish the pending evaluation, LM decision, publish the pose) + this kernel (evaluate the published pose of every job).
//
// Workgroup = (strip of tilePx consecutive pixels of the keyframe level, job).  The strip is processed in chunks of 1024 pixels
// (3.2 rows of a 320-pixel level), per chunk:
//   1. the chunk's keyframe planes (idepthVar, idepth, image) are read as one float4 per lane and plane — fully coalesced, each byte
//      of the planes is read exactly once;
//   2. the valid reference pixels (semi-dense: 30-60 %) are compacted into an LDS list of 16-byte entries (x | y << 16, 1 / idepth,
//      colour, variance) in a fixed order (wave, pixel slot, lane), so the evaluation runs with all lanes busy and re-reads nothing
//      from HBM;
//   3. every lane warps its (at most 4) list entries to the pose under evaluation; the workgroup takes the minimum and maximum image
//      row any in-image point samples;
//   4. the rows [min, max + 1] of the tracked frame's (gx, gy, I) texels are staged into LDS with coalesced 16-byte loads — the
//      IMAGE TILE of this chunk (the inter-frame motion only shifts it; its height is the chunk's 3-4 rows plus the spread of the
//      motion over the chunk) — as three planes, so that the four taps of getInterpolatedElement43 (C/util/globalFuncs.h:63-77) are
//      LDS reads; a tile that would not fit (large rotation, wide levels) falls back to global loads for that chunk;
//   5. residual / weights / normal equations of the lane's points in the reference's operation order (track_device.hpp), summed in
//      41 registers across all chunks of the strip.
// One workgroup reduction per strip -> sums[tile], topkey[tile] (the finishing launch re-evaluates the <= 3 tail points of the SSE
// tail drop, as on every multi-pass level).  The only global loads left inside the evaluation are the fallback taps; the only
// stores the refPixelWasGood bytes.
#include "track_device.hpp"
#include "tracker_batch.hpp"

#define TB_BLOCK 256
#define TB_CHUNK 1024           // pixels per chunk (4 per lane)
#define TB_WCAP 2880            // texels of the staged image tile (three float planes: 33.75 KB; 9 rows of a 320-pixel level)

__global__ __launch_bounds__(TB_BLOCK, 3) void k_track_eval_tiles(const TrackJob* __restrict__ jobs, const TrackState* __restrict__ st2, TrackScratch sc,
                                                               int parity) {
  constexpr int BLOCK = TB_BLOCK, WAVES = BLOCK / 64;
  constexpr int CPP = RS_END, RSLICE = BLOCK / CPP, RRUN = (BLOCK + RSLICE - 1) / RSLICE;
  // LDS: list (16 KB) + image tile (36 KB) during the chunks; the reduction scratch (42 KB) reuses the same bytes afterwards
  constexpr int LIST_WORDS = TB_CHUNK * 4, WIN_WORDS = 3 * TB_WCAP, RED_WORDS = CPP * (BLOCK + 1) + 8;
  constexpr int POOL_WORDS = LIST_WORDS + WIN_WORDS > RED_WORDS ? LIST_WORDS + WIN_WORDS : RED_WORDS;
  __shared__ __attribute__((aligned(16))) float s_pool[POOL_WORDS];
  __shared__ float s_sum[RSLICE][64];
  __shared__ int s_wtop[WAVES][3];
  __shared__ int s_top[3];
  __shared__ int s_cnt[WAVES];
  __shared__ int s_rowmin[WAVES], s_rowmax[WAVES];
  __shared__ float s_pose[16];          // R[9], t[3], aff_a, aff_b
  __shared__ int s_lvl;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const size_t j = blockIdx.y, rows = (size_t)sc.max_rows;
  const TrackJob& job = jobs[j];
  const TrackState* S = st2 + 2 * j + parity;
  if (tid == 0) s_lvl = S->done ? -1 : S->level;
  if (tid < 9) s_pose[tid] = S->R[tid];
  else if (tid < 12) s_pose[tid] = S->t[tid - 9];
  else if (tid == 12) s_pose[12] = S->aff_a;
  else if (tid == 13) s_pose[13] = S->aff_b;
  __syncthreads();
  const int level = s_lvl;
  if (level < 0) return;
  const TrackLevel& L = job.lv[level];
  const int nb = L.nblocks, tilePx = L.tilePx;
  if ((int)blockIdx.x >= nb) return;
  const int tile = xcd_tile((int)blockIdx.x, nb);
  EvalCtx a;
  a.kf_idepth = (gfloat*)L.kf_idepth; a.kf_idepthVar = (gfloat*)L.kf_idepthVar; a.kf_image = (gfloat*)L.kf_image; a.fr_grad = (gfloat*)L.fr_grad;
  a.pts_pos = nullptr; a.pts_colvar = nullptr; a.npts = -1; a.w = L.w; a.h = L.h;
  a.fx = L.fx; a.fy = L.fy; a.cx = L.cx; a.cy = L.cy; a.fxi = L.fxi; a.fyi = L.fyi; a.cxi = L.cxi; a.cyi = L.cyi;
#pragma unroll
  for (int i = 0; i < 9; i++) a.R[i] = s_pose[i];
#pragma unroll
  for (int i = 0; i < 3; i++) a.t[i] = s_pose[9 + i];
  a.aff_a = s_pose[12]; a.aff_b = s_pose[13];
  a.cameraPixelNoise2 = job.cameraPixelNoise2; a.var_weight = job.var_weight; a.huber_half = job.huber_half;
  gbyte* wasGood = (gbyte*)(L.writeMask ? job.wasGood : nullptr);
  const int w = a.w, work = a.w * a.h;
  const float inv_w = 1.0f / (float)w;
  v4f* s_list = (v4f*)s_pool;                       // [TB_CHUNK]: (xy bits, 1 / idepth, colour, variance)
  float* s_gx = s_pool + LIST_WORDS;                // [TB_WCAP] each
  float* s_gy = s_gx + TB_WCAP;
  float* s_gi = s_gy + TB_WCAP;

  float acc[RS_END];
#pragma unroll
  for (int k = 0; k < RS_END; k++) acc[k] = 0.f;
  int key0 = -1, key1 = -1, key2 = -1;
  const int nchunk = (tilePx + TB_CHUNK - 1) / TB_CHUNK;   // strips are multiples of 256 pixels: the last chunk may be partial
  const int stripEnd = min(tile * tilePx + tilePx, work);
  const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
  // planes of the first chunk; the next chunk's are requested before the current one is evaluated
  int i0 = tile * tilePx + (tid << 2);
  v4f vv = zero4, dd = zero4, ii = zero4;
  if (nchunk > 0 && i0 < stripEnd) { vv = *(gv4f*)(a.kf_idepthVar + i0); dd = *(gv4f*)(a.kf_idepth + i0); ii = *(gv4f*)(a.kf_image + i0); }
  for (int c = 0; c < nchunk; c++) {
    // ---- 2. compaction of this chunk's valid reference pixels ----------------------------------------------------------------
    int y = (int)((float)i0 * inv_w);
    int x = i0 - y * w;
    if (x < 0) { y--; x += w; }
    if (x >= w) { y++; x -= w; }
    const float vk[4] = {vv.x, vv.y, vv.z, vv.w}, dk[4] = {dd.x, dd.y, dd.z, dd.w}, ik[4] = {ii.x, ii.y, ii.z, ii.w};
    bool ok[4];
    int xs[4], ys[4];
    int wcount = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      xs[k] = x; ys[k] = y;
      ok[k] = i0 < stripEnd && !(x < 1 || x >= w - 1 || y < 1 || y >= a.h - 1) && !(vk[k] <= 0 || dk[k] == 0);
      wcount += __popcll(__ballot(ok[k]));
      if (++x >= w) { x = 0; y++; }
    }
    if (lane == 0) s_cnt[wave] = wcount;
    __syncthreads();                                 // (also: the previous chunk's evaluation is over — list and tile are free)
    int pos = 0, total = 0;
#pragma unroll
    for (int wv = 0; wv < WAVES; wv++) { const int cw = s_cnt[wv]; if (wv < wave) pos += cw; total += cw; }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const unsigned long long bal = __ballot(ok[k]);
      if (ok[k]) {
        const int p = pos + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        v4f e;
        e.x = __int_as_float(xs[k] | (ys[k] << 16)); e.y = 1.0f / dk[k]; e.z = ik[k]; e.w = vk[k];
        s_list[p] = e;
      }
      pos += __popcll(bal);
    }
    // the next chunk's planes travel while this one is evaluated
    const int i0n = i0 + TB_CHUNK;
    if (c + 1 < nchunk && i0n < stripEnd) { vv = *(gv4f*)(a.kf_idepthVar + i0n); dd = *(gv4f*)(a.kf_idepth + i0n); ii = *(gv4f*)(a.kf_image + i0n); }
    else { vv = zero4; dd = zero4; ii = zero4; }
    i0 = i0n;
    __syncthreads();
    if (total == 0) continue;
    // ---- 3. warp of this lane's entries; rows of the frame the chunk samples ----------------------------------------------------
    PointWarp q[4];
    v4f ent[4];
    bool live[4];
    int rmin = 0x7fffffff, rmax = -1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int e = tid + k * BLOCK;
      live[k] = e < total;
      ent[k] = s_list[live[k] ? e : 0];
      const int xy = __float_as_int(ent[k].x);
      const int px_ = xy & 0xffff, py_ = xy >> 16;
      const float inv = ent[k].y;
      eval_warp(a, inv * (a.fxi * px_ + a.cxi), inv * (a.fyi * py_ + a.cyi), inv * 1.0f, q[k]);
      if (live[k] && q[k].in_image) { const int iy = (int)q[k].v_new; rmin = min(rmin, iy); rmax = max(rmax, iy); }
    }
    {
      // wave minimum / maximum through DPP-free ballot-style reduction (values are small non-negative row numbers)
      int mn = rmin, mx = rmax;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        mn = min(mn, __shfl_xor(mn, off));
        mx = max(mx, __shfl_xor(mx, off));
      }
      if (lane == 0) { s_rowmin[wave] = mn; s_rowmax[wave] = mx; }
    }
    __syncthreads();
    int row0 = s_rowmin[0], row1 = s_rowmax[0];
#pragma unroll
    for (int wv = 1; wv < WAVES; wv++) { row0 = min(row0, s_rowmin[wv]); row1 = max(row1, s_rowmax[wv]); }
    const int nrows = row1 + 2 - row0;                // the taps of row iy touch rows iy and iy + 1
    const bool useWin = row1 >= 0 && nrows * w <= TB_WCAP;
    // ---- 4. the image tile: rows [row0, row0 + nrows) of the frame's texels -> LDS planes -------------------------------------
    if (useWin) {
      const int ntex = nrows * w;
      gv4f* src = (gv4f*)(a.fr_grad + 4 * (size_t)(row0 * w));
      for (int idx = tid; idx < ntex; idx += BLOCK) {
        const v4f tx = src[idx];
        s_gx[idx] = tx.x; s_gy[idx] = tx.y; s_gi[idx] = tx.z;
      }
    }
    __syncthreads();
    // ---- 5. evaluation of this lane's entries ----------------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (live[k]) {
        const int xy = __float_as_int(ent[k].x);
        const int px_ = xy & 0xffff, py_ = xy >> 16;
        const int pi = py_ * w + px_;
        acc[RS_NREF] += 1.f;
        if (!q[k].in_image) {
          if (wasGood) wasGood[pi] = 0;
        } else {
          PointTexels t;
          if (useWin) {
            const int b0 = ((int)q[k].v_new - row0) * w + (int)q[k].u_new;
            t.t00 = {s_gx[b0], s_gy[b0], s_gi[b0]};
            t.t10 = {s_gx[b0 + 1], s_gy[b0 + 1], s_gi[b0 + 1]};
            t.t01 = {s_gx[b0 + w], s_gy[b0 + w], s_gi[b0 + w]};
            t.t11 = {s_gx[b0 + w + 1], s_gy[b0 + w + 1], s_gi[b0 + w + 1]};
          } else {
            eval_fetch(a, q[k], true, t);
          }
          PointOut o;
          eval_finish(a, q[k], t, ent[k].y * 1.0f, ent[k].z, ent[k].w, o);
          if (wasGood) wasGood[pi] = o.good ? 1 : 0;
          top3_insert(px_ * a.h + py_, key0, key1, key2);
          accumulate_point(o, acc);
        }
      }
    }
  }
  __syncthreads();                                   // list / tile -> reduction scratch
  // ---- workgroup reduction (the transposed form of k_track_step) -> sums[tile], topkey[tile] ---------------------------------------
  float* s_red = s_pool;
  float* sums_out = sc.sums + j * 2 * RS_COLS * rows + (size_t)parity * RS_COLS * rows;
  int4* topkey_out = sc.topkey + j * 2 * rows + (size_t)parity * rows;
#pragma unroll
  for (int k = 0; k < CPP; k++) s_red[k * (BLOCK + 1) + tid] = acc[k];
  __syncthreads();
  {
    const int slice = tid / CPP, k = tid - slice * CPP;
    if (slice < RSLICE) {
      const float* row = s_red + k * (BLOCK + 1);
      const int j0 = slice * RRUN;
      float v[RRUN];
#pragma unroll
      for (int jj = 0; jj < RRUN; jj++) v[jj] = row[j0 + jj];   // the last run reads 2 words of the next row (allocated)
      float s = 0.f;
#pragma unroll
      for (int jj = 0; jj < RRUN; jj++) s += (j0 + jj < BLOCK) ? v[jj] : 0.f;
      s_sum[slice][k] = s;
    }
  }
  __syncthreads();
  if (tid < CPP) {
    float s = s_sum[0][tid];
#pragma unroll
    for (int sl = 1; sl < RSLICE; sl++) s += s_sum[sl][tid];
    sums_out[(size_t)tile * RS_COLS + tid] = s;
  }
  block_top3(key0, key1, key2, s_wtop, s_top);
  if (tid == 0) topkey_out[tile] = make_int4(s_top[0], s_top[1], s_top[2], -1);
}

int lsd_track_eval_tiles_launch(lsdhip_tracker* t, int grid, int n, const TrackScratch& sc, int parity) {
  hipLaunchKernelGGL(k_track_eval_tiles, dim3(grid, n), dim3(TB_BLOCK), 0, t->ctx->stream, (const TrackJob*)t->d_bjobs, (const TrackState*)t->d_bstate, sc,
                     parity);
  HIPCHK(hipGetLastError());
  return LSDHIP_OK;
}
