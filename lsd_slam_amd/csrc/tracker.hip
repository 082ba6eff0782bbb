// SE3Tracker on the device: one fused "residual kernel" per evaluation (K0 point generation + K1 warp/sample/mask +
// K2 weights + K3 normal equations) followed by a one-workgroup finalize kernel; the Levenberg-Marquardt control loop
// stays on the host exactly as in the reference.  gfx950 only.
//
// Reference behaviour restated:
//   TrackingReference::makePointCloud   C/Tracking/TrackingReference.cpp:128-138  (points generated on the fly from
//                                        the keyframe's idepth / idepthVar / image planes, no compacted arrays)
//   SE3Tracker::calcResidualAndBuffers  C/Tracking/SE3Tracker.cpp:885-1029
//   SE3Tracker::calcWeightsAndResidualSSE  :492-575   (op order of the SSE path; _mm_rcp_ps -> IEEE 1/x)
//   SE3Tracker::calculateWarpUpdateSSE  :1033-1130 + LGS6::updateSSE C/Tracking/LGSX.h:328-386
//   SE3Tracker::trackFrame              :280-486      (host)
//   SE3Tracker::trackFrameOnPermaref    :162-272, checkPermaRefOverlap :121-157
//
// Quirks kept on purpose: the SSE loops ignore the last size%4 in-image points (in the reference's
// x-outer point order) for K2/K3 — emulated by the finalize kernel; LGS6::updateSSE counts 6 constraints per group of 4.
//
// Data layout: keyframe planes idepth/idepthVar/image (3 x 4 B per pixel, row-major, coalesced per wave), tracked-frame
// texels float4 (gx, gy, I, 0) so that one bilinear tap is one 16-byte load.  Algorithmic bytes per evaluation at level l
// (SURVEY.md §8(d)): 20 N_l + [l==1] 5 N_l + 12 min(w_l h_l, 4 N_l).
#include "lsdhip_internal.hpp"

#define RES_BLOCK 256

// ---- wave64 sum via DPP (row_shr 1,2,3 / 4 / 8, row_bcast 15 / 31); result valid in lane 63 ----------------------
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, true));
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  float t = v + dpp_f<0x111, 0xf, 0xf>(v);  // row_shr:1
  t = t + dpp_f<0x112, 0xf, 0xf>(v);        // row_shr:2
  t = t + dpp_f<0x113, 0xf, 0xf>(v);        // row_shr:3
  t = t + dpp_f<0x114, 0xf, 0xe>(t);        // row_shr:4 bank_mask:0xe
  t = t + dpp_f<0x118, 0xf, 0xc>(t);        // row_shr:8 bank_mask:0xc
  t = t + dpp_f<0x142, 0xa, 0xf>(t);        // row_bcast:15 row_mask:0xa
  t = t + dpp_f<0x143, 0xc, 0xf>(t);        // row_bcast:31 row_mask:0xc
  return t;
}

// ---- per-point arithmetic -----------------------------------------------------------------------------------------
struct PointOut {
  bool in_image;
  bool good;
  float res, c1, c2, hw;      // residual, affine terms, Huber weight of the affine estimator
  float usage;                // min(1, z_ref / z_new)
  float werr;                 // wh * w_p * r^2 (K2)
  float w;                    // wh * w_p
  float J[6];
};

// (px,py,pz) = reference point, I_ref / var = its colour and inverse-depth variance.
__device__ __forceinline__ void eval_point(const ResidualArgs& a, float px, float py, float pz, float I_ref, float var, PointOut& o) {
  // Wxp = rotMat * p + transVec (Eigen coefficient product: ((r0*x + r1*y) + r2*z), then + t)
  float Wx = ((a.R[0] * px + a.R[1] * py) + a.R[2] * pz) + a.t[0];
  float Wy = ((a.R[3] * px + a.R[4] * py) + a.R[5] * pz) + a.t[1];
  float Wz = ((a.R[6] * px + a.R[7] * py) + a.R[8] * pz) + a.t[2];
  float u_new = (Wx / Wz) * a.fx + a.cx;
  float v_new = (Wy / Wz) * a.fy + a.cy;
  o.in_image = (u_new > 1 && v_new > 1 && u_new < a.w - 2 && v_new < a.h - 2);
  if (!o.in_image) return;

  // getInterpolatedElement43 (C/util/globalFuncs.h:63-77)
  int ix = (int)u_new;
  int iy = (int)v_new;
  float dx = u_new - ix;
  float dy = v_new - iy;
  float dxdy = dx * dy;
  const float4* bp = a.fr_grad + ix + iy * a.w;
  float4 t00 = bp[0], t10 = bp[1], t01 = bp[a.w], t11 = bp[1 + a.w];
  float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
  float rx = w11 * t11.x + w01 * t01.x + w10 * t10.x + w00 * t00.x;
  float ry = w11 * t11.y + w01 * t01.y + w10 * t10.y + w00 * t00.y;
  float rz = w11 * t11.z + w01 * t01.z + w10 * t10.z + w00 * t00.z;

  float c1 = a.aff_a * I_ref + a.aff_b;
  float c2 = rz;
  float residual = c1 - c2;
  o.res = residual; o.c1 = c1; o.c2 = c2;
  o.hw = fabsf(residual) < 5.0f ? 1 : 5.0f / fabsf(residual);
  o.good = residual * residual / (40.0f * 40.0f + 0.5f * 0.5f * (rx * rx + ry * ry)) < 1;
  float gx = a.fx * rx;   // buf_warped_dx
  float gy = a.fy * ry;   // buf_warped_dy
  float d = 1.0f / pz;    // buf_d
  float depthChange = pz / Wz;
  o.usage = depthChange < 1 ? depthChange : 1;

  // K2, SSE operation order with an IEEE reciprocal
  float pz2d = 1.0f / ((Wz * Wz) * d);
  float g0 = (Wz * a.t[0] - Wx * a.t[2]) * pz2d;
  float g1 = (Wz * a.t[1] - Wy * a.t[2]) * pz2d;
  float drpdd = g0 * gx + g1 * gy;
  float w_p = 1.0f / (a.cameraPixelNoise2 + drpdd * (drpdd * (a.var_weight * var)));
  float wr = residual * sqrtf(w_p);
  wr = fmaxf(wr, 0.0f - wr);
  float wh = (wr < a.huber_half) ? 1.0f : a.huber_half * (1.0f / wr);
  o.werr = wh * (wr * wr);
  o.w = wh * w_p;

  // K3, SSE operation order
  float z = 1.0f / Wz;
  o.J[0] = z * gx;
  o.J[1] = z * gy;
  float v1 = (Wx * gy) * z;
  float v2 = (Wy * gx) * z;
  o.J[5] = v1 - v2;
  float z2 = z * z;
  v1 = (Wx * gx) * z2;
  v2 = (Wy * gy) * z2;
  o.J[2] = 0.0f - (v1 + v2);
  o.J[3] = 0.0f - ((v2 * Wy) + (gy + v1 * Wy));
  o.J[4] = (gx + v1 * Wx) + v2 * Wx;
}

// fetch the reference point `i` (dense index into the keyframe level, or index into the explicit list)
__device__ __forceinline__ bool fetch_point(const ResidualArgs& a, int i, float& px, float& py, float& pz, float& I_ref,
                                            float& var, int& maskIdx) {
  if (a.npts >= 0) {
    if (i >= a.npts) return false;
    px = a.pts_pos[3 * i]; py = a.pts_pos[3 * i + 1]; pz = a.pts_pos[3 * i + 2];
    I_ref = a.pts_colvar[2 * i]; var = a.pts_colvar[2 * i + 1];
    maskIdx = -1;
    return true;
  }
  if (i >= a.w * a.h) return false;
  int x = i % a.w, y = i / a.w;
  if (x < 1 || x >= a.w - 1 || y < 1 || y >= a.h - 1) return false;
  var = a.kf_idepthVar[i];
  float id = a.kf_idepth[i];
  if (var <= 0 || id == 0) return false;
  float inv = 1.0f / id;
  px = inv * (a.fxi * x + a.cxi);
  py = inv * (a.fyi * y + a.cyi);
  pz = inv * 1.0f;
  I_ref = a.kf_image[i];
  maskIdx = i;
  return true;
}

// Residual kernel: one reference pixel per lane; 44 sums reduced wave -> workgroup -> partials[block].
__global__ __launch_bounds__(RES_BLOCK) void k_residual(ResidualArgs a) {
  const int tid = threadIdx.x;
  const int i = blockIdx.x * RES_BLOCK + tid;
  float acc[RS_NUM];
#pragma unroll
  for (int k = 0; k < RS_NUM; k++) acc[k] = 0.f;

  float px, py, pz, I_ref, var;
  int maskIdx;
  if (fetch_point(a, i, px, py, pz, I_ref, var, maskIdx)) {
    acc[RS_NREF] = 1.f;
    PointOut o;
    eval_point(a, px, py, pz, I_ref, var, o);
    if (!o.in_image) {
      if (a.wasGood && maskIdx >= 0) a.wasGood[maskIdx] = 0;
    } else {
      if (a.wasGood && maskIdx >= 0) a.wasGood[maskIdx] = o.good ? 1 : 0;
      acc[RS_M] = 1.f;
      acc[RS_SXX] = o.c1 * o.c1 * o.hw;
      acc[RS_SYY] = o.c2 * o.c2 * o.hw;
      acc[RS_SX] = o.c1 * o.hw;
      acc[RS_SY] = o.c2 * o.hw;
      acc[RS_SW] = o.hw;
      if (o.good) { acc[RS_GOOD] = 1.f; acc[RS_SUMRES2] = o.res * o.res; acc[RS_SUMSIGNED] = o.res; }
      else acc[RS_BAD] = 1.f;
      acc[RS_USAGE] = o.usage;
      acc[RS_WERR] = o.werr;
      int k = RS_A0;
#pragma unroll
      for (int r = 0; r < 6; r++) {
        float Jw = o.J[r] * o.w;
#pragma unroll
        for (int c = r; c < 6; c++) acc[k++] = Jw * o.J[c];
      }
      float resw = o.res * o.w;
#pragma unroll
      for (int r = 0; r < 6; r++) acc[RS_B0 + r] = resw * o.J[r];
      acc[RS_ERR] = resw * o.res;
    }
  }

  __shared__ float s_part[RES_BLOCK / 64][RS_NUM];
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int k = 0; k < RS_END; k++) {
    float s = wave_sum_to_lane63(acc[k]);
    if (lane == 63) s_part[wave][k] = s;
  }
  __syncthreads();
  if (tid < RS_NUM) {
    float s = 0.f;
    if (tid < RS_END) {
      s = s_part[0][tid];
#pragma unroll
      for (int wv = 1; wv < RES_BLOCK / 64; wv++) s += s_part[wv][tid];
    }
    a.partials[(size_t)blockIdx.x * RS_NUM + tid] = s;
  }
}

// Finalize: fixed-order sum of the per-workgroup partials, then the SSE tail drop: the last (M % 4) in-image points
// in the reference's point order (x outer, y inner over the keyframe level; or list order) are removed from the K2/K3 sums.
__global__ __launch_bounds__(256) void k_residual_finalize(ResidualArgs a) {
  __shared__ float s_sum[4][64];
  __shared__ float s_tot[RS_NUM];
  __shared__ int s_flag[256];
  __shared__ int s_chosen[4];
  __shared__ int s_nchosen;
  __shared__ float s_sub[3][32];
  const int tid = threadIdx.x;
  const int col = tid & 63, slice = tid >> 6;
  float s = 0.f;
  if (col < RS_NUM)
    for (int b = slice; b < a.nblocks; b += 4) s += a.partials[(size_t)b * RS_NUM + col];
  s_sum[slice][col] = s;
  __syncthreads();
  if (tid < RS_NUM) s_tot[tid] = ((s_sum[0][tid] + s_sum[1][tid]) + s_sum[2][tid]) + s_sum[3][tid];
  if (tid == 0) s_nchosen = 0;
  __syncthreads();

  const int M = (int)s_tot[RS_M];
  const int need = M & 3;
  if (need > 0) {
    // reverse walk in reference order, 256 candidates per round
    const int total = (a.npts >= 0) ? a.npts : (a.w - 2) * (a.h - 2);
    for (int base = 0; base < total; base += 256) {
      int r = base + tid;  // r-th point from the end
      int flag = 0, pidx = -1;
      if (r < total) {
        if (a.npts >= 0) pidx = a.npts - 1 - r;
        else {
          int q = total - 1 - r;           // forward position in x-outer / y-inner order over the interior
          int x = 1 + q / (a.h - 2), y = 1 + q % (a.h - 2);
          pidx = x + y * a.w;
        }
        float px, py, pz, I_ref, var;
        int maskIdx;
        if (fetch_point(a, pidx, px, py, pz, I_ref, var, maskIdx)) {
          PointOut o;
          eval_point(a, px, py, pz, I_ref, var, o);
          flag = o.in_image ? 1 : 0;
        }
      }
      s_flag[tid] = flag ? pidx + 1 : 0;
      __syncthreads();
      if (tid == 0) {
        int n = s_nchosen;
        for (int k = 0; k < 256 && n < need; k++)
          if (s_flag[k]) s_chosen[n++] = s_flag[k] - 1;
        s_nchosen = n;
      }
      __syncthreads();
      if (s_nchosen >= need) break;
    }
    if (tid < s_nchosen) {
      float px, py, pz, I_ref, var;
      int maskIdx;
      fetch_point(a, s_chosen[tid], px, py, pz, I_ref, var, maskIdx);
      PointOut o;
      eval_point(a, px, py, pz, I_ref, var, o);
      float* sub = s_sub[tid];
      sub[0] = o.werr;
      int k = 1;
      for (int r = 0; r < 6; r++) {
        float Jw = o.J[r] * o.w;
        for (int c = r; c < 6; c++) sub[k++] = Jw * o.J[c];
      }
      float resw = o.res * o.w;
      for (int r = 0; r < 6; r++) sub[k++] = resw * o.J[r];
      sub[k++] = resw * o.res;
    }
    __syncthreads();
    if (tid < 29) {
      // element 0 -> RS_WERR, 1..21 -> RS_A0.., 22..27 -> RS_B0.., 28 -> RS_ERR
      int dst = (tid == 0) ? RS_WERR : (tid <= 21 ? RS_A0 + tid - 1 : (tid <= 27 ? RS_B0 + tid - 22 : RS_ERR));
      float v = s_tot[dst];
      for (int k = 0; k < s_nchosen; k++) v -= s_sub[k][tid];
      s_tot[dst] = v;
    }
    __syncthreads();
  }
  if (tid < RS_NUM) a.out_record[tid] = s_tot[tid];
}

// checkPermaRefOverlap (SE3Tracker.cpp:121-157): usage only, explicit point list
__global__ __launch_bounds__(256) void k_overlap(const float* __restrict__ pos, int n, ResidualArgs a, float* __restrict__ out) {
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
  t->max_blocks = (c->w * c->h + RES_BLOCK - 1) / RES_BLOCK;
  HIPCHK(hipMalloc((void**)&t->d_partials, (size_t)t->max_blocks * RS_NUM * sizeof(float)));
  HIPCHK(hipHostMalloc((void**)&t->h_record, RS_NUM * sizeof(float), hipHostMallocMapped));
  HIPCHK(hipHostGetDevicePointer((void**)&t->d_record, t->h_record, 0));
  *out = t;
  return LSDHIP_OK;
}
extern "C" void lsdhip_tracker_destroy(lsdhip_tracker* t) {
  if (!t) return;
  (void)hipSetDevice(t->ctx->device);
  (void)hipStreamSynchronize(t->ctx->stream);
  (void)hipFree(t->d_partials);
  (void)hipHostFree(t->h_record);
  if (t->d_pts) (void)hipFree(t->d_pts);
  delete t;
}
extern "C" int lsdhip_tracker_set_max_its(lsdhip_tracker* t, const int its[LSD_LEVELS]) {
  if (!t || !its) return LSDHIP_E_ARG;
  for (int l = 0; l < LSD_LEVELS; l++) t->maxItsPerLvl[l] = its[l];
  return LSDHIP_OK;
}

struct EvalOut {       // what one evaluation leaves behind, in the reference's terms
  int warped_size;
  float retval;        // calcResidualAndBuffers return value
  float weightedError; // calcWeightsAndResidualSSE return value
  float A[36], b[6], lsError;
  double num_constraints;
};

// launch residual + finalize for one pose and turn the raw sums into the tracker's members
static int evaluate_pose(lsdhip_tracker* t, lsdhip_frame* kf, lsdhip_frame* frame, const float* pts_pos, const float* pts_colvar,
                         int npts, const lsdm::SE3fH& T, int level, bool writeMask, EvalOut* eo) {
  lsdhip_ctx* c = t->ctx;
  ResidualArgs a;
  memset(&a, 0, sizeof(a));
  const LevelIntr& in = c->intr[level];
  a.w = c->wl[level]; a.h = c->hl[level];
  a.fx = in.fx; a.fy = in.fy; a.cx = in.cx; a.cy = in.cy; a.fxi = in.fxi; a.fyi = in.fyi; a.cxi = in.cxi; a.cyi = in.cyi;
  a.fr_grad = frame->d_grad[level];
  if (npts >= 0) {
    a.pts_pos = pts_pos; a.pts_colvar = pts_colvar; a.npts = npts;
  } else {
    a.kf_idepth = kf->d_idepth[level]; a.kf_idepthVar = kf->d_idepthVar[level]; a.kf_image = kf->d_image[level];
    a.npts = -1;
  }
  a.wasGood = nullptr;
  if (writeMask) {
    int rc = lsd_frame_ensure_wasgood(frame);
    if (rc) return rc;
    a.wasGood = frame->d_wasGood;
  }
  lsdm::quatf_to_rot(T.q, a.R);
  a.t[0] = T.t[0]; a.t[1] = T.t[1]; a.t[2] = T.t[2];
  a.aff_a = t->affineEstimation_a; a.aff_b = t->affineEstimation_b;
  a.cameraPixelNoise2 = c->params.cameraPixelNoise2;
  a.var_weight = t->var_weight;
  a.huber_half = t->huber_d / 2;
  a.partials = t->d_partials;
  a.out_record = t->d_record;
  int work = npts >= 0 ? npts : a.w * a.h;
  a.nblocks = (work + RES_BLOCK - 1) / RES_BLOCK;
  if (a.nblocks < 1) a.nblocks = 1;
  if (a.nblocks > t->max_blocks) { lsd_set_error("residual grid exceeds scratch"); return LSDHIP_E_CAPACITY; }

  if (c->prof_on) HIPCHK(hipEventRecord(c->ev_a, c->stream));
  hipLaunchKernelGGL(k_residual, dim3(a.nblocks), dim3(RES_BLOCK), 0, c->stream, a);
  if (c->prof_on) HIPCHK(hipEventRecord(c->ev_b, c->stream));
  hipLaunchKernelGGL(k_residual_finalize, dim3(1), dim3(256), 0, c->stream, a);
  HIPCHK(hipStreamSynchronize(c->stream));
  const float* r = t->h_record;
  if (c->prof_on) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev_a, c->ev_b));
    c->prof_ms += ms;
    c->prof_launches++;
    double N = r[RS_NREF];
    double texels = 4.0 * N < (double)a.w * a.h ? 4.0 * N : (double)a.w * a.h;
    c->prof_bytes += 20.0 * N + (writeMask ? 5.0 * N : 0.0) + 12.0 * texels;
  }
  t->numEvaluations++;

  // calcResidualAndBuffers epilogue (SE3Tracker.cpp:1016-1028)
  int M = (int)r[RS_M];
  float refNum = r[RS_NREF];
  float goodCount = r[RS_GOOD], badCount = r[RS_BAD];
  t->pointUsage = r[RS_USAGE] / refNum;
  t->lastGoodCount = goodCount;
  t->lastBadCount = badCount;
  t->lastMeanRes = r[RS_SUMSIGNED] / goodCount;
  float sxx = r[RS_SXX], syy = r[RS_SYY], sx = r[RS_SX], sy = r[RS_SY], sw = r[RS_SW];
  t->affineEstimation_a_lastIt = sqrtf((syy - sy * sy / sw) / (sxx - sx * sx / sw));
  t->affineEstimation_b_lastIt = (sy - t->affineEstimation_a_lastIt * sx) / sw;
  eo->warped_size = M;
  eo->retval = r[RS_SUMRES2] / goodCount;
  // calcWeightsAndResidualSSE epilogue (:572-574)
  eo->weightedError = r[RS_WERR] / ((M >> 2) << 2);
  // LGS6::finish with the SSE constraint count (LGSX.h:319-325, :385)
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

static void fill_result(lsdhip_tracker* t, const lsdm::SE3dH& T, lsdhip_track_result* out) {
  lsdm::se3d_to7(T, out->frameToReference);
  out->pointUsage = t->pointUsage; out->lastGoodCount = t->lastGoodCount; out->lastBadCount = t->lastBadCount;
  out->lastMeanRes = t->lastMeanRes; out->lastResidual = t->lastResidual;
  out->affineEstimation_a = t->affineEstimation_a; out->affineEstimation_b = t->affineEstimation_b;
  out->diverged = t->diverged; out->trackingWasGood = t->trackingWasGood;
  out->numEvaluations = t->numEvaluations; out->numWarpUpdates = t->numWarpUpdates;
}

static const float MIN_GOODPERGOODBAD_PIXEL = 0.5f;
static const float MIN_GOODPERALL_PIXEL = 0.04f;
static const float MIN_GOODPERALL_PIXEL_ABSMIN = 0.01f;

// The LM loop shared by trackFrame (levels 4..1) and trackFrameOnPermaref (level 4 only).
// Returns LSDHIP_DIVERGED on divergence.
static int lm_level(lsdhip_tracker* t, lsdhip_frame* kf, lsdhip_frame* frame, const float* pts_pos, const float* pts_colvar, int npts,
                    int lvl, bool writeMask, float lambdaInitial, float stepSizeMin, float convergenceEps, int maxIts,
                    lsdm::SE3fH& referenceToFrame, float* lastResidualOut, bool trackFrameSemantics) {
  lsdhip_ctx* c = t->ctx;
  EvalOut ev;
  int rc = evaluate_pose(t, kf, frame, pts_pos, pts_colvar, npts, referenceToFrame, lvl, writeMask, &ev);
  if (rc) return rc;
  if (ev.warped_size < MIN_GOODPERALL_PIXEL_ABSMIN * (c->w >> lvl) * (c->h >> lvl)) return LSDHIP_DIVERGED;
  if (c->params.useAffineLightningEstimation) {
    t->affineEstimation_a = t->affineEstimation_a_lastIt;
    t->affineEstimation_b = t->affineEstimation_b_lastIt;
  }
  // NOTE: in the reference the first weights/LGS of a level are computed on residuals that used the *previous*
  // affine parameters (buffers are not recomputed after the assignment above); the fused record `ev` has exactly
  // those semantics because K2/K3 ran inside the same evaluation.
  float lastErr = ev.weightedError;
  float LM_lambda = lambdaInitial;
  EvalOut cur = ev;  // buffers of the last *accepted* pose feed calculateWarpUpdate

  for (int iteration = 0; iteration < maxIts; iteration++) {
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
      rc = evaluate_pose(t, kf, frame, pts_pos, pts_colvar, npts, new_referenceToFrame, lvl, writeMask, &nev);
      if (rc) return rc;
      if (nev.warped_size < MIN_GOODPERALL_PIXEL_ABSMIN * (c->w >> lvl) * (c->h >> lvl)) return LSDHIP_DIVERGED;
      float error = nev.weightedError;
      if (error < lastErr) {
        referenceToFrame = new_referenceToFrame;
        cur = nev;
        if (c->params.useAffineLightningEstimation) {
          t->affineEstimation_a = t->affineEstimation_a_lastIt;
          t->affineEstimation_b = t->affineEstimation_b_lastIt;
        }
        if (error / lastErr > convergenceEps) iteration = maxIts;
        lastErr = error;
        if (trackFrameSemantics) *lastResidualOut = error;
        if (LM_lambda <= 0.2) LM_lambda = 0;
        else LM_lambda *= t->lambdaSuccessFac;
        break;
      } else {
        float incdot = (inc[0] * inc[0] + (inc[1] * inc[1] + inc[2] * inc[2])) + (inc[3] * inc[3] + (inc[4] * inc[4] + inc[5] * inc[5]));
        if (!(incdot > stepSizeMin)) { iteration = maxIts; break; }
        if (LM_lambda == 0) LM_lambda = 0.2;
        else LM_lambda *= std::pow(t->lambdaFailFac, incTry);
      }
    }
  }
  if (!trackFrameSemantics) *lastResidualOut = lastErr;
  return LSDHIP_OK;
}

extern "C" int lsdhip_tracker_track(lsdhip_tracker* t, lsdhip_frame* kf, lsdhip_frame* frame, const double init[7],
                                    lsdhip_track_result* out) {
  if (!t || !kf || !frame || !init || !out) return LSDHIP_E_ARG;
  if (!kf->hasIDepth) { lsd_set_error("lsdhip_tracker_track: keyframe %d has no depth", kf->id); return LSDHIP_E_STATE; }
  lsdhip_ctx* c = t->ctx;
  HIPCHK(hipSetDevice(c->device));
  t->diverged = false;
  t->trackingWasGood = true;
  t->affineEstimation_a = 1; t->affineEstimation_b = 0;
  t->numEvaluations = 0; t->numWarpUpdates = 0;
  lsdm::SE3fH referenceToFrame = lsdm::se3f_from_d(lsdm::se3d_inverse(lsdm::se3d_from7(init)));
  float last_residual = 0;
  for (int lvl = LSD_TRACK_MAX_LEVEL - 1; lvl >= LSD_TRACK_MIN_LEVEL; lvl--) {
    int rc = lm_level(t, kf, frame, nullptr, nullptr, -1, lvl, lvl == LSD_TRACK_MIN_LEVEL, t->lambdaInitial[lvl], t->stepSizeMin[lvl],
                      t->convergenceEps[lvl], t->maxItsPerLvl[lvl], referenceToFrame, &last_residual, true);
    if (rc == LSDHIP_DIVERGED) {
      t->diverged = true;
      t->trackingWasGood = false;
      lsdm::SE3dH I; I.q = {1, 0, 0, 0}; I.t[0] = I.t[1] = I.t[2] = 0;
      fill_result(t, I, out);
      return LSDHIP_DIVERGED;
    }
    if (rc) return rc;
  }
  t->lastResidual = last_residual;
  t->trackingWasGood = !t->diverged && t->lastGoodCount / (c->wl[LSD_TRACK_MIN_LEVEL] * c->hl[LSD_TRACK_MIN_LEVEL]) > MIN_GOODPERALL_PIXEL &&
                       t->lastGoodCount / (t->lastGoodCount + t->lastBadCount) > MIN_GOODPERGOODBAD_PIXEL;
  if (t->trackingWasGood) kf->numFramesTrackedOnThis++;
  frame->initialTrackedResidual = t->lastResidual / t->pointUsage;
  lsdm::SE3dH f2r = lsdm::se3d_from_f(lsdm::se3f_inverse(referenceToFrame));
  frame->thisToParent_raw.q = f2r.q;
  frame->thisToParent_raw.t[0] = f2r.t[0]; frame->thisToParent_raw.t[1] = f2r.t[1]; frame->thisToParent_raw.t[2] = f2r.t[2];
  frame->thisToParent_raw.s = 1;
  frame->trackingParent = kf;
  frame->trackingParentID = kf->id;
  fill_result(t, f2r, out);
  return LSDHIP_OK;
}

extern "C" int lsdhip_tracker_evaluate(lsdhip_tracker* t, lsdhip_frame* kf, lsdhip_frame* frame, const float T7[7], int level,
                                       float aff_a, float aff_b, lsdhip_residual_record* out) {
  if (!t || !kf || !frame || !T7 || !out || level < 0 || level >= LSD_LEVELS) return LSDHIP_E_ARG;
  if (!kf->hasIDepth) { lsd_set_error("lsdhip_tracker_evaluate: keyframe has no depth"); return LSDHIP_E_STATE; }
  HIPCHK(hipSetDevice(t->ctx->device));
  lsdm::SE3fH T;
  T.q = {T7[0], T7[1], T7[2], T7[3]};
  T.t[0] = T7[4]; T.t[1] = T7[5]; T.t[2] = T7[6];
  t->affineEstimation_a = aff_a; t->affineEstimation_b = aff_b;
  EvalOut ev;
  int rc = evaluate_pose(t, kf, frame, nullptr, nullptr, -1, T, level, level == LSD_TRACK_MIN_LEVEL, &ev);
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
  HIPCHK(hipSetDevice(c->device));
  if ((n + RES_BLOCK - 1) / RES_BLOCK > t->max_blocks) return LSDHIP_E_CAPACITY;
  int rc = upload_points(t, pos, colvar, n);
  if (rc) return rc;
  lsdm::SE3fH referenceToFrame = lsdm::se3f_from_d(lsdm::se3d_from7(refToFrame));
  t->affineEstimation_a = 1; t->affineEstimation_b = 0;
  t->diverged = false; t->trackingWasGood = true;
  t->numEvaluations = 0; t->numWarpUpdates = 0;
  float lastErr = 0;
  const int L = LSD_QUICK_KF_CHECK_LVL;
  rc = lm_level(t, nullptr, frame, t->d_pts, t->d_pts + (size_t)t->pts_capacity * 3, n, L, false, t->lambdaInitialTestTrack,
                t->stepSizeMinTestTrack, t->convergenceEpsTestTrack, (int)t->maxItsTestTrack, referenceToFrame, &lastErr, false);
  if (rc == LSDHIP_DIVERGED) {
    t->diverged = true; t->trackingWasGood = false;
    lsdm::SE3dH I; I.q = {1, 0, 0, 0}; I.t[0] = I.t[1] = I.t[2] = 0;
    fill_result(t, I, out);
    return LSDHIP_DIVERGED;
  }
  if (rc) return rc;
  t->lastResidual = lastErr;
  t->trackingWasGood = !t->diverged && t->lastGoodCount / (c->wl[L] * c->hl[L]) > MIN_GOODPERALL_PIXEL &&
                       t->lastGoodCount / (t->lastGoodCount + t->lastBadCount) > MIN_GOODPERGOODBAD_PIXEL;
  fill_result(t, lsdm::se3d_from_f(referenceToFrame), out);
  return LSDHIP_OK;
}

extern "C" int lsdhip_tracker_check_overlap(lsdhip_tracker* t, const float* pos, int n, const double refToFrame[7], float* usage_out) {
  if (!t || !pos || n <= 0 || !refToFrame || !usage_out) return LSDHIP_E_ARG;
  lsdhip_ctx* c = t->ctx;
  HIPCHK(hipSetDevice(c->device));
  int rc = upload_points(t, pos, nullptr, n);
  if (rc) return rc;
  lsdm::SE3fH T = lsdm::se3f_from_d(lsdm::se3d_from7(refToFrame));
  const int L = LSD_QUICK_KF_CHECK_LVL;
  ResidualArgs a;
  memset(&a, 0, sizeof(a));
  const LevelIntr& in = c->intr[L];
  a.w = c->wl[L]; a.h = c->hl[L];
  a.fx = in.fx; a.fy = in.fy; a.cx = in.cx; a.cy = in.cy;
  lsdm::quatf_to_rot(T.q, a.R);
  a.t[0] = T.t[0]; a.t[1] = T.t[1]; a.t[2] = T.t[2];
  hipLaunchKernelGGL(k_overlap, dim3(1), dim3(256), 0, c->stream, t->d_pts, n, a, t->d_record);
  HIPCHK(hipStreamSynchronize(c->stream));
  t->pointUsage = t->h_record[0] / (float)n;
  *usage_out = t->pointUsage;
  return LSDHIP_OK;
}
