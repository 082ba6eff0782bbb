// ORACLE — TEST INFRASTRUCTURE ONLY (see lsd_oracle.hpp).  PARITY UNPINNED.
// Restates C/Tracking/SE3Tracker.cpp (scalar and ENABLE_SSE member functions) and C/Tracking/LGSX.h:184-402.
#include <xmmintrin.h>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "lsd_oracle.hpp"
#include <cstdio>
#include <cstdlib>

namespace orc {

#define SSEE(val, idx) (*(((float*)&val) + idx))

// settings.h:165-170
static const float MAX_DIFF_CONSTANT = 40.0f * 40.0f;
static const float MAX_DIFF_GRAD_MULT = 0.5f * 0.5f;
static const float MIN_GOODPERGOODBAD_PIXEL = 0.5f;
static const float MIN_GOODPERALL_PIXEL = 0.04f;
static const float MIN_GOODPERALL_PIXEL_ABSMIN = 0.01f;

// ---- LGSX.h:184-402 ------------------------------------------------------------------------
void LGS6::initialize() {
  memset(A, 0, sizeof(A));
  memset(b, 0, sizeof(b));
  memset(SSEData, 0, sizeof(SSEData));
  error = 0;
  num_constraints = 0;
}
// LGSX.h:390-396
void LGS6::update(const float J[6], float res, float weight) {
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) A[i * 6 + j] += J[i] * J[j] * weight;
  float rw = res * weight;
  for (int i = 0; i < 6; i++) b[i] -= J[i] * rw;
  error += res * res * weight;
  num_constraints += 1;
}
// LGSX.h:205-316: lane sums in the order ((l0 + l1) + l2) + l3, mirrored into both triangles.
void LGS6::finishNoDivide() {
  int k = 0;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++, k++) {
      const float* a = SSEData + 4 * k;
      float s = a[0] + a[1] + a[2] + a[3];
      A[i * 6 + j] += s;
      A[j * 6 + i] = A[i * 6 + j];
    }
  for (int i = 0; i < 6; i++) {
    const float* a = SSEData + 4 * (21 + i);
    b[i] -= a[0] + a[1] + a[2] + a[3];
  }
  const float* a = SSEData + 4 * 27;
  error += a[0] + a[1] + a[2] + a[3];
}
// LGSX.h:319-325
void LGS6::finish() {
  finishNoDivide();
  float n = (float)num_constraints;
  for (int i = 0; i < 36; i++) A[i] /= n;
  for (int i = 0; i < 6; i++) b[i] /= n;
  error /= n;
}
// LGSX.h:328-386 (num_constraints += 6 per group of 4 points is the reference's behaviour)
static inline void lgs6_updateSSE(LGS6& ls, const __m128 J[6], const __m128& res, const __m128& weight) {
  float* S = ls.SSEData;
  int k = 0;
  for (int i = 0; i < 6; i++) {
    __m128 Jiw = _mm_mul_ps(J[i], weight);
    for (int j = i; j < 6; j++, k++)
      _mm_store_ps(S + 4 * k, _mm_add_ps(_mm_load_ps(S + 4 * k), _mm_mul_ps(Jiw, J[j])));
  }
  __m128 resw = _mm_mul_ps(res, weight);
  for (int i = 0; i < 6; i++)
    _mm_store_ps(S + 4 * (21 + i), _mm_add_ps(_mm_load_ps(S + 4 * (21 + i)), _mm_mul_ps(resw, J[i])));
  _mm_store_ps(S + 4 * 27, _mm_add_ps(_mm_load_ps(S + 4 * 27), _mm_mul_ps(resw, res)));
  ls.num_constraints += 6;
}

// ---- SE3Tracker ----------------------------------------------------------------------------
static float* alloc16(size_t n) {
  void* p = nullptr;
  if (posix_memalign(&p, 16, n * sizeof(float)) != 0) abort();
  memset(p, 0, n * sizeof(float));
  return (float*)p;
}

SE3Tracker::SE3Tracker(int w, int h, const float[4], const Params& p) : width(w), height(h), params(p) {
  size_t n = (size_t)w * h;
  buf_warped_residual = alloc16(n); buf_warped_dx = alloc16(n); buf_warped_dy = alloc16(n);
  buf_warped_x = alloc16(n); buf_warped_y = alloc16(n); buf_warped_z = alloc16(n);
  buf_d = alloc16(n); buf_idepthVar = alloc16(n); buf_weight_p = alloc16(n);
}
SE3Tracker::~SE3Tracker() {
  free(buf_warped_residual); free(buf_warped_dx); free(buf_warped_dy); free(buf_warped_x); free(buf_warped_y);
  free(buf_warped_z); free(buf_d); free(buf_idepthVar); free(buf_weight_p);
}

// C/util/globalFuncs.h:63-77 (getInterpolatedElement43), Eigen expression evaluated per component,
// left to right: ((dxdy*a + (dy-dxdy)*b) + (dx-dxdy)*c) + (1-dx-dy+dxdy)*d
static inline void interp43(const float* mat4, float x, float y, int width, float out[3]) {
  int ix = (int)x;
  int iy = (int)y;
  float dx = x - ix;
  float dy = y - iy;
  float dxdy = dx * dy;
  const float* bp = mat4 + 4 * (ix + iy * width);
  float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
  for (int c = 0; c < 3; c++)
    out[c] = w11 * bp[4 * (1 + width) + c] + w01 * bp[4 * width + c] + w10 * bp[4 + c] + w00 * bp[c];
}

// SE3Tracker.cpp:885-1029
float SE3Tracker::calcResidualAndBuffers(const float* refPoint, const float* refColVar, const int* idxBuf, int refNum,
                                         Frame* frame, const SE3f& referenceToFrame, int level) {
  int w = frame->width(level);
  int h = frame->height(level);
  float fx_l = frame->K[level](0, 0), fy_l = frame->K[level](1, 1);
  float cx_l = frame->K[level](0, 2), cy_l = frame->K[level](1, 2);
  M3f rotMat = referenceToFrame.rotationMatrix();
  V3f transVec = referenceToFrame.t;
  const float* refPoint_max = refPoint + 3 * refNum;
  const float* frame_gradients = frame->gradients(level);
  int idx = 0;
  float sumResUnweighted = 0;
  uint8_t* isGoodOutBuffer = idxBuf != 0 ? frame->refPixelWasGood() : 0;
  int goodCount = 0, badCount = 0;
  float sumSignedRes = 0;
  float sxx = 0, syy = 0, sx = 0, sy = 0, sw = 0;
  float usageCount = 0;

  for (; refPoint < refPoint_max; refPoint += 3, refColVar += 2, idxBuf++) {
    V3f p = mk3<float>(refPoint[0], refPoint[1], refPoint[2]);
    V3f Rp = matvec(rotMat, p);
    V3f Wxp = mk3<float>(Rp[0] + transVec[0], Rp[1] + transVec[1], Rp[2] + transVec[2]);
    float u_new = (Wxp[0] / Wxp[2]) * fx_l + cx_l;
    float v_new = (Wxp[1] / Wxp[2]) * fy_l + cy_l;
    if (!(u_new > 1 && v_new > 1 && u_new < w - 2 && v_new < h - 2)) {
      if (isGoodOutBuffer != 0) isGoodOutBuffer[*idxBuf] = false;
      continue;
    }
    float resInterp[3];
    interp43(frame_gradients, u_new, v_new, w, resInterp);
    float c1 = affineEstimation_a * refColVar[0] + affineEstimation_b;
    float c2 = resInterp[2];
    float residual = c1 - c2;
    float weight = fabsf(residual) < 5.0f ? 1 : 5.0f / fabsf(residual);
    sxx += c1 * c1 * weight;
    syy += c2 * c2 * weight;
    sx += c1 * weight;
    sy += c2 * weight;
    sw += weight;
    bool isGood = residual * residual /
                      (MAX_DIFF_CONSTANT + MAX_DIFF_GRAD_MULT * (resInterp[0] * resInterp[0] + resInterp[1] * resInterp[1])) < 1;
    if (isGoodOutBuffer != 0) isGoodOutBuffer[*idxBuf] = isGood;
    buf_warped_x[idx] = Wxp[0];
    buf_warped_y[idx] = Wxp[1];
    buf_warped_z[idx] = Wxp[2];
    buf_warped_dx[idx] = fx_l * resInterp[0];
    buf_warped_dy[idx] = fy_l * resInterp[1];
    buf_warped_residual[idx] = residual;
    buf_d[idx] = 1.0f / refPoint[2];
    buf_idepthVar[idx] = refColVar[1];
    idx++;
    if (isGood) {
      sumResUnweighted += residual * residual;
      sumSignedRes += residual;
      goodCount++;
    } else
      badCount++;
    float depthChange = refPoint[2] / Wxp[2];
    usageCount += depthChange < 1 ? depthChange : 1;
  }
  buf_warped_size = idx;
  pointUsage = usageCount / (float)refNum;
  lastGoodCount = goodCount;
  lastBadCount = badCount;
  lastMeanRes = sumSignedRes / goodCount;
  affineEstimation_a_lastIt = sqrtf((syy - sy * sy / sw) / (sxx - sx * sx / sw));
  affineEstimation_b_lastIt = (sy - affineEstimation_a_lastIt * sx) / sw;
  numEvaluations++;
  if (level >= 0 && level < ORC_PYRAMID_LEVELS) { levelEvaluations[level]++; levelPoints[level] += refNum; levelWarped[level] += idx; }
  return sumResUnweighted / goodCount;
}

// SE3Tracker.cpp:749-790
float SE3Tracker::calcWeightsAndResidual(const SE3f& referenceToFrame) {
  float tx = referenceToFrame.t[0], ty = referenceToFrame.t[1], tz = referenceToFrame.t[2];
  float sumRes = 0;
  for (int i = 0; i < buf_warped_size; i++) {
    float px = buf_warped_x[i], py = buf_warped_y[i], pz = buf_warped_z[i];
    float d = buf_d[i];
    float rp = buf_warped_residual[i];
    float gx = buf_warped_dx[i], gy = buf_warped_dy[i];
    float s = settings.var_weight * buf_idepthVar[i];
    float g0 = (tx * pz - tz * px) / (pz * pz * d);
    float g1 = (ty * pz - tz * py) / (pz * pz * d);
    float drpdd = gx * g0 + gy * g1;
    float w_p = 1.0f / ((params.cameraPixelNoise2) + s * drpdd * drpdd);
    float weighted_rp = fabs(rp * sqrtf(w_p));
    float wh = fabs(weighted_rp < (settings.huber_d / 2) ? 1 : (settings.huber_d / 2) / weighted_rp);
    sumRes += wh * w_p * rp * rp;
    buf_weight_p[i] = wh * w_p;
  }
  return sumRes / buf_warped_size;
}

static inline __m128 rcp_sel(__m128 x, bool exact) {
  return exact ? _mm_div_ps(_mm_set1_ps(1.0f), x) : _mm_rcp_ps(x);
}

// SE3Tracker.cpp:492-575
float SE3Tracker::calcWeightsAndResidualSSE(const SE3f& referenceToFrame, bool exactRcp) {
  const __m128 txs = _mm_set1_ps(referenceToFrame.t[0]);
  const __m128 tys = _mm_set1_ps(referenceToFrame.t[1]);
  const __m128 tzs = _mm_set1_ps(referenceToFrame.t[2]);
  const __m128 zeros = _mm_set1_ps(0.0f);
  const __m128 ones = _mm_set1_ps(1.0f);
  const __m128 depthVarFacs = _mm_set1_ps((float)settings.var_weight);
  const __m128 sigma_i2s = _mm_set1_ps((float)params.cameraPixelNoise2);
  const __m128 huber_res_ponlys = _mm_set1_ps((float)(settings.huber_d / 2));
  __m128 sumResP = zeros;
  float sumRes = 0;
  for (int i = 0; i < buf_warped_size - 3; i += 4) {
    __m128 pzs = _mm_load_ps(buf_warped_z + i);
    __m128 pz2ds = rcp_sel(_mm_mul_ps(_mm_mul_ps(pzs, pzs), _mm_load_ps(buf_d + i)), exactRcp);
    __m128 g0s = _mm_sub_ps(_mm_mul_ps(pzs, txs), _mm_mul_ps(_mm_load_ps(buf_warped_x + i), tzs));
    g0s = _mm_mul_ps(g0s, pz2ds);
    __m128 g1s = _mm_sub_ps(_mm_mul_ps(pzs, tys), _mm_mul_ps(_mm_load_ps(buf_warped_y + i), tzs));
    g1s = _mm_mul_ps(g1s, pz2ds);
    __m128 drpdds = _mm_add_ps(_mm_mul_ps(g0s, _mm_load_ps(buf_warped_dx + i)), _mm_mul_ps(g1s, _mm_load_ps(buf_warped_dy + i)));
    __m128 w_ps = rcp_sel(
        _mm_add_ps(sigma_i2s, _mm_mul_ps(drpdds, _mm_mul_ps(drpdds, _mm_mul_ps(depthVarFacs, _mm_load_ps(buf_idepthVar + i))))),
        exactRcp);
    __m128 weighted_rps = _mm_mul_ps(_mm_load_ps(buf_warped_residual + i), _mm_sqrt_ps(w_ps));
    weighted_rps = _mm_max_ps(weighted_rps, _mm_sub_ps(zeros, weighted_rps));
    __m128 whs = _mm_cmplt_ps(weighted_rps, huber_res_ponlys);
    whs = _mm_or_ps(_mm_and_ps(whs, ones), _mm_andnot_ps(whs, _mm_mul_ps(huber_res_ponlys, rcp_sel(weighted_rps, exactRcp))));
    if (i + 3 < buf_warped_size) sumResP = _mm_add_ps(sumResP, _mm_mul_ps(whs, _mm_mul_ps(weighted_rps, weighted_rps)));
    _mm_store_ps(buf_weight_p + i, _mm_mul_ps(whs, w_ps));
  }
  sumRes = SSEE(sumResP, 0) + SSEE(sumResP, 1) + SSEE(sumResP, 2) + SSEE(sumResP, 3);
  return sumRes / ((buf_warped_size >> 2) << 2);
}

// SE3Tracker.cpp:1258-1299 (note the double literals 1.0 in v[3], v[4])
void SE3Tracker::calculateWarpUpdate(LGS6& ls) {
  ls.initialize();
  for (int i = 0; i < buf_warped_size; i++) {
    float px = buf_warped_x[i], py = buf_warped_y[i], pz = buf_warped_z[i];
    float r = buf_warped_residual[i];
    float gx = buf_warped_dx[i], gy = buf_warped_dy[i];
    float z = 1.0f / pz;
    float z_sqr = 1.0f / (pz * pz);
    float v[6];
    v[0] = z * gx + 0;
    v[1] = 0 + z * gy;
    v[2] = (-px * z_sqr) * gx + (-py * z_sqr) * gy;
    v[3] = (-px * py * z_sqr) * gx + (-(1.0 + py * py * z_sqr)) * gy;
    v[4] = (1.0 + px * px * z_sqr) * gx + (px * py * z_sqr) * gy;
    v[5] = (-py * z) * gx + (px * z) * gy;
    ls.update(v, r, buf_weight_p[i]);
  }
  ls.finish();
  numWarpUpdates++;
}

// SE3Tracker.cpp:1033-1130 (the scalar tail branch :1114-1122 is unreachable: the loop bound excludes it)
void SE3Tracker::calculateWarpUpdateSSE(LGS6& ls, bool exactRcp) {
  ls.initialize();
  for (int i = 0; i < buf_warped_size - 3; i += 4) {
    __m128 val1, val2, val3, val4;
    __m128 J[6];
    __m128 pz = _mm_load_ps(buf_warped_z + i);
    pz = rcp_sel(pz, exactRcp);
    __m128 gx = _mm_load_ps(buf_warped_dx + i);
    J[0] = _mm_mul_ps(pz, gx);
    __m128 gy = _mm_load_ps(buf_warped_dy + i);
    J[1] = _mm_mul_ps(pz, gy);
    __m128 px = _mm_load_ps(buf_warped_x + i);
    val1 = _mm_mul_ps(px, gy);
    val1 = _mm_mul_ps(val1, pz);
    __m128 py = _mm_load_ps(buf_warped_y + i);
    val2 = _mm_mul_ps(py, gx);
    val2 = _mm_mul_ps(val2, pz);
    J[5] = _mm_sub_ps(val1, val2);
    pz = _mm_mul_ps(pz, pz);
    val1 = _mm_mul_ps(px, gx);
    val1 = _mm_mul_ps(val1, pz);
    val2 = _mm_mul_ps(py, gy);
    val2 = _mm_mul_ps(val2, pz);
    val3 = _mm_add_ps(val1, val2);
    J[2] = _mm_sub_ps(_mm_setr_ps(0, 0, 0, 0), val3);
    val3 = _mm_mul_ps(val1, py);
    val4 = _mm_add_ps(gy, val3);
    val3 = _mm_mul_ps(val2, py);
    val4 = _mm_add_ps(val3, val4);
    J[3] = _mm_sub_ps(_mm_setr_ps(0, 0, 0, 0), val4);
    val3 = _mm_mul_ps(val1, px);
    val4 = _mm_add_ps(gx, val3);
    val3 = _mm_mul_ps(val2, px);
    J[4] = _mm_add_ps(val4, val3);
    if (i + 3 < buf_warped_size)
      lgs6_updateSSE(ls, J, _mm_load_ps(buf_warped_residual + i), _mm_load_ps(buf_weight_p + i));
  }
  ls.finish();
  numWarpUpdates++;
}

float SE3Tracker::callWeights(const SE3f& T) {
  if (mode == TRACKER_SCALAR) return calcWeightsAndResidual(T);
  return calcWeightsAndResidualSSE(T, mode == TRACKER_SSE_EXACT_RCP);
}
void SE3Tracker::callWarpUpdate(LGS6& ls) {
  if (mode == TRACKER_SCALAR) calculateWarpUpdate(ls);
  else calculateWarpUpdateSSE(ls, mode == TRACKER_SSE_EXACT_RCP);
}

void SE3Tracker::evaluate(TrackingReference* reference, Frame* frame, const SE3f& referenceToFrame, int level,
                          float aff_a, float aff_b, ResidualRecord* out) {
  reference->makePointCloud(level);
  affineEstimation_a = aff_a;
  affineEstimation_b = aff_b;
  float rv = calcResidualAndBuffers(reference->posData[level].data(), reference->colorAndVarData[level].data(),
                                    ORC_SE3TRACKING_MIN_LEVEL == level ? reference->pointPosInXYGrid[level].data() : 0,
                                    reference->numData[level], frame, referenceToFrame, level);
  out->warped_size = buf_warped_size;
  out->goodCount = lastGoodCount; out->badCount = lastBadCount; out->pointUsage = pointUsage;
  out->meanRes = lastMeanRes; out->retval = rv;
  out->affine_a_lastIt = affineEstimation_a_lastIt; out->affine_b_lastIt = affineEstimation_b_lastIt;
  out->weightedError = callWeights(referenceToFrame);
  LGS6 ls;
  callWarpUpdate(ls);
  memcpy(out->A, ls.A, sizeof(ls.A));
  memcpy(out->b, ls.b, sizeof(ls.b));
  out->lsError = ls.error;
  out->num_constraints = (double)ls.num_constraints;
}

// SE3Tracker.cpp:280-486
SE3d SE3Tracker::trackFrame(TrackingReference* reference, Frame* frame, const SE3d& frameToReference_initialEstimate) {
  diverged = false;
  trackingWasGood = true;
  affineEstimation_a = 1; affineEstimation_b = 0;
  SE3f referenceToFrame = frameToReference_initialEstimate.inverse().cast<float>();
  LGS6 ls;
  float last_residual = 0;

  for (int lvl = ORC_SE3TRACKING_MAX_LEVEL - 1; lvl >= ORC_SE3TRACKING_MIN_LEVEL; lvl--) {
    reference->makePointCloud(lvl);
    const int* idxb = ORC_SE3TRACKING_MIN_LEVEL == lvl ? reference->pointPosInXYGrid[lvl].data() : 0;
    calcResidualAndBuffers(reference->posData[lvl].data(), reference->colorAndVarData[lvl].data(), idxb,
                           reference->numData[lvl], frame, referenceToFrame, lvl);
    if (buf_warped_size < MIN_GOODPERALL_PIXEL_ABSMIN * (width >> lvl) * (height >> lvl)) {
      diverged = true;
      trackingWasGood = false;
      return SE3d();
    }
    if (params.useAffineLightningEstimation) {
      affineEstimation_a = affineEstimation_a_lastIt;
      affineEstimation_b = affineEstimation_b_lastIt;
    }
    float lastErr = callWeights(referenceToFrame);
    float LM_lambda = settings.lambdaInitial[lvl];

    for (int iteration = 0; iteration < settings.maxItsPerLvl[lvl]; iteration++) {
      callWarpUpdate(ls);
      int incTry = 0;
      while (true) {
        float b[6], A[36], inc[6];
        for (int i = 0; i < 6; i++) b[i] = -ls.b[i];
        memcpy(A, ls.A, sizeof(A));
        for (int i = 0; i < 6; i++) A[i * 6 + i] *= 1 + LM_lambda;
        ldlt6_solve(A, b, inc);
        incTry++;
        SE3f new_referenceToFrame = se3_exp<float>(inc) * referenceToFrame;
        calcResidualAndBuffers(reference->posData[lvl].data(), reference->colorAndVarData[lvl].data(), idxb,
                               reference->numData[lvl], frame, new_referenceToFrame, lvl);
        if (buf_warped_size < MIN_GOODPERALL_PIXEL_ABSMIN * (width >> lvl) * (height >> lvl)) {
          diverged = true;
          trackingWasGood = false;
          return SE3d();
        }
        float error = callWeights(new_referenceToFrame);
        static const bool lmTrace = getenv("ORC_LM_TRACE") != nullptr;   // developer diagnostics: accept / reject pattern per level
        if (lmTrace) fprintf(stderr, "LMT %d %c %g %d\n", lvl, error < lastErr ? 'A' : 'R', (double)LM_lambda, incTry);
        if (error < lastErr) {
          referenceToFrame = new_referenceToFrame;
          if (params.useAffineLightningEstimation) {
            affineEstimation_a = affineEstimation_a_lastIt;
            affineEstimation_b = affineEstimation_b_lastIt;
          }
          if (error / lastErr > settings.convergenceEps[lvl]) iteration = settings.maxItsPerLvl[lvl];
          last_residual = lastErr = error;
          if (LM_lambda <= 0.2) LM_lambda = 0;
          else LM_lambda *= settings.lambdaSuccessFac;
          break;
        } else {
          float incdot = 0;  // Vector6::dot — 6-element redux tree: (x0+(x1+x2)) + (x3+(x4+x5))
          incdot = (inc[0] * inc[0] + (inc[1] * inc[1] + inc[2] * inc[2])) + (inc[3] * inc[3] + (inc[4] * inc[4] + inc[5] * inc[5]));
          if (!(incdot > settings.stepSizeMin[lvl])) {
            iteration = settings.maxItsPerLvl[lvl];
            break;
          }
          if (LM_lambda == 0) LM_lambda = 0.2;
          else LM_lambda *= std::pow(settings.lambdaFailFac, incTry);
        }
      }
    }
  }

  lastResidual = last_residual;
  trackingWasGood = !diverged &&
                    lastGoodCount / (frame->width(ORC_SE3TRACKING_MIN_LEVEL) * frame->height(ORC_SE3TRACKING_MIN_LEVEL)) > MIN_GOODPERALL_PIXEL &&
                    lastGoodCount / (lastGoodCount + lastBadCount) > MIN_GOODPERGOODBAD_PIXEL;
  if (trackingWasGood) reference->keyframe->numFramesTrackedOnThis++;
  frame->initialTrackedResidual = lastResidual / pointUsage;
  SE3d f2r = referenceToFrame.inverse().cast<double>();
  frame->thisToParent_raw.q = f2r.q;
  qnormalize(frame->thisToParent_raw.q);   // sim3FromSE3 -> Sim3::setScale normalises the quaternion (rxso3.hpp:332-335)
  frame->thisToParent_raw.t = f2r.t;
  frame->thisToParent_raw.s = 1;
  frame->trackingParent = reference->keyframe;
  return f2r;
}

// SE3Tracker.cpp:162-272 (level QUICK_KF_CHECK_LVL only)
SE3d SE3Tracker::trackFrameOnPermaref(const float* permaRef_pos, const float* permaRef_colVar, int permaRefNumPts,
                                      Frame* frame, const SE3d& referenceToFrameOrg) {
  SE3f referenceToFrame = referenceToFrameOrg.cast<float>();
  affineEstimation_a = 1; affineEstimation_b = 0;
  LGS6 ls;
  diverged = false;
  trackingWasGood = true;
  const int L = ORC_QUICK_KF_CHECK_LVL;
  calcResidualAndBuffers(permaRef_pos, permaRef_colVar, 0, permaRefNumPts, frame, referenceToFrame, L);
  if (buf_warped_size < MIN_GOODPERALL_PIXEL_ABSMIN * (width >> L) * (height >> L)) {
    diverged = true; trackingWasGood = false; return SE3d();
  }
  if (params.useAffineLightningEstimation) { affineEstimation_a = affineEstimation_a_lastIt; affineEstimation_b = affineEstimation_b_lastIt; }
  float lastErr = callWeights(referenceToFrame);
  float LM_lambda = settings.lambdaInitialTestTrack;
  for (int iteration = 0; iteration < settings.maxItsTestTrack; iteration++) {
    callWarpUpdate(ls);
    int incTry = 0;
    while (true) {
      float b[6], A[36], inc[6];
      for (int i = 0; i < 6; i++) b[i] = -ls.b[i];
      memcpy(A, ls.A, sizeof(A));
      for (int i = 0; i < 6; i++) A[i * 6 + i] *= 1 + LM_lambda;
      ldlt6_solve(A, b, inc);
      incTry++;
      SE3f new_referenceToFrame = se3_exp<float>(inc) * referenceToFrame;
      calcResidualAndBuffers(permaRef_pos, permaRef_colVar, 0, permaRefNumPts, frame, new_referenceToFrame, L);
      if (buf_warped_size < MIN_GOODPERALL_PIXEL_ABSMIN * (width >> L) * (height >> L)) {
        diverged = true; trackingWasGood = false; return SE3d();
      }
      float error = callWeights(new_referenceToFrame);
      if (error < lastErr) {
        referenceToFrame = new_referenceToFrame;
        if (params.useAffineLightningEstimation) { affineEstimation_a = affineEstimation_a_lastIt; affineEstimation_b = affineEstimation_b_lastIt; }
        if (error / lastErr > settings.convergenceEpsTestTrack) iteration = settings.maxItsTestTrack;
        lastErr = error;
        if (LM_lambda <= 0.2) LM_lambda = 0;
        else LM_lambda *= settings.lambdaSuccessFac;
        break;
      } else {
        float incdot = (inc[0] * inc[0] + (inc[1] * inc[1] + inc[2] * inc[2])) + (inc[3] * inc[3] + (inc[4] * inc[4] + inc[5] * inc[5]));
        if (!(incdot > settings.stepSizeMinTestTrack)) { iteration = settings.maxItsTestTrack; break; }
        if (LM_lambda == 0) LM_lambda = 0.2;
        else LM_lambda *= std::pow(settings.lambdaFailFac, incTry);
      }
    }
  }
  lastResidual = lastErr;
  trackingWasGood = !diverged && lastGoodCount / (frame->width(L) * frame->height(L)) > MIN_GOODPERALL_PIXEL &&
                    lastGoodCount / (lastGoodCount + lastBadCount) > MIN_GOODPERGOODBAD_PIXEL;
  return referenceToFrame.cast<double>();
}

// SE3Tracker.cpp:121-157
float SE3Tracker::checkPermaRefOverlap(const float* permaRef_pos, int permaRefNumPts, Frame* reference,
                                       const SE3d& referenceToFrameOrg) {
  SE3f referenceToFrame = referenceToFrameOrg.cast<float>();
  const int L = ORC_QUICK_KF_CHECK_LVL;
  int w2 = reference->width(L) - 1;
  int h2 = reference->height(L) - 1;
  float fx_l = reference->K[L](0, 0), fy_l = reference->K[L](1, 1), cx_l = reference->K[L](0, 2), cy_l = reference->K[L](1, 2);
  M3f rotMat = referenceToFrame.rotationMatrix();
  V3f transVec = referenceToFrame.t;
  float usageCount = 0;
  for (int i = 0; i < permaRefNumPts; i++) {
    V3f p = mk3<float>(permaRef_pos[3 * i], permaRef_pos[3 * i + 1], permaRef_pos[3 * i + 2]);
    V3f Rp = matvec(rotMat, p);
    V3f Wxp = mk3<float>(Rp[0] + transVec[0], Rp[1] + transVec[1], Rp[2] + transVec[2]);
    float u_new = (Wxp[0] / Wxp[2]) * fx_l + cx_l;
    float v_new = (Wxp[1] / Wxp[2]) * fy_l + cy_l;
    if ((u_new > 0 && v_new > 0 && u_new < w2 && v_new < h2)) {
      float depthChange = p[2] / Wxp[2];
      usageCount += depthChange < 1 ? depthChange : 1;
    }
  }
  pointUsage = usageCount / (float)permaRefNumPts;
  return pointUsage;
}

}  // namespace orc
