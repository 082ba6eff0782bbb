// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
//
// CPU restatement of Sim3Tracker::trackFrameSim3, function by function:
//   calcSim3Buffers                     C/Tracking/Sim3Tracker.cpp:414-607   (the SSE/NEON entry points :384-411 call it)
//   calcSim3WeightsAndResidual[SSE]     :748-856 / :611-736
//   calcSim3LGS[SSE]                    :992-1047 / :858-983
//   LGS4 / LGS7                         C/Tracking/LGSX.h:45-176, :411-443
//   trackFrameSim3 (LM loop)            :149-378
//   Sim3::exp / operator* / inverse     thirdparty/Sophus/sophus/sim3.hpp:417-428, :160-163, :169-173, :608-650,
//                                       rxso3.hpp:416-425
//   Quaternionf::setFromTwoVectors      Eigen/src/Geometry/Quaternion.h (absent dependency; published algorithm restated)
// PINNED: the reference ships no test or fixture for this function, but its Tracking/Sim3Tracker.cpp compiles unchanged into
// oracle/_ref (stand-in Eigen / Sophus headers); tests/test_ref_pin_cpu.py::test_sim3_* compare this file with it bit for bit
// (fixed-transformation buffers / weights / 7x7 system, whole trackFrameSim3 calls incl. the Hessian; SSE and scalar builds).
// Shared, and therefore unpinned, algebra: Sim3::exp (orc_sim3_exp.hpp), the 7x7 LDL^T, setFromTwoVectors.
#include <xmmintrin.h>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "orc_sim3.hpp"
#include "orc_sim3_exp.hpp"

namespace orc {

static float* alloc16s(size_t n) {
  void* p = nullptr;
  if (posix_memalign(&p, 16, n * sizeof(float))) return nullptr;
  memset(p, 0, n * sizeof(float));
  return (float*)p;
}
static inline void interp43s(const float* mat4, float x, float y, int width, float out[3]) {   // globalFuncs.h:63-77
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
#define SSEE(v, i) (((const float*)&(v))[i])

// ---- Sim3 algebra (double), Sophus semantics on the (unit quaternion, scale, translation) representation ----------
static Sim3d sim3_mul(const Sim3d& a, const Sim3d& b) {   // sim3.hpp:160-163: t += rxso3 * other.t; rxso3 *= other.rxso3
  Sim3d r;
  V3d rt = qapply(a.q, b.t);
  r.t = mk3<double>(a.t[0] + a.s * rt[0], a.t[1] + a.s * rt[1], a.t[2] + a.s * rt[2]);
  r.q = qmul(a.q, b.q);
  qnormalize(r.q);     // the product of two scaled quaternions has norm s_a s_b; kept as unit quaternion + scale here
  r.s = a.s * b.s;
  return r;
}


// ---- LGS4 / LGS7 --------------------------------------------------------------------------------------------------
struct LGS4s {
  float A[16], b[4], error;
  size_t num_constraints;
  alignas(16) float SSEData[4 * 15];
  void initialize() { memset(A, 0, sizeof(A)); memset(b, 0, sizeof(b)); memset(SSEData, 0, sizeof(SSEData)); error = 0; num_constraints = 0; }
  void update(const float J[4], float res, float weight) {   // LGSX.h:166-172
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) A[i * 4 + j] += J[i] * J[j] * weight;
    float rw = res * weight;
    for (int i = 0; i < 4; i++) b[i] -= J[i] * rw;
    error += res * res * weight;
    num_constraints += 1;
  }
  void updateSSE(const __m128 J[4], const __m128& res, const __m128& weight) {   // LGSX.h:130-163
    int k = 0;
    for (int i = 0; i < 4; i++) {
      __m128 Jw = _mm_mul_ps(J[i], weight);
      for (int j = i; j < 4; j++, k++) _mm_store_ps(SSEData + 4 * k, _mm_add_ps(_mm_load_ps(SSEData + 4 * k), _mm_mul_ps(Jw, J[j])));
    }
    __m128 resw = _mm_mul_ps(res, weight);
    for (int i = 0; i < 4; i++) _mm_store_ps(SSEData + 4 * (10 + i), _mm_add_ps(_mm_load_ps(SSEData + 4 * (10 + i)), _mm_mul_ps(resw, J[i])));
    _mm_store_ps(SSEData + 4 * 14, _mm_add_ps(_mm_load_ps(SSEData + 4 * 14), _mm_mul_ps(resw, res)));
    num_constraints += 4;
  }
  void finishNoDivide() {   // LGSX.h:65-126
    int k = 0;
    for (int i = 0; i < 4; i++)
      for (int j = i; j < 4; j++, k++) {
        const float* a = SSEData + 4 * k;
        A[i * 4 + j] += a[0] + a[1] + a[2] + a[3];
        A[j * 4 + i] = A[i * 4 + j];
      }
    for (int i = 0; i < 4; i++) { const float* a = SSEData + 4 * (10 + i); b[i] -= a[0] + a[1] + a[2] + a[3]; }
    const float* a = SSEData + 4 * 14;
    error += a[0] + a[1] + a[2] + a[3];
  }
};
static void lgs6_updateSSE_s(LGS6& ls, const __m128 J[6], const __m128& res, const __m128& weight) {   // LGSX.h:328-386
  float* S = ls.SSEData;
  int k = 0;
  for (int i = 0; i < 6; i++) {
    __m128 Jiw = _mm_mul_ps(J[i], weight);
    for (int j = i; j < 6; j++, k++) _mm_store_ps(S + 4 * k, _mm_add_ps(_mm_load_ps(S + 4 * k), _mm_mul_ps(Jiw, J[j])));
  }
  __m128 resw = _mm_mul_ps(res, weight);
  for (int i = 0; i < 6; i++) _mm_store_ps(S + 4 * (21 + i), _mm_add_ps(_mm_load_ps(S + 4 * (21 + i)), _mm_mul_ps(resw, J[i])));
  _mm_store_ps(S + 4 * 27, _mm_add_ps(_mm_load_ps(S + 4 * 27), _mm_mul_ps(resw, res)));
  ls.num_constraints += 6;
}
static void lgs7_from(LGS7s& l7, const LGS6& l6, const LGS4s& l4) {   // LGSX.h:424-442
  memset(l7.A, 0, sizeof(l7.A));
  memset(l7.b, 0, sizeof(l7.b));
  for (int i = 0; i < 6; i++) { l7.b[i] = l6.b[i]; for (int j = 0; j < 6; j++) l7.A[i * 7 + j] = l6.A[i * 6 + j]; }
  const int remap[4] = {2, 3, 4, 6};
  for (int i = 0; i < 4; i++) {
    l7.b[remap[i]] += l4.b[i];
    for (int j = 0; j < 4; j++) l7.A[remap[i] * 7 + remap[j]] += l4.A[i * 4 + j];
  }
  l7.num_constraints = l6.num_constraints + l4.num_constraints;
}

// 7x7 LDL^T with diagonal pivoting (Eigen A.ldlt().solve(b)), same restatement as ldlt6_solve
static void ldlt7_solve(const float Ain[49], const float bin[7], float x[7]) {
  const int n = 7;
  float A[7][7];
  int perm[7];
  for (int i = 0; i < n; i++) { perm[i] = i; for (int j = 0; j < n; j++) A[i][j] = Ain[i * n + j]; }
  for (int k = 0; k < n; k++) {
    int piv = k;
    float best = std::fabs(A[k][k]);
    for (int i = k + 1; i < n; i++) { float v = std::fabs(A[i][i]); if (v > best) { best = v; piv = i; } }
    if (piv != k) {
      for (int j = 0; j < n; j++) { float t = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = t; }
      for (int i = 0; i < n; i++) { float t = A[i][k]; A[i][k] = A[i][piv]; A[i][piv] = t; }
      int tp = perm[k]; perm[k] = perm[piv]; perm[piv] = tp;
    }
    float d = A[k][k];
    for (int j = 0; j < k; j++) d -= A[k][j] * A[k][j] * A[j][j];
    A[k][k] = d;
    for (int i = k + 1; i < n; i++) {
      float v = A[i][k];
      for (int j = 0; j < k; j++) v -= A[i][j] * A[k][j] * A[j][j];
      A[i][k] = (d != 0.0f) ? v / d : 0.0f;
    }
  }
  float y[7];
  for (int i = 0; i < n; i++) y[i] = bin[perm[i]];
  for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
  for (int i = 0; i < n; i++) y[i] = (A[i][i] != 0.0f) ? y[i] / A[i][i] : 0.0f;
  for (int i = n - 1; i >= 0; i--) for (int j = i + 1; j < n; j++) y[i] -= A[j][i] * y[j];
  for (int i = 0; i < n; i++) x[perm[i]] = y[i];
}
void sim3_ldlt7_solve(const float A[49], const float b[7], float x[7]) { ldlt7_solve(A, b, x); }

// ---- Sim3Tracker --------------------------------------------------------------------------------------------------
Sim3Tracker::Sim3Tracker(int w, int h, const float K[4], const Params& p) : width(w), height(h), params(p) {
  (void)K;
  size_t n = (size_t)w * h;
  float** bufs[] = {&buf_warped_residual, &buf_warped_dx, &buf_warped_dy, &buf_warped_x, &buf_warped_y, &buf_warped_z, &buf_d,
                    &buf_residual_d, &buf_idepthVar, &buf_warped_idepthVar, &buf_weight_p, &buf_weight_d};
  for (float** b : bufs) *b = alloc16s(n);
}
Sim3Tracker::~Sim3Tracker() {
  float* bufs[] = {buf_warped_residual, buf_warped_dx, buf_warped_dy, buf_warped_x, buf_warped_y, buf_warped_z, buf_d,
                   buf_residual_d, buf_idepthVar, buf_warped_idepthVar, buf_weight_p, buf_weight_d};
  for (float* b : bufs) free(b);
}

// Eigen Quaternionf::setFromTwoVectors(a, b) followed by toRotationMatrix() * rotMatUnscaled (Sim3Tracker.cpp:455-464)
static void roll_matrix(const M3f& rotMatUnscaled, float& xRoll0, float& xRoll1, float& yRoll0, float& yRoll1) {
  V3f fwd = mk3<float>(0, 0, -1);
  V3f rf = matvec(rotMatUnscaled, fwd);
  // v0 = a.normalized(), v1 = b.normalized()
  float n0 = std::sqrt(dot3(rf, rf));
  V3f v0 = mk3<float>(rf[0] / n0, rf[1] / n0, rf[2] / n0);
  V3f v1 = fwd;
  float c = dot3(v1, v0);
  Quat<float> q;
  if (c < -1.0f + 1e-5f) {   // dummy_precision<float>: vectors nearly opposite — not reachable for a forward-looking pair of views
    q.w = 0; q.x = 1; q.y = 0; q.z = 0;
  } else {
    V3f axis = mk3<float>(v0[1] * v1[2] - v0[2] * v1[1], v0[2] * v1[0] - v0[0] * v1[2], v0[0] * v1[1] - v0[1] * v1[0]);
    float s = std::sqrt((1.0f + c) * 2.0f);
    float invs = 1.0f / s;
    q.x = axis[0] * invs; q.y = axis[1] * invs; q.z = axis[2] * invs;
    q.w = s * 0.5f;
  }
  M3f rollMat = matmat(qrot(q), rotMatUnscaled);
  xRoll0 = rollMat(0, 0); xRoll1 = rollMat(0, 1); yRoll0 = rollMat(1, 0); yRoll1 = rollMat(1, 1);
}

void Sim3Tracker::calcSim3Buffers(TrackingReference* reference, Frame* frame, const Sim3d& referenceToFrame, int level) {
  int w = frame->width(level), h = frame->height(level);
  float fx_l = frame->K[level](0, 0), fy_l = frame->K[level](1, 1), cx_l = frame->K[level](0, 2), cy_l = frame->K[level](1, 2);
  M3d Rd = referenceToFrame.rotationMatrix();
  M3f rotMat, rotMatUnscaled;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      rotMat.m[i][j] = (float)(referenceToFrame.s * Rd.m[i][j]);   // rxso3().matrix() = scale * R, cast to float
      rotMatUnscaled.m[i][j] = (float)Rd.m[i][j];
    }
  V3f transVec = mk3<float>((float)referenceToFrame.t[0], (float)referenceToFrame.t[1], (float)referenceToFrame.t[2]);
  float xRoll0, xRoll1, yRoll0, yRoll1;
  roll_matrix(rotMatUnscaled, xRoll0, xRoll1, yRoll0, yRoll1);

  const int refNum = reference->numData[level];
  const float* refPoint = reference->posData[level].data();
  const float* refColVar = reference->colorAndVarData[level].data();
  const float* refGrad = reference->gradData[level].data();
  const float* frame_idepth = frame->idepth(level);
  const float* frame_idepthVar = frame->idepthVar(level);
  const float* frame_grad = frame->gradients(level);

  float sxx = 0, syy = 0, sx = 0, sy = 0, sw = 0;
  float usageCount = 0;
  int idx = 0;
  for (int i = 0; i < refNum; i++) {
    V3f p = mk3<float>(refPoint[3 * i], refPoint[3 * i + 1], refPoint[3 * i + 2]);
    V3f Wxp = matvec(rotMat, p);
    Wxp = mk3<float>(Wxp[0] + transVec[0], Wxp[1] + transVec[1], Wxp[2] + transVec[2]);
    float u_new = (Wxp[0] / Wxp[2]) * fx_l + cx_l;
    float v_new = (Wxp[1] / Wxp[2]) * fy_l + cy_l;
    if (!(u_new > 1 && v_new > 1 && u_new < w - 2 && v_new < h - 2)) continue;
    buf_warped_x[idx] = Wxp[0]; buf_warped_y[idx] = Wxp[1]; buf_warped_z[idx] = Wxp[2];
    float resInterp[3];
    interp43s(frame_grad, u_new, v_new, w, resInterp);
    // USE_ESM_TRACKING == 1 (settings.h:85)
    float rotatedGradX = xRoll0 * refGrad[2 * i] + xRoll1 * refGrad[2 * i + 1];
    float rotatedGradY = yRoll0 * refGrad[2 * i] + yRoll1 * refGrad[2 * i + 1];
    buf_warped_dx[idx] = fx_l * 0.5f * (resInterp[0] + rotatedGradX);
    buf_warped_dy[idx] = fy_l * 0.5f * (resInterp[1] + rotatedGradY);
    float c1 = affineEstimation_a * refColVar[2 * i] + affineEstimation_b;
    float c2 = resInterp[2];
    float residual_p = c1 - c2;
    float weight = fabsf(residual_p) < 2.0f ? 1 : 2.0f / fabsf(residual_p);
    sxx += c1 * c1 * weight; syy += c2 * c2 * weight; sx += c1 * weight; sy += c2 * weight; sw += weight;
    buf_warped_residual[idx] = residual_p;
    buf_idepthVar[idx] = refColVar[2 * i + 1];
    int idx_rounded = (int)(u_new + 0.5f) + w * (int)(v_new + 0.5f);
    float var_frameDepth = frame_idepthVar[idx_rounded];
    float ref_idepth = 1.0f / Wxp[2];
    buf_d[idx] = 1.0f / p[2];
    if (var_frameDepth > 0) {
      buf_residual_d[idx] = ref_idepth - frame_idepth[idx_rounded];
      buf_warped_idepthVar[idx] = var_frameDepth;
    } else {
      buf_residual_d[idx] = -1;
      buf_warped_idepthVar[idx] = -1;
    }
    idx++;
    float depthChange = p[2] / Wxp[2];
    usageCount += depthChange < 1 ? depthChange : 1;
  }
  buf_warped_size = idx;
  pointUsage = usageCount / (float)refNum;
  affineEstimation_a_lastIt = sqrtf((syy - sy * sy / sw) / (sxx - sx * sx / sw));
  affineEstimation_b_lastIt = (sy - affineEstimation_a_lastIt * sx) / sw;
}

Sim3ResidualStruct Sim3Tracker::calcSim3WeightsAndResidual(const Sim3d& referenceToFrame) {
  float tx = (float)referenceToFrame.t[0], ty = (float)referenceToFrame.t[1], tz = (float)referenceToFrame.t[2];
  Sim3ResidualStruct sumRes;
  memset(&sumRes, 0, sizeof(sumRes));
  for (int i = 0; i < buf_warped_size; i++) {
    float px = buf_warped_x[i], py = buf_warped_y[i], pz = buf_warped_z[i], d = buf_d[i];
    float rp = buf_warped_residual[i], rd = buf_residual_d[i], gx = buf_warped_dx[i], gy = buf_warped_dy[i];
    float s = settings.var_weight * buf_idepthVar[i];
    float sv = settings.var_weight * buf_warped_idepthVar[i];
    float g0 = (tx * pz - tz * px) / (pz * pz * d);
    float g1 = (ty * pz - tz * py) / (pz * pz * d);
    float g2 = (pz - tz) / (pz * pz * d);
    float drpdd = gx * g0 + gy * g1;
    float w_p = 1.0f / (params.cameraPixelNoise2 + s * drpdd * drpdd);
    float w_d = 1.0f / (sv + g2 * g2 * s);
    float weighted_rd = fabs(rd * sqrtf(w_d));
    float weighted_rp = fabs(rp * sqrtf(w_p));
    float weighted_abs_res = sv > 0 ? weighted_rd + weighted_rp : weighted_rp;
    float wh = fabs(weighted_abs_res < settings.huber_d ? 1 : settings.huber_d / weighted_abs_res);
    if (sv > 0) { sumRes.sumResD += wh * w_d * rd * rd; sumRes.numTermsD++; }
    sumRes.sumResP += wh * w_p * rp * rp;
    sumRes.numTermsP++;
    buf_weight_p[i] = wh * w_p;
    buf_weight_d[i] = sv > 0 ? wh * w_d : 0;
  }
  sumRes.mean = (sumRes.sumResD + sumRes.sumResP) / (sumRes.numTermsD + sumRes.numTermsP);
  sumRes.meanD = sumRes.sumResD / sumRes.numTermsD;
  sumRes.meanP = sumRes.sumResP / sumRes.numTermsP;
  return sumRes;
}

static inline __m128 rcp_mode(__m128 v, bool exact) { return exact ? _mm_div_ps(_mm_set1_ps(1.0f), v) : _mm_rcp_ps(v); }

Sim3ResidualStruct Sim3Tracker::calcSim3WeightsAndResidualSSE(const Sim3d& referenceToFrame, bool exactRcp) {
  const __m128 txs = _mm_set1_ps((float)referenceToFrame.t[0]);
  const __m128 tys = _mm_set1_ps((float)referenceToFrame.t[1]);
  const __m128 tzs = _mm_set1_ps((float)referenceToFrame.t[2]);
  const __m128 zeros = _mm_set1_ps(0.0f), ones = _mm_set1_ps(1.0f);
  const __m128 depthVarFacs = _mm_set1_ps((float)settings.var_weight);
  const __m128 sigma_i2s = _mm_set1_ps((float)params.cameraPixelNoise2);
  const __m128 huber_ress = _mm_set1_ps((float)settings.huber_d);
  __m128 sumResP = zeros, sumResD = zeros, numTermsD = zeros;
  Sim3ResidualStruct sumRes;
  memset(&sumRes, 0, sizeof(sumRes));
  for (int i = 0; i < buf_warped_size - 3; i += 4) {
    __m128 pzs = _mm_load_ps(buf_warped_z + i);
    __m128 pz2ds = rcp_mode(_mm_mul_ps(_mm_mul_ps(pzs, pzs), _mm_load_ps(buf_d + i)), exactRcp);
    __m128 g0s = _mm_mul_ps(_mm_sub_ps(_mm_mul_ps(pzs, txs), _mm_mul_ps(_mm_load_ps(buf_warped_x + i), tzs)), pz2ds);
    __m128 g1s = _mm_mul_ps(_mm_sub_ps(_mm_mul_ps(pzs, tys), _mm_mul_ps(_mm_load_ps(buf_warped_y + i), tzs)), pz2ds);
    __m128 g2s = _mm_mul_ps(_mm_sub_ps(pzs, tzs), pz2ds);
    __m128 drpdds = _mm_add_ps(_mm_mul_ps(g0s, _mm_load_ps(buf_warped_dx + i)), _mm_mul_ps(g1s, _mm_load_ps(buf_warped_dy + i)));
    __m128 w_ps = rcp_mode(_mm_add_ps(sigma_i2s, _mm_mul_ps(drpdds, _mm_mul_ps(drpdds, _mm_mul_ps(depthVarFacs, _mm_load_ps(buf_idepthVar + i))))), exactRcp);
    __m128 w_ds = rcp_mode(_mm_add_ps(_mm_load_ps(buf_warped_idepthVar + i), _mm_mul_ps(g2s, _mm_mul_ps(g2s, _mm_mul_ps(depthVarFacs, _mm_load_ps(buf_idepthVar + i))))), exactRcp);
    __m128 weighted_rps = _mm_mul_ps(_mm_load_ps(buf_warped_residual + i), _mm_sqrt_ps(w_ps));
    weighted_rps = _mm_max_ps(weighted_rps, _mm_sub_ps(zeros, weighted_rps));
    __m128 weighted_rds = _mm_mul_ps(_mm_load_ps(buf_residual_d + i), _mm_sqrt_ps(w_ds));
    weighted_rds = _mm_max_ps(weighted_rds, _mm_sub_ps(zeros, weighted_rds));
    __m128 depthValid = _mm_cmplt_ps(zeros, _mm_load_ps(buf_warped_idepthVar + i));
    __m128 weighted_abs_ress = _mm_add_ps(_mm_and_ps(weighted_rds, depthValid), weighted_rps);
    __m128 whs = _mm_cmplt_ps(weighted_abs_ress, huber_ress);
    whs = _mm_or_ps(_mm_and_ps(whs, ones), _mm_andnot_ps(whs, _mm_mul_ps(huber_ress, rcp_mode(weighted_abs_ress, exactRcp))));
    if (i + 3 < buf_warped_size) {
      numTermsD = _mm_add_ps(numTermsD, _mm_and_ps(depthValid, ones));
      sumResD = _mm_add_ps(sumResD, _mm_and_ps(depthValid, _mm_mul_ps(whs, _mm_mul_ps(weighted_rds, weighted_rds))));
      sumResP = _mm_add_ps(sumResP, _mm_mul_ps(whs, _mm_mul_ps(weighted_rps, weighted_rps)));
    }
    _mm_store_ps(buf_weight_p + i, _mm_mul_ps(whs, w_ps));
    _mm_store_ps(buf_weight_d + i, _mm_and_ps(depthValid, _mm_mul_ps(whs, w_ds)));
  }
  sumRes.sumResP = SSEE(sumResP, 0) + SSEE(sumResP, 1) + SSEE(sumResP, 2) + SSEE(sumResP, 3);
  sumRes.numTermsP = (buf_warped_size >> 2) << 2;
  sumRes.sumResD = SSEE(sumResD, 0) + SSEE(sumResD, 1) + SSEE(sumResD, 2) + SSEE(sumResD, 3);
  sumRes.numTermsD = SSEE(numTermsD, 0) + SSEE(numTermsD, 1) + SSEE(numTermsD, 2) + SSEE(numTermsD, 3);
  sumRes.mean = (sumRes.sumResD + sumRes.sumResP) / (sumRes.numTermsD + sumRes.numTermsP);
  sumRes.meanD = sumRes.sumResD / sumRes.numTermsD;
  sumRes.meanP = sumRes.sumResP / sumRes.numTermsP;
  return sumRes;
}

void Sim3Tracker::calcSim3LGS(LGS7s& ls7) {
  LGS4s ls4;
  LGS6 ls6;
  ls6.initialize();
  ls4.initialize();
  for (int i = 0; i < buf_warped_size; i++) {
    float px = buf_warped_x[i], py = buf_warped_y[i], pz = buf_warped_z[i];
    float wp = buf_weight_p[i], wd = buf_weight_d[i], rp = buf_warped_residual[i], rd = buf_residual_d[i];
    float gx = buf_warped_dx[i], gy = buf_warped_dy[i];
    float z = 1.0f / pz;
    float z_sqr = 1.0f / (pz * pz);
    float v[6], v4[4];
    v[0] = z * gx + 0;
    v[1] = 0 + z * gy;
    v[2] = (-px * z_sqr) * gx + (-py * z_sqr) * gy;
    v[3] = (float)((-px * py * z_sqr) * gx + (-(1.0 + py * py * z_sqr)) * gy);   // the reference's 1.0 is a double literal
    v[4] = (float)((1.0 + px * px * z_sqr) * gx + (px * py * z_sqr) * gy);
    v[5] = (-py * z) * gx + (px * z) * gy;
    v4[0] = z_sqr; v4[1] = z_sqr * py; v4[2] = -z_sqr * px; v4[3] = z;
    ls6.update(v, rp, wp);
    ls4.update(v4, rd, wd);
  }
  ls4.finishNoDivide();
  ls6.finishNoDivide();
  lgs7_from(ls7, ls6, ls4);
}

void Sim3Tracker::calcSim3LGSSSE(LGS7s& ls7, bool exactRcp) {
  LGS4s ls4;
  LGS6 ls6;
  ls6.initialize();
  ls4.initialize();
  const __m128 zeros = _mm_set1_ps(0.0f);
  for (int i = 0; i < buf_warped_size - 3; i += 4) {
    __m128 val1, val2, val3, val4;
    __m128 J4[4], J6[6];
    __m128 pz = rcp_mode(_mm_load_ps(buf_warped_z + i), exactRcp);
    J4[3] = pz;
    __m128 gx = _mm_load_ps(buf_warped_dx + i);
    J6[0] = _mm_mul_ps(pz, gx);
    __m128 gy = _mm_load_ps(buf_warped_dy + i);
    J6[1] = _mm_mul_ps(pz, gy);
    __m128 px = _mm_load_ps(buf_warped_x + i);
    val1 = _mm_mul_ps(_mm_mul_ps(px, gy), pz);
    __m128 py = _mm_load_ps(buf_warped_y + i);
    val2 = _mm_mul_ps(_mm_mul_ps(py, gx), pz);
    J6[5] = _mm_sub_ps(val1, val2);
    pz = _mm_mul_ps(pz, pz);
    J4[0] = pz;
    J4[1] = _mm_mul_ps(pz, py);
    J4[2] = _mm_sub_ps(zeros, _mm_mul_ps(pz, px));
    val1 = _mm_mul_ps(_mm_mul_ps(px, gx), pz);
    val2 = _mm_mul_ps(_mm_mul_ps(py, gy), pz);
    val3 = _mm_sub_ps(zeros, _mm_add_ps(val1, val2));
    J6[2] = val3;
    val3 = _mm_mul_ps(val1, py);
    val4 = _mm_add_ps(gy, val3);
    val3 = _mm_mul_ps(val2, py);
    val4 = _mm_add_ps(val3, val4);
    J6[3] = _mm_sub_ps(zeros, val4);
    val3 = _mm_mul_ps(val1, px);
    val4 = _mm_add_ps(gx, val3);
    val3 = _mm_mul_ps(val2, px);
    J6[4] = _mm_add_ps(val4, val3);
    // `if (i + 3 < buf_warped_size)` is always true inside this loop: the last size % 4 points never reach the system
    ls4.updateSSE(J4, _mm_load_ps(buf_residual_d + i), _mm_load_ps(buf_weight_d + i));
    lgs6_updateSSE_s(ls6, J6, _mm_load_ps(buf_warped_residual + i), _mm_load_ps(buf_weight_p + i));
  }
  ls4.finishNoDivide();
  ls6.finishNoDivide();
  lgs7_from(ls7, ls6, ls4);
}

Sim3ResidualStruct Sim3Tracker::callWeights(const Sim3d& T) {
  if (mode == TRACKER_SCALAR) return calcSim3WeightsAndResidual(T);
  return calcSim3WeightsAndResidualSSE(T, mode == TRACKER_SSE_EXACT_RCP);
}
void Sim3Tracker::callLGS(LGS7s& ls7) {
  if (mode == TRACKER_SCALAR) calcSim3LGS(ls7);
  else calcSim3LGSSSE(ls7, mode == TRACKER_SSE_EXACT_RCP);
}

void Sim3Tracker::evaluate(TrackingReference* reference, Frame* frame, const Sim3d& referenceToFrame, int level, float a, float b,
                           Sim3EvalRecord* out) {
  affineEstimation_a = a; affineEstimation_b = b;
  reference->makePointCloud(level);
  calcSim3Buffers(reference, frame, referenceToFrame, level);
  Sim3ResidualStruct r = callWeights(referenceToFrame);
  LGS7s ls7;
  callLGS(ls7);
  out->warped_size = buf_warped_size;
  out->pointUsage = pointUsage;
  out->affine_a_lastIt = affineEstimation_a_lastIt; out->affine_b_lastIt = affineEstimation_b_lastIt;
  out->res = r;
  memcpy(out->A, ls7.A, sizeof(ls7.A));
  memcpy(out->b, ls7.b, sizeof(ls7.b));
  out->num_constraints = (double)ls7.num_constraints;
}

Sim3d Sim3Tracker::trackFrameSim3(TrackingReference* reference, Frame* frame, const Sim3d& frameToReference_initialEstimate, int startLevel,
                                  int finalLevel) {
  diverged = false;
  affineEstimation_a = 1; affineEstimation_b = 0;
  numEvaluations = 0;
  Sim3d referenceToFrame = frameToReference_initialEstimate.inverse();
  LGS7s ls7;
  memset(&ls7, 0, sizeof(ls7));
  Sim3ResidualStruct finalResidual;
  memset(&finalResidual, 0, sizeof(finalResidual));
  bool warp_update_up_to_date = false;
  for (int lvl = startLevel; lvl >= finalLevel; lvl--) {
    if (settings.maxItsPerLvl[lvl] == 0) continue;
    reference->makePointCloud(lvl);
    calcSim3Buffers(reference, frame, referenceToFrame, lvl);
    numEvaluations++;
    if (buf_warped_size < 0.5 * 0.01 * (width >> lvl) * (height >> lvl) || buf_warped_size < 10) { diverged = true; return Sim3d(); }
    Sim3ResidualStruct lastErr = callWeights(referenceToFrame);
    if (params.useAffineLightningEstimation) { affineEstimation_a = affineEstimation_a_lastIt; affineEstimation_b = affineEstimation_b_lastIt; }
    float LM_lambda = settings.lambdaInitial[lvl];
    warp_update_up_to_date = false;
    for (int iteration = 0; iteration < settings.maxItsPerLvl[lvl]; iteration++) {
      callLGS(ls7);
      warp_update_up_to_date = true;
      int incTry = 0;
      while (true) {
        float b[7], A[49], inc[7];
        const float nc = (float)ls7.num_constraints;
        for (int i = 0; i < 7; i++) b[i] = -ls7.b[i] / nc;
        for (int i = 0; i < 49; i++) A[i] = ls7.A[i] / nc;
        for (int i = 0; i < 7; i++) A[i * 7 + i] *= 1 + LM_lambda;
        ldlt7_solve(A, b, inc);
        incTry++;
        float absInc = 0;
        for (int i = 0; i < 7; i++) absInc += inc[i] * inc[i];
        if (!(absInc >= 0 && absInc < 1)) { memset(lastSim3Hessian, 0, sizeof(lastSim3Hessian)); return Sim3d(); }
        double incd[7];
        for (int i = 0; i < 7; i++) incd[i] = (double)inc[i];
        Sim3d new_referenceToFrame = sim3_mul(sim3_exp(incd), referenceToFrame);
        calcSim3Buffers(reference, frame, new_referenceToFrame, lvl);
        numEvaluations++;
        if (buf_warped_size < 0.5 * 0.01 * (width >> lvl) * (height >> lvl) || buf_warped_size < 10) { diverged = true; return Sim3d(); }
        Sim3ResidualStruct error = callWeights(new_referenceToFrame);
        if (error.mean < lastErr.mean) {
          referenceToFrame = new_referenceToFrame;
          warp_update_up_to_date = false;
          if (params.useAffineLightningEstimation) { affineEstimation_a = affineEstimation_a_lastIt; affineEstimation_b = affineEstimation_b_lastIt; }
          if (error.mean / lastErr.mean > settings.convergenceEps[lvl]) iteration = settings.maxItsPerLvl[lvl];
          finalResidual = lastErr = error;
          if (LM_lambda <= 0.2) LM_lambda = 0;
          else LM_lambda *= settings.lambdaSuccessFac;
          break;
        } else {
          if (!(absInc > settings.stepSizeMin[lvl])) { iteration = settings.maxItsPerLvl[lvl]; break; }
          if (LM_lambda == 0) LM_lambda = 0.2;
          else LM_lambda *= std::pow(settings.lambdaFailFac, incTry);
        }
      }
    }
  }
  if (!warp_update_up_to_date) {
    reference->makePointCloud(finalLevel);
    calcSim3Buffers(reference, frame, referenceToFrame, finalLevel);
    numEvaluations++;
    finalResidual = callWeights(referenceToFrame);
    callLGS(ls7);
  }
  memcpy(lastSim3Hessian, ls7.A, sizeof(ls7.A));
  if (referenceToFrame.s <= 0) { diverged = true; return Sim3d(); }
  lastResidual = finalResidual.mean;
  lastDepthResidual = finalResidual.meanD;
  lastPhotometricResidual = finalResidual.meanP;
  return referenceToFrame.inverse();
}

}  // namespace orc
