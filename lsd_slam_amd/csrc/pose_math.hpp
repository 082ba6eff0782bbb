// Float / double pose algebra shared by the host entry points and the on-device Levenberg-Marquardt step kernel:
// what SE3Tracker::trackFrame does between two evaluations (C/Tracking/SE3Tracker.cpp:356-363 — 6x6 LDL^T solve,
// Sophus SE3f::exp, left-multiplication, inverse) and the Sim3 inverse of Frame::prepareForStereoWith
// (C/DataStructures/Frame.cpp:295-311).  Semantics follow Sophus v0.9a (unit quaternion + translation, re-normalised
// after every product; thirdparty/Sophus/sophus/se3.hpp:160-172,262-270,406-428, so3.hpp:196-202,342-369).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

#define LSDM_HD __host__ __device__ inline

namespace lsdm {
struct Quatd { double w, x, y, z; };
struct Quatf { float w, x, y, z; };
struct SE3dH { Quatd q; double t[3]; };
struct SE3fH { Quatf q; float t[3]; };
struct Sim3dH { Quatd q; double t[3]; double s; };

template <typename Q> LSDM_HD Q q_conj(const Q& a) { Q r = {a.w, -a.x, -a.y, -a.z}; return r; }
template <typename Q> LSDM_HD Q q_mul(const Q& a, const Q& b) {
  Q r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
LSDM_HD void q_normalize(Quatf& q) {
  float n = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.w /= n; q.x /= n; q.y /= n; q.z /= n;
}
LSDM_HD void q_normalize(Quatd& q) {
  double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.w /= n; q.x /= n; q.y /= n; q.z /= n;
}
// v' = v + w*(2 q_v x v) + q_v x (2 q_v x v)   (Eigen QuaternionBase::_transformVector)
template <typename Q, typename T> LSDM_HD void q_rotate(const Q& q, const T v[3], T out[3]) {
  T ux = q.y * v[2] - q.z * v[1], uy = q.z * v[0] - q.x * v[2], uz = q.x * v[1] - q.y * v[0];
  ux += ux; uy += uy; uz += uz;
  T cx = q.y * uz - q.z * uy, cy = q.z * ux - q.x * uz, cz = q.x * uy - q.y * ux;
  out[0] = v[0] + q.w * ux + cx;
  out[1] = v[1] + q.w * uy + cy;
  out[2] = v[2] + q.w * uz + cz;
}
// Eigen QuaternionBase::toRotationMatrix, row-major
template <typename Q, typename T> LSDM_HD void q_to_rot(const Q& q, T R[9]) {
  T tx = T(2) * q.x, ty = T(2) * q.y, tz = T(2) * q.z;
  T twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  T txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  T tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = T(1) - (tyy + tzz); R[1] = txy - twz;          R[2] = txz + twy;
  R[3] = txy + twz;          R[4] = T(1) - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;          R[7] = tyz + twx;          R[8] = T(1) - (txx + tyy);
}
LSDM_HD void quatf_to_rot(const Quatf& q, float R[9]) { q_to_rot<Quatf, float>(q, R); }
LSDM_HD void quatd_to_rot(const Quatd& q, double R[9]) { q_to_rot<Quatd, double>(q, R); }

LSDM_HD SE3dH se3d_from7(const double p[7]) { SE3dH T; T.q = {p[0], p[1], p[2], p[3]}; T.t[0] = p[4]; T.t[1] = p[5]; T.t[2] = p[6]; return T; }
LSDM_HD void se3d_to7(const SE3dH& T, double p[7]) { p[0] = T.q.w; p[1] = T.q.x; p[2] = T.q.y; p[3] = T.q.z; p[4] = T.t[0]; p[5] = T.t[1]; p[6] = T.t[2]; }
LSDM_HD SE3dH se3d_inverse(const SE3dH& T) {
  SE3dH r;
  r.q = q_conj(T.q);
  double nt[3] = {T.t[0] * -1.0, T.t[1] * -1.0, T.t[2] * -1.0};
  q_rotate<Quatd, double>(r.q, nt, r.t);
  return r;
}
// Sophus SE3Group::cast<>() re-normalises the quaternion in the new scalar type (sophus/se3.hpp:144-149 ->
// so3.hpp:125-129 -> SO3Group(const Quaternion&) so3.hpp:630-633)
LSDM_HD SE3fH se3f_from_d(const SE3dH& T) {
  SE3fH r;
  r.q = {(float)T.q.w, (float)T.q.x, (float)T.q.y, (float)T.q.z};
  q_normalize(r.q);
  for (int i = 0; i < 3; i++) r.t[i] = (float)T.t[i];
  return r;
}
LSDM_HD SE3dH se3d_from_f(const SE3fH& T) {
  SE3dH r;
  r.q = {(double)T.q.w, (double)T.q.x, (double)T.q.y, (double)T.q.z};
  q_normalize(r.q);
  for (int i = 0; i < 3; i++) r.t[i] = (double)T.t[i];
  return r;
}
LSDM_HD SE3fH se3f_inverse(const SE3fH& T) {
  SE3fH r;
  r.q = q_conj(T.q);
  float nt[3] = {T.t[0] * -1.0f, T.t[1] * -1.0f, T.t[2] * -1.0f};
  q_rotate<Quatf, float>(r.q, nt, r.t);
  return r;
}
// Sophus operator*: fastMultiply + normalize
LSDM_HD SE3fH se3f_mul(const SE3fH& a, const SE3fH& b) {
  SE3fH r = a;
  float rt[3];
  q_rotate<Quatf, float>(a.q, b.t, rt);
  for (int i = 0; i < 3; i++) r.t[i] = a.t[i] + rt[i];
  r.q = q_mul(a.q, b.q);
  q_normalize(r.q);
  return r;
}
// Sophus SE3Group<float>::exp; tangent = (upsilon, omega)
LSDM_HD SE3fH se3f_exp(const float a[6]) {
  const float eps = static_cast<float>(1e-5);
  float ox = a[3], oy = a[4], oz = a[5];
  float theta_sq = ox * ox + (oy * oy + oz * oz);
  float theta = sqrtf(theta_sq);
  float half_theta = 0.5f * theta;
  float imag, real;
  if (theta < eps) {
    float theta_po4 = theta_sq * theta_sq;
    imag = 0.5f - static_cast<float>(1.0 / 48.0) * theta_sq + static_cast<float>(1.0 / 3840.0) * theta_po4;
    real = 1.0f - 0.5f * theta_sq + static_cast<float>(1.0 / 384.0) * theta_po4;
  } else {
    float s = sinf(half_theta);
    imag = s / theta;
    real = cosf(half_theta);
  }
  SE3fH r;
  r.q = {real, imag * ox, imag * oy, imag * oz};
  q_normalize(r.q);
  float Om[9] = {0, -oz, oy, oz, 0, -ox, -oy, ox, 0};
  float Om2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      float acc = Om[i * 3 + 0] * Om[0 * 3 + j];
      acc += Om[i * 3 + 1] * Om[1 * 3 + j];
      acc += Om[i * 3 + 2] * Om[2 * 3 + j];
      Om2[i * 3 + j] = acc;
    }
  float V[9];
  if (theta < eps) {
    q_to_rot<Quatf, float>(r.q, V);
  } else {
    float tsq = theta * theta;
    float ca = (1.0f - cosf(theta)) / tsq;
    float cb = (theta - sinf(theta)) / (tsq * theta);
    for (int i = 0; i < 9; i++) V[i] = (((i % 4) == 0 ? 1.0f : 0.0f) + ca * Om[i]) + cb * Om2[i];
  }
  for (int i = 0; i < 3; i++) {
    float acc = V[i * 3 + 0] * a[0];
    acc += V[i * 3 + 1] * a[1];
    acc += V[i * 3 + 2] * a[2];
    r.t[i] = acc;
  }
  return r;
}

LSDM_HD Sim3dH sim3_inverse(const Sim3dH& S) {
  Sim3dH r;
  r.q = q_conj(S.q);
  r.s = 1.0 / S.s;
  double rt[3];
  q_rotate<Quatd, double>(r.q, S.t, rt);
  for (int i = 0; i < 3; i++) r.t[i] = -(rt[i] * r.s);
  return r;
}

// LDL^T with diagonal pivoting on a 6x6 SPD-after-damping system (the reference: Eigen A.ldlt().solve(b)).
LSDM_HD void ldlt6_solve(const float Ain[36], const float bin[6], float x[6]) {
  float M[6][6];
  int p[6];
  for (int i = 0; i < 6; i++) { p[i] = i; for (int j = 0; j < 6; j++) M[i][j] = Ain[i * 6 + j]; }
  for (int k = 0; k < 6; k++) {
    int piv = k;
    float big = fabsf(M[k][k]);
    for (int i = k + 1; i < 6; i++) if (fabsf(M[i][i]) > big) { big = fabsf(M[i][i]); piv = i; }
    if (piv != k) {
      for (int j = 0; j < 6; j++) { float tmp = M[k][j]; M[k][j] = M[piv][j]; M[piv][j] = tmp; }
      for (int i = 0; i < 6; i++) { float tmp = M[i][k]; M[i][k] = M[i][piv]; M[i][piv] = tmp; }
      int tp = p[k]; p[k] = p[piv]; p[piv] = tp;
    }
    float d = M[k][k];
    for (int j = 0; j < k; j++) d -= M[k][j] * M[k][j] * M[j][j];
    M[k][k] = d;
    for (int i = k + 1; i < 6; i++) {
      float v = M[i][k];
      for (int j = 0; j < k; j++) v -= M[i][j] * M[k][j] * M[j][j];
      M[i][k] = d != 0.0f ? v / d : 0.0f;
    }
  }
  float y[6];
  for (int i = 0; i < 6; i++) {
    float v = 0.f;
    for (int j = 0; j < 6; j++) if (p[i] == j) v = bin[j];
    y[i] = v;
  }
  for (int i = 0; i < 6; i++) for (int j = 0; j < i; j++) y[i] -= M[i][j] * y[j];
  for (int i = 0; i < 6; i++) y[i] = M[i][i] != 0.0f ? y[i] / M[i][i] : 0.0f;
  for (int i = 5; i >= 0; i--) for (int j = i + 1; j < 6; j++) y[i] -= M[j][i] * y[j];
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) if (p[i] == j) x[j] = y[i];
}

// 3x3 inverse by cofactors of column 0 / determinant / scaled cofactors — the algorithm Eigen's Matrix3f::inverse()
// uses for fixed size 3, so that fxi, fyi, cxi, cyi carry the same bits as the reference's KInv
// (C/DataStructures/Frame.cpp:409-413, :454-459).
inline void inverse3_eigen(const float K[9], float Ki[9]) {
  auto cof = [&](int i, int j) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return K[i1 * 3 + j1] * K[i2 * 3 + j2] - K[i1 * 3 + j2] * K[i2 * 3 + j1];
  };
  float c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
  float det = c0 * K[0] + (c1 * K[3] + c2 * K[6]);
  float invdet = 1.0f / det;
  Ki[0] = c0 * invdet; Ki[1] = c1 * invdet; Ki[2] = c2 * invdet;
  Ki[3] = cof(0, 1) * invdet; Ki[4] = cof(1, 1) * invdet; Ki[7] = cof(1, 2) * invdet;
  Ki[5] = cof(2, 1) * invdet; Ki[6] = cof(0, 2) * invdet; Ki[8] = cof(2, 2) * invdet;
}
}  // namespace lsdm
