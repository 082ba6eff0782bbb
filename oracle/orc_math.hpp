// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// PARITY UNPINNED FOR THIS FILE beyond the Sophus test vectors: Eigen is absent from this machine and the
// vendored Sophus needs it, so this file restates the small amount of Eigen 3.2 / Sophus v0.9a arithmetic
// the hot path relies on (checked against the element list of sophus/test_se3.cpp in tests/test_oracle_cpu.py).
// oracle/_ref (the reference's own sources, see lsd_oracle.hpp) is compiled against THIS arithmetic too.
//
// Minimal float/double linear algebra with the *operation order* of the libraries the reference uses:
//  - fixed-size Matrix3f*Vector3f / Matrix3f*Matrix3f: Eigen 3.2 coefficient-based lazy product,
//    product_coeff_impl<DefaultTraversal>: ((a0*b0) + a1*b1) + a2*b2   (Eigen is an external dependency,
//    absent from /root/reference; version unpinned by the reference's CMake, era-appropriate = 3.2.x)
//  - Vector3f::dot / squaredNorm: Eigen redux_novec_unroller (binary tree): a0*b0 + (a1*b1 + a2*b2)
//  - Matrix3f::inverse(): Eigen compute_inverse<…,3> (cofactors of column 0, det, 1/det, cofactor*invdet)
//  - Sophus SE3f/SO3f: /root/reference/lsd_slam_core/thirdparty/Sophus/sophus/se3.hpp:160-172,262-270,
//    406-428 and so3.hpp:150-175,196-202,342-369; quaternion algebra as in Eigen/Geometry/Quaternion.h.
#pragma once
#include <cmath>
#include <cstring>

namespace orc {

template <typename T> struct Vec3 {
  T v[3];
  T& operator[](int i) { return v[i]; }
  const T& operator[](int i) const { return v[i]; }
};
template <typename T> struct Mat3 {
  T m[3][3];
  T& operator()(int r, int c) { return m[r][c]; }
  const T& operator()(int r, int c) const { return m[r][c]; }
};
typedef Vec3<float> V3f;
typedef Mat3<float> M3f;
typedef Vec3<double> V3d;
typedef Mat3<double> M3d;

template <typename T> static inline Vec3<T> mk3(T a, T b, T c) { Vec3<T> r; r.v[0] = a; r.v[1] = b; r.v[2] = c; return r; }

// Eigen 3.2 coefficient-based product: sequential accumulation over the inner index.
template <typename T> static inline Vec3<T> matvec(const Mat3<T>& A, const Vec3<T>& p) {
  Vec3<T> r;
  for (int i = 0; i < 3; i++) {
    T acc = A.m[i][0] * p.v[0];
    acc += A.m[i][1] * p.v[1];
    acc += A.m[i][2] * p.v[2];
    r.v[i] = acc;
  }
  return r;
}
template <typename T> static inline Mat3<T> matmat(const Mat3<T>& A, const Mat3<T>& B) {
  Mat3<T> r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      T acc = A.m[i][0] * B.m[0][j];
      acc += A.m[i][1] * B.m[1][j];
      acc += A.m[i][2] * B.m[2][j];
      r.m[i][j] = acc;
    }
  return r;
}
// Eigen redux tree for 3 elements: x0 + (x1 + x2)
template <typename T> static inline T dot3(const Vec3<T>& a, const Vec3<T>& b) {
  return a.v[0] * b.v[0] + (a.v[1] * b.v[1] + a.v[2] * b.v[2]);
}
template <typename T> static inline Mat3<T> transpose(const Mat3<T>& A) {
  Mat3<T> r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = A.m[j][i];
  return r;
}
template <typename T> static inline Mat3<T> scale(const Mat3<T>& A, T s) {
  Mat3<T> r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i][j] = A.m[i][j] * s;
  return r;
}

// Eigen compute_inverse<MatrixType,ResultType,3> (Eigen/src/LU/Inverse.h, 3.2.x)
template <typename T> static inline T cofactor3(const Mat3<T>& m, int i, int j) {
  int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m.m[i1][j1] * m.m[i2][j2] - m.m[i1][j2] * m.m[i2][j1];
}
template <typename T> static inline Mat3<T> inverse3(const Mat3<T>& m) {
  T c0 = cofactor3(m, 0, 0), c1 = cofactor3(m, 1, 0), c2 = cofactor3(m, 2, 0);
  // (cofactors_col0.cwiseProduct(matrix.col(0))).sum() -> redux tree x0 + (x1 + x2)
  T det = c0 * m.m[0][0] + (c1 * m.m[1][0] + c2 * m.m[2][0]);
  T invdet = T(1) / det;
  Mat3<T> r;
  r.m[0][0] = c0 * invdet; r.m[0][1] = c1 * invdet; r.m[0][2] = c2 * invdet;
  r.m[1][0] = cofactor3(m, 0, 1) * invdet;
  r.m[1][1] = cofactor3(m, 1, 1) * invdet;
  r.m[2][1] = cofactor3(m, 1, 2) * invdet;
  r.m[1][2] = cofactor3(m, 2, 1) * invdet;
  r.m[2][0] = cofactor3(m, 0, 2) * invdet;
  r.m[2][2] = cofactor3(m, 2, 2) * invdet;
  return r;
}

// ---------------------------------------------------------------------------------------------
// Unit quaternion (w, x, y, z) + translation: Sophus SE3Group<Scalar>.
template <typename T> struct Quat { T w, x, y, z; };

template <typename T> static inline Quat<T> qmul(const Quat<T>& a, const Quat<T>& b) {
  Quat<T> r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
template <typename T> static inline Quat<T> qconj(const Quat<T>& a) { Quat<T> r = {a.w, -a.x, -a.y, -a.z}; return r; }
template <typename T> static inline void qnormalize(Quat<T>& q) {  // so3.hpp:196-202
  T len = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.w /= len; q.x /= len; q.y /= len; q.z /= len;
}
// Eigen QuaternionBase::toRotationMatrix
template <typename T> static inline Mat3<T> qrot(const Quat<T>& q) {
  T tx = T(2) * q.x, ty = T(2) * q.y, tz = T(2) * q.z;
  T twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  T txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  T tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  Mat3<T> r;
  r.m[0][0] = T(1) - (tyy + tzz); r.m[0][1] = txy - twz;          r.m[0][2] = txz + twy;
  r.m[1][0] = txy + twz;          r.m[1][1] = T(1) - (txx + tzz); r.m[1][2] = tyz - twx;
  r.m[2][0] = txz - twy;          r.m[2][1] = tyz + twx;          r.m[2][2] = T(1) - (txx + tyy);
  return r;
}
// Eigen QuaternionBase::_transformVector: uv = 2*(q.vec x v); v + w*uv + q.vec x uv
template <typename T> static inline Vec3<T> qapply(const Quat<T>& q, const Vec3<T>& v) {
  Vec3<T> uv = mk3<T>(q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]);
  uv.v[0] += uv.v[0]; uv.v[1] += uv.v[1]; uv.v[2] += uv.v[2];
  Vec3<T> c = mk3<T>(q.y * uv[2] - q.z * uv[1], q.z * uv[0] - q.x * uv[2], q.x * uv[1] - q.y * uv[0]);
  return mk3<T>(v[0] + q.w * uv[0] + c[0], v[1] + q.w * uv[1] + c[1], v[2] + q.w * uv[2] + c[2]);
}

template <typename T> struct SE3 {
  Quat<T> q;
  Vec3<T> t;
  SE3() { q.w = 1; q.x = q.y = q.z = 0; t = mk3<T>(0, 0, 0); }
  Mat3<T> rotationMatrix() const { return qrot(q); }
  // se3.hpp:169-172
  SE3 inverse() const {
    SE3 r;
    r.q = qconj(q);
    Vec3<T> nt = mk3<T>(t[0] * T(-1), t[1] * T(-1), t[2] * T(-1));
    r.t = qapply(r.q, nt);
    return r;
  }
  // se3.hpp:160-163 (fastMultiply) + 267-270 (normalize)
  SE3 operator*(const SE3& o) const {
    SE3 r = *this;
    Vec3<T> rt = qapply(r.q, o.t);
    r.t = mk3<T>(r.t[0] + rt[0], r.t[1] + rt[1], r.t[2] + rt[2]);
    r.q = qmul(r.q, o.q);
    qnormalize(r.q);
    return r;
  }
  // se3.hpp:144-149 -> so3.hpp:125-129: the cast builds SO3Group<U>(quaternion.cast<U>()), and that constructor
  // re-normalises in the NEW scalar type (so3.hpp:630-633).  (Found by running the oracle against oracle/_ref.)
  template <typename U> SE3<U> cast() const {
    SE3<U> r;
    r.q.w = (U)q.w; r.q.x = (U)q.x; r.q.y = (U)q.y; r.q.z = (U)q.z;
    qnormalize(r.q);
    r.t = mk3<U>((U)t[0], (U)t[1], (U)t[2]);
    return r;
  }
};
typedef SE3<float> SE3f;
typedef SE3<double> SE3d;

template <typename T> static inline T sophus_eps();
template <> inline float sophus_eps<float>() { return static_cast<float>(1e-5); }    // sophus.hpp:52-56
template <> inline double sophus_eps<double>() { return static_cast<double>(1e-10); } // sophus.hpp:43-46

// so3.hpp:342-369 + se3.hpp:406-428.  Tangent order: (upsilon[3], omega[3]).
template <typename T> static inline SE3<T> se3_exp(const T a[6]) {
  Vec3<T> omega = mk3<T>(a[3], a[4], a[5]);
  T theta_sq = dot3(omega, omega);
  T theta = std::sqrt(theta_sq);
  T half_theta = T(0.5) * theta;
  T imag_factor, real_factor;
  if (theta < sophus_eps<T>()) {
    T theta_po4 = theta_sq * theta_sq;
    imag_factor = T(0.5) - T(1.0 / 48.0) * theta_sq + T(1.0 / 3840.0) * theta_po4;
    real_factor = T(1) - T(0.5) * theta_sq + T(1.0 / 384.0) * theta_po4;
  } else {
    T sin_half_theta = std::sin(half_theta);
    imag_factor = sin_half_theta / theta;
    real_factor = std::cos(half_theta);
  }
  SE3<T> r;
  r.q.w = real_factor; r.q.x = imag_factor * omega[0]; r.q.y = imag_factor * omega[1]; r.q.z = imag_factor * omega[2];
  qnormalize(r.q);  // SO3Group(const Quaternion&) ctor normalises (so3.hpp:631-633)

  Mat3<T> Om;  // so3.hpp:423-429 (hat)
  Om.m[0][0] = 0;         Om.m[0][1] = -omega[2]; Om.m[0][2] = omega[1];
  Om.m[1][0] = omega[2];  Om.m[1][1] = 0;         Om.m[1][2] = -omega[0];
  Om.m[2][0] = -omega[1]; Om.m[2][1] = omega[0];  Om.m[2][2] = 0;
  Mat3<T> Om2 = matmat(Om, Om);
  Mat3<T> V;
  if (theta < sophus_eps<T>()) {
    V = qrot(r.q);
  } else {
    T theta_sq2 = theta * theta;
    T ca = (T(1) - std::cos(theta)) / theta_sq2;
    T cb = (theta - std::sin(theta)) / (theta_sq2 * theta);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        V.m[i][j] = ((i == j ? T(1) : T(0)) + ca * Om.m[i][j]) + cb * Om2.m[i][j];
  }
  r.t = matvec(V, mk3<T>(a[0], a[1], a[2]));
  return r;
}

// SO3 log (so3.hpp:464-505 logAndTheta) + SE3 log (se3.hpp:443-476); used by tests only (pose distance).
template <typename T> static inline void se3_log(const SE3<T>& s, T out[6]) {
  T sq_n = s.q.x * s.q.x + s.q.y * s.q.y + s.q.z * s.q.z;
  T n = std::sqrt(sq_n);
  T w = s.q.w;
  T two_atan_nbyw_by_n;
  if (n < sophus_eps<T>()) {
    T sq_w = w * w;
    two_atan_nbyw_by_n = T(2) / w - T(2) * sq_n / (w * sq_w);
  } else {
    if (std::abs(w) < sophus_eps<T>()) {
      two_atan_nbyw_by_n = (w > 0 ? T(M_PI) : -T(M_PI)) / n;
    } else {
      two_atan_nbyw_by_n = T(2) * std::atan(n / w) / n;
    }
  }
  T theta = two_atan_nbyw_by_n * n;
  Vec3<T> omega = mk3<T>(two_atan_nbyw_by_n * s.q.x, two_atan_nbyw_by_n * s.q.y, two_atan_nbyw_by_n * s.q.z);
  Mat3<T> Om;
  Om.m[0][0] = 0;         Om.m[0][1] = -omega[2]; Om.m[0][2] = omega[1];
  Om.m[1][0] = omega[2];  Om.m[1][1] = 0;         Om.m[1][2] = -omega[0];
  Om.m[2][0] = -omega[1]; Om.m[2][1] = omega[0];  Om.m[2][2] = 0;
  Mat3<T> Om2 = matmat(Om, Om);
  Mat3<T> Vinv;
  if (std::abs(theta) < sophus_eps<T>()) {   // se3.hpp:566 (theta is negative for quaternions with w < 0: rotations beyond pi)
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
      Vinv.m[i][j] = (i == j ? T(1) : T(0)) - T(0.5) * Om.m[i][j] + T(1. / 12.) * Om2.m[i][j];
  } else {
    T half = T(0.5) * theta;
    T c = (T(1) - theta * std::cos(half) / (T(2) * std::sin(half))) / (theta * theta);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
      Vinv.m[i][j] = (i == j ? T(1) : T(0)) - T(0.5) * Om.m[i][j] + c * Om2.m[i][j];
  }
  Vec3<T> u = matvec(Vinv, s.t);
  out[0] = u[0]; out[1] = u[1]; out[2] = u[2];
  out[3] = omega[0]; out[4] = omega[1]; out[5] = omega[2];
}

// Sim3 as used by the hot path only: rotation, translation, scale (double); Sophus Sim3d semantics
// p' = s*R*p + t; inverse: R^T/s, -(R^T t)/s.
struct Sim3d {
  Quat<double> q;
  V3d t;
  double s;
  Sim3d() { q.w = 1; q.x = q.y = q.z = 0; t = mk3<double>(0, 0, 0); s = 1; }
  M3d rotationMatrix() const { return qrot(q); }
  Sim3d inverse() const {
    Sim3d r;
    r.q = qconj(q);
    r.s = 1.0 / s;
    V3d rt = qapply(r.q, t);
    r.t = mk3<double>(-(rt[0] * r.s), -(rt[1] * r.s), -(rt[2] * r.s));
    return r;
  }
};

// 6x6 symmetric solve with Eigen-LDLT-style diagonal pivoting (Eigen/src/Cholesky/LDLT.h, unblocked):
// the reference calls A.ldlt().solve(b) (SE3Tracker.cpp:359).  Restated: at step k pick the largest
// remaining |diagonal|, symmetric swap, then the usual LDL^T column update; solve P^T L D L^T P x = b.
static inline void ldlt6_solve(const float Ain[36], const float bin[6], float x[6]) {
  float A[6][6];
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) A[i][j] = Ain[i * 6 + j];
  int perm[6];
  for (int i = 0; i < 6; i++) perm[i] = i;
  const int n = 6;
  for (int k = 0; k < n; k++) {
    int piv = k; float best = std::fabs(A[k][k]);
    for (int i = k + 1; i < n; i++) { float v = std::fabs(A[i][i]); if (v > best) { best = v; piv = i; } }
    if (piv != k) {
      for (int j = 0; j < n; j++) { float tmp = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = tmp; }
      for (int i = 0; i < n; i++) { float tmp = A[i][k]; A[i][k] = A[i][piv]; A[i][piv] = tmp; }
      int tp = perm[k]; perm[k] = perm[piv]; perm[piv] = tp;
    }
    // A[k][k] -= sum_j<k L[k][j]^2 D[j]; L stored in lower part, D on the diagonal.
    float d = A[k][k];
    for (int j = 0; j < k; j++) d -= A[k][j] * A[k][j] * A[j][j];
    A[k][k] = d;
    for (int i = k + 1; i < n; i++) {
      float v = A[i][k];
      for (int j = 0; j < k; j++) v -= A[i][j] * A[k][j] * A[j][j];
      A[i][k] = (d != 0.0f) ? v / d : 0.0f;
    }
  }
  float y[6];
  for (int i = 0; i < n; i++) y[i] = bin[perm[i]];
  for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
  for (int i = 0; i < n; i++) y[i] = (A[i][i] != 0.0f) ? y[i] / A[i][i] : 0.0f;
  for (int i = n - 1; i >= 0; i--) for (int j = i + 1; j < n; j++) y[i] -= A[j][i] * y[j];
  for (int i = 0; i < n; i++) x[perm[i]] = y[i];
}

}  // namespace orc
