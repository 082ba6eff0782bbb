// ORACLE — TEST INFRASTRUCTURE ONLY.  STAND-IN for sophus/sim3.hpp over the oracle's Sim3 (unit quaternion + scale +
// translation; Sophus stores the scale as the norm of a non-unit quaternion, rxso3.hpp:300-312 — same group, see
// oracle/orc_math.hpp Sim3d and oracle/orc_sim3.cpp sim3_mul / sim3_exp).  Shared with the oracle on purpose.
#pragma once
#include "se3.hpp"

#include "../../../orc_sim3_exp.hpp"   // orc::sim3_exp: Sophus Sim3::exp restated once, shared with the oracle

namespace Sophus {

template <typename T, int Options = 0> class Sim3Group {
 public:
  typedef Eigen::Matrix<T, 3, 1> Point;
  typedef Eigen::Matrix<T, 7, 1> Tangent;
  orc::Quat<T> q;   // unit
  T s;
  Point t_;

  Sim3Group() { q.w = 1; q.x = q.y = q.z = 0; s = 1; t_.setZero(); }
  Sim3Group(const Eigen::Quaternion<T>& quat, const Point& t) : t_(t) {
    q = quat.q;
    T n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);   // scale = |quaternion| (rxso3.hpp:311-313)
    s = n;
    q.w /= n; q.x /= n; q.y /= n; q.z /= n;
  }
  void setScale(const T& scale) { s = scale; }           // normalise + scale the quaternion (rxso3.hpp:332-335)
  T scale() const { return s; }
  const Point& translation() const { return t_; }
  Point& translation() { return t_; }
  Eigen::Quaternion<T> quaternion() const { return Eigen::Quaternion<T>(q.w * s, q.x * s, q.y * s, q.z * s); }
  Eigen::Matrix<T, 3, 3> rotationMatrix() const {
    orc::Mat3<T> R = orc::qrot(q);
    Eigen::Matrix<T, 3, 3> r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) = R.m[i][j];
    return r;
  }
  // RxSO3::matrix() (rxso3.hpp:164-170): scale * (normalised quaternion).toRotationMatrix()
  struct RxSO3View {
    Eigen::Matrix<T, 3, 3> m;
    const Eigen::Matrix<T, 3, 3>& matrix() const { return m; }
  };
  RxSO3View rxso3() const {
    RxSO3View v;
    orc::Mat3<T> R = orc::qrot(q);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) v.m(i, j) = s * R.m[i][j];
    return v;
  }
  // oracle/orc_math.hpp Sim3d::inverse
  Sim3Group inverse() const {
    Sim3Group r;
    r.q = orc::qconj(q);
    r.s = T(1) / s;
    orc::Vec3<T> rt = orc::qapply(r.q, orc::mk3<T>(t_[0], t_[1], t_[2]));
    r.t_ = Point(-(rt[0] * r.s), -(rt[1] * r.s), -(rt[2] * r.s));
    return r;
  }
  // oracle/orc_sim3.cpp sim3_mul (sim3.hpp:160-163)
  Sim3Group operator*(const Sim3Group& b) const {
    Sim3Group r;
    orc::Vec3<T> rt = orc::qapply(q, orc::mk3<T>(b.t_[0], b.t_[1], b.t_[2]));
    r.t_ = Point(t_[0] + s * rt[0], t_[1] + s * rt[1], t_[2] + s * rt[2]);
    r.q = orc::qmul(q, b.q);
    orc::qnormalize(r.q);
    r.s = s * b.s;
    return r;
  }
  template <typename U> Sim3Group<U, Options> cast() const {
    Sim3Group<U, Options> r;
    r.q.w = (U)q.w; r.q.x = (U)q.x; r.q.y = (U)q.y; r.q.z = (U)q.z;
    r.s = (U)s;
    r.t_ = t_.template cast<U>();
    return r;
  }
  // only reached from printf debugging (C/Tracking/Sim3Tracker.cpp:273-275, behind printTrackingIterationInfo): not restated
  Tangent log() const { Tangent r; r.setZero(); return r; }
  static Sim3Group exp(const Tangent& a) {
    double ad[7];
    for (int i = 0; i < 7; i++) ad[i] = (double)a[i];
    orc::Sim3d e = orc::sim3_exp(ad);
    Sim3Group r;
    r.q.w = (T)e.q.w; r.q.x = (T)e.q.x; r.q.y = (T)e.q.y; r.q.z = (T)e.q.z;
    r.s = (T)e.s;
    r.t_ = Point((T)e.t[0], (T)e.t[1], (T)e.t[2]);
    return r;
  }
};
typedef Sim3Group<float> Sim3f;
typedef Sim3Group<double> Sim3d;

}  // namespace Sophus
