// ORACLE — TEST INFRASTRUCTURE ONLY.  STAND-IN for sophus/se3.hpp (the vendored Sophus under
// /root/reference/lsd_slam_core/thirdparty/Sophus needs the real Eigen, which is absent here).  Thin wrappers over the
// oracle's restatement of Sophus v0.9a (oracle/orc_math.hpp: se3.hpp:160-172,262-270,406-428, so3.hpp:150-202,342-369),
// which tests/test_oracle_cpu.py checks against the element list of the reference's own sophus/test_se3.cpp.  The group
// algebra is deliberately SHARED with the oracle (see shim/Eigen/Core).
#pragma once
#include <Eigen/Core>

namespace Sophus {

template <typename T, int Options = 0> class SE3Group {
 public:
  typedef Eigen::Matrix<T, 3, 1> Point;
  typedef Eigen::Matrix<T, 6, 1> Tangent;
  orc::SE3<T> g;
  Point t_;   // mirror of g.t so that translation() can hand out a reference

  SE3Group() { sync(); }
  explicit SE3Group(const orc::SE3<T>& o) : g(o) { sync(); }
  SE3Group(const Eigen::Quaternion<T>& q, const Point& t) {
    g.q = q.q;
    orc::qnormalize(g.q);      // SO3Group(const Quaternion&) normalises (so3.hpp:631-633)
    g.t = orc::mk3<T>(t[0], t[1], t[2]);
    sync();
  }
  void sync() { t_[0] = g.t[0]; t_[1] = g.t[1]; t_[2] = g.t[2]; }

  SE3Group inverse() const { return SE3Group(g.inverse()); }
  SE3Group operator*(const SE3Group& o) const { return SE3Group(g * o.g); }
  template <typename U> SE3Group<U, Options> cast() const { return SE3Group<U, Options>(g.template cast<U>()); }
  Eigen::Matrix<T, 3, 3> rotationMatrix() const {
    orc::Mat3<T> R = g.rotationMatrix();
    Eigen::Matrix<T, 3, 3> r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) = R.m[i][j];
    return r;
  }
  const Point& translation() const { return t_; }
  Eigen::Quaternion<T> unit_quaternion() const { return Eigen::Quaternion<T>(g.q); }
  static SE3Group exp(const Tangent& a) { return SE3Group(orc::se3_exp<T>(a.d)); }
  Tangent log() const {
    Tangent r;
    orc::se3_log<T>(g, r.d);
    return r;
  }
};
typedef SE3Group<float> SE3f;
typedef SE3Group<double> SE3d;

template <typename T, int Options = 0> class SO3Group {
 public:
  orc::Quat<T> q;
  SO3Group() { q.w = 1; q.x = q.y = q.z = 0; }
};
typedef SO3Group<float> SO3f;
typedef SO3Group<double> SO3d;

}  // namespace Sophus
