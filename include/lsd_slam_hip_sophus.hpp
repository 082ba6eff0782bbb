// lsd_slam_hip_sophus.hpp — conversions between the adapter's POD types (include/lsd_slam_hip.hpp) and the types
// lsd_slam_core already has: Sophus::SE3d / Sim3d (typedef'd SE3 / Sim3 in C/util/SophusUtil.h:30-47) and Eigen::Matrix3f.
// Include it AFTER <sophus/se3.hpp>, <sophus/sim3.hpp> (or C/util/SophusUtil.h); nothing here needs more of Sophus / Eigen
// than unit_quaternion() / quaternion() / translation() / scale() / setScale(), the (Quaternion, Vector3) constructors and
// operator() of a 3x3 matrix.
//
// Verified by tests/test_abi_cpu.py::test_sophus_adapter_compiles_against_reference_typedefs: the header is compiled together
// with the REFERENCE's own util/SophusUtil.h and DataStructures/FramePoseStruct.h.  In this container the Sophus / Eigen behind
// those typedefs are the stand-ins of oracle/ref/shim — the vendored thirdparty/Sophus needs the real Eigen (Eigen::Map,
// Eigen::QuaternionBase, Eigen::internal::traits specialisations, Matrix block expressions), which is absent here.
#pragma once
#include "lsd_slam_hip.hpp"

namespace lsd_slam_hip {

template <typename SophusSE3> inline SE3 toHip(const SophusSE3& T) {
  SE3 r;
  const auto q = T.unit_quaternion();
  r.q[0] = q.w(); r.q[1] = q.x(); r.q[2] = q.y(); r.q[3] = q.z();
  for (int i = 0; i < 3; i++) r.t[i] = T.translation()[i];
  return r;
}
template <typename SophusSE3> inline SophusSE3 fromHip(const SE3& T) {
  typedef decltype(SophusSE3().unit_quaternion()) Quat;
  typedef typename std::decay<decltype(SophusSE3().translation())>::type Vec3;
  return SophusSE3(Quat(T.q[0], T.q[1], T.q[2], T.q[3]), Vec3(T.t[0], T.t[1], T.t[2]));
}
// Sim3: Sophus keeps the scale as the norm of its quaternion; the adapter as a separate field
template <typename SophusSim3> inline Sim3 toHipSim3(const SophusSim3& S) {
  Sim3 r;
  const auto q = S.quaternion();
  const double s = S.scale();
  r.q[0] = q.w() / s; r.q[1] = q.x() / s; r.q[2] = q.y() / s; r.q[3] = q.z() / s;
  for (int i = 0; i < 3; i++) r.t[i] = S.translation()[i];
  r.s = s;
  return r;
}
template <typename SophusSim3> inline SophusSim3 fromHipSim3(const Sim3& S) {
  typedef decltype(SophusSim3().quaternion()) Quat;
  typedef typename std::decay<decltype(SophusSim3().translation())>::type Vec3;
  SophusSim3 r(Quat(S.q[0], S.q[1], S.q[2], S.q[3]), Vec3(S.t[0], S.t[1], S.t[2]));
  r.setScale(S.s);
  return r;
}
template <typename EigenMatrix3f> inline Mat3f toHipK(const EigenMatrix3f& K) {
  Mat3f m;
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m.m[r * 3 + c] = K(r, c);
  return m;
}

}  // namespace lsd_slam_hip
