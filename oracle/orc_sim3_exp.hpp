// ORACLE — TEST INFRASTRUCTURE ONLY.  Sophus Sim3::exp restated (thirdparty/Sophus/sophus/sim3.hpp:585-650: vendored with the reference,
// but it needs Eigen, which this image lacks; held to Sophus' own expMapTest vectors in tests/test_oracle_cpu.py), shared by the oracle's Sim3Tracker (orc_sim3.cpp) and by the stand-in Sophus header the reference build uses
// (oracle/ref/shim/sophus/sim3.hpp): like the SE3 algebra, this arithmetic is common to both sides of the pin.
#pragma once
#include <cmath>
#include "orc_math.hpp"

namespace orc {
inline Sim3d sim3_exp(const double a[7]) {
  const double eps = 1e-10;   // SophusConstants<double>::epsilon
  V3d omega = mk3<double>(a[3], a[4], a[5]);
  double sigma = a[6];
  double scale = std::exp(sigma);
  double theta_sq = dot3(omega, omega);
  double theta = std::sqrt(theta_sq);
  double half_theta = 0.5 * theta;
  double imag, real;
  if (theta < eps) {
    double theta_po4 = theta_sq * theta_sq;
    imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
    real = 1.0 - 0.5 * theta_sq + (1.0 / 384.0) * theta_po4;
  } else {
    imag = std::sin(half_theta) / theta;
    real = std::cos(half_theta);
  }
  Sim3d r;
  r.q.w = real; r.q.x = imag * omega[0]; r.q.y = imag * omega[1]; r.q.z = imag * omega[2];
  qnormalize(r.q);
  r.s = scale;
  M3d Om;
  Om.m[0][0] = 0;         Om.m[0][1] = -omega[2]; Om.m[0][2] = omega[1];
  Om.m[1][0] = omega[2];  Om.m[1][1] = 0;         Om.m[1][2] = -omega[0];
  Om.m[2][0] = -omega[1]; Om.m[2][1] = omega[0];  Om.m[2][2] = 0;
  M3d Om2 = matmat(Om, Om);
  double A, B, C;   // calcW, sim3.hpp:608-650
  if (std::abs(sigma) < eps) {
    C = 1.0;
    if (std::abs(theta) < eps) { A = 0.5; B = 1.0 / 6.0; }
    else { A = (1.0 - std::cos(theta)) / theta_sq; B = (theta - std::sin(theta)) / (theta_sq * theta); }
  } else {
    C = (scale - 1.0) / sigma;
    if (std::abs(theta) < eps) {
      double sigma_sq = sigma * sigma;
      A = ((sigma - 1.0) * scale + 1.0) / sigma_sq;
      B = ((0.5 * sigma * sigma - sigma + 1.0) * scale) / (sigma_sq * sigma);
    } else {
      double sa = scale * std::sin(theta), sb = scale * std::cos(theta), c = theta_sq + sigma * sigma;
      A = (sa * sigma + (1.0 - sb) * theta) / (theta * c);
      B = (C - ((sb - 1.0) * sigma + sa * theta) / c) * 1.0 / theta_sq;
    }
  }
  M3d W;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) W.m[i][j] = A * Om.m[i][j] + B * Om2.m[i][j] + C * (i == j ? 1.0 : 0.0);
  r.t = matvec(W, mk3<double>(a[0], a[1], a[2]));
  return r;
}
}  // namespace orc
