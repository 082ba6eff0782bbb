// Error-string plumbing of the C ABI (the pose algebra lives in pose_math.hpp, shared by host and device code).
#include <cstdarg>
#include "lsdhip_internal.hpp"

static thread_local std::string g_last_error;
void lsd_set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}
extern "C" const char* lsdhip_last_error(void) { return g_last_error.c_str(); }


// ---- host-side test hooks: the LM step arithmetic shared by the device kernel (pose_math.hpp), compiled for the host ----
extern "C" int lsdhip_host_se3f_step(const float increment[6], const float referenceToFrame[7], float out[7]) {
  if (!increment || !referenceToFrame || !out) return LSDHIP_E_ARG;
  lsdm::SE3fH T;
  T.q = {referenceToFrame[0], referenceToFrame[1], referenceToFrame[2], referenceToFrame[3]};
  T.t[0] = referenceToFrame[4]; T.t[1] = referenceToFrame[5]; T.t[2] = referenceToFrame[6];
  const lsdm::SE3fH r = lsdm::se3f_mul(lsdm::se3f_exp(increment), T);
  out[0] = r.q.w; out[1] = r.q.x; out[2] = r.q.y; out[3] = r.q.z; out[4] = r.t[0]; out[5] = r.t[1]; out[6] = r.t[2];
  return LSDHIP_OK;
}
extern "C" int lsdhip_host_ldlt6(const float A[36], const float b[6], float x[6]) {
  if (!A || !b || !x) return LSDHIP_E_ARG;
  lsdm::ldlt6_solve(A, b, x);
  return LSDHIP_OK;
}
