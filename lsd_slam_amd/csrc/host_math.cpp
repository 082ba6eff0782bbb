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

