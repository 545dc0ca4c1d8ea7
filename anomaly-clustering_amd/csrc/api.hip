// Error plumbing and version of the C ABI (include/ac_hip.h).
#include "common.h"

static thread_local char g_err[512] = "";

int ac_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int ac_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ac_fail(AC_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return AC_OK;
}

extern "C" const char* ac_last_error(void) { return g_err; }
extern "C" int ac_version(void) { return 100; }   // round 1
