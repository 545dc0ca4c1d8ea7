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
extern "C" int ac_version(void) { return AC_ABI_VERSION; }

// ---- per-device state.  The library keeps no data between calls; what it caches is keyed by the device the
// calling thread has current, so one process may drive several GPUs (one at a time per call).
int ac_device_cus() {
  static int cus[64];                              // 0 = not queried yet
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev >= 0 && dev < 64 && cus[dev] > 0) return cus[dev];
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  if (dev >= 0 && dev < 64) cus[dev] = n;
  return n;
}

int ac_set_max_lds(const void* kernel, int bytes, unsigned long long* done_mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return ac_fail(AC_ELAUNCH, "hipGetDevice failed");
  const bool tracked = done_mask && dev >= 0 && dev < 64;   // no mask: the size varies per call, set it every time
  if (tracked && (__atomic_load_n(done_mask, __ATOMIC_RELAXED) >> dev) & 1ull) return AC_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess)
    return ac_fail(AC_ELAUNCH, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d): %s", bytes, hipGetErrorString(e));
  if (tracked) __atomic_fetch_or(done_mask, 1ull << dev, __ATOMIC_RELAXED);
  return AC_OK;
}

static int g_gemm_policy = AC_GEMM_AUTO;
int ac_gemm_policy() { return __atomic_load_n(&g_gemm_policy, __ATOMIC_RELAXED); }
extern "C" int ac_set_gemm_policy(int policy) {
  const bool ok = policy == AC_GEMM_AUTO || policy == AC_GEMM_256x128 || policy == AC_GEMM_256x256
#ifdef AC_EXPERIMENTAL
                  || (policy >= 100 && policy < 200)
#endif
      ;
  if (!ok) return ac_fail(AC_EINVAL, "ac_set_gemm_policy: unknown policy %d", policy);
  __atomic_store_n(&g_gemm_policy, policy, __ATOMIC_RELAXED);
  return AC_OK;
}
