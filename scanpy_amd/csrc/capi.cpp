// Library-wide C-ABI plumbing: thread-local error string, version, device probe.
#include "common.h"

namespace scamd {
static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace scamd

extern "C" int scamd_abi_version(void) { return SCAMD_ABI_VERSION; }
extern "C" const char* scamd_last_error(void) { return scamd::g_err; }
extern "C" int scamd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
