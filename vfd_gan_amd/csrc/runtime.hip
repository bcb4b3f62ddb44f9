// Error plumbing and version of the C ABI (include/vfdgan_hip.h).
#include "common.hpp"

static thread_local char g_err[512] = "";

void vfd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int vfd_abi_version(void) { return VFD_ABI_VERSION; }
extern "C" const char* vfd_last_error(void) { return g_err; }
