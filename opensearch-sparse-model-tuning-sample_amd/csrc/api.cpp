// Error plumbing and version of the C ABI (include/sparse_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/sparse_hip.h"

static thread_local char g_err[512] = "";

void sm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* sm_last_error(void) { return g_err; }
extern "C" int sm_abi_version(void) { return SM_ABI_VERSION; }
