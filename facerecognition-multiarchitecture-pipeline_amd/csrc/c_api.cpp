// Error plumbing + version of the C ABI (include/frmap_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/frmap_hip.h"

static thread_local char g_err[512] = "";

void frmap_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int frmap_abi_version(void) { return 4; }
extern "C" const char* frmap_last_error(void) { return g_err; }
