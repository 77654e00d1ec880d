// ABI version + thread-local error text for libadm_hip.so.
#include <stdarg.h>

#include "adm_common.h"

static thread_local char g_err[512] = "";

void adm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int adm_abi_version(void) { return ADM_ABI_VERSION; }
extern "C" const char* adm_last_error(void) { return g_err; }
