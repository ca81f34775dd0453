// Library-level pieces of libtdvc_hip.so: ABI version and thread-local error text.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/tdvc_hip.h"

static thread_local char g_err[512] = "";

void tdvc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int tdvc_abi_version(void) { return 2; }   // 2: tdvc_dcn_desc.x_planar, fp32 conv entry points
extern "C" const char* tdvc_last_error(void) { return g_err; }
