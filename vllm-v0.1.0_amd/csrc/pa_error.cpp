// Per-thread error message store behind pa_last_error().
#include <stdarg.h>
#include <stdio.h>

#include "pa_common.h"

namespace pa {

static thread_local char g_error[512] = {0};

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
  return code;
}

void clear_error() { g_error[0] = 0; }

}  // namespace pa

extern "C" const char* pa_last_error(void) { return pa::g_error; }

extern "C" int pa_abi_version(void) { return PA_ABI_VERSION; }
