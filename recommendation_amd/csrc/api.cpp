// Host-only pieces of the C ABI: version string and per-thread error text.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/mi355rec.h"

static thread_local char g_err[512] = "";

void mi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mi_version(void) { return "mi355rec 0.1 (gfx950)"; }
extern "C" const char* mi_last_error(void) { return g_err; }
