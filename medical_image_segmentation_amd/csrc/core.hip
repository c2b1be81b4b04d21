// Library probe + thread-local error message.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void mis_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {
int mis_version(void) { return 100; /* 0.1.0 */ }
const char* mis_arch(void) { return "gfx950"; }
const char* mis_last_error(void) { return g_err; }
}
