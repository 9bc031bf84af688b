// common.hip — C-ABI plumbing shared by every entry point: version + thread-local error string.
#include <stdarg.h>
#include <string.h>

#include "gt_common.h"

static thread_local char g_err[512] = "";

void gt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int gt_version(void) { return 100; /* 0.1.0 */ }
extern "C" const char* gt_last_error(void) { return g_err; }
