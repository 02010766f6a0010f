// core.hip — error reporting and ABI bookkeeping for libvidil_hip.so.
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

void vidil_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vidil_last_error(void) { return g_err; }
extern "C" int vidil_abi_version(void) { return 8; }
// keep in sync with include/vidil_hip.h (tests/test_abi.py parses the header)
extern "C" int vidil_num_entry_points(void) { return 25; }
