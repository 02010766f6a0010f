// core.hip — error reporting and ABI bookkeeping for libvidil_hip.so.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void vidil_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Developer overrides ($VIDIL_GEMM4W, $VIDIL_GEMM4W128, $VIDIL_GEMM_CUS ...) are consulted on every GEMM launch, including
// launches being captured into the decode steps' graphs: each is looked up in the environment ONCE per process and
// remembered — unless $VIDIL_DEV_ENV is set when the first one is asked for: the tests and tools that flip these
// switches between launches set it, and then every query is a live getenv.
// Compute units of the current device, read once (256 on MI355X; also the fallback when no device answers, e.g. the host-only
// dispatch-name checks).  The tile-count thresholds of the GEMM dispatch are fractions of it: they were tuned as "workgroups
// per CU", and the persistent grids are sized from the same number.
// (per DEVICE, ADVICE r4: a process that drives more than one GPU — not how the path is deployed, one process per GPU, but not
//  forbidden either — must not size the second device's grids from the first one's CU count)
int vidil_cu_count() {
  static std::atomic<int> cache[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int v = cache[dev].load(std::memory_order_relaxed);
  if (v == 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8) v = 256;
    cache[dev].store(v, std::memory_order_relaxed);        // (racing threads store the same value)
  }
  return v;
}

// per-device one-time dynamic-LDS opt-in of a kernel (hipFuncSetAttribute applies to the device that is current when it is
// called).  `mask` holds one bit per device that HAS the attribute: the bit is set only after the call succeeded, so a failed
// opt-in is reported by this launch and retried by the next one, and it is an atomic word — two host threads driving
// different devices may both run the (idempotent) setup, neither loses the other's bit.
int vidil_lds_opt_in(std::atomic<unsigned long long>& mask, const void* kern, int bytes, const char* who) {
  int dev = 0;
  const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;      // (unknown device: set it every time)
  const unsigned long long bit = known ? 1ull << dev : 0ull;
  if (known && (mask.load(std::memory_order_acquire) & bit)) return VIDIL_OK;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    vidil_set_error("%s: hipFuncSetAttribute(%d B of dynamic LDS) failed: %s", who, bytes, hipGetErrorString(e));
    return VIDIL_ELAUNCH;
  }
  mask.fetch_or(bit, std::memory_order_release);
  return VIDIL_OK;
}

const char* vidil_dev_env(const char* name) {
  static const bool live = getenv("VIDIL_DEV_ENV") != nullptr;
  if (live) return getenv(name);
  struct Slot { const char* name; const char* val; bool known; };
  static Slot slots[16] = {};
  static int n = 0;
  for (int i = 0; i < n; ++i)
    if (slots[i].name == name || strcmp(slots[i].name, name) == 0) return slots[i].known ? slots[i].val : nullptr;
  const char* v = getenv(name);
  if (n < 16) {   // (launches come from one host thread per process: DESIGN.md §1 "Threading")
    slots[n].name = name;
    slots[n].val = v ? strdup(v) : nullptr;
    slots[n].known = v != nullptr;
    ++n;
    return slots[n - 1].val;
  }
  return v;
}

extern "C" const char* vidil_last_error(void) { return g_err; }
extern "C" int vidil_abi_version(void) { return 12; }
// keep in sync with include/vidil_hip.h (tests/test_abi.py parses the header)
extern "C" int vidil_num_entry_points(void) { return 29; }
