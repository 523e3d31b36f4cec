// Host-side runtime glue of libsdhip.so: ABI version and the thread-local
// error channel.  No global mutable state besides the per-thread message, so
// entry points are re-entrant from autograd's backward threads.
#include "sdhip_common.h"

#define SDHIP_ABI_VERSION 1

static thread_local char g_err[512] = "";

void sdhip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int sdhip_abi_version(void) { return SDHIP_ABI_VERSION; }
extern "C" const char* sdhip_last_error(void) { return g_err; }
