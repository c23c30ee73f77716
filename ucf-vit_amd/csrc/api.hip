// ABI bookkeeping: version and the thread-local error message.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void ucfvit_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int ucfvit_abi_version(void) { return UCFVIT_ABI_VERSION; }
extern "C" const char* ucfvit_last_error(void) { return g_err; }
