// Error plumbing and version for the C ABI (include/quantool_amd.h).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void qt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* qt_last_error(void) { return g_err; }
extern "C" int qt_version(void) { return 100; }
