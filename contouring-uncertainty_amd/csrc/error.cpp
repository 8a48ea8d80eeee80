// Thread-local last-error string + version/arch entry points of libcontour_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/contour_hip.h"

static thread_local char g_err[512] = "";

void cu_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* cu_last_error(void) { return g_err; }
extern "C" int cu_version(void) { return 100; }
extern "C" const char* cu_arch(void) { return "gfx950"; }
