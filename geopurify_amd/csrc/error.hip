// Error reporting + version for libgeopurify_hip.so
#include <stdarg.h>

#include "gp_common.h"

static thread_local char g_err[512] = "";

extern "C" void gp_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *gp_last_error(void) { return g_err; }
extern "C" int gp_version(void) { return 100; }
